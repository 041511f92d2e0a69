// Fused Conv2d(+folded BN bias)+activation(+residual) as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces, on the reference's hot path:
//   * every ultralytics/yolov5 v6.0 `Conv` / `Bottleneck` / `C3` / `SPPF` / `Detect.m[i]` convolution that
//     /root/reference/networks/yolo.py:70 (`self.model(inputs)`) executes, and
//   * every convolution of the DeepSORT appearance net, /root/reference/networks/deepsort/deep/model.py:5-98
//     (row B5; ReLU / residual-before-ReLU epilogues).
//
// Layout: activations NHWC (channel-sliced views: a buffer may be a slice [co, co+C) of a wider
// concat buffer with channel stride cs, which is how Concat costs nothing), weights [Cout][K] with
// K = (r, s, c) so that a 16-byte chunk of the im2col row is one contiguous NHWC read.
// GEMM orientation: D[channel][pixel] += W[channel][k] * X[pixel][k]; the MFMA "A" operand is the
// weight tile, "B" the im2col pixel tile, so each lane ends up with 4 consecutive output channels
// of one pixel -> one 8-byte (bf16) / 16-byte (f32) NHWC store per 16x16 tile.
//
// bf16 path : v_mfma_f32_16x16x32_bf16, fp32 accumulate, one RNE rounding on store.
// fp32 path : v_mfma_f32_16x16x4_f32 (exact fmaf chain) -- the tight-parity mode (SURVEY.md 8d ladder).
#include <cstdlib>

#include "vc_common.h"

namespace vc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

union Chunk {
    uint4 u;
    bf16x8 h;
    float f[4];
};

// LDS rows hold KC chunks of 16 B (KC = 4: 64-byte rows, KC = 8: 128-byte rows).  ds_read_b128 is serviced in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table).  The chunk index is XOR-swizzled so
// that the 16 lanes of every group land on 16 distinct 16-byte slots of the 256-byte bank row:
//   KC = 4: chunk ^ perm[(row>>2)&3], perm = {0,2,3,1};   KC = 8: chunk ^ (row & 7).
template <int KC>
__device__ __forceinline__ int lds_slot(int row, int chunk) {
    if constexpr (KC == 4) return row * 4 + (chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3));
    else return row * 8 + (chunk ^ (row & 7));
}

__device__ __forceinline__ float act_apply(float v, int act, bool precise) {
    if (act == ACT_SILU) {
        return precise ? v / (1.0f + expf(-v)) : v * __frcp_rn(1.0f + __expf(-v));
    }
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {       // v_cvt_pk_bf16_f32 (RNE)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// BP x BC output tile (pixels x channels), WP x WC wavefronts, KC 16-byte chunks of K per tile row, PD = prefetch
// distance (global loads of K tile kt+PD are issued before the MFMAs of tile kt).
//
// Address generation is kept off the VALU as far as possible (the first version spent ~2000 VALU instructions per
// wave against 144 MFMAs): both operands are fetched with raw buffer loads (SGPR descriptor + one 32-bit offset per
// lane), padding / tile-edge / K-padding taps are turned into out-of-range offsets that the hardware answers with
// zeros, the per-row validity of all kh*kw taps is one 64-bit mask computed once, the tap offset advances
// incrementally, and all LDS addresses are loop invariant.
template <int BP, int BC, int WP, int WC, int KC, int PD, bool F32>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvP p) {
    constexpr int ES = F32 ? 4 : 2;           // element size
    constexpr int CH = 16 / ES;               // elements per 16-byte chunk
    constexpr int BK = KC * CH;               // K elements per tile
    constexpr int CPT = KC / 4;               // chunks per thread per staged row
    constexpr int XI = BP / 64;               // pixel rows staged per thread
    constexpr int WI = (BC + 63) / 64;        // weight rows staged per thread
    constexpr int WTP = BP / WP, WTC = BC / WC;
    constexpr int PT = WTP / 16, CT = WTC / 16;
    constexpr uint32_t OOB = 0x80000000u;     // beyond every descriptor's num_records -> the load returns 0
    static_assert(WP * WC == 4, "4 waves per workgroup");
    static_assert(BP % 64 == 0 && WTP % 16 == 0 && WTC % 16 == 0, "tile shape");
    static_assert(KC == 4 || KC == 8, "K tile");

    __shared__ __attribute__((aligned(16))) uint4 lds[2][(BP + BC) * KC];

    // XCD-aware tile order: the dispatcher places block b on XCD b % 8; give each XCD a contiguous
    // range of tiles so the channel tiles that share one pixel tile hit the same private L2.
    const int nblk = gridDim.x;
    const int tiles_c = (p.Cout + BC - 1) / BC;
    int tile;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / tiles_c) * BP;
    const int n0 = (tile % tiles_c) * BC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kc0 = (tid & 3) * CPT, lrow = tid >> 2;
    const int HoWo = p.Ho * p.Wo;
    const int ntap = p.kh * p.kw;

    // descriptors: whole input buffer / whole packed weight buffer (sizes < 2 GiB, checked by the launcher)
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    // per staged pixel row: byte offset of its (iy0, ix0) corner and the validity mask of the kh*kw taps
    uint32_t xoff[XI];
    unsigned long long xmask[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = m0 + lrow + 64 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;
        xoff[i] = (uint32_t)((((b * p.H + iy0) * p.W + ix0) * p.in_cs + p.in_co) * ES);
        unsigned long long mk = 0;
        if (ok)
            for (int t = 0; t < ntap; ++t) {
                const int r = t / p.kw, s = t - r * p.kw;
                if ((unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + s) < (unsigned)p.W) mk |= 1ull << t;
            }
        xmask[i] = mk;
    }
    uint32_t woff[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) woff[i] = (uint32_t)(((n0 + lrow + 64 * i) * p.Kw + kc0 * CH) * ES);

    // (tap, c) of each of this thread's chunks and the tap's byte offset (r*W + s)*in_cs*ES, advanced by BK per K step
    int kc_c[CPT], kc_t[CPT], kc_s[CPT];
    uint32_t kc_off[CPT];
    const uint32_t tap_x = (uint32_t)(p.in_cs * ES), tap_y = (uint32_t)((p.W - p.kw + 1) * p.in_cs * ES);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int k = (kc0 + j) * CH;
        const int tap = k / p.Cin;
        const int r = tap / p.kw;
        kc_c[j] = k - tap * p.Cin;
        kc_t[j] = tap;
        kc_s[j] = tap - r * p.kw;
        kc_off[j] = (uint32_t)((r * p.W + kc_s[j]) * p.in_cs * ES);
    }
    const int nk = p.Kp / BK;

    // loop-invariant LDS slots
    int xslot[XI][CPT], wslot[WI][CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
#pragma unroll
        for (int i = 0; i < XI; ++i) xslot[i][j] = lds_slot<KC>(lrow + 64 * i, kc0 + j);
#pragma unroll
        for (int i = 0; i < WI; ++i) wslot[i][j] = BP * KC + lds_slot<KC>(lrow + 64 * i, kc0 + j);
    }

    u32x4 xr[PD][XI][CPT], wr[PD][WI][CPT];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
#pragma unroll
        for (int i = 0; i < XI; ++i)
#pragma unroll
            for (int j = 0; j < CPT; ++j) xr[d][i][j] = (u32x4){0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < WI; ++i)
#pragma unroll
            for (int j = 0; j < CPT; ++j) wr[d][i][j] = (u32x4){0, 0, 0, 0};
    }
    // global -> registers for K tile `kt` into register set `set`
#define VC_GLOAD(kt, set)                                                                                                \
    {                                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < CPT; ++j) {                                                                \
            const uint32_t tc = kc_off[j] + (uint32_t)(kc_c[j] * ES);                                                    \
            const int t = kc_t[j];                                                                                       \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                             \
                const uint32_t o = ((xmask[i] >> t) & 1ull) ? xoff[i] + tc : OOB;                                        \
                xr[set][i][j] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, (int)o, 0, 0);                               \
            }                                                                                                            \
            _Pragma("unroll") for (int i = 0; i < WI; ++i) {                                                             \
                if (BC % 64 == 0 || lrow + 64 * i < BC)                                                                  \
                    wr[set][i][j] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, (int)(woff[i] + (uint32_t)(((kt) * BK + j * CH) * ES)), 0, 0); \
            }                                                                                                            \
            int cc = kc_c[j] + BK;                                                                                       \
            while (cc >= p.Cin) {                                                                                        \
                cc -= p.Cin;                                                                                             \
                ++kc_t[j];                                                                                               \
                if (++kc_s[j] == p.kw) { kc_s[j] = 0; kc_off[j] += tap_y; } else { kc_off[j] += tap_x; }                 \
            }                                                                                                            \
            kc_c[j] = cc;                                                                                                \
        }                                                                                                                \
    }
#define VC_LSTORE(buf, set)                                                                                              \
    {                                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < CPT; ++j) {                                                                \
            _Pragma("unroll") for (int i = 0; i < XI; ++i) lds[buf][xslot[i][j]] = __builtin_bit_cast(uint4, xr[set][i][j]); \
            _Pragma("unroll") for (int i = 0; i < WI; ++i) {                                                             \
                if (BC % 64 == 0 || lrow + 64 * i < BC) lds[buf][wslot[i][j]] = __builtin_bit_cast(uint4, wr[set][i][j]); \
            }                                                                                                            \
        }                                                                                                                \
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;
    int xfrag[CPT][PT], wfrag[CPT][CT];
#pragma unroll
    for (int h = 0; h < CPT; ++h) {
#pragma unroll
        for (int i = 0; i < PT; ++i) xfrag[h][i] = lds_slot<KC>(wp * WTP + i * 16 + frow, h * 4 + fch);
#pragma unroll
        for (int i = 0; i < CT; ++i) wfrag[h][i] = BP * KC + lds_slot<KC>(wc * WTC + i * 16 + frow, h * 4 + fch);
    }

    // prologue: tiles 0 .. PD-1 in flight, tile 0 staged
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nk) VC_GLOAD(d, d);
    VC_LSTORE(0, 0);
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {                // register set indices are compile-time constants
            const int kt = kt0 + d;
            if (kt < nk) {
                const int buf = kt & 1;
                if (kt + PD < nk) VC_GLOAD(kt + PD, d);       // set d held tile kt, which is already in LDS
#pragma unroll
                for (int h = 0; h < CPT; ++h) {               // one MFMA K-step (4 chunks) per half of the tile row
                    Chunk xa[PT], wa[CT];
#pragma unroll
                    for (int i = 0; i < PT; ++i) xa[i].u = lds[buf][xfrag[h][i]];
#pragma unroll
                    for (int i = 0; i < CT; ++i) wa[i].u = lds[buf][wfrag[h][i]];
#pragma unroll
                    for (int a = 0; a < CT; ++a)
#pragma unroll
                        for (int b = 0; b < PT; ++b) {
                            if constexpr (F32) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[a].f[j], xa[b].f[j], acc[a][b], 0, 0, 0);
                            } else {
                                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[a].h, xa[b].h, acc[a][b], 0, 0, 0);
                            }
                        }
                }
                if (kt + 1 < nk) VC_LSTORE(buf ^ 1, (d + 1) % PD);
                __syncthreads();
            }
        }
    }

    // epilogue: D[channel = (lane>>4)*4 + reg][pixel = lane&15]
    const int nbase = n0 + wc * WTC + fch * 4;
#pragma unroll
    for (int b = 0; b < PT; ++b) {
        const int m = m0 + wp * WTP + b * 16 + frow;
        if (m >= p.M) continue;
        const size_t orow = (size_t)m * p.out_cs + p.out_co, rrow = (size_t)m * p.res_cs + p.res_co;
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = nbase + a * 16;
            if (n >= p.Cout) continue;
            const float4 bv = *(const float4*)(p.bias + n);
            float v[4] = {acc[a][b][0] + bv.x, acc[a][b][1] + bv.y, acc[a][b][2] + bv.z, acc[a][b][3] + bv.w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            const int nvalid = p.Cout - n >= 4 ? 4 : p.Cout - n;
            if (p.res_mode != RES_NONE) {
                if constexpr (F32) {
                    const float4 t = *(const float4*)((const float*)p.res + rrow + n);
                    rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
                } else {
                    const uint2 t = *(const uint2*)((const uint16_t*)p.res + rrow + n);
                    rv[0] = __uint_as_float(t.x << 16); rv[1] = __uint_as_float(t.x & 0xffff0000u);
                    rv[2] = __uint_as_float(t.y << 16); rv[3] = __uint_as_float(t.y & 0xffff0000u);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j];
                if (p.res_mode == RES_BEFORE_ACT) t += rv[j];
                t = act_apply(t, p.act, F32);
                if (p.res_mode == RES_AFTER_ACT) t += rv[j];
                v[j] = t;
            }
            if (F32 || p.out_f32) {
                float* o = (float*)p.out + orow + n;
                if (nvalid == 4) *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j < nvalid) o[j] = v[j];
                }
            } else {
                uint16_t* o = (uint16_t*)p.out + orow + n;
                if (nvalid == 4) {
                    *(uint2*)o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j < nvalid) o[j] = f32_to_bf16(v[j]);
                }
            }
        }
    }
}

#undef VC_GLOAD
#undef VC_LSTORE

int conv_k_tile(int prec) { return prec == PREC_F32 ? 32 : 64; }    // weights are padded to the widest K tile (KC = 8)

double conv_flops(const ConvP& p) { return 2.0 * (double)p.M * (double)p.Cout * (double)p.K; }

template <int BP, int BC, int WP, int WC, int KC, int PD>
static int launch_cfg(ConvP p, hipStream_t s) {
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    const int bk = KC * (p.prec == PREC_F32 ? 4 : 8);
    p.Kw = p.Kp;                              // weight row stride as packed
    p.Kp = (p.K + bk - 1) / bk * bk;          // K-loop extent: only the tiles that hold real taps
    if (p.prec == PREC_F32)
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KC, 1, true>), dim3(tiles), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, KC, PD, false>), dim3(tiles), dim3(256), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

template <int BP, int BC, int WP, int WC>
static int launch_kc(const ConvP& p, hipStream_t s) {
    // wide K tile (half the barriers per MFMA) unless its zero padding would waste more than 1/8 of the K loop
    const int ch = p.prec == PREC_F32 ? 4 : 8;
    const int k8 = (p.K + 8 * ch - 1) / (8 * ch) * (8 * ch);
    static const int force_kc = getenv("VC_CONV_KC") ? atoi(getenv("VC_CONV_KC")) : 0;      // tuning knobs (bench only)
    static const int pd = getenv("VC_CONV_PD") ? atoi(getenv("VC_CONV_PD")) : 1;
    const bool wide = force_kc ? force_kc == 8 : false;
    (void)k8;
    if (wide) return pd >= 2 ? launch_cfg<BP, BC, WP, WC, 8, 2>(p, s) : launch_cfg<BP, BC, WP, WC, 8, 1>(p, s);
    if (pd >= 3) return launch_cfg<BP, BC, WP, WC, 4, 3>(p, s);
    if (pd == 2) return launch_cfg<BP, BC, WP, WC, 4, 2>(p, s);
    return launch_cfg<BP, BC, WP, WC, 4, 1>(p, s);
}

int launch_conv(const ConvP& p, hipStream_t s) {
    const int ch = p.prec == PREC_F32 ? 4 : 8;
    VC_CHECK(p.Cin % ch == 0 && p.in_cs % ch == 0 && p.in_co % ch == 0, VC_ERR_ARG,
             "conv: input channels/stride/offset (%d,%d,%d) must be multiples of %d", p.Cin, p.in_cs, p.in_co, ch);
    VC_CHECK(p.out_cs % 4 == 0 && p.out_co % 4 == 0, VC_ERR_ARG, "conv: output stride/offset must be multiples of 4");
    VC_CHECK(p.res_mode == RES_NONE || (p.res_cs % 4 == 0 && p.res_co % 4 == 0), VC_ERR_ARG, "conv: residual alignment");
    VC_CHECK(p.Kp % conv_k_tile(p.prec) == 0 && p.Kp >= p.K, VC_ERR_ARG, "conv: bad K padding %d/%d", p.K, p.Kp);
    VC_CHECK(p.M > 0 && p.Cout > 0, VC_ERR_ARG, "conv: empty problem");
    VC_CHECK((size_t)p.B * p.H * p.W * p.in_cs * elem_size(p.prec) < (1ull << 31), VC_ERR_CAPACITY, "conv: input tensor exceeds the 2 GiB buffer descriptor");
    VC_CHECK((size_t)((p.Cout + 127) / 128 * 128) * p.Kp * elem_size(p.prec) < (1ull << 31), VC_ERR_CAPACITY, "conv: weights exceed 2 GiB");
    VC_CHECK(p.kh * p.kw <= 40, VC_ERR_ARG, "conv: at most 40 taps (validity mask is 64 bits incl. K padding)");
    // tile choice: narrow layers get tall pixel tiles; late (small-M) layers get 64x64 so the grid still covers 256 CUs
    if (p.Cout <= 32) return launch_kc<256, 32, 4, 1>(p, s);
    if (p.Cout <= 64) return launch_kc<128, 64, 2, 2>(p, s);
    const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (t128 >= 512) return launch_kc<128, 128, 2, 2>(p, s);
    return launch_kc<64, 64, 2, 2>(p, s);
}

}  // namespace vc
