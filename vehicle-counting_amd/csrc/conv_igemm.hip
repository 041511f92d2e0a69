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
#include "vc_common.h"

namespace vc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

union Chunk {
    uint4 u;
    bf16x8 h;
    float f[4];
};

// 64-byte LDS rows (4 chunks of 16 B).  ds_read_b128 is serviced in the lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table); XOR-ing the chunk with
// perm[(row>>2)&3], perm = {0,2,3,1}, puts the 16 lanes of every group on 16 distinct 16-byte slots.
__device__ __forceinline__ int lds_slot(int row, int chunk) {
    return row * 4 + (chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3));
}

__device__ __forceinline__ float act_apply(float v, int act, bool precise) {
    if (act == ACT_SILU) {
        return precise ? v / (1.0f + expf(-v)) : v * __frcp_rn(1.0f + __expf(-v));
    }
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

template <int BP, int BC, int WP, int WC, bool F32>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvP p) {
    constexpr int ES = F32 ? 4 : 2;           // element size
    constexpr int CH = 16 / ES;               // elements per 16-byte chunk
    constexpr int BK = 4 * CH;                // K elements per tile
    constexpr int XI = BP / 64;               // pixel rows staged per thread
    constexpr int WI = (BC + 63) / 64;        // weight rows staged per thread
    constexpr int WTP = BP / WP, WTC = BC / WC;
    constexpr int PT = WTP / 16, CT = WTC / 16;
    static_assert(WP * WC == 4, "4 waves per workgroup");
    static_assert(BP % 64 == 0 && WTP % 16 == 0 && WTC % 16 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) uint4 lds[2][(BP + BC) * 4];

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
    const int kc = tid & 3, lrow = tid >> 2;
    const int HoWo = p.Ho * p.Wo;

    int xiy0[XI], xix0[XI];
    const char* xbase[XI];
    bool xok[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = m0 + lrow + 64 * i;
        xok[i] = m < p.M;
        const int mm = xok[i] ? m : 0;
        const int b = mm / HoWo;
        const int rem = mm - b * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        xiy0[i] = oy * p.sh - p.ph;
        xix0[i] = ox * p.sw - p.pw;
        xbase[i] = (const char*)p.in + ((size_t)b * p.H * p.W * p.in_cs + p.in_co) * ES;
    }
    const char* wptr[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i)
        wptr[i] = (const char*)p.w + ((size_t)(n0 + lrow + 64 * i) * p.Kp + kc * CH) * ES;

    // (r, s, c) of this thread's chunk, advanced by BK every K step
    int c, r, s;
    {
        const int k = kc * CH;
        const int tap = k / p.Cin;
        c = k - tap * p.Cin;
        r = tap / p.kw;
        s = tap - r * p.kw;
    }
    const int nk = p.Kp / BK;

    uint4 xr[XI], wr[WI];
#pragma unroll
    for (int i = 0; i < XI; ++i) xr[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WI; ++i) wr[i] = make_uint4(0, 0, 0, 0);
    // global -> registers for K tile `kt` (issued one tile ahead of the MFMAs that consume it)
#define VC_GLOAD(kt)                                                                                          \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                      \
            const int iy = xiy0[i] + r, ix = xix0[i] + s;                                                     \
            const bool ok = xok[i] && r < p.kh && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W; \
            uint4 v = make_uint4(0, 0, 0, 0);                                                                 \
            if (ok) v = *(const uint4*)(xbase[i] + ((size_t)(iy * p.W + ix) * p.in_cs + c) * ES);             \
            xr[i] = v;                                                                                        \
        }                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) {                                                      \
            if (BC % 64 == 0 || lrow + 64 * i < BC) wr[i] = *(const uint4*)(wptr[i] + (size_t)(kt) * BK * ES); \
        }                                                                                                     \
        c += BK;                                                                                              \
        while (c >= p.Cin) {                                                                                  \
            c -= p.Cin;                                                                                       \
            if (++s == p.kw) { s = 0; ++r; }                                                                  \
        }                                                                                                     \
    }
#define VC_LSTORE(buf)                                                                                        \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) lds[buf][lds_slot(lrow + 64 * i, kc)] = xr[i];         \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) {                                                      \
            if (BC % 64 == 0 || lrow + 64 * i < BC) lds[buf][BP * 4 + lds_slot(lrow + 64 * i, kc)] = wr[i];  \
        }                                                                                                     \
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;

    VC_GLOAD(0);
    VC_LSTORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) VC_GLOAD(kt + 1);
        Chunk xa[PT], wa[CT];
#pragma unroll
        for (int i = 0; i < PT; ++i) xa[i].u = lds[buf][lds_slot(wp * WTP + i * 16 + frow, fch)];
#pragma unroll
        for (int i = 0; i < CT; ++i) wa[i].u = lds[buf][BP * 4 + lds_slot(wc * WTC + i * 16 + frow, fch)];
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
        if (kt + 1 < nk) VC_LSTORE(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[channel = (lane>>4)*4 + reg][pixel = lane&15]
#pragma unroll
    for (int b = 0; b < PT; ++b) {
        const int m = m0 + wp * WTP + b * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = n0 + wc * WTC + a * 16 + fch * 4;
            if (n >= p.Cout) continue;
            const float4 bv = *(const float4*)(p.bias + n);
            float v[4] = {acc[a][b][0] + bv.x, acc[a][b][1] + bv.y, acc[a][b][2] + bv.z, acc[a][b][3] + bv.w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            const int nvalid = p.Cout - n >= 4 ? 4 : p.Cout - n;
            if (p.res_mode != RES_NONE) {
                const size_t ro = (size_t)m * p.res_cs + p.res_co + n;
                if constexpr (F32) {
                    const float4 t = *(const float4*)((const float*)p.res + ro);
                    rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w;
                } else {
                    const uint2 t = *(const uint2*)((const uint16_t*)p.res + ro);
                    rv[0] = bf16_to_f32((uint16_t)(t.x & 0xffff)); rv[1] = bf16_to_f32((uint16_t)(t.x >> 16));
                    rv[2] = bf16_to_f32((uint16_t)(t.y & 0xffff)); rv[3] = bf16_to_f32((uint16_t)(t.y >> 16));
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
            const size_t oo = (size_t)m * p.out_cs + p.out_co + n;
            if (F32 || p.out_f32) {
                float* o = (float*)p.out + oo;
                if (nvalid == 4) *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                else for (int j = 0; j < nvalid; ++j) o[j] = v[j];
            } else {
                uint16_t* o = (uint16_t*)p.out + oo;
                if (nvalid == 4) {
                    uint2 t;
                    t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *(uint2*)o = t;
                } else for (int j = 0; j < nvalid; ++j) o[j] = f32_to_bf16(v[j]);
            }
        }
    }
}

#undef VC_GLOAD
#undef VC_LSTORE

int conv_k_tile(int prec) { return prec == PREC_F32 ? 16 : 32; }

double conv_flops(const ConvP& p) { return 2.0 * (double)p.M * (double)p.Cout * (double)p.K; }

template <int BP, int BC, int WP, int WC>
static int launch_cfg(const ConvP& p, hipStream_t s) {
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    if (p.prec == PREC_F32)
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, true>), dim3(tiles), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BP, BC, WP, WC, false>), dim3(tiles), dim3(256), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_conv(const ConvP& p, hipStream_t s) {
    const int ch = p.prec == PREC_F32 ? 4 : 8;
    VC_CHECK(p.Cin % ch == 0 && p.in_cs % ch == 0 && p.in_co % ch == 0, VC_ERR_ARG,
             "conv: input channels/stride/offset (%d,%d,%d) must be multiples of %d", p.Cin, p.in_cs, p.in_co, ch);
    VC_CHECK(p.out_cs % 4 == 0 && p.out_co % 4 == 0, VC_ERR_ARG, "conv: output stride/offset must be multiples of 4");
    VC_CHECK(p.res_mode == RES_NONE || (p.res_cs % 4 == 0 && p.res_co % 4 == 0), VC_ERR_ARG, "conv: residual alignment");
    VC_CHECK(p.Kp % conv_k_tile(p.prec) == 0 && p.Kp >= p.K, VC_ERR_ARG, "conv: bad K padding %d/%d", p.K, p.Kp);
    VC_CHECK(p.M > 0 && p.Cout > 0, VC_ERR_ARG, "conv: empty problem");
    // tile choice: narrow layers get tall pixel tiles; late (small-M) layers get 64x64 so the grid still covers 256 CUs
    if (p.Cout <= 32) return launch_cfg<256, 32, 4, 1>(p, s);
    if (p.Cout <= 64) return launch_cfg<128, 64, 2, 2>(p, s);
    const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (t128 >= 512) return launch_cfg<128, 128, 2, 2>(p, s);
    return launch_cfg<64, 64, 2, 2>(p, s);
}

}  // namespace vc
