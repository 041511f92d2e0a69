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
#include <algorithm>
#include <map>
#include <mutex>
#include <cstdlib>

#include "vc_common.h"
#include "conv_device.h"

namespace vc {

// BP x BC output tile (pixels x channels) per workgroup of WP x WC wavefronts; KC 16-byte chunks of K per tile row.
//
// Staging: both operands go global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA: no VGPR round trip, no ds_write).
// One wave-instruction writes 64 lanes x 16 B = 1 KiB lane-linearly, i.e. 64/KC consecutive tile rows, so the
// bank-conflict swizzle is applied on the SOURCE side: lane (row, c) fetches logical chunk c ^ swz(row)
// (cdna_hip_programming.md rule 21).  Padding / tile-edge / K-padding taps become out-of-range buffer offsets that the
// hardware answers with zeros (verified by the padded test cases); the per-row validity of all kh*kw taps is one 64-bit
// mask computed once, the tap offset advances incrementally, every LDS address is loop invariant: the K loop is
// {KC/4 x (PT+CT ds_read_b128, PT*CT MFMA)} + (XI+WI) DMA issues + one barrier.
// UP: the first p.up_C input channels of a pointwise conv come from a tensor of half the height and width, nearest-upsampled on the fly
// (nn.Upsample(None, 2, 'nearest') + Concat in front of C3.cv1 | cv2, YOLOv5 layers 11-13 and 15-17): the staged row of output pixel
// (b, y, x) reads pixel (b, y / 2, x / 2) of p.in_up for K tiles below up_C and the concat buffer for the rest, so the upsampled map
// is neither written nor read.  Same values in the same K order: bit-identical to upsample2x_kernel + this kernel.
// SK (round 6, bf16): deterministic split-K for launches with too few output tiles to fill the chip (batch 1 .. 8: 16 workgroups walking
// 72 K steps).  A work item is (output tile, K split s of p.ksplit): it multiplies K tiles [s nk / KS, (s + 1) nk / KS), stores its fp32
// accumulators to p.sk_ws and takes a ticket of its tile; the workgroup that draws the LAST ticket adds the KS partial sums in split
// order 0 .. KS - 1 (whichever workgroup arrives last: the same sum) and runs the epilogue.  No workgroup waits for another one.
// OCC (round 6, configurations 64 - 66): workgroups per CU the register allocation is bounded for.  Two 8-wave workgroups on one CU share no
// barrier: while one is in its epilogue (SiLU + stores, no loads issued, no MFMA) the other is in its K loop -- the phases a single 16-wave
// workgroup runs one after the other overlap across the pair.  (HIP: the second launch-bound is WAVES per SIMD, hence OCC * waves / 4.)
template <int BP, int BC, int WP, int WC, int KC, int NS, int PR, bool UP = false, bool SK = false, int OCC = 1>       // PR: PREC_BF16, PREC_F32 or PREC_FP8
__global__ __launch_bounds__(WP * WC * 64, OCC == 1 ? 1 : OCC * WP * WC / 4) void conv_igemm_kernel(const ConvP p_arg) {
    ConvP p = p_arg;
    if (p_arg.m_dev) {                        // device-side problem size (uniform): fewer pixels, fewer tiles
        const int mm = min(p_arg.M, *p_arg.m_dev);
        p.M = mm;
        p.ntiles = ((mm + BP - 1) / BP) * ((p_arg.Cout + BC - 1) / BC);
    }
    constexpr bool F32 = PR == PREC_F32, FP8 = PR == PREC_FP8;
    constexpr int ES = F32 ? 4 : FP8 ? 1 : 2; // element size
    static_assert(!FP8 || KC == 8, "fp8: one 128-byte LDS row = one K = 128 MX-scaled MFMA step");
    constexpr int CH = 16 / ES;               // elements per 16-byte chunk
    constexpr int BK = KC * CH;               // K elements per tile
    constexpr int RPI = 64 / KC;              // tile rows covered by one wave-instruction
    constexpr int NW = WP * WC;               // waves per workgroup: 4, or 8 / 16 for the 256-row tiles (configs 40-43)
    constexpr int PASS = NW * RPI;            // tile rows covered by one instruction of all waves
    constexpr int XI = BP / PASS;             // DMA instructions per thread for the pixel tile
    constexpr int WI = (BC + PASS - 1) / PASS;
    constexpr int WTP = BP / WP, WTC = BC / WC;
    constexpr int PT = WTP / 16, CT = WTC / 16;
    constexpr uint32_t OOB = 0x80000000u;     // beyond every descriptor's num_records -> the load returns 0
    static_assert(NW == 4 || NW == 8 || NW == 16, "waves per workgroup");
    static_assert(BP % PASS == 0 && BC % RPI == 0 && WTP % 16 == 0 && WTC % 16 == 0, "tile shape");
    constexpr int ROWS = BP + WI * PASS;       // rows of one stage: every wave issues the same XI + WI DMA instructions per tile
    constexpr int PER = XI + WI;
    static_assert(KC == 4 || KC == 8, "K tile");
    static_assert(NS >= 2 && (NS - 2) * PER <= 63, "ring depth: the counted s_waitcnt must fit vmcnt");

    __shared__ __attribute__((aligned(16))) uint4 lds[NS][ROWS * KC];
    if (p.ablate == 6) return;                                 // launch floor (diagnostics)
#define VC_TS(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    VC_TS(0);

    // Persistent workgroups: the grid is min(tiles, resident workgroups); workgroup b walks the tiles of the virtual blocks
    // b, b + G, b + 2G, ...  The K-tile ring runs THROUGH the tile boundaries -- while a tile's epilogue (bias, SiLU, stores)
    // executes, the first NS-1 K tiles of the next output tile are already in flight -- so the per-tile prologue arithmetic,
    // the first-tile latency and the store tail overlap with useful work instead of being paid serially by a fresh
    // workgroup per tile.
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (G is a multiple of 8, so all virtual blocks of a
    // workgroup share its XCD); each XCD gets a contiguous range of tiles so the channel tiles that share one pixel tile
    // hit the same private L2.
    const int ntiles = p.ntiles, G = gridDim.x;
    const int KS = SK ? p.ksplit : 1;         // (uniform)
    const int nitems = ntiles * KS;
    const int tiles_c = (p.Cout + BC - 1) / BC;
    const int tq = ntiles >> 3, tr = ntiles & 7;
#define VC_TILE_OF(v) ((((v) & 7) < tr ? ((v) & 7) * (tq + 1) : tr * (tq + 1) + (((v) & 7) - tr) * tq) + ((v) >> 3))

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int prow = wave * RPI + lane / KC;                 // this thread's row inside a PASS-row slab
    const int cpos = lane % KC;                              // physical chunk slot it fills
    const int kc0 = KC == 4 ? (cpos ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3)) : (cpos ^ (prow & 7));   // logical chunk
    const int HoWo = p.Ho * p.Wo;

    // descriptors: whole input buffer / whole packed weight buffer (sizes < 2 GiB, checked by the launcher)
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    const bool pointwise = p.kh == 1 && p.kw == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0;    // uniform
    const uint32_t tap_x = (uint32_t)(p.in_cs * ES), tap_y = (uint32_t)((p.W - p.kw + 1) * p.in_cs * ES);
    // When Cin is a multiple of the K tile (every layer but the stems) a K tile lies inside ONE tap, the same for every
    // lane: the tap walk is then scalar state (SALU) instead of a divergent per-lane loop.
    const bool ut = (p.Cin % BK) == 0;

    // ---- state of the tile being STAGED (it runs up to NS-1 K tiles ahead of the tile being multiplied) ----------------
    // per staged pixel row: byte offset of its (iy0, ix0) corner and the validity mask of the kh*kw taps; quotients come from
    // an exact float reciprocal with a +-1 fix-up, the mask from row/column ranges (no run-time integer divisions)
    uint32_t xoff[XI], xoffl[XI], woff[WI];
    uint32_t xoffu[UP ? XI : 1];              // UP: this lane's chunk in the half-size source
    const __amdgpu_buffer_rsrc_t usrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(UP ? p.in_up : p.in), 0, UP ? (int)((size_t)p.B * (p.H / 2) * (p.W / 2) * p.up_cs * ES) : 0, 0x00020000);
    unsigned long long xmask[XI];
    int kc_c = 0, kc_t = 0, kc_s = 0, u_tap = 0, u_s = 0, u_c = 0;
    uint32_t kc_off = 0, u_tapoff = 0;
    int s_v = blockIdx.x, s_kt = 0, s_kt1 = 0, s_issued = 0;
    const int nk = p.Kp / BK;
#define VC_TILE_STATE(v)                                                                                                  \
    if ((v) < nitems) {                                                                                                   \
        const int tile_ = VC_TILE_OF(SK ? (v) / KS : (v));                                                                \
        const int sp_ = SK ? (v) % KS : 0;                                                                                \
        s_kt = SK ? sp_ * nk / KS : 0;                                                                                    \
        s_kt1 = SK ? (sp_ + 1) * nk / KS : nk;                                                                            \
        const int kst_ = s_kt * BK;           /* first K element of this item */                                           \
        const int sm0 = (tile_ / tiles_c) * BP, sn0 = (tile_ % tiles_c) * BC;                                              \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                                  \
            const int m = sm0 + prow + PASS * i;                                                                          \
            const bool ok = m < p.M;                                                                                      \
            const int mm = ok ? m : 0;                                                                                    \
            if (pointwise) {                  /* 1x1 / stride 1: output pixel m reads input pixel m, one tap */            \
                xoff[i] = (uint32_t)((mm * p.in_cs + p.in_co) * ES);                                                      \
                xmask[i] = ok ? 1ull : 0ull;                                                                              \
                if constexpr (UP) {                                                                                       \
                    int b = (int)((float)mm * inv_howo);                                                                  \
                    b -= (b * HoWo > mm) ? 1 : 0;                                                                         \
                    b += ((b + 1) * HoWo <= mm) ? 1 : 0;                                                                  \
                    const int rem = mm - b * HoWo;                                                                        \
                    int oy = (int)((float)rem * inv_wo);                                                                  \
                    oy -= (oy * p.Wo > rem) ? 1 : 0;                                                                      \
                    oy += ((oy + 1) * p.Wo <= rem) ? 1 : 0;                                                               \
                    const int ox = rem - oy * p.Wo;                                                                       \
                    xoffu[i] = (uint32_t)((((b * (p.H / 2) + (oy >> 1)) * (p.W / 2) + (ox >> 1)) * p.up_cs + p.up_co + kc0 * CH) * ES); \
                }                                                                                                         \
            } else {                                                                                                      \
                int b = (int)((float)mm * inv_howo);    /* mm < 2^24: the float product is within 1 of the quotient */     \
                b -= (b * HoWo > mm) ? 1 : 0;                                                                             \
                b += ((b + 1) * HoWo <= mm) ? 1 : 0;                                                                      \
                const int rem = mm - b * HoWo;                                                                            \
                int oy = (int)((float)rem * inv_wo);                                                                      \
                oy -= (oy * p.Wo > rem) ? 1 : 0;                                                                          \
                oy += ((oy + 1) * p.Wo <= rem) ? 1 : 0;                                                                   \
                const int ox = rem - oy * p.Wo;                                                                           \
                const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;                                                 \
                xoff[i] = (uint32_t)((((b * p.H + iy0) * p.W + ix0) * p.in_cs + p.in_co) * ES);                           \
                const int r_lo = max(0, -iy0), r_hi = min(p.kh, p.H - iy0);                                               \
                const int s_lo = max(0, -ix0), s_hi = min(p.kw, p.W - ix0);                                               \
                unsigned long long mk = 0;                                                                                \
                if (ok && s_hi > s_lo) {                                                                                  \
                    const unsigned long long rowbits = ((1ull << (s_hi - s_lo)) - 1ull) << s_lo;                          \
                    for (int r = r_lo; r < r_hi; ++r) mk |= rowbits << (r * p.kw);                                        \
                }                                                                                                         \
                xmask[i] = mk;                                                                                            \
            }                                                                                                             \
            xoffl[i] = xoff[i] + (uint32_t)(kc0 * CH * ES);                                                               \
        }                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) woff[i] = (uint32_t)(((sn0 + prow + PASS * i) * p.Kw + kc0 * CH) * ES); \
        {   /* (tap, c) of this thread's chunk and the tap's byte offset (r*W + s)*in_cs*ES, advanced by BK per K step */   \
            const int k = kst_ + kc0 * CH;                                                                                \
            const int tap = k / p.Cin;                                                                                    \
            const int r = tap / p.kw;                                                                                     \
            kc_c = k - tap * p.Cin;                                                                                       \
            kc_t = tap;                                                                                                   \
            kc_s = tap - r * p.kw;                                                                                        \
            kc_off = (uint32_t)((r * p.W + kc_s) * p.in_cs * ES);                                                         \
        }                                                                                                                 \
        u_tap = 0; u_s = 0; u_c = 0; u_tapoff = 0;                                                                        \
        if (SK && kst_ > 0) {                 /* (uniform) the scalar tap walk starts inside the K range */                \
            const int tap = kst_ / p.Cin, r = tap / p.kw;                                                                 \
            u_c = kst_ - tap * p.Cin;                                                                                     \
            u_tap = min(tap, 63);                                                                                         \
            u_s = tap - r * p.kw;                                                                                         \
            u_tapoff = (uint32_t)((r * p.W + u_s) * p.in_cs * ES);                                                        \
        }                                                                                                                 \
    } else {                                  /* no tile left: everything staged from here on is out of range (zeros) */   \
        s_kt = 0; s_kt1 = nk;                                                                                             \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) { xmask[i] = 0ull; xoff[i] = 0; xoffl[i] = 0; }                    \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) woff[i] = OOB;                                                     \
        kc_c = 0; kc_t = 0; kc_s = 0; kc_off = 0; u_tap = 0; u_s = 0; u_c = 0; u_tapoff = 0;                              \
    }

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // stage K tile s_kt of the staged tile into ring slot `buf`, then advance (to the next tile of this workgroup at the end)
#define VC_STAGE_NEXT(buf)                                                                                               \
    if ((p.ablate == 1 || p.ablate == 3) && s_issued >= NS) {                                                                               \
    } else if (ut) {                                                                                                            \
        const uint32_t so = u_tapoff + (uint32_t)(u_c * ES);                                                             \
        const bool from_up = UP && u_c < p.up_C;      /* (uniform) this K tile lies in the upsampled half of the concat */   \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                                 \
            const uint32_t o = ((xmask[i] >> u_tap) & 1ull) ? (from_up ? xoffu[UP ? i : 0] : xoffl[i]) + so : OOB;        \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(from_up ? usrd : xsrd, (lds_ptr_t)&lds[buf][(PASS * i + uwave * RPI) * KC], 16, (int)o, 0, 0, 0); \
        }                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) {                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[buf][BP * KC + (PASS * i + uwave * RPI) * KC], 16,     \
                                                     (int)(woff[i] + (uint32_t)(s_kt * BK * ES)), 0, 0, 0);             \
        }                                                                                                                \
        u_c += BK;                                                                                                       \
        if (u_c == p.Cin) {                                                                                              \
            u_c = 0;                                                                                                     \
            u_tap = min(u_tap + 1, 63);                                                                                  \
            if (++u_s == p.kw) { u_s = 0; u_tapoff += tap_y; } else { u_tapoff += tap_x; }                               \
        }                                                                                                                \
    } else {                                                                                                             \
        const uint32_t tc = kc_off + (uint32_t)(kc_c * ES);                                                              \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                                 \
            const uint32_t o = ((xmask[i] >> min(kc_t, 63)) & 1ull) ? xoff[i] + tc : OOB;                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)&lds[buf][(PASS * i + uwave * RPI) * KC], 16, (int)o, 0, 0, 0); \
        }                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) {                                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[buf][BP * KC + (PASS * i + uwave * RPI) * KC], 16,     \
                                                     (int)(woff[i] + (uint32_t)(s_kt * BK * ES)), 0, 0, 0);             \
        }                                                                                                                \
        int cc = kc_c + BK;                                                                                              \
        while (cc >= p.Cin) {                                                                                            \
            cc -= p.Cin;                                                                                                 \
            ++kc_t;                                                                                                      \
            if (++kc_s == p.kw) { kc_s = 0; kc_off += tap_y; } else { kc_off += tap_x; }                                 \
        }                                                                                                                \
        kc_c = cc;                                                                                                       \
    }                                                                                                                    \
    ++s_issued;                                                                                                          \
    if (++s_kt == s_kt1) {                                                                                               \
        s_v += G;                                                                                                        \
        VC_TILE_STATE(s_v);                                                                                              \
    }

    VC_TILE_STATE(s_v);

    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;
    int xfrag[KC / 4][PT], wfrag[KC / 4][CT];
#pragma unroll
    for (int h = 0; h < KC / 4; ++h) {
#pragma unroll
        for (int i = 0; i < PT; ++i) xfrag[h][i] = 16 * lds_slot<KC>(wp * WTP + i * 16 + frow, h * 4 + fch);          // byte offsets in a stage
#pragma unroll
        for (int i = 0; i < CT; ++i) wfrag[h][i] = 16 * (BP * KC + lds_slot<KC>(wc * WTC + i * 16 + frow, h * 4 + fch));
    }

    // NS-stage ring: K tiles g+1 .. g+NS-1 of this workgroup's tile sequence are in flight while K tile g is multiplied.
    // LDS-DMA loads return in order, so "the next K tile has landed" is vmcnt <= (NS-2) * PER (the epilogue's loads and stores
    // also sit on the VM counter: they can only make this wait longer, never shorter than needed -- loads complete in order).
    //
    // The fragment reads are inline asm and the barrier is the raw s_barrier: hipcc counts an LDS-DMA as a pending LDS
    // write and puts `s_waitcnt vmcnt(0)` in front of every ds_read (and inside __syncthreads) it can see, which drains
    // the tile that was just issued and serialises DMA and MFMA.  The waits that are needed are written out below.
    VC_TS(1);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)&lds[0][0];
    constexpr uint32_t STAGE_BYTES = ROWS * KC * 16;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) { VC_STAGE_NEXT(st); }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER) : "memory");
    __builtin_amdgcn_s_barrier();
    VC_TS(2);
    int sbuf = NS - 1;
    uint32_t boff = lds_base;
    for (int v = blockIdx.x; v < nitems; v += G) {
        const int tile = VC_TILE_OF(SK ? v / KS : v);
        const int split = SK ? v % KS : 0;
        const int nk_item = SK ? (split + 1) * nk / KS - split * nk / KS : nk;
        const int m0 = (tile / tiles_c) * BP, n0 = (tile % tiles_c) * BC;
        f32x4 acc[CT][PT];
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk_item; ++kt) {
            VC_STAGE_NEXT(sbuf);                        // into the slot consumed last iteration (all waves passed its barrier)
            sbuf = sbuf + 1 == NS ? 0 : sbuf + 1;
            if constexpr (FP8) {
                // MX-scaled fp8 MFMA, K = 128 per instruction (twice the bf16 rate): a lane's 32 K-bytes are the chunks fch and 4 + fch of
                // its LDS row -- the same two ds_read_b128 the bf16 halves issue; any K assignment works as long as both operands use
                // the same one.  Block scales are 1 (E8M0 0x7f): per-channel weight scales are applied in the epilogue.
                u32x4v xr[2][PT], wr[2][CT];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < PT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[h][i]) : "v"(boff + xfrag[h][i]) : "memory");
#pragma unroll
                    for (int i = 0; i < CT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wr[h][i]) : "v"(boff + wfrag[h][i]) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                typedef int i32x8 __attribute__((ext_vector_type(8)));
                i32x8 xa[PT], wa[CT];
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    asm volatile("" : "+v"(xr[0][i])); asm volatile("" : "+v"(xr[1][i]));
                    xa[i] = (i32x8){(int)xr[0][i].x, (int)xr[0][i].y, (int)xr[0][i].z, (int)xr[0][i].w, (int)xr[1][i].x, (int)xr[1][i].y, (int)xr[1][i].z, (int)xr[1][i].w};
                }
#pragma unroll
                for (int i = 0; i < CT; ++i) {
                    asm volatile("" : "+v"(wr[0][i])); asm volatile("" : "+v"(wr[1][i]));
                    wa[i] = (i32x8){(int)wr[0][i].x, (int)wr[0][i].y, (int)wr[0][i].z, (int)wr[0][i].w, (int)wr[1][i].x, (int)wr[1][i].y, (int)wr[1][i].z, (int)wr[1][i].w};
                }
#pragma unroll
                for (int a = 0; a < CT; ++a)
#pragma unroll
                    for (int b = 0; b < PT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa[a], xa[b], acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else
#pragma unroll
            for (int h = 0; h < KC / 4; ++h) {
                u32x4v xr[PT], wr[CT];
#pragma unroll
                for (int i = 0; i < PT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[i]) : "v"(boff + xfrag[h][i]) : "memory");
#pragma unroll
                for (int i = 0; i < CT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wr[i]) : "v"(boff + wfrag[h][i]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                Chunk xa[PT], wa[CT];
#pragma unroll
                for (int i = 0; i < PT; ++i) { asm volatile("" : "+v"(xr[i])); xa[i].u = xr[i]; }      // the MFMAs below depend on the wait above
#pragma unroll
                for (int i = 0; i < CT; ++i) { asm volatile("" : "+v"(wr[i])); wa[i].u = wr[i]; }
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
            boff = boff + STAGE_BYTES == lds_base + NS * STAGE_BYTES ? lds_base : boff + STAGE_BYTES;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (SK) {
            // Partial sums out, ticket, and for the last arrival: all KS partial sums back in split order.  The hand-over crosses XCDs (each
            // has its own L2), but a release / acquire FENCE at agent scope writes back and invalidates a whole L2 (measured: the split
            // launches ran 3 - 8 x slower than the unsplit ones).  Instead the partial sums themselves travel with agent-scope atomic
            // stores / loads (sc1: write-through / cache-bypassing dword accesses), ordered against the ticket by the VM counter.
            constexpr int NT = NW * 64;
            float* ws = p.sk_ws + (size_t)tile * KS * (CT * PT * NT * 4);
            float* mine = ws + (size_t)split * (CT * PT * NT * 4);
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int b = 0; b < PT; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) __hip_atomic_store(mine + (((a * PT + b) * 4 + j) * NT + tid), acc[a][b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's partial sums have reached memory
            __shared__ int sk_ticket;
            __syncthreads();                                      // ... and every other thread's of the workgroup
            if (tid == 0) sk_ticket = __hip_atomic_fetch_add(p.sk_tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = sk_ticket == KS - 1;        // (uniform)
            if (!last) continue;
            if (tid == 0) __hip_atomic_store(p.sk_tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the tickets are zero again when the launch ends
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // four splits' partial sums are requested together (64 - 128 loads in flight per lane: one memory latency per group instead of one per
            // split -- the reads were the larger half of a split launch's time), then added in split order; a split past KS reads split
            // KS - 1 again and is not added
            for (int s0 = 0; s0 < KS; s0 += 4) {
                float v[4][CT * PT * 4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* part = ws + (size_t)min(s0 + u, KS - 1) * (CT * PT * NT * 4);
#pragma unroll
                    for (int i = 0; i < CT * PT * 4; ++i) v[u][i] = __hip_atomic_load(part + (i * NT + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool on = s0 + u < KS;              // (uniform)
#pragma unroll
                    for (int a = 0; a < CT; ++a)
#pragma unroll
                        for (int b = 0; b < PT; ++b)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[a][b][j] = on ? acc[a][b][j] + v[u][(a * PT + b) * 4 + j] : acc[a][b][j];
                }
            }
        }
        // epilogue: D[channel = (lane>>4)*4 + reg][pixel = lane&15]; the next tile's first K tiles are already in flight
        if constexpr (FP8) conv_epilogue_fp8<PT, CT>(p, acc, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow);
        else conv_epilogue<PT, CT, F32>(p, acc, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the K tiles issued past the last tile before the LDS is released
    VC_TS(3);
    if (p.dbg) { VC_TS(4); }
#undef VC_STAGE_NEXT
#undef VC_TILE_STATE
#undef VC_TILE_OF
#undef VC_TS
}

// ---- halo-staged 3x3 / stride 1 / pad 1 (bf16) ---------------------------------------------------------------------------
// The implicit GEMM above stages every output pixel's nine taps separately: each input line travels L2 -> LDS nine times, and
// the K loops of the 3x3 layers (93 % L2 hits) sit at half of the L2 bandwidth.  Here the K loop is turned inside out: outer
// loop over 32-channel slices, inner loop over the 9 taps.  A workgroup owns BP consecutive output pixels; per slice it stages
// the input rows those pixels touch ONCE -- rows g0-1 .. g1+1 of the flattened (batch, y) row space are contiguous in NHWC, so
// the patch is a plain run of `npix` pixels starting at pixel (g0-1)*W -- and the nine taps read their MFMA operand from that
// patch at pixel + dy*W + dx.  Taps that fall outside the image (also across the batch seam inside a patch) read a zero
// pixel instead: a 9-bit validity mask per lane, the addresses of all nine taps are loop invariant.  Weights stream through
// the same NS-stage LDS-DMA ring as above, one (tap, slice) tile of [BC][32] per step.  The MFMA / accumulation order per
// output equals the implicit GEMM's only up to the order of the K tiles (tap-major there, slice-major here): results agree
// to fp32 rounding, not bit for bit (same tolerance as between tile configurations with different K chunking... they are
// identical there; here the tests' bf16 / fp32 tolerances apply).
template <int BP, int BC, int WP, int WC, int NS, int XI>
__global__ __launch_bounds__(256, 1) void conv3x3_halo_kernel(const ConvP p) {
    constexpr int KC = 4, ES = 2, BK = 32;
    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (0)] = wall_clock64(); } while (0);      // diagnostics (VC_CONV_DBG): phase timestamps like conv_igemm_kernel
    constexpr int PASS = 64;                       // weight rows covered by one DMA instruction of all four waves (16 per wave)
    constexpr int WI = (BC + PASS - 1) / PASS;
    constexpr int WROWS = WI * PASS;
    constexpr int WTP = BP / WP, WTC = BC / WC, PT = WTP / 16, CT = WTC / 16;
    constexpr int ZP = XI * 64 - 1;                // index of the zero pixel: last pixel of a patch buffer, never reached by a patch
    constexpr int XCH = XI * 256;                  // 16-byte chunks per patch buffer
    constexpr uint32_t OOB = 0x80000000u;
    static_assert(WP * WC == 4 && WTP % 16 == 0 && WTC % 16 == 0, "tile shape");
    static_assert((NS - 2) * WI + XI <= 63, "counted vmcnt");
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * XCH + NS * WROWS * KC];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

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
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int W = p.W, H = p.H;

    // patch geometry (workgroup-uniform)
    const int g0 = m0 / W;
    const int g1 = (min(m0 + BP, p.M) - 1) / W;
    const int gp0 = (g0 - 1) * W;                  // first patch pixel (may be negative: row -1 of the first image)
    const int npix = (g1 - g0 + 3) * W;            // <= ZP, checked by the launcher

    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    // patch staging: instruction i of this wave fills chunks [(i*4 + wave)*64, +64); lane -> (patch pixel, chunk slot)
    uint32_t xsrc[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int e = (i * 4 + wave) * 64 + lane;
        const int pp = e >> 2, cpos = e & 3;
        const int chunk = cpos ^ ((pp >> 1) & 2);                                 // source-side swizzle, see xaddr below
        const int gp = gp0 + pp;
        xsrc[i] = (pp < npix && gp >= 0) ? (uint32_t)((gp * p.in_cs + p.in_co) * ES + chunk * 16) : OOB;
    }
    // weight staging: rows of 4 chunks, 16 rows per wave-instruction (as in conv_igemm_kernel with KC = 4)
    const int prow = wave * 16 + (lane >> 2);
    const int wchunk = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
    uint32_t woff[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) woff[i] = (uint32_t)(((n0 + prow + PASS * i) * p.Kw + wchunk * 8) * ES);

    // fragment addresses: this lane's pixel of every pixel tile, its nine taps (byte offsets inside a patch buffer)
    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;
    uint32_t xaddr[PT][9];
    {
        const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int m = m0 + wp * WTP + i * 16 + frow;
            const bool ok = m < p.M;
            const int mm = ok ? m : m0;
            int g = (int)((float)mm * inv_w);                                     // global row, +-1 fix-up (mm < 2^24)
            g -= (g * W > mm) ? 1 : 0;
            g += ((g + 1) * W <= mm) ? 1 : 0;
            const int x = mm - g * W;
            int b = (int)((float)g * inv_h);
            b -= (b * H > g) ? 1 : 0;
            b += ((b + 1) * H <= g) ? 1 : 0;
            const int y = g - b * H;
            const int pc = mm - gp0;                                              // patch index of the centre tap
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool valid = ok && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const int px = valid ? pc + dy * W + dx : ZP;
                // chunk slot = fch ^ 2 * bit 2 of the pixel index.  The fragment reads of a tap start at an ARBITRARY patch pixel
                // (lds_slot<4>'s permutation is conflict-free only for bases that are multiples of 16: 33 % of the LDS cycles of this
                // kernel were bank conflicts); this one keeps the four lanes of a ds_read_b128 group that share pixel & 3 on four
                // different 16-byte slots for every base (exhaustive check over bases and the hardware's lane groups).
                xaddr[i][t] = (uint32_t)((px * 4 + (fch ^ ((px >> 1) & 2))) * 16);
            }
        }
    }
    int wfrag[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wfrag[i] = 16 * lds_slot<4>(wc * WTC + i * 16 + frow, fch);

    f32x4 acc[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_inputs_settle<CT * PT>(&acc[0][0]);
    // the residual (ReID conv2 + shortcut, YOLO Bottleneck 3x3 + shortcut) is fetched NOW: this workgroup computes one tile and ends,
    // so the epilogue's residual reads had a full memory latency to themselves (64 -> 64 at 25 x 25: 0.135 ms with, 0.100 ms without)
    u32x2r rpre[PT][CT];
    // (only where the 2 * PT * CT registers it holds through the K loop do not cost a wave of occupancy)
    const bool have_res = PT * CT <= 8 && p.res_mode != RES_NONE && conv_epilogue_fast_bf16<PT, CT>(p) &&
                          ((p.act == ACT_SILU && p.res_mode == RES_AFTER_ACT) || (p.act == ACT_RELU && p.res_mode == RES_BEFORE_ACT));
    if (have_res) conv_residual_fetch<PT, CT>(p, rpre, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow);

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)&lds[0];
    constexpr uint32_t XBYTES = XCH * 16, WSTAGE = WROWS * KC * 16;
    const uint32_t wring = lds_base + 2 * XBYTES;
    const int nslices = p.Cin / BK;
    const int nk = nslices * 9;                    // K tiles, slice-major: kt = slice * 9 + tap

    // zero pixels (one per patch buffer): plain LDS stores, ordered before the first barrier
    if (tid < 8) lds[(tid >> 2) * XCH + ZP * 4 + (tid & 3)] = make_uint4(0u, 0u, 0u, 0u);

#define VC_XSTAGE(slice, xb)                                                                                              \
    {                                                                                                                     \
        const uint32_t so = (slice) < nslices ? (uint32_t)((slice) * BK * ES) : OOB;                                       \
        _Pragma("unroll") for (int i = 0; i < XI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)&lds[(xb) * XCH + (i * 4 + uwave) * 64], 16,           \
                                                     (int)((xsrc[i] | so) >= OOB ? OOB : xsrc[i] + so), 0, 0, 0);          \
    }
#define VC_WSTAGE(kt, st)                                                                                                 \
    {                                                                                                                     \
        const int kk = (kt);                                                                                               \
        const int sl = kk / 9, tp = kk - sl * 9;                                                                           \
        const uint32_t ko = kk < nk ? (uint32_t)((tp * p.Cin + sl * BK) * ES) : OOB;                                       \
        _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[2 * XCH + (st) * WROWS * KC + (PASS * i + uwave * 16) * KC], 16, \
                                                     (int)(ko >= OOB ? OOB : woff[i] + ko), 0, 0, 0);                       \
    }

    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (1)] = wall_clock64(); } while (0);
    VC_XSTAGE(0, 0);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) VC_WSTAGE(st, st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (2)] = wall_clock64(); } while (0);
    int kt = 0, sbuf = NS - 1;
    uint32_t woffs = wring;                        // LDS address of the weight stage being multiplied
    for (int slice = 0; slice < nslices; ++slice) {
        const uint32_t xb = lds_base + (uint32_t)(slice & 1) * XBYTES;
        VC_XSTAGE(slice + 1, (slice + 1) & 1);     // next slice's patch: its buffer was last read one slice ago
#pragma unroll
        for (int t = 0; t < 9; ++t, ++kt) {
            VC_WSTAGE(kt + NS - 1, sbuf);
            sbuf = sbuf + 1 == NS ? 0 : sbuf + 1;
            u32x4v xr[PT], wr[CT];
#pragma unroll
            for (int i = 0; i < PT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[i]) : "v"(xb + xaddr[i][t]) : "memory");
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wr[i]) : "v"(woffs + wfrag[i]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < PT; ++i) asm volatile("" : "+v"(xr[i]));
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("" : "+v"(wr[i]));
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int b = 0; b < PT; ++b) mfma_bf16_inplace(acc[a][b], wr[a], xr[b]);
            woffs = woffs + WSTAGE == wring + NS * WSTAGE ? wring : woffs + WSTAGE;
            // weight tile kt+1 has landed once at most the newer weight tiles -- and, while it is still older than this
            // slice's patch prefetch, that prefetch -- are outstanding
            if (t < NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI + XI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_results_settle<CT * PT>(&acc[0][0]);
    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (3)] = wall_clock64(); } while (0);
#undef VC_XSTAGE
#undef VC_WSTAGE
    conv_epilogue<PT, CT, false>(p, acc, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow, rpre, have_res);
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (4)] = wall_clock64(); } while (0); }
}

// ---- halo-staged 3x3 / stride 2 / pad 1 (bf16) ---------------------------------------------------------------------------
// The down-sampling layers (YOLO 3-P3 .. 7-P5 and the two head 3x3/s2) ran on the implicit GEMM at 480 - 880 TFLOP/s: nine separately
// staged taps per output pixel, 9/4 L2 -> LDS trips per input line.  Same inside-out K loop as conv3x3_halo_kernel (patch staged once,
// taps read it at shifted addresses, weights through the LDS-DMA ring), with what stride 2 changes:
//   * the tile is a RECTANGLE of th x tw output pixels (th * tw <= 128, chosen by the launcher per layer), not a run of the flattened
//     pixel index: a run of 128 outputs of an 80-wide map touches seven 160-pixel input rows, the 8 x 16 rectangle a 17 x 33 patch
//     (1.10 x the 4 inputs per output nothing can avoid).  Rows are rows of the flattened (batch, y) space (H = 2 Ho, so input row =
//     2 * output row - 1 + dy across images too): no ragged tiles at image bottoms.  The 128 pixel slots of the MFMA tiles are the
//     rectangle row-major; slots past th * tw, past the map's right edge or the last row are masked lanes.
//   * the patch is staged BY PARITY CLASS: tap (dy, dx) of output (r, c) reads patch pixel (2r + dy, 2c + dx), so the taps with
//     (dy & 1, dx & 1) = (rp, cp) touch only patch rows of parity rp and columns of parity cp -- 4, 2, 2 and 1 taps for the classes
//     even/even, even/odd, odd/even, odd/odd.  One class at a time is in LDS, (th + 1) x (tw + 1) pixels, and within it the 16 lanes
//     of an MFMA fragment read CONSECUTIVE pixels (r + dy/2, c + dx/2): the stride-2 gather becomes the stride-1 kernel's access
//     pattern.  A quarter of the patch at a time is also what lets a pixel be staged with 64 channels (its full 128-byte line, 20 KB
//     per class) instead of a 32-channel slice of the whole patch (37 KB): the first version fetched half lines, and the second half
//     came from HBM again 12 us later (FETCH_SIZE 822 MB for the 419 MB input of YOLO layer 3: HBM bound at 2 x the traffic).
//   K order: 64-channel group, parity class, 32-channel half, tap -- 18 MFMA steps per group; the weight ring follows that order.
// One patch buffer: the class loads are exposed to the workgroup and covered by the other workgroups of the CU (a second buffer for the
// next class was built and measured: never faster, the LDS it takes costs a workgroup per CU).
// Output rows are not consecutive in memory, so the epilogue takes the lane's pixel index per MFMA tile (mrow).
template <int PT, int CT, int ACT>
__device__ __forceinline__ void conv_epilogue_bf16_rows(const ConvP& p, f32x4 (&acc)[CT][PT], const int (&mrow)[PT], int nbase) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t osrd2 = __builtin_amdgcn_make_buffer_rsrc(p.split > 0 ? p.out2 : p.out, 0, 0x7ffffff0, 0x00020000);
    float4 bias[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) bias[a] = nbase + a * 16 < p.Cout ? *(const float4*)(p.bias + nbase + a * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool odd = ((threadIdx.x >> 4) & 1) != 0;
#pragma unroll
    for (int b = 0; b < PT; b += 2) {
        const int m = odd ? mrow[b + 1] : mrow[b];                  // after the lane-pair exchange below (conv_epilogue_bf16)
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = nbase + a * 16;
            u32x2 P[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[4] = {acc[a][b + t][0] + bias[a].x, acc[a][b + t][1] + bias[a].y, acc[a][b + t][2] + bias[a].z, acc[a][b + t][3] + bias[a].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j];
                    if constexpr (ACT == ACT_SILU) x = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
                    if constexpr (ACT == ACT_RELU) x = x > 0.f ? x : 0.f;
                    v[j] = x;
                }
                P[t] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
            const u32x2 sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
            const u32x2 sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
            const u32x4 o4 = {sx.x, sy.x, sx.y, sy.y};
            const int nn = odd ? n - 4 : n;
            const bool ok = n < p.Cout && m < p.M;
            if (p.split == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(o4, osrd, ok ? (m * p.out_cs + p.out_co + nn) * 2 : (int)0x80000000u, 0, 0);
            } else {                                                   // two destinations (C3.cv1 | cv2), as conv_epilogue_bf16
                const bool second = nn >= p.split;
                __builtin_amdgcn_raw_buffer_store_b128(o4, osrd, (ok && !second) ? (m * p.out_cs + p.out_co + nn) * 2 : (int)0x80000000u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o4, osrd2, (ok && second) ? (m * p.out2_cs + p.out2_co + nn - p.split) * 2 : (int)0x80000000u, 0, 0);
            }
        }
    }
}

// ---- conv3x3s2_halo_kernel<..., F2>: the pointwise conv that is the ONLY reader of this conv's output, on the tile while it is on chip ----------
// YOLOv5s layer 3 (Conv 64 -> 128, 3x3 / s2, 80^2) is read by C3.cv1 | cv2 of layer 4 (one 1x1 launch, 128 -> 64 | 64) and by nothing else: as two
// launches its 210 MB (128 frames) are written and read straight back.  A wave of the 256 x 128 tile holds ALL 128 channels of its 64 pixels, so
// after the 3x3's own epilogue (bias, SiLU, bf16 rounding: the values the stand-alone launch would have stored) the packed outputs are re-laid
// into MFMA B operands across the wave's four 16-lane rows (ds_bpermute: the LDS crossbar, no LDS memory), the 1x1's 32 KB of weights wait as
// ready-made fragments in the LDS the K loop has finished with, and the 1x1 runs its four K steps in the stand-alone kernel's order (ks = 0..3 from a
// zero accumulator, one v_mfma_f32_16x16x32_bf16 per step and tile): bit-identical to the two launches (tests/test_gpu_round5.py).
// O[t2][a] of a lane in 16-lane row r (conv_epilogue_bf16_rows' lane-pair exchange): pixel tile 2 t2 + (r & 1), channels 16 a + 8 (r >> 1) .. + 7.
// B operand of K step ks for pixel tile t, row kg: channels 32 ks + 8 kg .. + 7 = O[t >> 1][2 ks + (kg >> 1)] of row 2 (kg & 1) + (t & 1).
template <int PT, int CT>
__device__ __forceinline__ void s2_pointwise_stage(const ConvP& p, const ConvP& q, f32x4 (&acc)[CT][PT], const int (&mrow)[PT], uint4* lds) {
    static_assert(PT == 4 && CT == 8, "the 256 x 128 tile with four waves side by side in pixels");
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fch = lane >> 4;
    // the 1x1's weights: fragment f = ks * 8 + a2 (channel tile a2, K step ks), this wave fetches f = 4 i + wave
    uint4 wreg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = i * 4 + wave, ks = f >> 3, a2 = f & 7;
        wreg[i] = *(const uint4*)((const char*)q.w + ((size_t)(a2 * 16 + frow) * q.Kw + ks * 32 + fch * 8) * 2);
    }
    // the 3x3's epilogue, kept in registers
    uint4 O[2][CT];
    {
        float4 bias[CT];
#pragma unroll
        for (int a = 0; a < CT; ++a) bias[a] = *(const float4*)(p.bias + a * 16 + fch * 4);
#pragma unroll
        for (int b = 0; b < PT; b += 2)
#pragma unroll
            for (int a = 0; a < CT; ++a) {
                u32x2 P[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v[4] = {acc[a][b + t][0] + bias[a].x, acc[a][b + t][1] + bias[a].y, acc[a][b + t][2] + bias[a].z, acc[a][b + t][3] + bias[a].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[j]));
                    P[t] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
                const u32x2 sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
                const u32x2 sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
                O[b >> 1][a] = make_uint4(sx.x, sy.x, sx.y, sy.y);
            }
    }
    __syncthreads();                                   // every wave's LDS-DMA writes (the ring's look-ahead past the last step) have landed
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[(i * 4 + wave) * 64 + lane] = wreg[i];
    __syncthreads();
    f32x4 acc2[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool upper = (fch >> 1) != 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        Chunk xb[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int src = (frow + 16 * (2 * (fch & 1) + (t & 1))) * 4;
            const uint4 lo = O[t >> 1][2 * ks], hi = O[t >> 1][2 * ks + 1];
            const uint32_t l4[4] = {lo.x, lo.y, lo.z, lo.w}, h4[4] = {hi.x, hi.y, hi.z, hi.w};
            uint32_t d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)l4[j]);
                const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)h4[j]);
                d[j] = upper ? y : x;
            }
            xb[t].u = (u32x4v){d[0], d[1], d[2], d[3]};
        }
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            Chunk wf;
            const uint4 wv = lds[(ks * 8 + a) * 64 + lane];
            wf.u = (u32x4v){wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int t = 0; t < PT; ++t) acc2[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.h, xb[t].h, acc2[a][t], 0, 0, 0);
        }
    }
    conv_epilogue_bf16_rows<PT, CT, ACT_SILU>(q, acc2, mrow, fch * 4);
}

struct S2Steps {          // the 18 steps of a 64-channel group: tap, 32-channel half, parity class, position within the class
    int tap[18], sub[18], cls[18], pos[18], len[18];
};
constexpr S2Steps s2_steps() {
    S2Steps t{};
    const int taps[4][4] = {{0, 2, 6, 8}, {1, 7, -1, -1}, {3, 5, -1, -1}, {4, -1, -1, -1}};
    const int ntap[4] = {4, 2, 2, 1};
    int n = 0;
    for (int c = 0; c < 4; ++c)
        for (int sub = 0; sub < 2; ++sub)
            for (int k = 0; k < ntap[c]; ++k) {
                t.tap[n] = taps[c][k]; t.sub[n] = sub; t.cls[n] = c; t.pos[n] = sub * ntap[c] + k; t.len[n] = 2 * ntap[c];
                ++n;
            }
    return t;
}

template <int BP, int BC, int WP, int WC, int NS, bool F2 = false>     // F2: followed on the tile by the pointwise conv q that alone reads its output
__global__ __launch_bounds__(256, BP * BC <= 128 * 128 ? 3 : 2) void conv3x3s2_halo_kernel(const ConvP p, const ConvP q) {
    constexpr int KC = 4, ES = 2;
    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (0)] = wall_clock64(); } while (0);      // diagnostics (VC_CONV_DBG): phase timestamps like conv_igemm_kernel
    constexpr int XI = BP / 128 * 5;               // DMA instructions per class: 5 x 256 chunks = 160 pixels of 128 bytes per 128 outputs
    constexpr int PASS = 64;
    constexpr int WI = (BC + PASS - 1) / PASS;
    constexpr int WROWS = WI * PASS;
    constexpr int WTP = BP / WP, WTC = BC / WC, PT = WTP / 16, CT = WTC / 16;
    constexpr int ZP = XI * 32 - 1;                // the zero pixel: last pixel of a patch buffer, never reached by a class
    constexpr int XCH = XI * 256;
    constexpr uint32_t OOB = 0x80000000u;
    constexpr S2Steps ST = s2_steps();
    static_assert(WP * WC == 4 && WTP % 32 == 0 && WTC % 16 == 0, "tile shape");
    static_assert((NS - 2) * WI + XI <= 63 && NS <= 6, "counted vmcnt");
    __shared__ __attribute__((aligned(16))) uint4 lds[XCH + NS * WROWS * KC];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int nblk = gridDim.x;
    const int tiles_c = (p.Cout + BC - 1) / BC;
    int tile;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int TH = p.s2_th, TW = p.s2_tw;          // tile rows x columns (outputs), TH * TW <= BP, (TH + 1) * (TW + 1) <= ZP
    const int CW = TW + 1;                         // row length of a class in the patch buffer
    const int Wo = p.Wo, Ho = p.Ho, W = p.W;
    const int G = p.B * Ho, GR = p.B * p.H;        // rows of the flattened (batch, y) spaces
    const int ctiles = (Wo + TW - 1) / TW;
    const int ptile = tile / tiles_c;
    const int n0 = (tile - ptile * tiles_c) * BC;
    const int g_top = (ptile / ctiles) * TH;
    const int x0 = (ptile % ctiles) * TW;
    const int gr0 = 2 * g_top - 1, c0 = 2 * x0 - 1;   // input row / column of the patch's corner
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int ncg = p.Cin / 64;

    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    // patch staging: buffer pixel lp = i * CW + j holds patch pixel (2i + rp, 2j + cp) of the class being staged, 8 chunks of 16 B;
    // lane -> (lp, chunk slot), the class and the channel group only add uniform offsets
    int xij[XI];                                   // i << 16 | j << 8 | chunk * 16 (i = 255: no such pixel)
    {
        const float inv_cw = 1.0f / (float)CW;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int e = (k * 4 + wave) * 64 + lane;
            const int lp = e >> 3, cpos = e & 7;
            const int chunk = cpos ^ (((lp >> 1) & 3) << 1);                          // source-side swizzle, see the fragment reads
            const int i = (int)(((float)lp + 0.5f) * inv_cw);                         // lp < 320: exact
            const int j = lp - i * CW;
            xij[k] = ((i <= TH ? i : 255) << 16) | (j << 8) | (chunk * 16);
        }
    }
    const int row_bytes = W * p.in_cs * ES, px_bytes = p.in_cs * ES;
    const int corner = ((gr0 * W + c0) * p.in_cs + p.in_co) * ES;
    const int prow = wave * 16 + (lane >> 2);
    const int wchunk = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
    uint32_t woff[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) woff[i] = (uint32_t)(((n0 + prow + PASS * i) * p.Kw + wchunk * 8) * ES);

    // fragment addresses: this lane's output pixel (r, c) of every MFMA pixel tile reads buffer pixel (r + dy/2, c + dx/2) of the class
    // of tap (dy, dx): four addresses per tile.  The only taps that can leave the image are dy = 0 on an image's first row and dx = 0 on
    // the first column; masked lanes read the zero pixel everywhere.
    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;
    int xp[PT];                                    // buffer pixel of (r, c) in every class
    uint32_t flg = 0;                              // per MFMA tile i, bits 3i .. 3i+2: masked lane, image's first row, first column
    constexpr uint32_t ZADDR = (uint32_t)ZP * 128;
    const float inv_tw = 1.0f / (float)TW, inv_ho = 1.0f / (float)Ho;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int q = wp * WTP + i * 16 + frow;                                       // slot of the tile, row-major over TH x TW
        const int r = (int)(((float)q + 0.5f) * inv_tw);
        const int c = q - r * TW;
        const int g = g_top + r, x = x0 + c;
        const bool ok = r < TH && g < G && x < Wo;
        int b = (int)((float)g * inv_ho);                                             // image of the row, +-1 fix-up (g < 2^24)
        b -= (b * Ho > g) ? 1 : 0;
        b += ((b + 1) * Ho <= g) ? 1 : 0;
        flg |= ((ok ? 0u : 1u) | (g - b * Ho == 0 ? 2u : 0u) | (x == 0 ? 4u : 0u)) << (3 * i);
        xp[i] = r * CW + c;
    }
    int wfrag[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wfrag[i] = 16 * lds_slot<4>(wc * WTC + i * 16 + frow, fch);

    f32x4 acc[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_inputs_settle<CT * PT>(&acc[0][0]);

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)&lds[0];
    constexpr uint32_t XBYTES = XCH * 16, WSTAGE = WROWS * KC * 16;
    const uint32_t wring = lds_base + XBYTES;

    if (tid < 8) lds[ZP * 8 + tid] = make_uint4(0u, 0u, 0u, 0u);

    // stage class `cls` (row parity cls >> 1, column parity cls & 1) of channel group `cg`
#define VC_XCLASS(cg, cls)                                                                                            \
    {                                                                                                                     \
        const int rp = (cls) >> 1, cp = (cls) & 1;                                                                         \
        const int add = rp * row_bytes + cp * px_bytes + (cg) * 128;                                                       \
        const bool live = (cg) < ncg;                                                                                      \
        _Pragma("unroll") for (int k = 0; k < XI; ++k) {                                                                   \
            int ij = xij[k];                                                                                               \
            asm volatile("" : "+v"(ij));               /* opaque: no per-class copies hoisted out of the group loop */      \
            const int i = ij >> 16, j = (ij >> 8) & 255;                                                                   \
            const int xo = corner + 2 * i * row_bytes + 2 * j * px_bytes + (ij & 255);                                     \
            const bool in = live && i <= TH - rp && j <= TW - cp && (unsigned)(gr0 + 2 * i + rp) < (unsigned)GR &&         \
                            (unsigned)(c0 + 2 * j + cp) < (unsigned)W;                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)&lds[(k * 4 + uwave) * 64], 16,                        \
                                                     in ? xo + add : (int)OOB, 0, 0, 0);                                    \
        }                                                                                                                 \
    }
    // weight tile of step `st` (0 .. 17) of channel group `cg` into ring stage `rs`
#define VC_WSTEP(cg, st, rs)                                                                                              \
    {                                                                                                                     \
        const uint32_t ko = (cg) < ncg ? (uint32_t)((ST.tap[st] * p.Cin + (cg) * 64 + ST.sub[st] * 32) * ES) : OOB;        \
        _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[XCH + (rs) * WROWS * KC + (PASS * i + uwave * 16) * KC], 16,  \
                                                     (int)(ko >= OOB ? OOB : woff[i] + ko), 0, 0, 0);                       \
    }

    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (1)] = wall_clock64(); } while (0);
    VC_XCLASS(0, 0);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) VC_WSTEP(0, st, st);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (2)] = wall_clock64(); } while (0);
    int sbuf = NS - 1;
    uint32_t woffs = wring;
    for (int cg = 0; cg < ncg; ++cg) {
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int cls = ST.cls[s], tap = ST.tap[s], dy = tap / 3, dx = tap % 3, v = (dy >> 1) * 2 + (dx >> 1);
            const uint32_t xb = lds_base;
            { const int s2 = (s + NS - 1) % 18, carry = (s + NS - 1) / 18; VC_WSTEP(cg + carry, s2, sbuf); }
            sbuf = sbuf + 1 == NS ? 0 : sbuf + 1;
            u32x4v xr[PT], wr[CT];
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                // 128-byte pixels: slot = chunk ^ 2 * ((P >> 1) & 3).  A ds_read_b128 lane group holds 8 consecutive-or-nearly pixels of one
                // 16-byte chunk index and 8 of the next; per half of the 256-byte bank row (pixel parity) that is 4 + 4 pixels whose
                // (P >> 1) & 3 are all different: 8 different slots for every base pixel.  Computed per step from the tile's one base
                // pixel (VALU is idle here; 18 steps x PT precomputed addresses cost a wave of occupancy)
                int P = xp[i];
                asm volatile("" : "+v"(P));        // opaque: or the addresses of all 18 steps are hoisted out of the group loop
                P += (v >> 1) * CW + (v & 1);
                const uint32_t bad = flg & ((1u | (dy == 0 ? 2u : 0u) | (dx == 0 ? 4u : 0u)) << (3 * i));
                const uint32_t a = bad ? ZADDR + (uint32_t)fch * 16 : (uint32_t)(P * 128 + ((fch ^ (((P >> 1) & 3) << 1)) * 16));
                asm volatile("ds_read_b128 %0, %1" : "=v"(xr[i]) : "v"(xb + (a ^ (uint32_t)(ST.sub[s] << 6))) : "memory");
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wr[i]) : "v"(woffs + wfrag[i]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < PT; ++i) asm volatile("" : "+v"(xr[i]));
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("" : "+v"(wr[i]));
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int b = 0; b < PT; ++b) mfma_bf16_inplace(acc[a][b], wr[a], xr[b]);
            woffs = woffs + WSTAGE == wring + NS * WSTAGE ? wring : woffs + WSTAGE;
            const bool last_of_class = ST.pos[s] == ST.len[s] - 1;
            if (last_of_class) {
                // the one patch buffer: every wave is done with this class's taps (barrier), then the next class is fetched and waited
                // for with everything before it (the ring's tiles are older) -- the other workgroups of the CU fill the gap
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
                if (cls < 3 || cg + 1 < ncg) {
                    __builtin_amdgcn_s_barrier();
                    if (cls < 3) { VC_XCLASS(cg, cls + 1); } else { VC_XCLASS(cg + 1, 0); }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_results_settle<CT * PT>(&acc[0][0]);
    do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (3)] = wall_clock64(); } while (0);
#undef VC_XCLASS
#undef VC_WSTEP
    const int nbase = n0 + wc * WTC + fch * 4;
    int mrow[PT];                                  // this lane's output pixel per MFMA tile (M = none)
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int q = wp * WTP + i * 16 + frow;
        const int r = (int)(((float)q + 0.5f) * inv_tw);
        mrow[i] = (flg >> (3 * i)) & 1u ? p.M : (g_top + r) * Wo + x0 + q - r * TW;
    }
    if constexpr (F2) {
        if (p.ablate == 8) conv_epilogue_bf16_rows<PT, CT, ACT_SILU>(p, acc, mrow, nbase);     // diagnostics: also store the 3x3's own output
        s2_pointwise_stage<PT, CT>(p, q, acc, mrow, &lds[0]);
    } else {
        if (p.act == ACT_SILU) conv_epilogue_bf16_rows<PT, CT, ACT_SILU>(p, acc, mrow, nbase);
        else if (p.act == ACT_RELU) conv_epilogue_bf16_rows<PT, CT, ACT_RELU>(p, acc, mrow, nbase);
        else conv_epilogue_bf16_rows<PT, CT, ACT_NONE>(p, acc, mrow, nbase);
    }
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (4)] = wall_clock64(); } while (0); }
}

// ---- 1x1 / stride 1 with the weights in registers (bf16) -------------------------------------------------------------------
// The narrow pointwise layers (K <= 128) are bound by everything but the matrix work: two K tiles per output tile, each with its
// DMA issue, vmcnt wait and workgroup barrier, around 16 MFMAs.  With K*N this small a wave can keep its share of the weight
// matrix in registers (CT x KS fragments = 32 / 64 VGPRs) for the whole launch and read its MFMA "B" operand -- lane (pixel,
// 16-byte chunk of the pixel's channel run) -- straight from global memory: no LDS, no barrier, waves fully independent, the
// next pixel block's fragments are fetched before this block's MFMAs and epilogue.  A wave owns one channel group of CT*16
// outputs (NG = Cout / (CT*16) groups, 1, 2 or 4) and walks pixel blocks of PT*16 pixels; the NG waves that share a pixel block
// run side by side in one workgroup (the second to fourth read of a pixel hits L2/L1).  K order, MFMA operand order and the
// epilogue are those of conv_igemm_kernel: bit-identical results.
template <int CT, int KS, int PT, int OCC, int ACT>     // OCC = waves per SIMD the register budget is sized for
__global__ __launch_bounds__(256, OCC) void conv1x1_direct_kernel(const ConvP p) {
    constexpr uint32_t OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fch = lane >> 4;
    const int NG = p.Cout / (CT * 16);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;           // 4 % NG == 0: a workgroup holds whole sets of groups
    const int g = gw % NG, stride = nw / NG;
    const int nblk = (p.M + PT * 16 - 1) / (PT * 16);
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * 2), 0x00020000);
    Chunk wf[CT][KS];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wf[ct][ks].u = *(const u32x4v*)((const char*)p.w + ((size_t)((g * CT + ct) * 16 + frow) * p.Kw + ks * 32 + fch * 8) * 2);
    float4 bias[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) bias[a] = *(const float4*)(p.bias + (g * CT + a) * 16 + fch * 4);
    u32x4 x[PT][KS], xn[PT][KS];
    auto fetch = [&](int blk, u32x4 (&dst)[PT][KS]) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = (blk * PT + pt) * 16 + frow;
            const uint32_t base = (blk < nblk && m < p.M) ? (uint32_t)((m * p.in_cs + p.in_co + fch * 8) * 2) : OOB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dst[pt][ks] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, (int)(base >= OOB ? OOB : base + ks * 64), 0, 0);
        }
    };
    int blk = gw / NG;
    fetch(blk, x);
    for (; blk < nblk; blk += stride) {
        fetch(blk + stride, xn);
        f32x4 acc[CT][PT];
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < CT; ++a)
#pragma unroll
                for (int b = 0; b < PT; ++b) {
                    Chunk xa;
                    xa.u = (u32x4v){x[b][ks].x, x[b][ks].y, x[b][ks].z, x[b][ks].w};
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a][ks].h, xa.h, acc[a][b], 0, 0, 0);
                }
        conv_epilogue_bf16<PT, CT, ACT, RES_NONE>(p, acc, bias, blk * PT * 16, g * CT * 16 + fch * 4, frow);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) x[pt][ks] = xn[pt][ks];
    }
}

// The same kernel for the fp8 path (round 6): pointwise layers with K <= 256 (Bottleneck.cv1, C3.cv1 | cv2, C3.cv3 of the 320^2 - 80^2 levels of
// YOLOv5l at 1280^2, BASELINE.json configs[4]).  Through the implicit GEMM a K = 128 layer is ONE K step per tile: every 64-pixel tile pays
// its tile bookkeeping, a barrier, a 16 KB weight tile re-streamed through LDS for 8 KB of pixels, and the fp8 layers ran no faster than the
// bf16 ones on half the bytes (128 -> 128 at 160^2: 57 us for 105 MB).  Here the weights of a wave's CT x 16 channels sit in registers as MFMA
// A operands (KS steps of K = 128: 8 registers per fragment), the pixels stream global -> registers, one fetch ahead.  K assignment inside a
// step as in conv_igemm_kernel's fp8 branch (a lane's 32 K-bytes = chunks fch and 4 + fch of the 128-byte slice, both operands): the same
// products in the same MFMA, bit-identical results.  Cin = 64: KS = 1, the upper half of the step is out-of-range offsets (zeros).
template <int CT, int KS, int PT, int OCC>
__global__ __launch_bounds__(256, OCC) void conv1x1_direct_fp8_kernel(const ConvP p) {
    constexpr uint32_t OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fch = lane >> 4;
    const int NG = (p.Cout + CT * 16 - 1) / (CT * 16);
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;           // 4 % NG == 0: a workgroup holds whole sets of groups
    const int g = gw % NG, stride = nw / NG;
    const int nblk = (p.M + PT * 16 - 1) / (PT * 16);
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw), 0x00020000);
    u32x4 wlo[CT][KS], whi[CT][KS];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int row = (g * CT + ct) * 16 + frow;                   // rows past Cout are zero rows of the padded weight buffer
            const int off = row * p.Kw + ks * 128 + fch * 16;
            wlo[ct][ks] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, off, 0, 0);
            whi[ct][ks] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, off + 64, 0, 0);
        }
    u32x4 xl[PT][KS], xh[PT][KS], nl[PT][KS], nh[PT][KS];
    const bool half = p.Cin <= 128 * KS - 64;                           // (uniform) Cin = 64: only the first chunk of the last step holds data
    auto fetch = [&](int blk, u32x4 (&lo)[PT][KS], u32x4 (&hi)[PT][KS]) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = (blk * PT + pt) * 16 + frow;
            const uint32_t base = (blk < nblk && m < p.M) ? (uint32_t)(m * p.in_cs + p.in_co + fch * 16) : OOB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                lo[pt][ks] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, (int)(base >= OOB ? OOB : base + ks * 128), 0, 0);
                hi[pt][ks] = __builtin_amdgcn_raw_buffer_load_b128(xsrd, (int)((base >= OOB || (half && ks == KS - 1)) ? OOB : base + ks * 128 + 64), 0, 0);
            }
        }
    };
    int blk = gw / NG;
    fetch(blk, xl, xh);
    for (; blk < nblk; blk += stride) {
        fetch(blk + stride, nl, nh);
        f32x4 acc[CT][PT];
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < CT; ++a) {
                const i32x8 wa = {(int)wlo[a][ks].x, (int)wlo[a][ks].y, (int)wlo[a][ks].z, (int)wlo[a][ks].w, (int)whi[a][ks].x, (int)whi[a][ks].y, (int)whi[a][ks].z, (int)whi[a][ks].w};
#pragma unroll
                for (int b = 0; b < PT; ++b) {
                    const i32x8 xa = {(int)xl[b][ks].x, (int)xl[b][ks].y, (int)xl[b][ks].z, (int)xl[b][ks].w, (int)xh[b][ks].x, (int)xh[b][ks].y, (int)xh[b][ks].z, (int)xh[b][ks].w};
                    acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xa, acc[a][b], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        conv_epilogue_fp8<PT, CT>(p, acc, blk * PT * 16, g * CT * 16 + fch * 4, frow);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { xl[pt][ks] = nl[pt][ks]; xh[pt][ks] = nh[pt][ks]; }
    }
}

// ---- 1x1 / stride 1, streaming form: weights in LDS, a wave owns ALL output channels of its pixels (bf16) -----------------------
// The wide-map pointwise layers (K, N <= 256 at 80^2 / 40^2) move 2 - 7 times the bytes their MFMAs are worth in time, and both forms
// above leave them at ~3.3 TB/s: the implicit GEMM pays a DMA issue, a counted wait and a workgroup barrier per K tile around a dozen
// MFMAs, and the register-weight kernel reads every pixel NG times with 32 KB of unique bytes in flight per CU.  tools/ubench/stream_bw
// puts the ceiling for THIS access shape (fragment loads, epilogue-shaped stores) at 4.5 - 5.0 TB/s.  Here the whole weight matrix sits
// in LDS as ready-made MFMA fragments (CT x KS KB, up to 128 KB: one workgroup of eight waves per CU), a wave takes PT x 16 pixels,
// reads their channel runs straight from global memory into the MFMA "B" operand and keeps all CT x 16 outputs of those pixels in
// its accumulators: every input byte is read once, nothing is staged, no barrier after the prologue.  The pixel fragments of K step
// ks are re-requested for the wave's NEXT block as soon as the step's MFMAs have consumed them, so a block's loads fly under the rest
// of the K loop and the whole epilogue of the block before (8 waves x 16 KB in flight per CU).  K order, operand order and epilogue
// are those of conv_igemm_kernel: bit-identical results.
// Measured (128 frames, isolated, autotuner's timing): 256 -> 256 at 40^2 0.054 - 0.056 ms against 0.060 - 0.062 for the best staged tile,
// 256 -> 128 at 80^2 0.149 - 0.158 against 0.159 - 0.166, 128 -> 128 at 80^2 0.112 - 0.116 (NP = 1, PT = 2) against 0.120 - 0.125; a tie on the
// smaller maps -- 5 - 9 %, not the 30 % the access-shape ceiling would allow.  Neither a second fragment set (a block's loads in flight for
// two block times) nor counting the epilogue's stores as allowed-outstanding (loads and stores do retire in issue order here:
// tools/ubench/vmcnt_order, 0 of 3e9) moved it, so what is left is not staging, load latency or store acknowledgement; both removed.
// END TO END the kernel LOSES: one workgroup with up to 132 KB of LDS per CU keeps the ReID queue's workgroups off the CUs it runs on --
// 17.96 / 18.27 k frames/s with it against 18.55 / 18.81 k without (alternating 60-step runs, one box) although the conv stage sum drops
// from 6.40 to 6.35 ms.  The autotuner therefore offers it only under VC_CONV_STREAM=1 (conv_stream_cfg); it stays for the tests and as the
// measured answer to "would reading every byte once with nothing staged reach the copy rate".
template <int CT, int KS, int PT, int NP, int ACT>
__global__ __launch_bounds__(512, 1) void conv1x1_stream_kernel(const ConvP p) {
    constexpr uint32_t OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) uint4 wl[CT * KS * 64 + CT * 4];      // weight fragments [ct][ks][lane], then the bias
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fch = lane >> 4;
    for (int f = wave; f < CT * KS; f += 8) {                 // LDS order [ks][ct]: a K step's fragments are one ds_read offset apart
        const int ks = f / CT, ct = f - ks * CT;
        wl[f * 64 + lane] = *(const uint4*)((const char*)p.w + ((size_t)(ct * 16 + frow) * p.Kw + ks * 32 + fch * 8) * 2);
    }
    float* bl = (float*)(wl + CT * KS * 64);
    for (int i = threadIdx.x; i < CT * 16; i += 512) bl[i] = p.bias[i];
    __syncthreads();
    const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
    const int nblk = (p.M + PT * 16 - 1) / (PT * 16);
    const uint32_t wl_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)wl + lane * 16;
    // The pixel fragments are loaded by hand too (global_load_dwordx4 + counted s_waitcnt): with loads and stores both pending, hipcc's
    // wait insertion falls back to vmcnt(0) in front of the first MFMA of every block, which drains the next block's loads AND this
    // block's stores once per iteration.  Loads return in order: when step ks of a block's first pass starts, the loads younger than its
    // fragments are the (KS - 1 - ks) * PT of the later K steps (requested during the previous block's last pass), so
    // "vmcnt <= (KS - 1 - ks) * PT" means they have landed; the epilogue's stores also sit on the counter and can only make the wait longer.
    // Rows past M are clamped to the last row (read, multiplied, dropped by the epilogue's m < M).
    u32x4v x[PT][KS];
    const char* inb = (const char*)p.in + (size_t)p.in_co * 2 + fch * 16;
    const char* ra[PT];
    auto rows_of = [&](int blk) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) ra[pt] = inb + (size_t)min((blk * PT + pt) * 16 + frow, p.M - 1) * p.in_cs * 2;
    };
#define VC_XLOAD(pt, ks) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(x[pt][ks]) : "v"(ra[pt]), "n"((ks) * 64))
    rows_of(gw);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) VC_XLOAD(pt, ks);
    constexpr int CTP = CT / NP;                          // channel tiles per pass: the accumulators of one pass are CTP x PT x 4 registers
    for (int blk = gw; blk < nblk; blk += nw) {
        rows_of(blk + nw);                            // the next block of this wave (past the end: the last row again)
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            f32x4 acc[CTP][PT];
#pragma unroll
            for (int a = 0; a < CTP; ++a)
#pragma unroll
                for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // The weight fragments are read by hand, one ahead of the MFMAs that use them: left to the compiler, the loop-invariant LDS
            // reads are hoisted out of the block loop (CT x KS x 4 registers: 170 - 550 spills).  lgkmcnt(1) = everything but the newest
            // LDS operation has landed, whatever else the compiler has in flight (LDS returns in order): the wait can only be too strict.
            uint32_t wa = wl_addr + np * CTP * 1024;
            asm volatile("" : "+v"(wa));
            u32x4v wcur, wnext;
            asm volatile("ds_read_b128 %0, %1" : "=v"(wcur) : "v"(wa));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (np == 0) {                        // first pass over this block: its fragments of step ks must have landed
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KS - 1 - ks) * PT));
#pragma unroll
                    for (int b = 0; b < PT; ++b) asm volatile("" : "+v"(x[b][ks]));
                }
#pragma unroll
                for (int a = 0; a < CTP; ++a) {
                    if (a + 1 < CTP) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wnext) : "v"(wa), "n"((a + 1) * 1024));
                    else if (ks + 1 < KS) { wa += CT * 1024; asm volatile("ds_read_b128 %0, %1" : "=v"(wnext) : "v"(wa)); }
                    if (a + 1 < CTP || ks + 1 < KS) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wcur));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wcur));
                    Chunk wf;
                    wf.u = wcur;
#pragma unroll
                    for (int b = 0; b < PT; ++b) {
                        Chunk xa;
                        xa.u = x[b][ks];
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.h, xa.h, acc[a][b], 0, 0, 0);
                    }
                    wcur = wnext;
                }
                if (np == NP - 1) {                   // last pass: this K step's fragments are dead, request the next block's
#pragma unroll
                    for (int b = 0; b < PT; ++b) VC_XLOAD(b, ks);
                }
            }
#pragma unroll
            for (int a = 0; a < CTP; ++a) {
                const float4 b1[1] = {*(const float4*)(bl + (np * CTP + a) * 16 + fch * 4)};
                conv_epilogue_bf16<PT, 1, ACT, RES_NONE>(p, reinterpret_cast<f32x4(&)[1][PT]>(acc[a]), b1, blk * PT * 16, (np * CTP + a) * 16 + fch * 4, frow);
            }
        }
    }
}
#undef VC_XLOAD

bool conv_stream_cfg(int cfg) { return cfg >= 50 && cfg <= 54; }

int conv_k_tile(int prec) { return prec == PREC_F32 ? 32 : prec == PREC_FP8 ? 128 : 64; }    // weights are padded to the widest K tile (KC = 8)

double conv_flops(const ConvP& p) { return 2.0 * (double)p.M * (double)p.Cout * (double)p.K; }

// ---- tile configurations -------------------------------------------------------------------------------------------
// One list drives both the table the autotuner walks and the dispatch switch.  Rings deeper than 2 exist for bf16 only;
// the fp32 parity path maps them to the 2-stage instantiation of the same tile.
#define VC_CONV_CFGS(X)                                                                                          \
    X(0, 256, 32, 4, 1, 4, 2)   X(1, 128, 64, 2, 2, 4, 2)   X(2, 128, 128, 2, 2, 4, 2)  X(3, 64, 64, 2, 2, 4, 2)      \
    X(4, 128, 64, 2, 2, 8, 2)   X(5, 128, 128, 2, 2, 8, 2)  X(6, 64, 64, 2, 2, 8, 2)    X(7, 256, 64, 4, 1, 4, 2)     \
    X(8, 256, 64, 4, 1, 8, 2)   X(9, 256, 128, 2, 2, 4, 2)  X(10, 256, 128, 2, 2, 8, 2) X(11, 64, 128, 1, 4, 4, 2)    \
    X(12, 64, 128, 1, 4, 8, 2)  X(13, 256, 32, 4, 1, 8, 2)                                                          \
    X(14, 64, 64, 2, 2, 4, 4)   X(15, 64, 64, 2, 2, 8, 3)   X(16, 64, 64, 2, 2, 8, 4)   X(17, 128, 64, 2, 2, 4, 4)    \
    X(18, 128, 64, 2, 2, 8, 3)  X(19, 128, 128, 2, 2, 4, 4) X(20, 128, 128, 2, 2, 8, 3) X(21, 64, 128, 1, 4, 4, 4)    \
    X(22, 64, 128, 1, 4, 8, 3)  X(23, 256, 32, 4, 1, 4, 4)  X(24, 256, 64, 4, 1, 4, 4)  X(25, 256, 64, 4, 1, 8, 3)    \
    X(26, 256, 128, 2, 2, 4, 4) X(27, 256, 128, 2, 2, 8, 3)
// halo-staged 3x3 / s1 / p1 (bf16): Y(index, BP, BC, WP, WC, NS)
// 36 - 39: the four waves side by side in pixels (wave tiles 32 x 64 and 64 x 128): half the per-tile tap-address set-up per MFMA
// of the 2 x 2 arrangement -- the narrow layers issue 7 VALU instructions per MFMA, most of them set-up and epilogue (measured:
// 64 -> 64 at 25^2 -8 %, 128 -> 128 at 40^2 -10 %; 256 x 64 and 128 x 128 tiles in this arrangement gained nothing)
#define VC_HALO_CFGS(Y) Y(28, 128, 64, 2, 2, 2) Y(29, 128, 64, 2, 2, 3) Y(30, 128, 128, 2, 2, 2) Y(31, 128, 128, 2, 2, 3) \
                        Y(36, 128, 64, 4, 1, 2) Y(37, 128, 64, 4, 1, 3) Y(38, 256, 128, 4, 1, 2) Y(39, 256, 128, 4, 1, 3)
// 16-wave workgroups on 256 x 256 tiles: half the staged bytes (and LDS-DMA instructions, ~150 issue cycles each) per MFMA of the
// 128 x 128 tile and a 3- or 4-deep ring in 96 / 128 KB (measured per 128 frames: 3x3/s2 128->256 at 80^2 213 -> 169 us, 256->512
// 202 -> 148 us, 1x1 512->512 at 20^2 67 -> 56 us; 256 x 128, 512 x 128 and 512 x 64 tiles with 8 / 16 waves gained nothing)
// 43: the same tile on 128-byte rows (2 x 64 KB): the only 256 x 256 tile of the fp8 path (its K = 128 MFMA step needs KC = 8)
#define VC_CONV_BIG_CFGS(X) X(40, 256, 256, 4, 4, 4, 2) X(41, 256, 256, 4, 4, 4, 3) X(42, 256, 256, 4, 4, 4, 4) X(43, 256, 256, 4, 4, 8, 2)
// halo-staged 3x3 / s2 / p1 (bf16), rectangular tiles of BP pixels: V(index, BP, BC, WP, WC, NS)
// (measured on YOLOv5s, 128 frames: 3-P3 64->128 at 80^2 0.253 -> 0.198 ms with 256-pixel tiles, 18-P4 128->128 0.117 -> 0.096, 5-P4 a tie;
// the 20^2 layers stay on the 256 x 256 implicit GEMM; 2 x 2 waves on 256 pixels never won)
#define VC_S2HALO_CFGS(V) V(44, 128, 128, 2, 2, 2) V(45, 128, 128, 2, 2, 3) V(46, 256, 128, 4, 1, 3) V(47, 128, 128, 2, 2, 4) \
                          V(48, 128, 256, 2, 2, 2) V(49, 256, 128, 4, 1, 2)
struct ConvCfg { int bp, bc, wp, wc, kc, ns; };
#define VC_X(i, bp, bc, wp, wc, kc, ns) {bp, bc, wp, wc, kc, ns},
static const ConvCfg kCfg[] = {VC_CONV_CFGS(VC_X)};
#undef VC_X
// weights-in-registers 1x1 (bf16): Z(index, CT, KS, PT, OCC)
#define VC_DIRECT_CFGS(Z) Z(32, 2, 1, 4, 4) Z(33, 4, 2, 4, 2) Z(34, 4, 2, 2, 3) Z(35, 4, 4, 2, 2)
// weights-in-LDS streaming 1x1 (bf16): S(index, CT, KS, PT, NP): 128 -> 128, 256 -> 256, 256 -> 128, 128 -> 256 channels; NP passes over the
// block's fragments, each for CT / NP channel tiles, keep accumulators + fragments + epilogue inside 256 registers at two waves per SIMD
#define VC_STREAM_CFGS(S) S(50, 8, 4, 4, 2) S(51, 16, 8, 2, 2) S(52, 8, 8, 2, 1) S(53, 16, 4, 2, 2) S(54, 8, 4, 2, 1)
// deep rings on the small tiles (round 6): a launch of 28 workgroups walking 36 K steps is bound by the latency of its LDS-DMA loads (~1.2 us from
// L2 / HBM on an otherwise idle chip) divided by the tiles in flight; six or eight stages instead of three
#define VC_CONV_DEEP_CFGS(X) X(60, 64, 64, 2, 2, 8, 6) X(61, 64, 64, 2, 2, 8, 8) X(62, 128, 64, 2, 2, 8, 6) X(63, 64, 128, 1, 4, 8, 6)
// split-K instances of the implicit GEMM (bf16, round 6): K(index, BP, BC, WP, WC, KC, NS); offered when the tiles alone cannot fill the chip
#define VC_SK_CFGS(K) K(56, 64, 64, 2, 2, 8, 3) K(57, 64, 64, 2, 2, 8, 4) K(58, 128, 64, 2, 2, 8, 3) K(59, 64, 128, 1, 4, 8, 3)
// paired 8-wave workgroups, two per CU (round 6, conv_igemm_kernel<..., OCC = 2>): P(index, BP, BC, WP, WC, KC, NS)
#define VC_PAIR_CFGS(P) P(64, 256, 128, 4, 2, 4, 3) P(65, 128, 256, 2, 4, 4, 3) P(66, 256, 128, 4, 2, 4, 2)
// 67 - 68: two 4-wave workgroups per CU with 128 x 64 wave tiles (12 fragment reads per 32 MFMAs: 96 B / clk of LDS reads where the 64 x 64 wave
// tile asks for the LDS's whole 128 B / clk), 256 registers per wave
#define VC_PAIR4_CFGS(P) P(67, 256, 128, 2, 2, 4, 3) P(68, 256, 128, 2, 2, 4, 2)
// weights-in-registers 1x1 of the fp8 path (conv1x1_direct_fp8_kernel): F(index, CT, KS, PT, OCC); K <= 128 KS
#define VC_DIRECT8_CFGS(F) F(69, 4, 1, 2, 2) F(70, 4, 2, 2, 2) F(71, 4, 1, 4, 2) F(72, 8, 1, 2, 2)
int conv_num_cfgs() { return (int)(sizeof(kCfg) / sizeof(kCfg[0])) + 4 + 4 + 4 + 4 + 6 + 5 + 1 + 4 + 4 + 3 + 2 + 4; }   // + the halo-staged 3x3 (28-31, 36-39), the direct 1x1 (32-35), the 16-wave 256 x 256 tiles (40-43), the halo-staged 3x3/s2 (44-49), the streaming 1x1 (50-54), conv3x3_halo_v2_kernel (55), the split-K tiles (56-59), the deep rings (60-63) the paired workgroups (64-68) and the fp8 direct 1x1 (69-72)

// resident workgroups of one kernel instantiation on the whole device (occupancy x CUs), queried once
static int device_cus() {
    static const int n = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        return cus;
    }();
    return n;
}

template <class K>
static int resident_workgroups(K kernel, int threads = 256) {
    int per_cu = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu * cus;
}

// Grid of a persistent launch.  Workgroup b walks tiles b, b + G, ...: the launch lasts ceil(tiles / G) tile times.  The old rule -- every slot
// but a reserve of 64 for the other streams' kernels (never fewer than 256 slots) -- can cost a whole extra round on the configurations with
// several workgroups per CU (3200 tiles on 448 of 512 slots: eight rounds where seven do) and always occupies every slot it may, whatever the
// tile count.  Rounds first: the fewest rounds the chip allows (the reserve is given up only when that saves a round), then the smallest
// grid that still finishes in that many rounds (800 tiles in four rounds: 200 workgroups, not 256 -- the slots that are not needed stay free
// for the ReID queue and the tracker), a multiple of 8 for the XCD-aware tile order.  Measured + 0.5 % end to end, two alternations.
// VC_CONV_BALANCED=0: the old rule (A/B switch).
static int persistent_grid(int tiles, int slots_hw, int reserve, int slots_override) {
    static const bool balanced = !(getenv("VC_CONV_BALANCED") && atoi(getenv("VC_CONV_BALANCED")) == 0);
    if (slots_override > 0) return tiles > slots_override ? std::max(8, slots_override / 8 * 8) : tiles;
    const int cap = std::max(8, std::max(256, slots_hw - reserve) / 8 * 8);
    if (tiles <= cap) return tiles;
    if (!balanced) return cap;
    const int full = std::max(cap, slots_hw / 8 * 8);
    const int r_cap = (tiles + cap - 1) / cap, r_full = (tiles + full - 1) / full;
    const int rounds = std::min(r_cap, r_full);
    const int g = ((tiles + rounds - 1) / rounds + 7) / 8 * 8;
    return std::min(g, full);
}

template <int BP, int BC, int WP, int WC, int KC, int NS, int OCC = 1>
static int launch_one(ConvP p, hipStream_t s) {
    if (OCC != 1 && (p.prec != PREC_BF16 || p.in_up)) return VC_ERR_ARG;      // quietly: the paired-workgroup tiles exist for plain bf16 only
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    const int bk = KC * (p.prec == PREC_F32 ? 4 : p.prec == PREC_FP8 ? 16 : 8);
    p.Kw = p.Kp;                              // weight row stride as packed
    p.Kp = (p.K + bk - 1) / bk * bk;          // K-loop extent: only the tiles that hold real taps
    p.ntiles = tiles;
    static const int dyn_lds = getenv("VC_CONV_DYN_LDS") ? atoi(getenv("VC_CONV_DYN_LDS")) : 0;   // diagnostics: caps workgroups per CU
    static const bool persist = !(getenv("VC_CONV_PERSIST") && atoi(getenv("VC_CONV_PERSIST")) == 0);
    const int slots_override = p.slots;                                                              // tests: force long tile walks (ConvP::slots)
    // A persistent grid that fills every workgroup slot of the chip leaves no room for the kernels of the other streams (ReID next to the
    // detector, the tracker walk), which then wait for a conv launch to end: 64 slots are left free (round 2, 128-frame steps:
    // 0 / 32 / 64 / 96 / 128 free slots = 14.9 / 15.1 / 15.6 / 15.6 / 15.4 k frames/s; 256 free slots cost 9 % of conv time).
    static const int slots_reserve = getenv("VC_CONV_RESERVE") ? atoi(getenv("VC_CONV_RESERVE")) : 64;
    if (p.in_up && p.prec != PREC_BF16) return VC_ERR_ARG;        // (conv_check refuses it with a message)
    bool handled = false;
    if constexpr (OCC == 1) {
    handled = p.prec == PREC_F32 || p.prec == PREC_FP8 || p.in_up;
    if (p.prec == PREC_F32) {
        static const int slots_hw = resident_workgroups(conv_igemm_kernel<BP, BC, WP, WC, KC, 2, PREC_F32>, WP * WC * 64);
        const int grid = (persist && !dyn_lds) ? persistent_grid(tiles, slots_hw, slots_reserve, slots_override) : tiles;
        launch_timed(p, conv_igemm_kernel<BP, BC, WP, WC, KC, 2, PREC_F32>, dim3(grid), dim3(WP * WC * 64), dyn_lds, s, p);
    } else if (p.prec == PREC_FP8) {
        if constexpr (KC == 8 && (BP / WP / 16) % 2 == 0) {        // the fp8 epilogue pairs pixel tiles (PT even)
            static const int slots_hw = resident_workgroups(conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_FP8>, WP * WC * 64);
            const int grid = (persist && !dyn_lds) ? persistent_grid(tiles, slots_hw, slots_reserve, slots_override) : tiles;
            launch_timed(p, conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_FP8>, dim3(grid), dim3(WP * WC * 64), dyn_lds, s, p);
        } else {
            return VC_ERR_ARG;                                       // quietly: the autotuner skips it (fp8 runs on the 128-byte-row tiles only)
        }
    } else if (p.in_up) {
        // the upsample fold-in is instantiated for the tiles the wide pointwise layers use (256 x 256 on 16 waves, 128 / 256 x 128 on 2 x 2)
        constexpr bool UP_OK = (BP == 256 && BC == 256 && KC == 4 && NS <= 4) ||
                               (WP == 2 && WC == 2 && BC == 128 && (BP == 128 || BP == 256) && (KC == 8 || (KC == 4 && NS == 2 && BP == 128)));
        if constexpr (UP_OK) {
            static const int slots_hw = resident_workgroups(conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_BF16, true>, WP * WC * 64);
            const int grid = (persist && !dyn_lds) ? persistent_grid(tiles, slots_hw, slots_reserve, slots_override) : tiles;
            launch_timed(p, conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_BF16, true>, dim3(grid), dim3(WP * WC * 64), dyn_lds, s, p);
        } else {
            return VC_ERR_ARG;                                       // quietly: the autotuner skips it
        }
    }
    }
    if (!handled) {
        static const int slots_hw = resident_workgroups(conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_BF16, false, false, OCC>, WP * WC * 64);
        const int grid = (persist && !dyn_lds) ? persistent_grid(tiles, slots_hw, slots_reserve, slots_override) : tiles;
        launch_timed(p, conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_BF16, false, false, OCC>, dim3(grid), dim3(WP * WC * 64), dyn_lds, s, p);
    }
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// Split-K workspace: fp32 partial sums + one ticket per output tile, one set per stream (launches of one stream are ordered; the detector's and
// the ReID net's streams run side by side).  Tickets are zeroed once: the kernel leaves them at zero.
struct SkWorkspace { float* ws = nullptr; int* tickets = nullptr; };
static constexpr size_t SK_WS_BYTES = 32u << 20;
static constexpr int SK_MAX_TILES = 16384;
static int sk_workspace(hipStream_t s, SkWorkspace* out) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SkWorkspace> table;
    int dev = 0;
    VC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    SkWorkspace& w = table[{dev, s}];
    if (!w.ws) {
        static const bool uncached = !(getenv("VC_SK_UNCACHED") && atoi(getenv("VC_SK_UNCACHED")) == 0);      // (A/B switch)
        if (uncached) {       // memory no XCD's L2 keeps a copy of: the hand-over between workgroups of different XCDs cannot meet a stale line
            VC_HIP(hipExtMallocWithFlags((void**)&w.ws, SK_WS_BYTES, hipDeviceMallocUncached));
            VC_HIP(hipExtMallocWithFlags((void**)&w.tickets, sizeof(int) * SK_MAX_TILES, hipDeviceMallocUncached));
        } else {
            VC_HIP(hipMalloc((void**)&w.ws, SK_WS_BYTES));
            VC_HIP(hipMalloc((void**)&w.tickets, sizeof(int) * SK_MAX_TILES));
        }
        VC_HIP(hipMemset(w.tickets, 0, sizeof(int) * SK_MAX_TILES));
    }
    *out = w;
    return VC_OK;
}

template <int BP, int BC, int WP, int WC, int KC, int NS>
static int launch_one_sk(ConvP p, hipStream_t s) {
    static const bool enabled = !(getenv("VC_CONV_SK") && atoi(getenv("VC_CONV_SK")) == 0);     // (A/B switch)
    if (!enabled || p.prec != PREC_BF16 || p.in_up || p.m_dev) return VC_ERR_ARG;      // quietly: the autotuner skips it
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    const int bk = KC * 8;
    p.Kw = p.Kp;
    p.Kp = (p.K + bk - 1) / bk * bk;
    p.ntiles = tiles;
    const int nk = p.Kp / bk;
    static const int slots_hw = resident_workgroups(conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_BF16, false, true>, WP * WC * 64);
    // worth it only when the tiles leave most of the chip idle and every split still walks a few K tiles
    constexpr size_t item_bytes = (size_t)BP * BC * 4;
    // (measured, tools/sk_time.py: a K step of a 64 x 64 tile costs ~0.43 us -- LDS-DMA issue, not latency: rings of 6 / 8 stages, configurations
    // 60 - 63, change nothing -- a launch ~6 us whatever its size, and the last arrival reads the KS partial sums one memory latency after the
    // other: splits of at least four K steps, at most eight of them)
    int ks = std::min(std::min(slots_hw / std::max(tiles, 1), nk / 4), 8);
    ks = std::min<long>(ks, (long)(SK_WS_BYTES / (item_bytes * (size_t)std::max(tiles, 1))));
    if (ks < 2 || tiles > SK_MAX_TILES) return VC_ERR_ARG;
    SkWorkspace w;
    VC_TRY(sk_workspace(s, &w));
    p.ksplit = ks; p.sk_ws = w.ws; p.sk_tickets = w.tickets;
    launch_timed(p, conv_igemm_kernel<BP, BC, WP, WC, KC, NS, PREC_BF16, false, true>, dim3(tiles * ks), dim3(WP * WC * 64), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

static int conv_heuristic(const ConvP& p) {
    if (p.prec == PREC_FP8) {                  // 128-byte-row tiles only (KC = 8, even pixel tiles per wave)
        if (p.Cout <= 64) return 4;
        const long t = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
        return t >= 512 ? 5 : 6;
    }
    if (p.in_up) return 2;                     // the upsample fold-in is instantiated for a subset of the tiles (launch_one): 128 x 128 is one of them
    // narrow layers get tall pixel tiles; late (small-M) layers get small tiles so the grid still covers 256 CUs
    if (p.Cout <= 32) return 0;
    if (p.Cout <= 64) return 1;
    const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    return t128 >= 512 ? 2 : 3;
}

template <int BP, int BC, int WP, int WC, int NS>
static int launch_halo(ConvP p, hipStream_t s) {
    if (!halo_applicable(p, BP)) return VC_ERR_ARG;                   // quietly: the autotuner skips it, launch_conv falls back
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    p.Kw = p.Kp;
    const int px = halo_patch_pixels(p, BP);
    if (px <= 4 * 64 - 1) launch_timed(p, conv3x3_halo_kernel<BP, BC, WP, WC, NS, 4>, dim3(tiles), dim3(256), 0, s, p);
    else if (px <= 7 * 64 - 1) launch_timed(p, conv3x3_halo_kernel<BP, BC, WP, WC, NS, 7>, dim3(tiles), dim3(256), 0, s, p);
    else launch_timed(p, conv3x3_halo_kernel<BP, BC, WP, WC, NS, 11>, dim3(tiles), dim3(256), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// tile rectangle of the stride-2 halo kernel: the th x tw (th * tw <= 128) whose parity classes ((th + 1) x (tw + 1) pixels) fit the patch
// buffer and that covers the map with the fewest tiles (then the smallest patch): 8 x 16 at 80 columns, 16 x 8 at 40, 25 x 5 at 20
// (the row space is batch * Ho deep)
static bool s2halo_geom(const ConvP& p, int bp, int* th_out, int* tw_out) {
    const long G = (long)p.B * p.Ho;
    long best = -1;
    for (int tw = 1; tw <= std::min(p.Wo, 254); ++tw) {
        const int th = (int)std::min<long>(std::min(bp / tw, 254), G);
        const int cls = (th + 1) * (tw + 1);
        if (th < 1 || cls > bp / 128 * 5 * 32 - 1) continue;
        const long tiles = (long)((p.Wo + tw - 1) / tw) * ((G + th - 1) / th);
        const long cost = tiles * 4096 + cls;
        if (best < 0 || cost < best) { best = cost; *th_out = th; *tw_out = tw; }
    }
    return best >= 0;
}
static bool s2halo_applicable(const ConvP& p) {
    if (p.prec != PREC_BF16 || p.kh != 3 || p.kw != 3 || p.sh != 2 || p.sw != 2 || p.ph != 1 || p.pw != 1) return false;
    if (p.Cin % 64 != 0 || p.in_cs % 8 != 0 || p.in_co % 8 != 0 || p.H != 2 * p.Ho || p.W != 2 * p.Wo) return false;
    if (p.out_f32 || p.res_mode != RES_NONE || p.split != 0 || p.m_dev || p.Cout % 8 != 0 || p.out_cs % 8 != 0 || p.out_co % 8 != 0) return false;
    return p.act == ACT_SILU || p.act == ACT_RELU || p.act == ACT_NONE;
}
template <int BP, int BC, int WP, int WC, int NS>
static int launch_s2halo(ConvP p, hipStream_t s) {
    static const bool enabled = !(getenv("VC_CONV_S2HALO") && atoi(getenv("VC_CONV_S2HALO")) == 0);   // A/B switch
    if (!enabled || !s2halo_applicable(p) || !s2halo_geom(p, BP, &p.s2_th, &p.s2_tw)) return VC_ERR_ARG;  // quietly, like launch_halo
    const long G = (long)p.B * p.Ho;
    const long tiles = (long)((p.Wo + p.s2_tw - 1) / p.s2_tw) * ((G + p.s2_th - 1) / p.s2_th) * ((p.Cout + BC - 1) / BC);
    p.Kw = p.Kp;
    p.ntiles = (int)tiles;
    launch_timed(p, conv3x3s2_halo_kernel<BP, BC, WP, WC, NS>, dim3((unsigned)tiles), dim3(256), 0, s, p, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// p: a 3x3 / s2 conv with 128 output channels + SiLU; q: the 1x1 / s1 conv (128 -> 128 channels, SiLU, one or two destinations) that reads p's
// output -- and is its ONLY reader (the caller's knowledge: p's output is not written).  s2_pointwise_stage above.
bool s2halo_pw_applicable(const ConvP& p, const ConvP& q) {
    int th, tw;
    if (!s2halo_applicable(p) || p.Cout != 128 || p.act != ACT_SILU || !s2halo_geom(p, 256, &th, &tw)) return false;
    if (q.prec != PREC_BF16 || q.kh != 1 || q.kw != 1 || q.sh != 1 || q.sw != 1 || q.ph != 0 || q.pw != 0 || q.Cin != 128 || q.K != 128 || q.Cout != 128) return false;
    if (q.act != ACT_SILU || q.res_mode != RES_NONE || q.out_f32 || q.m_dev || q.in_up || q.Kp < 128) return false;
    if (q.in != p.out || q.in_co != p.out_co || q.in_cs != p.out_cs || q.B != p.B || q.H != p.Ho || q.W != p.Wo || q.M != p.M) return false;
    if (q.out_cs % 8 != 0 || q.out_co % 8 != 0 || (q.split != 0 && (q.split % 8 != 0 || q.out2_cs % 8 != 0 || q.out2_co % 8 != 0))) return false;
    return true;
}
int launch_s2halo_pw(ConvP p, ConvP q, hipStream_t s) {
    if (!s2halo_pw_applicable(p, q) || !s2halo_geom(p, 256, &p.s2_th, &p.s2_tw)) return VC_ERR_ARG;
    const long G = (long)p.B * p.Ho;
    const long tiles = (long)((p.Wo + p.s2_tw - 1) / p.s2_tw) * ((G + p.s2_th - 1) / p.s2_th);
    p.Kw = p.Kp; q.Kw = q.Kp;
    p.ntiles = (int)tiles;
    static const bool also_store = getenv("VC_S2PW_STORE") && atoi(getenv("VC_S2PW_STORE")) != 0;   // diagnostics
    if (also_store) p.ablate = 8;
    launch_timed(p, conv3x3s2_halo_kernel<256, 128, 4, 1, 2, true>, dim3((unsigned)tiles), dim3(256), 0, s, p, q);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

static bool direct1x1_applicable(const ConvP& p, int ct, int ks) {
    if (p.prec != PREC_BF16 || p.kh != 1 || p.kw != 1 || p.sh != 1 || p.sw != 1 || p.ph != 0 || p.pw != 0) return false;
    if (p.Cin != ks * 32 || p.K != p.Cin || p.Ho != p.H || p.Wo != p.W || p.in_cs % 8 != 0 || p.in_co % 8 != 0 || p.in_up) return false;
    // the 16-byte-store epilogue only (conv_epilogue_bf16's preconditions), SiLU or no activation, no residual
    if (p.out_f32 || p.res_mode != RES_NONE || (p.act != ACT_SILU && p.act != ACT_NONE) || p.out_cs % 8 != 0 || p.out_co % 8 != 0) return false;
    if (p.split != 0 && (p.split % 8 != 0 || p.out2_cs % 8 != 0 || p.out2_co % 8 != 0)) return false;
    const int ng = p.Cout / (ct * 16);
    return p.Cout % (ct * 16) == 0 && (ng == 1 || ng == 2 || ng == 4);
}

template <int CT, int KS, int PT, int OCC>
static int launch_direct1x1(ConvP p, hipStream_t s) {
    static const bool enabled = !(getenv("VC_CONV_DIRECT") && atoi(getenv("VC_CONV_DIRECT")) == 0);   // A/B switch
    if (!enabled || !direct1x1_applicable(p, CT, KS)) return VC_ERR_ARG;          // quietly, like launch_halo
    p.Kw = p.Kp;
    const int ng = p.Cout / (CT * 16);
    const int nblk = (p.M + PT * 16 - 1) / (PT * 16);
    const int need = (nblk * ng + 3) / 4;
    static const int slots_hw = resident_workgroups(conv1x1_direct_kernel<CT, KS, PT, OCC, ACT_SILU>);
    static const int slots_reserve = getenv("VC_CONV_RESERVE") ? atoi(getenv("VC_CONV_RESERVE")) : 64;
    const int slots_override = p.slots;
    const int slots = slots_override > 0 ? slots_override : std::max(256, slots_hw - slots_reserve);
    p.ntiles = nblk * ng;
    if (p.act == ACT_SILU) launch_timed(p, conv1x1_direct_kernel<CT, KS, PT, OCC, ACT_SILU>, dim3(std::min(need, slots)), dim3(256), 0, s, p);
    else launch_timed(p, conv1x1_direct_kernel<CT, KS, PT, OCC, ACT_NONE>, dim3(std::min(need, slots)), dim3(256), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}


static bool direct8_applicable(const ConvP& p, int ct, int ks) {
    if (p.prec != PREC_FP8 || p.kh != 1 || p.kw != 1 || p.sh != 1 || p.sw != 1 || p.ph != 0 || p.pw != 0) return false;
    if (p.K != p.Cin || p.Cin > ks * 128 || p.Cin <= (ks - 1) * 128 || p.Cin % 64 != 0 || (p.Cin % 128 != 0 && p.Cin != 64)) return false;
    if (p.Ho != p.H || p.Wo != p.W || p.in_cs % 16 != 0 || p.in_co % 16 != 0 || p.in_up || p.m_dev || !p.scale || p.Kw < ks * 128) return false;
    const int ng = (p.Cout + ct * 16 - 1) / (ct * 16);
    return ng == 1 || ng == 2 || ng == 4;                         // (the channel tail of a group is masked by the epilogue; its weight rows are zero padding)
}

template <int CT, int KS, int PT, int OCC>
static int launch_direct8(ConvP p, hipStream_t s) {
    static const bool enabled = !(getenv("VC_CONV_DIRECT8") && atoi(getenv("VC_CONV_DIRECT8")) == 0);   // A/B switch
    p.Kw = p.Kp;
    if (!enabled || !direct8_applicable(p, CT, KS)) return VC_ERR_ARG;             // quietly, like launch_halo
    const int ng = (p.Cout + CT * 16 - 1) / (CT * 16);
    const int nblk = (p.M + PT * 16 - 1) / (PT * 16);
    const int need = (nblk * ng + 3) / 4;
    static const int slots_hw = resident_workgroups(conv1x1_direct_fp8_kernel<CT, KS, PT, OCC>);
    static const int slots_reserve = getenv("VC_CONV_RESERVE") ? atoi(getenv("VC_CONV_RESERVE")) : 64;
    const int slots = p.slots > 0 ? p.slots : std::max(256, slots_hw - slots_reserve);
    p.ntiles = nblk * ng;
    launch_timed(p, conv1x1_direct_fp8_kernel<CT, KS, PT, OCC>, dim3(std::min(need, slots)), dim3(256), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

template <int CT, int KS, int PT, int NP>
static int launch_stream1x1(ConvP p, hipStream_t s) {
    if (!direct1x1_applicable(p, CT, KS) || p.Cout != CT * 16) return VC_ERR_ARG;                     // quietly, like launch_halo
    p.Kw = p.Kp;
    const int nblk = (p.M + PT * 16 - 1) / (PT * 16);
    p.ntiles = nblk;
    const int grid = std::max(1, std::min((nblk + 7) / 8, p.slots > 0 ? std::max(1, p.slots / 8) : device_cus()));   // persistent, one workgroup per CU
    if (p.act == ACT_SILU) launch_timed(p, conv1x1_stream_kernel<CT, KS, PT, NP, ACT_SILU>, dim3(grid), dim3(512), 0, s, p);
    else launch_timed(p, conv1x1_stream_kernel<CT, KS, PT, NP, ACT_NONE>, dim3(grid), dim3(512), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_conv_cfg(const ConvP& p, int cfg, hipStream_t s) {
    if (cfg < 0 || cfg >= conv_num_cfgs()) cfg = conv_heuristic(p);
    switch (cfg) {
#define VC_Y(i, bp, bc, wp, wc, ns) case i: return launch_halo<bp, bc, wp, wc, ns>(p, s);
        VC_HALO_CFGS(VC_Y)
#undef VC_Y
        case 55: return launch_halo_v2_cfg(p, cfg, s);        // conv_halo_v2.hip
#define VC_V(i, bp, bc, wp, wc, ns) case i: return launch_s2halo<bp, bc, wp, wc, ns>(p, s);
        VC_S2HALO_CFGS(VC_V)
#undef VC_V
#define VC_Z(i, ct, ks, pt, occ) case i: return launch_direct1x1<ct, ks, pt, occ>(p, s);
        VC_DIRECT_CFGS(VC_Z)
#undef VC_Z
#define VC_S(i, ct, ks, pt, np) case i: return launch_stream1x1<ct, ks, pt, np>(p, s);
        VC_STREAM_CFGS(VC_S)
#undef VC_S
#define VC_X(i, bp, bc, wp, wc, kc, ns) case i: return launch_one<bp, bc, wp, wc, kc, ns>(p, s);
        VC_CONV_CFGS(VC_X)
        VC_CONV_BIG_CFGS(VC_X)
        VC_CONV_DEEP_CFGS(VC_X)
#undef VC_X
#define VC_K(i, bp, bc, wp, wc, kc, ns) case i: return launch_one_sk<bp, bc, wp, wc, kc, ns>(p, s);
        VC_SK_CFGS(VC_K)
#undef VC_K
#define VC_P(i, bp, bc, wp, wc, kc, ns) case i: return launch_one<bp, bc, wp, wc, kc, ns, 2>(p, s);
        VC_PAIR_CFGS(VC_P)
        VC_PAIR4_CFGS(VC_P)
#undef VC_P
#define VC_F(i, ct, ks, pt, occ) case i: return launch_direct8<ct, ks, pt, occ>(p, s);
        VC_DIRECT8_CFGS(VC_F)
#undef VC_F
    }
    return VC_ERR_ARG;
}

int conv_check(const ConvP& p);

int launch_conv(const ConvP& p, hipStream_t s) {
    VC_TRY(conv_check(p));
    const int rc = launch_conv_cfg(p, p.cfg, s);
    // A tuned choice is keyed by the power-of-two bucket of M, but some families' applicability depends on the exact batch (v2_applicable:
    // whole rows per tile; the upsample fold-in's tile subset): a launcher refuses before it launches anything, and the heuristic's
    // implicit-GEMM tile takes every shape conv_check admits.
    if (rc == VC_ERR_ARG && p.cfg >= 0 && p.cfg < conv_num_cfgs() && !getenv("VC_CONV_STRICT")) return launch_conv_cfg(p, -1, s);   // (VC_CONV_STRICT: the tile-configuration tests want the refusal)
    return rc;
}

int conv_check(const ConvP& p) {
    const int ch = p.prec == PREC_F32 ? 4 : p.prec == PREC_FP8 ? 16 : 8;
    if (p.prec == PREC_FP8)
        VC_CHECK(p.scale && p.Cout % 8 == 0 && p.out_cs % 8 == 0 && p.out_co % 8 == 0 && (p.split == 0 || (p.split % 8 == 0 && p.out2_cs % 8 == 0 && p.out2_co % 8 == 0)) &&
                 (p.res_mode == RES_NONE || (p.res_cs % 4 == 0 && p.res_co % 4 == 0)), VC_ERR_ARG, "conv fp8: channel scales / 8-channel alignment of the outputs");
    VC_CHECK(p.Cin % ch == 0 && p.in_cs % ch == 0 && p.in_co % ch == 0, VC_ERR_ARG,
             "conv: input channels/stride/offset (%d,%d,%d) must be multiples of %d", p.Cin, p.in_cs, p.in_co, ch);
    VC_CHECK(p.out_cs % 4 == 0 && p.out_co % 4 == 0, VC_ERR_ARG, "conv: output stride/offset must be multiples of 4");
    VC_CHECK(p.split == 0 || (p.split % 4 == 0 && p.out2 && p.out2_cs % 4 == 0 && p.out2_co % 4 == 0 && p.res_mode == RES_NONE), VC_ERR_ARG,
             "conv: bad split destination");
    VC_CHECK(p.res_mode == RES_NONE || (p.res_cs % 4 == 0 && p.res_co % 4 == 0), VC_ERR_ARG, "conv: residual alignment");
    VC_CHECK(p.Kp % conv_k_tile(p.prec) == 0 && p.Kp >= p.K, VC_ERR_ARG, "conv: bad K padding %d/%d", p.K, p.Kp);
    VC_CHECK(p.M > 0 && p.Cout > 0, VC_ERR_ARG, "conv: empty problem");
    VC_CHECK(!p.in_up || (p.prec == PREC_BF16 && p.kh == 1 && p.kw == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.H % 2 == 0 && p.W % 2 == 0 &&
                          p.up_C > 0 && p.up_C < p.Cin && p.up_C % 64 == 0 && p.Cin % 64 == 0 && p.up_cs % 8 == 0 && p.up_co % 8 == 0 && !p.m_dev), VC_ERR_ARG,
             "conv: the upsample fold-in needs a bf16 pointwise conv on an even-sized map with 64-channel-aligned halves");
    VC_CHECK((size_t)p.B * p.H * p.W * p.in_cs * elem_size(p.prec) < (1ull << 31), VC_ERR_CAPACITY, "conv: input tensor exceeds the 2 GiB buffer descriptor");
    VC_CHECK((size_t)((p.Cout + 127) / 128 * 128) * p.Kp * elem_size(p.prec) < (1ull << 31), VC_ERR_CAPACITY, "conv: weights exceed 2 GiB");
    VC_CHECK(p.kh * p.kw <= 40, VC_ERR_ARG, "conv: at most 40 taps (validity mask is 64 bits incl. K padding)");
    VC_CHECK(p.M < (1 << 24), VC_ERR_CAPACITY, "conv: more than 2^24 output pixels in one launch");
    VC_CHECK((size_t)p.M * std::max(p.out_cs, std::max(p.res_cs, p.out2_cs)) * ((p.prec == PREC_F32 || p.out_f32) ? 4 : (p.prec == PREC_FP8 && !p.out_bf16) ? 1 : 2) < (1ull << 31),
             VC_ERR_CAPACITY, "conv: output tensor exceeds 2 GiB (32-bit buffer offsets)");
    return VC_OK;
}

}  // namespace vc
