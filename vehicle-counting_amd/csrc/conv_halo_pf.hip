// Software-pipelined halo-staged 3x3 / stride 1 / pad 1 convolution (bf16), tile configurations 55 - 60 of conv_igemm.hip's table.
// Same rows of the reference as conv_igemm.hip: the 3x3 convolutions of ultralytics/yolov5 v6.0 `Bottleneck` blocks that
// /root/reference/networks/yolo.py:70 executes and of the DeepSORT appearance net's BasicBlocks,
// /root/reference/networks/deepsort/deep/model.py:5-98 (row B5).
#include <algorithm>
#include <cstdlib>

#include "vc_common.h"
#include "conv_device.h"

namespace vc {

// ---- halo-staged 3x3 / stride 1 / pad 1, software-pipelined, 4 or 8 waves (bf16) -------------------------------------------------
// conv3x3_halo_kernel runs one workgroup of four waves per CU on the wide layers (its two patch buffers take 57 - 114 KB): ONE wave per
// SIMD, and every step of that wave is a dependent chain -- DMA issue (an LDS-DMA instruction holds the wave's issue for 60 - 150 cycles,
// MI355X_MICROARCH.md), fragment reads, a full LDS latency, 16 - 32 MFMAs, counted wait, barrier -- with nothing else on the SIMD to
// cover it.  Same data flow here (patch per 32-channel slice staged once, taps read it at shifted addresses, weights through the LDS-DMA
// ring, identical MFMA order: bit-identical results), two changes:
//   * the fragments of step kt + 1 are read into a second register set BEFORE the MFMAs of step kt are issued (the ring is one stage
//     deeper in what must have landed: stage kt + 2 at the end of step kt), so the MFMAs start right behind the barrier and the LDS
//     latency runs under them;
//   * WP x WC may be 8 waves (two per SIMD, <= 256 registers each): while one wave of a SIMD issues its DMA / fragment reads the other
//     one's MFMAs keep the matrix pipe busy.
// nslices is walked two slices (18 steps) per loop iteration so that the two register sets alternate at compile time.
template <int BP, int BC, int WP, int WC, int XI>
__global__ __launch_bounds__(WP * WC * 64, 1) void conv3x3_halo_pf_kernel(const ConvP p) {
    constexpr int NW = WP * WC;
    constexpr int KC = 4, ES = 2, BK = 32;
    constexpr int NS = 6;                          // ring depth: divides the 18 steps of a loop iteration, so every ring address is an immediate
    constexpr int PASS = NW * 16;                  // weight rows covered by one DMA instruction of all waves (16 per wave)
    constexpr int WI = (BC + PASS - 1) / PASS;
    constexpr int WROWS = WI * PASS;
    constexpr int WTP = BP / WP, WTC = BC / WC, PT = WTP / 16, CT = WTC / 16;
    constexpr int ZP = XI * NW * 16 - 1;           // index of the zero pixel: last pixel of a patch buffer, never reached by a patch
    constexpr int XCH = XI * NW * 64;              // 16-byte chunks per patch buffer
    constexpr int RCH = NS * WROWS * KC;           // chunks of the weight ring (it comes first: its stage offsets and the patch parity
    constexpr uint32_t OOB = 0x80000000u;          // offset both fit the 16-bit immediate of ds_read)
    constexpr uint32_t XBYTES = XCH * 16, WSTAGE = WROWS * KC * 16;
    static_assert((NW == 4 || NW == 8) && WTP % 16 == 0 && WTC % 16 == 0 && PT % 2 == 0, "tile shape");
    static_assert((NS - 3) * WI + XI <= 63, "counted vmcnt");
    static_assert(PT + CT <= 15, "counted lgkmcnt");
    static_assert(XBYTES < 65536 && NS * WSTAGE < 65536, "ds_read immediates");
    __shared__ __attribute__((aligned(16))) uint4 lds[RCH + 2 * XCH];
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

    const int g0 = m0 / W;
    const int g1 = (min(m0 + BP, p.M) - 1) / W;
    const int gp0 = (g0 - 1) * W;
    const int npix = (g1 - g0 + 3) * W;            // <= ZP, checked by the launcher

    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    uint32_t xsrc[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int e = (i * NW + wave) * 64 + lane;
        const int pp = e >> 2, cpos = e & 3;
        const int chunk = cpos ^ ((pp >> 1) & 2);
        const int gp = gp0 + pp;
        xsrc[i] = (pp < npix && gp >= 0) ? (uint32_t)((gp * p.in_cs + p.in_co) * ES + chunk * 16) : OOB;
    }
    const int prow = wave * 16 + (lane >> 2);
    const int wchunk = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
    uint32_t woff[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) woff[i] = (uint32_t)(((n0 + prow + PASS * i) * p.Kw + wchunk * 8) * ES);

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)&lds[0];
    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;
    uint32_t xaddr[PT][9];                         // LDS byte address of this lane's fragment of every tap in patch buffer 0
    {
        const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int m = m0 + wp * WTP + i * 16 + frow;
            const bool ok = m < p.M;
            const int mm = ok ? m : m0;
            int g = (int)((float)mm * inv_w);
            g -= (g * W > mm) ? 1 : 0;
            g += ((g + 1) * W <= mm) ? 1 : 0;
            const int x = mm - g * W;
            int b = (int)((float)g * inv_h);
            b -= (b * H > g) ? 1 : 0;
            b += ((b + 1) * H <= g) ? 1 : 0;
            const int y = g - b * H;
            const int pc = mm - gp0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool valid = ok && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const int px = valid ? pc + dy * W + dx : ZP;
                xaddr[i][t] = lds_base + (uint32_t)((RCH + px * 4 + (fch ^ ((px >> 1) & 2))) * 16);      // conv3x3_halo_kernel's conflict-free slot
            }
        }
    }
    uint32_t wfrag[CT];                            // ... and of every channel tile in ring stage 0
#pragma unroll
    for (int i = 0; i < CT; ++i) wfrag[i] = lds_base + 16 * lds_slot<4>(wc * WTC + i * 16 + frow, fch);

    f32x4 acc[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_inputs_settle<CT * PT>(&acc[0][0]);
    u32x2r rpre[PT][CT];
    const bool have_res = PT * CT <= 8 && p.res_mode != RES_NONE && conv_epilogue_fast_bf16<PT, CT>(p) &&
                          ((p.act == ACT_SILU && p.res_mode == RES_AFTER_ACT) || (p.act == ACT_RELU && p.res_mode == RES_BEFORE_ACT));
    if (have_res) conv_residual_fetch<PT, CT>(p, rpre, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow);

    const int nslices = p.Cin / BK;

    if (tid < 8) lds[RCH + (tid >> 2) * XCH + ZP * 4 + (tid & 3)] = make_uint4(0u, 0u, 0u, 0u);

    // patch of 32-channel slice `slice` into patch buffer `xb` (0 / 1, a literal)
#define VC_XSTAGE(slice, xb)                                                                                              \
    {                                                                                                                     \
        uint32_t so = (slice) < nslices ? (uint32_t)((slice) * BK * ES) : OOB;                                             \
        asm volatile("" : "+s"(so));              /* opaque: or every unrolled step keeps its own offset registers */      \
        _Pragma("unroll") for (int i = 0; i < XI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)&lds[RCH + (xb) * XCH + (i * NW + uwave) * 64], 16,    \
                                                     (int)((xsrc[i] | so) >= OOB ? OOB : xsrc[i] + so), 0, 0, 0);          \
    }
    // weight tile (slice `sl`, tap `tp`) into ring stage `st` (a literal)
#define VC_WSTAGE(sl, tp, st)                                                                                             \
    {                                                                                                                     \
        uint32_t ko = (sl) < nslices ? (uint32_t)(((tp) * p.Cin + (sl) * BK) * ES) : OOB;                                  \
        asm volatile("" : "+s"(ko));                                                                                       \
        _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[(st) * WROWS * KC + (PASS * i + uwave * 16) * KC], 16, \
                                                     (int)(ko >= OOB ? OOB : woff[i] + ko), 0, 0, 0);                       \
    }
    // fragments of tap `t` of patch buffer `xb` and of ring stage `st` into a register set (t, xb, st literals: immediates)
#define VC_FRAGS(XR, WR, xb, t, st)                                                                                       \
    {                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < PT; ++i)                                                                     \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(XR[i]) : "v"(xaddr[i][t]), "n"((xb) * XBYTES) : "memory"); \
        _Pragma("unroll") for (int i = 0; i < CT; ++i)                                                                     \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(WR[i]) : "v"(wfrag[i]), "n"((st) * WSTAGE) : "memory");     \
    }

    VC_XSTAGE(0, 0);
    VC_WSTAGE(0, 0, 0) VC_WSTAGE(0, 1, 1) VC_WSTAGE(0, 2, 2) VC_WSTAGE(0, 3, 3) VC_WSTAGE(0, 4, 4)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * WI) : "memory");       // patch 0, weight stages 0 and 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4v xrA[PT], wrA[CT], xrB[PT], wrB[CT];
    const int abl = p.ablate;                      // timing experiments (VC_CONV_ABLATE bits): 1 no weight DMA, 2 no patch DMA, 4 no fragment reads, 8 no barrier, 16 no MFMAs
    VC_FRAGS(xrA, wrA, 0, 0, 0);
    // Step kt = 9 * slice + t with slice = 2 j + par: ring stage of the tile being multiplied = kt % 6 = (9 par + t) % 6.
    // DMA of weight tile kt + 5 (into the stage read a step ago), fragments of step kt + 1 into (XN, WN), MFMAs of step kt on (XC, WC_),
    // then "tile kt + 2 has landed" -- at most the three newer tiles and, while tile kt + 2 is older than it, this slice's patch
    // prefetch outstanding (LDS-DMA loads return in order) -- and the barrier.
#define VC_STEP(t, par, sl, XC, WC_, XN, WN)                                                                              \
    {                                                                                                                     \
        if ((t) == 0 && !(abl & 2)) VC_XSTAGE((sl) + 1, 1 - (par));                                                        \
        if (!(abl & 1)) VC_WSTAGE((sl) + ((t) + 5) / 9, ((t) + 5) % 9, (9 * (par) + (t) + 5) % 6);                         \
        if (abl & 4) {                                                                                                    \
        } else if ((t) < 8 || (sl) + 1 < nslices) { /* (no look-ahead behind the last step: nothing is in flight at the exit) */ \
            VC_FRAGS(XN, WN, ((t) == 8 ? 1 - (par) : (par)), ((t) + 1) % 9, (9 * (par) + (t) + 1) % 6);                    \
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PT + CT) : "memory"); /* LDS returns in order: the set read a step ago has landed */ \
        } else {                                                                                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        }                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) asm volatile("" : "+v"(XC[i]));                                     \
        _Pragma("unroll") for (int i = 0; i < CT; ++i) asm volatile("" : "+v"(WC_[i]));                                    \
        if (!(abl & 16)) {                                                                                                \
        _Pragma("unroll") for (int a = 0; a < CT; ++a)                                                                     \
            _Pragma("unroll") for (int b = 0; b < PT; ++b) mfma_bf16_inplace(acc[a][b], WC_[a], XC[b]);                    \
        }                                                                                                                 \
        if ((t) < NS - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * WI + XI) : "memory");                        \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * WI) : "memory");                                          \
        if (!(abl & 8)) __builtin_amdgcn_s_barrier();                                                                      \
    }
#define VC_SLICE(par, sl, X0, W0, X1, W1)                                                                                 \
    VC_STEP(0, par, sl, X0, W0, X1, W1) VC_STEP(1, par, sl, X1, W1, X0, W0) VC_STEP(2, par, sl, X0, W0, X1, W1)            \
    VC_STEP(3, par, sl, X1, W1, X0, W0) VC_STEP(4, par, sl, X0, W0, X1, W1) VC_STEP(5, par, sl, X1, W1, X0, W0)            \
    VC_STEP(6, par, sl, X0, W0, X1, W1) VC_STEP(7, par, sl, X1, W1, X0, W0) VC_STEP(8, par, sl, X0, W0, X1, W1)
    for (int slice = 0; slice < nslices; slice += 2) {
        VC_SLICE(0, slice, xrA, wrA, xrB, wrB)
        if (slice + 1 >= nslices) break;
        VC_SLICE(1, slice + 1, xrB, wrB, xrA, wrA)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_results_settle<CT * PT>(&acc[0][0]);
#undef VC_SLICE
#undef VC_STEP
#undef VC_FRAGS
#undef VC_XSTAGE
#undef VC_WSTAGE
    conv_epilogue<PT, CT, false>(p, acc, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow, rpre, have_res);
}

// software-pipelined halo-staged 3x3 / s1 / p1 (conv3x3_halo_pf_kernel; bf16): P(index, BP, BC, WP, WC) -- 8 waves (two per SIMD) or 4
#define VC_HALO_PF_CFGS(P) P(55, 256, 128, 4, 2) P(56, 256, 128, 8, 1) P(57, 256, 128, 4, 1) P(58, 128, 128, 2, 2) \
                           P(59, 128, 128, 4, 2) P(60, 128, 128, 2, 4)
template <int BP, int BC, int WP, int WC>
static int launch_halo_pf(ConvP p, hipStream_t s) {
    constexpr int NW = WP * WC;
    constexpr int X0 = NW == 8 ? 2 : 4, X1 = NW == 8 ? 4 : 7, X2 = NW == 8 ? 6 : 11;      // patch buffer sizes: XI x NW x 16 pixels of 64 B
    static const bool enabled = !(getenv("VC_CONV_HALO_PF") && atoi(getenv("VC_CONV_HALO_PF")) == 0);   // A/B switch
    if (!enabled || !halo_applicable(p, BP)) return VC_ERR_ARG;       // quietly: the autotuner skips it, launch_conv falls back
    const int px = halo_patch_pixels(p, BP);
    if (px > X2 * NW * 16 - 1) return VC_ERR_ARG;
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    p.Kw = p.Kp;
    if (px <= X0 * NW * 16 - 1) launch_timed(p, conv3x3_halo_pf_kernel<BP, BC, WP, WC, X0>, dim3(tiles), dim3(NW * 64), 0, s, p);
    else if (px <= X1 * NW * 16 - 1) launch_timed(p, conv3x3_halo_pf_kernel<BP, BC, WP, WC, X1>, dim3(tiles), dim3(NW * 64), 0, s, p);
    else launch_timed(p, conv3x3_halo_pf_kernel<BP, BC, WP, WC, X2>, dim3(tiles), dim3(NW * 64), 0, s, p);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ---- halo-staged 3x3 / stride 1 / pad 1, PERSISTENT workgroups (bf16) ---------------------------------------------------------------
// What bounds conv3x3_halo_kernel on the wide layers (measured by removal on 128 -> 128 at 40^2, 128 frames: tools/experiments/
// halo_pf_ablate.sh): a 256 x 128 tile carries 9 us of MFMA work, and its workgroup spends as long again OUTSIDE the K loop -- ~4 us
// until the first patch and weight tiles have landed, ~5 us in the epilogue (128 values per lane: bias, SiLU's two quarter-rate
// transcendentals, residual, pack, stores) -- with the matrix pipe idle, because the two workgroups that share a CU start together and
// stay in lockstep (both in their K loops, then both in their epilogues), and the launch's last round runs half empty.
// Here a workgroup walks tiles v = b, b + G, ... and the staging never stops at a tile boundary: the first patch slice and the first
// weight tiles of the NEXT tile are requested during the last slice of this one (the weight ring and the two patch buffers simply carry
// on), so a tile's epilogue runs with the next tile's operands in flight.  The second workgroup of a CU starts half a tile late
// (`dephase`): from then on one workgroup's epilogue and tile set-up run under the other's K loop.  Per-output arithmetic, MFMA and K
// order are conv3x3_halo_kernel's: bit-identical results.
template <int BP, int BC, int WP, int WC, int NS, int XI>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_ps_kernel(const ConvP p, const int dephase_ticks) {
    constexpr int KC = 4, ES = 2, BK = 32;
    constexpr int PASS = 64;
    constexpr int WI = (BC + PASS - 1) / PASS;
    constexpr int WROWS = WI * PASS;
    constexpr int WTP = BP / WP, WTC = BC / WC, PT = WTP / 16, CT = WTC / 16;
    constexpr int ZP = XI * 64 - 1;
    constexpr int XCH = XI * 256;
    constexpr uint32_t OOB = 0x80000000u;
    static_assert(WP * WC == 4 && WTP % 16 == 0 && WTC % 16 == 0, "tile shape");
    static_assert((NS - 2) * WI + XI <= 63 && NS - 1 <= 9, "counted vmcnt / look-ahead of at most one slice");
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * XCH + NS * WROWS * KC];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int ntiles = p.ntiles, G = gridDim.x;
    const int tiles_c = (p.Cout + BC - 1) / BC;
    const int tq = ntiles >> 3, tr = ntiles & 7;
    // XCD-aware order (conv_igemm_kernel): block b runs on XCD b % 8, G is a multiple of 8, each XCD walks a contiguous range of tiles
#define VC_TILE_OF(v) ((((v) & 7) < tr ? ((v) & 7) * (tq + 1) : tr * (tq + 1) + (((v) & 7) - tr) * tq) + ((v) >> 3))
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int W = p.W, H = p.H;
    const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;

    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    const int prow = wave * 16 + (lane >> 2);
    const int wchunk = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
    const int wp = wave % WP, wc = wave / WP;
    const int frow = lane & 15, fch = lane >> 4;
    int wfrag[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) wfrag[i] = 16 * lds_slot<4>(wc * WTC + i * 16 + frow, fch);

    // staging sources of a tile: patch chunk offsets (per DMA instruction of this lane) and weight row offsets; past the last tile
    // everything is out of range (the hardware answers with zeros, the counted waits stay uniform)
    uint32_t xsrc[XI], woff[WI], woff_n[WI];
#define VC_TILE_XSRC(v, XS)                                                                                               \
    if ((v) < ntiles) {                                                                                                   \
        const int tile_ = VC_TILE_OF(v);                                                                                  \
        const int sm0 = (tile_ / tiles_c) * BP;                                                                            \
        const int sg0 = sm0 / W, sg1 = (min(sm0 + BP, p.M) - 1) / W;                                                       \
        const int sgp0 = (sg0 - 1) * W, snpix = (sg1 - sg0 + 3) * W;                                                       \
        const int sgq0 = (p.ablate & 32) ? (sg0 % H - 1) * W : sgp0;      /* timing experiment: every patch from image 0 (L2 hits) */ \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                                                   \
            const int e = (i * 4 + wave) * 64 + lane;                                                                      \
            const int pp = e >> 2, cpos = e & 3;                                                                           \
            const int chunk = cpos ^ ((pp >> 1) & 2);                                                                      \
            const int gp = sgq0 + pp;                                                                                      \
            XS[i] = (pp < snpix && gp >= 0) ? (uint32_t)((gp * p.in_cs + p.in_co) * ES + chunk * 16) : OOB;                \
        }                                                                                                                 \
    } else {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) XS[i] = OOB;                                                        \
    }
#define VC_TILE_WOFF(v, WO)                                                                                               \
    if ((v) < ntiles) {                                                                                                   \
        const int sn0 = (VC_TILE_OF(v) % tiles_c) * BC;                                                                    \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) WO[i] = (uint32_t)(((sn0 + prow + PASS * i) * p.Kw + wchunk * 8) * ES); \
    } else {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < WI; ++i) WO[i] = OOB;                                                        \
    }

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)&lds[0];
    constexpr uint32_t XBYTES = XCH * 16, WSTAGE = WROWS * KC * 16;
    const uint32_t wring = lds_base + 2 * XBYTES;
    const int nslices = p.Cin / BK;

    if (tid < 8) lds[(tid >> 2) * XCH + ZP * 4 + (tid & 3)] = make_uint4(0u, 0u, 0u, 0u);

    // patch of slice `sl` (< nslices) of the tile whose sources are XS into buffer xb
#define VC_XSTAGE(XS, sl, xb)                                                                                             \
    {                                                                                                                     \
        uint32_t so = (uint32_t)((sl) * BK * ES);                                                                          \
        asm volatile("" : "+s"(so));                                                                                       \
        _Pragma("unroll") for (int i = 0; i < XI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)&lds[(xb) * XCH + (i * 4 + uwave) * 64], 16,           \
                                                     (int)(XS[i] >= OOB ? OOB : XS[i] + so), 0, 0, 0);                     \
    }
    // weight tile (slice sl, tap tp) of the tile whose row offsets are WO into ring stage st
#define VC_WSTAGE(WO, sl, tp, st)                                                                                         \
    {                                                                                                                     \
        uint32_t ko = (uint32_t)(((tp) * p.Cin + (sl) * BK) * ES);                                                         \
        asm volatile("" : "+s"(ko));                                                                                       \
        _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[2 * XCH + (st) * WROWS * KC + (PASS * i + uwave * 16) * KC], 16, \
                                                     (int)(WO[i] >= OOB ? OOB : WO[i] + ko), 0, 0, 0);                      \
    }

    if (dephase_ticks > 0 && ((blockIdx.x / (unsigned)(G / 2)) & 1)) {          // the second half of the grid = the second workgroup of every CU
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)dephase_ticks) __builtin_amdgcn_s_sleep(32);
    }

#define VC_TS(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)      /* diagnostics (VC_CONV_DBG) */
    VC_TS(0);
    int v = blockIdx.x;
    int dbg_i = 1;
    VC_TILE_XSRC(v, xsrc);
    VC_TILE_WOFF(v, woff);
    VC_TILE_WOFF(v + G, woff_n);
    VC_XSTAGE(xsrc, 0, 0);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) VC_WSTAGE(woff, st / 9, st % 9, st);      // (NS - 1 <= 9: slice 0)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    VC_TS(1);
    // diagnostics (VC_CONV_DBG): per-step cycle stamps of wave 0 of four workgroups, first tile: trace[wg][step][5] behind the phase stamps
    long long* trace = (p.dbg && (blockIdx.x & 127) == 0 && blockIdx.x < 512 && tid == 0) ? p.dbg + 400000 + (blockIdx.x >> 7) * 4096 : nullptr;
#define VC_TR(j) do { if (trace && v == (int)blockIdx.x) trace[(slice * 9 + t) * 5 + (j)] = (long long)__builtin_readcyclecounter(); } while (0)
    const int abl = p.ablate;                      // timing experiments (VC_CONV_ABLATE bits): 1 no weight DMA, 4 no weight fragment reads, 8 barrier per slice only, 16 no MFMAs
    int sbuf = NS - 1, gs = 0;                     // ring stage the next weight tile goes to; slices consumed so far (patch buffer = gs & 1)
    uint32_t woffs = wring;
    for (; v < ntiles; v += G) {
        const int tile = VC_TILE_OF(v);
        const int m0 = (tile / tiles_c) * BP, n0 = (tile % tiles_c) * BC;
        const int gp0 = (m0 / W - 1) * W;
        uint32_t xaddr[PT][9];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int m = m0 + wp * WTP + i * 16 + frow;
            const bool ok = m < p.M;
            const int mm = ok ? m : m0;
            int g = (int)((float)mm * inv_w);
            g -= (g * W > mm) ? 1 : 0;
            g += ((g + 1) * W <= mm) ? 1 : 0;
            const int x = mm - g * W;
            int b = (int)((float)g * inv_h);
            b -= (b * H > g) ? 1 : 0;
            b += ((b + 1) * H <= g) ? 1 : 0;
            const int y = g - b * H;
            const int pc = mm - gp0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool valid = ok && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const int px = valid ? pc + dy * W + dx : ZP;
                xaddr[i][t] = (uint32_t)((px * 4 + (fch ^ ((px >> 1) & 2))) * 16);
            }
        }
        f32x4 acc[CT][PT];
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
            for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_inputs_settle<CT * PT>(&acc[0][0]);
        if (dbg_i < 6) { VC_TS(dbg_i + 1); }       // stamps 2 / 5: tile set-up done

        for (int slice = 0; slice < nslices; ++slice, ++gs) {
            const uint32_t xb = lds_base + (uint32_t)(gs & 1) * XBYTES;
            // the other patch buffer was last read a slice ago: next slice of this tile, or the first slice of the next one
            // (this tile's patch sources are dead once its last slice has been requested: they become the next tile's)
            if (slice + 1 < nslices) { VC_XSTAGE(xsrc, slice + 1, (gs + 1) & 1); } else { VC_TILE_XSRC(v + G, xsrc); VC_XSTAGE(xsrc, 0, (gs + 1) & 1); }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                VC_TR(0);
                {   // weight tile NS - 1 steps ahead: tap (t + NS - 1) % 9 of this slice / the next one / slice 0 of the next tile
                    const int wsl = slice + (t + NS - 1) / 9;
                    if (abl & 1) { } else
                    if (wsl < nslices) { VC_WSTAGE(woff, wsl, (t + NS - 1) % 9, sbuf); } else { VC_WSTAGE(woff_n, 0, (t + NS - 1) % 9, sbuf); }
                }
                sbuf = sbuf + 1 == NS ? 0 : sbuf + 1;
                VC_TR(1);
                u32x4v xr[PT], wr[CT];
#pragma unroll
                for (int i = 0; i < PT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[i]) : "v"(xb + xaddr[i][t]) : "memory");
#pragma unroll
                for (int i = 0; i < CT; ++i) if (!(abl & 4)) asm volatile("ds_read_b128 %0, %1" : "=v"(wr[i]) : "v"(woffs + wfrag[i]) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                VC_TR(2);
#pragma unroll
                for (int i = 0; i < PT; ++i) asm volatile("" : "+v"(xr[i]));
#pragma unroll
                for (int i = 0; i < CT; ++i) asm volatile("" : "+v"(wr[i]));
#pragma unroll
                for (int a = 0; a < CT; ++a)
#pragma unroll
                    for (int b = 0; b < PT; ++b) if (!(abl & 16)) mfma_bf16_inplace(acc[a][b], wr[a], xr[b]);
                woffs = woffs + WSTAGE == wring + NS * WSTAGE ? wring : woffs + WSTAGE;
                VC_TR(3);
                // the next weight tile has landed once at most the newer ones -- and, while it is still older than this slice's patch
                // request, that request -- are outstanding (the epilogue's loads and stores of the tile before sit on the counter too and
                // can only make the wait longer: everything retires in issue order)
                if (t < NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI + XI) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * WI) : "memory");
                VC_TR(4);
                if (!(abl & 8) || t == 8) __builtin_amdgcn_s_barrier();
            }
        }
        mfma_results_settle<CT * PT>(&acc[0][0]);
        if (dbg_i < 6) { VC_TS(dbg_i + 2); }       // stamps 3 / 6: K loop done
        conv_epilogue<PT, CT, false>(p, acc, m0 + wp * WTP, n0 + wc * WTC + fch * 4, frow);
        if (dbg_i < 6) { if (p.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); VC_TS(dbg_i + 3); dbg_i += 3; }      // stamps 4 / 7: stores retired
#pragma unroll
        for (int i = 0; i < WI; ++i) woff[i] = woff_n[i];
        VC_TILE_WOFF(v + 2 * G, woff_n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead requests past the last tile (zeros) before the LDS is released
#undef VC_XSTAGE
#undef VC_WSTAGE
#undef VC_TILE_XSRC
#undef VC_TILE_WOFF
#undef VC_TS
#undef VC_TR
#undef VC_TILE_OF
}

// persistent halo-staged 3x3 / s1 / p1 (conv3x3_halo_ps_kernel; bf16): Q(index, BP, BC, WP, WC, NS)
#define VC_HALO_PS_CFGS(Q) Q(61, 256, 128, 4, 1, 3) Q(62, 128, 128, 2, 2, 3) Q(63, 128, 128, 2, 2, 2) Q(64, 256, 128, 4, 1, 4)

template <int BP, int BC, int WP, int WC, int NS>
static int launch_halo_ps(ConvP p, hipStream_t s) {
    static const bool enabled = !(getenv("VC_CONV_HALO_PS") && atoi(getenv("VC_CONV_HALO_PS")) == 0);      // A/B switch
    if (!enabled || !halo_applicable(p, BP)) return VC_ERR_ARG;       // quietly: the autotuner skips it, launch_conv falls back
    const int tiles = ((p.M + BP - 1) / BP) * ((p.Cout + BC - 1) / BC);
    p.Kw = p.Kp;
    p.ntiles = tiles;
    static const int cus = [] {
        int dev = 0; hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }();
    // experiments (read once): workgroups of the persistent grid, start delay of its second half in 10 ns ticks (-1: half a tile's MFMA time)
    static const int grid_env = getenv("VC_HALO_PS_GRID") ? atoi(getenv("VC_HALO_PS_GRID")) : 0;
    static const int delay_env = getenv("VC_HALO_PS_DELAY") ? atoi(getenv("VC_HALO_PS_DELAY")) : -1;
    int grid = p.slots > 0 ? p.slots : grid_env > 0 ? grid_env : 2 * cus;
    grid = std::max(8, std::min(grid, (tiles + 7) / 8 * 8) / 8 * 8);
    // half a tile's matrix work at 16 cycles per MFMA and ~2.2 GHz, in 10 ns ticks
    const double tile_mfma_us = (double)(p.Cin / 32 * 9) * (BP / WP / 16) * (BC / WC / 16) * 16.0 / 2200.0;      // MFMAs per wave and tile x 16 cycles
    int delay = delay_env >= 0 ? delay_env : (int)(tile_mfma_us * 100.0 / 2.0);
    if (tiles <= grid / 2) delay = 0;
    const int px = halo_patch_pixels(p, BP);
    if (px <= 4 * 64 - 1) launch_timed(p, conv3x3_halo_ps_kernel<BP, BC, WP, WC, NS, 4>, dim3(grid), dim3(256), 0, s, p, delay);
    else if (px <= 7 * 64 - 1) launch_timed(p, conv3x3_halo_ps_kernel<BP, BC, WP, WC, NS, 7>, dim3(grid), dim3(256), 0, s, p, delay);
    else launch_timed(p, conv3x3_halo_ps_kernel<BP, BC, WP, WC, NS, 11>, dim3(grid), dim3(256), 0, s, p, delay);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_halo_pf_cfg(const ConvP& p, int cfg, hipStream_t s) {
    switch (cfg) {
#define VC_P(i, bp, bc, wp, wc) case i: return launch_halo_pf<bp, bc, wp, wc>(p, s);
        VC_HALO_PF_CFGS(VC_P)
#undef VC_P
#define VC_Q(i, bp, bc, wp, wc, ns) case i: return launch_halo_ps<bp, bc, wp, wc, ns>(p, s);
        VC_HALO_PS_CFGS(VC_Q)
#undef VC_Q
    }
    return VC_ERR_ARG;
}

}  // namespace vc
