// Halo-staged 3x3 / stride 1 / pad 1 convolution (bf16), second form: row-aligned tiles, streamed weight fragments, look-ahead
// fragment reads and a patch fetch spread over the K loop.  Tile configuration 55 of conv_igemm.hip's table.
// Same rows of the reference as conv_igemm.hip: the 3x3 convolutions of ultralytics/yolov5 v6.0 `Bottleneck` blocks that
// /root/reference/networks/yolo.py:70 executes and of the DeepSORT appearance net's BasicBlocks,
// /root/reference/networks/deepsort/deep/model.py:5-98 (row B5).
//
// What per-step cycle stamps of conv3x3_halo_kernel showed on 128 -> 128 at 40^2 (128 frames, two workgroups per CU): a K step of 32 MFMAs per wave (512 matrix-pipe cycles) takes ~1800 cycles -- ~250 to issue the
// step's LDS-DMA instructions, ~500 for the twelve fragment reads and their `lgkmcnt(0)`, ~900 for the MFMAs while the other workgroup's
// wave on the SIMD issues its own, ~100 + ~300 in the counted wait and the barrier -- and the first step of every 32-channel slice
// ~4600 more: all workgroups of the chip request their 28 KB patches in the same microsecond, an HBM-bound burst (~11 B/clk/CU) that
// every wave sits in because the DMA instructions do not issue faster than the memory system accepts them.
//
// Same data flow (patch per 32-channel slice staged once, the nine taps read it at shifted addresses, weight tiles [128][32] through an
// LDS-DMA ring, MFMA and K order unchanged: results bit-identical to conv3x3_halo_kernel), four changes:
//   * tiles are whole rows of the flattened (batch, y) row space -- R rows with R * W <= 256 -- so the patch is exactly R + 2 rows
//     (320 pixels of 64 B at most: 20 KB per buffer instead of 28 KB, which is what lets two workgroups with a deeper ring share a CU);
//   * the patch of the next slice is requested one 4 KB piece per step over the first five steps of a slice instead of all at once:
//     the chip's HBM demand is even over the K loop, and a piece is only waited for a step after its request;
//   * a wave's eight weight fragments of a step stream through four rotating registers (request fragment a + 3, multiply with
//     fragment a), which frees the registers for
//   * the first fragments of step kt + 1 being requested during the last MFMAs of step kt (the ring is one tile deeper in what has
//     landed): the MFMAs of a step start right behind the barrier.
#include <algorithm>
#include <cstdlib>

#include "vc_common.h"
#include "conv_device.h"

namespace vc {

static constexpr int V2_XI = 5;                 // patch pieces (DMA instructions per wave) per slice: 5 x 64 pixels
static constexpr int V2_PATCH_PX = V2_XI * 64;  // 320

template <int NS>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_v2_kernel(const ConvP p, const int R) {
    constexpr int KC = 4, ES = 2, BK = 32, BC = 128, PT = 4, CT = 8, XI = V2_XI, WI = 2;
    constexpr int WROWS = 128;
    constexpr int XCH = XI * 256;                  // 16-byte chunks of a patch buffer, followed by its zero pixel (4 chunks)
    constexpr int XBC = XCH + 4;
    constexpr int RCH = NS * WROWS * KC;           // the weight ring comes first
    constexpr int ZP = XI * 64;                    // index of a buffer's zero pixel
    constexpr uint32_t OOB = 0x80000000u;
    constexpr uint32_t XBYTES = XBC * 16, WSTAGE = WROWS * KC * 16;
    static_assert(NS >= 4 && NS <= 6, "ring depth");
    static_assert(XBYTES + 4096 < 65536, "ds_read immediates");
    __shared__ __attribute__((aligned(16))) uint4 lds[RCH + 2 * XBC];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int nblk = gridDim.x;
    const int tiles_c = (p.Cout + BC - 1) / BC;
    int tile;
    {
        const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 8] = wall_clock64();
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int W = p.W, H = p.H;
    const int ptile = tile / tiles_c;
    const int n0 = (tile - ptile * tiles_c) * BC;
    const int g0 = ptile * R;
    const int nrow = min(R, p.B * H - g0);
    const int m0 = g0 * W, bp = nrow * W;          // this tile's pixels [m0, m0 + bp), bp <= 256
    const int gp0 = m0 - W;                        // first patch pixel (row g0 - 1; negative for the first tile)
    const int npix = (nrow + 2) * W;               // <= 320, checked by the launcher

    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.in), 0, (int)((size_t)p.B * p.H * p.W * p.in_cs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)((size_t)((p.Cout + 127) / 128 * 128) * p.Kw * ES), 0x00020000);

    // patch staging: piece j of this wave fills chunks [(j*4 + wave)*64, +64); lane -> (patch pixel, chunk slot)
    uint32_t xsrc[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int e = (j * 4 + wave) * 64 + lane;
        const int pp = e >> 2, cpos = e & 3;
        const int chunk = cpos ^ ((pp >> 1) & 2);                                 // source-side swizzle (conv3x3_halo_kernel)
        const int gp = gp0 + pp;
        xsrc[j] = (pp < npix && gp >= 0) ? (uint32_t)((gp * p.in_cs + p.in_co) * ES + chunk * 16) : OOB;
    }
    const int prow = wave * 16 + (lane >> 2);
    const int wchunk = (lane & 3) ^ ((0x78 >> (((prow >> 2) & 3) * 2)) & 3);
    uint32_t woff[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) woff[i] = (uint32_t)(((n0 + prow + 64 * i) * p.Kw + wchunk * 8) * ES);

    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_ptr_t)&lds[0];
    const int frow = lane & 15, fch = lane >> 4;
    // weight fragments: channel tile i of ring stage st sits i * 1024 + st * WSTAGE bytes behind this
    const uint32_t wfrag0 = lds_base + 16 * lds_slot<4>(frow, fch);

    f32x4 acc[CT][PT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_inputs_settle<CT * PT>(&acc[0][0]);

    const int nslices = p.Cin / BK;
    const int nk = nslices * 9;
    if (tid < 8) lds[RCH + (tid >> 2) * XBC + ZP * 4 + (tid & 3)] = make_uint4(0u, 0u, 0u, 0u);

    // piece j of the patch of slice `sl` into patch buffer xb (0 / 1)
#define VC_XPIECE(sl, xb, j)                                                                                              \
    {                                                                                                                     \
        uint32_t so = (sl) < nslices ? (uint32_t)((sl) * BK * ES) : OOB;                                                   \
        asm volatile("" : "+s"(so));                                                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)&lds[RCH + (xb) * XBC + ((j) * 4 + uwave) * 64], 16,       \
                                                 (int)((xsrc[j] | so) >= OOB ? OOB : xsrc[j] + so), 0, 0, 0);              \
    }
    // weight tile of K step kk (slice kk / 9, tap kk % 9) into ring stage st
#define VC_WSTAGE(kk_, st)                                                                                                \
    {                                                                                                                     \
        const int kk = (kk_);                                                                                              \
        const int sl_ = kk / 9, tp_ = kk - sl_ * 9;                                                                        \
        uint32_t ko = kk < nk ? (uint32_t)((tp_ * p.Cin + sl_ * BK) * ES) : OOB;                                           \
        asm volatile("" : "+s"(ko));                                                                                       \
        _Pragma("unroll") for (int i = 0; i < WI; ++i)                                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrd, (lds_ptr_t)&lds[(st) * WROWS * KC + (64 * i + uwave * 16) * KC], 16, \
                                                     (int)(ko >= OOB ? OOB : woff[i] + ko), 0, 0, 0);                       \
    }
#define VC_RDX(dst, i, t, xb) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(xaddr[i][t]), "n"((xb) * XBYTES) : "memory")
#define VC_RDW(dst, base, a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"((a) * 1024) : "memory")
#define VC_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define VC_MFMA4(a, wreg, X) { asm volatile("" : "+v"(wreg)); mfma_bf16_inplace(acc[a][0], wreg, X[0]); mfma_bf16_inplace(acc[a][1], wreg, X[1]); \
                               mfma_bf16_inplace(acc[a][2], wreg, X[2]); mfma_bf16_inplace(acc[a][3], wreg, X[3]); }

    // prologue: the whole patch of slice 0, weight tiles 0 .. NS - 2; patch and tiles 0, 1 must have landed
#pragma unroll
    for (int j = 0; j < XI; ++j) VC_XPIECE(0, 0, j);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) VC_WSTAGE(st, st);
    // (the fragment addresses are computed while the first patch and weight tiles are on their way)
    uint32_t xaddr[PT][9];                         // this lane's fragment of every tap in patch buffer 0 (buffer 1: + XBYTES as an immediate)
    {
        const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int q = wave * 64 + i * 16 + frow;
            const bool ok = q < bp;
            const int mm = m0 + (ok ? q : 0);
            int g = (int)((float)mm * inv_w);                                     // global row, +-1 fix-up (mm < 2^24)
            g -= (g * W > mm) ? 1 : 0;
            g += ((g + 1) * W <= mm) ? 1 : 0;
            const int x = mm - g * W;
            int b = (int)((float)g * inv_h);
            b -= (b * H > g) ? 1 : 0;
            b += ((b + 1) * H <= g) ? 1 : 0;
            const int y = g - b * H;
            const int pc = q + W;                                                 // patch index of the centre tap
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool valid = ok && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
                const int px = valid ? pc + dy * W + dx : ZP;
                xaddr[i][t] = lds_base + (uint32_t)((RCH + px * 4 + (fch ^ ((px >> 1) & 2))) * 16);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * WI) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 8 + 1] = wall_clock64();

    // diagnostics (VC_CONV_DBG): phase stamps per workgroup (100 MHz) and per-step cycle stamps of wave 0 of four workgroups
#define VC_TS(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
    long long* trace = (p.dbg && (blockIdx.x & 127) == 0 && blockIdx.x < 512 && tid == 0) ? p.dbg + 400000 + (blockIdx.x >> 7) * 4096 : nullptr;
#define VC_TR(j) do { if (trace) trace[kt * 5 + (j)] = (long long)__builtin_readcyclecounter(); } while (0)
    VC_TS(2);
    u32x4v XA[PT], XB[PT], WR[4];
    int kt = 0, sbuf = NS - 1;                     // ring stage the next weight tile goes to
    uint32_t wcur = wfrag0;                        // fragment base of the weight tile being multiplied (stage kt % NS)
    uint32_t wnxt = wfrag0 + WSTAGE;               // ... of tile kt + 1
    // step 0's first fragments, in the order every step leaves them for the next one: w0, w1, x0, x1, w2, x2, x3
    VC_RDW(WR[0], wcur, 0); VC_RDW(WR[1], wcur, 1); VC_RDX(XA[0], 0, 0, 0); VC_RDX(XA[1], 1, 0, 0); VC_RDW(WR[2], wcur, 2); VC_RDX(XA[2], 2, 0, 0); VC_RDX(XA[3], 3, 0, 0);

    // One K step (tap t of a slice of parity par, i.e. patch buffer par): XC = this step's pixel fragments, XN = the next step's.
    // Weight fragment a of the step lives in WR[a % 4]; fragment a + 3 is requested into the register of fragment a - 1 once the MFMAs of
    // fragment a have been issued (two MFMA groups = 128 cycles of cover for the read; no read ever targets a register an MFMA that
    // has not been issued yet still needs).  LDS returns in order, so the counted waits name exactly the requests still allowed in flight.
    // np = patch pieces requested during steps t - 1 and t (younger than weight tile kt + 2).
#define VC_STEP(t, par, sl, XC, XN)                                                                                       \
    {                                                                                                                     \
        constexpr int np = (((t) >= 1 && (t) - 1 < XI) ? 1 : 0) + (((t) < XI) ? 1 : 0);                                    \
        constexpr int tn = ((t) + 1) % 9, xbn = ((t) == 8 ? 1 - (par) : (par));                                            \
        VC_TR(0);                                                                                                          \
        VC_LGKM(0);                                                 /* w0, w1, x0, x1, w2, x2, x3 */                        \
        VC_TR(1);                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < PT; ++i) asm volatile("" : "+v"(XC[i]));                                     \
        VC_MFMA4(0, WR[0], XC) VC_RDW(WR[3], wcur, 3);                                                                     \
        VC_WSTAGE(kt + NS - 1, sbuf);                               /* into the stage that was multiplied a step ago */     \
        sbuf = sbuf + 1 == NS ? 0 : sbuf + 1;                                                                              \
        if ((t) < XI) VC_XPIECE((sl) + 1, 1 - (par), ((t) < XI ? (t) : 0));   /* the other patch buffer was last read a slice ago */  \
        VC_TR(2);                                                                                                          \
        VC_MFMA4(1, WR[1], XC) VC_RDW(WR[0], wcur, 4);                                                                     \
        VC_MFMA4(2, WR[2], XC) VC_RDW(WR[1], wcur, 5);                                                                     \
        VC_LGKM(2); VC_MFMA4(3, WR[3], XC) VC_RDW(WR[2], wcur, 6);                                                         \
        VC_LGKM(2); VC_MFMA4(4, WR[0], XC) VC_RDW(WR[3], wcur, 7);                                                         \
        /* the first fragments of step kt + 1 (its weight tile has been in LDS since the last barrier); behind the last step they  \
           read a stage and a patch buffer nobody needs any more */                                                         \
        VC_LGKM(2); VC_MFMA4(5, WR[1], XC) VC_RDW(WR[0], wnxt, 0);                                                         \
        VC_LGKM(2); VC_MFMA4(6, WR[2], XC) VC_RDW(WR[1], wnxt, 1); VC_RDX(XN[0], 0, tn, xbn); VC_RDX(XN[1], 1, tn, xbn);    \
        VC_LGKM(4); VC_MFMA4(7, WR[3], XC) VC_RDW(WR[2], wnxt, 2); VC_RDX(XN[2], 2, tn, xbn); VC_RDX(XN[3], 3, tn, xbn);    \
        VC_TR(3);                                                                                                          \
        wcur = wnxt;                                                                                                       \
        wnxt = wnxt + WSTAGE == wfrag0 + NS * WSTAGE ? wfrag0 : wnxt + WSTAGE;                                             \
        /* weight tile kt + 2 has landed once at most the NS - 3 newer tiles and the patch pieces requested behind it are outstanding */ \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * WI + np) : "memory");                                          \
        VC_TR(4);                                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                                     \
        ++kt;                                                                                                             \
    }
#define VC_SLICE(par, sl, X0, X1)                                                                                         \
    VC_STEP(0, par, sl, X0, X1) VC_STEP(1, par, sl, X1, X0) VC_STEP(2, par, sl, X0, X1) VC_STEP(3, par, sl, X1, X0) VC_STEP(4, par, sl, X0, X1) \
    VC_STEP(5, par, sl, X1, X0) VC_STEP(6, par, sl, X0, X1) VC_STEP(7, par, sl, X1, X0) VC_STEP(8, par, sl, X0, X1)
    for (int slice = 0; slice < nslices; slice += 2) {          // (an even number of slices: checked by the launcher; two register sets alternate)
        VC_SLICE(0, slice, XA, XB)
        VC_SLICE(1, slice + 1, XB, XA)
    }
    // the look-ahead requests past the last K step: DMA (zeros) before the LDS is released, fragment reads before their registers are reused
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)"
                 : "+v"(XA[0]), "+v"(XA[1]), "+v"(XA[2]), "+v"(XA[3]), "+v"(WR[0]), "+v"(WR[1]), "+v"(WR[2]) :: "memory");
    mfma_results_settle<CT * PT>(&acc[0][0]);
    VC_TS(3);
#undef VC_SLICE
#undef VC_STEP
#undef VC_MFMA4
#undef VC_LGKM
#undef VC_RDW
#undef VC_RDX
#undef VC_WSTAGE
#undef VC_XPIECE
    ConvP pe = p;
    pe.M = min(p.M, m0 + bp);                      // the pixel slots behind this tile's rows belong to the next tile
    conv_epilogue<PT, CT, false>(pe, acc, m0 + wave * 64, n0 + fch * 4, frow);
    if (p.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); VC_TS(4); }
#undef VC_TS
#undef VC_TR
}

// rows per tile: as many whole rows as fit 256 output pixels and a 320-pixel patch
static int v2_rows(const ConvP& p) {
    const int by_out = 256 / p.W, by_patch = V2_PATCH_PX / p.W - 2;
    return std::min(std::min(by_out, by_patch), p.B * p.H);
}
static bool v2_applicable(const ConvP& p) {
    if (p.prec != PREC_BF16 || p.kh != 3 || p.kw != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1) return false;
    if (p.Cin % 64 != 0 || p.in_cs % 8 != 0 || p.in_co % 8 != 0 || p.Ho != p.H || p.Wo != p.W || p.m_dev) return false;
    if (p.W < 1 || p.W > 64) return false;
    const int r = v2_rows(p);
    return r >= 1 && r * p.W >= 160;               // (tiles that fill less than 5/8 of the MFMA slots are left to the other kernels)
}

int launch_halo_v2_cfg(const ConvP& p_in, int cfg, hipStream_t s) {
    static const bool enabled = !(getenv("VC_CONV_HALO_V2") && atoi(getenv("VC_CONV_HALO_V2")) == 0);      // A/B switch
    if (cfg != 55 || !enabled || !v2_applicable(p_in)) return VC_ERR_ARG;      // quietly: the autotuner skips it, launch_conv falls back
    ConvP p = p_in;
    const int R = v2_rows(p);
    const int rows = p.B * p.H;
    const int tiles = ((rows + R - 1) / R) * ((p.Cout + 127) / 128);
    p.Kw = p.Kp;
    p.ntiles = tiles;
    launch_timed(p, conv3x3_halo_v2_kernel<4>, dim3(tiles), dim3(256), 0, s, p, R);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
