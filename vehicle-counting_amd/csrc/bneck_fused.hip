// A 64-channel YOLOv5 Bottleneck in one kernel (bf16): b1 = SiLU(cv1 y) (1x1, 64 -> 64), m = [y +] SiLU(cv2 b1) (3x3 / pad 1, 64 -> 64)
// -- models/common.py::Bottleneck of ultralytics/yolov5 v6.0 (e = 1.0 inside C3), reached from /root/reference/networks/yolo.py:70.
// YOLOv5s runs it three times at 80 x 80 (both bottlenecks of the second backbone C3, with the shortcut; the one of the P3 head C3,
// without).  Unfused, b1 (105 MB per 128 frames) is written and read back and the 1x1 is a launch of its own at the HBM roof.
//
// Same construction as c3_fused.hip: a workgroup owns an 8 x 16 tile of output pixels, computes b1 on the tile's 10 x 18 halo region
// into LDS (pixels outside the image hold 0: the 3x3 pads b1 with zeros) and convolves it from there.  The weights of both layers
// (8 + 72 KB in MFMA fragment order) stay in LDS for the whole launch, which leaves room for ONE workgroup per CU: eight waves, two
// per SIMD.  LDS: b1 of two tiles on their halo regions (two 32-channel planes each, 2 x 24 KB), weights 80 KB = 128 KB.
// MFMA operand order and k order (tap-major, then the two 32-channel halves of a tap) equal conv_igemm_kernel's; the epilogues are
// conv_epilogue_bf16's expressions.
//
// CV3 instance: the LAST Bottleneck of a 64-channel C3 also runs the block's cv3 (1x1, [m | y2] 128 -> 128) on its tile: m (105 MB per
// 128 frames at 80 x 80) is neither written nor read back and the 1x1 -- a launch at the HBM roof of its own -- disappears.  The
// consumer waves re-lay their bf16 m values from the accumulator layout (lane = 4 channels of a pixel) into MFMA B operands (lane = 8
// consecutive channels of a pixel, all four k quarters the same pixel) with three rounds of v_permlane16/32_swap, no LDS round trip;
// y2 is read from HBM in operand order; cv3's weights (32 KB) take the rest of the LDS (157 of 160 KB).  k order = channel order:
// bit-identical to the separate launch.  Measured per 128 frames at 80 x 80: 0.195 ms against 0.100 + 0.117 for the two launches.
// (cv3 as a third pipeline stage on the PRODUCER waves -- m handed over in the b1 buffer the consumers have just read, two workgroup
// barriers per tile -- was built, bit-identical, and measured 0.28 ms: with 144 SiLU values per lane and tile the kernel is bound by the
// quarter-rate transcendentals of its three epilogues, not by which wave runs them, and the second barrier costs more than it balances.)
#include <algorithm>

#include "kernels.h"

namespace vc {

typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef float f32x2b __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8b __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2b __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
union ChunkB { uint4 u; bf16x8b h; };

#define BN_TH 8
#define BN_TW 16
#define BN_RW (BN_TW + 2)              // 18
#define BN_NH ((BN_TH + 2) * BN_RW)    // 180 halo pixels
#define BN_NT ((BN_NH + 15) / 16)      // 12 pixel tiles of the halo region
#define BN_NP BN_NH                    // 180 pixel slots (the 12th pixel tile's slots 180 .. 191 are never stored or read)
#define BN_NW 8
#define BN_PLANE (BN_NP * 64)          // bytes of one 32-channel plane

__device__ __forceinline__ int bn_addr(int px, int chunk) { return (px * 4 + (chunk ^ ((px >> 1) & 2))) * 16; }   // byte offset, 64-byte pixels
__device__ __forceinline__ f32x2b bn_sigmoid2(f32x2b x) {
    const f32x2b t = x * (f32x2b){-1.442695040888963387f, -1.442695040888963387f};
    const f32x2b d = (f32x2b){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (f32x2b){1.0f, 1.0f};
    return (f32x2b){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ f32x2b bn_silu2(f32x2b x) {
    const f32x2b t = x * (f32x2b){-1.442695040888963387f, -1.442695040888963387f};
    const f32x2b d = (f32x2b){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (f32x2b){1.0f, 1.0f};
    return x * (f32x2b){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

struct BnArgs {
    const uint4 *w1, *w2;
    const float *b1, *b2;
    int kw1, kw2;                          // weight row strides in 16-byte chunks
    const uint16_t* x; int in_cs, in_co;
    uint16_t* y; int out_cs, out_co;
    int B, H, W, tiles_x, tiles_y, res;
    // CV3 instance: cv3's weights / bias, the y2 half of its input (first channel, pixel stride), its output
    const uint4* w3; const float* b3; int kw3;
    const uint16_t* y2; int y2_cs;
    uint16_t* z; int z_cs, z_co;
};

// Two groups of four waves work on DIFFERENT tiles at the same time: the producers (waves 4-7) compute b1 of tile k + 1 -- few MFMAs,
// many SiLU transcendentals, operands straight from global memory (a 1x1 needs no staging) -- while the consumers (waves 0-3) run the
// 3x3 of tile k from the b1 buffer the producers filled one step earlier: MFMA- and LDS-heavy.  Every SIMD holds one wave of each
// group, so one group's MFMAs cover the other's quarter-rate v_exp / v_rcp; b1 is double-buffered, one workgroup barrier per tile.
template <bool CV3>
__global__ __launch_bounds__(BN_NW * 64) void bneck_fused_kernel(const BnArgs a) {
    const bool RES = a.res != 0;                            // launch-uniform: the block has a shortcut
    __shared__ uint4 bs[2][2 * BN_NP * 4];                 // 2 x 22.5 KB: b1 of two tiles, [32-channel plane][pixel slot][4 chunks]
    __shared__ uint4 w1s[2 * 4 * 64];                      // 8 KB: cv1, [k step][channel tile][lane]
    __shared__ uint4 w2s[18 * 4 * 64];                     // 72 KB: cv2, [k step = 2 tap + half][channel tile][lane]
    __shared__ uint4 w3s[CV3 ? 4 * 8 * 64 : 1];            // CV3: 32 KB: cv3, [k step][channel tile][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    constexpr int NT = BN_NW * 64;
    for (int i = threadIdx.x; i < 2 * 4 * 64; i += NT) {
        const int l = i & 63, ct = (i >> 6) & 3, s = i >> 8;
        w1s[i] = a.w1[(size_t)(ct * 16 + (l & 15)) * a.kw1 + 4 * s + (l >> 4)];
    }
    for (int i = threadIdx.x; i < 18 * 4 * 64; i += NT) {
        const int l = i & 63, ct = (i >> 6) & 3, s = i >> 8;
        w2s[i] = a.w2[(size_t)(ct * 16 + (l & 15)) * a.kw2 + 4 * s + (l >> 4)];
    }
    if constexpr (CV3) {
        for (int i = threadIdx.x; i < 4 * 8 * 64; i += NT) {
            const int l = i & 63, ct = (i >> 6) & 7, s = i >> 9;
            w3s[i] = a.w3[(size_t)(ct * 16 + (l & 15)) * a.kw3 + 4 * s + (l >> 4)];
        }
    }
    const bool producer = wave >= 4;
    const int gw = wave & 3;
    float4 bv[4];                                           // producers: cv1's bias, consumers: cv2's
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = *(const float4*)((producer ? a.b1 : a.b2) + ct * 16 + kq * 4);
    const int ntiles = a.B * a.tiles_y * a.tiles_x;
    const int nk = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;     // tiles of this workgroup
    const bool odd = (kq & 1) != 0;
    __syncthreads();                                        // weights are in LDS

    // producers: the y fragments (MFMA B operands) of a tile's three pixel tiles of this wave, loaded one tile ahead
    uint4 yf[3][2], yn[3][2];
    auto load_y = [&](int t, uint4 (&dst)[3][2]) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = (gw * 3 + j) * 16 + col;
            const int ry = (n * 3641) >> 16, rx = n - ry * BN_RW;             // n / 18 for n < 192
            const int gy = ty * BN_TH - 1 + ry, gx = tx * BN_TW - 1 + rx;
            dst[j][0] = make_uint4(0u, 0u, 0u, 0u); dst[j][1] = dst[j][0];
            if (n < BN_NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const uint16_t* px = a.x + (((size_t)b * a.H + gy) * a.W + gx) * a.in_cs + a.in_co + kq * 8;
                dst[j][0] = *(const uint4*)px; dst[j][1] = *(const uint4*)(px + 32);
            }
        }
    };
    if (producer && nk > 0) load_y(blockIdx.x, yf);
    for (int k = 0; k <= nk; ++k) {
        if (producer) {
            if (k < nk) {
                // ---- cv1 (1x1) of tile k on its halo region -> bs[k & 1]; pixels outside the image hold 0 (the 3x3 pads b1 with zeros) ------
                const int t = blockIdx.x + k * gridDim.x;
                if (k + 1 < nk) load_y(t + gridDim.x, yn);
                const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y;
                char* bsb = (char*)bs[k & 1];
                // all 24 MFMAs first (12 independent accumulators), then the epilogues: the matrix pipe drains while the VALU / transcendental
                // units work through the earlier accumulators
                f32x4b pacc[3][4];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    ChunkB y0f, y1f;
                    y0f.u = yf[j][0]; y1f.u = yf[j][1];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        ChunkB w0, w1;
                        w0.u = w1s[(0 * 4 + ct) * 64 + lane]; w1.u = w1s[(1 * 4 + ct) * 64 + lane];
                        pacc[j][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.h, y0f.h, (f32x4b){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        pacc[j][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.h, y1f.h, pacc[j][ct], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int n = (gw * 3 + j) * 16 + col;
                    const int ry = (n * 3641) >> 16, rx = n - ry * BN_RW;
                    const int gy = ty * BN_TH - 1 + ry, gx = tx * BN_TW - 1 + rx;
                    const bool inimg = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const f32x4b acc = pacc[j][ct];
                        const f32x2b lo = bn_silu2((f32x2b){acc[0], acc[1]} + (f32x2b){bv[ct].x, bv[ct].y});
                        const f32x2b hi = bn_silu2((f32x2b){acc[2], acc[3]} + (f32x2b){bv[ct].z, bv[ct].w});
                        const bf16x2b p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                        uint2 v = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
                        if (!inimg) v = make_uint2(0u, 0u);
                        // channel tile ct = channels 16 ct .. 16 ct + 15: plane ct >> 1, chunks 2 (ct & 1) and 2 (ct & 1) + 1 of the pixel
                        if (n < BN_NH) *(uint2*)(bsb + (ct >> 1) * BN_PLANE + bn_addr(n, (ct & 1) * 2 + (kq >> 1)) + (kq & 1) * 8) = v;
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) { yf[j][0] = yn[j][0]; yf[j][1] = yn[j][1]; }
            }
        } else if (k >= 1) {
            // ---- cv2 (3x3) of tile k - 1 from bs[(k - 1) & 1] [+ shortcut] -> HBM: wave gw owns rows 2 gw, 2 gw + 1, all 64 channels -------
            const int t = blockIdx.x + (k - 1) * gridDim.x;
            const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
            const int oy0 = ty * BN_TH, ox0 = tx * BN_TW;
            const char* bsb = (const char*)bs[(k - 1) & 1];
            uint2 rs[4][2];                                 // the shortcut: this lane's 4 channels of its pixel per (channel tile, row), read ahead
            if (RES) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int oy = oy0 + gw * 2 + q, ox = ox0 + col;
                    const bool ok = oy < a.H && ox < a.W;
                    const uint16_t* px = a.x + (((size_t)b * a.H + min(oy, a.H - 1)) * a.W + min(ox, a.W - 1)) * a.in_cs + a.in_co + kq * 4;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) rs[ct][q] = ok ? *(const uint2*)(px + ct * 16) : make_uint2(0u, 0u);
                }
            }
            f32x4b acc[4][2];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[ct][q] = (f32x4b){0.f, 0.f, 0.f, 0.f};
            // 18 k steps (tap-major, then the tap's two 32-channel halves), software-pipelined by hand: the six fragment reads of step
            // s + 1 are issued BEFORE the eight MFMAs of step s (the scheduling barriers keep hipcc from sinking them back next to their
            // uses, where every pair of MFMAs waited for an LDS read)
            ChunkB bfr[2][2], wfr[2][4];
            auto load_step = [&](int st, ChunkB (&bf)[2], ChunkB (&wf)[4]) {
                const int tp = st >> 1, hf = st & 1, tyy = tp / 3, txx = tp - tyy * 3;
#pragma unroll
                for (int q = 0; q < 2; ++q) bf[q].u = *(const uint4*)(bsb + hf * BN_PLANE + bn_addr((gw * 2 + q + tyy) * BN_RW + col + txx, kq));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) wf[ct].u = w2s[(st * 4 + ct) * 64 + lane];
            };
            load_step(0, bfr[0], wfr[0]);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + 1 < 18) load_step(st + 1, bfr[(st + 1) & 1], wfr[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[st & 1][ct].h, bfr[st & 1][q].h, acc[ct][q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            uint4 y2f[2][2];                                // CV3: this lane's y2 operands, [row][k step], read ahead of the 3x3's epilogue
            if constexpr (CV3) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int oy = oy0 + gw * 2 + q, ox = ox0 + col;
                    const bool ok = oy < a.H && ox < a.W;
                    const uint16_t* px = a.y2 + (((size_t)b * a.H + min(oy, a.H - 1)) * a.W + min(ox, a.W - 1)) * a.y2_cs + kq * 8;
                    y2f[q][0] = ok ? *(const uint4*)px : make_uint4(0u, 0u, 0u, 0u);
                    y2f[q][1] = ok ? *(const uint4*)(px + 32) : make_uint4(0u, 0u, 0u, 0u);
                }
            }
            uint4 mo[4];                                    // CV3: m as stored below, lane (col, kq) = channels 16 ct + 8 (kq >> 1) .. + 7 of row kq & 1
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                uint2 P[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x2b xl = (f32x2b){acc[ct][q][0], acc[ct][q][1]} + (f32x2b){bv[ct].x, bv[ct].y};
                    const f32x2b xh = (f32x2b){acc[ct][q][2], acc[ct][q][3]} + (f32x2b){bv[ct].z, bv[ct].w};
                    const f32x2b rl = bn_sigmoid2(xl), rh = bn_sigmoid2(xh);
                    f32x2b lo, hi;
                    if (RES) {          // added AFTER the activation, as ONE fused multiply-add (x * sigmoid(x) + y rounded once): what conv_epilogue_bf16 compiles to
                        const uint2 r = rs[ct][q];
                        lo = __builtin_elementwise_fma(xl, rl, (f32x2b){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)});
                        hi = __builtin_elementwise_fma(xh, rh, (f32x2b){__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)});
                    } else {
                        lo = xl * rl; hi = xh * rh;
                    }
                    const bf16x2b p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                    P[q] = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
                }
                const u32x2b sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
                const u32x2b sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
                const uint4 o4 = make_uint4(sx.x, sy.x, sx.y, sy.y);
                if constexpr (CV3) {
                    mo[ct] = o4;
                } else {
                    const int oy = oy0 + gw * 2 + (odd ? 1 : 0), ox = ox0 + col;
                    if (oy < a.H && ox < a.W)
                        *(uint4*)(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.out_cs + a.out_co + ct * 16 + (kq & ~1) * 4) = o4;
                }
            }
            if constexpr (CV3) {
                // ---- cv3 on the tile: z = SiLU(W3 [m | y2] + b3), 128 -> 128 --------------------------------------------------------------
                // m operands.  Write c8 for the 8-channel chunk index (channels 8 c8 .. 8 c8 + 7); mo[ct] holds c8 = 2 ct + (kq >> 1) of row
                // kq & 1.  Round A (rows of 16 lanes, odd <-> even, between ct and ct + 2) makes every register ONE pixel row:
                //   X = (mo[0], mo[2]) -> row 0 with c8 = [0, 4, 1, 5] over kq, Y -> row 1 alike; (mo[1], mo[3]) -> [2, 6, 3, 7].
                // Round B (halves of 32 lanes between the two registers of a row) -> [0, 4, 2, 6] / [1, 5, 3, 7]; round C (rows of 16 again)
                // -> [0, 1, 2, 3] / [4, 5, 6, 7]: k step 0 and k step 1 of the row in channel order.
                uint4 T[2][2];                              // [row][k step]
                {
                    uint32_t r0[4][4] = {{mo[0].x, mo[0].y, mo[0].z, mo[0].w}, {mo[1].x, mo[1].y, mo[1].z, mo[1].w},
                                         {mo[2].x, mo[2].y, mo[2].z, mo[2].w}, {mo[3].x, mo[3].y, mo[3].z, mo[3].w}};
                    uint32_t t[2][2][4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const u32x2b a02 = __builtin_amdgcn_permlane16_swap(r0[0][d], r0[2][d], false, false);   // row 0 [0,4,1,5], row 1 alike
                        const u32x2b a13 = __builtin_amdgcn_permlane16_swap(r0[1][d], r0[3][d], false, false);   // row 0 [2,6,3,7], row 1 alike
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const u32x2b bq = __builtin_amdgcn_permlane32_swap(a02[q], a13[q], false, false);    // [0,4,2,6] / [1,5,3,7]
                            const u32x2b cq = __builtin_amdgcn_permlane16_swap(bq.x, bq.y, false, false);          // [0,1,2,3] / [4,5,6,7]
                            t[q][0][d] = cq.x; t[q][1][d] = cq.y;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int st = 0; st < 2; ++st) T[q][st] = make_uint4(t[q][st][0], t[q][st][1], t[q][st][2], t[q][st][3]);
                }
                f32x4b zacc[8][2];
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int q = 0; q < 2; ++q) zacc[ct][q] = (f32x4b){0.f, 0.f, 0.f, 0.f};
                // (the weight fragment reads pipelined by hand one group of four ahead of the MFMAs, as in the 3x3, and the eight bias loads hoisted:
                // 0.195 -> 0.265 ms -- the compiler's own schedule is the better one at 256 VGPRs; removed)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    ChunkB bq[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) bq[q].u = st < 2 ? T[q][st] : y2f[q][st - 2];
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) {
                        ChunkB wf;
                        wf.u = w3s[(st * 8 + ct) * 64 + lane];
#pragma unroll
                        for (int q = 0; q < 2; ++q) zacc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.h, bq[q].h, zacc[ct][q], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    const float4 b3 = *(const float4*)(a.b3 + ct * 16 + kq * 4);
                    uint2 P[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x2b lo = bn_silu2((f32x2b){zacc[ct][q][0], zacc[ct][q][1]} + (f32x2b){b3.x, b3.y});
                        const f32x2b hi = bn_silu2((f32x2b){zacc[ct][q][2], zacc[ct][q][3]} + (f32x2b){b3.z, b3.w});
                        const bf16x2b p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                        P[q] = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
                    }
                    const u32x2b sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
                    const u32x2b sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
                    const uint4 o4 = make_uint4(sx.x, sy.x, sx.y, sy.y);
                    const int oy = oy0 + gw * 2 + (odd ? 1 : 0), ox = ox0 + col;
                    if (oy < a.H && ox < a.W)
                        *(uint4*)(a.z + (((size_t)b * a.H + oy) * a.W + ox) * a.z_cs + a.z_co + ct * 16 + (kq & ~1) * 4) = o4;
                }
            }
        }
        __syncthreads();                                    // bs[k & 1] is complete, bs[(k - 1) & 1] is free again
    }
}

// pm1 = Bottleneck.cv1, pm2 = Bottleneck.cv2 as engine.hip::yolo_c3 builds them
bool bneck_fused_applicable(const ConvP& pm1, const ConvP& pm2) {
    if (!(pm1.prec == PREC_BF16 && pm1.kh == 1 && pm1.kw == 1 && pm1.sh == 1 && pm1.sw == 1 && pm1.ph == 0 && pm1.pw == 0 && pm1.Cin == 64 && pm1.Cout == 64 &&
          pm1.act == ACT_SILU && pm1.res_mode == RES_NONE && !pm1.out_f32 && pm1.split == 0))
        return false;
    if (!(pm2.prec == PREC_BF16 && pm2.kh == 3 && pm2.kw == 3 && pm2.sh == 1 && pm2.sw == 1 && pm2.ph == 1 && pm2.pw == 1 && pm2.Cin == 64 && pm2.Cout == 64 &&
          pm2.act == ACT_SILU && (pm2.res_mode == RES_NONE || pm2.res_mode == RES_AFTER_ACT) && !pm2.out_f32 && pm2.split == 0))
        return false;
    if (!(pm2.in == pm1.out && pm2.in_co == pm1.out_co && pm2.in_cs == pm1.out_cs && pm2.H == pm1.H && pm2.W == pm1.W && pm2.B == pm1.B)) return false;
    if (pm2.res_mode == RES_AFTER_ACT && !(pm2.res == pm1.in && pm2.res_co == pm1.in_co && pm2.res_cs == pm1.in_cs)) return false;   // the shortcut is the block's input
    // the block's input must not be its output buffer (tiles read halo pixels that neighbouring tiles write)
    if (pm2.out == pm1.in && pm2.out_co < pm1.in_co + 64 && pm1.in_co < pm2.out_co + 64) return false;
    return pm1.in_cs % 8 == 0 && pm1.in_co % 8 == 0 && pm2.out_cs % 8 == 0 && pm2.out_co % 8 == 0 && pm1.Kp >= 64 && pm2.Kp >= 576;
}

// p3 = C3.cv3 reading [m | y2]: the Bottleneck's output slice followed by 64 more channels of the same buffer
bool bneck_cv3_fused_applicable(const ConvP& pm1, const ConvP& pm2, const ConvP& p3) {
    if (!bneck_fused_applicable(pm1, pm2)) return false;
    if (!(p3.prec == PREC_BF16 && p3.kh == 1 && p3.kw == 1 && p3.sh == 1 && p3.sw == 1 && p3.ph == 0 && p3.pw == 0 && p3.Cin == 128 && p3.Cout == 128 &&
          p3.act == ACT_SILU && p3.res_mode == RES_NONE && !p3.out_f32 && p3.split == 0 && !p3.m_dev))
        return false;
    if (!(p3.in == pm2.out && p3.in_co == pm2.out_co && p3.in_cs == pm2.out_cs && p3.in_cs >= p3.in_co + 128 && p3.H == pm2.H && p3.W == pm2.W && p3.B == pm2.B)) return false;
    // cv3's output must not overlap anything the kernel still reads (x with its halo, y2)
    if (p3.out == pm1.in && p3.out_co < pm1.in_co + 64 && pm1.in_co < p3.out_co + 128) return false;
    if (p3.out == p3.in && p3.out_co < p3.in_co + 128 && p3.in_co < p3.out_co + 128) return false;
    return p3.out_cs % 8 == 0 && p3.out_co % 8 == 0 && p3.Kp >= 128;
}

// Persistent grid of the Bottleneck kernels: one workgroup holds 128 KB of a CU's LDS, so a grid on all 256 CUs shuts the other conv queue's
// workgroups out of the whole chip for the length of the launch (three launches of 0.1 - 0.19 ms per 128-frame step).  An eighth of the CUs
// is left free: the kernel alone is ~14 % slower (28.6 instead of 25 tile rounds at 80^2), the step is 2.5 - 3.5 % faster (60-step runs on
// one box, alternating: 18.36 / 18.42 / 18.47 k frames/s with 256 workgroups, 18.90 / 19.10 k with 224, 18.91 k with 192, 18.66 k with 240).
// The same cap on front_fused_kernel (150 KB) and reid_block_fused_kernel (152 KB) stayed inside the run-to-run spread.
static int bneck_grid_cap(int cus) {
    static const int forced = getenv("VC_BN_GRID") ? atoi(getenv("VC_BN_GRID")) : 0;      // A/B switch
    return forced > 0 ? forced : cus - cus / 8;
}

int launch_bneck_cv3_fused(const ConvP& pm1, const ConvP& pm2, const ConvP& p3, hipStream_t s) {
    if (!bneck_cv3_fused_applicable(pm1, pm2, p3)) return VC_ERR_ARG;
    BnArgs a{};
    a.w1 = (const uint4*)pm1.w; a.w2 = (const uint4*)pm2.w; a.b1 = pm1.bias; a.b2 = pm2.bias;
    a.kw1 = pm1.Kp / 8; a.kw2 = pm2.Kp / 8;
    a.x = (const uint16_t*)pm1.in; a.in_cs = pm1.in_cs; a.in_co = pm1.in_co;
    a.y = nullptr;
    a.w3 = (const uint4*)p3.w; a.b3 = p3.bias; a.kw3 = p3.Kp / 8;
    a.y2 = (const uint16_t*)p3.in + p3.in_co + 64; a.y2_cs = p3.in_cs;
    a.z = (uint16_t*)p3.out; a.z_cs = p3.out_cs; a.z_co = p3.out_co;
    a.B = pm1.B; a.H = pm1.H; a.W = pm1.W;
    a.tiles_x = (a.W + BN_TW - 1) / BN_TW; a.tiles_y = (a.H + BN_TH - 1) / BN_TH;
    const int ntiles = a.B * a.tiles_x * a.tiles_y;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    a.res = pm2.res_mode == RES_AFTER_ACT ? 1 : 0;
    launch_timed(pm1, bneck_fused_kernel<true>, dim3(std::min(ntiles, bneck_grid_cap(cus))), dim3(BN_NW * 64), 0, s, a);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_bneck_fused(const ConvP& pm1, const ConvP& pm2, hipStream_t s) {
    if (!bneck_fused_applicable(pm1, pm2)) return VC_ERR_ARG;
    BnArgs a{};
    a.w1 = (const uint4*)pm1.w; a.w2 = (const uint4*)pm2.w; a.b1 = pm1.bias; a.b2 = pm2.bias;
    a.kw1 = pm1.Kp / 8; a.kw2 = pm2.Kp / 8;
    a.x = (const uint16_t*)pm1.in; a.in_cs = pm1.in_cs; a.in_co = pm1.in_co;
    a.y = (uint16_t*)pm2.out; a.out_cs = pm2.out_cs; a.out_co = pm2.out_co;
    a.B = pm1.B; a.H = pm1.H; a.W = pm1.W;
    a.tiles_x = (a.W + BN_TW - 1) / BN_TW; a.tiles_y = (a.H + BN_TH - 1) / BN_TH;
    const int ntiles = a.B * a.tiles_x * a.tiles_y;
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n;
    }();
    const int grid = std::min(ntiles, bneck_grid_cap(cus));                // persistent, one workgroup per CU (128 KB of LDS)
    a.res = pm2.res_mode == RES_AFTER_ACT ? 1 : 0;
    launch_timed(pm1, bneck_fused_kernel<false>, dim3(grid), dim3(BN_NW * 64), 0, s, a);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
