// YOLOv5 layers 0 + 1 in one kernel (bf16): the 6x6 / stride-2 stem and the 3x3 / stride-2 conv that follows it
// (models/yolov5s.yaml v6.0: [-1, 1, Conv, [64, 6, 2, 2]], [-1, 1, Conv, [128, 3, 2]] at width 0.5 -> 32 and 64 channels), reached
// from /root/reference/networks/yolo.py:70.
//
// Unfused, the stem's output (B x 320 x 320 x 32 bf16 = 839 MB per 128 frames) is written to HBM and read straight back: both
// layers sit on the HBM roof and together move 2.3 GB per 128 frames.  Here a workgroup owns a 16 x 16 tile of layer-1 output,
// computes the 33 x 33 layer-0 pixels it needs (halo recomputed: +6 % stem FLOPs) into LDS and convolves them from there:
// HBM traffic = the u8 frames in (157 MB) + layer 1 out (419 MB).  Eight waves (two per SIMD) share one 16 x 16 output tile: the stem's
// weights live in registers (32 x 160 -> 10 fragments), the conv's (64 x 288 -> 36 fragments) in LDS in fragment order, the patch
// of the next tile travels through registers while the current tile is computed (as in stem_direct.hip).
//
// Layer-0 tile in LDS: two planes by column parity (the stride-2 taps of one output row then read 16 CONSECUTIVE pixels of a plane),
// 64 bytes per pixel, 16-byte chunks XOR-swizzled with conv3x3_halo_kernel's formula (conflict-free for any base pixel).
// MFMA operand order and k order equal conv_igemm_kernel's for both layers: bit-identical to the unfused path.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "kernels.h"

namespace vc {

typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8f __attribute__((ext_vector_type(8)));
union ChunkF { uint4 u; bf16x8f h; };

// (-DFF_TH=4 -DFF_NW=4: 4 x 16 tiles, 70 KB of LDS, TWO workgroups of four waves per CU whose phases drift apart -- the recipe of
// c3_fused.hip -- is bit-identical and was measured 0.78 against 0.53 ms per 128 frames: 22 patch rows per 4 output rows instead of 70
// per 16, 12 % more stem halo, four barriers per quarter of the work)
#ifndef FF_TH
#define FF_TH 16                  // layer-1 output tile rows (even)
#endif
#ifndef FF_NW
#define FF_NW 8                   // waves per workgroup; FF_NW / (FF_TH / 2) = channel splits of the conv phase (1, 2 or 4)
#endif
#define FF_TW 16
#define FF_RH (2 * FF_TH + 1)     // layer-0 region rows (33)
#define FF_PW (FF_TW + 1)         // layer-0 pixels per parity plane row (17: columns 0, 2, .., 32 / 1, 3, .., 31 + one unused)
#define FF_PLANE (FF_RH * FF_PW)  // 561
#define FF_NPIX (2 * FF_PLANE)    // 1122 layer-0 pixel slots
#define FF_NT0 ((FF_NPIX + 15) / 16)   // 71 pixel-tile slots of the LDS tile
#define FF_NT0R (2 * FF_RH + (FF_RH + 15) / 16)   // 69 row-aligned stem pixel tiles
#define FF_PR (2 * FF_RH + 4)     // input patch rows (70)
#define FF_PC (2 * FF_TW + 1 + 2) // input patch pixel pairs per row (35)
#define FF_PP 36                  // LDS row pitch of the patch in 16-byte chunks
#define FF_RAWC 15                // U8: 16-byte chunks of one raw source row of the patch (35 pixel pairs x 6 B = 210 B + up to 15 B of alignment slack)
#define FF_RAWP (FF_RAWC * 16)    // byte pitch of a raw row in LDS

typedef float f32x2f __attribute__((ext_vector_type(2)));
// SiLU of two values: the multiplies and the add as packed fp32 operations (same IEEE results as the scalar forms)
// (the SiLU's multiplies / add as plain fp32 VALU through inline assembly instead of the packed forms hipcc emits: 0.506 -> 0.494 ms,
// inside the run-to-run spread: removed)
__device__ __forceinline__ f32x2f ff_silu2(f32x2f x) {
    const f32x2f t = x * (f32x2f){-1.442695040888963387f, -1.442695040888963387f};
    const f32x2f d = (f32x2f){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (f32x2f){1.0f, 1.0f};
    return x * (f32x2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ int ff_l0_addr(int px, int chunk) { return (px * 4 + (chunk ^ ((px >> 1) & 2))) * 16; }   // byte offset in the layer-0 tile

// Resize mode (RS, round 4): frames that are NOT at network scale (the reference's only real geometry, 1280 x 720 -> 384 x 640, Q8).
// The letterbox's INTER_LINEAR resize (cv::resize on u8: 11-bit fixed point, aux_kernels.hip::letterbox_kernel) is evaluated when the
// patch is built: the source rows / columns the tile's 70 x 70 tensor pixels sample are fetched as aligned 16-byte chunks into the raw
// staging area (rs_rows x rs_chunks, planned by the launcher with the kernel's own lin_coef), the per-column and per-row taps and
// coefficients of the tile are computed once into two small LDS tables, and every patch pixel is two rows x two taps x three bytes
// from LDS.  Same integer arithmetic per pixel as the stand-alone kernel: layer 1 is bit-identical to letterbox + stem + conv.
struct FrontRS { int rows, chunks; double sx, sy; };     // staged source rows per tile, 16-byte chunks per staged row, source / tensor scale
#define FF_RS_NRAW 8              // raw chunks per thread in resize mode: rows x chunks <= 4 096 (1280 x 720 -> 640 x 360 stages 140 rows x 28 chunks = 3 920); more spills registers

template <bool U8, bool DIAG = false, bool RS = false>
__global__ __launch_bounds__(FF_NW * 64, FF_NW == 8 ? 1 : 2) void front_fused_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w0, const float* __restrict__ b0,
                                                             const uint4* __restrict__ w1, const float* __restrict__ b1, uint16_t* __restrict__ y,
                                                             int B, int H, int Wp, int H0, int W0, int H1, int W1, int kw8_0, int kw8_1, int out_cs,
                                                             int out_co, int tiles_x, int tiles_y, const uint8_t* __restrict__ src8, LetterboxGeom g, long long* dbg, int abl_arg, FrontRS rs) {
    const int abl = DIAG ? abl_arg : 0;        // the production instance folds every ablation branch away (they fragment the MFMA loops)
    // abl (VC_FF_ABLATE, diagnostics with WRONG results; 0 in production): 1 no transcendentals, 2 no stem MFMAs, 4 no stem LDS stores,
    // 8 no output stores, 16 no stem phase, 32 no conv phase, 64 no patch writes, 128 no global fetch, 256 no patch reads in the stem
    __shared__ uint4 patch[FF_PR * FF_PP];                 // 40.3 KB
    __shared__ uint4 l0t[FF_NT0 * 16 * 4];                 // 72.7 KB: [pixel slot][4 chunks], swizzled
    __shared__ uint4 w1s[9 * 4 * 64];                      // 36.9 KB: layer-1 weights, [tap][channel tile][lane] = one fragment load per wave
    __shared__ int4 rs_col[RS ? 2 * FF_PC + 2 : 1];        // resize mode: per tensor column of the patch (byte offset of tap 0 / tap 1 in a staged row, a0, a1 or -1)
    __shared__ int4 rs_row[RS ? FF_PR + 2 : 1];            // per tensor row of the patch (byte offset of source row y0 / y1 in the staging area, b0, b1 or -1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    // weights in registers for the whole launch
    ChunkF wf0[5][2];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wf0[s][ct].u = w0[(size_t)(ct * 16 + col) * kw8_0 + 4 * s + kq];
    for (int i = threadIdx.x; i < 9 * 4 * 64; i += FF_NW * 64) {     // fragment (tap t, channel tile ct) of lane l = chunk 4t + l/16 of channel 16ct + l%16
        const int l = i & 63, ct = (i >> 6) & 3, t = i >> 8;
        w1s[i] = w1[(size_t)(ct * 16 + (l & 15)) * kw8_1 + 4 * t + (l >> 4)];
    }
    // conv phase: a wave owns two output rows (row pair rp) and CTW of the 4 channel tiles (from ct0)
    constexpr int RP = FF_TH / 2, CS = FF_NW / RP, CTW = 4 / CS;
    static_assert(FF_TH % 2 == 0 && FF_NW % RP == 0 && (CS == 1 || CS == 2 || CS == 4), "conv phase split");
    const int rp = wave % RP, ct0 = (wave / RP) * CTW;
    float4 bv0[2], bv1[CTW];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv0[ct] = *(const float4*)(b0 + ct * 16 + kq * 4);
#pragma unroll
    for (int c = 0; c < CTW; ++c) bv1[c] = *(const float4*)(b1 + (ct0 + c) * 16 + kq * 4);
    int koff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int q = min(4 * s + kq, 17);                  // padded chunks multiply zero weights
        koff[s] = (q / 3) * FF_PP + (q % 3);
    }
    const int ntiles = B * tiles_y * tiles_x;
    constexpr int NT = FF_NW * 64;
    constexpr int NPRE = (FF_PR * FF_PC + NT - 1) / NT;    // 5 chunks per thread
    // The next tile's patch travels through registers while this tile is computed.  U8: the registers hold the RAW source bytes
    // (three 16-bit loads per pixel pair) and the conversion (pad 114, R/B swap, exact /255, RNE to bf16) happens when the patch is
    // written to LDS one tile later -- converting at fetch time would make every wave wait for its loads right there.
    uint4 pre[NPRE];
    // U8: the patch's source rows are fetched as ALIGNED 16-byte chunks (70 rows x 15 chunks = 1 050 loads per tile, two or three per
    // thread) instead of three 16-bit loads per pixel pair (7 350 per tile): the 2-byte-granular loads occupied the texture-address
    // unit for ~1 lane per cycle and were the largest single item of the kernel (tools/ff_ablate.py: no fetch = -0.17 of 0.67 ms).
    // The raw bytes wait in registers during the tile's compute like `pre` did, go to a raw staging area in LDS at the top of the next
    // tile (aliased with the layer-0 tile, which is dead between a tile's conv phase and the next tile's stem phase) and are converted
    // from there.  Out-of-buffer chunks (before the first / after the last frame) are buffer loads past num_records: zeros.
    constexpr int NRAW = RS ? FF_RS_NRAW : (FF_PR * FF_RAWC + NT - 1) / NT;   // 3 (resize mode: 10)
    uint4 praw[NRAW];
    const __amdgpu_buffer_rsrc_t s8rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src8), 0, U8 ? (int)(((size_t)B * g.src_h * g.src_w * 3 + 3) & ~(size_t)3) : 0, 0x00020000);   // whole dwords: the range check is per dword, and the last pixels of an odd-sized clip share theirs with up to 3 bytes past the end (same page: the engine requires a 4-byte-aligned base for this kernel)
    auto row_start = [&](int b, int gy0, int gx0, int pr, bool& valid) -> long long {       // byte offset of the source pixel under the patch row's first pixel
        const int uy = 2 * gy0 - 2 + pr - g.top;
        valid = uy >= 0 && uy < g.unpad_h;
        return (((long long)b * g.src_h + uy) * g.src_w + (2 * (gx0 - 1) - g.left)) * 3;
    };
    // resize mode: first source row / column a tile samples = tap 0 of its first tensor row / column that lies inside the resized image
    auto rs_origin = [&](int gy0, int gx0, int& srow0, int& scol0) {
        const int uy = min(max(2 * gy0 - 2 - g.top, 0), g.unpad_h - 1), ux = min(max(2 * (gx0 - 1) - g.left, 0), g.unpad_w - 1);
        int s1, c0, c1;
        lin_coef(uy, g.src_h, rs.sy, srow0, s1, c0, c1, false);
        lin_coef(ux, g.src_w, rs.sx, scol0, s1, c0, c1, true);
    };
    auto fetch_raw = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int gy0 = 2 * ty * FF_TH - 1, gx0 = 2 * tx * FF_TW - 1;
        if constexpr (RS) {
            int srow0, scol0;
            rs_origin(gy0, gx0, srow0, scol0);
            const int total = rs.rows * rs.chunks;
#pragma unroll
            for (int k = 0; k < NRAW; ++k) {
                const int c = threadIdx.x + k * NT;
                const int r = c / rs.chunks, j = c - r * rs.chunks;
                const int sr = min(srow0 + r, g.src_h - 1);                            // rows past the image are never sampled (taps clamp): any valid row will do
                const long long rs0 = (((long long)b * g.src_h + sr) * g.src_w + scol0) * 3;
                typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
                u32x4r v = {0u, 0u, 0u, 0u};
                if (c < total && !(abl & 128)) v = __builtin_amdgcn_raw_buffer_load_b128(s8rd, (int)(((rs0 >> 4) << 4) + 16 * j), 0, 0);
                praw[k] = make_uint4(v.x, v.y, v.z, v.w);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NRAW; ++k) {
            const int c = threadIdx.x + k * NT;
            const int pr = c / FF_RAWC, j = c - pr * FF_RAWC;
            bool valid;
            const long long rs = row_start(b, gy0, gx0, pr, valid);
            const long long off = ((rs >> 4) << 4) + 16 * j;                       // floor to 16 bytes (arithmetic shift: also for rs < 0)
            typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
            u32x4r v = {0u, 0u, 0u, 0u};
            if (c < FF_PR * FF_RAWC && valid && !(abl & 128)) v = __builtin_amdgcn_raw_buffer_load_b128(s8rd, (int)off, 0, 0);
            praw[k] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    auto fetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int gy0 = 2 * ty * FF_TH - 1, gx0 = 2 * tx * FF_TW - 1;          // first layer-0 row / column of the region
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * NT;
            const int pr = i / FF_PC, pc = i - pr * FF_PC;
            const int iy = 2 * gy0 - 2 + pr, ip = gx0 - 1 + pc;
            pre[k] = make_uint4(0u, 0u, 0u, 0u);                            // .w = 0: outside the network input (zero padding)
            if (i < FF_PR * FF_PC && iy >= 0 && iy < H && ip >= 0 && ip < Wp && !(abl & 128)) {
                if constexpr (!U8) {
                    pre[k] = x[((size_t)b * H + iy) * Wp + ip];
                } else {
                    const int uy = iy - g.top, ux = 2 * ip - g.left;
                    pre[k].w = 1u;                                          // inside the input, letterbox padding (114) unless the bytes say otherwise
                    if (uy >= 0 && uy < g.unpad_h && ux >= 0 && ux + 1 < g.unpad_w) {
                        // (one 2-byte-aligned 8-byte load per pair instead of three 16-bit loads was measured 10 % SLOWER: 0.62 -> 0.68 ms)
                        const uint16_t* q = (const uint16_t*)(src8 + (((size_t)b * g.src_h + uy) * g.src_w + ux) * 3);
                        pre[k].x = q[0]; pre[k].y = q[1]; pre[k].z = q[2];
                        pre[k].w = 2u;
                    }
                }
            }
        }
    };
    auto patch_chunk = [&](const uint4& r) -> uint4 {
        if constexpr (!U8) return r;
        if (r.w == 0u) return make_uint4(0u, 0u, 0u, 0u);
        int pv[6] = {114, 114, 114, 114, 114, 114};
        if (r.w == 2u) { pv[0] = r.x & 255; pv[1] = r.x >> 8; pv[2] = r.y & 255; pv[3] = r.y >> 8; pv[4] = r.z & 255; pv[5] = r.z >> 8; }
        const int a0 = g.swap_rb ? pv[2] : pv[0], a2 = g.swap_rb ? pv[0] : pv[2];
        const int c0 = g.swap_rb ? pv[5] : pv[3], c2 = g.swap_rb ? pv[3] : pv[5];
        uint4 o;
        o.x = pack2_bf16(div255_exact((float)a0), div255_exact((float)pv[1]));
        o.y = pack2_bf16(div255_exact((float)a2), 0.f);
        o.z = pack2_bf16(div255_exact((float)c0), div255_exact((float)pv[4]));
        o.w = pack2_bf16(div255_exact((float)c2), 0.f);
        return o;
    };
    typedef unsigned int u32x2f __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2f __attribute__((ext_vector_type(2)));
    const bool odd = (kq & 1) != 0;
    char* l0b = (char*)l0t;
    if ((int)blockIdx.x < ntiles) { if constexpr (U8) fetch_raw(blockIdx.x); else fetch(blockIdx.x); }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int oy0 = ty * FF_TH, ox0 = tx * FF_TW;
        const int gy0 = 2 * oy0 - 1, gx0 = 2 * ox0 - 1;
        long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
        if (dbg) ts0 = wall_clock64();
        __syncthreads();                                    // the previous tile's LDS reads are done
        if constexpr (U8 && RS) {
            char* rawb = (char*)l0t;                        // staged source rows: [rs.rows][rs.chunks * 16] bytes in the (dead) layer-0 tile
            const int total = rs.rows * rs.chunks, pitch = rs.chunks * 16;
#pragma unroll
            for (int k = 0; k < NRAW; ++k) {
                const int c = threadIdx.x + k * NT;
                if (c < total) *(uint4*)(rawb + c * 16) = praw[k];
            }
            int srow0, scol0;
            rs_origin(gy0, gx0, srow0, scol0);
            if (threadIdx.x < 2 * FF_PC) {                  // tensor column j of the patch: taps relative to the staged row's first fetched pixel
                const int j = threadIdx.x, ix = 2 * (gx0 - 1) + j, ux = ix - g.left;
                int4 e = make_int4(0, 0, -1, -1);           // a1 = -1: letterbox padding (114) or outside the tensor
                if (ix >= 0 && ix < 2 * Wp && ux >= 0 && ux < g.unpad_w) {
                    int x0, x1, a0, a1;
                    lin_coef(ux, g.src_w, rs.sx, x0, x1, a0, a1, true);
                    e = make_int4((x0 - scol0) * 3, (x1 - scol0) * 3, a0, a1);
                }
                rs_col[j] = e;
            } else if (threadIdx.x >= 128 && threadIdx.x < 128 + FF_PR) {
                const int pr = threadIdx.x - 128, iy = 2 * gy0 - 2 + pr, uy = iy - g.top;
                int4 e = make_int4(0, 0, -1, -1);
                if (iy >= 0 && iy < H && uy >= 0 && uy < g.unpad_h) {
                    int y0, y1, c0, c1;
                    lin_coef(uy, g.src_h, rs.sy, y0, y1, c0, c1, false);
                    // byte position of source pixel scol0 of a staged row inside its first 16-byte chunk: differs from row to row unless src_w * 3 is a multiple of 16
                    const long long q0 = (((long long)b * g.src_h + y0) * g.src_w + scol0) * 3, q1 = (((long long)b * g.src_h + y1) * g.src_w + scol0) * 3;
                    e = make_int4((y0 - srow0) * pitch + (int)(q0 & 15), (y1 - srow0) * pitch + (int)(q1 & 15), c0, c1);
                }
                rs_row[pr] = e;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int i = threadIdx.x + k * NT;
                if (i < FF_PR * FF_PC && !(abl & 64)) {
                    const int pr = i / FF_PC, pc = i - pr * FF_PC;
                    const int iy = 2 * gy0 - 2 + pr, ip = gx0 - 1 + pc;
                    uint4 o = make_uint4(0u, 0u, 0u, 0u);                        // outside the network input: the stem's zero padding
                    if (iy >= 0 && iy < H && ip >= 0 && ip < Wp) {
                        const int4 rw = rs_row[pr];
                        int pv[6] = {114, 114, 114, 114, 114, 114};
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int4 cl = rs_col[2 * pc + h];
                            if (rw.w >= 0 && cl.w >= 0) {
                                // (two aligned dword reads per source row and a funnel shift instead of six byte reads: measured no faster -- 0.477
                                // against 0.460 ms per 128 frames at 1280 x 720; the kernel's extra time over the same-scale mode is the 3.7 x larger raw
                                // fetch per tile, not these reads -- and a row offset of 3 mod 4 needs a third dword; removed)
                                const uint8_t* r0 = (const uint8_t*)rawb + rw.x;
                                const uint8_t* r1 = (const uint8_t*)rawb + rw.y;
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    const int h0 = r0[cl.x + c] * cl.z + r0[cl.y + c] * cl.w;
                                    const int h1 = r1[cl.x + c] * cl.z + r1[cl.y + c] * cl.w;
                                    const int v = (((rw.z * (h0 >> 4)) >> 16) + ((rw.w * (h1 >> 4)) >> 16) + 2) >> 2;
                                    pv[3 * h + c] = min(max(v, 0), 255);
                                }
                                if (g.swap_rb) { const int tsw = pv[3 * h]; pv[3 * h] = pv[3 * h + 2]; pv[3 * h + 2] = tsw; }
                            }
                        }
                        o.x = pack2_bf16(div255_exact((float)pv[0]), div255_exact((float)pv[1]));
                        o.y = pack2_bf16(div255_exact((float)pv[2]), 0.f);
                        o.z = pack2_bf16(div255_exact((float)pv[3]), div255_exact((float)pv[4]));
                        o.w = pack2_bf16(div255_exact((float)pv[5]), 0.f);
                    }
                    patch[pr * FF_PP + pc] = o;
                }
            }
        } else if constexpr (U8) {
            char* rawb = (char*)l0t;                        // raw source rows: [FF_PR][FF_RAWP] bytes in the (dead) layer-0 tile
#pragma unroll
            for (int k = 0; k < NRAW; ++k) {
                const int c = threadIdx.x + k * NT;
                if (c < FF_PR * FF_RAWC) *(uint4*)(rawb + c * 16) = praw[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int i = threadIdx.x + k * NT;
                if (i < FF_PR * FF_PC && !(abl & 64)) {
                    const int pr = i / FF_PC, pc = i - pr * FF_PC;
                    const int iy = 2 * gy0 - 2 + pr, ip = gx0 - 1 + pc;
                    uint4 r = make_uint4(0u, 0u, 0u, 0u);                        // .w = 0: outside the network input (zero padding)
                    if (iy >= 0 && iy < H && ip >= 0 && ip < Wp) {
                        const int ux = 2 * ip - g.left;
                        bool valid;
                        const long long rs = row_start(b, gy0, gx0, pr, valid);
                        r.w = 1u;                                                // inside the input: letterbox padding (114) unless the bytes say otherwise
                        if (valid && ux >= 0 && ux + 1 < g.unpad_w) {
                            const uint16_t* q = (const uint16_t*)(rawb + pr * FF_RAWP + (int)(rs - ((rs >> 4) << 4)) + 6 * pc);
                            r.x = q[0]; r.y = q[1]; r.z = q[2]; r.w = 2u;
                        }
                    }
                    patch[pr * FF_PP + pc] = patch_chunk(r);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int i = threadIdx.x + k * NT;
                if (i < FF_PR * FF_PC && !(abl & 64)) { const int pr = i / FF_PC; patch[pr * FF_PP + (i - pr * FF_PC)] = patch_chunk(pre[k]); }
            }
        }
        __syncthreads();
        if (dbg) ts1 = wall_clock64();
        if (t + (int)gridDim.x < ntiles) { if constexpr (U8) fetch_raw(t + gridDim.x); else fetch(t + gridDim.x); }
        // ---- layer 0 on the region.  Pixel tiles are ROW-ALIGNED so that a tile's coordinates cost no division: tile r < 33 = row r,
        // even columns 0 .. 30; tile 33 + r = row r, odd columns 1 .. 31; tiles 66 .. 68 = the 33 pixels of column 32, one row per lane.
        // LDS slot of a pixel: plane (column parity) * 561 + row * 17 + column / 2, as before. ------------------------------------------
        const bool interior = gy0 >= 0 && gy0 + FF_RH <= H0 && gx0 >= 0 && gx0 + 2 * FF_TW + 1 <= W0;     // block-uniform: no zero padding of layer 0 in this tile
        // (four pixel tiles per pass and wave -- twice the independent work per dependency chain -- measured 0.535 against 0.528 ms: the passes
        // already overlap)
        for (int tb = wave * 2; tb < FF_NT0R && !(abl & 16); tb += 2 * FF_NW) {    // two pixel tiles per pass and wave
            int ly[2], lx[2], slot[2];
            bool live[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int tq = tb + q;
                if (tq < 2 * FF_RH) {
                    const int pl = tq >= FF_RH ? 1 : 0;
                    ly[q] = tq - pl * FF_RH; lx[q] = 2 * col + pl; live[q] = true;
                } else {
                    const int r = (tq - 2 * FF_RH) * 16 + col;
                    ly[q] = min(r, FF_RH - 1); lx[q] = 2 * FF_TW; live[q] = tq < FF_NT0R && r < FF_RH;
                }
                slot[q] = (lx[q] & 1) * FF_PLANE + ly[q] * FF_PW + (lx[q] >> 1);
            }
            f32x4f acc[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[ct][q] = (f32x4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                ChunkF xf[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) xf[q].u = (abl & 256) ? make_uint4(ly[q], lx[q], s, q) : patch[(2 * ly[q]) * FF_PP + lx[q] + koff[s]];
                if (abl & 2) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) { acc[0][q][0] += __uint_as_float(xf[q].u.x); acc[1][q][1] += __uint_as_float(xf[q].u.w); }
                    continue;
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[s][ct].h, xf[q].h, acc[ct][q], 0, 0, 0);
            }
            // bias + SiLU + bf16; lane pairs swap halves across the two pixel tiles (conv_epilogue_bf16): 16 bytes = 8 channels of one pixel
            bool inside[2] = {true, true};
            if (!interior) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // layer 1 pads layer 0 with ZEROS: region pixels outside the layer-0 map hold 0, not SiLU(bias)
                    const int gy = gy0 + ly[q], gx = gx0 + lx[q];
                    inside[q] = gy >= 0 && gy < H0 && gx >= 0 && gx < W0;
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                uint2 P[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x2f lo = (f32x2f){acc[ct][q][0], acc[ct][q][1]} + (f32x2f){bv0[ct].x, bv0[ct].y};
                    f32x2f hi = (f32x2f){acc[ct][q][2], acc[ct][q][3]} + (f32x2f){bv0[ct].z, bv0[ct].w};
                    if (!(abl & 1)) { lo = ff_silu2(lo); hi = ff_silu2(hi); }
                    const bf16x2f p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                    P[q].x = inside[q] ? __builtin_bit_cast(uint32_t, p0) : 0u; P[q].y = inside[q] ? __builtin_bit_cast(uint32_t, p1) : 0u;
                }
                const u32x2f sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
                const u32x2f sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
                const uint4 o4 = make_uint4(sx.x, sy.x, sx.y, sy.y);
                const int q = odd ? 1 : 0;                   // even lanes keep tile 0's pixel, odd lanes tile 1's (same `col`)
                // the pair (lane, lane ^ 16) shares `col`, hence the same pixel of each tile; liveness and slot are per lane's own tile
                if (live[q] && !(abl & 4)) *(uint4*)(l0b + ff_l0_addr(slot[q], ct * 2 + (kq >> 1))) = o4;
            }
        }
        if (dbg) ts2 = wall_clock64();
        __syncthreads();
        // ---- layer 1 from the LDS tile: a wave owns two output rows (one pixel tile each) and CTW of the 4 channel tiles (all of them when
        // there are as many waves as row pairs); weights from LDS ---
        if (!(abl & 32)) {
            f32x4f acc[CTW][2];
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[c][q] = (f32x4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int r = tp / 3, s = tp % 3;
                ChunkF xf[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int oyl = rp * 2 + q;
                    const int px = (s & 1) * FF_PLANE + (2 * oyl + r) * FF_PW + col + (s >> 1);
                    xf[q].u = *(const uint4*)(l0b + ff_l0_addr(px, kq));
                }
#pragma unroll
                for (int c = 0; c < CTW; ++c) {
                    ChunkF wv;
                    wv.u = w1s[(tp * 4 + ct0 + c) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv.h, xf[q].h, acc[c][q], 0, 0, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int ct = ct0 + c;
                const float4 bb = bv1[c];
                uint2 P[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x2f lo = (f32x2f){acc[c][q][0], acc[c][q][1]} + (f32x2f){bb.x, bb.y};
                    f32x2f hi = (f32x2f){acc[c][q][2], acc[c][q][3]} + (f32x2f){bb.z, bb.w};
                    if (!(abl & 1)) { lo = ff_silu2(lo); hi = ff_silu2(hi); }
                    const bf16x2f p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                    P[q].x = __builtin_bit_cast(uint32_t, p0); P[q].y = __builtin_bit_cast(uint32_t, p1);
                }
                const u32x2f sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
                const u32x2f sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
                const uint4 o4 = make_uint4(sx.x, sy.x, sx.y, sy.y);
                const int oy = oy0 + rp * 2 + (odd ? 1 : 0), ox = ox0 + col;
                if (oy < H1 && ox < W1 && !(abl & 8))
                    *(uint4*)(y + (((size_t)b * H1 + oy) * W1 + ox) * out_cs + out_co + ct * 16 + (kq & ~1) * 4) = o4;
            }
        }
        if (dbg && lane == 0) {                            // VC_FF_DBG: per-wave phase sums (100 MHz ticks): patch, stem, barrier wait, conv
            ts3 = wall_clock64();
            long long* d = dbg + ((size_t)blockIdx.x * FF_NW + wave) * 8;
            d[0] += ts1 - ts0; d[1] += ts2 - ts1; d[3] += ts3 - ts2;
        }
    }
}

// p0 = the stem as launch_conv sees it (stem_direct_applicable), p1 = the conv that consumes its output
bool front_fused_applicable(const ConvP& p0, const ConvP& p1) {
    if (!stem_direct_applicable(p0) || p0.Cout != 32) return false;
    return p1.prec == PREC_BF16 && p1.in == p0.out && p1.in_cs == p0.out_cs && p1.in_co == p0.out_co && p1.Cin == 32 && p1.Cout == 64 && p1.kh == 3 &&
           p1.kw == 3 && p1.sh == 2 && p1.sw == 2 && p1.ph == 1 && p1.pw == 1 && p1.act == ACT_SILU && p1.res_mode == RES_NONE && !p1.out_f32 && p1.split == 0 &&
           p1.B == p0.B && p1.H == p0.Ho && p1.W == p0.Wo && p1.K == 288 && p1.out_cs % 8 == 0 && p1.out_co % 8 == 0;
}

// Resize mode: what a tile has to stage.  For every tile row / column of the layer-1 output the source rows / columns its 70 tensor rows
// / columns sample, with the kernel's own lin_coef (host and device agree bit for bit); the staging area is the layer-0 tile.
static bool front_rs_plan(const LetterboxGeom& g, FrontRS& rs) {
    if (g.unpad_h < 1 || g.unpad_w < 1 || g.net_h % 4 || g.net_w % 4) return false;
    rs.sx = 1.0 / ((double)g.unpad_w / (double)g.src_w); rs.sy = 1.0 / ((double)g.unpad_h / (double)g.src_h);   // letterbox_kernel's scales
    const int H1 = g.net_h / 4, W1 = g.net_w / 4;
    const int tiles_x = (W1 + FF_TW - 1) / FF_TW, tiles_y = (H1 + FF_TH - 1) / FF_TH;
    int rows = 1, bytes = 16;
    for (int ty = 0; ty < tiles_y; ++ty) {
        const int gy0 = 2 * ty * FF_TH - 1;
        const int u0 = std::min(std::max(2 * gy0 - 2 - g.top, 0), g.unpad_h - 1), u1 = std::min(std::max(2 * gy0 - 2 + FF_PR - 1 - g.top, 0), g.unpad_h - 1);
        int a0, a1, b0, b1, c0, c1;
        lin_coef(u0, g.src_h, rs.sy, a0, a1, c0, c1, false);
        lin_coef(u1, g.src_h, rs.sy, b0, b1, c0, c1, false);
        rows = std::max(rows, b1 - a0 + 1);
    }
    for (int tx = 0; tx < tiles_x; ++tx) {
        const int gx0 = 2 * tx * FF_TW - 1;
        const int u0 = std::min(std::max(2 * (gx0 - 1) - g.left, 0), g.unpad_w - 1), u1 = std::min(std::max(2 * (gx0 - 1) + 2 * FF_PC - 1 - g.left, 0), g.unpad_w - 1);
        int a0, a1, b0, b1, c0, c1;
        lin_coef(u0, g.src_w, rs.sx, a0, a1, c0, c1, true);
        lin_coef(u1, g.src_w, rs.sx, b0, b1, c0, c1, true);
        bytes = std::max(bytes, (b1 - a0 + 1) * 3 + 15);           // + the worst position of the first pixel inside its 16-byte chunk
    }
    rs.rows = rows; rs.chunks = (bytes + 15) / 16;
    // monotone taps: lin_coef's s0 / s1 never decrease with d, so the first / last tensor row bound every row in between
    return (long)rs.rows * rs.chunks <= (long)FF_RS_NRAW * FF_NW * 64 && (size_t)rs.rows * rs.chunks * 16 <= sizeof(uint4) * FF_NT0 * 16 * 4;
}

bool front_fused_resize_ok(const LetterboxGeom& g) {
    FrontRS rs;
    return front_rs_plan(g, rs);
}

int launch_front_fused(const ConvP& p0, const ConvP& p1, const uint8_t* src8, const LetterboxGeom& g, hipStream_t s) {
    const int tiles_x = (p1.Wo + FF_TW - 1) / FF_TW, tiles_y = (p1.Ho + FF_TH - 1) / FF_TH;
    const int ntiles = p1.B * tiles_x * tiles_y;
    static const int slots_reserve = getenv("VC_CONV_RESERVE") ? atoi(getenv("VC_CONV_RESERVE")) : 64;
    static const int slots_hw = [] {
        int per_cu = 1, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, front_fused_kernel<true>, FF_NW * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        return per_cu * cus;
    }();
    // (tiles handed out by an atomic counter instead of the static stride -- late-starting workgroups, e.g. behind the tracker's, would not carry
    // a full share -- measured 7.09 against 7.13 ms per bench step over five alternating runs: inside the spread, not kept)
    const int grid = std::min(ntiles, std::max(256, slots_hw - slots_reserve));   // persistent: every workgroup walks tiles; at one workgroup per CU nothing is held back (as in launch_one)
    const uint4* x = (const uint4*)p0.in;
    uint16_t* y = (uint16_t*)p1.out;
    static const bool dbg_on = getenv("VC_FF_DBG") != nullptr;
    const int abl = p0.ablate;                                                        // diagnostics only (engine option "ff_ablate", tools/ff_ablate.py)
    long long* dbg = nullptr;
    if (dbg_on && hipMalloc((void**)&dbg, (size_t)grid * FF_NW * 64) == hipSuccess) hipMemsetAsync(dbg, 0, (size_t)grid * FF_NW * 64, s);
    const bool resize = src8 && !(g.unpad_h == g.src_h && g.unpad_w == g.src_w && g.src_w % 2 == 0 && g.left % 2 == 0);
    FrontRS rs{0, 0, 1.0, 1.0};
    if (resize && !front_rs_plan(g, rs)) { set_error("front_fused: the resize geometry %dx%d -> %dx%d does not fit the staging area", g.src_h, g.src_w, g.unpad_h, g.unpad_w); return VC_ERR_ARG; }
#define FF_ARGS x, (const uint4*)p0.w, p0.bias, (const uint4*)p1.w, p1.bias, y, p0.B, p0.H, p0.W, p0.Ho, p0.Wo, p1.Ho, p1.Wo, p0.Kp / 8, p1.Kp / 8, p1.out_cs, p1.out_co, tiles_x, tiles_y, src8, g, dbg, abl, rs
    if (resize) launch_timed(p0, front_fused_kernel<true, false, true>, dim3(grid), dim3(FF_NW * 64), 0, s, FF_ARGS);
    else if (src8 && abl) launch_timed(p0, front_fused_kernel<true, true>, dim3(grid), dim3(FF_NW * 64), 0, s, FF_ARGS);
    else if (src8) launch_timed(p0, front_fused_kernel<true>, dim3(grid), dim3(FF_NW * 64), 0, s, FF_ARGS);
    else launch_timed(p0, front_fused_kernel<false>, dim3(grid), dim3(FF_NW * 64), 0, s, FF_ARGS);
#undef FF_ARGS
    VC_HIP(hipGetLastError());
    if (dbg) {
        hipStreamSynchronize(s);
        std::vector<long long> h((size_t)grid * FF_NW * 8);
        hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        hipFree(dbg);
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < h.size(); ++i) a[i & 7] += (double)h[i];
        const double per = (double)ntiles / grid * grid * FF_NW;      // (tiles per workgroup) x waves
        fprintf(stderr, "[vc ff dbg] %d tiles on %d workgroups; us per tile and wave: patch %.2f stem %.2f conv %.2f\n", ntiles, grid, a[0] / per / 100.0, a[1] / per / 100.0, a[3] / per / 100.0);
    }
    return VC_OK;
}

}  // namespace vc
