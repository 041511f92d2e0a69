// YOLOv5's first C3 block in one kernel (bf16): C3(64 -> 64, n = 1, shortcut) at 160 x 160 for a 640 x 640 input -- models/common.py::C3 /
// Bottleneck of ultralytics/yolov5 v6.0 (yolov5s.yaml layer 2 at width 0.5), reached from /root/reference/networks/yolo.py:70:
//     y1 = SiLU(cv1 x)   y2 = SiLU(cv2 x)          1x1, 64 -> 32 each (one fused launch in the unfused graph: engine.hip::yolo_c3)
//     b1 = SiLU(m.cv1 y1)                           1x1, 32 -> 32
//     m  = y1 + SiLU(m.cv2 b1)                      3x3 / pad 1, 32 -> 32, shortcut added after the activation
//     out = SiLU(cv3 [m | y2])                      1x1, 64 -> 64
// Unfused, the four launches move 2.5 GB per 128 frames (x in, y1 / y2 / b1 / m out and back in, out) and every one of them sits on
// the HBM roof; here a workgroup owns a 8 x 16 tile of output pixels, keeps y1 and b1 on the tile's 10 x 18 halo region, y2 and m on
// the tile itself in LDS, and touches HBM for x (with halo: 1.4 x) and the output only: 0.5 + 0.42 GB per 128 frames.
//
// Occupancy is the design constraint ("SiLU is two quarter-rate transcendentals per value"): one big workgroup per CU runs
// all its waves through the same phase at the same time and leaves the epilogues' v_exp / v_rcp exposed.  So: FOUR waves per workgroup,
// 80 KB of LDS, two workgroups per CU that drift apart -- one's MFMA / LDS phases cover the other's epilogues.
//
// LDS (81 920 B): x as two 32-channel planes on the halo region (24 KB; plane 0 is reused for b1 once cv1 / cv2 are done),
// y1 on the halo region (12 KB; m overwrites y1 in place on the interior), y2 on the interior (8 KB), all weights in MFMA fragment
// order (36 KB).  Every 32-channel tensor uses front_fused.hip's pixel layout (64 bytes per pixel, 16-byte chunks XOR-swizzled).
// MFMA operand order and k order equal conv_igemm_kernel's for all four convolutions, the epilogues are conv_epilogue_bf16's
// expressions: bit-identical to the unfused path (tests/test_gpu_nets.py::test_c3_fused_bit_identical).
#include <algorithm>

#include "kernels.h"

namespace vc {

typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef float f32x2c __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2c __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));
union ChunkC { uint4 u; bf16x8c h; };

#define C3_TH 8
#define C3_TW 16
#define C3_RW (C3_TW + 2)              // 18
#define C3_NH ((C3_TH + 2) * C3_RW)    // 180 halo pixels
#define C3_NT ((C3_NH + 15) / 16)      // 12 pixel tiles of the halo region
#define C3_NP (C3_NT * 16)             // 192 pixel slots
#define C3_NW 4

__device__ __forceinline__ int c3_addr(int px, int chunk) { return (px * 4 + (chunk ^ ((px >> 1) & 2))) * 16; }   // byte offset, 64-byte pixels
__device__ __forceinline__ f32x2c c3_silu2(f32x2c x) {
    const f32x2c t = x * (f32x2c){-1.442695040888963387f, -1.442695040888963387f};
    const f32x2c d = (f32x2c){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (f32x2c){1.0f, 1.0f};
    return x * (f32x2c){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// bias + SiLU on the four accumulator values of a lane -> 4 bf16 (8 bytes)
template <bool NOTRANS = false>
__device__ __forceinline__ uint2 c3_act4(const f32x4c& a, const float4& b) {
    f32x2c lo = (f32x2c){a[0], a[1]} + (f32x2c){b.x, b.y};
    f32x2c hi = (f32x2c){a[2], a[3]} + (f32x2c){b.z, b.w};
    if constexpr (!NOTRANS) { lo = c3_silu2(lo); hi = c3_silu2(hi); }
    const bf16x2c p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
    return make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
}

struct C3Args {
    const uint4 *w12, *wm1, *wm2, *w3;
    const float *b12, *bm1, *bm2, *b3;
    int kw12, kwm1, kwm2, kw3;             // weight row strides in 16-byte chunks
    const uint16_t* x; int in_cs, in_co;
    uint16_t* y; int out_cs, out_co;
    int B, H, W, tiles_x, tiles_y;
    int abl;                               // diagnostics (VC_C3_ABLATE, wrong results): 1 no transcendentals, 2 no global fetch, 4 no output stores,
};                                         // 8 no 3x3 pass, 16 no cv12 pass, 32 no m.cv1 pass, 64 no cv3 pass

template <bool DIAG>
__global__ __launch_bounds__(C3_NW * 64, 2) void c3_fused_kernel(const C3Args a) {
    const int abl = DIAG ? a.abl : 0;
    __shared__ uint4 xa[2 * C3_NP * 4];                    // 24 KB: x planes [k step][pixel slot][4 chunks]; plane 0 becomes b1
    __shared__ uint4 yb[C3_NP * 4];                        // 12 KB: y1 on the halo region, m in place on the interior
    __shared__ uint4 yc[C3_TH * C3_TW * 4];                // 8 KB: y2 on the interior
    __shared__ uint4 w12s[2 * 4 * 64], wm1s[2 * 64], wm2s[9 * 2 * 64], w3s[2 * 4 * 64];     // 36 KB, [k step][channel tile][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    constexpr int NT = C3_NW * 64;
    // fragment (k step s, channel tile ct) of lane l = chunk 4 s + l / 16 of channel 16 ct + l % 16
    for (int i = threadIdx.x; i < 2 * 4 * 64; i += NT) {
        const int l = i & 63, ct = (i >> 6) & 3, s = i >> 8;
        w12s[i] = a.w12[(size_t)(ct * 16 + (l & 15)) * a.kw12 + 4 * s + (l >> 4)];
        w3s[i] = a.w3[(size_t)(ct * 16 + (l & 15)) * a.kw3 + 4 * s + (l >> 4)];
    }
    for (int i = threadIdx.x; i < 2 * 64; i += NT) {
        const int l = i & 63, ct = i >> 6;
        wm1s[i] = a.wm1[(size_t)(ct * 16 + (l & 15)) * a.kwm1 + (l >> 4)];
    }
    for (int i = threadIdx.x; i < 9 * 2 * 64; i += NT) {
        const int l = i & 63, ct = (i >> 6) & 1, t = i >> 7;
        wm2s[i] = a.wm2[(size_t)(ct * 16 + (l & 15)) * a.kwm2 + 4 * t + (l >> 4)];
    }
    float4 b12[4], bm1[2], bm2[2], b3[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { b12[ct] = *(const float4*)(a.b12 + ct * 16 + kq * 4); b3[ct] = *(const float4*)(a.b3 + ct * 16 + kq * 4); }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) { bm1[ct] = *(const float4*)(a.bm1 + ct * 16 + kq * 4); bm2[ct] = *(const float4*)(a.bm2 + ct * 16 + kq * 4); }

    auto act4 = [&](const f32x4c& v, const float4& bb) -> uint2 { if (DIAG && (abl & 1)) return c3_act4<true>(v, bb); return c3_act4<false>(v, bb); };
    const int ntiles = a.B * a.tiles_y * a.tiles_x;
    constexpr int NPRE = (C3_NH * 8 + NT - 1) / NT;        // 6 chunks of the x halo tile per thread
    uint4 pre[NPRE];
    auto fetch = [&](int t) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
        const int y0 = ty * C3_TH - 1, x0 = tx * C3_TW - 1;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * NT;
            const int n = i >> 3, c = i & 7;
            const int ry = (n * 3641) >> 16, rx = n - ry * C3_RW;             // n / 18 for n < 192
            const int gy = y0 + ry, gx = x0 + rx;
            pre[k] = make_uint4(0u, 0u, 0u, 0u);
            if (n < C3_NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && !(abl & 2))
                pre[k] = *(const uint4*)(a.x + (((size_t)b * a.H + gy) * a.W + gx) * a.in_cs + a.in_co + c * 8);
        }
    };
    char* xab = (char*)xa;
    char* ybb = (char*)yb;
    char* ycb = (char*)yc;
    const bool odd = (kq & 1) != 0;
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
        const int oy0 = ty * C3_TH, ox0 = tx * C3_TW;
        // x halo tile -> LDS (region A is free: the previous tile's 3x3 pass, its last reader, ended before that tile's cv3 pass)
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * NT;
            const int n = i >> 3, c = i & 7;
            if (n < C3_NH) *(uint4*)(xab + (c >> 2) * (C3_NP * 64) + c3_addr(n, c & 3)) = pre[k];
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);                // in flight during the four passes

        // ---- cv1 | cv2 on the halo region: y1 -> yb (all of it), y2 -> yc (interior pixels only) ------------------------------------
        for (int pt = wave; pt < C3_NT && !(abl & 16); pt += C3_NW) {
            const int n = pt * 16 + col;
            ChunkC x0f, x1f;
            x0f.u = *(const uint4*)(xab + c3_addr(n, kq));
            x1f.u = *(const uint4*)(xab + C3_NP * 64 + c3_addr(n, kq));
            f32x4c acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                ChunkC w0, w1;
                w0.u = w12s[(0 * 4 + ct) * 64 + lane]; w1.u = w12s[(1 * 4 + ct) * 64 + lane];
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.h, x0f.h, (f32x4c){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.h, x1f.h, acc[ct], 0, 0, 0);
            }
            const int ry = (n * 3641) >> 16, rx = n - ry * C3_RW;
            const bool interior = n < C3_NH && ry >= 1 && ry <= C3_TH && rx >= 1 && rx <= C3_TW;
            const int q = (ry - 1) * C3_TW + (rx - 1);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const uint2 v1 = act4(acc[ct], b12[ct]);
                if (n < C3_NH) *(uint2*)(ybb + c3_addr(n, ct * 2 + (kq >> 1)) + (kq & 1) * 8) = v1;
                const uint2 v2 = act4(acc[2 + ct], b12[2 + ct]);
                if (interior) *(uint2*)(ycb + c3_addr(q, ct * 2 + (kq >> 1)) + (kq & 1) * 8) = v2;
            }
        }
        __syncthreads();
        // ---- m.cv1 on the halo region: b1 -> plane 0 of region A; pixels outside the image hold 0 (the 3x3 pads b1 with zeros) --------
        for (int pt = wave; pt < C3_NT && !(abl & 32); pt += C3_NW) {
            const int n = pt * 16 + col;
            ChunkC yf;
            yf.u = *(const uint4*)(ybb + c3_addr(n, kq));
            const int ry = (n * 3641) >> 16, rx = n - ry * C3_RW;
            const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
            const bool inimg = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                ChunkC w;
                w.u = wm1s[ct * 64 + lane];
                const f32x4c acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, yf.h, (f32x4c){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                uint2 v = act4(acc, bm1[ct]);
                if (!inimg) v = make_uint2(0u, 0u);
                if (n < C3_NH) *(uint2*)(xab + c3_addr(n, ct * 2 + (kq >> 1)) + (kq & 1) * 8) = v;
            }
        }
        __syncthreads();
        // ---- m.cv2 (3x3) on the interior + shortcut: wave w owns rows 2w, 2w + 1; m overwrites y1 in place -----------------------------
        if (!(abl & 8)) {
            f32x4c acc[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[ct][q] = (f32x4c){0.f, 0.f, 0.f, 0.f};
            // nine taps, software-pipelined by hand: the four fragment reads of tap s + 1 are issued before the four MFMAs of tap s
            ChunkC bfr[2][2], wfr[2][2];
            auto load_tap = [&](int tp, ChunkC (&bf)[2], ChunkC (&wf)[2]) {
                const int tyy = tp / 3, txx = tp - tyy * 3;
#pragma unroll
                for (int q = 0; q < 2; ++q) bf[q].u = *(const uint4*)(xab + c3_addr((wave * 2 + q + tyy) * C3_RW + col + txx, kq));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) wf[ct].u = wm2s[(tp * 2 + ct) * 64 + lane];
            };
            load_tap(0, bfr[0], wfr[0]);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                if (tp + 1 < 9) load_tap(tp + 1, bfr[(tp + 1) & 1], wfr[(tp + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[tp & 1][ct].h, bfr[tp & 1][q].h, acc[ct][q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int nc = (wave * 2 + q + 1) * C3_RW + col + 1;              // this pixel in the halo region
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    uint2* slot = (uint2*)(ybb + c3_addr(nc, ct * 2 + (kq >> 1)) + (kq & 1) * 8);
                    const uint2 r = *slot;                                        // y1, the shortcut (added AFTER the activation)
                    const f32x2c lo = c3_silu2((f32x2c){acc[ct][q][0], acc[ct][q][1]} + (f32x2c){bm2[ct].x, bm2[ct].y}) +
                                      (f32x2c){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
                    const f32x2c hi = c3_silu2((f32x2c){acc[ct][q][2], acc[ct][q][3]} + (f32x2c){bm2[ct].z, bm2[ct].w}) +
                                      (f32x2c){__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
                    const bf16x2c p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                    *slot = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
                }
            }
        }
        __syncthreads();
        // ---- cv3 on [m | y2]: rows 2w, 2w + 1, all 64 channels -> HBM ---------------------------------------------------------------------
        if (!(abl & 64)) {
            f32x4c acc[4][2];
            ChunkC mf[2], yf[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                mf[q].u = *(const uint4*)(ybb + c3_addr((wave * 2 + q + 1) * C3_RW + col + 1, kq));
                yf[q].u = *(const uint4*)(ycb + c3_addr((wave * 2 + q) * C3_TW + col, kq));
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                ChunkC w0, w1;
                w0.u = w3s[(0 * 4 + ct) * 64 + lane]; w1.u = w3s[(1 * 4 + ct) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.h, mf[q].h, (f32x4c){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.h, yf[q].h, acc[ct][q], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const uint2 P0 = act4(acc[ct][0], b3[ct]), P1 = act4(acc[ct][1], b3[ct]);
                // lane pairs (lane, lane ^ 16) swap halves across the two rows: 16 bytes = 8 channels of ONE pixel (conv_epilogue_bf16)
                const u32x2c sx = __builtin_amdgcn_permlane16_swap(P0.x, P1.x, false, false);
                const u32x2c sy = __builtin_amdgcn_permlane16_swap(P0.y, P1.y, false, false);
                const uint4 o4 = make_uint4(sx.x, sy.x, sx.y, sy.y);
                const int oy = oy0 + wave * 2 + (odd ? 1 : 0), ox = ox0 + col;
                if (oy < a.H && ox < a.W && !(abl & 4))
                    *(uint4*)(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.out_cs + a.out_co + ct * 16 + (kq & ~1) * 4) = o4;
            }
        }
    }
}

// the four launches of engine.hip::yolo_c3 for n = 1 with a shortcut: cv1 | cv2 (one launch, two destinations), m.cv1, m.cv2, cv3
bool c3_fused_applicable(const ConvP& p12, const ConvP& pm1, const ConvP& pm2, const ConvP& p3) {
    auto pw = [](const ConvP& p, int cin, int cout) {
        return p.prec == PREC_BF16 && p.kh == 1 && p.kw == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.Cin == cin && p.Cout == cout &&
               p.act == ACT_SILU && p.res_mode == RES_NONE && !p.out_f32;
    };
    if (!pw(p12, 64, 64) || p12.split != 32 || !pw(pm1, 32, 32) || pm1.split != 0 || !pw(p3, 64, 64) || p3.split != 0) return false;
    if (!(pm2.prec == PREC_BF16 && pm2.kh == 3 && pm2.kw == 3 && pm2.sh == 1 && pm2.sw == 1 && pm2.ph == 1 && pm2.pw == 1 && pm2.Cin == 32 && pm2.Cout == 32 &&
          pm2.act == ACT_SILU && pm2.res_mode == RES_AFTER_ACT && !pm2.out_f32 && pm2.split == 0))
        return false;
    // data flow: y1 = p12.out feeds m.cv1 and is m.cv2's shortcut; b1 = pm1.out feeds m.cv2; [m | y2] = p3's input
    const bool chain = pm1.in == p12.out && pm1.in_co == p12.out_co && pm2.in == pm1.out && pm2.in_co == pm1.out_co && pm2.res == p12.out && pm2.res_co == p12.out_co &&
                       pm2.out == p3.in && pm2.out_co == p3.in_co && p12.out2 == p3.in && p12.out2_co == p3.in_co + 32 && pm2.out_cs == p3.in_cs && p12.out2_cs == p3.in_cs;
    const bool shape = pm1.H == p12.H && pm2.H == p12.H && p3.H == p12.H && pm1.W == p12.W && pm2.W == p12.W && p3.W == p12.W && pm1.B == p12.B && p3.B == p12.B;
    return chain && shape && p12.in_cs % 8 == 0 && p12.in_co % 8 == 0 && p3.out_cs % 8 == 0 && p3.out_co % 8 == 0 && p12.Kp >= 64 && pm1.Kp >= 32 && pm2.Kp >= 288 && p3.Kp >= 64;
}

int launch_c3_fused(const ConvP& p12, const ConvP& pm1, const ConvP& pm2, const ConvP& p3, hipStream_t s) {
    if (!c3_fused_applicable(p12, pm1, pm2, p3)) return VC_ERR_ARG;
    C3Args a{};
    a.w12 = (const uint4*)p12.w; a.wm1 = (const uint4*)pm1.w; a.wm2 = (const uint4*)pm2.w; a.w3 = (const uint4*)p3.w;
    a.b12 = p12.bias; a.bm1 = pm1.bias; a.bm2 = pm2.bias; a.b3 = p3.bias;
    a.kw12 = p12.Kp / 8; a.kwm1 = pm1.Kp / 8; a.kwm2 = pm2.Kp / 8; a.kw3 = p3.Kp / 8;
    a.x = (const uint16_t*)p12.in; a.in_cs = p12.in_cs; a.in_co = p12.in_co;
    a.y = (uint16_t*)p3.out; a.out_cs = p3.out_cs; a.out_co = p3.out_co;
    a.B = p12.B; a.H = p12.H; a.W = p12.W;
    a.tiles_x = (a.W + C3_TW - 1) / C3_TW; a.tiles_y = (a.H + C3_TH - 1) / C3_TH;
    const int ntiles = a.B * a.tiles_x * a.tiles_y;
    static const int slots_hw = [] {
        int per_cu = 2, dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, c3_fused_kernel<false>, C3_NW * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        return per_cu * cus;
    }();
    static const int slots_reserve = getenv("VC_CONV_RESERVE") ? atoi(getenv("VC_CONV_RESERVE")) : 64;
    const int grid = std::min(ntiles, std::max(256, slots_hw - slots_reserve / 2));   // persistent; two workgroups per CU: half the usual number of slots stays free
    a.abl = p12.ablate;                                                              // diagnostics only (engine option "c3_ablate", tools/ff_ablate.py)
    if (a.abl) launch_timed(p12, c3_fused_kernel<true>, dim3(grid), dim3(C3_NW * 64), 0, s, a);
    else launch_timed(p12, c3_fused_kernel<false>, dim3(grid), dim3(C3_NW * 64), 0, s, a);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
