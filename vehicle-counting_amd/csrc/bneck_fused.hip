// A 64-channel YOLOv5 Bottleneck in one kernel (bf16): b1 = SiLU(cv1 y) (1x1, 64 -> 64), m = [y +] SiLU(cv2 b1) (3x3 / pad 1, 64 -> 64)
// -- models/common.py::Bottleneck of ultralytics/yolov5 v6.0 (e = 1.0 inside C3), reached from /root/reference/networks/yolo.py:70.
// YOLOv5s runs it three times at 80 x 80 (both bottlenecks of the second backbone C3, with the shortcut; the one of the P3 head C3,
// without).  Unfused, b1 (105 MB per 128 frames) is written and read back and the 1x1 is a launch of its own at the HBM roof.
//
// Same construction as c3_fused.hip: a workgroup owns an 8 x 16 tile of output pixels, computes b1 on the tile's 10 x 18 halo region
// into LDS (pixels outside the image hold 0: the 3x3 pads b1 with zeros) and convolves it from there.  The weights of both layers
// (8 + 72 KB in MFMA fragment order) stay in LDS for the whole launch, which leaves room for ONE workgroup per CU: eight waves, two
// per SIMD.  LDS: y on the halo region (two 32-channel planes, 24 KB), b1 likewise (24 KB), weights 80 KB = 128 KB.
// MFMA operand order and k order (tap-major, then the two 32-channel halves of a tap) equal conv_igemm_kernel's; the epilogues are
// conv_epilogue_bf16's expressions.
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
#define BN_NP (BN_NT * 16)             // 192 pixel slots
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
};

__global__ __launch_bounds__(BN_NW * 64) void bneck_fused_kernel(const BnArgs a) {
    const bool RES = a.res != 0;                            // launch-uniform: the block has a shortcut
    __shared__ uint4 ys[2 * BN_NP * 4];                    // 24 KB: the block's input y, [32-channel plane][pixel slot][4 chunks]
    __shared__ uint4 bs[2 * BN_NP * 4];                    // 24 KB: b1 likewise
    __shared__ uint4 w1s[2 * 4 * 64];                      // 8 KB: cv1, [k step][channel tile][lane]
    __shared__ uint4 w2s[18 * 4 * 64];                     // 72 KB: cv2, [k step = 2 tap + half][channel tile][lane]
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
    // cv1: a wave's unit = (pixel tile, channel half); cv2: rows 2 (w & 3), 2 (w & 3) + 1 and channel half w >> 2
    const int rp = wave & 3, ch0 = (wave >> 2) * 2;
    float4 bv1[4], bv2[2];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv1[ct] = *(const float4*)(a.b1 + ct * 16 + kq * 4);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv2[ct] = *(const float4*)(a.b2 + (ch0 + ct) * 16 + kq * 4);

    const int ntiles = a.B * a.tiles_y * a.tiles_x;
    constexpr int NPRE = (BN_NH * 8 + NT - 1) / NT;        // 3 chunks of the y halo tile per thread
    uint4 pre[NPRE];
    auto fetch = [&](int t) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
        const int y0 = ty * BN_TH - 1, x0 = tx * BN_TW - 1;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * NT;
            const int n = i >> 3, c = i & 7;
            const int ry = (n * 3641) >> 16, rx = n - ry * BN_RW;             // n / 18 for n < 192
            const int gy = y0 + ry, gx = x0 + rx;
            pre[k] = make_uint4(0u, 0u, 0u, 0u);
            if (n < BN_NH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                pre[k] = *(const uint4*)(a.x + (((size_t)b * a.H + gy) * a.W + gx) * a.in_cs + a.in_co + c * 8);
        }
    };
    char* ysb = (char*)ys;
    char* bsb = (char*)bs;
    const bool odd = (kq & 1) != 0;
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, b = t / (a.tiles_x * a.tiles_y);
        const int oy0 = ty * BN_TH, ox0 = tx * BN_TW;
        __syncthreads();                                    // the previous tile's 3x3 pass (reads bs, ys) is done
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * NT;
            const int n = i >> 3, c = i & 7;
            if (n < BN_NH) *(uint4*)(ysb + (c >> 2) * BN_PLANE + bn_addr(n, c & 3)) = pre[k];
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);                // in flight during both passes

        // ---- cv1 (1x1) on the halo region: 12 pixel tiles x 2 channel halves = 24 units over 8 waves ---------------------------------
        for (int u = wave; u < 2 * BN_NT; u += BN_NW) {
            const int pt = u >> 1, cth = (u & 1) * 2;
            const int n = pt * 16 + col;
            ChunkB y0f, y1f;
            y0f.u = *(const uint4*)(ysb + bn_addr(n, kq));
            y1f.u = *(const uint4*)(ysb + BN_PLANE + bn_addr(n, kq));
            const int ry = (n * 3641) >> 16, rx = n - ry * BN_RW;
            const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
            const bool inimg = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int ct = cth + c2;
                ChunkB w0, w1;
                w0.u = w1s[(0 * 4 + ct) * 64 + lane]; w1.u = w1s[(1 * 4 + ct) * 64 + lane];
                f32x4b acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0.h, y0f.h, (f32x4b){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1.h, y1f.h, acc, 0, 0, 0);
                const float4 bb = ct == 0 ? bv1[0] : ct == 1 ? bv1[1] : ct == 2 ? bv1[2] : bv1[3];
                const f32x2b lo = bn_silu2((f32x2b){acc[0], acc[1]} + (f32x2b){bb.x, bb.y});
                const f32x2b hi = bn_silu2((f32x2b){acc[2], acc[3]} + (f32x2b){bb.z, bb.w});
                const bf16x2b p0 = {(__bf16)lo.x, (__bf16)lo.y}, p1 = {(__bf16)hi.x, (__bf16)hi.y};
                uint2 v = make_uint2(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
                if (!inimg) v = make_uint2(0u, 0u);
                // channel tile ct = channels 16 ct .. 16 ct + 15: plane ct >> 1, chunks 2 (ct & 1) and 2 (ct & 1) + 1 of the pixel
                if (n < BN_NH) *(uint2*)(bsb + (ct >> 1) * BN_PLANE + bn_addr(n, (ct & 1) * 2 + (kq >> 1)) + (kq & 1) * 8) = v;
            }
        }
        __syncthreads();
        // ---- cv2 (3x3) on the interior [+ shortcut] -> HBM ---------------------------------------------------------------------------------
        {
            f32x4b acc[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[ct][q] = (f32x4b){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int tyy = tp / 3, txx = tp % 3;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    ChunkB bf[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) bf[q].u = *(const uint4*)(bsb + hf * BN_PLANE + bn_addr((rp * 2 + q + tyy) * BN_RW + col + txx, kq));
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        ChunkB w;
                        w.u = w2s[((tp * 2 + hf) * 4 + ch0 + ct) * 64 + lane];
#pragma unroll
                        for (int q = 0; q < 2; ++q) acc[ct][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.h, bf[q].h, acc[ct][q], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                uint2 P[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x2b xl = (f32x2b){acc[ct][q][0], acc[ct][q][1]} + (f32x2b){bv2[ct].x, bv2[ct].y};
                    const f32x2b xh = (f32x2b){acc[ct][q][2], acc[ct][q][3]} + (f32x2b){bv2[ct].z, bv2[ct].w};
                    const f32x2b rl = bn_sigmoid2(xl), rh = bn_sigmoid2(xh);
                    f32x2b lo, hi;
                    if (RES) {                              // the shortcut: this pixel of y, added AFTER the activation -- as ONE fused multiply-add
                        const int nc = (rp * 2 + q + 1) * BN_RW + col + 1;       // (x * sigmoid(x) + y rounded once), which is what conv_epilogue_bf16 compiles to
                        const int c = ch0 + ct;
                        const uint2 r = *(const uint2*)(ysb + (c >> 1) * BN_PLANE + bn_addr(nc, (c & 1) * 2 + (kq >> 1)) + (kq & 1) * 8);
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
                const int oy = oy0 + rp * 2 + (odd ? 1 : 0), ox = ox0 + col;
                if (oy < a.H && ox < a.W)
                    *(uint4*)(a.y + (((size_t)b * a.H + oy) * a.W + ox) * a.out_cs + a.out_co + (ch0 + ct) * 16 + (kq & ~1) * 4) = o4;
            }
        }
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
    const int grid = std::min(ntiles, cus);                // persistent, one workgroup per CU (128 KB of LDS)
    a.res = pm2.res_mode == RES_AFTER_ACT ? 1 : 0;
    launch_timed(pm1, bneck_fused_kernel, dim3(grid), dim3(BN_NW * 64), 0, s, a);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
