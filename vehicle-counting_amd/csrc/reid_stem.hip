// DeepSORT ReID stem (deep/model.py:51-58): conv3x3(3 -> 64, bias, BN folded) + ReLU + MaxPool2d(3, 2, padding=1) on 50x50 crops,
// fused for the bf16 path.
//
// As two launches the 50x50x64 conv output (320 KB per crop) is written to HBM and read straight back by the pool: at ~1000
// crops per batch that is 0.7 GB of traffic for a 95 MB result.  Here a work item is (crop, band of two pooled rows): the
// workgroup stages the 7 input rows the band needs (6 KB), runs the same MFMA sequence as conv_igemm_kernel on the 5 conv rows
// (bias + ReLU + bf16 rounding as there), keeps them in LDS (32 KB) and writes only the pooled rows.  max() commutes with the
// monotonic bf16 rounding, so the result is bit-identical to conv -> store -> pool.
#include "vc_common.h"

namespace vc {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
union ChunkR { uint4 u; bf16x8r h; };

#define RS_S 50               // crop side
#define RS_P 25               // pooled side
#define RS_PW 52              // patch width in pixels (x = -1 .. 50)
#define RS_ROWS 5             // conv rows per band
#define RS_BANDS 13           // bands of two pooled rows per crop

// max of two bf16 pairs that are >= +0 (after ReLU): non-negative floats order like their bit patterns, so it is one packed
// unsigned 16-bit max (v_pk_max_u16)
__device__ __forceinline__ uint32_t max_bf16x2_nonneg(uint32_t a, uint32_t b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

__global__ __launch_bounds__(256) void reid_stem_pool_kernel(const uint4* __restrict__ x /* [k][50][50] chunks of 8 bf16 */, const uint4* __restrict__ w,
                                                             const float* __restrict__ bias, uint32_t* __restrict__ y /* [k][25][25][32] bf16x2 */, int k,
                                                             int Kw8) {
    __shared__ uint4 patch[7 * RS_PW];
    // conv rows of the band, [row][col][32 channel pairs].  A pixel is 32 words = one pass over the banks, and a ds_write_b64
    // group is 16 consecutive pixels at one channel offset (16-way conflict as is): the index of the 16-byte chunk inside the
    // pixel is XORed with (pixel & 7), which leaves 2 lanes per bank pair and keeps the chunks the pool reads whole.
    __shared__ __attribute__((aligned(16))) uint32_t cbuf[RS_ROWS * RS_S * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    // weights: 3 k-steps x 4 channel tiles, lane (channel = ct*16 + col, chunk = 4*s + kq); chunks 9..15 of the packed rows are zero
    ChunkR wf[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) wf[s][ct].u = w[(size_t)(ct * 16 + col) * Kw8 + 4 * s + kq];
    float4 bv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) bv[ct] = *(const float4*)(bias + ct * 16 + kq * 4);
    int koff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int q = min(4 * s + kq, 8);            // padded chunks multiply zero weights: any finite operand will do
        koff[s] = (q / 3) * RS_PW + (q % 3);
    }
    const int nitems = k * RS_BANDS;
    uint4 pre[2];                                    // the next item's patch travels through registers during this item's work
    auto fetch = [&](int item) {
        const int crop = item / RS_BANDS, band = item - crop * RS_BANDS;
        const int cy0 = 4 * band - 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = threadIdx.x + q * 256;
            const int pr = i / RS_PW, pc = i - pr * RS_PW;
            const int iy = cy0 - 1 + pr, ix = pc - 1;
            pre[q] = make_uint4(0u, 0u, 0u, 0u);
            if (i < 7 * RS_PW && iy >= 0 && iy < RS_S && ix >= 0 && ix < RS_S) pre[q] = x[((size_t)crop * RS_S + iy) * RS_S + ix];
        }
    };
    if ((int)blockIdx.x < nitems) fetch(blockIdx.x);
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int crop = item / RS_BANDS, band = item - crop * RS_BANDS;
        const int cy0 = 4 * band - 1;                // first conv row of the band; input rows cy0-1 .. cy0+5
        __syncthreads();                             // the previous item's reads of patch / cbuf are done
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = threadIdx.x + q * 256;
            if (i < 7 * RS_PW) patch[i] = pre[q];
        }
        __syncthreads();
        if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
        // conv: 250 pixels = 16 tiles of 16, four per wave
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int p = min((wave * 4 + tt) * 16 + col, RS_ROWS * RS_S - 1);
            const int cr = p / RS_S, cx = p - cr * RS_S;
            f32x4r acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4r){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                ChunkR xf;
                xf.u = patch[cr * RS_PW + cx + koff[s]];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][ct].h, xf.h, acc[ct], 0, 0, 0);
            }
            if ((wave * 4 + tt) * 16 + col < RS_ROWS * RS_S) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    float v0 = acc[ct][0] + bv[ct].x, v1 = acc[ct][1] + bv[ct].y, v2 = acc[ct][2] + bv[ct].z, v3 = acc[ct][3] + bv[ct].w;
                    v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;   // ReLU, never -0
                    typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));
                    const bf16x2r p0 = {(__bf16)v0, (__bf16)v1}, p1 = {(__bf16)v2, (__bf16)v3};
                    uint2 pk;
                    pk.x = __builtin_bit_cast(uint32_t, p0); pk.y = __builtin_bit_cast(uint32_t, p1);
                    *(uint2*)&cbuf[p * 32 + ((ct * 8 + kq * 2) ^ ((p & 7) << 2))] = pk;
                }
            }
        }
        __syncthreads();
        // pool: 2 pooled rows x 25 columns x 8 chunks of 8 channels; a thread owns one 16-byte chunk of one pooled pixel.  Taps
        // outside the image are clamped onto the window's own border tap (max is idempotent), so the loop has no branches.
        for (int u = threadIdx.x; u < 2 * RS_P * 8; u += 256) {
            const int prow = u / (RS_P * 8), rem = u - prow * (RS_P * 8);
            const int px = rem >> 3, c4 = rem & 7;
            const int py = 2 * band + prow;
            if (py >= RS_P) continue;
            uint4 m = make_uint4(0u, 0u, 0u, 0u);    // all values are >= +0
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int cr = max(2 * py - 1 + dy, 0) - cy0;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int cpix = cr * RS_S + max(2 * px + dx, 0);
                    const uint4 v = *(const uint4*)&cbuf[cpix * 32 + ((c4 ^ (cpix & 7)) << 2)];
                    m.x = max_bf16x2_nonneg(m.x, v.x); m.y = max_bf16x2_nonneg(m.y, v.y);
                    m.z = max_bf16x2_nonneg(m.z, v.z); m.w = max_bf16x2_nonneg(m.w, v.w);
                }
            }
            ((uint4*)y)[(((size_t)crop * RS_P + py) * RS_P + px) * 8 + c4] = m;
        }
    }
}

// conv = the stem as launch_conv sees it; the pooled output goes to a dense [k][25][25][64] bf16 buffer
bool reid_stem_applicable(const ConvP& p, int out_cs, int out_co) {
    return p.prec == PREC_BF16 && p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.ph == 1 && p.pw == 1 && p.Cin == 8 && p.in_cs == 8 &&
           p.in_co == 0 && p.H == RS_S && p.W == RS_S && p.Cout == 64 && p.act == ACT_RELU && p.res_mode == RES_NONE && !p.out_f32 && p.split == 0 &&
           p.Kp >= 96 && out_cs == 64 && out_co == 0;
}

int launch_reid_stem_pool(const ConvP& p, void* pooled, hipStream_t s) {
    const int k = p.B;
    if (k <= 0) return VC_OK;
    const int grid = std::min(k * RS_BANDS, 256 * 3 - 64);      // 3 workgroups per CU can be resident; 64 slots stay free for the tracker stream
    launch_timed(p, reid_stem_pool_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)p.in, (const uint4*)p.w, p.bias, (uint32_t*)pooled, k, p.Kp / 8);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
