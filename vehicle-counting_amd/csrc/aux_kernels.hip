// HBM-bound kernels around the convolutions: letterbox, SPPF pooling, nearest upsample, ReID crop/resize,
// max/avg pooling and the L2 norm.  All are streaming kernels: 16-byte vector accesses along the NHWC
// channel axis, one pass over the data, grid-stride over (pixel, channel-vector) items.
//
// Reference rows: A5 (AutoShape letterbox, ultralytics/yolov5 v6.0 utils/augmentations.py
// restated in oracle/imageops.py), A6 (SPPF / Upsample / Concat), B3-B4
// (/root/reference/networks/deepsort/deep_sort.py:89-95,119-129 and deep/feature_extractor.py:26-39), B5 pools
// (/root/reference/networks/deepsort/deep/model.py:58,70,93).
#include "kernels.h"

#pragma clang fp contract(off)   // restated arithmetic must round like the reference's separate mul/add

namespace vc {

static inline int grid_for(long items, int block) {
    long g = (items + block - 1) / block;
    if (g > 256L * 8) g = 256L * 8;       // grid-stride beyond 8 blocks per CU
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------ vector helpers
template <bool F32> struct Vec;
template <> struct Vec<true> {            // 4 x f32
    static constexpr int N = 4;
    float v[4];
    __device__ static Vec load(const void* p) { Vec r; const float4 t = *(const float4*)p; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r; }
    __device__ void store(void* p) const { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<false> {           // 8 x bf16, held as f32
    static constexpr int N = 8;
    float v[8];
    __device__ static Vec load(const void* p) {
        Vec r; const uint4 t = *(const uint4*)p; const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { r.v[2 * i] = bf16_to_f32((uint16_t)(w[i] & 0xffff)); r.v[2 * i + 1] = bf16_to_f32((uint16_t)(w[i] >> 16)); }
        return r;
    }
    __device__ void store(void* p) const {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
        *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// ------------------------------------------------------------------------------------------ letterbox (A5)
// cv::resize INTER_LINEAR on 8-bit data: 11-bit fixed point coefficients (lin_coef, vc_common.h; see oracle/imageops.py).
template <bool F32>
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, void* __restrict__ dst, int B, LetterboxGeom g) {
    const long total = (long)B * g.net_h * g.net_w;
    const bool resize = !(g.unpad_h == g.src_h && g.unpad_w == g.src_w);
    const double sx = 1.0 / ((double)g.unpad_w / (double)g.src_w), sy = 1.0 / ((double)g.unpad_h / (double)g.src_h);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % g.net_w);
        const int y = (int)((i / g.net_w) % g.net_h);
        const int b = (int)(i / ((long)g.net_w * g.net_h));
        int px[3] = {114, 114, 114};
        const int ux = x - g.left, uy = y - g.top;
        if (ux >= 0 && ux < g.unpad_w && uy >= 0 && uy < g.unpad_h) {
            const uint8_t* im = src + (size_t)b * g.src_h * g.src_w * 3;
            if (!resize) {
                const uint8_t* q = im + ((size_t)uy * g.src_w + ux) * 3;
                px[0] = q[0]; px[1] = q[1]; px[2] = q[2];
            } else {
                int x0, x1, a0, a1, y0, y1, b0, b1;
                lin_coef(ux, g.src_w, sx, x0, x1, a0, a1, true);
                lin_coef(uy, g.src_h, sy, y0, y1, b0, b1, false);
                const uint8_t* r0 = im + (size_t)y0 * g.src_w * 3;
                const uint8_t* r1 = im + (size_t)y1 * g.src_w * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
                    const int h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
                    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    px[c] = min(max(v, 0), 255);
                }
            }
        }
        if (g.swap_rb && ux >= 0 && ux < g.unpad_w && uy >= 0 && uy < g.unpad_h) { const int t = px[0]; px[0] = px[2]; px[2] = t; }
        const float f0 = (float)px[0] / 255.f, f1 = (float)px[1] / 255.f, f2 = (float)px[2] / 255.f;
        if (F32) {
            ((float4*)dst)[i] = make_float4(f0, f1, f2, 0.f);
        } else {
            uint2 t;
            t.x = (uint32_t)f32_to_bf16(f0) | ((uint32_t)f32_to_bf16(f1) << 16);
            t.y = (uint32_t)f32_to_bf16(f2);
            ((uint2*)dst)[i] = t;
        }
    }
}

// No-resize case (source already at network scale, e.g. 640x640 frames): pad + channel swap + /255 only.  One thread makes
// four consecutive output pixels: a 12-byte source read (three dwords when the run is inside the image and 4-byte aligned)
// and one 32-byte (bf16) / 64-byte (fp32) store.  Same values as the general kernel ((float)u8 / 255.f, RNE to bf16).
template <bool F32>
__global__ __launch_bounds__(256) void letterbox_copy_kernel(const uint8_t* __restrict__ src, void* __restrict__ dst, int B, LetterboxGeom g) {
    const int wq = g.net_w >> 2;                                          // net_w is a multiple of 32
    const long total = (long)B * g.net_h * wq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xq = (int)(i % wq);
        const int y = (int)((i / wq) % g.net_h);
        const int b = (int)(i / ((long)wq * g.net_h));
        const int uy = y - g.top;
        const int ux0 = xq * 4 - g.left;
        const uint8_t* q0 = src + (((size_t)b * g.src_h + uy) * g.src_w + ux0) * 3;
        int pv[12];
        if (uy >= 0 && uy < g.unpad_h && ux0 >= 0 && ux0 + 3 < g.unpad_w && ((uintptr_t)q0 & 3) == 0) {
            const uint32_t w0 = ((const uint32_t*)q0)[0], w1 = ((const uint32_t*)q0)[1], w2 = ((const uint32_t*)q0)[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { pv[k] = (w0 >> (8 * k)) & 255; pv[4 + k] = (w1 >> (8 * k)) & 255; pv[8 + k] = (w2 >> (8 * k)) & 255; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ux = ux0 + k;
                const bool in = ux >= 0 && ux < g.unpad_w && uy >= 0 && uy < g.unpad_h;
#pragma unroll
                for (int c = 0; c < 3; ++c) pv[3 * k + c] = in ? q0[3 * k + c] : 114;
            }
        }
        float f[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p0 = g.swap_rb ? pv[3 * k + 2] : pv[3 * k], p2 = g.swap_rb ? pv[3 * k] : pv[3 * k + 2];
            f[k][0] = div255_exact((float)p0); f[k][1] = div255_exact((float)pv[3 * k + 1]); f[k][2] = div255_exact((float)p2);
        }
        const size_t o = ((size_t)b * g.net_h + y) * g.net_w + (size_t)xq * 4;
        if (F32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ((float4*)dst)[o + k] = make_float4(f[k][0], f[k][1], f[k][2], 0.f);
        } else {
            uint4 t0, t1;                                                 // v_cvt_pk_bf16_f32: RNE like f32_to_bf16
            t0.x = pack2_bf16(f[0][0], f[0][1]); t0.y = pack2_bf16(f[0][2], 0.f);
            t0.z = pack2_bf16(f[1][0], f[1][1]); t0.w = pack2_bf16(f[1][2], 0.f);
            t1.x = pack2_bf16(f[2][0], f[2][1]); t1.y = pack2_bf16(f[2][2], 0.f);
            t1.z = pack2_bf16(f[3][0], f[3][1]); t1.w = pack2_bf16(f[3][2], 0.f);
            uint4* d = (uint4*)((uint2*)dst + o);
            d[0] = t0; d[1] = t1;
        }
    }
}

int launch_letterbox(const uint8_t* src, void* dst, int B, const LetterboxGeom& g, int prec, hipStream_t s) {
    const long total = (long)B * g.net_h * g.net_w;
    if (g.unpad_h == g.src_h && g.unpad_w == g.src_w && g.net_w % 4 == 0) {
        if (prec == PREC_F32) hipLaunchKernelGGL(letterbox_copy_kernel<true>, dim3(grid_for(total / 4, 256)), dim3(256), 0, s, src, dst, B, g);
        else hipLaunchKernelGGL(letterbox_copy_kernel<false>, dim3(grid_for(total / 4, 256)), dim3(256), 0, s, src, dst, B, g);
        VC_HIP(hipGetLastError());
        return VC_OK;
    }
    if (prec == PREC_F32) hipLaunchKernelGGL(letterbox_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst, B, g);
    else hipLaunchKernelGGL(letterbox_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst, B, g);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------ SPPF pooling (A6)
// Three chained MaxPool2d(5,1,2) == max over 5x5, 9x9, 13x13 windows of x (padding behaves as -inf).
template <bool F32>
__global__ __launch_bounds__(256) void sppf_pool_kernel(View cat, int C) {
    using V = Vec<F32>;
    constexpr int ES = F32 ? 4 : 2;
    const int cv = C / V::N;
    const long total = (long)cat.B * cat.H * cat.W * cv;
    char* base = (char*)cat.ptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V::N;
        long pix = i / cv;
        const int x = (int)(pix % cat.W);
        const int y = (int)((pix / cat.W) % cat.H);
        const int b = (int)(pix / ((long)cat.W * cat.H));
        V m1, m2, m3;
#pragma unroll
        for (int j = 0; j < V::N; ++j) m1.v[j] = m2.v[j] = m3.v[j] = -INFINITY;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= cat.H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= cat.W) continue;
                const V t = V::load(base + ((((size_t)b * cat.H + yy) * cat.W + xx) * cat.cs + cat.co + c) * ES);
                const int r = max(abs(dx), abs(dy));
#pragma unroll
                for (int j = 0; j < V::N; ++j) {
                    m3.v[j] = fmaxf(m3.v[j], t.v[j]);
                    if (r <= 4) m2.v[j] = fmaxf(m2.v[j], t.v[j]);
                    if (r <= 2) m1.v[j] = fmaxf(m1.v[j], t.v[j]);
                }
            }
        }
        char* o = base + ((((size_t)b * cat.H + y) * cat.W + x) * cat.cs + cat.co + c) * ES;
        m1.store(o + (size_t)C * ES);
        m2.store(o + (size_t)2 * C * ES);
        m3.store(o + (size_t)3 * C * ES);
    }
}

// LDS version: one workgroup per (frame, channel vector) keeps the whole H x W plane in LDS and applies the three chained
// 5x5 max-pools as separable row / column passes (6 passes over 400 positions at 20x20 instead of 169 global reads per output).
template <bool F32>
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(View cat, int C) {
    using V = Vec<F32>;
    constexpr int ES = F32 ? 4 : 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];          // two planes of [H*W][N] floats
    const int cv = C / V::N;
    const int b = blockIdx.x / cv, c = (blockIdx.x % cv) * V::N;
    const int H = cat.H, W = cat.W, n = H * W;
    float* P = sm;
    float* T = sm + (size_t)n * V::N;
    char* base = (char*)cat.ptr + (((size_t)b * n) * cat.cs + cat.co + c) * ES;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const V v = V::load(base + (size_t)i * cat.cs * ES);
#pragma unroll
        for (int j = 0; j < V::N; ++j) P[i * V::N + j] = v.v[j];
    }
    __syncthreads();
    for (int round = 1; round <= 3; ++round) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {              // row pass
            const int y = i / W, x = i - y * W;
            const int x0 = max(x - 2, 0), x1 = min(x + 2, W - 1);
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                float m = -INFINITY;
                for (int xx = x0; xx <= x1; ++xx) m = fmaxf(m, P[(y * W + xx) * V::N + j]);
                T[i * V::N + j] = m;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {              // column pass + store of this round's slice
            const int y = i / W, x = i - y * W;
            const int y0 = max(y - 2, 0), y1 = min(y + 2, H - 1);
            V o;
#pragma unroll
            for (int j = 0; j < V::N; ++j) {
                float m = -INFINITY;
                for (int yy = y0; yy <= y1; ++yy) m = fmaxf(m, T[(yy * W + x) * V::N + j]);
                o.v[j] = m;
            }
            o.store(base + ((size_t)i * cat.cs + (size_t)round * C) * ES);
#pragma unroll
            for (int j = 0; j < V::N; ++j) P[i * V::N + j] = o.v[j];      // own position only: no hazard with other threads' T reads
        }
        __syncthreads();
    }
}

// Wide LDS version (the one that runs at 20x20): one workgroup per (frame, 32 words of channels = 64 bf16 / 32 fp32 channels).
// A lane owns one 4-byte word of a position, so a position's slice is one coalesced 128-byte read/write and every LDS access
// of a wave is 64 consecutive words; the three chained 5x5 max-pools are separable row / column passes on the LDS plane.
// max() of bf16 values is exact on the two halves of a word (no rounding anywhere).
// Order-preserving keys: a float's bits compare like unsigned integers once negative values have all bits flipped and
// non-negative ones the sign bit set.  bf16 pairs are keyed per half, so the 30 max() of a position are one v_pk_max_u16
// (v_max_u32 for fp32) each instead of an unpack / two fmaxf / repack; keys are turned back into values on the way out.
// (No NaNs reach this kernel: its input is the SiLU output of a finite convolution.)
// fp8 (e4m3fn) planes: max-pooling only selects, so it runs on an order-preserving integer key of the byte (sign-magnitude ->
// biased unsigned) and stores the winner's byte; 4 channels per thread.
__device__ __forceinline__ unsigned int fp8_key4(unsigned int v) {           // per byte: b ^ (b & 0x80 ? 0xff : 0x80)
    const unsigned int neg = (v >> 7) & 0x01010101u;
    return v ^ (0x80808080u | (neg * 0x7fu));
}
__device__ __forceinline__ unsigned int fp8_unkey4(unsigned int k) {        // inverse: key >= 0x80 was a positive byte (k ^ 0x80), else negative (k ^ 0xff)
    const unsigned int pos = (k >> 7) & 0x01010101u;
    return k ^ (0xffffffffu ^ (pos * 0x7fu));
}
__device__ __forceinline__ unsigned int umax4(unsigned int a, unsigned int b) {
    unsigned int r = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const unsigned int x = (a >> (8 * i)) & 0xff, y = (b >> (8 * i)) & 0xff; r |= (x > y ? x : y) << (8 * i); }
    return r;
}
template <bool F32>
__device__ __forceinline__ uint32_t to_key(uint32_t v) {
    if (F32) return v ^ (((int32_t)v >> 31) | 0x80000000u);
    const uint32_t neg = (v >> 15) & 0x00010001u;                     // sign of each half
    return v ^ ((neg * 0x7fffu) | 0x80008000u);                        // negative: flip all 16 bits, else flip the sign bit
}
template <bool F32>
__device__ __forceinline__ uint32_t from_key(uint32_t k) {
    if (F32) return k ^ ((~((int32_t)k >> 31)) | 0x80000000u);
    const uint32_t pos = (k >> 15) & 0x00010001u;                     // key has its top bit set <=> the value was non-negative
    return k ^ (((pos ^ 0x00010001u) * 0x7fffu) | 0x80008000u);
}
template <bool F32>
__device__ __forceinline__ uint32_t kmax(uint32_t a, uint32_t b) {
    if (F32) return a > b ? a : b;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// element type of the plane: 0 = fp32, 1 = bf16 pairs, 2 = fp8 (e4m3fn) quads
template <int ET> __device__ __forceinline__ uint32_t word_key(uint32_t v) { return ET == 2 ? fp8_key4(v) : to_key<ET == 0>(v); }
template <int ET> __device__ __forceinline__ uint32_t word_unkey(uint32_t k) { return ET == 2 ? fp8_unkey4(k) : from_key<ET == 0>(k); }
template <int ET> __device__ __forceinline__ uint32_t word_max(uint32_t a, uint32_t b) {
    if (ET == 2) {
        typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u8x4, a), __builtin_bit_cast(u8x4, b)));
    }
    return kmax<ET == 0>(a, b);
}

template <int ET, int WS>          // WS 32-bit words of a pixel per workgroup (32: 128-byte slices ... 4: 16-byte slices)
__global__ __launch_bounds__(1024) void sppf_pool_wide_kernel(View cat, int C) {
    constexpr int ES = ET == 0 ? 4 : (ET == 1 ? 2 : 1);
    constexpr int CW = WS * 4 / ES;                                      // channels per workgroup
    constexpr int NPG = 1024 / WS;                                       // positions per pass (16 waves hide the LDS latency)
    extern __shared__ __attribute__((aligned(16))) uint32_t smw[];       // two planes of [H*W][WS] words
    const int groups = C / CW;
    const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * CW;
    const int H = cat.H, W = cat.W, n = H * W;
    uint32_t* P = smw;
    uint32_t* T = smw + (size_t)n * WS;
    const int w = threadIdx.x % WS, pg = threadIdx.x / WS;               // word of the slice, position group
    char* base = (char*)cat.ptr + (((size_t)b * n) * cat.cs + cat.co + c0) * ES + w * 4;
    const size_t pstride = (size_t)cat.cs * ES;
    for (int i = pg; i < n; i += NPG) P[i * WS + w] = word_key<ET>(*(const uint32_t*)(base + i * pstride));
    __syncthreads();
    const uint32_t NEG = 0u;                                             // smallest key (below -inf)
    for (int round = 1; round <= 3; ++round) {
        for (int i = pg; i < n; i += NPG) {                               // row pass
            const int y = i / W, x = i - y * W;
            uint32_t m = NEG;
            for (int xx = max(x - 2, 0); xx <= min(x + 2, W - 1); ++xx) m = word_max<ET>(m, P[(y * W + xx) * WS + w]);
            T[i * WS + w] = m;
        }
        __syncthreads();
        for (int i = pg; i < n; i += NPG) {                               // column pass + store of this round's slice
            const int y = i / W, x = i - y * W;
            uint32_t m = NEG;
            for (int yy = max(y - 2, 0); yy <= min(y + 2, H - 1); ++yy) m = word_max<ET>(m, T[(yy * W + x) * WS + w]);
            *(uint32_t*)(base + i * pstride + (size_t)round * C * ES) = word_unkey<ET>(m);
            P[i * WS + w] = m;                                           // own position only: no hazard with other threads' T reads
        }
        __syncthreads();
    }
}

// Register form (the one that runs at 20x20 and 40x40 since round 5).  pool5 o pool5 = pool9 and pool5 o pool5 o pool5 = pool13 of x, and a square
// window maximum is a column maximum of row maxima, so the three outputs are col_r(row_r(x)) for r = 2, 4, 6.  A thread owns ONE ROW of one
// 4-byte word of the channel slice in registers (W global loads in flight at once), builds the row maxima of all three radii from shared
// partial windows (pairs -> quads -> octets: 7 packed max per position for the three radii together), hands them to the thread that owns the
// COLUMN through one LDS plane (one radius at a time: 27 KB per workgroup at 20x20, every workgroup of a 128-frame batch resident at once), and
// that thread finishes the column maxima in registers and stores.  Per position: 3 LDS writes + 3 LDS reads and 5 workgroup barriers per launch,
// against 33 LDS accesses per position and 7 barriers in sppf_pool_wide_kernel: 65 -> ~25 us per 128 frames at 20x20 x 256 channels
// (26 MB in, 79 MB out: ~17 us at the chip's copy rate).  Maxima select, so the result is that of the other forms bit for bit.
template <int ET, int RAD, int MAXD>
__device__ __forceinline__ void window_max(const uint32_t (&p)[MAXD + 12], uint32_t (&out)[MAXD]) {     // p[i] = position i - 6; out[x] = max p[x+6-RAD .. x+6+RAD]
    uint32_t a[MAXD + 11], b[MAXD + 9];
#pragma unroll
    for (int i = 0; i < MAXD + 11; ++i) a[i] = word_max<ET>(p[i], p[i + 1]);
#pragma unroll
    for (int i = 0; i < MAXD + 9; ++i) b[i] = word_max<ET>(a[i], a[i + 2]);                               // p[i .. i+3]
    if (RAD == 2) {
#pragma unroll
        for (int x = 0; x < MAXD; ++x) out[x] = word_max<ET>(b[x + 4], p[x + 8]);
    } else {
        uint32_t c[MAXD + 5];
#pragma unroll
        for (int i = 0; i < MAXD + 5; ++i) c[i] = word_max<ET>(b[i], b[i + 4]);                           // p[i .. i+7]
#pragma unroll
        for (int x = 0; x < MAXD; ++x)
            out[x] = RAD == 4 ? word_max<ET>(c[x + 2], p[x + 10]) : word_max<ET>(word_max<ET>(c[x], b[x + 8]), p[x + 12]);
    }
}

template <int ET, int MAXD>
__global__ __launch_bounds__(MAXD > 20 ? 320 : 640) void sppf_pool_sep_kernel(View cat, int C, int WS, int rstride) {
    constexpr int ES = ET == 0 ? 4 : (ET == 1 ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) uint32_t smw[];       // [H][rstride]: position x, word w of row y at y * rstride + x * WS + w
    const int cw = WS * 4 / ES, groups = C / cw;
    const int b = blockIdx.x / groups, c0 = (blockIdx.x % groups) * cw;
    const int H = cat.H, W = cat.W;
    const int w = threadIdx.x % WS, q = threadIdx.x / WS;                // q: this thread's row in the row pass, its column in the column passes
    char* base = (char*)cat.ptr + (((size_t)b * H * W) * cat.cs + cat.co + c0) * ES + w * 4;
    const size_t pstride = (size_t)cat.cs * ES;
    uint32_t R[3][MAXD];
    if (q < H) {
        uint32_t p[MAXD + 12];
#pragma unroll
        for (int i = 0; i < MAXD + 12; ++i) p[i] = 0u;                   // key 0 sits below every value's key: the padding of MaxPool2d
#pragma unroll
        for (int x = 0; x < MAXD; ++x)
            if (x < W) p[x + 6] = *(const uint32_t*)(base + (size_t)(q * W + x) * pstride);
#pragma unroll
        for (int x = 0; x < MAXD; ++x) p[x + 6] = x < W ? word_key<ET>(p[x + 6]) : 0u;
        // the three radii share their partial windows
        uint32_t a[MAXD + 11], bq[MAXD + 9], c[MAXD + 5];
#pragma unroll
        for (int i = 0; i < MAXD + 11; ++i) a[i] = word_max<ET>(p[i], p[i + 1]);
#pragma unroll
        for (int i = 0; i < MAXD + 9; ++i) bq[i] = word_max<ET>(a[i], a[i + 2]);
#pragma unroll
        for (int i = 0; i < MAXD + 5; ++i) c[i] = word_max<ET>(bq[i], bq[i + 4]);
#pragma unroll
        for (int x = 0; x < MAXD; ++x) {
            R[0][x] = word_max<ET>(bq[x + 4], p[x + 8]);
            R[1][x] = word_max<ET>(c[x + 2], p[x + 10]);
            R[2][x] = word_max<ET>(word_max<ET>(c[x], bq[x + 8]), p[x + 12]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k) __syncthreads();                                          // the column passes of the radius before have read the plane
        if (q < H) {
#pragma unroll
            for (int x = 0; x < MAXD; ++x)
                if (x < W) smw[q * rstride + x * WS + w] = R[k][x];
        }
        __syncthreads();
        if (q < W) {
            uint32_t p[MAXD + 12], o[MAXD];
#pragma unroll
            for (int i = 0; i < MAXD + 12; ++i) p[i] = 0u;
#pragma unroll
            for (int y = 0; y < MAXD; ++y)
                if (y < H) p[y + 6] = smw[y * rstride + q * WS + w];
            if (k == 0) window_max<ET, 2, MAXD>(p, o); else if (k == 1) window_max<ET, 4, MAXD>(p, o); else window_max<ET, 6, MAXD>(p, o);
            char* ob = base + (size_t)(k + 1) * C * ES + (size_t)q * pstride;
#pragma unroll
            for (int y = 0; y < MAXD; ++y)
                if (y < H) *(uint32_t*)(ob + (size_t)y * W * pstride) = word_unkey<ET>(o[y]);
        }
    }
}

// fp8 fallback for planes that do not fit LDS: every output reads its 13 x 13 window (4 channels per thread)
__global__ __launch_bounds__(256) void sppf_pool_fp8_kernel(View cat, int C) {
    const int cv = C / 4;
    const long total = (long)cat.B * cat.H * cat.W * cv;
    char* base = (char*)cat.ptr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 4;
        long pix = i / cv;
        const int x = (int)(pix % cat.W);
        const int y = (int)((pix / cat.W) % cat.H);
        const int b = (int)(pix / ((long)cat.W * cat.H));
        unsigned int m1 = 0, m2 = 0, m3 = 0;                              // key 0 is below every finite value's key
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= cat.H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= cat.W) continue;
                const unsigned int k = fp8_key4(*(const unsigned int*)(base + (((size_t)b * cat.H + yy) * cat.W + xx) * cat.cs + cat.co + c));
                const int r = max(abs(dx), abs(dy));
                m3 = umax4(m3, k);
                if (r <= 4) m2 = umax4(m2, k);
                if (r <= 2) m1 = umax4(m1, k);
            }
        }
        char* o = base + (((size_t)b * cat.H + y) * cat.W + x) * cat.cs + cat.co + c;
        *(unsigned int*)(o + (size_t)C) = fp8_unkey4(m1);
        *(unsigned int*)(o + (size_t)2 * C) = fp8_unkey4(m2);
        *(unsigned int*)(o + (size_t)3 * C) = fp8_unkey4(m3);
    }
}

int launch_sppf_pool(const View& cat, int C, int prec, int form, hipStream_t s) {
    const int es = prec == PREC_F32 ? 4 : (prec == PREC_FP8 ? 1 : 2);
    VC_CHECK(C % 4 == 0 && cat.cs % 4 == 0 && cat.co % 4 == 0, VC_ERR_ARG, "sppf: channel alignment");
    // form 1 (the engine's default): the register form wherever the plane is at most 40 x 40 -- every SPPF input of 640^2 ... 1280^2 frames
    const int maxd = std::max(cat.H, cat.W);
    if (form == 1 && maxd <= 40 && (cat.cs * es) % 4 == 0 && (cat.co * es) % 4 == 0) {
        static const int ws_env = getenv("VC_SPPF_WS") ? atoi(getenv("VC_SPPF_WS")) : 0;       // A/B switch
        int ws = 0, rstride = 0;
        for (int cand : {32, 16, 8, 4}) {
            if (ws || (ws_env && cand != ws_env)) continue;
            const int rs = cat.W * cand + ((cand - (cat.W * cand) % 64) % 64 + 64) % 64;      // a wave's 64 / cand rows land on distinct banks
            if (C % (cand * 4 / es) == 0 && maxd * cand <= (maxd > 20 ? 320 : 640) && (size_t)cat.H * rs * 4 <= 56 * 1024) { ws = cand; rstride = rs; }
        }
        if (ws) {
            const int blocks = cat.B * (C / (ws * 4 / es));
            const int threads = (maxd * ws + 63) / 64 * 64;
            const size_t lds = (size_t)cat.H * rstride * 4;
#define VC_SPPF_SEP(ET) { if (maxd <= 20) hipLaunchKernelGGL((sppf_pool_sep_kernel<ET, 20>), dim3(blocks), dim3(threads), lds, s, cat, C, ws, rstride); \
                          else hipLaunchKernelGGL((sppf_pool_sep_kernel<ET, 40>), dim3(blocks), dim3(threads), lds, s, cat, C, ws, rstride); }
            if (prec == PREC_F32) VC_SPPF_SEP(0) else if (prec == PREC_FP8) VC_SPPF_SEP(2) else VC_SPPF_SEP(1)
#undef VC_SPPF_SEP
            VC_HIP(hipGetLastError());
            return VC_OK;
        }
    }
    // The wide LDS kernel with the widest slice that fits: 64-byte (or narrower) slices when two such workgroups fit a CU's LDS
    // (every workgroup of a 128-frame batch is then resident at once -- the kernel is six LDS passes and seven barriers long, i.e.
    // latency-bound), wider ones otherwise.
    const size_t plane = (size_t)2 * cat.H * cat.W * 4;                  // bytes per word of slice width
    int ws = 0;
    for (int cand : {16, 8, 4})
        if (!ws && plane * cand <= 76 * 1024 && C % (cand * 4 / es) == 0) ws = cand;
    for (int cand : {32, 16, 8, 4})
        if (!ws && plane * cand <= 150 * 1024 && C % (cand * 4 / es) == 0) ws = cand;
    if (ws && (cat.cs * es) % 4 == 0 && (cat.co * es) % 4 == 0) {
        const int cw = ws * 4 / es;
        const size_t lds_wide = plane * ws;
        const int blocks = cat.B * (C / cw);
        auto go = [&](auto kernel) {
            if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide) != hipSuccess) return (int)VC_ERR_HIP;
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(1024), lds_wide, s, cat, C);
            return (int)VC_OK;
        };
        int rc = VC_ERR_ARG;
#define VC_SPPF_GO(ET) rc = ws == 32 ? go(sppf_pool_wide_kernel<ET, 32>) : ws == 16 ? go(sppf_pool_wide_kernel<ET, 16>) : ws == 8 ? go(sppf_pool_wide_kernel<ET, 8>) : go(sppf_pool_wide_kernel<ET, 4>)
        if (prec == PREC_F32) { VC_SPPF_GO(0); } else if (prec == PREC_FP8) { VC_SPPF_GO(2); } else { VC_SPPF_GO(1); }
#undef VC_SPPF_GO
        VC_CHECK(rc == VC_OK, VC_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the SPPF pool kernel");
        VC_HIP(hipGetLastError());
        return VC_OK;
    }
    if (prec == PREC_FP8) {
        const long total = (long)cat.B * cat.H * cat.W * (C / 4);
        hipLaunchKernelGGL(sppf_pool_fp8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, cat, C);
        VC_HIP(hipGetLastError());
        return VC_OK;
    }
    const int n = prec == PREC_F32 ? 4 : 8;
    VC_CHECK(C % n == 0 && cat.cs % n == 0 && cat.co % n == 0, VC_ERR_ARG, "sppf: channel alignment");
    const size_t lds = (size_t)2 * cat.H * cat.W * n * sizeof(float);
    if (lds <= 150 * 1024) {
        const int blocks = cat.B * (C / n);
        if (prec == PREC_F32) hipLaunchKernelGGL(sppf_pool_lds_kernel<true>, dim3(blocks), dim3(256), lds, s, cat, C);
        else hipLaunchKernelGGL(sppf_pool_lds_kernel<false>, dim3(blocks), dim3(256), lds, s, cat, C);
        VC_HIP(hipGetLastError());
        return VC_OK;
    }
    const long total = (long)cat.B * cat.H * cat.W * (C / n);
    if (prec == PREC_F32) hipLaunchKernelGGL(sppf_pool_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, cat, C);
    else hipLaunchKernelGGL(sppf_pool_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, cat, C);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------ nearest 2x upsample
template <int ES>
__global__ __launch_bounds__(256) void upsample2x_kernel(View src, View dst) {
    constexpr int N = 16 / ES;
    const int cv = src.C / N;
    const long total = (long)dst.B * dst.H * dst.W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * N;
        long pix = i / cv;
        const int x = (int)(pix % dst.W);
        const int y = (int)((pix / dst.W) % dst.H);
        const int b = (int)(pix / ((long)dst.W * dst.H));
        const uint4 t = *(const uint4*)((const char*)src.ptr + ((((size_t)b * src.H + (y >> 1)) * src.W + (x >> 1)) * src.cs + src.co + c) * ES);
        *(uint4*)((char*)dst.ptr + ((((size_t)b * dst.H + y) * dst.W + x) * dst.cs + dst.co + c) * ES) = t;
    }
}

// bf16 -> fp8 (e4m3fn): 8 channels per thread
__global__ __launch_bounds__(256) void bf16_to_fp8_kernel(View src, View dst, float inv_scale) {
    const int cv = src.C / 8;
    const long total = (long)src.B * src.H * src.W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8;
        const long pix = i / cv;
        const uint4 t = *(const uint4*)((const char*)src.ptr + ((size_t)pix * src.cs + src.co + c) * 2);
        const unsigned int w[4] = {t.x, t.y, t.z, t.w};
        unsigned int o[2] = {0u, 0u};
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = __builtin_amdgcn_fmed3f(__uint_as_float(w[j] << 16) * inv_scale, -448.0f, 448.0f);
            f[2 * j + 1] = __builtin_amdgcn_fmed3f(__uint_as_float(w[j] & 0xffff0000u) * inv_scale, -448.0f, 448.0f);
        }
        o[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0u, false); o[0] = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o[0], true);
        o[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0u, false); o[1] = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], o[1], true);
        *(uint2*)((char*)dst.ptr + (size_t)pix * dst.cs + dst.co + c) = make_uint2(o[0], o[1]);
    }
}

int launch_bf16_to_fp8(const View& src, const View& dst, float inv_scale, hipStream_t s) {
    VC_CHECK(src.C % 8 == 0 && src.cs % 8 == 0 && src.co % 8 == 0 && dst.cs % 8 == 0 && dst.co % 8 == 0, VC_ERR_ARG, "bf16 -> fp8: alignment");
    const long total = (long)src.B * src.H * src.W * (src.C / 8);
    hipLaunchKernelGGL(bf16_to_fp8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst, inv_scale);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_upsample2x(const View& src, const View& dst, int prec, hipStream_t s) {
    if (prec == PREC_FP8) {
        VC_CHECK(src.C % 16 == 0 && src.cs % 16 == 0 && src.co % 16 == 0 && dst.cs % 16 == 0 && dst.co % 16 == 0, VC_ERR_ARG, "upsample fp8: alignment");
        VC_CHECK(dst.H == 2 * src.H && dst.W == 2 * src.W && dst.B == src.B, VC_ERR_ARG, "upsample: shape");
        const long total = (long)dst.B * dst.H * dst.W * (src.C / 16);
        hipLaunchKernelGGL(upsample2x_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst);
        VC_HIP(hipGetLastError());
        return VC_OK;
    }
    const int n = prec == PREC_F32 ? 4 : 8;
    VC_CHECK(src.C % n == 0 && src.cs % n == 0 && src.co % n == 0 && dst.cs % n == 0 && dst.co % n == 0, VC_ERR_ARG, "upsample: alignment");
    VC_CHECK(dst.H == 2 * src.H && dst.W == 2 * src.W && dst.B == src.B, VC_ERR_ARG, "upsample: shape");
    const long total = (long)dst.B * dst.H * dst.W * (src.C / n);
    if (prec == PREC_F32) hipLaunchKernelGGL(upsample2x_kernel<4>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst);
    else hipLaunchKernelGGL(upsample2x_kernel<2>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------ ReID crop + resize (B3/B4)
// cv2.resize(im.astype(float32)/255., (50,50)) on float data: horizontal pass then vertical pass in f32,
// every product and sum rounded separately (this file compiles with fp contract(off); the __fmul_rn / __fadd_rn wrappers of the HIP
// headers are plain operators compiled with contraction ALLOWED, and the backend fused them differently in the two kernels below),
// then torchvision Normalize; restated in oracle/imageops.py::resize_linear_f32 + oracle/reid.py::preprocess_crops.
__device__ __forceinline__ void lin_coef_f(int d, int src, double scale, int& s0, int& s1, float& c0, float& c1, bool horizontal) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        s0 = s; s1 = min(s + 1, src - 1);
    } else {
        s0 = min(max(s, 0), src - 1); s1 = min(max(s + 1, 0), src - 1);
    }
    c0 = 1.f - f; c1 = f;
}

template <bool F32>
__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t* __restrict__ frames, int H, int W, const int* __restrict__ crops5,
                                                          int k, void* __restrict__ dst, int cpad) {
    constexpr int S = VC_REID_SIZE;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const long total = (long)k * S * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S), n = (int)(i / (S * S));
        const int* cr = crops5 + n * 5;
        const int x1 = cr[1], y1 = cr[2], cw = cr[3] - cr[1], chh = cr[4] - cr[2];
        float out[3] = {0.f, 0.f, 0.f};
        if (cw > 0 && chh > 0) {
            const uint8_t* im = frames + (size_t)cr[0] * H * W * 3;
            float v[3];
            if (cw == S && chh == S) {
                const uint8_t* q = im + ((size_t)(y1 + y) * W + x1 + x) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (float)q[c] / 255.f;
            } else {
                int sx0, sx1, sy0, sy1; float a0, a1, b0, b1;
                lin_coef_f(x, cw, 1.0 / ((double)S / (double)cw), sx0, sx1, a0, a1, true);
                lin_coef_f(y, chh, 1.0 / ((double)S / (double)chh), sy0, sy1, b0, b1, false);
                const uint8_t* r0 = im + ((size_t)(y1 + sy0) * W + x1) * 3;
                const uint8_t* r1 = im + ((size_t)(y1 + sy1) * W + x1) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float p00 = (float)r0[sx0 * 3 + c] / 255.f, p01 = (float)r0[sx1 * 3 + c] / 255.f;
                    const float p10 = (float)r1[sx0 * 3 + c] / 255.f, p11 = (float)r1[sx1 * 3 + c] / 255.f;
                    const float h0 = p00 * a0 + p01 * a1;
                    const float h1 = p10 * a0 + p11 * a1;
                    v[c] = h0 * b0 + h1 * b1;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = (v[c] - mean[c]) / stdv[c];
        }
        if (F32) {
            float* o = (float*)dst + (size_t)i * cpad;
            for (int c = 0; c < cpad; ++c) o[c] = c < 3 ? out[c] : 0.f;
        } else {
            uint16_t* o = (uint16_t*)dst + (size_t)i * cpad;
            for (int c = 0; c < cpad; ++c) o[c] = c < 3 ? f32_to_bf16(out[c]) : (uint16_t)0;
        }
    }
}

// The same arithmetic with one workgroup per crop (the throughput path: bf16, 8 channels per pixel): the interpolation taps and
// weights of the 50 columns and 50 rows are computed once per crop into LDS (two fp64 divisions each) instead of once per
// output pixel, u8 / 255 is the two-instruction exact form (div255_exact, vc_common.h), a pixel leaves as one 16-byte store.
// Bit-identical to crop_resize_kernel<false> (tests/test_gpu_nets.py::test_crop_resize_per_crop_kernel).
__global__ __launch_bounds__(256) void crop_resize_wg_kernel(const uint8_t* __restrict__ frames, int H, int W, const int* __restrict__ crops5,
                                                            uint4* __restrict__ dst) {
    constexpr int S = VC_REID_SIZE;
    __shared__ int tap[2][S][2];
    __shared__ float wgt[2][S][2];
    const int n = blockIdx.x;
    const int* cr = crops5 + n * 5;
    const int x1 = cr[1], y1 = cr[2], cw = cr[3] - cr[1], chh = cr[4] - cr[2];
    uint4* out = dst + (size_t)n * S * S;
    if (!(cw > 0 && chh > 0)) {
        for (int i = threadIdx.x; i < S * S; i += blockDim.x) out[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    if (threadIdx.x < 2 * S) {
        const int axis = threadIdx.x / S, d = threadIdx.x - axis * S;
        int s0, s1; float c0, c1;
        if (axis == 0) lin_coef_f(d, cw, 1.0 / ((double)S / (double)cw), s0, s1, c0, c1, true);
        else lin_coef_f(d, chh, 1.0 / ((double)S / (double)chh), s0, s1, c0, c1, false);
        tap[axis][d][0] = s0; tap[axis][d][1] = s1; wgt[axis][d][0] = c0; wgt[axis][d][1] = c1;
    }
    __syncthreads();
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const uint8_t* im = frames + ((size_t)cr[0] * H + y1) * W * 3 + (size_t)x1 * 3;
    const bool same = cw == S && chh == S;
    for (int i = threadIdx.x; i < S * S; i += blockDim.x) {
        const int y = i / S, x = i - y * S;
        float v[3];
        if (same) {
            const uint8_t* q = im + ((size_t)y * W + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = div255_exact((float)q[c]);
        } else {
            const int sx0 = tap[0][x][0], sx1 = tap[0][x][1], sy0 = tap[1][y][0], sy1 = tap[1][y][1];
            const float a0 = wgt[0][x][0], a1 = wgt[0][x][1], b0 = wgt[1][y][0], b1 = wgt[1][y][1];
            const uint8_t* r0 = im + (size_t)sy0 * W * 3;
            const uint8_t* r1 = im + (size_t)sy1 * W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p00 = div255_exact((float)r0[sx0 * 3 + c]), p01 = div255_exact((float)r0[sx1 * 3 + c]);
                const float p10 = div255_exact((float)r1[sx0 * 3 + c]), p11 = div255_exact((float)r1[sx1 * 3 + c]);
                const float h0 = p00 * a0 + p01 * a1;
                const float h1 = p10 * a0 + p11 * a1;
                v[c] = h0 * b0 + h1 * b1;
            }
        }
        const uint32_t o0 = f32_to_bf16((v[0] - mean[0]) / stdv[0]), o1 = f32_to_bf16((v[1] - mean[1]) / stdv[1]), o2 = f32_to_bf16((v[2] - mean[2]) / stdv[2]);
        out[i] = make_uint4(o0 | (o1 << 16), o2, 0u, 0u);
    }
}

int launch_crop_resize(const uint8_t* frames, int H, int W, const int* crops5, int k, void* dst, int cpad, int prec, hipStream_t s, bool per_pixel) {
    if (k <= 0) return VC_OK;
    if (prec != PREC_F32 && cpad == 8 && !per_pixel) {
        hipLaunchKernelGGL(crop_resize_wg_kernel, dim3(k), dim3(256), 0, s, frames, H, W, crops5, (uint4*)dst);
        VC_HIP(hipGetLastError());
        return VC_OK;
    }
    const long total = (long)k * VC_REID_SIZE * VC_REID_SIZE;
    if (prec == PREC_F32) hipLaunchKernelGGL(crop_resize_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, H, W, crops5, k, dst, cpad);
    else hipLaunchKernelGGL(crop_resize_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, H, W, crops5, k, dst, cpad);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

template <bool F32>
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, int k, int C, int H, int W, void* __restrict__ dst, int cpad) {
    const long total = (long)k * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int hw = (int)(i % ((long)H * W));
        const int n = (int)(i / ((long)H * W));
        for (int c = 0; c < cpad; ++c) {
            const float v = c < C ? x[((size_t)n * C + c) * H * W + hw] : 0.f;
            if (F32) ((float*)dst)[(size_t)i * cpad + c] = v;
            else ((uint16_t*)dst)[(size_t)i * cpad + c] = f32_to_bf16(v);
        }
    }
}

int launch_nchw_to_nhwc_pad(const float* x, int k, int C, int H, int W, void* dst, int cpad, int prec, hipStream_t s) {
    if (k <= 0) return VC_OK;
    const long total = (long)k * H * W;
    if (prec == PREC_F32) hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, x, k, C, H, W, dst, cpad);
    else hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, x, k, C, H, W, dst, cpad);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------ MaxPool2d(3, 2, padding=1)
template <bool F32>
__global__ __launch_bounds__(256) void maxpool3s2_kernel(View src, View dst) {
    using V = Vec<F32>;
    constexpr int ES = F32 ? 4 : 2;
    const int cv = src.C / V::N;
    const long total = (long)dst.B * dst.H * dst.W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V::N;
        long pix = i / cv;
        const int x = (int)(pix % dst.W);
        const int y = (int)((pix / dst.W) % dst.H);
        const int b = (int)(pix / ((long)dst.W * dst.H));
        V m;
#pragma unroll
        for (int j = 0; j < V::N; ++j) m.v[j] = -INFINITY;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * y + dy;
            if (yy < 0 || yy >= src.H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = 2 * x + dx;
                if (xx < 0 || xx >= src.W) continue;
                const V t = V::load((const char*)src.ptr + ((((size_t)b * src.H + yy) * src.W + xx) * src.cs + src.co + c) * ES);
#pragma unroll
                for (int j = 0; j < V::N; ++j) m.v[j] = fmaxf(m.v[j], t.v[j]);
            }
        }
        m.store((char*)dst.ptr + ((((size_t)b * dst.H + y) * dst.W + x) * dst.cs + dst.co + c) * ES);
    }
}

int launch_maxpool3s2(const View& src, const View& dst, int prec, hipStream_t s) {
    const int n = prec == PREC_F32 ? 4 : 8;
    VC_CHECK(src.C % n == 0 && src.cs % n == 0 && dst.cs % n == 0, VC_ERR_ARG, "maxpool: alignment");
    const long total = (long)dst.B * dst.H * dst.W * (src.C / n);
    if (total <= 0) return VC_OK;
    if (prec == PREC_F32) hipLaunchKernelGGL(maxpool3s2_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst);
    else hipLaunchKernelGGL(maxpool3s2_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, dst);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ------------------------------------------------------------------------------------------ AvgPool(4,4) + L2 normalise
// One workgroup per crop: 256 threads x 2 channels; block-wide sum of squares through wave shuffles + LDS.
template <bool F32>
__global__ __launch_bounds__(256) void avgpool_l2norm_kernel(View src, float* __restrict__ out) {
    constexpr int ES = F32 ? 4 : 2;
    __shared__ float part[4];
    const int n = blockIdx.x, t = threadIdx.x;
    float a[2] = {0.f, 0.f};
    const int npix = src.H * src.W;    // 16
    for (int q = 0; q < npix; ++q) {
        const char* ptr = (const char*)src.ptr + (((size_t)n * npix + q) * src.cs + src.co + 2 * t) * ES;
        if (F32) { const float2 v = *(const float2*)ptr; a[0] += v.x; a[1] += v.y; }
        else { const uint32_t v = *(const uint32_t*)ptr; a[0] += bf16_to_f32((uint16_t)(v & 0xffff)); a[1] += bf16_to_f32((uint16_t)(v >> 16)); }
    }
    a[0] /= (float)npix; a[1] /= (float)npix;
    float ss = a[0] * a[0] + a[1] * a[1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((t & 63) == 0) part[t >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(part[0] + part[1] + part[2] + part[3]);
    out[(size_t)n * VC_FEAT_DIM + 2 * t] = a[0] / nrm;
    out[(size_t)n * VC_FEAT_DIM + 2 * t + 1] = a[1] / nrm;
}

int launch_avgpool_l2norm(const View& src, float* out, int prec, hipStream_t s) {
    VC_CHECK(src.C == VC_FEAT_DIM, VC_ERR_ARG, "avgpool: expects 512 channels");
    if (src.B <= 0) return VC_OK;
    if (prec == PREC_F32) hipLaunchKernelGGL(avgpool_l2norm_kernel<true>, dim3(src.B), dim3(256), 0, s, src, out);
    else hipLaunchKernelGGL(avgpool_l2norm_kernel<false>, dim3(src.B), dim3(256), 0, s, src, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
