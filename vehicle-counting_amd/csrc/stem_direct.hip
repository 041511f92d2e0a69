// YOLOv5 stem (6x6 / stride 2 / pad 2 over 3 input channels, models/yolo.py v6.0 layer 0) as a direct
// convolution for the bf16 path.
//
// The implicit-GEMM kernel treats the stem like any other layer: every output pixel's 6x3 taps (pairs of 4-channel pixels,
// DESIGN.md section 3) are gathered global -> LDS as 18 separate 16-byte pieces, 2.1 GB through the vector cache per 64
// frames for a layer whose algorithmic traffic is 0.63 GB.  Here a workgroup stages the input patch of an 8 x 32 output tile
// ONCE (20 rows x 34 pixel pairs, 11 KB), keeps the whole weight matrix in registers across the tiles it walks, and feeds the
// same MFMA sequence from the patch: the layer becomes what it is, an HBM stream (read 8 B, write 64 B per input pixel).
// The MFMA operand order and the k order are those of conv_igemm_kernel, so the results are bit-identical to it.
#include "kernels.h"

namespace vc {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
union ChunkS { uint4 u; bf16x8s h; };

#define STEM_TH 8
#define STEM_TW 32
#define STEM_PR 20            // patch rows  = 2*TH + 4
#define STEM_PC 34            // patch pairs = TW + 2
#define STEM_PP 36            // LDS row pitch in 16-byte chunks

// U8: the patch comes straight from the u8 frames (letterbox_copy_kernel's arithmetic applied per fetched pixel pair: pad value
// 114, optional R/B swap, exact /255, RNE to bf16), so the letterboxed tensor is never written or read.
template <int CT, bool U8>   // Cout = CT * 16
__global__ __launch_bounds__(256) void stem_direct_kernel(const uint4* __restrict__ x, const uint4* __restrict__ w, const float* __restrict__ bias,
                                                          uint16_t* __restrict__ y, int B, int H, int Wp, int Ho, int Wo, int Kw8 /* weight row stride in chunks */,
                                                          int out_cs, int out_co, int tiles_x, int tiles_y, const uint8_t* __restrict__ src8, LetterboxGeom g,
                                                          uint8_t* __restrict__ y8, int q_cs, int q_co, float q_inv_scale) {
    __shared__ uint4 patch[STEM_PR * STEM_PP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    // weights: 5 k-steps x CT channel tiles, lane (channel = ct*16 + col, chunk = 4*s + kq); chunks 18..23 of the packed rows are zero
    ChunkS wf[5][CT];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wf[s][ct].u = w[(size_t)(ct * 16 + col) * Kw8 + 4 * s + kq];
    float4 bv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bv[ct] = *(const float4*)(bias + ct * 16 + kq * 4);
    // LDS offsets of this lane's operand chunk per k-step, relative to the pixel's patch origin (row 2*ly, pair lx)
    int koff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int q = min(4 * s + kq, 17);          // padded chunks multiply zero weights: any finite operand will do
        koff[s] = (q / 3) * STEM_PP + (q % 3);
    }
    const int ntiles = B * tiles_y * tiles_x;
    // patch of the NEXT tile travels through registers while this tile is multiplied and stored (3 chunks per thread)
    constexpr int NPRE = (STEM_PR * STEM_PC + 255) / 256;
    uint4 pre[NPRE];
    auto fetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * 256;
            const int pr = i / STEM_PC, pc = i - pr * STEM_PC;
            const int iy = 2 * oy0 - 2 + pr, ip = ox0 - 1 + pc;
            pre[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < STEM_PR * STEM_PC && iy >= 0 && iy < H && ip >= 0 && ip < Wp) {
                if constexpr (!U8) {
                    pre[k] = x[((size_t)b * H + iy) * Wp + ip];
                } else {
                    const int uy = iy - g.top, ux = 2 * ip - g.left;                  // left is even here: a pair is inside or outside as a whole
                    int pv[6] = {114, 114, 114, 114, 114, 114};
                    if (uy >= 0 && uy < g.unpad_h && ux >= 0 && ux + 1 < g.unpad_w) {
                        const uint16_t* q = (const uint16_t*)(src8 + (((size_t)b * g.src_h + uy) * g.src_w + ux) * 3);      // 2-byte aligned (even width)
                        const uint32_t h0 = q[0], h1 = q[1], h2 = q[2];
                        pv[0] = h0 & 255; pv[1] = h0 >> 8; pv[2] = h1 & 255; pv[3] = h1 >> 8; pv[4] = h2 & 255; pv[5] = h2 >> 8;
                    }
                    const int a0 = g.swap_rb ? pv[2] : pv[0], a2 = g.swap_rb ? pv[0] : pv[2];
                    const int b0 = g.swap_rb ? pv[5] : pv[3], b2 = g.swap_rb ? pv[3] : pv[5];
                    pre[k].x = pack2_bf16(div255_exact((float)a0), div255_exact((float)pv[1]));
                    pre[k].y = pack2_bf16(div255_exact((float)a2), 0.f);
                    pre[k].z = pack2_bf16(div255_exact((float)b0), div255_exact((float)pv[4]));
                    pre[k].w = pack2_bf16(div255_exact((float)b2), 0.f);
                }
            }
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int oy0 = ty * STEM_TH, ox0 = tx * STEM_TW;
        __syncthreads();                             // the previous tile's reads are done
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < STEM_PR * STEM_PC) { const int pr = i / STEM_PC; patch[pr * STEM_PP + (i - pr * STEM_PC)] = pre[k]; }
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        f32x4s acc[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = (f32x4s){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            ChunkS xf[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int ly = wave * 2 + (pt >> 1), lx = (pt & 1) * 16 + col;
                xf[pt].u = patch[(2 * ly) * STEM_PP + lx + koff[s]];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][ct].h, xf[pt].h, acc[ct][pt], 0, 0, 0);
        }
        // epilogue: bias + SiLU (same expression as conv_igemm.hip::act_apply, fast path).  A lane holds 4 consecutive channels of a
        // pixel; lane pairs (lane, lane ^ 16) swap halves across two pixel tiles so that each lane stores 8 channels (16 bytes) of
        // ONE pixel -- half the store instructions (see conv_igemm.hip::conv_epilogue).
        const bool odd = (kq & 1) != 0;
#pragma unroll
        for (int pt = 0; pt < 4; pt += 2) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                uint2 P[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v[4] = {acc[ct][pt + t][0] + bv[ct].x, acc[ct][pt + t][1] + bv[ct].y, acc[ct][pt + t][2] + bv[ct].z, acc[ct][pt + t][3] + bv[ct].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[j]));
                    typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
                    const bf16x2s p0 = {(__bf16)v[0], (__bf16)v[1]}, p1 = {(__bf16)v[2], (__bf16)v[3]};
                    P[t].x = __builtin_bit_cast(uint32_t, p0); P[t].y = __builtin_bit_cast(uint32_t, p1);
                }
                typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
                const u32x2s sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);      // lane <-> lane ^ 16, see conv_epilogue_bf16
                const u32x2s sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
                const uint4 o4 = make_uint4(sx.x, sy.x, sx.y, sy.y);
                const int ptm = pt + (odd ? 1 : 0);
                const int oy = oy0 + wave * 2 + (ptm >> 1), ox = ox0 + (ptm & 1) * 16 + col;
                if (oy < Ho && ox < Wo) {
                    *(uint4*)(y + (((size_t)b * Ho + oy) * Wo + ox) * out_cs + out_co + ct * 16 + (kq & ~1) * 4) = o4;
                    if (y8) {       // fp8 engine: the e4m3 copy the next layer reads, from the bf16-rounded values (= bf16_to_fp8_kernel on y)
                        const unsigned int wv[4] = {o4.x, o4.y, o4.z, o4.w};
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f[2 * j] = __builtin_amdgcn_fmed3f(__uint_as_float(wv[j] << 16) * q_inv_scale, -448.0f, 448.0f);
                            f[2 * j + 1] = __builtin_amdgcn_fmed3f(__uint_as_float(wv[j] & 0xffff0000u) * q_inv_scale, -448.0f, 448.0f);
                        }
                        unsigned int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0u, false); q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], q0, true);
                        unsigned int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0u, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], q1, true);
                        *(uint2*)(y8 + (((size_t)b * Ho + oy) * Wo + ox) * q_cs + q_co + ct * 16 + (kq & ~1) * 4) = make_uint2(q0, q1);
                    }
                }
            }
        }
    }
}

// p describes the stem as launch_conv sees it (pair view: H x W/2 x 8 channels, 6 x 3 kernel, stride (2,1), pad (2,1)).
bool stem_direct_applicable(const ConvP& p) {
    return p.prec == PREC_BF16 && p.kh == 6 && p.kw == 3 && p.sh == 2 && p.sw == 1 && p.ph == 2 && p.pw == 1 && p.Cin == 8 && p.in_cs == 8 &&
           p.in_co == 0 && p.act == ACT_SILU && p.res_mode == RES_NONE && !p.out_f32 && p.split == 0 && p.Cout % 16 == 0 && p.Cout >= 16 &&
           p.Cout <= 64 && p.Kp >= 160 && p.out_cs % 8 == 0 && p.out_co % 8 == 0;
}

static int launch_stem_impl(const ConvP& p, const uint8_t* src8, const LetterboxGeom& g, hipStream_t s, const View* q8 = nullptr, float q_inv_scale = 1.0f) {
    uint8_t* y8 = q8 ? (uint8_t*)q8->ptr : nullptr;
    const int q_cs = q8 ? q8->cs : 0, q_co = q8 ? q8->co : 0;
    const int tiles_x = (p.Wo + STEM_TW - 1) / STEM_TW, tiles_y = (p.Ho + STEM_TH - 1) / STEM_TH;
    const int ntiles = p.B * tiles_x * tiles_y;
    const int grid = std::min(ntiles, 256 * 3 - 64); // 3 workgroups per CU can be resident (VGPRs); 64 slots stay free for the tracker stream (conv_igemm.hip)
    const uint4* x = (const uint4*)p.in; const uint4* w = (const uint4*)p.w;
    uint16_t* y = (uint16_t*)p.out;
    const int kw8 = p.Kp / 8;
#define VC_STEM_LAUNCH(CT)                                                                                                          \
    if (src8) launch_timed(p, stem_direct_kernel<CT, true>, dim3(grid), dim3(256), 0, s, x, w, p.bias, y, p.B, p.H, p.W, p.Ho, p.Wo, kw8, \
                           p.out_cs, p.out_co, tiles_x, tiles_y, src8, g, y8, q_cs, q_co, q_inv_scale);                                  \
    else launch_timed(p, stem_direct_kernel<CT, false>, dim3(grid), dim3(256), 0, s, x, w, p.bias, y, p.B, p.H, p.W, p.Ho, p.Wo, kw8,    \
                      p.out_cs, p.out_co, tiles_x, tiles_y, src8, g, y8, q_cs, q_co, q_inv_scale);
    switch (p.Cout / 16) {
        case 1: VC_STEM_LAUNCH(1) break;
        case 2: VC_STEM_LAUNCH(2) break;
        case 3: VC_STEM_LAUNCH(3) break;
        case 4: VC_STEM_LAUNCH(4) break;
        default: return VC_ERR_ARG;
    }
#undef VC_STEM_LAUNCH
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_stem_direct(const ConvP& p, hipStream_t s, const View* q8, float q_inv_scale) { return launch_stem_impl(p, nullptr, LetterboxGeom{}, s, q8, q_inv_scale); }

// no-resize geometry only (the frame is already at network scale), pairs never straddle the image edge (even left pad and
// width), 2-byte aligned pair reads (even source width)
bool stem_u8_applicable(const ConvP& p, const LetterboxGeom& g) {
    return stem_direct_applicable(p) && g.unpad_h == g.src_h && g.unpad_w == g.src_w && g.src_w % 2 == 0 && g.left % 2 == 0 &&
           g.net_h == p.H && g.net_w == 2 * p.W;
}

int launch_stem_direct_u8(const ConvP& p, const uint8_t* frames, const LetterboxGeom& g, hipStream_t s, const View* q8, float q_inv_scale) {
    if (!frames || !stem_u8_applicable(p, g)) return VC_ERR_ARG;
    return launch_stem_impl(p, frames, g, s, q8, q_inv_scale);
}

}  // namespace vc
