// Device-side pieces shared by the convolution kernels (conv_igemm.hip, conv_halo_v2.hip): MFMA fragment types, the LDS swizzle,
// the in-place MFMA statement and the epilogues (bias, activation, residual, bf16 / fp8 pack, concat-slice and split-destination stores).
#pragma once
#include "vc_common.h"

namespace vc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
union Chunk {
    u32x4v u;
    bf16x8 h;
    float f[4];
};

// LDS rows hold KC chunks of 16 B (KC = 4: 64-byte rows, KC = 8: 128-byte rows).  ds_read_b128 is serviced in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table).  The chunk index is XOR-swizzled so
// that the 16 lanes of every group land on 16 distinct 16-byte slots of the 256-byte bank row:
//   KC = 4: chunk ^ perm[(row>>2)&3], perm = {0,2,3,1};   KC = 8: chunk ^ (row & 7).
template <int KC>
__device__ __forceinline__ int lds_slot(int row, int chunk) {
    if constexpr (KC == 4) return row * 4 + (chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3));
    else return row * 8 + (chunk ^ (row & 7));
}

__device__ __forceinline__ float act_apply(float v, int act, bool precise) {
    if (act == ACT_SILU) {
        return precise ? v / (1.0f + expf(-v)) : v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    }
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {       // v_cvt_pk_bf16_f32 (RNE)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// MFMA with the accumulator updated in place, as inline assembly.  With the builtin, the register allocator renames the
// accumulators of the halo kernel's 9-tap unrolled loop (vdst != src C on a third of the MFMAs) and repairs the rotation
// with ~90 v_accvgpr_read/write copies + s_nop stalls per 144 MFMAs.  The compiler cannot see that this statement is an
// MFMA, so the wait states it would insert before a VALU reads the result are provided by mfma_results_settle().
__device__ __forceinline__ void mfma_bf16_inplace(f32x4& c, const u32x4v& a, const u32x4v& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// The other direction: the compiler is free to materialise the accumulators' zeros with v_mov right in front of the first
// inline-asm MFMA, which then reads src C before the VALU write has landed (seen: the last two v_mov of a tile).  Tying the
// accumulators to an asm statement forces the zeros into registers here, the s_nop covers the VALU-write -> MFMA-read wait states.
template <int N>
__device__ __forceinline__ void mfma_inputs_settle(f32x4* acc) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(acc[i]));
    asm volatile("s_nop 4" ::: "memory");
}
// >= 18 wait states between the last MFMA and the first VALU read of an accumulator (CDNA3/4 ISA: XDL write VGPR -> VALU
// read, 8-pass MFMA: 11), tied to every accumulator so that no read is scheduled above it.
template <int N>
__device__ __forceinline__ void mfma_results_settle(f32x4* acc) {
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(acc[i]));
}

// bf16 epilogue with 16-byte stores.  The MFMA layout leaves 4 consecutive channels of one pixel in a lane (an 8-byte store);
// lane pairs (lane, lane ^ 16) hold channels [8g, 8g+4) and [8g+4, 8g+8) of the SAME pixels, so for two pixel tiles they swap
// halves -- the even lane ends with 8 channels of the first tile's pixel, the odd lane with 8 channels of the second tile's
// pixel -- and each stores one dwordx4.  Half the store instructions: the store tail of these kernels is issue-bound
// (MI355X_MICROARCH.md, "epilogue store tail").  Residual reads of the whole wave tile are issued before the first value is
// touched (one memory latency per tile instead of one per 16x16 block).  SiLU = x * rcp(1 + exp2(-x * log2 e)): v_exp_f32 and
// v_rcp_f32 directly (1 ulp each, far below the bf16 rounding that follows) -- a correctly rounded division costs 11 more
// VALU instructions per value, and the epilogue is the larger part of the 1x1 layers.
// Preconditions (checked by the caller): Cout, channel strides and offsets multiples of 8.
typedef unsigned int u32x2r __attribute__((ext_vector_type(2)));
// the residual values of a wave tile, as conv_epilogue_bf16 reads them (one 8-byte load per 16 x 16 block and lane)
template <int PT, int CT>
__device__ __forceinline__ void conv_residual_fetch(const ConvP& p, u32x2r (&R)[PT][CT], int mbase, int nbase, int frow) {
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
    for (int b = 0; b < PT; ++b) {
        const int m = mbase + b * 16 + frow;
        const int rrow = m * p.res_cs + p.res_co;
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = nbase + a * 16;
            R[b][a] = __builtin_amdgcn_raw_buffer_load_b64(rsrd, (n < p.Cout && m < p.M) ? (rrow + n) * 2 : 0, 0, 0);
        }
    }
}
// Rpre / have_pre: the same values fetched ahead by the caller (conv3x3_halo_kernel reads them before its K loop: a one-tile workgroup
// has nothing to overlap the residual's memory latency with at the end of its life). By reference and a flag, not a pointer that may be
// null: a selected pointer sends the array to scratch memory.
template <int PT, int CT, int ACT, int RES>
__device__ __forceinline__ void conv_epilogue_bf16(const ConvP& p, f32x4 (&acc)[CT][PT], const float4 (&bias)[CT], int mbase, int nbase, int frow,
                                                   const u32x2r (&Rpre)[PT][CT], bool have_pre) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t osrd2 = __builtin_amdgcn_make_buffer_rsrc(p.split > 0 ? p.out2 : p.out, 0, 0x7ffffff0, 0x00020000);
    u32x2 R[PT][CT];
    if constexpr (RES != RES_NONE) {
        if (have_pre) {
#pragma unroll
            for (int b = 0; b < PT; ++b)
#pragma unroll
                for (int a = 0; a < CT; ++a) R[b][a] = Rpre[b][a];
        } else {
            conv_residual_fetch<PT, CT>(p, R, mbase, nbase, frow);
        }
    }
    const bool odd = ((threadIdx.x >> 4) & 1) != 0;
    const bool no_store = p.ablate == 2 || p.ablate == 3;      // timing experiment (VC_CONV_ABLATE): everything but the stores
#pragma unroll
    for (int b = 0; b < PT; b += 2) {
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = nbase + a * 16;
            const bool grp_ok = n < p.Cout;                     // the same for both lanes of a pair (Cout % 8 == 0)
            u32x2 P[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[4] = {acc[a][b + t][0] + bias[a].x, acc[a][b + t][1] + bias[a].y, acc[a][b + t][2] + bias[a].z, acc[a][b + t][3] + bias[a].w};
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (RES != RES_NONE) {
                    const u32x2 r = R[b + t][a];
                    rv[0] = __uint_as_float(r.x << 16); rv[1] = __uint_as_float(r.x & 0xffff0000u);
                    rv[2] = __uint_as_float(r.y << 16); rv[3] = __uint_as_float(r.y & 0xffff0000u);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j];
                    if constexpr (RES == RES_BEFORE_ACT) x += rv[j];
                    if constexpr (ACT == ACT_SILU) x = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
                    if constexpr (ACT == ACT_RELU) x = x > 0.f ? x : 0.f;
                    if constexpr (RES == RES_AFTER_ACT) x += rv[j];
                    v[j] = x;
                }
                P[t] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
            // v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second, i.e. exactly the exchange
            // between lane and lane ^ 16 described above, in one VALU instruction per dword (no LDS-pipe round trip)
            const u32x2 sx = __builtin_amdgcn_permlane16_swap(P[0].x, P[1].x, false, false);
            const u32x2 sy = __builtin_amdgcn_permlane16_swap(P[0].y, P[1].y, false, false);
            const u32x4 o4 = {sx.x, sy.x, sx.y, sy.y};
            const int m = mbase + (b + (odd ? 1 : 0)) * 16 + frow;
            const int nn = odd ? n - 4 : n;
            // masked lanes store to an out-of-range offset, which the buffer unit drops: no exec-mask branches around the stores
            const bool ok = grp_ok && m < p.M && !no_store;
            const int off1 = (m * p.out_cs + p.out_co + nn) * 2;
            if (p.split == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(o4, osrd, ok ? off1 : (int)0x80000000u, 0, 0);
            } else {
                const bool second = nn >= p.split;
                __builtin_amdgcn_raw_buffer_store_b128(o4, osrd, (ok && !second) ? off1 : (int)0x80000000u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o4, osrd2, (ok && second) ? (m * p.out2_cs + p.out2_co + nn - p.split) * 2 : (int)0x80000000u, 0, 0);
            }
        }
    }
}

// fp8 (OCP e4m3fn) epilogue of the MX-scaled path: value = acc * scale[channel] + bias[channel] with scale = weight scale x activation
// scale (the block scales of the MFMA itself are 1), activation, residual (stored fp8, times the activation scale), then either
// fp8 again (divide by the activation scale, clamp to +-448: the conversion does not saturate by itself, v_cvt_pk_fp8_f32) or,
// for the Detect heads, the dequantised value as bf16.  Lane pairs (lane, lane ^ 16) exchange halves exactly as in
// conv_epilogue_bf16, so fp8 stores are 8 bytes (8 channels of one pixel).  Preconditions: Cout, strides, offsets multiples of 8.
template <int PT, int CT>
__device__ __forceinline__ void conv_epilogue_fp8(const ConvP& p, f32x4 (&acc)[CT][PT], int mbase, int nbase, int frow) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t osrd2 = __builtin_amdgcn_make_buffer_rsrc(p.split > 0 ? p.out2 : p.out, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res_mode != RES_NONE ? p.res : (const void*)p.bias), 0, 0x7ffffff0, 0x00020000);
    float4 bias[CT], scl[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) {
        const bool in = nbase + a * 16 < p.Cout;
        bias[a] = in ? *(const float4*)(p.bias + nbase + a * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        scl[a] = in ? *(const float4*)(p.scale + nbase + a * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool odd = ((threadIdx.x >> 4) & 1) != 0;
    const float inv_s = p.inv_act_scale, rs = p.act_scale;
#pragma unroll
    for (int b = 0; b < PT; b += 2) {
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = nbase + a * 16;
            const bool grp_ok = n < p.Cout;
            unsigned int P[2];
            u32x2 Q[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = mbase + (b + t) * 16 + frow;
                float v[4] = {acc[a][b + t][0] * scl[a].x + bias[a].x, acc[a][b + t][1] * scl[a].y + bias[a].y,
                              acc[a][b + t][2] * scl[a].z + bias[a].z, acc[a][b + t][3] * scl[a].w + bias[a].w};
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.res_mode != RES_NONE) {
                    const unsigned int r = __builtin_amdgcn_raw_buffer_load_b32(rsrd, (grp_ok && m < p.M) ? m * p.res_cs + p.res_co + n : 0, 0, 0);
                    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(r, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(r, true);
                    rv[0] = lo[0] * rs; rv[1] = lo[1] * rs; rv[2] = hi[0] * rs; rv[3] = hi[1] * rs;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j];
                    if (p.res_mode == RES_BEFORE_ACT) x += rv[j];
                    if (p.act == ACT_SILU) x = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
                    else if (p.act == ACT_RELU) x = x > 0.f ? x : 0.f;
                    if (p.res_mode == RES_AFTER_ACT) x += rv[j];
                    v[j] = x;
                }
                if (p.out_bf16) {
                    Q[t] = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * inv_s, -448.0f, 448.0f);
                    unsigned int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0u, false);
                    P[t] = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                }
            }
            if (p.out_bf16) {                                     // dequantised bf16 output (Detect logits): 8-byte stores per tile
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int m = mbase + (b + t) * 16 + frow;
                    const bool ok = grp_ok && m < p.M;
                    __builtin_amdgcn_raw_buffer_store_b64(Q[t], osrd, ok ? (m * p.out_cs + p.out_co + n) * 2 : (int)0x80000000u, 0, 0);
                }
                continue;
            }
            const u32x2 sx = __builtin_amdgcn_permlane16_swap(P[0], P[1], false, false);
            const int m = mbase + (b + (odd ? 1 : 0)) * 16 + frow;
            const int nn = odd ? n - 4 : n;
            const bool ok = grp_ok && m < p.M;
            const int off1 = m * p.out_cs + p.out_co + nn;
            if (p.split == 0) {
                __builtin_amdgcn_raw_buffer_store_b64(sx, osrd, ok ? off1 : (int)0x80000000u, 0, 0);
            } else {
                const bool second = nn >= p.split;
                __builtin_amdgcn_raw_buffer_store_b64(sx, osrd, (ok && !second) ? off1 : (int)0x80000000u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(sx, osrd2, (ok && second) ? m * p.out2_cs + p.out2_co + nn - p.split : (int)0x80000000u, 0, 0);
            }
        }
    }
}

// Epilogue shared by the conv kernels: D[channel = (lane>>4)*4 + reg][pixel = lane&15] -> bias, activation, residual, bf16
// pack, concat-slice / split-destination store.  mbase = first pixel of the wave's tile, nbase = this lane's first channel.
template <int PT, int CT, int ACT, int RES>
__device__ __forceinline__ void conv_epilogue_bf16(const ConvP& p, f32x4 (&acc)[CT][PT], const float4 (&bias)[CT], int mbase, int nbase, int frow) {
    u32x2r none[PT][CT];                                   // never read
    conv_epilogue_bf16<PT, CT, ACT, RES>(p, acc, bias, mbase, nbase, frow, none, false);
}
template <int PT, int CT>
__device__ __forceinline__ bool conv_epilogue_fast_bf16(const ConvP& p) {        // the preconditions of conv_epilogue_bf16 (uniform)
    return !p.out_f32 && p.Cout % 8 == 0 && p.out_cs % 8 == 0 && p.out_co % 8 == 0 &&
           (p.split == 0 || (p.split % 8 == 0 && p.out2_cs % 8 == 0 && p.out2_co % 8 == 0));
}
template <int PT, int CT, bool F32>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[CT][PT], int mbase, int nbase, int frow, const u32x2r (&Rpre)[PT][CT], bool have_pre) {
    float4 bias[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) bias[a] = nbase + a * 16 < p.Cout ? *(const float4*)(p.bias + nbase + a * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    // 32-bit element offsets + buffer stores (SGPR descriptors): no 64-bit address arithmetic per tile
    const __amdgpu_buffer_rsrc_t osrd = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t osrd2 = __builtin_amdgcn_make_buffer_rsrc(p.split > 0 ? p.out2 : p.out, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res_mode != RES_NONE ? p.res : p.bias), 0, 0x7ffffff0, 0x00020000);
    const bool wide_out = F32 || p.out_f32;
    const int OES = wide_out ? 4 : 2;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // bf16 fast path: conv_epilogue_bf16 above
    if constexpr (!F32 && PT % 2 == 0) {
        const bool fast = conv_epilogue_fast_bf16<PT, CT>(p);
        if (fast) {
            // activation / residual mode are launch constants: one straight-line instance per combination the networks use
            // (no per-value scalar branches), anything else takes the general loop below
            const int key = p.act * 4 + p.res_mode;
            if (key == ACT_SILU * 4 + RES_NONE) { conv_epilogue_bf16<PT, CT, ACT_SILU, RES_NONE>(p, acc, bias, mbase, nbase, frow, Rpre, false); return; }
            if (key == ACT_SILU * 4 + RES_AFTER_ACT) { conv_epilogue_bf16<PT, CT, ACT_SILU, RES_AFTER_ACT>(p, acc, bias, mbase, nbase, frow, Rpre, have_pre); return; }
            if (key == ACT_RELU * 4 + RES_NONE) { conv_epilogue_bf16<PT, CT, ACT_RELU, RES_NONE>(p, acc, bias, mbase, nbase, frow, Rpre, false); return; }
            if (key == ACT_RELU * 4 + RES_BEFORE_ACT) { conv_epilogue_bf16<PT, CT, ACT_RELU, RES_BEFORE_ACT>(p, acc, bias, mbase, nbase, frow, Rpre, have_pre); return; }
            if (key == ACT_NONE * 4 + RES_NONE) { conv_epilogue_bf16<PT, CT, ACT_NONE, RES_NONE>(p, acc, bias, mbase, nbase, frow, Rpre, false); return; }
        }
    }
#pragma unroll
    for (int b = 0; b < PT; ++b) {
        const int m = mbase + b * 16 + frow;
        if (m >= p.M) continue;
        const int orow = m * p.out_cs + p.out_co, rrow = m * p.res_cs + p.res_co, orow2 = m * p.out2_cs + p.out2_co;
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int n = nbase + a * 16;
            if (n >= p.Cout) continue;
            float v[4] = {acc[a][b][0] + bias[a].x, acc[a][b][1] + bias[a].y, acc[a][b][2] + bias[a].z, acc[a][b][3] + bias[a].w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            const int nvalid = p.Cout - n >= 4 ? 4 : p.Cout - n;
            if (p.res_mode != RES_NONE) {
                if constexpr (F32) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrd, (rrow + n) * 4, 0, 0);
                    rv[0] = __uint_as_float(t.x); rv[1] = __uint_as_float(t.y); rv[2] = __uint_as_float(t.z); rv[3] = __uint_as_float(t.w);
                } else {
                    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrd, (rrow + n) * 2, 0, 0);
                    rv[0] = __uint_as_float(t.x << 16); rv[1] = __uint_as_float(t.x & 0xffff0000u);
                    rv[2] = __uint_as_float(t.y << 16); rv[3] = __uint_as_float(t.y & 0xffff0000u);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j];
                if (p.res_mode == RES_BEFORE_ACT) t += rv[j];
                t = act_apply(t, p.act, F32);
                if (p.res_mode == RES_AFTER_ACT) t += rv[j];
                v[j] = t;
            }
            const bool second = p.split > 0 && n >= p.split;            // uniform per 4-channel group
            const int eoff = (second ? orow2 + (n - p.split) : orow + n) * OES;
            if (nvalid == 4) {
                if (wide_out) {
                    const u32x4 t = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    if (second) __builtin_amdgcn_raw_buffer_store_b128(t, osrd2, eoff, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(t, osrd, eoff, 0, 0);
                } else {
                    const u32x2 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    if (second) __builtin_amdgcn_raw_buffer_store_b64(t, osrd2, eoff, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b64(t, osrd, eoff, 0, 0);
                }
            } else {                                                      // ragged channel tail (e.g. Detect's 255 outputs)
                char* ob = (char*)(second ? p.out2 : p.out) + eoff;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nvalid) {
                        if (wide_out) ((float*)ob)[j] = v[j];
                        else ((uint16_t*)ob)[j] = f32_to_bf16(v[j]);
                    }
            }
        }
    }
}

template <int PT, int CT, bool F32>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x4 (&acc)[CT][PT], int mbase, int nbase, int frow) {
    u32x2r none[PT][CT];                                   // never read
    conv_epilogue<PT, CT, F32>(p, acc, mbase, nbase, frow, none, false);
}

// ---- halo-staged 3x3 / s1 / p1: what the launchers of both halo kernels share ---------------------------------------------
// patch buffer sizes: XI x 64 pixels of 64 B (the last one is the zero pixel); the smallest that holds the layer's patch is
// used, because the patch buffers decide how many workgroups share a CU (2 x 16 / 28 / 44 KB)
static inline int halo_patch_pixels(const ConvP& p, int bp) {
    const int rows = (bp - 1 + p.W - 1) / p.W + 1 + 2;               // worst case: a tile that starts at the end of a row
    return rows * p.W;
}
static inline bool halo_applicable(const ConvP& p, int bp) {
    if (p.prec != PREC_BF16 || p.kh != 3 || p.kw != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1) return false;
    if (p.Cin % 32 != 0 || p.in_cs % 8 != 0 || p.in_co % 8 != 0 || p.Ho != p.H || p.Wo != p.W) return false;
    return halo_patch_pixels(p, bp) <= 11 * 64 - 1;
}

}  // namespace vc
