// A 64-channel ReID BasicBlock in one kernel (bf16): t = ReLU(conv1 x), y = ReLU(conv2 t + x), both 3x3 / pad 1, 64 -> 64, on the
// 25 x 25 maps of layer1 -- /root/reference/networks/deepsort/deep/model.py:5-38 (BasicBlock.forward, is_downsample = False) as
// make_layers(64, 64, 2) builds it twice (:61), reached from deep_sort.py:119-129 -> feature_extractor.py:42-47.
//
// Unfused, each of the two blocks is two launches that move t (80 KB per crop) to HBM and back and read x a second time for the
// residual: 615 MB per 1 536 crops and block against 246 MB for x in + y out, at 22 - 28 % of the MFMA rate (profiles/r03_*: the four
// layer1 convs are 0.46 ms of a 128-frame step).  Here a workgroup owns a whole crop: x sits in LDS on a zero-padded 27 x 27 raster
// (101 KB with three spare zero rows, 128-byte pixels, 16-byte chunks XOR-swizzled with conv3x3_halo_kernel's formula), conv1's 625 x 64 outputs stay in the
// accumulators of the eight waves (5 pixel tiles x 4 channel tiles each) until every wave has finished reading x, then t overwrites
// x in place and conv2 runs from the same raster; the residual is re-read from global memory (L2-resident: the workgroup fetched it a
// few microseconds earlier).  Weights stream one kernel ROW at a time (three taps, 24 KB in MFMA fragment order) by LDS-DMA into a
// two-stage ring, one row ahead of their use; one workgroup barrier per row, six rows per block; the fragment reads are software-pipelined
// by hand (inline ds_read_b128 with counted waits).  One workgroup per CU (149 KB of LDS), persistent over the crops.
//
// MFMA operand order and k order (tap-major, then the tap's two 32-channel halves) equal conv_igemm_kernel's, the epilogues are
// conv_epilogue_bf16's expressions (bias, residual BEFORE the activation, ReLU, v_cvt_pk_bf16_f32): bit-identical to the two launches.
#include <algorithm>

#include "kernels.h"

namespace vc {

typedef float f32x4r __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
union ChunkR { uint4 u; bf16x8r h; };

#define RB_HW 25                      // map side (50 x 50 crops after the stem's 3 / 2 / 1 max pool)
#define RB_PW (RB_HW + 2)             // padded raster row
#define RB_NPIX (RB_HW * RB_HW)       // 625
#define RB_NSLOT (RB_PW * (RB_PW + 3))   // 810 pixel slots: the 27 x 27 padded map + three all-zero rows that the 15 lanes past pixel 624 read
#define RB_ZSLOT (RB_PW * RB_PW)      // their window's top-left corner (rows 27 - 29, columns 0 - 2: never written)
#define RB_NW 8
#define RB_PT 5                       // pixel tiles per wave: 8 x 5 x 16 = 640 >= 625
#define RB_TAPW 512                   // uint4 of one tap's weights: 64 channels x 64 k x 2 B
#define RB_ROWB (3 * RB_TAPW * 16)    // bytes of one ring stage = one kernel row (three taps)

struct RbArgs {
    const uint4 *w1, *w2;
    const float *b1, *b2;
    int kw1, kw2;                      // weight row strides in 16-byte chunks
    const uint16_t* x; int in_cs, in_co;
    uint16_t* y; int out_cs, out_co;
    int k;                             // crops
};

__device__ __forceinline__ int rb_addr(int slot, int chunk) { return slot * 128 + ((chunk ^ (((slot >> 1) & 3) << 1)) << 4); }   // byte offset in the raster

__global__ __launch_bounds__(RB_NW * 64) void reid_block_fused_kernel(const RbArgs a) {
    __shared__ uint4 xs[RB_NSLOT * 8];                     // 101 KB: x, then t, [pixel slot][8 chunks], swizzled
    __shared__ uint4 wr[2][3 * RB_TAPW];                   // 48 KB: weight ring, [stage = kernel row][tap of the row][k half][channel tile][lane]
    typedef __attribute__((address_space(3))) void* lds_ptr_q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int col = lane & 15, kq = lane >> 4;
    char* xb = (char*)xs;
    const uint32_t lds_x = (uint32_t)(uintptr_t)(lds_ptr_q)&xs[0], lds_w = (uint32_t)(uintptr_t)(lds_ptr_q)&wr[0][0];
    for (int i = tid; i < RB_NSLOT * 8; i += RB_NW * 64) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    // this lane's pixel of each of the wave's five pixel tiles: raster slot of its 3x3 window's top-left corner
    int slot0[RB_PT];
#pragma unroll
    for (int pt = 0; pt < RB_PT; ++pt) {
        const int p = (wave * RB_PT + pt) * 16 + col;
        const int py = p / RB_HW, px = p - py * RB_HW;
        slot0[pt] = p < RB_NPIX ? py * RB_PW + px : RB_ZSLOT;
    }
    // weights by LDS-DMA, one kernel row (three taps) per ring stage: lane l of wave w fetches fragment element tid = 64 w + l of every tap --
    // k half tid / 256, channel tile (tid / 64) % 4, fragment lane tid % 64 = chunk 8 tap + 4 (k half) + l / 16 of channel 16 ct + l % 16
    const __amdgpu_buffer_rsrc_t w1rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.w1), 0, 64 * a.kw1 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.w2), 0, 64 * a.kw2 * 16, 0x00020000);
    const int wrow = ((tid >> 6) & 3) * 16 + (tid & 15), wchunk = 4 * (tid >> 8) + ((tid >> 4) & 3);
    const int wo1 = (wrow * a.kw1 + wchunk) * 16, wo2 = (wrow * a.kw2 + wchunk) * 16;
    auto wdma = [&](int rw, int stage) {                    // row rw = 0 .. 2 of conv1, 3 .. 5 of conv2
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tp = (rw < 3 ? rw : rw - 3) * 3 + j;
            lds_ptr_q dst = (lds_ptr_q)&wr[stage][j * RB_TAPW + uwave * 64];
            if (rw < 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rd, dst, 16, wo1 + tp * 128, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rd, dst, 16, wo2 + tp * 128, 0, 0, 0);
        }
    };
    __syncthreads();
    for (int crop = blockIdx.x; crop < a.k; crop += gridDim.x) {
        // ---- x of this crop into the raster's interior; row 0 of conv1 into ring stage 0 ------------------------------------------
        wdma(0, 0);
        const uint16_t* xc = a.x + (size_t)crop * RB_NPIX * a.in_cs + a.in_co;
        for (int i = tid; i < RB_NPIX * 8; i += RB_NW * 64) {
            const int p = i >> 3, c = i & 7;
            const int py = p / RB_HW, px = p - py * RB_HW;
            *(uint4*)(xb + rb_addr((py + 1) * RB_PW + px + 1, c)) = *(const uint4*)(xc + (size_t)p * a.in_cs + c * 8);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4r acc[4][RB_PT];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int pt = 0; pt < RB_PT; ++pt) acc[ct][pt] = (f32x4r){0.f, 0.f, 0.f, 0.f};
        // six steps of one kernel ROW each (three taps): conv1 rows 0 - 2, conv2 rows 0 - 2; one workgroup barrier per row
#pragma unroll
        for (int rw = 0; rw < 6; ++rw) {
            if (rw == 3) {
                // ---- t = ReLU(conv1 x + b1) over x (every wave is past its last read of x: the barrier that ended row 2) ------------------
                float4 bv1[4];                              // (loaded here, not kept across the tap loops: registers the MFMA loops need)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bv1[ct] = *(const float4*)(a.b1 + ct * 16 + kq * 4);
#pragma unroll
                for (int pt = 0; pt < RB_PT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        float v[4] = {acc[ct][pt][0] + bv1[ct].x, acc[ct][pt][1] + bv1[ct].y, acc[ct][pt][2] + bv1[ct].z, acc[ct][pt][3] + bv1[ct].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                        const uint2 o = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
                        // channels 16 ct + 4 kq .. + 3 of the pixel: chunk 2 ct + kq / 2, its half kq % 2
                        if (slot0[pt] < RB_ZSLOT) *(uint2*)(xb + rb_addr(slot0[pt] + RB_PW + 1, ct * 2 + (kq >> 1)) + (kq & 1) * 8) = o;
                        acc[ct][pt] = (f32x4r){0.f, 0.f, 0.f, 0.f};
                    }
                __syncthreads();
            }
            // the NEXT row's weights into the other stage: it was last read during row rw - 1, which every wave left at the previous barrier
            if (rw + 1 < 6) wdma(rw + 1, (rw + 1) & 1);
            const int dy = rw < 3 ? rw : rw - 3;
            // Six half-steps per row (tap dx = hs / 2, k half s = hs % 2), software-pipelined by hand over two fragment register sets: the
            // nine ds_read_b128 of half-step hs + 2 are issued right after the MFMAs of half-step hs, and the wait in front of a half-step's
            // MFMAs is counted (LDS reads return in order: the nine newer ones may still be in flight).  Inline assembly with explicit
            // s_waitcnt like the halo kernels: left to hipcc, every fragment read was followed by s_waitcnt lgkmcnt(0) and ONE MFMA (the
            // addresses shared a register), an LDS round trip per MFMA -- 0.20 ms per block, no faster than the two launches.
            const uint32_t wsb = lds_w + (uint32_t)(rw & 1) * RB_ROWB + (uint32_t)lane * 16;
            u32x4q wf[2][4], xf[2][RB_PT];
            uint32_t xa[RB_PT];                             // raster byte offset of this lane's pixel for the current tap, k half 0 (half 1 = ^ 64: chunk bit 2)
#define RB_ISSUE(hs, set)                                                                                                                          \
            {                                                                                                                                       \
                constexpr int dx_ = (hs) >> 1, s_ = (hs) & 1;                                                                                        \
                if (s_ == 0) {                                                                                                                       \
                    _Pragma("unroll") for (int pt = 0; pt < RB_PT; ++pt) {                                                                           \
                        int sl = slot0[pt];                                                                                                          \
                        asm volatile("" : "+v"(sl));       /* opaque: or the addresses of all 18 taps x 5 tiles are hoisted out of the crop loop */    \
                        xa[pt] = (uint32_t)rb_addr(sl + dy * RB_PW + dx_, kq);                                                                \
                    }                                                                                                                                \
                }                                                                                                                                    \
                _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                                                                     \
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[set][ct]) : "v"(wsb), "n"((dx_ * RB_TAPW + (s_ * 4 + ct) * 64) * 16) : "memory"); \
                _Pragma("unroll") for (int pt = 0; pt < RB_PT; ++pt)                                                                                 \
                    asm volatile("ds_read_b128 %0, %1" : "=v"(xf[set][pt]) : "v"(lds_x + (xa[pt] ^ (uint32_t)(s_ << 6))) : "memory");                            \
            }
            RB_ISSUE(0, 0);
            RB_ISSUE(1, 1);
#pragma unroll
            for (int hs = 0; hs < 6; ++hs) {
                if (hs < 5) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) asm volatile("" : "+v"(wf[hs & 1][ct]));      // the MFMAs below may not move above the wait
#pragma unroll
                for (int pt = 0; pt < RB_PT; ++pt) asm volatile("" : "+v"(xf[hs & 1][pt]));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int pt = 0; pt < RB_PT; ++pt)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, wf[hs & 1][ct]), __builtin_bit_cast(bf16x8r, xf[hs & 1][pt]), acc[ct][pt], 0, 0, 0);
                if (hs == 0) RB_ISSUE(2, 0);
                if (hs == 1) RB_ISSUE(3, 1);
                if (hs == 2) RB_ISSUE(4, 0);
                if (hs == 3) RB_ISSUE(5, 1);
            }
#undef RB_ISSUE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  // the next row's weights have landed
            __syncthreads();
        }
        // ---- y = ReLU(conv2 t + b2 + x): the residual from global memory (this workgroup read the crop a moment ago) -----------------
        uint16_t* yc = a.y + (size_t)crop * RB_NPIX * a.out_cs + a.out_co;
        float4 bv2[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) bv2[ct] = *(const float4*)(a.b2 + ct * 16 + kq * 4);
#pragma unroll
        for (int pt = 0; pt < RB_PT; ++pt) {
            const int pp = (wave * RB_PT + pt) * 16 + col;
            const bool valid = pp < RB_NPIX;
            const int p = valid ? pp : 0;
            uint2 r[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) r[ct] = *(const uint2*)(xc + (size_t)p * a.in_cs + ct * 16 + kq * 4);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float rv[4] = {__uint_as_float(r[ct].x << 16), __uint_as_float(r[ct].x & 0xffff0000u), __uint_as_float(r[ct].y << 16),
                                     __uint_as_float(r[ct].y & 0xffff0000u)};
                float v[4] = {acc[ct][pt][0] + bv2[ct].x, acc[ct][pt][1] + bv2[ct].y, acc[ct][pt][2] + bv2[ct].z, acc[ct][pt][3] + bv2[ct].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += rv[j]; v[j] = v[j] > 0.f ? v[j] : 0.f; }
                if (valid) *(uint2*)(yc + (size_t)p * a.out_cs + ct * 16 + kq * 4) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
            }
        }
        // (the loop's last barrier separates this crop's reads of t from the next crop's writes of x; the epilogue above touches no LDS)
    }
}

// p1 = BasicBlock.conv1, p2 = BasicBlock.conv2 as engine.hip::reid_forward_chunk builds them (no downsample branch)
bool reid_block_fused_applicable(const ConvP& p1, const ConvP& p2) {
    auto conv3 = [](const ConvP& p) {
        return p.prec == PREC_BF16 && p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.ph == 1 && p.pw == 1 && p.Cin == 64 && p.Cout == 64 && p.act == ACT_RELU &&
               !p.out_f32 && p.split == 0 && !p.m_dev && p.H == RB_HW && p.W == RB_HW && p.Ho == RB_HW && p.Wo == RB_HW && p.Kp >= 576;
    };
    if (!conv3(p1) || !conv3(p2) || p1.res_mode != RES_NONE || p2.res_mode != RES_BEFORE_ACT) return false;
    if (!(p2.in == p1.out && p2.in_cs == p1.out_cs && p2.in_co == p1.out_co && p2.B == p1.B)) return false;
    if (!(p2.res == p1.in && p2.res_cs == p1.in_cs && p2.res_co == p1.in_co)) return false;          // the residual is the block's input
    if (p2.out == p1.in) return false;                                                               // other crops' workgroups still read it
    return p1.in_cs % 8 == 0 && p1.in_co % 8 == 0 && p2.out_cs % 4 == 0 && p2.out_co % 4 == 0;
}

int launch_reid_block_fused(const ConvP& p1, const ConvP& p2, hipStream_t s) {
    if (!reid_block_fused_applicable(p1, p2)) return VC_ERR_ARG;
    RbArgs a{};
    a.w1 = (const uint4*)p1.w; a.w2 = (const uint4*)p2.w; a.b1 = p1.bias; a.b2 = p2.bias;
    a.kw1 = p1.Kp / 8; a.kw2 = p2.Kp / 8;
    a.x = (const uint16_t*)p1.in; a.in_cs = p1.in_cs; a.in_co = p1.in_co;
    a.y = (uint16_t*)p2.out; a.out_cs = p2.out_cs; a.out_co = p2.out_co;
    a.k = p1.B;
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n;
    }();
    launch_timed(p1, reid_block_fused_kernel, dim3(std::min(a.k, cus)), dim3(RB_NW * 64), 0, s, a);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
