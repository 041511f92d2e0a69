// Shared declarations for libvcount_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/vcount_hip.h"

namespace vc {

// ---- error plumbing: every C-ABI entry returns a status, message kept per thread ---------------
void set_error(const char* fmt, ...);
const char* last_error();

#define VC_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            vc::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            return VC_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

#define VC_CHECK(cond, code, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            vc::set_error(__VA_ARGS__);           \
            return (code);                        \
        }                                         \
    } while (0)

#define VC_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != VC_OK) return _s;   \
    } while (0)

// ---- element types --------------------------------------------------------------------------
enum Prec : int { PREC_BF16 = 0, PREC_F32 = 1, PREC_FP8 = 2 };   // PREC_FP8: OCP e4m3fn activations + weights on the MX-scaled K = 128 MFMA
static inline int elem_size(int prec) { return prec == PREC_BF16 ? 2 : prec == PREC_FP8 ? 1 : 4; }

// round-to-nearest-even f32 -> OCP e4m3fn (1-4-3, bias 7, max 448, no infinities), saturating; host-side weight packing and tests
static inline uint8_t f32_to_e4m3(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    float a = f < 0 ? -f : f;
    if (!(a == a)) return (uint8_t)(sign | 0x7f);
    if (a >= 448.0f) return (uint8_t)(sign | 0x7e);               // saturate to +-448
    if (a < 0.0009765625f) return sign;                            // below half of the smallest subnormal (2^-10)
    int e;
    const float m = frexpf(a, &e);                                 // a = m * 2^e, m in [0.5, 1)
    int exp = e - 1;                                               // a = (2m) * 2^exp, 2m in [1, 2)
    if (exp < -6) {                                                // subnormal: value = k * 2^-9, k = 0..7
        const float k = nearbyintf(a * 512.0f);
        return (uint8_t)(sign | (k >= 8.0f ? 0x08 : (uint8_t)k));
    }
    float frac = nearbyintf((2.0f * m - 1.0f) * 8.0f);             // 3 mantissa bits, RNE (default rounding mode)
    if (frac >= 8.0f) { frac = 0.0f; ++exp; }
    if (exp > 8 || (exp == 8 && frac > 6.0f)) return (uint8_t)(sign | 0x7e);
    return (uint8_t)(sign | ((exp + 7) << 3) | (int)frac);
}
static inline float e4m3_to_f32(uint8_t b) {
    const int e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? (float)m * 0.001953125f : ldexpf(1.0f + (float)m / 8.0f, e - 7);
    if (e == 15 && m == 7) v = NAN;
    return (b & 0x80) ? -v : v;
}

// round-to-nearest-even f32 -> bf16 (matches torch .bfloat16() and v_cvt_pk_bf16_f32)
__host__ __device__ static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

#ifdef __HIPCC__
// (float)p / 255.f for an integer p in 0..255 without the division sequence: q = p * r, one residual correction
// q' = fma(fma(-255, q, p), r, q) with r = RN(1/255).  Equal to the IEEE quotient for all 256 inputs (checked exhaustively in
// exact arithmetic, tools/div255_check.py; the fp32 detector test compares the network input bit for bit).
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    typedef __bf16 bf16x2a __attribute__((ext_vector_type(2)));
    const bf16x2a v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float div255_exact(float p) {
    const float r = 1.0f / 255.0f;
    const float q = p * r;
    return __fmaf_rn(__fmaf_rn(-255.0f, q, p), r, q);
}
#endif

// cv::resize INTER_LINEAR on 8-bit data (the letterbox of AutoShape): source taps s0 / s1 and their 11-bit fixed-point
// coefficients c0 / c1 for destination index d; restated in oracle/imageops.py.  Host and device evaluate the same IEEE operations
// (no contraction: the reference rounds the multiply and the subtraction separately), so a launcher can plan with it what a kernel
// will fetch.  Used by letterbox_kernel (aux_kernels.hip) and by front_fused_kernel's resize mode.
__host__ __device__ inline void lin_coef(int d, int src, double scale, int& s0, int& s1, int& c0, int& c1, bool horizontal) {
#pragma clang fp contract(off)
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (horizontal) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src - 1) { s = src - 1; f = 0.f; }
        s0 = s; s1 = s + 1 < src - 1 ? s + 1 : src - 1;
    } else {
        s0 = s < 0 ? 0 : (s > src - 1 ? src - 1 : s);
        s1 = s + 1 < 0 ? 0 : (s + 1 > src - 1 ? src - 1 : s + 1);
    }
    const float w0 = (1.f - f) * 2048.f, w1 = f * 2048.f;
#if defined(__HIP_DEVICE_COMPILE__)
    int r0 = __float2int_rn(w0), r1 = __float2int_rn(w1);
#else
    int r0 = (int)nearbyintf(w0), r1 = (int)nearbyintf(w1);           // round to nearest even, like v_cvt_i32_f32's default mode
#endif
    c0 = r0 < -32768 ? -32768 : (r0 > 32767 ? 32767 : r0);
    c1 = r1 < -32768 ? -32768 : (r1 > 32767 ? 32767 : r1);
}

// ---- convolution as implicit GEMM (conv_igemm.hip) ----------------------------------------------
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };
enum ResMode : int { RES_NONE = 0, RES_AFTER_ACT = 1, RES_BEFORE_ACT = 2 };

struct ConvP {
    const void* in;      // NHWC activations, element type = prec
    const void* w;       // packed weights [Cout_pad][Kp], K order (r, s, c)
    const float* bias;   // [Cout_pad] f32
    const void* res;     // optional residual, NHWC, element type = prec
    void* out;           // NHWC, element type = prec (or f32 when out_f32)
    int B, H, W, Cin;    // logical input (Cin = channels consumed per tap; multiple of 8 (bf16) / 4 (f32))
    int in_cs, in_co;    // channel stride / offset of the input buffer (elements)
    int Ho, Wo, Cout;
    int out_cs, out_co;
    int res_cs, res_co;
    int kh, kw, sh, sw, ph, pw;
    int K, Kp;           // kh*kw*Cin and the padded row length of the packed weights (multiple of conv_k_tile)
    int Kw;              // set by launch_conv: weight row stride (= caller's Kp); Kp then becomes the K-loop extent
    int act, res_mode, out_f32, prec;
    // fp8 path (PREC_FP8): per-output-channel scale [Cout_pad] = weight scale x activation scale, applied to the accumulator; activation
    // scale of the fp8 tensors (real = stored x act_scale); out_bf16 = store the dequantised result as bf16 (Detect heads)
    const float* scale;
    float act_scale, inv_act_scale;
    int out_bf16;
    int M;               // B*Ho*Wo
    int cfg;             // tile configuration index (conv_igemm.hip kCfg), -1 = heuristic
    hipEvent_t ev_start, ev_stop;   // optional (in-flight profiling): receive the kernel's own start / stop timestamps (hipExtLaunchKernel)
    int ntiles;          // set by the launcher: output tiles of the chosen configuration (the grid may be smaller: persistent)
    // optional second destination: output channels >= split go to out2 (two 1x1 convs over the same input fused into
    // one launch, e.g. C3.cv1 + C3.cv2); split is a multiple of 4, 0 = single destination
    void* out2;
    int split, out2_cs, out2_co;
    const int* m_dev;    // optional: the number of valid output pixels lives on the DEVICE (a compacted batch whose size the host does not
                         // know, e.g. the sparse Detect head): the kernel processes min(M, *m_dev) pixels; M bounds the launch.  Only the
                         // implicit-GEMM family (conv_igemm_kernel) honours it
    // optional (bf16 pointwise convs, conv_igemm_kernel<..., UP>): input channels [0, up_C) are the nearest-neighbour 2 x upsampling of this
    // [B][H/2][W/2] tensor (channel stride up_cs, offset up_co) instead of channels [in_co, in_co + up_C) of `in` (Upsample + Concat folded in)
    const void* in_up;
    int up_C, up_cs, up_co;
    // split-K (conv_igemm_kernel<..., SK>, set by its launcher): K splits per output tile, fp32 partial sums [tile][split][...], one ticket per tile
    int ksplit;
    float* sk_ws;
    int* sk_tickets;
    long long* dbg;      // diagnostics only (VC_CONV_DBG): per-workgroup phase timestamps [tiles][8], 100 MHz clock; null in production
    int s2_th, s2_tw;    // set by the launcher of conv3x3s2_halo_kernel: its output tile rectangle (rows x columns)
    int slots;           // tests only: > 0 forces the persistent grid to this many workgroups (long tile walks); set by vc_conv2d_host from
                         // VC_CONV_SLOTS -- the launchers themselves never read the environment
    int ablate;          // diagnostics only (VC_CONV_ABLATE, timing experiments with wrong results): 1 = no staging DMA after the first tiles,
                         // 2 = no output stores, 3 = both, 6 = return at once (launch floor of the grid); per-phase times come from dbg
};

// launch, optionally with the dispatch's own start/stop timestamps written to p.ev_start / p.ev_stop (what rocprofv3 reports as the
// kernel duration; a hipEventRecord pair around a launch also counts the time the dispatch waits behind other queues)
template <class K, class... A>
static inline void launch_timed(const ConvP& p, K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t s, A... args) {
    if (p.ev_start) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, p.ev_start, p.ev_stop, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
}

int launch_conv(const ConvP& p, hipStream_t s);
int launch_conv_cfg(const ConvP& p, int cfg, hipStream_t s);     // no argument checks: for the autotuner
int launch_halo_v2_cfg(const ConvP& p, int cfg, hipStream_t s);  // conv_halo_v2.hip: tile configuration 55
bool s2halo_pw_applicable(const ConvP& p, const ConvP& q);      // conv3x3s2_halo_kernel<..., F2>: a 3x3 / s2 conv and the pointwise conv that alone reads it, one launch
int launch_s2halo_pw(ConvP p, ConvP q, hipStream_t s);
int conv_num_cfgs();
bool conv_stream_cfg(int cfg);                               // conv1x1_stream_kernel variants: offered to the autotuner only under VC_CONV_STREAM=1
bool stem_direct_applicable(const ConvP& p);                    // stem_direct.hip: YOLO 6x6/s2 stem in bf16
struct View;
int launch_stem_direct(const ConvP& p, hipStream_t s, const View* q8 = nullptr, float q_inv_scale = 1.0f);   // q8: also write the e4m3 copy of the output
struct LetterboxGeom;
// the same with the letterbox folded into the patch fetch: reads the u8 frames directly (no-resize geometry, even source width)
bool stem_u8_applicable(const ConvP& p, const LetterboxGeom& g);
int launch_stem_direct_u8(const ConvP& p, const uint8_t* frames, const LetterboxGeom& g, hipStream_t s, const View* q8 = nullptr, float q_inv_scale = 1.0f);
// front_fused.hip: YOLO layers 0 + 1 (stem + 3x3 / s2) in one kernel, the stem's output never leaves the CU
bool c3_fused_applicable(const ConvP& p12, const ConvP& pm1, const ConvP& pm2, const ConvP& p3);     // c3_fused.hip: the first C3 block (64 -> 64, n = 1) in one kernel
int launch_c3_fused(const ConvP& p12, const ConvP& pm1, const ConvP& pm2, const ConvP& p3, hipStream_t s);
bool bneck_fused_applicable(const ConvP& pm1, const ConvP& pm2);                                     // bneck_fused.hip: a 64-channel Bottleneck (1x1 + 3x3 [+ shortcut]) in one kernel
int launch_bneck_fused(const ConvP& pm1, const ConvP& pm2, hipStream_t s);
bool bneck_cv3_fused_applicable(const ConvP& pm1, const ConvP& pm2, const ConvP& p3);                // the same + the C3's cv3 (1x1 over [m | y2], 128 -> 128) on the tile
int launch_bneck_cv3_fused(const ConvP& pm1, const ConvP& pm2, const ConvP& p3, hipStream_t s);
bool front_fused_applicable(const ConvP& p0, const ConvP& p1);
int launch_front_fused(const ConvP& p0, const ConvP& p1, const uint8_t* frames_u8 /* nullable */, const LetterboxGeom& g, hipStream_t s);
bool front_fused_resize_ok(const LetterboxGeom& g);            // u8 frames that need the letterbox RESIZE: does the tile's source footprint fit the kernel's staging area?
bool reid_block_fused_applicable(const ConvP& p1, const ConvP& p2);          // reid_block_fused.hip: a 64-channel BasicBlock (two 3x3 + residual) on a 25 x 25 map in one kernel
int launch_reid_block_fused(const ConvP& p1, const ConvP& p2, hipStream_t s);
bool reid_stem_applicable(const ConvP& p, int out_cs, int out_co);          // reid_stem.hip: conv1 + ReLU + MaxPool fused (bf16)
int launch_reid_stem_pool(const ConvP& p, void* pooled, hipStream_t s);
int conv_k_tile(int prec);    // K elements per tile (weights are padded to a multiple of it)
double conv_flops(const ConvP& p);

}  // namespace vc
