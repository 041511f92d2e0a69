// Batched DeepSORT numerics on the device: Kalman predict / initiate / update, Mahalanobis gating folded into the
// appearance (gallery cosine) cost, and the IoU cost.  fp64 where the reference is fp64, fp32 where it is fp32.
//
// Reference (paths relative to /root/reference/networks/deepsort/sort/):
//   kalman_filter.py:55-85   initiate            -> kalman_initiate_kernel
//   kalman_filter.py:87-121  predict             -> kalman_predict_kernel   (F P F^T is exact: F is 0/1)
//   kalman_filter.py:123-152 project             -> project4()
//   kalman_filter.py:154-186 update              -> kalman_update_kernel    (4x4 Cholesky of S, K = P H^T S^-1)
//   kalman_filter.py:188-229 gating_distance     -> maha4()
//   nn_matching.py:31-54,78-96,160-177 distance  -> appearance_cost_kernel  (re-normalise, 1 - a.b, min over samples)
//   linear_assignment.py:148-192 gate_cost_matrix-> appearance_cost_kernel  (chi2inv95[4] = 9.4877 -> 1e5)
//   iou_matching.py:7-81     iou / iou_cost      -> iou_cost_kernel
//   track.py:82-96           to_tlwh             -> mean_to_tlwh()
// One wavefront owns one (track, detection) dot-product stream; reductions are wave shuffles, no atomics.
#include "kernels.h"

#pragma clang fp contract(off)

namespace vc {

#define VC_W_POS (1.0 / 20)
#define VC_W_VEL (1.0 / 160)
#define VC_CHI2_95_4 9.4877
#define VC_GATED 1e5

__device__ __forceinline__ void kalman_initiate_dev(double* m, double* P, const double* z) {
    const double h = z[3];
    for (int k = 0; k < 4; ++k) { m[k] = z[k]; m[4 + k] = 0.0; }
    const double sp = (2 * VC_W_POS) * h, sv = (10 * VC_W_VEL) * h;
    const double sd[8] = {sp, sp, 1e-2, sp, sv, sv, 1e-5, sv};
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) P[r * 8 + c] = r == c ? sd[r] * sd[r] : 0.0;
}

__global__ __launch_bounds__(64) void kalman_initiate_kernel(TrackPool tp, const int* slots, const double* xyah, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kalman_initiate_dev(tp.mean + (size_t)slots[i] * 8, tp.cov + (size_t)slots[i] * 64, xyah + (size_t)i * 4);
}

__device__ __forceinline__ void kalman_predict_dev(double* m, double* P) {
    const double h = m[3];
    const double sp = VC_W_POS * h, sv = VC_W_VEL * h;
    const double sd[8] = {sp, sp, 1e-2, sp, sv, sv, 1e-5, sv};
    double T[64];
    // T = P F^T : column j < 4 gains column j+4
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) T[r * 8 + c] = c < 4 ? P[r * 8 + c] + P[r * 8 + c + 4] : P[r * 8 + c];
    // P' = F T + Q : row r < 4 gains row r+4
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            double v = r < 4 ? T[r * 8 + c] + T[(r + 4) * 8 + c] : T[r * 8 + c];
            if (r == c) v += sd[r] * sd[r];
            P[r * 8 + c] = v;
        }
    for (int k = 0; k < 4; ++k) m[k] = m[k] + m[k + 4];
}

__global__ __launch_bounds__(64) void kalman_predict_kernel(TrackPool tp, const int* slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kalman_predict_dev(tp.mean + (size_t)slots[i] * 8, tp.cov + (size_t)slots[i] * 64);
}

// S = H P H^T + R (4x4), projected mean = mean[:4]
__device__ __forceinline__ void project4(const double* m, const double* P, double S[16]) {
    const double sp = VC_W_POS * m[3];
    const double sd[4] = {sp, sp, 1e-1, sp};
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) S[r * 4 + c] = P[r * 8 + c] + (r == c ? sd[r] * sd[r] : 0.0);
}

__device__ __forceinline__ void chol4(const double S[16], double L[16]) {
    for (int i = 0; i < 16; ++i) L[i] = 0.0;
    for (int j = 0; j < 4; ++j) {
        double d = S[j * 4 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 4 + k] * L[j * 4 + k];
        d = sqrt(d);
        L[j * 4 + j] = d;
        for (int i = j + 1; i < 4; ++i) {
            double v = S[i * 4 + j];
            for (int k = 0; k < j; ++k) v -= L[i * 4 + k] * L[j * 4 + k];
            L[i * 4 + j] = v / d;
        }
    }
}

__device__ __forceinline__ void kalman_update_dev(double* m, double* P, const double* z) {
    double S[16], L[16], K[32];
    project4(m, P, S);
    chol4(S, L);
    // K^T = S^-1 (P H^T)^T : for each state row r solve S k = P[r, 0:4]
    for (int r = 0; r < 8; ++r) {
        double y[4];
        for (int a = 0; a < 4; ++a) {            // L y = b
            double v = P[r * 8 + a];
            for (int k = 0; k < a; ++k) v -= L[a * 4 + k] * y[k];
            y[a] = v / L[a * 4 + a];
        }
        for (int a = 3; a >= 0; --a) {           // L^T x = y
            double v = y[a];
            for (int k = a + 1; k < 4; ++k) v -= L[k * 4 + a] * K[r * 4 + k];
            K[r * 4 + a] = v / L[a * 4 + a];
        }
    }
    double innov[4];
    for (int a = 0; a < 4; ++a) innov[a] = z[a] - m[a];
    double nm[8];
    for (int r = 0; r < 8; ++r) {
        double v = 0.0;
        for (int a = 0; a < 4; ++a) v += innov[a] * K[r * 4 + a];
        nm[r] = m[r] + v;
    }
    // P' = P - K (S K^T)
    double SKt[32];                               // 4 x 8
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 8; ++c) {
            double v = 0.0;
            for (int k = 0; k < 4; ++k) v += S[a * 4 + k] * K[c * 4 + k];
            SKt[a * 8 + c] = v;
        }
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            double v = 0.0;
            for (int a = 0; a < 4; ++a) v += K[r * 4 + a] * SKt[a * 8 + c];
            P[r * 8 + c] = P[r * 8 + c] - v;
        }
    for (int r = 0; r < 8; ++r) m[r] = nm[r];
}

// ---- one-wavefront forms of initiate / predict / update (lane = covariance element (r, c) = (lane >> 3, lane & 7)) ---
// Every output element is produced by exactly the operation sequence of the single-thread forms above (same operands,
// same order, no cross-lane reductions), so the results are bit-identical; what changes is that the ~70 dependent fp64
// divisions / square roots of an update are spread over the lanes.  Call with the 64 threads of ONE wave; kbuf is 32
// doubles of LDS.
__device__ __forceinline__ void kalman_initiate_wave(double* m, double* P, const double* z, int lane) {
    const int r = lane >> 3, c = lane & 7;
    const double h = z[3];
    const double sp = (2 * VC_W_POS) * h, sv = (10 * VC_W_VEL) * h;
    const double sd = r == 2 ? 1e-2 : r == 6 ? 1e-5 : r < 4 ? sp : sv;
    P[lane] = r == c ? sd * sd : 0.0;
    if (lane < 8) m[lane] = lane < 4 ? z[lane] : 0.0;
}

__device__ __forceinline__ void kalman_predict_wave(double* m, double* P, int lane) {
    const int r = lane >> 3, c = lane & 7;
    const double h = m[3];
    const double sp = VC_W_POS * h, sv = VC_W_VEL * h;
    const double sd = r == 2 ? 1e-2 : r == 6 ? 1e-5 : r < 4 ? sp : sv;
    // T = P F^T (column j < 4 gains column j+4), P' = F T + Q (row r < 4 gains row r+4)
    const double t0 = c < 4 ? P[r * 8 + c] + P[r * 8 + c + 4] : P[r * 8 + c];
    const int r4 = (r + 4) & 7;
    const double t1 = c < 4 ? P[r4 * 8 + c] + P[r4 * 8 + c + 4] : P[r4 * 8 + c];
    double v = r < 4 ? t0 + t1 : t0;
    if (r == c) v += sd * sd;
    const double mk = lane < 4 ? m[lane] + m[lane + 4] : 0.0;
    // every load above feeds a value stored below, so all lanes' loads have returned before the first store issues
    P[lane] = v;
    if (lane < 4) m[lane] = mk;
}

__device__ __forceinline__ void kalman_update_wave(double* m, double* P, const double* z, int lane, volatile double* kbuf) {
    const int r = lane >> 3, c = lane & 7;
    double S[16], L[16];
    project4(m, P, S);
    chol4(S, L);
    const double prc = P[lane];
    double nm = 0.0;
    if (lane < 8) {                               // K^T = S^-1 (P H^T)^T : state row `lane`
        double y[4], k[4];
        for (int a = 0; a < 4; ++a) {            // L y = b
            double v = P[lane * 8 + a];
            for (int q = 0; q < a; ++q) v -= L[a * 4 + q] * y[q];
            y[a] = v / L[a * 4 + a];
        }
        for (int a = 3; a >= 0; --a) {           // L^T x = y
            double v = y[a];
            for (int q = a + 1; q < 4; ++q) v -= L[q * 4 + a] * k[q];
            k[a] = v / L[a * 4 + a];
        }
        double v = 0.0;
        for (int a = 0; a < 4; ++a) { v += (z[a] - m[a]) * k[a]; kbuf[lane * 4 + a] = k[a]; }
        nm = m[lane] + v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double Kr[4], Kc[4];
    for (int a = 0; a < 4; ++a) { Kr[a] = kbuf[r * 4 + a]; Kc[a] = kbuf[c * 4 + a]; }
    double v = 0.0;                               // P' = P - K (S K^T)
    for (int a = 0; a < 4; ++a) {
        double skt = 0.0;
        for (int q = 0; q < 4; ++q) skt += S[a * 4 + q] * Kc[q];
        v += Kr[a] * skt;
    }
    P[lane] = prc - v;
    if (lane < 8) m[lane] = nm;
}

__global__ __launch_bounds__(64) void kalman_update_kernel(TrackPool tp, const int* slots, const double* xyah, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kalman_update_dev(tp.mean + (size_t)slots[i] * 8, tp.cov + (size_t)slots[i] * 64, xyah + (size_t)i * 4);
}

__device__ __forceinline__ double maha4(const double* m, const double L[16], const double* z) {
    double y[4], acc = 0.0;
    for (int a = 0; a < 4; ++a) {
        double v = z[a] - m[a];
        for (int k = 0; k < a; ++k) v -= L[a * 4 + k] * y[k];
        y[a] = v / L[a * 4 + a];
        acc += y[a] * y[a];
    }
    return acc;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// store to host-mapped pinned memory: system-scope write-through, complete once the wave's vmcnt drains
__device__ __forceinline__ void host_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Descriptor loads (pinned host memory, or device memory in the KAT entry points).  Plain vector loads: a kernel lives for
// one step, so nothing it reads from pinned memory can be stale, and neighbouring lanes coalesce into a few PCIe reads
// (per-lane system-scope atomic loads do not coalesce: 16k read transactions per step, +10 us).
__device__ __forceinline__ unsigned long long sys_load_u64(const void* p) { return *(const volatile unsigned long long*)p; }
__device__ __forceinline__ unsigned sys_load_u32(const void* p) { return *(const volatile unsigned*)p; }
__device__ __forceinline__ double sys_load_f64(const double* p) { return *(const volatile double*)p; }

// LDS scratch of one tracker workgroup
struct TrackShared {
    float smax[4][16];
    float red[2];
    int featrow[16];                 // feature rows / xyah of the 16 detections being scored
    double xyah[16][4];
    unsigned long long rec[12];      // the chain record being executed
    double kbuf[32];                 // Kalman gain rows (kalman_update_wave)
};

// One workgroup (4 waves) per job = one confirmed track against a contiguous range of detections.
// cost[d] = min_s (1 - <g_s, f_d/|f_d|>) with the gallery rows g_s stored already normalised (store_gallery_row):
// a (S x 512) x (512 x D) product on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, an exact fmaf chain), one 16-sample
// tile per wave against 16 detections at a time; |f_d|^2 falls out of the same operand loads.  The Mahalanobis gate of
// linear_assignment.py:148-192 is folded in (lane d of wave 0).
__device__ __forceinline__ void appearance_row_dev(const TrackPool& tp, const CostJob& jb, const float* feat, const int* det_feat_row,
                                                   const double* det_xyah, double* out, TrackShared& sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int S = jb.gal_count, D = jb.det_n;
    const double* m = tp.mean + (size_t)jb.slot * 8;
    const float* gal = tp.gallery + (size_t)jb.slot * tp.budget_cap * VC_FEAT_DIM;
    for (int d0 = 0; d0 < D; d0 += 16) {
        // the chunk's descriptors -> LDS (one round trip to pinned memory for all 16 detections)
        if (threadIdx.x < 16) {
            const int d = jb.det_off + min(d0 + (int)threadIdx.x, D - 1);
            sh.featrow[threadIdx.x] = (int)sys_load_u32(det_feat_row + d);
        } else if (threadIdx.x < 80) {
            const int t = threadIdx.x - 16, d = jb.det_off + min(d0 + (t >> 2), D - 1);
            sh.xyah[t >> 2][t & 3] = sys_load_f64(det_xyah + (size_t)d * 4 + (t & 3));
        }
        __syncthreads();
        const float* fptr = feat + (size_t)sh.featrow[col] * VC_FEAT_DIM + kq * 4;
        float best = -INFINITY, ss = 0.f;
        for (int st = wave; st * 16 < S; st += 4) {
            const float* gptr = gal + (size_t)min(st * 16 + col, S - 1) * VC_FEAT_DIM + kq * 4;
            // four independent accumulators (k mod 4 chunks): a 32-deep MFMA dependency chain instead of 128
            f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            float s2 = 0.f;
#pragma unroll 8
            for (int k0 = 0; k0 < VC_FEAT_DIM; k0 += 16) {
                const float4 a = *(const float4*)(gptr + k0), b = *(const float4*)(fptr + k0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc3, 0, 0, 0);
                s2 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
            }
            const f32x4_t acc = (acc0 + acc1) + (acc2 + acc3);
            ss = s2;
#pragma unroll
            for (int r = 0; r < 4; ++r)                       // acc[r] = <g_{st*16 + kq*4 + r}, f_{d0 + col}>
                if (st * 16 + kq * 4 + r < S) best = fmaxf(best, acc[r]);
        }
        best = fmaxf(best, __shfl_xor(best, 16));
        best = fmaxf(best, __shfl_xor(best, 32));
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (lane < 16) sh.smax[wave][lane] = best;
        __syncthreads();
        if (wave == 0 && lane < 16 && d0 + lane < D) {
            const float bmax = fmaxf(fmaxf(sh.smax[0][lane], sh.smax[1][lane]), fmaxf(sh.smax[2][lane], sh.smax[3][lane]));
            const float cosv = bmax * (1.0f / sqrtf(ss));
            double Sg[16], L[16];                          // gate: only these 16 lanes need the 4x4 Cholesky factor (fp64, ~100 dependent ops)
            project4(m, tp.cov + (size_t)jb.slot * 64, Sg);
            chol4(Sg, L);
            const double g2 = maha4(m, L, sh.xyah[lane]);
            host_store(out + jb.out_off + d0 + lane, g2 > VC_CHI2_95_4 ? VC_GATED : (double)(1.0f - cosv));
        }
        __syncthreads();
    }
}

// gallery row = feature / |feature|_2 (nn_matching.py:45-47 normalises at distance time; the result is the same vector)
__device__ __forceinline__ void store_gallery_row(float* dst, const float* src, int t /* threads 0..127 work */, float* red /* [2] */) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < 128) v = ((const float4*)src)[t];
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (t < 128 && (t & 63) == 0) red[t >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1]);
    if (t < 128) ((float4*)dst)[t] = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
}

__global__ __launch_bounds__(256) void appearance_cost_kernel(TrackPool tp, const CostJob* jobs, const float* feat, const int* det_feat_row,
                                                              const double* det_xyah, double* out) {
    __shared__ TrackShared sh;
    appearance_row_dev(tp, jobs[blockIdx.x], feat, det_feat_row, det_xyah, out, sh);
}

__device__ __forceinline__ double iou_tlwh(const double* b, const double* c);
__device__ __forceinline__ void mean_to_tlwh(const double* m, double t[4]) {
    t[2] = m[2] * m[3];
    t[3] = m[3];
    t[0] = m[0] - t[2] / 2;
    t[1] = m[1] - t[3] / 2;
}

__global__ __launch_bounds__(64) void iou_cost_kernel(TrackPool tp, const CostJob* jobs, const double* __restrict__ det_tlwh,
                                                      double* __restrict__ out) {
    const CostJob jb = jobs[blockIdx.x];
    double b[4];
    mean_to_tlwh(tp.mean + (size_t)jb.slot * 8, b);
    for (int d = threadIdx.x; d < jb.det_n; d += blockDim.x) {
        if (jb.tsu > 1) { out[jb.out_off + d] = VC_GATED; continue; }
        out[jb.out_off + d] = 1.0 - iou_tlwh(b, det_tlwh + (size_t)(jb.det_off + d) * 4);
    }
}

__global__ __launch_bounds__(128) void gallery_write_kernel(TrackPool tp, const int* __restrict__ sps, const float* __restrict__ feat) {
    __shared__ float red[2];
    const int* e = sps + (size_t)blockIdx.x * 3;
    store_gallery_row(tp.gallery + ((size_t)e[0] * tp.budget_cap + e[1]) * VC_FEAT_DIM, feat + (size_t)e[2] * VC_FEAT_DIM, threadIdx.x, red);
}

// ---- per-frame tracker work (tracker.hip: track_launch) -------------------------------------------------------------
// A "chain" is everything one track slot needs between two host matching steps: first the slot's pending operation of
// frame f -- Kalman update (kind 1) / initiate (kind 2) / nothing (kind 0), the gallery ring write of the matched feature
// and the posterior mean of output-eligible tracks to the host -- then the slot's cost job of frame f+1: Kalman predict in
// place, (confirmed tracks) the appearance + gate row and (IoU candidates) the IoU row against its tracker's detections.
// A slot belongs to exactly one chain per step, so chains never wait for each other.  Records, detections and results
// live in host-mapped pinned memory: no copy operations.
__device__ __forceinline__ void run_chain(const TrackPool& tp, const TrackChainRec* rec_ptr, const float* feat_ops, const float* feat_jobs,
                                          double* mean_out, const int* det_feat_row, const double* det_xyah, const double* det_tlwh,
                                          double* out, TrackShared& sh) {
    if (threadIdx.x < 12) sh.rec[threadIdx.x] = sys_load_u64((const unsigned long long*)rec_ptr + threadIdx.x);
    __syncthreads();
    const TrackChainRec& rec = *(const TrackChainRec*)sh.rec;
    const TrackOpB op = rec.op;
    const TrackJobA jb = rec.job;
    __syncthreads();                       // sh.rec is free again
    if (op.kind >= 0) {
        double* m = tp.mean + (size_t)op.slot * 8;
        if (threadIdx.x < 64) {            // wave 0
            if (op.kind == 1) kalman_update_wave(m, tp.cov + (size_t)op.slot * 64, op.z, threadIdx.x, sh.kbuf);
            else if (op.kind == 2) kalman_initiate_wave(m, tp.cov + (size_t)op.slot * 64, op.z, threadIdx.x);
        }
        if (op.feat_row >= 0)              // block-uniform
            store_gallery_row(tp.gallery + ((size_t)op.slot * tp.budget_cap + op.gal_pos) * VC_FEAT_DIM,
                              feat_ops + (size_t)op.feat_row * VC_FEAT_DIM, threadIdx.x, sh.red);
        __syncthreads();
        if (op.out_row >= 0 && threadIdx.x < 8) host_store(mean_out + (size_t)op.out_row * 8 + threadIdx.x, m[threadIdx.x]);
        __syncthreads();
    }
    if (jb.slot >= 0) {
        if (threadIdx.x < 64) kalman_predict_wave(tp.mean + (size_t)jb.slot * 8, tp.cov + (size_t)jb.slot * 64, threadIdx.x);
        __syncthreads();
        if (jb.app_off >= 0) {
            const CostJob cj{jb.slot, jb.gal_count, jb.det_off, jb.det_n, jb.app_off, jb.tsu};
            appearance_row_dev(tp, cj, feat_jobs, det_feat_row, det_xyah, out, sh);
        }
        if (jb.iou_off >= 0) {
            double b[4];
            mean_to_tlwh(tp.mean + (size_t)jb.slot * 8, b);
            for (int d = threadIdx.x; d < jb.det_n; d += blockDim.x) {
                double c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = sys_load_f64(det_tlwh + (size_t)(jb.det_off + d) * 4 + k);
                host_store(out + jb.iou_off + d, jb.tsu > 1 ? VC_GATED : 1.0 - iou_tlwh(b, c));
            }
        }
    }
}

// One launch per step (the blocking entry points and the profiling mode): one workgroup per chain; the last workgroup to
// finish publishes `seq` to a pinned word, which the host polls instead of paying a stream synchronisation.
__global__ __launch_bounds__(256) void track_step_kernel(TrackPool tp, const TrackChainRec* recs, const float* feat_ops,
                                                         const float* feat_jobs, double* mean_out, const int* det_feat_row,
                                                         const double* det_xyah, const double* det_tlwh, double* out,
                                                         unsigned* counter, unsigned* done_flag, unsigned seq) {
    __shared__ TrackShared sh;
    run_chain(tp, recs + blockIdx.x, feat_ops, feat_jobs, mean_out, det_feat_row, det_xyah, det_tlwh, out, sh);
    // completion: results went to pinned memory as system-scope write-through stores (host_store), so a workgroup only
    // has to wait for its own stores to be acknowledged
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // relaxed on purpose: an agent/system RELEASE here is a write-back of the XCD's whole L2 (full of the detector's
        // dirty output lines) per workgroup; the results are already write-through and acknowledged (vmcnt(0) above)
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int launch_track_step(const TrackPool& tp, const TrackChainRec* recs, int nchains, const float* feat_ops, const float* feat_jobs,
                      double* mean_out, const int* det_feat_row, const double* det_xyah, const double* det_tlwh, double* out,
                      unsigned* counter, unsigned* done_flag, unsigned seq, hipStream_t s) {
    if (nchains <= 0) return VC_OK;
    hipLaunchKernelGGL(track_step_kernel, dim3(nchains), dim3(256), 0, s, tp, recs, feat_ops, feat_jobs, mean_out, det_feat_row,
                       det_xyah, det_tlwh, out, counter, done_flag, seq);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

__device__ __forceinline__ double iou_tlwh(const double* b, const double* c) {
    const double tlx = fmax(b[0], c[0]), tly = fmax(b[1], c[1]);
    const double brx = fmin(b[0] + b[2], c[0] + c[2]), bry = fmin(b[1] + b[3], c[1] + c[3]);
    const double w = fmax(0.0, brx - tlx), h = fmax(0.0, bry - tly);
    const double inter = w * h;
    return inter / (b[2] * b[3] + c[2] * c[3] - inter);
}

__global__ __launch_bounds__(64) void iou_boxes_kernel(const double* a, int t, const double* b, int d, double* out) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += blockDim.x) out[(size_t)i * d + j] = iou_tlwh(a + (size_t)i * 4, b + (size_t)j * 4);
}

__global__ __launch_bounds__(64) void gating_values_kernel(TrackPool tp, int slot, const double* z, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* m = tp.mean + (size_t)slot * 8;
    double S[16], L[16];
    project4(m, tp.cov + (size_t)slot * 64, S);
    chol4(S, L);
    out[i] = maha4(m, L, z + (size_t)i * 4);
}

int launch_gating_values(const TrackPool& tp, int slot, const double* z, int n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(gating_values_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tp, slot, z, n, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_iou_boxes(const double* a, int t, const double* b, int d, double* out, hipStream_t s) {
    hipLaunchKernelGGL(iou_boxes_kernel, dim3(t), dim3(64), 0, s, a, t, b, d, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_kalman_predict(TrackPool& tp, const int* slots, int n, hipStream_t s) {
    if (n <= 0) return VC_OK;
    hipLaunchKernelGGL(kalman_predict_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tp, slots, n);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_kalman_initiate(TrackPool& tp, const int* slots, const double* xyah, int n, hipStream_t s) {
    if (n <= 0) return VC_OK;
    hipLaunchKernelGGL(kalman_initiate_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tp, slots, xyah, n);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_kalman_update(TrackPool& tp, const int* slots, const double* xyah, int n, hipStream_t s) {
    if (n <= 0) return VC_OK;
    hipLaunchKernelGGL(kalman_update_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tp, slots, xyah, n);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_appearance_cost(const TrackPool& tp, const CostJob* jobs, int njobs, const float* feat, const int* det_feat_row,
                           const double* det_xyah, double* out, hipStream_t s) {
    if (njobs <= 0) return VC_OK;
    hipLaunchKernelGGL(appearance_cost_kernel, dim3(njobs), dim3(256), 0, s, tp, jobs, feat, det_feat_row, det_xyah, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_iou_cost(const TrackPool& tp, const CostJob* jobs, int njobs, const double* det_tlwh, double* out, hipStream_t s) {
    if (njobs <= 0) return VC_OK;
    hipLaunchKernelGGL(iou_cost_kernel, dim3(njobs), dim3(64), 0, s, tp, jobs, det_tlwh, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
__global__ __launch_bounds__(256) void gather_means_kernel(TrackPool tp, const int* __restrict__ slots, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * 8) out[i] = tp.mean[(size_t)slots[i >> 3] * 8 + (i & 7)];
}
int launch_gather_means(const TrackPool& tp, const int* slots, int n, double* out, hipStream_t s) {
    if (n <= 0) return VC_OK;
    hipLaunchKernelGGL(gather_means_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, s, tp, slots, n, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_gallery_write(TrackPool& tp, const int* slot_pos_src, int n, const float* feat, hipStream_t s) {
    if (n <= 0) return VC_OK;
    hipLaunchKernelGGL(gallery_write_kernel, dim3(n), dim3(128), 0, s, tp, slot_pos_src, feat);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
