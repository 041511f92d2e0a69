// DeepSORT on the device, one workgroup per (camera, class) tracker for a whole batch of frames: no host in the per-frame loop.
//
// track_batch_kernel: workgroup b owns tracker plans[b].tracker and walks its tasks (= the frames of the batch in which the
// class has detections, modules/track.py:50-59) in order.  Per task:
//   P0  all 4 waves   Track.predict counters + Kalman predict (one wave per track, one covariance element per lane),
//                     gated appearance rows of the confirmed tracks (fp32 matrix cores) and IoU rows of the IoU candidates
//   P1  wave 0        Tracker._match: cascade + IoU stage on exact rectangular assignments (track_core.h), slots for new tracks
//   P2  all 4 waves   Kalman update / initiate + gallery ring writes, one wave per matched / new track
//   P3  wave 0        Track FSM, list maintenance, rows [x1,y1,x2,y2,id,label] of the confirmed tracks (deep_sort.py:46-58)
// Tracker state (header, ordered slot list, per-slot records, fp64 Kalman pool, gallery rings) lives in device memory between
// batches; detections and tasks arrive in one host-to-device copy per batch, rows leave in one copy back.
//
// Reference (paths relative to /root/reference/networks/deepsort/sort/):
//   kalman_filter.py:55-229  initiate / predict / project / update / gating_distance   -> kalman_*_wave, project4/chol4/maha4
//   nn_matching.py:31-54,78-96,160-177 distance  -> appearance_row_dev (re-normalise, 1 - a.b, min over samples)
//   linear_assignment.py:148-192 gate_cost_matrix-> appearance_row_dev (chi2inv95[4] = 9.4877 -> 1e5)
//   iou_matching.py:7-81     iou / iou_cost      -> iou rows;  track.py:82-96 to_tlwh -> mean_to_tlwh
//   tracker.py / linear_assignment.py / track.py -> track_core.h
// The single-function entry points of the parity tests (vc_kalman_*_host, vc_cosine_cost_host, ...) launch kat_* kernels that
// call exactly the __device__ functions the batch kernel calls.
#include <algorithm>

#include "kernels.h"

#pragma clang fp contract(off)

#include "track_core.h"

namespace vc {

using namespace tc;

// ---- one-wavefront forms of initiate / predict / update (lane = covariance element (r, c) = (lane >> 3, lane & 7)) ---
// Every output element is produced by exactly the operation sequence of the single-thread forms of track_math.h (same operands,
// same order, no cross-lane reductions), so the results are bit-identical; what changes is that the ~70 dependent fp64
// divisions / square roots of an update are spread over the lanes.  Call with the 64 threads of ONE wave; kbuf is 32
// doubles of LDS private to that wave.
__device__ __forceinline__ void kalman_initiate_wave(double* m, double* P, const double* z, int lane) {
    const int r = lane >> 3, c = lane & 7;
    const double h = z[3];
    const double sp = (2 * VC_W_POS) * h, sv = (10 * VC_W_VEL) * h;
    const double sd = r == 2 ? 1e-2 : r == 6 ? 1e-5 : r < 4 ? sp : sv;
    P[lane] = r == c ? sd * sd : 0.0;
    if (lane < 8) m[lane] = lane < 4 ? z[lane] : 0.0;
}

// m2 / P2 (optional): a second copy of the predicted state (the step's LDS copy, read by the cost rows, the update and the rows)
__device__ __forceinline__ void kalman_predict_wave(double* m, double* P, int lane, double* m2 = nullptr, double* P2 = nullptr) {
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
    if (m2) {
        P2[lane] = v;
        if (lane < 8) m2[lane] = lane < 4 ? mk : m[lane];
    }
}

// ms / Ps (optional): where the prior is read from (the step's LDS copy of what m / P hold); the posterior goes to m / P and,
// the mean, to ms as well
__device__ __forceinline__ void kalman_update_wave(double* mg, double* Pg, const double* z, int lane, volatile double* kbuf, double* ms = nullptr,
                                                   const double* Ps = nullptr) {
    const int r = lane >> 3, c = lane & 7;
    const double* m = ms ? ms : mg;
    const double* P = Ps ? Ps : Pg;
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
    Pg[lane] = prc - v;
    if (lane < 8) { mg[lane] = nm; if (ms) ms[lane] = nm; }
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define VC_SMALL_D 12       // detections per step up to which the appearance rows run one wave per track (appearance_row_wave)
#define VC_TRACK_MAX_WAVES 8
#define VC_STEP_STATE 64         // tracks whose Kalman state a step keeps in LDS (larger steps read the pool in global memory)
#define VC_TRACK_DYN_LDS VC_TRACK_DYN_LDS_BYTES

// LDS scratch of one tracker workgroup
struct TrackShared {
    float smax[4][16];
    int featrow[16];                 // feature rows / xyah of the 16 detections being scored
    double xyah[16][4];
    double kbuf[VC_TRACK_MAX_WAVES][32];   // Kalman gain rows, one block per wave (kalman_update_wave)
    double small_c[256], small_t[256];   // small assignment problems stay in LDS (StepWork::small_c / small_t)
    double cost_small[2][512];           // the step's appearance / IoU rows when T x D <= 512 (else the workgroup's global scratch)
    int ctl[8];                      // P1 -> P2/P3 hand-over: n_match, n_un, n_new, newdets is w.left, error
    // the tracker's header and track list live here for the whole walk (global copies are refreshed when the walk ends): a step
    // then starts and finishes without a dependent global-memory round trip for either
    TrackerHdr hdr;
    int list[TC_HARD_CAP];
    // the step's copy of the Kalman state of up to VC_STEP_STATE tracks (list position t): written by the predict, read by the cost
    // rows and the update, posterior mean read by the row emission -- no global-memory round trip between the phases of a step
    double st_mean[VC_STEP_STATE][8];
    double st_cov[VC_STEP_STATE][64];
};

// One workgroup (4 waves) per job = one confirmed track against a contiguous range of detections.
// cost[d] = min_s (1 - <g_s, f_d/|f_d|>) with the gallery rows g_s stored already normalised (gallery_store_wave):
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
        if (threadIdx.x < 16) {
            const int d = jb.det_off + min(d0 + (int)threadIdx.x, D - 1);
            sh.featrow[threadIdx.x] = det_feat_row[d];
        } else if (threadIdx.x < 80) {
            const int t = threadIdx.x - 16, d = jb.det_off + min(d0 + (t >> 2), D - 1);
            sh.xyah[t >> 2][t & 3] = det_xyah[(size_t)d * 4 + (t & 3)];
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
            out[jb.out_off + d0 + lane] = g2 > VC_CHI2_95_4 ? VC_GATED : (double)(1.0f - cosv);
        }
        __syncthreads();
    }
}

// The same row for a SMALL number of detections, one wave per track: v_mfma_f32_4x4x1_16b_f32 = 16 independent 4x4 outer products
// per instruction, block b = samples 4b .. 4b+3 (A operand: lane l feeds sample l), columns = 4 detections (B operand: lane l feeds
// detection l % 4); measured layout (tools/ubench/mfma4x4_probe.hip): D[vgpr i] at lane l = A[lane 4*(l/4) + i] * B[lane l].
// 64 samples x 4 detections per instruction with nothing wasted, where the 16x16x4 form above spends 16 columns on 2-3 detections
// and needs all four waves for one track.  Accumulator c takes the elements k = c (mod 4) in increasing k, summed
// (acc0 + acc1) + (acc2 + acc3), and |f|^2 is built from the same four partial sums as above: bit-identical to appearance_row_dev.
__device__ __forceinline__ void appearance_row_wave(const TrackPool& tp, const CostJob& jb, const float* feat, const int* det_feat_row,
                                                    const double* det_xyah, double* out, int lane) {
    const int S = jb.gal_count, D = jb.det_n;
    const double* m = tp.mean + (size_t)jb.slot * 8;
    const float* gal = tp.gallery + (size_t)jb.slot * tp.budget_cap * VC_FEAT_DIM;
    double Lc[16];
    {   // the gate's Cholesky factor depends on the track only (fp64, ~100 dependent operations): once, not per detection group
        double Sg[16];
        project4(m, tp.cov + (size_t)jb.slot * 64, Sg);
        chol4(Sg, Lc);
    }
    // eight detections per pass over the gallery: lane l feeds detection d0 + l % 4 (group A) and d0 + 4 + l % 4 (group B)
    for (int d0 = 0; d0 < D; d0 += 8) {
        const bool two = d0 + 4 < D;                               // wave-uniform
        const int dA = jb.det_off + min(d0 + (lane & 3), D - 1), dB = jb.det_off + min(d0 + 4 + (lane & 3), D - 1);
        const float* fa = feat + (size_t)det_feat_row[dA] * VC_FEAT_DIM;
        const float* fb = feat + (size_t)det_feat_row[dB] * VC_FEAT_DIM;
        float bestA = -INFINITY, bestB = -INFINITY, ssA = 0.f, ssB = 0.f;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const float* gp = gal + (size_t)min(s0 + lane, S - 1) * VC_FEAT_DIM;
            f32x4_t A0 = {0.f, 0.f, 0.f, 0.f}, A1 = A0, A2 = A0, A3 = A0, B0 = A0, B1 = A0, B2 = A0, B3 = A0;
            float qa[4] = {0.f, 0.f, 0.f, 0.f}, qb[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < VC_FEAT_DIM / 4; c0 += 4) {
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const float4 a = *(const float4*)(gp + (c0 + kq) * 4);
                    const float4 b = *(const float4*)(fa + (c0 + kq) * 4);
                    A0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b.x, A0, 0, 0, 0);
                    A1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b.y, A1, 0, 0, 0);
                    A2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b.z, A2, 0, 0, 0);
                    A3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b.w, A3, 0, 0, 0);
                    qa[kq] += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
                    if (two) {
                        const float4 b2 = *(const float4*)(fb + (c0 + kq) * 4);
                        B0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b2.x, B0, 0, 0, 0);
                        B1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b2.y, B1, 0, 0, 0);
                        B2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b2.z, B2, 0, 0, 0);
                        B3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b2.w, B3, 0, 0, 0);
                        qb[kq] += b2.x * b2.x + b2.y * b2.y + b2.z * b2.z + b2.w * b2.w;
                    }
                }
            }
            const f32x4_t accA = (A0 + A1) + (A2 + A3), accB = (B0 + B1) + (B2 + B3);
            ssA = (qa[0] + qa[1]) + (qa[2] + qa[3]);
            ssB = (qb[0] + qb[1]) + (qb[2] + qb[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i)                           // acc[i] = <g_{s0 + 4*(lane/4) + i}, f_{d + lane%4}>
                if (s0 + (lane & ~3) + i < S) { bestA = fmaxf(bestA, accA[i]); bestB = fmaxf(bestB, accB[i]); }
        }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { bestA = fmaxf(bestA, __shfl_xor(bestA, o)); bestB = fmaxf(bestB, __shfl_xor(bestB, o)); }
        // lanes 0-3 finish group A, lanes 4-7 group B (lane & 3 = detection within the group, same value in every block)
        const int d = d0 + lane;                                   // lanes 0..7 -> detections d0 .. d0 + 7
        if (lane < 8 && d < D) {
            const float best = lane < 4 ? bestA : bestB, ss = lane < 4 ? ssA : ssB;
            const float cosv = best * (1.0f / sqrtf(ss));
            const double g2 = maha4(m, Lc, det_xyah + (size_t)(jb.det_off + d) * 4);
            out[jb.out_off + d] = g2 > VC_CHI2_95_4 ? VC_GATED : (double)(1.0f - cosv);
        }
    }
}

// gallery row = feature / |feature|_2 (nn_matching.py:45-47 normalises at distance time; the result is the same vector).
// One wave: lane l holds elements [4l, 4l+4) and [256 + 4l, 256 + 4l + 4).  Returns |feature|^2.
__device__ __forceinline__ float gallery_store_wave(float* dst, const float* src, int lane) {
    const float4 a = ((const float4*)src)[lane], b = ((const float4*)src)[64 + lane];
    float ss = (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float nrm = sqrtf(ss);
    ((float4*)dst)[lane] = make_float4(a.x / nrm, a.y / nrm, a.z / nrm, a.w / nrm);
    ((float4*)dst)[64 + lane] = make_float4(b.x / nrm, b.y / nrm, b.z / nrm, b.w / nrm);
    return ss;
}

// ---- appearance dots, hoisted out of the sequential loop (TrackDotPlan in kernels.h) ----------------------------------------------
// plan: one workgroup per tracker of the batch.  Numbers the samples the tracker holds now (track order, ring position) as table
// rows, records the row of every ring entry (gal_row) and the entry of every row (row_src), reserves the table in the arena.
__global__ __launch_bounds__(256) void track_plan_kernel(const TrackBatchArgs a) {
    __shared__ int s_cnt[TC_HARD_CAP];
    __shared__ int s_base[TC_HARD_CAP + 1];
    __shared__ TrackDotPlan s_dp;
    const TrackWgPlan plan = a.plans[blockIdx.x];
    const TrackerHdr* hdr = a.hdrs + plan.tracker;
    const int* list = a.lists + (size_t)plan.tracker * a.list_cap;
    const int T = min(hdr->n_tracks, (int)TC_HARD_CAP), SC = a.pool.budget_cap;
    for (int t = threadIdx.x; t < T; t += blockDim.x) s_cnt[t] = min(a.recs[list[t]].gal_count, SC);
    __syncthreads();
    if (threadIdx.x == 0) {                                  // T <= 512 and a batch has one plan pass: a serial prefix sum is fine
        int acc = 0;
        for (int t = 0; t < T; ++t) { s_base[t] = acc; acc += s_cnt[t]; }
        s_base[T] = acc;
        TrackDotPlan dp{};
        dp.n_old_rows = acc; dp.n_dets = plan.det_n;
        const long long rows = (long long)acc + plan.det_n, need = rows * plan.det_n;
        dp.det_tiles = (plan.det_n + 15) / 16;
        dp.tiles = (int)((rows + 15) / 16) * dp.det_tiles;
        dp.use_table = 0;
        if (plan.det_n > 0 && need < (1ll << 31)) {
            const unsigned long long off = __hip_atomic_fetch_add((unsigned long long*)a.dot_ctl, (unsigned long long)need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int roff = __hip_atomic_fetch_add(a.dot_ctl + 2, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((long long)off + need <= a.dot_arena_floats && roff + acc <= a.row_src_cap) {
                dp.use_table = 1; dp.table_off = (long long)off; dp.row_src_off = roff;
                dp.tile_begin = __hip_atomic_fetch_add(a.dot_ctl + 3, dp.tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        s_dp = dp;
        a.dot_plans[blockIdx.x] = dp;
    }
    __syncthreads();
    if (!s_dp.use_table) return;                             // no room (or nothing to score): the batch kernel computes the rows itself
    for (int t = 0; t < T; ++t) {
        const int slot = list[t], n = s_cnt[t], base = s_base[t];
        for (int pos = threadIdx.x; pos < n; pos += blockDim.x) {
            a.gal_row[(size_t)slot * SC + pos] = base + pos;
            a.row_src[s_dp.row_src_off + base + pos] = slot * SC + pos;
        }
    }
}

// norm: one wave per detection of the batch: the normalised feature a gallery write would store, and |feature|^2
__global__ __launch_bounds__(256) void track_norm_kernel(const TrackBatchArgs a) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= a.n_det_total) return;
    const float ss = gallery_store_wave(a.nfeat + (size_t)g * VC_FEAT_DIM, a.feat + (size_t)a.det_featrow[g] * VC_FEAT_DIM, lane);
    if (lane == 0) a.det_ss[g] = ss;
}

// dots: persistent waves pull 16 x 16 tiles (rows x detections) of all trackers' tables; one v_mfma_f32_16x16x4_f32 chain per
// tile exactly as in appearance_row_dev (four accumulators by k mod 4), so a table entry equals what that function computes.
__global__ __launch_bounds__(256) void track_dots_kernel(const TrackBatchArgs a, int n_wg) {
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    const int total = a.dot_ctl[3];
    for (;;) {
        int x = 0;
        if (lane == 0) x = __hip_atomic_fetch_add(a.dot_ctl + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = __shfl(x, 0);
        if (x >= total) return;
        int p = 0;
        for (; p < n_wg; ++p) {                                           // wave-uniform search (n_wg <= a few hundred, L2-resident)
            const TrackDotPlan& q = a.dot_plans[p];
            if (q.use_table && x >= q.tile_begin && x < q.tile_begin + q.tiles) break;
        }
        if (p == n_wg) continue;
        const TrackDotPlan dp = a.dot_plans[p];
        const TrackWgPlan plan = a.plans[p];
        const int local = x - dp.tile_begin, rt = local / dp.det_tiles, dt = local - rt * dp.det_tiles;
        const int rows = dp.n_old_rows + dp.n_dets, row0 = rt * 16, d0 = dt * 16;
        const int r = min(row0 + col, rows - 1), d = min(d0 + col, dp.n_dets - 1);
        const float* aptr = (r < dp.n_old_rows ? a.pool.gallery + (size_t)a.row_src[dp.row_src_off + r] * VC_FEAT_DIM
                                                : a.nfeat + (size_t)(plan.det_begin + r - dp.n_old_rows) * VC_FEAT_DIM) + kq * 4;
        const float* bptr = a.feat + (size_t)a.det_featrow[plan.det_begin + d] * VC_FEAT_DIM + kq * 4;
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
#pragma unroll 8
        for (int k0 = 0; k0 < VC_FEAT_DIM; k0 += 16) {
            const float4 av = *(const float4*)(aptr + k0), bv = *(const float4*)(bptr + k0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc3, 0, 0, 0);
        }
        const f32x4_t acc = (acc0 + acc1) + (acc2 + acc3);                // acc[i] = <row row0 + kq*4 + i, detection d0 + col>
        float* tab = a.dot_arena + dp.table_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = row0 + kq * 4 + i;
            if (rr < rows && d0 + col < dp.n_dets) tab[(size_t)rr * dp.n_dets + d0 + col] = acc[i];
        }
    }
}

// One wave per track: the gated appearance row from the precomputed table.  Lane = detection (coalesced reads of a table row),
// loop over the track's ring entries; cost = 1 - max_s <g_s, f_d> / |f_d|.
// adm_out (StepWork::adm of the track): 1 when some detection of the step is within max_dist of the track, else 0
__device__ __forceinline__ void appearance_row_table(const TrackBatchArgs& a, const TrackDotPlan& dp, const CostJob& jb, int det_local0,
                                                     double* out, int lane, const double* m, const double* Pm, double max_dist, unsigned char* adm_out) {
    const TrackPool& tp = a.pool;
    const int S = jb.gal_count, D = jb.det_n, SC = tp.budget_cap;
    const float* tab = a.dot_arena + dp.table_off;
    double Lc[16];
    {
        double Sg[16];
        project4(m, Pm, Sg);
        chol4(Sg, Lc);
    }
    // lanes = (sample group, detection): with D detections the wave splits into 64 / pow2(D) sample groups, each lane walks every
    // SL-th ring entry (independent loads, all in flight together) and the groups are merged with a few lane exchanges -- a step of
    // a light scene (D ~ 6, S ~ 60) is 8 table reads per lane behind one memory latency instead of 60 behind fifteen
    bool adm = false;                                    // (uniform)
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int nd = min(64, D - d0);
        int sh = 0;
        while ((1 << sh) < nd) ++sh;
        const int DL = 1 << sh, SL = 64 >> sh, dl = lane & (DL - 1), sg = lane >> sh;
        const bool ok = dl < nd;
        // The Mahalanobis gate first (linear_assignment.py:148-192): a track whose object has left the scene -- most tracks of a crowded
        // tracker between their last match and max_age -- gates every detection out, and its row is VC_GATED without the two dependent
        // memory round trips (ring -> table rows) below.  The values written are the same either way.
        const bool mine = ok && sg == 0;
        const int g = jb.det_off + d0 + dl;
        double g2 = 0.0;
        if (mine) g2 = maha4(m, Lc, a.det_xyah + (size_t)g * 4);
        if (!__any(mine && !(g2 > VC_CHI2_95_4))) {
            if (mine) out[jb.out_off + d0 + dl] = VC_GATED;
            continue;
        }
        const int* grow = a.gal_row + (size_t)jb.slot * SC;
        const float* col = tab + det_local0 + d0 + min(dl, nd - 1);
        float best = -INFINITY;
        // 16 ring entries at a time: their 16 table rows are read behind ONE memory latency (the row numbers first, all of them, then the
        // 16 independent table reads), instead of eight dependent rounds per 60-sample gallery -- at 50 detections x 90 tracks per step
        // (BASELINE.json configs[2]) the cost rows were 172 us of a 345 us step
        for (int s0 = sg; s0 < S; s0 += 16 * SL) {
            int r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int s = s0 + k * SL; r[k] = grow[s < S ? s : sg]; }
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = col[(size_t)r[k] * dp.n_dets];
#pragma unroll
            for (int k = 0; k < 16; ++k) best = fmaxf(best, v[k]);      // entries past S repeat entry sg: the maximum is unchanged
        }
        for (int o = DL; o < 64; o <<= 1) best = fmaxf(best, __shfl_xor(best, o));
        double val = VC_GATED;
        if (mine) {
            const float cosv = best * (1.0f / sqrtf(a.det_ss[g]));
            val = g2 > VC_CHI2_95_4 ? VC_GATED : (double)(1.0f - cosv);
            out[jb.out_off + d0 + dl] = val;
        }
        adm = adm || __any(mine && !(val > max_dist));
    }
    if (lane == 0) *adm_out = adm ? 1 : 0;
}

// ---- slot pool: during a kernel slots are only TAKEN from the free stack (filled before the launch) and freed slots are only
// APPENDED to a separate list; merge_free_kernel moves them over between batches.  A slot therefore never changes owner inside
// a kernel: the XCDs' L2s are not coherent with each other, and a slot freed by a workgroup on one XCD and re-initialised by a
// workgroup on another would end with whichever dirty line is written back last.
__device__ __forceinline__ bool pool_take(const TrackBatchArgs& a, int n, int* out, int lane) {
    int base = 0;
    if (lane == 0) {
        const int t = __hip_atomic_fetch_sub(a.free_top, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < n) { __hip_atomic_fetch_add(a.free_top, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); base = -1; }
        else base = t - n;
    }
    base = __shfl(base, 0);
    if (base < 0) return false;
    for (int i = lane; i < n; i += 64) out[i] = a.free_stack[base + i];
    return true;
}

__device__ __forceinline__ void report_error(const TrackBatchArgs& a, TrackerHdr* hdr, int code, int tracker, int task) {
    hdr->err = code;                 // the workgroup's copy (sh.hdr); written back with the header when the walk ends
    if (__hip_atomic_exchange(a.status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { a.status[1] = tracker; a.status[2] = task; }
}

extern __shared__ __attribute__((aligned(16))) char track_dyn_lds[];

// NW waves per workgroup; TABLE: every tracker of the launch has its appearance dots in the precomputed table (the normal case) --
// that instance carries no MFMA code and no fallback paths, which is what lets it run 8 waves (one track per wave in P0 / P2).
template <int NW, bool TABLE>
__global__ __launch_bounds__(NW * 64) void track_batch_kernel(const TrackBatchArgs a) {
    __shared__ TrackShared sh;
    // Serial, latency-bound work next to the detector's MFMA-heavy waves on the same CUs: highest issue priority, so that a step's
    // chain of dependent instructions does not queue behind them (the tracker's ~50 workgroups take a negligible share of issue slots)
    __builtin_amdgcn_s_setprio(3);
    // Work arrays: in LDS for steps with up to a.cap tracks + detections (what the host expects from the trackers' recent sizes),
    // in the workgroup's global scratch for the rare larger step (same code, L2 latency instead of LDS latency) up to TC_HARD_CAP.
    StepWork w_lds, w_glb;
    step_work_carve(w_lds, track_dyn_lds, a.cap);
    const TrackWgPlan plan = a.plans[blockIdx.x];
    const TrackDotPlan dp = a.dot_plans[blockIdx.x];
    TrackerHdr* const ghdr = a.hdrs + plan.tracker;
    int* const glist = a.lists + (size_t)plan.tracker * a.list_cap;
    TrackerHdr* hdr = &sh.hdr;
    int* list = sh.list;
    if (threadIdx.x == 0) sh.hdr = *ghdr;
    {
        const int n0 = min(ghdr->n_tracks, (int)TC_HARD_CAP);
        for (int t = threadIdx.x; t < n0; t += NW * 64) sh.list[t] = glist[t];
    }
    __syncthreads();
    char* wg_scratch = (char*)a.scratch + (size_t)blockIdx.x * a.scratch_per_wg;
    double* const cost_app_g = (double*)wg_scratch;         // 4 matrices of TC_MAT doubles: T x D <= TC_MAT whenever T + D <= TC_HARD_CAP
    double* const cost_iou_g = cost_app_g + TC_MAT;
    double* cbuf = cost_iou_g + TC_MAT;
    double* tbuf = cbuf + TC_MAT;
    step_work_carve(w_glb, wg_scratch + 4 * TC_MAT * sizeof(double), TC_HARD_CAP);
    w_lds.small_c = w_glb.small_c = sh.small_c; w_lds.small_t = w_glb.small_t = sh.small_t; w_lds.small_n = w_glb.small_n = 256;
    // dense batches: the matrix the assignment scans lives in LDS behind the step's work arrays (the host sizes it, tracker.hip)
    w_lds.lmat = w_glb.lmat = a.lmat_doubles > 0 ? (double*)(track_dyn_lds + step_work_bytes(a.cap)) : nullptr;
    w_lds.lmat_n = w_glb.lmat_n = a.lmat_doubles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Lanes L{lane, 64};
    const TrackPool& tp = a.pool;
    if (TABLE && !dp.use_table && plan.det_n > 0 && hdr->err == TERR_NONE) {      // cannot happen (the host's bound covers the plan kernel's need)
        if (threadIdx.x == 0) report_error(a, hdr, TERR_TABLE, plan.tracker, plan.task_begin);
        __syncthreads();
    }

#define VC_TTS(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)task * 16 + (i)] = wall_clock64(); } while (0)
    int row_base = 0, row_left = 0;                          // wave 0: the part of the row arena this workgroup has reserved
    TrackTask tk_next = a.tasks[plan.task_begin];
    for (int task = plan.task_begin; task < plan.task_end; ++task) {
        VC_TTS(0);
        const TrackTask tk = tk_next;
        if (task + 1 < plan.task_end) tk_next = a.tasks[task + 1];       // in flight during this step
        const int T = hdr->n_tracks, D = tk.det_n;
        const int prior = hdr->err;
        const StepWork& w = T + D <= a.cap ? w_lds : w_glb;
        const bool small_cost = T * D <= 512 && !a.dbg_costs;            // block-uniform
        double* cost_app = small_cost ? sh.cost_small[0] : cost_app_g;
        double* cost_iou = small_cost ? sh.cost_small[1] : cost_iou_g;
        // every capacity check comes before anything is mutated: a refused step leaves the tracker exactly as it was
        if (prior != TERR_NONE || T + D > TC_HARD_CAP || T + D > a.list_cap) {
            __syncthreads();
            if (threadIdx.x == 0) {
                if (prior == TERR_NONE) report_error(a, hdr, TERR_TRACK_CAP, plan.tracker, task);
                a.task_row_n[task] = 0; a.task_row_off[task] = 0; a.task_ntracks[task] = T; a.task_T[task] = T;
            }
            __syncthreads();
            continue;
        }
        // ---- P0: Track.predict (track.py:112-124) + cost rows -------------------------------------------------------------
        // Steps of up to VC_STEP_STATE tracks (lean instance): a wave predicts a track into the pool AND into the step's LDS copy and
        // goes straight on to that track's cost rows -- no workgroup barrier and no global-memory round trip in between.
        const bool use_st = TABLE && T <= VC_STEP_STATE;                 // block-uniform
        for (int t = wave; t < T; t += NW) {
            const int slot = list[t];
            if (lane == 0) {
                TrackRecD& r = a.recs[slot];
                const int age = r.age + 1, tsu = r.tsu + 1;
                r.age = age; r.tsu = tsu;
                w.slot[t] = slot; w.state[t] = r.state; w.tsu[t] = tsu; w.galc[t] = r.gal_count; w.galh[t] = r.gal_head;
                w.hits[t] = r.hits; w.id[t] = r.id; w.adm[t] = 1;
            }
            kalman_predict_wave(tp.mean + (size_t)slot * 8, tp.cov + (size_t)slot * 64, lane, use_st ? sh.st_mean[t] : nullptr, use_st ? sh.st_cov[t] : nullptr);
            if (use_st && D > 0) {
                wave_sync();                                             // the wave's own LDS writes (work arrays, state copy)
                const int st = w.state[t], tsu = w.tsu[t];
                if (st == CONFIRMED) {
                    const CostJob cj{slot, w.galc[t], tk.det_off, D, t * D, tsu};
                    appearance_row_table(a, dp, cj, tk.det_off - plan.det_begin, cost_app, lane, sh.st_mean[t], sh.st_cov[t], hdr->max_dist, &w.adm[t]);
                }
                if (!(st == CONFIRMED && tsu != 1)) {
                    double b[4];
                    mean_to_tlwh(sh.st_mean[t], b);
                    for (int d = lane; d < D; d += 64)
                        cost_iou[(size_t)t * D + d] = tsu > 1 ? VC_GATED : 1.0 - iou_tlwh(b, a.det_tlwh + (size_t)(tk.det_off + d) * 4);
                }
            }
        }
        if (!use_st) __syncthreads();
        VC_TTS(1);
        if (use_st) {
        } else         if (D > 0 && (TABLE || dp.use_table || D <= VC_SMALL_D)) {
            // one wave per track, no workgroup barriers: rows from the precomputed dot table (the normal case), or computed here
            // for a few detections when the table did not fit the arena
            for (int t = wave; t < T; t += NW) {
                const int st = w.state[t], tsu = w.tsu[t], slot = w.slot[t];
                if (st == CONFIRMED) {
                    const CostJob cj{slot, w.galc[t], tk.det_off, D, t * D, tsu};
                    if (TABLE || dp.use_table) appearance_row_table(a, dp, cj, tk.det_off - plan.det_begin, cost_app, lane, tp.mean + (size_t)slot * 8, tp.cov + (size_t)slot * 64, hdr->max_dist, &w.adm[t]);
                    else if constexpr (!TABLE) appearance_row_wave(tp, cj, a.feat, a.det_featrow, a.det_xyah, cost_app, lane);
                }
                if (!(st == CONFIRMED && tsu != 1)) {
                    double b[4];
                    mean_to_tlwh(tp.mean + (size_t)slot * 8, b);
                    for (int d = lane; d < D; d += 64)
                        cost_iou[(size_t)t * D + d] = tsu > 1 ? VC_GATED : 1.0 - iou_tlwh(b, a.det_tlwh + (size_t)(tk.det_off + d) * 4);
                }
            }
        } else if (!TABLE && D > 0) {
            for (int t = 0; t < T; ++t) {                               // block-uniform
                const int st = w.state[t], tsu = w.tsu[t], slot = w.slot[t];
                if (st == CONFIRMED) {
                    const CostJob cj{slot, w.galc[t], tk.det_off, D, t * D, tsu};
                    if constexpr (!TABLE) appearance_row_dev(tp, cj, a.feat, a.det_featrow, a.det_xyah, cost_app, sh);
                }
                if (!(st == CONFIRMED && tsu != 1)) {                   // IoU candidates only (tracker.py:118-120)
                    double b[4];
                    mean_to_tlwh(tp.mean + (size_t)slot * 8, b);
                    for (int d = threadIdx.x; d < D; d += blockDim.x)
                        cost_iou[(size_t)t * D + d] = tsu > 1 ? VC_GATED : 1.0 - iou_tlwh(b, a.det_tlwh + (size_t)(tk.det_off + d) * 4);
                }
            }
        }
        __syncthreads();
        VC_TTS(2);
        // ---- P1: matching (wave 0) --------------------------------------------------------------------------------------------
        if (wave == 0) {
            int n_match = 0, n_un = 0, n_new = 0, err = TERR_NONE;
            int* newdets = nullptr;
            match_step(L, w, *hdr, T, D, cost_app, cost_iou, cbuf, tbuf, n_match, n_un, newdets, n_new, err, a.dbg ? a.dbg + (size_t)task * 16 + 8 : nullptr, a.no_reg == 0);
            if (err == TERR_NONE && n_new > 0 && !pool_take(a, n_new, w.newslot, lane)) err = TERR_POOL;
            if (lane == 0) { sh.ctl[0] = n_match; sh.ctl[1] = n_un; sh.ctl[2] = n_new; sh.ctl[3] = newdets == w.left ? 1 : 0; sh.ctl[4] = err; }
        }
        __syncthreads();
        VC_TTS(3);
        const int n_match = sh.ctl[0], n_un = sh.ctl[1], n_new = sh.ctl[2], err = sh.ctl[4];
        const int* newdets = sh.ctl[3] ? w.left : w.un_cols;
        if (err != TERR_NONE) {                                          // pool exhausted / infeasible assignment: the tracker stops here
            if (threadIdx.x == 0) {
                report_error(a, hdr, err, plan.tracker, task);
                a.task_row_n[task] = 0; a.task_row_off[task] = 0; a.task_ntracks[task] = T; a.task_T[task] = T;
            }
            __syncthreads();
            continue;
        }
        // ---- P2: Kalman update / initiate + gallery ring writes, one wave per operation ------------------------------------------
        for (int k = wave; k < n_match + n_new; k += NW) {
            if (k < n_match) {                                           // Track.update (track.py:126-145)
                const int t = w.match_t[k], g = tk.det_off + w.match_d[k], slot = w.slot[t];
                kalman_update_wave(tp.mean + (size_t)slot * 8, tp.cov + (size_t)slot * 64, a.det_xyah + (size_t)g * 4, lane, sh.kbuf[wave],
                                   use_st ? sh.st_mean[t] : nullptr, use_st ? sh.st_cov[t] : nullptr);
                float* dst = tp.gallery + ((size_t)slot * tp.budget_cap + w.galh[t]) * VC_FEAT_DIM;
                ((float4*)dst)[lane] = ((const float4*)(a.nfeat + (size_t)g * VC_FEAT_DIM))[lane];              // the normalised feature (track_norm_kernel)
                ((float4*)dst)[64 + lane] = ((const float4*)(a.nfeat + (size_t)g * VC_FEAT_DIM))[64 + lane];
                if (lane == 0) a.gal_row[(size_t)slot * tp.budget_cap + w.galh[t]] = dp.n_old_rows + g - plan.det_begin;
            } else {                                                     // _initiate_track (tracker.py:133-139)
                const int i = k - n_match, g = tk.det_off + newdets[i], slot = w.newslot[i];
                kalman_initiate_wave(tp.mean + (size_t)slot * 8, tp.cov + (size_t)slot * 64, a.det_xyah + (size_t)g * 4, lane);
                float* dst = tp.gallery + (size_t)slot * tp.budget_cap * VC_FEAT_DIM;
                ((float4*)dst)[lane] = ((const float4*)(a.nfeat + (size_t)g * VC_FEAT_DIM))[lane];
                ((float4*)dst)[64 + lane] = ((const float4*)(a.nfeat + (size_t)g * VC_FEAT_DIM))[64 + lane];
                if (lane == 0) a.gal_row[(size_t)slot * tp.budget_cap] = dp.n_old_rows + g - plan.det_begin;
            }
        }
        __syncthreads();
        VC_TTS(4);
        // ---- P3: FSM, list maintenance, rows (wave 0) -----------------------------------------------------------------------------
        if (wave == 0) {
            const int n = finish_step(L, w, hdr, list, a.recs, T, n_match, n_un, n_new, [&](const int* slots, int nd) {
                int pos = 0;
                if (lane == 0) pos = __hip_atomic_fetch_add(a.freed_count, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pos = __shfl(pos, 0);
                for (int i = lane; i < nd; i += 64) a.freed[pos + i] = slots[i];
            });
            // rows: the arena is handed out in chunks (one atomic per VC_ROW_CHUNK rows instead of one per step)
            const int m = compact(L, T, [&](int t) { return w.state[t] == CONFIRMED && w.tsu[t] <= 1; }, [](int, int) {});
            if (m > row_left) {
                const int grab = max(m, VC_ROW_CHUNK);
                int b0 = 0;
                if (lane == 0) b0 = __hip_atomic_fetch_add(a.row_cursor, grab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                row_base = __shfl(b0, 0); row_left = grab;
            }
            const int base = row_base;
            row_base += m; row_left -= m;
            if (base + m > a.rows_cap) {
                if (lane == 0) { report_error(a, hdr, TERR_ROWS, plan.tracker, task); a.task_row_n[task] = 0; a.task_row_off[task] = 0; }
            } else {
                auto mean_of = [&](int t) -> const double* { return use_st ? sh.st_mean[t] : tp.mean + (size_t)w.slot[t] * 8; };
                emit_rows(L, w, mean_of, T, a.frame_w, a.frame_h, tk.label, [&](int pos, const long long* row) {
                    long long* o = a.rows + (size_t)(base + pos) * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) o[c] = row[c];
                });
                if (lane == 0) { a.task_row_n[task] = m; a.task_row_off[task] = base; }
            }
            if (lane == 0) { a.task_ntracks[task] = n; a.task_T[task] = T; }
        }
        __syncthreads();
        VC_TTS(5);
        if (a.dbg && threadIdx.x == 0) { a.dbg[(size_t)task * 16 + 6] = T; a.dbg[(size_t)task * 16 + 7] = D; }
    }
    // the walk is over: header and list go back to global memory (every path above ends in a workgroup barrier)
    {
        const int n1 = min(sh.hdr.n_tracks, (int)TC_HARD_CAP);
        for (int t = threadIdx.x; t < n1; t += NW * 64) glist[t] = sh.list[t];
        if (threadIdx.x == 0) *ghdr = sh.hdr;
    }
#undef VC_TTS
}

// freed slots of the last batch -> free stack (between batches, one workgroup)
__global__ __launch_bounds__(256) void merge_free_kernel(int* free_top, int* free_stack, int* freed_count, const int* freed) {
    const int n = *freed_count, top = *free_top;
    for (int i = threadIdx.x; i < n; i += blockDim.x) free_stack[top + i] = freed[i];
    __syncthreads();
    if (threadIdx.x == 0) { *free_top = top + n; *freed_count = 0; }
}

size_t track_scratch_per_wg() { return ((size_t)4 * TC_MAT * sizeof(double) + step_work_bytes(TC_HARD_CAP) + 255) & ~(size_t)255; }

int launch_track_batch(const TrackBatchArgs& a, int n_wg, hipStream_t s) {
    if (n_wg <= 0) return VC_OK;
    const size_t lds = step_work_bytes(a.cap) + (size_t)a.lmat_doubles * sizeof(double);
    static const bool lds_ok = [] {                          // dynamic LDS beyond 64 KB (cap = 512) has to be allowed per kernel
        return hipFuncSetAttribute((const void*)track_batch_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, VC_TRACK_DYN_LDS) == hipSuccess &&
               hipFuncSetAttribute((const void*)track_batch_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, VC_TRACK_DYN_LDS) == hipSuccess;
    }();
    VC_CHECK(lds_ok, VC_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the tracker kernel");
    VC_CHECK(a.cap % 8 == 0 && a.cap <= TC_HARD_CAP && lds <= VC_TRACK_DYN_LDS, VC_ERR_CAPACITY, "tracker step capacity %d does not fit the workgroup's LDS", a.cap);
    VC_CHECK(a.scratch_per_wg >= track_scratch_per_wg(), VC_ERR_ARG, "tracker scratch too small");
    if (a.n_det_total > 0) {
        hipLaunchKernelGGL(track_plan_kernel, dim3(n_wg), dim3(256), 0, s, a);
        hipLaunchKernelGGL(track_norm_kernel, dim3((a.n_det_total + 3) / 4), dim3(256), 0, s, a);
        hipLaunchKernelGGL(track_dots_kernel, dim3(std::min(2048, std::max(64, a.n_det_total))), dim3(256), 0, s, a, n_wg);
        VC_HIP(hipGetLastError());
    }
    if (a.all_tables) hipLaunchKernelGGL((track_batch_kernel<8, true>), dim3(n_wg), dim3(512), lds, s, a);
    else hipLaunchKernelGGL((track_batch_kernel<4, false>), dim3(n_wg), dim3(256), lds, s, a);
    VC_HIP(hipGetLastError());
    hipLaunchKernelGGL(merge_free_kernel, dim3(1), dim3(256), 0, s, a.free_top, a.free_stack, a.freed_count, a.freed);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// ---- single-function kernels of the parity tests: the batch kernel's own __device__ functions on caller-supplied state --------
template <int WHICH>      // 0 initiate, 1 predict, 2 update
__global__ __launch_bounds__(64) void kat_kalman_kernel(TrackPool tp, const double* z, int n) {
    __shared__ double kbuf[32];
    const int i = blockIdx.x, lane = threadIdx.x;
    double* m = tp.mean + (size_t)i * 8;
    double* P = tp.cov + (size_t)i * 64;
    if (WHICH == 0) kalman_initiate_wave(m, P, z + (size_t)i * 4, lane);
    else if (WHICH == 1) kalman_predict_wave(m, P, lane);
    else kalman_update_wave(m, P, z + (size_t)i * 4, lane, kbuf);
}

int launch_kat_kalman(TrackPool& tp, int which, const double* z, int n, hipStream_t s) {
    if (n <= 0) return VC_OK;
    if (which == 0) hipLaunchKernelGGL(kat_kalman_kernel<0>, dim3(n), dim3(64), 0, s, tp, z, n);
    else if (which == 1) hipLaunchKernelGGL(kat_kalman_kernel<1>, dim3(n), dim3(64), 0, s, tp, z, n);
    else hipLaunchKernelGGL(kat_kalman_kernel<2>, dim3(n), dim3(64), 0, s, tp, z, n);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

__global__ __launch_bounds__(256) void kat_appearance_kernel(TrackPool tp, const CostJob* jobs, const float* feat, const int* det_feat_row,
                                                            const double* det_xyah, double* out) {
    __shared__ TrackShared sh;
    appearance_row_dev(tp, jobs[blockIdx.x], feat, det_feat_row, det_xyah, out, sh);
}

int launch_appearance_cost(const TrackPool& tp, const CostJob* jobs, int njobs, const float* feat, const int* det_feat_row,
                           const double* det_xyah, double* out, hipStream_t s) {
    if (njobs <= 0) return VC_OK;
    hipLaunchKernelGGL(kat_appearance_kernel, dim3(njobs), dim3(256), 0, s, tp, jobs, feat, det_feat_row, det_xyah, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

__global__ __launch_bounds__(64) void kat_gallery_write_kernel(TrackPool tp, const int* __restrict__ sps, const float* __restrict__ feat) {
    const int* e = sps + (size_t)blockIdx.x * 3;
    gallery_store_wave(tp.gallery + ((size_t)e[0] * tp.budget_cap + e[1]) * VC_FEAT_DIM, feat + (size_t)e[2] * VC_FEAT_DIM, threadIdx.x);
}

int launch_gallery_write(TrackPool& tp, const int* slot_pos_src, int n, const float* feat, hipStream_t s) {
    if (n <= 0) return VC_OK;
    hipLaunchKernelGGL(kat_gallery_write_kernel, dim3(n), dim3(64), 0, s, tp, slot_pos_src, feat);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

__global__ __launch_bounds__(64) void kat_iou_boxes_kernel(const double* a, int t, const double* b, int d, double* out) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < d; j += blockDim.x) out[(size_t)i * d + j] = iou_tlwh(a + (size_t)i * 4, b + (size_t)j * 4);
}

__global__ __launch_bounds__(64) void kat_gating_kernel(TrackPool tp, int slot, const double* z, int n, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* m = tp.mean + (size_t)slot * 8;
    double S[16], L[16];
    project4(m, tp.cov + (size_t)slot * 64, S);
    chol4(S, L);
    out[i] = maha4(m, L, z + (size_t)i * 4);
}

// scipy.optimize.linear_sum_assignment through the batch kernel's own solver (one wave, work arrays in LDS)
__global__ __launch_bounds__(64) void kat_lap_kernel(const double* cost, int nr, int nc, int cap, double* tbuf, int* out_rows, int* out_cols, int* out_n) {
    StepWork w;
    step_work_carve(w, track_dyn_lds, cap);
    const Lanes L{(int)threadIdx.x, 64};
    int err = TERR_NONE;
    const int np = lap_solve(L, w, cost, nr, nc, tbuf, err);
    for (int k = L.lane; k < np; k += 64) { out_rows[k] = w.ri[k]; out_cols[k] = w.ci[k]; }
    if (L.lane == 0) *out_n = err == TERR_NONE ? np : -1;
}

int launch_kat_lap(const double* cost, int nr, int nc, double* tbuf, int* out_rows, int* out_cols, int* out_n, hipStream_t s) {
    const int cap = (std::max(nr, nc) + 7) / 8 * 8;
    static const bool lds_ok = hipFuncSetAttribute((const void*)kat_lap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VC_TRACK_DYN_LDS) == hipSuccess;
    VC_CHECK(lds_ok && step_work_bytes(cap) <= VC_TRACK_DYN_LDS, VC_ERR_CAPACITY, "lap: at most 512 rows / columns");
    hipLaunchKernelGGL(kat_lap_kernel, dim3(1), dim3(64), step_work_bytes(cap), s, cost, nr, nc, cap, tbuf, out_rows, out_cols, out_n);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_gating_values(const TrackPool& tp, int slot, const double* z, int n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(kat_gating_kernel, dim3((n + 63) / 64), dim3(64), 0, s, tp, slot, z, n, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}
int launch_iou_boxes(const double* a, int t, const double* b, int d, double* out, hipStream_t s) {
    hipLaunchKernelGGL(kat_iou_boxes_kernel, dim3(t), dim3(64), 0, s, a, t, b, d, out);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
