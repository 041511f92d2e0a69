// DeepSORT numerics shared by the device tracker (track_kernels.hip) and the host build of the tracker's control logic that
// the CPU tests compile (tests/native/track_core_host.cpp): serial forms of the Kalman filter, the gate and the IoU.
// fp64 where the reference is fp64.  Compile with floating-point contraction OFF (the operation order is the reference's).
//
// Reference (paths relative to /root/reference/networks/deepsort/sort/):
//   kalman_filter.py:55-85   initiate            kalman_filter.py:87-121  predict   (F P F^T is exact: F is 0/1)
//   kalman_filter.py:123-152 project             kalman_filter.py:154-186 update    (4x4 Cholesky of S, K = P H^T S^-1)
//   kalman_filter.py:188-229 gating_distance     iou_matching.py:7-81     iou       track.py:82-96 to_tlwh
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VC_HD __host__ __device__ __forceinline__
#else
#define VC_HD inline
#endif

namespace vc {

#define VC_W_POS (1.0 / 20)
#define VC_W_VEL (1.0 / 160)
#define VC_CHI2_95_4 9.4877
#define VC_GATED 1e5

VC_HD void kalman_initiate_dev(double* m, double* P, const double* z) {
    const double h = z[3];
    for (int k = 0; k < 4; ++k) { m[k] = z[k]; m[4 + k] = 0.0; }
    const double sp = (2 * VC_W_POS) * h, sv = (10 * VC_W_VEL) * h;
    const double sd[8] = {sp, sp, 1e-2, sp, sv, sv, 1e-5, sv};
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) P[r * 8 + c] = r == c ? sd[r] * sd[r] : 0.0;
}

VC_HD void kalman_predict_dev(double* m, double* P) {
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

// S = H P H^T + R (4x4), projected mean = mean[:4]
VC_HD void project4(const double* m, const double* P, double S[16]) {
    const double sp = VC_W_POS * m[3];
    const double sd[4] = {sp, sp, 1e-1, sp};
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) S[r * 4 + c] = P[r * 8 + c] + (r == c ? sd[r] * sd[r] : 0.0);
}

VC_HD void chol4(const double S[16], double L[16]) {
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

VC_HD void kalman_update_dev(double* m, double* P, const double* z) {
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

VC_HD double maha4(const double* m, const double L[16], const double* z) {
    double y[4], acc = 0.0;
    for (int a = 0; a < 4; ++a) {
        double v = z[a] - m[a];
        for (int k = 0; k < a; ++k) v -= L[a * 4 + k] * y[k];
        y[a] = v / L[a * 4 + a];
        acc += y[a] * y[a];
    }
    return acc;
}

VC_HD void mean_to_tlwh(const double* m, double t[4]) {
    t[2] = m[2] * m[3];
    t[3] = m[3];
    t[0] = m[0] - t[2] / 2;
    t[1] = m[1] - t[3] / 2;
}

VC_HD double iou_tlwh(const double* b, const double* c) {
    const double tlx = fmax(b[0], c[0]), tly = fmax(b[1], c[1]);
    const double brx = fmin(b[0] + b[2], c[0] + c[2]), bry = fmin(b[1] + b[3], c[1] + c[3]);
    const double w = fmax(0.0, brx - tlx), h = fmax(0.0, bry - tly);
    const double inter = w * h;
    return inter / (b[2] * b[3] + c[2] * c[3] - inter);
}

}  // namespace vc
