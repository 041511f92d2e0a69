// TEST HARNESS (never linked into libvcount_hip.so): the device tracker's control logic -- csrc/track_core.h: cascade, exact
// assignment with SciPy's tie-breaking, track FSM, list maintenance, row emission -- compiled for the host with one "lane" and
// driven through plain serial forms of the step's numerics (csrc/track_math.h), so that the CPU test suite can run the SAME
// source the GPU executes against the reference's golden tracker traces.  Built by tests/test_track_core_host.py with
//   g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC
#include <cmath>
#include <cstring>
#include <vector>

#include "../../vehicle-counting_amd/csrc/track_core.h"

using namespace vc;
using namespace vc::tc;

namespace {
constexpr int FEAT = 512;

struct HostTracker {
    TrackerHdr hdr{};
    int cap = 0, pool = 0;
    std::vector<int> list, free_stack;
    std::vector<TrackRecD> recs;
    std::vector<double> mean, cov;
    std::vector<float> gallery;            // [pool][budget][512], rows stored normalised like the device pool
    std::vector<char> work;
    std::vector<double> cost_app, cost_iou, cbuf, tbuf;
    std::vector<long long> rows;
    int n_rows = 0;
};

void gallery_store(float* dst, const float* src) {
    float ss = 0.f;
    for (int i = 0; i < FEAT; ++i) ss += src[i] * src[i];
    const float nrm = sqrtf(ss);
    for (int i = 0; i < FEAT; ++i) dst[i] = src[i] / nrm;
}
}  // namespace

extern "C" {

void* tch_create(double max_dist, double max_iou_distance, int max_age, int n_init, int budget, int cap, int pool) {
    HostTracker* t = new HostTracker();
    t->hdr.max_dist = max_dist; t->hdr.max_iou_distance = max_iou_distance; t->hdr.max_age = max_age; t->hdr.n_init = n_init;
    t->hdr.nn_budget = budget; t->hdr.next_id = 1; t->hdr.n_tracks = 0; t->hdr.err = 0;
    t->cap = cap; t->pool = pool;
    t->list.assign(cap, -1);
    t->recs.resize(pool);
    t->mean.assign((size_t)pool * 8, 0.0); t->cov.assign((size_t)pool * 64, 0.0);
    t->gallery.assign((size_t)pool * budget * FEAT, 0.f);
    for (int i = 0; i < pool; ++i) t->free_stack.push_back(pool - 1 - i);
    t->work.resize(step_work_bytes(cap) + 64);
    t->cost_app.resize((size_t)cap * cap); t->cost_iou.resize((size_t)cap * cap); t->cbuf.resize((size_t)cap * cap); t->tbuf.resize((size_t)cap * cap);
    t->rows.resize((size_t)cap * 6);
    return t;
}

void tch_destroy(void* h) { delete (HostTracker*)h; }

// Tracker.predict() + Tracker.update(detections): tlwh [k][4] f64, feat [k][512] f32.  W/H: frame size for the emitted rows.
int tch_step(void* hp, const double* tlwh, const float* feat, int k, int W, int H, int label) {
    HostTracker& h = *(HostTracker*)hp;
    const Lanes L{0, 1};
    StepWork w;
    step_work_carve(w, h.work.data(), h.cap);
    const int T = h.hdr.n_tracks, D = k, S = h.hdr.nn_budget;
    if (T + D > h.cap) return TERR_TRACK_CAP;                         // checked before anything is mutated
    std::vector<double> xyah((size_t)D * 4);
    for (int d = 0; d < D; ++d) {                                     // detection.py:42-50 to_xyah
        const double* t = tlwh + (size_t)d * 4;
        xyah[d * 4] = t[0] + t[2] / 2; xyah[d * 4 + 1] = t[1] + t[3] / 2; xyah[d * 4 + 2] = t[2] / t[3]; xyah[d * 4 + 3] = t[3];
    }
    // load + predict (track.py:112-124) + cost rows
    for (int t = 0; t < T; ++t) {
        const int slot = h.list[t];
        TrackRecD& r = h.recs[slot];
        r.age += 1; r.tsu += 1;
        w.slot[t] = slot; w.state[t] = r.state; w.tsu[t] = r.tsu; w.galc[t] = r.gal_count; w.galh[t] = r.gal_head; w.hits[t] = r.hits; w.id[t] = r.id; w.adm[t] = 1;
        double* m = &h.mean[(size_t)slot * 8];
        double* P = &h.cov[(size_t)slot * 64];
        kalman_predict_dev(m, P);
        if (D > 0 && r.state == CONFIRMED) {
            double Sg[16], Lc[16];
            project4(m, P, Sg);
            chol4(Sg, Lc);
            for (int d = 0; d < D; ++d) {
                const float* f = feat + (size_t)d * FEAT;
                float ss = 0.f;
                for (int i = 0; i < FEAT; ++i) ss += f[i] * f[i];
                float best = -INFINITY;
                for (int s = 0; s < r.gal_count; ++s) {
                    const float* g = &h.gallery[((size_t)slot * S + s) * FEAT];
                    float acc = 0.f;
                    for (int i = 0; i < FEAT; ++i) acc += g[i] * f[i];
                    best = fmaxf(best, acc);
                }
                const float cosv = best * (1.0f / sqrtf(ss));
                const double g2 = maha4(m, Lc, &xyah[(size_t)d * 4]);
                h.cost_app[(size_t)t * D + d] = g2 > VC_CHI2_95_4 ? VC_GATED : (double)(1.0f - cosv);
            }
            bool any = false;                                         // StepWork::adm, as appearance_row_table writes it on the device
            for (int d = 0; d < D; ++d) any = any || !(h.cost_app[(size_t)t * D + d] > h.hdr.max_dist);
            w.adm[t] = any ? 1 : 0;
        }
        if (D > 0 && !(r.state == CONFIRMED && r.tsu != 1)) {
            double b[4];
            mean_to_tlwh(m, b);
            for (int d = 0; d < D; ++d) h.cost_iou[(size_t)t * D + d] = r.tsu > 1 ? VC_GATED : 1.0 - iou_tlwh(b, tlwh + (size_t)d * 4);
        }
    }
    int n_match = 0, n_un = 0, n_new = 0, err = 0;
    int* newdets = nullptr;
    match_step(L, w, h.hdr, T, D, h.cost_app.data(), h.cost_iou.data(), h.cbuf.data(), h.tbuf.data(), n_match, n_un, newdets, n_new, err);
    if (err) return err;
    if ((int)h.free_stack.size() < n_new) return TERR_POOL;
    for (int i = 0; i < n_new; ++i) { w.newslot[i] = h.free_stack.back(); h.free_stack.pop_back(); }
    // Kalman update / initiate + gallery ring writes
    for (int q = 0; q < n_match; ++q) {
        const int t = w.match_t[q], d = w.match_d[q], slot = w.slot[t];
        kalman_update_dev(&h.mean[(size_t)slot * 8], &h.cov[(size_t)slot * 64], &xyah[(size_t)d * 4]);
        gallery_store(&h.gallery[((size_t)slot * S + w.galh[t]) * FEAT], feat + (size_t)d * FEAT);
    }
    for (int i = 0; i < n_new; ++i) {
        const int d = newdets[i], slot = w.newslot[i];
        kalman_initiate_dev(&h.mean[(size_t)slot * 8], &h.cov[(size_t)slot * 64], &xyah[(size_t)d * 4]);
        gallery_store(&h.gallery[((size_t)slot * S + 0) * FEAT], feat + (size_t)d * FEAT);
    }
    const int n = finish_step(L, w, &h.hdr, h.list.data(), h.recs.data(), T, n_match, n_un, n_new, [&](const int* slots, int nd) { for (int i = 0; i < nd; ++i) h.free_stack.push_back(slots[i]); });
    (void)n;
    h.n_rows = emit_rows(L, w, [&](int t) -> const double* { return &h.mean[(size_t)w.slot[t] * 8]; }, T, W, H, label,
                         [&](int pos, const long long* row) { memcpy(&h.rows[(size_t)pos * 6], row, 6 * sizeof(long long)); });
    return 0;
}

int tch_state(void* hp, long long* ids, int* state, int* hits, int* age, int* tsu, double* mean8, double* covdiag8, int* gal) {
    HostTracker& h = *(HostTracker*)hp;
    for (int t = 0; t < h.hdr.n_tracks; ++t) {
        const int slot = h.list[t];
        const TrackRecD& r = h.recs[slot];
        ids[t] = r.id; state[t] = r.state; hits[t] = r.hits; age[t] = r.age; tsu[t] = r.tsu;
        gal[t] = r.state == CONFIRMED ? r.gal_count : 0;
        memcpy(mean8 + (size_t)t * 8, &h.mean[(size_t)slot * 8], 64);
        for (int i = 0; i < 8; ++i) covdiag8[(size_t)t * 8 + i] = h.cov[(size_t)slot * 64 + i * 9];
    }
    return h.hdr.n_tracks;
}

int tch_rows(void* hp, long long* rows6) {
    HostTracker& h = *(HostTracker*)hp;
    memcpy(rows6, h.rows.data(), (size_t)h.n_rows * 6 * sizeof(long long));
    return h.n_rows;
}

// pyset_difference_order alone: list(set(conf) - set(k for k in conf if matched[k])) in CPython's iteration order
int tch_pyset_order(const int* conf, int n1, const unsigned char* matched, int n2, int* out) {
    const int slots = pyset_need_slots(n1);
    std::vector<short> m0(slots), m1(slots);
    return pyset_difference_order(Lanes{0, 1}, conf, n1, [&](int k) { return matched[k] != 0; }, n2, m0.data(), m1.data(), out);
}

// lap_solve alone: pairs sorted by row like scipy.optimize.linear_sum_assignment
int tch_lap(const double* cost, int nr, int nc, int* rows, int* cols) {
    const int cap = ((nr > nc ? nr : nc) + 7) / 8 * 8;
    std::vector<char> buf(step_work_bytes(cap) + 64);
    StepWork w;
    step_work_carve(w, buf.data(), cap);
    std::vector<double> t((size_t)nr * nc + 1);
    int err = 0;
    const int np = lap_solve(Lanes{0, 1}, w, cost, nr, nc, t.data(), err);
    if (err) return -1;
    for (int k = 0; k < np; ++k) { rows[k] = w.ri[k]; cols[k] = w.ci[k]; }
    return np;
}

}  // extern "C"
