// DeepSORT tracker: host-side track lifecycle + assignment around the batched device numerics of
// track_kernels.hip.  One Tracker per (camera, class) like the reference (modules/track.py:16); all of a frame's
// trackers are stepped together so a frame costs two host<->device round trips regardless of the class count.
//
// Reference (paths relative to /root/reference/networks/deepsort/):
//   sort/tracker.py:40-139        Tracker.predict / update / _match / _initiate_track
//   sort/track.py:4-175           Track state machine (Tentative -> Confirmed -> Deleted), counters
//   sort/linear_assignment.py     min_cost_matching :13-77, matching_cascade :80-145 (gate folded into the kernel)
//   sort/nn_matching.py:137-154   partial_fit: per-target sample lists trimmed to `budget` (device ring buffer here)
//   sort/preprocessing.py:6-73    non_max_suppression (quirk Q6)
//   deep_sort.py:25-59            DeepSort.update; modules/track.py:30-70 VideoTracker.run
//   scipy.optimize.linear_sum_assignment (third-party; Crouse's shortest augmenting path, restated in lap_solve)
#include <algorithm>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>
#include <map>
#include <numeric>

#include "engine.h"

namespace vc {

enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };

// ------------------------------------------------------------------------------------------------ LSAP
// Rectangular linear sum assignment, same algorithm and tie-breaking as SciPy's rectangular_lsap
// (D. F. Crouse, "On implementing 2D rectangular assignment algorithms", 2016): returns pairs sorted by row.
static int lsap_core(int nr, int nc, const double* cost, std::vector<int>& col4row) {
    const double INF = std::numeric_limits<double>::infinity();
    // scratch reused across calls (a frame makes 10-20 small assignments; the heap traffic was a third of the host step)
    static thread_local std::vector<double> u, v, spc;
    static thread_local std::vector<int> path, row4col, remaining;
    static thread_local std::vector<char> SR, SC;
    u.assign(nr, 0.0); v.assign(nc, 0.0); spc.resize(nc);
    path.assign(nc, -1); row4col.assign(nc, -1); remaining.resize(nc);
    SR.resize(nr); SC.resize(nc);
    col4row.assign(nr, -1);
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INF);
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INF) return -1;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r) if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j) if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    return 0;
}

int lap_solve(const double* cost, int nr, int nc, std::vector<int>& rows, std::vector<int>& cols) {
    rows.clear(); cols.clear();
    if (nr == 0 || nc == 0) return VC_OK;
    static thread_local std::vector<int> c4r;
    static thread_local std::vector<double> t;
    if (nc < nr) {                       // SciPy transposes so that rows <= cols
        t.resize((size_t)nr * nc);
        for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) t[(size_t)j * nr + i] = cost[(size_t)i * nc + j];
        VC_CHECK(lsap_core(nc, nr, t.data(), c4r) == 0, VC_ERR_ARG, "lap: infeasible cost matrix");
        std::vector<int> order(nc);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return c4r[a] < c4r[b]; });
        for (int v : order) { rows.push_back(c4r[v]); cols.push_back(v); }
    } else {
        VC_CHECK(lsap_core(nr, nc, cost, c4r) == 0, VC_ERR_ARG, "lap: infeasible cost matrix");
        for (int i = 0; i < nr; ++i) { rows.push_back(i); cols.push_back(c4r[i]); }
    }
    return VC_OK;
}

// sort/linear_assignment.py:52-77 on a dense sub-matrix given as full rows + selected columns
struct MatchOut { std::vector<std::pair<int, int>> matches; std::vector<int> un_rows, un_cols; };
static int min_cost_matching(const std::vector<const double*>& row_ptr, const std::vector<int>& rows, const std::vector<int>& cols,
                             double max_cost, MatchOut& out) {
    out.matches.clear(); out.un_rows.clear(); out.un_cols.clear();
    const int nr = (int)rows.size(), nc = (int)cols.size();
    if (nr == 0 || nc == 0) { out.un_rows = rows; out.un_cols = cols; return VC_OK; }
    static thread_local std::vector<double> c;
    static thread_local std::vector<int> ri, ci;
    static thread_local std::vector<char> col_used, row_used;
    c.resize((size_t)nr * nc);
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) {
            const double v = row_ptr[i][cols[j]];
            c[(size_t)i * nc + j] = v > max_cost ? max_cost + 1e-5 : v;
        }
    VC_TRY(lap_solve(c.data(), nr, nc, ri, ci));
    col_used.assign(nc, 0); row_used.assign(nr, 0);
    for (size_t k = 0; k < ri.size(); ++k) { row_used[ri[k]] = 1; col_used[ci[k]] = 1; }
    for (int j = 0; j < nc; ++j) if (!col_used[j]) out.un_cols.push_back(cols[j]);
    for (int i = 0; i < nr; ++i) if (!row_used[i]) out.un_rows.push_back(rows[i]);
    for (size_t k = 0; k < ri.size(); ++k) {
        if (c[(size_t)ri[k] * nc + ci[k]] > max_cost) { out.un_rows.push_back(rows[ri[k]]); out.un_cols.push_back(cols[ci[k]]); }
        else out.matches.emplace_back(rows[ri[k]], cols[ci[k]]);
    }
    return VC_OK;
}

// sort/preprocessing.py:6-73 (overlap = inter / area(other), +1 pixel, '>' threshold; quirk Q6)
void dsort_nms(const double* tlwh, const double* scores, int n, double max_overlap, std::vector<int>& keep) {
    keep.clear();
    if (n == 0) return;
    std::vector<double> x1(n), y1(n), x2(n), y2(n), area(n);
    for (int i = 0; i < n; ++i) {
        x1[i] = tlwh[i * 4]; y1[i] = tlwh[i * 4 + 1]; x2[i] = tlwh[i * 4 + 2] + tlwh[i * 4]; y2[i] = tlwh[i * 4 + 3] + tlwh[i * 4 + 1];
        area[i] = (x2[i] - x1[i] + 1) * (y2[i] - y1[i] + 1);
    }
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return scores[a] < scores[b]; });
    while (!idx.empty()) {
        const int i = idx.back();
        idx.pop_back();
        keep.push_back(i);
        std::vector<int> rest;
        for (int j : idx) {
            const double w = std::max(0.0, std::min(x2[i], x2[j]) - std::max(x1[i], x1[j]) + 1);
            const double h = std::max(0.0, std::min(y2[i], y2[j]) - std::max(y1[i], y1[j]) + 1);
            if (!((w * h) / area[j] > max_overlap)) rest.push_back(j);
        }
        idx.swap(rest);
    }
}

// ------------------------------------------------------------------------------------------------ pool
static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

int tracker_init_pool(vc_engine* e) {
    const size_t T = e->cfg.max_tracks, S = e->cfg.nn_budget_cap;
    VC_CHECK(T >= 1 && S >= 1, VC_ERR_ARG, "max_tracks and nn_budget_cap must be positive");
    e->pool.max_tracks = (int)T; e->pool.budget_cap = (int)S;
    VC_TRY(dev_alloc(e, (void**)&e->pool.mean, T * 8 * sizeof(double)));
    VC_TRY(dev_alloc(e, (void**)&e->pool.cov, T * 64 * sizeof(double)));
    VC_TRY(dev_alloc(e, (void**)&e->pool.gallery, T * S * VC_FEAT_DIM * sizeof(float)));
    e->free_slots.resize(T);
    for (size_t i = 0; i < T; ++i) e->free_slots[i] = (int)(T - 1 - i);
    e->det_cap = std::max(e->cfg.max_det * 2, 1024);
    e->cost_cap = (size_t)2 * T * 512;
    const size_t D = e->det_cap;
    // one pinned staging block per phase, mirrored on the device: a phase costs ONE host->device copy
    e->stage_cap = align16((T + D) * 4) + 2 * align16(D * 32) + align16(2 * T * sizeof(CostJob)) +      // phase A
                   2 * align16(T * 4) + 2 * align16((T + D) * 32) + align16((T + D) * 12) + 256;          // phase B (superset)
    e->stage_cap = std::max(e->stage_cap, (T + D) * sizeof(TrackChainRec) + 256);
    e->slot_chain.assign(T, -1);
    VC_TRY(dev_alloc(e, (void**)&e->d_track_counter, 64));
    VC_HIP(hipMemset(e->d_track_counter, 0, 64));
    VC_TRY(host_alloc(e, (void**)&e->h_track_flag, 64));
    memset(e->h_track_flag, 0, 64);
    VC_HIP(hipHostGetDevicePointer((void**)&e->hd_track_flag, e->h_track_flag, 0));
    // pinned + device-mapped: the per-frame kernels read their descriptors from and write their results to host memory
    VC_TRY(host_alloc(e, (void**)&e->h_stage, e->stage_cap));
    VC_TRY(host_alloc(e, (void**)&e->h_stage2, e->stage_cap));
    VC_TRY(host_alloc(e, (void**)&e->h_cost, e->cost_cap * sizeof(double)));
    VC_TRY(host_alloc(e, (void**)&e->h_mean, T * 8 * sizeof(double)));
    VC_HIP(hipHostGetDevicePointer((void**)&e->hd_stage, e->h_stage, 0));
    VC_HIP(hipHostGetDevicePointer((void**)&e->hd_stage2, e->h_stage2, 0));
    VC_HIP(hipHostGetDevicePointer((void**)&e->hd_cost, e->h_cost, 0));
    VC_HIP(hipHostGetDevicePointer((void**)&e->hd_mean, e->h_mean, 0));
    VC_TRY(dev_alloc(e, (void**)&e->d_feat_in, D * VC_FEAT_DIM * sizeof(float)));
    return VC_OK;
}

static int slot_alloc(vc_engine* e) {
    if (e->free_slots.empty()) return -1;
    const int slot = e->free_slots.back();
    e->free_slots.pop_back();
    return slot;
}
static void slot_free(vc_engine* e, int slot) { e->free_slots.push_back(slot); }

static void tlwh_to_xyah(const double* t, double* o) {      // sort/detection.py:42-50
    o[0] = t[0] + t[2] / 2; o[1] = t[1] + t[3] / 2; o[2] = t[2] / t[3]; o[3] = t[3];
}

// bump allocator over the pinned staging block; device addresses mirror host offsets
struct Stage {
    char* h; char* d; size_t cap, off = 0;
    template <class T> T* take(size_t n, T** dev) {
        T* p = (T*)(h + off);
        *dev = (T*)(d + off);
        off = align16(off + n * sizeof(T));
        return p;
    }
};

// One tracker step for a set of trackers (all classes of one frame), split at its only data dependency on the device:
//   track_prepare_a  (host)    Tracker.predict bookkeeping, one cost job per live track (predict + appearance/gate + IoU rows)
//   track_launch     (device)  the pending operations of the PREVIOUS frame and the cost jobs of this one, one kernel
//   track_wait       (host)    poll the completion word; the cost rows (and the previous frame's posterior means) are in
//                              pinned memory
//   track_host_b     (host)    matching cascade + IoU matching (exact LSAP), Track.update / mark_missed / _initiate_track:
//                              the Kalman update / initiate / gallery writes become the pending operations of this frame
// so a frame costs ONE launch and ONE round trip (vc_stream_run); the blocking entry points flush the pending
// operations with a second launch.
int track_prepare_a(vc_engine* e, StepCtx& c) {
    const int njobs = (int)c.ids.size();
    int n_tracks = 0;
    c.n_dets = 0;
    c.det_base.assign(njobs, 0);
    for (int j = 0; j < njobs; ++j) {
        VC_CHECK(c.ids[j] >= 0 && c.ids[j] < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id %d", c.ids[j]);
        n_tracks += (int)e->trackers[c.ids[j]]->tracks.size();
        c.det_base[j] = c.n_dets;
        c.n_dets += (int)c.prep[j].conf.size();
    }
    const int n_dets = c.n_dets;
    VC_CHECK(n_dets <= e->det_cap, VC_ERR_CAPACITY, "%d detections in one step exceed capacity %d", n_dets, e->det_cap);
    // descriptors + detections go into the pinned, device-mapped staging block: the kernel reads them over the host link
    // (a few hundred bytes) and writes the cost rows straight into pinned memory -- no copy operations in the phase
    Stage st{e->h_stage, e->hd_stage, e->stage_cap};
    int* d_featrow; double *d_xyah, *d_tlwh; TrackJobA* d_jobs;
    int* h_featrow = st.take<int>(n_dets, &d_featrow);
    double* h_xyah = st.take<double>((size_t)n_dets * 4, &d_xyah);
    double* h_tlwh = st.take<double>((size_t)n_dets * 4, &d_tlwh);
    TrackJobA* h_jobs = st.take<TrackJobA>(n_tracks, &d_jobs);
    VC_CHECK(st.off <= e->stage_cap, VC_ERR_CAPACITY, "tracker staging block too small");
    for (int j = 0; j < njobs; ++j) {
        const Prepared& p = c.prep[j];
        for (size_t i = 0; i < p.conf.size(); ++i) {
            const int g = c.det_base[j] + (int)i;
            memcpy(h_tlwh + (size_t)g * 4, &p.tlwh[i * 4], 4 * sizeof(double));
            tlwh_to_xyah(&p.tlwh[i * 4], h_xyah + (size_t)g * 4);
            h_featrow[g] = p.feat_rows[i];
        }
    }
    // one job per live track: Kalman predict + (confirmed) appearance row + (IoU candidate) IoU row over its tracker's dets
    c.out = 0;
    c.app_job.assign(njobs, {});
    c.iou_job.assign(njobs, {});
    int q = 0;
    for (int j = 0; j < njobs; ++j) {
        Tracker& tk = *e->trackers[c.ids[j]];
        const int k = (int)c.prep[j].conf.size();
        c.app_job[j].assign(tk.tracks.size(), -1);
        c.iou_job[j].assign(tk.tracks.size(), -1);
        for (size_t t = 0; t < tk.tracks.size(); ++t) {
            TrackRec& tr = tk.tracks[t];
            tr.age += 1; tr.tsu += 1;                                      // sort/track.py:112-124
            TrackJobA jb{tr.slot, tr.gal_count, c.det_base[j], k, -1, -1, tr.tsu, 0};
            if (k > 0 && tr.state == CONFIRMED) { jb.app_off = (int)c.out; c.app_job[j][t] = jb.app_off; c.out += k; }
            if (k > 0 && !(tr.state == CONFIRMED && tr.tsu != 1)) {         // IoU candidates only (sort/tracker.py:118-120)
                jb.iou_off = (int)c.out; c.iou_job[j][t] = jb.iou_off; c.out += k;
            }
            h_jobs[q++] = jb;
        }
    }
    VC_CHECK(c.out <= e->cost_cap, VC_ERR_CAPACITY, "cost matrices (%zu entries) exceed capacity %zu", c.out, e->cost_cap);
    c.det_xyah.assign(h_xyah, h_xyah + (size_t)n_dets * 4);
    c.featrow.assign(h_featrow, h_featrow + n_dets);
    c.n_jobs = n_tracks;
    c.h_jobs = h_jobs; c.d_jobs = d_jobs; c.d_featrow = d_featrow; c.d_xyah = d_xyah; c.d_tlwh = d_tlwh;
    return VC_OK;
}

// Chain records of one step: the pending operations of `cb` (may be null) merged with the cost jobs of `ca` (may be
// null), one record per touched slot, written to the second pinned staging block.
static int build_chain_recs(vc_engine* e, const StepCtx* cb, const StepCtx* ca, const TrackChainRec** d_recs, int* n_out) {
    const int nops = cb ? (int)cb->ops.size() : 0, njobs = ca ? ca->n_jobs : 0;
    std::vector<TrackChainRec>& tmp = e->chain_scratch;
    tmp.clear();
    TrackChainRec blank{};
    blank.op.kind = -1; blank.op.slot = -1; blank.op.feat_row = -1; blank.op.out_row = -1;
    blank.job.slot = -1; blank.job.app_off = -1; blank.job.iou_off = -1;
    for (int i = 0; i < nops; ++i) {
        e->slot_chain[cb->ops[i].slot] = (int)tmp.size();
        tmp.push_back(blank);
        tmp.back().op = cb->ops[i];
    }
    for (int q = 0; q < njobs; ++q) {
        const int slot = ca->h_jobs[q].slot, at = e->slot_chain[slot];
        if (at >= 0) tmp[at].job = ca->h_jobs[q];
        else { tmp.push_back(blank); tmp.back().job = ca->h_jobs[q]; }
    }
    for (int i = 0; i < nops; ++i) e->slot_chain[cb->ops[i].slot] = -1;
    const int n = (int)tmp.size();
    VC_CHECK((size_t)n * sizeof(TrackChainRec) <= e->stage_cap, VC_ERR_CAPACITY, "tracker staging block too small");
    TrackChainRec* h = (TrackChainRec*)e->h_stage2;
    if (n) memcpy(h, tmp.data(), (size_t)n * sizeof(TrackChainRec));
    *d_recs = (const TrackChainRec*)e->hd_stage2;
    *n_out = n;
    return VC_OK;
}

// Launch the pending operations of `cb` (may be null) and the cost jobs of `ca` (may be null) as one kernel, one
// workgroup per touched slot.
int track_launch(vc_engine* e, const StepCtx* cb, const float* feat_b, const StepCtx* ca, const float* feat_a) {
    const TrackChainRec* d_recs; int nch;
    VC_TRY(build_chain_recs(e, cb, ca, &d_recs, &nch));
    if (nch == 0) { e->track_inflight = false; return VC_OK; }
    e->track_seq += 1;
    ProfScope ps(e, VC_PROF_TRACK);
    VC_TRY(launch_track_step(e->pool, d_recs, nch, feat_b, feat_a, e->hd_mean, ca ? ca->d_featrow : nullptr, ca ? ca->d_xyah : nullptr,
                             ca ? ca->d_tlwh : nullptr, e->hd_cost, e->d_track_counter, e->hd_track_flag, e->track_seq, e->stream));
    e->track_inflight = true;
    return VC_OK;
}

// Wait for the last track_launch: poll the completion word the kernel publishes to pinned memory (a stream
// synchronisation costs several times the kernel itself); fall back to the stream if it does not arrive.
int track_wait(vc_engine* e) {
    if (!e->track_inflight) return VC_OK;
    e->track_inflight = false;
    if (!e->profiling) {
        volatile unsigned* flag = (volatile unsigned*)e->h_track_flag;
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0;; ++spin) {
            if (*flag == e->track_seq) { std::atomic_thread_fence(std::memory_order_acquire); return VC_OK; }
            __builtin_ia32_pause();
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    VC_HIP(hipStreamSynchronize(e->stream));
    VC_CHECK(*(volatile unsigned*)e->h_track_flag == e->track_seq, VC_ERR_HIP, "tracker kernel did not publish its completion word");
    return VC_OK;
}

int track_host_b(vc_engine* e, StepCtx& c) {
    const int njobs = (int)c.ids.size();
    std::vector<TrackOpB>& ops = c.ops;              // one per updated / initiated / output-only track
    ops.clear();
    auto add_op = [&](int slot, int kind, const double* z, int gal_pos, int feat_row) {
        TrackOpB op{};
        op.slot = slot; op.kind = kind; op.gal_pos = gal_pos; op.feat_row = feat_row; op.out_row = -1;
        if (z) memcpy(op.z, z, 4 * sizeof(double));
        ops.push_back(op);
    };
    for (int j = 0; j < njobs; ++j) {
        Tracker& tk = *e->trackers[c.ids[j]];
        const Prepared& pr = c.prep[j];
        const int k = (int)pr.conf.size();
        const int nt = (int)tk.tracks.size();
        // scratch reused across trackers and frames (no heap traffic in the per-frame step)
        static thread_local std::vector<int> confirmed, unconfirmed, left, lvl, un_a, cand, un_tracks;
        static thread_local std::vector<std::pair<int, int>> matches;
        static thread_local std::vector<char> matched_track;
        static thread_local std::vector<const double*> rp;
        static thread_local MatchOut mo;
        confirmed.clear(); unconfirmed.clear(); matches.clear();
        for (int t = 0; t < nt; ++t) (tk.tracks[t].state == CONFIRMED ? confirmed : unconfirmed).push_back(t);
        left.resize(k);
        std::iota(left.begin(), left.end(), 0);
        // matching_cascade, sort/linear_assignment.py:124-145
        matched_track.assign(nt, 0);
        int max_tsu = 0;
        for (int t : confirmed) max_tsu = std::max(max_tsu, tk.tracks[t].tsu);
        for (int level = 0; level < tk.p.max_age && level < max_tsu; ++level) {      // levels above the oldest track are empty
            if (left.empty()) break;
            lvl.clear();
            for (int t : confirmed) if (tk.tracks[t].tsu == 1 + level) lvl.push_back(t);
            if (lvl.empty()) continue;
            rp.clear();
            for (int t : lvl) rp.push_back(e->h_cost + c.app_job[j][t]);
            VC_TRY(min_cost_matching(rp, lvl, left, tk.p.max_dist, mo));
            for (auto& m : mo.matches) { matches.push_back(m); matched_track[m.first] = 1; }
            left = mo.un_cols;
        }
        un_a.clear();                                            // set(confirmed) - matched, ascending ("set order")
        for (int t : confirmed) if (!matched_track[t]) un_a.push_back(t);
        // IoU stage, sort/tracker.py:118-127
        cand = unconfirmed; un_tracks.clear();
        for (int t : un_a) (tk.tracks[t].tsu == 1 ? cand : un_tracks).push_back(t);
        {
            rp.clear();
            if (k > 0) for (int t : cand) rp.push_back(e->h_cost + c.iou_job[j][t]);
            else rp.assign(cand.size(), nullptr);
            VC_TRY(min_cost_matching(rp, cand, left, tk.p.max_iou_distance, mo));
        }
        for (auto& m : mo.matches) matches.push_back(m);
        for (int t : mo.un_rows) un_tracks.push_back(t);
        const std::vector<int>& un_dets = mo.un_cols;

        // Track.update, sort/track.py:126-145
        for (auto& m : matches) {
            TrackRec& tr = tk.tracks[m.first];
            const int g = c.det_base[j] + m.second;
            add_op(tr.slot, 1, &c.det_xyah[(size_t)g * 4], tr.gal_head, c.featrow[g]);
            tr.gal_head = (tr.gal_head + 1) % tk.p.nn_budget;
            tr.gal_count = std::min(tr.gal_count + 1, tk.p.nn_budget);
            tr.last_conf = pr.conf[m.second];
            tr.hits += 1; tr.tsu = 0;
            if (tr.state == TENTATIVE && tr.hits >= tk.p.n_init) tr.state = CONFIRMED;
        }
        // Track.mark_missed, sort/track.py:147-153
        for (int t : un_tracks) {
            TrackRec& tr = tk.tracks[t];
            if (tr.state == TENTATIVE) tr.state = DELETED;
            else if (tr.tsu > tk.p.max_age) tr.state = DELETED;
        }
        // _initiate_track, sort/tracker.py:133-139
        for (int d : un_dets) {
            const int new_slot = slot_alloc(e);
            VC_CHECK(new_slot >= 0, VC_ERR_CAPACITY, "track pool exhausted (max_tracks = %d)", e->cfg.max_tracks);
            TrackRec tr{};
            tr.id = tk.next_id++; tr.state = TENTATIVE; tr.hits = 1; tr.age = 1; tr.tsu = 0;
            tr.slot = new_slot;
            const int g = c.det_base[j] + d;
            add_op(tr.slot, 2, &c.det_xyah[(size_t)g * 4], 0, c.featrow[g]);
            tr.gal_head = 1 % tk.p.nn_budget; tr.gal_count = 1;
            tr.last_conf = pr.conf[d];
            tk.tracks.push_back(tr);
        }
        // drop deleted tracks, sort/tracker.py:80
        std::vector<TrackRec> alive;
        for (TrackRec& tr : tk.tracks) {
            if (tr.state == DELETED) slot_free(e, tr.slot);
            else alive.push_back(tr);
        }
        tk.tracks.swap(alive);
    }
    // deep_sort.py:46-58 output eligibility (confirmed, time_since_update <= 1): those tracks' posterior means go back
    c.emit.clear();
    c.mean_offsets.assign(njobs + 1, 0);
    std::vector<int>& op_of_slot = e->slot_chain;    // scratch, -1 outside a call
    const size_t n_real_ops = ops.size();
    for (size_t i = 0; i < n_real_ops; ++i) op_of_slot[ops[i].slot] = (int)i;
    int n_out = 0;
    for (int j = 0; j < njobs; ++j) {
        c.mean_offsets[j] = n_out;
        for (const TrackRec& tr : e->trackers[c.ids[j]]->tracks) {
            if (!c.all_means && (tr.state != CONFIRMED || tr.tsu > 1)) continue;
            if (tr.state == CONFIRMED && tr.tsu <= 1) c.emit.push_back(StepCtx::Emit{n_out, tr.id, j < (int)c.labels.size() ? c.labels[j] : 0});
            const int at = op_of_slot[tr.slot];
            if (at < 0) { add_op(tr.slot, 0, nullptr, 0, -1); ops.back().out_row = n_out; }
            else ops[at].out_row = n_out;
            ++n_out;
        }
    }
    for (size_t i = 0; i < n_real_ops; ++i) op_of_slot[ops[i].slot] = -1;
    c.mean_offsets[njobs] = n_out;
    return VC_OK;      // c.ops are pending: e->h_mean / c.emit are valid after the launch that carries them has completed
}

// Blocking form used by the per-frame entry points: cost jobs, host matching, then the operations in a second launch.
int track_step_blocking(vc_engine* e, StepCtx& c, const float* d_feat) {
    VC_TRY(track_prepare_a(e, c));
    VC_TRY(track_launch(e, nullptr, nullptr, &c, d_feat));
    VC_TRY(track_wait(e));
    VC_TRY(track_host_b(e, c));
    VC_TRY(track_launch(e, &c, d_feat, nullptr, nullptr));
    VC_TRY(track_wait(e));
    return VC_OK;
}

// deep_sort.py:46-58: confirmed tracks seen within one frame -> int rows [x1,y1,x2,y2,id,label] (box = Kalman posterior, Q7)
void emit_rows(const StepCtx& c, const double* means, std::vector<int64_t>& rows6) {
    for (const StepCtx::Emit& em : c.emit) {
        const double* m = means + (size_t)em.row * 8;
        const double w = m[2] * m[3], h = m[3];                         // sort/track.py:82-96 to_tlwh
        const double x = m[0] - w / 2, y = m[1] - h / 2;
        rows6.push_back(std::max((int64_t)x, (int64_t)0));              // deep_sort.py:97-108 int() + clamp
        rows6.push_back(std::max((int64_t)y, (int64_t)0));
        rows6.push_back(std::min((int64_t)(x + w), (int64_t)c.W - 1));
        rows6.push_back(std::min((int64_t)(y + h), (int64_t)c.H - 1));
        rows6.push_back(em.id);
        rows6.push_back(em.label);
    }
}

// DeepSort.update minus the embedding: confidence filter, tlwh, DeepSORT NMS -> detections in pick order
void prepare_dets(const double* xyxy, const double* conf, const int* rows, int k, const vc_tracker_params& p, Prepared& out) {
    std::vector<double> tl, cf;
    std::vector<int> fr;
    for (int i = 0; i < k; ++i) {
        if (!(conf[i] > p.min_confidence)) continue;                 // deep_sort.py:31 (features were computed for all, Q5)
        const double* b = xyxy + (size_t)i * 4;
        const double w = b[2] - b[0], h = b[3] - b[1];               // _xyxy_to_xywh :78-87
        const double cx = b[0] + w / 2, cy = b[1] + h / 2;
        tl.push_back(cx - w / 2.); tl.push_back(cy - h / 2.); tl.push_back(w); tl.push_back(h);   // _xywh_to_tlwh :68-75
        cf.push_back(conf[i]);
        fr.push_back(rows[i]);
    }
    std::vector<int> keep;
    dsort_nms(tl.data(), cf.data(), (int)cf.size(), p.nms_max_overlap, keep);
    out.tlwh.clear(); out.conf.clear(); out.feat_rows.clear();
    for (int i : keep) {
        for (int c = 0; c < 4; ++c) out.tlwh.push_back(tl[(size_t)i * 4 + c]);
        out.conf.push_back(cf[i]);
        out.feat_rows.push_back(fr[i]);
    }
}

static void xyxy_to_cxcywh(const double* b, double* o) {
    const double w = b[2] - b[0], h = b[3] - b[1];
    o[0] = b[0] + w / 2; o[1] = b[1] + h / 2; o[2] = w; o[3] = h;
}

static void crop_corners_i(const double* b, int W, int H, int* c) {  // deep_sort.py:89-95
    c[0] = std::max((int)(b[0] - b[2] / 2), 0); c[2] = std::min((int)(b[0] + b[2] / 2), W - 1);
    c[1] = std::max((int)(b[1] - b[3] / 2), 0); c[3] = std::min((int)(b[1] + b[3] / 2), H - 1);
}

// Build the step context of one frame: per tracker the prepared (filtered, NMS'ed) detections.
void build_ctx(vc_engine* e, StepCtx& c, int H, int W, const std::vector<int>& tracker_ids, const std::vector<int>& labels,
               const std::vector<std::vector<int>>& groups, const double* xyxy, const double* conf, int feat_row0) {
    const int nj = (int)tracker_ids.size();
    c.ids = tracker_ids; c.labels = labels; c.H = H; c.W = W; c.all_means = false;
    c.prep.assign(nj, Prepared{});
    for (int j = 0; j < nj; ++j) {
        const auto& g = groups[j];
        std::vector<double> bx(g.size() * 4), cf(g.size());
        std::vector<int> rows(g.size());
        for (size_t i = 0; i < g.size(); ++i) {
            memcpy(&bx[i * 4], xyxy + (size_t)g[i] * 4, 4 * sizeof(double));
            cf[i] = conf[g[i]];
            rows[i] = feat_row0 + g[i];
        }
        prepare_dets(bx.data(), cf.data(), rows.data(), (int)g.size(), e->trackers[tracker_ids[j]]->p, c.prep[j]);
    }
}

// Shared by vc_deepsort_update / vc_videotracker_run: one frame already on the device, blocking.
// groups: per tracker the indices (into xyxy/conf) of its boxes.  Output rows [x1,y1,x2,y2,id,label].
int frame_track(vc_engine* e, const uint8_t* d_frame_base, int frame_index, int H, int W, const std::vector<int>& tracker_ids,
                const std::vector<int>& labels, const std::vector<std::vector<int>>& groups, const double* xyxy, const double* conf,
                int n, std::vector<int64_t>& rows6) {
    VC_CHECK(n <= e->cfg.max_crops, VC_ERR_CAPACITY, "%d crops exceed max_crops %d", n, e->cfg.max_crops);
    for (int i = 0; i < n; ++i) {
        double c[4]; int q[4];
        xyxy_to_cxcywh(xyxy + (size_t)i * 4, c);
        crop_corners_i(c, W, H, q);
        VC_CHECK(q[2] > q[0] && q[3] > q[1], VC_ERR_ARG, "box %d gives an empty crop (the reference's cv2.resize raises here)", i);
        int* h = e->h_crops + (size_t)i * 5;
        h[0] = frame_index; h[1] = q[0]; h[2] = q[1]; h[3] = q[2]; h[4] = q[3];
    }
    VC_HIP(hipMemcpyAsync(e->d_crops, e->h_crops, (size_t)n * 5 * sizeof(int), hipMemcpyHostToDevice, e->stream));
    VC_TRY(run_reid_dev(e, d_frame_base, H, W, n));
    StepCtx c;
    build_ctx(e, c, H, W, tracker_ids, labels, groups, xyxy, conf, 0);
    VC_TRY(track_step_blocking(e, c, e->d_feat));
    emit_rows(c, e->h_mean, rows6);
    return VC_OK;
}

}  // namespace vc

// ================================================================================================ C ABI
using namespace vc;

extern "C" {

int vc_tracker_create(vc_engine* e, const vc_tracker_params* p, int* id) {
    VC_CHECK(e && p && id, VC_ERR_ARG, "null argument");
    VC_CHECK(p->nn_budget >= 1 && p->nn_budget <= e->cfg.nn_budget_cap, VC_ERR_CAPACITY,
             "nn_budget %d outside [1, nn_budget_cap=%d] (an unbounded budget is not supported)", p->nn_budget, e->cfg.nn_budget_cap);
    VC_CHECK(p->max_age >= 1 && p->n_init >= 1, VC_ERR_ARG, "max_age and n_init must be >= 1");
    std::unique_ptr<Tracker> t(new Tracker());
    t->p = *p;
    e->trackers.push_back(std::move(t));
    *id = (int)e->trackers.size() - 1;
    return VC_OK;
}

int vc_tracker_reset(vc_engine* e, int id) {
    if (e) async_wait_all(e);            // tracker state belongs to the worker thread while asynchronous batches run
    VC_CHECK(e && id >= 0 && id < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id");
    Tracker& tk = *e->trackers[id];
    for (const TrackRec& t : tk.tracks) slot_free(e, t.slot);
    tk.tracks.clear();
    tk.next_id = 1;
    return VC_OK;
}

int vc_tracker_step(vc_engine* e, int id, const double* tlwh, const double* conf, const float* feat, int k) {
    if (e) async_wait_all(e);            // tracker state belongs to the worker thread while asynchronous batches run
    VC_CHECK(e && id >= 0 && id < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id");
    VC_CHECK(k == 0 || (tlwh && conf && feat), VC_ERR_ARG, "null argument");
    VC_CHECK(k <= e->det_cap, VC_ERR_CAPACITY, "%d detections exceed capacity %d", k, e->det_cap);
    VC_HIP(hipSetDevice(e->cfg.device));
    if (k > 0) VC_HIP(hipMemcpyAsync(e->d_feat_in, feat, (size_t)k * VC_FEAT_DIM * sizeof(float), hipMemcpyHostToDevice, e->stream));
    StepCtx c;
    c.ids = {id}; c.labels = {0}; c.all_means = false;
    c.prep.assign(1, Prepared{});
    c.prep[0].tlwh.assign(tlwh, tlwh + (size_t)k * 4);
    c.prep[0].conf.assign(conf, conf + k);
    c.prep[0].feat_rows.resize(k);
    std::iota(c.prep[0].feat_rows.begin(), c.prep[0].feat_rows.end(), 0);
    VC_TRY(track_step_blocking(e, c, e->d_feat_in));
    return VC_OK;
}

int vc_tracker_count(vc_engine* e, int id, int* n) {
    if (e) async_wait_all(e);            // tracker state belongs to the worker thread while asynchronous batches run
    VC_CHECK(e && n && id >= 0 && id < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id");
    *n = (int)e->trackers[id]->tracks.size();
    return VC_OK;
}

int vc_tracker_state(vc_engine* e, int id, int cap, int64_t* ids, int* state, int* hits, int* age, int* tsu, double* mean8,
                     double* cov64, int* gallery_count) {
    if (e) async_wait_all(e);
    VC_CHECK(e && id >= 0 && id < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id");
    const Tracker& tk = *e->trackers[id];
    VC_CHECK((int)tk.tracks.size() <= cap, VC_ERR_CAPACITY, "need room for %zu tracks", tk.tracks.size());
    VC_HIP(hipStreamSynchronize(e->stream));
    for (size_t t = 0; t < tk.tracks.size(); ++t) {
        const TrackRec& tr = tk.tracks[t];
        if (ids) ids[t] = tr.id;
        if (state) state[t] = tr.state;
        if (hits) hits[t] = tr.hits;
        if (age) age[t] = tr.age;
        if (tsu) tsu[t] = tr.tsu;
        if (gallery_count) gallery_count[t] = tr.state == CONFIRMED ? tr.gal_count : 0;
        if (mean8) VC_HIP(hipMemcpy(mean8 + t * 8, e->pool.mean + (size_t)tr.slot * 8, 8 * sizeof(double), hipMemcpyDeviceToHost));
        if (cov64) VC_HIP(hipMemcpy(cov64 + t * 64, e->pool.cov + (size_t)tr.slot * 64, 64 * sizeof(double), hipMemcpyDeviceToHost));
    }
    return VC_OK;
}

// ---- tracker snapshot / restore (tracker state for stream migration) -----------------------------------------
// Everything Tracker.predict/update reads: parameters, id counter, and per track the FSM counters, the Kalman mean / covariance
// (fp64, bit for bit) and the valid rows of the appearance gallery ring.  Little-endian, packed; restoring on another engine (or
// GPU) and feeding the same detections continues the stream with identical ids and states (tests/test_gpu_tracker.py).
namespace {
struct SnapHeader { char magic[8]; int32_t feat_dim, n_tracks; int64_t next_id; double max_dist, min_confidence, nms_max_overlap, max_iou_distance; int32_t max_age, n_init, nn_budget, pad; };
struct SnapTrack { int64_t id; int32_t state, hits, age, tsu, gal_count, gal_head; double last_conf; double mean[8]; double cov[64]; };
const char kSnapMagic[8] = {'V', 'C', 'T', 'R', 'K', '0', '1', 0};
}  // namespace

int vc_tracker_snapshot(vc_engine* e, int id, void* buf, size_t cap, size_t* size) {
    if (e) async_wait_all(e);
    VC_CHECK(e && size && id >= 0 && id < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id");
    const Tracker& tk = *e->trackers[id];
    size_t need = sizeof(SnapHeader);
    for (const TrackRec& tr : tk.tracks) need += sizeof(SnapTrack) + (size_t)std::min(tr.gal_count, tk.p.nn_budget) * VC_FEAT_DIM * sizeof(float);
    *size = need;
    if (!buf) return VC_OK;                       // size query
    VC_CHECK(cap >= need, VC_ERR_CAPACITY, "snapshot needs %zu bytes", need);
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_HIP(hipStreamSynchronize(e->stream));
    char* o = (char*)buf;
    SnapHeader h{};
    memcpy(h.magic, kSnapMagic, 8);
    h.feat_dim = VC_FEAT_DIM; h.n_tracks = (int32_t)tk.tracks.size(); h.next_id = tk.next_id;
    h.max_dist = tk.p.max_dist; h.min_confidence = tk.p.min_confidence; h.nms_max_overlap = tk.p.nms_max_overlap;
    h.max_iou_distance = tk.p.max_iou_distance; h.max_age = tk.p.max_age; h.n_init = tk.p.n_init; h.nn_budget = tk.p.nn_budget;
    memcpy(o, &h, sizeof(h)); o += sizeof(h);
    for (const TrackRec& tr : tk.tracks) {
        SnapTrack t{};
        t.id = tr.id; t.state = tr.state; t.hits = tr.hits; t.age = tr.age; t.tsu = tr.tsu; t.gal_count = tr.gal_count; t.gal_head = tr.gal_head;
        t.last_conf = tr.last_conf;
        VC_HIP(hipMemcpy(t.mean, e->pool.mean + (size_t)tr.slot * 8, sizeof(t.mean), hipMemcpyDeviceToHost));
        VC_HIP(hipMemcpy(t.cov, e->pool.cov + (size_t)tr.slot * 64, sizeof(t.cov), hipMemcpyDeviceToHost));
        memcpy(o, &t, sizeof(t)); o += sizeof(t);
        const size_t rows = (size_t)std::min(tr.gal_count, tk.p.nn_budget);
        if (rows) VC_HIP(hipMemcpy(o, e->pool.gallery + (size_t)tr.slot * e->pool.budget_cap * VC_FEAT_DIM, rows * VC_FEAT_DIM * sizeof(float), hipMemcpyDeviceToHost));
        o += rows * VC_FEAT_DIM * sizeof(float);
    }
    return VC_OK;
}

int vc_tracker_restore(vc_engine* e, int id, const void* buf, size_t size) {
    if (e) async_wait_all(e);
    VC_CHECK(e && buf && id >= 0 && id < (int)e->trackers.size(), VC_ERR_NOTFOUND, "bad tracker id");
    VC_CHECK(size >= sizeof(SnapHeader), VC_ERR_ARG, "snapshot truncated");
    const char* in = (const char*)buf;
    SnapHeader h;
    memcpy(&h, in, sizeof(h)); in += sizeof(h);
    VC_CHECK(memcmp(h.magic, kSnapMagic, 8) == 0 && h.feat_dim == VC_FEAT_DIM && h.n_tracks >= 0, VC_ERR_ARG, "not a tracker snapshot");
    VC_CHECK(h.nn_budget >= 1 && h.nn_budget <= e->cfg.nn_budget_cap, VC_ERR_CAPACITY, "snapshot nn_budget %d exceeds nn_budget_cap %d", h.nn_budget, e->cfg.nn_budget_cap);
    // validate the whole blob before touching the tracker
    {
        const char* q = in;
        for (int i = 0; i < h.n_tracks; ++i) {
            VC_CHECK((size_t)(q - (const char*)buf) + sizeof(SnapTrack) <= size, VC_ERR_ARG, "snapshot truncated");
            SnapTrack t;
            memcpy(&t, q, sizeof(t)); q += sizeof(t);
            VC_CHECK(t.gal_count >= 0 && t.gal_head >= 0 && t.gal_head < h.nn_budget, VC_ERR_ARG, "snapshot corrupt (gallery ring)");
            q += (size_t)std::min(t.gal_count, h.nn_budget) * VC_FEAT_DIM * sizeof(float);
        }
        VC_CHECK((size_t)(q - (const char*)buf) == size, VC_ERR_ARG, "snapshot size mismatch");
    }
    Tracker& tk = *e->trackers[id];
    VC_CHECK((int)e->free_slots.size() + (int)tk.tracks.size() >= h.n_tracks, VC_ERR_CAPACITY, "track pool too small for %d tracks", h.n_tracks);
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_HIP(hipStreamSynchronize(e->stream));
    for (const TrackRec& t : tk.tracks) slot_free(e, t.slot);
    tk.tracks.clear();
    tk.p.max_dist = h.max_dist; tk.p.min_confidence = h.min_confidence; tk.p.nms_max_overlap = h.nms_max_overlap;
    tk.p.max_iou_distance = h.max_iou_distance; tk.p.max_age = h.max_age; tk.p.n_init = h.n_init; tk.p.nn_budget = h.nn_budget;
    tk.next_id = h.next_id;
    for (int i = 0; i < h.n_tracks; ++i) {
        SnapTrack t;
        memcpy(&t, in, sizeof(t)); in += sizeof(t);
        TrackRec tr{};
        tr.id = t.id; tr.state = t.state; tr.hits = t.hits; tr.age = t.age; tr.tsu = t.tsu; tr.gal_count = t.gal_count; tr.gal_head = t.gal_head;
        tr.last_conf = t.last_conf;
        tr.slot = slot_alloc(e);
        VC_HIP(hipMemcpy(e->pool.mean + (size_t)tr.slot * 8, t.mean, sizeof(t.mean), hipMemcpyHostToDevice));
        VC_HIP(hipMemcpy(e->pool.cov + (size_t)tr.slot * 64, t.cov, sizeof(t.cov), hipMemcpyHostToDevice));
        const size_t rows = (size_t)std::min(t.gal_count, h.nn_budget);
        if (rows) VC_HIP(hipMemcpy(e->pool.gallery + (size_t)tr.slot * e->pool.budget_cap * VC_FEAT_DIM, in, rows * VC_FEAT_DIM * sizeof(float), hipMemcpyHostToDevice));
        in += rows * VC_FEAT_DIM * sizeof(float);
        tk.tracks.push_back(tr);
    }
    return VC_OK;
}

int vc_deepsort_update(vc_engine* e, int id, const uint8_t* bgr, int h, int w, const double* bbox_xyxy, const double* conf, int k,
                       int64_t* out_rows7, int cap_rows, int* out_m) {
    if (e) async_wait_all(e);
    VC_CHECK(e && bgr && out_m && id >= 0 && id < (int)e->trackers.size(), VC_ERR_ARG, "bad argument");
    VC_CHECK(k >= 1 && bbox_xyxy && conf, VC_ERR_ARG, "DeepSort.update needs at least one box (the reference only calls it then)");
    VC_HIP(hipSetDevice(e->cfg.device));
    const size_t bytes = (size_t)h * w * 3;
    VC_CHECK(bytes <= e->d_frames_bytes, VC_ERR_CAPACITY, "frame exceeds the staging buffer");
    VC_HIP(hipMemcpyAsync(e->d_frames, bgr, bytes, hipMemcpyHostToDevice, e->stream));
    std::vector<int> all(k);
    std::iota(all.begin(), all.end(), 0);
    std::vector<int64_t> rows6;
    VC_TRY(frame_track(e, e->d_frames, 0, h, w, {id}, {0}, {all}, bbox_xyxy, conf, k, rows6));
    const int m = (int)(rows6.size() / 6);
    VC_CHECK(m <= cap_rows, VC_ERR_CAPACITY, "need room for %d rows", m);
    for (int i = 0; i < m; ++i) {
        for (int c = 0; c < 5; ++c) out_rows7[i * 7 + c] = rows6[(size_t)i * 6 + c];
        out_rows7[i * 7 + 5] = -1;       // track_feat slot: features of confirmed tracks were just cleared (quirk Q7)
        out_rows7[i * 7 + 6] = 0;        // int(confidence) with confidence in (0, 1)
    }
    *out_m = m;
    return VC_OK;
}

int vc_videotracker_run(vc_engine* e, const int* trackers, int num_classes, const uint8_t* bgr, int h, int w, const double* boxes_xywh,
                        const int64_t* labels, const double* scores, int n, int64_t* out_rows6, int cap_rows, int* out_m) {
    if (e) async_wait_all(e);
    VC_CHECK(e && trackers && bgr && out_m, VC_ERR_ARG, "null argument");
    VC_CHECK(n >= 1 && boxes_xywh && labels && scores, VC_ERR_ARG, "VideoTracker.run needs at least one box (quirk Q1)");
    VC_HIP(hipSetDevice(e->cfg.device));
    const size_t bytes = (size_t)h * w * 3;
    VC_CHECK(bytes <= e->d_frames_bytes, VC_ERR_CAPACITY, "frame exceeds the staging buffer");
    VC_HIP(hipMemcpyAsync(e->d_frames, bgr, bytes, hipMemcpyHostToDevice, e->stream));
    std::vector<double> xyxy((size_t)n * 4);
    for (int i = 0; i < n; ++i) {                               // modules/track.py:39-41
        xyxy[i * 4] = boxes_xywh[i * 4]; xyxy[i * 4 + 1] = boxes_xywh[i * 4 + 1];
        xyxy[i * 4 + 2] = boxes_xywh[i * 4 + 2] + boxes_xywh[i * 4]; xyxy[i * 4 + 3] = boxes_xywh[i * 4 + 3] + boxes_xywh[i * 4 + 1];
    }
    std::vector<int> ids, labs;
    std::vector<std::vector<int>> groups;
    for (int c = 0; c < num_classes; ++c) {                     // modules/track.py:50-59: classes without boxes are not stepped
        std::vector<int> g;
        for (int i = 0; i < n; ++i) if (labels[i] == c) g.push_back(i);
        if (g.empty()) continue;
        ids.push_back(trackers[c]); labs.push_back(c); groups.push_back(g);
    }
    // boxes of classes outside [0, num_classes) are ignored exactly like the reference's mask loop; embed only used boxes
    std::vector<int64_t> rows6;
    if (!ids.empty()) {
        std::vector<double> used_xyxy, used_conf;
        std::vector<std::vector<int>> g2(groups.size());
        for (size_t j = 0; j < groups.size(); ++j)
            for (int i : groups[j]) {
                g2[j].push_back((int)used_conf.size());
                for (int c = 0; c < 4; ++c) used_xyxy.push_back(xyxy[(size_t)i * 4 + c]);
                used_conf.push_back(scores[i]);
            }
        VC_TRY(frame_track(e, e->d_frames, 0, h, w, ids, labs, g2, used_xyxy.data(), used_conf.data(), (int)used_conf.size(), rows6));
    }
    const int m = (int)(rows6.size() / 6);
    VC_CHECK(m <= cap_rows, VC_ERR_CAPACITY, "need room for %d rows", m);
    memcpy(out_rows6, rows6.data(), rows6.size() * sizeof(int64_t));
    *out_m = m;
    return VC_OK;
}

// ---- single-function entry points (parity tests) --------------------------------------------------------------
static int with_pool(int n, TrackPool& tp, std::vector<void*>& allocs, int** d_slots) {
    tp.max_tracks = n; tp.budget_cap = 1;
    VC_HIP(hipMalloc((void**)&tp.mean, (size_t)n * 8 * sizeof(double))); allocs.push_back(tp.mean);
    VC_HIP(hipMalloc((void**)&tp.cov, (size_t)n * 64 * sizeof(double))); allocs.push_back(tp.cov);
    tp.gallery = nullptr;
    std::vector<int> s(n);
    std::iota(s.begin(), s.end(), 0);
    VC_HIP(hipMalloc((void**)d_slots, (size_t)n * sizeof(int))); allocs.push_back(*d_slots);
    VC_HIP(hipMemcpy(*d_slots, s.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    return VC_OK;
}
static void free_all(std::vector<void*>& a) { for (void* p : a) hipFree(p); a.clear(); }

#define VC_HOST_FINISH(st)                                                                                              \
    if ((st) == VC_OK && hipDeviceSynchronize() != hipSuccess) { set_error("kernel failed: %s", hipGetErrorString(hipGetLastError())); (st) = VC_ERR_HIP; }

int vc_kalman_initiate_host(const double* xyah, int n, double* mean8, double* cov64) {
    VC_CHECK(xyah && mean8 && cov64 && n > 0, VC_ERR_ARG, "bad argument");
    TrackPool tp{}; std::vector<void*> al; int* ds = nullptr; double* dz = nullptr;
    int st = with_pool(n, tp, al, &ds);
    if (st == VC_OK && hipMalloc((void**)&dz, (size_t)n * 32) != hipSuccess) { set_error("alloc"); st = VC_ERR_HIP; } else al.push_back(dz);
    if (st == VC_OK && hipMemcpy(dz, xyah, (size_t)n * 32, hipMemcpyHostToDevice) != hipSuccess) { set_error("copy"); st = VC_ERR_HIP; }
    if (st == VC_OK) st = launch_kalman_initiate(tp, ds, dz, n, nullptr);
    VC_HOST_FINISH(st);
    if (st == VC_OK) { hipMemcpy(mean8, tp.mean, (size_t)n * 64, hipMemcpyDeviceToHost); hipMemcpy(cov64, tp.cov, (size_t)n * 512, hipMemcpyDeviceToHost); }
    free_all(al);
    return st;
}

static int kalman_inout(double* mean8, double* cov64, const double* z4, int n, int which) {
    VC_CHECK(mean8 && cov64 && n > 0, VC_ERR_ARG, "bad argument");
    TrackPool tp{}; std::vector<void*> al; int* ds = nullptr; double* dz = nullptr;
    int st = with_pool(n, tp, al, &ds);
    if (st == VC_OK) { hipMemcpy(tp.mean, mean8, (size_t)n * 64, hipMemcpyHostToDevice); hipMemcpy(tp.cov, cov64, (size_t)n * 512, hipMemcpyHostToDevice); }
    if (st == VC_OK && z4) {
        if (hipMalloc((void**)&dz, (size_t)n * 32) != hipSuccess) { set_error("alloc"); st = VC_ERR_HIP; } else { al.push_back(dz); hipMemcpy(dz, z4, (size_t)n * 32, hipMemcpyHostToDevice); }
    }
    if (st == VC_OK) st = which == 0 ? launch_kalman_predict(tp, ds, n, nullptr) : launch_kalman_update(tp, ds, dz, n, nullptr);
    VC_HOST_FINISH(st);
    if (st == VC_OK) { hipMemcpy(mean8, tp.mean, (size_t)n * 64, hipMemcpyDeviceToHost); hipMemcpy(cov64, tp.cov, (size_t)n * 512, hipMemcpyDeviceToHost); }
    free_all(al);
    return st;
}
int vc_kalman_predict_host(double* mean8, double* cov64, int n) { return kalman_inout(mean8, cov64, nullptr, n, 0); }
int vc_kalman_update_host(double* mean8, double* cov64, const double* z4, int n) {
    VC_CHECK(z4, VC_ERR_ARG, "null measurement");
    return kalman_inout(mean8, cov64, z4, n, 1);
}

// gating distance of ONE track against n_meas measurements, through the appearance-cost kernel with an empty
// gallery: returns the squared Mahalanobis distances reconstructed from the gate (exact values via a side channel)
int vc_kalman_gating_host(const double* mean8, const double* cov64, const double* z4, int n_meas, double* out) {
    VC_CHECK(mean8 && cov64 && z4 && out && n_meas > 0, VC_ERR_ARG, "bad argument");
    // Reuse the device code path: a 1-track pool, gallery of one zero... the kernel only exposes the gated cost, so
    // the distances themselves are produced by a dedicated tiny launch below.
    TrackPool tp{}; std::vector<void*> al; int* ds = nullptr; double *dz = nullptr, *dout = nullptr;
    int st = with_pool(1, tp, al, &ds);
    if (st == VC_OK) { hipMemcpy(tp.mean, mean8, 64, hipMemcpyHostToDevice); hipMemcpy(tp.cov, cov64, 512, hipMemcpyHostToDevice); }
    if (st == VC_OK && (hipMalloc((void**)&dz, (size_t)n_meas * 32) != hipSuccess || hipMalloc((void**)&dout, (size_t)n_meas * 8) != hipSuccess)) { set_error("alloc"); st = VC_ERR_HIP; }
    if (dz) al.push_back(dz);
    if (dout) al.push_back(dout);
    if (st == VC_OK) { hipMemcpy(dz, z4, (size_t)n_meas * 32, hipMemcpyHostToDevice); st = launch_gating_values(tp, 0, dz, n_meas, dout, nullptr); }
    VC_HOST_FINISH(st);
    if (st == VC_OK) hipMemcpy(out, dout, (size_t)n_meas * 8, hipMemcpyDeviceToHost);
    free_all(al);
    return st;
}

int vc_iou_cost_host(const double* track_tlwh, int t, const double* det_tlwh, int d, double* out_iou) {
    VC_CHECK(track_tlwh && det_tlwh && out_iou && t > 0 && d > 0, VC_ERR_ARG, "bad argument");
    // tracks are given as boxes: build means (cx, cy, a, h) whose to_tlwh() reproduces them is lossy, so the kernel is
    // driven through its box-level twin
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    std::vector<void*> al;
    int st = VC_OK;
    if (hipMalloc((void**)&da, (size_t)t * 32) != hipSuccess || hipMalloc((void**)&db, (size_t)d * 32) != hipSuccess ||
        hipMalloc((void**)&dout, (size_t)t * d * 8) != hipSuccess) { set_error("alloc"); st = VC_ERR_HIP; }
    if (da) al.push_back(da);
    if (db) al.push_back(db);
    if (dout) al.push_back(dout);
    if (st == VC_OK) { hipMemcpy(da, track_tlwh, (size_t)t * 32, hipMemcpyHostToDevice); hipMemcpy(db, det_tlwh, (size_t)d * 32, hipMemcpyHostToDevice);
                       st = launch_iou_boxes(da, t, db, d, dout, nullptr); }
    VC_HOST_FINISH(st);
    if (st == VC_OK) hipMemcpy(out_iou, dout, (size_t)t * d * 8, hipMemcpyDeviceToHost);
    free_all(al);
    return st;
}

int vc_cosine_cost_host(const float* gallery, const int* gal_count, int t, int s_cap, const float* feat, int d, double* out) {
    VC_CHECK(gallery && gal_count && feat && out && t > 0 && d > 0 && s_cap > 0, VC_ERR_ARG, "bad argument");
    TrackPool tp{}; std::vector<void*> al; int* ds = nullptr;
    int st = with_pool(t, tp, al, &ds);
    tp.budget_cap = s_cap;
    float* dfeat = nullptr; double *dz = nullptr, *dout = nullptr; CostJob* dj = nullptr; int* drow = nullptr;
    if (st == VC_OK && (hipMalloc((void**)&tp.gallery, (size_t)t * s_cap * VC_FEAT_DIM * 4) != hipSuccess || hipMalloc((void**)&dfeat, (size_t)d * VC_FEAT_DIM * 4) != hipSuccess ||
                        hipMalloc((void**)&dz, (size_t)d * 32) != hipSuccess || hipMalloc((void**)&dout, (size_t)t * d * 8) != hipSuccess ||
                        hipMalloc((void**)&dj, (size_t)t * sizeof(CostJob)) != hipSuccess || hipMalloc((void**)&drow, (size_t)d * 4) != hipSuccess)) { set_error("alloc"); st = VC_ERR_HIP; }
    for (void* p : {(void*)tp.gallery, (void*)dfeat, (void*)dz, (void*)dout, (void*)dj, (void*)drow}) if (p) al.push_back(p);
    if (st == VC_OK) {
        // a wide-open gate: identity covariance scaled up, measurement = mean
        std::vector<double> mean((size_t)t * 8, 0.0), cov((size_t)t * 64, 0.0), z((size_t)d * 4, 0.0);
        for (int i = 0; i < t; ++i) { mean[i * 8 + 3] = 100.0; for (int k = 0; k < 8; ++k) cov[(size_t)i * 64 + k * 9] = 1e6; }
        for (int i = 0; i < d; ++i) z[i * 4 + 3] = 100.0;
        std::vector<CostJob> jobs(t);
        std::vector<int> rows(d);
        std::iota(rows.begin(), rows.end(), 0);
        for (int i = 0; i < t; ++i) jobs[i] = CostJob{i, gal_count[i], 0, d, i * d, 1};
        hipMemcpy(tp.mean, mean.data(), mean.size() * 8, hipMemcpyHostToDevice); hipMemcpy(tp.cov, cov.data(), cov.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dfeat, feat, (size_t)d * VC_FEAT_DIM * 4, hipMemcpyHostToDevice);
        {   // samples enter the device gallery the way the tracker stores them (normalised rows)
            float* draw = nullptr; int* dsps = nullptr;
            std::vector<int> sps;
            for (int i = 0; i < t; ++i) for (int q = 0; q < gal_count[i]; ++q) { sps.push_back(i); sps.push_back(q); sps.push_back(i * s_cap + q); }
            if (hipMalloc((void**)&draw, (size_t)t * s_cap * VC_FEAT_DIM * 4) == hipSuccess && hipMalloc((void**)&dsps, sps.size() * 4 + 16) == hipSuccess) {
                al.push_back(draw); al.push_back(dsps);
                hipMemcpy(draw, gallery, (size_t)t * s_cap * VC_FEAT_DIM * 4, hipMemcpyHostToDevice);
                hipMemcpy(dsps, sps.data(), sps.size() * 4, hipMemcpyHostToDevice);
                st = launch_gallery_write(tp, dsps, (int)sps.size() / 3, draw, nullptr);
            } else { set_error("alloc"); st = VC_ERR_HIP; }
        }
        hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dj, jobs.data(), jobs.size() * sizeof(CostJob), hipMemcpyHostToDevice);
        hipMemcpy(drow, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
        if (st == VC_OK) st = launch_appearance_cost(tp, dj, t, dfeat, drow, dz, dout, nullptr);
    }
    VC_HOST_FINISH(st);
    if (st == VC_OK) hipMemcpy(out, dout, (size_t)t * d * 8, hipMemcpyDeviceToHost);
    free_all(al);
    return st;
}

int vc_dsort_nms_host(const double* tlwh, const double* scores, int n, double max_overlap, int* keep, int* n_keep) {
    VC_CHECK(n_keep && (n == 0 || (tlwh && scores && keep)), VC_ERR_ARG, "null argument");
    std::vector<int> k;
    dsort_nms(tlwh, scores, n, max_overlap, k);
    for (size_t i = 0; i < k.size(); ++i) keep[i] = k[i];
    *n_keep = (int)k.size();
    return VC_OK;
}

int vc_lap_host(const double* cost, int nr, int nc, int* rows, int* cols, int* n_assigned) {
    VC_CHECK(cost && rows && cols && n_assigned && nr >= 0 && nc >= 0, VC_ERR_ARG, "bad argument");
    std::vector<int> r, c;
    VC_TRY(lap_solve(cost, nr, nc, r, c));
    for (size_t i = 0; i < r.size(); ++i) { rows[i] = r[i]; cols[i] = c[i]; }
    *n_assigned = (int)r.size();
    return VC_OK;
}

}  // extern "C"
