// DeepSORT tracker, host side: the tracker state is device-resident and every step runs inside track_batch_kernel
// (track_kernels.hip); the host only prepares a batch's detections -- confidence filter and DeepSORT NMS, which do not depend
// on tracker state -- packs them with the (frame, class) task list into one host-to-device copy, launches ONE kernel per batch
// and reads the rows the kernel wrote into pinned memory.  One tracker per (camera, class) like the reference
// (modules/track.py:16).
//
// Reference (paths relative to /root/reference/networks/deepsort/):
//   deep_sort.py:25-59            DeepSort.update (confidence filter :31, tlwh :68-87, NMS :37-41, predict/update :44-45, rows :46-58)
//   sort/preprocessing.py:6-73    non_max_suppression (quirk Q6)                      -> dsort_nms (host: state-independent)
//   sort/tracker.py, sort/track.py, sort/linear_assignment.py, sort/nn_matching.py, sort/kalman_filter.py, sort/iou_matching.py
//                                 -> track_core.h + track_kernels.hip (device)
//   modules/track.py:30-70        VideoTracker.run (one DeepSORT per class, classes without boxes are not stepped)
#include <algorithm>
#include <cstring>
#include <cmath>
#include <numeric>
#include <climits>
#include <cstdint>
#include <cstdlib>

#include "engine.h"
#include "track_core.h"

namespace vc {

using namespace tc;

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

// ------------------------------------------------------------------------------------------------ pool
static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

int tracker_init_pool(vc_engine* e) {
    const size_t T = e->cfg.max_tracks, S = e->cfg.nn_budget_cap;
    VC_CHECK(T >= 1 && S >= 1, VC_ERR_ARG, "max_tracks and nn_budget_cap must be positive");
    e->max_trackers = e->cfg.max_trackers > 0 ? e->cfg.max_trackers : 256;
    VC_CHECK(e->max_trackers <= 65535, VC_ERR_ARG, "max_trackers %d: a tracker handle addresses at most 65535 slots", e->max_trackers);
    e->list_cap = e->cfg.tracks_per_tracker > 0 ? e->cfg.tracks_per_tracker : 512;
    e->list_cap = (int)std::min<size_t>((size_t)e->list_cap, T);
    e->pool.max_tracks = (int)T; e->pool.budget_cap = (int)S;
    VC_TRY(dev_alloc(e, (void**)&e->pool.mean, T * 8 * sizeof(double)));
    VC_TRY(dev_alloc(e, (void**)&e->pool.cov, T * 64 * sizeof(double)));
    VC_TRY(dev_alloc(e, (void**)&e->pool.gallery, T * S * VC_FEAT_DIM * sizeof(float)));
    VC_TRY(dev_alloc(e, (void**)&e->d_hdrs, (size_t)e->max_trackers * sizeof(TrackerHdr)));
    VC_TRY(dev_alloc(e, (void**)&e->d_lists, (size_t)e->max_trackers * e->list_cap * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&e->d_recs, T * sizeof(TrackRecD)));
    VC_TRY(dev_alloc(e, (void**)&e->d_free, 64));
    VC_TRY(dev_alloc(e, (void**)&e->d_free_stack, T * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&e->d_freed, T * sizeof(int)));
    VC_HIP(hipMemset(e->d_hdrs, 0, (size_t)e->max_trackers * sizeof(TrackerHdr)));
    std::vector<int> st(T);
    for (size_t i = 0; i < T; ++i) st[i] = (int)(T - 1 - i);
    VC_HIP(hipMemcpy(e->d_free_stack, st.data(), T * sizeof(int), hipMemcpyHostToDevice));
    const int ctl[2] = {(int)T, 0};
    VC_HIP(hipMemcpy(e->d_free, ctl, sizeof(ctl), hipMemcpyHostToDevice));
    // hoisted appearance dots (track_kernels.hip): table arena (grown on demand by track_enqueue, up to VC_DOT_ARENA_MB /
    // vc_engine_set_option "dot_arena_mb": 1 GB by default; 0 forces the in-walk appearance rows), row bookkeeping
    e->dot_arena_max_floats = getenv("VC_DOT_ARENA_MB") ? (size_t)atol(getenv("VC_DOT_ARENA_MB")) * 262144 : (size_t)256 << 20;
    e->dot_arena_floats = 0;
    VC_TRY(dev_alloc(e, (void**)&e->d_dot_arena, 16));
    e->row_src_cap = (int)std::min<size_t>(T * S, (size_t)1 << 24);
    VC_TRY(dev_alloc(e, (void**)&e->d_row_src, (size_t)e->row_src_cap * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&e->d_gal_row, T * S * sizeof(int)));
    VC_TRY(dev_alloc(e, (void**)&e->d_dot_ctl, 64));
    e->det_cap = std::max(e->cfg.max_det * 2, 1024);
    VC_TRY(dev_alloc(e, (void**)&e->d_feat_in, (size_t)e->det_cap * VC_FEAT_DIM * sizeof(float)));
    for (TrackStage& s : e->tstage) {
        VC_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        VC_TRY(dev_alloc(e, (void**)&s.d_cursor, 64));
    }
    return VC_OK;
}

// (re)size a staging block; old blocks stay on the engine's free-at-destroy lists (sizes grow geometrically)
static int stage_reserve(vc_engine* e, TrackStage& s, size_t in_bytes, size_t out_bytes) {
    if (in_bytes > s.in_cap) {
        const size_t cap = std::max(in_bytes * 3 / 2, (size_t)1 << 16);
        VC_TRY(host_alloc(e, (void**)&s.h_in, cap));
        VC_TRY(dev_alloc(e, (void**)&s.d_in, cap));
        s.in_cap = cap;
    }
    if (out_bytes > s.out_cap) {
        const size_t cap = std::max(out_bytes * 3 / 2, (size_t)1 << 16);
        VC_TRY(host_alloc(e, (void**)&s.h_out, cap));
        VC_HIP(hipHostGetDevicePointer((void**)&s.hd_out, s.h_out, 0));
        s.out_cap = cap;
    }
    return VC_OK;
}

int track_idle(vc_engine* e) {
    for (TrackStage& s : e->tstage)
        if (s.busy) VC_HIP(hipEventSynchronize(s.done));
    return VC_OK;
}

// output block layout (pinned, written by the kernel)
struct OutLayout { size_t rows, row_off, row_n, ntracks, tT, status, total; };
static OutLayout out_layout(int n_tasks, int rows_cap) {
    OutLayout o;
    size_t p = 0;
    o.rows = p; p = align16(p + (size_t)rows_cap * 6 * sizeof(long long));
    o.row_off = p; p = align16(p + (size_t)n_tasks * 4);
    o.row_n = p; p = align16(p + (size_t)n_tasks * 4);
    o.ntracks = p; p = align16(p + (size_t)n_tasks * 4);
    o.tT = p; p = align16(p + (size_t)n_tasks * 4);
    o.status = p; p = align16(p + 16);
    o.total = p;
    return o;
}

int track_enqueue(vc_engine* e, int st, const std::vector<std::vector<FrameClassDets>>& frames, const float* d_feat, int W, int H,
                  int rows_cap, hipEvent_t wait) {
    TrackStage& s = e->tstage[st];
    VC_CHECK(!s.busy, VC_ERR_STATE, "tracker staging slot %d is still in flight", st);
    // tasks in (frame, class) order -- the order rows are handed back in
    s.tasks.clear(); s.tracker_dets.clear(); s.last_task_of.clear();
    std::vector<const FrameClassDets*> src;
    for (size_t f = 0; f < frames.size(); ++f)
        for (const FrameClassDets& g : frames[f]) {
            VC_CHECK(tracker_ok(e->trackers, g.tracker), VC_ERR_NOTFOUND, "bad tracker id %d", g.tracker);
            s.tasks.push_back(TrackTaskHost{g.tracker, g.label, (int)f, 0, (int)g.dets.conf.size()});
            src.push_back(&g);
        }
    const int n_tasks = (int)s.tasks.size();
    s.n_tasks = n_tasks; s.rows_cap = rows_cap; s.b = (int)frames.size(); s.W = W; s.H = H; s.n_wg = 0; s.n_dets = 0;
    if (n_tasks == 0) { s.busy = true; VC_HIP(hipEventRecord(s.done, e->stream)); return VC_OK; }
    // device order: grouped by tracker, frames ascending inside a group (stable sort of the frame-major list); the detection arrays
    // follow the same order, so every tracker's detections of the batch are one contiguous range
    std::vector<int> order(n_tasks);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return s.tasks[a].tracker < s.tasks[b].tracker; });
    s.dev_index.assign(n_tasks, 0);
    for (int k = 0; k < n_tasks; ++k) s.dev_index[order[k]] = k;
    std::vector<TrackWgPlan> plans;
    int need = 8, n_dets = 0;
    long long mat_need = 0;              // entries of the largest assignment problem a step of this batch is likely to pose (tracks x detections)
    // Do the appearance tables of ALL trackers fit the arena for certain?  Bound from what the host knows: the tracks of the last
    // collected batch (known_tracks) hold at most a full gallery each, and every detection of a batch still in flight (pending_dets)
    // adds at most ONE gallery row -- it either extends one track's ring or starts a track with a single sample (tracker.py:82-91,
    // 133-139).  (Round 2 charged a full gallery per pending detection: at 256 detections per frame that bound was 1.3 GB per tracker,
    // nothing "fitted", and the batch ran on the fallback instance.)  Then the lean kernel instance (no fallback code, 8 waves) runs.
    long long table_floats = 0, table_rows = 0;
    bool all_tables = true;
    for (int k = 0; k < n_tasks;) {
        const int tr = s.tasks[order[k]].tracker;
        int k1 = k, dets = 0, dmax = 0;
        const int det_begin = n_dets;
        while (k1 < n_tasks && s.tasks[order[k1]].tracker == tr) {
            TrackTaskHost& t = s.tasks[order[k1]];
            t.det_off = n_dets; n_dets += t.det_n;
            dets += t.det_n; dmax = std::max(dmax, t.det_n);
            ++k1;
        }
        plans.push_back(TrackWgPlan{tr, k, k1, det_begin, dets, 0, 0, 0});
        {
            const Tracker& tk = *e->trackers[tr];
            const long long old_rows = (long long)tk.known_tracks * std::min(e->pool.budget_cap, tk.p.nn_budget) + tk.pending_dets;
            const long long need_f = (old_rows + dets) * dets;
            if (need_f >= (1ll << 31)) all_tables = false;
            table_floats += need_f; table_rows += old_rows;
        }
        // LDS capacity from the tracker's recent size (the kernel falls back to global-memory work arrays for a larger step)
        need = std::max(need, 2 * e->trackers[tr]->known_tracks + 2 * dmax + 16);
        mat_need = std::max(mat_need, (long long)(e->trackers[tr]->known_tracks + 32) * dmax);
        s.tracker_dets.emplace_back(tr, dets);
        s.last_task_of.push_back(k1 - 1);
        k = k1;
    }
    s.n_dets = n_dets;
    // work-array capacity: 32, else the next multiple of 64 (a power of two, as before round 3, doubled the arrays' LDS for a 310-entry
    // step and left no room for the assignment matrix next to them)
    const int cap = need <= 32 ? 32 : std::min((int)TC_HARD_CAP, (need + 63) / 64 * 64);
    const int n_wg = (int)plans.size();
    s.n_wg = n_wg;
    s.step_cap = cap;
    // input block
    size_t p = 0;
    const size_t o_tasks = p; p = align16(p + (size_t)n_tasks * sizeof(TrackTask));
    const size_t o_plans = p; p = align16(p + (size_t)n_wg * sizeof(TrackWgPlan));
    const size_t o_tlwh = p; p = align16(p + (size_t)n_dets * 32);
    const size_t o_xyah = p; p = align16(p + (size_t)n_dets * 32);
    const size_t o_frow = p; p = align16(p + (size_t)n_dets * 4);
    // The kernel reserves rows in chunks: a workgroup that needs m rows and has fewer left abandons the rest of its chunk and grabs
    // max(m, VC_ROW_CHUNK) fresh ones, so the rows it abandons are always fewer than the rows of the step that follows -- consumption
    // is bounded by twice the rows emitted plus one unfinished chunk per workgroup (`cap + chunk` was a spurious TERR_ROWS
    // for steady steps of 65..127 rows).
    rows_cap = 2 * rows_cap + n_wg * VC_ROW_CHUNK;
    s.rows_cap = rows_cap;
    const OutLayout ol = out_layout(n_tasks, rows_cap);
    VC_TRY(stage_reserve(e, s, p, ol.total));
    TrackTask* ht = (TrackTask*)(s.h_in + o_tasks);
    for (int k = 0; k < n_tasks; ++k) {
        const TrackTaskHost& t = s.tasks[order[k]];
        ht[k] = TrackTask{t.tracker, t.det_off, t.det_n, t.frame, t.label, 0, 0, 0};
    }
    memcpy(s.h_in + o_plans, plans.data(), (size_t)n_wg * sizeof(TrackWgPlan));
    {
        double* tl = (double*)(s.h_in + o_tlwh);
        double* xy = (double*)(s.h_in + o_xyah);
        int* fr = (int*)(s.h_in + o_frow);
        for (int k = 0; k < n_tasks; ++k) {
            const FrameClassDets& c = *src[order[k]];
            int g = s.tasks[order[k]].det_off;
            for (size_t i = 0; i < c.dets.conf.size(); ++i, ++g) {
                const double* t = &c.dets.tlwh[i * 4];
                memcpy(tl + (size_t)g * 4, t, 32);
                double* o = xy + (size_t)g * 4;                           // sort/detection.py:42-50 to_xyah
                o[0] = t[0] + t[2] / 2; o[1] = t[1] + t[3] / 2; o[2] = t[2] / t[3]; o[3] = t[3];
                fr[g] = c.dets.feat_rows[i];
            }
        }
    }
    // buffers of the hoisted appearance dots (engine-wide: the tracker stream runs one batch at a time)
    if ((size_t)n_wg > e->dot_plans_cap || (size_t)n_dets > e->nfeat_cap) {
        VC_TRY(track_idle(e));
        VC_HIP(hipStreamSynchronize(e->stream));
        if ((size_t)n_wg > e->dot_plans_cap) { e->dot_plans_cap = (size_t)n_wg * 2; VC_TRY(dev_alloc(e, (void**)&e->d_dot_plans, e->dot_plans_cap * sizeof(TrackDotPlan))); }
        if ((size_t)n_dets > e->nfeat_cap) {
            e->nfeat_cap = (size_t)n_dets * 3 / 2 + 64;
            VC_TRY(dev_alloc(e, (void**)&e->d_nfeat, e->nfeat_cap * VC_FEAT_DIM * sizeof(float)));
            VC_TRY(dev_alloc(e, (void**)&e->d_det_ss, e->nfeat_cap * sizeof(float)));
        }
    }
    const size_t scratch = (size_t)n_wg * track_scratch_per_wg();
    if (scratch > e->track_scratch_bytes) {
        VC_TRY(track_idle(e));                                                // the old block may still be in use
        const size_t bytes = scratch * 3 / 2;
        VC_TRY(dev_alloc(e, (void**)&e->d_track_scratch, bytes));
        e->track_scratch_bytes = bytes;
    }
    // appearance-table arena: grown to what this batch needs (bounded by dot_arena_max_floats).  When every tracker's table fits for
    // certain the lean kernel instance runs; otherwise the arena still grows towards the bound (it used to stay at its
    // 16-byte initial size then, and the per-tracker fit check on the device sent EVERY tracker to the in-walk appearance rows although
    // most tables would have fitted) and the general instance serves the trackers whose tables fit from it.
    const bool tables_fit = all_tables && table_floats <= (long long)e->dot_arena_max_floats && table_rows <= (long long)e->row_src_cap;
    const size_t want_now = (size_t)std::min<long long>(std::max<long long>(table_floats, 0), (long long)e->dot_arena_max_floats);
    if (want_now > e->dot_arena_floats) {
        VC_TRY(track_idle(e));
        VC_HIP(hipStreamSynchronize(e->stream));
        const size_t want = std::min(e->dot_arena_max_floats, std::max(want_now * 3 / 2, (size_t)1 << 22));
        VC_TRY(dev_realloc(e, (void**)&e->d_dot_arena, want * sizeof(float)));
        e->dot_arena_floats = want;
    }
    for (auto& td : s.tracker_dets) e->trackers[td.first]->pending_dets += td.second;
    hipStream_t ts = e->stream;
    if (wait) VC_HIP(hipStreamWaitEvent(ts, wait, 0));
    VC_HIP(hipMemcpyAsync(s.d_in, s.h_in, p, hipMemcpyHostToDevice, ts));
    VC_HIP(hipMemsetAsync(s.d_cursor, 0, 64, ts));           // [0] row cursor, [4..6] status
    VC_HIP(hipMemsetAsync(e->d_dot_ctl, 0, 64, ts));
    VC_HIP(hipMemsetAsync(e->d_dot_plans, 0, (size_t)n_wg * sizeof(TrackDotPlan), ts));
    TrackBatchArgs a{};
    a.pool = e->pool;
    a.hdrs = e->d_hdrs; a.lists = e->d_lists; a.list_cap = e->list_cap; a.recs = e->d_recs;
    a.free_top = e->d_free; a.freed_count = e->d_free + 1; a.free_stack = e->d_free_stack; a.freed = e->d_freed;
    a.plans = (const TrackWgPlan*)(s.d_in + o_plans); a.tasks = (const TrackTask*)(s.d_in + o_tasks);
    a.det_tlwh = (const double*)(s.d_in + o_tlwh); a.det_xyah = (const double*)(s.d_in + o_xyah); a.det_featrow = (const int*)(s.d_in + o_frow);
    a.feat = d_feat;
    a.dot_plans = e->d_dot_plans; a.dot_arena = e->d_dot_arena; a.dot_arena_floats = (long long)e->dot_arena_floats;
    a.row_src = e->d_row_src; a.row_src_cap = e->row_src_cap; a.gal_row = e->d_gal_row; a.nfeat = e->d_nfeat; a.det_ss = e->d_det_ss;
    a.dot_ctl = e->d_dot_ctl; a.n_det_total = n_dets;
    a.rows = (long long*)(s.hd_out + ol.rows); a.rows_cap = rows_cap; a.row_cursor = s.d_cursor;
    a.task_row_off = (int*)(s.hd_out + ol.row_off); a.task_row_n = (int*)(s.hd_out + ol.row_n);
    a.task_ntracks = (int*)(s.hd_out + ol.ntracks); a.task_T = (int*)(s.hd_out + ol.tT);
    a.status = s.d_cursor + 4;                               // device memory (atomics), copied next to the rows below
    a.scratch = e->d_track_scratch; a.scratch_per_wg = track_scratch_per_wg(); a.cap = cap; a.frame_w = W; a.frame_h = H;
    a.all_tables = tables_fit ? 1 : 0;
    // Dense steps (BASELINE.json configs[2]: ~90 tracks x ~70 detections): the matrix the assignment scans goes into LDS behind the work
    // arrays -- as much as the kernel's dynamic LDS allows; light batches ask for none (their problems fit the 2 KB static buffers, and
    // a small LDS footprint lets conv workgroups share the tracker's CUs).
    a.lmat_doubles = 0;
    if (mat_need > 256) {
        const long long room = ((long long)VC_TRACK_DYN_LDS_BYTES - (long long)step_work_bytes(cap)) / 8;
        a.lmat_doubles = (int)std::max(0ll, std::min(room, (mat_need + 63) / 64 * 64));
    }
    static const bool no_reg = getenv("VC_TRACK_NO_REG") != nullptr;
    a.no_reg = no_reg ? 1 : 0;
    a.dbg_costs = st == 3 ? 1 : 0;                               // blocking entry points: vc_tracker_debug_costs may read the rows back
    static const bool dbg_on = getenv("VC_TRACK_DBG") != nullptr;        // diagnostics: phase times of every task, printed per batch
    long long* dbg = nullptr;
    if (dbg_on && hipMalloc((void**)&dbg, (size_t)n_tasks * 128) == hipSuccess) { hipMemsetAsync(dbg, 0, (size_t)n_tasks * 128, ts); a.dbg = dbg; }
    {
        ProfScope ps(e, VC_PROF_TRACK);
        VC_TRY(launch_track_batch(a, n_wg, ts));
    }
    if (dbg) {
        std::vector<long long> h((size_t)n_tasks * 16);
        hipStreamSynchronize(ts);
        hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        hipFree(dbg);
        double acc[5] = {0, 0, 0, 0, 0}, sumT = 0, sumD = 0;
        int maxwg = 0;
        for (const TrackWgPlan& pl : plans) maxwg = std::max(maxwg, pl.task_end - pl.task_begin);
        for (int k = 0; k < n_tasks; ++k) {
            for (int i = 0; i < 5; ++i) acc[i] += (double)(h[(size_t)k * 16 + i + 1] - h[(size_t)k * 16 + i]) / 100.0;
            sumT += (double)h[(size_t)k * 16 + 6]; sumD += (double)h[(size_t)k * 16 + 7];
        }
        fprintf(stderr, "[vc track dbg] %d tasks in %d workgroups (longest %d tasks), LDS cap %d; us per task: predict %.1f cost %.1f match %.1f apply %.1f finish %.1f; mean T %.1f D %.1f\n",
                n_tasks, n_wg, maxwg, cap, acc[0] / n_tasks, acc[1] / n_tasks, acc[2] / n_tasks, acc[3] / n_tasks, acc[4] / n_tasks, sumT / n_tasks, sumD / n_tasks);
        long long t_lo = INT64_MAX, t_hi = 0;
        double worst = 0; const TrackWgPlan* wp = nullptr;
        for (const TrackWgPlan& pl : plans) {
            const long long b0 = h[(size_t)pl.task_begin * 16], b1 = h[(size_t)(pl.task_end - 1) * 16 + 5];
            t_lo = std::min(t_lo, b0); t_hi = std::max(t_hi, b1);
            if ((double)(b1 - b0) > worst) { worst = (double)(b1 - b0); wp = &pl; }
        }
        if (wp) {
            double a2[5] = {0, 0, 0, 0, 0}, sT = 0, sD = 0, gaps = 0;
            const int nt = wp->task_end - wp->task_begin;
            for (int k = wp->task_begin; k < wp->task_end; ++k) {
                for (int i = 0; i < 5; ++i) a2[i] += (double)(h[(size_t)k * 16 + i + 1] - h[(size_t)k * 16 + i]) / 100.0;
                sT += (double)h[(size_t)k * 16 + 6]; sD += (double)h[(size_t)k * 16 + 7];
                if (k > wp->task_begin) gaps += (double)(h[(size_t)k * 16] - h[(size_t)(k - 1) * 16 + 5]) / 100.0;
            }
            double mp[3] = {0, 0, 0};
            for (int k = wp->task_begin; k < wp->task_end; ++k)
                for (int i = 0; i < 3; ++i) mp[i] += (double)(h[(size_t)k * 16 + 9 + i] - h[(size_t)k * 16 + 8 + i]) / 100.0;
            fprintf(stderr, "[vc track dbg]   match split of that workgroup, us per task: setup %.1f cascade %.1f iou stage %.1f\n", mp[0] / nt, mp[1] / nt, mp[2] / nt);
            fprintf(stderr, "[vc track dbg]   kernel span %.0f us; slowest workgroup (tracker %d): %d tasks, %.0f us, per task predict %.1f cost %.1f match %.1f apply %.1f finish %.1f gap %.1f; mean T %.1f D %.1f\n",
                    (double)(t_hi - t_lo) / 100.0, wp->tracker, nt, worst / 100.0, a2[0] / nt, a2[1] / nt, a2[2] / nt, a2[3] / nt, a2[4] / nt, gaps / nt, sT / nt, sD / nt);
        }
    }
    VC_HIP(hipMemcpyAsync(s.h_out + ol.status, s.d_cursor + 4, 16, hipMemcpyDeviceToHost, ts));
    VC_HIP(hipEventRecord(s.done, ts));
    s.busy = true;
    return VC_OK;
}

int track_collect(vc_engine* e, int st, int64_t* out_rows6, int cap_rows_per_frame, int* out_m) {
    TrackStage& s = e->tstage[st];
    VC_CHECK(s.busy, VC_ERR_STATE, "no tracker batch in staging slot %d", st);
    VC_HIP(hipEventSynchronize(s.done));
    s.busy = false;
    for (int f = 0; f < s.b; ++f) out_m[f] = 0;
    if (s.n_tasks == 0) return VC_OK;
    const OutLayout ol = out_layout(s.n_tasks, s.rows_cap);
    const int* row_off = (const int*)(s.h_out + ol.row_off);
    const int* row_n = (const int*)(s.h_out + ol.row_n);
    const int* ntr = (const int*)(s.h_out + ol.ntracks);
    const int* status = (const int*)(s.h_out + ol.status);
    const long long* rows = (const long long*)(s.h_out + ol.rows);
    for (size_t i = 0; i < s.tracker_dets.size(); ++i) {                        // size bounds for the next batches
        Tracker& tk = *e->trackers[s.tracker_dets[i].first];
        tk.pending_dets -= s.tracker_dets[i].second;
        tk.known_tracks = ntr[s.last_task_of[i]];
    }
    if (status[0] != TERR_NONE) {
        const char* what = status[0] == TERR_TRACK_CAP ? "live tracks + detections exceed the per-tracker capacity (vc_engine_config.tracks_per_tracker)"
                         : status[0] == TERR_POOL     ? "track pool exhausted (vc_engine_config.max_tracks)"
                         : status[0] == TERR_ROWS     ? "more output rows than the caller's buffers hold (cap_rows)"
                         : status[0] == TERR_TABLE    ? "appearance table missing although the host's bound said it fits (internal)"
                                                      : "infeasible assignment problem";
        set_error("tracker %d: %s; the tracker is stopped until vc_tracker_reset", status[1], what);
        return VC_ERR_CAPACITY;
    }
    for (int i = 0; i < s.n_tasks; ++i) {                                       // (frame, class) order
        const int k = s.dev_index[i], f = s.tasks[i].frame, m = row_n[k];
        VC_CHECK(out_m[f] + m <= cap_rows_per_frame, VC_ERR_CAPACITY, "frame %d needs room for %d rows", f, out_m[f] + m);
        memcpy(out_rows6 + ((size_t)f * cap_rows_per_frame + out_m[f]) * 6, rows + (size_t)row_off[k] * 6, (size_t)m * 6 * sizeof(int64_t));
        out_m[f] += m;
    }
    return VC_OK;
}

static void xyxy_to_cxcywh(const double* b, double* o) {
    const double w = b[2] - b[0], h = b[3] - b[1];
    o[0] = b[0] + w / 2; o[1] = b[1] + h / 2; o[2] = w; o[3] = h;
}

static void crop_corners_i(const double* b, int W, int H, int* c) {  // deep_sort.py:89-95
    c[0] = std::max((int)(b[0] - b[2] / 2), 0); c[2] = std::min((int)(b[0] + b[2] / 2), W - 1);
    c[1] = std::max((int)(b[1] - b[3] / 2), 0); c[3] = std::min((int)(b[1] + b[3] / 2), H - 1);
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
    std::vector<std::vector<FrameClassDets>> frames(1);
    int total_tracks = 0;
    for (size_t j = 0; j < tracker_ids.size(); ++j) {
        const auto& g = groups[j];
        std::vector<double> bx(g.size() * 4), cf(g.size());
        std::vector<int> rows(g.size());
        for (size_t i = 0; i < g.size(); ++i) {
            memcpy(&bx[i * 4], xyxy + (size_t)g[i] * 4, 4 * sizeof(double));
            cf[i] = conf[g[i]];
            rows[i] = g[i];
        }
        const int ts = tracker_slot(e->trackers, tracker_ids[j]);             // (handles, as the caller holds them)
        VC_CHECK(ts >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle %d", tracker_ids[j]);
        FrameClassDets fc{labels[j], ts, {}};
        prepare_dets(bx.data(), cf.data(), rows.data(), (int)g.size(), e->trackers[ts]->p, fc.dets);
        total_tracks += e->trackers[ts]->known_tracks + (int)fc.dets.conf.size();
        frames[0].push_back(std::move(fc));
    }
    const int cap = std::max(total_tracks, 16);
    VC_TRY(track_enqueue(e, 3, frames, e->d_feat, W, H, cap, nullptr));
    std::vector<int64_t> buf((size_t)cap * 6);
    int m = 0;
    VC_TRY(track_collect(e, 3, buf.data(), cap, &m));
    rows6.assign(buf.begin(), buf.begin() + (size_t)m * 6);
    return VC_OK;
}

// ---- tracker state on the host (blocking paths: the engine is idle) ---------------------------------------------------------------
struct HostTrackerState { TrackerHdr hdr; std::vector<int> list; std::vector<TrackRecD> recs; };

static int download_tracker(vc_engine* e, int id, HostTrackerState& st) {
    VC_TRY(track_idle(e));
    VC_HIP(hipStreamSynchronize(e->stream));
    VC_HIP(hipMemcpy(&st.hdr, e->d_hdrs + id, sizeof(TrackerHdr), hipMemcpyDeviceToHost));
    const int n = st.hdr.n_tracks;
    st.list.resize(n); st.recs.resize(n);
    if (n) VC_HIP(hipMemcpy(st.list.data(), e->d_lists + (size_t)id * e->list_cap, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    for (int t = 0; t < n; ++t) VC_HIP(hipMemcpy(&st.recs[t], e->d_recs + st.list[t], sizeof(TrackRecD), hipMemcpyDeviceToHost));
    return VC_OK;
}

// free-slot stack, host side (engine idle): take n slots / give slots back
static int host_take_slots(vc_engine* e, int n, std::vector<int>& out) {
    int ctl[2];
    VC_HIP(hipMemcpy(ctl, e->d_free, sizeof(ctl), hipMemcpyDeviceToHost));
    VC_CHECK(ctl[0] >= n, VC_ERR_CAPACITY, "track pool too small for %d more tracks (max_tracks = %d)", n, e->cfg.max_tracks);
    out.resize(n);
    if (n) VC_HIP(hipMemcpy(out.data(), e->d_free_stack + (ctl[0] - n), (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    ctl[0] -= n;
    VC_HIP(hipMemcpy(e->d_free, ctl, sizeof(int), hipMemcpyHostToDevice));
    return VC_OK;
}
static int host_give_slots(vc_engine* e, const std::vector<int>& slots) {
    if (slots.empty()) return VC_OK;
    int ctl[2];
    VC_HIP(hipMemcpy(ctl, e->d_free, sizeof(ctl), hipMemcpyDeviceToHost));
    VC_HIP(hipMemcpy(e->d_free_stack + ctl[0], slots.data(), slots.size() * sizeof(int), hipMemcpyHostToDevice));
    ctl[0] += (int)slots.size();
    VC_HIP(hipMemcpy(e->d_free, ctl, sizeof(int), hipMemcpyHostToDevice));
    return VC_OK;
}

static TrackerHdr make_hdr(const vc_tracker_params& p) {
    TrackerHdr h{};
    h.max_dist = p.max_dist; h.max_iou_distance = p.max_iou_distance; h.next_id = 1;
    h.max_age = p.max_age; h.n_init = p.n_init; h.nn_budget = p.nn_budget; h.n_tracks = 0; h.err = TERR_NONE;
    return h;
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
    int slot = -1;                       // an id given back by vc_tracker_destroy is used again before the table grows
    for (int i = 0; i < (int)e->trackers.size() && slot < 0; ++i) if (e->trackers[i]->released) slot = i;
    VC_CHECK(slot >= 0 || (int)e->trackers.size() < e->max_trackers, VC_ERR_CAPACITY,
             "more than max_trackers (%d) trackers (vc_tracker_destroy gives an id back)", e->max_trackers);
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));           // batches in flight index e->trackers
    const TrackerHdr h = make_hdr(*p);
    if (slot < 0) { slot = (int)e->trackers.size(); e->trackers.push_back(std::unique_ptr<Tracker>(new Tracker())); }
    VC_HIP(hipMemcpy(e->d_hdrs + slot, &h, sizeof(h), hipMemcpyHostToDevice));
    Tracker& t = *e->trackers[slot];
    t.p = *p; t.known_tracks = 0; t.pending_dets = 0; t.released = false;
    *id = slot | (t.gen << 16);
    return VC_OK;
}

// The reference builds a new VideoTracker (one DeepSort per class) for every video (modules/__init__.py:32-36) and drops the old one:
// the drop-in's DeepSort gives its tracker back here, so that a process that walks a folder of videos does not run out of ids.
int vc_tracker_destroy(vc_engine* e, int handle) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_TRY(vc_tracker_reset(e, handle)); // waits for the batches in flight, returns the track slots to the pool
    e->trackers[id]->released = true;
    e->trackers[id]->gen = (e->trackers[id]->gen + 1) & 0x7fff;      // handles of this incarnation are refused from now on
    return VC_OK;
}

int vc_tracker_reset(vc_engine* e, int handle) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    HostTrackerState st;
    VC_TRY(download_tracker(e, id, st));
    VC_TRY(host_give_slots(e, st.list));
    const TrackerHdr h = make_hdr(e->trackers[id]->p);
    VC_HIP(hipMemcpy(e->d_hdrs + id, &h, sizeof(h), hipMemcpyHostToDevice));
    e->trackers[id]->known_tracks = 0;
    e->trackers[id]->pending_dets = 0;
    return VC_OK;
}

int vc_tracker_step(vc_engine* e, int handle, const double* tlwh, const double* conf, const float* feat, int k) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_CHECK(k == 0 || (tlwh && conf && feat), VC_ERR_ARG, "null argument");
    VC_CHECK(k >= 0 && k <= e->det_cap, VC_ERR_CAPACITY, "%d detections exceed capacity %d", k, e->det_cap);
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    if (k > 0) VC_HIP(hipMemcpyAsync(e->d_feat_in, feat, (size_t)k * VC_FEAT_DIM * sizeof(float), hipMemcpyHostToDevice, e->stream));
    std::vector<std::vector<FrameClassDets>> frames(1);
    FrameClassDets fc{0, id, {}};
    fc.dets.tlwh.assign(tlwh, tlwh + (size_t)k * 4);
    fc.dets.conf.assign(conf, conf + k);
    fc.dets.feat_rows.resize(k);
    std::iota(fc.dets.feat_rows.begin(), fc.dets.feat_rows.end(), 0);
    frames[0].push_back(std::move(fc));
    const int cap = e->trackers[id]->known_tracks + k + 16;
    VC_TRY(track_enqueue(e, 3, frames, e->d_feat_in, 1 << 30, 1 << 30, cap, nullptr));
    std::vector<int64_t> buf((size_t)cap * 6);
    int m = 0;
    return track_collect(e, 3, buf.data(), cap, &m);
}

int vc_tracker_count(vc_engine* e, int handle, int* n) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && n && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    HostTrackerState st;
    VC_TRY(download_tracker(e, id, st));
    *n = st.hdr.n_tracks;
    return VC_OK;
}

int vc_tracker_state(vc_engine* e, int handle, int cap, int64_t* ids, int* state, int* hits, int* age, int* tsu, double* mean8,
                     double* cov64, int* gallery_count) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    HostTrackerState st;
    VC_TRY(download_tracker(e, id, st));
    VC_CHECK(st.hdr.n_tracks <= cap, VC_ERR_CAPACITY, "need room for %d tracks", st.hdr.n_tracks);
    for (int t = 0; t < st.hdr.n_tracks; ++t) {
        const TrackRecD& tr = st.recs[t];
        const int slot = st.list[t];
        if (ids) ids[t] = tr.id;
        if (state) state[t] = tr.state;
        if (hits) hits[t] = tr.hits;
        if (age) age[t] = tr.age;
        if (tsu) tsu[t] = tr.tsu;
        if (gallery_count) gallery_count[t] = tr.state == CONFIRMED ? tr.gal_count : 0;
        if (mean8) VC_HIP(hipMemcpy(mean8 + (size_t)t * 8, e->pool.mean + (size_t)slot * 8, 8 * sizeof(double), hipMemcpyDeviceToHost));
        if (cov64) VC_HIP(hipMemcpy(cov64 + (size_t)t * 64, e->pool.cov + (size_t)slot * 64, 64 * sizeof(double), hipMemcpyDeviceToHost));
    }
    return VC_OK;
}

// The cost matrices of the LAST blocking step of this engine (vc_tracker_step / vc_deepsort_update on ONE tracker), as the batch
// kernel left them in its scratch: app[t * D + d] = gated appearance cost of list position t (valid for tracks that were
// confirmed when the step began), iou[t * D + d] = 1 - IoU (valid for the IoU candidates).  T = tracks before the step.
int vc_tracker_debug_costs(vc_engine* e, int cap_entries, double* app, double* iou, int* T, int* D) {
    VC_CHECK(e && app && iou && T && D, VC_ERR_ARG, "null argument");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    const TrackStage& s = e->tstage[3];
    VC_CHECK(s.n_tasks == 1 && s.n_wg == 1 && !s.busy, VC_ERR_STATE, "the last blocking call did not step exactly one tracker");
    const OutLayout ol = out_layout(1, s.rows_cap);
    *T = ((const int*)(s.h_out + ol.tT))[0];
    *D = s.tasks[0].det_n;
    const size_t n = (size_t)*T * *D;
    VC_CHECK(n <= (size_t)cap_entries, VC_ERR_CAPACITY, "need room for %zu entries", n);
    VC_HIP(hipStreamSynchronize(e->stream));
    const size_t mat = TC_MAT;                                       // scratch of workgroup 0: [appearance | IoU | ...], rows of D entries
    if (n) {
        VC_HIP(hipMemcpy(app, e->d_track_scratch, n * sizeof(double), hipMemcpyDeviceToHost));
        VC_HIP(hipMemcpy(iou, e->d_track_scratch + mat, n * sizeof(double), hipMemcpyDeviceToHost));
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

int vc_tracker_snapshot(vc_engine* e, int handle, void* buf, size_t cap, size_t* size) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && size && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    HostTrackerState st;
    VC_TRY(download_tracker(e, id, st));
    const vc_tracker_params& tp = e->trackers[id]->p;
    size_t need = sizeof(SnapHeader);
    for (const TrackRecD& tr : st.recs) need += sizeof(SnapTrack) + (size_t)std::min(tr.gal_count, tp.nn_budget) * VC_FEAT_DIM * sizeof(float);
    *size = need;
    if (!buf) return VC_OK;                       // size query
    VC_CHECK(cap >= need, VC_ERR_CAPACITY, "snapshot needs %zu bytes", need);
    char* o = (char*)buf;
    SnapHeader h{};
    memcpy(h.magic, kSnapMagic, 8);
    h.feat_dim = VC_FEAT_DIM; h.n_tracks = st.hdr.n_tracks; h.next_id = st.hdr.next_id;
    h.max_dist = tp.max_dist; h.min_confidence = tp.min_confidence; h.nms_max_overlap = tp.nms_max_overlap;
    h.max_iou_distance = tp.max_iou_distance; h.max_age = tp.max_age; h.n_init = tp.n_init; h.nn_budget = tp.nn_budget;
    memcpy(o, &h, sizeof(h)); o += sizeof(h);
    for (int i = 0; i < st.hdr.n_tracks; ++i) {
        const TrackRecD& tr = st.recs[i];
        const int slot = st.list[i];
        SnapTrack t{};
        t.id = tr.id; t.state = tr.state; t.hits = tr.hits; t.age = tr.age; t.tsu = tr.tsu; t.gal_count = tr.gal_count; t.gal_head = tr.gal_head;
        t.last_conf = 0.0;
        VC_HIP(hipMemcpy(t.mean, e->pool.mean + (size_t)slot * 8, sizeof(t.mean), hipMemcpyDeviceToHost));
        VC_HIP(hipMemcpy(t.cov, e->pool.cov + (size_t)slot * 64, sizeof(t.cov), hipMemcpyDeviceToHost));
        memcpy(o, &t, sizeof(t)); o += sizeof(t);
        const size_t rows = (size_t)std::min(tr.gal_count, tp.nn_budget);
        if (rows) VC_HIP(hipMemcpy(o, e->pool.gallery + (size_t)slot * e->pool.budget_cap * VC_FEAT_DIM, rows * VC_FEAT_DIM * sizeof(float), hipMemcpyDeviceToHost));
        o += rows * VC_FEAT_DIM * sizeof(float);
    }
    return VC_OK;
}

int vc_tracker_restore(vc_engine* e, int handle, const void* buf, size_t size) {
    const int id = e ? tracker_slot(e->trackers, handle) : -1;
    VC_CHECK(e && buf && id >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle");
    VC_CHECK(size >= sizeof(SnapHeader), VC_ERR_ARG, "snapshot truncated");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    const char* in = (const char*)buf;
    SnapHeader h;
    memcpy(&h, in, sizeof(h)); in += sizeof(h);
    VC_CHECK(memcmp(h.magic, kSnapMagic, 8) == 0 && h.feat_dim == VC_FEAT_DIM && h.n_tracks >= 0, VC_ERR_ARG, "not a tracker snapshot");
    VC_CHECK(h.nn_budget >= 1 && h.nn_budget <= e->cfg.nn_budget_cap, VC_ERR_CAPACITY, "snapshot nn_budget %d exceeds nn_budget_cap %d", h.nn_budget, e->cfg.nn_budget_cap);
    VC_CHECK(h.max_age >= 1 && h.n_init >= 1 && h.next_id >= 1, VC_ERR_ARG, "snapshot corrupt (parameters)");
    VC_CHECK(h.n_tracks <= e->list_cap, VC_ERR_CAPACITY, "snapshot holds %d tracks, tracks_per_tracker is %d", h.n_tracks, e->list_cap);
    // validate the whole blob before touching the tracker
    {
        const char* q = in;
        for (int i = 0; i < h.n_tracks; ++i) {
            VC_CHECK((size_t)(q - (const char*)buf) + sizeof(SnapTrack) <= size, VC_ERR_ARG, "snapshot truncated");
            SnapTrack t;
            memcpy(&t, q, sizeof(t)); q += sizeof(t);
            VC_CHECK(t.gal_count >= 0 && t.gal_count <= h.nn_budget && t.gal_head >= 0 && t.gal_head < h.nn_budget, VC_ERR_ARG, "snapshot corrupt (gallery ring)");
            VC_CHECK(t.state == TENTATIVE || t.state == CONFIRMED, VC_ERR_ARG, "snapshot corrupt (track state %d)", t.state);
            VC_CHECK(t.hits >= 0 && t.age >= 0 && t.tsu >= 0 && t.id >= 1 && t.id < h.next_id, VC_ERR_ARG, "snapshot corrupt (track counters)");
            q += (size_t)t.gal_count * VC_FEAT_DIM * sizeof(float);
        }
        VC_CHECK((size_t)(q - (const char*)buf) == size, VC_ERR_ARG, "snapshot size mismatch");
    }
    HostTrackerState old;
    VC_TRY(download_tracker(e, id, old));
    VC_TRY(host_give_slots(e, old.list));
    std::vector<int> slots;
    VC_TRY(host_take_slots(e, h.n_tracks, slots));
    Tracker& tk = *e->trackers[id];
    tk.p.max_dist = h.max_dist; tk.p.min_confidence = h.min_confidence; tk.p.nms_max_overlap = h.nms_max_overlap;
    tk.p.max_iou_distance = h.max_iou_distance; tk.p.max_age = h.max_age; tk.p.n_init = h.n_init; tk.p.nn_budget = h.nn_budget;
    TrackerHdr dh = make_hdr(tk.p);
    dh.next_id = h.next_id; dh.n_tracks = h.n_tracks;
    for (int i = 0; i < h.n_tracks; ++i) {
        SnapTrack t;
        memcpy(&t, in, sizeof(t)); in += sizeof(t);
        TrackRecD tr{};
        tr.id = t.id; tr.state = t.state; tr.hits = t.hits; tr.age = t.age; tr.tsu = t.tsu; tr.gal_count = t.gal_count; tr.gal_head = t.gal_head;
        const int slot = slots[i];
        VC_HIP(hipMemcpy(e->d_recs + slot, &tr, sizeof(tr), hipMemcpyHostToDevice));
        VC_HIP(hipMemcpy(e->pool.mean + (size_t)slot * 8, t.mean, sizeof(t.mean), hipMemcpyHostToDevice));
        VC_HIP(hipMemcpy(e->pool.cov + (size_t)slot * 64, t.cov, sizeof(t.cov), hipMemcpyHostToDevice));
        const size_t rows = (size_t)t.gal_count;
        if (rows) VC_HIP(hipMemcpy(e->pool.gallery + (size_t)slot * e->pool.budget_cap * VC_FEAT_DIM, in, rows * VC_FEAT_DIM * sizeof(float), hipMemcpyHostToDevice));
        in += rows * VC_FEAT_DIM * sizeof(float);
    }
    if (h.n_tracks) VC_HIP(hipMemcpy(e->d_lists + (size_t)id * e->list_cap, slots.data(), (size_t)h.n_tracks * sizeof(int), hipMemcpyHostToDevice));
    VC_HIP(hipMemcpy(e->d_hdrs + id, &dh, sizeof(dh), hipMemcpyHostToDevice));
    tk.known_tracks = h.n_tracks;
    tk.pending_dets = 0;
    return VC_OK;
}

int vc_deepsort_update(vc_engine* e, int id, const uint8_t* bgr, int h, int w, const double* bbox_xyxy, const double* conf, int k,
                       int64_t* out_rows7, int cap_rows, int* out_m) {
    VC_CHECK(e && bgr && out_m && tracker_slot(e->trackers, id) >= 0, VC_ERR_ARG, "bad argument (null pointer, bad or stale tracker handle)");
    VC_CHECK(k >= 1 && bbox_xyxy && conf, VC_ERR_ARG, "DeepSort.update needs at least one box (the reference only calls it then)");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
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
    VC_CHECK(e && trackers && bgr && out_m, VC_ERR_ARG, "null argument");
    VC_CHECK(n >= 1 && boxes_xywh && labels && scores, VC_ERR_ARG, "VideoTracker.run needs at least one box (quirk Q1)");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
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

// ---- single-function entry points (parity tests): the batch kernel's device functions on caller-supplied state --------------------
static int with_pool(int n, TrackPool& tp, std::vector<void*>& allocs) {
    tp.max_tracks = n; tp.budget_cap = 1;
    VC_HIP(hipMalloc((void**)&tp.mean, (size_t)n * 8 * sizeof(double))); allocs.push_back(tp.mean);
    VC_HIP(hipMalloc((void**)&tp.cov, (size_t)n * 64 * sizeof(double))); allocs.push_back(tp.cov);
    tp.gallery = nullptr;
    return VC_OK;
}
static void free_all(std::vector<void*>& a) { for (void* p : a) hipFree(p); a.clear(); }

#define VC_HOST_FINISH(st)                                                                                              \
    if ((st) == VC_OK && hipDeviceSynchronize() != hipSuccess) { set_error("kernel failed: %s", hipGetErrorString(hipGetLastError())); (st) = VC_ERR_HIP; }

// which: 0 initiate (z -> mean, cov), 1 predict, 2 update (z)
static int kalman_kat(double* mean8, double* cov64, const double* z4, int n, int which) {
    VC_CHECK(mean8 && cov64 && n > 0, VC_ERR_ARG, "bad argument");
    TrackPool tp{}; std::vector<void*> al; double* dz = nullptr;
    int st = with_pool(n, tp, al);
    if (st == VC_OK && which != 0) { hipMemcpy(tp.mean, mean8, (size_t)n * 64, hipMemcpyHostToDevice); hipMemcpy(tp.cov, cov64, (size_t)n * 512, hipMemcpyHostToDevice); }
    if (st == VC_OK && z4) {
        if (hipMalloc((void**)&dz, (size_t)n * 32) != hipSuccess) { set_error("alloc"); st = VC_ERR_HIP; } else { al.push_back(dz); hipMemcpy(dz, z4, (size_t)n * 32, hipMemcpyHostToDevice); }
    }
    if (st == VC_OK) st = launch_kat_kalman(tp, which, dz, n, nullptr);
    VC_HOST_FINISH(st);
    if (st == VC_OK) { hipMemcpy(mean8, tp.mean, (size_t)n * 64, hipMemcpyDeviceToHost); hipMemcpy(cov64, tp.cov, (size_t)n * 512, hipMemcpyDeviceToHost); }
    free_all(al);
    return st;
}
int vc_kalman_initiate_host(const double* xyah, int n, double* mean8, double* cov64) {
    VC_CHECK(xyah, VC_ERR_ARG, "null measurement");
    return kalman_kat(mean8, cov64, xyah, n, 0);
}
int vc_kalman_predict_host(double* mean8, double* cov64, int n) { return kalman_kat(mean8, cov64, nullptr, n, 1); }
int vc_kalman_update_host(double* mean8, double* cov64, const double* z4, int n) {
    VC_CHECK(z4, VC_ERR_ARG, "null measurement");
    return kalman_kat(mean8, cov64, z4, n, 2);
}

// squared Mahalanobis distances of ONE track against n_meas measurements (the gate's project4 / chol4 / maha4)
int vc_kalman_gating_host(const double* mean8, const double* cov64, const double* z4, int n_meas, double* out) {
    VC_CHECK(mean8 && cov64 && z4 && out && n_meas > 0, VC_ERR_ARG, "bad argument");
    TrackPool tp{}; std::vector<void*> al; double *dz = nullptr, *dout = nullptr;
    int st = with_pool(1, tp, al);
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
    TrackPool tp{}; std::vector<void*> al;
    int st = with_pool(t, tp, al);
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

// scipy.optimize.linear_sum_assignment through the tracker kernel's own solver (track_core.h lap_solve, one wave on the device)
int vc_lap_host(const double* cost, int nr, int nc, int* rows, int* cols, int* n_assigned) {
    VC_CHECK(cost && rows && cols && n_assigned && nr >= 0 && nc >= 0, VC_ERR_ARG, "bad argument");
    *n_assigned = 0;
    if (nr == 0 || nc == 0) return VC_OK;
    VC_CHECK(nr <= 512 && nc <= 512, VC_ERR_CAPACITY, "lap: at most 512 rows / columns");
    std::vector<void*> al;
    double *dc = nullptr, *dt = nullptr; int *dr = nullptr, *dq = nullptr, *dn = nullptr;
    const int k = std::min(nr, nc);
    int st = VC_OK;
    if (hipMalloc((void**)&dc, (size_t)nr * nc * 8) != hipSuccess || hipMalloc((void**)&dt, (size_t)nr * nc * 8) != hipSuccess ||
        hipMalloc((void**)&dr, (size_t)k * 4) != hipSuccess || hipMalloc((void**)&dq, (size_t)k * 4) != hipSuccess || hipMalloc((void**)&dn, 4) != hipSuccess) { set_error("alloc"); st = VC_ERR_HIP; }
    for (void* p : {(void*)dc, (void*)dt, (void*)dr, (void*)dq, (void*)dn}) if (p) al.push_back(p);
    if (st == VC_OK) { hipMemcpy(dc, cost, (size_t)nr * nc * 8, hipMemcpyHostToDevice); st = launch_kat_lap(dc, nr, nc, dt, dr, dq, dn, nullptr); }
    VC_HOST_FINISH(st);
    if (st == VC_OK) {
        int n = 0;
        hipMemcpy(&n, dn, 4, hipMemcpyDeviceToHost);
        if (n < 0) { set_error("lap: infeasible cost matrix"); st = VC_ERR_ARG; }
        else { hipMemcpy(rows, dr, (size_t)n * 4, hipMemcpyDeviceToHost); hipMemcpy(cols, dq, (size_t)n * 4, hipMemcpyDeviceToHost); *n_assigned = n; }
    }
    free_all(al);
    return st;
}

}  // extern "C"
