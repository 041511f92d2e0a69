// Fused stream path: the per-frame body of CountingPipeline.run (/root/reference/modules/__init__.py:54-84) for a
// batch of device-resident frames of one camera: detect (batched) -> marshal like networks/yolo.py:72-97 ->
// skip empty frames (quirk Q1) -> crops + ReID for every box of the batch in one launch -> per frame, in order,
// one batched tracker step over the classes that have boxes (modules/track.py:50-59).
#include <algorithm>
#include <cmath>
#include <numeric>

#include <chrono>
#include <cstdlib>
#include <thread>

#include "engine.h"

using namespace vc;

namespace {
// host-timeline instrumentation (VC_TIMING=1): where a vc_stream_run call spends its wall time
struct Tm {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long calls = 0;
    std::chrono::steady_clock::time_point t;
    bool on = getenv("VC_TIMING") != nullptr;
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void lap(int i) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
    void report() {
        if (!on || ++calls % 10) return;
        fprintf(stderr, "[vc timing, us per call] wait_det+reid_issue %.0f prepare %.0f enqueue %.0f - %.0f %.0f %.0f %.0f %.0f\n",
                acc[0] / 10, acc[1] / 10, acc[2] / 10, acc[3] / 10, acc[4] / 10, acc[5] / 10, acc[6] / 10, acc[7] / 10);
        for (double& a : acc) a = 0;                   // windowed: the last 10 calls
    }
} g_tm;
}  // namespace

namespace {

// DataFrame.to_json(double_precision=10) -> json.loads (quirk Q9), same rounding as oracle/yolov5.py::marshal_like_reference
inline double round10(double v) { return std::nearbyint(v * 1e10) / 1e10; }

typedef vc_engine::FrameDets FrameDets;     // boxes as VideoTracker.run sees them (xywh -> xyxy round trip included)

void marshal(const float* det6, int n, FrameDets& out) {
    out.xyxy.clear(); out.conf.clear(); out.label.clear();
    for (int i = 0; i < n; ++i) {
        const float* d = det6 + (size_t)i * 6;
        const double x1 = round10((double)d[0]), y1 = round10((double)d[1]), x2 = round10((double)d[2]), y2 = round10((double)d[3]);
        const double w = x2 - x1, h = y2 - y1;                         // networks/yolo.py:82
        out.xyxy.push_back(x1); out.xyxy.push_back(y1);
        out.xyxy.push_back(w + x1); out.xyxy.push_back(h + y1);        // modules/track.py:39-41
        out.conf.push_back(round10((double)d[4]));
        out.label.push_back((int)d[5]);
    }
}

}  // namespace

extern "C" {

int vc_stream_inject(vc_engine* e, const float* det6, const int* count, int b, int n) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    if (!det6 || !count || b <= 0) { e->inject_b = 0; e->inject_det.clear(); e->inject_count.clear(); return VC_OK; }
    VC_CHECK(n <= e->cfg.max_det, VC_ERR_CAPACITY, "injected detections per frame exceed max_det");
    e->inject_det.assign(det6, det6 + (size_t)b * n * 6);
    e->inject_count.assign(count, count + b);
    e->inject_b = b; e->inject_n = n;
    return VC_OK;
}

// Enqueue the detector (letterbox .. NMS, results to pinned memory) for a batch on the detector stream and return
// immediately.  At most two submissions may be outstanding; vc_stream_run consumes them in order.
int vc_stream_submit(vc_engine* e, const void* frames_dev, int b, int h, int w) {
    VC_CHECK(e && frames_dev, VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_detector, VC_ERR_STATE, "engine not finalized");
    VC_CHECK(e->pending.size() < 2, VC_ERR_STATE, "two submissions are already in flight: call vc_stream_run first");
    VC_HIP(hipSetDevice(e->cfg.device));
    for (int i = 0; i < 4; ++i)                          // a batch staged by vc_stream_stage_host: the detector starts behind its copy
        if (e->d_ingest[i] && frames_dev == e->d_ingest[i] && e->ingest_staged[i]) {
            VC_HIP(hipStreamWaitEvent(e->dstream, e->ev_ingest[i], 0));
            e->ingest_staged[i] = false;
        }
    const int slot = (int)(e->submit_seq++ & 1);
    const int md = e->cfg.max_det;
    VC_TRY(run_detector_dev(e, (const uint8_t*)frames_dev, b, h, w, /*swap_rb=*/true));
    VC_HIP(hipMemcpyAsync(e->h_det2[slot], e->post.det, (size_t)b * md * 6 * sizeof(float), hipMemcpyDeviceToHost, e->dstream));
    VC_HIP(hipMemcpyAsync(e->h_det_count2[slot], e->post.det_count, b * sizeof(int), hipMemcpyDeviceToHost, e->dstream));
    VC_HIP(hipEventRecord(e->ev_det[slot], e->dstream));
    vc_engine::Pending pd{};
    pd.frames = frames_dev; pd.b = b; pd.h = h; pd.w = w; pd.slot = slot;
    if (e->inject_b > 0) { pd.inj_det = e->inject_det; pd.inj_cnt = e->inject_count; pd.inj_b = e->inject_b; pd.inj_n = e->inject_n; }   // this batch's own rectangles
    e->pending.push_back(std::move(pd));
    return VC_OK;
}

// Ingest from host memory (the reference decodes on the host and hands numpy frames over, modules/datasets.py:47-61):
// vc_stream_stage_host copies the batch into one of four device staging slots on the engine's copy stream and returns; a later
// vc_stream_submit of the returned address enqueues the detector behind the copy.  Staged ONE BATCH FURTHER AHEAD than it is submitted
// (stage i + 2, submit i + 1, run i, collect i - 1) the copy runs under the detector of the batch before it; vc_stream_submit_host
// (stage + submit in one call) leaves the copy in front of its own detector, overlapped only with the ReID of the batch before --
// 157 MB per 128 frames of 640 x 640 is 3 - 5 ms of PCIe in front of a 7 ms step.  *frames_dev_out receives the device address to pass to
// vc_stream_run / vc_stream_run_async for this batch.  `frames_host` should be pinned (hipHostMalloc / hipHostRegister / torch
// pin_memory) for the copy to overlap; it may be reused as soon as the call returns only if it is pinned AND the caller keeps
// it untouched until the batch's rows have been collected -- plain pageable memory is copied synchronously by the runtime.
int vc_stream_stage_host(vc_engine* e, const uint8_t* frames_host, int b, int h, int w, void** frames_dev_out) {
    VC_CHECK(e && frames_host && frames_dev_out, VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_detector, VC_ERR_STATE, "engine not finalized");
    VC_CHECK(b >= 1 && b <= e->cfg.max_batch && h >= 1 && w >= 1 && h <= e->cfg.max_frame_h && w <= e->cfg.max_frame_w, VC_ERR_CAPACITY,
             "batch of %d frames %dx%d exceeds max_batch / max_frame_h / max_frame_w", b, h, w);
    VC_HIP(hipSetDevice(e->cfg.device));
    const size_t bytes = (size_t)b * h * w * 3, slot_bytes = (size_t)e->cfg.max_batch * e->cfg.max_frame_h * e->cfg.max_frame_w * 3;
    if (!e->cstream) {
        VC_HIP(hipStreamCreateWithFlags(&e->cstream, hipStreamNonBlocking));
        for (int i = 0; i < 4; ++i) {
            VC_TRY(dev_alloc(e, (void**)&e->d_ingest[i], slot_bytes));
            VC_HIP(hipEventCreateWithFlags(&e->ev_ingest[i], hipEventDisableTiming));
        }
    }
    // four slots: one batch staged ahead + <= 2 submissions + the batch whose rows are still to be collected
    const int slot = (int)(e->ingest_seq & 3);
    VC_CHECK(!e->ingest_staged[slot], VC_ERR_STATE, "four host batches are staged and none has been submitted: call vc_stream_submit");
    for (const auto& pd : e->pending)
        VC_CHECK(pd.frames != e->d_ingest[slot], VC_ERR_STATE, "the staging slot's previous batch is still waiting for vc_stream_run: at most four host batches may be alive");
    for (const auto& job : e->jobs)                      // its crops may still be being cut on the ReID stream
        VC_CHECK(job.frames != e->d_ingest[slot], VC_ERR_STATE, "the staging slot's previous batch has not been collected: at most four host batches may be alive");
    ++e->ingest_seq;
    VC_HIP(hipMemcpyAsync(e->d_ingest[slot], frames_host, bytes, hipMemcpyHostToDevice, e->cstream));
    VC_HIP(hipEventRecord(e->ev_ingest[slot], e->cstream));
    e->ingest_staged[slot] = true;
    *frames_dev_out = e->d_ingest[slot];
    return VC_OK;
}

int vc_stream_submit_host(vc_engine* e, const uint8_t* frames_host, int b, int h, int w, void** frames_dev_out) {
    VC_CHECK(e && e->pending.size() < 2, e ? VC_ERR_STATE : VC_ERR_ARG, "two submissions are already in flight: call vc_stream_run first");
    VC_TRY(vc_stream_stage_host(e, frames_host, b, h, w, frames_dev_out));
    return vc_stream_submit(e, *frames_dev_out, b, h, w);
}

}  // extern "C"

namespace {

// Second pipeline stage of a submission whose detector has finished: marshal the detections like networks/yolo.py:72-97,
// cut the crops of every box of the batch (deep_sort.py:89-95,119-129) and enqueue the ReID net on its own stream.
int issue_reid(vc_engine* e, vc_engine::Pending& pd) {
    const int md = e->cfg.max_det, b = pd.b, h = pd.h, w = pd.w;
    const float* h_det = e->h_det2[pd.slot];
    const int* h_cnt = e->h_det_count2[pd.slot];
    pd.fd.assign(b, FrameDets{});
    pd.row0.assign(b, 0);
    for (int f = 0; f < b; ++f) {
        VC_CHECK(h_cnt[f] >= 0 || pd.inj_b > 0, VC_ERR_CAPACITY, "frame %d: more than max_candidates (%d) boxes passed conf_thres; raise vc_engine_config.max_candidates", f,
                 e->cfg.max_candidates);
        if (pd.inj_b > 0) {
            const int fi = f % pd.inj_b;
            marshal(pd.inj_det.data() + (size_t)fi * pd.inj_n * 6, pd.inj_cnt[fi], pd.fd[f]);
        } else {
            marshal(h_det + (size_t)f * md * 6, h_cnt[f], pd.fd[f]);
        }
    }
    // every check comes before a feature / crop slot is taken: a refused batch leaves the three-buffer rotation untouched
    std::vector<int>& crops = e->crop_scratch;
    crops.clear();
    int k = 0;
    for (int f = 0; f < b; ++f) {
        pd.row0[f] = k;
        FrameDets& d = pd.fd[f];
        VC_CHECK(k + (int)d.conf.size() <= e->cfg.max_crops, VC_ERR_CAPACITY,
                 "the batch has more boxes than max_crops (%d): raise vc_engine_config.max_crops", e->cfg.max_crops);
        for (size_t i = 0; i < d.conf.size(); ++i) {
            const double* bx = &d.xyxy[i * 4];
            const double bw = bx[2] - bx[0], bh = bx[3] - bx[1];                 // deep_sort.py:78-87
            const double cx = bx[0] + bw / 2, cy = bx[1] + bh / 2;
            int c[5];
            c[0] = f;
            c[1] = std::max((int)(cx - bw / 2), 0); c[3] = std::min((int)(cx + bw / 2), w - 1);    // deep_sort.py:89-95
            c[2] = std::max((int)(cy - bh / 2), 0); c[4] = std::min((int)(cy + bh / 2), h - 1);
            VC_CHECK(c[3] > c[1] && c[4] > c[2], VC_ERR_ARG,
                     "frame %d box %zu gives an empty crop (the reference's cv2.resize raises here)", f, i);
            crops.insert(crops.end(), c, c + 5);
            ++k;
        }
    }
    pd.fslot = (int)(e->reid_seq++ % 3);
    int* hc = e->h_crops2[pd.fslot];
    if (k > 0) {
        memcpy(hc, crops.data(), (size_t)k * 5 * sizeof(int));
        VC_HIP(hipMemcpyAsync(e->d_crops2[pd.fslot], hc, (size_t)k * 5 * sizeof(int), hipMemcpyHostToDevice, e->rstream));
        VC_TRY(run_reid_on(e, (const uint8_t*)pd.frames, h, w, k, e->d_crops2[pd.fslot], e->d_feat2[pd.fslot], e->rstream));
    }
    VC_HIP(hipEventRecord(e->ev_reid[pd.fslot], e->rstream));
    pd.stage = 1;
    return VC_OK;
}

// A submission whose detections cannot be embedded (candidate overflow, more boxes than max_crops, an empty crop) is DROPPED: the
// error is reported once, by the call that CONSUMES that submission (vc_stream_run* / vc_stream_embed for its frames), and the stream
// continues with the next one (it used to stay at the front of the queue and fail every following call).
int issue_reid_or_drop(vc_engine* e, size_t idx) {
    const int st = issue_reid(e, e->pending[idx]);
    if (st != VC_OK) e->pending.erase(e->pending.begin() + idx);
    return st;
}

// Look-ahead: if the detector of the next submission has finished, start its ReID now (it then overlaps the tracking in progress).
// Never fails on behalf of the look-ahead submission (a drop of batch n + 1 used to come back as the status of the
// run_async / collect call made for batch n -- whose rows were then lost and whose job stayed queued behind the caller's back).  A
// submission that cannot be embedded stays queued at stage 0 with `embed_refused` set; take_front runs the same checks again when the
// caller asks for THAT batch, reports the error there and drops it.  issue_reid checks everything before it takes a feature / crop
// slot, so the refused attempt leaves no state behind.
void try_issue_next(vc_engine* e) {
    int next = -1;
    int embedded = 0;                                   // batches that own one of the three feature buffers
    for (size_t i = 0; i < e->pending.size(); ++i) {
        if (e->pending[i].stage == 0) { next = (int)i; break; }
        ++embedded;
    }
    if (next < 0 || e->pending[next].embed_refused) return;
    embedded += (int)e->jobs.size();
    if (embedded >= 3) return;
    if (hipEventQuery(e->ev_det[e->pending[next].slot]) != hipSuccess) return;
    if (issue_reid(e, e->pending[next]) != VC_OK) e->pending[next].embed_refused = true;
}

}  // namespace

extern "C" {

// The tracker work of one batch whose ReID has been enqueued (pd.stage == 1): per frame, in class order, the detections
// VideoTracker.run would hand to that class's DeepSort.update (modules/track.py:50-59; frames without boxes are skipped,
// modules/__init__.py:68-69, Q1), prepared on the host (confidence filter + DeepSORT NMS do not depend on tracker state), then ONE
// kernel on the tracker stream that steps every tracker through the whole batch (track_kernels.hip).  No host round trip per frame.
// Multi-camera batches: frame f of the batch belongs to camera cam_of_frame[f] (nullptr: one camera) and is stepped on that camera's
// trackers, trackers[cam * num_classes + c]; the frames of one camera appear in the batch in stream order.  One engine (one weight
// copy, one detector launch per layer) then serves S cameras with B / S frames of latency each, and the tracker kernel walks
// S x num_classes trackers in parallel, B / S steps each (a new VideoTracker per video, /root/reference/modules/__init__.py:29-36).
static int enqueue_batch_tracking(vc_engine* e, vc_engine::Pending& pd, const int* all_trackers, int num_classes, const int* cam_of_frame, int n_cam,
                                  int stage, int cap_rows_per_frame, std::vector<int>& ndet) {
    const int b = pd.b;
    ndet.assign(b, 0);
    std::vector<std::vector<FrameClassDets>> frames(b);
    std::vector<std::vector<int>> by_class(num_classes);
    for (int f = 0; f < b; ++f) {
        const int cam = cam_of_frame ? cam_of_frame[f] : 0;
        VC_CHECK(cam >= 0 && cam < n_cam, VC_ERR_ARG, "frame %d: camera index %d outside [0, %d)", f, cam, n_cam);
        const int* trackers = all_trackers + (size_t)cam * num_classes;
        FrameDets& d = pd.fd[f];
        ndet[f] = (int)d.conf.size();
        if (d.conf.empty()) continue;                                                // Q1
        for (size_t i = 0; i < d.label.size(); ++i)
            if (d.label[i] >= 0 && d.label[i] < num_classes) by_class[d.label[i]].push_back((int)i);
        for (int c = 0; c < num_classes; ++c) {
            std::vector<int>& g = by_class[c];
            if (g.empty()) continue;
            const int ts = tracker_slot(e->trackers, trackers[c]);
            VC_CHECK(ts >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle %d for class %d", trackers[c], c);
            std::vector<double> bx(g.size() * 4), cf(g.size());
            std::vector<int> rows(g.size());
            for (size_t i = 0; i < g.size(); ++i) {
                memcpy(&bx[i * 4], &d.xyxy[(size_t)g[i] * 4], 4 * sizeof(double));
                cf[i] = d.conf[g[i]];
                rows[i] = pd.row0[f] + g[i];
            }
            FrameClassDets fc{c, ts, {}};
            prepare_dets(bx.data(), cf.data(), rows.data(), (int)g.size(), e->trackers[ts]->p, fc.dets);
            frames[f].push_back(std::move(fc));
            g.clear();
        }
    }
    g_tm.lap(1);
    VC_TRY(track_enqueue(e, stage, frames, e->d_feat2[pd.fslot], pd.w, pd.h, b * cap_rows_per_frame, e->ev_reid[pd.fslot]));
    g_tm.lap(2);
    return VC_OK;
}

// Pop the oldest submission once its detector has finished and its ReID is enqueued.
static int take_front(vc_engine* e, const void* frames_dev, int b, int h, int w, vc_engine::Pending& out) {
    if (e->pending.empty()) VC_TRY(vc_stream_submit(e, frames_dev, b, h, w));
    {
        const vc_engine::Pending& fr = e->pending.front();
        VC_CHECK(fr.frames == frames_dev && fr.b == b && fr.h == h && fr.w == w, VC_ERR_STATE,
                 "vc_stream_run must consume submissions in the order they were made");
    }
    g_tm.start();
    if (e->pending.front().stage == 0) {
        VC_HIP(hipEventSynchronize(e->ev_det[e->pending.front().slot]));
        VC_TRY(issue_reid_or_drop(e, 0));
    }
    out = std::move(e->pending.front());
    e->pending.erase(e->pending.begin());
    g_tm.lap(0);
    return VC_OK;
}

}  // extern "C" (helpers above have internal linkage)

namespace vc {
// Blocking entry points that touch tracker state wait for every tracker batch in flight (their rows stay in pinned memory until
// vc_stream_collect picks them up).
int async_wait_all(vc_engine* e) { return track_idle(e); }
}  // namespace vc

extern "C" {

// Asynchronous form: the batch's tracker work is enqueued on the tracker stream (it starts on the GPU when the batch's ReID has
// finished) and the call returns; the caller goes on to submit / embed the following batches; rows are picked up in order with
// vc_stream_collect.  At most two batches may be outstanding.
int vc_stream_run_async(vc_engine* e, const int* trackers, int num_classes, const void* frames_dev, int b, int h, int w,
                        int cap_rows_per_frame) {
    return vc_stream_run_async_multi(e, trackers, 1, num_classes, nullptr, frames_dev, b, h, w, cap_rows_per_frame);
}

int vc_stream_run_async_multi(vc_engine* e, const int* trackers, int n_cam, int num_classes, const int* cam_of_frame, const void* frames_dev, int b, int h,
                              int w, int cap_rows_per_frame) {
    VC_CHECK(e && trackers && frames_dev && cap_rows_per_frame > 0 && n_cam >= 1 && (n_cam == 1 || cam_of_frame), VC_ERR_ARG, "bad argument");
    VC_CHECK(e->finalized && e->cfg.with_detector && e->cfg.with_reid, VC_ERR_STATE, "engine not finalized");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_CHECK(e->jobs.size() < 2, VC_ERR_STATE, "two asynchronous batches are already outstanding: call vc_stream_collect");
    vc_engine::Pending pd;
    VC_TRY(take_front(e, frames_dev, b, h, w, pd));
    vc_engine::AsyncJob job;
    job.stage = (int)(e->tstage_seq++ % 3); job.b = b; job.cap = cap_rows_per_frame; job.frames = frames_dev;
    VC_TRY(enqueue_batch_tracking(e, pd, trackers, num_classes, cam_of_frame, n_cam, job.stage, cap_rows_per_frame, job.ndet));
    e->jobs.push_back(std::move(job));
    g_tm.report();
    try_issue_next(e);
    return VC_OK;
}

// Results of the oldest asynchronous batch (blocks until its tracker kernel has finished).  While waiting, the ReID of the
// next submission is started as soon as its detector has finished.
int vc_stream_collect(vc_engine* e, int64_t* out_rows6, int cap_rows_per_frame, int* out_m, int* out_ndet, int b) {
    VC_CHECK(e && out_rows6 && out_m, VC_ERR_ARG, "null argument");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_CHECK(!e->jobs.empty(), VC_ERR_STATE, "no asynchronous batch is outstanding");
    const vc_engine::AsyncJob& front = e->jobs.front();
    VC_CHECK(front.b == b && front.cap == cap_rows_per_frame, VC_ERR_ARG, "collect: batch of %d frames x %d rows expected", front.b, front.cap);
    while (hipEventQuery(e->tstage[front.stage].done) == hipErrorNotReady) {
        try_issue_next(e);
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    vc_engine::AsyncJob job = std::move(e->jobs.front());
    e->jobs.pop_front();
    VC_TRY(track_collect(e, job.stage, out_rows6, cap_rows_per_frame, out_m));
    if (out_ndet) memcpy(out_ndet, job.ndet.data(), (size_t)b * sizeof(int));
    try_issue_next(e);
    return VC_OK;
}

int vc_stream_run(vc_engine* e, const int* trackers, int num_classes, const void* frames_dev, int b, int h, int w,
                  int64_t* out_rows6, int cap_rows_per_frame, int* out_m, int* out_ndet) {
    VC_CHECK(e && trackers && frames_dev && out_rows6 && out_m, VC_ERR_ARG, "null argument");
    VC_CHECK(e->jobs.empty(), VC_ERR_STATE, "asynchronous batches are outstanding: vc_stream_collect them first");
    VC_TRY(vc_stream_run_async(e, trackers, num_classes, frames_dev, b, h, w, cap_rows_per_frame));
    return vc_stream_collect(e, out_rows6, cap_rows_per_frame, out_m, out_ndet, b);
}

// ---- one stream on several GPUs: the stateless front end shards by frame chunk over the ranks -----------------------
// Front half of the fused path for the OLDEST submission: wait for its detector, marshal the detections like networks/yolo.py:72-97,
// cut the crops and run the ReID net for every box (deep_sort.py:119-129, Q5), and hand back what VideoTracker.run works on: rows
// [frame index in the batch, x1, y1, x2, y2, conf, label] (float64; a frame without boxes contributes nothing, Q1) plus the device
// address of the matching [n][512] float32 embeddings.  The embeddings stay valid until the third following vc_stream_embed /
// vc_stream_run* call (three feature buffers rotate).
int vc_stream_embed(vc_engine* e, const void* frames_dev, int b, int h, int w, double* out_rows7, int cap_rows, int* out_n, const float** out_feat_dev) {
    VC_CHECK(e && frames_dev && out_rows7 && out_n && out_feat_dev, VC_ERR_ARG, "null argument");
    VC_CHECK(e->finalized && e->cfg.with_detector && e->cfg.with_reid, VC_ERR_STATE, "engine not finalized");
    VC_HIP(hipSetDevice(e->cfg.device));
    vc_engine::Pending pd;
    VC_TRY(take_front(e, frames_dev, b, h, w, pd));
    VC_HIP(hipEventSynchronize(e->ev_reid[pd.fslot]));
    int n = 0;
    for (int f = 0; f < b; ++f) n += (int)pd.fd[f].conf.size();
    VC_CHECK(n <= cap_rows, VC_ERR_CAPACITY, "vc_stream_embed: %d rows, room for %d", n, cap_rows);
    double* o = out_rows7;
    for (int f = 0; f < b; ++f) {
        const FrameDets& d = pd.fd[f];
        for (size_t i = 0; i < d.conf.size(); ++i, o += 7) {
            o[0] = (double)f; memcpy(o + 1, &d.xyxy[i * 4], 4 * sizeof(double)); o[5] = d.conf[i]; o[6] = (double)d.label[i];
        }
    }
    *out_n = n;
    *out_feat_dev = e->d_feat2[pd.fslot];
    return VC_OK;
}

// VideoTracker.run (modules/track.py:30-70) for a run of frames whose detections AND embeddings are supplied -- the tracker rank of the
// frame-sharded front end.  rows7 [n][7] = [frame key, x1, y1, x2, y2, conf, label] sorted by frame key (any ascending integer, e.g. the
// 1-based frame id); row i's embedding is feat_dev[i].  One launch of track_batch_kernel steps every (frame, class with boxes) in order.
// Output per distinct frame key, ascending: out_keys[j], out_m[j] rows [x1, y1, x2, y2, track id, label] at out_rows6[j * cap_rows_per_frame].
int vc_videotracker_run_features(vc_engine* e, const int* trackers, int num_classes, const double* rows7, const float* feat_dev, int n, int h, int w,
                                 int64_t* out_rows6, int cap_rows_per_frame, int* out_m, int64_t* out_keys, int cap_frames, int* out_n_frames) {
    VC_CHECK(e && trackers && out_rows6 && out_m && out_keys && out_n_frames && (n == 0 || (rows7 && feat_dev)) && n >= 0, VC_ERR_ARG, "bad argument");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_TRY(async_wait_all(e));
    std::vector<std::vector<FrameClassDets>> frames;
    std::vector<int64_t> keys;
    std::vector<std::vector<int>> by_class(num_classes);
    for (int i0 = 0; i0 < n;) {
        const double key = rows7[(size_t)i0 * 7];
        int i1 = i0;
        while (i1 < n && rows7[(size_t)i1 * 7] == key) ++i1;
        VC_CHECK(keys.empty() || (int64_t)key > keys.back(), VC_ERR_ARG, "rows must be sorted by frame key");
        keys.push_back((int64_t)key);
        frames.emplace_back();
        for (int i = i0; i < i1; ++i) {
            const int lb = (int)rows7[(size_t)i * 7 + 6];
            if (lb >= 0 && lb < num_classes) by_class[lb].push_back(i);
        }
        for (int c = 0; c < num_classes; ++c) {                                        // modules/track.py:50-59
            std::vector<int>& g = by_class[c];
            if (g.empty()) continue;
            const int ts = tracker_slot(e->trackers, trackers[c]);
            VC_CHECK(ts >= 0, VC_ERR_NOTFOUND, "bad or stale tracker handle %d for class %d", trackers[c], c);
            std::vector<double> bx(g.size() * 4), cf(g.size());
            for (size_t k = 0; k < g.size(); ++k) {
                memcpy(&bx[k * 4], rows7 + (size_t)g[k] * 7 + 1, 4 * sizeof(double));
                cf[k] = rows7[(size_t)g[k] * 7 + 5];
            }
            FrameClassDets fc{c, ts, {}};
            prepare_dets(bx.data(), cf.data(), g.data(), (int)g.size(), e->trackers[ts]->p, fc.dets);
            frames.back().push_back(std::move(fc));
            g.clear();
        }
        i0 = i1;
    }
    const int nf = (int)frames.size();
    VC_CHECK(nf <= cap_frames, VC_ERR_CAPACITY, "%d frames, room for %d", nf, cap_frames);
    *out_n_frames = nf;
    for (int j = 0; j < nf; ++j) out_keys[j] = keys[j];
    if (nf == 0) return VC_OK;
    VC_TRY(track_enqueue(e, 3, frames, feat_dev, w, h, nf * cap_rows_per_frame, nullptr));
    return track_collect(e, 3, out_rows6, cap_rows_per_frame, out_m);
}

// Abandon everything in flight on the stream path: waits for the GPU, discards the rows of uncollected batches (their tracker steps
// HAVE run: reset the trackers as well if the clip is to be replayed) and empties the submission queue.  The way out of any error
// state of vc_stream_* without destroying the engine.
int vc_stream_reset(vc_engine* e) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    VC_HIP(hipSetDevice(e->cfg.device));
    VC_HIP(hipStreamSynchronize(e->dstream)); VC_HIP(hipStreamSynchronize(e->rstream)); VC_HIP(hipStreamSynchronize(e->stream));
    while (!e->jobs.empty()) {
        const vc_engine::AsyncJob job = std::move(e->jobs.front());
        e->jobs.pop_front();
        std::vector<int64_t> rows((size_t)job.b * job.cap * 6);
        std::vector<int> m(job.b);
        (void)track_collect(e, job.stage, rows.data(), job.cap, m.data());      // bookkeeping (pending_dets, known_tracks); errors are dropped with the rows
    }
    e->pending.clear();
    if (e->cstream) VC_HIP(hipStreamSynchronize(e->cstream));         // staged host batches are abandoned with the rest
    for (bool& st : e->ingest_staged) st = false;
    return VC_OK;
}

// Zone filter of VideoCounting.run (/root/reference/modules/track.py:102-104 -> utilities/counting/bb_polygon.py:14-114):
// inside[i] = any corner of boxes[i] = (x1,y1,x2,y2) lies in the polygon (ray cast to y = 1e9, exact double compares).
// Host-only (the post-pass runs once per video on a few thousand rows); same arithmetic as counting.py.
namespace {
struct P2 { double x, y; };
inline int orient(P2 p, P2 q, P2 r) {
    const double v = (q.y - p.y) * (r.x - q.x) - (q.x - p.x) * (r.y - q.y);
    return v == 0 ? 0 : (v > 0 ? 1 : 2);
}
inline bool on_segment(P2 p, P2 q, P2 r) {
    return q.x <= std::max(p.x, r.x) && q.x >= std::min(p.x, r.x) && q.y <= std::max(p.y, r.y) && q.y >= std::min(p.y, r.y);
}
inline bool intersect(P2 p1, P2 q1, P2 p2, P2 q2) {
    const int o1 = orient(p1, q1, p2), o2 = orient(p1, q1, q2), o3 = orient(p2, q2, p1), o4 = orient(p2, q2, q1);
    if (o1 != o2 && o3 != o4) return true;
    if (o1 == 0 && on_segment(p1, p2, q1)) return true;
    if (o2 == 0 && on_segment(p1, q2, q1)) return true;
    if (o3 == 0 && on_segment(p2, p1, q2)) return true;
    return o4 == 0 && on_segment(p2, q1, q2);
}
bool point_in_polygon(const P2* poly, int n, P2 pt) {
    const P2 far{pt.x, 1e9};
    int count = 0;
    for (int i = 0; i < n; ++i) {
        const P2 a = poly[i], b = poly[(i + 1) % n];
        if (intersect(a, b, pt, far)) {
            if (orient(a, pt, b) == 0) return on_segment(a, pt, b);
            ++count;
        }
    }
    return count % 2 == 1;
}
}  // namespace

int vc_zone_filter_host(const double* polygon_xy, int n_points, const int64_t* boxes_xyxy, int n, uint8_t* inside) {
    VC_CHECK(polygon_xy && n_points >= 1 && (n == 0 || (boxes_xyxy && inside)), VC_ERR_ARG, "bad argument");
    const P2* poly = (const P2*)polygon_xy;
    for (int i = 0; i < n; ++i) {
        const double x1 = (double)boxes_xyxy[i * 4], y1 = (double)boxes_xyxy[i * 4 + 1], x2 = (double)boxes_xyxy[i * 4 + 2], y2 = (double)boxes_xyxy[i * 4 + 3];
        inside[i] = point_in_polygon(poly, n_points, P2{x1, y1}) || point_in_polygon(poly, n_points, P2{x2, y1}) ||
                    point_in_polygon(poly, n_points, P2{x2, y2}) || point_in_polygon(poly, n_points, P2{x1, y2});
    }
    return VC_OK;
}

// Greedy class-offset NMS on an explicit candidate list (reference order), through the same three kernels the
// detector uses.  Output rows [x1,y1,x2,y2,conf,cls] (network pixels, no rescale).
int vc_nms_host(const float* boxes4, const float* conf, const int* cls, int n, float iou, int max_det, int max_cand, float* out6,
                int* out_n) {
    VC_CHECK(boxes4 && conf && cls && out6 && out_n && n >= 0, VC_ERR_ARG, "bad argument");
    VC_CHECK(max_cand % 64 == 0 && max_cand >= 64 && max_cand <= 8192 && n <= max_cand, VC_ERR_ARG, "max_cand must be a multiple of 64 in [64,8192] and >= n");
    vc_engine tmp;
    DetectPostBuffers pb{};
    float* geom = nullptr;
    int st = VC_OK;
    const size_t mc = max_cand;
    auto A = [&](void** p, size_t bytes) { if (st == VC_OK) st = dev_alloc(&tmp, p, bytes); };
    A((void**)&pb.cand_box, mc * 16); A((void**)&pb.cand_conf, mc * 4); A((void**)&pb.cand_cls, mc * 4); A((void**)&pb.cand_idx, mc * 4);
    A((void**)&pb.cand_count, 4); A((void**)&pb.sort_box, mc * 16); A((void**)&pb.sort_conf, mc * 4); A((void**)&pb.sort_cls, mc * 4);
    A((void**)&pb.mask, mc * (mc / 64) * 8); A((void**)&pb.det, (size_t)max_det * 24); A((void**)&pb.det_count, 4); A((void**)&pb.overflow, 4);
    A((void**)&geom, 20);
    if (st == VC_OK && hipMemset(pb.overflow, 0, 4) != hipSuccess) { set_error("memset failed"); st = VC_ERR_HIP; }
    if (st == VC_OK) {
        std::vector<int> idx(n);
        std::iota(idx.begin(), idx.end(), 0);
        const float g[5] = {1.f, 0.f, 0.f, 1e30f, 1e30f};
        bool ok = hipMemcpy(pb.cand_box, boxes4, (size_t)n * 16, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(pb.cand_conf, conf, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(pb.cand_cls, cls, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(pb.cand_idx, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(pb.cand_count, &n, 4, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(geom, g, 20, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) { set_error("upload failed"); st = VC_ERR_HIP; }
    }
    if (st == VC_OK) st = launch_nms(1, max_cand, max_det, iou, geom, pb, nullptr);
    if (st == VC_OK && hipDeviceSynchronize() != hipSuccess) { set_error("nms kernels failed: %s", hipGetErrorString(hipGetLastError())); st = VC_ERR_HIP; }
    if (st == VC_OK) {
        if (hipMemcpy(out_n, pb.det_count, 4, hipMemcpyDeviceToHost) != hipSuccess || *out_n < 0 ||
            hipMemcpy(out6, pb.det, (size_t)std::min(*out_n, max_det) * 24, hipMemcpyDeviceToHost) != hipSuccess) { set_error("download failed"); st = VC_ERR_HIP; }
    }
    for (void* q : tmp.allocs) (void)hipFree(q);
    tmp.allocs.clear();
    return st;
}

}  // extern "C"
