// Engine internals shared between engine.hip (networks), tracker.hip (DeepSORT) and stream.hip (fused path).
#pragma once
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

namespace vc {

struct ConvParam {
    std::string name;
    int O = 0, I = 0, kh = 0, kw = 0;
    bool set = false;
    bool pair_stem = false;          // YOLO stem in bf16: two 4-channel pixels form one 16-byte chunk
    int fuse_a = -1, fuse_b = -1;    // >= 0: internal parameter = rows of params[fuse_a] then params[fuse_b] (same input, 1x1)
    bool hidden = false;             // internal (fused) parameters are not enumerated to the caller
    std::vector<float> w, b;         // host copies (OIHW, BN folded) until finalize()
    void* d_w = nullptr;             // packed [Cout_pad][Kp]
    float* d_b = nullptr;            // [Cout_pad]
    int cin_eff = 0, K = 0, Kp = 0, cout_pad = 0;
};

struct Op {
    enum Kind { CONV, SPPF, UPSAMPLE, MAXPOOL } kind;
    ConvP conv{};
    View a{}, b{};
    int C = 0;
    int cout_logical = 0;            // > 0: the launch covers zero-padded output channels, FLOPs count this many
    int param = -1;                  // index into Net::params for CONV
};

struct Net {
    std::vector<ConvParam> params;
    std::map<std::string, int> index;
    int add(const std::string& name, int O, int I, int kh, int kw) {
        ConvParam p;
        p.name = name; p.O = O; p.I = I; p.kh = kh; p.kw = kw;
        index[name] = (int)params.size();
        params.push_back(p);
        return (int)params.size() - 1;
    }
};

struct ProfCat {
    double ms = 0, flops = 0, bytes = 0;
    int64_t launches = 0;
};

// host-side tracker record (tracker.hip)
struct TrackRec {
    int64_t id;
    int state, hits, age, tsu;
    int slot;                        // index into the device TrackPool
    int gal_count, gal_head;         // ring of the last `budget` features
    double last_conf;
};

struct Tracker {
    vc_tracker_params p;
    std::vector<TrackRec> tracks;
    int64_t next_id = 1;
};

}  // namespace vc

struct vc_engine {
    vc_engine_config cfg{};
    int prec = 0;
    hipStream_t stream = nullptr;    // ReID + tracker
    hipStream_t dstream = nullptr;   // detector (runs ahead of the tracker on the next batch)
    hipStream_t rstream = nullptr;   // ReID of the next batch (stream path), concurrent with detector and tracker
    hipEvent_t ev_det[2] = {nullptr, nullptr};
    hipEvent_t ev_reid[3] = {nullptr, nullptr, nullptr};
    bool finalized = false;
    std::vector<void*> allocs;       // everything hipMalloc'ed, freed on destroy
    std::vector<void*> host_allocs;  // hipHostMalloc'ed

    // ---- detector --------------------------------------------------------------------------------
    vc::Net yolo;
    int ch[5] = {0, 0, 0, 0, 0}, rep[4] = {0, 0, 0, 0};
    std::map<std::string, vc::View> ybuf;        // named activation buffers (max shape)
    vc::View layer_view[24];
    uint8_t* d_frames = nullptr;                 // staging for host images / stream frames
    size_t d_frames_bytes = 0;
    float* d_logits[3] = {nullptr, nullptr, nullptr};
    float anchors[3][6];                         // Detect anchors in pixels (default: the COCO set of yolov5{s,m,l}.yaml)
    vc::DetectPostBuffers post{};
    float* d_geom = nullptr;                     // [max_batch][5] gain, padw, padh, src_w, src_h
    float* d_pred_debug = nullptr;
    bool want_pred_debug = false;
    int last_B = 0, last_nh = 0, last_nw = 0, last_ntotal = 0;
    // letterbox folded into the stem (stem_direct.hip, U8): source of the running / last detector pass; the letterboxed tensor
    // ybuf["in"] is then only produced on demand (vc_detect_debug_layer(-1))
    const uint8_t* stem_src = nullptr;
    vc::LetterboxGeom stem_geom{};
    bool in_stale = false;
    float* h_det = nullptr;                      // pinned [max_batch][max_det][6]
    int* h_det_count = nullptr;                  // pinned [max_batch]
    float* h_det2[2] = {nullptr, nullptr};       // pinned detector outputs of the (up to) two submissions in flight
    int* h_det_count2[2] = {nullptr, nullptr};
    float* h_geom = nullptr;                     // pinned [2][max_batch][5]
    unsigned geom_seq = 0, submit_seq = 0;
    // stream path: three feature / crop buffers -- the batch being tracked (possibly by the worker thread), the batch
    // whose ReID is running, and the one after it
    float* d_feat2[3] = {nullptr, nullptr, nullptr};
    int* d_crops2[3] = {nullptr, nullptr, nullptr};
    int* h_crops2[3] = {nullptr, nullptr, nullptr};
    unsigned reid_seq = 0;
    struct FrameDets { std::vector<double> xyxy, conf; std::vector<int> label; };
    struct Pending {
        const void* frames; int b, h, w, slot;
        int stage = 0;                   // 0: detector enqueued, 1: ReID enqueued as well
        int fslot = 0;                   // feature / crop buffer of this batch
        std::vector<FrameDets> fd;       // per frame, as VideoTracker.run sees them
        std::vector<int> row0;           // first feature row of each frame
    };
    std::vector<Pending> pending;
    // asynchronous tracking (vc_stream_run_async / vc_stream_collect): the tracker loop of a batch runs on a worker thread
    struct AsyncJob {
        Pending pd;
        std::vector<int> trackers;
        int num_classes = 0, b = 0, h = 0, w = 0, cap = 0;
        std::vector<int64_t> rows6;
        std::vector<int> m, ndet;
        int status = 0;
        std::string err;
        bool done = false;
    };
    std::thread worker;
    std::mutex jmu;
    std::condition_variable jcv;
    std::deque<std::unique_ptr<AsyncJob>> jobs;      // submission order; front = next to collect
    bool worker_quit = false;

    // ---- ReID ---------------------------------------------------------------------------------------
    vc::Net reid;
    std::map<std::string, vc::View> rbuf;
    int* d_crops = nullptr;                      // [max_crops][5]
    int* h_crops = nullptr;
    float* d_feat = nullptr;                     // [max_crops][512]
    float* h_feat = nullptr;
    float* d_reid_in_nchw = nullptr;

    // ---- tracker pool ---------------------------------------------------------------------------------
    vc::TrackPool pool{};
    std::vector<int> free_slots;
    std::vector<std::unique_ptr<vc::Tracker>> trackers;
    // per-step scratch: one pinned staging block mirrored on the device, cost matrices, posterior means
    // pinned + device-mapped (hd_* = device alias): phase A block, phase B block, cost rows, posterior means
    char* h_stage = nullptr; char* hd_stage = nullptr; size_t stage_cap = 0;
    char* h_stage2 = nullptr; char* hd_stage2 = nullptr;
    double* h_cost = nullptr; double* hd_cost = nullptr;
    double* h_mean = nullptr; double* hd_mean = nullptr;
    float* d_feat_in = nullptr;                  // features handed in from the host (vc_tracker_step)
    std::vector<int> slot_chain;                 // scratch of track_launch: slot -> chain index, -1 outside a call
    unsigned* d_track_counter = nullptr;         // workgroups of the running tracker kernel that have finished
    unsigned* h_track_flag = nullptr; unsigned* hd_track_flag = nullptr;   // pinned completion word (sequence number)
    unsigned track_seq = 0;
    bool track_inflight = false;
    std::vector<vc::TrackChainRec> chain_scratch;
    size_t cost_cap = 0;
    int det_cap = 0;

    // ---- measurement ----------------------------------------------------------------------------------
    bool profiling = false;
    // in-flight conv profiling (vc_profile_enable(e, 2)): event pairs recorded on the launch stream WITHOUT synchronising, so the
    // timed steps keep their three overlapping streams; resolved by vc_profile_read.  Only the thread that issues convs uses it.
    bool prof_async = false;
    struct ProfPair { hipEvent_t a, b; double flops, bytes; };
    std::vector<ProfPair> prof_pairs;
    size_t prof_used = 0;
    double prof_conv_union_ms = 0, prof_conv_span_ms = 0;     // set when the in-flight pairs are resolved (vc_profile_read)
    vc::ProfCat prof[VC_PROF_NCAT];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, int> tuned;            // conv autotune cache: shape signature -> tile config (engine.hip::tune_key)
    bool tuned_dirty = false;
    std::string op_log;                          // per-launch lines "name M N K tile ms" while profiling
    double last_ms = 0;

    // ---- stream injection -----------------------------------------------------------------------------
    std::vector<float> inject_det;
    std::vector<int> inject_count;
    int inject_b = 0, inject_n = 0;
};

namespace vc {
// engine.hip
int dev_alloc(vc_engine* e, void** p, size_t bytes);
int host_alloc(vc_engine* e, void** p, size_t bytes);
int run_detector_dev(vc_engine* e, const uint8_t* d_frames, int B, int h, int w, bool swap_rb);   // frames same size, on device
int run_reid_dev(vc_engine* e, const uint8_t* d_frames, int H, int W, int k);                      // crops in e->d_crops -> e->d_feat
int run_reid_on(vc_engine* e, const uint8_t* d_frames, int H, int W, int k, const int* d_crops, float* feat_out, hipStream_t rs);
int prof_launch(vc_engine* e, int cat, double flops, double bytes, int status);
struct ProfScope {
    vc_engine* e; int cat; double flops, bytes;
    hipStream_t s;
    ProfScope(vc_engine* e_, int cat_, double flops_ = 0, double bytes_ = 0, hipStream_t s_ = nullptr);
    ~ProfScope();
};
// tracker.hip
int tracker_init_pool(vc_engine* e);
struct Prepared { std::vector<double> tlwh, conf; std::vector<int> feat_rows; };   // filtered + NMS'ed detections of one tracker
struct StepCtx {
    std::vector<int> ids, labels;          // trackers stepped together (the classes of one frame) and their labels
    std::vector<Prepared> prep;
    int W = 0, H = 0;
    bool all_means = false;
    // phase A products
    int n_dets = 0, n_app = 0, n_iou = 0;
    size_t out = 0;
    std::vector<int> det_base, featrow;
    std::vector<std::vector<int>> app_job, iou_job;
    std::vector<double> det_xyah;
    int n_jobs = 0;                        // cost jobs staged in e->h_stage (+ their device aliases)
    const TrackJobA* h_jobs = nullptr; const TrackJobA* d_jobs = nullptr;
    const int* d_featrow = nullptr; const double* d_xyah = nullptr; const double* d_tlwh = nullptr;
    // phase B products
    std::vector<TrackOpB> ops;             // pending device operations (carried by the next track_launch)
    struct Emit { int row; int64_t id; int label; };
    std::vector<Emit> emit;
    std::vector<int> mean_offsets;
};
int track_prepare_a(vc_engine* e, StepCtx& c);
int track_host_b(vc_engine* e, StepCtx& c);
int track_launch(vc_engine* e, const StepCtx* cb, const float* feat_b, const StepCtx* ca, const float* feat_a);
int track_wait(vc_engine* e);
int async_wait_all(vc_engine* e);          // stream.hip: block until the worker thread has finished every queued batch
void emit_rows(const StepCtx& c, const double* means, std::vector<int64_t>& rows6);
void build_ctx(vc_engine* e, StepCtx& c, int H, int W, const std::vector<int>& tracker_ids, const std::vector<int>& labels,
               const std::vector<std::vector<int>>& groups, const double* xyxy, const double* conf, int feat_row0);
int frame_track(vc_engine* e, const uint8_t* d_frame_base, int frame_index, int H, int W, const std::vector<int>& tracker_ids,
                const std::vector<int>& labels, const std::vector<std::vector<int>>& groups, const double* xyxy, const double* conf,
                int n, std::vector<int64_t>& rows6);
int lap_solve(const double* cost, int nr, int nc, std::vector<int>& row_of_col_rows, std::vector<int>& cols);
void dsort_nms(const double* tlwh, const double* scores, int n, double max_overlap, std::vector<int>& keep);
}  // namespace vc
