/* libvcount_hip.so -- C ABI of the MI355X-native detect + track hot path of kaylode/vehicle-counting.
 *
 * The reference is pure Python and has no FFI layer; its seam is the duck-typed stage API that
 * modules/__init__.py:54-84 (CountingPipeline.run) calls.  Each group of entry points below is what a
 * ctypes binding of one of those Python operators would call (see INTEGRATION.md for the stubs):
 *
 *   vc_detect*            <- modules/detect.py:30-60 ImageDetect.run -> networks/detector.py:36-38
 *                            -> networks/yolo.py:68-99 YoloBackbone.detect (AutoShape forward: letterbox, YOLOv5
 *                            v6.0 conv stack, Detect decode, NMS, scale_coords)
 *   vc_embed              <- networks/deepsort/deep/feature_extractor.py:42-47 Extractor.__call__ on the crops
 *                            cut by networks/deepsort/deep_sort.py:119-129 DeepSort._get_features
 *   vc_tracker_* / vc_deepsort_update
 *                         <- networks/deepsort/deep_sort.py:25-59 DeepSort.update
 *                            (sort/tracker.py:50-91 Tracker.predict/update and everything below it)
 *   vc_videotracker_run   <- modules/track.py:30-70 VideoTracker.run (one DeepSORT per class)
 *   vc_stream_*           <- the per-frame body of modules/__init__.py:54-84 for device-resident frames
 *   vc_*_host             <- single-function entry points used by the parity tests (one reference function each)
 *
 * Conventions: every function returns an int status (VC_OK == 0); vc_last_error() gives the message for
 * the calling thread.  Handles are opaque, thread-compatible (not thread-safe), own their device memory
 * and HIP streams, and never call back into the caller.  All array arguments are caller-owned plain
 * host pointers unless the name says `_dev`; outputs are caller-allocated.  No torch types anywhere.
 */
#ifndef VCOUNT_HIP_H
#define VCOUNT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_OK 0
#define VC_ERR_ARG 1       /* bad argument / shape */
#define VC_ERR_HIP 2       /* HIP runtime error (no device, OOM, launch failure) */
#define VC_ERR_STATE 3     /* call order (e.g. run before finalize) */
#define VC_ERR_CAPACITY 4  /* a configured capacity (tracks, candidates, crops) was exceeded */
#define VC_ERR_NOTFOUND 5  /* unknown parameter name / handle id */

#define VC_PREC_BF16 0     /* bf16 activations/weights, fp32 accumulate (throughput mode) */
#define VC_PREC_F32 1      /* fp32 everywhere (tight-parity mode) */
#define VC_PREC_FP8 2      /* detector convs on the MX-scaled fp8 MFMA (OCP e4m3fn activations, per-channel-scaled e4m3fn weights,
                              fp32 accumulate); the 3-channel stem, the Detect logits and the ReID net stay bf16 */

#define VC_FEAT_DIM 512    /* networks/deepsort/deep/model.py: embedding width */
#define VC_REID_SIZE 50    /* feature_extractor.py:18 */

int vc_version(void);
const char* vc_last_error(void);
int vc_device_count(int* n);

/* ------------------------------------------------------------------------------------------------
 * Engine: one per process / GPU.  Holds the detector, the ReID net, the tracker pool and the streams.
 * ---------------------------------------------------------------------------------------------- */
typedef struct vc_engine vc_engine;

typedef struct vc_engine_config {
    int device;            /* HIP device ordinal */
    int precision;         /* VC_PREC_* */
    int yolo_variant;      /* 0 = yolov5s, 1 = yolov5m, 2 = yolov5l  (configs/configs.yaml: model_name) */
    int num_classes;       /* detector classes == tracker fan-out (modules/__init__.py:33) */
    int img_size;          /* AutoShape `size`, 640 in the reference (quirk Q8) */
    int max_batch;         /* frames per detect launch */
    int max_frame_h, max_frame_w; /* largest source frame */
    float conf_thres;      /* configs/configs.yaml: min_conf 0.25 */
    float iou_thres;       /* configs/configs.yaml: min_iou 0.45 */
    int max_det;           /* configs/configs.yaml: max_det 300 */
    int max_candidates;    /* per-frame cap on boxes entering NMS (upstream max_nms = 30000) */
    int max_crops;         /* ReID crops per launch */
    int max_tracks;        /* live tracks over all trackers */
    int nn_budget_cap;     /* largest NN_BUDGET any tracker will ask for */
    int with_detector;     /* 0: skip building the YOLO plan (track-only users) */
    int with_reid;         /* 0: skip building the ReID plan */
    int max_trackers;      /* (camera, class) trackers the engine can hold (device-resident headers + track lists) */
    int tracks_per_tracker;/* live tracks one tracker may hold (<= 512: the device step keeps its work arrays in LDS) */
} vc_engine_config;

int vc_engine_config_default(vc_engine_config* cfg);
int vc_engine_create(const vc_engine_config* cfg, vc_engine** out);
int vc_engine_destroy(vc_engine* e);

/* Parameters: conv layers addressed by the checkpoint's own names ("model.0.conv", "layer2.0.conv1", ...);
 * weights OIHW fp32 with BatchNorm already folded (vehicle-counting_amd/weights.py: fold_bn). */
#define VC_NET_YOLO 0
#define VC_NET_REID 1
int vc_engine_param_count(const vc_engine* e, int net, int* n);
int vc_engine_param_info(const vc_engine* e, int net, int index, char* name, int name_cap, int dims[4] /* O,I,kh,kw */);
int vc_engine_set_param(vc_engine* e, int net, const char* name, const float* w_oihw, const float* bias);
/* Detect anchors in pixels, 3 levels x 3 anchors x (w, h) = `model.24.anchors * stride` of an ultralytics checkpoint (custom
 * weights loaded by /root/reference/networks/yolo.py:58 may carry autoanchor values).  Default: the COCO set of yolov5{s,m,l}.yaml. */
int vc_engine_set_anchors(vc_engine* e, const float* anchors18);
int vc_engine_finalize(vc_engine* e); /* packs + uploads weights; detector/ReID calls are valid afterwards */
/* Kernel-selection switches of a live engine (defaults come from the environment at vc_engine_create: VC_C3_FUSED, VC_BNECK_FUSED,
 * VC_FRONT_FUSED, VC_CROP_PER_PIXEL, VC_DOT_ARENA_MB).  Names: "c3_fused", "bneck_fused", "bneck_cv3" (0 / 1), "front_fused" (0 off, 1 stream path,
 * 2 always), "crop_per_pixel", "sparse_head", "reid_block_fused", "fuse_upsample", "sppf_sep", "fuse_s2_pw", "head_side" (0 / 1; head_side: the P3 / P4 Detect-head ops on their own stream
 * beside the neck layers that follow them), "dot_arena_mb" (largest appearance-table arena the tracker may allocate; 0 = compute the
 * appearance rows inside the walk).  The parity tests use it to compare a fused kernel with the launches it replaces. */
int vc_engine_set_option(vc_engine* e, const char* name, int value);

/* ---- detect: ImageDetect.run ------------------------------------------------------------------ */
/* rgb[i]: H[i] x W[i] x 3 uint8 RGB.  out_det: n * max_det * 6 floats [x1,y1,x2,y2,conf,cls] in source pixels
 * (what AutoShape returns in results.xyxy); out_count[i] = detections of image i.  All images of one call
 * share the AutoShape inference shape (models/common.py v6.0). */
int vc_detect(vc_engine* e, const uint8_t* const* rgb, const int* h, const int* w, int n, float* out_det, int* out_count);
/* Diagnostics for tensor-level parity: the network tensor the detector saw, raw head, candidates. */
int vc_detect_debug_shape(const vc_engine* e, int* net_h, int* net_w, int* n_candidates);
int vc_detect_debug_layer(vc_engine* e, int layer /* 0..23, or -1 = preprocessed input */, float* out_nhwc, size_t cap_floats,
                          int dims[4] /* B,H,W,C */);
int vc_detect_debug_pred(vc_engine* e, float* out /* B * n_candidates * (5+nc) */, size_t cap_floats);

/* ---- embed: Extractor.__call__ over DeepSort._get_features crops -------------------------------- */
/* bgr: H x W x 3 uint8 (the `ori_img` the reference crops from).  boxes_cxcywh: k x 4 float64 centre boxes
 * (deep_sort.py:28 bbox_xywh).  out_feat: k x 512 float32, unit L2 norm. */
int vc_embed(vc_engine* e, const uint8_t* bgr, int h, int w, const double* boxes_cxcywh, int k, float* out_feat);
/* Diagnostics: the k x 50 x 50 x 3 network input the last vc_embed built (Extractor._preprocess: resize, /255, Normalize), as float32. */
int vc_embed_debug_input(vc_engine* e, int k, float* out_nhwc, size_t cap_floats, int dims[4] /* k,50,50,3 */);
int vc_embed_tensor(vc_engine* e, const float* x_nchw /* k x 3 x 50 x 50 */, int k, float* out_feat);

/* ---- tracker: sort/tracker.py + deep_sort.py ------------------------------------------------------ */
typedef struct vc_tracker_params {
    double max_dist;          /* MAX_DIST          (cosine matching threshold) */
    double min_confidence;    /* MIN_CONFIDENCE */
    double nms_max_overlap;   /* NMS_MAX_OVERLAP */
    double max_iou_distance;  /* MAX_IOU_DISTANCE */
    int max_age;              /* MAX_AGE */
    int n_init;               /* N_INIT */
    int nn_budget;            /* NN_BUDGET (<= engine nn_budget_cap) */
} vc_tracker_params;

/* tracker_id is an opaque handle (slot | generation << 16): every entry point that takes one refuses (VC_ERR_NOTFOUND / VC_ERR_ARG) a handle
 * whose tracker has been destroyed, also after its slot has been handed to a later vc_tracker_create. */
int vc_tracker_create(vc_engine* e, const vc_tracker_params* p, int* tracker_id);
int vc_tracker_reset(vc_engine* e, int tracker_id);
/* Gives the tracker's tracks and its slot back.  The reference drops its VideoTracker after every video (modules/__init__.py:32-36); the
 * drop-in's DeepSort calls this when it is closed. */
int vc_tracker_destroy(vc_engine* e, int tracker_id);
/* Tracker.predict() + Tracker.update(detections) for detections already filtered/NMS'ed by the caller.
 * tlwh: k x 4 f64, conf: k f64, feat: k x 512 f32 (host). */
int vc_tracker_step(vc_engine* e, int tracker_id, const double* tlwh, const double* conf, const float* feat, int k);
/* Snapshot of the live tracks in list order (sort/tracker.py: self.tracks). Any pointer may be NULL. */
int vc_tracker_count(vc_engine* e, int tracker_id, int* n);
int vc_tracker_state(vc_engine* e, int tracker_id, int cap, int64_t* ids, int* state, int* hits, int* age, int* tsu,
                     double* mean8, double* cov64, int* gallery_count);
/* Cost matrices of the last blocking step that stepped exactly ONE tracker (vc_tracker_step, vc_deepsort_update), as the tracker
 * kernel computed them: app[t*D + d] = gated min-cosine cost (sort/nn_matching.py:160-177 + linear_assignment.py:148-192) of list
 * position t, valid for the tracks that were confirmed when the step began; iou[t*D + d] = 1 - IoU (sort/iou_matching.py:40-81), valid
 * for the IoU candidates.  T = live tracks before the step, D = detections of the step.  Parity tests read the HOT kernel's numbers. */
int vc_tracker_debug_costs(vc_engine* e, int cap_entries, double* app, double* iou, int* T, int* D);
/* Tracker state for stream migration (the reference keeps it in Python objects, sort/tracker.py:40-60 and
 * sort/track.py:64-80, and cannot move a stream): parameters, id counter, per track the FSM counters, fp64 Kalman mean and
 * covariance and the valid gallery rows.  vc_tracker_snapshot with buf == NULL only reports the size.  vc_tracker_restore
 * replaces the state (and parameters) of an existing tracker of any engine whose nn_budget_cap >= the snapshot's nn_budget. */
int vc_tracker_snapshot(vc_engine* e, int tracker_id, void* buf, size_t cap, size_t* size);
int vc_tracker_restore(vc_engine* e, int tracker_id, const void* buf, size_t size);
/* DeepSort.update: boxes xyxy (k x 4 f64) + confidences on a BGR frame -> rows [x1,y1,x2,y2,track_id,-1,0]. */
int vc_deepsort_update(vc_engine* e, int tracker_id, const uint8_t* bgr, int h, int w, const double* bbox_xyxy,
                       const double* conf, int k, int64_t* out_rows7, int cap_rows, int* out_m);
/* VideoTracker.run: trackers[c] is the tracker id of class c.  boxes xywh (top-left), labels, scores as
 * ImageDetect.run returns them.  Output rows [x1,y1,x2,y2,track_id,label]. */
int vc_videotracker_run(vc_engine* e, const int* trackers, int num_classes, const uint8_t* bgr, int h, int w,
                        const double* boxes_xywh, const int64_t* labels, const double* scores, int n,
                        int64_t* out_rows6, int cap_rows, int* out_m);

/* ---- fused stream path: the per-frame body of CountingPipeline.run on device-resident frames -------- */
/* frames_dev: device pointer to B x H x W x 3 uint8 *BGR* frames (cv2.VideoCapture order; the RGB view
 * the detector needs is taken on the fly).  Results per frame: rows [x1,y1,x2,y2,track_id,label]. */
/* vc_stream_submit enqueues the detector for a batch and returns at once (at most two outstanding); vc_stream_run consumes
 * submissions in order (submitting itself when none is pending), so `submit(i+1); run(i)` overlaps detect(i+1) with track(i). */
int vc_stream_submit(vc_engine* e, const void* frames_dev, int b, int h, int w);
/* Frames in (pinned) HOST memory, as the reference's loader delivers them (modules/datasets.py:47-61).  vc_stream_stage_host copies the
 * batch to one of four device staging slots on the engine's copy stream and returns at once; *frames_dev_out is the device address to
 * hand to vc_stream_submit and then vc_stream_run / vc_stream_run_async for this batch (valid until its rows have been collected; at
 * most four host batches may be alive).  vc_stream_submit of a staged address enqueues the detector behind the copy.  Staging one
 * batch further ahead than submitting -- stage(i+2); submit(i+1); run(i); collect(i-1) -- puts the PCIe copy under the detector of the
 * batch before.  vc_stream_submit_host = stage + submit in one call (the copy then sits in front of its own detector). */
int vc_stream_stage_host(vc_engine* e, const uint8_t* frames_host, int b, int h, int w, void** frames_dev_out);
int vc_stream_submit_host(vc_engine* e, const uint8_t* frames_host, int b, int h, int w, void** frames_dev_out);
int vc_stream_run(vc_engine* e, const int* trackers, int num_classes, const void* frames_dev, int b, int h, int w,
                  int64_t* out_rows6, int cap_rows_per_frame, int* out_m /* b */, int* out_ndet /* b, may be NULL */);

/* Asynchronous form of vc_stream_run: the batch's tracker work is ONE kernel enqueued on the engine's tracker stream (no host
 * thread); the call returns as soon as the batch's ReID and tracker kernel are enqueued.  At most two batches may be outstanding.
 * vc_stream_collect returns the rows of the OLDEST outstanding batch (same layout as vc_stream_run) and blocks until they are
 * ready.  Calls that touch tracker state (vc_tracker_*, vc_deepsort_update, vc_videotracker_run) first wait for outstanding batches.
 * A submission that cannot be embedded (more than max_candidates boxes passed conf_thres, more boxes than max_crops, an empty crop)
 * is reported ONCE by the call that finds it and dropped; vc_stream_reset abandons everything in flight. */
int vc_stream_run_async(vc_engine* e, const int* trackers, int num_classes, const void* frames_dev, int b, int h, int w,
                        int cap_rows_per_frame);
int vc_stream_collect(vc_engine* e, int64_t* out_rows6, int cap_rows_per_frame, int* out_m, int* out_ndet, int b);
/* The same for a batch that interleaves S cameras (the reference builds a new VideoTracker per video, modules/__init__.py:29-36; one
 * engine then serves S videos): frame f belongs to camera cam_of_frame[f] in [0, n_cam) and is stepped on trackers[cam * num_classes +
 * label]; the frames of one camera must appear in stream order inside the batch and across batches.  Rows come back per frame exactly
 * as from vc_stream_run_async (vc_stream_collect).  Per-camera latency is b / n_cam frames, and the tracker kernel walks
 * n_cam x num_classes trackers in parallel, b / n_cam steps each. */
int vc_stream_run_async_multi(vc_engine* e, const int* trackers /* n_cam x num_classes */, int n_cam, int num_classes, const int* cam_of_frame /* b */,
                              const void* frames_dev, int b, int h, int w, int cap_rows_per_frame);
int vc_stream_reset(vc_engine* e);

/* ---- one stream on several GPUs: frame-sharded front end (ordering contract of modules/__init__.py:54-84) -------- */
/* Front half of the fused path for the OLDEST submission (vc_stream_submit): detections marshalled like networks/yolo.py:72-97 +
 * crops + ReID for every box (deep_sort.py:119-129).  out_rows7: n x [frame index in the batch, x1, y1, x2, y2, conf, label] float64 --
 * the boxes VideoTracker.run works on; frames without boxes contribute nothing (Q1).  *out_feat_dev: DEVICE address of the matching
 * n x 512 float32 embeddings, valid until the third following vc_stream_embed / vc_stream_run* call. */
int vc_stream_embed(vc_engine* e, const void* frames_dev, int b, int h, int w, double* out_rows7, int cap_rows, int* out_n, const float** out_feat_dev);
/* Variable-length all-gather of such rows + embeddings over RCCL / xGMI on the engine's stream (vc_comm_init first): rows of all ranks,
 * rank-major, counts per rank, and the DEVICE address of the gathered embeddings (valid until the next call). */
int vc_allgather_rows(vc_engine* e, const double* rows7, const float* feat_dev, int n, double* out_rows7, int cap_rows, int* out_counts /* world */,
                      const float** out_feat_dev);
/* VideoTracker.run (modules/track.py:30-70) for a run of frames whose detections and embeddings are supplied: rows7 sorted by frame key
 * (column 0, ascending integers), row i's embedding at feat_dev[i] (device).  One tracker kernel launch for the whole run.  Per distinct
 * key j, ascending: out_keys[j], out_m[j] rows [x1, y1, x2, y2, track_id, label] at out_rows6 + j * cap_rows_per_frame * 6. */
int vc_videotracker_run_features(vc_engine* e, const int* trackers, int num_classes, const double* rows7, const float* feat_dev, int n, int h, int w,
                                 int64_t* out_rows6, int cap_rows_per_frame, int* out_m, int64_t* out_keys, int cap_frames, int* out_n_frames);
/* Detection injection for throughput studies: replaces the detector's NMS output of the batches SUBMITTED from now on
 * (vc_stream_submit / vc_stream_submit_host capture it) with caller boxes after the conv stack has run. NULL clears. */
int vc_stream_inject(vc_engine* e, const float* det6 /* b x n x 6 */, const int* count, int b, int n);

/* ---- counting: VideoCounting.run + count_frame_directions, and the one collective of the multi-GPU design -------------------- */
/* modules/track.py:81-137 (zone filter: any box corner inside the polygon; per (label, track) first / last box; direction =
 * utilities/counting/utils.py:139-152 find_best_match_direction) and utilities/counting/utils.py:276-297 (count[direction][label] += 1
 * at each track's last frame).  dir_lines: n_dir x (x0, y0, x1, y1) in the order of the zone file's direction shapes. */
typedef struct vc_counter vc_counter;
int vc_counter_create(const double* polygon_xy, int n_points, const double* dir_lines, int n_dir, int num_classes, vc_counter** out);
int vc_counter_destroy(vc_counter* c);
int vc_counter_add(vc_counter* c, const int64_t* frames, const int64_t* track_ids, const int64_t* labels, const int64_t* boxes_xyxy, int n);
int vc_counter_tracks(const vc_counter* c, int* n);
int vc_counts(const vc_counter* c, int32_t* out /* n_dir x num_classes */);
/* utilities/counting/utils.py:154-198 save_tracking_to_csv as columns (colour excluded, Q10): one row per (track, frame) that passed
 * the zone filter, ordered by label, then by the track's first appearance, then by arrival.  direction = index of the direction line. */
int vc_counter_rows_count(const vc_counter* c, int64_t* n);
int vc_counter_rows(const vc_counter* c, int64_t cap, int64_t* track_id, int64_t* frame_id, int64_t* box4, int64_t* label, int32_t* direction,
                    double* fpoint2, double* lpoint2, int64_t* fframe, int64_t* lframe);
/* One all-gather of the per-camera count tensors over RCCL / xGMI on the engine's stream.  Rank 0 creates the 128-byte
 * id (vc_comm_unique_id) and hands it to the other ranks by any side channel (the Python shim broadcasts it with torch.distributed);
 * every rank then calls vc_comm_init.  out receives world x n values, rank-major. */
int vc_comm_unique_id(void* out128);
int vc_comm_init(vc_engine* e, int rank, int world, const void* id128);
int vc_comm_destroy(vc_engine* e);
int vc_allgather_counts(vc_engine* e, const int32_t* local, int n, int32_t* out);

/* ---- visualisation egress (off the hot path) --------------------------------------------------------- */
/* utilities/counting/utils.py:299-331 visualize_merged: draws the annotation overlay into b BGR u8 frames (h x w x 3) in device
 * memory.  prims12: n x 12 int32 [type, x0, y0, x1, y1, t, colour (B | G << 8 | R << 16), glyph bits lo, glyph bits hi, 0, 0, 0] with
 * type 0 line (thickness t), 1 disc (radius t), 2 rectangle outline (thickness t), 3 filled box, 4 glyph (5 x 7 bitmap, scale t);
 * frame f owns prims [frame_first[f], frame_first[f + 1]) and paints them in order.  The host side (overlay.py) builds the lists
 * the way the reference's draw_* helpers are called.  Pixel parity with OpenCV's rasteriser / fonts is not claimed. */
int vc_overlay(vc_engine* e, void* frames_dev, int b, int h, int w, const int32_t* prims12, const int32_t* frame_first);

/* Host half of vc_allgather_rows: RCCL gathers equal-sized blocks, so every rank contributes `max_rows` rows (its own counts[r] rows
 * followed by padding) and the receive buffer is rank-major [world][max_rows][row_bytes].  This compacts such a padded buffer into
 * the first sum(counts) rows of `out` in rank-major order -- for frame chunks dealt round-robin to the ranks that is frame order
 * (the ordering contract of /root/reference/modules/__init__.py:54-84).  Pure host function (no GPU): it is the logic a world of
 * 2 / 3 / 8 ranks exercises and a single-GPU box cannot, so the CPU tests drive it directly.  vc_allgather_rows calls it for the rows
 * and uses vc_gather_offsets for the device-side copies of the embeddings.  Returns the row count through *out_total. */
int vc_gather_compact_host(const void* padded, int world, int max_rows, size_t row_bytes, const int* counts, void* out, size_t out_cap_rows,
                           int64_t* out_total);
/* src_off[r] / dst_off[r]: first row of rank r's block in the padded buffer / in the compacted output (rows, not bytes). */
int vc_gather_offsets(int world, int max_rows, const int* counts, int64_t* src_off, int64_t* dst_off, int64_t* out_total);

/* Conv autotune choices of this engine as text ("<shape key> <tile config>\n" per line, the VC_TUNE_CACHE file format), and their
 * import into another engine (keys already present are overwritten; call before the first launch of those shapes).  Multi-GPU
 * launches tune on rank 0 and broadcast the text (parallel.share_tune_cache): N ranks then neither time the candidates N times nor
 * disagree on the member of a near-tie (bf16 results are identical across tile configurations of one kernel family only). */
int vc_tune_export(vc_engine* e, char* buf, size_t cap, size_t* size);   /* buf may be NULL: *size receives the bytes needed (incl. NUL) */
int vc_tune_import(vc_engine* e, const char* text);

/* ---- measurement ---------------------------------------------------------------------------------- */
#define VC_PROF_CONV 0       /* all implicit-GEMM conv launches */
#define VC_PROF_DETECT_AUX 1 /* letterbox, pools, upsample, decode, NMS */
#define VC_PROF_REID_AUX 2   /* crop/resize, pools, L2 norm */
#define VC_PROF_TRACK 3      /* Kalman, cost matrices */
#define VC_PROF_NCAT 4
int vc_profile_enable(vc_engine* e, int on);   /* 0 off; 1 blocking events around every launch (serialises the streams);
                                                  2 in-flight event pairs around conv launches only, resolved by vc_profile_read */   /* brackets every launch with hipEvents on its own stream; disables graphs */
int vc_profile_read(vc_engine* e, int category, double* total_ms, int64_t* launches, double* flops, double* bytes);
/* After vc_profile_read(VC_PROF_CONV) resolved an in-flight (mode 2) region: milliseconds during which at least one conv kernel was
 * running, and the window from the first conv start to the last conv stop. */
int vc_profile_conv_busy(vc_engine* e, double* union_ms, double* span_ms);
/* vc_profile_read reports the work the launches EXECUTE (the sparse Detect head: the gathered rows only).  This returns the same
 * category with the sparse head credited as the dense Detect.m[i] it replaces -- the reference's algorithmic work -- as a separate
 * figure (bench.py: roofline.algorithmic_dense_*).  Call vc_profile_read first. */
int vc_profile_read_dense(vc_engine* e, int category, double* flops_dense, double* bytes_dense);
int vc_profile_reset(vc_engine* e);
int vc_profile_ops(vc_engine* e, char* buf, size_t cap);   /* per-conv-launch lines "conv M= N= K= ... ms= tflops=" recorded while profiling */
int vc_engine_sync(vc_engine* e);

/* ---- single-function entry points (parity tests; each runs the device kernel on host arrays) ------- */
typedef struct vc_conv_desc {
    int b, h, w, cin, cout, kh, kw, stride, pad;
    int act;         /* 0 none, 1 SiLU, 2 ReLU */
    int res_mode;    /* 0 none, 1 add after act, 2 add before act */
    int precision;   /* VC_PREC_* */
} vc_conv_desc;
/* x NHWC f32, w OIHW f32, bias f32[cout], res NHWC f32 or NULL, y NHWC f32 (bf16 mode rounds x, w, res, y). */
int vc_conv2d_host(const vc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res, float* y);
int vc_kalman_initiate_host(const double* xyah, int n, double* mean8, double* cov64);
int vc_kalman_predict_host(double* mean8, double* cov64, int n);
int vc_kalman_update_host(double* mean8, double* cov64, const double* z4, int n);
int vc_kalman_gating_host(const double* mean8, const double* cov64, const double* z4, int n_meas, double* out);
int vc_iou_cost_host(const double* track_tlwh, int t, const double* det_tlwh, int d, double* out_iou);
int vc_cosine_cost_host(const float* gallery, const int* gal_count, int t, int s_cap, const float* feat, int d, double* out);
int vc_dsort_nms_host(const double* tlwh, const double* scores, int n, double max_overlap, int* keep, int* n_keep);
int vc_lap_host(const double* cost, int nr, int nc, int* row4col_rows, int* cols, int* n_assigned);
int vc_letterbox_host(const uint8_t* rgb, int h, int w, int net_h, int net_w, int precision, float* out_nhwc3);
/* VideoCounting.run zone filter (modules/track.py:102-104): inside[i] = any corner of boxes[i] in the polygon. Host only. */
int vc_zone_filter_host(const double* polygon_xy, int n_points, const int64_t* boxes_xyxy, int n, uint8_t* inside);
/* candidates (already conf-filtered, in the reference's candidate order): boxes xyxy, conf, class -> kept rows */
int vc_nms_host(const float* boxes4, const float* conf, const int* cls, int n, float iou, int max_det, int max_cand,
                float* out6, int* out_n);

#ifdef __cplusplus
}
#endif
#endif /* VCOUNT_HIP_H */
