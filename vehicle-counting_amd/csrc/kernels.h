// Launch wrappers for the HBM-bound kernels around the convolutions (aux_kernels.hip, detect_post.hip,
// track_kernels.hip).  All take the stream they run on; none allocates or synchronises.
#pragma once
#include "vc_common.h"

namespace vc {

// A channel-sliced NHWC view: element (b,y,x,c) lives at ptr[((b*H+y)*W+x)*cs + co + c].
struct View {
    void* ptr;
    int B, H, W, C;
    int cs, co;
    int es;               // element size in bytes (0: the engine's default), set where the buffer is allocated
};

// ---- detect side -----------------------------------------------------------------------------------
struct LetterboxGeom {
    int src_h, src_w;     // source frame
    int net_h, net_w;     // network tensor
    int unpad_h, unpad_w; // resized image size inside the tensor
    int top, left;        // padding offsets
    int swap_rb;          // 1: source is BGR, network wants RGB
};
// src: B x src_h x src_w x 3 u8.  dst: B x net_h x net_w x 4 (prec), channel 3 = 0, values /255.
int launch_letterbox(const uint8_t* src, void* dst, int B, const LetterboxGeom& g, int prec, hipStream_t s);
// SPPF: x = view slice [0,C); writes maxpool5, maxpool5∘2, maxpool5∘3 into slices [C,2C), [2C,3C), [3C,4C) of the same buffer.
int launch_sppf_pool(const View& cat, int C, int prec, int form, hipStream_t s);   // form 1: register form where it applies, 0: the LDS-plane forms
int launch_upsample2x(const View& src, const View& dst, int prec, hipStream_t s);
// bf16 NHWC -> OCP e4m3fn, value x inv_scale, clamped to +-448 (the stem's output entering the fp8 layers)
int launch_bf16_to_fp8(const View& src, const View& dst, float inv_scale, hipStream_t s);

// Detect decode + candidate filter.  logits: B x ny x nx x lc_stride (channel a*(5+nc)+o), f32 in the fp32 parity mode, bf16 in
// the bf16 mode (like every other activation there).
struct DecodeLevel {
    const void* logits;
    int bf16;             // element type of `logits`
    int ny, nx, cs;       // cs = channel stride of the logits buffer
    float stride;
    float anchor_w[3], anchor_h[3];
    int base;             // index of this level's first candidate in the (level, anchor, y, x) order
};
struct DetectPostBuffers {
    // per frame f: cand_*[f*max_cand + i]
    float* cand_box;      // 4 floats xyxy (network pixels)
    float* cand_conf;
    int* cand_cls;
    int* cand_idx;        // position in the reference's flattened prediction (tie-break key)
    int* cand_count;      // [B]
    float* sort_box;      // same arrays after the (conf desc, idx asc) sort
    float* sort_conf;
    int* sort_cls;
    unsigned long long* mask;   // [B][max_cand][max_cand/64] suppression bit matrix
    float* det;           // [B][max_det][6] xyxy(source pixels) conf cls
    int* det_count;       // [B]
    int* overflow;        // [B] set when more than max_cand candidates passed the threshold
};
int launch_decode(const DecodeLevel* lv, int nlv, int B, int nc, float conf, int max_cand, DetectPostBuffers& pb,
                  float* pred_debug /* nullable: B x n_total x (5+nc) */, int n_total, hipStream_t s);
// sparse Detect head (detect_post.hip): pixels whose objectness can pass conf_thres -> list + gathered feature rows; decode of the gathered logits
int launch_head_compact(const void* obj /* [M][8] bf16 */, const View& x, int M, int pix_per_frame, float conf, int cap, int* count, int* list, void* xc,
                        int* overflow, hipStream_t s);
int launch_decode_sparse(const DecodeLevel* lv, const int* counts, int* const* lists, const int* caps, int nc, float conf, int max_cand, DetectPostBuffers& pb,
                         hipStream_t s);
struct ScaleGeom { int net_h, net_w, src_h, src_w; };
void scale_geom_host(const ScaleGeom& g, float out5[5]);   // gain, padw, padh, src_w, src_h as float32
// geom_dev: device [B][5] from scale_geom_host
int launch_nms(int B, int max_cand, int max_det, float iou, const float* geom_dev, DetectPostBuffers& pb, hipStream_t s);

// ---- ReID side -------------------------------------------------------------------------------------
// frames: F x H x W x 3 u8 BGR.  crop[i] = (frame, x1, y1, x2, y2) end-exclusive int corners.
// dst: k x 50 x 50 x cpad (prec), channels >= 3 zero.  ((v/255 bilinear) - mean) / std, BGR order kept (quirk Q3).
int launch_crop_resize(const uint8_t* frames, int H, int W, const int* crops5, int k, void* dst, int cpad, int prec,
                       hipStream_t s, bool per_pixel = false);     // per_pixel: the per-pixel instance of the bf16 path (parity tests)
// x_nchw: k x 3 x 50 x 50 f32 already-normalised tensor -> same NHWC/cpad layout (vc_embed_tensor)
int launch_nchw_to_nhwc_pad(const float* x, int k, int C, int H, int W, void* dst, int cpad, int prec, hipStream_t s);
int launch_maxpool3s2(const View& src, const View& dst, int prec, hipStream_t s);
// src: k x 4 x 4 x 512 -> out k x 512 f32, AvgPool2d((4,4),1) then x / ||x||_2
int launch_avgpool_l2norm(const View& src, float* out, int prec, hipStream_t s);

// ---- track side (track_kernels.hip) ---------------------------------------------------------------------
namespace tc { struct TrackerHdr; struct TrackRecD; }   // track_core.h (included where the definitions are needed: it must be
                                                        // compiled with floating-point contraction off, unlike this header's users)
struct TrackPool {
    double* mean;     // [max_tracks][8]
    double* cov;      // [max_tracks][64]
    float* gallery;   // [max_tracks][budget_cap][512]
    int max_tracks, budget_cap;
};
// one confirmed track against a contiguous range of detections (appearance_row_dev)
struct CostJob {
    int slot;        // track slot
    int gal_count;   // valid gallery rows of that slot
    int det_off;     // first detection (index into the batch's detection arrays)
    int det_n;       // detections
    int out_off;     // offset into the output cost array
    int tsu;         // time since update
};
// One tracker step = (tracker, frame): Tracker.predict() + Tracker.update(detections) on the prepared (confidence-filtered,
// DeepSORT-NMS'ed, deep_sort.py:31-41) detections [det_off, det_off + det_n) of the batch's detection arrays.
struct TrackTask { int tracker, det_off, det_n, frame, label, pad0, pad1, pad2; };
// workgroup b of the batch kernel owns one tracker and runs its tasks [task_begin, task_end) in order; the tracker's detections of
// the batch are the contiguous range [det_begin, det_begin + det_n) of the detection arrays (frames ascending)
struct TrackWgPlan { int tracker, task_begin, task_end, det_begin, det_n, pad0, pad1, pad2; };
// Appearance dot products hoisted out of the sequential tracker loop (track_kernels.hip, "dots"): every gallery sample a tracker
// can hold during the batch is either a sample it held when the batch began or the normalised feature of one of the batch's own
// detections, so <sample, detection feature> for ALL pairs is state-independent and is computed up front by a grid-wide kernel.
// table[row * n_dets + d]: rows [0, n_old_rows) = the old samples (track order, ring position), rows n_old_rows + e = detection e.
#define VC_TRACK_DYN_LDS_BYTES (96 * 1024)   // dynamic LDS the tracker kernels may ask for: step work arrays (62.5 KB at 512 tracks + detections) + the assignment matrix
#define VC_ROW_CHUNK 128          // rows of the output arena a tracker workgroup reserves at a time
struct TrackDotPlan { long long table_off; int n_old_rows, n_dets, row_src_off, tile_begin, tiles, det_tiles, use_table, pad; };
struct TrackBatchArgs {
    TrackPool pool;
    tc::TrackerHdr* hdrs;            // [max_trackers]
    int* lists;                      // [max_trackers][list_cap]: pool slots of the live tracks in list order (tracker.py: self.tracks)
    int list_cap;
    tc::TrackRecD* recs;             // [max_tracks] per-slot record
    int* free_top;                   // free slots: stack [0, *free_top), only popped inside a kernel
    int* free_stack;
    int* freed_count;                // slots freed by the running kernel (appended; merged into the stack after it)
    int* freed;
    const TrackWgPlan* plans;
    const TrackTask* tasks;
    const double* det_tlwh;          // [n_det][4]
    const double* det_xyah;          // [n_det][4]
    const int* det_featrow;          // [n_det] row of `feat`
    const float* feat;               // [rows][512] embeddings of the batch
    TrackDotPlan* dot_plans;         // [n_wg], written by track_plan_kernel
    float* dot_arena; long long dot_arena_floats;
    int* row_src; int row_src_cap;   // per old table row: slot * budget_cap + ring position
    int* gal_row;                    // [max_tracks][budget_cap]: table row of every gallery ring entry (valid during a batch)
    float* nfeat;                    // [n_det][512] normalised detection features (what a gallery write stores)
    float* det_ss;                   // [n_det] |feature|^2
    int* dot_ctl;                    // [0..1] arena cursor (long long), [2] row_src cursor, [3] tile total, [4] tile cursor
    int n_det_total;
    long long* rows;                 // [rows_cap][6] output arena: x1,y1,x2,y2,id,label
    int rows_cap;
    int* row_cursor;
    int* task_row_off;               // [n_tasks] first row of the task in the arena
    int* task_row_n;                 // [n_tasks] rows of the task
    int* task_ntracks;               // [n_tasks] live tracks after the step
    int* task_T;                     // [n_tasks] live tracks before the step (rows of the step's cost matrices)
    int* status;                     // [0] first error (tc::TERR_*), [1] tracker, [2] task
    double* scratch;                 // per workgroup track_scratch_per_wg() bytes: appearance rows, IoU rows, gathered sub-matrix,
    size_t scratch_per_wg;           // transpose (4 x 65536 doubles) + the work arrays of steps too large for the LDS
    int cap;                         // tracks + detections per step whose work arrays live in LDS (a multiple of 8, <= 512)
    int frame_w, frame_h;
    int all_tables;                  // host-side bound: every tracker's appearance table fits the arena (lean kernel instance)
    int dbg_costs;                   // 1: keep the cost rows in the global scratch (vc_tracker_debug_costs reads them back)
    int lmat_doubles;                // dynamic LDS behind the step work arrays for the assignment's matrix (0: small_c / global scratch only)
    int no_reg;                      // diagnostics (VC_TRACK_NO_REG): steps of <= 64 x 64 take the LDS-list matching path as well
    long long* dbg;                  // diagnostics (VC_TRACK_DBG): per task 8 timestamps (100 MHz): start, predict, cost rows, match, apply, finish
};
size_t track_scratch_per_wg();
int launch_track_batch(const TrackBatchArgs& a, int n_wg, hipStream_t s);
// test entry points: the batch kernel's own device functions on caller-supplied state
int launch_kat_kalman(TrackPool& tp, int which /* 0 initiate, 1 predict, 2 update */, const double* z, int n, hipStream_t s);
int launch_kat_lap(const double* cost, int nr, int nc, double* tbuf, int* out_rows, int* out_cols, int* out_n, hipStream_t s);
//   out[out_off + i] = gate(slot, det i) > 9.4877 ? 1e5 : min_s (1 - <g_s/|g_s|, f_i/|f_i|>); feature of detection g is feat[det_feat_row[g]]
int launch_appearance_cost(const TrackPool& tp, const CostJob* jobs, int njobs, const float* feat /* [rows][512] */,
                           const int* det_feat_row /* [nd] */, const double* det_xyah /* [nd][4] */, double* out, hipStream_t s);
int launch_gating_values(const TrackPool& tp, int slot, const double* z, int n, double* out, hipStream_t s);
int launch_iou_boxes(const double* a, int t, const double* b, int d, double* out, hipStream_t s);
// gallery append: normalised copy of feat[src[i]] into gallery[slot[i]][pos[i]]
int launch_gallery_write(TrackPool& tp, const int* slot_pos_src /* n x 3 */, int n, const float* feat, hipStream_t s);

}  // namespace vc
