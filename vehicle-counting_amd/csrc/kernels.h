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
int launch_sppf_pool(const View& cat, int C, int prec, hipStream_t s);
int launch_upsample2x(const View& src, const View& dst, int prec, hipStream_t s);

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
struct ScaleGeom { int net_h, net_w, src_h, src_w; };
void scale_geom_host(const ScaleGeom& g, float out5[5]);   // gain, padw, padh, src_w, src_h as float32
// geom_dev: device [B][5] from scale_geom_host
int launch_nms(int B, int max_cand, int max_det, float iou, const float* geom_dev, DetectPostBuffers& pb, hipStream_t s);

// ---- ReID side -------------------------------------------------------------------------------------
// frames: F x H x W x 3 u8 BGR.  crop[i] = (frame, x1, y1, x2, y2) end-exclusive int corners.
// dst: k x 50 x 50 x cpad (prec), channels >= 3 zero.  ((v/255 bilinear) - mean) / std, BGR order kept (quirk Q3).
int launch_crop_resize(const uint8_t* frames, int H, int W, const int* crops5, int k, void* dst, int cpad, int prec,
                       hipStream_t s);
// x_nchw: k x 3 x 50 x 50 f32 already-normalised tensor -> same NHWC/cpad layout (vc_embed_tensor)
int launch_nchw_to_nhwc_pad(const float* x, int k, int C, int H, int W, void* dst, int cpad, int prec, hipStream_t s);
int launch_maxpool3s2(const View& src, const View& dst, int prec, hipStream_t s);
// src: k x 4 x 4 x 512 -> out k x 512 f32, AvgPool2d((4,4),1) then x / ||x||_2
int launch_avgpool_l2norm(const View& src, float* out, int prec, hipStream_t s);

// ---- track side (track_kernels.hip) ---------------------------------------------------------------------
struct TrackPool {
    double* mean;     // [max_tracks][8]
    double* cov;      // [max_tracks][64]
    float* gallery;   // [max_tracks][budget_cap][512]
    int max_tracks, budget_cap;
};
// slots: pool indices.  All kernels are batched over an index list living in device memory.
int launch_kalman_predict(TrackPool& tp, const int* slots, int n, hipStream_t s);
// initiate: meas xyah per new track
int launch_kalman_initiate(TrackPool& tp, const int* slots, const double* xyah, int n, hipStream_t s);
int launch_kalman_update(TrackPool& tp, const int* slots, const double* xyah, int n, hipStream_t s);
// gating distances: for pair list job j: track slot ts[j] against measurements zs[zoff[j] .. zoff[j]+zn[j]) -> out[ooff[j] + i]
struct CostJob {
    int slot;        // track slot
    int gal_count;   // valid gallery rows of that slot (0: skip appearance)
    int det_off;     // first detection (index into the frame's detection arrays)
    int det_n;       // detections
    int out_off;     // offset into the output cost array
    int tsu;         // time since update (IoU rows with tsu > 1 are filled with 1e5)
};
// fused per-frame tracker kernels (one launch per phase; descriptors and results may live in host-mapped pinned memory)
struct TrackJobA { int slot, gal_count, det_off, det_n, app_off, iou_off, tsu, pad; };          // app_off / iou_off < 0: no such row
struct TrackOpB { int slot, kind, gal_pos, feat_row, out_row, pad0, pad1, pad2; double z[4]; }; // kind 0 none, 1 update, 2 initiate
struct TrackChainRec { TrackOpB op; TrackJobA job; };   // per touched slot: op.kind < 0 = no operation, job.slot < 0 = no cost job (96 B)
int launch_track_step(const TrackPool& tp, const TrackChainRec* recs, int nchains, const float* feat_ops, const float* feat_jobs,
                      double* mean_out, const int* det_feat_row, const double* det_xyah, const double* det_tlwh, double* out,
                      unsigned* counter, unsigned* done_flag, unsigned seq, hipStream_t s);
// appearance cost (min cosine distance over the gallery) with Mahalanobis gating folded in:
//   out[out_off + i] = gate(slot, det i) > 9.4877 ? 1e5 : min_s (1 - <g_s/|g_s|, f_i/|f_i|>)
// feature of detection g (global index det_off + i) is feat[det_feat_row[g]]
int launch_appearance_cost(const TrackPool& tp, const CostJob* jobs, int njobs, const float* feat /* [rows][512] */,
                           const int* det_feat_row /* [nd] */, const double* det_xyah /* [nd][4] */, double* out, hipStream_t s);
// test twins: raw squared Mahalanobis distances of one track; IoU between two box lists (tlwh)
int launch_gating_values(const TrackPool& tp, int slot, const double* z, int n, double* out, hipStream_t s);
int launch_iou_boxes(const double* a, int t, const double* b, int d, double* out, hipStream_t s);
//   out[out_off + i] = tsu > 1 ? 1e5 : 1 - IoU(track tlwh, det tlwh)
int launch_iou_cost(const TrackPool& tp, const CostJob* jobs, int njobs, const double* det_tlwh, double* out, hipStream_t s);
// out[i][0:8] = mean[slots[i]]
int launch_gather_means(const TrackPool& tp, const int* slots, int n, double* out, hipStream_t s);
// gallery append: copy feat[src[i]] into gallery[slot[i]][pos[i]]
int launch_gallery_write(TrackPool& tp, const int* slot_pos_src /* n x 3 */, int n, const float* feat, hipStream_t s);

}  // namespace vc
