// Detect head decode, candidate filter, class-offset greedy NMS and scale_coords -- rows A7-A9 of SURVEY.md.
//
// Restates (ultralytics/yolov5 v6.0, reached from /root/reference/networks/yolo.py:70):
//   models/yolo.py::Detect.forward (inference)      -> decode_kernel
//   utils/general.py::non_max_suppression           -> decode_kernel filter + rank_sort + nms_mask + nms_scan
//   torchvision.ops.nms (stable descending sort, greedy, IoU > thr suppressed, all in float32)
//   utils/general.py::scale_coords / clip_coords    -> nms_scan_kernel tail
// and the thresholds the reference sets at networks/yolo.py:62-66 (conf, iou, max_det, multi_label=False).
// The float32 operation order follows oracle/yolov5.py so that kept/suppressed decisions are bit-identical
// for identical logits; contraction is disabled for the same reason.
#include <algorithm>

#include "kernels.h"

#pragma clang fp contract(off)

namespace vc {

#define VC_MAX_WH 4096.0f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool BF16>
__global__ __launch_bounds__(256) void decode_kernel(const DecodeLevel l0, const DecodeLevel l1, const DecodeLevel l2, int B, int nc,
                                                     float conf_thres, int max_cand, DetectPostBuffers pb, float* pred_debug,
                                                     int n_total) {
    const int no = nc + 5;
    const long total = (long)B * n_total;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int b = (int)(g / n_total);
        const int i = (int)(g % n_total);
        const DecodeLevel& lv = i >= l2.base ? l2 : (i >= l1.base ? l1 : l0);
        const int li = i - lv.base;
        const int x = li % lv.nx;
        const int y = (li / lv.nx) % lv.ny;
        const int a = li / (lv.nx * lv.ny);
        const size_t qoff = (((size_t)b * lv.ny + y) * lv.nx + x) * lv.cs + a * no;
        auto ld = [&](int c) -> float {
            if constexpr (BF16) return __uint_as_float((uint32_t)((const uint16_t*)lv.logits)[qoff + c] << 16);
            else return ((const float*)lv.logits)[qoff + c];
        };
        const float obj = sigmoidf_(ld(4));
        if (pred_debug == nullptr && !(obj > conf_thres)) continue;
        const float sx = sigmoidf_(ld(0)), sy = sigmoidf_(ld(1)), sw = sigmoidf_(ld(2)), sh = sigmoidf_(ld(3));
        const float cx = (sx * 2.0f - 0.5f + (float)x) * lv.stride;
        const float cy = (sy * 2.0f - 0.5f + (float)y) * lv.stride;
        const float tw = sw * 2.0f, th = sh * 2.0f;
        const float w = tw * tw * lv.anchor_w[a];
        const float h = th * th * lv.anchor_h[a];
        float best = -1.0f;
        int bj = 0;
        float* dbg = pred_debug ? pred_debug + (size_t)g * no : nullptr;
        if (dbg) { dbg[0] = cx; dbg[1] = cy; dbg[2] = w; dbg[3] = h; dbg[4] = obj; }
        for (int c = 0; c < nc; ++c) {
            const float sc = sigmoidf_(ld(5 + c));
            if (dbg) dbg[5 + c] = sc;
            const float v = sc * obj;
            if (v > best) { best = v; bj = c; }
        }
        if (!(obj > conf_thres) || !(best > conf_thres)) continue;
        const int pos = atomicAdd(pb.cand_count + b, 1);
        if (pos >= max_cand) { pb.overflow[b] = 1; continue; }
        const size_t o = (size_t)b * max_cand + pos;
        const float hw = w / 2.0f, hh = h / 2.0f;
        pb.cand_box[o * 4 + 0] = cx - hw;
        pb.cand_box[o * 4 + 1] = cy - hh;
        pb.cand_box[o * 4 + 2] = cx + hw;
        pb.cand_box[o * 4 + 3] = cy + hh;
        pb.cand_conf[o] = best;
        pb.cand_cls[o] = bj;
        pb.cand_idx[o] = i;
    }
}

// ---- sparse Detect head (bf16 engines) ----------------------------------------------------------------------------------------------
// Detect.m[i] is a 1x1 conv with 3 x (5 + nc) = 255 outputs per pixel, and non_max_suppression drops every anchor whose objectness is
// not above conf_thres before it looks at anything else (utils/general.py: `xc = prediction[..., 4] > conf_thres`).  Dense, the three head
// convs write 0.55 GB of logits per 128 frames at 640 x 640 that the decode reads the three objectness values of and throws away.  Sparse:
// an 8-channel conv (the three objectness rows of Detect.m[i], engine.hip) over every pixel, this kernel picks the pixels where an
// anchor can pass -- the SAME float test as decode_kernel -- and gathers their feature vectors; the full 255-channel conv then runs on
// the gathered rows only (conv_igemm_kernel with a device-side row count) and decode_sparse_kernel decodes those.  Same dot products in
// the same order for every surviving anchor: the detections are identical to the dense path's.
__global__ __launch_bounds__(256) void head_compact_kernel(const uint16_t* __restrict__ obj, const uint16_t* __restrict__ x, int x_cs, int x_co, int C, int M,
                                                           int pix_per_frame, float conf_thres, int cap, int* __restrict__ count, int* __restrict__ list,
                                                           uint16_t* __restrict__ xc, int* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int cpr = C / 8;                                   // 16-byte chunks per feature row (16 / 32 / 64 / ...)
    for (int m0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64; m0 < M; m0 += nwaves * 64) {
        const int m = m0 + lane;
        bool pass = false;
        if (m < M) {
            const uint2 o = *(const uint2*)(obj + (size_t)m * 8);
            const float l0 = __uint_as_float(o.x << 16), l1 = __uint_as_float(o.x & 0xffff0000u), l2 = __uint_as_float(o.y << 16);
            pass = sigmoidf_(l0) > conf_thres || sigmoidf_(l1) > conf_thres || sigmoidf_(l2) > conf_thres;
        }
        const unsigned long long mask = __ballot(pass);
        if (mask == 0) continue;
        const int n = __popcll(mask);
        int base = 0;
        if (lane == 0) base = atomicAdd(count, n);
        base = __shfl(base, 0);
        const int slot = base + __popcll(mask & lt);
        if (pass) {
            if (slot < cap) list[slot] = m;
            else overflow[m / pix_per_frame] = 1;           // more gathered pixels than the level's buffer holds: reported like a candidate overflow
        }
        // the flagged pixels' feature rows, 64 / cpr rows per round (cpr lanes x 16 bytes each)
        const int rpr = cpr >= 64 ? 1 : 64 / cpr;
        for (int k0 = 0; k0 < n; k0 += rpr) {
            const int k = k0 + lane / cpr, c = lane % cpr;
            if (lane / cpr < rpr && k < n && base + k < cap) {
                unsigned long long rest = mask;              // lane index of the k-th set bit
                for (int q = 0; q < k; ++q) rest &= rest - 1;
                const int src = m0 + (__ffsll((long long)rest) - 1);
                for (int cc = c; cc < cpr; cc += 64) {
                    const uint4 v = *(const uint4*)(x + (size_t)src * x_cs + x_co + cc * 8);
                    *(uint4*)(xc + (size_t)(base + k) * C + cc * 8) = v;
                }
            }
        }
    }
}

// decode_kernel's arithmetic on the gathered logits: SIXTEEN lanes per (gathered pixel, anchor) -- the class scores are what costs (80 precise
// sigmoids per anchor, one thread per anchor walked them one after the other: 29 us per pass on the detector queue's tail); lane s takes classes
// s, s + 16, ... in ascending order, the group reduces (score, class) with "larger score, then smaller class" = the serial loop's strict `>`.
// Candidate slots: a workgroup takes a CONTIGUOUS run of the gathered anchors (16 per pass, up to 8 passes when there are more anchors than the
// grid has groups), parks its survivors in LDS and asks for their slots with ONE atomicAdd per frame it met -- the gather order follows the pixel
// index, so a run is one or two frames.  One returning atomic per candidate on a frame's counter is a chain of ~90 ns steps: with the ~8000
// candidates per frame of the dense workloads that chain WAS the kernel (735 us per 32 frames of m1024-bf16, 548 us at 1280 x 720).  Which slot a
// candidate gets does not matter: rank_sort_kernel orders a frame's candidates by (score, flattened index), a total order.
#define VC_DS_MAXP 8
__global__ __launch_bounds__(256) void decode_sparse_kernel(const DecodeLevel l0, const DecodeLevel l1, const DecodeLevel l2, const int* __restrict__ counts,
                                                            const int* __restrict__ list0, const int* __restrict__ list1, const int* __restrict__ list2,
                                                            int cap0, int cap1, int cap2, int nc, float conf_thres, int max_cand, DetectPostBuffers pb) {
    __shared__ float4 s_box[16 * VC_DS_MAXP];
    __shared__ float s_conf[16 * VC_DS_MAXP];
    __shared__ int s_cls[16 * VC_DS_MAXP], s_idx[16 * VC_DS_MAXP], s_b[16 * VC_DS_MAXP], s_base[16 * VC_DS_MAXP];
    __shared__ int s_n;
    const int no = nc + 5;
    const int n0 = min(counts[0], cap0), n1 = min(counts[1], cap1), n2 = min(counts[2], cap2);
    const long total = 3l * ((long)n0 + n1 + n2);
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long per_pass = 16l * gridDim.x;
    const int passes = (int)min((long)VC_DS_MAXP, max(1l, (total + per_pass - 1) / per_pass));
    const long run = 16l * passes;
    for (long g0 = (long)blockIdx.x * run; g0 < total; g0 += (long)gridDim.x * run) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        for (int k = 0; k < passes; ++k) {
            const long g = g0 + k * 16 + grp;
            if (g >= total) continue;
            long e = g / 3;
            const int a = (int)(g - e * 3);
            const int level = e < n0 ? 0 : (e < (long)n0 + n1 ? 1 : 2);
            const DecodeLevel& lv = level == 0 ? l0 : (level == 1 ? l1 : l2);
            const int s = (int)(level == 0 ? e : (level == 1 ? e - n0 : e - n0 - n1));
            const int m = (level == 0 ? list0 : (level == 1 ? list1 : list2))[s];
            const int ppf = lv.ny * lv.nx;
            const int b = m / ppf, rem = m - b * ppf, y = rem / lv.nx, x = rem - y * lv.nx;
            const size_t qoff = (size_t)s * lv.cs + a * no;
            auto ld = [&](int c) -> float { return __uint_as_float((uint32_t)((const uint16_t*)lv.logits)[qoff + c] << 16); };
            const float obj = sigmoidf_(ld(4));
            if (!(obj > conf_thres)) continue;                   // (the same for the sixteen lanes of a group)
            float best = -1.0f;
            int bj = 0;
            for (int c = sub; c < nc; c += 16) {
                const float v = sigmoidf_(ld(5 + c)) * obj;
                if (v > best) { best = v; bj = c; }
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {
                const float ob = __shfl_xor(best, d);
                const int oj = __shfl_xor(bj, d);
                if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
            }
            if (sub != 0 || !(best > conf_thres)) continue;
            const float sx = sigmoidf_(ld(0)), sy = sigmoidf_(ld(1)), sw = sigmoidf_(ld(2)), sh = sigmoidf_(ld(3));
            const float cx = (sx * 2.0f - 0.5f + (float)x) * lv.stride;
            const float cy = (sy * 2.0f - 0.5f + (float)y) * lv.stride;
            const float tw = sw * 2.0f, th = sh * 2.0f;
            const float w = tw * tw * lv.anchor_w[a];
            const float h = th * th * lv.anchor_h[a];
            const float hw = w / 2.0f, hh = h / 2.0f;
            const int i = atomicAdd(&s_n, 1);
            s_box[i] = make_float4(cx - hw, cy - hh, cx + hw, cy + hh);
            s_conf[i] = best; s_cls[i] = bj; s_b[i] = b;
            s_idx[i] = lv.base + (a * lv.ny + y) * lv.nx + x;      // the anchor's position in the reference's flattened prediction
        }
        __syncthreads();
        const int n = s_n, t = threadIdx.x;
        int first = 0, rank = 0;
        if (t < n) {
            const int b = s_b[t];
            int cnt = 0;
            first = -1;
            for (int j = 0; j < n; ++j)
                if (s_b[j] == b) { if (first < 0) first = j; rank += j < t ? 1 : 0; ++cnt; }
            if (first == t) s_base[t] = atomicAdd(pb.cand_count + b, cnt);
        }
        __syncthreads();
        if (t < n) {
            const int b = s_b[t], pos = s_base[first] + rank;
            if (pos >= max_cand) pb.overflow[b] = 1;
            else {
                const size_t o = (size_t)b * max_cand + pos;
                *(float4*)(pb.cand_box + o * 4) = s_box[t];
                pb.cand_conf[o] = s_conf[t];
                pb.cand_cls[o] = s_cls[t];
                pb.cand_idx[o] = s_idx[t];
            }
        }
        // (the next run's `s_n = 0` sits behind this run's last LDS read: the barrier at the top of the loop orders the candidate arrays)
        __syncthreads();
    }
}

// Order candidates like `scores.sort(stable=True, descending=True)` over the reference's candidate order:
// rank = #{j : conf_j > conf_i or (conf_j == conf_i and idx_j < idx_i)}.
__global__ __launch_bounds__(256) void rank_sort_kernel(int max_cand, DetectPostBuffers pb) {
    __shared__ float s_conf[1024];
    __shared__ int s_idx[1024];
    const int b = blockIdx.y;
    const int n = min(pb.cand_count[b], max_cand);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= n) return;
    const size_t base = (size_t)b * max_cand;
    const float ci = i < n ? pb.cand_conf[base + i] : 0.f;
    const int ii = i < n ? pb.cand_idx[base + i] : 0;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024 && j0 + t < n; t += blockDim.x) {
            s_conf[t] = pb.cand_conf[base + j0 + t];
            s_idx[t] = pb.cand_idx[base + j0 + t];
        }
        __syncthreads();
        const int m = min(1024, n - j0);
        for (int t = 0; t < m; ++t) {
            const float cj = s_conf[t];
            rank += (cj > ci || (cj == ci && s_idx[t] < ii)) ? 1 : 0;
        }
    }
    if (i < n) {
        const size_t o = base + rank;
        const float4 bx = *(const float4*)(pb.cand_box + (base + i) * 4);
        *(float4*)(pb.sort_box + o * 4) = bx;
        pb.sort_conf[o] = ci;
        pb.sort_cls[o] = pb.cand_cls[base + i];
    }
}

// mask[i][w] bit j: box (w*64+j) > i in sorted order and IoU(i, j) > thr, boxes offset by cls * 4096 (float32 add).
__global__ __launch_bounds__(64) void nms_mask_kernel(int max_cand, float iou_thres, DetectPostBuffers pb) {
    __shared__ float4 cb[64];
    const int b = blockIdx.z;
    const int n = min(pb.cand_count[b], max_cand);
    const size_t base = (size_t)b * max_cand;
    const int t = threadIdx.x;
    // the grid is a fixed 8 x 8 tiles per frame (the candidate count lives on the device): block-stride over the 64 x 64 tiles
    // of the upper triangle -- a (max_cand/64)^2 grid costs 60 us of empty workgroups at max_cand = 4096
    for (int rb = blockIdx.y; rb * 64 < n; rb += gridDim.y) {
        const int row0 = rb * 64;
        const int i = row0 + t;
        float ix1 = 0.f, iy1 = 0.f, ix2 = 0.f, iy2 = 0.f, iarea = 0.f;
        if (i < n) {
            const float4 v = *(const float4*)(pb.sort_box + (base + i) * 4);
            const float off = (float)pb.sort_cls[base + i] * VC_MAX_WH;
            ix1 = v.x + off; iy1 = v.y + off; ix2 = v.z + off; iy2 = v.w + off;
            iarea = (ix2 - ix1) * (iy2 - iy1);
        }
        for (int cbk = blockIdx.x; cbk * 64 < n; cbk += gridDim.x) {
            const int col0 = cbk * 64;
            if (col0 + 63 < row0) continue;                                 // tile strictly below the diagonal (block-uniform)
            __syncthreads();
            if (col0 + t < n) {
                const float4 v = *(const float4*)(pb.sort_box + (base + col0 + t) * 4);
                const float off = (float)pb.sort_cls[base + col0 + t] * VC_MAX_WH;
                cb[t] = make_float4(v.x + off, v.y + off, v.z + off, v.w + off);
            }
            __syncthreads();
            if (i >= n) continue;
            unsigned long long bits = 0;
            const int m = min(64, n - col0);
            for (int j = 0; j < m; ++j) {
                if (col0 + j <= i) continue;
                const float4 c = cb[j];
                const float xx1 = fmaxf(ix1, c.x), yy1 = fmaxf(iy1, c.y), xx2 = fminf(ix2, c.z), yy2 = fminf(iy2, c.w);
                const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
                const float inter = w * h;
                const float carea = (c.z - c.x) * (c.w - c.y);
                const float ovr = inter / (iarea + carea - inter);
                if (ovr > iou_thres) bits |= 1ull << j;
            }
            pb.mask[(base + i) * (size_t)(max_cand / 64) + cbk] = bits;
        }
    }
}

#define VC_NMS_LDS_ROWS 512      // frames with at most this many candidates scan their suppression matrix from LDS

// One workgroup per frame walks the sorted candidates, keeps the un-suppressed ones (at most max_det), and writes
// the surviving boxes mapped back to source pixels (scale_coords + clip_coords).  The walk is a chain of dependent
// row reads (one per kept box): the rows of a frame with <= VC_NMS_LDS_ROWS candidates are first copied to LDS with
// coalesced loads so the chain runs at LDS latency instead of L2/HBM latency.
__global__ __launch_bounds__(256) void nms_scan_kernel(int max_cand, int max_det, const float* __restrict__ geom, DetectPostBuffers pb) {
    extern __shared__ unsigned long long nms_lds[];     // [max_cand / 64] (unused since round 5), rows[VC_NMS_LDS_ROWS][VC_NMS_LDS_ROWS / 64], kept_idx[max_det]
    const int b = blockIdx.x, lane = threadIdx.x;
    const float gain = geom[b * 5 + 0], padw = geom[b * 5 + 1], padh = geom[b * 5 + 2], src_w = geom[b * 5 + 3], src_h = geom[b * 5 + 4];
    const int n = min(pb.cand_count[b], max_cand);
    const int words = max_cand / 64;
    const int nw = (n + 63) / 64;
    unsigned long long* rows = nms_lds + words;
    const size_t base = (size_t)b * max_cand;
    const bool in_lds = n <= VC_NMS_LDS_ROWS;
    if (in_lds)
        for (int e = lane; e < n * nw; e += blockDim.x) {
            const int r = e / nw, w = e - r * nw;
            // words left of the diagonal tile are never written by nms_mask_kernel; they are not read below either (w >= i >> 6)
            rows[e] = w >= (r >> 6) ? pb.mask[(base + r) * (size_t)words + w] : 0ull;
        }
    __syncthreads();
    // The greedy walk itself: ONE wave, the frame's `removed` bit vector in registers (lane l holds words l and l + 64: 8192 candidates), the
    // word that holds candidate i read with v_readlane, the next survivor found with a count-trailing-zeros -- no barrier and no global
    // access per kept box (the first form took two workgroup barriers and a dependent global read per kept box: ~1.5 us each, 48 us per pass
    // at ~20 boxes per frame, all of it on the tail of the detector queue).  Kept indices go to LDS; all threads write the boxes afterwards.
    int* kept_idx = (int*)(rows + (size_t)VC_NMS_LDS_ROWS * (VC_NMS_LDS_ROWS / 64));
    __shared__ int kept_n;
    if (threadIdx.x < 64) {
        unsigned long long rm0 = 0ull, rm1 = 0ull;
        int kept = 0;
        for (int w = 0; w < nw && kept < max_det; ++w) {
            const unsigned long long valid = (w == nw - 1 && (n & 63)) ? ((1ull << (n & 63)) - 1ull) : ~0ull;
            unsigned long long done = 0ull;             // candidates of this word already visited
            while (kept < max_det) {
                const unsigned long long src = w < 64 ? rm0 : rm1;
                const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)src, w & 63);
                const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(src >> 32), w & 63);
                const unsigned long long alive = ~(((unsigned long long)hi << 32) | lo) & valid & ~done;      // uniform
                if (!alive) break;
                const int bit = __builtin_ctzll(alive);
                const int i = w * 64 + bit;
                done |= bit == 63 ? ~0ull : ((2ull << bit) - 1ull);
                if (lane == 0) kept_idx[kept] = i;
                ++kept;
                // candidate i suppresses the later ones its mask row names (words left of the diagonal are never written: not read)
                const unsigned long long* mrow = in_lds ? rows + (size_t)i * nw : pb.mask + (base + i) * (size_t)words;
                if (lane >= w && lane < nw) rm0 |= mrow[lane];
                if (lane + 64 >= w && lane + 64 < nw) rm1 |= mrow[lane + 64];
            }
        }
        if (lane == 0) kept_n = kept;
    }
    __syncthreads();
    const int kept = kept_n;
    for (int k = threadIdx.x; k < kept; k += blockDim.x) {
        const int i = kept_idx[k];
        const float4 v = *(const float4*)(pb.sort_box + (base + i) * 4);
        float x1 = (v.x - padw) / gain, y1 = (v.y - padh) / gain, x2 = (v.z - padw) / gain, y2 = (v.w - padh) / gain;
        x1 = fminf(fmaxf(x1, 0.f), src_w); x2 = fminf(fmaxf(x2, 0.f), src_w);
        y1 = fminf(fmaxf(y1, 0.f), src_h); y2 = fminf(fmaxf(y2, 0.f), src_h);
        float* o = pb.det + ((size_t)b * max_det + k) * 6;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = pb.sort_conf[base + i]; o[5] = (float)pb.sort_cls[base + i];
    }
    // more candidates passed the confidence test than max_candidates holds: which ones reached NMS depended on atomicAdd arrival
    // order, so the frame's result is not the reference's -- reported as a negative count (the host turns it into VC_ERR_CAPACITY)
    if (lane == 0) pb.det_count[b] = pb.overflow[b] ? -1 : kept;
}

int launch_decode(const DecodeLevel* lv, int nlv, int B, int nc, float conf, int max_cand, DetectPostBuffers& pb, float* pred_debug,
                  int n_total, hipStream_t s) {
    VC_CHECK(nlv == 3, VC_ERR_ARG, "decode: expects the 3 detection levels of YOLOv5");
    VC_HIP(hipMemsetAsync(pb.cand_count, 0, sizeof(int) * B, s));
    VC_HIP(hipMemsetAsync(pb.overflow, 0, sizeof(int) * B, s));
    const long total = (long)B * n_total;
    long grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    VC_CHECK(lv[0].bf16 == lv[1].bf16 && lv[1].bf16 == lv[2].bf16, VC_ERR_ARG, "decode: mixed logits types");
    if (lv[0].bf16) hipLaunchKernelGGL(decode_kernel<true>, dim3((int)grid), dim3(256), 0, s, lv[0], lv[1], lv[2], B, nc, conf, max_cand, pb, pred_debug, n_total);
    else hipLaunchKernelGGL(decode_kernel<false>, dim3((int)grid), dim3(256), 0, s, lv[0], lv[1], lv[2], B, nc, conf, max_cand, pb, pred_debug, n_total);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

int launch_head_compact(const void* obj, const View& x, int M, int pix_per_frame, float conf, int cap, int* count, int* list, void* xc, int* overflow, hipStream_t s) {
    VC_CHECK(x.C % 8 == 0 && x.cs % 8 == 0 && x.co % 8 == 0, VC_ERR_ARG, "head compaction: channel counts must be multiples of 8");
    const int waves = (M + 63) / 64;
    const int grid = std::min(std::max((waves + 3) / 4, 1), 256 * 8);
    hipLaunchKernelGGL(head_compact_kernel, dim3(grid), dim3(256), 0, s, (const uint16_t*)obj, (const uint16_t*)x.ptr, x.cs, x.co, x.C, M, pix_per_frame, conf, cap,
                       count, list, (uint16_t*)xc, overflow);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

// lv[i].logits = the gathered logits of level i ([cap_i][cs]); counts: device int[3]
int launch_decode_sparse(const DecodeLevel* lv, const int* counts, int* const* lists, const int* caps, int nc, float conf, int max_cand, DetectPostBuffers& pb,
                         hipStream_t s) {
    hipLaunchKernelGGL(decode_sparse_kernel, dim3(256 * 4), dim3(256), 0, s, lv[0], lv[1], lv[2], counts, lists[0], lists[1], lists[2], caps[0], caps[1], caps[2], nc,
                       conf, max_cand, pb);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

void scale_geom_host(const ScaleGeom& g, float out5[5]) {
    // scale_coords scalars exactly as python computes them (double), rounded to f32 when applied to the f32 tensor
    const double gain = std::min((double)g.net_h / g.src_h, (double)g.net_w / g.src_w);
    const double padw = (g.net_w - g.src_w * gain) / 2, padh = (g.net_h - g.src_h * gain) / 2;
    out5[0] = (float)gain; out5[1] = (float)padw; out5[2] = (float)padh; out5[3] = (float)g.src_w; out5[4] = (float)g.src_h;
}

int launch_nms(int B, int max_cand, int max_det, float iou, const float* geom_dev, DetectPostBuffers& pb, hipStream_t s) {
    VC_CHECK(max_cand % 64 == 0 && max_cand <= 8192, VC_ERR_ARG, "nms: max_candidates must be a multiple of 64 and <= 8192");
    hipLaunchKernelGGL(rank_sort_kernel, dim3(max_cand / 256, B), dim3(256), 0, s, max_cand, pb);
    const int tiles = std::min(max_cand / 64, 8);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(tiles, tiles, B), dim3(64), 0, s, max_cand, iou, pb);
    // kept_idx holds one entry per kept box: never more than min(max_det, max_cand) (upstream's max_nms is 30 000)
    const size_t lds = sizeof(unsigned long long) * (max_cand / 64 + VC_NMS_LDS_ROWS * (VC_NMS_LDS_ROWS / 64)) + sizeof(int) * std::min(max_det, max_cand);
    if (lds > 48 * 1024) {
        static size_t raised = 0;
        if (lds > raised) { VC_HIP(hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); raised = lds; }
    }
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(256), lds, s, max_cand, max_det, geom_dev, pb);
    VC_HIP(hipGetLastError());
    return VC_OK;
}

}  // namespace vc
