// Counting behind the C ABI (vc_counts / vc_allgather_counts): VideoCounting.run's zone filter, per-track first /
// last box, direction assignment and the end state of count_frame_directions, plus the one collective of the multi-GPU design --
// an all-gather of the per-camera count tensors int32[n_dir][n_cls] over RCCL/xGMI on the engine's stream.
//
// Reference: /root/reference/modules/track.py:81-137 (VideoCounting.run), utilities/counting/bb_polygon.py:14-124,
// utilities/counting/utils.py:139-152 (find_best_match_direction: strict '>' from 0, first key as fallback, Q11),
// utilities/counting/utils.py:276-297 (count_frame_directions: count[direction][label] += 1 where lframe == frame_id).
// Host arithmetic only (a few thousand rows per video); the same double-precision operations as vehicle-counting_amd/counting.py.
#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

#include <rccl/rccl.h>

#include "engine.h"

struct vc_counter {
    std::vector<double> polygon;             // x0, y0, x1, y1, ...
    std::vector<double> dirs;                // per direction x0, y0, x1, y1
    int num_classes = 0;
    // per label: the tracks in order of first appearance (the reference's dict insertion order = the CSV's row order), every
    // (frame, box) row of a track in arrival order
    struct Rec { int64_t id; std::vector<int64_t> frames, boxes; int at_last; };
    std::vector<std::vector<Rec>> tracks;
    std::vector<std::map<int64_t, size_t>> index;  // per label: track id -> position in tracks[label]
    size_t n_rows = 0;
};

// utilities/counting/utils.py:139-152 find_best_match_direction on the centres of a track's first and last box: index of the
// direction with the largest cosine, strict '>' from 0, the first direction as fallback (Q11)
static int best_direction(const vc_counter* c, const int64_t* fbox, const int64_t* lbox, double* fpoint, double* lpoint) {
    const double fx = (double)(fbox[2] + fbox[0]) / 2, fy = (double)(fbox[3] + fbox[1]) / 2;
    const double lx = (double)(lbox[2] + lbox[0]) / 2, ly = (double)(lbox[3] + lbox[1]) / 2;
    if (fpoint) { fpoint[0] = fx; fpoint[1] = fy; }
    if (lpoint) { lpoint[0] = lx; lpoint[1] = ly; }
    const double ax = lx - fx, ay = ly - fy;
    const int nd = (int)(c->dirs.size() / 4);
    double best = 0;
    int bi = 0;
    for (int d = 0; d < nd; ++d) {
        const double* p = &c->dirs[(size_t)d * 4];
        const double bx = p[2] - p[0], by = p[3] - p[1];
        const double den = std::sqrt(ax * ax + ay * ay) * std::sqrt(bx * bx + by * by);
        const double score = (ax * bx + ay * by) / den;           // 0/0 -> NaN, x/0 -> +-inf, like the reference's numpy division
        if (score > best) { best = score; bi = d; }
    }
    return bi;
}

using namespace vc;

extern "C" {

int vc_counter_create(const double* polygon_xy, int n_points, const double* dir_lines, int n_dir, int num_classes, vc_counter** out) {
    VC_CHECK(polygon_xy && n_points >= 1 && dir_lines && n_dir >= 1 && num_classes >= 1 && out, VC_ERR_ARG, "bad argument");
    vc_counter* c = new vc_counter();
    c->polygon.assign(polygon_xy, polygon_xy + (size_t)n_points * 2);
    c->dirs.assign(dir_lines, dir_lines + (size_t)n_dir * 4);
    c->num_classes = num_classes;
    c->tracks.resize(num_classes);
    c->index.resize(num_classes);
    *out = c;
    return VC_OK;
}

int vc_counter_destroy(vc_counter* c) {
    delete c;
    return VC_OK;
}

// VideoCounting.run's loop body for n rows (modules/track.py:100-116): rows whose box has no corner inside the zone are dropped
int vc_counter_add(vc_counter* c, const int64_t* frames, const int64_t* track_ids, const int64_t* labels, const int64_t* boxes_xyxy, int n) {
    VC_CHECK(c && (n == 0 || (frames && track_ids && labels && boxes_xyxy)), VC_ERR_ARG, "bad argument");
    std::vector<uint8_t> inside(std::max(n, 1));
    VC_TRY(vc_zone_filter_host(c->polygon.data(), (int)(c->polygon.size() / 2), boxes_xyxy, n, inside.data()));
    for (int i = 0; i < n; ++i) {
        if (!inside[i]) continue;
        VC_CHECK(labels[i] >= 0 && labels[i] < c->num_classes, VC_ERR_ARG, "row %d: label %lld outside [0, %d)", i, (long long)labels[i], c->num_classes);
        auto& idx = c->index[labels[i]];
        auto& v = c->tracks[labels[i]];
        auto it = idx.find(track_ids[i]);
        if (it == idx.end()) {
            idx.emplace(track_ids[i], v.size());
            v.emplace_back();
            vc_counter::Rec& r = v.back();
            r.id = track_ids[i]; r.at_last = 1;
            r.frames.push_back(frames[i]);
            r.boxes.insert(r.boxes.end(), boxes_xyxy + (size_t)i * 4, boxes_xyxy + (size_t)i * 4 + 4);
        } else {
            vc_counter::Rec& r = v[it->second];
            if (frames[i] == r.frames.back()) r.at_last += 1; else r.at_last = 1;
            r.frames.push_back(frames[i]);
            r.boxes.insert(r.boxes.end(), boxes_xyxy + (size_t)i * 4, boxes_xyxy + (size_t)i * 4 + 4);
        }
        c->n_rows += 1;
    }
    return VC_OK;
}

int vc_counter_tracks(const vc_counter* c, int* n) {
    VC_CHECK(c && n, VC_ERR_ARG, "null argument");
    size_t t = 0;
    for (const auto& v : c->tracks) t += v.size();
    *n = (int)t;
    return VC_OK;
}

// counts[d * num_classes + label]: tracks (rows at the track's last frame) per best-matching direction and class
int vc_counts(const vc_counter* c, int32_t* out) {
    VC_CHECK(c && out, VC_ERR_ARG, "null argument");
    const int nd = (int)(c->dirs.size() / 4);
    std::fill(out, out + (size_t)nd * c->num_classes, 0);
    for (int label = 0; label < c->num_classes; ++label)
        for (const vc_counter::Rec& r : c->tracks[label]) {
            const int bi = best_direction(c, r.boxes.data(), r.boxes.data() + r.boxes.size() - 4, nullptr, nullptr);
            out[(size_t)bi * c->num_classes + label] += r.at_last;
        }
    return VC_OK;
}

int vc_counter_rows_count(const vc_counter* c, int64_t* n) {
    VC_CHECK(c && n, VC_ERR_ARG, "null argument");
    *n = (int64_t)c->n_rows;
    return VC_OK;
}

// The table of save_tracking_to_csv (utilities/counting/utils.py:154-198) as columns: one row per (track, frame), ordered by label,
// then by the track's first appearance, then by arrival.  direction = index into the direction lines given at creation.
int vc_counter_rows(const vc_counter* c, int64_t cap, int64_t* track_id, int64_t* frame_id, int64_t* box4, int64_t* label, int32_t* direction,
                    double* fpoint2, double* lpoint2, int64_t* fframe, int64_t* lframe) {
    VC_CHECK(c && track_id && frame_id && box4 && label && direction && fpoint2 && lpoint2 && fframe && lframe, VC_ERR_ARG, "null argument");
    VC_CHECK(cap >= (int64_t)c->n_rows, VC_ERR_CAPACITY, "%lld rows, room for %lld", (long long)c->n_rows, (long long)cap);
    size_t k = 0;
    for (int lab = 0; lab < c->num_classes; ++lab)
        for (const vc_counter::Rec& r : c->tracks[lab]) {
            double fp[2], lp[2];
            const int d = best_direction(c, r.boxes.data(), r.boxes.data() + r.boxes.size() - 4, fp, lp);
            for (size_t i = 0; i < r.frames.size(); ++i, ++k) {
                track_id[k] = r.id; frame_id[k] = r.frames[i]; label[k] = lab; direction[k] = d;
                std::copy(r.boxes.begin() + i * 4, r.boxes.begin() + i * 4 + 4, box4 + k * 4);
                fpoint2[k * 2] = fp[0]; fpoint2[k * 2 + 1] = fp[1]; lpoint2[k * 2] = lp[0]; lpoint2[k * 2 + 1] = lp[1];
                fframe[k] = r.frames.front(); lframe[k] = r.frames.back();
            }
        }
    return VC_OK;
}

// ---- the count all-gather over RCCL ---------------------------------------------------------------------------------------
int vc_comm_unique_id(void* out128) {
    VC_CHECK(out128, VC_ERR_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    VC_CHECK(r == ncclSuccess, VC_ERR_HIP, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(out128, &id, 128);
    return VC_OK;
}

int vc_comm_init(vc_engine* e, int rank, int world, const void* id128) {
    VC_CHECK(e && id128 && world >= 1 && rank >= 0 && rank < world, VC_ERR_ARG, "bad argument");
    VC_CHECK(!e->comm, VC_ERR_STATE, "communicator already initialised");
    VC_HIP(hipSetDevice(e->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, world, id, rank);
    VC_CHECK(r == ncclSuccess, VC_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, ncclGetErrorString(r));
    e->comm = comm; e->comm_rank = rank; e->comm_world = world;
    return VC_OK;
}

int vc_comm_destroy(vc_engine* e) {
    VC_CHECK(e, VC_ERR_ARG, "null engine");
    if (e->comm) { ncclCommDestroy((ncclComm_t)e->comm); e->comm = nullptr; }
    return VC_OK;
}

// every rank contributes n int32 (its cameras' count tensors, rank-major in the result): out = world * n values
int vc_allgather_counts(vc_engine* e, const int32_t* local, int n, int32_t* out) {
    VC_CHECK(e && local && out && n >= 1, VC_ERR_ARG, "bad argument");
    VC_CHECK(e->comm, VC_ERR_STATE, "vc_comm_init first");
    VC_HIP(hipSetDevice(e->cfg.device));
    const size_t need = (size_t)n * (e->comm_world + 1) * sizeof(int32_t);
    if (need > e->comm_buf_bytes) {
        VC_TRY(dev_alloc(e, (void**)&e->d_comm_buf, need * 2));
        e->comm_buf_bytes = need * 2;
    }
    int32_t* d_in = (int32_t*)e->d_comm_buf;
    int32_t* d_out = d_in + n;
    VC_HIP(hipMemcpyAsync(d_in, local, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    const ncclResult_t r = ncclAllGather(d_in, d_out, (size_t)n, ncclInt32, (ncclComm_t)e->comm, e->stream);
    VC_CHECK(r == ncclSuccess, VC_ERR_HIP, "ncclAllGather: %s", ncclGetErrorString(r));
    VC_HIP(hipMemcpyAsync(out, d_out, (size_t)n * e->comm_world * 4, hipMemcpyDeviceToHost, e->stream));
    VC_HIP(hipStreamSynchronize(e->stream));
    return VC_OK;
}

// Variable-length all-gather of detection rows and their embeddings (the frame-sharded front end): every rank hands in
// n rows [7] float64 (host) with the device address of the matching [n][512] float32 embeddings and receives the rows of ALL ranks,
// rank-major (a rank's rows keep their order), their counts per rank, and the device address of the gathered embeddings in the same
// order.  Two collectives on the engine's stream: the counts, then rows + embeddings padded to the largest count (ncclAllGather wants
// equal contributions).  The gathered embeddings stay valid until the next vc_allgather_rows.
int vc_gather_offsets(int world, int max_rows, const int* counts, int64_t* src_off, int64_t* dst_off, int64_t* out_total) {
    VC_CHECK(world >= 1 && max_rows >= 0 && counts && src_off && dst_off && out_total, VC_ERR_ARG, "bad argument");
    int64_t off = 0;
    for (int r = 0; r < world; ++r) {
        VC_CHECK(counts[r] >= 0 && counts[r] <= max_rows, VC_ERR_ARG, "rank %d contributes %d rows, blocks hold %d", r, counts[r], max_rows);
        src_off[r] = (int64_t)r * max_rows;                  // RCCL's receive buffer: equal blocks, rank-major
        dst_off[r] = off;                                    // compacted: rank-major, no padding
        off += counts[r];
    }
    *out_total = off;
    return VC_OK;
}

int vc_gather_compact_host(const void* padded, int world, int max_rows, size_t row_bytes, const int* counts, void* out, size_t out_cap_rows,
                           int64_t* out_total) {
    VC_CHECK(world >= 1 && counts && out_total && row_bytes > 0, VC_ERR_ARG, "bad argument");
    std::vector<int64_t> src(world), dst(world);
    VC_TRY(vc_gather_offsets(world, max_rows, counts, src.data(), dst.data(), out_total));
    VC_CHECK((size_t)*out_total <= out_cap_rows, VC_ERR_CAPACITY, "gather: %lld rows, room for %zu", (long long)*out_total, out_cap_rows);
    VC_CHECK(*out_total == 0 || (padded && out), VC_ERR_ARG, "null buffer");
    for (int r = 0; r < world; ++r)
        if (counts[r] > 0)
            memmove((char*)out + (size_t)dst[r] * row_bytes, (const char*)padded + (size_t)src[r] * row_bytes, (size_t)counts[r] * row_bytes);
    return VC_OK;
}

int vc_allgather_rows(vc_engine* e, const double* rows7, const float* feat_dev, int n, double* out_rows7, int cap_rows, int* out_counts,
                      const float** out_feat_dev) {
    VC_CHECK(e && out_rows7 && out_counts && out_feat_dev && n >= 0 && (n == 0 || (rows7 && feat_dev)), VC_ERR_ARG, "bad argument");
    VC_CHECK(e->comm, VC_ERR_STATE, "vc_comm_init first");
    VC_HIP(hipSetDevice(e->cfg.device));
    const int world = e->comm_world;
    ncclComm_t comm = (ncclComm_t)e->comm;
    hipStream_t s = e->stream;
    auto reserve = [&](void** p, size_t* cap, size_t need) -> int {
        if (need > *cap) { VC_TRY(dev_alloc(e, p, need * 3 / 2 + 256)); *cap = need * 3 / 2 + 256; }
        return VC_OK;
    };
    VC_TRY(reserve(&e->d_comm_buf, &e->comm_buf_bytes, (size_t)(world + 1) * sizeof(int32_t)));
    int32_t* d_n = (int32_t*)e->d_comm_buf;
    VC_HIP(hipMemcpyAsync(d_n, &n, 4, hipMemcpyHostToDevice, s));
    ncclResult_t r = ncclAllGather(d_n, d_n + 1, 1, ncclInt32, comm, s);
    VC_CHECK(r == ncclSuccess, VC_ERR_HIP, "ncclAllGather(counts): %s", ncclGetErrorString(r));
    VC_HIP(hipMemcpyAsync(out_counts, d_n + 1, (size_t)world * 4, hipMemcpyDeviceToHost, s));
    VC_HIP(hipStreamSynchronize(s));
    int maxn = 0, total = 0;
    for (int k = 0; k < world; ++k) { maxn = std::max(maxn, out_counts[k]); total += out_counts[k]; }
    VC_CHECK(total <= cap_rows, VC_ERR_CAPACITY, "vc_allgather_rows: %d rows, room for %d", total, cap_rows);
    *out_feat_dev = nullptr;
    if (total == 0) return VC_OK;
    const size_t rb = (size_t)maxn * 7 * sizeof(double), fb = (size_t)maxn * VC_FEAT_DIM * sizeof(float);
    VC_TRY(reserve(&e->d_gather_send, &e->gather_send_bytes, rb + fb));
    VC_TRY(reserve(&e->d_gather_recv, &e->gather_recv_bytes, (rb + fb) * world));
    VC_TRY(reserve(&e->d_gather_feat, &e->gather_feat_bytes, (size_t)total * VC_FEAT_DIM * sizeof(float)));
    if ((size_t)world * rb > e->h_gather_bytes) { VC_TRY(host_alloc(e, (void**)&e->h_gather, (size_t)world * rb * 2)); e->h_gather_bytes = (size_t)world * rb * 2; }
    char* send = (char*)e->d_gather_send;
    char* recv = (char*)e->d_gather_recv;
    if (n > 0) {
        VC_HIP(hipMemcpyAsync(send, rows7, (size_t)n * 7 * sizeof(double), hipMemcpyHostToDevice, s));
        VC_HIP(hipMemcpyAsync(send + rb, feat_dev, (size_t)n * VC_FEAT_DIM * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    ncclGroupStart();
    r = ncclAllGather(send, recv, (size_t)maxn * 7, ncclDouble, comm, s);
    const ncclResult_t r2 = ncclAllGather(send + rb, recv + (size_t)world * rb, (size_t)maxn * VC_FEAT_DIM, ncclFloat, comm, s);
    ncclGroupEnd();
    VC_CHECK(r == ncclSuccess && r2 == ncclSuccess, VC_ERR_HIP, "ncclAllGather(rows): %s", ncclGetErrorString(r != ncclSuccess ? r : r2));
    VC_HIP(hipMemcpyAsync(e->h_gather, recv, (size_t)world * rb, hipMemcpyDeviceToHost, s));
    std::vector<int64_t> src_off(world), dst_off(world);
    int64_t total_rows = 0;
    VC_TRY(vc_gather_offsets(world, maxn, out_counts, src_off.data(), dst_off.data(), &total_rows));
    for (int k = 0; k < world; ++k)                         // compact the embeddings on the device, rank-major
        if (out_counts[k] > 0)
            VC_HIP(hipMemcpyAsync((char*)e->d_gather_feat + (size_t)dst_off[k] * VC_FEAT_DIM * sizeof(float),
                                  recv + (size_t)world * rb + (size_t)src_off[k] * VC_FEAT_DIM * sizeof(float),
                                  (size_t)out_counts[k] * VC_FEAT_DIM * sizeof(float), hipMemcpyDeviceToDevice, s));
    VC_HIP(hipStreamSynchronize(s));
    VC_TRY(vc_gather_compact_host(e->h_gather, world, maxn, 7 * sizeof(double), out_counts, out_rows7, (size_t)cap_rows, &total_rows));
    *out_feat_dev = (const float*)e->d_gather_feat;
    return VC_OK;
}

}  // extern "C"
