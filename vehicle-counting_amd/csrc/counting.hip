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
    struct Rec { int64_t fbox[4], lbox[4], lframe; int at_last; };
    std::vector<std::map<int64_t, Rec>> tracks;   // per label: track id -> record (counts do not depend on the order)
};

using namespace vc;

extern "C" {

int vc_counter_create(const double* polygon_xy, int n_points, const double* dir_lines, int n_dir, int num_classes, vc_counter** out) {
    VC_CHECK(polygon_xy && n_points >= 1 && dir_lines && n_dir >= 1 && num_classes >= 1 && out, VC_ERR_ARG, "bad argument");
    vc_counter* c = new vc_counter();
    c->polygon.assign(polygon_xy, polygon_xy + (size_t)n_points * 2);
    c->dirs.assign(dir_lines, dir_lines + (size_t)n_dir * 4);
    c->num_classes = num_classes;
    c->tracks.resize(num_classes);
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
        auto& m = c->tracks[labels[i]];
        auto it = m.find(track_ids[i]);
        if (it == m.end()) {
            vc_counter::Rec r{};
            std::copy(boxes_xyxy + (size_t)i * 4, boxes_xyxy + (size_t)i * 4 + 4, r.fbox);
            std::copy(r.fbox, r.fbox + 4, r.lbox);
            r.lframe = frames[i]; r.at_last = 1;
            m.emplace(track_ids[i], r);
        } else {
            vc_counter::Rec& r = it->second;
            std::copy(boxes_xyxy + (size_t)i * 4, boxes_xyxy + (size_t)i * 4 + 4, r.lbox);
            if (frames[i] == r.lframe) r.at_last += 1; else { r.lframe = frames[i]; r.at_last = 1; }
        }
    }
    return VC_OK;
}

int vc_counter_tracks(const vc_counter* c, int* n) {
    VC_CHECK(c && n, VC_ERR_ARG, "null argument");
    size_t t = 0;
    for (const auto& m : c->tracks) t += m.size();
    *n = (int)t;
    return VC_OK;
}

// counts[d * num_classes + label]: tracks (rows at the track's last frame) per best-matching direction and class
int vc_counts(const vc_counter* c, int32_t* out) {
    VC_CHECK(c && out, VC_ERR_ARG, "null argument");
    const int nd = (int)(c->dirs.size() / 4);
    std::fill(out, out + (size_t)nd * c->num_classes, 0);
    for (int label = 0; label < c->num_classes; ++label)
        for (const auto& kv : c->tracks[label]) {
            const vc_counter::Rec& r = kv.second;
            const double fx = (double)(r.fbox[2] + r.fbox[0]) / 2, fy = (double)(r.fbox[3] + r.fbox[1]) / 2;
            const double lx = (double)(r.lbox[2] + r.lbox[0]) / 2, ly = (double)(r.lbox[3] + r.lbox[1]) / 2;
            const double ax = lx - fx, ay = ly - fy;
            double best = 0;
            int bi = 0;
            for (int d = 0; d < nd; ++d) {
                const double* p = &c->dirs[(size_t)d * 4];
                const double bx = p[2] - p[0], by = p[3] - p[1];
                const double den = std::sqrt(ax * ax + ay * ay) * std::sqrt(bx * bx + by * by);
                const double score = (ax * bx + ay * by) / den;           // 0/0 -> NaN, x/0 -> +-inf, like the reference's numpy division
                if (score > best) { best = score; bi = d; }
            }
            out[(size_t)bi * c->num_classes + label] += r.at_last;
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

}  // extern "C"
