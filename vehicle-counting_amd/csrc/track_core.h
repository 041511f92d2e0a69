// Control logic of one DeepSORT tracker step, written once for the device (track_kernels.hip: executed by wavefront 0 of the
// tracker's workgroup, data in LDS) and for the host build the CPU tests compile (tests/native/track_core_host.cpp, one "lane").
//
// Reference (paths relative to /root/reference/networks/deepsort/sort/):
//   linear_assignment.py:13-77   min_cost_matching  (clamp > max to max + 1e-5, rectangular LSA, reject > max)
//   linear_assignment.py:80-145  matching_cascade   (levels = time_since_update - 1; gate already folded into the cost rows)
//   tracker.py:93-131            Tracker._match     (cascade on confirmed tracks, IoU stage on the rest)
//   tracker.py:58-91             Tracker.update     (update / mark_missed / _initiate_track, drop deleted, gallery bookkeeping)
//   track.py:112-153             Track.predict counters, update, mark_missed (the FSM)
//   scipy.optimize.linear_sum_assignment (third-party: Crouse's rectangular shortest augmenting path, restated in lsap_core with
//   its tie-breaking -- pinned by tests/test_cabi.py::test_lap_matches_scipy_including_ties on 1500 matrices)
//
// Execution model: every lane of the cooperating group runs the same control flow on the same (uniform) values -- scalars live in
// registers, lists in LDS -- and only the loops marked "parallel" split their index range over the lanes.  A store of a uniform
// value by all lanes is harmless; read-modify-write of shared words is done by lane 0.  wave_sync() orders one parallel section
// against the next.  On the host there is one lane and every helper degenerates to the plain serial loop.
#pragma once
#include <limits.h>

#include "track_math.h"

namespace vc {
namespace tc {

enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };
enum { TERR_NONE = 0, TERR_TRACK_CAP = 1, TERR_POOL = 2, TERR_ROWS = 3, TERR_LAP = 4, TERR_TABLE = 5 };
// most live tracks + detections one tracker step can hold; a step's cost matrices are [T][D] with T + D <= TC_HARD_CAP, i.e. at
// most TC_MAT = (TC_HARD_CAP / 2)^2 entries
enum { TC_HARD_CAP = 512, TC_MAT = 256 * 256 };

struct TrackRecD { long long id; int state, hits, age, tsu, gal_count, gal_head; };                       // per pool slot, 32 B
struct TrackerHdr { double max_dist, max_iou_distance; long long next_id; int max_age, n_init, nn_budget, n_tracks, err, pad; };   // 48 B

struct Lanes { int lane, n; };

VC_HD void wave_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

VC_HD int imin(int a, int b) { return a < b ? a : b; }
VC_HD int imax(int a, int b) { return a > b ? a : b; }

VC_HD unsigned long long wave_or(Lanes L, unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int o = 1; o < 64; o <<= 1) x |= __shfl_xor(x, o);
#endif
    (void)L;
    return x;
}

VC_HD int wave_max(Lanes L, int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int o = 1; o < 64; o <<= 1) x = imax(x, __shfl_xor(x, o));
#endif
    (void)L;
    return x;
}

// out positions of the indices i in [0, n) with flag(i), ascending: store(position, i) is called once per kept index; returns the
// count.  Device: ballot + popcount prefix per 64 indices (the group is one full wavefront).
template <class F, class S>
VC_HD int compact(Lanes L, int n, F flag, S store) {
#if defined(__HIP_DEVICE_COMPILE__)
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + L.lane;
        const bool f = i < n && flag(i);
        const unsigned long long m = __ballot(f);
        if (f) store(base + __popcll(m & ((1ull << L.lane) - 1ull)), i);
        base += __popcll(m);
    }
    return base;
#else
    (void)L;
    int c = 0;
    for (int i = 0; i < n; ++i)
        if (flag(i)) store(c++, i);
    return c;
#endif
}

// ---- rectangular linear sum assignment -----------------------------------------------------------------------------------
// The inner scan of SciPy's augmenting-path search picks, over the remaining columns in list order,
//     if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
// i.e. the minimum; among equal minima the LAST unassigned column of the list if there is one, else the FIRST column.  That is
// the maximum of a total order on (value, unassigned, list position), so the scan can be split over lanes and merged in any order.
struct Best { double v; int un; int it; };
VC_HD bool better(const Best& a, const Best& b) {
    if (a.v < b.v) return true;
    if (a.v > b.v) return false;
    if (a.un != b.un) return a.un > b.un;
    return a.un ? a.it > b.it : a.it < b.it;
}
VC_HD Best wave_best(Lanes L, Best x) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int o = 1; o < 64; o <<= 1) {
        Best y;
        y.v = __shfl_xor(x.v, o); y.un = __shfl_xor(x.un, o); y.it = __shfl_xor(x.it, o);
        if (better(y, x)) x = y;
    }
#endif
    (void)L;
    return x;
}

struct LapWork { double *u, *v, *spc; int *path, *row4col, *remaining, *col4row; unsigned char *SR, *SC; };

// nr <= nc.  cost row-major [nr][nc].  Result in w.col4row[0..nr).  Returns 0, or -1 for an infeasible matrix.
VC_HD int lsap_core(Lanes L, int nr, int nc, const double* cost, const LapWork& w) {
    const double INF = (double)INFINITY;
    for (int i = L.lane; i < nr; i += L.n) { w.u[i] = 0.0; w.col4row[i] = -1; }
    for (int j = L.lane; j < nc; j += L.n) { w.v[j] = 0.0; w.path[j] = -1; w.row4col[j] = -1; }
    wave_sync();
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = L.lane; it < nc; it += L.n) { w.remaining[it] = nc - it - 1; w.SC[it] = 0; w.spc[it] = INF; }
        for (int r = L.lane; r < nr; r += L.n) w.SR[r] = 0;
        wave_sync();
        while (sink == -1) {
            w.SR[i] = 1;
            const double ui = w.u[i];
            Best b = {INF, 0, INT_MAX};
            for (int it = L.lane; it < num_remaining; it += L.n) {               // parallel
                const int j = w.remaining[it];
                const double r = minVal + cost[(size_t)i * nc + j] - ui - w.v[j];
                double s = w.spc[j];
                if (r < s) { w.path[j] = i; w.spc[j] = r; s = r; }
                const Best c = {s, w.row4col[j] == -1 ? 1 : 0, it};
                if (better(c, b)) b = c;
            }
            b = wave_best(L, b);
            wave_sync();
            minVal = b.v;
            if (minVal == INF) return -1;
            const int index = b.it;
            const int j = w.remaining[index];
            const int rj = w.row4col[j];
            const int last = w.remaining[num_remaining - 1];
            --num_remaining;
            wave_sync();                                   // every lane has read remaining[index] and the tail entry
            if (rj == -1) sink = j; else i = rj;
            w.SC[j] = 1;
            w.remaining[index] = last;
            wave_sync();
        }
        if (L.lane == 0) w.u[cur] += minVal;
        for (int r = L.lane; r < nr; r += L.n)
            if (w.SR[r] && r != cur) w.u[r] += minVal - w.spc[w.col4row[r]];
        for (int j = L.lane; j < nc; j += L.n)
            if (w.SC[j]) w.v[j] -= minVal - w.spc[j];
        wave_sync();
        int j = sink;                                      // augment along the alternating path (uniform, serial)
        while (true) {
            const int r = w.path[j];
            const int prev = w.col4row[r];
            wave_sync();
            if (L.lane == 0) { w.row4col[j] = r; w.col4row[r] = j; }
            j = prev;
            if (r == cur) break;
        }
        wave_sync();
    }
    return 0;
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same solver for problems with at most 64 rows and 64 columns, state in registers: lane j owns column j (v, spc, path,
// row4col, its position in the `remaining` list, SC) and lane i owns row i (u, col4row, SR); scalars move with wave shuffles, the
// cost matrix comes from cost(i, lane) = entry (row i, column `lane`), evaluated by every lane (lanes >= nc: any finite or
// infinite value, ignored).  Same scan order, same tie-breaking, same dual updates as lsap_core above (remaining[it] = nc - it - 1
// initially, swap-remove on selection), hence the same assignment; what it saves are the LDS round trips and wave-level
// synchronisations between the dependent steps of the search.  The scan's total order (value, unassigned, list position) is
// carried as (value, key) with key = unassigned ? 64 + position : 63 - position, larger key wins among equal values; the
// reduction runs over the pow2(nc) lanes that can hold a column and lane 0 broadcasts.  (Loading a lane's column of a small matrix
// into registers up front -- eight reads behind one latency instead of one dependent read per scan -- was measured slower: the
// select chain per scan costs more than the LDS read it replaces; 14.0 vs 12.9 us of matching per step.)
// On return lane i < nr holds col4row (every row is assigned) and lane j < nc holds row4col (-1: unassigned).
struct BestK { double v; int key; };
__device__ __forceinline__ bool better_k(const BestK& a, const BestK& b) {
    if (a.v < b.v) return true;
    if (a.v > b.v) return false;
    return a.key > b.key;
}
// Values of a wave-uniform lane: v_readlane (a few cycles) instead of a trip through the LDS crossbar (ds_bpermute, ~100 cycles in a
// dependent chain -- and the search below is one long dependent chain).
__device__ __forceinline__ int lane_get(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ double lane_get(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// one reduction step with the partner chosen by a DPP control: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140 -- after the four of them every lane of a 16-lane row holds the row's best
template <int CTRL>
__device__ __forceinline__ void best_step_dpp(BestK& b) {
    BestK y;
    y.v = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(b.v), CTRL, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(b.v), CTRL, 0xf, 0xf, true));
    y.key = __builtin_amdgcn_mov_dpp(b.key, CTRL, 0xf, 0xf, true);
    if (better_k(y, b)) b = y;
}
// the steps across 16-lane rows: partner lane ^ 16 / lane ^ 32 through v_permlane16_swap / v_permlane32_swap (gfx950: VALU rate, no trip
// through the LDS crossbar -- three ds_bpermute per step were most of a scan's latency for problems wider than 16 columns).
// permlane16_swap(x, x) = ([x0 x0 x2 x2], [x1 x1 x3 x3]) by rows: an even row finds its partner in the second result, an odd row in the first.
__device__ __forceinline__ int partner16(int x, bool odd_row) {
    typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
    const u32x2s r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    return (int)(odd_row ? r.x : r.y);
}
__device__ __forceinline__ int partner32(int x, bool upper) {
    typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
    const u32x2s r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (int)(upper ? r.x : r.y);
}
// best over lanes [0, width), width a power of two; valid in lane 0
__device__ __forceinline__ void best_reduce(BestK& b, int width) {
    if (width > 1) best_step_dpp<0xB1>(b);
    if (width > 2) best_step_dpp<0x4E>(b);
    if (width > 4) best_step_dpp<0x141>(b);
    if (width > 8) best_step_dpp<0x140>(b);
    if (width > 16) {
        const bool odd = (__lane_id() >> 4) & 1;
        BestK y;
        y.v = __hiloint2double(partner16(__double2hiint(b.v), odd), partner16(__double2loint(b.v), odd));
        y.key = partner16(b.key, odd);
        if (better_k(y, b)) b = y;
    }
    if (width > 32) {
        const bool up = __lane_id() >= 32;
        BestK y;
        y.v = __hiloint2double(partner32(__double2hiint(b.v), up), partner32(__double2loint(b.v), up));
        y.key = partner32(b.key, up);
        if (better_k(y, b)) b = y;
    }
}
template <class CostFn>
__device__ __forceinline__ int lsap_reg(int lane, int nr, int nc, CostFn cost, int& col4row, int& row4col) {
    const double INF = (double)INFINITY;
    double u = 0.0, v = 0.0;
    int path = -1;
    col4row = -1; row4col = -1;
    int width = 1;
    while (width < nc) width <<= 1;
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0, spc = INF;
        int i = cur, num_remaining = nc, sink = -1;
        int pos = lane < nc ? nc - 1 - lane : -1;          // column `lane` sits at list position nc - 1 - lane
        bool SR = false, SC = false;
        while (sink == -1) {
            if (lane == i) SR = true;
            const double ui = lane_get(u, i);
            const double cij = cost(i, lane);
            BestK b = {INF, INT_MIN};
            if (pos >= 0) {
                const double r = minVal + cij - ui - v;
                if (r < spc) { path = i; spc = r; }
                b.v = spc; b.key = row4col == -1 ? 64 + pos : 63 - pos;
            }
            best_reduce(b, width);
            minVal = lane_get(b.v, 0);
            const int key = lane_get(b.key, 0);
            if (minVal == INF) return -1;
            const int it = key >= 64 ? key - 64 : 63 - key;
            const int jstar = __ffsll((unsigned long long)__ballot(pos == it)) - 1;
            const int rj = lane_get(row4col, jstar);
            if (pos == num_remaining - 1) pos = it;        // swap-remove: the list's last entry takes the freed position ...
            if (lane == jstar) { pos = -1; SC = true; }    // ... and the chosen column leaves the list
            --num_remaining;
            if (rj == -1) sink = jstar; else i = rj;
        }
        const double spc_of_my_col = __shfl(spc, col4row >= 0 ? col4row : 0);
        if (lane == cur) u += minVal;
        else if (SR && lane < nr) u += minVal - spc_of_my_col;
        if (SC) v -= minVal - spc;
        int j = sink;                                        // augment along the alternating path
        while (true) {
            const int r = lane_get(path, j);
            const int prev = lane_get(col4row, r);
            if (lane == j) row4col = r;
            if (lane == r) col4row = j;
            j = prev;
            if (r == cur) break;
        }
    }
    return 0;
}
// The same solver for up to 128 rows x 128 columns: lane l owns columns l and l + 64 and rows l and l + 64 (two copies of the
// per-column / per-row state), the scan merges a lane's two columns before the cross-lane reduction, list positions take seven bits
// (key = unassigned ? 128 + position : 127 - position).  Same scan order, tie-breaking and dual updates as lsap_core / lsap_reg.  A
// step of BASELINE.json configs[2] (about 90 tracks x 70 detections per class) is 67 x 87 after the transpose: on the LDS-list
// solver, with the matrix in global memory, that was 120-230 us of a 250-390 us step.
template <class CostFn>
__device__ __forceinline__ int lsap_reg2(int lane, int nr, int nc, CostFn cost, int (&col4row)[2], int (&row4col)[2]) {
    const double INF = (double)INFINITY;
    double u[2] = {0.0, 0.0}, v[2] = {0.0, 0.0};
    int path[2] = {-1, -1};
    col4row[0] = col4row[1] = -1; row4col[0] = row4col[1] = -1;
    auto sel = [](const auto (&a)[2], int h) { return h ? a[1] : a[0]; };
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0, spc[2] = {INF, INF};
        int i = cur, num_remaining = nc, sink = -1;
        int pos[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) pos[h] = lane + 64 * h < nc ? nc - 1 - (lane + 64 * h) : -1;
        bool SR[2] = {false, false}, SC[2] = {false, false};
        while (sink == -1) {
            const int ih = i >> 6, il = i & 63;
            if (lane == il) { if (ih) SR[1] = true; else SR[0] = true; }
            const double ui = lane_get(ih ? u[1] : u[0], il);
            BestK b = {INF, INT_MIN};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (pos[h] >= 0) {
                    const double r = minVal + cost(i, lane, h) - ui - v[h];
                    if (r < spc[h]) { path[h] = i; spc[h] = r; }
                    const BestK c = {spc[h], row4col[h] == -1 ? 128 + pos[h] : 127 - pos[h]};
                    if (better_k(c, b)) b = c;
                }
            }
            best_reduce(b, 64);
            minVal = lane_get(b.v, 0);
            const int key = lane_get(b.key, 0);
            if (minVal == INF) return -1;
            const int it = key >= 128 ? key - 128 : 127 - key;
            const unsigned long long m0 = __ballot(pos[0] == it), m1 = __ballot(pos[1] == it);
            const int jstar = m0 ? __ffsll(m0) - 1 : 64 + __ffsll(m1) - 1;
            const int jh = jstar >> 6, jl = jstar & 63;
            const int rj = lane_get(jh ? row4col[1] : row4col[0], jl);
#pragma unroll
            for (int h = 0; h < 2; ++h) if (pos[h] == num_remaining - 1) pos[h] = it;        // swap-remove: the list's last entry takes the freed position ...
            if (lane == jl) { if (jh) { pos[1] = -1; SC[1] = true; } else { pos[0] = -1; SC[0] = true; } }    // ... and the chosen column leaves the list
            --num_remaining;
            if (rj == -1) sink = jstar; else i = rj;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {                        // dual update of row lane + 64 g
            const int c = col4row[g];
            const double s0 = __shfl(spc[0], c >= 0 ? c & 63 : 0), s1 = __shfl(spc[1], c >= 0 ? c & 63 : 0);
            const double spc_of_my_col = (c >> 6) & 1 ? s1 : s0;
            const int row = lane + 64 * g;
            if (row == cur) u[g] += minVal;
            else if (SR[g] && row < nr) u[g] += minVal - spc_of_my_col;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) if (SC[h]) v[h] -= minVal - spc[h];
        int j = sink;                                        // augment along the alternating path
        while (true) {
            const int r = lane_get((j >> 6) ? path[1] : path[0], j & 63);
            const int prev = lane_get((r >> 6) ? col4row[1] : col4row[0], r & 63);
            if (lane == (j & 63)) { if (j >> 6) row4col[1] = r; else row4col[0] = r; }
            if (lane == (r & 63)) { if (r >> 6) col4row[1] = j; else col4row[0] = j; }
            j = prev;
            if (r == cur) break;
        }
    }
    (void)sel;
    return 0;
}
// cost row-major [nr][nc] in memory, nr <= nc <= 128; col4row_out[0..nr)
__device__ __forceinline__ int lsap_core_wave128(int lane, int nr, int nc, const double* cost, int* col4row_out) {
    int c4r[2], r4c[2];
    if (lsap_reg2(lane, nr, nc, [&](int i, int l, int h) { const int j = l + 64 * h; return cost[(size_t)i * nc + (j < nc ? j : 0)]; }, c4r, r4c) != 0) return -1;
    if (lane < nr) col4row_out[lane] = c4r[0];
    if (lane + 64 < nr) col4row_out[lane + 64] = c4r[1];
    return 0;
}
// cost row-major [nr][nc] in memory (LDS for the small problems this is used for); col4row_out[0..nr)
__device__ __forceinline__ int lsap_core_wave64(int lane, int nr, int nc, const double* cost, int* col4row_out) {
    int c4r, r4c;
    const int cl = lane < nc ? lane : 0;
    if (lsap_reg(lane, nr, nc, [&](int i, int) { return cost[(size_t)i * nc + cl]; }, c4r, r4c) != 0) return -1;
    if (lane < nr) col4row_out[lane] = c4r;
    return 0;
}
#endif

// per-step work arrays (LDS on the device), each with room for `cap` entries
struct StepWork {
    int cap;
    int *slot, *state, *tsu, *galc, *galh, *hits;                 // per live track (list position t)
    long long* id;
    int *confirmed, *unconfirmed, *left, *rows, *un_rows, *un_cols, *un_tracks, *match_t, *match_d, *ri, *ci, *newslot;
    unsigned char *row_used, *col_used, *matched;
    unsigned char* adm;             // per live track: 0 = no detection of the step is within the appearance threshold of this (confirmed) track
                                    // (written with the track's cost row; 1 = unknown / some detection is: always a safe value)
    LapWork lap;
    double *small_c, *small_t;      // gathered sub-matrix / its transpose when it has at most small_n entries (LDS on the device:
    int small_n;                    // the assignment's scans then never leave the CU); 0 = always use the caller's buffers
    double* lmat;                   // one larger LDS matrix (lmat_n entries) for the problems that outgrow small_c / small_t: it holds
    int lmat_n;                     // the matrix the solver scans -- the gathered one, or its transpose when rows > columns
};

// bytes of one StepWork with capacity cap (all arrays 8-byte aligned: cap is a multiple of 8)
VC_HD size_t step_work_bytes(int cap) { return (size_t)cap * (4 * (6 + 12 + 4) + 8 * 4 + 6); }

VC_HD void step_work_carve(StepWork& w, void* base, int cap) {
    char* p = (char*)base;
    w.cap = cap;
    w.small_c = nullptr; w.small_t = nullptr; w.small_n = 0; w.lmat = nullptr; w.lmat_n = 0;
    auto D = [&](double*& q) { q = (double*)p; p += (size_t)cap * 8; };
    auto I = [&](int*& q) { q = (int*)p; p += (size_t)cap * 4; };
    auto B = [&](unsigned char*& q) { q = (unsigned char*)p; p += (size_t)cap; };
    D(w.lap.u); D(w.lap.v); D(w.lap.spc);
    w.id = (long long*)p; p += (size_t)cap * 8;
    I(w.lap.path); I(w.lap.row4col); I(w.lap.remaining); I(w.lap.col4row);
    I(w.slot); I(w.state); I(w.tsu); I(w.galc); I(w.galh); I(w.hits);
    I(w.confirmed); I(w.unconfirmed); I(w.left); I(w.rows); I(w.un_rows); I(w.un_cols); I(w.un_tracks); I(w.match_t); I(w.match_d);
    I(w.ri); I(w.ci); I(w.newslot);
    B(w.lap.SR); B(w.lap.SC); B(w.row_used); B(w.col_used); B(w.matched); B(w.adm);
}

// The transposed solve (nr > nc): t holds the matrix as [nc][nr]; pairs come back sorted by ORIGINAL row like SciPy's.
VC_HD int lap_solve_tr(Lanes L, const StepWork& w, const double* t, int nr, int nc, int& err) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (nr <= 64) { if (lsap_core_wave64(L.lane, nc, nr, t, w.lap.col4row) != 0) { err = TERR_LAP; return 0; } wave_sync(); }
    else if (nr <= 128) { if (lsap_core_wave128(L.lane, nc, nr, t, w.lap.col4row) != 0) { err = TERR_LAP; return 0; } wave_sync(); }
    else
#endif
    if (lsap_core(L, nc, nr, t, w.lap) != 0) { err = TERR_LAP; return 0; }
    int* r2c = w.lap.path;                                   // free again after the solve: original row -> original column
    for (int i = L.lane; i < nr; i += L.n) r2c[i] = -1;
    wave_sync();
    for (int v = L.lane; v < nc; v += L.n) r2c[w.lap.col4row[v]] = v;
    wave_sync();
    const int np = compact(L, nr, [&](int i) { return r2c[i] >= 0; }, [&](int pos, int i) { w.ri[pos] = i; w.ci[pos] = r2c[i]; });
    wave_sync();
    return np;
}

// scipy.optimize.linear_sum_assignment on c [nr][nc]: pairs (ri[k], ci[k]) sorted by row; returns their number (min(nr, nc)).
VC_HD int lap_solve(Lanes L, const StepWork& w, const double* c, int nr, int nc, double* tbuf, int& err) {
    if (nr == 1 || nc == 1) {
        // One row (or one column): the augmenting-path search reduces to its first scan -- the minimum entry, and among equal
        // minima the smallest index (the scan walks the columns in descending order and keeps the LAST unassigned minimum; for one
        // column SciPy solves the transpose).  Most cascade levels of a light scene hold a single track.
        const int n = nr * nc;
        Best b = {(double)INFINITY, 0, INT_MAX};
        for (int e = L.lane; e < n; e += L.n) {
            const Best x = {c[e], 0, e};
            if (better(x, b)) b = x;
        }
        b = wave_best(L, b);
        if (b.it == INT_MAX) { err = TERR_LAP; return 0; }
        if (L.lane == 0) { w.ri[0] = nr == 1 ? 0 : b.it; w.ci[0] = nr == 1 ? b.it : 0; }
        wave_sync();
        return 1;
    }
    if (nc < nr) {                                           // SciPy transposes so that rows <= columns
        for (int e = L.lane; e < nr * nc; e += L.n) {       // parallel
            const int i = e / nc, j = e - i * nc;
            tbuf[(size_t)j * nr + i] = c[e];
        }
        wave_sync();
        return lap_solve_tr(L, w, tbuf, nr, nc, err);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (nc <= 64) { if (lsap_core_wave64(L.lane, nr, nc, c, w.lap.col4row) != 0) { err = TERR_LAP; return 0; } wave_sync(); }
    else if (nc <= 128) { if (lsap_core_wave128(L.lane, nr, nc, c, w.lap.col4row) != 0) { err = TERR_LAP; return 0; } wave_sync(); }
    else
#endif
    if (lsap_core(L, nr, nc, c, w.lap) != 0) { err = TERR_LAP; return 0; }
    for (int i = L.lane; i < nr; i += L.n) { w.ri[i] = i; w.ci[i] = w.lap.col4row[i]; }
    wave_sync();
    return nr;
}

// ---- CPython's list(set(a) - set(b)) --------------------------------------------------------------------------------------------------
// linear_assignment.py:144 builds the cascade's unmatched tracks as `list(set(track_indices) - set(k for k, _ in matches))`, and
// tracker.py:118-120 keeps that list's order for the confirmed tracks that enter the IoU stage -- the order of the IoU cost matrix's rows,
// hence of the rejected pairs min_cost_matching appends to the unmatched detections, hence of the ids new tracks receive.  A Python set of
// small ints iterates in hash-table slot order: ascending as long as every key is smaller than the table, anything else once a key wraps
// around (a tracker with 40 confirmed tracks whose tracks 13 and 45 miss a frame yields [45, 13]).  The reference's results therefore
// depend on CPython's set implementation (Objects/setobject.c, unchanged in the respects below from 3.7 to 3.12: 8-slot minimum table,
// hash(int) = int, 9 linear probes then i = 5 i + 1 + perturb with perturb >>= 5, growth to the first power of two above 4 x used when
// fill x 5 >= mask x 3; set_difference copies the left set when it is more than four times larger than the right one and otherwise
// adds the surviving keys to a new set in the left set's iteration order).  Restated here for keys in [0, 512); tables of int16,
// -1 = empty.  tests/test_track_core_host.py checks it against real Python sets.
struct PySetSim { short* tab; short* spare; int mask, fill; };
VC_HD void pyset_clear(Lanes L, short* t, int n) {
    for (int i = L.lane; i < n; i += L.n) t[i] = -1;
    wave_sync();
}
VC_HD void pyset_insert_clean(short* tab, int mask, int key) {     // set_insert_clean: the table holds no equal key and no dummy
    unsigned perturb = (unsigned)key;
    int i = key & mask;
    while (true) {
        if (tab[i] < 0) { tab[i] = (short)key; return; }
        if (i + 9 <= mask)
            for (int j = 1; j <= 9; ++j)
                if (tab[i + j] < 0) { tab[i + j] = (short)key; return; }
        perturb >>= 5;
        i = (int)(((unsigned)i * 5u + 1u + perturb) & (unsigned)mask);
    }
}
VC_HD void pyset_resize(Lanes L, PySetSim& s, int minused) {         // set_table_resize: re-insert in slot order
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    short* old = s.tab;
    const int oldn = s.mask + 1;
    s.tab = s.spare; s.spare = old; s.mask = newsize - 1;
    pyset_clear(L, s.tab, newsize);
    for (int i = 0; i < oldn; ++i) {
        const int k = old[i];
        if (k >= 0) pyset_insert_clean(s.tab, s.mask, k);      // every lane performs the same accesses: program order is enough
    }
}
VC_HD void pyset_add(Lanes L, PySetSim& s, int key) {               // set_add_entry for a key that is not in the set yet
    pyset_insert_clean(s.tab, s.mask, key);                         // (same probe sequence: no dummies, no equal keys)
    ++s.fill;
    if (s.fill * 5 >= s.mask * 3) pyset_resize(L, s, s.fill * 4);
}
VC_HD int pyset_seq_mask(int n) { return n < 5 ? 7 : n < 19 ? 31 : n < 77 ? 127 : n < 307 ? 511 : 2047; }   // mask after n adds to an empty set
// conf[0 .. n1): ascending keys (the confirmed tracks' list positions); matched(key): key is in the right-hand set, n2 of them.
// out[0 .. n1 - n2): the surviving keys in CPython's iteration order.  mem0 / mem1: two tables of at least `slots` int16 each, where
// slots >= pyset_need_slots(n1).  Returns the number of survivors.
VC_HD int pyset_copy_mask(int n1) { int ns = 8; if (n1 * 5 >= 21) while (ns <= 2 * n1) ns <<= 1; return ns - 1; }   // set_merge into an empty set
VC_HD int pyset_need_slots(int n1) { return imax(pyset_seq_mask(n1), pyset_copy_mask(n1)) + 1; }
// is the result simply ascending?  n1 keys with maximum max1 on the left, n2 of them removed, the largest survivor is maxu
VC_HD bool pyset_difference_sorted(int n1, int n2, int max1, int maxu) {
#ifdef VC_PYSET_ALWAYS_SORTED            // tests only (negative control): the pre-round-3 behaviour, ascending order whatever the keys
    return true;
#endif
    const int mask1 = pyset_seq_mask(n1);
    const bool s1_sorted = max1 <= mask1;                   // ascending whenever no key wraps around in a table that decides the order
    const bool copy_path = (n1 >> 2) > n2;                  // set_difference: copy the left set and discard / build a new set
    const int copy_mask = pyset_copy_mask(n1);
    return copy_path ? (copy_mask == mask1 ? s1_sorted : max1 <= copy_mask) : (s1_sorted && maxu <= pyset_seq_mask(n1 - n2));
}
template <class M>
VC_HD int pyset_difference_order(Lanes L, const int* conf, int n1, M matched, int n2, short* mem0, short* mem1, int* out) {
    const int r = n1 - n2;
    if (r <= 0 || n1 <= 0) return 0;
    const int max1 = conf[n1 - 1];
    const int mask1 = pyset_seq_mask(n1);
    const bool copy_path = (n1 >> 2) > n2;
    const int copy_mask = pyset_copy_mask(n1);
    const bool s1_sorted = max1 <= mask1;
    int maxu = -1;
    for (int q = n1 - 1; q >= 0; --q) if (!matched(conf[q])) { maxu = conf[q]; break; }
    if (pyset_difference_sorted(n1, n2, max1, maxu))
        return compact(L, n1, [&](int q) { return !matched(conf[q]); }, [&](int pos, int q) { out[pos] = conf[q]; });
    // the left set, as set(list) builds it
    PySetSim a{mem0, mem1, 7, 0};
    if (!s1_sorted) {
        pyset_clear(L, a.tab, 8);
        for (int q = 0; q < n1; ++q) pyset_add(L, a, conf[q]);
    }
    // its iteration order (slot order; ascending when nothing wrapped), filtered, goes through the result set
    auto for_each_a = [&](auto&& f) {
        if (s1_sorted) { for (int q = 0; q < n1; ++q) f(conf[q]); }
        else { for (int i = 0; i <= a.mask; ++i) { const int k = a.tab[i]; if (k >= 0) f(k); } }
    };
    int n = 0;
    if (copy_path) {
        if (!s1_sorted && copy_mask == a.mask) {            // same table size: the copy is slot for slot
            for_each_a([&](int k) { if (!matched(k)) { if (L.lane == 0) out[n] = k; ++n; } });
            wave_sync();
            return n;
        }
        short* rt = s1_sorted ? mem0 : a.spare;             // set_merge into an empty set: set_insert_clean in the source's slot order
        pyset_clear(L, rt, copy_mask + 1);
        for_each_a([&](int k) { pyset_insert_clean(rt, copy_mask, k); });
        for (int i = 0; i <= copy_mask; ++i) { const int k = rt[i]; if (k >= 0 && !matched(k)) { if (L.lane == 0) out[n] = k; ++n; } }
        wave_sync();
        return n;
    }
    // new set: survivors are added in the left set's iteration order.  The left table must stay intact while the result grows, so the
    // result ping-pongs between the second table's halves... it is at most as large as the left one: use its own pair of buffers
    // carved from the spare table when the left set was simulated, else the two tables themselves.
    int keys_n = 0;
    for_each_a([&](int k) { if (!matched(k)) { if (L.lane == 0) out[keys_n] = k; ++keys_n; } });      // insertion order, staged in out[]
    wave_sync();
    PySetSim rs{mem0, mem1, 7, 0};
    pyset_clear(L, rs.tab, 8);
    for (int q = 0; q < keys_n; ++q) pyset_add(L, rs, out[q]);
    for (int i = 0; i <= rs.mask; ++i) { const int k = rs.tab[i]; if (k >= 0) { if (L.lane == 0) out[n] = k; ++n; } }
    wave_sync();
    return n;
}

// linear_assignment.py:52-77 on the sub-matrix cost[rows[i] * ld + cols[j]].  Accepted pairs are appended to match_t / match_d
// (n_match advances); unmatched rows / columns go to un_rows / un_cols in the reference's list order (unassigned ones first,
// in index order, then the rejected pairs in row order).
VC_HD void min_cost_matching(Lanes L, const StepWork& w, const int* rows, int nr, const int* cols, int nc, const double* cost, int ld,
                             double max_cost, double* cbuf, double* tbuf, int& n_match, int* un_rows, int& n_ur, int* un_cols, int& n_uc,
                             int& err) {
    if (nr == 0 || nc == 0) {
        for (int i = L.lane; i < nr; i += L.n) un_rows[i] = rows[i];
        for (int j = L.lane; j < nc; j += L.n) un_cols[j] = cols[j];
        n_ur = nr; n_uc = nc;
        wave_sync();
        return;
    }
    // The sub-matrix is gathered ONCE, clamped, straight into the orientation the solver scans (SciPy solves the transpose when there
    // are more rows than columns) and into the fastest memory that holds it: the 2 KB static LDS buffer, the batch's LDS matrix
    // (w.lmat, dense scenes), else the workgroup's global scratch.  Eight independent reads per lane and round: the cost rows of a
    // dense step live in global memory, and one dependent round trip per element made this loop the larger part of the matching.
    const bool tr = nr > nc && nc > 1;
    const int n = nr * nc;
    double* mat = n <= w.small_n ? w.small_c : (w.lmat && n <= w.lmat_n ? w.lmat : (tr ? tbuf : cbuf));
    unsigned long long adm = 0;                             // does ANY pair of the sub-matrix survive the threshold?
    for (int e0 = L.lane; e0 < n; e0 += 8 * L.n) {          // parallel gather + clamp
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * L.n, ee = e < n ? e : 0, i = ee / nc, j = ee - i * nc;
            v[k] = cost[(size_t)rows[i] * ld + cols[j]];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + k * L.n;
            if (e < n) {
                const int i = e / nc, j = e - i * nc;
                mat[tr ? (size_t)j * nr + i : (size_t)e] = v[k] > max_cost ? max_cost + 1e-5 : v[k];
                if (!(v[k] > max_cost)) adm = 1;
            }
        }
    }
    if (wave_or(L, adm) == 0) {
        // Every entry is the clamp value: on a constant matrix SciPy's solver pairs row i with column i for i < min(nr, nc) (checked
        // against scipy.optimize.linear_sum_assignment for every shape up to 40 x 40 and in tests/test_track_core_host.py), all of these
        // pairs are rejected, and the reference's lists come out rotated: the unassigned tail first, then the rejected pairs in row
        // order (linear_assignment.py:63-76).  A cascade level whose tracks have lost their objects -- most levels of a crowded scene
        // -- costs one pass over its sub-matrix instead of an assignment.
        const int k = imin(nr, nc);
        for (int p = L.lane; p < nc; p += L.n) un_cols[p] = p < nc - k ? cols[k + p] : cols[p - (nc - k)];
        for (int p = L.lane; p < nr; p += L.n) un_rows[p] = p < nr - k ? rows[k + p] : rows[p - (nr - k)];
        n_ur = nr; n_uc = nc;
        wave_sync();
        return;
    }
    for (int i = L.lane; i < nr; i += L.n) w.row_used[i] = 0;
    for (int j = L.lane; j < nc; j += L.n) w.col_used[j] = 0;
    wave_sync();
    const int np = tr ? lap_solve_tr(L, w, mat, nr, nc, err) : lap_solve(L, w, mat, nr, nc, tbuf, err);
    for (int k = L.lane; k < np; k += L.n) { w.row_used[w.ri[k]] = 1; w.col_used[w.ci[k]] = 1; }
    wave_sync();
    n_uc = compact(L, nc, [&](int j) { return !w.col_used[j]; }, [&](int pos, int j) { un_cols[pos] = cols[j]; });
    n_ur = compact(L, nr, [&](int i) { return !w.row_used[i]; }, [&](int pos, int i) { un_rows[pos] = rows[i]; });
    auto rejected = [&](int k) { return cost[(size_t)rows[w.ri[k]] * ld + cols[w.ci[k]]] > max_cost; };     // the clamp maps exactly these to max + 1e-5
    const int nrej = compact(L, np, rejected, [&](int pos, int k) { un_rows[n_ur + pos] = rows[w.ri[k]]; un_cols[n_uc + pos] = cols[w.ci[k]]; });
    const int nacc = compact(L, np, [&](int k) { return !rejected(k); },
                             [&](int pos, int k) { w.match_t[n_match + pos] = rows[w.ri[k]]; w.match_d[n_match + pos] = cols[w.ci[k]]; });
    n_ur += nrej; n_uc += nrej; n_match += nacc;
    wave_sync();
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- the same matching for steps with at most 64 tracks and 64 detections: every list lives in registers -----------------------
// A list is one register per lane (lane p = list position p); lists are built by PUSHING: lane l sends its value to lane `dst`
// (ds_permute), lanes with nothing to send aim at a position the list does not use.
__device__ __forceinline__ int lane_push(int dst, int value) { return __builtin_amdgcn_ds_permute(dst << 2, value); }

// min_cost_matching (linear_assignment.py:52-77) on rows (track indices, nr of them) x cols (detection indices, nc of them) with
// cost_at(track, detection).  Accepted pairs are appended to w.match_t / w.match_d in row order (n_match advances); acc_row: this
// lane's row was matched; un_rows / un_cols: the reference's unmatched lists (unassigned entries in list order, then the rejected
// pairs in row order).
template <class CostAt>
__device__ __forceinline__ int mcm_reg(int lane, int rows, int nr, int cols, int nc, CostAt cost_at, double max_cost, const StepWork& w,
                                       int& n_match, int& un_rows, int& n_ur, int& un_cols, int& n_uc, bool& acc_row) {
    acc_row = false;
    if (nr == 0 || nc == 0) { un_rows = rows; n_ur = nr; un_cols = cols; n_uc = nc; return TERR_NONE; }
    auto clampc = [&](double v) { return v > max_cost ? max_cost + 1e-5 : v; };
    int acol, arow;                                         // row lane: position of its column (-1 none); column lane: of its row
    if (nr <= nc) {
        int c4r, r4c;
        if (lsap_reg(lane, nr, nc, [&](int i, int) { return clampc(cost_at(lane_get(rows, i), cols)); }, c4r, r4c) != 0) return TERR_LAP;
        acol = lane < nr ? c4r : -1; arow = lane < nc ? r4c : -1;
    } else {                                                // SciPy transposes so that rows <= columns
        int c4r, r4c;
        if (lsap_reg(lane, nc, nr, [&](int i, int) { return clampc(cost_at(rows, lane_get(cols, i))); }, c4r, r4c) != 0) return TERR_LAP;
        arow = lane < nc ? c4r : -1; acol = lane < nr ? r4c : -1;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int my_det = __shfl(cols, acol >= 0 ? acol : 0);                       // row lanes: detection of the assigned column
    const bool is_row = lane < nr, is_col = lane < nc;
    const bool rej_row = is_row && acol >= 0 && clampc(cost_at(rows, my_det)) > max_cost;
    acc_row = is_row && acol >= 0 && !rej_row;
    const bool un_row = is_row && acol < 0, un_col = is_col && arow < 0;
    const unsigned long long m_ur = __ballot(un_row), m_rej = __ballot(rej_row), m_acc = __ballot(acc_row), m_uc = __ballot(un_col);
    const int n_ur0 = __popcll(m_ur), nrej = __popcll(m_rej), n_uc0 = __popcll(m_uc);
    const int rank_rej = __popcll(m_rej & lt);
    un_rows = lane_push(un_row ? __popcll(m_ur & lt) : rej_row ? n_ur0 + rank_rej : 63, rows);      // lane 63 is past the list unless all 64 lanes send
    const int uc_a = lane_push(un_col ? __popcll(m_uc & lt) : 63, cols);
    const int uc_b = lane_push(rej_row ? n_uc0 + rank_rej : (n_uc0 > 0 ? 0 : 63), my_det);
    un_cols = lane < n_uc0 ? uc_a : uc_b;
    if (acc_row) { const int k = n_match + __popcll(m_acc & lt); w.match_t[k] = rows; w.match_d[k] = my_det; }
    n_match += __popcll(m_acc);
    n_ur = n_ur0 + nrej; n_uc = n_uc0 + nrej;
    if (lane >= n_ur) un_rows = 0;
    if (lane >= n_uc) un_cols = 0;
    return TERR_NONE;
}

__device__ __forceinline__ void match_step_wave64(int lane, const StepWork& w, const TrackerHdr& h, int T, int D, const double* cost_app,
                                                  const double* cost_iou, int& n_match, int& n_un, int*& newdets, int& n_new, int& err, long long* prof) {
    if (prof && lane == 0) prof[0] = wall_clock64();
    const int st = lane < T ? w.state[lane] : -1, tsu = lane < T ? w.tsu[lane] : 0;
    const bool conf = st == CONFIRMED;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int left = lane < D ? lane : 0, n_left = D;
    bool matched = false;
    n_match = 0;
    const Lanes L{lane, 64};
    const int n_levels = imin(h.max_age, wave_max(L, conf ? tsu : 0));
    if (prof && lane == 0) prof[1] = wall_clock64();
    auto app_at = [&](int t, int d) { return cost_app[(size_t)t * D + d]; };
    auto iou_at = [&](int t, int d) { return cost_iou[(size_t)t * D + d]; };
    // Stale levels.  A confirmed track that missed its last frames usually has no detection of the step within the threshold (its object
    // has left; the gate turned its whole row into VC_GATED).  If that holds for EVERY track of a level, the level's clamped matrix is
    // constant, SciPy pairs row i with column i, every pair is rejected and min_cost_matching's lists come out with the first
    // min(nl, n_left) columns moved behind the others (min_cost_matching above, the same shortcut on the gathered matrix): nothing is
    // matched and `left` is rotated by nl when nl < n_left, unchanged otherwise.  Such a level costs a ballot here instead of a cost
    // load, a wave reduction and the list exchange (~1.3 us each, 15 - 20 of them per step in a crowded tracker); rotations of successive
    // stale levels add up and are applied once, in front of the next level that has to look at `left`.
    bool adm = false;
    if (conf && tsu >= 2)
        for (int d = 0; d < D; ++d) adm = adm || !(cost_app[(size_t)lane * D + d] > h.max_dist);
    int rot = 0;                                             // pending rotation: the list is left[(p + rot) % n_left]
    auto settle = [&]() {
        if (rot != 0) {
            const int src = lane < n_left ? (lane + rot >= n_left ? lane + rot - n_left : lane + rot) : lane;
            left = __shfl(left, src);
            rot = 0;
        }
    };
    // matching_cascade (linear_assignment.py:80-145): level = time_since_update - 1, most recently seen tracks first
    for (int level = 0; level < n_levels && n_left > 0; ++level) {
        const unsigned long long rm = __ballot(conf && tsu == 1 + level);
        if (rm == 0) continue;
        const int nl = __popcll(rm), rank = __popcll(rm & lt);
        const bool in = (rm >> lane) & 1ull;
        if (level >= 1 && __ballot(in && adm) == 0) {            // (adm is known for time_since_update >= 2, i.e. from level 1 on)
            if (nl < n_left) { rot += nl; if (rot >= n_left) rot -= n_left; }
            continue;
        }
        settle();
        if (nl == 1 || n_left == 1) {
            // One track in the level, or one detection left (most levels of a light scene): the assignment is the minimum of a
            // vector, smallest index among equal minima (SciPy's scan keeps the LAST unassigned minimum of a list that holds the
            // columns in descending order; for one column it solves the transpose), and min_cost_matching's list handling
            // reduces to removing the column when the pair is accepted or moving it to the END of the list when it is rejected
            // (linear_assignment.py:69-76 appends the rejected pairs after the unassigned entries).
            const int t1 = __ffsll(rm) - 1;                                   // nl == 1: the track
            const int d1 = lane_get(left, 0);                                 // n_left == 1: the detection
            const int n = nl == 1 ? n_left : nl;
            // nl == 1: lane e scores detection left[e]; n_left == 1: the lane of track t scores itself (key = -rank keeps list order)
            const bool cand = nl == 1 ? lane < n_left : in;
            const double v = cost_app[(size_t)(nl == 1 ? t1 : (in ? lane : 0)) * D + (nl == 1 ? left : d1)];
            BestK b = {(double)INFINITY, INT_MIN};
            if (cand) { b.v = v > h.max_dist ? h.max_dist + 1e-5 : v; b.key = -(nl == 1 ? lane : rank); }
            int width = 1;
            while (width < (nl == 1 ? n : T)) width <<= 1;
            best_reduce(b, width);
            const double bv = lane_get(b.v, 0);
            const int bi = -lane_get(b.key, 0);                               // list position (nl == 1) / rank in the level (n_left == 1)
            const bool accepted = !(bv > h.max_dist);
            const int ci = nl == 1 ? bi : 0;                                  // position of the column in `left`
            const int col = lane_get(left, ci);
            int trk = t1;
            if (nl != 1) trk = __ffsll((unsigned long long)__ballot(in && rank == bi)) - 1;
            const int nxt = __shfl_down(left, 1);
            if (lane >= ci && lane < n_left - 1) left = nxt;                   // the column leaves its position ...
            if (!accepted && lane == n_left - 1) left = col;                  // ... and goes to the end when the pair was rejected
            if (accepted) {
                if (lane == 0) { w.match_t[n_match] = trk; w.match_d[n_match] = col; }
                if (lane == trk) matched = true;
                ++n_match; --n_left;
                if (lane >= n_left) left = 0;
            }
            (void)n;
            continue;
        }
        int rows = lane_push(in ? rank : 63, lane);
        if (lane >= nl) rows = 0;
        int un_rows, n_ur, un_cols, n_uc;
        bool acc;
        const int e = mcm_reg(lane, rows, nl, left, n_left, app_at, h.max_dist, w, n_match, un_rows, n_ur, un_cols, n_uc, acc);
        if (e != TERR_NONE) { err = e; return; }
        const bool mine = __shfl((int)acc, in ? rank : 0) != 0;
        if (in && mine) matched = true;
        left = un_cols; n_left = n_uc;
    }
    settle();
    if (prof && lane == 0) prof[2] = wall_clock64();
    // IoU stage (tracker.py:118-127): unconfirmed tracks, then the confirmed tracks missed for exactly one frame
    const unsigned long long um = __ballot(lane < T && !conf), rcm = __ballot(conf && !matched && tsu == 1), unm = __ballot(conf && !matched && tsu != 1);
    const int n_unconf = __popcll(um), nr = n_unconf + __popcll(rcm);
    const bool is_u = (um >> lane) & 1ull, is_r = (rcm >> lane) & 1ull;
    int rows;
    // The confirmed tracks missed for one frame enter in the order of the reference's `list(set(track_indices) - set(matched))`
    // (linear_assignment.py:144, tracker.py:118-120): ascending unless a key wraps around in CPython's hash table (pyset_difference_order)
    const unsigned long long cm = __ballot(conf), uam = rcm | unm;
    const int n1 = __popcll(cm);
    if (uam == 0 || pyset_difference_sorted(n1, n_match, 63 - __builtin_clzll(cm), 63 - __builtin_clzll(uam))) {
        rows = lane_push(is_u ? __popcll(um & lt) : is_r ? n_unconf + __popcll(rcm & lt) : 63, lane);
    } else {
        if (conf) w.confirmed[__popcll(cm & lt)] = lane;
        if (lane < T) w.matched[lane] = matched ? 1 : 0;
        wave_sync();
        const int n_ua = pyset_difference_order(Lanes{lane, 64}, w.confirmed, n1, [&](int t) { return w.matched[t] != 0; }, n_match, (short*)w.small_c,
                                                (short*)w.small_t, w.ri);
        wave_sync();
        const int key = lane < n_ua ? w.ri[lane] : 0;
        const int tsu_of_key = __shfl(tsu, key);             // every lane takes part: a shuffle reads nothing from a lane that is masked off
        const bool rec = lane < n_ua && tsu_of_key == 1;
        const unsigned long long recm = __ballot(rec);
        const int rows_u = lane_push(is_u ? __popcll(um & lt) : 63, lane);
        const int rows_r = lane_push(rec ? n_unconf + __popcll(recm & lt) : (n_unconf > 0 ? 0 : 63), key);
        rows = lane < n_unconf ? rows_u : rows_r;
    }
    if (lane >= nr) rows = 0;
    if ((unm >> lane) & 1ull) w.un_tracks[__popcll(unm & lt)] = lane;
    n_un = __popcll(unm);
    int un_rows, n_ur, un_cols, n_uc;
    bool acc;
    const int e = mcm_reg(lane, rows, nr, left, n_left, iou_at, h.max_iou_distance, w, n_match, un_rows, n_ur, un_cols, n_uc, acc);
    if (e != TERR_NONE) { err = e; return; }
    if (lane < n_ur) w.un_tracks[n_un + lane] = un_rows;
    n_un += n_ur;
    if (lane < n_uc) w.left[lane] = un_cols;
    newdets = w.left;
    n_new = n_uc;
    wave_sync();
    if (prof && lane == 0) prof[3] = wall_clock64();
}
#endif

// Tracker._match on the step's cost rows (cost_app: gated appearance rows of the confirmed tracks, cost_iou: IoU rows of the IoU
// candidates; both [T][D], rows of other tracks are never read).  Outputs: matches in w.match_t / w.match_d, missed tracks in
// w.un_tracks, the detections that start new tracks in *newdets (one of w.left / w.un_cols), all in the reference's list order.
VC_HD long long tc_clock() {
#if defined(__HIP_DEVICE_COMPILE__)
    return wall_clock64();
#else
    return 0;
#endif
}

VC_HD void match_step(Lanes L, const StepWork& w, const TrackerHdr& h, int T, int D, const double* cost_app, const double* cost_iou,
                      double* cbuf, double* tbuf, int& n_match, int& n_un, int*& newdets, int& n_new, int& err, long long* prof = nullptr,
                      bool allow_reg = true) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (allow_reg && T <= 64 && D <= 64) { match_step_wave64(L.lane, w, h, T, D, cost_app, cost_iou, n_match, n_un, newdets, n_new, err, prof); return; }
#endif
    if (prof && L.lane == 0) prof[0] = tc_clock();
    const int n_conf = compact(L, T, [&](int t) { return w.state[t] == CONFIRMED; }, [&](int pos, int t) { w.confirmed[pos] = t; });
    const int n_unconf = compact(L, T, [&](int t) { return w.state[t] != CONFIRMED; }, [&](int pos, int t) { w.unconfirmed[pos] = t; });
    int mt = 0;
    for (int t = L.lane; t < T; t += L.n) { w.matched[t] = 0; if (w.state[t] == CONFIRMED) mt = imax(mt, w.tsu[t]); }
    const int max_tsu = wave_max(L, mt);
    for (int d = L.lane; d < D; d += L.n) w.left[d] = d;
    wave_sync();
    int* left = w.left;
    int* other = w.un_cols;
    int n_left = D;
    n_match = 0;
    if (prof && L.lane == 0) prof[1] = tc_clock();
    // matching_cascade: level = time_since_update - 1, most recently seen tracks first.  Only the levels that hold a track are
    // visited (a 64-bit presence mask for the first 64 levels, a plain scan above).
    unsigned long long lm = 0;
    for (int q = L.lane; q < n_conf; q += L.n) { const int lv = w.tsu[w.confirmed[q]] - 1; if (lv >= 0 && lv < 64) lm |= 1ull << lv; }
    const unsigned long long level_mask = wave_or(L, lm);
    const int n_levels = imin(h.max_age, max_tsu);
    for (int level = 0; level < n_levels; ++level) {
        if (n_left == 0) break;
        if (level < 64 && !((level_mask >> level) & 1ull)) continue;
        const int nl = compact(L, n_conf, [&](int q) { return w.tsu[w.confirmed[q]] == 1 + level; }, [&](int pos, int q) { w.rows[pos] = w.confirmed[q]; });
        if (nl == 0) continue;
        wave_sync();
        {
            // A level whose tracks are all stale (w.adm: no detection of the step within the threshold, known since the cost rows were written)
            // is the constant-matrix case of min_cost_matching without gathering its sub-matrix: nothing matched, the first nl unmatched
            // detections move behind the others when nl < n_left (match_step_wave64 has the same shortcut in registers).  In a dense scene
            // (90 tracks, 50 detections) two thirds of the cascade's time went into gathering such levels from global memory.
            unsigned long long any = 0;
            for (int q = L.lane; q < nl; q += L.n) any |= w.adm[w.rows[q]] ? 1ull : 0ull;
            if (wave_or(L, any) == 0) {
                if (nl < n_left) {
                    for (int e = L.lane; e < n_left; e += L.n) other[e] = left[e + nl < n_left ? e + nl : e + nl - n_left];
                    int* sw = left; left = other; other = sw;
                    wave_sync();
                }
                continue;
            }
        }
        if (nl == 1 || n_left == 1) {
            // One track in the level, or one detection left: the assignment is the minimum of a vector (smallest index among equal
            // minima, lap_solve), and min_cost_matching's list handling reduces to removing the column when the pair is accepted
            // or moving it to the END of the list when it is rejected (linear_assignment.py:69-76 appends rejected pairs last).
            const int n = nl == 1 ? n_left : nl;
            Best b = {(double)INFINITY, 0, INT_MAX};
            for (int e = L.lane; e < n; e += L.n) {
                const int t = nl == 1 ? w.rows[0] : w.rows[e], d = nl == 1 ? left[e] : left[0];
                const double v = cost_app[(size_t)t * D + d];
                const Best x = {v > h.max_dist ? h.max_dist + 1e-5 : v, 0, e};
                if (better(x, b)) b = x;
            }
            b = wave_best(L, b);
            const bool accepted = !(b.v > h.max_dist);
            const int t = nl == 1 ? w.rows[0] : w.rows[b.it];
            const int ci = nl == 1 ? b.it : 0, col = left[ci];
            for (int e = L.lane; e < n_left; e += L.n)
                if (e != ci) other[e < ci ? e : e - 1] = left[e];
            if (!accepted && L.lane == 0) other[n_left - 1] = col;
            if (accepted && L.lane == 0) { w.match_t[n_match] = t; w.match_d[n_match] = col; w.matched[t] = 1; }
            if (accepted) { ++n_match; --n_left; }
            int* sw = left; left = other; other = sw;
            wave_sync();
            continue;
        }
        const int m0 = n_match;
        int n_ur = 0, n_uc = 0;
        min_cost_matching(L, w, w.rows, nl, left, n_left, cost_app, D, h.max_dist, cbuf, tbuf, n_match, w.un_rows, n_ur, other, n_uc, err);
        for (int k = m0 + L.lane; k < n_match; k += L.n) w.matched[w.match_t[k]] = 1;
        int* sw = left; left = other; other = sw;
        n_left = n_uc;
        wave_sync();
    }
    if (prof && L.lane == 0) prof[2] = tc_clock();
    // IoU stage (tracker.py:118-127): unconfirmed tracks + confirmed tracks that were missed for exactly one frame
    for (int q = L.lane; q < n_unconf; q += L.n) w.rows[q] = w.unconfirmed[q];
    // the cascade's unmatched tracks in the order of the reference's `list(set(track_indices) - set(matched))` (linear_assignment.py:144):
    // that order is the row order of the IoU problem for the tracks missed for exactly one frame (tracker.py:118-120)
    const int need = pyset_need_slots(n_conf);
    short* const m0 = (w.small_c && need <= w.small_n * 4) ? (short*)w.small_c : (short*)cbuf;
    short* const m1 = (w.small_t && need <= w.small_n * 4) ? (short*)w.small_t : (short*)tbuf;
    wave_sync();
    const int n_ua = pyset_difference_order(L, w.confirmed, n_conf, [&](int t) { return w.matched[t] != 0; }, n_match, m0, m1, w.ri);
    wave_sync();
    const int n_recent = compact(L, n_ua, [&](int q) { return w.tsu[w.ri[q]] == 1; }, [&](int pos, int q) { w.rows[n_unconf + pos] = w.ri[q]; });
    n_un = compact(L, n_ua, [&](int q) { return w.tsu[w.ri[q]] != 1; }, [&](int pos, int q) { w.un_tracks[pos] = w.ri[q]; });
    wave_sync();
    int n_ur = 0, n_uc = 0;
    min_cost_matching(L, w, w.rows, n_unconf + n_recent, left, n_left, cost_iou, D, h.max_iou_distance, cbuf, tbuf, n_match, w.un_rows, n_ur, other,
                      n_uc, err);
    for (int k = L.lane; k < n_ur; k += L.n) w.un_tracks[n_un + k] = w.un_rows[k];
    n_un += n_ur;
    newdets = other;
    n_new = n_uc;
    wave_sync();
    if (prof && L.lane == 0) prof[3] = tc_clock();
}

// Track.update / mark_missed / _initiate_track bookkeeping + Tracker.update's list maintenance, after the Kalman and gallery
// writes of the step have been applied to the pool.  Survivors keep their order, new tracks are appended in `newdets` order with
// consecutive ids.  The slots of deleted tracks are handed to free_slots(slots, n).  Returns the number of live tracks.
template <class FreeSlots>
VC_HD int finish_step(Lanes L, const StepWork& w, TrackerHdr* hdr, int* list, TrackRecD* recs, int T, int n_match, int n_un, int n_new,
                      FreeSlots free_slots) {
    const TrackerHdr h = *hdr;
    for (int k = L.lane; k < n_match; k += L.n) {            // Track.update (track.py:126-145); a track is matched at most once
        const int t = w.match_t[k];
        const int hits = w.hits[t] + 1;                      // the record's counters were loaded into the work arrays with the step
        w.hits[t] = hits;
        w.tsu[t] = 0;
        w.galh[t] = (w.galh[t] + 1) % h.nn_budget;
        w.galc[t] = imin(w.galc[t] + 1, h.nn_budget);
        if (w.state[t] == TENTATIVE && hits >= h.n_init) w.state[t] = CONFIRMED;
    }
    for (int k = L.lane; k < n_un; k += L.n) {               // Track.mark_missed (track.py:147-153)
        const int t = w.un_tracks[k];
        if (w.state[t] == TENTATIVE) w.state[t] = DELETED;
        else if (w.tsu[t] > h.max_age) w.state[t] = DELETED;
    }
    wave_sync();
    for (int t = L.lane; t < T; t += L.n) {                  // write the counters back (age and time_since_update were advanced at load)
        TrackRecD& r = recs[w.slot[t]];
        r.state = w.state[t]; r.tsu = w.tsu[t]; r.gal_count = w.galc[t]; r.gal_head = w.galh[t]; r.hits = w.hits[t];
    }
    const int n_surv = compact(L, T, [&](int t) { return w.state[t] != DELETED; }, [&](int pos, int t) { list[pos] = w.slot[t]; });
    const int n_del = compact(L, T, [&](int t) { return w.state[t] == DELETED; }, [&](int pos, int t) { w.rows[pos] = w.slot[t]; });
    wave_sync();
    if (n_del > 0) free_slots(w.rows, n_del);
    for (int i = L.lane; i < n_new; i += L.n) {              // _initiate_track (tracker.py:133-139)
        TrackRecD r;
        r.id = h.next_id + i; r.state = TENTATIVE; r.hits = 1; r.age = 1; r.tsu = 0; r.gal_count = 1; r.gal_head = 1 % h.nn_budget;
        recs[w.newslot[i]] = r;
        list[n_surv + i] = w.newslot[i];
    }
    if (L.lane == 0) { hdr->n_tracks = n_surv + n_new; hdr->next_id = h.next_id + n_new; }
    wave_sync();
    return n_surv + n_new;
}

// deep_sort.py:46-58 + 97-108: rows [x1, y1, x2, y2, id, label] of the confirmed tracks seen within the last frame (box = Kalman
// posterior, int() truncation, clamped to the frame), in list order.  Works on the step's arrays after finish_step (positions of the
// step's START: survivors keep their order and the tracks born in this step are tentative, so this is the list order of the rows).
// emit(position, row6) receives them; returns the count.
template <class MeanOf, class Emit>
VC_HD int emit_rows(Lanes L, const StepWork& w, MeanOf mean_of, int T, int W, int H, int label, Emit emit) {
    return compact(L, T, [&](int t) { return w.state[t] == CONFIRMED && w.tsu[t] <= 1; },
                   [&](int pos, int t) {
                       const double* m = mean_of(t);                                  // posterior mean of list position t
                       const double bw = m[2] * m[3], bh = m[3];                     // track.py:82-96 to_tlwh
                       const double x = m[0] - bw / 2, y = m[1] - bh / 2;
                       long long row[6];
                       const long long x1 = (long long)x, y1 = (long long)y, x2 = (long long)(x + bw), y2 = (long long)(y + bh);
                       row[0] = x1 > 0 ? x1 : 0; row[1] = y1 > 0 ? y1 : 0;
                       row[2] = x2 < W - 1 ? x2 : W - 1; row[3] = y2 < H - 1 ? y2 : H - 1;
                       row[4] = w.id[t]; row[5] = label;
                       emit(pos, row);
                   });
}

}  // namespace tc
}  // namespace vc
