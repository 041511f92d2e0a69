"""CPU ORACLE (test infrastructure, never a product path) -- DeepSORT track side, rows B1-B13 of SURVEY.md section 8.

A NumPy/SciPy restatement of the reference's tracker arithmetic.  It is written around flat
records and explicit index sets instead of the reference's object graph, but every numeric
step follows the cited reference line so that results agree bit for bit where the reference
is deterministic (fp64 Kalman, SciPy LSA) and to f32 rounding where BLAS order is free
(cosine GEMM).  Pinned against golden vectors produced by importing the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference map (all paths relative to /root/reference/networks/deepsort):
  KalmanCV            sort/kalman_filter.py:23-229
  dsort_nms           sort/preprocessing.py:6-73
  iou_cost_matrix     sort/iou_matching.py:7-81
  cosine_nn_cost      sort/nn_matching.py:31-54,78-96,160-177
  assign_min_cost     sort/linear_assignment.py:13-77
  cascade             sort/linear_assignment.py:80-145
  gate                sort/linear_assignment.py:148-192
  TrackerState.*      sort/tracker.py:40-139, sort/track.py:4-175
  DeepSortOracle      deep_sort.py:15-129
  VideoTrackerOracle  ../../modules/track.py:9-70
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

CHI2_95_4DOF = 9.4877          # sort/kalman_filter.py:11-20 (chi2inv95[4])
GATED_COST = 1e5               # sort/linear_assignment.py:9  (INFTY_COST)
TENTATIVE, CONFIRMED, DELETED = 1, 2, 3   # sort/track.py:14-16


# --------------------------------------------------------------------------- B8 Kalman
class KalmanCV:
    """8-state constant-velocity filter over (cx, cy, aspect, h); sort/kalman_filter.py:23-229."""

    W_POS = 1.0 / 20            # :52
    W_VEL = 1.0 / 160           # :53

    def __init__(self):
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0          # :44-47, dt = 1
        self.H = np.eye(4, 8)               # :48

    def initiate(self, xyah):
        h = xyah[3]
        mean = np.r_[xyah, np.zeros(4)]
        std = np.array([2 * self.W_POS * h, 2 * self.W_POS * h, 1e-2, 2 * self.W_POS * h,
                        10 * self.W_VEL * h, 10 * self.W_VEL * h, 1e-5, 10 * self.W_VEL * h])  # :76-84
        return mean, np.diag(std * std)

    def predict(self, mean, cov):
        h = mean[3]
        std = np.array([self.W_POS * h, self.W_POS * h, 1e-2, self.W_POS * h,
                        self.W_VEL * h, self.W_VEL * h, 1e-5, self.W_VEL * h])              # :107-116
        q = np.diag(std * std)
        return self.F @ mean, np.linalg.multi_dot((self.F, cov, self.F.T)) + q               # :119-121

    def project(self, mean, cov):
        h = mean[3]
        std = np.array([self.W_POS * h, self.W_POS * h, 1e-1, self.W_POS * h])               # :142-146
        r = np.diag(std * std)
        return self.H @ mean, np.linalg.multi_dot((self.H, cov, self.H.T)) + r               # :149-152

    def update(self, mean, cov, z):
        pm, pc = self.project(mean, cov)
        cf = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)                     # :176-177
        gain = scipy.linalg.cho_solve(cf, (cov @ self.H.T).T, check_finite=False).T          # :178-180
        innov = z - pm
        return mean + innov @ gain.T, cov - np.linalg.multi_dot((gain, pc, gain.T))          # :183-186

    def gating(self, mean, cov, zs):
        pm, pc = self.project(mean, cov)
        L = np.linalg.cholesky(pc)                                                           # :223
        y = scipy.linalg.solve_triangular(L, (zs - pm).T, lower=True, check_finite=False)    # :225-227
        return np.sum(y * y, axis=0)


# --------------------------------------------------------------------------- B6 helpers
def tlwh_to_xyah(tlwh):
    """sort/detection.py:42-50."""
    r = np.array(tlwh, dtype=np.float64)
    r[:2] += r[2:] / 2
    r[2] /= r[3]
    return r


def mean_to_tlwh(mean):
    """sort/track.py:82-96."""
    r = mean[:4].copy()
    r[2] *= r[3]
    r[:2] -= r[2:] / 2
    return r


# --------------------------------------------------------------------------- B7 DeepSORT "NMS"
def dsort_nms(tlwh, max_overlap, scores):
    """Greedy suppression with overlap = inter / area(other), +1 pixel convention.

    sort/preprocessing.py:6-73 (quirk Q6).  Returns picked indices, highest score first.
    """
    if len(tlwh) == 0:
        return []
    b = np.asarray(tlwh, dtype=np.float64)
    x1, y1 = b[:, 0], b[:, 1]
    x2, y2 = b[:, 2] + b[:, 0], b[:, 3] + b[:, 1]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(scores)            # ascending; default quicksort as the reference
    keep = []
    while len(order):
        i = order[-1]
        rest = order[:-1]
        keep.append(int(i))
        w = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        ov = (w * h) / area[rest]
        order = rest[~(ov > max_overlap)]
    return keep


# --------------------------------------------------------------------------- B13 IoU
def iou_one_to_many(box, cands):
    """sort/iou_matching.py:7-38, boxes are tlwh, no +1."""
    tl = np.maximum(box[:2], cands[:, :2])
    br = np.minimum(box[:2] + box[2:], cands[:, :2] + cands[:, 2:])
    wh = np.maximum(0.0, br - tl)
    inter = wh[:, 0] * wh[:, 1]
    return inter / (box[2] * box[3] + cands[:, 2] * cands[:, 3] - inter)


# --------------------------------------------------------------------------- B11 cosine
def cosine_nn_cost(samples, feats):
    """min over gallery samples of (1 - cos); sort/nn_matching.py:31-54,78-96."""
    a = np.asarray(samples)
    b = np.asarray(feats)
    a = a / np.linalg.norm(a, axis=1, keepdims=True)
    b = b / np.linalg.norm(b, axis=1, keepdims=True)
    return (1.0 - a @ b.T).min(axis=0)


# --------------------------------------------------------------------------- B12 assignment
def assign_min_cost(cost, max_cost, rows, cols):
    """Threshold-clamp, rectangular LSA, reject > max; sort/linear_assignment.py:52-77.

    `cost` is (len(rows), len(cols)); returns (matches, unmatched_rows, unmatched_cols) as ids
    drawn from `rows`/`cols`, in the reference's list order.
    """
    if len(rows) == 0 or len(cols) == 0:
        return [], list(rows), list(cols)
    c = np.array(cost, dtype=np.float64, copy=True)
    c[c > max_cost] = max_cost + 1e-5
    ri, ci = linear_sum_assignment(c)
    m, ur, uc = [], [], []
    cset, rset = set(ci.tolist()), set(ri.tolist())
    for j, cid in enumerate(cols):
        if j not in cset:
            uc.append(cid)
    for i, rid in enumerate(rows):
        if i not in rset:
            ur.append(rid)
    for i, j in zip(ri, ci):
        if c[i, j] > max_cost:
            ur.append(rows[i])
            uc.append(cols[j])
        else:
            m.append((rows[i], cols[j]))
    return m, ur, uc


class _Track:
    __slots__ = ("mean", "cov", "tid", "hits", "age", "tsu", "state", "feats", "confs")


class TrackerState:
    """One class's tracker (sort/tracker.py:40-139) + gallery (sort/nn_matching.py:99-177)."""

    def __init__(self, max_cos, budget, max_iou_distance=0.7, max_age=70, n_init=3):
        self.kf = KalmanCV()
        self.max_cos, self.budget = max_cos, budget
        self.max_iou, self.max_age, self.n_init = max_iou_distance, max_age, n_init
        self.tracks: list[_Track] = []
        self.next_id = 1
        self.gallery: dict[int, list[np.ndarray]] = {}
        self.last_matches = []          # diagnostics for the golden traces

    # sort/tracker.py:50-56 + sort/track.py:112-124
    def predict(self):
        for t in self.tracks:
            t.mean, t.cov = self.kf.predict(t.mean, t.cov)
            t.age += 1
            t.tsu += 1

    def _appearance_cost(self, dets, tidx, didx):
        feats = np.array([dets[i]["feature"] for i in didx])
        cost = np.zeros((len(tidx), len(didx)))
        for r, k in enumerate(tidx):
            cost[r] = cosine_nn_cost(self.gallery[self.tracks[k].tid], feats)        # nn_matching.py:174-176
        zs = np.asarray([tlwh_to_xyah(dets[i]["tlwh"]) for i in didx])
        for r, k in enumerate(tidx):
            g = self.kf.gating(self.tracks[k].mean, self.tracks[k].cov, zs)
            cost[r, g > CHI2_95_4DOF] = GATED_COST                                    # linear_assignment.py:187-191
        return cost

    def _iou_cost(self, dets, tidx, didx):
        cost = np.zeros((len(tidx), len(didx)))
        cands = np.asarray([dets[i]["tlwh"] for i in didx])
        for r, k in enumerate(tidx):
            if self.tracks[k].tsu > 1:
                cost[r] = GATED_COST                                                 # iou_matching.py:74-76
            else:
                cost[r] = 1.0 - iou_one_to_many(mean_to_tlwh(self.tracks[k].mean), cands)
        return cost

    def _match(self, dets):
        conf = [i for i, t in enumerate(self.tracks) if t.state == CONFIRMED]
        unconf = [i for i, t in enumerate(self.tracks) if t.state != CONFIRMED]
        # cascade, sort/linear_assignment.py:124-145
        left = list(range(len(dets)))
        m_a = []
        for level in range(self.max_age):
            if not left:
                break
            lvl = [k for k in conf if self.tracks[k].tsu == 1 + level]
            if not lvl:
                continue
            m, _, left = assign_min_cost(self._appearance_cost(dets, lvl, left), self.max_cos, lvl, left)
            m_a += m
        um_a = list(set(conf) - set(k for k, _ in m_a))
        # IoU stage, sort/tracker.py:118-127
        cand = unconf + [k for k in um_a if self.tracks[k].tsu == 1]
        um_a = [k for k in um_a if self.tracks[k].tsu != 1]
        if len(cand) == 0 or len(left) == 0:
            m_b, um_b, left2 = [], cand, left
        else:
            m_b, um_b, left2 = assign_min_cost(self._iou_cost(dets, cand, left), self.max_iou, cand, left)
        return m_a + m_b, list(set(um_a + um_b)), left2

    def update(self, dets):
        """dets: list of {"tlwh": f64[4], "conf": float, "feature": f32[512]}; sort/tracker.py:58-91."""
        matches, um_tracks, um_dets = self._match(dets)
        self.last_matches = [(int(a), int(b)) for a, b in matches]
        for k, d in matches:                                   # sort/track.py:126-145
            t = self.tracks[k]
            t.mean, t.cov = self.kf.update(t.mean, t.cov, tlwh_to_xyah(dets[d]["tlwh"]))
            t.feats.append(dets[d]["feature"])
            t.confs.append(dets[d]["conf"])
            t.hits += 1
            t.tsu = 0
            if t.state == TENTATIVE and t.hits >= self.n_init:
                t.state = CONFIRMED
        for k in um_tracks:                                    # sort/track.py:147-153
            t = self.tracks[k]
            if t.state == TENTATIVE or t.tsu > self.max_age:
                t.state = DELETED
        for d in um_dets:                                      # sort/tracker.py:133-139
            t = _Track()
            t.mean, t.cov = self.kf.initiate(tlwh_to_xyah(dets[d]["tlwh"]))
            t.tid, t.hits, t.age, t.tsu, t.state = self.next_id, 1, 1, 0, TENTATIVE
            t.feats, t.confs = [dets[d]["feature"]], [dets[d]["conf"]]
            self.tracks.append(t)
            self.next_id += 1
        self.tracks = [t for t in self.tracks if t.state != DELETED]
        # gallery, sort/tracker.py:82-91 + sort/nn_matching.py:137-154
        active = [t.tid for t in self.tracks if t.state == CONFIRMED]
        for t in self.tracks:
            if t.state != CONFIRMED:
                continue
            for f in t.feats:
                g = self.gallery.setdefault(t.tid, [])
                g.append(f)
                if self.budget is not None:
                    self.gallery[t.tid] = g[-self.budget:]
            t.feats = []
        self.gallery = {k: self.gallery[k] for k in active}


# --------------------------------------------------------------------------- B2-B4 glue
def xyxy_to_cxcywh(b):
    """deep_sort.py:78-87."""
    o = np.array(b, dtype=np.float64, copy=True)
    o[:, 2] = b[:, 2] - b[:, 0]
    o[:, 3] = b[:, 3] - b[:, 1]
    o[:, 0] = o[:, 0] + o[:, 2] / 2
    o[:, 1] = o[:, 1] + o[:, 3] / 2
    return o


def crop_corners(cxcywh, width, height):
    """int() truncation + clamp to [0, W-1]/[0, H-1]; deep_sort.py:89-95 (quirk Q4)."""
    x, y, w, h = cxcywh
    return (max(int(x - w / 2), 0), max(int(y - h / 2), 0),
            min(int(x + w / 2), width - 1), min(int(y + h / 2), height - 1))


class DeepSortOracle:
    """deep_sort.py:15-59.  `embed(crops)->(k,512) f32` stands in for Extractor.__call__."""

    def __init__(self, embed, max_dist=0.2, min_confidence=0.3, nms_max_overlap=1.0,
                 max_iou_distance=0.7, max_age=70, n_init=3, nn_budget=100):
        self.embed = embed
        self.min_conf, self.nms_ov = min_confidence, nms_max_overlap
        self.trk = TrackerState(max_dist, nn_budget, max_iou_distance, max_age, n_init)

    def update(self, bbox_xyxy, confidences, ori_img):
        H, W = ori_img.shape[:2]
        cxcywh = xyxy_to_cxcywh(np.asarray(bbox_xyxy, dtype=np.float64))
        crops = []
        for b in cxcywh:                                         # :119-129, features for ALL boxes (Q5)
            x1, y1, x2, y2 = crop_corners(b, W, H)
            crops.append(ori_img[y1:y2, x1:x2])
        feats = self.embed(crops) if crops else np.zeros((0, 512), np.float32)
        tlwh = cxcywh.copy()                                     # :68-75
        tlwh[:, 0] = cxcywh[:, 0] - cxcywh[:, 2] / 2.0
        tlwh[:, 1] = cxcywh[:, 1] - cxcywh[:, 3] / 2.0
        dets = [{"tlwh": tlwh[i].astype(np.float64), "conf": float(c),
                 "feature": np.asarray(feats[i], dtype=np.float32)}
                for i, c in enumerate(confidences) if c > self.min_conf]              # :31
        keep = dsort_nms(np.array([d["tlwh"] for d in dets]), self.nms_ov,
                         np.array([d["conf"] for d in dets]))                         # :34-37
        dets = [dets[i] for i in keep]
        self.trk.predict()
        self.trk.update(dets)
        rows = []
        for t in self.trk.tracks:                                # :46-58 (Q7)
            if t.state != CONFIRMED or t.tsu > 1:
                continue
            x, y, w, h = mean_to_tlwh(t.mean)
            rows.append([max(int(x), 0), max(int(y), 0), min(int(x + w), W - 1), min(int(y + h), H - 1),
                         t.tid, -1, int(t.confs[-1]) if t.confs else -1])
        return np.array(rows, dtype=np.int64) if rows else []


class VideoTrackerOracle:
    """modules/track.py:9-70: one DeepSORT per class (Q2), classes without detections are not stepped (Q1)."""

    def __init__(self, num_classes, tracking_config, embed):
        c = tracking_config
        self.num_classes = num_classes
        self.ds = [DeepSortOracle(embed, max_dist=c["MAX_DIST"], min_confidence=c["MIN_CONFIDENCE"],
                                  nms_max_overlap=c["NMS_MAX_OVERLAP"], max_iou_distance=c["MAX_IOU_DISTANCE"],
                                  max_age=c["MAX_AGE"], n_init=c["N_INIT"], nn_budget=c["NN_BUDGET"])
                   for _ in range(num_classes)]

    def run(self, image, boxes, labels, scores):
        xyxy = np.array(boxes, dtype=np.float64, copy=True)
        xyxy[:, 2] += xyxy[:, 0]
        xyxy[:, 3] += xyxy[:, 1]
        out = {"tracks": [], "boxes": [], "labels": [], "scores": []}
        for c in range(self.num_classes):
            m = labels == c
            if m.sum() > 0:
                for row in self.ds[c].update(xyxy[m], scores[m], image):
                    out["tracks"].append(row[4])
                    out["boxes"].append(row[:4])
                    out["labels"].append(c)
        out["boxes"] = np.array(out["boxes"])
        return out
