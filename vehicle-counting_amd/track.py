"""DeepSort / VideoTracker / VideoCounting: drop-ins for /root/reference/networks/deepsort/deep_sort.py:15-59 and
/root/reference/modules/track.py:9-137 on top of the HIP engine (same constructor arguments, same return shapes)."""
from __future__ import annotations

import numpy as np

from .counting import find_best_match_direction, load_zone_anno, save_tracking_to_csv, zone_mask


class DeepSort:
    """deep_sort.py:15-59.  `model_path` is accepted for signature compatibility; the ReID weights live in the engine."""

    def __init__(self, model_path, max_dist=0.2, min_confidence=0.3, nms_max_overlap=1.0, max_iou_distance=0.7, max_age=70,
                 n_init=3, nn_budget=100, use_cuda=True, engine=None):
        if engine is None:
            raise ValueError("DeepSort needs the HIP engine (there is no CPU path)")
        self.engine = engine
        self.min_confidence, self.nms_max_overlap = min_confidence, nms_max_overlap
        self.tracker_id = engine.tracker_create(max_dist=max_dist, min_confidence=min_confidence, nms_max_overlap=nms_max_overlap,
                                                max_iou_distance=max_iou_distance, max_age=max_age, n_init=n_init, nn_budget=nn_budget)

    def close(self):
        """Gives the tracker (its tracks, galleries and its slot) back to the engine; the handle is dead afterwards (the engine refuses it
        even once the slot belongs to a later tracker).  The reference simply drops its DeepSort objects when CountingPipeline builds the
        next video's VideoTracker (modules/__init__.py:32-36); the drop-in's pipeline closes its stages explicitly (pipeline._video)."""
        tid, self.tracker_id = self.tracker_id, None
        if tid is not None and getattr(self.engine, "_h", None):
            self.engine.tracker_destroy(tid)

    def __del__(self):
        # A finaliser runs at arbitrary points (also inside another engine call): it must not wait for batches in flight the way
        # vc_tracker_destroy does.  It only queues the handle; the engine gives queued handles back the next time a tracker is created.
        try:
            tid, self.tracker_id = self.tracker_id, None
            if tid is not None and getattr(self.engine, "_h", None):
                self.engine.tracker_destroy_later(tid)
        except Exception:          # interpreter shutdown, engine already destroyed
            pass

    def update(self, bbox_xyxy, confidences, ori_img):
        rows = self.engine.deepsort_update(self.tracker_id, bbox_xyxy, confidences, ori_img)
        return rows if len(rows) > 0 else []          # deep_sort.py:57-59

    def update_with_features(self, bbox_xyxy, confidences, features, height, width):
        """DeepSort.update with the embeddings supplied by the caller (frame-sharded front end, SURVEY.md 8f.1: another rank
        ran the detector and the ReID net).  Same steps as deep_sort.py:25-59 minus `_get_features`: confidence filter (:31),
        tlwh (:68-87), DeepSORT NMS in pick order (:37-41), tracker predict/update (:44-45), rows of the confirmed tracks seen
        within one frame as clamped ints (:47-56, 97-108)."""
        from . import engine as E
        b = np.asarray(bbox_xyxy, np.float64).reshape(-1, 4)
        c = np.asarray(confidences, np.float64).reshape(-1)
        f = np.asarray(features, np.float32).reshape(-1, 512)
        keep = c > self.min_confidence
        b, c, f = b[keep], c[keep], f[keep]
        w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
        cx, cy = b[:, 0] + w / 2, b[:, 1] + h / 2
        tlwh = np.stack([cx - w / 2., cy - h / 2., w, h], 1) if len(b) else np.zeros((0, 4))
        pick = E.dsort_nms(tlwh, c, self.nms_max_overlap) if len(b) else []
        self.engine.tracker_step(self.tracker_id, tlwh[pick], c[pick], f[pick])
        st = self.engine.tracker_state(self.tracker_id, with_cov=False)
        rows = []
        for i in range(len(st["ids"])):
            if st["state"][i] != 2 or st["tsu"][i] > 1:          # confirmed, time_since_update <= 1
                continue
            m = st["mean"][i]
            tw, th = m[2] * m[3], m[3]
            x, y = m[0] - tw / 2, m[1] - th / 2
            rows.append([max(int(x), 0), max(int(y), 0), min(int(x + tw), width - 1), min(int(y + th), height - 1), int(st["ids"][i])])
        return np.asarray(rows, np.int64).reshape(-1, 5)


class VideoTracker:
    """modules/track.py:9-70: one DeepSORT per class; `run` steps only the classes that have boxes in the frame."""

    def __init__(self, num_classes, cam_config, video_info, deepsort_chepoint=None, engine=None):
        cfg = cam_config["tracking_config"]
        self.num_classes = num_classes
        self.video_info = video_info
        self.num_frames = video_info.get("num_frames") if video_info else None
        self.engine = engine
        self.deepsort = [DeepSort(deepsort_chepoint, max_dist=cfg["MAX_DIST"], min_confidence=cfg["MIN_CONFIDENCE"],
                                  nms_max_overlap=cfg["NMS_MAX_OVERLAP"], max_iou_distance=cfg["MAX_IOU_DISTANCE"],
                                  max_age=cfg["MAX_AGE"], n_init=cfg["N_INIT"], nn_budget=cfg["NN_BUDGET"], use_cuda=1,
                                  engine=engine) for _ in range(num_classes)]
        self.tracker_ids = [d.tracker_id for d in self.deepsort]

    def close(self):
        for d in self.deepsort:
            d.close()
        self.tracker_ids = None            # a closed VideoTracker drives nothing (its handles are dead in the engine as well)

    def run(self, image, boxes, labels, scores):
        if self.tracker_ids is None:
            raise RuntimeError("VideoTracker.run after close()")
        rows = self.engine.videotracker_run(self.tracker_ids, image, boxes, labels, scores)
        return {"tracks": [int(r[4]) for r in rows], "boxes": rows[:, :4].copy() if len(rows) else np.array([]),
                "labels": [int(r[5]) for r in rows], "scores": []}

    def rows_to_result(self, rows):
        return {"tracks": [int(r[4]) for r in rows], "boxes": rows[:, :4].copy() if len(rows) else np.array([]),
                "labels": [int(r[5]) for r in rows], "scores": []}


class VideoCounting:
    """modules/track.py:73-137 (zone filter -> per (label, track) lists -> direction -> CSV)."""

    def __init__(self, class_names, zone_path, minimum_length=4):
        self.class_names = class_names
        self.num_classes = len(class_names)
        self.track_dict = [{} for _ in range(self.num_classes)]
        self.minimum_length = minimum_length            # stored and never used by the reference either (Q15)
        self.zone_path = zone_path
        self.polygons, self.directions = load_zone_anno(zone_path)

    def run(self, frames, tracks, labels, boxes, output_path=None, finalize=True):
        """The reference calls this once per video with the whole lists.  finalize=False only appends the rows (zone filter +
        per-track lists) so that a stream can be fed batch by batch while the GPU works on the next one; the last call
        (possibly with empty lists) assigns the directions -- the resulting track_dict is the same as for one call."""
        inside = zone_mask(self.polygons, boxes) if len(boxes) else []
        for keep, frame_id, track_id, label_id, box in zip(inside, frames, tracks, labels, boxes):
            if keep:                                     # check_bbox_intersect_polygon (modules/track.py:104)
                rec = self.track_dict[label_id].setdefault(track_id, {"boxes": [], "frames": [], "color": ""})
                rec["boxes"].append(box)
                rec["frames"].append(frame_id)
        if not finalize:
            return self.track_dict
        for label_id in range(self.num_classes):
            for rec in self.track_dict[label_id].values():
                fb, lb = rec["boxes"][0], rec["boxes"][-1]
                first = ((fb[2] + fb[0]) / 2, (fb[3] + fb[1]) / 2)
                last = ((lb[2] + lb[0]) / 2, (lb[3] + lb[1]) / 2)
                rec["direction"] = find_best_match_direction((first, last), self.directions)
        if output_path is not None:
            save_tracking_to_csv(self.track_dict, output_path)
        return self.track_dict
