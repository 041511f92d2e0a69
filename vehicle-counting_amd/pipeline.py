"""CountingPipeline: the per-video driver of /root/reference/modules/__init__.py:7-100 behind the same stage objects.

Video decode/encode (cv2.VideoCapture / VideoWriter, modules/datasets.py) is out of scope: frames come from a
`FrameSource` over an in-memory BGR array (or any iterable of such batches) that honours the reference's input contract
-- RGB frame for the detector, BGR original for the tracker, 1-based frame ids (modules/datasets.py:47-76).

Drivers with identical results:
  run()                the reference's loop, one frame at a time through ImageDetect.run / VideoTracker.run (host frames);
  run_stream()         frames resident in HBM, B frames per `vc_stream_run` call (detect batched, trackers stepped in order);
  run_frame_sharded()  ONE stream on several GPUs: detect + ReID shard by frame chunk over the ranks, the
                       per-detection payloads are gathered in frame order, rank 0 runs the sequential tracker.
"""
from __future__ import annotations

import os

import numpy as np

from .counting import count_directions, csv_records
from .detect import ImageDetect
from .track import VideoCounting, VideoTracker


class FrameSource:
    """In-memory video: (T, H, W, 3) uint8 BGR, like cv2.VideoCapture.read() delivers frames."""

    def __init__(self, frames_bgr, name="cam_04.mp4", fps=10):
        self.frames = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
        t, h, w, _ = self.frames.shape
        self.video_info = {"name": name, "width": w, "height": h, "fps": fps, "num_frames": t}

    def __len__(self):
        return len(self.frames)

    def __iter__(self):
        for i, f in enumerate(self.frames):
            yield {"imgs": [f[:, :, ::-1]], "ori_imgs": [f], "frames": [i + 1]}       # BGR->RGB view, 1-based id


class CountingPipeline:
    def __init__(self, args, config, cam_config, engine=None, class_names=None, synthetic=False):
        # the ReID checkpoint of the reference's cam_configs.yaml (`checkpoint: .../ckpt.t7`, handed to every DeepSort at
        # modules/__init__.py:36) becomes the engine's ReID parameters -- one engine owns both networks
        ck = cam_config.get("checkpoint") if isinstance(cam_config, dict) else getattr(cam_config, "checkpoint", None)
        self.detector = ImageDetect(args, config, engine=engine, class_names=class_names, synthetic=synthetic, reid_checkpoint=ck)
        self.engine = self.detector.engine
        self.class_names = self.detector.class_names
        self.saved_path = getattr(args, "output_path", None)
        self.cam_config = cam_config
        self.config = config

    def _stages(self, cam_name, video_info, zone_path):
        cam = self.cam_config["cam"][cam_name] if isinstance(self.cam_config, dict) else self.cam_config.cam[cam_name]
        tracker = VideoTracker(len(self.class_names), cam, video_info, engine=self.engine)
        counter = VideoCounting(class_names=self.class_names, zone_path=zone_path)
        return tracker, counter

    def _finish(self, counter, obj, cam_name):
        out = os.path.join(self.saved_path, cam_name + ".csv") if self.saved_path else None
        td = counter.run(frames=obj["frames"], tracks=obj["tracks"], labels=obj["labels"], boxes=obj["boxes"], output_path=out)
        rows = csv_records(td)
        counts = count_directions(rows, list(counter.directions.keys()), len(self.class_names))
        return rows, counts

    def run(self, source, cam_name, zone_path):
        """modules/__init__.py:28-100 for one video."""
        tracker, counter = self._stages(cam_name, source.video_info, zone_path)
        obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
        for batch in source:
            if batch is None:
                continue
            preds = self.detector.run(batch)
            for i in range(len(batch["ori_imgs"])):
                boxes, labels, scores = preds["boxes"][i], preds["labels"][i], preds["scores"][i]
                if len(boxes) == 0:                         # :68-69 (Q1)
                    continue
                res = tracker.run(batch["ori_imgs"][i], boxes, labels, scores)
                for j in range(len(res["boxes"])):
                    obj["frames"].append(batch["frames"][i])
                    obj["tracks"].append(res["tracks"][j])
                    obj["labels"].append(res["labels"][j])
                    obj["boxes"].append(res["boxes"][j])
        return self._finish(counter, obj, cam_name)

    def run_stream(self, source, cam_name, zone_path, batch=16, asynchronous=False):
        """asynchronous=True: the tracker kernel of batch n runs on the engine's tracker stream while batch n+1 is submitted and
        embedded (`stream_run_async` / `stream_collect`); rows are identical, they arrive one batch later."""
        import torch
        tracker, counter = self._stages(cam_name, source.video_info, zone_path)
        obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
        frames = source.frames
        t, h, w, _ = frames.shape
        dev = torch.from_numpy(frames).to(f"cuda:{self.engine.cfg.device}")      # tensor container only
        starts = list(range(0, t, batch))

        def record(f0, rows, fidx):
            obj["frames"].extend((f0 + 1 + fidx).tolist())
            obj["tracks"].extend(rows[:, 4].tolist())
            obj["labels"].extend(rows[:, 5].tolist())
            obj["boxes"].extend(list(rows[:, :4].copy()))

        self.engine.stream_submit(dev[0:min(batch, t)].data_ptr(), min(batch, t), h, w)
        for n, f0 in enumerate(starts):
            b = min(batch, t - f0)
            if n + 1 < len(starts):                     # detect the next batch while this one is tracked
                g0 = starts[n + 1]
                self.engine.stream_submit(dev[g0:g0 + min(batch, t - g0)].data_ptr(), min(batch, t - g0), h, w)
            if asynchronous:
                self.engine.stream_run_async(tracker.tracker_ids, dev[f0:f0 + b].data_ptr(), b, h, w)
                if n > 0:
                    record(starts[n - 1], *self.engine.stream_collect()[:2])
            else:
                record(f0, *self.engine.stream_run_packed(tracker.tracker_ids, dev[f0:f0 + b].data_ptr(), b, h, w)[:2])
        if asynchronous and starts:
            record(starts[-1], *self.engine.stream_collect()[:2])
        return self._finish(counter, obj, cam_name)

    def run_frame_sharded(self, source, cam_name, zone_path, chunk=8, device=None):
        """One camera stream, the stateless front end sharded by frame chunk over the ranks of torch.distributed (chunk j on rank
        j % world), one ordered gather of [frame, x1, y1, x2, y2, conf, label, feature(512)] rows per round, the tracker and the
        counting on rank 0 (tracker state never shards below a camera).  Returns (rows, counts) on rank 0, (None, None) elsewhere.
        The detections are marshalled exactly as in run(): ImageDetect.run -> xywh -> xyxy (modules/track.py:39-41)."""
        import torch.distributed as dist

        from . import parallel
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        frames = source.frames
        t, h, w, _ = frames.shape
        mine = parallel.shard_frames(t, rank, world, chunk)
        n_rounds = (len(range(0, t, chunk)) + world - 1) // world
        tracker, counter = self._stages(cam_name, source.video_info, zone_path) if rank == 0 else (None, None)
        obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
        for r in range(n_rounds):
            local = []
            if r < len(mine):
                for f in range(*mine[r]):
                    bgr = frames[f]
                    preds = self.detector.run({"imgs": [bgr[:, :, ::-1]], "ori_imgs": [bgr], "frames": [f + 1]})
                    boxes, labels, scores = preds["boxes"][0], preds["labels"][0], preds["scores"][0]
                    if len(boxes) == 0:                                              # modules/__init__.py:68-69 (Q1)
                        continue
                    xyxy = np.asarray(boxes, np.float64).copy()
                    xyxy[:, 2] += xyxy[:, 0]; xyxy[:, 3] += xyxy[:, 1]               # modules/track.py:39-41
                    bw, bh = xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]        # deep_sort.py:78-87
                    cxcywh = np.stack([xyxy[:, 0] + bw / 2, xyxy[:, 1] + bh / 2, bw, bh], 1)
                    feat = self.engine.embed(bgr, cxcywh)
                    local.append(np.concatenate([np.full((len(xyxy), 1), f + 1, np.float64), xyxy, np.asarray(scores, np.float64)[:, None],
                                                 np.asarray(labels, np.float64)[:, None], feat.astype(np.float64)], 1))
            rows = parallel.gather_rows(np.concatenate(local, 0) if local else np.zeros((0, 7 + 512)), device=device)
            if rank != 0:
                continue
            for fid in np.unique(rows[:, 0]) if len(rows) else []:                   # ascending frame ids
                fr = rows[rows[:, 0] == fid]
                lab = fr[:, 6].astype(np.int64)
                for c in range(len(self.class_names)):                               # modules/track.py:50-59
                    sel = fr[lab == c]
                    if len(sel) == 0:
                        continue
                    out = tracker.deepsort[c].update_with_features(sel[:, 1:5], sel[:, 5], sel[:, 7:].astype(np.float32), h, w)
                    for row in out:
                        obj["frames"].append(int(fid)); obj["tracks"].append(int(row[4])); obj["labels"].append(c); obj["boxes"].append(row[:4].copy())
        if rank != 0:
            return None, None
        return self._finish(counter, obj, cam_name)
