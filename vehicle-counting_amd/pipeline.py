"""CountingPipeline: the per-video driver of /root/reference/modules/__init__.py:7-100 behind the same stage objects.

Video decode/encode (cv2.VideoCapture / VideoWriter, modules/datasets.py) is out of scope: frames come from a
`FrameSource` over an in-memory BGR array (or any iterable of such batches) that honours the reference's input contract
-- RGB frame for the detector, BGR original for the tracker, 1-based frame ids (modules/datasets.py:47-76).

Two drivers with identical results:
  run()         the reference's loop, one frame at a time through ImageDetect.run / VideoTracker.run (host frames);
  run_stream()  frames resident in HBM, B frames per `vc_stream_run` call (detect batched, trackers stepped in order).
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
    def __init__(self, args, config, cam_config, engine=None, class_names=None):
        self.detector = ImageDetect(args, config, engine=engine, class_names=class_names)
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
        """asynchronous=True: the tracker loop of batch n runs on the engine's worker thread while batch n+1 is submitted and
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
