"""CountingPipeline: the per-video driver of /root/reference/modules/__init__.py:7-100 behind the same stage objects.

Video decode/encode (cv2.VideoCapture / VideoWriter, modules/datasets.py) is out of scope: frames come from a
`FrameSource` over an in-memory BGR array (or any iterable of such batches) that honours the reference's input contract
-- RGB frame for the detector, BGR original for the tracker, 1-based frame ids (modules/datasets.py:47-76).

Drivers with identical results:
  run()                the reference's loop, one frame at a time through ImageDetect.run / VideoTracker.run (host frames);
  run_pipelined()      the same loop over the same loader (any iterable of the reference's batch dicts, batch_size 1 included), but the
                       stage calls are asynchronous: the detector of batch n+1 runs behind the ReID net and the tracker of batch n;
  run_stream()         frames resident in HBM, B frames per `vc_stream_run` call (detect batched, trackers stepped in order);
  run_frame_sharded()  ONE stream on several GPUs: detect + ReID shard by frame chunk over the ranks, the
                       per-detection payloads are gathered in frame order, rank 0 runs the sequential tracker.
"""
from __future__ import annotations

import contextlib
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

    @contextlib.contextmanager
    def _video(self, *trackers):
        """One video's stage objects: whatever happens inside, the engine-side trackers are given back (the reference drops the video's
        VideoTracker, modules/__init__.py:32-36), and after an exception the submissions still in flight are abandoned so that the
        next video starts on an idle engine."""
        try:
            yield
        except BaseException:
            try:
                self.engine.stream_reset()
            except Exception:
                pass
            raise
        finally:
            for t in trackers:
                if t is not None:
                    t.close()

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
        with self._video(tracker):
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

    def run_pipelined(self, source, cam_name, zone_path):
        """modules/__init__.py:28-100 for one video with the per-frame body software-pipelined over the engine's streams.

        Same input as run(): an iterable of the reference's loader batches ({'imgs', 'ori_imgs', 'frames'}, modules/datasets.py:47-76;
        batch_size = 1 at :93), host frames.  Same results (rows in the same order, Q1 skip included): what changes is WHEN a batch's
        stage work is issued -- batch n+2 is copied to the device and batch n+1's detector pass is enqueued before batch n's detections are
        marshalled, embedded and tracked (`vc_stream_stage_host` / `vc_stream_submit` / `vc_stream_run_async` / `vc_stream_collect`), so
        the ~60 dependent launches of a batch-1 detector pass overlap with the ReID net and the tracker walk of the frame before instead
        of following them.  ImageDetect.run / VideoTracker.run stay blocking calls for callers that use the stage objects directly;
        this driver is the drop-in for CountingPipeline.run itself."""
        import collections
        tracker, counter = self._stages(cam_name, source.video_info, zone_path)
        obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
        it = (b for b in source if b is not None)
        staged = collections.deque()        # (frame ids, host array, device address, b, h, w): copied, detector not yet enqueued
        submitted = collections.deque()     # detector enqueued, not yet handed to ReID + tracker
        running = collections.deque()       # tracker enqueued, rows not yet collected

        def stage():
            batch = next(it, None)
            if batch is None:
                return
            arr = np.ascontiguousarray(np.stack([np.asarray(f) for f in batch["ori_imgs"]]), dtype=np.uint8)   # (b, h, w, 3) BGR as the loader delivers it
            b, h, w, _ = arr.shape
            ptr = self.engine.stream_stage_host(arr.ctypes.data, b, h, w)
            staged.append((np.asarray(batch["frames"], dtype=np.int64), arr, ptr, b, h, w))

        def submit():
            if staged:
                ids, arr, ptr, b, h, w = staged.popleft()
                self.engine.stream_submit(ptr, b, h, w)
                submitted.append((ids, arr, ptr, b, h, w))

        def collect():
            ids, _arr = running.popleft()
            rows, fidx = self.engine.stream_collect()[:2]
            obj["frames"].extend(ids[fidx].tolist())
            obj["tracks"].extend(rows[:, 4].tolist())
            obj["labels"].extend(rows[:, 5].tolist())
            obj["boxes"].extend(list(rows[:, :4].copy()))

        with self._video(tracker):
            stage(); stage()                                 # staging order = batch order (four host slots, round-robin)
            submit()
            while submitted:
                stage()                                      # copy batch n+2 under the detector of batch n+1
                submit()                                     # detect batch n+1 while batch n is embedded and tracked
                ids, arr, ptr, b, h, w = submitted.popleft()
                self.engine.stream_run_async(tracker.tracker_ids, ptr, b, h, w)
                running.append((ids, arr))
                if len(running) > 1:
                    collect()
            while running:
                collect()
        return self._finish(counter, obj, cam_name)

    def run_stream(self, source, cam_name, zone_path, batch=16, asynchronous=False, host_frames=False):
        """asynchronous=True: the tracker kernel of batch n runs on the engine's tracker stream while batch n+1 is submitted and
        embedded (`stream_run_async` / `stream_collect`); rows are identical, they arrive one batch later.
        host_frames=True: the frames stay in (pinned) host memory, as the reference's loader delivers them, and cross PCIe batch by
        batch -- batch n+2 is staged (`stream_stage_host`) while the detector works on batch n+1, so a video of any length needs four
        batches of device memory; otherwise the whole clip is uploaded once."""
        import torch
        tracker, counter = self._stages(cam_name, source.video_info, zone_path)
        obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
        frames = source.frames
        t, h, w, _ = frames.shape
        starts = list(range(0, t, batch))
        size = lambda n: min(batch, t - starts[n])
        ptr = {}
        if host_frames:
            host = torch.from_numpy(frames).pin_memory()
            stage = lambda n: ptr.__setitem__(n, self.engine.stream_stage_host(host[starts[n]:starts[n] + size(n)].data_ptr(), size(n), h, w))
        else:
            dev = torch.from_numpy(frames).to(f"cuda:{self.engine.cfg.device}")      # tensor container only
            stage = lambda n: ptr.__setitem__(n, dev[starts[n]:starts[n] + size(n)].data_ptr())

        def record(f0, rows, fidx):
            obj["frames"].extend((f0 + 1 + fidx).tolist())
            obj["tracks"].extend(rows[:, 4].tolist())
            obj["labels"].extend(rows[:, 5].tolist())
            obj["boxes"].extend(list(rows[:, :4].copy()))

        with self._video(tracker):
            for n in range(min(2, len(starts))):            # staging order = batch order (the engine hands its four host slots out round-robin)
                stage(n)
            if starts:
                self.engine.stream_submit(ptr[0], size(0), h, w)
            for n, f0 in enumerate(starts):
                b = size(n)
                if n + 2 < len(starts):                     # copy batch n+2 (host frames) under the detector of batch n+1
                    stage(n + 2)
                if n + 1 < len(starts):                     # detect the next batch while this one is tracked
                    self.engine.stream_submit(ptr[n + 1], size(n + 1), h, w)
                if asynchronous:
                    self.engine.stream_run_async(tracker.tracker_ids, ptr[n], b, h, w)
                    if n > 0:
                        record(starts[n - 1], *self.engine.stream_collect()[:2])
                else:
                    record(f0, *self.engine.stream_run_packed(tracker.tracker_ids, ptr[n], b, h, w)[:2])
                ptr.pop(n - 1, None)
            if asynchronous and starts:
                record(starts[-1], *self.engine.stream_collect()[:2])
        return self._finish(counter, obj, cam_name)

    def run_streams(self, sources, cam_names, zone_paths, batch=16):
        """S videos at once on ONE engine (the reference runs them one after another, each with a new VideoTracker,
        modules/__init__.py:28-36): the frames of the cameras are interleaved round-robin into batches of `batch` frames, the
        detector and the ReID net see one batch, every camera's frames are stepped on that camera's own trackers
        (`vc_stream_run_async_multi`).  Per-camera results are identical to S separate `run_stream` calls; per-camera latency is
        batch / S frames.  All sources must share one frame size.  Returns [(rows, counts)] in camera order."""
        import torch
        S = len(sources)
        shapes = {s.frames.shape[1:] for s in sources}
        assert len(shapes) == 1, "run_streams: all cameras must deliver frames of one size"
        h, w, _ = next(iter(shapes))
        stages = [self._stages(n, s.video_info, z) for n, s, z in zip(cam_names, sources, zone_paths)]
        tids = np.array([st[0].tracker_ids for st in stages], np.int32)                 # [S][num_classes]
        order = []                                                                       # (camera, frame) round-robin, exhausted cameras drop out
        for t in range(max(len(s) for s in sources)):
            order.extend((c, t) for c in range(S) if t < len(sources[c]))
        cams = np.array([c for c, _ in order], np.int32)
        fidx = np.array([t for _, t in order], np.int64)
        dev = torch.from_numpy(np.stack([sources[c].frames[t] for c, t in order])).to(f"cuda:{self.engine.cfg.device}")
        objs = [{"frames": [], "tracks": [], "labels": [], "boxes": []} for _ in range(S)]
        starts = list(range(0, len(order), batch))

        def record(f0, rows, fb):
            g = f0 + fb                                                                  # global position of each row's frame
            for c in range(S):
                sel = cams[g] == c
                o = objs[c]
                o["frames"].extend((fidx[g[sel]] + 1).tolist())
                o["tracks"].extend(rows[sel, 4].tolist())
                o["labels"].extend(rows[sel, 5].tolist())
                o["boxes"].extend(list(rows[sel, :4].copy()))

        def span(n):
            f0 = starts[n]
            return f0, min(batch, len(order) - f0)

        with self._video(*[st[0] for st in stages]):
            f0, b = span(0)
            self.engine.stream_submit(dev[f0:f0 + b].data_ptr(), b, h, w)
            for n in range(len(starts)):
                f0, b = span(n)
                if n + 1 < len(starts):
                    g0, gb = span(n + 1)
                    self.engine.stream_submit(dev[g0:g0 + gb].data_ptr(), gb, h, w)
                self.engine.stream_run_async_multi(tids, cams[f0:f0 + b], dev[f0:f0 + b].data_ptr(), b, h, w)
                if n > 0:
                    record(starts[n - 1], *self.engine.stream_collect()[:2])
            if starts:
                record(starts[-1], *self.engine.stream_collect()[:2])
        return [self._finish(st[1], o, n) for st, o, n in zip(stages, objs, cam_names)]

    def run_frame_sharded(self, source, cam_name, zone_path, chunk=8, device=None):
        """ONE camera stream on several GPUs, on the product's own batched path: the stateless front end shards by
        frame chunk over the ranks (chunk j on rank j % world): every rank submits its chunk to the batched detector
        (`vc_stream_submit`) and takes the marshalled boxes + device-resident embeddings of the whole chunk back
        (`vc_stream_embed`: one detector pass and one ReID pass per chunk); one variable-length RCCL all-gather per round behind the
        C ABI (`vc_allgather_rows`: rows over PCIe-free xGMI, embeddings device to device) brings every round's rows to all ranks in
        frame order; rank 0 steps the sequential tracker on the gathered round with ONE tracker kernel launch
        (`vc_videotracker_run_features`) and runs the counting -- tracker state never shards below a camera.  The ordering contract
        is the reference's (modules/__init__.py:54-84): frames reach the tracker in ascending order, empty frames are skipped (Q1).
        Returns (rows, counts) on rank 0, (None, None) elsewhere."""
        import torch
        import torch.distributed as dist

        from . import parallel
        rank, world = parallel.ensure_comm(self.engine)
        frames = source.frames
        t, h, w, _ = frames.shape
        mine = parallel.shard_frames(t, rank, world, chunk)
        n_rounds = (len(range(0, t, chunk)) + world - 1) // world
        tracker, counter = self._stages(cam_name, source.video_info, zone_path) if rank == 0 else (None, None)
        obj = {"frames": [], "tracks": [], "labels": [], "boxes": []}
        devs = [torch.from_numpy(frames[a:b]).to(f"cuda:{self.engine.cfg.device}") for a, b in mine]     # only this rank's chunks
        with self._video(tracker):
            if mine:
                self.engine.stream_submit(devs[0].data_ptr(), len(devs[0]), h, w)
            for r in range(n_rounds):
                if r + 1 < len(mine):                                                # detector of the next chunk runs behind this round's ReID / gather
                    self.engine.stream_submit(devs[r + 1].data_ptr(), len(devs[r + 1]), h, w)
                if r < len(mine):
                    rows7, feat = self.engine.stream_embed(devs[r].data_ptr(), len(devs[r]), h, w)
                    rows7[:, 0] += mine[r][0] + 1                                     # 1-based global frame id (modules/datasets.py:61)
                else:
                    rows7, feat = np.zeros((0, 7)), 0
                all_rows, all_feat, _ = self.engine.allgather_rows(rows7, feat, world)   # rank-major = frame order within a round
                if rank != 0:
                    continue
                for fid, rows in self.engine.videotracker_run_features(tracker.tracker_ids, all_rows, all_feat, h, w):
                    obj["frames"].extend([fid] * len(rows))
                    obj["tracks"].extend(rows[:, 4].tolist())
                    obj["labels"].extend(rows[:, 5].tolist())
                    obj["boxes"].extend(list(rows[:, :4].copy()))
        if rank != 0:
            return None, None
        return self._finish(counter, obj, cam_name)
