"""Engine: owns one `vc_engine` (one GPU) and exposes its entry points with numpy in / numpy out."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .weights import fold_reid

YOLO_VARIANT_ID = {"yolov5s": 0, "yolov5m": 1, "yolov5l": 2}


class Engine:
    """One per process/GPU.  `yolo_sd`: {name+'.weight'/'.bias': ndarray} (BN folded); `reid_sd`: un-fused ReID state_dict."""

    def __init__(self, yolo_sd=None, reid_sd=None, *, device=0, precision="bf16", model_name="yolov5s", num_classes=80,
                 img_size=640, max_batch=16, max_frame_hw=(720, 1280), conf_thres=0.25, iou_thres=0.45, max_det=300,
                 max_candidates=4096, max_crops=1024, max_tracks=4096, nn_budget_cap=100, max_trackers=256):
        lib = L.lib()
        cfg = L.EngineConfig()
        L.check(lib.vc_engine_config_default(C.byref(cfg)))
        cfg.device = device
        cfg.precision = L.PREC_ID[precision]
        cfg.yolo_variant = YOLO_VARIANT_ID[model_name]
        cfg.num_classes, cfg.img_size, cfg.max_batch = num_classes, img_size, max_batch
        cfg.max_frame_h, cfg.max_frame_w = max_frame_hw
        cfg.conf_thres, cfg.iou_thres, cfg.max_det, cfg.max_candidates = conf_thres, iou_thres, max_det, max_candidates
        cfg.max_crops, cfg.max_tracks, cfg.nn_budget_cap = max_crops, max_tracks, nn_budget_cap
        cfg.max_trackers = max_trackers
        cfg.with_detector = 1 if yolo_sd is not None else 0
        cfg.with_reid = 1 if reid_sd is not None else 0
        self.cfg = cfg
        self.precision = precision
        self._h = C.c_void_p()
        self._deferred_destroy = []          # tracker handles queued by finalisers (tracker_destroy_later)
        L.check(lib.vc_engine_create(C.byref(cfg), C.byref(self._h)))
        if yolo_sd is not None:
            self._upload(L.NET_YOLO, lambda n: (yolo_sd[n + ".weight"], yolo_sd[n + ".bias"]))
            if "model.24.anchors_px" in yolo_sd:             # checkpoint.load_yolov5_checkpoint: Detect anchors x stride (pixels)
                a = L.f32(yolo_sd["model.24.anchors_px"]).reshape(18)
                L.check(lib.vc_engine_set_anchors(self._h, L.ptr(a, C.c_float)))
        if reid_sd is not None:
            folded = fold_reid(reid_sd)
            self._upload(L.NET_REID, lambda n: folded[n])
        L.check(lib.vc_engine_finalize(self._h))

    # ---------------------------------------------------------------- lifecycle
    def _upload(self, net, getter):
        lib = L.lib()
        n = C.c_int()
        L.check(lib.vc_engine_param_count(self._h, net, C.byref(n)))
        for i in range(n.value):
            name = C.create_string_buffer(128)
            dims = (C.c_int * 4)()
            L.check(lib.vc_engine_param_info(self._h, net, i, name, 128, dims))
            w, b = getter(name.value.decode())
            w, b = L.f32(w), L.f32(b)
            if tuple(w.shape) != tuple(dims):
                raise ValueError(f"{name.value.decode()}: expected weight shape {tuple(dims)}, got {w.shape}")
            L.check(lib.vc_engine_set_param(self._h, net, name.value, L.ptr(w, C.c_float), L.ptr(b, C.c_float)))

    def close(self):
        if self._h:
            L.lib().vc_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        L.check(L.lib().vc_engine_sync(self._h))

    def set_option(self, name, value):
        """Kernel-selection switch of the live engine ("c3_fused", "bneck_fused", "bneck_cv3", "front_fused", "sparse_head", "reid_block_fused", "crop_per_pixel", "dot_arena_mb"; diagnostics: "ff_ablate", "c3_ablate")."""
        L.check(L.lib().vc_engine_set_option(self._h, name.encode(), int(value)))

    def stream_reset(self):
        """Abandon everything in flight on the stream path (after an error, or to replay a clip)."""
        L.check(L.lib().vc_stream_reset(self._h))
        self._async_shapes = []

    # ---------------------------------------------------------------- detect (AutoShape forward)
    def detect(self, imgs_rgb):
        """list of HxWx3 uint8 RGB -> list of (n,6) float32 [x1,y1,x2,y2,conf,cls] in source pixels."""
        n = len(imgs_rgb)
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs_rgb]
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        md = self.cfg.max_det
        out = np.zeros((n, md, 6), np.float32)
        cnt = np.zeros(n, np.int32)
        L.check(L.lib().vc_detect(self._h, ptrs, hs, ws, n, L.ptr(out, C.c_float), L.ptr(cnt, C.c_int)))
        return [out[i, :cnt[i]].copy() for i in range(n)]

    def debug_layer(self, layer, batch=1):
        dims = (C.c_int * 4)()
        cap = batch * max(640, int(self.cfg.img_size)) ** 2 * 32     # largest layer: (size/2)^2 x 64..128 channels
        buf = np.zeros(cap, np.float32)
        L.check(L.lib().vc_detect_debug_layer(self._h, layer, L.ptr(buf, C.c_float), cap, dims))
        b, h, w, c = dims
        return buf[: b * h * w * c].reshape(b, h, w, c).copy()

    def debug_pred(self, arm=False):
        if arm:
            L.check(L.lib().vc_detect_debug_pred(self._h, None, 0))
            return None
        nh, nw, nt = C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().vc_detect_debug_shape(self._h, C.byref(nh), C.byref(nw), C.byref(nt)))
        no = self.cfg.num_classes + 5
        buf = np.zeros(self.cfg.max_batch * nt.value * no, np.float32)
        L.check(L.lib().vc_detect_debug_pred(self._h, L.ptr(buf, C.c_float), buf.size))
        return buf.reshape(self.cfg.max_batch, nt.value, no)

    # ---------------------------------------------------------------- embed (Extractor)
    def embed(self, bgr, boxes_cxcywh):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        boxes = L.f64(boxes_cxcywh).reshape(-1, 4)
        out = np.zeros((len(boxes), L.FEAT_DIM), np.float32)
        L.check(L.lib().vc_embed(self._h, L.ptr(bgr, C.c_uint8), bgr.shape[0], bgr.shape[1], L.ptr(boxes, C.c_double),
                                 len(boxes), L.ptr(out, C.c_float)))
        return out

    def embed_input(self, k):
        """The k x 50 x 50 x 3 network input the last embed() built (diagnostics)."""
        buf = np.empty((k, 50, 50, 3), np.float32)
        dims = (C.c_int * 4)()
        L.check(L.lib().vc_embed_debug_input(self._h, k, L.ptr(buf, C.c_float), buf.size, dims))
        return buf

    def embed_tensor(self, x_nchw):
        x = L.f32(x_nchw)
        out = np.zeros((len(x), L.FEAT_DIM), np.float32)
        L.check(L.lib().vc_embed_tensor(self._h, L.ptr(x, C.c_float), len(x), L.ptr(out, C.c_float)))
        return out

    def pretune(self, crop_counts=(32, 64, 128, 256, 512, 1024)):
        """Run the ReID net once per problem-size bucket so the conv autotuner (engine.hip::tuned_cfg) has picked its tile
        configurations before any timed work; the detector's convs are tuned by the first (warm-up) batch."""
        for k in crop_counts:
            if k <= self.cfg.max_crops and self.cfg.with_reid:
                self.embed_tensor(np.zeros((k, 3, 50, 50), np.float32))

    # ---------------------------------------------------------------- tracker
    def tracker_create(self, max_dist=0.2, min_confidence=0.3, nms_max_overlap=1.0, max_iou_distance=0.7, max_age=70,
                       n_init=3, nn_budget=100):
        while self._deferred_destroy:                      # handles queued by DeepSort.__del__
            L.check(L.lib().vc_tracker_destroy(self._h, self._deferred_destroy.pop()))
        p = L.TrackerParams(max_dist, min_confidence, nms_max_overlap, max_iou_distance, max_age, n_init, nn_budget)
        tid = C.c_int()
        L.check(L.lib().vc_tracker_create(self._h, C.byref(p), C.byref(tid)))
        return tid.value

    def tracker_destroy(self, tid):
        if self._h:
            L.check(L.lib().vc_tracker_destroy(self._h, tid))

    def tracker_destroy_later(self, tid):
        """From a finaliser: the handle is given back at the next tracker_create (vc_tracker_destroy waits for the batches in flight)."""
        self._deferred_destroy.append(tid)

    def tracker_reset(self, tid):
        L.check(L.lib().vc_tracker_reset(self._h, tid))

    def tracker_step(self, tid, tlwh, conf, feat):
        tlwh, conf, feat = L.f64(tlwh).reshape(-1, 4), L.f64(conf).reshape(-1), L.f32(feat).reshape(-1, L.FEAT_DIM)
        L.check(L.lib().vc_tracker_step(self._h, tid, L.ptr(tlwh, C.c_double), L.ptr(conf, C.c_double),
                                        L.ptr(feat, C.c_float), len(conf)))

    def tracker_state(self, tid, with_cov=True):
        n = C.c_int()
        L.check(L.lib().vc_tracker_count(self._h, tid, C.byref(n)))
        n = n.value
        ids = np.zeros(n, np.int64)
        st, hits, age, tsu, gal = (np.zeros(n, np.int32) for _ in range(5))
        mean = np.zeros((n, 8))
        cov = np.zeros((n, 8, 8)) if with_cov else None
        L.check(L.lib().vc_tracker_state(self._h, tid, max(n, 1), L.ptr(ids, C.c_int64), L.ptr(st, C.c_int), L.ptr(hits, C.c_int),
                                         L.ptr(age, C.c_int), L.ptr(tsu, C.c_int), L.ptr(mean, C.c_double),
                                         L.ptr(cov, C.c_double), L.ptr(gal, C.c_int)))
        return {"ids": ids, "state": st, "hits": hits, "age": age, "tsu": tsu, "mean": mean, "cov": cov, "gallery": gal}

    def tracker_debug_costs(self, cap=1 << 18):
        """(appearance [T, D], iou [T, D]) cost matrices of the last blocking single-tracker step, as the tracker kernel computed
        them (rows are only meaningful for the tracks that took part: confirmed tracks / IoU candidates)."""
        app, iou = np.zeros(cap), np.zeros(cap)
        t, d = C.c_int(), C.c_int()
        L.check(L.lib().vc_tracker_debug_costs(self._h, cap, L.ptr(app, C.c_double), L.ptr(iou, C.c_double), C.byref(t), C.byref(d)))
        n = t.value * d.value
        return app[:n].reshape(t.value, d.value).copy(), iou[:n].reshape(t.value, d.value).copy()

    def tracker_snapshot(self, tid) -> bytes:
        """Serialised tracker state (parameters, id counter, Kalman state, galleries) for stream migration."""
        n = C.c_size_t()
        L.check(L.lib().vc_tracker_snapshot(self._h, tid, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        L.check(L.lib().vc_tracker_snapshot(self._h, tid, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def tracker_restore(self, tid, blob: bytes):
        """Replace tracker `tid`'s state and parameters with a snapshot (possibly taken on another engine / GPU)."""
        buf = C.create_string_buffer(bytes(blob), len(blob))
        L.check(L.lib().vc_tracker_restore(self._h, tid, buf, len(blob)))

    def deepsort_update(self, tid, bbox_xyxy, confidences, ori_img):
        img = np.ascontiguousarray(ori_img, dtype=np.uint8)
        b, c = L.f64(bbox_xyxy).reshape(-1, 4), L.f64(confidences).reshape(-1)
        cap = self.cfg.max_tracks
        rows = np.zeros((cap, 7), np.int64)
        m = C.c_int()
        L.check(L.lib().vc_deepsort_update(self._h, tid, L.ptr(img, C.c_uint8), img.shape[0], img.shape[1], L.ptr(b, C.c_double),
                                           L.ptr(c, C.c_double), len(c), L.ptr(rows, C.c_int64), cap, C.byref(m)))
        return rows[: m.value].copy()

    def videotracker_run(self, tracker_ids, image, boxes_xywh, labels, scores):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        tr = np.ascontiguousarray(tracker_ids, dtype=np.int32)
        b, s = L.f64(boxes_xywh).reshape(-1, 4), L.f64(scores).reshape(-1)
        lab = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
        cap = self.cfg.max_tracks
        rows = np.zeros((cap, 6), np.int64)
        m = C.c_int()
        L.check(L.lib().vc_videotracker_run(self._h, L.ptr(tr, C.c_int), len(tr), L.ptr(img, C.c_uint8), img.shape[0], img.shape[1],
                                            L.ptr(b, C.c_double), L.ptr(lab, C.c_int64), L.ptr(s, C.c_double), len(s),
                                            L.ptr(rows, C.c_int64), cap, C.byref(m)))
        return rows[: m.value].copy()

    # ---------------------------------------------------------------- fused stream path
    def _stream_call(self, tracker_ids, frames_dev_ptr, b, h, w, cap_rows):
        key = (b, cap_rows)
        bufs = self._stream_bufs.get(key) if hasattr(self, "_stream_bufs") else None
        if bufs is None:                                  # output buffers are reused across calls (1.5 MB at b = 64)
            if not hasattr(self, "_stream_bufs"):
                self._stream_bufs = {}
            bufs = self._stream_bufs[key] = (np.empty((b, cap_rows, 6), np.int64), np.zeros(b, np.int32), np.zeros(b, np.int32))
        rows, m, nd = bufs
        tr = np.ascontiguousarray(tracker_ids, dtype=np.int32)
        L.check(L.lib().vc_stream_run(self._h, L.ptr(tr, C.c_int), len(tr), C.c_void_p(frames_dev_ptr), b, h, w,
                                      L.ptr(rows, C.c_int64), cap_rows, L.ptr(m, C.c_int), L.ptr(nd, C.c_int)))
        return rows, m, nd

    def stream_run(self, tracker_ids, frames_dev_ptr, b, h, w, cap_rows=512):
        """frames_dev_ptr: integer device address of B x H x W x 3 uint8 BGR frames (e.g. torch_tensor.data_ptr()).
        Returns (per-frame row arrays [m_i, 6] = x1,y1,x2,y2,track id,label; detections per frame)."""
        rows, m, nd = self._stream_call(tracker_ids, frames_dev_ptr, b, h, w, cap_rows)
        return [rows[i, : m[i]].copy() for i in range(b)], nd.copy()

    def stream_run_packed(self, tracker_ids, frames_dev_ptr, b, h, w, cap_rows=512):
        """Same step, rows of all frames packed: (rows [sum m, 6], frame index of each row [sum m], detections per frame)."""
        rows, m, nd = self._stream_call(tracker_ids, frames_dev_ptr, b, h, w, cap_rows)
        keep = np.arange(cap_rows)[None, :] < m[:, None]
        return rows[keep], np.repeat(np.arange(b), m), nd.copy()

    def stream_run_async(self, tracker_ids, frames_dev_ptr, b, h, w, cap_rows=512):
        """Enqueue the batch's ReID and its tracker kernel (one launch on the engine's tracker stream, no host thread) and return;
        `stream_collect` picks the rows up (in order)."""
        tr = np.ascontiguousarray(tracker_ids, dtype=np.int32)
        L.check(L.lib().vc_stream_run_async(self._h, L.ptr(tr, C.c_int), len(tr), C.c_void_p(frames_dev_ptr), b, h, w, cap_rows))
        self._async_shapes = getattr(self, "_async_shapes", [])
        self._async_shapes.append((b, cap_rows))

    def stream_run_async_multi(self, tracker_ids, cam_of_frame, frames_dev_ptr, b, h, w, cap_rows=512):
        """Multi-camera batch: tracker_ids [n_cam][num_classes], cam_of_frame [b] (each camera's frames in stream order); rows are
        collected with `stream_collect` per frame of the batch, as for one camera."""
        tr = np.ascontiguousarray(tracker_ids, dtype=np.int32)
        assert tr.ndim == 2, "tracker_ids must be [n_cam][num_classes]"
        cams = np.ascontiguousarray(cam_of_frame, dtype=np.int32).reshape(-1)
        assert len(cams) == b
        L.check(L.lib().vc_stream_run_async_multi(self._h, L.ptr(tr, C.c_int), tr.shape[0], tr.shape[1], L.ptr(cams, C.c_int),
                                                  C.c_void_p(frames_dev_ptr), b, h, w, cap_rows))
        self._async_shapes = getattr(self, "_async_shapes", [])
        self._async_shapes.append((b, cap_rows))

    def stream_collect(self):
        """Rows of the oldest asynchronous batch, packed like `stream_run_packed` (blocks until its tracker loop is done)."""
        if not getattr(self, "_async_shapes", None):
            raise L.VcError(3, "stream_collect: no asynchronous batch is in flight (stream_run_async first)")
        b, cap_rows = self._async_shapes.pop(0)
        rows, m, nd = np.empty((b, cap_rows, 6), np.int64), np.zeros(b, np.int32), np.zeros(b, np.int32)
        L.check(L.lib().vc_stream_collect(self._h, L.ptr(rows, C.c_int64), cap_rows, L.ptr(m, C.c_int), L.ptr(nd, C.c_int), b))
        keep = np.arange(cap_rows)[None, :] < m[:, None]
        return rows[keep], np.repeat(np.arange(b), m), nd

    def stream_submit(self, frames_dev_ptr, b, h, w):
        """Enqueue the detector for a batch (returns immediately); the matching stream_run consumes it."""
        L.check(L.lib().vc_stream_submit(self._h, C.c_void_p(frames_dev_ptr), b, h, w))

    def stream_stage_host(self, frames_host_ptr, b, h, w):
        """Enqueue ONLY the host-to-device copy of a batch of (pinned) host frames; returns the device address to pass to stream_submit
        (which starts the detector behind the copy) and then stream_run / stream_run_async.  Stage batch i + 2 before submitting
        batch i + 1 and the copy runs under the detector of the batch before it."""
        out = C.c_void_p()
        L.check(L.lib().vc_stream_stage_host(self._h, C.c_void_p(frames_host_ptr), b, h, w, C.byref(out)))
        return out.value

    def stream_submit_host(self, frames_host_ptr, b, h, w):
        """Enqueue the host-to-device copy of a batch of (pinned) host frames and the detector behind it; returns the device
        address to pass to stream_run / stream_run_async for this batch."""
        out = C.c_void_p()
        L.check(L.lib().vc_stream_submit_host(self._h, C.c_void_p(frames_host_ptr), b, h, w, C.byref(out)))
        return out.value

    # ---------------------------------------------------------------- frame-sharded front end (one stream on several GPUs)
    def stream_embed(self, frames_dev_ptr, b, h, w):
        """Front half of the fused path for the oldest submission: (rows [n, 7] float64 = frame index in the batch, x1, y1, x2, y2,
        conf, label -- the boxes VideoTracker.run works on -- and the DEVICE address of the matching [n, 512] float32 embeddings)."""
        cap = b * self.cfg.max_det
        rows = np.zeros((cap, 7), np.float64)
        n, feat = C.c_int(), C.c_void_p()
        L.check(L.lib().vc_stream_embed(self._h, C.c_void_p(frames_dev_ptr), b, h, w, L.ptr(rows, C.c_double), cap, C.byref(n), C.byref(feat)))
        return rows[: n.value].copy(), feat.value or 0

    def allgather_rows(self, rows7, feat_dev_ptr, world):
        """RCCL all-gather (C ABI, vc_comm_init first) of every rank's rows + embeddings: (rows of all ranks, rank-major; device
        address of the gathered embeddings in the same order; rows per rank)."""
        rows = L.f64(rows7).reshape(-1, 7)
        cap = world * self.cfg.max_batch * self.cfg.max_det
        out = np.zeros((cap, 7), np.float64)
        counts = np.zeros(world, np.int32)
        feat = C.c_void_p()
        L.check(L.lib().vc_allgather_rows(self._h, L.ptr(rows, C.c_double), C.c_void_p(feat_dev_ptr or None), len(rows), L.ptr(out, C.c_double), cap,
                                          L.ptr(counts, C.c_int), C.byref(feat)))
        return out[: int(counts.sum())].copy(), feat.value or 0, counts

    def videotracker_run_features(self, tracker_ids, rows7, feat_dev_ptr, h, w, cap_rows=512):
        """VideoTracker.run for a run of frames with supplied detections (rows7 sorted by frame key) and device-resident embeddings:
        [(frame key, rows [m, 6] = x1, y1, x2, y2, track id, label)] per distinct key, ascending; one tracker kernel launch."""
        rows = L.f64(rows7).reshape(-1, 7)
        tr = np.ascontiguousarray(tracker_ids, dtype=np.int32)
        nf_cap = max(len(np.unique(rows[:, 0])), 1)
        out = np.empty((nf_cap, cap_rows, 6), np.int64)
        m, keys, nf = np.zeros(nf_cap, np.int32), np.zeros(nf_cap, np.int64), C.c_int()
        L.check(L.lib().vc_videotracker_run_features(self._h, L.ptr(tr, C.c_int), len(tr), L.ptr(rows, C.c_double), C.c_void_p(feat_dev_ptr or None),
                                                     len(rows), h, w, L.ptr(out, C.c_int64), cap_rows, L.ptr(m, C.c_int), L.ptr(keys, C.c_int64),
                                                     nf_cap, C.byref(nf)))
        return [(int(keys[j]), out[j, : m[j]].copy()) for j in range(nf.value)]

    def stream_inject(self, det6=None, counts=None):
        if det6 is None:
            L.check(L.lib().vc_stream_inject(self._h, None, None, 0, 0))
            return
        d = L.f32(det6)
        c = np.ascontiguousarray(counts, dtype=np.int32)
        L.check(L.lib().vc_stream_inject(self._h, L.ptr(d, C.c_float), L.ptr(c, C.c_int), d.shape[0], d.shape[1]))

    # ---------------------------------------------------------------- measurement
    def profile(self, on):
        """False/0 off; True/1 blocking per-launch events for every stage (serialises the streams); 2 in-flight event pairs
        around the conv launches only (no host waits: usable inside a timed region), resolved by profile_read(PROF_CONV)."""
        L.check(L.lib().vc_profile_enable(self._h, 2 if on == 2 else (1 if on else 0)))

    def profile_reset(self):
        L.check(L.lib().vc_profile_reset(self._h))

    def profile_ops(self):
        buf = C.create_string_buffer(1 << 20)
        L.check(L.lib().vc_profile_ops(self._h, buf, len(buf)))
        return buf.value.decode()

    def profile_conv_busy(self):
        """(union_ms, span_ms) of the conv launches of the last resolved in-flight profiling region."""
        u, sp = C.c_double(), C.c_double()
        L.check(L.lib().vc_profile_conv_busy(self._h, C.byref(u), C.byref(sp)))
        return u.value, sp.value

    def profile_read_dense(self, cat):
        """(flops, bytes) of the category with the sparse Detect head credited as the dense head it replaces (after profile_read)."""
        fl, by = C.c_double(), C.c_double()
        L.check(L.lib().vc_profile_read_dense(self._h, cat, C.byref(fl), C.byref(by)))
        return fl.value, by.value

    def tune_export(self) -> str:
        """Conv autotune choices of this engine as text (the VC_TUNE_CACHE file format)."""
        n = C.c_size_t()
        L.check(L.lib().vc_tune_export(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        L.check(L.lib().vc_tune_export(self._h, buf, n.value, C.byref(n)))
        return buf.value.decode()

    def tune_import(self, text: str):
        """Adopt another engine's autotune choices (before the first launch of those shapes)."""
        L.check(L.lib().vc_tune_import(self._h, text.encode()))

    def profile_read(self, cat):
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        L.check(L.lib().vc_profile_read(self._h, cat, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}


# ---- single-function entry points (used by the parity tests) -------------------------------------------------------
def conv2d(x_nhwc, w_oihw, bias, *, stride=1, pad=0, act=0, res=None, res_mode=0, precision="bf16"):
    x, w, b = L.f32(x_nhwc), L.f32(w_oihw), L.f32(bias)
    B, H, W, Ci = x.shape
    Co, _, kh, kw = w.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    d = L.ConvDesc(B, H, W, Ci, Co, kh, kw, stride, pad, act, res_mode, L.PREC_ID[precision])
    y = np.zeros((B, Ho, Wo, Co), np.float32)
    r = L.f32(res) if res is not None else None
    L.check(L.lib().vc_conv2d_host(C.byref(d), L.ptr(x, C.c_float), L.ptr(w, C.c_float), L.ptr(b, C.c_float),
                                   L.ptr(r, C.c_float), L.ptr(y, C.c_float)))
    return y


def kalman_initiate(xyah):
    z = L.f64(xyah).reshape(-1, 4)
    m, c = np.zeros((len(z), 8)), np.zeros((len(z), 8, 8))
    L.check(L.lib().vc_kalman_initiate_host(L.ptr(z, C.c_double), len(z), L.ptr(m, C.c_double), L.ptr(c, C.c_double)))
    return m, c


def kalman_predict(mean, cov):
    m, c = L.f64(mean).reshape(-1, 8).copy(), L.f64(cov).reshape(-1, 8, 8).copy()
    L.check(L.lib().vc_kalman_predict_host(L.ptr(m, C.c_double), L.ptr(c, C.c_double), len(m)))
    return m, c


def kalman_update(mean, cov, z):
    m, c, z = L.f64(mean).reshape(-1, 8).copy(), L.f64(cov).reshape(-1, 8, 8).copy(), L.f64(z).reshape(-1, 4)
    L.check(L.lib().vc_kalman_update_host(L.ptr(m, C.c_double), L.ptr(c, C.c_double), L.ptr(z, C.c_double), len(m)))
    return m, c


def kalman_gating(mean, cov, zs):
    m, c, z = L.f64(mean).reshape(8), L.f64(cov).reshape(8, 8), L.f64(zs).reshape(-1, 4)
    out = np.zeros(len(z))
    L.check(L.lib().vc_kalman_gating_host(L.ptr(m, C.c_double), L.ptr(c, C.c_double), L.ptr(z, C.c_double), len(z), L.ptr(out, C.c_double)))
    return out


def iou_matrix(track_tlwh, det_tlwh):
    a, b = L.f64(track_tlwh).reshape(-1, 4), L.f64(det_tlwh).reshape(-1, 4)
    out = np.zeros((len(a), len(b)))
    L.check(L.lib().vc_iou_cost_host(L.ptr(a, C.c_double), len(a), L.ptr(b, C.c_double), len(b), L.ptr(out, C.c_double)))
    return out


def cosine_cost(galleries, feats):
    """galleries: list of (s_i, 512) arrays -> (T, D) min cosine distance."""
    t, s_cap = len(galleries), max(len(g) for g in galleries)
    gal = np.zeros((t, s_cap, L.FEAT_DIM), np.float32)
    cnt = np.zeros(t, np.int32)
    for i, g in enumerate(galleries):
        gal[i, : len(g)] = g
        cnt[i] = len(g)
    f = L.f32(feats).reshape(-1, L.FEAT_DIM)
    out = np.zeros((t, len(f)))
    L.check(L.lib().vc_cosine_cost_host(L.ptr(gal, C.c_float), L.ptr(cnt, C.c_int), t, s_cap, L.ptr(f, C.c_float), len(f), L.ptr(out, C.c_double)))
    return out


def dsort_nms(tlwh, scores, max_overlap):
    b, s = L.f64(tlwh).reshape(-1, 4), L.f64(scores).reshape(-1)
    keep = np.zeros(max(len(s), 1), np.int32)
    n = C.c_int()
    L.check(L.lib().vc_dsort_nms_host(L.ptr(b, C.c_double), L.ptr(s, C.c_double), len(s), float(max_overlap), L.ptr(keep, C.c_int), C.byref(n)))
    return keep[: n.value].tolist()


def lap(cost):
    c = L.f64(cost)
    nr, nc = c.shape
    k = max(min(nr, nc), 1)
    r, q = np.zeros(k, np.int32), np.zeros(k, np.int32)
    n = C.c_int()
    L.check(L.lib().vc_lap_host(L.ptr(c, C.c_double), nr, nc, L.ptr(r, C.c_int), L.ptr(q, C.c_int), C.byref(n)))
    return r[: n.value].copy(), q[: n.value].copy()


def letterbox(rgb, net_h, net_w, precision="f32"):
    im = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.zeros((net_h, net_w, 3), np.float32)
    L.check(L.lib().vc_letterbox_host(L.ptr(im, C.c_uint8), im.shape[0], im.shape[1], net_h, net_w,
                                      L.PREC_BF16 if precision == "bf16" else L.PREC_F32, L.ptr(out, C.c_float)))
    return out


def nms(boxes, conf, cls, iou=0.45, max_det=300, max_cand=4096):
    b, c = L.f32(boxes).reshape(-1, 4), L.f32(conf).reshape(-1)
    k = np.ascontiguousarray(cls, dtype=np.int32).reshape(-1)
    out = np.zeros((max_det, 6), np.float32)
    n = C.c_int()
    L.check(L.lib().vc_nms_host(L.ptr(b, C.c_float), L.ptr(c, C.c_float), L.ptr(k, C.c_int), len(c), iou, max_det, max_cand,
                                L.ptr(out, C.c_float), C.byref(n)))
    return out[: n.value].copy()
