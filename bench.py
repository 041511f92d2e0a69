"""bench.py -- end-to-end frames/s of the detect + NMS + ReID + track hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload s640-bf16|m1024-bf16|l1280-fp8]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of B synthetic frames of ONE camera stream per GPU: letterbox -> YOLOv5
conv stack -> decode -> NMS -> crops -> ReID CNN -> per-class DeepSORT (device-resident, one kernel per batch), frames already
resident in HBM.  Each rank owns its own camera stream (weak scaling); the only collective is the all-gather of
the per-camera count tensors at the end.  Rank 0 prints ONE JSON line.

Workloads (config.workload names the one measured):
  s640-bf16   BASELINE.json configs[1] -- the headline: YOLOv5s, 640x640 frames, bf16 convs, B = 128, 512-frame clip, 12 objects
  m1024-bf16  BASELINE.json configs[2] -- YOLOv5m, 1024x1024 frames, bf16 convs, B = 32, 256 ground-truth rectangles per frame
              injected after the conv stack has run (<= 256 detections per frame through NMS + ReID + DeepSORT)
  l1280-fp8   BASELINE.json configs[4] -- YOLOv5l, 1280x1280 frames, fp8 (MX-scaled MFMA) detector convs, B = 16, ground-truth
              rectangles injected after the conv stack has run (the seeded random head of the deep variant saturates)
With the default workload on one GPU the line also carries `extra_points`: the same pipeline at K = 32 and K = 256 detections per
frame (detection injection; K = 256 with its per-stage split), the fp32 engine (the mode whose CSV equals the
oracle's exactly), short runs of the m1024-bf16 and l1280-fp8 workloads with their own `roofline`, and `value_host_frames`, the
PCIe-inclusive rate with the frames in pinned host memory.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import vehicle_counting_amd.engine as E  # noqa: E402
from vehicle_counting_amd import _lib as L  # noqa: E402
from vehicle_counting_amd import parallel  # noqa: E402
from vehicle_counting_amd.counting import NativeCounter, count_directions, csv_records  # noqa: E402
from vehicle_counting_amd.coded import coded_frames, coded_yolo  # noqa: E402
from vehicle_counting_amd.synth import synth_frames, synth_tracks  # noqa: E402
from vehicle_counting_amd.track import VideoCounting  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

NC = 80
TRACK = dict(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
PEAK_TFLOPS = {"bf16": 2500.0, "fp8": 5000.0, "f32": 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md; fp8 = MX-scaled K = 128 form; f32 = v_mfma_f32_16x16x4_f32, the vector rate)
PEAK_HBM_GBS = 8000.0                              # HBM3E (MI355X_MICROARCH.md)
ACHIEVABLE_HBM_GBS = 6300.0                        # what a read + write stream reaches (tools/ubench/stream_bw.hip)
ZONE = os.path.join(ROOT, "tests", "golden", "cam_04_halfres.json")
ZONE_720P = os.path.join(ROOT, "tests", "golden", "cam_04.json")           # the reference's own zone file (demo/sample/cam_04.json, 1280 x 720)

WORKLOADS = {
    # 256 frames per vc_stream_* call (round 6; 128 before): +4.4 % frames/s on the same box, two alternations (per-launch fixed costs and tile
    # quantisation of the 20 x 20 level over twice the pixels); the batch sweep below keeps the 128-frame point
    "s640-bf16": dict(model="yolov5s", size=640, precision="bf16", B=int(os.environ.get("VC_BENCH_B", 256)),
                      clip=int(os.environ.get("VC_BENCH_CLIP", 512)), n_obj=12, inject=0, obj_shift=1.0,
                      desc="YOLOv5s 640x640 single camera stream per GPU, bf16 convs (BASELINE.json configs[1])"),
    "m1024-bf16": dict(model="yolov5m", size=1024, precision="bf16", B=int(os.environ.get("VC_BENCH_B", 32)), clip=64, n_obj=256, inject=256,
                       det_scale=16.0, obj_shift=2.0,   # the random head's own output: ~1 % of the 64 512 candidates pass conf_thres (oracle-calibrated), decoded + NMSed, then replaced
                       desc="YOLOv5m 1024x1024 single camera stream per GPU, bf16 convs, 256 ground-truth rectangles per frame injected after the "
                            "conv stack (BASELINE.json configs[2]: <= 256 detections per frame through NMS + DeepSORT ReID)"),
    "l1280-fp8": dict(model="yolov5l", size=1280, precision="fp8", B=int(os.environ.get("VC_BENCH_B", 32)), clip=64, n_obj=16, inject=16,   # 32 frames per call (16 until round 6: + 8 %; 64 would put layer 1's output past the 2 GiB a buffer descriptor addresses)
                      det_scale=1.0, obj_shift=-24.0,   # calibrated like the 640 workload: 20-80 boxes per frame survive the random head's NMS (tools/head_calib.py)
                      desc="YOLOv5l 1280x1280 single camera stream per GPU, fp8 MX-MFMA detector convs (BASELINE.json configs[4]), "
                           "16 ground-truth rectangles per frame injected after the conv stack"),
}


def host_cpu_info():
    """`lscpu` of the box the CPU baseline runs on: model, sockets, physical cores, threads per core."""
    import subprocess
    info = {}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU(s)", "NUMA node(s)"):
                info[k.strip()] = v.strip()
    except Exception as ex:                                             # no lscpu: say so in the line
        info["error"] = str(ex)[:100]
    try:
        info["physical_cores"] = int(info["Socket(s)"]) * int(info["Core(s) per socket"])
    except Exception:
        info["physical_cores"] = os.cpu_count() or 1
    return info


# The same detector on the reference's real geometry (1280 x 720 frames -> 384 x 640 tensor, Q8): the seeded random head is calibrated per
# frame geometry (tools/head_calib.py, VC_FRAME_HW=720,1280: obj_shift 6 -> 1 box per frame, 8 -> 21) so that the point carries the
# headline's ~15 detections per frame through ReID and the tracker
WL_720P = dict(WORKLOADS["s640-bf16"], obj_shift=7.8, desc="YOLOv5s, 1280x720 frames (384x640 tensor, Q8), single camera stream per GPU, bf16 convs")


def cpu_baseline(ysd, rsd, frames, n_frames, repeats=3):
    """The oracle (CPU port of the reference path) timed on this box's host cores over a bounded sample: `repeats` runs of the same
    n_frames-frame sample on one thread per PHYSICAL core (lscpu), the median reported, every run listed."""
    from oracle import pipeline as op
    cfg = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    cpu = host_cpu_info()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(max(1, cpu["physical_cores"]))
    try:
        op.run_video(frames[:1], ysd, rsd, cfg, ZONE, nc=NC)            # warm the CPU kernels
        runs, nd = [], None
        load = os.getloadavg()[0]                                       # other tenants on the box's host cores show up here (and in the spread of the runs)
        t_all = time.perf_counter()
        for _ in range(repeats):
            t0 = time.perf_counter()
            _, _, nd = op.run_video(frames[:n_frames], ysd, rsd, cfg, ZONE, nc=NC)
            runs.append(n_frames / (time.perf_counter() - t0))
            if time.perf_counter() - t_all > 40.0:                      # bounded: a loaded host gets fewer repeats, not a longer bench
                break
    finally:
        torch.set_num_threads(threads_before)
    return {"value": float(np.median(runs)), "unit": "frames/s", "cores": cpu["physical_cores"], "kind": "port",
            "runs_fps": [round(r, 3) for r in runs], "best_fps": round(max(runs), 3), "host_load_1min_before": round(load, 1), "lscpu": cpu,
            "sample": f"median of {len(runs)} runs over the first {n_frames} frames of the rank-0 stream through oracle/pipeline.py (torch-CPU fp32 "
                      f"YOLOv5s + ReID, NumPy/SciPy DeepSORT, batch 1), torch threads = physical cores, {int(np.mean(nd))} det/frame, "
                      f"{sum(n_frames / r for r in runs):.1f} s of CPU work"}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command under torch.distributed.run with one rank per GPU
    (127.0.0.1 rendezvous, free port), exactly what the driver's `python -m torch.distributed.run --nproc-per-node N` does.
    Fails loudly when fewer than N GPUs are visible -- a silent single-rank run would print a meaningless scaling point."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {ndev}; refusing to run fewer ranks", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def injected_detections(n_frames, hw, n_obj, seed):
    """Ground-truth rectangles of the synthetic stream as detector output rows [x1, y1, x2, y2, conf, cls] (SURVEY.md 8d injection)."""
    tr = synth_tracks(n_frames, hw[0], hw[1], n_obj=n_obj, seed=seed, bounce=True)
    det = np.zeros((n_frames, n_obj, 6), np.float32)
    for f, (xywh, labels, scores) in enumerate(tr):
        det[f, :, 0:2] = xywh[:, 0:2]
        det[f, :, 2:4] = xywh[:, 0:2] + xywh[:, 2:4]
        det[f, :, 4] = scores
        det[f, :, 5] = labels
    return det, np.full(n_frames, n_obj, np.int32)


class Stream:
    """One camera stream on one engine: the three overlapped stages of the fused path."""

    def __init__(self, wl, rank, local, dev, n_obj=None, inject=None, B=None, clip=None, precision=None, n_cam=1, frame_hw=None, zone=None, distinct_cams=False, coded=False):
        wl = dict(wl, precision=precision or wl["precision"])
        self.wl, self.dev = wl, dev
        self.B = B or wl["B"]
        self.H, self.W = frame_hw or (wl["size"], wl["size"])     # frame_hw: frames of another geometry through the same engine (1280 x 720 -> 384 x 640 tensor, Q8)
        self.zone = zone or ZONE
        n_obj = n_obj or wl["n_obj"]
        inject = wl["inject"] if inject is None else inject
        clip = clip or wl["clip"]
        clip = max(self.B, clip // self.B * self.B)                  # whole batches, so a batch is one contiguous run of frames
        # coded: the well-conditioned detector of vehicle_counting_amd/coded.py on its plate-carrying clip -- the detector's OWN boxes (no
        # injection) are exactly one per object, in every precision (tests/test_gpu_coded.py holds this configuration to the oracle's CSV)
        self.ysd = coded_yolo(wl["model"], nc=NC) if coded else synth_yolo(wl["model"], nc=NC, seed=1702, det_scale=wl.get("det_scale", 4.0), obj_shift=wl["obj_shift"])
        self.rsd = synth_reid(1702)
        per_frame = max(160 if distinct_cams else 64, 2 * max(inject, n_obj))
        max_crops = max(512, self.B * per_frame)                     # (small batches: a single busy frame of the random head can carry > 64 boxes)
        self.eng = E.Engine(self.ysd, self.rsd, device=local, precision=wl["precision"], model_name=wl["model"], num_classes=NC,
                            img_size=wl["size"], max_batch=self.B, max_frame_hw=(self.H, self.W), max_crops=max_crops,
                            max_tracks=max(32768 if distinct_cams else 8192, 64 * inject), nn_budget_cap=60, max_candidates=8192 if wl["size"] > 640 else 4096,
                            max_trackers=max(256, n_cam * NC))
        k = 32
        sizes = []
        while k <= max_crops:
            sizes.append(k); k *= 2
        # n_cam > 1: S cameras interleaved frame by frame in every batch (vc_stream_run_async_multi): one engine, B / S frames of latency
        # per camera, S x 80 trackers walked in parallel by the tracker kernel
        self.n_cam = n_cam
        assert self.B % n_cam == 0 and clip % n_cam == 0
        self.trackers = [[self.eng.tracker_create(**TRACK) for _ in range(NC)] for _ in range(n_cam)]
        # every camera shows the rank's clip (same content, independent trackers): the point stays comparable with the single-camera
        # headline -- other seeds draw 4 x more boxes from the random head
        one = (coded_frames(clip // n_cam, self.H, self.W, n_obj=n_obj, seed=1702 + rank, size=wl["size"])[0] if coded else
               synth_frames(clip // n_cam, self.H, self.W, n_obj=n_obj, seed=1702 + rank, bounce=True))
        per_cam = [one] * n_cam
        if distinct_cams:                                            # eight different scenes (other seeds: other objects, and more boxes from the random head)
            per_cam = [one] + [synth_frames(clip // n_cam, self.H, self.W, n_obj=n_obj, seed=1702 + rank + 101 * c, bounce=True) for c in range(1, n_cam)]
        self.frames = per_cam[0] if n_cam == 1 else np.stack(per_cam, 1).reshape((clip,) + per_cam[0].shape[1:])     # frame j: camera j % S, time j // S
        self.cams = np.tile(np.arange(n_cam, dtype=np.int32), self.B // n_cam)
        self.d_frames = torch.from_numpy(self.frames).to(dev)       # resident in HBM before the timed region
        self.clip = clip
        self.inject = injected_detections(clip, (self.H, self.W), inject, 1702 + rank) if inject else None
        assert not (inject and n_cam > 1)
        self.counters = [VideoCounting([str(c) for c in range(NC)], self.zone) for _ in range(n_cam)]    # the Python restatement: checks the native counter after the timed region
        self.ncounters = [NativeCounter(self.zone, NC) for _ in range(n_cam)]     # the count tensors behind the C ABI (vc_counter_* / vc_counts), one per camera
        self.ncounter = self.ncounters[0]
        self.gather_via = None
        try:                                                                 # bring the RCCL communicator up before anything is timed
            parallel.allgather_counts_native(self.eng, np.zeros((1, len(self.ncounter.direction_keys), NC), np.int32))
        except L.VcError as ex:
            print(f"bench.py: vc_comm_init failed ({ex})", file=sys.stderr)
        self.ndet = [0, 0]
        self.nrows = 0
        self.kept = [[] for _ in range(n_cam)]
        self.host = None
        self._dev_ptr = {}
        self.t_submit, self.lat = {}, []                            # submit -> rows-on-the-host latency of every recorded batch

        def tune():                                                  # conv autotune: every ReID size bucket + one detector batch
            self.eng.pretune(tuple(sizes))
            self.eng.stream_submit(self.batch_ptr(0), self.B, self.H, self.W)
            self.eng.sync()
            self.eng.stream_reset()
        # N ranks: rank 0 tunes, the others adopt its choices (one timing pass instead of N, identical tile configurations everywhere)
        parallel.share_tune_cache(self.eng, tune)

    def use_host_frames(self):
        """Frames in pinned host memory: every batch crosses PCIe inside the timed region (vc_stream_submit_host)."""
        self.host = torch.from_numpy(self.frames).pin_memory()

    def batch_ptr(self, i):
        f0 = (i * self.B) % self.clip
        return self.d_frames[f0:f0 + self.B].data_ptr()

    def submit(self, i):
        self.t_submit[i] = time.perf_counter()
        if self.inject is not None:                                  # batch i gets the rectangles of its own frames (captured at submit time)
            f0 = (i * self.B) % self.clip
            self.eng.stream_inject(self.inject[0][f0:f0 + self.B], self.inject[1][f0:f0 + self.B])
        if self.host is not None:
            if i not in self._dev_ptr:
                self.stage(i)
            self.eng.stream_submit(self._dev_ptr[i], self.B, self.H, self.W)
        else:
            self.eng.stream_submit(self.batch_ptr(i), self.B, self.H, self.W)

    def stage(self, i):
        """Host frames: the PCIe copy of batch i, issued one batch further ahead than its detector (vc_stream_stage_host)."""
        f0 = (i * self.B) % self.clip
        self._dev_ptr[i] = self.eng.stream_stage_host(self.host[f0:f0 + self.B].data_ptr(), self.B, self.H, self.W)

    def run_async(self, i):
        ptr = self._dev_ptr.pop(i) if self.host is not None else self.batch_ptr(i)
        if self.n_cam == 1:
            self.eng.stream_run_async(self.trackers[0], ptr, self.B, self.H, self.W)
        else:
            self.eng.stream_run_async_multi(self.trackers, self.cams, ptr, self.B, self.H, self.W)

    def collect(self, i, record):
        rows, fidx, nd = self.eng.stream_collect()
        t_sub = self.t_submit.pop(i, None)
        if record:
            if t_sub is not None:
                self.lat.append(time.perf_counter() - t_sub)
            self.ndet[0] += int(nd.sum()); self.ndet[1] += self.B
            self.nrows += len(rows)
            # VideoCounting.run for one batch behind the C ABI (zone filter, per-track rows), on the host while the GPU works on the
            # next batches; the rows are kept so that the Python VideoCounting can check the result after the timed region
            S = self.n_cam
            g = i * self.B + fidx                                            # position in the interleaved stream: camera g % S, time g // S
            for c in range(S):
                sel = slice(None) if S == 1 else (g % S == c)
                fr, r = g[sel] // S + 1, rows[sel]
                self.ncounters[c].add(fr, r[:, 4], r[:, 5], r[:, :4])
                self.kept[c].append((fr, r))

    def run_steps(self, first, n, record):
        """Three overlapped stages: detector of batch i+1 (own stream), ReID of batch i (own stream), tracker kernel of batch i
        (tracker stream); rows come back one batch late.  Everything is collected before the function returns, so the timed
        region contains the whole work of its n steps."""
        if n <= 0:
            return
        if self.host is not None:                                    # staging order = batch order (the four slots are handed out round-robin)
            self.stage(first)
            if n > 1 and not os.environ.get("VC_BENCH_HOST_INLINE"):
                self.stage(first + 1)
        self.submit(first)
        for i in range(first, first + n):
            if self.host is not None and i + 2 < first + n and not os.environ.get("VC_BENCH_HOST_INLINE"):   # (A/B switch: copy in front of its own detector)
                self.stage(i + 2)                                    # copy of batch i + 2 under the detector of batch i + 1
            if i + 1 < first + n:
                self.submit(i + 1)
            self.run_async(i)
            if i > first:
                self.collect(i - 1, record)
        self.collect(first + n - 1, record)

    def sync(self, world):
        self.eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    def timed(self, first, steps, world):
        """(seconds, post-pass ms, gathered counts) for `steps` steps starting at batch `first`, counting post-pass included."""
        self.sync(world)
        gc.collect(); gc.disable()                 # a generation-2 collection over the per-track box lists costs 40 ms when it lands in the serial tail
        t0 = time.perf_counter()
        self.run_steps(first, steps, True)
        t_post = time.perf_counter()
        tables = [nc.table() for nc in self.ncounters]                      # save_tracking_to_csv's table: directions, first / last points, every row
        local_counts = np.stack([nc.counts() for nc in self.ncounters])    # int32 [cameras of this rank][n_dir][n_cls]
        try:                                                               # the one collective: ncclAllGather on the engine's stream (C ABI)
            all_counts = parallel.allgather_counts_native(self.eng, local_counts)
            self.gather_via = "vc_allgather_counts (RCCL, C ABI)"
        except L.VcError as ex:                                            # loud, recorded in the JSON line -- never silent
            print(f"bench.py: C-ABI count all-gather failed ({ex}); using torch.distributed.all_gather", file=sys.stderr)
            all_counts = parallel.allgather_counts(local_counts, device=self.dev if world > 1 else None)
            self.gather_via = f"torch.distributed.all_gather (vc_allgather_counts failed: {ex})"
        self.sync(world)
        dt = time.perf_counter() - t0
        gc.enable()
        post_ms = (time.perf_counter() - t_post) * 1e3
        # outside the timed region: the Python VideoCounting + csv_records + count_directions over the same rows must agree
        for c in range(self.n_cam):
            counter, table = self.counters[c], tables[c]
            for fr, rows in self.kept[c]:
                counter.run(fr.tolist(), rows[:, 4].tolist(), rows[:, 5].tolist(), np.ascontiguousarray(rows[:, :4]), finalize=False)
            self.kept[c] = []
            recs = csv_records(counter.run([], [], [], np.zeros((0, 4), np.int64)))
            dirs = list(counter.directions.keys())
            ref = parallel.counts_to_tensor(count_directions(recs, dirs, NC), dirs, NC)
            same_rows = len(recs) == len(table["frame_id"]) and all(
                r["track_id"] == int(table["track_id"][k]) and r["frame_id"] == int(table["frame_id"][k]) and r["direction"] == table["direction"][k]
                and r["box"] == table["box"][k].tolist() for k, r in enumerate(recs))
            if not np.array_equal(ref, local_counts[c]) or not same_rows:
                raise SystemExit("bench.py: vc_counts / vc_counter_rows disagree with VideoCounting + csv_records + count_directions")
        return dt, post_ms, all_counts


def measure(st, warmup, steps, world, peak_tflops, first=0, traffic=None, traffic_src=None):
    """`warmup` untimed steps, then EXACTLY `steps` timed ones (barrier + synchronize on both sides, max over ranks), with the
    conv launches of the timed steps bracketed by in-flight HIP event pairs on their own streams; afterwards two extra steps with
    blocking events give the per-stage split.  Returns (seconds, post-pass ms, gathered counts, roofline, stage_ms_per_step)."""
    eng = st.eng
    st.run_steps(first, warmup, False)
    st.sync(world)
    eng.profile_reset()
    if not os.environ.get("VC_BENCH_NO_EVENTS"):  # (A/B switch: what the event pairs themselves cost)
        eng.profile(2)                             # in-flight event pairs around every conv launch of the timed steps (no host waits)
    dt, post_ms, all_counts = st.timed(first + warmup, steps, world)
    if world > 1:
        t = torch.tensor([dt], device=st.dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # roofline of the dominant kernel (the conv kernels): HIP events around every conv launch of the TIMED steps, recorded on
    # the stream the kernel is launched on and resolved after the region (the launches overlap with the other two streams, as in
    # the rocprofv3 trace of this command).  The per-stage split comes from two extra steps with blocking events.
    conv_timed = eng.profile_read(L.PROF_CONV)
    dense_flops, dense_bytes = eng.profile_read_dense(L.PROF_CONV)
    conv_union_ms, conv_span_ms = eng.profile_conv_busy()
    eng.profile(0)
    eng.profile(True); eng.profile_reset()
    nxt = first + warmup + steps
    for i in range(2):
        st.run_steps(nxt + i, 1, False)
    eng.sync()
    conv = eng.profile_read(L.PROF_CONV)
    cats = {n: eng.profile_read(c) for n, c in (("conv", L.PROF_CONV), ("detect_aux", L.PROF_DETECT_AUX),
                                                ("reid_aux", L.PROF_REID_AUX), ("track", L.PROF_TRACK))}
    eng.profile(False)
    isolated = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
    # Roofline of the dominant kernels (all conv launches of a step), SURVEY.md 8(d): bound = whichever of
    # flops / peak_mfma and bytes / peak_hbm is larger.  Time base = the wall clock of the timed region (ms_per_step): conv
    # launches of the detector and ReID streams overlap each other, so a sum of per-launch durations counts that time twice
    # (kept below as achieved_sum_of_overlapped_durations); flops-per-step / ms_per_step can be re-derived from the driver's own
    # clock and from profiles/ (launches per step x rocprofv3 average duration is <= ms_per_step).
    step_s = dt / steps
    n_meas_steps = max(conv_timed["launches"] / max(conv["launches"] / 2.0, 1.0), 1e-9)      # timed steps covered by the event pool
    flops_step, bytes_step = conv_timed["flops"] / n_meas_steps, conv_timed["bytes"] / n_meas_steps
    mfma_tflops, hbm_gbs = flops_step / step_s / 1e12, bytes_step / step_s / 1e9
    mfma_frac, hbm_frac = mfma_tflops / peak_tflops, hbm_gbs / PEAK_HBM_GBS
    overlapped = conv_timed["flops"] / (conv_timed["ms"] * 1e-3) / 1e12 if conv_timed["ms"] > 0 else 0.0
    hbm_bound = hbm_frac > mfma_frac
    roofline = {
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": hbm_gbs if hbm_bound else mfma_tflops, "peak": PEAK_HBM_GBS if hbm_bound else peak_tflops,
        "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": max(hbm_frac, mfma_frac), "traffic": traffic,
        "mfma_frac": mfma_frac, "mfma_tflops": mfma_tflops, "mfma_peak_tflops": peak_tflops, "hbm_frac": hbm_frac, "hbm_gbs": hbm_gbs,
        "time_base": "wall clock of the timed region: algorithmic conv work per step / ms_per_step",
        "kernel": "vc::conv_igemm_kernel<*> / conv3x3_halo_kernel<*> / conv3x3_halo_v2_kernel<*> / conv3x3s2_halo_kernel<*> / conv1x1_direct_kernel<*> / stem_direct_kernel<*> / front_fused_kernel<*> / c3_fused_kernel / bneck_fused_kernel / reid_block_fused_kernel / reid_stem_pool_kernel (all detector + ReID conv launches of a step)",
        "launches_per_step": conv["launches"] / 2.0,
        "algorithmic_gflop_per_step": flops_step / 1e9, "algorithmic_bytes_per_launch": conv_timed["bytes"] / max(conv_timed["launches"], 1),
        "algorithmic_note": "EXECUTED work of the launches: the sparse Detect head counts the gathered rows it computes (row counts read back from the device); the dense head it replaces is in algorithmic_dense_*, not in frac",
        "algorithmic_dense_gflop_per_step": dense_flops / n_meas_steps / 1e9, "algorithmic_dense_bytes_per_launch": dense_bytes / max(conv_timed["launches"], 1),
        "avg_launch_us": conv_timed["ms"] * 1e3 / max(conv_timed["launches"], 1),
        "avg_launch_note": "HIP start/stop timestamps of every conv dispatch of the timed steps (hipExtLaunchKernel); launches of the detector and ReID streams overlap, so launches_per_step x avg_launch_us may exceed ms_per_step",
        "achieved_sum_of_overlapped_durations_tflops": overlapped, "frac_sum_of_overlapped_durations": overlapped / peak_tflops,
        "achieved_isolated_tflops": isolated,
        "conv_running_frac_of_timed_window": conv_union_ms / conv_span_ms if conv_span_ms > 0 else None,
        "timed_launches_measured": int(conv_timed["launches"]),     # capped by the engine's pool of event pairs
        "traffic_note": ("HBM bytes per conv launch from %s (rocprofv3 --pmc FETCH_SIZE x2 (gfx950) + WRITE_SIZE, separate passes of this command)" % traffic_src) if traffic_src else None,
    }
    # the same algorithmic work over the AVERAGE LAUNCH DURATION (the contract's per-launch form; detector and ReID launches overlap, so
    # this is below the wall-clock `frac`), and which HBM peak `peak` is
    n_l = max(conv_timed["launches"], 1)
    avg_s = conv_timed["ms"] * 1e-3 / n_l
    if avg_s > 0:
        roofline["per_launch_mfma_frac"] = conv_timed["flops"] / n_l / avg_s / 1e12 / peak_tflops
        roofline["per_launch_hbm_frac"] = conv_timed["bytes"] / n_l / avg_s / 1e9 / PEAK_HBM_GBS
        roofline["per_launch_frac"] = max(roofline["per_launch_mfma_frac"], roofline["per_launch_hbm_frac"])
    roofline["hbm_peak_used"] = ("spec 8.0 TB/s (MI355X_MICROARCH.md); the copy rate tools/ubench/stream_bw reaches on this chip is %.1f TB/s, "
                                 "the floor profiles/*_layer_bounds.md prices HBM-bound launches with" % (ACHIEVABLE_HBM_GBS / 1e3))
    return dt, post_ms, all_counts, roofline, {k: v["ms"] / 2 for k, v in cats.items()}


def backbone_mfma_busy():
    """north_star's own quantity -- SQ_VALU_MFMA_BUSY_CYCLES of the DETECTOR's conv kernels alone, duration-weighted -- cannot be counted inside
    this process (PMC passes are separate rocprofv3 runs): the newest profiles/rNN_detector_only_mfma_util.md (tools/detector_only_mfma.sh) is quoted,
    source named, like `traffic`."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_detector_only_mfma_util.md")), reverse=True):
        with open(path) as f:
            text = f.read()
        m = re.search(r"duration-weighted: ([\d.]+) % MFMA busy", text)
        if m:
            out = {"backbone_mfma_busy": float(m.group(1)) / 100.0, "backbone_mfma_busy_src": "profiles/" + os.path.basename(path) +
                   " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz), detector conv kernels only, 128 frames of 640 x 640)"}
            m2 = re.search(r"at the measured shader clock of ([\d.]+) GHz: ([\d.]+) %", text)
            if m2:
                out["backbone_mfma_busy_at_measured_clock"] = float(m2.group(2)) / 100.0
                out["measured_shader_clock_ghz"] = float(m2.group(1))
            return out
    return {"backbone_mfma_busy": None, "backbone_mfma_busy_src": None}


def quick_point(wl, rank, local, dev, world, **kw):
    """quick_point_ with failures reported in the point itself: an extra operating point never takes the headline line down."""
    try:
        return quick_point_(wl, rank, local, dev, world, **kw)
    except Exception as ex:                                                       # loud, in the JSON line
        print(f"bench.py: extra point failed: {ex}", file=sys.stderr)
        torch.cuda.empty_cache()
        return {"error": str(ex)[:300]}


def quick_point_(wl, rank, local, dev, world, **kw):
    """Throughput of one extra operating point: a short run of the same pipeline with other stream parameters.  full=True adds the
    point's own `roofline` and per-stage split (same procedure as the headline)."""
    steps, warm = kw.pop("steps", 8), kw.pop("warmup", 2)
    host, full = kw.pop("host", False), kw.pop("full", False)
    st = Stream(wl, rank, local, dev, **kw)
    if host:
        st.use_host_frames()
    extra = {}
    if full:
        dt, _, _, roof, stages = measure(st, warm, steps, world, PEAK_TFLOPS[st.wl["precision"]])
        keep = ("bound", "achieved", "peak", "unit", "frac", "mfma_frac", "mfma_tflops", "mfma_peak_tflops", "hbm_frac", "hbm_gbs", "launches_per_step",
                "algorithmic_gflop_per_step", "algorithmic_bytes_per_launch", "algorithmic_dense_gflop_per_step", "avg_launch_us", "achieved_isolated_tflops")
        extra = {"roofline": {k: roof[k] for k in keep}, "stage_ms_per_step": stages, "ms_per_step": dt / steps * 1e3,
                 "dtype": st.wl["precision"], "workload": st.wl["desc"]}
    else:
        st.run_steps(0, warm, False)
        dt, _, _ = st.timed(warm, steps, world)
    out = {"value": steps * st.B / dt, "unit": "frames/s", "frames_per_step": st.B, "steps": steps,
           "det_per_frame": st.ndet[0] / max(st.ndet[1], 1), "tracked_rows": st.nrows,
           # submit -> rows on the host, per batch: what a frame waits in the three-stage pipeline (detector of batch i + 1 and ReID /
           # tracker of batch i overlap, rows come back one batch late)
           "latency_ms_submit_to_rows": float(np.mean(st.lat)) * 1e3 if st.lat else None}
    out.update(extra)
    st.eng.close()
    del st
    torch.cuda.empty_cache()
    return out


class RefLoaderFrames:
    """What /root/reference/modules/datasets.py:47-76 hands the loop, batch_size = 1 (:93): {'imgs': [RGB copy], 'ori_imgs': [BGR frame],
    'frames': [1-based id]} -- cv2.cvtColor makes `img` its own contiguous array.  Decode / colour conversion are the harness's input
    contract, so the batches are built before the timed region."""

    def __init__(self, frames_bgr, fps=10):
        t, h, w, _ = frames_bgr.shape
        self.frames = frames_bgr
        self.video_info = {"name": "synthetic.mp4", "width": w, "height": h, "fps": fps, "num_frames": t}
        self.batches = [{"imgs": [np.ascontiguousarray(f[:, :, ::-1])], "ori_imgs": [f], "frames": [i + 1]} for i, f in enumerate(frames_bgr)]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def dropin_point_(wl, local, frame_hw, n_frames, zone, n_obj=12, seed=1702, pipelined=False):
    """Frames/s of the reference's OWN loop through the drop-in classes: CountingPipeline.run = /root/reference/modules/__init__.py:54-84,
    host frames, ImageDetect.run one image at a time (batch_size = 1, modules/datasets.py:93), VideoTracker.run per frame, VideoCounting
    at the end -- every call crosses the C ABI with host buffers and blocks (PCIe both ways inside the timed region)."""
    from types import SimpleNamespace

    from vehicle_counting_amd.pipeline import CountingPipeline
    H, W = frame_hw
    ysd = synth_yolo(wl["model"], nc=NC, seed=1702, det_scale=wl.get("det_scale", 4.0), obj_shift=wl["obj_shift"])
    eng = E.Engine(ysd, synth_reid(1702), device=local, precision=wl["precision"], model_name=wl["model"], num_classes=NC, img_size=wl["size"],
                   max_batch=1, max_frame_hw=(H, W), max_crops=512, max_tracks=8192, nn_budget_cap=60, max_candidates=4096, max_trackers=256)
    args = SimpleNamespace(weight=None, output_path=None, mapping=None)
    config = SimpleNamespace(model_name=wl["model"], min_conf=0.25, min_iou=0.45, max_det=300)
    cam = {"cam": {"cam": {"tracking_config": dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)}}}
    pipe = CountingPipeline(args, config, cam, engine=eng, class_names=[str(c) for c in range(NC)])
    frames = synth_frames(n_frames, H, W, n_obj=n_obj, seed=seed, bounce=True)
    eng.pretune(tuple(range(1, 65)))                                     # ReID conv autotune for every crop count a frame of this clip can bring (size buckets)
    warm = RefLoaderFrames(frames[:24])
    src = RefLoaderFrames(frames)
    run = pipe.run_pipelined if pipelined else pipe.run
    run(warm, "cam", zone)                                               # conv autotune, first-use allocations; its trackers are discarded
    eng.sync(); torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    rows, counts = run(src, "cam", zone)                                 # a new VideoTracker per video, like modules/__init__.py:32-36
    eng.sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if pipelined:
        eng.close()
        torch.cuda.empty_cache()
        return {"value": n_frames / dt, "unit": "frames/s", "frames": n_frames, "frame_hw": [H, W], "batch_size": 1, "ms_per_frame": dt / n_frames * 1e3,
                "csv_rows": len(rows), "dtype": wl["precision"],
                "path": "CountingPipeline.run_pipelined: the same loader and the same per-frame body, stage calls asynchronous (batch n+1's detector behind batch n's ReID + tracker), host frames"}
    # the same loop once more with the two stage calls timed separately
    tracker, _ = pipe._stages("cam", src.video_info, zone)
    t_det = t_trk = 0.0
    n_det = 0
    for batch in src.batches[:64]:
        a = time.perf_counter()
        preds = pipe.detector.run(batch)
        b = time.perf_counter()
        if len(preds["boxes"][0]):
            tracker.run(batch["ori_imgs"][0], preds["boxes"][0], preds["labels"][0], preds["scores"][0])
        c = time.perf_counter()
        t_det += b - a; t_trk += c - b; n_det += len(preds["boxes"][0])
    nb = len(src.batches[:64])
    eng.close()
    torch.cuda.empty_cache()
    return {"value": n_frames / dt, "unit": "frames/s", "frames": n_frames, "frame_hw": [H, W], "batch_size": 1, "ms_per_frame": dt / n_frames * 1e3,
            "ms_ImageDetect_run": t_det / nb * 1e3, "ms_VideoTracker_run": t_trk / nb * 1e3, "det_per_frame": n_det / nb, "csv_rows": len(rows),
            "dtype": wl["precision"], "path": "CountingPipeline.run -> ImageDetect.run (vc_detect) + VideoTracker.run (vc_videotracker_run) per frame, host frames, blocking"}


def dropin_point(*a, **kw):
    try:
        return dropin_point_(*a, **kw)
    except Exception as ex:
        print(f"bench.py: drop-in point failed: {ex}", file=sys.stderr)
        torch.cuda.empty_cache()
        return {"error": str(ex)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="s640-bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra operating points (K = 32 / 256, fp32, m1024-bf16, l1280-fp8, host frames)")
    ap.add_argument("--extras", default="", help="comma-separated subset of the extra operating points to run (default: all)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))            # `python bench.py --gpus N`: become N ranks (one per GPU) over RCCL
    rank, world, local = parallel.init_from_env("nccl" if args.gpus > 1 else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly one rank per GPU")
    ndev = torch.cuda.device_count()
    if ndev < max(local + 1, 1) or (world > 1 and ndev < world):
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local} of {world}, but only {ndev} device(s) are visible (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    wl = WORKLOADS[args.workload]
    peak_tflops = PEAK_TFLOPS[wl["precision"]]

    st = Stream(wl, rank, local, dev)
    eng, B = st.eng, st.B
    # HBM traffic of the conv kernels: rocprofv3 PMC passes cannot run inside this process; tools/pmc_traffic.py stores the
    # per-launch figure of the same command under profiles/ (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)
    traffic, traffic_src = None, None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tp = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if args.workload == "s640-bf16" and os.path.exists(tp):
            with open(tp) as f:
                traffic, traffic_src = json.load(f)["conv_all"]["hbm_bytes_per_launch"], f"profiles/{rnd}_pmc_traffic.json"
            break
    dt, post_ms, all_counts, roofline, stage_ms = measure(st, args.warmup, args.steps, world, peak_tflops, traffic=traffic, traffic_src=traffic_src)
    if args.workload == "s640-bf16":
        roofline.update(backbone_mfma_busy())

    out = None
    if rank == 0:
        out = {
            "metric": "end-to-end frames/sec (detect+NMS+ReID+track), YOLOv5s 640px" if args.workload == "s640-bf16" else
                      "end-to-end frames/sec (detect+NMS+ReID+track), " + wl["desc"],
            "value": world * args.steps * B / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": wl["precision"], "data": "synthetic",
            "config": {"workload": wl["desc"], "workload_id": args.workload,
                       "frames_per_step": B, "frame_hw": [st.H, st.W], "clip_frames": st.clip, "objects": wl["n_obj"], "num_classes": NC,
                       "det_per_frame": st.ndet[0] / max(st.ndet[1], 1), "detection_injection": bool(wl["inject"]),
                       "weights": "seeded synthetic (no checkpoints available)", "streams": world, "ranks": world,
                       "counts_allgather_shape": list(all_counts.shape), "counts_allgather_via": st.gather_via, "tracked_rows": int(st.nrows),
                       "counting": "vc_counter_add per batch, vc_counter_rows + vc_counts after the last batch (C ABI), inside the timed region", "counting_postpass_ms_total": post_ms, "tracker": "device-resident (one kernel per batch, no host round trip per frame)"},
            "roofline": roofline,
            "stage_ms_per_step": stage_ms,
        }
    ysd, rsd, frames = st.ysd, st.rsd, st.frames
    eng.close()
    del st
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and args.workload == "s640-bf16" and not args.no_extras:
        # other operating points of SURVEY.md 8(d), short runs of the same pipeline (extra keys; `value` above is the headline)
        qp = lambda w=wl, **kw: (lambda: quick_point(w, rank, local, dev, world, **kw))
        points = {
            "K32_injected": qp(n_obj=32, inject=32, clip=256),
            "K256_injected": qp(n_obj=256, inject=256, B=32, clip=128, steps=24, warmup=3, full=True),
            # the engine mode whose CSV is identical to the oracle's (tests/test_gpu_bench_config.py): fp32 MFMA convs, same stream
            "s640_fp32_exact_csv": qp(precision="f32", B=64, clip=128, steps=12, full=True),
            # BASELINE.json configs[2] and configs[4], short runs of `--workload m1024-bf16` / `--workload l1280-fp8`
            "m1024_bf16": qp(WORKLOADS["m1024-bf16"], steps=16, warmup=2, full=True),
            "l1280_fp8": qp(WORKLOADS["l1280-fp8"], steps=16, warmup=2, full=True),
            # the same two precisions on the well-conditioned detector, its own detections all the way (no injection): the configuration
            # tests/test_gpu_coded.py holds to the oracle's CSV
            "s640_bf16_coded": qp(coded=True, inject=0, clip=256, steps=48, warmup=3, full=True),
            "l1280_fp8_coded": qp(WORKLOADS["l1280-fp8"], coded=True, inject=0, n_obj=12, steps=16, warmup=2, full=True),
            # BASELINE.json configs[3]'s 8 cameras on ONE GPU: 8 x 16 frames interleaved in every 128-frame batch (vc_stream_run_async_multi)
            "s640_8cam_one_gpu": qp(n_cam=8, clip=512, steps=48, warmup=3, full=True),
            "s640_8cam_distinct_clips": qp(n_cam=8, clip=512, steps=12, warmup=3, full=True, distinct_cams=True),
            # the boundary the reference itself calls: batch_size = 1 through the drop-in classes, host frames
            "dropin_bs1": lambda: dropin_point(wl, local, (640, 640), 256, ZONE),
            "dropin_bs1_720p": lambda: dropin_point(WL_720P, local, (720, 1280), 256, ZONE_720P),
            # the same loader and loop body with the stage calls asynchronous (CountingPipeline.run_pipelined): same rows, frame n+1's detector
            # behind frame n's ReID + tracker
            "dropin_bs1_pipelined": lambda: dropin_point(wl, local, (640, 640), 256, ZONE, pipelined=True),
            "dropin_bs1_720p_pipelined": lambda: dropin_point(WL_720P, local, (720, 1280), 256, ZONE_720P, pipelined=True),
            # the reference's only real input geometry (Q8, networks/yolo.py:69-70; demo/sample/cam_04.json): 1280 x 720 frames -> 384 x 640 tensor
            "s720p_bf16": qp(WL_720P, frame_hw=(720, 1280), zone=ZONE_720P, clip=256, steps=48, warmup=3, full=True),
            # the same frames with the clip's 12 ground-truth rectangles injected after the detector has run in full (the random head needs
            # obj_shift 7.8 at this geometry: thousands of candidates per frame reach the NMS and persistent noise boxes load the tracker --
            # artefacts of the synthetic head, not of the geometry)
            "s720p_bf16_K12_injected": qp(frame_hw=(720, 1280), zone=ZONE_720P, n_obj=12, inject=12, clip=256, steps=48, warmup=3, full=True),
            # batch-size sweep of the headline stream (frames per vc_stream_* call) with the submit -> rows latency of a batch
            "batch_sweep": lambda: {f"B{b}": quick_point(wl, rank, local, dev, world, B=b, clip=256, steps=max(8, min(128, 512 // b)), warmup=max(3, min(16, 64 // b)))
                                    for b in (1, 8, 16, 32, 128, 256)},
        }
        only = set(args.extras.split(",")) if args.extras else None
        out["extra_points"] = {k: f() for k, f in points.items() if only is None or k in only}
        if only is None or "host_frames" in only:
            hp = quick_point(wl, rank, local, dev, world, host=True, steps=30, warmup=6)     # the clip is 4 batches: every pinned page has crossed PCIe once before the timed steps
            out["value_host_frames"] = hp["value"]
            out["value_host_frames_note"] = "same workload with the frames in pinned host memory: every batch is copied over PCIe inside the timed region (vc_stream_stage_host one batch ahead of vc_stream_submit: the copy runs under the detector of the batch before)"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == "s640-bf16":
            out["cpu_baseline"] = cpu_baseline(ysd, rsd, frames, args.cpu_frames)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
