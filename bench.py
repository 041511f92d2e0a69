"""bench.py -- end-to-end frames/s of the detect + NMS + ReID + track hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of B = 64 synthetic 640x640 frames of ONE camera stream per GPU
(BASELINE.json configs[1]: YOLOv5s 640x640, bf16 convs): letterbox -> YOLOv5s conv stack -> decode -> NMS -> crops ->
ReID CNN -> per-class DeepSORT step, frames already resident in HBM.  Each rank owns its own camera stream (weak
scaling); the only collective is the all-gather of the per-camera count tensors at the end.
Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import math
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
from vehicle_counting_amd.counting import count_directions, csv_records  # noqa: E402
from vehicle_counting_amd.synth import synth_frames  # noqa: E402
from vehicle_counting_amd.track import VideoCounting  # noqa: E402
from vehicle_counting_amd.weights import synth_reid, synth_yolo  # noqa: E402

B = int(os.environ.get("VC_BENCH_B", 128))  # frames per step (one batch of the camera stream; 64 -> 128 -> 256: 11.0 -> 11.6 -> 11.8 k frames/s)
H = W = 640
NC = 80
N_OBJ = 12
CLIP = int(os.environ.get("VC_BENCH_CLIP", 512))   # distinct synthetic frames per stream (F = 512), cycled
ASYNC = os.environ.get("VC_BENCH_ASYNC", "1") != "0"     # tracker loop on the engine's worker thread (vc_stream_run_async)
TRACK = dict(max_dist=0.2, min_confidence=0.25, nms_max_overlap=0.5, max_iou_distance=0.6, max_age=30, n_init=3, nn_budget=60)
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0          # HBM3E (MI355X_MICROARCH.md)
ZONE = os.path.join(ROOT, "tests", "golden", "cam_04_halfres.json")


def cpu_baseline(ysd, rsd, frames, n_frames):
    """The oracle (CPU port of the reference path) timed on this box's host cores over a bounded sample."""
    from oracle import pipeline as op
    cfg = dict(MAX_DIST=0.2, MIN_CONFIDENCE=0.25, NMS_MAX_OVERLAP=0.5, MAX_IOU_DISTANCE=0.6, MAX_AGE=30, N_INIT=3, NN_BUDGET=60)
    op.run_video(frames[:1], ysd, rsd, cfg, ZONE, nc=NC)            # warm the CPU kernels
    t0 = time.perf_counter()
    _, _, nd = op.run_video(frames[:n_frames], ysd, rsd, cfg, ZONE, nc=NC)
    dt = time.perf_counter() - t0
    gc.enable()
    return {"value": n_frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {n_frames} frames of the rank-0 stream through oracle/pipeline.py (torch-CPU fp32 YOLOv5s + ReID, "
                      f"NumPy/SciPy DeepSORT), {int(np.mean(nd))} det/frame, {dt:.1f} s"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=12)
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

    ysd = synth_yolo("yolov5s", nc=NC, seed=1702, det_scale=4.0, obj_shift=1.0)
    rsd = synth_reid(1702)
    eng = E.Engine(ysd, rsd, device=local, precision="bf16", model_name="yolov5s", num_classes=NC, max_batch=B,
                   max_frame_hw=(H, W), max_crops=B * 64, max_tracks=8192, nn_budget_cap=60)
    eng.pretune((32, 64, 128, 256, 512, 1024, 2048, 4096))                      # conv autotune for every ReID size bucket
    trackers = [eng.tracker_create(**TRACK) for _ in range(NC)]
    frames = synth_frames(CLIP, H, W, n_obj=N_OBJ, seed=1702 + rank, bounce=True)          # one camera stream per rank
    d_frames = torch.from_numpy(frames).to(dev)                                 # resident in HBM before the timed region
    LOOP = math.lcm(CLIP, B)                  # the cycled clip laid out so that every batch is one contiguous run of frames
    if LOOP > CLIP:
        d_frames = d_frames.repeat(LOOP // CLIP, 1, 1, 1)
    ndet_total = [0, 0]

    def batch_ptr(i):
        f0 = (i * B) % LOOP
        return d_frames[f0:f0 + B].data_ptr()

    def step(i, record, prefetch=True):
        if prefetch:
            eng.stream_submit(batch_ptr(i + 1), B, H, W)      # detector of the NEXT batch runs on its own stream while this one is tracked
        rows, fidx, nd = eng.stream_run_packed(trackers, batch_ptr(i), B, H, W)
        if record:
            ndet_total[0] += int(nd.sum()); ndet_total[1] += B
            count_rows(i * B + 1, rows, fidx)

    counter = VideoCounting([str(c) for c in range(NC)], ZONE)
    nrows_total = [0]

    def count_rows(f0, rows, fidx):
        """VideoCounting's zone filter + per-track lists for one batch, on the host while the GPU works on the next batches."""
        nrows_total[0] += len(rows)
        counter.run((f0 + fidx).tolist(), rows[:, 4].tolist(), rows[:, 5].tolist(), np.ascontiguousarray(rows[:, :4]), finalize=False)

    step_marks = []

    def collect(i, record):
        rows, fidx, nd = eng.stream_collect()
        if record:
            step_marks.append(time.perf_counter())
            ndet_total[0] += int(nd.sum()); ndet_total[1] += B
            count_rows(i * B + 1, rows, fidx)

    def run_steps(first, n, record):
        """Three overlapped stages: detector of batch i+1 (own stream), ReID of batch i (own stream), tracker loop of batch
        i-1 (the engine's worker thread + tracker stream); rows come back one batch late.  Everything is collected before
        the function returns, so the timed region contains the whole work of its n steps."""
        if n <= 0:
            return
        if ASYNC:
            for i in range(first, first + n):
                eng.stream_submit(batch_ptr(i + 1), B, H, W)
                eng.stream_run_async(trackers, batch_ptr(i), B, H, W)
                if i > first:
                    collect(i - 1, record)
            collect(first + n - 1, record)
        else:
            for i in range(first, first + n):
                step(i, record)

    def sync_all():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    eng.stream_submit(batch_ptr(0), B, H, W)
    run_steps(0, args.warmup, False)
    sync_all()
    eng.profile_reset(); eng.profile(2)            # in-flight event pairs around every conv launch of the timed steps (no host waits)
    gc.collect(); gc.disable()                     # a generation-2 collection over the per-track box lists costs 40 ms when it lands in the serial tail
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, True)
    # end of run: per-camera counts (VideoCounting) merged with the one collective of the design
    t_post = time.perf_counter()
    tt = [time.perf_counter()]
    tt.append(time.perf_counter())
    td = counter.run([], [], [], np.zeros((0, 4), np.int64))      # every batch was appended as it was collected: directions only
    tt.append(time.perf_counter())
    rows = csv_records(td)
    tt.append(time.perf_counter())
    dirs = list(counter.directions.keys())
    local_counts = parallel.counts_to_tensor(count_directions(rows, dirs, NC), dirs, NC)[None]
    all_counts = parallel.allgather_counts(local_counts, device=dev if world > 1 else None)
    tt.append(time.perf_counter())
    sync_all()
    dt = time.perf_counter() - t0
    post_ms = (time.perf_counter() - t_post) * 1e3
    tt.append(time.perf_counter())
    if os.environ.get('VC_BENCH_DBG') and len(step_marks) > 1: print('ms between collects: ' + ' '.join('%.1f' % ((b - a) * 1e3) for a, b in zip([t0] + step_marks[:-1], step_marks)), file=sys.stderr)
    if os.environ.get('VC_BENCH_DBG'): print('post-pass ms: setup %.1f run %.1f csv %.1f count+gather %.1f final sync %.1f' % tuple((b - a) * 1e3 for a, b in zip(tt[:-1], tt[1:])), file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # roofline of the dominant kernel (the conv kernels): HIP events around every conv launch of the TIMED steps, recorded on
    # the stream the kernel is launched on and resolved after the region (the launches overlap with the other two streams, as in
    # the rocprofv3 trace of this command).  The per-stage split comes from two extra steps with blocking events.
    conv_timed = eng.profile_read(L.PROF_CONV)
    conv_union_ms, conv_span_ms = eng.profile_conv_busy()
    eng.profile(0)
    step(args.warmup + args.steps, False, prefetch=False)       # drain the submission left in flight
    eng.profile(True); eng.profile_reset()
    for i in range(2):
        step(args.warmup + args.steps + 1 + i, False, prefetch=False)
    eng.sync()
    conv = eng.profile_read(L.PROF_CONV)
    cats = {n: eng.profile_read(c) for n, c in (("conv", L.PROF_CONV), ("detect_aux", L.PROF_DETECT_AUX),
                                                ("reid_aux", L.PROF_REID_AUX), ("track", L.PROF_TRACK))}
    eng.profile(False)
    isolated = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0

    # HBM traffic of the conv kernels: rocprofv3 PMC passes cannot run inside this process; tools/pmc_traffic.py stores the
    # per-launch figure of the same command under profiles/ (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes)
    traffic, traffic_src = None, None
    for rnd in ("r02", "r01"):
        tp = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if os.path.exists(tp):
            with open(tp) as f:
                traffic, traffic_src = json.load(f)["conv_all"]["hbm_bytes_per_launch"], f"profiles/{rnd}_pmc_traffic.json"
            break

    # Roofline of the dominant kernels (all conv launches of a step), SURVEY.md 8(d): bound = whichever of
    # flops / peak_mfma and bytes / peak_hbm is larger.  Time base = the wall clock of the timed region (ms_per_step): conv
    # launches of the detector and ReID streams overlap each other, so a sum of per-launch durations counts that time twice
    # (kept below as achieved_sum_of_overlapped_durations); flops-per-step / ms_per_step can be re-derived from the driver's own
    # clock and from profiles/ (launches per step x rocprofv3 average duration is <= ms_per_step).
    step_s = dt / args.steps
    n_meas_steps = max(conv_timed["launches"] / max(conv["launches"] / 2.0, 1.0), 1e-9)      # timed steps covered by the event pool
    flops_step, bytes_step = conv_timed["flops"] / n_meas_steps, conv_timed["bytes"] / n_meas_steps
    mfma_tflops, hbm_gbs = flops_step / step_s / 1e12, bytes_step / step_s / 1e9
    mfma_frac, hbm_frac = mfma_tflops / PEAK_BF16_TFLOPS, hbm_gbs / PEAK_HBM_GBS
    overlapped = conv_timed["flops"] / (conv_timed["ms"] * 1e-3) / 1e12 if conv_timed["ms"] > 0 else 0.0
    hbm_bound = hbm_frac > mfma_frac
    roofline = {
        "bound": "hbm" if hbm_bound else "mfma",
        "achieved": hbm_gbs if hbm_bound else mfma_tflops, "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_TFLOPS,
        "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": max(hbm_frac, mfma_frac), "traffic": traffic,
        "mfma_frac": mfma_frac, "mfma_tflops": mfma_tflops, "hbm_frac": hbm_frac, "hbm_gbs": hbm_gbs,
        "time_base": "wall clock of the timed region: algorithmic conv work per step / ms_per_step",
        "kernel": "vc::conv_igemm_kernel<*> / conv3x3_halo_kernel<*> / conv1x1_direct_kernel<*> / stem_direct_kernel<*> / reid_stem_pool_kernel (all YOLOv5s + ReID conv launches of a step)",
        "launches_per_step": conv["launches"] / 2.0,
        "algorithmic_gflop_per_step": flops_step / 1e9, "algorithmic_bytes_per_launch": conv_timed["bytes"] / max(conv_timed["launches"], 1),
        "avg_launch_us": conv_timed["ms"] * 1e3 / max(conv_timed["launches"], 1),
        "avg_launch_note": "HIP start/stop timestamps of every conv dispatch of the timed steps (hipExtLaunchKernel); launches of the detector and ReID streams overlap, so launches_per_step x avg_launch_us may exceed ms_per_step",
        "achieved_sum_of_overlapped_durations_tflops": overlapped, "frac_sum_of_overlapped_durations": overlapped / PEAK_BF16_TFLOPS,
        "achieved_isolated_tflops": isolated,
        "conv_running_frac_of_timed_window": conv_union_ms / conv_span_ms if conv_span_ms > 0 else None,
        "timed_launches_measured": int(conv_timed["launches"]),     # capped by the engine's pool of event pairs
        "traffic_note": "HBM bytes per conv launch from %s (rocprofv3 --pmc FETCH_SIZE x2 (gfx950) + WRITE_SIZE, separate passes of this command)" % traffic_src,
    }

    if rank == 0:
        out = {
            "metric": "end-to-end frames/sec (detect+NMS+ReID+track), YOLOv5s 640px",
            "value": world * args.steps * B / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "YOLOv5s 640x640 single camera stream per GPU, bf16 convs (BASELINE.json configs[1])",
                       "frames_per_step": B, "frame_hw": [H, W], "num_classes": NC, "det_per_frame": ndet_total[0] / max(ndet_total[1], 1),
                       "weights": "seeded synthetic (no checkpoints available)", "streams": world,
                       "counts_allgather_shape": list(all_counts.shape), "tracked_rows": int(nrows_total[0]),
                       "counting_postpass_ms_total": post_ms},
            "roofline": roofline,
            "stage_ms_per_step": {k: v["ms"] / 2 for k, v in cats.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ysd, rsd, frames, args.cpu_frames)
        print(json.dumps(out))
    eng.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
