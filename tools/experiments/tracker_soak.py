"""Device-resident DeepSORT against the oracle's TrackerState on seeded random scenes with random tracker parameters.
VC_SOAK_SEEDS=a,b (range), VC_SOAK_ARENA=0 for the in-walk instance.  Prints SOAK_OK or the first divergence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import vehicle_counting_amd.engine as E
from oracle import deepsort as od
from vehicle_counting_amd.weights import synth_reid

STATE = {od.TENTATIVE: 1, od.CONFIRMED: 2} if hasattr(od, "TENTATIVE") else None


def scene(seed):
    """seed >= 1000: "stale" scenes (round 6) -- 12 to 40 objects that are all seen for a while and then leave one by one while a few
    stay and new ones arrive, long max_age: most cascade levels of a step hold only tracks whose objects are gone (the rotation
    shortcut of match_step_wave64 / min_cost_matching), and the order of the remaining detections decides the ids of the new tracks."""
    rng = np.random.default_rng(seed)
    stale = seed >= 1000
    n = int(rng.integers(12, 40)) if stale else int(rng.integers(2, 70)); T = int(rng.integers(40, 60)) if stale else int(rng.integers(15, 50))
    p = dict(max_dist=float(rng.uniform(0.05, 0.35)), max_iou_distance=float(rng.uniform(0.4, 0.95)), max_age=int(rng.integers(20, 40) if stale else rng.integers(1, 40)),
             n_init=int(rng.integers(1, 5)), budget=int(rng.choice([1, 2, 5, 30, 60])))
    protos = rng.standard_normal((n, 512)).astype(np.float32); protos /= np.linalg.norm(protos, axis=1, keepdims=True)
    twins = rng.random(n) < 0.2                                    # look-alikes: appearance ambiguous, motion decides
    for i in np.nonzero(twins)[0][1:]: protos[i] = protos[np.nonzero(twins)[0][0]]
    pos = rng.uniform([50, 50], [1200, 650], (n, 2)); vel = rng.uniform(-9, 9, (n, 2)); wh = rng.uniform([20, 20], [140, 170], (n, 2))
    born = rng.integers(0, T // 2 + 1, n); dies = born + rng.integers(3, T + 5, n); pvis = rng.uniform(0.6, 1.0)
    if stale:
        late = rng.random(n) < 0.25                                # a quarter arrives while the others are leaving
        born = np.where(late, rng.integers(T // 3, T - 5, n), rng.integers(0, 4, n))
        dies = np.where(late | (rng.random(n) < 0.15), T + 1, rng.integers(8, T - 5, n))
        pvis = rng.uniform(0.9, 1.0)
    noise = float(rng.choice([0.005, 0.02, 0.06]))
    frames = []
    for t in range(T):
        dets = []
        for i in range(n):
            if not (born[i] <= t < dies[i]) or rng.random() > pvis: continue
            c = pos[i] + vel[i] * t + rng.normal(0, 0.7, 2); s = wh[i] * (1 + rng.normal(0, 0.02, 2))
            f = protos[i] + noise * rng.standard_normal(512).astype(np.float32); f = (f / np.linalg.norm(f)).astype(np.float32)
            dets.append({"tlwh": np.concatenate([c - s / 2, s]), "conf": float(rng.uniform(0.3, 0.99)), "feature": f})
        for _ in range(int(rng.integers(0, 3))):                   # clutter
            f = rng.standard_normal(512).astype(np.float32); f /= np.linalg.norm(f)
            dets.append({"tlwh": np.concatenate([rng.uniform([0, 0], [1200, 650]), rng.uniform([20, 20], [120, 150])]), "conf": float(rng.uniform(0.3, 0.9)), "feature": f})
        if t == T // 3: dets = []                                  # an empty frame (Q1 does not apply at this level: predict + age only)
        order = rng.permutation(len(dets)); frames.append([dets[k] for k in order])
    return p, frames


def run(eng, seed):
    p, frames = scene(seed)
    ref = od.TrackerState(p["max_dist"], p["budget"], max_iou_distance=p["max_iou_distance"], max_age=p["max_age"], n_init=p["n_init"])
    tid = eng.tracker_create(max_dist=p["max_dist"], max_iou_distance=p["max_iou_distance"], max_age=p["max_age"], n_init=p["n_init"], nn_budget=p["budget"])
    for t, dets in enumerate(frames):
        ref.predict(); ref.update(dets)
        eng.tracker_step(tid, np.array([d["tlwh"] for d in dets]).reshape(-1, 4), np.array([d["conf"] for d in dets]),
                         np.array([d["feature"] for d in dets], dtype=np.float32).reshape(-1, 512))
        s = eng.tracker_state(tid, with_cov=False)
        ids = np.array([k.tid for k in ref.tracks], np.int64)
        ok = (np.array_equal(s["ids"], ids) and np.array_equal(s["hits"], [k.hits for k in ref.tracks]) and np.array_equal(s["age"], [k.age for k in ref.tracks])
              and np.array_equal(s["tsu"], [k.tsu for k in ref.tracks]) and np.array_equal(s["state"], [k.state for k in ref.tracks]))
        if ok and len(ids): ok = np.allclose(s["mean"], np.array([k.mean for k in ref.tracks]), rtol=1e-9, atol=1e-9)
        if ok: ok = np.array_equal(s["gallery"], [len(ref.gallery.get(k.tid, [])) for k in ref.tracks])
        if not ok:
            print("DIVERGED seed", seed, "frame", t, p, "tracks", len(ids), "dets", len(dets)); eng.tracker_destroy(tid); return False
    eng.tracker_destroy(tid)
    return True


if __name__ == "__main__":
    a, b = (int(v) for v in os.environ.get("VC_SOAK_SEEDS", "0,20").split(","))
    eng = E.Engine(None, synth_reid(1702), precision="f32", max_crops=128, max_frame_hw=(720, 1280), max_tracks=512, nn_budget_cap=60)
    if os.environ.get("VC_SOAK_ARENA") == "0": eng.set_option("dot_arena_mb", 0)
    bad = [s for s in range(a, b) if not run(eng, s)]
    print("SOAK_OK" if not bad else "SOAK_FAILED", b - a, "scenes", bad)
