"""GPU: single-kernel parity through the C ABI (vc_*_host) against the oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from oracle import deepsort as od  # noqa: E402
from oracle import imageops as oi  # noqa: E402
from oracle import yolov5 as oy  # noqa: E402


def bf16(x):
    return torch.as_tensor(x).bfloat16().float()


def torch_conv(x_nhwc, w, b, stride, pad, act, res, res_mode, precision):
    x = torch.as_tensor(x_nhwc).permute(0, 3, 1, 2)
    w, b = torch.as_tensor(w), torch.as_tensor(b)
    r = None if res is None else torch.as_tensor(res).permute(0, 3, 1, 2)
    if precision == "bf16":
        x, w = bf16(x), bf16(w)
        r = None if r is None else bf16(r)
    y = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    if r is not None and res_mode == 2:
        y = y + r.double()
    y = F.silu(y) if act == 1 else (F.relu(y) if act == 2 else y)
    if r is not None and res_mode == 1:
        y = y + r.double()
    y = y.float()
    if precision == "bf16":
        y = bf16(y)
    return y.permute(0, 2, 3, 1).numpy()


# (B, H, W, Cin, Cout, k, stride, pad, act, res_mode)   -- SURVEY.md 7.2 representative layers + edge shapes
CONV_CASES = [
    (1, 40, 40, 256, 512, 3, 2, 1, 1, 0),      # 7.Conv   K=2304, M=400
    (1, 160, 160, 64, 32, 1, 1, 0, 1, 0),      # 1x1      K=64,   M=25600
    (2, 64, 64, 3, 32, 6, 2, 2, 1, 0),         # stem shape (generic channel-padded path)
    (2, 40, 40, 64, 64, 3, 1, 1, 1, 1),        # bottleneck: residual after SiLU
    (3, 25, 25, 64, 128, 3, 2, 1, 2, 0),       # ReID downsample conv1 (odd spatial)
    (3, 13, 13, 128, 128, 3, 1, 1, 2, 2),      # ReID conv2: residual before ReLU
    (3, 25, 25, 64, 128, 1, 2, 0, 0, 0),       # ReID 1x1 stride-2 shortcut, no activation
    (1, 20, 20, 512, 255, 1, 1, 0, 0, 0),      # Detect head: Cout not a multiple of 4
    (1, 7, 9, 48, 96, 3, 1, 1, 1, 0),          # yolov5m-like channel counts (Cin not a power of two), ragged M
    (5, 50, 50, 3, 64, 3, 1, 1, 2, 0),         # ReID stem
]


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case, precision):
    B, H, W, Ci, Co, k, s, p, act, rm = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
    w = (rng.standard_normal((Co, Ci, k, k), dtype=np.float32) / np.sqrt(Ci * k * k)).astype(np.float32)
    b = rng.standard_normal(Co, dtype=np.float32) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rng.standard_normal((B, Ho, Wo, Co), dtype=np.float32) if rm else None
    y = E.conv2d(x, w, b, stride=s, pad=p, act=act, res=res, res_mode=rm, precision=precision)
    ref = torch_conv(x, w, b, s, p, act, res, rm, precision)
    assert y.shape == ref.shape
    if precision == "f32":
        # tolerance (SURVEY.md 8d ladder step 1): fp32 MFMA fmaf chain vs fp64-accumulated reference
        np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
    else:
        # identical bf16-rounded operands, fp32 accumulate: at most one bf16 ulp (2^-8 relative) on the stored result
        np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("slots", [0, 8])
@pytest.mark.parametrize("cfg", list(range(32)) + list(range(36, 50)) + [55] + list(range(60, 69)))     # (56 - 59, split-K: tests/test_gpu_round6.py; 64 - 68: paired workgroups)
def test_conv2d_every_tile_config(cfg, slots, monkeypatch):
    """Every entry of conv_igemm.hip's tile table (tile shape x K chunk x ring depth) on a padded 3x3 with a ragged
    pixel tail, a ragged channel tail and a K extent shorter than the deepest ring, and on a strided 1x1."""
    s2halo = 44 <= cfg <= 49
    v2 = cfg == 55                                               # conv3x3_halo_v2_kernel (conv_halo_v2.hip)
    halo = cfg in (28, 29, 30, 31, 36, 37, 38, 39) or s2halo or v2   # 40-43 are implicit-GEMM tiles with 16 waves per workgroup
    monkeypatch.setenv("VC_CONV_CFG", str(cfg))
    if slots:
        if halo:
            pytest.skip("the halo variants are not persistent")
        monkeypatch.setenv("VC_CONV_SLOTS", str(slots))       # 8 persistent workgroups walk all tiles: the K ring crosses tile boundaries
    cases = [(2, 23, 19, 24, 72, 3, 1, 1, 1, 1), (1, 9, 9, 8, 130, 3, 1, 1, 0, 0), (3, 20, 20, 136, 40, 1, 2, 0, 1, 2)]
    if halo:
        # halo-staged 3x3 / s1 / p1 variants (bf16 path; Cin a multiple of 32): ragged tiles, patches that cross the batch seam
        # (7x7 and 4x5 maps: one tile spans several images), residual before / after the activation, 2 and 4 channel slices
        cases = [(2, 23, 19, 64, 72, 3, 1, 1, 1, 1), (5, 7, 7, 32, 40, 3, 1, 1, 2, 2), (9, 4, 5, 128, 130, 3, 1, 1, 1, 0),
                 (1, 40, 160, 32, 64, 3, 1, 1, 1, 0)]
        if cfg >= 38:
            cases[-1] = (1, 40, 96, 32, 64, 3, 1, 1, 1, 0)       # 256-pixel tiles: five rows of 160 pixels exceed the largest patch buffer
    if s2halo:
        # halo-staged 3x3 / s2 / p1 (rectangular tiles, patch staged by parity class; Cin a multiple of 64): 20 x 16, 4 x 3 (a tile spans
        # several images), the 80-column geometry of YOLO layer 3 (8 x 16 tiles), two channel groups with a ragged channel tail, and a
        # 20-column map (25 x 5 tiles across images) with three groups
        cases = [(2, 40, 32, 64, 72, 3, 2, 1, 1, 0), (5, 8, 6, 64, 40, 3, 2, 1, 2, 0), (1, 160, 160, 64, 128, 3, 2, 1, 1, 0),
                 (3, 20, 24, 128, 136, 3, 2, 1, 0, 0), (2, 40, 40, 192, 64, 3, 2, 1, 1, 0)]
    if v2:
        # row-aligned tiles (R rows with R * W <= 256 and a patch of at most 320 pixels; Cin a multiple of 64): 13 rows of 19 with a ragged
        # last tile and a ragged channel tail, one tile that holds five whole 7 x 7 images, 4 x 5 images (a tile spans nine), the 40 x 40
        # geometry of the detector (6 rows per tile, 8-row patch = exactly 320 pixels) with four slices and two channel tiles, residual
        # before / after the activation
        cases = [(2, 23, 19, 64, 72, 3, 1, 1, 1, 1), (5, 7, 7, 64, 40, 3, 1, 1, 2, 2), (9, 4, 5, 128, 130, 3, 1, 1, 1, 0),
                 (3, 40, 40, 128, 256, 3, 1, 1, 1, 2), (7, 13, 13, 192, 128, 3, 1, 1, 2, 1)]
    for case in cases:
        B, H, W, Ci, Co, k, s, p, act, rm = case
        rng = np.random.default_rng(cfg * 131 + H)
        x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
        w = (rng.standard_normal((Co, Ci, k, k), dtype=np.float32) / np.sqrt(Ci * k * k)).astype(np.float32)
        b = rng.standard_normal(Co, dtype=np.float32) * 0.1
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        res = rng.standard_normal((B, Ho, Wo, Co), dtype=np.float32) if rm else None
        for precision in (("bf16",) if halo else ("bf16", "f32")):
            y = E.conv2d(x, w, b, stride=s, pad=p, act=act, res=res, res_mode=rm, precision=precision)
            ref = torch_conv(x, w, b, s, p, act, res, rm, precision)
            if precision == "f32":
                np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4)
            else:
                np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("slots", [0, 8])
@pytest.mark.parametrize("cfg,ci", [(32, 32), (33, 64), (34, 64), (35, 128)])
def test_conv1x1_direct_configs(cfg, ci, slots, monkeypatch):
    """Weights-in-registers 1x1 variants (conv1x1_direct_kernel): 1, 2 and 4 channel groups, ragged pixel tails, SiLU and no
    activation, a short persistent grid; against the torch reference and bit for bit against the implicit-GEMM kernel."""
    for co in ((32, 64, 128) if cfg == 32 else (64, 128, 256)):
        for (B, H, W) in ((3, 7, 9), (2, 40, 41)):
            for act in (1, 0):
                rng = np.random.default_rng(cfg * 131 + co + H)
                x = rng.standard_normal((B, H, W, ci), dtype=np.float32)
                w = (rng.standard_normal((co, ci, 1, 1), dtype=np.float32) / np.sqrt(ci)).astype(np.float32)
                b = rng.standard_normal(co, dtype=np.float32) * 0.1
                monkeypatch.setenv("VC_CONV_CFG", str(cfg))
                if slots:
                    monkeypatch.setenv("VC_CONV_SLOTS", str(slots))
                y = E.conv2d(x, w, b, stride=1, pad=0, act=act, precision="bf16")
                monkeypatch.delenv("VC_CONV_SLOTS", raising=False)
                monkeypatch.setenv("VC_CONV_CFG", "3")
                y_igemm = E.conv2d(x, w, b, stride=1, pad=0, act=act, precision="bf16")
                ref = torch_conv(x, w, b, 1, 0, act, None, 0, "bf16")
                np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2e-3)
                np.testing.assert_array_equal(y, y_igemm)


@pytest.mark.parametrize("slots", [0, 8])
@pytest.mark.parametrize("cfg,ci,co", [(50, 128, 128), (51, 256, 256), (52, 256, 128), (53, 128, 256), (54, 128, 128)])
def test_conv1x1_stream_configs(cfg, ci, co, slots, monkeypatch):
    """Weights-in-LDS streaming 1x1 variants (conv1x1_stream_kernel: a wave owns every output channel of its pixels, the pixel operand
    comes straight from global memory): ragged pixel tails, fewer blocks than waves, SiLU and no activation, a short persistent grid;
    against the torch reference and bit for bit against the implicit-GEMM kernel."""
    for (B, H, W) in ((3, 7, 9), (2, 40, 41), (1, 3, 5)):
        for act in (1, 0):
            rng = np.random.default_rng(cfg * 131 + co + H)
            x = rng.standard_normal((B, H, W, ci), dtype=np.float32)
            w = (rng.standard_normal((co, ci, 1, 1), dtype=np.float32) / np.sqrt(ci)).astype(np.float32)
            b = rng.standard_normal(co, dtype=np.float32) * 0.1
            monkeypatch.setenv("VC_CONV_CFG", str(cfg))
            if slots:
                monkeypatch.setenv("VC_CONV_SLOTS", str(slots))
            y = E.conv2d(x, w, b, stride=1, pad=0, act=act, precision="bf16")
            monkeypatch.delenv("VC_CONV_SLOTS", raising=False)
            monkeypatch.setenv("VC_CONV_CFG", "3")
            y_igemm = E.conv2d(x, w, b, stride=1, pad=0, act=act, precision="bf16")
            ref = torch_conv(x, w, b, 1, 0, act, None, 0, "bf16")
            np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=2e-3)
            np.testing.assert_array_equal(y, y_igemm)


@pytest.mark.parametrize("hw", [(720, 1280, 384, 640), (333, 500, 448, 640), (640, 640, 640, 640), (100, 60, 640, 384)])
def test_letterbox(hw):
    h, w, nh, nw = hw
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = E.letterbox(img, nh, nw, "f32")
    ref = oi.letterbox(img, nh, nw).astype(np.float32) / np.float32(255)
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("n", [0, 1, 5, 200, 1500])
def test_nms(n):
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 600, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 200, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    conf = rng.uniform(0.25, 1, n).astype(np.float32)
    cls = rng.integers(0, 3, n).astype(np.int32)
    if n >= 5:
        conf[3] = conf[4]                  # tie -> stable order
        boxes[1] = boxes[0]                # duplicate
        cls[1] = cls[0]
    got = E.nms(boxes, conf, cls, iou=0.45, max_det=300, max_cand=2048)
    off = (cls[:, None].astype(np.float32) * np.float32(oy.MAX_WH)).astype(np.float32)
    keep = oy.box_iou_greedy_nms((boxes + off).astype(np.float32), conf, 0.45)[:300]
    ref = np.concatenate([boxes[keep], conf[keep, None], cls[keep, None].astype(np.float32)], 1) if n else np.zeros((0, 6), np.float32)
    np.testing.assert_array_equal(got, ref)


def test_kalman_kats(golden_dir):
    g = np.load(os.path.join(golden_dir, "kalman.npz"))
    m, c = E.kalman_initiate(g["meas"])
    np.testing.assert_array_equal(m, g["init_m"])
    np.testing.assert_array_equal(c, g["init_c"])
    # predict is exact (F is 0/1): check one step from the golden initial states against the oracle
    kf = od.KalmanCV()
    m1, c1 = E.kalman_predict(g["init_m"], g["init_c"])
    for i in range(len(m1)):
        om, oc = kf.predict(g["init_m"][i], g["init_c"][i])
        np.testing.assert_array_equal(m1[i], om)
        np.testing.assert_array_equal(c1[i], oc)
    # update / gating: LAPACK's operation order is not reproducible bit for bit -> 1e-9
    um, uc = E.kalman_update(g["pred_m"], g["pred_c"], g["zs"])
    np.testing.assert_allclose(um, g["upd_m"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(uc, g["upd_c"], rtol=1e-9, atol=1e-9)
    for i in range(len(g["gate"])):
        gd = E.kalman_gating(g["pred_m"][i], g["pred_c"][i], g["gate_in"][i])
        np.testing.assert_allclose(gd, g["gate"][i], rtol=1e-9, atol=1e-12)


def test_iou_and_cosine(golden_dir):
    g = np.load(os.path.join(golden_dir, "iou_cost.npz"))
    for ci in range(4):
        np.testing.assert_array_equal(E.iou_matrix(g[f"c{ci}_a"], g[f"c{ci}_b"]), g[f"c{ci}_iou"])
    g = np.load(os.path.join(golden_dir, "cosine.npz"))
    cost = E.cosine_cost([g["gallery1"], g["gallery3"]], g["query"])
    np.testing.assert_allclose(cost, g["cost"], rtol=0, atol=2e-6)     # f32 dot products, free summation order
