"""GPU: round-6 kernels.

  * split-K instances of the implicit GEMM (tile configurations 56 - 59, conv_igemm_kernel<..., SK>): the small-M layers of a batch-1 pass
    against a float64 reference on the bf16-rounded operands (the ladder of tests/test_gpu_kernels.py::test_conv2d), run to run bit for
    bit (the partial sums are added in split order whichever workgroup arrives last), and again after other launches have used the same
    workspace (the tickets are left at zero)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vehicle_counting_amd.engine as E  # noqa: E402
from test_gpu_kernels import torch_conv  # noqa: E402

# (B, H, W, Cin, Cout, k, stride, pad, act, res_mode): few output tiles, long K
SK_CASES = [
    (1, 20, 20, 256, 256, 3, 1, 1, 1, 1),       # YOLOv5s Bottleneck 3x3 at 20 x 20, one frame: K = 2304, residual after SiLU
    (2, 7, 7, 256, 256, 3, 1, 1, 2, 2),         # ReID layer3 block, two crops: residual before ReLU, a tile spans both images
    (3, 4, 4, 512, 512, 3, 1, 1, 2, 0),         # ReID layer4: M = 48, K = 4608
    (1, 20, 20, 1024, 512, 1, 1, 0, 1, 0),      # SPPF.cv2: pointwise, K = 1024
    (1, 40, 40, 128, 256, 3, 2, 1, 1, 0),       # stride 2
    (1, 9, 7, 24, 72, 5, 1, 2, 0, 0),           # Cin not a multiple of the K tile (per-lane tap walk starting inside the K range), 5 x 5, ragged tails, no activation
    (1, 20, 20, 512, 255, 1, 1, 0, 0, 0),       # Detect head: Cout not a multiple of 4
]


@pytest.mark.parametrize("cfg", [56, 57, 58, 59])
@pytest.mark.parametrize("case", SK_CASES)
def test_split_k_tiles(case, cfg, monkeypatch):
    B, H, W, Ci, Co, k, s, p, act, rm = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
    w = (rng.standard_normal((Co, Ci, k, k), dtype=np.float32) / np.sqrt(Ci * k * k)).astype(np.float32)
    b = rng.standard_normal(Co, dtype=np.float32) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rng.standard_normal((B, Ho, Wo, Co), dtype=np.float32) if rm else None
    ref = torch_conv(x, w, b, s, p, act, res, rm, "bf16")
    monkeypatch.setenv("VC_CONV_CFG", str(cfg))
    monkeypatch.setenv("VC_CONV_STRICT", "1")                     # no fall-back to the heuristic tile: the split-K kernel itself has to take every case
    y = [E.conv2d(x, w, b, stride=s, pad=p, act=act, res=res, res_mode=rm, precision="bf16") for _ in range(3)]
    np.testing.assert_allclose(y[0], ref, rtol=2 ** -7, atol=2e-3)
    np.testing.assert_array_equal(y[0], y[1])
    np.testing.assert_array_equal(y[0], y[2])
    monkeypatch.setenv("VC_CONV_CFG", "15")                       # the same tile without the split: equal up to the order of the fp32 sums
    np.testing.assert_allclose(E.conv2d(x, w, b, stride=s, pad=p, act=act, res=res, res_mode=rm, precision="bf16"), y[0], rtol=2 ** -7, atol=2e-3)


def test_split_k_workspace_reuse_is_clean(monkeypatch):
    """The split-K workspace is one allocation per stream that every split-K launch reuses: the same two convolutions with different
    data, alternating 150 times, must give their own bits every time (a partial sum read from a stale cache line of the other
    launch would not)."""
    B, H, W, Ci, Co, k = 2, 7, 7, 256, 256, 3
    rng = np.random.default_rng(7)
    data = []
    for _ in range(2):
        x = rng.standard_normal((B, H, W, Ci), dtype=np.float32)
        w = (rng.standard_normal((Co, Ci, k, k), dtype=np.float32) / np.sqrt(Ci * k * k)).astype(np.float32)
        data.append((x, w, rng.standard_normal(Co, dtype=np.float32) * 0.1))
    monkeypatch.setenv("VC_CONV_CFG", "57")
    monkeypatch.setenv("VC_CONV_STRICT", "1")
    first = [E.conv2d(x, w, b, stride=1, pad=1, act=1, precision="bf16") for x, w, b in data]
    assert not np.array_equal(first[0], first[1])
    for it in range(150):
        for i, (x, w, b) in enumerate(data):
            np.testing.assert_array_equal(E.conv2d(x, w, b, stride=1, pad=1, act=1, precision="bf16"), first[i], err_msg=f"iteration {it}, case {i}")
