"""Checkpoint ingestion: a YOLOv5-style `.pt` that pickles classes from a package that is NOT importable
at load time, and a `ckpt.t7`-style ReID file, go through the stub unpickler and come out as the engine's parameter dict."""
import sys
import types

import numpy as np
import pytest
import torch

from vehicle_counting_amd.checkpoint import load_reid_checkpoint, load_yolov5_checkpoint
from vehicle_counting_amd.weights import YOLO_BN_EPS, fold_bn, synth_reid, yolo_conv_table


def _build_fake_upstream_model(nc, seed):
    """A module tree with upstream's parameter names (model.N....conv / .bn, model.24.m.i), built from classes living in
    throw-away `models.common` / `models.yolo` modules -- exactly what an ultralytics .pt pickles."""
    common, yolo = types.ModuleType("models.common"), types.ModuleType("models.yolo")
    pkg = types.ModuleType("models")
    sys.modules.update({"models": pkg, "models.common": common, "models.yolo": yolo})

    def cls(mod, name):
        c = type(name, (torch.nn.Module,), {"__module__": mod.__name__})
        setattr(mod, name, c)
        return c

    Conv, Block, Model, Detect = cls(common, "Conv"), cls(common, "C3"), cls(yolo, "Model"), cls(yolo, "Detect")
    g = torch.Generator().manual_seed(seed)
    root = Model()
    expect = {}
    for name, ci, co, k in yolo_conv_table("yolov5s", nc):
        parts = name.split(".")
        node = root
        for p in parts[:-1]:                     # e.g. model.2.m.0.cv1.conv -> containers model, 2, m, 0, cv1
            if p not in node._modules:
                node.add_module(p, Conv() if p.startswith("cv") else Block())
            node = node._modules[p]
        if name.startswith("model.24."):
            conv = torch.nn.Conv2d(ci, co, 1)
            conv.weight.data = torch.randn(conv.weight.shape, generator=g) * 0.1
            conv.bias.data = torch.randn(co, generator=g)
            if "m" not in root._modules["model"]._modules["24"]._modules:
                pass
            node.add_module(parts[-1], conv)
            expect[name] = (conv.weight.detach().numpy().copy(), conv.bias.detach().numpy().copy())
            continue
        conv = torch.nn.Conv2d(ci, co, k, bias=False)
        conv.weight.data = torch.randn(conv.weight.shape, generator=g) * 0.1
        bn = torch.nn.BatchNorm2d(co, eps=YOLO_BN_EPS)
        bn.weight.data = torch.rand(co, generator=g) + 0.5
        bn.bias.data = torch.randn(co, generator=g) * 0.1
        bn.running_mean = torch.randn(co, generator=g) * 0.1
        bn.running_var = torch.rand(co, generator=g) + 0.5
        node.add_module("conv", conv)
        node.add_module("bn", bn)
        expect[name] = fold_bn(conv.weight.detach().numpy(), None, bn.weight.detach().numpy(), bn.bias.detach().numpy(),
                               bn.running_mean.numpy(), bn.running_var.numpy(), YOLO_BN_EPS)
    return root, expect


def test_yolov5_pt_without_upstream_package(tmp_path):
    nc = 8
    model, expect = _build_fake_upstream_model(nc, 3)
    path = tmp_path / "yolov5s.pt"
    torch.save({"epoch": -1, "model": model.half(), "ema": None, "optimizer": None}, path)     # upstream ships fp16
    for m in ("models", "models.common", "models.yolo"):
        sys.modules.pop(m)                                                                       # the loader must not need them
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=False)                                 # the plain loader does
    sd = load_yolov5_checkpoint(path, "yolov5s")
    assert len(sd) == 2 * len(expect)
    for name, (w, b) in expect.items():
        half = lambda a: torch.from_numpy(np.asarray(a, np.float32)).half().float().numpy()     # the file stores fp16
        if name.startswith("model.24."):
            np.testing.assert_array_equal(sd[name + ".weight"], half(w))
            np.testing.assert_array_equal(sd[name + ".bias"], half(b))
        else:
            assert sd[name + ".weight"].shape == w.shape and sd[name + ".bias"].shape == b.shape
            np.testing.assert_allclose(sd[name + ".weight"], w, rtol=2e-3, atol=2e-4)           # fold of fp16-rounded parts
            np.testing.assert_allclose(sd[name + ".bias"], b, rtol=2e-3, atol=2e-3)
    with pytest.raises(ValueError):
        load_yolov5_checkpoint(path, "yolov5m")


def test_reid_ckpt_t7(tmp_path):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_reid(5).items()}
    path = tmp_path / "ckpt.t7"
    torch.save({"net_dict": sd, "acc": 0.9, "epoch": 40}, path)
    got = load_reid_checkpoint(path)
    assert set(got) == set(sd)
    for k in sd:
        np.testing.assert_array_equal(got[k], sd[k].float().numpy())


def test_detect_anchors_travel_with_the_checkpoint(tmp_path):
    """Custom (autoanchor) checkpoints carry their own Detect anchors: `model.24.anchors` is in stride units."""
    nc = 8
    model, _ = _build_fake_upstream_model(nc, 4)
    det = model._modules["model"]._modules["24"]
    anchors_px = np.array([[9, 11, 21, 19, 17, 41], [43, 32, 39, 70, 86, 64], [65, 131, 134, 130, 300, 290]], np.float32)
    det.register_buffer("anchors", torch.from_numpy(anchors_px.reshape(3, 3, 2) / np.array([8, 16, 32], np.float32)[:, None, None]))
    path = tmp_path / "custom.pt"
    torch.save({"model": model}, path)
    for m in ("models", "models.common", "models.yolo"):
        sys.modules.pop(m)
    sd = load_yolov5_checkpoint(path, "yolov5s")
    np.testing.assert_allclose(sd["model.24.anchors_px"], anchors_px, rtol=1e-6)


class _Hostile:
    def __reduce__(self):
        import builtins
        return (builtins.eval, ("__import__('os').environ.__setitem__('VC_PWNED', '1')",))


def test_hostile_pickle_does_not_execute(tmp_path):
    """builtins.eval / exec / getattr, os.system, torch.hub.load are not on the allowlist: they resolve to inert stubs."""
    import os
    import pickle
    from vehicle_counting_amd.checkpoint import _StubUnpickler, _allowed
    for mod, name in (("builtins", "eval"), ("builtins", "exec"), ("builtins", "getattr"), ("os", "system"), ("posix", "system"),
                      ("torch.hub", "load"), ("subprocess", "Popen"), ("torch", "load"), ("numpy", "load")):
        assert not _allowed(mod, name), (mod, name)
    for mod, name in (("collections", "OrderedDict"), ("torch._utils", "_rebuild_tensor_v2"), ("torch", "HalfStorage"),
                      ("torch", "float16"), ("torch.nn.modules.conv", "Conv2d"), ("builtins", "set")):
        assert _allowed(mod, name), (mod, name)
    path = tmp_path / "evil.pkl"
    with open(path, "wb") as f:
        pickle.dump({"model": _Hostile()}, f)
    os.environ.pop("VC_PWNED", None)
    with open(path, "rb") as f:
        obj = _StubUnpickler(f).load()
    assert "VC_PWNED" not in os.environ
    assert isinstance(obj["model"], torch.nn.Module)             # the eval call became the construction of a stub module


class _NestedHostile:
    """Outer pickle: REDUCE torch.storage._load_from_bytes(<inner torch.save blob whose pickle calls builtins.eval>).  The real
    _load_from_bytes is torch.load(..., weights_only=False) with the default pickle module."""

    def __init__(self, inner):
        self.inner = inner

    def __reduce__(self):
        return (torch.storage._load_from_bytes, (self.inner,))


def test_nested_load_from_bytes_payload_does_not_execute(tmp_path):
    import io
    import os
    from vehicle_counting_amd.checkpoint import _allowed
    assert not _allowed("torch.storage", "_load_from_bytes")
    buf = io.BytesIO()
    torch.save({"w": _Hostile()}, buf)
    path = tmp_path / "nested.pt"
    torch.save({"net_dict": _NestedHostile(buf.getvalue())}, path)
    os.environ.pop("VC_PWNED", None)
    try:
        load_reid_checkpoint(path)
    except Exception:
        pass                                                    # a TypeError about the stub is fine; running the payload is not
    assert "VC_PWNED" not in os.environ
    # a tensor pickled with plain pickle (the legitimate user of _load_from_bytes) still loads
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    path2 = tmp_path / "plain.pt"
    torch.save({"net_dict": {"a": _NestedHostile(_tensor_blob(t))}}, path2)
    got = load_reid_checkpoint(path2)
    np.testing.assert_array_equal(got["a"], t.numpy())


def _tensor_blob(t):
    import io
    b = io.BytesIO()
    torch.save(t, b)
    return b.getvalue()


def test_imagedetect_refuses_to_run_without_weights():
    import types
    from vehicle_counting_amd.detect import ImageDetect
    cfg = types.SimpleNamespace(model_name="yolov5s", min_conf=0.25, min_iou=0.45, max_det=300)
    with pytest.raises(ValueError, match="no --weight"):
        ImageDetect(types.SimpleNamespace(weight=None, mapping=None), cfg)
