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
