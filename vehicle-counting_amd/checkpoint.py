"""Checkpoint ingestion without the upstream packages.

The reference loads its detector through `torch.hub.load('ultralytics/yolov5', ...)` (/root/reference/networks/yolo.py:58)
and its ReID net from `ckpt.t7['net_dict']` (/root/reference/networks/deepsort/deep/feature_extractor.py:13-14).  A
YOLOv5 v6.0 `.pt` file pickles a whole `models.yolo.Model` object; unpickling it normally needs the ultralytics/yolov5
source tree on sys.path.  Here every class the pickle names that is not importable is replaced by an empty
`torch.nn.Module` subclass: pickle restores an nn.Module by filling `__dict__` (`_parameters`, `_buffers`, `_modules`),
which is all `state_dict()` needs -- no upstream code runs.  The result is the fused {name.weight, name.bias} dict
`Engine` consumes (BatchNorm folded like upstream's `model.fuse()`).
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

from .weights import fold_yolo_state_dict, yolo_conv_table

_SAFE_PREFIXES = ("torch", "collections", "numpy", "builtins", "__builtin__", "_codecs", "copy_reg", "copyreg")


class _StubUnpickler(pickle.Unpickler):
    """Resolves torch / numpy / stdlib globals normally; anything else (models.yolo.Model, models.common.Conv, ...) becomes
    an attribute-bag nn.Module subclass named after the original class."""

    _made = {}

    def find_class(self, module, name):
        if module.split(".")[0] in _SAFE_PREFIXES:
            return super().find_class(module, name)          # incl. the protocol-2 names (__builtin__.set, copy_reg, ...)
        key = (module, name)
        if key not in self._made:
            self._made[key] = type(name, (torch.nn.Module,), {"__module__": module, "__init__": lambda self, *a, **k: torch.nn.Module.__init__(self),
                                                              "forward": lambda self, *a, **k: None})
        return self._made[key]


class _StubPickle:
    """The `pickle_module` interface torch.load expects."""
    Unpickler = _StubUnpickler
    load = staticmethod(lambda f, **kw: _StubUnpickler(f, **kw).load())
    __name__ = "pickle"


def _state_dict_of(obj):
    if isinstance(obj, torch.nn.Module):
        return {k: v.detach().float().cpu().numpy() for k, v in obj.state_dict().items()}
    if isinstance(obj, dict):
        return {k: (v.detach().float().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in obj.items()}
    raise TypeError(f"cannot take a state_dict from {type(obj).__name__}")


def load_yolov5_checkpoint(path, variant="yolov5s", nc=None):
    """`.pt` from the ultralytics/yolov5 v6.0 release (dict with 'model' / 'ema' holding a pickled Model, fp16 or fp32),
    or a plain state_dict file -> fused conv parameters {name+'.weight', name+'.bias'} (float32 numpy).  Checks the layer
    table of `variant` (names and shapes) so a wrong variant / class count fails here, not inside the engine."""
    ck = torch.load(path, map_location="cpu", pickle_module=_StubPickle, weights_only=False)
    if isinstance(ck, dict) and ("model" in ck or "ema" in ck):
        ck = ck.get("ema") or ck["model"]
    sd = fold_yolo_state_dict(_state_dict_of(ck))
    if nc is None:
        nc = sd["model.24.m.0.weight"].shape[0] // 3 - 5
    for name, ci, co, k in yolo_conv_table(variant, nc):
        w = sd.get(name + ".weight")
        if w is None or tuple(w.shape) != (co, ci, k, k):
            raise ValueError(f"{path}: {name}.weight is {None if w is None else tuple(w.shape)}, {variant} (nc={nc}) needs {(co, ci, k, k)}")
    return sd


def load_reid_checkpoint(path):
    """`ckpt.t7` of the reference's DeepSORT extractor: {'net_dict': state_dict} (feature_extractor.py:13-14), or a bare
    state_dict -> float32 numpy state_dict for `Engine(reid_sd=...)` (BatchNorm folded there, weights.fold_reid)."""
    ck = torch.load(path, map_location="cpu", pickle_module=_StubPickle, weights_only=False)
    if isinstance(ck, dict) and "net_dict" in ck:
        ck = ck["net_dict"]
    return _state_dict_of(ck)
