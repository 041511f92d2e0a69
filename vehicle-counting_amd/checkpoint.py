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

import io
import pickle

import numpy as np
import torch

from .weights import fold_yolo_state_dict, yolo_conv_table

# Exact globals a tensor / state_dict / nn.Module pickle needs.  Everything else -- including builtins.eval / exec / getattr,
# torch.hub.load, os.system ... -- is NOT resolved: it becomes an inert stub class, so a crafted checkpoint cannot run code.
_BUILTIN_OK = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice",
               "range", "object"}
_EXACT_OK = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("_codecs", "encode"),
    ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"),
    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.nn.parameter", "Parameter"), ("torch", "Tensor"), ("torch", "Size"), ("torch", "device"),
    ("torch.serialization", "_get_layout"), ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
}


def _allowed(module, name):
    if module in ("builtins", "__builtin__"):
        return name in _BUILTIN_OK
    if (module, name) in _EXACT_OK:
        return True
    if module == "torch":                                   # dtypes (torch.float16, ...) and the legacy typed storages
        return name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)
    if module.startswith("torch.nn.modules."):             # plain layers (Conv2d, BatchNorm2d, SiLU, Sequential, ...): classes only
        import importlib
        try:
            obj = getattr(importlib.import_module(module), name, None)
        except ImportError:
            return False
        return isinstance(obj, type) and issubclass(obj, torch.nn.Module)
    return False


def _load_from_bytes_restricted(b):
    """Stand-in for torch.storage._load_from_bytes (what a tensor pickled with plain `pickle.dumps` reduces to).  The original is
    `torch.load(io.BytesIO(b), weights_only=False)` with the UNRESTRICTED pickle module: an allowlisted call that would run any
    inner payload.  The nested blob goes back through the same stub unpickler instead."""
    return torch.load(io.BytesIO(b), map_location="cpu", pickle_module=_StubPickle, weights_only=False)


class _StubUnpickler(pickle.Unpickler):
    """Resolves only the allowlisted torch / numpy / stdlib globals above; anything else (models.yolo.Model,
    models.common.Conv, ... and anything hostile) becomes an attribute-bag nn.Module subclass named after the original."""

    _made = {}

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _load_from_bytes_restricted
        if _allowed(module, name):
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
    raw = _state_dict_of(ck)
    sd = fold_yolo_state_dict(raw)
    # Detect anchors: `anchors` is stored in stride units (models/yolo.py::Detect divides by the stride at build time); v6.0 also
    # keeps `anchor_grid` in pixels.  Custom (autoanchor) checkpoints differ from the COCO defaults, so they travel with the weights.
    if "model.24.anchor_grid" in raw and np.asarray(raw["model.24.anchor_grid"]).size == 18:
        sd["model.24.anchors_px"] = np.asarray(raw["model.24.anchor_grid"], np.float32).reshape(3, 6)
    elif "model.24.anchors" in raw:
        sd["model.24.anchors_px"] = (np.asarray(raw["model.24.anchors"], np.float32).reshape(3, 3, 2) *
                                     np.array([8.0, 16.0, 32.0], np.float32)[:, None, None]).reshape(3, 6)
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
