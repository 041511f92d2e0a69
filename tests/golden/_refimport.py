"""Import shims for the read-only reference tree (generation time only; never on the GPU box).

The reference (kaylode/vehicle-counting) needs NumPy<1.24 aliases and imports cv2 in a few
modules where it is unused for the functions we call.  We import the
individual modules by path so the heavyweight package __init__ files are never executed.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = os.environ.get("VC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True  # the reference tree is read-only


def _shim():
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    if "cv2" not in sys.modules:
        class _Cv2Stub(types.ModuleType):
            """cv2 is absent; the reference only touches cv2 constants at import time on our path."""
            def __getattr__(self, name):
                if name.startswith("__"):
                    raise AttributeError(name)
                return 0
        sys.modules["cv2"] = _Cv2Stub("cv2")


def load_sort():
    """Return the reference's `sort` package (tracker, kalman_filter, ...)."""
    _shim()
    p = os.path.join(REF, "networks", "deepsort")
    if p not in sys.path:
        sys.path.insert(0, p)
    import sort.detection, sort.iou_matching, sort.kalman_filter, sort.linear_assignment  # noqa
    import sort.nn_matching, sort.preprocessing, sort.track, sort.tracker  # noqa
    import sort
    return sort


def _load_file(name, path):
    _shim()
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reid_model():
    return _load_file("ref_reid_model", os.path.join(REF, "networks", "deepsort", "deep", "model.py"))


def load_bb_polygon():
    return _load_file("ref_bb_polygon", os.path.join(REF, "utilities", "counting", "bb_polygon.py"))


def load_counting_utils():
    """utilities/counting/utils.py does `from .bb_polygon import *`-style relative imports."""
    _shim()
    pkg_dir = os.path.join(REF, "utilities", "counting")
    pkg = types.ModuleType("ref_counting")
    pkg.__path__ = [pkg_dir]
    sys.modules["ref_counting"] = pkg
    for sub in ("bb_polygon", "utils"):
        spec = importlib.util.spec_from_file_location(f"ref_counting.{sub}", os.path.join(pkg_dir, sub + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"ref_counting.{sub}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, sub, mod)
    return pkg.utils
