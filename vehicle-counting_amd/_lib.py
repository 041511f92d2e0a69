"""ctypes binding of libvcount_hip.so (C ABI in include/vcount_hip.h).  No fallback: if the library is absent or a
call fails, a VcError is raised -- the product never computes the hot path on the CPU."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# hardware queues of the HIP runtime: the engine's four streams must not share one (see vc_engine_create); a default, set as early as
# possible because the runtime reads it when it builds its queue pool
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LIB_PATH = os.environ.get("VC_LIB_PATH") or os.path.join(_HERE, "libvcount_hip.so")   # override: A/B runs of two builds on one box

VC_OK = 0
PREC_BF16, PREC_F32, PREC_FP8 = 0, 1, 2
PREC_ID = {"bf16": 0, "f32": 1, "fp8": 2}
NET_YOLO, NET_REID = 0, 1
FEAT_DIM = 512
PROF_CONV, PROF_DETECT_AUX, PROF_REID_AUX, PROF_TRACK = 0, 1, 2, 3


class VcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvcount_hip status {code}: {msg}")
        self.code = code


class EngineConfig(C.Structure):
    _fields_ = [("device", C.c_int), ("precision", C.c_int), ("yolo_variant", C.c_int), ("num_classes", C.c_int),
                ("img_size", C.c_int), ("max_batch", C.c_int), ("max_frame_h", C.c_int), ("max_frame_w", C.c_int),
                ("conf_thres", C.c_float), ("iou_thres", C.c_float), ("max_det", C.c_int), ("max_candidates", C.c_int),
                ("max_crops", C.c_int), ("max_tracks", C.c_int), ("nn_budget_cap", C.c_int),
                ("with_detector", C.c_int), ("with_reid", C.c_int), ("max_trackers", C.c_int), ("tracks_per_tracker", C.c_int)]


class TrackerParams(C.Structure):
    _fields_ = [("max_dist", C.c_double), ("min_confidence", C.c_double), ("nms_max_overlap", C.c_double),
                ("max_iou_distance", C.c_double), ("max_age", C.c_int), ("n_init", C.c_int), ("nn_budget", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("b", C.c_int), ("h", C.c_int), ("w", C.c_int), ("cin", C.c_int), ("cout", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("act", C.c_int), ("res_mode", C.c_int),
                ("precision", C.c_int)]


_P = C.POINTER
_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_pf, _pd, _pi, _pl, _pu8 = _P(C.c_float), _P(C.c_double), _P(C.c_int), _P(C.c_int64), _P(C.c_uint8)

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/vcount_hip.h one to one
SIGNATURES = {
    "vc_version": [],
    "vc_last_error": [],
    "vc_device_count": [_pi],
    "vc_engine_config_default": [_P(EngineConfig)],
    "vc_engine_create": [_P(EngineConfig), _P(_vp)],
    "vc_engine_destroy": [_vp],
    "vc_engine_param_count": [_vp, _i, _pi],
    "vc_engine_param_info": [_vp, _i, _i, C.c_char_p, _i, _pi],
    "vc_engine_set_param": [_vp, _i, C.c_char_p, _pf, _pf],
    "vc_engine_set_anchors": [_vp, _pf],
    "vc_engine_finalize": [_vp],
    "vc_engine_set_option": [_vp, C.c_char_p, _i],
    "vc_engine_sync": [_vp],
    "vc_detect": [_vp, _P(_vp), _pi, _pi, _i, _pf, _pi],
    "vc_detect_debug_shape": [_vp, _pi, _pi, _pi],
    "vc_detect_debug_layer": [_vp, _i, _pf, C.c_size_t, _pi],
    "vc_embed_debug_input": [_vp, _i, _pf, C.c_size_t, _pi],
    "vc_detect_debug_pred": [_vp, _pf, C.c_size_t],
    "vc_embed": [_vp, _pu8, _i, _i, _pd, _i, _pf],
    "vc_embed_tensor": [_vp, _pf, _i, _pf],
    "vc_tracker_create": [_vp, _P(TrackerParams), _pi],
    "vc_tracker_reset": [_vp, _i],
    "vc_tracker_destroy": [_vp, _i],
    "vc_tracker_step": [_vp, _i, _pd, _pd, _pf, _i],
    "vc_tracker_count": [_vp, _i, _pi],
    "vc_tracker_state": [_vp, _i, _i, _pl, _pi, _pi, _pi, _pi, _pd, _pd, _pi],
    "vc_tracker_debug_costs": [_vp, _i, _pd, _pd, _pi, _pi],
    "vc_tracker_snapshot": [_vp, _i, _vp, C.c_size_t, _P(C.c_size_t)],
    "vc_tracker_restore": [_vp, _i, _vp, C.c_size_t],
    "vc_deepsort_update": [_vp, _i, _pu8, _i, _i, _pd, _pd, _i, _pl, _i, _pi],
    "vc_videotracker_run": [_vp, _pi, _i, _pu8, _i, _i, _pd, _pl, _pd, _i, _pl, _i, _pi],
    "vc_stream_run": [_vp, _pi, _i, _vp, _i, _i, _i, _pl, _i, _pi, _pi],
    "vc_stream_inject": [_vp, _pf, _pi, _i, _i],
    "vc_stream_submit": [_vp, _vp, _i, _i, _i],
    "vc_stream_stage_host": [_vp, _vp, _i, _i, _i, _P(_vp)],
    "vc_stream_submit_host": [_vp, _vp, _i, _i, _i, _P(_vp)],
    "vc_stream_run_async": [_vp, _pi, _i, _vp, _i, _i, _i, _i],
    "vc_stream_collect": [_vp, _pl, _i, _pi, _pi, _i],
    "vc_stream_run_async_multi": [_vp, _pi, _i, _i, _pi, _vp, _i, _i, _i, _i],
    "vc_stream_reset": [_vp],
    "vc_stream_embed": [_vp, _vp, _i, _i, _i, _pd, _i, _pi, _P(_vp)],
    "vc_allgather_rows": [_vp, _pd, _vp, _i, _pd, _i, _pi, _P(_vp)],
    "vc_videotracker_run_features": [_vp, _pi, _i, _pd, _vp, _i, _i, _i, _pl, _i, _pi, _pl, _i, _pi],
    "vc_counter_create": [_pd, _i, _pd, _i, _i, _P(_vp)],
    "vc_counter_destroy": [_vp],
    "vc_counter_add": [_vp, _pl, _pl, _pl, _pl, _i],
    "vc_counter_tracks": [_vp, _pi],
    "vc_counts": [_vp, _pi],
    "vc_counter_rows_count": [_vp, _pl],
    "vc_counter_rows": [_vp, C.c_int64, _pl, _pl, _pl, _pl, _pi, _pd, _pd, _pl, _pl],
    "vc_comm_unique_id": [_vp],
    "vc_comm_init": [_vp, _i, _i, _vp],
    "vc_comm_destroy": [_vp],
    "vc_allgather_counts": [_vp, _pi, _i, _pi],
    "vc_overlay": [_vp, _vp, _i, _i, _i, _pi, _pi],
    "vc_profile_enable": [_vp, _i],
    "vc_profile_conv_busy": [_vp, _pd, _pd],
    "vc_profile_read_dense": [_vp, _i, _pd, _pd],
    "vc_gather_compact_host": [_vp, _i, _i, C.c_size_t, _pi, _vp, C.c_size_t, _pl],
    "vc_gather_offsets": [_i, _i, _pi, _pl, _pl, _pl],
    "vc_tune_export": [_vp, C.c_char_p, C.c_size_t, _P(C.c_size_t)],
    "vc_tune_import": [_vp, C.c_char_p],
    "vc_profile_read": [_vp, _i, _pd, _pl, _pd, _pd],
    "vc_profile_reset": [_vp],
    "vc_profile_ops": [_vp, C.c_char_p, C.c_size_t],
    "vc_conv2d_host": [_P(ConvDesc), _pf, _pf, _pf, _pf, _pf],
    "vc_kalman_initiate_host": [_pd, _i, _pd, _pd],
    "vc_kalman_predict_host": [_pd, _pd, _i],
    "vc_kalman_update_host": [_pd, _pd, _pd, _i],
    "vc_kalman_gating_host": [_pd, _pd, _pd, _i, _pd],
    "vc_iou_cost_host": [_pd, _i, _pd, _i, _pd],
    "vc_cosine_cost_host": [_pf, _pi, _i, _i, _pf, _i, _pd],
    "vc_dsort_nms_host": [_pd, _pd, _i, _d, _pi, _pi],
    "vc_lap_host": [_pd, _i, _i, _pi, _pi, _pi],
    "vc_letterbox_host": [_pu8, _i, _i, _i, _i, _i, _pf],
    "vc_zone_filter_host": [_pd, _i, _pl, _i, _pu8],
    "vc_nms_host": [_pf, _pf, _pi, _i, _f, _i, _i, _pf, _pi],
}
_RESTYPE = {"vc_last_error": C.c_char_p}

_lib = None


def lib():
    """Load (once) and return the shared library with all prototypes set."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VcError(-1, f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and loads it by path; libvcount_hip.so links /opt/rocm's.
        # If this library is loaded FIRST, a later `import torch` brings a second runtime into the process and its device enumeration
        # fails ("No HIP GPUs are available" at the first torch.cuda call -- found by running tests/test_gpu_round4.py alone).  Loaded
        # after torch, the dynamic linker resolves this library's dependency to the copy that is already there.  So when torch is
        # installed it is imported here, before the CDLL; a process that never uses torch is not affected either way.
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = l
    return _lib


def check(code):
    if code != VC_OK:
        raise VcError(code, lib().vc_last_error().decode("utf-8", "replace"))


def ptr(a, ctype):
    """Pointer to a C-contiguous numpy array of the matching dtype (or NULL for None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(_P(ctype))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
