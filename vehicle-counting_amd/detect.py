"""ImageDetect: drop-in for /root/reference/modules/detect.py:8-60 (and, below it, networks/detector.py:36-38 and
networks/yolo.py:46-99) on top of the HIP engine.

Same constructor arguments (`args`, `config`), same attributes (`class_names`, `device`), same `run(batch)` contract:
`batch['imgs']` is a list of HxWx3 uint8 RGB frames (modules/datasets.py:59-61), the result is
{"boxes": [f64 (n,4) xywh top-left, source pixels], "labels": [int (n,)], "scores": [f64 (n,)]} with zero-length
arrays for empty images (networks/yolo.py:91-96) so that `len(boxes) == 0` works upstream (modules/__init__.py:68).
"""
from __future__ import annotations

import os

import numpy as np

from .engine import Engine
from .weights import synth_reid, synth_yolo

COCO_NAMES = [f"class{i}" for i in range(80)]


def load_detector_weights(path, model_name="yolov5s"):
    """Fused detector parameters {name.weight, name.bias[, model.24.anchors_px]} from what `--weight` may point at:
      * `.pt` / `.pth`  -- an ultralytics/yolov5 v6.0 checkpoint (what /root/reference/networks/yolo.py:58 hands to torch.hub),
                           through the stub unpickler of checkpoint.py (no upstream package needed);
      * `.npz` / `.safetensors` -- a flat {name: array} dict, BatchNorm fused or not (weights.fold_yolo_state_dict folds it)."""
    from .weights import fold_yolo_state_dict
    ext = os.path.splitext(path)[1].lower()
    if ext in (".pt", ".pth"):
        from .checkpoint import load_yolov5_checkpoint
        return load_yolov5_checkpoint(path, model_name)
    if ext == ".npz":
        raw = dict(np.load(path))
    elif ext == ".safetensors":
        from safetensors.numpy import load_file
        raw = load_file(path)
    else:
        raise ValueError(f"--weight {path}: expected .pt/.pth (yolov5 v6.0 checkpoint), .npz or .safetensors")
    sd = fold_yolo_state_dict(raw)
    if "model.24.anchors_px" in raw:
        sd["model.24.anchors_px"] = np.asarray(raw["model.24.anchors_px"], np.float32)
    return sd


def load_reid_weights(path):
    """Un-fused ReID state_dict from `ckpt.t7` (feature_extractor.py:13-14) or a flat .npz / .safetensors of the same names."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        return dict(np.load(path))
    if ext == ".safetensors":
        from safetensors.numpy import load_file
        return load_file(path)
    from .checkpoint import load_reid_checkpoint
    return load_reid_checkpoint(path)


class ImageDetect:
    """`synthetic=True` (or args.synthetic) asks for seeded synthetic weights explicitly -- benchmarks and parity tests, where no
    checkpoint exists.  Without it a missing `--weight` / ReID checkpoint raises: the reference would download pretrained
    weights (utilities/utils.py:203-208), which this build cannot, and silently tracking on random weights is never wanted."""

    def __init__(self, args, config, engine=None, class_names=None, synthetic=False, reid_checkpoint=None):
        self.mapping_dict = getattr(args, "mapping", None)           # modules/detect.py:12 (dead by default, Q13)
        model_name = config.model_name or "yolov5s"
        synthetic = synthetic or bool(getattr(args, "synthetic", False))
        if engine is None:
            weight = getattr(args, "weight", None)
            reid_checkpoint = reid_checkpoint or getattr(args, "reid_checkpoint", None)
            if weight:
                ysd = load_detector_weights(weight, model_name)
            elif synthetic:
                ysd = synth_yolo(model_name)
            else:
                raise ValueError("ImageDetect: no --weight given (a yolov5 v6.0 .pt, .npz or .safetensors); pass synthetic=True to "
                                 "run on seeded synthetic weights")
            if reid_checkpoint:
                rsd = load_reid_weights(reid_checkpoint)
            elif synthetic:
                rsd = synth_reid()
            else:
                raise ValueError("ImageDetect: no ReID checkpoint given (cam_config 'checkpoint' / deepsort ckpt.t7); pass "
                                 "synthetic=True to run on seeded synthetic weights")
            num_classes = ysd["model.24.m.0.weight"].shape[0] // 3 - 5          # from the checkpoint, not a default
            engine = Engine(ysd, rsd, model_name=model_name, num_classes=num_classes,
                            conf_thres=config.min_conf, iou_thres=config.min_iou, max_det=config.max_det,
                            precision=getattr(args, "precision", "bf16"))
        self.engine = engine
        self.device = f"hip:{engine.cfg.device}"
        self.class_names = list(class_names) if class_names is not None else COCO_NAMES[: engine.cfg.num_classes]
        if self.mapping_dict is not None:                               # modules/detect.py:17-21
            self.included_classes = list(self.mapping_dict.keys())
            ids = sorted(np.unique(list(self.mapping_dict.values())))
            self.class_names = [self.class_names[i] for i in ids]

    @staticmethod
    def _marshal(det):
        """networks/yolo.py:72-97: pandas xyxy -> to_json (10 decimals, Q9) -> json.loads -> xywh float64."""
        if len(det) == 0:
            return {"bboxes": np.array(()), "classes": np.array(()), "scores": np.array(())}
        d = np.round(det.astype(np.float64), 10)
        boxes = np.stack((d[:, 0], d[:, 1], d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]), 1)
        return {"bboxes": boxes, "classes": det[:, 5].astype(np.int64), "scores": d[:, 4]}

    def run(self, batch):
        dets = self.engine.detect(batch["imgs"])
        boxes_result, labels_result, scores_result = [], [], []
        for det in dets:
            out = self._marshal(det)
            if self.mapping_dict is not None and len(out["classes"]):        # modules/detect.py:41-46
                keep = [i for i, c in enumerate(out["classes"]) if c in self.included_classes]
                out["classes"] = np.array([self.mapping_dict[int(c) - 1] for c in out["classes"][keep]])
                out["scores"], out["bboxes"] = out["scores"][keep], out["bboxes"][keep]
            boxes_result.append(out["bboxes"])
            labels_result.append(out["classes"])
            scores_result.append(out["scores"])
        return {"boxes": boxes_result, "labels": labels_result, "scores": scores_result}
