"""ImageDetect: drop-in for /root/reference/modules/detect.py:8-60 (and, below it, networks/detector.py:36-38 and
networks/yolo.py:46-99) on top of the HIP engine.

Same constructor arguments (`args`, `config`), same attributes (`class_names`, `device`), same `run(batch)` contract:
`batch['imgs']` is a list of HxWx3 uint8 RGB frames (modules/datasets.py:59-61), the result is
{"boxes": [f64 (n,4) xywh top-left, source pixels], "labels": [int (n,)], "scores": [f64 (n,)]} with zero-length
arrays for empty images (networks/yolo.py:91-96) so that `len(boxes) == 0` works upstream (modules/__init__.py:68).
"""
from __future__ import annotations

import numpy as np

from .engine import Engine
from .weights import synth_reid, synth_yolo

COCO_NAMES = [f"class{i}" for i in range(80)]


def load_flat_weights(path):
    """{name: ndarray} from .npz or .safetensors (a converted yolov5 v6.0 / ckpt.t7 checkpoint, BN un-fused or fused)."""
    if path.endswith(".npz"):
        return dict(np.load(path))
    from safetensors.numpy import load_file
    return load_file(path)


class ImageDetect:
    def __init__(self, args, config, engine=None, class_names=None):
        self.mapping_dict = getattr(args, "mapping", None)           # modules/detect.py:12 (dead by default, Q13)
        model_name = config.model_name or "yolov5s"
        if engine is None:
            weight = getattr(args, "weight", None)
            ysd = load_flat_weights(weight) if weight else synth_yolo(model_name)
            engine = Engine(ysd, synth_reid(), model_name=model_name,
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
