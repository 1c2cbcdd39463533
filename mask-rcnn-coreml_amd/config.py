"""Configuration: the conversion-time JSON keys and the runtime URL singleton.

Mirrors two things in the reference:

* the conversion config (``README.md:85-92`` of the reference; loaded at
  ``Sources/maskrcnn/Python/Conversion/task.py:166-169``): ``architecture``, ``input_image_shape``,
  ``num_classes``, ``pre_nms_max_proposals``, ``max_proposals`` — plus the Matterport defaults the
  converter bakes into the custom-layer parameter dicts (``task.py:25-67``);
* ``MaskRCNNConfig.defaultConfig`` (``Sources/Mask-RCNN-CoreML/MaskRCNNConfig.swift:10-18``): a
  process-global holder of ``anchorsURL``, ``compiledClassifierModelURL``, ``compiledMaskModelURL``
  that must be set before the main model is loaded (``Example/Source/AppDelegate.swift:18-20``).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, asdict
from typing import Optional, Tuple


@dataclass
class ModelConfig:
    architecture: str = "resnet101"                  # or "resnet50"
    input_image_shape: Tuple[int, int, int] = (1024, 1024, 3)
    num_classes: int = 81
    pre_nms_max_proposals: int = 6000
    max_proposals: int = 1000
    # Matterport defaults that the converter copies from the Keras layers (task.py:29-34,60-65)
    max_detections: int = 100
    bounding_box_std_dev: Tuple[float, float, float, float] = (0.1, 0.1, 0.2, 0.2)
    rpn_nms_threshold: float = 0.7
    detection_min_confidence: float = 0.7
    detection_nms_threshold: float = 0.3
    classifier_pool_size: int = 7
    mask_pool_size: int = 14
    # anchors (generator lives in the un-vendored third-party package; Matterport layout)
    anchor_scales: Tuple[int, ...] = (32, 64, 128, 256, 512)
    anchor_ratios: Tuple[float, ...] = (0.5, 1.0, 2.0)
    backbone_strides: Tuple[int, ...] = (4, 8, 16, 32, 64)
    anchor_stride: int = 1
    # image bias applied by the Core ML image input (task.py:73-75)
    mean_rgb: Tuple[float, float, float] = (123.7, 116.8, 103.9)

    @property
    def image_height(self) -> int:
        return int(self.input_image_shape[0])

    @property
    def image_width(self) -> int:
        return int(self.input_image_shape[1])

    def feature_shapes(self):
        import math
        return [(int(math.ceil(self.image_height / s)), int(math.ceil(self.image_width / s)))
                for s in self.backbone_strides]

    def num_anchors(self) -> int:
        return sum(h * w * len(self.anchor_ratios) // (self.anchor_stride ** 2)
                   for h, w in self.feature_shapes())

    @classmethod
    def from_json(cls, path: str) -> "ModelConfig":
        with open(path) as f:
            d = json.load(f)
        return cls.from_dict(d)

    @classmethod
    def from_dict(cls, d: dict) -> "ModelConfig":
        c = cls()
        for k, v in d.items():
            if hasattr(c, k):
                cur = getattr(c, k)
                setattr(c, k, tuple(v) if isinstance(cur, tuple) else type(cur)(v))
        return c

    def to_dict(self) -> dict:
        return asdict(self)

    # --- the custom-layer parameter dictionaries exactly as task.py emits them -------------------
    def proposal_layer_params(self) -> dict:          # task.py:25-35
        p = {"bboxStdDev_count": len(self.bounding_box_std_dev)}
        for i, v in enumerate(self.bounding_box_std_dev):
            p[f"bboxStdDev_{i}"] = float(v)
        p["preNMSMaxProposals"] = int(self.pre_nms_max_proposals)
        p["maxProposals"] = int(self.max_proposals)
        p["nmsIOUThreshold"] = float(self.rpn_nms_threshold)
        return p

    def pyramid_params(self, pool_size: int) -> dict:  # task.py:37-44
        # NB task.py:41-42 writes image_shape[0] as imageWidth and [1] as imageHeight (sic).
        return {"poolSize": int(pool_size),
                "imageWidth": int(self.input_image_shape[0]),
                "imageHeight": int(self.input_image_shape[1])}

    def detection_layer_params(self) -> dict:          # task.py:57-67
        p = {"bboxStdDev_count": len(self.bounding_box_std_dev)}
        for i, v in enumerate(self.bounding_box_std_dev):
            p[f"bboxStdDev_{i}"] = float(v)
        p["maxDetections"] = int(self.max_detections)
        p["scoreThreshold"] = float(self.detection_min_confidence)
        p["nmsIOUThreshold"] = float(self.detection_nms_threshold)
        return p


class MaskRCNNConfig:
    """Runtime singleton (MaskRCNNConfig.swift:10-18).  Setting a field forwards it to the native
    library's process-global config (``mrcnn_config_set_*``) so that layers created through the C ABI
    by a non-Python host see the same state."""

    _default: Optional["MaskRCNNConfig"] = None

    def __init__(self):
        self._anchors = None
        self._classifier = None
        self._mask = None

    @classmethod
    def defaultConfig(cls) -> "MaskRCNNConfig":
        if cls._default is None:
            cls._default = cls()
        return cls._default

    def _push(self, which: str, value):
        from . import _lib
        fn = getattr(_lib.lib(), f"mrcnn_config_set_{which}")
        _lib.check(fn(None if value is None else str(value).encode()))

    @property
    def anchorsURL(self):
        return self._anchors

    @anchorsURL.setter
    def anchorsURL(self, v):
        self._anchors = v
        self._push("anchors_path", v)

    @property
    def compiledClassifierModelURL(self):
        return self._classifier

    @compiledClassifierModelURL.setter
    def compiledClassifierModelURL(self, v):
        self._classifier = v
        self._push("classifier_path", v)

    @property
    def compiledMaskModelURL(self):
        return self._mask

    @compiledMaskModelURL.setter
    def compiledMaskModelURL(self, v):
        self._mask = v
        self._push("mask_path", v)
