"""Result decoding: the public ``Detection`` type and ``IOU`` of the reference library target
(``Sources/Mask-RCNN-CoreML/Detection.swift:15-99``, ``Utils.swift:232``), plus the mask
paste that the example app performs when drawing (``Example/Source/DetectionRenderer.swift:13-24``).
Host-side like the reference; the arithmetic lives in libmaskrcnn_hip.so (``mrcnn_detections_decode``,
``mrcnn_mask_to_u8``, ``mrcnn_iou``).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

from . import _lib


@dataclass
class Detection:
    index: int
    boundingBox: Tuple[float, float, float, float]   # CGRect(x, y, width, height), normalized
    classId: int
    score: float
    mask: Optional[np.ndarray]                       # 28×28 uint8 (the CGImage mask's bytes) or None

    @staticmethod
    def detectionsFromFeatureValue(featureValue: np.ndarray, maskFeatureValue: Optional[np.ndarray] = None) -> List["Detection"]:
        """featureValue: "detections" (N,6) float32; maskFeatureValue: "mask" (N,28,28) float32."""
        det = np.ascontiguousarray(featureValue, dtype=np.float32)
        if det.ndim != 2 or det.shape[1] < 6:
            return []
        n = det.shape[0]
        recs = (_lib.DetectionRecord * max(1, n))()
        cnt = C.c_int64(0)
        _lib.check(_lib.lib().mrcnn_detections_decode(det.ctypes.data, n, det.shape[1], recs, n, C.byref(cnt)))
        out = []
        for k in range(cnt.value):
            r = recs[k]
            mask = None
            if maskFeatureValue is not None and maskFeatureValue.shape[0] > r.index:
                mask = Detection.maskFromFeatureValue(maskFeatureValue, r.index)
            out.append(Detection(int(r.index), (r.x, r.y, r.w, r.h), int(r.class_id), float(r.score), mask))
        return out

    @staticmethod
    def maskFromFeatureValue(maskFeatureValue: np.ndarray, atIndex: int) -> Optional[np.ndarray]:
        if maskFeatureValue.shape[0] <= atIndex:
            return None
        m = np.ascontiguousarray(maskFeatureValue[atIndex], dtype=np.float32)
        out = np.empty(m.shape, dtype=np.uint8)
        _lib.check(_lib.lib().mrcnn_mask_to_u8(m.ctypes.data, m.size, out.ctypes.data))
        return out


def IOU(a_xywh, b_xywh) -> float:
    """``IOU(_ a: CGRect, _ b: CGRect) -> Float`` with rects as (x, y, width, height)."""
    def yxyx(r):
        x, y, w, h = r
        return np.array([y, x, y + h, x + w], dtype=np.float32)
    a, b = yxyx(a_xywh), yxyx(b_xywh)
    f32p = C.POINTER(C.c_float)
    return float(_lib.lib().mrcnn_iou(a.ctypes.data_as(f32p), b.ctypes.data_as(f32p)))


def paste_masks(detections: np.ndarray, masks: np.ndarray, image_h: int, image_w: int, threshold: float = 0.5) -> np.ndarray:
    """Full-resolution binary instance masks (n, image_h, image_w) uint8 from "detections" (n,6) and
    "mask" (n,S,S): each mask is resized to its box and thresholded on the GPU (``mrcnn_paste_masks``) —
    the step DetectionRenderer.renderMask (Example/Source/DetectionRenderer.swift:13-24) leaves to
    CoreGraphics when drawing, and the one mask AP needs."""
    det = np.ascontiguousarray(detections, dtype=np.float32)
    m = np.ascontiguousarray(masks, dtype=np.float32)
    n = det.shape[0]
    out = np.empty((n, image_h, image_w), dtype=np.uint8)
    _lib.check(_lib.lib().mrcnn_paste_masks(det.ctypes.data, det.shape[1], m.ctypes.data, n, m.shape[1], image_h, image_w,
                                            C.c_float(threshold), _lib.HOST, out.ctypes.data))
    return out
