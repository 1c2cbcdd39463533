"""Result decoding: the public ``Detection`` type and ``IOU`` of the reference library target
(``Sources/Mask-RCNN-CoreML/Detection.swift:15-99``, ``Utils.swift:232``), plus the GPU-free mask
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


def paste_mask(mask28: np.ndarray, box_xywh, image_w: int, image_h: int, threshold: float = 0.5) -> np.ndarray:
    """Resize a 28×28 probability mask to its box (bilinear) and threshold it into a full-image
    boolean mask — what DetectionRenderer.renderMask does with CoreGraphics when drawing."""
    x, y, w, h = box_xywh
    x0, y0 = int(round(x * image_w)), int(round(y * image_h))
    bw, bh = max(1, int(round(w * image_w))), max(1, int(round(h * image_h)))
    ys = (np.arange(bh) + 0.5) * mask28.shape[0] / bh - 0.5
    xs = (np.arange(bw) + 0.5) * mask28.shape[1] / bw - 0.5
    y0i = np.clip(np.floor(ys).astype(int), 0, mask28.shape[0] - 1); y1i = np.clip(y0i + 1, 0, mask28.shape[0] - 1)
    x0i = np.clip(np.floor(xs).astype(int), 0, mask28.shape[1] - 1); x1i = np.clip(x0i + 1, 0, mask28.shape[1] - 1)
    fy = np.clip(ys - np.floor(ys), 0, 1)[:, None]; fx = np.clip(xs - np.floor(xs), 0, 1)[None, :]
    m = mask28.astype(np.float32)
    top = m[y0i][:, x0i] * (1 - fx) + m[y0i][:, x1i] * fx
    bot = m[y1i][:, x0i] * (1 - fx) + m[y1i][:, x1i] * fx
    r = top * (1 - fy) + bot * fy
    full = np.zeros((image_h, image_w), dtype=bool)
    ys0, xs0 = max(0, y0), max(0, x0)
    ys1, xs1 = min(image_h, y0 + bh), min(image_w, x0 + bw)
    if ys1 > ys0 and xs1 > xs0:
        full[ys0:ys1, xs0:xs1] = r[ys0 - y0:ys1 - y0, xs0 - x0:xs1 - x0] >= threshold
    return full
