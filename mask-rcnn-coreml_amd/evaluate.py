"""Caller harness: the counterpart of ``maskrcnn evaluate``.

Mirrors ``evaluate(...)`` of ``Sources/maskrcnn/EvaluateCommand.swift:134-200``:
model load OUTSIDE the loop (:146-156), then for each image (first ``limit`` images sorted by id,
``COCO.swift:60-78``; the reference uses limit 5, :165): `.scaleFit` letterbox to the model's input
size (:157) → predict → wall-clock seconds around exactly that (:167,179, printed :193) → detections
with probability > 0.7 as ``results.proto`` records (:203-248, boxes normalized in the letterboxed
frame, masks dropped, classLabel "test").  Letterbox, predict and everything in between run on the GPU
through the C ABI; this file only sequences calls.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Iterable, List, Optional, Tuple

import numpy as np

from . import _lib
from .models import MaskRCNN, load_maskrcnn
from .results_pb import PBResult, detections_to_pb, encode_results


def letterbox_geometry(h: int, w: int, H: int, W: int) -> Tuple[int, int, int, int]:
    nh, nw, py, px = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().mrcnn_letterbox_geometry(h, w, H, W, C.byref(nh), C.byref(nw), C.byref(py), C.byref(px)))
    return nh.value, nw.value, py.value, px.value


def letterbox(image: np.ndarray, H: int, W: int) -> np.ndarray:
    """image (h,w,3) uint8 → (H,W,3) uint8, `.scaleFit` on the GPU."""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    out = np.empty((H, W, 3), dtype=np.uint8)
    _lib.check(_lib.lib().mrcnn_letterbox_rgb(img.ctypes.data, img.shape[0], img.shape[1], _lib.HOST, out.ctypes.data, H, W))
    return out


def unletterbox_boxes(detections: np.ndarray, h: int, w: int, H: int, W: int) -> np.ndarray:
    """Normalized (y1,x1,y2,x2) in the letterboxed frame → normalized in the source image (host arithmetic)."""
    nh, nw, py, px = letterbox_geometry(h, w, H, W)
    d = np.array(detections, dtype=np.float64, copy=True)
    d[:, [0, 2]] = np.clip((d[:, [0, 2]] * H - py) / nh, 0.0, 1.0)
    d[:, [1, 3]] = np.clip((d[:, [1, 3]] * W - px) / nw, 0.0, 1.0)
    return d


def evaluate(model: MaskRCNN, images: Iterable[Tuple[int, np.ndarray]], dataset_id: str = "coco",
             limit: Optional[int] = 5, verbose: bool = True):
    """images: (image_id, HxWx3 uint8).  Returns (results.proto bytes, [seconds per image], [PBResult])."""
    items = sorted(images, key=lambda it: it[0])           # sortById:true
    if limit is not None:
        items = items[:limit]
    H, W = model.image_height, model.image_width
    out: List[PBResult] = []
    secs: List[float] = []
    for image_id, img in items:
        t0 = time.perf_counter()
        lb = letterbox(img, H, W)
        r = model.prediction(lb)
        t1 = time.perf_counter()
        out.append(PBResult(dataset_id, str(image_id), int(img.shape[1]), int(img.shape[0]), detections_to_pb(r["detections"])))
        secs.append(t1 - t0)
        if verbose:
            print(t1 - t0)                                 # EvaluateCommand.swift:193
    return encode_results(out), secs, out


def evaluate_from_dir(model_dir: str, images, **kw):
    return evaluate(load_maskrcnn(model_dir, max_batch=1), images, **kw)
