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
    # The engine's normalized coordinates follow Matterport's norm_boxes (anchors.py, k_paste_masks):
    # pixel = norm * (size - 1), far edge + 1.  Denormalize in the letterboxed frame, remove the padding, scale the
    # content back to the source size, renormalize with (size - 1) and the far edge - 1.
    sy, sx = h / nh, w / nw
    y1 = (d[:, 0] * (H - 1) - py) * sy
    x1 = (d[:, 1] * (W - 1) - px) * sx
    y2 = (d[:, 2] * (H - 1) + 1.0 - py) * sy
    x2 = (d[:, 3] * (W - 1) + 1.0 - px) * sx
    hy, wx = max(h - 1, 1), max(w - 1, 1)
    d[:, 0] = np.clip(y1 / hy, 0.0, 1.0)
    d[:, 1] = np.clip(x1 / wx, 0.0, 1.0)
    d[:, 2] = np.clip((y2 - 1.0) / hy, 0.0, 1.0)
    d[:, 3] = np.clip((x2 - 1.0) / wx, 0.0, 1.0)
    # zero-padded rows stay all-zero (like mrcnn_unletterbox_boxes): an all-zero box mapped through the letterbox would come out
    # with a non-zero far edge and read as a detection to a consumer that finds the padding by all-zero rows
    pad = ~np.any(np.asarray(detections, dtype=np.float64)[:, :4] != 0.0, axis=1)
    d[pad, :4] = 0.0
    return d


def detection_agreement(det_a: np.ndarray, det_b: np.ndarray, box_tol: float = 1e-4, masks_a=None, masks_b=None) -> dict:
    """End-to-end agreement of two engines' outputs for ONE image: det_* (n,6) rows (y1,x1,y2,x2,classId,score),
    zero-padded.  A row of A is matched to at most one row of B with the SAME class id and every box coordinate
    within `box_tol` (order-insensitive: scores that differ in the last bits may swap neighbours).  Returns counts,
    the matched fraction over max(nA, nB), the largest score difference over matched pairs and — when the 28×28
    masks are given — the largest mask difference over matched pairs whose mask is present on both sides, plus the
    number of pairs where the mask layer's removeZeros rule (a pooled row with an exactly-zero sample is skipped,
    TimeDistributedMaskLayer.swift:52) emptied the mask on one side only; pairs behind the first such row are reported
    separately (`*_behind_presence_mismatch`)."""
    a = np.asarray(det_a, dtype=np.float32)
    b = np.asarray(det_b, dtype=np.float32)
    ia = np.flatnonzero(a[:, 5] > 0)
    ib = np.flatnonzero(b[:, 5] > 0)
    used = np.zeros(len(ib), dtype=bool)
    matched, same_row, dscore, presence = 0, 0, 0.0, 0
    pairs = []                       # (row in A, row in B, mask difference or None when the mask is present on one side only)
    for i in ia:
        ok = (~used) & (b[ib, 4] == a[i, 4]) & (np.abs(b[ib, :4] - a[i, :4]).max(axis=1) <= box_tol)
        js = np.flatnonzero(ok)
        if js.size == 0:
            continue
        j = js[np.argmin(np.abs(b[ib[js], 5] - a[i, 5]))]
        used[j] = True
        matched += 1
        same_row += int(ib[j] == i)
        dscore = max(dscore, float(abs(b[ib[j], 5] - a[i, 5])))
        if masks_a is not None and masks_b is not None:
            ma, mb = np.asarray(masks_a[i], np.float32), np.asarray(masks_b[ib[j]], np.float32)
            if (ma == 0).all() != (mb == 0).all():
                presence += 1            # the reference's removeZeros rule dropped the mask on one side only (an exact-zero sample)
                pairs.append((int(i), int(ib[j]), None))
            else:
                pairs.append((int(i), int(ib[j]), float(np.abs(ma - mb).max())))
    # A row dropped by removeZeros on ONE side shifts the class lookup of every later row of that side (the mask layer
    # indexes `detections` with the compact index, TimeDistributedMaskLayer.swift:71): rows behind the first such row are
    # compared separately — their masks belong to different classes by the reference's own rule.
    cliff = min([min(i, j) for i, j, d in pairs if d is None], default=None)
    before = [d for i, j, d in pairs if d is not None and (cliff is None or max(i, j) < cliff)]
    after = [d for i, j, d in pairs if d is not None and not (cliff is None or max(i, j) < cliff)]
    denom = max(len(ia), len(ib))
    return {"n_a": int(len(ia)), "n_b": int(len(ib)), "matched": int(matched), "same_row": int(same_row),
            "fraction": (matched / denom) if denom else 1.0, "max_score_diff": dscore,
            "max_mask_diff": max(before, default=0.0), "mask_presence_mismatch": int(presence),
            "max_mask_diff_behind_presence_mismatch": max(after, default=0.0), "masks_behind_presence_mismatch": len(after)}


def mask_flip_causes(det_a: np.ndarray, det_b: np.ndarray, pooled_a: np.ndarray, pooled_b: np.ndarray, masks_a, masks_b,
                     box_tol: float = 1e-4, coord_tol: float = 1e-6) -> dict:
    """Why the masks of two engines differ in PRESENCE for one image — the reference's removeZeros cliff, proven from the taps.

    The mask layer skips a detection whose pooled 14x14x256 row holds an exactly-zero sample (TimeDistributedClassifierLayer.swift:
    116-127 via TimeDistributedMaskLayer.swift:52).  A sample is exactly zero when the sampler lands outside the map (a box clipped
    to the frame whose last sample position sits an ulp beyond the edge: a whole line of zeros) or when the bilinear interpolation
    of a sign change cancels exactly (chance: ~1e-7 per sample, 5 million samples per image).  Two evaluations that differ in the
    last bits can therefore disagree on whether a row is kept, and — since the kept rows are written COMPACTED and the class is
    looked up at the compact index (TimeDistributedMaskLayer.swift:58-89, :71) — every later mask of that image then sits one row
    further up on one side, under a class that may differ.

    det_* (n,6), pooled_* = the engines' own pooled_mask taps (n rows), masks_* (n,S,S).  Checked:
      * write set, per side: with n_kept = the rows whose pooled row has no exact zero, a row holds a mask iff it is kept AND its
        index is below n_kept (:83 writes the kept rows at their original index, :87-89 then zero rows [n_kept, D)) — the
        reference's rule applied to the side's OWN samples, so every presence difference between the sides follows from
        zero-sample differences, nothing else;
      * every matched pair (same class id, box within box_tol) whose zero predicate differs — a FLIP — is explained iff the two
        boxes agree within coord_tol per coordinate (last bits; they may be identical, the pyramids differ in the last bits too)
        and the zero side's row is not zero altogether.
    Returns {"flips", "explained", "unexplained": [...], "write_set_ok", "presence_mismatch", "masks_behind_flip"} —
    masks_behind_flip = matched pairs behind the first flip: their rows / class lookups are shifted on one side, compared by nobody."""
    a = np.asarray(det_a, np.float32); b = np.asarray(det_b, np.float32)
    pa = np.asarray(pooled_a, np.float32).reshape(a.shape[0], -1); pb = np.asarray(pooled_b, np.float32).reshape(b.shape[0], -1)
    ka = np.asarray(masks_a, np.float32).reshape(a.shape[0], -1); kb = np.asarray(masks_b, np.float32).reshape(b.shape[0], -1)
    ia = np.flatnonzero(a[:, 5] > 0); ib = np.flatnonzero(b[:, 5] > 0)
    zero_a = np.array([bool(np.any(pa[i] == 0.0)) for i in range(a.shape[0])]); zero_b = np.array([bool(np.any(pb[j] == 0.0)) for j in range(b.shape[0])])

    def write_set_ok(zero, k):
        # TimeDistributedMaskLayer.swift:58-89: the i-th KEPT row's mask goes to its original row (:83), then rows [n_kept, D) are
        # zeroed (:87-89) — which also wipes a kept row whose original index is >= n_kept
        kept = int(np.sum(~zero))
        has = np.any(k != 0, axis=1)
        want = (~zero) & (np.arange(zero.size) < kept)
        return bool(np.array_equal(has, want))
    ws_ok = write_set_ok(zero_a, ka) and write_set_ok(zero_b, kb)
    used = np.zeros(len(ib), dtype=bool)
    flips, explained, unexplained, first, presence = 0, 0, [], None, 0
    pairs = []
    for i in ia:
        ok = (~used) & (b[ib, 4] == a[i, 4]) & (np.abs(b[ib, :4] - a[i, :4]).max(axis=1) <= box_tol)
        js = np.flatnonzero(ok)
        if js.size == 0:
            continue
        jj = js[np.argmin(np.abs(b[ib[js], 5] - a[i, 5]))]
        used[jj] = True
        j = int(ib[jj])
        pairs.append((int(i), j))
        presence += int(bool(np.any(ka[i] != 0)) != bool(np.any(kb[j] != 0)))
        if zero_a[i] == zero_b[j]:
            continue
        flips += 1
        first = min(i, j) if first is None else min(first, i, j)
        dbox = np.abs(a[i, :4].astype(np.float64) - b[j, :4].astype(np.float64))
        zrow = pa[i] if zero_a[i] else pb[j]
        why = {"row_a": int(i), "row_b": j, "zero_sample_a": bool(zero_a[i]), "zero_sample_b": bool(zero_b[j]),
               "zeros_in_row": int(np.sum(zrow == 0.0)), "row_len": int(zrow.size),
               "box_a": a[i, :4].tolist(), "box_b": b[j, :4].tolist(), "max_coord_diff": float(dbox.max())}
        if dbox.max() <= coord_tol and why["zeros_in_row"] < why["row_len"]:
            explained += 1
        else:
            unexplained.append(why)
    behind = sum(1 for i, j in pairs if first is not None and max(i, j) > first)
    return {"flips": flips, "explained": explained, "unexplained": unexplained, "write_set_ok": ws_ok, "presence_mismatch": int(presence),
            "masks_behind_flip": int(behind)}


def evaluate(model: MaskRCNN, images: Iterable[Tuple[int, np.ndarray]], dataset_id: str = "coco",
             limit: Optional[int] = 5, verbose: bool = True, calibrate: bool = False):
    """images: (image_id, HxWx3 uint8).  Returns (results.proto bytes, [seconds per image], [PBResult]).
    calibrate: split modes only — one calibration predict on the first image before the timed loop (model set-up, like the load
    the reference keeps outside its timing, EvaluateCommand.swift:146-156): mrcnn_model_calibrate_split."""
    items = sorted(images, key=lambda it: it[0])           # sortById:true
    if limit is not None:
        items = items[:limit]
    H, W = model.image_height, model.image_width
    if calibrate and items and model.compute_dtype in ("f32x3", "f32s"):
        model.calibrate_split(letterbox(items[0][1], H, W)[None])
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


def evaluate_coco(model: MaskRCNN, annotations_json: str, load_image, dataset_id: str = "coco", limit: Optional[int] = 5, verbose: bool = True):
    """`maskrcnn evaluate` over a COCO annotation file (EvaluateCommand.swift:159-200): the first `limit` images sorted by id
    (`coco.makeImageIterator(limit: 5, sortById: true)`, :165), each loaded by `load_image(COCOImage) -> (h,w,3) uint8` (the
    reference reads `<dataset dir>/<file_name>`; decoding image files is left to the host — no codec ships here)."""
    from .coco import COCO
    coco = COCO(annotations_json)
    items = [(im.id, load_image(im)) for im, _ in coco.makeImageIterator(limit=limit, sortById=True)]
    return evaluate(model, items, dataset_id=dataset_id, limit=None, verbose=verbose)
