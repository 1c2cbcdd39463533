"""ctypes front-end of the CPU oracle (oracle/mrcnn_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of mrcnn_oracle.c.  PARITY STATUS: parity unpinned (the
reference has no tests/fixtures and cannot run here); pinned by hand-computed known-answer cases and
brute-force re-implementations in tests/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmrcnn_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mrcnn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p, i64p, i32p, u32p, f64p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint8))
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_iou.restype = C.c_float
        L.orc_iou.argtypes = [f32p, f32p]
        L.orc_nms.restype = C.c_int64
        L.orc_nms.argtypes = [f32p, i64p, C.c_int64, C.c_float, C.c_int64, i64p]
        L.orc_sorted_indices_desc.argtypes = [f32p, C.c_int64, u32p]
        L.orc_apply_box_deltas.argtypes = [f32p, f32p, C.c_int64]
        L.orc_clip_boxes.argtypes = [f32p, C.c_int64]
        L.orc_proposal_layer.restype = C.c_int64
        L.orc_proposal_layer.argtypes = [f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_float, f32p,
                                         f32p, C.c_int64, u32p, f32p, i64p]
        L.orc_roi_levels.argtypes = [f32p, C.c_int64, C.c_int64, C.c_double, C.c_double, i32p]
        L.orc_pyramid_roi_align.argtypes = [f32p, C.c_int64, C.c_int64, C.POINTER(f32p), i64p, i64p, C.c_int64,
                                            C.c_int64, C.c_double, C.c_double, f32p, C.c_int64]
        L.orc_classifier_postprocess.argtypes = [f64p, f64p, C.c_int64, C.c_int64, f32p, C.c_int64]
        L.orc_detection_layer.restype = C.c_int64
        L.orc_detection_layer.argtypes = [f32p, f32p, C.c_int64, f32p, C.c_int64, C.c_float, C.c_float, f32p,
                                          C.c_int64]
        L.orc_mask_valid_rows.restype = C.c_int64
        L.orc_mask_valid_rows.argtypes = [f32p, C.c_int64, C.c_int64, C.c_int64, i64p]
        L.orc_mask_layer_write.argtypes = [f64p, C.c_int64, i64p, C.c_int64, C.c_int64, f32p, C.c_int64,
                                           C.c_int64, f32p, C.c_int64]
        L.orc_detections_decode.restype = C.c_int64
        L.orc_detections_decode.argtypes = [f32p, C.c_int64, C.c_int64, i64p, f64p, i64p, f64p]
        L.orc_mask_to_u8.argtypes = [f64p, C.c_int64, u8p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def expf(x: np.ndarray) -> np.ndarray:
    x = _f32(x)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.orc_expf(C.c_float(float(flat_in[i])))
    return out


def iou(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().orc_iou(_p(a, C.c_float), _p(b, C.c_float)))


def sorted_indices_desc(v) -> np.ndarray:
    v = _f32(v)
    idx = np.empty(v.size, dtype=np.uint32)
    lib().orc_sorted_indices_desc(_p(v, C.c_float), v.size, _p(idx, C.c_uint32))
    return idx


def apply_box_deltas(boxes, deltas) -> np.ndarray:
    b = _f32(boxes).copy()
    d = _f32(deltas)
    lib().orc_apply_box_deltas(_p(b, C.c_float), _p(d, C.c_float), b.shape[0])
    return b


def clip_boxes(boxes) -> np.ndarray:
    b = _f32(boxes).copy()
    lib().orc_clip_boxes(_p(b, C.c_float), b.shape[0])
    return b


def nms(boxes, indices, iou_threshold: float, max_keep: int) -> np.ndarray:
    b = _f32(boxes)
    ind = np.ascontiguousarray(indices, dtype=np.int64)
    sel = np.empty(max(1, max_keep), dtype=np.int64)
    n = lib().orc_nms(_p(b, C.c_float), _p(ind, C.c_int64), ind.size, C.c_float(iou_threshold), max_keep,
                      _p(sel, C.c_int64))
    return sel[:n].copy()


def proposal_layer(probs, deltas, anchors, pre_nms=6000, max_proposals=1000, nms_thr=0.7,
                   std=(0.1, 0.1, 0.2, 0.2), out_stride=4, out=None, debug=False):
    """ProposalLayer.evaluate.  probs (A,2), deltas (A,4), anchors (A,4) → rois (max_proposals, stride)."""
    probs, deltas, anchors = _f32(probs), _f32(deltas), _f32(anchors)
    A = probs.shape[0]
    n = min(A, pre_nms)
    if out is None:
        out = np.full((max_proposals, out_stride), np.float32(np.nan), dtype=np.float32)
    std4 = _f32(std)
    topk = np.empty(max(1, n), dtype=np.uint32)
    boxes = np.empty((max(1, n), 4), dtype=np.float32)
    keep = np.empty(max(1, max_proposals), dtype=np.int64)
    nk = lib().orc_proposal_layer(_p(probs, C.c_float), _p(deltas, C.c_float), _p(anchors, C.c_float), A, pre_nms,
                                  max_proposals, C.c_float(nms_thr), _p(std4, C.c_float), _p(out, C.c_float),
                                  out_stride, _p(topk, C.c_uint32), _p(boxes, C.c_float), _p(keep, C.c_int64))
    if debug:
        return out, {"count": int(nk), "topk_idx": topk[:n].copy(), "boxes": boxes[:n].copy(),
                     "keep": keep[:nk].copy()}
    return out


def roi_levels(rois, image_w: float, image_h: float) -> np.ndarray:
    r = _f32(rois)
    lv = np.empty(r.shape[0], dtype=np.int32)
    lib().orc_roi_levels(_p(r, C.c_float), r.shape[0], r.shape[1], float(image_w), float(image_h), _p(lv, C.c_int32))
    return lv


def pyramid_roi_align(rois, fmaps, pool: int, image_w: float, image_h: float) -> np.ndarray:
    """rois (n, >=4); fmaps: 4 arrays (C, H, W) → (n, C, pool, pool)."""
    r = _f32(rois)
    fm = [_f32(f) for f in fmaps]
    Cc = fm[0].shape[0]
    H = np.array([f.shape[1] for f in fm], dtype=np.int64)
    W = np.array([f.shape[2] for f in fm], dtype=np.int64)
    ptrs = (C.POINTER(C.c_float) * len(fm))(*[_p(f, C.c_float) for f in fm])
    out = np.empty((r.shape[0], Cc, pool, pool), dtype=np.float32)
    lib().orc_pyramid_roi_align(_p(r, C.c_float), r.shape[0], r.shape[1], ptrs, _p(H, C.c_int64), _p(W, C.c_int64),
                                Cc, pool, float(image_w), float(image_h), _p(out, C.c_float), Cc * pool * pool)
    return out


def classifier_postprocess(probs, bbox) -> np.ndarray:
    """probs (n, nc), bbox (n, nc*4) → (n, 6) rows (dy,dx,dh,dw,classId,score)."""
    p = np.ascontiguousarray(probs, dtype=np.float64)
    b = np.ascontiguousarray(bbox, dtype=np.float64).reshape(p.shape[0], -1)
    out = np.empty((p.shape[0], 6), dtype=np.float32)
    lib().orc_classifier_postprocess(_p(p, C.c_double), _p(b, C.c_double), p.shape[0], p.shape[1],
                                     _p(out, C.c_float), 6)
    return out


def detection_layer(rois, cls, max_detections=100, score_thr=0.7, nms_thr=0.3,
                    std=(0.1, 0.1, 0.2, 0.2), out_stride=6, return_count=False):
    r = _f32(rois)
    assert r.shape[1] == 4
    c = _f32(cls)
    assert c.shape[1] == 6
    std4 = _f32(std)
    out = np.full((max_detections, out_stride), np.float32(np.nan), dtype=np.float32)
    nd = lib().orc_detection_layer(_p(r, C.c_float), _p(c, C.c_float), r.shape[0], _p(std4, C.c_float),
                                   max_detections, C.c_float(score_thr), C.c_float(nms_thr), _p(out, C.c_float),
                                   out_stride)
    return (out, int(nd)) if return_count else out


def mask_valid_rows(pooled) -> np.ndarray:
    p = _f32(pooled)
    n = p.shape[0]
    row = int(np.prod(p.shape[1:]))
    m = np.empty(max(1, n), dtype=np.int64)
    k = lib().orc_mask_valid_rows(_p(p, C.c_float), n, row, row, _p(m, C.c_int64))
    return m[:k].copy()


def mask_layer_write(masks_kept, index_mapping, detections, out) -> np.ndarray:
    """masks_kept (k, nc, mh, mw); detections (D, 6); out (D, mh*mw) modified in place and returned."""
    mk = np.ascontiguousarray(masks_kept, dtype=np.float64)
    im = np.ascontiguousarray(index_mapping, dtype=np.int64)
    d = _f32(detections)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    k = im.size
    nc = mk.shape[1] if k else 1
    mlen = out.shape[1]
    lib().orc_mask_layer_write(_p(mk, C.c_double), k, _p(im, C.c_int64), nc, mlen, _p(d, C.c_float), d.shape[0],
                               d.shape[1], _p(out, C.c_float), mlen)
    return out


def detections_decode(det):
    d = _f32(det)
    n = d.shape[0]
    idx = np.empty(max(1, n), dtype=np.int64)
    xywh = np.empty((max(1, n), 4), dtype=np.float64)
    cls = np.empty(max(1, n), dtype=np.int64)
    sc = np.empty(max(1, n), dtype=np.float64)
    k = lib().orc_detections_decode(_p(d, C.c_float), n, d.shape[1], _p(idx, C.c_int64), _p(xywh, C.c_double),
                                    _p(cls, C.c_int64), _p(sc, C.c_double))
    return idx[:k].copy(), xywh[:k].copy(), cls[:k].copy(), sc[:k].copy()


def mask_to_u8(mask) -> np.ndarray:
    m = np.ascontiguousarray(mask, dtype=np.float64)
    out = np.empty(m.shape, dtype=np.uint8)
    lib().orc_mask_to_u8(_p(m, C.c_double), m.size, _p(out, C.c_uint8))
    return out


def paste_masks(detections, masks, image_h: int, image_w: int, threshold: float = 0.5) -> np.ndarray:
    """numpy restatement of the mask paste (reference: DetectionRenderer.swift:13-24 leaves the
    resize to CoreGraphics, so the conventions are OURS and unpinned): Matterport denorm_boxes pixels
    (np.around = round-half-even, +1 on the far edge), bilinear with half-pixel centres and edge clamp in
    float32, `>= threshold`.  Returns (n, image_h, image_w) uint8."""
    f = np.float32
    det = _f32(detections)
    m = _f32(masks)
    n, S = det.shape[0], m.shape[1]
    out = np.zeros((n, image_h, image_w), dtype=np.uint8)
    for i in range(n):
        if not det[i, 5] > 0:
            continue
        y1 = int(np.around(np.float64(det[i, 0]) * (image_h - 1)))
        x1 = int(np.around(np.float64(det[i, 1]) * (image_w - 1)))
        y2 = int(np.around(np.float64(det[i, 2]) * (image_h - 1) + 1.0))
        x2 = int(np.around(np.float64(det[i, 3]) * (image_w - 1) + 1.0))
        bh, bw = y2 - y1, x2 - x1
        if bh <= 0 or bw <= 0:
            continue
        ys = np.arange(max(y1, 0), min(y2, image_h))
        xs = np.arange(max(x1, 0), min(x2, image_w))
        if ys.size == 0 or xs.size == 0:
            continue
        sy = ((ys - y1).astype(f) + f(0.5)) * (f(S) / f(bh)) - f(0.5)
        sx = ((xs - x1).astype(f) + f(0.5)) * (f(S) / f(bw)) - f(0.5)
        sy = np.minimum(np.maximum(sy, f(0)), f(S - 1)).astype(f)
        sx = np.minimum(np.maximum(sx, f(0)), f(S - 1)).astype(f)
        ya = np.floor(sy).astype(np.int64); yb = np.minimum(ya + 1, S - 1)
        xa = np.floor(sx).astype(np.int64); xc = np.minimum(xa + 1, S - 1)
        fy = (sy - ya.astype(f)).astype(f)[:, None]
        fx = (sx - xa.astype(f)).astype(f)[None, :]
        a = m[i][ya][:, xa]; b = m[i][ya][:, xc]; c = m[i][yb][:, xa]; d = m[i][yb][:, xc]
        top = (a + ((b - a).astype(f) * fx).astype(f)).astype(f)
        bot = (c + ((d - c).astype(f) * fx).astype(f)).astype(f)
        v = (top + ((bot - top).astype(f) * fy).astype(f)).astype(f)
        out[i][np.ix_(ys, xs)] = (v >= f(threshold)).astype(np.uint8)
    return out


def letterbox(image, H: int, W: int) -> np.ndarray:
    """numpy restatement of the GPU letterbox (`.scaleFit`, EvaluateCommand.swift:157; Vision's resampler is
    closed → OUR convention, unpinned): aspect-preserving bilinear (half-pixel centres, edge clamp, float32,
    round-half-up), centred, black borders."""
    f = np.float32
    img = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = img.shape[:2]
    sc = min(W / w, H / h)
    nh = min(H, max(1, int(np.floor(h * sc + 0.5))))
    nw = min(W, max(1, int(np.floor(w * sc + 0.5))))
    py, px = (H - nh) // 2, (W - nw) // 2
    ry, rx = f(h) / f(nh), f(w) / f(nw)
    sy = (np.arange(nh).astype(f) + f(0.5)) * ry - f(0.5)
    sx = (np.arange(nw).astype(f) + f(0.5)) * rx - f(0.5)
    sy = np.minimum(np.maximum(sy, f(0)), f(h - 1)).astype(f)
    sx = np.minimum(np.maximum(sx, f(0)), f(w - 1)).astype(f)
    ya = np.floor(sy).astype(np.int64); yb = np.minimum(ya + 1, h - 1)
    xa = np.floor(sx).astype(np.int64); xb = np.minimum(xa + 1, w - 1)
    fy = (sy - ya.astype(f)).astype(f)[:, None, None]
    fx = (sx - xa.astype(f)).astype(f)[None, :, None]
    src = img.astype(f)
    a = src[ya][:, xa]; b = src[ya][:, xb]; c = src[yb][:, xa]; d = src[yb][:, xb]
    top = (a + ((b - a).astype(f) * fx).astype(f)).astype(f)
    bot = (c + ((d - c).astype(f) * fx).astype(f)).astype(f)
    v = (top + ((bot - top).astype(f) * fy).astype(f)).astype(f)
    out = np.zeros((H, W, 3), dtype=np.uint8)
    out[py:py + nh, px:px + nw] = np.floor(v + f(0.5)).astype(np.uint8)
    return out
