"""The three-model surface: ``MaskRCNN``, ``Classifier``, ``Mask``.

Mirrors the classes Xcode generates from the reference's three ``.mlmodel`` artefacts
(``Example/iOS Example.xcodeproj/project.pbxproj:25-28``; specs emitted by
``Sources/maskrcnn/Python/Conversion/task.py:69-116``): ``MaskRCNN().prediction(image:)`` →
``detections`` (100,6) + ``mask`` (100,28,28); ``Classifier.prediction(feature_map:)`` →
``probabilities``, ``bounding_boxes``; ``Mask.prediction(feature_map:)`` → ``masks``.
``MaskRCNN.predict`` is the batched extension used by the bench / multi-GPU driver.
Nothing here computes: all work is ``mrcnn_*`` calls into libmaskrcnn_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from .config import MaskRCNNConfig

# f32s / f32x3: fp32 tensors, 2- / 3-part split-fp16 MFMA.  "default" = MRCNN_DEFAULT (include/maskrcnn_hip.h): what the artefact is prepared
# for — f32x3 with the stored exponents when `convert --calibrate` wrote them into MaskRCNN.mrcw, f32 otherwise — the mode a host that names no
# precision gets (`MaskRCNN()` of ViewController.swift:37).
DTYPES = {"default": _lib.DEFAULT, "f32": _lib.F32, "f16": _lib.F16, "f32s": _lib.F32S, "f32x3": _lib.F32X3}
DTYPE_NAMES = {v: k for k, v in DTYPES.items() if k != "default"}
STAGES = ["Trunk", "Proposal-Eval", "PyramidROIAlign-Eval", "TimeDistributedClassifierLayer-Eval", "Detection-Eval",
          "PyramidROIAlign-Eval-Mask", "TimeDistributedMask-Eval"]


class _Model:
    KIND = -1

    def __init__(self, path: str, max_batch: int, compute_dtype: int = _lib.DEFAULT):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().mrcnn_model_load(self.KIND, os.fspath(path).encode(), int(max_batch), compute_dtype,
                                               C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mrcnn_model_destroy(h)
            except Exception:
                pass
            self._h = None

    def get_int(self, key: str) -> int:
        v = C.c_int64(0)
        _lib.check(_lib.lib().mrcnn_model_get_int(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def enable_graph(self, on: bool = True):
        """hipGraph replay of predict's launch sequence (default off; see mrcnn_model_enable_graph)."""
        _lib.check(_lib.lib().mrcnn_model_enable_graph(self._h, int(on)))

    def set_stream(self, hip_stream: int):
        _lib.check(_lib.lib().mrcnn_model_set_stream(self._h, C.c_void_p(hip_stream)))


class MaskRCNN(_Model):
    """``MaskRCNN(path)``: loads MaskRCNN.mrcw; anchors / Classifier / Mask come from
    ``MaskRCNNConfig.defaultConfig()`` which must be set first (AppDelegate.swift:18-20)."""
    KIND = _lib.MODEL_MASKRCNN

    def __init__(self, path: str, max_batch: int = 1, compute_dtype: str = "default"):
        cfg = MaskRCNNConfig.defaultConfig()
        for name in ("anchorsURL", "compiledClassifierModelURL", "compiledMaskModelURL"):
            if getattr(cfg, name) is None:
                # the reference force-unwraps and crashes (ProposalLayer.swift:68); we raise
                raise _lib.MrcnnError(6, f"MaskRCNNConfig.defaultConfig().{name} must be set before loading MaskRCNN")
        super().__init__(path, max_batch, DTYPES[compute_dtype])
        self.compute_dtype = DTYPE_NAMES[self.get_int("compute_dtype")]      # the RESOLVED mode ("default" never stays)
        self.compute_dtype_defaulted = bool(self.get_int("compute_dtype_defaulted"))
        self.max_batch = max_batch
        self.image_height = self.get_int("image_height")
        self.image_width = self.get_int("image_width")
        self.max_detections = self.get_int("max_detections")
        self.max_proposals = self.get_int("max_proposals")
        self.num_classes = self.get_int("num_classes")
        self.mask_size = self.get_int("mask_size")

    # -- reference-shaped single-image call ---------------------------------------------------------
    def prediction(self, image: np.ndarray) -> Dict[str, np.ndarray]:
        """image (H,W,3) uint8 RGB → {"detections": (maxDet,6), "mask": (maxDet,28,28)}."""
        det, mask = self.predict(image[None])
        return {"detections": det[0], "mask": mask[0]}

    # -- batched extension --------------------------------------------------------------------------
    def predict(self, images):
        """images (B,H,W,3) uint8 — numpy (host) or torch CUDA tensor (device, results stay on the GPU)."""
        if isinstance(images, np.ndarray):
            imgs = np.ascontiguousarray(images, dtype=np.uint8)
            B, H, W, _ = imgs.shape
            det = np.empty((B, self.max_detections, 6), dtype=np.float32)
            mask = np.empty((B, self.max_detections, self.mask_size, self.mask_size), dtype=np.float32)
            _lib.check(_lib.lib().mrcnn_maskrcnn_predict(self._h, imgs.ctypes.data, B, H, W, _lib.HOST, det.ctypes.data,
                                                         mask.ctypes.data))
            return det, mask
        import torch
        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous()
        B, H, W, _ = images.shape
        det = torch.empty((B, self.max_detections, 6), dtype=torch.float32, device=images.device)
        mask = torch.empty((B, self.max_detections, self.mask_size, self.mask_size), dtype=torch.float32, device=images.device)
        _lib.check(_lib.lib().mrcnn_maskrcnn_predict(self._h, images.data_ptr(), B, H, W, _lib.DEVICE, det.data_ptr(),
                                                     mask.data_ptr()))
        return det, mask

    def predict_scalefit(self, images: np.ndarray):
        """images (B,h,w,3) uint8 of ANY size: `.scaleFit` letterbox fused into the pre-processing kernel, then predict.
        Boxes are normalized in the letterboxed frame (evaluate.unletterbox_boxes / mrcnn_unletterbox_boxes map them back)."""
        imgs = np.ascontiguousarray(images, dtype=np.uint8)
        B, h, w, _ = imgs.shape
        det = np.empty((B, self.max_detections, 6), dtype=np.float32)
        mask = np.empty((B, self.max_detections, self.mask_size, self.mask_size), dtype=np.float32)
        _lib.check(_lib.lib().mrcnn_maskrcnn_predict_scalefit(self._h, imgs.ctypes.data, B, h, w, _lib.HOST, det.ctypes.data, mask.ctypes.data))
        return det, mask

    def predict_into(self, images, det, mask, sync: bool = True):
        """Device tensors in, pre-allocated device tensors out (bench loop: no allocation, optional no sync)."""
        B, H, W, _ = images.shape
        fn = _lib.lib().mrcnn_maskrcnn_predict if sync else None
        if sync:
            _lib.check(fn(self._h, images.data_ptr(), B, H, W, _lib.DEVICE, det.data_ptr(), mask.data_ptr()))
        else:
            _lib.check(_lib.lib().mrcnn_maskrcnn_predict_async(self._h, images.data_ptr(), B, H, W, det.data_ptr(), mask.data_ptr()))

    def predict_host_into(self, images: np.ndarray, det: np.ndarray, mask: np.ndarray):
        """Host buffers in and out without allocation (pinned memory gives asynchronous PCIe copies): the H2D of the batch
        and the D2H of the records are part of the call — bench.py's h2d_included leg."""
        B, H, W, _ = images.shape
        _lib.check(_lib.lib().mrcnn_maskrcnn_predict(self._h, images.ctypes.data, B, H, W, _lib.HOST, det.ctypes.data, mask.ctypes.data))

    def submit(self, images: np.ndarray):
        """Pipelined host entry (mrcnn_maskrcnn_submit): the H2D copy of this batch overlaps the predict of the previous one.
        `images` must stay alive (and unchanged) until the matching collect()."""
        B, H, W, _ = images.shape
        _lib.check(_lib.lib().mrcnn_maskrcnn_submit(self._h, images.ctypes.data, B, H, W))

    def collect(self, det: np.ndarray, mask: np.ndarray) -> int:
        """Results of the OLDEST submission into the given host buffers; returns its batch size."""
        n = C.c_int(0)
        _lib.check(_lib.lib().mrcnn_maskrcnn_collect(self._h, det.ctypes.data, mask.ctypes.data, C.byref(n)))
        return int(n.value)

    def check_range(self) -> bool:
        """After predict_into(sync=False): synchronises the model's stream and reports whether the last predict left
        the fp16 range (its results are then not valid) — the async counterpart of the error predict() raises."""
        t = C.c_int(0)
        _lib.check(_lib.lib().mrcnn_model_check_range(self._h, C.byref(t)))
        return bool(t.value)

    # -- scale-aware split (include/maskrcnn_hip.h: mrcnn_model_calibrate_split) ----------------------
    def calibrate_split(self, images, apply: bool = True) -> Dict[str, int]:
        """One calibration predict on `images` (numpy (B,H,W,3) uint8 or a CUDA tensor): a power-of-two pre-scale per tensor
        group so that the split modes carry every activation like fp32 whatever the checkpoint's scale.  apply=False only
        diagnoses.  Returns the totals of the pass (see split_report for the per-group view)."""
        if isinstance(images, np.ndarray):
            imgs = np.ascontiguousarray(images, dtype=np.uint8)
            B, H, W, _ = imgs.shape
            _lib.check(_lib.lib().mrcnn_model_calibrate_split(self._h, imgs.ctypes.data, B, H, W, _lib.HOST, int(apply)))
        else:
            B, H, W, _ = images.shape
            _lib.check(_lib.lib().mrcnn_model_calibrate_split(self._h, images.data_ptr(), B, H, W, _lib.DEVICE, int(apply)))
        return {k: self.get_int(k) for k in ("split_small_inputs", "split_inexact_inputs", "split_inputs_counted", "split_min_exponent",
                                             "split_max_exponent", "split_calibrated")}

    def split_report(self):
        """[{name, exponent, fixed, absmax, small_inputs, inexact_inputs, inputs_counted}] per tensor group."""
        out = []
        st = _lib.SplitGroupStat()
        for i in range(self.get_int("split_groups")):
            _lib.check(_lib.lib().mrcnn_model_split_group_stat(self._h, i, C.byref(st)))
            out.append({"name": st.name.decode(), "exponent": int(st.exponent), "fixed": bool(st.fixed), "absmax": float(st.absmax),
                        "small_inputs": int(st.small_inputs), "inexact_inputs": int(st.inexact_inputs), "inputs_counted": int(st.inputs_counted)})
        return out

    @property
    def split_exponents(self) -> np.ndarray:
        n = C.c_int(0)
        _lib.check(_lib.lib().mrcnn_model_get_split_exponents(self._h, None, 0, C.byref(n)))
        e = np.zeros(n.value, np.int32)
        _lib.check(_lib.lib().mrcnn_model_get_split_exponents(self._h, e.ctypes.data, e.size, C.byref(n)))
        return e

    @split_exponents.setter
    def split_exponents(self, exps):
        e = np.ascontiguousarray(exps, dtype=np.int32)
        _lib.check(_lib.lib().mrcnn_model_set_split_exponents(self._h, e.ctypes.data, e.size))

    # -- parity / profiling hooks -------------------------------------------------------------------
    def read_tensor(self, name: str, image_index: int = 0) -> np.ndarray:
        cnt = C.c_int64(0)
        L = _lib.lib()
        st = L.mrcnn_model_read_tensor(self._h, name.encode(), image_index, None, 0, C.byref(cnt))
        if cnt.value <= 0:
            _lib.check(st)
        buf = np.empty(cnt.value, dtype=np.float32)
        _lib.check(L.mrcnn_model_read_tensor(self._h, name.encode(), image_index, buf.ctypes.data, buf.size, C.byref(cnt)))
        return buf

    def conv_profile_enable(self, on: bool = True):
        _lib.check(_lib.lib().mrcnn_model_conv_profile_enable(self._h, int(on)))

    def conv_profile(self):
        """{tile: (launches, total_ms, total_algorithmic_flops)} since conv_profile_enable()."""
        out = {}
        for tile, name in enumerate(("128x128", "128x64", "128x32", "128x128w4", "256x256pp", "128xNhalo", "128x256tail", "bneck", "c3h")):
            n, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
            _lib.check(_lib.lib().mrcnn_model_conv_profile_get(self._h, tile, C.byref(n), C.byref(ms), C.byref(fl)))
            out[name] = (int(n.value), float(ms.value), float(fl.value))
        return out

    def conv_profile_bytes(self):
        """{tile: total ALGORITHMIC bytes} of the launches conv_profile() counts (every operand across HBM once)."""
        out = {}
        for tile, name in enumerate(("128x128", "128x64", "128x32", "128x128w4", "256x256pp", "128xNhalo", "128x256tail", "bneck", "c3h")):
            b = C.c_double(0)
            _lib.check(_lib.lib().mrcnn_model_conv_profile_bytes(self._h, tile, C.byref(b)))
            out[name] = float(b.value)
        return out

    def conv_profile_groups(self):
        """{"backbone" | "other": (launches, total_ms, total_algorithmic_flops)} — conv1 + res2..res5 against everything else."""
        out = {}
        for grp, name in ((1, "backbone"), (0, "other")):
            n, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
            _lib.check(_lib.lib().mrcnn_model_conv_profile_group(self._h, grp, C.byref(n), C.byref(ms), C.byref(fl)))
            out[name] = (int(n.value), float(ms.value), float(fl.value))
        return out

    def conv_profile_shapes(self):
        """[(M, N, K, tile, launches, total_ms, total_algorithmic_flops, total_algorithmic_bytes)] per distinct GEMM shape."""
        L = _lib.lib()
        n = C.c_int(0)
        _lib.check(L.mrcnn_model_conv_profile_shapes(self._h, None, 0, C.byref(n)))
        buf = (_lib.ConvShapeStat * max(n.value, 1))()
        _lib.check(L.mrcnn_model_conv_profile_shapes(self._h, buf, n.value, C.byref(n)))
        return [(r.M, r.N, r.K, r.tile, r.launches, r.total_ms, r.total_flops, r.total_bytes) for r in buf[:n.value]]

    def enable_timing(self, on: bool = True):
        _lib.check(_lib.lib().mrcnn_model_enable_timing(self._h, int(on)))

    def stage_ms(self) -> Dict[str, float]:
        out = {}
        for s in STAGES:
            v = C.c_float(0)
            _lib.check(_lib.lib().mrcnn_model_stage_ms(self._h, s.encode(), C.byref(v)))
            out[s] = float(v.value)
        return out


class Classifier(_Model):
    KIND = _lib.MODEL_CLASSIFIER

    def __init__(self, path: str, max_rows: int = 1000, compute_dtype: str = "default"):
        super().__init__(path, max_rows, DTYPES[compute_dtype])
        self.num_classes = self.get_int("num_classes")

    def prediction(self, feature_map: np.ndarray) -> Dict[str, np.ndarray]:
        """feature_map (256,7,7) or (n,256,7,7) float32 CHW → probabilities (n,nc), bounding_boxes (n,nc*4)."""
        fm = np.ascontiguousarray(feature_map, dtype=np.float32)
        single = fm.ndim == 3
        if single:
            fm = fm[None]
        n = fm.shape[0]
        probs = np.empty((n, self.num_classes), dtype=np.float32)
        bbox = np.empty((n, self.num_classes * 4), dtype=np.float32)
        _lib.check(_lib.lib().mrcnn_classifier_predict(self._h, fm.ctypes.data, n, _lib.HOST, probs.ctypes.data, bbox.ctypes.data))
        if single:
            return {"probabilities": probs[0], "bounding_boxes": bbox[0]}
        return {"probabilities": probs, "bounding_boxes": bbox}


class Mask(_Model):
    KIND = _lib.MODEL_MASK

    def __init__(self, path: str, max_rows: int = 100, compute_dtype: str = "default"):
        super().__init__(path, max_rows, DTYPES[compute_dtype])
        self.num_classes = self.get_int("num_classes")

    def prediction(self, feature_map: np.ndarray) -> Dict[str, np.ndarray]:
        """feature_map (256,14,14) or (n,256,14,14) float32 CHW → masks (n,nc,28,28)."""
        fm = np.ascontiguousarray(feature_map, dtype=np.float32)
        single = fm.ndim == 3
        if single:
            fm = fm[None]
        n = fm.shape[0]
        masks = np.empty((n, self.num_classes, 2 * fm.shape[2], 2 * fm.shape[3]), dtype=np.float32)
        _lib.check(_lib.lib().mrcnn_mask_predict(self._h, fm.ctypes.data, n, _lib.HOST, masks.ctypes.data))
        return {"masks": masks[0] if single else masks}


def load_maskrcnn(model_dir: str, max_batch: int = 1, compute_dtype: str = "default") -> MaskRCNN:
    """Sets MaskRCNNConfig from a directory holding MaskRCNN.mrcw / Classifier.mrcw / Mask.mrcw /
    anchors.bin (the four artefacts of DownloadCommand.swift:10-32) and loads the main model —
    the sequence of EvaluateCommand.swift:144-153."""
    cfg = MaskRCNNConfig.defaultConfig()
    cfg.anchorsURL = os.path.join(model_dir, "anchors.bin")
    cfg.compiledClassifierModelURL = os.path.join(model_dir, "Classifier.mrcw")
    cfg.compiledMaskModelURL = os.path.join(model_dir, "Mask.mrcw")
    return MaskRCNN(os.path.join(model_dir, "MaskRCNN.mrcw"), max_batch=max_batch, compute_dtype=compute_dtype)
