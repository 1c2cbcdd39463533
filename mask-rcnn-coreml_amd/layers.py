"""The five custom layers, with the reference's own interface.

Host-side mirror of the ``MLCustomLayer`` plugins of the reference
(``Sources/Mask-RCNN-CoreML/{Proposal,PyramidROIAlign,TimeDistributedClassifier,Detection,
TimeDistributedMask}Layer.swift``): same class names, the same four methods
(``init(parameters:)``, ``setWeightData``, ``outputShapes(forInputShapes:)``,
``evaluate(inputs:outputs:)``), the same parameter keys (``Conversion/task.py:25-67``) and the same
error behaviour in spirit (Swift ``throws`` → ``MrcnnError``).  All arithmetic happens in
libmaskrcnn_hip.so through ``mrcnn_layer_*``; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib


class MLMultiArray:
    """5-D [sequence, batch, channel, height, width] Float32 array with element strides — the slice
    of Core ML's MLMultiArray the layers use (shape, strides, dataPointer).  Wraps a numpy array
    (host) or a torch CUDA tensor (device, used in place)."""

    def __init__(self, storage, shape: Sequence[int] | None = None, strides: Sequence[int] | None = None):
        self.storage = storage
        if isinstance(storage, np.ndarray):
            if storage.dtype != np.float32:
                raise TypeError("MLMultiArray: Float32 only (the layers assert it, ProposalLayer.swift:108)")
            if not storage.flags.c_contiguous:
                raise ValueError("MLMultiArray: storage must be C-contiguous; express views through `strides`")
            self.memspace = _lib.HOST
            self._ptr = storage.ctypes.data
            nat_shape = storage.shape
        else:  # torch tensor on the GPU
            import torch
            if not isinstance(storage, torch.Tensor) or storage.dtype != torch.float32 or not storage.is_contiguous():
                raise TypeError("MLMultiArray: expected a contiguous float32 numpy array or torch tensor")
            self.memspace = _lib.DEVICE if storage.is_cuda else _lib.HOST
            self._ptr = storage.data_ptr()
            nat_shape = tuple(storage.shape)
        if shape is None:
            if len(nat_shape) > 5:
                raise ValueError("more than 5 dimensions")
            shape = self._pad(nat_shape)
        self.shape = tuple(int(s) for s in shape)
        if strides is None:
            st, acc = [], 1
            for s in reversed(self.shape):
                st.append(acc)
                acc *= s
            strides = tuple(reversed(st))
        self.strides = tuple(int(s) for s in strides)

    @staticmethod
    def _pad(nat):
        # lower-rank arrays are right-aligned like Core ML does: (n, k) → [n, 1, k, 1, 1]
        if len(nat) == 1:
            return (nat[0], 1, 1, 1, 1)
        if len(nat) == 2:
            return (nat[0], 1, nat[1], 1, 1)
        if len(nat) == 3:
            return (1, 1) + tuple(nat)
        if len(nat) == 4:
            return (nat[0], 1) + tuple(nat[1:])
        return tuple(nat)

    def c_tensor(self) -> _lib.Tensor:
        t = _lib.Tensor()
        t.data = self._ptr
        t.dtype = _lib.F32
        t.memspace = self.memspace
        for i in range(5):
            t.shape[i] = self.shape[i]
            t.strides[i] = self.strides[i]
        return t


class _Layer:
    CLASS_NAME = ""

    def __init__(self, parameters: dict | None = None):
        self.parameters = dict(parameters or {})
        arr, n = _lib.make_params(self.parameters)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().mrcnn_layer_create(self.CLASS_NAME.encode(), arr, n, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mrcnn_layer_destroy(h)
            except Exception:
                pass
            self._h = None

    def setWeightData(self, weights: List[bytes]):
        _lib.check(_lib.lib().mrcnn_layer_set_weight_data(self._h, None, None, 0))   # no-op in all five layers

    def outputShapes(self, forInputShapes: Sequence[Sequence[int]]):
        n = len(forInputShapes)
        ins = ((C.c_int64 * 5) * n)()
        for i, s in enumerate(forInputShapes):
            for j in range(5):
                ins[i][j] = int(s[j])
        outs = ((C.c_int64 * 5) * 4)()
        n_out = C.c_int(0)
        _lib.check(_lib.lib().mrcnn_layer_output_shapes(self._h, ins, n, outs, C.byref(n_out)))
        return [[int(outs[i][j]) for j in range(5)] for i in range(n_out.value)]

    def evaluate(self, inputs: Sequence[MLMultiArray], outputs: Sequence[MLMultiArray]):
        ti = (_lib.Tensor * len(inputs))(*[a.c_tensor() for a in inputs])
        to = (_lib.Tensor * len(outputs))(*[a.c_tensor() for a in outputs])
        _lib.check(_lib.lib().mrcnn_layer_evaluate(self._h, ti, len(inputs), to, len(outputs)))


class ProposalLayer(_Layer):
    """ProposalLayer.swift:52 — parameters bboxStdDev_count/_i, preNMSMaxProposals, maxProposals,
    nmsIOUThreshold.  Reads MaskRCNNConfig.defaultConfig.anchorsURL at init (:68)."""
    CLASS_NAME = "ProposalLayer"


class PyramidROIAlignLayer(_Layer):
    """PyramidROIAlignLayer.swift:40 — parameters poolSize, imageWidth, imageHeight."""
    CLASS_NAME = "PyramidROIAlignLayer"


class TimeDistributedClassifierLayer(_Layer):
    """TimeDistributedClassifierLayer.swift:14 — runs Classifier over the ROI axis."""
    CLASS_NAME = "TimeDistributedClassifierLayer"


class DetectionLayer(_Layer):
    """DetectionLayer.swift:52 — parameters bboxStdDev_*, maxDetections, scoreThreshold, nmsIOUThreshold."""
    CLASS_NAME = "DetectionLayer"


class TimeDistributedMaskLayer(_Layer):
    """TimeDistributedMaskLayer.swift:14 — runs Mask over the detection axis, keeps the detected class."""
    CLASS_NAME = "TimeDistributedMaskLayer"
