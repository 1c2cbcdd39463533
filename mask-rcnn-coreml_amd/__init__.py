"""MI355X-native Mask-RCNN inference hot path behind the reference's API surface.

Drop-in for the hot path of edouardlp/Mask-RCNN-CoreML: the ``MaskRCNN`` /
``Classifier`` / ``Mask`` three-model surface, the five custom layers and the ``anchors.bin``
layout, executed by hand-written HIP (gfx950) kernels in ``csrc/`` through the C ABI declared in
``include/maskrcnn_hip.h``.  The package name contains a hyphen (it mirrors the reference's
``Mask-RCNN-CoreML`` target); import it with ``importlib.import_module("mask-rcnn-coreml_amd")`` or
through the ``maskrcnn_amd`` alias module at the repo root.

Submodules are imported lazily so that pure-host pieces (config, anchors, weights) work without the
native library; anything that computes fails loudly if ``libmaskrcnn_hip.so`` is missing.
"""
from .config import ModelConfig, MaskRCNNConfig  # noqa: F401

__all__ = ["ModelConfig", "MaskRCNNConfig"]


def __getattr__(name):
    import importlib
    lazy = {
        "MaskRCNN": ".models", "Classifier": ".models", "Mask": ".models",
        "ProposalLayer": ".layers", "PyramidROIAlignLayer": ".layers",
        "TimeDistributedClassifierLayer": ".layers", "DetectionLayer": ".layers",
        "TimeDistributedMaskLayer": ".layers", "MLMultiArray": ".layers",
        "Detection": ".detection", "IOU": ".detection", "COCO": ".coco",
    }
    if name in lazy:
        return getattr(importlib.import_module(lazy[name], __name__), name)
    raise AttributeError(name)
