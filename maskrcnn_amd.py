"""Alias: ``import maskrcnn_amd`` == the hyphenated package ``mask-rcnn-coreml_amd``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("mask-rcnn-coreml_amd")
sys.modules[__name__] = _pkg
