#!/usr/bin/env python
"""Micro-benchmark of the conv kernel family on the trunk's layer shapes (through the C ABI)."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
SHAPES = [  # (name, batch, h, w, cin, cout, k, stride)
    ("C2 1x1 64->256 @256", 8, 256, 256, 64, 256, 1, 1),
    ("C2 3x3 64->64 @256", 8, 256, 256, 64, 64, 3, 1),
    ("C3 3x3 128->128 @128", 8, 128, 128, 128, 128, 3, 1),
    ("C4 1x1 1024->256 @64", 8, 64, 64, 1024, 256, 1, 1),
    ("C4 3x3 256->256 @64", 8, 64, 64, 256, 256, 3, 1),
    ("C4 1x1 256->1024 @64", 8, 64, 64, 256, 1024, 1, 1),
    ("C5 3x3 512->512 @32", 8, 32, 32, 512, 512, 3, 1),
    ("FPN 3x3 256->256 @256", 8, 256, 256, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @256", 8, 256, 256, 256, 512, 3, 1),
    ("RPN 3x3 256->512 @128", 8, 128, 128, 256, 512, 3, 1),
]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dtype = {"f32": L.F32, "f16": L.F16, "f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
for name, b, h, w, ci, co, k, s in SHAPES:
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(b, h, w, ci, co, k, s, iters, dtype, C.byref(ms), C.byref(fl)))
    print(f"{name:28s} {ms.value*1e3:9.1f} us  {fl.value/ms.value/1e9:7.1f} TFLOP/s", flush=True)
