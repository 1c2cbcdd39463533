#!/usr/bin/env python
"""One conv shape through mrcnn_bench_conv_dtype: conv_one.py batch h w cin cout k stride [iters] [f32|f16|f32s]"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
a = [int(x) for x in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
dt = {"f32": L.F32, "f16": L.F16, "f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[9] if len(sys.argv) > 9 else "f32"]
ms, fl = C.c_float(0), C.c_double(0)
L.check(L.lib().mrcnn_bench_conv_dtype(*a, iters, dt, C.byref(ms), C.byref(fl)))
print(f"{a} {ms.value*1e3:.1f} us {fl.value/ms.value/1e9:.1f} TFLOP/s")
