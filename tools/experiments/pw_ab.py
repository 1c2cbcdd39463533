#!/usr/bin/env python
"""A/B of the 1x1 layers of the split modes: the 128-row kernel vs the persistent pointwise kernel (kernels_conv_pw.hip), interleaved
rounds in one process.  MRCNN_BENCH_RESIDUAL=1 adds the shortcut.  usage: pw_ab.py [rounds] [iters] [dtype]"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
SHAPES = [  # (name, batch, h, w, cin, cout, k, stride)
    ("C4 1x1 256->1024 @64", 8, 64, 64, 256, 1024, 1, 1),
    ("C4 1x1 1024->256 @64", 8, 64, 64, 1024, 256, 1, 1),
    ("C3 1x1 128->512 @128", 8, 128, 128, 128, 512, 1, 1),
    ("C3 1x1 512->128 @128", 8, 128, 128, 512, 128, 1, 1),
    ("FPN 1x1 256->256 @256", 8, 256, 256, 256, 256, 1, 1),
    ("C5 1x1 512->2048 @32", 8, 32, 32, 512, 2048, 1, 1),
    ("C5 1x1 2048->512 @32", 8, 32, 32, 2048, 512, 1, 1),
    ("box fc1 12544->1024 x8000", 1, 1, 8000, 12544, 1024, 1, 1),
    ("box fc2 1024->1024 x8000", 1, 1, 8000, 1024, 1024, 1, 1),
    ("mask deconv-like 256->1024 x156800", 800, 14, 14, 256, 1024, 1, 1),
]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DT = {"f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[3] if len(sys.argv) > 3 else "f32x3"]


def run(shape, pw):
    L.check(lib.mrcnn_debug_set(b"conv_pw", pw))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(*shape[1:], iters, DT, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3, fl.value / ms.value / 1e9


for sh in SHAPES:
    res = {0: [], 2: []}
    for r in range(rounds):
        for pw in (0, 2):
            res[pw].append(run(sh, pw))
    b0 = min(res[0]); b1 = min(res[2])
    print(f"{sh[0]:36s} 128-row {b0[0]:8.1f} us {b0[1]:7.1f} TF | pointwise {b1[0]:8.1f} us {b1[1]:7.1f} TF | x{b0[0] / b1[0]:.2f}", flush=True)
L.check(lib.mrcnn_debug_set(b"conv_pw", 1))
