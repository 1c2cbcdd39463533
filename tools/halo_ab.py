#!/usr/bin/env python
"""A/B of the 3x3 layers of the split modes: the 128-row kernel family vs the persistent halo kernel (kernels_conv_halo.hip),
interleaved rounds in one process.  usage: halo_ab.py [rounds] [iters] [dtype]"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
SHAPES = [  # (name, batch, h, w, cin, cout, k, stride)
    ("RPN 3x3 256->512 @256", 8, 256, 256, 256, 512, 3, 1),
    ("FPN 3x3 256->256 @256", 8, 256, 256, 256, 256, 3, 1),
    ("mask 3x3 256->256 800x14x14", 800, 14, 14, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @128", 8, 128, 128, 256, 512, 3, 1),
    ("FPN 3x3 256->256 @128", 8, 128, 128, 256, 256, 3, 1),
    ("C4 3x3 256->256 @64", 8, 64, 64, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @64", 8, 64, 64, 256, 512, 3, 1),
    ("C5 3x3 512->512 @32", 8, 32, 32, 512, 512, 3, 1),
    ("C3 3x3 128->128 @128", 8, 128, 128, 128, 128, 3, 1),
    ("C2 3x3 64->64 @256", 8, 256, 256, 64, 64, 3, 1),
    ("C4 3x3 256->256 @64 batch 1", 1, 64, 64, 256, 256, 3, 1),
]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DT = {"f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[3] if len(sys.argv) > 3 else "f32x3"]


def run(shape, halo):
    L.check(lib.mrcnn_debug_set(b"conv_halo", halo))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(*shape[1:], iters, DT, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3, fl.value / ms.value / 1e9


for sh in SHAPES:
    res = {0: [], 1: []}
    for r in range(rounds):
        for halo in (0, 1):
            res[halo].append(run(sh, halo))
    b0 = min(res[0]); b1 = min(res[1])
    print(f"{sh[0]:30s} 128-row {b0[0]:8.1f} us {b0[1]:7.1f} TF | halo {b1[0]:8.1f} us {b1[1]:7.1f} TF | x{b0[0] / b1[0]:.2f}", flush=True)
L.check(lib.mrcnn_debug_set(b"conv_halo", 1))
