#!/usr/bin/env python
"""A/B of the fp16 conv kernels on the trunk's big layer shapes: the 128-row kernel vs the 256-row ping-pong kernel,
interleaved rounds in one process (cdna guide §5.4 rule 24).  usage: conv_ab.py [rounds] [iters] [min_tiles] [dtype] [pp_dbg]
(pp_dbg: a third column with the ping-pong kernel under that conv_pp_dbg value, e.g. 512 = the other DMA addressing form)"""
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
    ("C4 1x1 1024->256 @64", 8, 64, 64, 1024, 256, 1, 1),
    ("C4 1x1 256->1024 @64", 8, 64, 64, 256, 1024, 1, 1),
    ("C3 1x1 512->128 @128", 8, 128, 128, 512, 128, 1, 1),
    ("FPN 1x1 256->256 @256", 8, 256, 256, 256, 256, 1, 1),
    ("C5 3x3 512->512 @32", 8, 32, 32, 512, 512, 3, 1),
]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
min_tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 1
DT = {"f16": L.F16, "f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[4] if len(sys.argv) > 4 else "f16"]
alt_dbg = int(sys.argv[5]) if len(sys.argv) > 5 else None


def run(shape, pp, dbg=0):
    L.check(lib.mrcnn_debug_set(b"conv_pp", pp))
    L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", dbg))
    L.check(lib.mrcnn_debug_set(b"conv_pp_min_tiles", min_tiles))
    L.check(lib.mrcnn_debug_set(b"conv_pp_min_fill", 0))
    L.check(lib.mrcnn_debug_set(b"conv_pp_min_kt", 1))
    L.check(lib.mrcnn_debug_set(b"conv_pp_split", 1))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(*shape[1:], iters, DT, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3, fl.value / ms.value / 1e9


for sh in SHAPES:
    res = {0: [], 1: [], 2: []}
    for r in range(rounds):
        for pp in (0, 1):
            res[pp].append(run(sh, pp))
        if alt_dbg is not None:
            res[2].append(run(sh, 1, alt_dbg))
    b0 = min(res[0]); b1 = min(res[1])
    alt = ""
    if alt_dbg is not None:
        b2 = min(res[2])
        alt = f" | dbg {alt_dbg} {b2[0]:8.1f} us {b2[1]:7.1f} TF x{b0[0] / b2[0]:.2f}"
    print(f"{sh[0]:30s} 128-row {b0[0]:8.1f} us {b0[1]:7.1f} TF | ping-pong {b1[0]:8.1f} us {b1[1]:7.1f} TF | x{b0[0] / b1[0]:.2f}{alt}", flush=True)
