#!/usr/bin/env python
"""A/B of one policy knob (mrcnn_debug_set) on single layers through the micro-benchmark hook, interleaved rounds in one process.
usage: knob_ab.py <knob> <v0> <v1> <dtype> [rounds] [iters] [shape-set]      shape-set: pw (the long-K 1x1 layers; default) | all | hbm (the short-K 1x1 layers)"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
PW = [  # (name, batch, h, w, cin, cout, k, stride)
    ("C4 2a 1024->256 @64 b8", 8, 64, 64, 1024, 256, 1, 1),
    ("C4 2a 1024->256 @64 b4", 4, 64, 64, 1024, 256, 1, 1),
    ("C4 2a 1024->256 @64 b2", 2, 64, 64, 1024, 256, 1, 1),
    ("C4 2a 1024->256 @64 b1", 1, 64, 64, 1024, 256, 1, 1),
    ("C5 2a 2048->512 @32 b8", 8, 32, 32, 2048, 512, 1, 1),
    ("C5 2a 2048->512 @32 b1", 1, 32, 32, 2048, 512, 1, 1),
    ("C5 sc 1024->2048 @32 b8", 8, 32, 32, 1024, 2048, 1, 1),
    ("C5 sc 1024->2048 @32 b1", 1, 32, 32, 1024, 2048, 1, 1),
    ("P5 lat 2048->256 @32 b8", 8, 32, 32, 2048, 256, 1, 1),
    ("P5 lat 2048->256 @32 b1", 1, 32, 32, 2048, 256, 1, 1),
    ("fc1 12544->1024 x8000", 8000, 1, 1, 12544, 1024, 1, 1),
    ("fc1 12544->1024 x1000", 1000, 1, 1, 12544, 1024, 1, 1),
    ("fc2 1024->1024 x8000", 8000, 1, 1, 1024, 1024, 1, 1),
    ("fc2 1024->1024 x1000", 1000, 1, 1, 1024, 1024, 1, 1),
]
ALL = PW + [
    ("C4 2c 256->1024 @64 b8", 8, 64, 64, 256, 1024, 1, 1),
    ("C4 2c 256->1024 @64 b1", 1, 64, 64, 256, 1024, 1, 1),
    ("C3 2a 512->128 @128 b8", 8, 128, 128, 512, 128, 1, 1),
    ("C4 3x3 256->256 @64 b8", 8, 64, 64, 256, 256, 3, 1),
    ("C4 3x3 256->256 @64 b1", 1, 64, 64, 256, 256, 3, 1),
    ("C5 3x3 512->512 @32 b1", 1, 32, 32, 512, 512, 3, 1),
    ("C3 3x3 128->128 @128 b8", 8, 128, 128, 128, 128, 3, 1),
    ("mask 3x3 256->256 x100", 100, 14, 14, 256, 256, 3, 1),
    ("mask 3x3 256->256 x200", 200, 14, 14, 256, 256, 3, 1),
    ("mask 3x3 256->256 x800", 800, 14, 14, 256, 256, 3, 1),
    ("C2 3x3 64->64 @256 b8", 8, 256, 256, 64, 64, 3, 1),
    ("C2 3x3 64->64 @256 b1", 1, 256, 256, 64, 64, 3, 1),
]
HBM = [  # the short-K 1x1 layers whose roof is HBM in the fp32-tensor modes (round 6: roofline.by_tile_class[*].bound)
    ("C2 2a 256->64 @256 b8", 8, 256, 256, 256, 64, 1, 1),
    ("C2 2c 64->256 @256 b8", 8, 256, 256, 64, 256, 1, 1),
    ("P2 lat 256->256 @256 b8", 8, 256, 256, 256, 256, 1, 1),
    ("C3 2a 512->128 @128 b8", 8, 128, 128, 512, 128, 1, 1),
    ("C3 2c 128->512 @128 b8", 8, 128, 128, 128, 512, 1, 1),
    ("P3 lat 512->256 @128 b8", 8, 128, 128, 512, 256, 1, 1),
    ("C4 2a 1024->256 @64 b8", 8, 64, 64, 1024, 256, 1, 1),
    ("C4 2c 256->1024 @64 b8", 8, 64, 64, 256, 1024, 1, 1),
]
knob, v0, v1 = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
DT = {"f32": L.F32, "f16": L.F16, "f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[4]]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
shapes = {"all": ALL, "hbm": HBM}.get(sys.argv[7] if len(sys.argv) > 7 else "pw", PW)


def run(shape, v):
    L.check(lib.mrcnn_debug_set(knob, v))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(*shape[1:], iters, DT, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3, fl.value / ms.value / 1e9


print(f"# {sys.argv[1]} {v0} vs {v1}, {sys.argv[4]}, best of {rounds} rounds x {iters} launches")
for sh in shapes:
    res = {v0: [], v1: []}
    for r in range(rounds):
        for v in (v0, v1):
            res[v].append(run(sh, v))
    b0, b1 = min(res[v0]), min(res[v1])
    print(f"{sh[0]:28s} {v0}: {b0[0]:8.1f} us {b0[1]:7.1f} TF | {v1}: {b1[0]:8.1f} us {b1[1]:7.1f} TF | x{b0[0] / b1[0]:.2f}", flush=True)
