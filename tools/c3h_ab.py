"""A/B of the fp16 3x3 kernel (kernels_conv3x3_h.hip, "conv_c3h" 1) against the 128-row / ping-pong kernels ("conv_c3h" 0) through the
micro-benchmark hook, interleaved rounds in one process.   usage: python tools/c3h_ab.py [rounds] [iters]"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
SHAPES = [  # (name, batch, h, w, cin, cout, k, stride)
    ("RPN 3x3 256->512 @256 b8", 8, 256, 256, 256, 512, 3, 1),
    ("FPN 3x3 256->256 @256 b8", 8, 256, 256, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @128 b8", 8, 128, 128, 256, 512, 3, 1),
    ("FPN 3x3 256->256 @128 b8", 8, 128, 128, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @64 b8", 8, 64, 64, 256, 512, 3, 1),
    ("C4 3x3 256->256 @64 b8", 8, 64, 64, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @32 b8", 8, 32, 32, 256, 512, 3, 1),
    ("C5 3x3 512->512 @32 b8", 8, 32, 32, 512, 512, 3, 1),
    ("RPN 3x3 256->512 @256 b1", 1, 256, 256, 256, 512, 3, 1),
    ("FPN 3x3 256->256 @256 b1", 1, 256, 256, 256, 256, 3, 1),
    ("RPN 3x3 256->512 @64 b1", 1, 64, 64, 256, 512, 3, 1),
    ("mask 3x3 256->256 x800", 800, 14, 14, 256, 256, 3, 1),
]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def run(shape, v):
    L.check(lib.mrcnn_debug_set(b"conv_c3h", 2 if v else 0))      # 2: every eligible layer on the kernel
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(*shape[1:], iters, L.F16, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3, fl.value / ms.value / 1e9


print(f"# conv_c3h 0 vs 1, f16, best of {rounds} rounds x {iters} launches")
for sh in SHAPES:
    res = {0: [], 1: []}
    for r in range(rounds):
        for v in (0, 1):
            res[v].append(run(sh, v))
    b0, b1 = min(res[0]), min(res[1])
    print(f"{sh[0]:28s} old: {b0[0]:8.1f} us {b0[1]:7.1f} TF | c3h: {b1[0]:8.1f} us {b1[1]:7.1f} TF | x{b0[0] / b1[0]:.2f}", flush=True)
