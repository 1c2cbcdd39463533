#!/usr/bin/env python
"""Fixed cost vs per-K-tile cost of the conv kernels: 1x1 convs M = 8*256*256, N = 512, Cin = 256..2048 (KT = 4..32)."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
L.check(lib.mrcnn_debug_set(b"conv_pp_min_tiles", 1)); L.check(lib.mrcnn_debug_set(b"conv_pp_min_fill", 0)); L.check(lib.mrcnn_debug_set(b"conv_pp_min_kt", 1))
L.check(lib.mrcnn_debug_set(b"conv_pp_min_kt", 1))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
def run(cin, pp, dbg, iters=10):
    L.check(lib.mrcnn_debug_set(b"conv_pp", pp)); L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", dbg))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(8, 256, 256, cin, N, 1, 1, iters, L.F16, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3
cins = [128, 256, 512, 1024, 2048]
for name, pp, dbg in [("128-row", 0, 0), ("pp", 1, 0), ("pp nodma nords", 1, 12), ("pp nomma nords", 1, 24), ("pp nomma nodma", 1, 20), ("pp only barriers", 1, 28)]:
    ts = [run(c, pp, dbg) for c in cins]
    kts = [c // 64 for c in cins]
    slope = (ts[-1] - ts[-2]) / (kts[-1] - kts[-2])
    fixed = ts[-1] - slope * kts[-1]
    print(f"{name:20s} " + " ".join(f"KT{k}:{t:7.1f}" for k, t in zip(kts, ts)) + f" | per K-tile {slope:6.2f} us, fixed {fixed:7.1f} us  (tiles/CU: {8*256*256//256*N//256/256:.0f})", flush=True)
