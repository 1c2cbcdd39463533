#!/usr/bin/env python
"""Where the per-tile fixed cost of the ping-pong kernel goes: 1x1 conv M = 8*256*256, N = 512, Cin = 128 (KT = 2), fp16."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
for k, v in ((b"conv_pp_min_tiles", 1), (b"conv_pp_min_fill", 0), (b"conv_pp_min_kt", 1)):
    L.check(lib.mrcnn_debug_set(k, v))
def run(cin, n, pp, dbg, iters=10):
    L.check(lib.mrcnn_debug_set(b"conv_pp", pp)); L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", dbg))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(8, 256, 256, cin, n, 1, 1, iters, L.F16, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3
for n in (512, 256):
    for cin in (128, 2048):
        print(f"N={n} Cin={cin}: " + "  ".join(f"{name} {run(cin, n, pp, dbg):7.1f}" for name, pp, dbg in
              [("128-row", 0, 0), ("pp", 1, 0), ("pp-nostores", 1, 32), ("pp-noepilogue", 1, 64), ("barriers-only", 1, 28),
               ("barriers-only-nostores", 1, 28 + 32), ("barriers-only-noepilogue", 1, 28 + 64)]), flush=True)
