#!/usr/bin/env python
"""Ablations of the ping-pong conv kernel on one shape (measurement-only switches, see kernels_conv_pp.hip a.dbg):
usage: pp_ablate.py [shape index] — prints us / TF for: 128-row kernel, ping-pong, and the ping-pong kernel with
1 no setprio, 2 no stagger, 4 no DMA in the main loop, 8 no fragment reads, 16 no MFMAs (and combinations)."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
SH = [(8, 256, 256, 256, 512, 3, 1), (8, 128, 128, 256, 256, 3, 1), (8, 64, 64, 256, 256, 3, 1)]
sh = SH[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
DT = {"f16": L.F16, "f32s": L.F32S, "f32x3": L.F32X3}[sys.argv[2] if len(sys.argv) > 2 else "f16"]
L.check(lib.mrcnn_debug_set(b"conv_pp_split", 1))
L.check(lib.mrcnn_debug_set(b"conv_pp_min_tiles", 1)); L.check(lib.mrcnn_debug_set(b"conv_pp_min_fill", 0)); L.check(lib.mrcnn_debug_set(b"conv_pp_min_kt", 1))

def run(pp, dbg, iters=10):
    L.check(lib.mrcnn_debug_set(b"conv_pp", pp)); L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", dbg))
    ms, fl = C.c_float(0), C.c_double(0)
    L.check(lib.mrcnn_bench_conv_dtype(*sh, iters, DT, C.byref(ms), C.byref(fl)))
    return ms.value * 1e3, fl.value / ms.value / 1e9

print("shape", sh)
for rnd in range(2):
    for name, pp, dbg in [("128-row", 0, 0), ("pp", 1, 0), ("pp noprio", 1, 1), ("pp nostagger", 1, 2), ("pp nostagger noprio", 1, 3),
                          ("pp nodma", 1, 4), ("pp nords", 1, 8), ("pp nodma nords", 1, 12), ("pp nomma", 1, 16),
                          ("pp nomma nords", 1, 24), ("pp nomma nodma", 1, 20), ("pp only barriers", 1, 28), ("pp cheap split (split modes)", 1, 128), ("pp cheap split nodma nords", 1, 128 + 12)]:
        us, tf = run(pp, dbg)
        print(f"  {name:22s} {us:9.1f} us {tf:8.1f} TF", flush=True)
