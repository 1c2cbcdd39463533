#!/usr/bin/env python
"""Ablations of the halo kernel on one shape (measurement only: results are wrong under most of them).
usage: halo_ablate.py [dtype] [shape...]"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
dt = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
DT = {"f32s": L.F32S, "f32x3": L.F32X3}[dt]
shape = [int(x) for x in sys.argv[2:9]] if len(sys.argv) > 8 else [8, 256, 256, 256, 512, 3, 1]
names = {0: "shipped", 32: "slab loads from a cache-hot source", 64: "no slab loads (split + writes of stale registers)", 128: "no split (raw bits written)",
         256: "no plane writes (loads + split only)", 64 + 128: "no loads, no split (writes only)", 64 + 256: "no loads, no writes (split only)", 128 + 256: "loads only", 1: "no filter DMA", 2: "no barriers", 4: "no fragment reads", 8: "no MFMAs", 16: "no slab staging",
         17: "no DMA, no staging", 3: "no DMA, no barriers", 19: "no DMA/staging/barriers", 23: "MFMAs only", 12: "no reads, no MFMAs (DMA + staging + barriers)"}
for rnd in range(2):
    for dbg, nm in names.items():
        L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", dbg))
        ms, fl = C.c_float(0), C.c_double(0)
        L.check(lib.mrcnn_bench_conv_dtype(*shape, 10, DT, C.byref(ms), C.byref(fl)))
        if rnd == 1:
            print(f"dbg {dbg:3d} {nm:46s} {ms.value * 1e3:8.1f} us {fl.value / ms.value / 1e9:7.1f} TF", flush=True)
L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", 0))
