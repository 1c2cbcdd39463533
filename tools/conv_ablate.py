#!/usr/bin/env python
"""Ablations of the 128-row kernels on one shape (measurement build: make -C mask-rcnn-coreml_amd/csrc ablate; results are wrong
under the switches).  usage: MRCNN_HIP_LIB=.../libmaskrcnn_hip_ablate.so [MRCNN_BENCH_RESIDUAL=1] conv_ablate.py dtype b h w cin cout k stride"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
lib = L.lib()
DT = {"f32s": L.F32S, "f32x3": L.F32X3, "f16": L.F16, "f32": L.F32}[sys.argv[1]]
shape = [int(x) for x in sys.argv[2:9]]
names = {0: "shipped", 256: "no MFMAs", 512: "no DMA in the main loop", 768: "no MFMAs, no DMA", 1024: "no epilogue", 2048: "no residual loads",
         4096: "no stores", 6144: "no residual, no stores", 1024 + 256: "no MFMAs, no epilogue (DMA + barriers)", 1024 + 512: "no DMA, no epilogue (MFMAs)"}
L.check(lib.mrcnn_debug_set(b"conv_pp", 0))
if os.environ.get("STAGGER"):
    names = {0: "shipped"}
    for us in (2, 4, 6, 8, 10, 12, 16, 20, 30):
        names[(us * 100) << 16] = f"second slots start {us} us late"
for rnd in range(2):
    for dbg, nm in names.items():
        L.check(lib.mrcnn_debug_set(b"conv_pp_dbg", dbg))
        ms, fl = C.c_float(0), C.c_double(0)
        L.check(lib.mrcnn_bench_conv_dtype(*shape, 20, DT, C.byref(ms), C.byref(fl)))
        if rnd == 1:
            print(f"dbg {dbg:5d} {nm:44s} {ms.value * 1e3:8.1f} us {fl.value / ms.value / 1e9:7.1f} TF", flush=True)
