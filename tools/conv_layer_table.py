#!/usr/bin/env python
"""Per-GEMM-shape table of the conv family inside a real predict (live HIP events through the C ABI):
   conv_layer_table.py [f32|f16|f32s|f32x3] [steps] [batch]   — BASELINE configs[1] (R101 1024², batch 8), synthetic weights."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import importlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mask-rcnn-coreml_amd")
models = importlib.import_module("mask-rcnn-coreml_amd.models")
weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = pkg.ModelConfig()
d = tempfile.mkdtemp()
weights.save_synthetic_models(d, cfg, seed=0)
m = models.load_maskrcnn(d, max_batch=B, compute_dtype=dtype)
img = np.random.default_rng(0).integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)
m.predict(img)
m.conv_profile_enable(True)
for _ in range(steps):
    m.predict(img)
rows = m.conv_profile_shapes()
tot = sum(r[5] for r in rows)
# MB = ALGORITHMIC bytes per launch (every operand across HBM once), GB/s against them; bound = the larger of flops / MFMA peak and bytes / 8 TB/s
peak = {"f32": 157.3e12, "f16": 2500e12, "f32s": 1250e12, "f32x3": 833.3e12}[dtype]
print(f"{'M':>8} {'N':>5} {'K':>6} tile  n/step   us/launch  TFLOP/s  share      MB    GB/s bound")
for M, N, K, tile, n, ms, fl, by in sorted(rows, key=lambda r: -r[5]):
    bound = "hbm" if by / 8e12 > fl / peak else "mfma"
    print(f"{M:8d} {N:5d} {K:6d} {('128', '64', '32', '128w4', 'pp256', 'halo', 'tail', 'bneck', 'c3h')[tile]:>5} {n / steps:7.1f} {ms / n * 1e3:11.1f} {fl / ms / 1e9:8.1f} {ms / tot * 100:6.1f}% {by / n / 1e6:7.1f} {by / ms / 1e6:7.0f} {bound:>5}")
m.conv_profile_enable(False)
import time
m.predict(img)
t0 = time.perf_counter()
for _ in range(5):
    m.predict(img)
print(f"batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per predict (host buffers in/out)")
print(f"total conv {tot / steps:.2f} ms/step, {sum(r[6] for r in rows) / tot / 1e9:.1f} TFLOP/s")
