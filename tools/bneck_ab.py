"""A/B of the fused identity bottleneck (kernels_bneck.hip) against the three launches, at the trunk's shapes (fp16 mode).
usage: python tools/bneck_ab.py [batch] [iters]"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
T = importlib.import_module("test_gpu_bneck")

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print(f"batch {batch}, {iters} iterations; us per block (fused | three launches), TFLOP/s algorithmic")
for name, C, H in [s for s in (("C4", 256, 64), ("C3", 128, 128), ("C2", 64, 256)) if not __import__("os").environ.get("BNECK_ONLY") or s[0] in __import__("os").environ["BNECK_ONLY"]]:
    if C == 512:
        continue
    x, w1, w2, w3, bn = T.make(C, batch, H, H, seed=1)
    fl = 2.0 * batch * H * H * 17 * C * C
    a, ms_f = T.bneck(x, w1, w2, w3, bn, True, iters)
    b, ms_3 = T.bneck(x, w1, w2, w3, bn, False, iters)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print(f"{name}  C={C:3d} {H}x{H}: fused {ms_f * 1e3:8.1f} us ({fl / ms_f / 1e9:7.1f} TF)   three {ms_3 * 1e3:8.1f} us ({fl / ms_3 / 1e9:7.1f} TF)   x{ms_3 / ms_f:.2f}   bit-identical: {same}", flush=True)

if not __import__("os").environ.get("BNECK_ONLY") or "C2a" in __import__("os").environ["BNECK_ONLY"]:
    x, w1, w2, w3, ws, bn = T.make_first(64, batch, 256, 256, seed=1)
    fl = 2.0 * batch * 256 * 256 * 18 * 64 * 64
    a, ms_f = T.bneck_first(x, w1, w2, w3, ws, bn, True, iters)
    b, ms_4 = T.bneck_first(x, w1, w2, w3, ws, bn, False, iters)
    print(f"C2a C= 64 256x256 (entry block): fused {ms_f * 1e3:8.1f} us ({fl / ms_f / 1e9:7.1f} TF)   four  {ms_4 * 1e3:8.1f} us ({fl / ms_4 / 1e9:7.1f} TF)   x{ms_4 / ms_f:.2f}   bit-identical: {np.array_equal(a.view(np.uint32), b.view(np.uint32))}", flush=True)
