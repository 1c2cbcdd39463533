import sys, importlib
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
T = importlib.import_module("test_gpu_bneck")
for (C, B, H, W) in [(256, 3, 16, 16), (256, 1, 16, 16), (256, 3, 8, 16), (128, 3, 32, 32), (64, 3, 64, 64), (256, 8, 64, 64)]:
    x, w1, w2, w3, bn = T.make(C, B, H, W, seed=3)
    a, _ = T.bneck(x, w1, w2, w3, bn, True)
    a2, _ = T.bneck(x, w1, w2, w3, bn, True)
    b, _ = T.bneck(x, w1, w2, w3, bn, False)
    d = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
    d2 = np.flatnonzero(a.view(np.uint32) != a2.view(np.uint32))
    print(C, B, H, W, "fused!=three:", d.size, "fused!=fused:", d2.size, (np.unravel_index(d[0], a.shape) if d.size else ""), flush=True)
    if d.size:
        idx = np.array(np.unravel_index(d, a.shape)).T
        print("  images", np.unique(idx[:, 0]), "rows", np.unique(idx[:, 1]), "cols", np.unique(idx[:, 2]), "ch range", idx[:, 3].min(), idx[:, 3].max())
