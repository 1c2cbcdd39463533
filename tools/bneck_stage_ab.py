"""A/B of C4's identity blocks as ONE launch (kernels_bneck.hip, STAGE form) against one fused launch per block (fp16 mode).
usage: python tools/bneck_stage_ab.py [batch] [layers] [iters] [rounds]
Both forms include the same 134-MB input copy + two memsets per run (mrcnn_bottleneck_stage_nhwc); `copy` is that overhead measured alone (0 layers are not
accepted, so it is the 1-block run minus one block)."""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
T = importlib.import_module("test_gpu_bneck")

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
x, w1, w2, w3, bn = T.make_stage(n, batch, 64, 64, seed=1)
fl = 2.0 * batch * 64 * 64 * 17 * 256 * 256 * n
best = {0: 1e9, 1: 1e9}
same = True
for r in range(rounds):
    a, ms1, f1 = T.bneck_stage(x, w1, w2, w3, bn, 1, iters)
    b, ms0, f0 = T.bneck_stage(x, w1, w2, w3, bn, 0, iters)
    same = same and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and f1 == 0 and f0 == 0
    best[1] = min(best[1], ms1); best[0] = min(best[0], ms0)
    print(f"round {r}: stage {ms1 * 1e3:8.1f} us   per-block {ms0 * 1e3:8.1f} us", flush=True)
_, ms_one, _ = T.bneck_stage(x, w1[:1], w2[:1], w3[:1], [b_[:1] for b_ in bn], 0, iters)
print(f"C4 batch {batch}, {n} blocks: ONE launch {best[1] * 1e3:8.1f} us ({best[1] * 1e3 / n:6.1f} per block, {fl / best[1] / 1e9:7.1f} TF)   per-block launches {best[0] * 1e3:8.1f} us "
      f"({best[0] * 1e3 / n:6.1f} per block, {fl / best[0] / 1e9:7.1f} TF)   x{best[0] / best[1]:.3f}   bit-identical: {same}   (1-block run incl. input copy: {ms_one * 1e3:.1f} us)")
