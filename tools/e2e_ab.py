#!/usr/bin/env python
"""Whole-model A/B of a kernel-policy knob, interleaved rounds in one process (cdna guide rule 24):
   e2e_ab.py <dtype> <key> <value A> <value B> [rounds] [steps]      e.g.  e2e_ab.py f32x3 conv_tn4 0 -1"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import importlib, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mask-rcnn-coreml_amd")
models = importlib.import_module("mask-rcnn-coreml_amd.models")
weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
L = importlib.import_module("mask-rcnn-coreml_amd._lib")
dtype, key, va, vb = sys.argv[1], sys.argv[2].encode(), int(sys.argv[3]), int(sys.argv[4])
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
B = int(os.environ.get("BATCH", "8"))
cfg = pkg.ModelConfig(architecture="resnet101", input_image_shape=(1024, 1024, 3), num_classes=81)
d = tempfile.mkdtemp(prefix="mrcnn_ab_")
weights.save_synthetic_models(d, cfg, seed=0, forced_load=True)
m = models.load_maskrcnn(d, max_batch=B, compute_dtype=dtype)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
images = torch.from_numpy(rng.integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)).to(dev)
det = torch.empty((B, m.max_detections, 6), dtype=torch.float32, device=dev)
mask = torch.empty((B, m.max_detections, m.mask_size, m.mask_size), dtype=torch.float32, device=dev)
def run(v):
    L.check(L.lib().mrcnn_debug_set(key, v))
    m.predict_into(images, det, mask, sync=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.predict_into(images, det, mask, sync=True)
    return (time.perf_counter() - t0) / steps * 1e3
ra, rb = [], []
for r in range(rounds):
    ra.append(run(va)); rb.append(run(vb))
print(f"{dtype} {key.decode()}={va}: {min(ra):.3f} ms/step ({B * 1e3 / min(ra):.1f} img/s)   {key.decode()}={vb}: {min(rb):.3f} ms/step ({B * 1e3 / min(rb):.1f} img/s)   "
      f"B/A speed x{min(ra) / min(rb):.3f}   rounds A {[round(x, 2) for x in ra]} B {[round(x, 2) for x in rb]}")
