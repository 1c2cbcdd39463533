#!/usr/bin/env python
"""Soak: the same batch of 8 (the headline batch: the fp16 mode then runs its large layers on the ping-pong kernel) through predict N times per compute mode; every output must be bit-identical to the first
(races in the DMA ring / barriers / atomics would show up as run-to-run differences).  soak_determinism.py [iters] [batch] [modes]
batch 1 exercises the single-image forms: shared-tile K chunks (whichever block arrives last folds the partial sums), the halo kernel's latency form."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mask-rcnn-coreml_amd")
models = importlib.import_module("mask-rcnn-coreml_amd.models")
weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "f32x3", "f32s", "f16"]
cfg = pkg.ModelConfig()
d = tempfile.mkdtemp()
weights.save_synthetic_models(d, cfg, seed=0)
img = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)).cuda()
bad = 0
for mode in MODES:
    m = models.load_maskrcnn(d, max_batch=B, compute_dtype=mode)
    det = torch.empty((B, m.max_detections, 6), device="cuda")
    mask = torch.empty((B, m.max_detections, m.mask_size, m.mask_size), device="cuda")
    m.predict_into(img, det, mask)
    d0, m0 = det.clone(), mask.clone()
    p0 = torch.from_numpy(m.read_tensor("P2", B - 1)).cuda()
    diffs = 0
    for i in range(iters):
        m.predict_into(img, det, mask)
        if not (torch.equal(det, d0) and torch.equal(mask, m0)):
            diffs += 1
        if i % 50 == 49 and not torch.equal(torch.from_numpy(m.read_tensor("P2", B - 1)).cuda(), p0):
            diffs += 1
    print(f"{mode}: batch {B}, {iters} repeats, {diffs} differing", flush=True)
    bad += diffs
    del m
sys.exit(1 if bad else 0)
