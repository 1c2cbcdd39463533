#!/usr/bin/env python
"""hipGraph replay vs plain stream launches for predict: graph_probe.py [iters]  (R101 1024², synthetic weights)."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import importlib
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mask-rcnn-coreml_amd")
models = importlib.import_module("mask-rcnn-coreml_amd.models")
weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = pkg.ModelConfig()
d = tempfile.mkdtemp()
weights.save_synthetic_models(d, cfg, seed=0)
for dtype in ("f32", "f16"):
    for B in (1, 8):
        m = models.load_maskrcnn(d, max_batch=B, compute_dtype=dtype)
        img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)).cuda()
        det = torch.empty((B, m.max_detections, 6), device="cuda")
        mask = torch.empty((B, m.max_detections, m.mask_size, m.mask_size), device="cuda")
        res = {}
        for on in (False, True):
            m.enable_graph(on)
            for _ in range(3):
                m.predict_into(img, det, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                m.predict_into(img, det, mask)
            dt = (time.perf_counter() - t0) / iters * 1e3
            res[on] = (dt, det.cpu().numpy().copy(), mask.cpu().numpy().copy())
        same = np.array_equal(res[False][1], res[True][1]) and np.array_equal(res[False][2], res[True][2])
        print(f"{dtype} batch {B}: stream {res[False][0]:.3f} ms   graph {res[True][0]:.3f} ms   identical={same}   "
              f"graph_launches={m.get_int('graph_launches')}", flush=True)
        del m
