#!/usr/bin/env python
"""Probe: batch 8 as two half-batches on two model instances / streams vs one batch-8 predict."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("mask-rcnn-coreml_amd")
models = importlib.import_module("mask-rcnn-coreml_amd.models")
weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = pkg.ModelConfig()
d = tempfile.mkdtemp()
weights.save_synthetic_models(d, cfg, seed=0)
dev = torch.device("cuda", 0)
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (8, 1024, 1024, 3), dtype=np.uint8)).to(dev)
def run(nsplit, steps=8):
    b = 8 // nsplit
    ms = [models.load_maskrcnn(d, max_batch=b, compute_dtype=dt) for _ in range(nsplit)]
    det = [torch.empty((b, 100, 6), dtype=torch.float32, device=dev) for _ in range(nsplit)]
    msk = [torch.empty((b, 100, 28, 28), dtype=torch.float32, device=dev) for _ in range(nsplit)]
    ims = [img[i * b:(i + 1) * b].contiguous() for i in range(nsplit)]
    def step():
        for i in range(nsplit): ms[i].predict_into(ims[i], det[i], msk[i], sync=False)
        torch.cuda.synchronize()
    for _ in range(2): step()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    dtm = (time.perf_counter() - t0) / steps
    print(f"{dt} split {nsplit}: {dtm*1e3:.2f} ms/step  {8/dtm:.1f} img/s", flush=True)
    return torch.cat(det).cpu().numpy()
a = run(1); b = run(2); c = run(4)
print("identical results:", np.array_equal(a, b), np.array_equal(a, c))
