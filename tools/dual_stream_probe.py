#!/usr/bin/env python
"""Does the chip run two half batches on two streams faster than one whole batch on one?  (memory-bound 1x1 layers of one half under the
   power-bound 3x3 layers of the other).  Two model handles (own arena + stream each), one host thread per handle, free-running.
   dual_stream_probe.py <dtype> [batch] [steps]"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import importlib, os, sys, tempfile, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("mask-rcnn-coreml_amd")
models = importlib.import_module("mask-rcnn-coreml_amd.models")
weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = pkg.ModelConfig(architecture="resnet101", input_image_shape=(1024, 1024, 3), num_classes=81)
d = tempfile.mkdtemp(prefix="mrcnn_dual_")
weights.save_synthetic_models(d, cfg, seed=0, forced_load=True)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
images = torch.from_numpy(rng.integers(0, 256, (B, 1024, 1024, 3), dtype=np.uint8)).to(dev)

MASKS = {"lohi": ["ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0", "0,0,0,0,ffffffff,ffffffff,ffffffff,ffffffff"],
         "evenodd": [",".join(["55555555"] * 8), ",".join(["aaaaaaaa"] * 8)],
         "bytes": [",".join(["00ff00ff"] * 8), ",".join(["ff00ff00"] * 8)],
         "half16": [",".join(["0000ffff"] * 8), ",".join(["ffff0000"] * 8)]}
masking = os.environ.get("DUAL_MASK", "")

def handle(nb, which=-1):
    if masking and which >= 0: os.environ["MRCNN_CU_MASK_PROBE"] = MASKS[masking][which]
    else: os.environ.pop("MRCNN_CU_MASK_PROBE", None)
    m = models.load_maskrcnn(d, max_batch=nb, compute_dtype=dtype)
    det = torch.empty((nb, m.max_detections, 6), dtype=torch.float32, device=dev)
    mask = torch.empty((nb, m.max_detections, m.mask_size, m.mask_size), dtype=torch.float32, device=dev)
    return m, det, mask

def loop(m, img, det, mask, n):
    for _ in range(n):
        m.predict_into(img, det, mask, sync=True)

whole = handle(B)
loop(whole[0], images, whole[1], whole[2], 3)
t0 = time.perf_counter(); loop(whole[0], images, whole[1], whole[2], steps); t_whole = (time.perf_counter() - t0) / steps
print(f"{dtype} one handle, batch {B}: {t_whole * 1e3:.3f} ms/step = {B / t_whole:.1f} images/s", flush=True)
for parts in ((2,) if masking else (2, 4)):
    nb = B // parts
    hs = [handle(nb, i) for i in range(parts)]
    for i, h in enumerate(hs): loop(h[0], images[i * nb:(i + 1) * nb], h[1], h[2], 3)
    # one handle alone at the part's batch
    t0 = time.perf_counter(); loop(hs[0][0], images[:nb], hs[0][1], hs[0][2], steps); t_one = (time.perf_counter() - t0) / steps
    ths = [threading.Thread(target=loop, args=(h[0], images[i * nb:(i + 1) * nb], h[1], h[2], steps)) for i, h in enumerate(hs)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    t_all = (time.perf_counter() - t0) / steps
    print(f"{dtype} {parts} handles x batch {nb}, one thread each, free-running: {t_all * 1e3:.3f} ms per round of {B} = {B / t_all:.1f} images/s "
          f"(one such handle alone: {t_one * 1e3:.3f} ms = {nb / t_one:.1f} images/s)   vs whole x{t_whole / t_all:.3f}", flush=True)
    del hs
