#!/usr/bin/env python
"""Distance of every compute mode's trunk to a float64 evaluation of the same graph.

For N full-size images (default 2: R101, 1024², the bench's synthetic weights) the pyramid levels P2..P5 and the RPN
outputs of the HIP engine in modes f32 / f32x3 / f32s / f16 are compared with oracle.network.trunk_fp64 (torch CPU,
float64 — fp16-exact weights widened), next to the torch-CPU fp32 network itself.  Error = max |x - x64| / max |x64|
per tensor.  Writes JSON (default gpurun_out/fp64_trunk_parity.json; the committed copy lives under profiles/).
"""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import argparse
import importlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fp64_trunk_parity.json"))
    args = ap.parse_args()
    pkg = importlib.import_module("mask-rcnn-coreml_amd")
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    weights = importlib.import_module("mask-rcnn-coreml_amd.weights")
    from oracle.network import load_oracle_model
    cfg = pkg.ModelConfig(architecture=args.arch, input_image_shape=(args.size, args.size, 3))
    d = tempfile.mkdtemp(prefix="mrcnn_fp64_")
    weights.save_synthetic_models(d, cfg, seed=0, forced_load=True)
    images = np.random.default_rng(5).integers(0, 256, (args.images, args.size, args.size, 3), dtype=np.uint8)
    om = load_oracle_model(d, cfg)
    names = ["P2", "P3", "P4", "P5", "rpn_deltas", "rpn_probs"]
    shapes = cfg.feature_shapes()
    A = cfg.num_anchors()

    def rel(x, ref):
        return float(np.abs(x.astype(np.float64) - ref).max() / max(1e-300, np.abs(ref).max()))

    ref, cpu32 = [], []
    for i in range(args.images):
        pyr, probs, deltas = om.trunk_fp64(images[i:i + 1])
        ref.append([p[0] for p in pyr] + [deltas[0], probs[0]])
        pyr, probs, deltas = om.trunk(images[i:i + 1])
        cpu32.append([p[0] for p in pyr] + [deltas[0], probs[0]])
    out = {"images": args.images, "size": args.size, "arch": args.arch, "error": "max|x-x64|/max|x64| per tensor, worst image",
           "modes": {}}
    out["modes"]["torch_cpu_fp32"] = {n: max(rel(cpu32[i][k], ref[i][k]) for i in range(args.images)) for k, n in enumerate(names)}
    for mode in ("f32", "f32x3", "f32s", "f16"):
        m = models.load_maskrcnn(d, max_batch=args.images, compute_dtype=mode)
        m.predict(images)
        errs = {n: 0.0 for n in names}
        for i in range(args.images):
            for l in range(4):
                h, w = shapes[l]
                x = m.read_tensor(f"P{l + 2}", i).reshape(h, w, 256).transpose(2, 0, 1)
                errs[names[l]] = max(errs[names[l]], rel(x, ref[i][l]))
            errs["rpn_deltas"] = max(errs["rpn_deltas"], rel(m.read_tensor("rpn_deltas", i).reshape(A, 4), ref[i][4]))
            errs["rpn_probs"] = max(errs["rpn_probs"], rel(m.read_tensor("rpn_probs", i).reshape(A, 2), ref[i][5]))
        out["modes"][mode] = errs
        del m
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    for mode, e in out["modes"].items():
        print(f"{mode:16s} " + "  ".join(f"{n} {v:.2e}" for n, v in e.items()))


if __name__ == "__main__":
    main()
