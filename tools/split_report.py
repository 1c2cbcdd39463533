#!/usr/bin/env python
"""The scale-aware split's calibration report on the headline model (R101+FPN 1024^2, synthetic weights): per tensor group the
maximum |a| of the calibration batch, the exponent chosen (max |a| * 2^e in [2^11, 2^12)) and the diagnostic counters
(include/maskrcnn_hip.h: mrcnn_model_calibrate_split).  usage: split_report.py [dtype] [images]"""
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

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = pkg.ModelConfig(architecture="resnet101")
d = tempfile.mkdtemp(prefix="mrcnn_split_")
weights.save_synthetic_models(d, cfg, seed=0)
m = models.load_maskrcnn(d, max_batch=n, compute_dtype=dtype)
imgs = np.random.default_rng(1).integers(0, 256, (n, 1024, 1024, 3), dtype=np.uint8)
tot = m.calibrate_split(imgs)
print(f"# {dtype}, {n} calibration images; totals: {tot}")
print(f"{'group':28s} {'exp':>4s} {'max |a|':>12s} {'stored max':>11s} {'counted':>12s} {'< 2^-8 max':>12s} {'< 0.5 stored':>13s}")
for g in m.split_report():
    if g["fixed"]:
        continue
    print(f"{g['name']:28s} {g['exponent']:4d} {g['absmax']:12.5g} {g['absmax'] * 2.0 ** g['exponent']:11.1f} {g['inputs_counted']:12d} "
          f"{g['small_inputs']:12d} {g['inexact_inputs']:13d}")
