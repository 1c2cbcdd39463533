#!/usr/bin/env python
"""Generates tests/golden/layers_v1.npz: seeded inputs and the CPU oracle's outputs for the custom
layers.  The reference ships no golden vectors and cannot run here, so these are
SELF-generated fixtures: they freeze the oracle's behaviour (which tests/test_oracle_kat.py pins
against hand-computed answers and brute-force restatements).  Re-run only on a deliberate change:
    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

pkg = importlib.import_module("mask-rcnn-coreml_amd")
anchors_mod = importlib.import_module("mask-rcnn-coreml_amd.anchors")

rng = np.random.default_rng(20260928)
cfg = pkg.ModelConfig(input_image_shape=(128, 128, 3))
anchors = anchors_mod.generate_anchors(cfg)
A = anchors.shape[0]
fg = (np.round(rng.random(A) * 256) / 256).astype(np.float32)          # ties included
probs = np.stack([1 - fg, fg], 1).astype(np.float32)
deltas = rng.standard_normal((A, 4)).astype(np.float32)
rois, dbg = orc.proposal_layer(probs, deltas, anchors, 300, 64, 0.7, debug=True)
fm = [rng.standard_normal((8, s, s)).astype(np.float32) for s in (32, 16, 8, 4)]
pooled = orc.pyramid_roi_align(rois[:16], fm, 7, 128, 128)
n = 96
y1 = rng.random(n) * 0.8; x1 = rng.random(n) * 0.8
det_rois = np.stack([y1, x1, y1 + rng.random(n) * 0.2, x1 + rng.random(n) * 0.2], 1).astype(np.float32)
det_cls = np.zeros((n, 6), np.float32)
det_cls[:, :4] = rng.standard_normal((n, 4))
det_cls[:, 4] = rng.integers(0, 5, n)
det_cls[:, 5] = rng.random(n)
detections = orc.detection_layer(det_rois, det_cls, 16, 0.7, 0.3)
exp_x = np.concatenate([rng.standard_normal(256) * 2, [0, 1, -1, 88.0, -100.0]]).astype(np.float32)
rows = np.array([0, 1, 2, 3071, 3072, A - 1])
np.savez_compressed(os.path.join(os.path.dirname(__file__), "layers_v1.npz"),
                    probs=probs, deltas=deltas, rois=rois, topk_idx_head=dbg["topk_idx"][:64],
                    fm0=fm[0], fm1=fm[1], fm2=fm[2], fm3=fm[3], pooled=pooled,
                    det_rois=det_rois, det_cls=det_cls, detections=detections,
                    exp_x=exp_x, exp_y=orc.expf(exp_x), anchor_rows=rows, anchor_values=anchors[rows])
print("wrote layers_v1.npz")
