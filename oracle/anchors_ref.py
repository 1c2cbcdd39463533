"""Independent restatement of the anchor generator, for the oracle side only.

TEST INFRASTRUCTURE ONLY.  The product has two generators (mask-rcnn-coreml_amd/anchors.py: the published Matterport
meshgrid formulation; csrc/api.hip mrcnn_generate_anchors: C loops).  This third one shares no code with either: plain
Python loops over (level, y, x, ratio) with the closed form written out, float64 arithmetic, one rounding to float32 —
so that an ordering or normalisation mistake in the product cannot hide behind a shared helper.  What it restates is the
published Matterport Mask R-CNN algorithm of the un-vendored `edouardlp/Mask-RCNN-Keras` package the reference's
converter dumps to anchors.bin (Sources/maskrcnn/Python/Conversion/task.py:173-176; order and normalisation are an
ASSUMPTION): scales (32, 64, 128, 256, 512) on strides (4, 8, 16, 32, 64), ratios (0.5, 1, 2), anchor
stride 1; box = centre ± size/2 in pixels with centre = index · stride; normalised (box − (0,0,1,1)) / (H−1, W−1, H−1, W−1).
"""
import math

import numpy as np

SCALES = (32.0, 64.0, 128.0, 256.0, 512.0)
STRIDES = (4, 8, 16, 32, 64)
RATIOS = (0.5, 1.0, 2.0)


def generate(image_h: int, image_w: int) -> np.ndarray:
    rows = []
    for scale, stride in zip(SCALES, STRIDES):
        fh = int(math.ceil(image_h / stride))
        fw = int(math.ceil(image_w / stride))
        for y in range(fh):
            cy = float(y * stride)
            for x in range(fw):
                cx = float(x * stride)
                for ratio in RATIOS:
                    bh = scale / math.sqrt(ratio)
                    bw = scale * math.sqrt(ratio)
                    rows.append(((cy - 0.5 * bh - 0.0) / (image_h - 1), (cx - 0.5 * bw - 0.0) / (image_w - 1),
                                 (cy + 0.5 * bh - 1.0) / (image_h - 1), (cx + 0.5 * bw - 1.0) / (image_w - 1)))
    return np.asarray(rows, dtype=np.float64).astype(np.float32)
