"""anchors.bin: generator, reader and validator.

``anchors.bin`` is a named drop-in contract of the reference: a raw little-endian float32 dump
``anchors.tofile(path)`` (``Sources/maskrcnn/Python/Conversion/task.py:173,176``) of shape (A, 4),
normalized (y1, x1, y2, x2), read back as ``Float`` by ``ProposalLayer`` (``ProposalLayer.swift:68,
146-149``).  The generator itself lives in the un-vendored, un-pinned third-party package
``edouardlp/Mask-RCNN-Keras`` (``Conversion/requirements.txt:4``); what follows restates the published
Matterport Mask R-CNN algorithm it derives from (``generate_pyramid_anchors`` + ``norm_boxes``):
level-major P2→P6, then y, then x, then ratio; pixel boxes normalized by
``(box - [0,0,1,1]) / [h-1, w-1, h-1, w-1]``.  Order and normalization are therefore an ASSUMPTION
(SURVEY.md §8b "unpinned"); MaskRCNNConfig.swift:14 carries a TODO to "generate the anchors on
demand" — this module is that generator.
"""
from __future__ import annotations

import math
import os
import numpy as np

from .config import ModelConfig


def _level_anchors(scale, ratios, shape, feature_stride, anchor_stride):
    scales, ratios = np.meshgrid(np.array([scale], dtype=np.float64), np.array(ratios, dtype=np.float64))
    scales = scales.flatten()
    ratios = ratios.flatten()
    heights = scales / np.sqrt(ratios)
    widths = scales * np.sqrt(ratios)
    shifts_y = np.arange(0, shape[0], anchor_stride) * feature_stride
    shifts_x = np.arange(0, shape[1], anchor_stride) * feature_stride
    shifts_x, shifts_y = np.meshgrid(shifts_x, shifts_y)
    box_widths, box_centers_x = np.meshgrid(widths, shifts_x)
    box_heights, box_centers_y = np.meshgrid(heights, shifts_y)
    centers = np.stack([box_centers_y, box_centers_x], axis=2).reshape([-1, 2])
    sizes = np.stack([box_heights, box_widths], axis=2).reshape([-1, 2])
    return np.concatenate([centers - 0.5 * sizes, centers + 0.5 * sizes], axis=1)


def generate_anchors(config: ModelConfig) -> np.ndarray:
    """(A, 4) float32 normalized (y1, x1, y2, x2), level-major."""
    h, w = config.image_height, config.image_width
    levels = []
    for scale, stride, shape in zip(config.anchor_scales, config.backbone_strides, config.feature_shapes()):
        levels.append(_level_anchors(scale, config.anchor_ratios, shape, stride, config.anchor_stride))
    a = np.concatenate(levels, axis=0)
    scale = np.array([h - 1, w - 1, h - 1, w - 1], dtype=np.float64)
    shift = np.array([0, 0, 1, 1], dtype=np.float64)
    return ((a - shift) / scale).astype(np.float32)


def write_anchors_bin(path: str, config: ModelConfig) -> np.ndarray:
    a = generate_anchors(config)
    a.astype("<f4").tofile(path)
    return a


def read_anchors_bin(path: str, expected_count: int | None = None) -> np.ndarray:
    """Reads anchors.bin and validates the layout (size multiple of 16 B, finite, y2>y1, x2>x1)."""
    n = os.path.getsize(path)
    if n % 16 != 0:
        raise ValueError(f"{path}: size {n} is not a multiple of 16 bytes (A x 4 float32)")
    a = np.fromfile(path, dtype="<f4").reshape(-1, 4)
    if expected_count is not None and a.shape[0] != expected_count:
        raise ValueError(f"{path}: {a.shape[0]} anchors, expected {expected_count}")
    if not np.isfinite(a).all():
        raise ValueError(f"{path}: non-finite anchor coordinates")
    if not ((a[:, 2] > a[:, 0]).all() and (a[:, 3] > a[:, 1]).all()):
        raise ValueError(f"{path}: anchors must satisfy y2>y1 and x2>x1")
    return a
