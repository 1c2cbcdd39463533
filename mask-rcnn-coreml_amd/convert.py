"""``maskrcnn convert``: Keras HDF5 weights + config.json → the artefacts the engine loads
(SURVEY.md §8f-1).

Counterpart of ``Sources/maskrcnn/ConvertCommand.swift`` (``--config``, ``--weights``,
``--output_dir``, :9-11; defaults ``model/config.json`` / ``model/weights.h5`` / ``products/``,
:36-70) and of ``Python/Conversion/task.py:120-178``, which writes ``MaskRCNN.mlmodel``,
``Mask.mlmodel``, ``Classifier.mlmodel`` and ``anchors.bin``.  Here the products are
``MaskRCNN.mrcw``, ``Classifier.mrcw``, ``Mask.mrcw`` and ``anchors.bin`` (generated on demand — the
reference's own TODO, ``task.py:160``, ``MaskRCNNConfig.swift:14``).

Weights come from a Matterport-layout Keras checkpoint (``README.md:83-85``: "only models trained
using Matterport's Mask-RCNN implementation are supported"): tensors are looked up by their Keras
``<layer>/<weight>`` names and re-laid-out from Keras to Core ML order, the same transposes
coremltools' Keras converter applies:

    Conv2D           kernel (kh, kw, I, O)  →  (O, I, kh, kw)
    Conv2DTranspose  kernel (kh, kw, O, I)  →  (I, O, kh, kw)
    Dense            kernel (I, O)          →  (O, I)
    BatchNorm        gamma, beta, moving_mean, moving_variance  →  gamma, beta, mean, variance

and cast to fp16 like ``task.py:90,102,114`` (``weights_dtype="f32"`` keeps them exact).  A missing
tensor or a shape that disagrees with config.json is an error naming the tensor; extra tensors in the
checkpoint (training-only layers) are reported, not fatal.  Host tooling — no GPU involved.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, List, Tuple

import numpy as np

from .anchors import write_anchors_bin
from .config import ModelConfig
from .hdf5 import read_keras_weights
from .weights import base_meta, tensor_shapes, write_mrcw

_BN_KERAS = {"gamma": "gamma", "beta": "beta", "mean": "moving_mean", "variance": "moving_variance"}
_DENSE = ("mrcnn_class_logits", "mrcnn_bbox_fc")
_DECONV = ("mrcnn_mask_deconv",)


class ConversionError(ValueError):
    pass


def keras_name(mrcw_name: str) -> str:
    layer, w = mrcw_name.split("/")
    return f"{layer}/{_BN_KERAS.get(w, w)}" if w in _BN_KERAS else mrcw_name


def from_keras_layout(mrcw_name: str, a: np.ndarray) -> np.ndarray:
    layer, w = mrcw_name.split("/")
    if w != "kernel":
        return a
    if layer in _DENSE:
        if a.ndim != 2:
            raise ConversionError(f"{mrcw_name}: Dense kernel must be 2-D, checkpoint has {a.shape}")
        return a.T
    if a.ndim != 4:
        raise ConversionError(f"{mrcw_name}: convolution kernel must be 4-D, checkpoint has {a.shape}")
    return a.transpose(3, 2, 0, 1)          # Conv2D → (O,I,kh,kw); Conv2DTranspose → (I,O,kh,kw)


def to_keras_layout(mrcw_name: str, a: np.ndarray) -> np.ndarray:
    """Inverse of ``from_keras_layout`` (used to write test checkpoints)."""
    layer, w = mrcw_name.split("/")
    if w != "kernel":
        return a
    return a.T if layer in _DENSE else a.transpose(2, 3, 1, 0)


def convert_tensors(keras: Dict[str, np.ndarray], cfg: ModelConfig, weights_dtype: str = "f16"):
    """→ ({"MaskRCNN": (meta, tensors), "Classifier": …, "Mask": …}, [unused checkpoint names])."""
    if weights_dtype not in ("f16", "f32"):
        raise ConversionError(f"weights_dtype {weights_dtype!r}: expected 'f16' or 'f32'")
    dt = np.dtype("<f2") if weights_dtype == "f16" else np.dtype("<f4")
    used = set()
    out = {}
    for kind, shapes in tensor_shapes(cfg).items():
        tensors: Dict[str, np.ndarray] = {}
        for name, shape in shapes.items():
            kn = keras_name(name)
            if kn not in keras:
                raise ConversionError(f"{kind}: checkpoint has no '{kn}' (architecture={cfg.architecture}, "
                                      f"num_classes={cfg.num_classes}); is config.json the one the model was trained with?")
            a = from_keras_layout(name, np.asarray(keras[kn]))
            if tuple(a.shape) != tuple(shape):
                raise ConversionError(f"{kind}: '{kn}' has shape {tuple(keras[kn].shape)} → {tuple(a.shape)}, "
                                      f"config.json implies {tuple(shape)}")
            if not np.all(np.isfinite(a)):
                raise ConversionError(f"{kind}: '{kn}' contains non-finite values")
            with np.errstate(over="ignore"):
                a16 = np.ascontiguousarray(a, dtype=np.float32).astype(dt)
            if dt.itemsize == 2 and not np.all(np.isfinite(a16)):
                raise ConversionError(f"{kind}: '{kn}' overflows fp16 (max |w| = {np.abs(a).max():g}); use weights_dtype='f32'")
            tensors[name] = a16
            used.add(kn)
        out[kind] = (base_meta(cfg, kind), tensors)
    return out, sorted(set(keras) - used)


def convert(config_path: str = "model/config.json", weights_path: str = "model/weights.h5",
            output_dir: str = "products", weights_dtype: str = "f16", verbose: bool = True) -> Dict[str, str]:
    """Writes the four products into ``output_dir`` and returns their paths."""
    cfg = ModelConfig.from_json(config_path) if config_path else ModelConfig()
    keras = read_keras_weights(weights_path)
    models, unused = convert_tensors(keras, cfg, weights_dtype)
    os.makedirs(output_dir, exist_ok=True)
    paths: Dict[str, str] = {}
    for kind, (meta, tensors) in models.items():
        paths[kind] = os.path.join(output_dir, f"{kind}.mrcw")
        write_mrcw(paths[kind], meta, tensors)
    paths["anchors"] = os.path.join(output_dir, "anchors.bin")
    write_anchors_bin(paths["anchors"], cfg)
    if verbose:
        n = sum(len(t) for _, t in models.values())
        print(f"converted {n} tensors ({weights_dtype}) from {weights_path} → {output_dir}; "
              f"{len(unused)} checkpoint tensors unused" + (f" (e.g. {unused[:3]})" if unused else ""))
    return paths


def load_calibration_images(path: str, H: int, W: int, limit: int = 8) -> np.ndarray:
    """Images for ``--calibrate``: a ``.npy`` file holding uint8 (N, H, W, 3), or a directory of image files (anything PIL opens),
    each letterboxed to H x W the way the evaluate loop hands images over (``.scaleFit``, EvaluateCommand.swift:152-157)."""
    if os.path.isfile(path) and path.endswith(".npy"):
        a = np.load(path)
        if a.dtype != np.uint8 or a.ndim != 4 or a.shape[1:] != (H, W, 3):
            raise ConversionError(f"{path}: expected uint8 (N, {H}, {W}, 3), found {a.dtype} {a.shape}")
        return np.ascontiguousarray(a[:limit])
    if not os.path.isdir(path):
        raise ConversionError(f"--calibrate {path}: neither a .npy file nor a directory of images")
    from PIL import Image
    from .evaluate import letterbox
    out = []
    for name in sorted(os.listdir(path)):
        try:
            img = np.asarray(Image.open(os.path.join(path, name)).convert("RGB"), dtype=np.uint8)
        except Exception:
            continue
        out.append(letterbox(img, H, W))
        if len(out) == limit:
            break
    if not out:
        raise ConversionError(f"--calibrate {path}: no readable image")
    return np.stack(out)


def store_split_exponents(model_dir: str, names: List[str], exponents: List[int]) -> str:
    """Rewrites ``MaskRCNN.mrcw`` with the exponent of every tensor group as ``split_exp.<group>`` metadata: mrcnn_model_load
    applies them in the split modes, so a drop-in ``MaskRCNN().prediction(image)`` (ViewController.swift:37) runs calibrated
    without an extra call.  Tensors and every other key are untouched."""
    from .weights import read_mrcw
    path = os.path.join(model_dir, "MaskRCNN.mrcw")
    meta, tensors = read_mrcw(path)
    meta = {k: v for k, v in meta.items() if not k.startswith("split_exp.")}
    for n, e in zip(names, exponents):
        meta["split_exp." + n] = int(e)
    write_mrcw(path, meta, {k: np.array(v) for k, v in tensors.items()})
    return path


def calibrate_artefact(model_dir: str, images: np.ndarray, compute_dtype: str = "f32x3", verbose: bool = True) -> Dict[str, int]:
    """Loads the converted artefacts on the GPU in a split mode, calibrates the scale-aware split on ``images`` (uint8 (N, H, W, 3))
    and stores the exponent vector in the artefact.  Needs a gfx950 device (the calibration IS a predict of the engine)."""
    from .models import load_maskrcnn
    n = int(images.shape[0])
    m = load_maskrcnn(model_dir, max_batch=n, compute_dtype=compute_dtype)
    totals = m.calibrate_split(images)
    rep = [g for g in m.split_report() if not g["fixed"]]
    del m
    store_split_exponents(model_dir, [g["name"] for g in rep], [g["exponent"] for g in rep])
    if verbose:
        print(f"calibrated the split on {n} images: exponents {totals['split_min_exponent']:+d} .. {totals['split_max_exponent']:+d} "
              f"over {len(rep)} tensor groups, stored in {os.path.join(model_dir, 'MaskRCNN.mrcw')}")
    return totals


def main(argv: List[str] = None) -> int:
    ap = argparse.ArgumentParser(prog="maskrcnn convert", description="Converts trained model to .mrcw")
    ap.add_argument("--config", default="model/config.json", help="Path to config JSON file")
    ap.add_argument("--weights", default="model/weights.h5", help="Path to HDF5 weights file")
    ap.add_argument("--output_dir", default="products", help="Path to output directory")
    ap.add_argument("--weights_dtype", default="f16", choices=("f16", "f32"))
    ap.add_argument("--calibrate", default=None, metavar="IMAGES",
                    help="a .npy of uint8 (N,H,W,3) images or a directory of image files: calibrates the scale-aware split of the "
                         "MRCNN_F32X3 / MRCNN_F32S modes on them (needs the GPU) and stores the exponents in MaskRCNN.mrcw")
    a = ap.parse_args(argv)
    convert(a.config, a.weights, a.output_dir, a.weights_dtype)
    if a.calibrate:
        cfg = ModelConfig.from_json(a.config) if a.config else ModelConfig()
        H, W = int(cfg.input_image_shape[0]), int(cfg.input_image_shape[1])
        calibrate_artefact(a.output_dir, load_calibration_images(a.calibrate, H, W))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
