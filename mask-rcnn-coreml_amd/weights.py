"""The .mrcw model artefact: what stands where the reference has a ``.mlmodel``.

The reference ships three Core ML artefacts produced by ``Sources/maskrcnn/Python/Conversion/
task.py`` — ``MaskRCNN.mlmodel`` (:69-92), ``Mask.mlmodel`` (:94-104), ``Classifier.mlmodel``
(:106-116) — whose weights are cast to fp16 (:90,102,114) and whose custom layers carry parameter
dictionaries (:25-67).  A ``.mrcw`` file carries the same information for the HIP engine:

    "MRCW" u32 version(=1) u32 n_meta u32 n_tensors
    meta   : u16 klen, key, u8 type (0=int64, 1=float64, 2=utf8), value (8 B | u32 len + bytes)
    tensor : u16 nlen, name, u8 dtype (0=f32, 2=f16), u8 ndim, u32 dims[ndim], u64 offset, u64 nbytes
    (pad to 64 B) data blob; every tensor 64-B aligned; offsets relative to the blob start

Tensor names are the Matterport Keras layer names (the network definition lives in the un-vendored
third-party package) with Core ML weight layouts:
``<conv>/kernel`` [O, I, kh, kw], ``<conv>/bias`` [O], ``<deconv>/kernel`` [I, O, kh, kw],
``<dense>/kernel`` [O, I], ``<bn>/{gamma,beta,mean,variance}`` [C]; all fp16 like the reference.
Metadata keys: ``kind`` (MaskRCNN | Classifier | Mask), the conversion config keys of
``README.md:87-92``, and ``<LayerClass>[.<instance>].<param>`` for every custom-layer parameter of
``task.py:25-67``.

There is no network here, hence no released weights: ``synthetic_models`` makes seeded weights of
the right architecture (SURVEY.md §8d "synthetic inputs").
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

from .config import ModelConfig

MAGIC = b"MRCW"
VERSION = 1
_DT = {np.dtype("<f4"): 0, np.dtype("<f2"): 2}
_DT_INV = {0: np.dtype("<f4"), 2: np.dtype("<f2")}
BN_EPS = 1e-3  # Keras BatchNormalization default, used by Matterport's BatchNorm


def write_mrcw(path: str, meta: Dict[str, object], tensors: Dict[str, np.ndarray]) -> None:
    hdr = bytearray()
    hdr += MAGIC + struct.pack("<III", VERSION, len(meta), len(tensors))
    for k, v in meta.items():
        kb = k.encode()
        hdr += struct.pack("<H", len(kb)) + kb
        if isinstance(v, bool) or isinstance(v, (int, np.integer)):
            hdr += struct.pack("<Bq", 0, int(v))
        elif isinstance(v, (float, np.floating)):
            hdr += struct.pack("<Bd", 1, float(v))
        else:
            sb = str(v).encode()
            hdr += struct.pack("<BI", 2, len(sb)) + sb
    off = 0
    entries = []
    for name, arr in tensors.items():
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in _DT:
            raise TypeError(f"{name}: dtype {arr.dtype} not supported")
        off = (off + 63) // 64 * 64
        entries.append((name, arr, off))
        off += arr.nbytes
    for name, arr, o in entries:
        nb = name.encode()
        hdr += struct.pack("<H", len(nb)) + nb
        hdr += struct.pack("<BB", _DT[arr.dtype], arr.ndim)
        hdr += struct.pack(f"<{arr.ndim}I", *arr.shape)
        hdr += struct.pack("<QQ", o, arr.nbytes)
    pad = (-len(hdr)) % 64
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(b"\0" * pad)
        pos = 0
        for name, arr, o in entries:
            if o > pos:
                f.write(b"\0" * (o - pos))
            f.write(arr.tobytes())
            pos = o + arr.nbytes


def read_mrcw(path: str) -> Tuple[Dict[str, object], Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != MAGIC:
        raise ValueError(f"{path}: not a .mrcw file")
    ver, n_meta, n_t = struct.unpack_from("<III", buf, 4)
    if ver != VERSION:
        raise ValueError(f"{path}: unsupported version {ver}")
    p = 16
    meta: Dict[str, object] = {}
    for _ in range(n_meta):
        (kl,) = struct.unpack_from("<H", buf, p); p += 2
        k = buf[p:p + kl].decode(); p += kl
        t = buf[p]; p += 1
        if t == 0:
            (v,) = struct.unpack_from("<q", buf, p); p += 8
        elif t == 1:
            (v,) = struct.unpack_from("<d", buf, p); p += 8
        else:
            (sl,) = struct.unpack_from("<I", buf, p); p += 4
            v = buf[p:p + sl].decode(); p += sl
        meta[k] = v
    ents = []
    for _ in range(n_t):
        (nl,) = struct.unpack_from("<H", buf, p); p += 2
        name = buf[p:p + nl].decode(); p += nl
        dt, nd = buf[p], buf[p + 1]; p += 2
        dims = struct.unpack_from(f"<{nd}I", buf, p); p += 4 * nd
        off, nb = struct.unpack_from("<QQ", buf, p); p += 16
        ents.append((name, dt, dims, off, nb))
    base = (p + 63) // 64 * 64
    tensors = {}
    for name, dt, dims, off, nb in ents:
        tensors[name] = np.frombuffer(buf, dtype=_DT_INV[dt], count=nb // _DT_INV[dt].itemsize,
                                      offset=base + off).reshape(dims)
    return meta, tensors


# ------------------------------------------------------------------------------------------------
# Network topology (Matterport layout): (name, kind, cin, cout, k) lists shared by the synthetic
# generator, the engine's plan builder (C++ has its own copy) and the oracle's torch definition.
# ------------------------------------------------------------------------------------------------
def resnet_stage_blocks(architecture: str):
    assert architecture in ("resnet50", "resnet101")
    n4 = {"resnet50": 5, "resnet101": 22}[architecture]
    return {2: ["a", "b", "c"], 3: ["a", "b", "c", "d"],
            4: ["a"] + [chr(98 + i) for i in range(n4)], 5: ["a", "b", "c"]}


def trunk_layers(cfg: ModelConfig):
    """[(conv_name, bn_name|None, cin, cout, k)] for MaskRCNN.mrcw."""
    L = [("conv1", "bn_conv1", 3, 64, 7)]
    cin = 64
    for stage, (f1, f3) in zip((2, 3, 4, 5), ((64, 256), (128, 512), (256, 1024), (512, 2048))):
        for b in resnet_stage_blocks(cfg.architecture)[stage]:
            p = f"{stage}{b}"
            L.append((f"res{p}_branch2a", f"bn{p}_branch2a", cin, f1, 1))
            L.append((f"res{p}_branch2b", f"bn{p}_branch2b", f1, f1, 3))
            L.append((f"res{p}_branch2c", f"bn{p}_branch2c", f1, f3, 1))
            if b == "a":
                L.append((f"res{p}_branch1", f"bn{p}_branch1", cin, f3, 1))
            cin = f3
    for name, c in (("fpn_c5p5", 2048), ("fpn_c4p4", 1024), ("fpn_c3p3", 512), ("fpn_c2p2", 256)):
        L.append((name, None, c, 256, 1))
    for name in ("fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5"):
        L.append((name, None, 256, 256, 3))
    na = len(cfg.anchor_ratios)
    L.append(("rpn_conv_shared", None, 256, 512, 3))
    L.append(("rpn_class_raw", None, 512, 2 * na, 1))
    L.append(("rpn_bbox_pred", None, 512, 4 * na, 1))
    return L


def tensor_shapes(cfg: ModelConfig) -> Dict[str, Dict[str, Tuple[int, ...]]]:
    """{"MaskRCNN" | "Classifier" | "Mask": {tensor name: shape}} — the complete tensor inventory the
    engine's plan builder asks a set of .mrcw files for (Core ML layouts, see the module docstring)."""
    def bn(c):
        return {k: (c,) for k in ("gamma", "beta", "mean", "variance")}

    main: Dict[str, Tuple[int, ...]] = {}
    for conv, b, cin, cout, k in trunk_layers(cfg):
        main[f"{conv}/kernel"] = (cout, cin, k, k)
        main[f"{conv}/bias"] = (cout,)
        if b:
            for kk, sh in bn(cout).items():
                main[f"{b}/{kk}"] = sh
    nc, ps = cfg.num_classes, cfg.classifier_pool_size
    cls: Dict[str, Tuple[int, ...]] = {"mrcnn_class_conv1/kernel": (1024, 256, ps, ps), "mrcnn_class_conv1/bias": (1024,),
                                       "mrcnn_class_conv2/kernel": (1024, 1024, 1, 1), "mrcnn_class_conv2/bias": (1024,),
                                       "mrcnn_class_logits/kernel": (nc, 1024), "mrcnn_class_logits/bias": (nc,),
                                       "mrcnn_bbox_fc/kernel": (nc * 4, 1024), "mrcnn_bbox_fc/bias": (nc * 4,)}
    for i in (1, 2):
        for kk, sh in bn(1024).items():
            cls[f"mrcnn_class_bn{i}/{kk}"] = sh
    mask: Dict[str, Tuple[int, ...]] = {"mrcnn_mask_deconv/kernel": (256, 256, 2, 2), "mrcnn_mask_deconv/bias": (256,),
                                        "mrcnn_mask/kernel": (nc, 256, 1, 1), "mrcnn_mask/bias": (nc,)}
    for i in range(1, 5):
        mask[f"mrcnn_mask_conv{i}/kernel"] = (256, 256, 3, 3)
        mask[f"mrcnn_mask_conv{i}/bias"] = (256,)
        for kk, sh in bn(256).items():
            mask[f"mrcnn_mask_bn{i}/{kk}"] = sh
    return {"MaskRCNN": main, "Classifier": cls, "Mask": mask}


def _he(rng, shape, fan_in, gain=1.0):
    return (rng.standard_normal(shape, dtype=np.float32) * np.float32(gain * np.sqrt(2.0 / fan_in)))


def _bn(rng, c, gamma_scale=1.0):
    return {"gamma": (rng.uniform(0.8, 1.2, c) * gamma_scale).astype(np.float32),
            "beta": rng.uniform(-0.1, 0.1, c).astype(np.float32),
            "mean": rng.uniform(-0.1, 0.1, c).astype(np.float32),
            "variance": rng.uniform(0.8, 1.2, c).astype(np.float32)}


def _f16(a):
    return np.asarray(a, dtype=np.float32).astype("<f2")


def base_meta(cfg: ModelConfig, kind: str) -> Dict[str, object]:
    m: Dict[str, object] = {"kind": kind, "architecture": cfg.architecture,
                            "image_height": cfg.image_height, "image_width": cfg.image_width,
                            "num_classes": cfg.num_classes,
                            "pre_nms_max_proposals": cfg.pre_nms_max_proposals,
                            "max_proposals": cfg.max_proposals,
                            "num_anchors_per_location": len(cfg.anchor_ratios),
                            "mean_r": cfg.mean_rgb[0], "mean_g": cfg.mean_rgb[1], "mean_b": cfg.mean_rgb[2],
                            "bn_eps": BN_EPS}
    if kind == "MaskRCNN":
        for k, v in cfg.proposal_layer_params().items():
            m[f"ProposalLayer.{k}"] = v
        for k, v in cfg.pyramid_params(cfg.classifier_pool_size).items():
            m[f"PyramidROIAlignLayer.classifier.{k}"] = v
        for k, v in cfg.pyramid_params(cfg.mask_pool_size).items():
            m[f"PyramidROIAlignLayer.mask.{k}"] = v
        for k, v in cfg.detection_layer_params().items():
            m[f"DetectionLayer.{k}"] = v
    return m


def synthetic_models(cfg: ModelConfig, seed: int = 0, forced_load: bool = True):
    """Seeded weights for the three models.  He-normal kernels rounded through fp16 (task.py:90),
    mildly non-trivial BatchNorm statistics; the last BN of every residual branch is damped so that
    activations stay O(1..100) through 33 blocks (also keeps an fp16 pipeline in range).

    ``forced_load``: scale the RPN / classifier output layers so that the data-dependent stages are
    fully loaded (>= max_proposals survive NMS, >= max_detections rows pass the 0.7 score filter) —
    with plain He-init the softmaxes sit near uniform and the heads would idle.
    Returns {"MaskRCNN": (meta, tensors), "Classifier": (...), "Mask": (...)}.
    """
    rng = np.random.default_rng(seed)
    t: Dict[str, np.ndarray] = {}
    for conv, bn, cin, cout, k in trunk_layers(cfg):
        gain = 1.0
        if conv in ("rpn_class_raw",):
            gain = 0.3 if forced_load else 0.05
        if conv in ("rpn_bbox_pred",):
            gain = 0.15
        if conv.startswith("fpn_") or conv.startswith("rpn_conv"):
            gain = 0.7   # no ReLU in front of the FPN convs: He gain would double the variance
        if conv == "conv1":
            gain = 1.0 / 74.0   # raw pixels minus mean have std ~74: bring the stem to O(1)
        t[f"{conv}/kernel"] = _f16(_he(rng, (cout, cin, k, k), cin * k * k, gain))
        t[f"{conv}/bias"] = _f16(rng.uniform(-0.05, 0.05, cout))
        if bn:
            damp = 0.25 if conv.endswith("branch2c") else 1.0
            for kk, v in _bn(rng, cout, damp).items():
                t[f"{bn}/{kk}"] = _f16(v)
    main = (base_meta(cfg, "MaskRCNN"), t)

    nc = cfg.num_classes
    c: Dict[str, np.ndarray] = {}
    ps = cfg.classifier_pool_size
    c["mrcnn_class_conv1/kernel"] = _f16(_he(rng, (1024, 256, ps, ps), 256 * ps * ps))
    c["mrcnn_class_conv1/bias"] = _f16(rng.uniform(-0.05, 0.05, 1024))
    for kk, v in _bn(rng, 1024).items():
        c[f"mrcnn_class_bn1/{kk}"] = _f16(v)
    c["mrcnn_class_conv2/kernel"] = _f16(_he(rng, (1024, 1024, 1, 1), 1024))
    c["mrcnn_class_conv2/bias"] = _f16(rng.uniform(-0.05, 0.05, 1024))
    for kk, v in _bn(rng, 1024).items():
        c[f"mrcnn_class_bn2/{kk}"] = _f16(v)
    lg = 1.5 if forced_load else 0.3
    lk = _he(rng, (nc, 1024), 1024, lg)
    lk -= lk.mean(axis=1, keepdims=True)   # zero-sum rows: the common-mode of the post-ReLU features
    c["mrcnn_class_logits/kernel"] = _f16(lk)  # cancels, so the argmax class varies from ROI to ROI
    bias = rng.uniform(-0.05, 0.05, nc)
    if forced_load:
        bias[0] -= 4.0            # push background down so most rows keep a foreground argmax
    c["mrcnn_class_logits/bias"] = _f16(bias)
    bk = _he(rng, (nc * 4, 1024), 1024, 0.15)
    bk -= bk.mean(axis=1, keepdims=True)
    c["mrcnn_bbox_fc/kernel"] = _f16(bk)
    c["mrcnn_bbox_fc/bias"] = _f16(rng.uniform(-0.05, 0.05, nc * 4))
    cls = (base_meta(cfg, "Classifier"), c)

    m: Dict[str, np.ndarray] = {}
    for i in range(1, 5):
        m[f"mrcnn_mask_conv{i}/kernel"] = _f16(_he(rng, (256, 256, 3, 3), 256 * 9))
        m[f"mrcnn_mask_conv{i}/bias"] = _f16(rng.uniform(-0.05, 0.05, 256))
        for kk, v in _bn(rng, 256).items():
            m[f"mrcnn_mask_bn{i}/{kk}"] = _f16(v)
    m["mrcnn_mask_deconv/kernel"] = _f16(_he(rng, (256, 256, 2, 2), 256))
    m["mrcnn_mask_deconv/bias"] = _f16(rng.uniform(-0.05, 0.05, 256))
    mk = _he(rng, (nc, 256, 1, 1), 256, 0.25)
    mk -= mk.mean(axis=1, keepdims=True)
    m["mrcnn_mask/kernel"] = _f16(mk)
    m["mrcnn_mask/bias"] = _f16(rng.uniform(-0.05, 0.05, nc))
    mask = (base_meta(cfg, "Mask"), m)
    return {"MaskRCNN": main, "Classifier": cls, "Mask": mask}


def save_synthetic_models(out_dir: str, cfg: ModelConfig, seed: int = 0, forced_load: bool = True):
    """Writes MaskRCNN.mrcw, Classifier.mrcw, Mask.mrcw and anchors.bin (the four artefacts the
    reference downloads, DownloadCommand.swift:10-32) and returns their paths."""
    from .anchors import write_anchors_bin
    os.makedirs(out_dir, exist_ok=True)
    models = synthetic_models(cfg, seed, forced_load)
    paths = {}
    for kind, (meta, tensors) in models.items():
        p = os.path.join(out_dir, f"{kind}.mrcw")
        write_mrcw(p, meta, tensors)
        paths[kind] = p
    paths["anchors"] = os.path.join(out_dir, "anchors.bin")
    write_anchors_bin(paths["anchors"], cfg)
    return paths
