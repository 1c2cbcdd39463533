"""Oracle network: a plain torch-CPU fp32 definition of the graph the reference executes.

TEST INFRASTRUCTURE ONLY (see mrcnn_oracle.c).  PARITY STATUS: parity unpinned — the layer list is
the Matterport Mask R-CNN layout (ResNet-50/101 + FPN + RPN, box head, mask head) defined in the
un-vendored, un-pinned third-party package ``edouardlp/Mask-RCNN-Keras`` that the reference's
converter imports (``Sources/maskrcnn/Python/Conversion/task.py:12-13,171-173``); this repo has no
copy of it, so the topology below is restated from the published Matterport model and declared an
assumption.  Data layout is Core ML's: NCHW activations, OIHW kernels,
fp16-stored weights (task.py:90) computed in fp32 (Core ML CPU path).

The custom layers are executed by oracle/oracle.py (C restatement of the Swift sources).
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from oracle import oracle as orc  # noqa: E402

_pkg = importlib.import_module("mask-rcnn-coreml_amd")
weights_mod = importlib.import_module("mask-rcnn-coreml_amd.weights")     # the .mrcw container reader only
BN_EPS = 1e-3       # Keras BatchNormalization epsilon of the published Matterport graph (stated here, not imported)


def _stage_blocks(architecture: str):
    """Residual block letters per stage of the published ResNet-50 / ResNet-101 (stage 4: 6 / 23 blocks) — the oracle's own
    table, so that a wrong layer list in the product cannot be shared by the checker."""
    n4 = {"resnet50": 6, "resnet101": 23}[architecture]
    return {2: ["a", "b", "c"], 3: ["a", "b", "c", "d"], 4: [chr(ord("a") + i) for i in range(n4)], 5: ["a", "b", "c"]}


class _W:
    def __init__(self, tensors, dtype=torch.float32):
        self.t = {k: torch.from_numpy(np.asarray(v, dtype=np.float32).copy()).to(dtype) for k, v in tensors.items()}

    def conv(self, x, name, stride=1, padding=0):
        return F.conv2d(x, self.t[f"{name}/kernel"], self.t[f"{name}/bias"], stride=stride, padding=padding)

    def bn(self, x, name):
        return F.batch_norm(x, self.t[f"{name}/mean"], self.t[f"{name}/variance"], self.t[f"{name}/gamma"],
                            self.t[f"{name}/beta"], training=False, eps=BN_EPS)


class OracleMaskRCNN:
    """MaskRCNN.mlmodel + Classifier.mlmodel + Mask.mlmodel + the five custom layers, on CPU."""

    def __init__(self, cfg, main_tensors, classifier_tensors, mask_tensors, anchors):
        self.cfg = cfg
        self.w = _W(main_tensors)
        self.wc = _W(classifier_tensors)
        self.wm = _W(mask_tensors)
        self.anchors = np.ascontiguousarray(anchors, dtype=np.float32)

    # ---- MaskRCNN.mlmodel built-in layers ------------------------------------------------------
    def preprocess(self, images_u8):
        """images (B,H,W,3) uint8 RGB → (B,3,H,W) fp32 minus per-channel mean (task.py:73-75)."""
        x = torch.from_numpy(np.ascontiguousarray(images_u8)).to(torch.float32)
        mean = torch.tensor(self.cfg.mean_rgb, dtype=torch.float32)
        return (x - mean).permute(0, 3, 1, 2).contiguous()

    def _block(self, x, stage, block, stride, has_shortcut):
        w = self.w
        p = f"{stage}{block}"
        y = F.relu(w.bn(w.conv(x, f"res{p}_branch2a", stride=stride), f"bn{p}_branch2a"))
        y = F.relu(w.bn(w.conv(y, f"res{p}_branch2b", padding=1), f"bn{p}_branch2b"))
        y = w.bn(w.conv(y, f"res{p}_branch2c"), f"bn{p}_branch2c")
        sc = w.bn(w.conv(x, f"res{p}_branch1", stride=stride), f"bn{p}_branch1") if has_shortcut else x
        return F.relu(y + sc)

    def backbone(self, x):
        w = self.w
        x = F.pad(x, (3, 3, 3, 3))
        x = F.relu(w.bn(w.conv(x, "conv1", stride=2), "bn_conv1"))
        x = F.pad(x, (0, 1, 0, 1), value=float("-inf"))          # Keras 'same' pool: pad bottom/right
        x = F.max_pool2d(x, 3, 2)
        blocks = _stage_blocks(self.cfg.architecture)
        feats = []
        for stage in (2, 3, 4, 5):
            for b in blocks[stage]:
                first = b == "a"
                x = self._block(x, stage, b, 2 if (first and stage > 2) else 1, first)
            feats.append(x)
        return feats                                              # C2..C5

    def fpn(self, feats):
        w = self.w
        c2, c3, c4, c5 = feats
        p5 = w.conv(c5, "fpn_c5p5")
        p4 = F.interpolate(p5, scale_factor=2, mode="nearest") + w.conv(c4, "fpn_c4p4")
        p3 = F.interpolate(p4, scale_factor=2, mode="nearest") + w.conv(c3, "fpn_c3p3")
        p2 = F.interpolate(p3, scale_factor=2, mode="nearest") + w.conv(c2, "fpn_c2p2")
        p2 = w.conv(p2, "fpn_p2", padding=1)
        p3 = w.conv(p3, "fpn_p3", padding=1)
        p4 = w.conv(p4, "fpn_p4", padding=1)
        p5 = w.conv(p5, "fpn_p5", padding=1)
        p6 = p5[:, :, ::2, ::2]                                   # MaxPool(pool 1, stride 2)
        return [p2, p3, p4, p5, p6]

    def rpn(self, pyramid):
        w = self.w
        probs, deltas = [], []
        for p in pyramid:
            s = F.relu(w.conv(p, "rpn_conv_shared", padding=1))
            lg = w.conv(s, "rpn_class_raw").permute(0, 2, 3, 1)   # (B,H,W,2*na) → (B, H*W*na, 2)
            lg = lg.reshape(lg.shape[0], -1, 2)
            probs.append(F.softmax(lg, dim=-1))
            bb = w.conv(s, "rpn_bbox_pred").permute(0, 2, 3, 1)
            deltas.append(bb.reshape(bb.shape[0], -1, 4))
        return torch.cat(probs, 1), torch.cat(deltas, 1)

    def trunk(self, images_u8):
        with torch.no_grad():
            x = self.preprocess(images_u8)
            pyr = self.fpn(self.backbone(x))
            probs, deltas = self.rpn(pyr)
        return [p.numpy() for p in pyr[:4]], probs.numpy(), deltas.numpy()

    def trunk_fp64(self, images_u8):
        """The same graph evaluated in float64 (weights are the fp16-exact values widened): the ground truth that the
        fp32 engines' summation-order / split-precision errors are measured against."""
        w32 = self.w
        try:
            self.w = _W({k: v.numpy() for k, v in w32.t.items()}, torch.float64)
            with torch.no_grad():
                x = self.preprocess(images_u8).to(torch.float64)
                pyr = self.fpn(self.backbone(x))
                probs, deltas = self.rpn(pyr)
            return [p.numpy() for p in pyr[:4]], probs.numpy(), deltas.numpy()
        finally:
            self.w = w32

    # ---- Classifier.mlmodel / Mask.mlmodel -----------------------------------------------------
    def classifier_model(self, fmap):
        """feature_map (n,256,7,7) → probabilities (n,nc), bounding_boxes (n, nc*4)."""
        w = self.wc
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(fmap, dtype=np.float32))
            x = F.relu(w.bn(w.conv(x, "mrcnn_class_conv1"), "mrcnn_class_bn1"))
            x = F.relu(w.bn(w.conv(x, "mrcnn_class_conv2"), "mrcnn_class_bn2"))
            x = x.reshape(x.shape[0], -1)
            logits = F.linear(x, w.t["mrcnn_class_logits/kernel"], w.t["mrcnn_class_logits/bias"])
            probs = F.softmax(logits, dim=-1)
            bbox = F.linear(x, w.t["mrcnn_bbox_fc/kernel"], w.t["mrcnn_bbox_fc/bias"])
        return probs.numpy(), bbox.numpy()

    def mask_model(self, fmap):
        """feature_map (n,256,14,14) → masks (n,nc,28,28)."""
        w = self.wm
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(fmap, dtype=np.float32))
            if x.shape[0] == 0:
                return np.zeros((0, self.cfg.num_classes, 2 * x.shape[2], 2 * x.shape[3]), np.float32)
            for i in range(1, 5):
                x = F.relu(w.bn(w.conv(x, f"mrcnn_mask_conv{i}", padding=1), f"mrcnn_mask_bn{i}"))
            x = F.relu(F.conv_transpose2d(x, w.t["mrcnn_mask_deconv/kernel"], w.t["mrcnn_mask_deconv/bias"], stride=2))
            x = torch.sigmoid(w.conv(x, "mrcnn_mask"))
        return x.numpy()

    # ---- stages after the trunk (each usable on its own with taps from the HIP engine) ---------
    def proposals(self, probs, deltas, debug=False):
        c = self.cfg
        return orc.proposal_layer(probs, deltas, self.anchors, c.pre_nms_max_proposals, c.max_proposals,
                                  c.rpn_nms_threshold, c.bounding_box_std_dev, debug=debug)

    def roi_align(self, rois, pyramid, pool):
        c = self.cfg
        pp = c.pyramid_params(pool)
        return orc.pyramid_roi_align(rois, pyramid, pool, pp["imageWidth"], pp["imageHeight"])

    def classify(self, pooled):
        probs, bbox = self.classifier_model(pooled)
        return orc.classifier_postprocess(probs, bbox), probs, bbox

    def detect(self, rois, cls6):
        c = self.cfg
        return orc.detection_layer(rois, cls6, c.max_detections, c.detection_min_confidence,
                                   c.detection_nms_threshold, c.bounding_box_std_dev)

    def masks(self, pooled_mask, detections, out=None, valid_from=None):
        """valid_from: the rows the removeZeros predicate is evaluated on when they differ from what the mask model is
        fed (fp16 engine: the predicate sees the fp32 samples, the model their fp16 rounding)."""
        mapping = orc.mask_valid_rows(pooled_mask if valid_from is None else valid_from)
        m = self.mask_model(pooled_mask[mapping])
        if out is None:
            out = np.zeros((detections.shape[0], m.shape[2] * m.shape[3] if m.size else 784), np.float32)
        return orc.mask_layer_write(m, mapping, detections, out)

    def predict(self, images_u8, taps=False):
        """images (B,H,W,3) uint8 → detections (B,max_det,6), masks (B,max_det,28,28)."""
        pyr, probs, deltas = self.trunk(images_u8)
        B = probs.shape[0]
        dets, masks, tap = [], [], []
        for b in range(B):
            pb = [p[b] for p in pyr]
            rois = self.proposals(probs[b], deltas[b])
            pooled = self.roi_align(rois, pb, self.cfg.classifier_pool_size)
            cls6, cprobs, cbbox = self.classify(pooled)
            det = self.detect(rois, cls6)
            pooled_m = self.roi_align(det, pb, self.cfg.mask_pool_size)
            mk = self.masks(pooled_m, det)
            dets.append(det)
            masks.append(mk.reshape(det.shape[0], 2 * self.cfg.mask_pool_size, 2 * self.cfg.mask_pool_size))
            if taps:
                tap.append({"rois": rois, "pooled": pooled, "cls6": cls6, "probs": cprobs, "bbox": cbbox,
                            "pooled_mask": pooled_m})
        out = (np.stack(dets), np.stack(masks))
        if taps:
            return out + ({"pyramid": pyr, "rpn_probs": probs, "rpn_deltas": deltas, "per_image": tap},)
        return out


def load_oracle_model(model_dir: str, cfg=None) -> OracleMaskRCNN:
    anchors_mod = importlib.import_module("mask-rcnn-coreml_amd.anchors")
    config_mod = importlib.import_module("mask-rcnn-coreml_amd.config")
    meta, main = weights_mod.read_mrcw(os.path.join(model_dir, "MaskRCNN.mrcw"))
    _, cls = weights_mod.read_mrcw(os.path.join(model_dir, "Classifier.mrcw"))
    _, msk = weights_mod.read_mrcw(os.path.join(model_dir, "Mask.mrcw"))
    if cfg is None:
        cfg = config_mod.ModelConfig(architecture=meta["architecture"],
                                     input_image_shape=(meta["image_height"], meta["image_width"], 3),
                                     num_classes=meta["num_classes"],
                                     pre_nms_max_proposals=meta["pre_nms_max_proposals"],
                                     max_proposals=meta["max_proposals"],
                                     max_detections=meta["DetectionLayer.maxDetections"])
    anchors = anchors_mod.read_anchors_bin(os.path.join(model_dir, "anchors.bin"), cfg.num_anchors())
    return OracleMaskRCNN(cfg, main, cls, msk, anchors)
