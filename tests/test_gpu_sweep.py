"""GPU parity sweep: the box-path layers on many small seeded configurations with the awkward inputs
the fixed tests do not reach — non-square anchor grids, pre_nms / maxProposals of 1, thresholds at
their extremes, quantised scores (ties everywhere), huge deltas (expf overflow → ±inf → clip),
ROIs outside / on the border of the image, degenerate and inverted ROIs, batches of rows smaller
than a wavefront.  Everything here is integer / index / IEEE box work: BIT-EXACT vs the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ML = None


@pytest.fixture(autouse=True)
def _ml(pkg):
    global ML
    ML = pkg.MLMultiArray


def _proposal_case(pkg, anchors_mod, orc, tmp_path, seed):
    rng = np.random.default_rng(1000 + seed)
    h = int(rng.choice([64, 128, 192, 256]))
    w = int(rng.choice([64, 128, 192, 256]))
    cfg = pkg.ModelConfig(input_image_shape=(h, w, 3))
    p = str(tmp_path / f"a_{seed}.bin")
    anchors = anchors_mod.write_anchors_bin(p, cfg)
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = p
    A = anchors.shape[0]
    pre = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, A, 2 * A]))
    maxp = int(rng.choice([1, 2, 17, 64, 100, 333]))
    thr = float(rng.choice([0.0, 0.3, 0.5, 0.7, 0.95, 1.0]))
    mode = seed % 5
    fg = rng.random(A)
    if mode == 1:
        fg = np.round(fg * 4) / 4                     # 5 distinct scores
    elif mode == 2:
        fg = np.full(A, 0.5)                          # one score
    elif mode == 3:
        fg = fg ** 8                                  # crowded near zero (denormal-ish keys)
    fg = fg.astype(np.float32)
    probs = np.stack([1 - fg, fg], 1).astype(np.float32)
    scale = [1.0, 0.05, 3.0, 30.0, 0.5][mode]         # 30: dh·0.2 = 6σ → exp overflow territory with the tail
    deltas = (rng.standard_normal((A, 4)) * scale).astype(np.float32)
    if mode == 3:
        deltas[rng.integers(0, A, 50), 2] = 1000.0    # expf(200) = inf → inf-sized box → clipped to the window
        deltas[rng.integers(0, A, 50), 3] = -1000.0   # expf(-200) = 0 → zero width
    params = dict(cfg.proposal_layer_params(), preNMSMaxProposals=pre, maxProposals=maxp, nmsIOUThreshold=thr)
    std = (0.1, 0.1, 0.2, 0.2)
    if seed % 4 == 3:                                 # bboxStdDev_* from the parameter dictionary (ProposalLayer.swift:70-80)
        std = tuple(float(np.float32(v)) for v in rng.choice([0.05, 0.1, 0.2, 0.5], 4))
        params.update({"bboxStdDev_count": 4, **{f"bboxStdDev_{i}": std[i] for i in range(4)}})
    stride = int(rng.choice([4, 4, 5, 8]))
    out = np.full((maxp, stride), np.float32(-3.0), dtype=np.float32)
    pkg.ProposalLayer(params).evaluate([ML(probs), ML(deltas)], [ML(out, shape=(maxp, 1, stride, 1, 1))])
    want = np.full((maxp, stride), np.float32(-3.0), dtype=np.float32)
    want = orc.proposal_layer(probs, deltas, anchors, pre, maxp, thr, std=std, out_stride=stride, out=want)
    np.testing.assert_array_equal(out, want, err_msg=f"seed {seed}: {h}x{w} A={A} pre={pre} maxp={maxp} thr={thr} mode={mode}")
    return int((np.abs(want[:, :4]).sum(1) > 0).sum())


def test_proposal_layer_sweep(pkg, anchors_mod, orc, tmp_path):
    kept = [_proposal_case(pkg, anchors_mod, orc, tmp_path, s) for s in range(150)]
    assert max(kept) > 50 and min(kept) <= 1          # both loaded and nearly empty outcomes occurred


def test_detection_layer_sweep(pkg, orc):
    seen_counts = []
    for seed in range(150):
        rng = np.random.default_rng(2000 + seed)
        n = int(rng.choice([1, 2, 31, 64, 65, 200, 1000]))
        nc = int(rng.choice([2, 3, 21, 81]))
        maxd = int(rng.choice([1, 5, 100]))
        sthr = float(rng.choice([0.0, 0.5, 0.7, 0.99]))
        nthr = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
        y1 = rng.random(n) * 0.8; x1 = rng.random(n) * 0.8
        size = [0.2, 0.6, 0.02][seed % 3]             # moderate overlap / everything overlaps / almost none
        rois = np.stack([y1, x1, y1 + rng.random(n) * size, x1 + rng.random(n) * size], 1).astype(np.float32)
        cls = np.zeros((n, 6), dtype=np.float32)
        cls[:, :4] = (rng.standard_normal((n, 4)) * [1.0, 1.0, 2.0, 2.0]).astype(np.float32)
        cls[:, 4] = rng.integers(0, nc, n).astype(np.float32)         # class 0 rows are background: never detected
        sc = rng.random(n)
        if seed % 4 == 1:
            sc = np.round(sc * 3) / 3                 # ties in the final score sort
        cls[:, 5] = sc.astype(np.float32)
        if n > 4:
            rois[1] = 0                               # padding ROI
            rois[2] = [0.9, 0.9, 0.1, 0.1]            # inverted ROI
        std = (0.1, 0.1, 0.2, 0.2) if seed % 5 else tuple(float(np.float32(v)) for v in rng.choice([0.05, 0.1, 0.3], 4))
        params = {"bboxStdDev_count": 4, "bboxStdDev_0": std[0], "bboxStdDev_1": std[1], "bboxStdDev_2": std[2], "bboxStdDev_3": std[3],
                  "maxDetections": maxd, "scoreThreshold": sthr, "nmsIOUThreshold": nthr}
        out = np.full((maxd, 6), np.float32(np.nan), dtype=np.float32)
        pkg.DetectionLayer(params).evaluate([ML(rois), ML(cls)], [ML(out)])
        want = orc.detection_layer(rois, cls, maxd, sthr, nthr, std=std)
        np.testing.assert_array_equal(out, want, err_msg=f"seed {seed}: n={n} nc={nc} maxd={maxd} score>={sthr} iou>{nthr}")
        seen_counts.append(int((want[:, 5] > 0).sum()))
    assert max(seen_counts) >= 50 and min(seen_counts) == 0      # loaded and empty outcomes both occurred


def test_pyramid_roi_align_sweep(pkg, orc):
    levels_seen = set()
    for seed in range(60):
        rng = np.random.default_rng(3000 + seed)
        C = int(rng.choice([4, 8, 32, 256]))
        base = int(rng.choice([16, 32, 64]))
        ar = [(1, 1), (1, 2), (2, 1)][seed % 3]        # non-square maps
        sizes = [(max(base * ar[0] >> l, 1), max(base * ar[1] >> l, 1)) for l in range(4)]
        fm = [rng.standard_normal((C, hh, ww)).astype(np.float32) for hh, ww in sizes]
        n = int(rng.choice([1, 3, 64, 130]))
        pool = int(rng.choice([1, 2, 7, 14]))
        img_h, img_w = 1024 * ar[0], 1024 * ar[1]
        y1 = rng.random(n) * 1.2 - 0.1; x1 = rng.random(n) * 1.2 - 0.1            # some start outside the image
        hh = rng.random(n) ** 3 * 1.1; ww = rng.random(n) ** 3 * 1.1              # tiny … larger than the image
        rois = np.stack([y1, x1, y1 + hh, x1 + ww], 1).astype(np.float32)
        if n > 3:
            rois[0] = 0
            rois[1] = [0.3, 0.3, 0.3, 0.3]            # zero area
            rois[2] = [1.0, 1.0, 1.0 + 1e-3, 1.0 + 1e-3]   # entirely on the far border
        layer = pkg.PyramidROIAlignLayer({"poolSize": pool, "imageWidth": img_w, "imageHeight": img_h})
        out = np.full((n, 1, C, pool, pool), np.float32(np.nan), dtype=np.float32)
        layer.evaluate([ML(rois)] + [ML(f) for f in fm], [ML(out)])
        want = orc.pyramid_roi_align(rois, fm, pool, img_w, img_h)
        np.testing.assert_array_equal(out.reshape(want.shape), want, err_msg=f"seed {seed}: C={C} sizes={sizes} n={n} pool={pool}")
        levels_seen |= set(np.unique(orc.roi_levels(rois, img_w, img_h)).tolist())
    assert levels_seen >= {-1, 0, 1, 2, 3}


def test_engine_config_sweep(pkg, orc, weights_mod, tmp_path_factory):
    """The fused engine on a dozen seeded small configurations — non-square inputs, class counts 2..81, proposal / detection
    budgets from tiny to larger than the anchor count allows, loaded and sparse weights, batch 1..3, all four compute modes (the headline f32x3 first) —
    each through the full staged parity of test_gpu_engine (index / box stages bit-exact on the GPU's taps)."""
    import importlib
    from oracle.network import load_oracle_model
    from test_gpu_engine import _check_stages
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    rng = np.random.default_rng(77)
    seen_modes = set()
    for case in range(12):
        h = int(rng.choice([128, 192, 256]))
        w = int(rng.choice([128, 192, 256]))
        kw = dict(architecture="resnet50", input_image_shape=(h, w, 3), num_classes=int(rng.choice([2, 5, 21, 81])),
                  pre_nms_max_proposals=int(rng.choice([50, 300, 6000])), max_proposals=int(rng.choice([16, 64, 200])),
                  max_detections=int(rng.choice([4, 16, 100])))
        cfg = pkg.ModelConfig(**kw)
        d = str(tmp_path_factory.mktemp(f"sweep{case}"))
        weights_mod.save_synthetic_models(d, cfg, seed=100 + case, forced_load=bool(case % 3))
        mode = ("f32x3", "f32", "f32s", "f16")[case % 4]
        seen_modes.add(mode)
        B = 1 + case % 3
        om = load_oracle_model(d)
        m = models.load_maskrcnn(d, max_batch=B, compute_dtype=mode)
        images = np.random.default_rng(200 + case).integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        det, mask = m.predict(images)
        trunk = om.trunk(images)
        for b in range(B):
            try:
                d_b, m_b = _check_stages(pkg, orc, om, m, cfg, images, b, True, trunk, f16=(mode == "f16"))
            except AssertionError as e:
                raise AssertionError(f"case {case}: {kw} mode={mode} batch={B} image={b}: {e}") from e
            np.testing.assert_array_equal(det[b], d_b)
            np.testing.assert_array_equal(mask[b].reshape(cfg.max_detections, -1), m_b)
        del m
    assert seen_modes == {"f32x3", "f32", "f32s", "f16"}
