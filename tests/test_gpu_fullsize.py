"""The headline workload at its own batch (BASELINE configs[1]: ResNet101+FPN, 1024², batch 8 on one MI355X) and the
1536² stress configuration at batch 2, in every compute mode:

  * per-image results of the batch call are BIT-EQUAL to eight batch-1 calls — at batch 8 the GEMM M dimension,
    hence the tile selection / grid-fill narrowing of the conv kernels, differs from batch 1 (conv_forward), so this
    is the test that the kernels' results do not depend on it (the multi-GPU sharding contract);
  * staged oracle parity (same bars as tests/test_gpu_engine.py) on the first and the last image of the batch.

VERDICT r1 "Next round" item 1(a).
"""
import numpy as np
import pytest

from conftest import rand_images, make_model_dir
from test_gpu_engine import _check_stages

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_model(tmp_path_factory, pkg, weights_mod):
    return make_model_dir(tmp_path_factory, pkg, weights_mod, "full8", architecture="resnet101")


@pytest.fixture(scope="module")
def full_images():
    return rand_images(8, 1024, 1024, seed=21)


@pytest.fixture(scope="module")
def full_oracle(full_model, full_images):
    """The CPU network on images 0 and 7 only (rows 0 and 1 of the returned trunk)."""
    from oracle.network import load_oracle_model
    d, cfg = full_model
    om = load_oracle_model(d)
    return om, om.trunk(full_images[[0, 7]])


@pytest.mark.parametrize("mode", ["f32", "f32s", "f32x3", "f16"])
def test_headline_batch8_full_size(pkg, orc, full_model, full_images, full_oracle, mode):
    models = __import__("importlib").import_module("mask-rcnn-coreml_amd.models")
    d, cfg = full_model
    om, trunk = full_oracle
    m = models.load_maskrcnn(d, max_batch=8, compute_dtype=mode)
    if mode in ("f32x3", "f32s"):
        m.calibrate_split(full_images[:2])             # the split modes as bench.py runs them: calibrated exponents (round 4)
    det, mask = m.predict(full_images)
    f16 = mode == "f16"
    for b, tb in ((0, 0), (7, 1)):
        d_b, m_b = _check_stages(pkg, orc, om, m, cfg, full_images, b, True, trunk, f16=f16, tb=tb)
        np.testing.assert_array_equal(det[b], d_b)
        np.testing.assert_array_equal(mask[b].reshape(cfg.max_detections, -1), m_b)
        assert int(m.read_tensor("keep_count", b)[0]) == cfg.max_proposals          # "forced full load" really loads
        assert int((d_b[:, 5] > 0).sum()) == cfg.max_detections
    for b in range(8):
        d1, m1 = m.predict(full_images[b:b + 1])
        np.testing.assert_array_equal(d1[0], det[b], err_msg=f"image {b}: batch-8 result differs from its batch-1 result")
        np.testing.assert_array_equal(m1[0], mask[b])
    # and the other way round: a different batch size in between must not leave state behind
    det4, mask4 = m.predict(full_images[2:6])
    np.testing.assert_array_equal(det4, det[2:6])
    np.testing.assert_array_equal(mask4, mask[2:6])
    if mode in ("f32x3", "f32s"):
        # the fused bottleneck tail (C4's 3x3 + 1x1 + shortcut in one persistent launch; opt-in: measured slower)
        # sums in the order of the two launches it replaces: bit-identical, all 23 blocks of C4, at the batch that fills the chip
        L = __import__("importlib").import_module("mask-rcnn-coreml_amd._lib")
        # ... and in the same predict the stage-entry shortcuts run as their own launches ("conv_scfuse" 0) instead of inside branch2c's
        # (late round 4): one comparison pins both fusions at the full-size grids
        try:
            L.check(L.lib().mrcnn_debug_set(b"conv_tail", 1))
            L.check(L.lib().mrcnn_debug_set(b"conv_scfuse", 0))
            det_t, mask_t = m.predict(full_images)
            p5 = [m.read_tensor("P5", b) for b in (0, 7)]
        finally:
            L.check(L.lib().mrcnn_debug_set(b"conv_tail", 0))
            L.check(L.lib().mrcnn_debug_set(b"conv_scfuse", 1))
        np.testing.assert_array_equal(det_t, det)
        np.testing.assert_array_equal(mask_t, mask)
        m.predict(full_images)
        for k, b in enumerate((0, 7)):
            np.testing.assert_array_equal(m.read_tensor("P5", b), p5[k])
    if mode == "f16":
        # round 6: C4's 22 identity blocks as ONE launch whose tiles wait for their neighbours' previous block (kernels_bneck.hip, STAGE form;
        # opt-in: measured equal, profiles/r06_bneck_stage_ab.txt) — the same tile arithmetic, so the whole predict agrees bit for bit, twice
        L = __import__("importlib").import_module("mask-rcnn-coreml_amd._lib")
        try:
            L.check(L.lib().mrcnn_debug_set(b"conv_bneck_stage", 1))
            for _ in range(2):
                det_s, mask_s = m.predict(full_images)
                np.testing.assert_array_equal(det_s, det)
                np.testing.assert_array_equal(mask_s, mask)
            p4 = [m.read_tensor("P4", b) for b in (0, 7)]
        finally:
            L.check(L.lib().mrcnn_debug_set(b"conv_bneck_stage", 0))
        m.predict(full_images)
        for k, b in enumerate((0, 7)):
            np.testing.assert_array_equal(m.read_tensor("P4", b), p4[k])
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


@pytest.mark.parametrize("mode", ["f32x3", "f32", "f16"])
def test_config5_1536_batch2(pkg, orc, tmp_path_factory, weights_mod, mode):
    """BASELINE configs[4] (1536², 2 classes, pre_nms 12000) at batch 2: batch independence + staged parity on image 1."""
    from oracle.network import load_oracle_model
    models = __import__("importlib").import_module("mask-rcnn-coreml_amd.models")
    d, cfg = make_model_dir(tmp_path_factory, pkg, weights_mod, "c5b2" + mode, architecture="resnet101",
                            input_image_shape=(1536, 1536, 3), num_classes=2, pre_nms_max_proposals=12000)
    om = load_oracle_model(d)
    m = models.load_maskrcnn(d, max_batch=2, compute_dtype=mode)
    images = rand_images(2, 1536, 1536, seed=23)
    det, mask = m.predict(images)
    trunk = om.trunk(images[1:2])
    d_b, m_b = _check_stages(pkg, orc, om, m, cfg, images, 1, True, trunk, f16=(mode == "f16"), tb=0)
    np.testing.assert_array_equal(det[1], d_b)
    for b in range(2):
        d1, m1 = m.predict(images[b:b + 1])
        np.testing.assert_array_equal(d1[0], det[b])
        np.testing.assert_array_equal(m1[0], mask[b])
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


@pytest.mark.parametrize("mode", ["f32x3", "f16"])
@pytest.mark.parametrize("name,kw", [("resnet50", dict(architecture="resnet50")),
                                     ("c5_1536", dict(architecture="resnet101", input_image_shape=(1536, 1536, 3), num_classes=2, pre_nms_max_proposals=12000))])
def test_other_configs_batch8_equals_eight_batch1_calls(pkg, tmp_path_factory, weights_mod, name, kw, mode):
    """the bench numbers of BASELINE configs[2] (ResNet-50) and configs[4] (1536², 2 classes, pre_nms 12000) are
    batch-8 numbers, while their parity tests ran at batch 1 / 2.  Per-image results must not depend on the batch THERE either (the
    tile shapes, the halo kernel's tile widths and the fused-head policy all change with M): batch 8 bit-equal to eight batch-1 calls,
    in the headline mode (calibrated split) and in fp16; every image yields its full load of proposals."""
    models = __import__("importlib").import_module("mask-rcnn-coreml_amd.models")
    d, cfg = make_model_dir(tmp_path_factory, pkg, weights_mod, "b8" + name + mode, **kw)
    m = models.load_maskrcnn(d, max_batch=8, compute_dtype=mode)
    images = rand_images(8, cfg.image_height, cfg.image_width, seed=53)
    if mode == "f32x3":
        m.calibrate_split(images[:1])
    det, mask = m.predict(images)
    assert int(m.read_tensor("keep_count", 0)[0]) == cfg.max_proposals and int((det[7, :, 5] > 0).sum()) > 0
    for b in range(8):
        d1, m1 = m.predict(images[b:b + 1])
        np.testing.assert_array_equal(d1[0], det[b], err_msg=f"{name} {mode} image {b}: batch-8 result differs from its batch-1 result")
        np.testing.assert_array_equal(m1[0], mask[b])
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


N_E2E = 16          # oracle images of the end-to-end tests (~5 s of host time each; bench.py / profiles/ carry 64 and 256)


@pytest.fixture(scope="module")
def e2e_oracle(full_model):
    """The CPU oracle's predict (with its pooled_mask taps) on N_E2E full-size images, shared by the end-to-end tests."""
    from oracle.network import load_oracle_model
    d, cfg = full_model
    images = rand_images(N_E2E, 1024, 1024, seed=31)
    om = load_oracle_model(d)
    dets, masks, pooled = [], [], []
    for i in range(0, N_E2E, 4):                              # four images per trunk call: bounded host memory
        dd, kk, taps = om.predict(images[i:i + 4], taps=True)
        dets.append(dd); masks.append(kk)
        pooled += [t["pooled_mask"] for t in taps["per_image"]]
    return images, np.concatenate(dets), np.concatenate(masks), pooled


def _hip_predict_with_taps(m, images, B=8):
    dets, masks, pooled = [], [], []
    for i in range(0, len(images), B):
        d_, k_ = m.predict(images[i:i + B])
        dets.append(d_); masks.append(k_)
        for b in range(d_.shape[0]):
            pooled.append(m.read_tensor("pooled_mask", b).reshape(d_.shape[1], -1))
    return np.concatenate(dets), np.concatenate(masks), pooled


def test_headline_end_to_end_agreement_with_the_oracle(pkg, full_model, e2e_oracle):
    """The end-to-end figure bench.py prints as parity_e2e, as a test: HIP predict in the headline mode (f32x3, calibrated split)
    vs the CPU oracle's predict on 16 full-size images of the headline workload.  Two fp32 evaluations that sum in different
    orders can swap a near-tie — profiles/r03_parity_e2e_256.json: 25 593 / 25 600 detections, 249 / 256 images fully matched —
    so the bar is a FRACTION (>= 0.999 of the detections: at most one of 1 600), not equality at one seed; matched
    pairs agree in score within 1e-5 and in mask (where both sides have one, in front of any removeZeros flip) within 2e-4.
    VERDICT r3 item 4: every mask that is present on one side only must be the reference's removeZeros cliff, PROVEN from the
    taps (evaluate.mask_flip_causes) — each side's masks are exactly the compacted rows its OWN pooled samples keep (no exact 0.0
    in the row), and where the two sides' predicates differ the boxes agree to the last bits; anything else fails."""
    import importlib
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    ev = importlib.import_module("mask-rcnn-coreml_amd.evaluate")
    d, cfg = full_model
    images, od, ok, opool = e2e_oracle
    m = models.load_maskrcnn(d, max_batch=8, compute_dtype="f32x3")
    m.calibrate_split(images[:2])
    hd, hk, hpool = _hip_predict_with_taps(m, images)
    tot = matched = flips = behind = 0
    head_tot = head_matched = 0          # the first four images of the fixed seed: ALL matched, next to the fractional bar over all 16
    for b in range(N_E2E):
        a = ev.detection_agreement(hd[b], od[b], 1e-4, hk[b], ok[b])
        assert a["n_a"] == a["n_b"] == cfg.max_detections, (b, a)
        assert a["max_score_diff"] < 1e-5 and a["max_mask_diff"] < 2e-4, (b, a)
        tot += a["n_a"]; matched += a["matched"]
        if b < 4:
            head_tot += a["n_a"]; head_matched += a["matched"]
        why = ev.mask_flip_causes(hd[b], od[b], hpool[b], opool[b], hk[b], ok[b])
        assert why["write_set_ok"], f"image {b}: a side's masks are not the compacted rows its own zero-sample predicate keeps"
        assert not why["unexplained"], f"image {b}: a kept / dropped difference that is NOT the removeZeros cliff: {why['unexplained']}"
        flips += why["flips"]; behind += why["masks_behind_flip"]
    assert matched >= 0.999 * tot, (matched, tot)
    # a fixed-seed equality beside the fraction: these 400 detections have matched one for one since round 2; a K-order change that
    # swaps a near-tie HERE is worth a look at the taps before the expectation is moved
    assert head_matched == head_tot, (head_matched, head_tot)
    print(f"e2e f32x3 vs oracle: {matched}/{tot} detections, {flips} removeZeros flips (all explained), {behind} masks behind a flip")
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


F16_MATCHED_FIRST4 = 397        # detections of the first four end-to-end images the fp16 mode shares with the fp32 oracle within 2e-3 (measured on two boxes, round 6: 397 / 400, 1575 / 1600 over all sixteen)


def test_fp16_mode_end_to_end_bar(pkg, full_model, e2e_oracle):
    """BASELINE configs[3] (fp16 tensors + fp16 MFMA) has an end-to-end bar of its own, the one
    include/maskrcnn_hip.h states for MRCNN_F16: at full size, batch 8, against the fp32 CPU oracle at least 95 % of the
    detections have a partner with the same class id and a box within 2e-3 (normalized), matched scores agree within 5e-4 and
    matched masks (in front of any removeZeros flip) within 3e-2; every image yields its full 100 detections.  The TEST holds the mode to
    what it measures (>= 98 %, 3e-4, 2e-2; round 6), tighter than the header's promise."""
    import importlib
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    ev = importlib.import_module("mask-rcnn-coreml_amd.evaluate")
    d, cfg = full_model
    images, od, ok, _ = e2e_oracle
    m = models.load_maskrcnn(d, max_batch=8, compute_dtype="f16")
    hd, hk, _ = _hip_predict_with_taps(m, images)
    tot = matched = matched4 = 0
    worst_score = worst_mask = 0.0
    for b in range(N_E2E):
        a = ev.detection_agreement(hd[b], od[b], 2e-3, hk[b], ok[b])
        assert a["n_a"] == a["n_b"] == cfg.max_detections, (b, a)
        tot += a["n_a"]; matched += a["matched"]
        matched4 += a["matched"] if b < 4 else 0
        worst_score = max(worst_score, a["max_score_diff"]); worst_mask = max(worst_mask, a["max_mask_diff"])
    print(f"e2e f16 vs oracle: {matched}/{tot} detections within 2e-3 ({matched4}/400 on the first four images), score diff {worst_score:.2e}, mask diff {worst_mask:.2e}")
    # Round 6: the bar is what is MEASURED, not what fp16 tensors could get away with — 1575 / 1600 (98.4 %), 1.0e-4, 1.4e-2 at this
    # round's kernels (round 3's 1584 predates the compact stem and the fused blocks): a K-order change that costs a point of agreement must show.
    assert matched >= 0.98 * tot, (matched, tot)
    assert worst_score < 3e-4 and worst_mask < 2e-2, (worst_score, worst_mask)
    # ... and on the first four images (fixed seeds, a deterministic engine) the count itself is pinned
    if F16_MATCHED_FIRST4 is not None:
        assert matched4 == F16_MATCHED_FIRST4, matched4
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


def test_removezeros_flips_between_two_engines_are_all_the_cliff(pkg, full_model):
    """The same proof on MORE images than the CPU oracle affords in a test: 64 full-size images, the headline mode (f32x3) against
    the exact-fp32 MFMA engine — two independent summation orders, both with taps.  Detection agreement >= 0.999, and every
    kept / dropped difference is an exact-zero sample on one side between boxes that agree to the last bits (the reference's
    semantics: TimeDistributedMaskLayer.swift:52-89, TimeDistributedClassifierLayer.swift:116-127); "masks behind a flip" — rows whose
    class lookup the compact-index rule shifts on one side — are counted, not compared."""
    import importlib
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    ev = importlib.import_module("mask-rcnn-coreml_amd.evaluate")
    d, cfg = full_model
    images = rand_images(64, 1024, 1024, seed=47)
    m3 = models.load_maskrcnn(d, max_batch=8, compute_dtype="f32x3")
    m3.calibrate_split(images[:2])
    d3, k3, p3 = _hip_predict_with_taps(m3, images)
    del m3
    m1 = models.load_maskrcnn(d, max_batch=8, compute_dtype="f32")
    d1, k1, p1 = _hip_predict_with_taps(m1, images)
    tot = matched = flips = behind = 0
    for b in range(64):
        a = ev.detection_agreement(d3[b], d1[b], 1e-4, k3[b], k1[b])
        tot += max(a["n_a"], a["n_b"]); matched += a["matched"]
        assert a["max_score_diff"] < 1e-5 and a["max_mask_diff"] < 2e-4, (b, a)
        why = ev.mask_flip_causes(d3[b], d1[b], p3[b], p1[b], k3[b], k1[b])
        assert why["write_set_ok"], f"image {b}: a side's masks are not the compacted rows its own zero-sample predicate keeps"
        assert not why["unexplained"], f"image {b}: {why['unexplained']}"
        flips += why["flips"]; behind += why["masks_behind_flip"]
    print(f"f32x3 vs f32 engine, 64 images: {matched}/{tot} detections, {flips} removeZeros flips (all explained), {behind} masks behind a flip")
    assert matched >= 0.999 * tot, (matched, tot)
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


def test_roi_align_fp16_tiny_samples_keep_their_rows(pkg, orc):
    """in fp16 a bilinear sample below ~3e-8 rounds to zero on store; the mask layer's removeZeros
    rule (a row is kept iff EVERY element != 0) must still see the fp32 value.  A map holding the smallest fp16
    subnormal gives interpolated samples far below the fp16 range: stored as 0, flagged as non-zero."""
    import ctypes as C
    L = __import__("importlib").import_module("mask-rcnn-coreml_amd._lib")
    lib = L.lib()
    Ch, P = 8, 14
    tiny = np.float16(6e-8)                                    # smallest positive fp16 subnormal (5.96e-8)
    maps = [np.full((s, s, Ch), tiny, np.float16) for s in (32, 16, 8, 4)]
    maps[0][:, 1::2, :] = np.float16(0)                        # P2 alternates tiny / 0 along x: samples in between are
                                                               # tiny*(1-lx) — non-zero in fp32, about half of them < 2^-25
    rois = np.array([[0.10, 0.05, 0.45, 0.40],                 # small box -> level P2 (log2(sqrt(wh)*128/224... ) + 4 = 2)
                     [0.0, 0.0, 0.0, 0.0]], np.float32)       # padding ROI
    Hs = (C.c_int * 4)(*[x.shape[0] for x in maps])
    Ws = (C.c_int * 4)(*[x.shape[1] for x in maps])
    ptrs = (C.c_void_p * 4)(*[x.ctypes.data for x in maps])
    out = np.empty((2, P, P, Ch), np.float16)
    flags = np.full(2, -1, np.int32)
    L.check(lib.mrcnn_roi_align_nhwc(ptrs, Hs, Ws, Ch, L.F16, rois.ctypes.data, 4, 2, P, 128.0, 128.0, L.HOST, out.ctypes.data,
                                     flags.ctypes.data))
    pyr32 = [np.ascontiguousarray(x.astype(np.float32).transpose(2, 0, 1)) for x in maps]
    want = orc.pyramid_roi_align(rois, pyr32, P, 128.0, 128.0)          # fp32 samples (n, C, P, P)
    np.testing.assert_array_equal(out.transpose(0, 3, 1, 2).astype(np.float32), want.astype(np.float16).astype(np.float32))
    valid32 = orc.mask_valid_rows(want)
    valid16 = orc.mask_valid_rows(want.astype(np.float16).astype(np.float32))
    want_flags = np.zeros(2, np.int32)
    want_flags[valid32] = 1
    np.testing.assert_array_equal(flags, want_flags)
    # the fixture really exercises the hazard: fp32 says "keep" where the rounded row would have been dropped
    assert want_flags[0] == 1 and want_flags[1] == 0
    assert 0 not in valid16, "fixture does not produce a sample that rounds to zero in fp16"
