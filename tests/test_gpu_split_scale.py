"""The scale-aware split (round 4; include/maskrcnn_hip.h: mrcnn_model_calibrate_split).

The split modes (MRCNN_F32X3, MRCNN_F32S) carry an fp32 activation exactly only while 0.5 <= |a| < 65504; below, to 2^-25
absolute.  The reference's CPU path is fp32 activations x fp16 weights (Conversion/task.py:90) at ANY scale, so a checkpoint
whose tensors sit at 1e-3 must not lose accuracy here.  Every tensor a split convolution reads is therefore stored as
2^e * value with a per-group exponent e folded into the producer's scale / shift; these tests pin

  * the mechanism at kernel level: the scale curve of one convolution is FLAT once the input is pre-scaled;
  * the engine: with calibrated exponents every staged oracle comparison still holds (taps, ROIAlign level multipliers, the
    fused RPN heads' multiplier, both heads), per-image results stay independent of the batch, and a second handle given the
    same exponent vector reproduces the bits (the sharding contract);
  * a checkpoint whose activations shrink by 2^-6 per stage (2^-28 at C5): uncalibrated the trunk collapses, calibrated it is
    fp32-grade; one whose activations leave the fp16 range: uncalibrated the watchdog fails the predict, calibrated it runs;
  * the diagnostic counters.
"""
import importlib
import os

import numpy as np
import pytest

from conftest import rand_images, make_model_dir
from test_gpu_conv_kernels import conv, torch_ref
from test_gpu_engine import _check_stages, _rel, _nhwc_to_chw

pytestmark = pytest.mark.gpu
L = importlib.import_module("mask-rcnn-coreml_amd._lib")


# ---------------------------------------------------------------------------------------------------------------------
# kernel level: one convolution, the same tensor at 2^0 ... 2^-20, with and without the power-of-two pre-scale
# ---------------------------------------------------------------------------------------------------------------------
def _prescale_exponent(x):
    """The engine's rule (Model::calibrate_split): max |a| * 2^e in [2^11, 2^12)."""
    _, k = np.frexp(float(np.abs(x).max()))
    return 12 - int(k)


@pytest.mark.parametrize("dtype", ["f32x3", "f32s"])
def test_prescaled_split_curve_is_flat(dtype):
    """VERDICT r3 item 2, done-criterion: <= 4e-6 at every scale down to 2^-20 (the un-prescaled curve loses a decade per 2^-4,
    tests/test_gpu_conv_kernels.py::test_split_modes_scale_curve_stays_inside_the_documented_bound)."""
    B, H, W, Ci, Co, k = 2, 32, 32, 256, 128, 3
    rng = np.random.default_rng(3)
    x = np.maximum(rng.standard_normal((B, H, W, Ci)), 0).astype(np.float32) * 4.0
    w = (rng.standard_normal((Co, k, k, Ci)) * np.sqrt(2.0 / (k * k * Ci))).astype(np.float16).astype(np.float32)
    one, zero = np.ones(Co, np.float32), np.zeros(Co, np.float32)
    plain, scaled = [], []
    for e in (0, -4, -8, -12, -16, -20):
        xs = np.ldexp(x, e).astype(np.float32)
        ref = torch_ref(xs, w, k, 1, one, zero, None, 0, dtype=dtype)
        got = conv(xs, w, k, 1, one, zero, None, act=0, dtype=dtype).astype(np.float64)
        plain.append(float(np.abs(got - ref).max() / np.abs(ref).max()))
        # what the engine does: the tensor is STORED as 2^p * value, the consumer's scale carries 2^-p (both exact)
        p = _prescale_exponent(xs)
        got = conv(np.ldexp(xs, p).astype(np.float32), w, k, 1, np.ldexp(one, -p).astype(np.float32), zero, None, act=0, dtype=dtype).astype(np.float64)
        scaled.append(float(np.abs(got - ref).max() / np.abs(ref).max()))
    bar = 4e-6 if dtype == "f32x3" else 4e-6 + 2.0 ** -21        # the two-part split carries 22-23 bits at any scale
    assert max(scaled) <= bar, scaled
    assert max(scaled) <= 2.5 * min(scaled) + 1e-7, scaled          # flat
    assert plain[-1] > 100 * scaled[-1], (plain, scaled)            # ... where the un-prescaled split is not


# ---------------------------------------------------------------------------------------------------------------------
# engine level
# ---------------------------------------------------------------------------------------------------------------------
def _models():
    return importlib.import_module("mask-rcnn-coreml_amd.models")


def test_calibrated_engine_keeps_every_staged_parity(pkg, orc, small_model):
    """After calibration the tensors in HBM are scaled — taps hand out true values, the ROIAlign sampler brings pyramid levels
    at different exponents to its output's, the fused heads undo theirs: every stage still equals the oracle on the GPU's own
    taps (bit-exact for the box / index stages)."""
    from oracle.network import load_oracle_model
    d, cfg = small_model
    om = load_oracle_model(d)
    B = 3
    m = _models().load_maskrcnn(d, max_batch=B, compute_dtype="f32x3")
    images = rand_images(B, cfg.image_height, cfg.image_width, seed=1)
    det0, mask0 = m.predict(images)
    assert not np.any(m.split_exponents)                                   # fresh handle: every exponent 0
    totals = m.calibrate_split(images)
    assert totals["split_calibrated"] == 1 and totals["split_inputs_counted"] > 0
    exps = m.split_exponents
    rep = m.split_report()
    assert any(e != 0 for e in exps), "the calibration changed nothing: the test has no teeth"
    for g, e in zip(rep, exps):
        assert g["exponent"] == e
        if g["fixed"]:
            assert e == 0
        elif g["absmax"] > 0:
            assert 2.0 ** 11 <= g["absmax"] * 2.0 ** e < 2.0 ** 12, g
    det, mask = m.predict(images)
    trunk = om.trunk(images)
    for b in range(B):
        d_b, m_b = _check_stages(pkg, orc, om, m, cfg, images, b, True, trunk)
        np.testing.assert_array_equal(det[b], d_b)
        np.testing.assert_array_equal(mask[b].reshape(cfg.max_detections, -1), m_b)
    # the calibrated engine stays within summation noise of the uncalibrated one on this O(1-100) model
    assert np.abs(det[..., :4] - det0[..., :4]).max() < 1e-4 or not np.array_equal(det[..., 4], det0[..., 4])
    # per-image results do not depend on the batch (the sharding contract), calibrated
    for b in range(B):
        d1, m1 = m.predict(images[b:b + 1])
        np.testing.assert_array_equal(d1[0], det[b])
        np.testing.assert_array_equal(m1[0], mask[b])
    # a second handle given the same exponent vector reproduces the bits (what a sharded job does on every rank)
    m2 = _models().load_maskrcnn(d, max_batch=B, compute_dtype="f32x3")
    m2.split_exponents = exps
    det2, mask2 = m2.predict(images)
    np.testing.assert_array_equal(det2, det)
    np.testing.assert_array_equal(mask2, mask)
    # exponents of the groups fp32 arithmetic consumes cannot be set
    bad = exps.copy()
    bad[[i for i, g in enumerate(rep) if g["fixed"]][0]] = 3
    with pytest.raises(L.MrcnnError):
        m2.split_exponents = bad


def test_calibration_is_refused_where_it_means_nothing(small_model):
    d, cfg = small_model
    images = rand_images(1, cfg.image_height, cfg.image_width, seed=1)
    for dt in ("f32", "f16"):
        m = _models().load_maskrcnn(d, max_batch=1, compute_dtype=dt)
        with pytest.raises(L.MrcnnError) as e:
            m.calibrate_split(images)
        assert e.value.code == 5                                           # MRCNN_ERR_UNSUPPORTED


def _rescaled_model(tmp_path_factory, pkg, weights_mod, name, conv1_exp, stage_exp):
    """The small synthetic model made positively homogeneous (every trunk bias, BatchNorm beta and BatchNorm mean zeroed: conv →
    scale → ReLU only) and then moved by powers of two: conv1's BatchNorm gamma times 2^conv1_exp, and the two BatchNorms that
    produce a stage's first block output (branch2c and branch1) times 2^stage_exp — so every tensor of stage st sits at
    2^(conv1_exp + (st - 1) * stage_exp) of the homogeneous model's O(1-100), the FPN levels at their stage's scale.  The RPN
    heads keep their biases (proposals and detections still come out); the oracle reads the same files."""
    d, cfg = make_model_dir(tmp_path_factory, pkg, weights_mod, name, architecture="resnet50", input_image_shape=(128, 128, 3),
                            num_classes=21, pre_nms_max_proposals=300, max_proposals=64, max_detections=16)
    path = os.path.join(d, "MaskRCNN.mrcw")
    meta, t = weights_mod.read_mrcw(path)
    t = dict(t)
    for k in list(t):
        if k.endswith("/beta") or k.endswith("/mean") or (k.endswith("/bias") and not k.startswith(("rpn_class_raw", "rpn_bbox_pred"))):
            t[k] = np.zeros_like(t[k])

    def mul(bn, e):
        t[f"{bn}/gamma"] = np.ldexp(np.asarray(t[f"{bn}/gamma"], np.float32), e).astype("<f2")
    mul("bn_conv1", conv1_exp)
    for st in (2, 3, 4, 5):
        mul(f"bn{st}a_branch2c", stage_exp)
        mul(f"bn{st}a_branch1", stage_exp)
    weights_mod.write_mrcw(path, meta, t)
    return d, cfg


def _trunk_errors(m, om, cfg, images):
    m.predict(images)
    pyr, _, _ = om.trunk(images)
    shapes = cfg.feature_shapes()
    return [_rel(_nhwc_to_chw(m.read_tensor(f"P{l + 2}", 0), shapes[l][0], shapes[l][1], 256), pyr[l][0]) for l in range(4)]


def test_a_checkpoint_with_tiny_activations_is_fp32_grade_once_calibrated(pkg, weights_mod, tmp_path_factory):
    """activations shrinking by 2^-6 per stage (conv1 2^-4, C2 2^-10 ... C5 2^-28 of the stock model's O(1-100)).
    fp32 does not care (the oracle network is fp32 throughout); the uncalibrated split modes collapse on the deep levels; with
    calibrated exponents they sit at fp32 summation noise on every level, and the diagnostic says so beforehand."""
    from oracle.network import load_oracle_model
    d, cfg = _rescaled_model(tmp_path_factory, pkg, weights_mod, "tiny_act", -4, -6)
    om = load_oracle_model(d)
    images = rand_images(1, 128, 128, seed=2)
    m32 = _models().load_maskrcnn(d, max_batch=1, compute_dtype="f32")
    e32 = _trunk_errors(m32, om, cfg, images)
    assert max(e32) < 2e-5, e32                                           # the exact-fp32 engine: scale-free
    for dt in ("f32x3", "f32s"):
        m = _models().load_maskrcnn(d, max_batch=1, compute_dtype=dt)
        raw = _trunk_errors(m, om, cfg, images)
        diag = m.calibrate_split(images, apply=False)                      # diagnose only: the exponents stay 0 ...
        assert not np.any(m.split_exponents) and diag["split_calibrated"] == 0
        assert diag["split_max_exponent"] == 0 and diag["split_min_exponent"] == 0
        proposed = [g for g in m.split_report() if not g["fixed"] and g["absmax"] > 0]
        assert min(g["absmax"] for g in proposed) < 1e-4                   # ... and the report shows where the checkpoint sits
        m.calibrate_split(images)
        cal = _trunk_errors(m, om, cfg, images)
        bar = 2e-5 if dt == "f32x3" else 4e-5
        assert max(cal) < bar, (dt, cal)
        assert max(cal) < 4 * max(e32) + 1e-6, (dt, cal, e32)              # as good as the exact-fp32 engine
        assert max(raw) > 50 * max(cal), (dt, raw, cal)                    # and the uncalibrated mode is NOT: the test has teeth
        assert m.get_int("split_max_exponent") >= 20


def test_a_checkpoint_that_leaves_the_fp16_range_runs_once_calibrated(pkg, weights_mod, tmp_path_factory):
    """Activations 2^12 times the stock model's.  Round 4 failed every uncalibrated predict (MRCNN_ERR_UNSUPPORTED); since round 5 the
    FIRST predict recovers by itself — the batch is measured where it sits on the device, the exponents are lowered, the batch is
    computed again ("range_recoveries") — and later predicts run inside the range.  An explicit calibration gives the same grade."""
    from oracle.network import load_oracle_model
    d, cfg = _rescaled_model(tmp_path_factory, pkg, weights_mod, "huge_act", 12, 0)
    om = load_oracle_model(d)
    images = rand_images(1, 128, 128, seed=2)
    m = _models().load_maskrcnn(d, max_batch=1, compute_dtype="f32x3")
    first = _trunk_errors(m, om, cfg, images)                              # (an uncalibrated predict: trips, recovers, returns valid results)
    assert max(first) < 2e-5, first
    assert m.get_int("range_overflows") == 1 and m.get_int("range_recoveries") == 1
    assert m.get_int("split_min_exponent") < 0
    again = _trunk_errors(m, om, cfg, images)
    assert max(again) < 2e-5 and m.get_int("range_recoveries") == 1        # the lowered exponents hold: no second recovery
    m.calibrate_split(images)
    cal = _trunk_errors(m, om, cfg, images)
    assert max(cal) < 2e-5, cal
    assert m.get_int("range_overflows") == 1                               # calibration passes are not the host's predicts


def test_an_input_far_above_the_calibrated_range_recovers_and_matches_the_oracle(pkg, weights_mod, tmp_path_factory):
    """the headline mode must not fail on data the reference's fp32 path handles (Conversion/task.py:90).  A
    positively homogeneous model is calibrated on a nearly mean-coloured image (|pixel - mean| < 1: every activation ~2^7 below
    what a real image produces); the next image drives the activations 2^6 .. 2^9 above the calibrated maximum — past the 16x head
    room.  The predict recovers (no error), agrees with the fp32 oracle end to end, both through the synchronous entry and through
    submit / collect; images of one batch still do not depend on their batch mates."""
    from oracle.network import load_oracle_model
    d, cfg = _rescaled_model(tmp_path_factory, pkg, weights_mod, "recover", 0, 0)
    om = load_oracle_model(d)
    flat = np.empty((1, 128, 128, 3), np.uint8)
    flat[...] = np.array([124, 117, 104], np.uint8)                         # mean pixel (123.7, 116.8, 103.9) + < 1
    images = rand_images(2, 128, 128, seed=9)
    for entry in ("predict", "collect"):
        m = _models().load_maskrcnn(d, max_batch=2, compute_dtype="f32x3")
        m.calibrate_split(flat)
        hi = int(m.get_int("split_max_exponent"))
        assert hi >= 12, hi                                                 # the calibration image really was tiny
        if entry == "predict":
            det, mask = m.predict(images)
        else:
            det = np.empty((2, m.max_detections, 6), np.float32)
            mask = np.empty((2, m.max_detections, m.mask_size, m.mask_size), np.float32)
            m.submit(images)
            assert m.collect(det, mask) == 2
        assert m.get_int("range_recoveries") == 1 and m.get_int("range_overflows") == 1
        assert int(m.get_int("split_max_exponent")) < hi
        pyr, _, _ = om.trunk(images)
        shapes = cfg.feature_shapes()
        for b in range(2):
            err = [_rel(_nhwc_to_chw(m.read_tensor(f"P{l + 2}", b), shapes[l][0], shapes[l][1], 256), pyr[l][b]) for l in range(4)]
            assert max(err) < 2e-5, (entry, b, err)
        # batch independence with the recovered exponents
        d1, m1 = m.predict(images[1:2])
        np.testing.assert_array_equal(d1[0], det[1])
        np.testing.assert_array_equal(m1[0], mask[1])
        assert m.get_int("range_recoveries") == 1


def test_a_sharded_predict_tells_every_rank_that_a_rank_recovered(pkg, weights_mod, tmp_path_factory):
    """a range recovery lowers the split exponents of ONE rank's handle for good — from then on its bits for an image differ from
    its peers'.  The exchange therefore carries, beside the status word, how often the rank's predict recovered (word 2 of the slot's
    trailer): mrcnn_dist_recovered reports it on every rank, so that the host can redistribute the lowered vector.  World 1 through RCCL:
    the call that recovers reports [1] and returns the recovered (oracle-grade) results, the next call reports [0]."""
    dmod = importlib.import_module("mask-rcnn-coreml_amd.dist")
    d, cfg = _rescaled_model(tmp_path_factory, pkg, weights_mod, "recover_dist", 0, 0)
    flat = np.empty((1, 128, 128, 3), np.uint8)
    flat[...] = np.array([124, 117, 104], np.uint8)
    images = rand_images(2, 128, 128, seed=9)
    m = _models().load_maskrcnn(d, max_batch=2, compute_dtype="f32x3")
    m.calibrate_split(flat)
    nd = dmod.NativeDist(0, 1, dmod.NativeDist.unique_id())
    try:
        assert nd.recovered() == [0]
        det, mask = nd.predict_sharded(m, images)
        assert nd.recovered() == [1] and m.get_int("range_recoveries") == 1
        want_d, want_m = m.predict(images)                                  # (the handle now holds the lowered exponents: same bits)
        np.testing.assert_array_equal(det, want_d)
        np.testing.assert_array_equal(mask, want_m)
        nd.predict_sharded(m, images)
        assert nd.recovered() == [0] and m.get_int("range_recoveries") == 1
    finally:
        nd.close()
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


def test_exponents_stored_in_the_artefact_make_the_drop_in_load_calibrated(pkg, weights_mod, tmp_path_factory):
    """`convert --calibrate` stores the exponent vector in MaskRCNN.mrcw; mrcnn_model_load applies it, so
    MaskRCNN().prediction(image) (ViewController.swift:37) needs no extra call.  The reloaded model reproduces the calibrated one
    bit for bit, a batch equals its single-image calls, and the tiny-activation checkpoint is fp32-grade straight from load."""
    from oracle.network import load_oracle_model
    convert = importlib.import_module("mask-rcnn-coreml_amd.convert")
    d, cfg = _rescaled_model(tmp_path_factory, pkg, weights_mod, "stored_exp", -4, -6)
    om = load_oracle_model(d)
    images = rand_images(3, 128, 128, seed=2)
    m = _models().load_maskrcnn(d, max_batch=3, compute_dtype="f32x3")
    assert m.get_int("split_exponents_from_artefact") == 0 and m.get_int("split_calibrated") == 0
    m.calibrate_split(images[:2])
    want_e = m.split_exponents.copy()
    det, mask = m.predict(images)
    del m
    convert.calibrate_artefact(d, images[:2], verbose=False)               # what `convert --calibrate` runs after writing the artefacts
    meta, _ = weights_mod.read_mrcw(os.path.join(d, "MaskRCNN.mrcw"))
    assert sum(k.startswith("split_exp.") for k in meta) >= 50
    for dt in ("f32x3", "f32s"):
        m2 = _models().load_maskrcnn(d, max_batch=3, compute_dtype=dt)
        assert m2.get_int("split_exponents_from_artefact") == 1 and m2.get_int("split_calibrated") == 1
        np.testing.assert_array_equal(m2.split_exponents, want_e)
        d2, k2 = m2.predict(images)
        if dt == "f32x3":
            np.testing.assert_array_equal(d2, det)
            np.testing.assert_array_equal(k2, mask)
        for b in range(3):
            d1, k1 = m2.predict(images[b:b + 1])
            np.testing.assert_array_equal(d1[0], d2[b])
            np.testing.assert_array_equal(k1[0], k2[b])
        err = _trunk_errors(m2, om, cfg, images[:1])
        assert max(err) < (2e-5 if dt == "f32x3" else 4e-5), (dt, err)
        assert m2.get_int("range_recoveries") == 0
    m16 = _models().load_maskrcnn(d, max_batch=1, compute_dtype="f16")      # modes without a split ignore the stored vector
    assert m16.get_int("split_exponents_from_artefact") == 0
    del m16, m2
    # Round 6: a host that names NO precision — `MaskRCNN()` in ViewController.swift:37; here load_maskrcnn(d) / MRCNN_DEFAULT —
    # gets the mode the artefact is prepared for: the stored exponents make it f32x3, and the handle agrees bit for bit with the explicitly
    # loaded and explicitly calibrated one above, and with the CPU oracle's staged parity
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    md = _models().load_maskrcnn(d, max_batch=3)
    assert md.compute_dtype == "f32x3" and md.compute_dtype_defaulted and md.get_int("compute_dtype") == L.F32X3
    assert md.get_int("split_exponents_from_artefact") == 1 and md.get_int("split_calibrated") == 1
    np.testing.assert_array_equal(md.split_exponents, want_e)
    dd, kd = md.predict(images)
    np.testing.assert_array_equal(dd, det)
    np.testing.assert_array_equal(kd, mask)
    assert max(_trunk_errors(md, om, cfg, images[:1])) < 2e-5
    del md


def test_a_default_load_of_an_uncalibrated_artefact_is_the_exact_fp32_mode(pkg, orc, small_model):
    """MRCNN_DEFAULT on an artefact WITHOUT stored exponents (and on the stand-alone Classifier / Mask models) resolves to MRCNN_F32 —
    the scale-invariant engine — and is the same handle an explicit "f32" load gives; staged oracle parity holds as for that mode."""
    from test_gpu_engine import _check_stages
    from oracle.network import load_oracle_model
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    d, cfg = small_model
    images = rand_images(2, cfg.image_height, cfg.image_width, seed=9)
    m0 = _models().load_maskrcnn(d, max_batch=2)
    assert m0.compute_dtype == "f32" and m0.compute_dtype_defaulted and m0.get_int("compute_dtype") == L.F32
    m1 = _models().load_maskrcnn(d, max_batch=2, compute_dtype="f32")
    assert not m1.compute_dtype_defaulted
    d0, k0 = m0.predict(images)
    d1, k1 = m1.predict(images)
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(k0, k1)
    om = load_oracle_model(d)
    _check_stages(pkg, orc, om, m0, cfg, images, 1, True, om.trunk(images[1:2]), tb=0)
    c = _models().Classifier(os.path.join(d, "Classifier.mrcw"), max_rows=4)
    assert c.get_int("compute_dtype") == L.F32 and c.get_int("compute_dtype_defaulted") == 1
    import ctypes as C
    h = C.c_void_p()
    assert L.lib().mrcnn_model_load(1, os.path.join(d, "Classifier.mrcw").encode(), 4, 7, C.byref(h)) != 0 and not h.value      # an unknown compute dtype is still refused


def test_split_counters_mean_what_the_header_says(small_model):
    """split_inexact_inputs counts the non-zero STORED inputs below 0.5 (the ones a three-part split carries to 2^-25 absolute);
    calibration moves every tensor's maximum to [2^11, 2^12), so far fewer inputs are inexact afterwards."""
    d, cfg = small_model
    images = rand_images(2, cfg.image_height, cfg.image_width, seed=4)
    m = _models().load_maskrcnn(d, max_batch=2, compute_dtype="f32x3")
    m.calibrate_split(images)
    after = {k: m.get_int(k) for k in ("split_small_inputs", "split_inexact_inputs", "split_inputs_counted")}
    assert 0 < after["split_inexact_inputs"] <= after["split_small_inputs"] <= after["split_inputs_counted"]
    # calibrated: the stored inputs below 0.5 are below 2^-12 of their tensor's maximum, a strict subset of those below 2^-8 of it
    for g in m.split_report():
        assert g["inexact_inputs"] <= g["small_inputs"] <= g["inputs_counted"], g


def test_calibration_from_a_device_tensor_equals_calibration_from_host_memory(small_model):
    """calibrate_split with MRCNN_DEVICE images used to stage its throw-away outputs in pageable host vectors behind a
    device-to-device copy kind.  The passes now copy nothing out; a CUDA tensor and the same images in host memory must give the same
    exponents, the same diagnostics and bit-equal predicts; a batch beyond max_batch is refused before anything runs."""
    import torch
    d, cfg = small_model
    images = rand_images(2, cfg.image_height, cfg.image_width, seed=6)
    ma = _models().load_maskrcnn(d, max_batch=2, compute_dtype="f32x3")
    mb = _models().load_maskrcnn(d, max_batch=2, compute_dtype="f32x3")
    ra = ma.calibrate_split(images)
    rb = mb.calibrate_split(torch.from_numpy(images).to("cuda:0"))
    assert ra == rb
    np.testing.assert_array_equal(ma.split_exponents, mb.split_exponents)
    da, ka = ma.predict(images)
    db, kb = mb.predict(images)
    np.testing.assert_array_equal(da, db)
    np.testing.assert_array_equal(ka, kb)
    assert ma.get_int("range_overflows") == 0 and ma.get_int("predict_calls") == 1          # calibration passes are not the host's predicts
    with pytest.raises(L.MrcnnError):
        ma.calibrate_split(rand_images(3, cfg.image_height, cfg.image_width, seed=6))
