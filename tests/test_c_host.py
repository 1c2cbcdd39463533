"""The C ABI from a non-Python host: include/maskrcnn_hip.h must be a self-contained C99 / C++17 header,
examples/maskrcnn_predict.c (the `maskrcnn evaluate` flow in plain C, standing where the Swift host would)
must build against libmaskrcnn_hip.so with nothing but that header — and, on a GPU box, produce the very
numbers the Python mirror produces through ctypes."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import HAS_GPU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "mask-rcnn-coreml_amd")
SO = os.path.join(LIBDIR, "libmaskrcnn_hip.so")


def _build_example(tmp_path, name="maskrcnn_predict"):
    if not os.path.exists(SO):
        pytest.skip("libmaskrcnn_hip.so not built (run python __graft_entry__.py)")
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", INC,
           os.path.join(ROOT, "examples", name + ".c"), "-L", LIBDIR, "-lmaskrcnn_hip",
           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_cxx(tmp_path):
    src = tmp_path / "use.c"
    src.write_text('#include "maskrcnn_hip.h"\n#include "maskrcnn_hip_test.h"\nint main(void) { mrcnn_tensor t; mrcnn_param p; mrcnn_detection d; mrcnn_conv_shape_stat s;'
                   ' (void)t; (void)p; (void)d; (void)s; return (int)MRCNN_OK; }\n')
    for cc, std in (("gcc", "-std=c99"), ("gcc", "-std=c11"), ("g++", "-std=c++17")):
        lang = ["-x", "c++"] if cc == "g++" else []
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", INC] + lang + [str(src)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (cc, std, r.stderr)
    hdr = open(os.path.join(INC, "maskrcnn_hip.h")).read()
    import re
    assert sorted(re.findall(r"^#include\s+(\S+)", hdr, re.M)) == ["<stddef.h>", "<stdint.h>"]     # no HIP / C++ / torch headers
    thdr = open(os.path.join(INC, "maskrcnn_hip_test.h")).read()
    assert re.findall(r"^#include\s+(\S+)", thdr, re.M) == ['"maskrcnn_hip.h"']


def test_c_host_builds_and_fails_loudly_without_gpu(tmp_path):
    exe = _build_example(tmp_path)
    if HAS_GPU:
        pytest.skip("GPU present: covered by test_c_host_matches_python_mirror")
    (tmp_path / "x.rgb").write_bytes(bytes(4 * 4 * 3))
    r = subprocess.run([exe, str(tmp_path), str(tmp_path / "x.rgb"), "4", "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3                                      # MRCNN_ERR_HIP
    assert "no CPU fallback" in r.stderr and r.stdout == ""


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16", "f32s", "f32x3"])
def test_c_host_matches_python_mirror(pkg, small_model, tmp_path, dtype):
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    ev = importlib.import_module("mask-rcnn-coreml_amd.evaluate")
    exe = _build_example(tmp_path)
    d, cfg = small_model
    img = np.random.default_rng(21).integers(0, 256, (100, 150, 3), dtype=np.uint8)      # non-square: letterboxed
    (tmp_path / "img.rgb").write_bytes(img.tobytes())
    env = dict(os.environ)
    r = subprocess.run([exe, d, str(tmp_path / "img.rgb"), "100", "150", dtype], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("seconds ") and float(lines[0].split()[1]) > 0
    n = int(lines[1].split()[1])
    rows = [l.split() for l in lines[2:]]
    assert len(rows) == n

    m = models.load_maskrcnn(d, max_batch=1, compute_dtype=dtype)
    out = m.prediction(ev.letterbox(img, m.image_height, m.image_width))
    dets = pkg.Detection.detectionsFromFeatureValue(out["detections"], out["mask"])
    assert n == len(dets) and n > 0
    for row, dd in zip(rows, dets):
        idx, cls = int(row[0]), int(row[1])
        score, x, y, w, h, msum = (float(v) for v in row[2:])
        assert (idx, cls) == (dd.index, dd.classId)
        assert score == float(dd.score)
        bx, by, bw, bh = dd.boundingBox
        assert (x, y, w, h) == (float(bx), float(by), float(bw), float(bh))
        assert msum == float(np.asarray(out["mask"][idx], dtype=np.float64).sum())



@pytest.mark.gpu
def test_c_host_that_names_no_precision_runs_the_mode_the_artefact_is_prepared_for(pkg, weights_mod, tmp_path_factory, tmp_path):
    """Round 6: `maskrcnn_predict <dir> <image> <h> <w>` — no fifth argument, as `MaskRCNN()` in ViewController.swift:37 names
    no precision — passes MRCNN_DEFAULT.  On an artefact calibrated the way `convert --calibrate` does the plain-C host must print the numbers
    of the Python mirror's f32x3 handle (stored exponents applied), on an uncalibrated one those of the exact-fp32 handle; the host runs as a
    PRODUCTION process: without MRCNN_TEST_KNOBS in its environment (and with a stray MRCNN_HALO=0, which it must ignore)."""
    from conftest import make_model_dir
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    ev = importlib.import_module("mask-rcnn-coreml_amd.evaluate")
    convert = importlib.import_module("mask-rcnn-coreml_amd.convert")
    exe = _build_example(tmp_path)
    img = np.random.default_rng(23).integers(0, 256, (90, 120, 3), dtype=np.uint8)
    (tmp_path / "img.rgb").write_bytes(img.tobytes())
    env = {k: v for k, v in os.environ.items() if k != "MRCNN_TEST_KNOBS"}
    env["MRCNN_HALO"] = "0"
    for calibrated, want_mode in ((False, "f32"), (True, "f32x3")):
        d, cfg = make_model_dir(tmp_path_factory, pkg, weights_mod, "chost_default" + str(int(calibrated)), architecture="resnet50",
                                input_image_shape=(128, 128, 3), num_classes=21, pre_nms_max_proposals=300, max_proposals=64, max_detections=16)
        if calibrated:
            convert.calibrate_artefact(d, np.random.default_rng(7).integers(0, 256, (2, 128, 128, 3), dtype=np.uint8), verbose=False)
        r = subprocess.run([exe, d, str(tmp_path / "img.rgb"), "90", "120"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        n = int(lines[1].split()[1])
        rows = [l.split() for l in lines[2:]]
        m = models.load_maskrcnn(d, max_batch=1)                      # the mirror names no precision either
        assert m.compute_dtype == want_mode and m.compute_dtype_defaulted
        out = m.prediction(ev.letterbox(img, m.image_height, m.image_width))
        dets = pkg.Detection.detectionsFromFeatureValue(out["detections"], out["mask"])
        assert n == len(dets) == len(rows) and n > 0
        for row, dd in zip(rows, dets):
            assert (int(row[0]), int(row[1])) == (dd.index, dd.classId) and float(row[2]) == float(dd.score)
            assert tuple(float(v) for v in row[3:7]) == tuple(float(v) for v in dd.boundingBox)
            assert float(row[7]) == float(np.asarray(out["mask"][int(row[0])], dtype=np.float64).sum())
        del m
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


def test_mgpu_c_host_builds_and_fails_loudly_without_gpu(tmp_path):
    exe = _build_example(tmp_path, "maskrcnn_predict_mgpu")
    if HAS_GPU:
        pytest.skip("GPU present: covered by test_mgpu_c_host_world_one")
    (tmp_path / "x.rgb").write_bytes(bytes(4 * 4 * 3))
    r = subprocess.run([exe, str(tmp_path), str(tmp_path / "x.rgb"), "1"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, RANK="0", WORLD_SIZE="1"))
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and r.stdout == ""        # MRCNN_ERR_HIP from mrcnn_dist_unique_id


@pytest.mark.gpu
def test_mgpu_c_host_world_one(pkg, small_model, tmp_path):
    """The plain-C multi-GPU host at world size 1 (one GPU here): rendezvous id, ncclCommInitRank, shard, predict, ncclAllGather —
    its printed records must be those of the Python mirror's batched predict."""
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    exe = _build_example(tmp_path, "maskrcnn_predict_mgpu")
    d, cfg = small_model
    B = 3
    imgs = np.random.default_rng(22).integers(0, 256, (B, cfg.image_height, cfg.image_width, 3), dtype=np.uint8)
    (tmp_path / "imgs.rgb").write_bytes(imgs.tobytes())
    r = subprocess.run([exe, d, str(tmp_path / "imgs.rgb"), str(B), "f32x3"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, RANK="0", WORLD_SIZE="1"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    while lines and not lines[0].startswith("world "):        # RCCL prints its version banner on stdout at communicator creation
        lines.pop(0)
    assert lines and lines[0].startswith("world 1 batch 3 seconds ")
    m = models.load_maskrcnn(d, max_batch=B, compute_dtype="f32x3")
    det, mask = m.predict(imgs)
    pos = 1
    total = 0
    for b in range(B):
        dets = pkg.Detection.detectionsFromFeatureValue(det[b], mask[b])
        assert lines[pos] == f"image {b} detections {len(dets)}"
        for row, dd in zip((l.split() for l in lines[pos + 1:pos + 1 + len(dets)]), dets):
            assert (int(row[0]), int(row[1])) == (dd.index, dd.classId) and float(row[2]) == float(dd.score)
            assert float(row[7]) == float(np.asarray(mask[b][dd.index], dtype=np.float64).sum())
        pos += 1 + len(dets)
        total += len(dets)
    assert pos == len(lines) and total > 0
    pkg.MaskRCNNConfig.defaultConfig().anchorsURL = None


def test_stream_c_host_builds(tmp_path):
    _build_example(tmp_path, "maskrcnn_predict_stream")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,calibrate", [("f32x3", True), ("f16", False)])
def test_pipelined_host_entry_equals_the_synchronous_one(pkg, small_model, tmp_path, dtype, calibrate):
    """mrcnn_maskrcnn_submit / _collect (the copy of batch i + 1 under the predict of batch i, two batches in
    flight) — through the plain-C host of examples/maskrcnn_predict_stream.c and through the Python mirror — returns the bits of
    the synchronous mrcnn_maskrcnn_predict for every image of a stream of batches, ragged last batch included."""
    models = importlib.import_module("mask-rcnn-coreml_amd.models")
    exe = _build_example(tmp_path, "maskrcnn_predict_stream")
    d, cfg = small_model
    nb, B = 5, 2
    imgs = np.random.default_rng(27).integers(0, 256, (nb * B, cfg.image_height, cfg.image_width, 3), dtype=np.uint8)
    (tmp_path / "imgs.rgb").write_bytes(imgs.tobytes())
    args = [exe, d, str(tmp_path / "imgs.rgb"), str(nb), str(B), dtype] + (["calibrate"] if calibrate else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("image ")]
    assert len(lines) == nb * B and r.stdout.strip().splitlines()[-1].startswith(f"batches {nb} batch {B} seconds ")

    m = models.load_maskrcnn(d, max_batch=B, compute_dtype=dtype)
    if calibrate:
        m.calibrate_split(imgs[:B])
    want = [m.predict(imgs[i:i + B]) for i in range(0, nb * B, B)]
    for i, line in enumerate(lines):
        t = line.split()
        det, mask = want[i // B][0][i % B], want[i // B][1][i % B]
        assert int(t[1]) == i and int(t[3]) == int((det[:, 5].astype(np.float64) > 0.7).sum())
        assert float(t[5]) == float(det.astype(np.float64).sum() + mask.astype(np.float64).sum()) or \
            abs(float(t[5]) - (det.astype(np.float64).sum() + mask.astype(np.float64).sum())) < 1e-6 * abs(float(t[5]))
    # the Python mirror: same loop, bit-equal records, a ragged last batch, and the two-in-flight limit
    det = np.empty((B, cfg.max_detections, 6), np.float32)
    mask = np.empty((B, cfg.max_detections, m.mask_size, m.mask_size), np.float32)
    batches = [np.ascontiguousarray(imgs[i:i + B]) for i in range(0, nb * B, B)] + [np.ascontiguousarray(imgs[:1])]
    m.submit(batches[0])
    for i in range(len(batches)):
        if i + 1 < len(batches):
            m.submit(batches[i + 1])
        n = m.collect(det, mask)
        ref = m.predict(batches[i]) if i == len(batches) - 1 else want[i]
        assert n == batches[i].shape[0]
        np.testing.assert_array_equal(det[:n], ref[0])
        np.testing.assert_array_equal(mask[:n], ref[1])
    L = importlib.import_module("mask-rcnn-coreml_amd._lib")
    m.submit(batches[0]); m.submit(batches[1])
    with pytest.raises(L.MrcnnError):
        m.submit(batches[2])
    m.collect(det, mask); m.collect(det, mask)
    with pytest.raises(L.MrcnnError):
        m.collect(det, mask)
