"""`maskrcnn convert` counterpart: the pure-Python HDF5 subset reader pinned against
a file written by the real h5py/libhdf5 (tests/golden/make_keras_h5.py), the Keras → .mrcw tensor
re-layout, and — when the image's conda interpreter with h5py is around — the whole
weights.h5 + config.json → products/ path on a full ResNet-50 checkpoint."""
import gzip
import importlib
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CONDA_PY = "/opt/conda/bin/python3.9"


@pytest.fixture(scope="module")
def hdf5():
    return importlib.import_module("mask-rcnn-coreml_amd.hdf5")


@pytest.fixture(scope="module")
def conv():
    return importlib.import_module("mask-rcnn-coreml_amd.convert")


@pytest.fixture()
def tiny_h5(tmp_path):
    p = tmp_path / "keras_tiny.h5"
    with gzip.open(os.path.join(GOLDEN, "keras_tiny.h5.gz")) as f:
        p.write_bytes(f.read())
    return str(p)


def test_hdf5_reader_matches_h5py_written_file(hdf5, tiny_h5):
    exp = np.load(os.path.join(GOLDEN, "keras_tiny.npz"))
    f = hdf5.File(tiny_h5)
    # root attrs as Keras 2.1.6 writes them; 298 groups → a two-level B-tree under the root group
    assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.1.6"
    assert [n.decode() for n in f.attrs["layer_names"]] == list(exp["layer_names"])
    assert sorted(f.keys()) == sorted(exp["layer_names"])
    assert [w.decode() for w in f["conv1"].attrs["weight_names"]] == ["conv1/kernel:0", "conv1/bias:0"]
    assert len(f["activation_0"].attrs["weight_names"]) == 0 and f["activation_0"].keys() == []
    # nested path lookup and dataset description
    d = f["rpn_model/rpn_conv_shared/kernel:0"]
    assert d.shape == (3, 3, 8, 16) and d.dtype == np.dtype("<f4")
    with pytest.raises(KeyError):
        f["conv1/conv1/nope"]
    # every dataset bit-equal, dtype included (f32, f64, f16, i32, u8, a scalar)
    w = hdf5.read_keras_weights(tiny_h5)
    names = [k for k in exp.files if k != "layer_names"]
    assert set(names) == set(w) and len(names) == 409
    for k in names:
        assert w[k].dtype == exp[k].dtype and w[k].shape == exp[k].shape, k
        np.testing.assert_array_equal(w[k], exp[k], err_msg=k)


def test_hdf5_reader_rejects_what_it_does_not_parse(hdf5, tmp_path, tiny_h5):
    p = tmp_path / "not.h5"
    p.write_bytes(b"MRCW" + b"\0" * 64)
    with pytest.raises(hdf5.HDF5FormatError, match="not an HDF5 file"):
        hdf5.File(str(p))
    raw = bytearray(open(tiny_h5, "rb").read())
    raw[8] = 2                                        # superblock version of libver='latest' files
    p.write_bytes(bytes(raw))
    with pytest.raises(hdf5.HDF5FormatError, match="superblock version 2"):
        hdf5.File(str(p))
    p.write_bytes(bytes(open(tiny_h5, "rb").read()[:4096]))   # truncated file: addresses beyond EOF
    with pytest.raises((hdf5.HDF5FormatError, ValueError, IndexError)):
        hdf5.read_keras_weights(str(p))
    # corrupted metadata never hangs or crashes the interpreter: a clean exception, or (for flips that land in
    # unused bytes / dataset payload) a normal read
    good = open(tiny_h5, "rb").read()
    rng = np.random.default_rng(5)
    outcomes = {"ok": 0, "rejected": 0}
    for trial in range(60):
        raw = bytearray(good)
        for pos in rng.integers(0, min(len(raw), 200_000), 8):
            raw[pos] ^= 1 << int(rng.integers(0, 8))
        p.write_bytes(bytes(raw))
        try:
            hdf5.read_keras_weights(str(p))
            outcomes["ok"] += 1
        except (hdf5.HDF5FormatError, ValueError, IndexError, KeyError, UnicodeDecodeError, OverflowError, struct.error, RecursionError, MemoryError):
            outcomes["rejected"] += 1
    assert outcomes["ok"] + outcomes["rejected"] == 60


def _keras_checkpoint(conv, models):
    """The Keras-side dict a Matterport checkpoint would hold for these .mrcw tensors (fp32)."""
    k = {}
    for kind, (_, tensors) in models.items():
        for name, a in tensors.items():
            k[conv.keras_name(name)] = np.ascontiguousarray(conv.to_keras_layout(name, a.astype(np.float32)))
    return k


def test_convert_tensors_roundtrip_and_errors(pkg, weights_mod, conv):
    cfg = pkg.ModelConfig(architecture="resnet50", input_image_shape=(128, 128, 3), num_classes=4)
    models = weights_mod.synthetic_models(cfg, seed=11)
    keras = _keras_checkpoint(conv, models)
    # Keras layouts: HWIO convs, (I,O) dense, HWOI transposed conv, moving_* BatchNorm names
    assert keras["conv1/kernel"].shape == (7, 7, 3, 64) and keras["mrcnn_class_logits/kernel"].shape == (1024, 4)
    assert keras["mrcnn_mask_deconv/kernel"].shape == (2, 2, 256, 256) and "bn_conv1/moving_variance" in keras
    keras["mrcnn_class_conv1/extra_training_only"] = np.zeros(3, np.float32)
    out, unused = conv.convert_tensors(keras, cfg)
    assert unused == ["mrcnn_class_conv1/extra_training_only"]
    for kind in ("MaskRCNN", "Classifier", "Mask"):
        meta, tensors = out[kind]
        assert meta == models[kind][0]
        assert sorted(tensors) == sorted(models[kind][1])
        for n, a in tensors.items():
            assert a.dtype == np.float16 and a.flags["C_CONTIGUOUS"]
            np.testing.assert_array_equal(a, models[kind][1][n], err_msg=n)
    out32, _ = conv.convert_tensors(keras, cfg, weights_dtype="f32")
    assert out32["Mask"][1]["mrcnn_mask/kernel"].dtype == np.float32
    # a non-square kernel proves the axis order (not just a shape-preserving shuffle)
    a = np.arange(2 * 3 * 5 * 7, dtype=np.float32).reshape(2, 3, 5, 7)         # (kh,kw,I,O)
    b = conv.from_keras_layout("x/kernel", a)
    assert b.shape == (7, 5, 2, 3) and b[6, 4, 1, 2] == a[1, 2, 4, 6]
    np.testing.assert_array_equal(conv.to_keras_layout("x/kernel", b), a)
    # errors name the tensor
    bad = dict(keras); del bad["res3d_branch2b/kernel"]
    with pytest.raises(conv.ConversionError, match="res3d_branch2b/kernel"):
        conv.convert_tensors(bad, cfg)
    with pytest.raises(conv.ConversionError, match="mrcnn_class_logits/kernel.*implies"):
        conv.convert_tensors(keras, pkg.ModelConfig(architecture="resnet50", num_classes=81))
    with pytest.raises(conv.ConversionError, match="res4g_branch2a"):
        conv.convert_tensors(keras, pkg.ModelConfig(architecture="resnet101", num_classes=4))
    bad = dict(keras); bad["fpn_p3/bias"] = np.full(256, 1e6, np.float32)
    with pytest.raises(conv.ConversionError, match="fpn_p3/bias.*overflows fp16"):
        conv.convert_tensors(bad, cfg)
    bad["fpn_p3/bias"][0] = np.nan
    with pytest.raises(conv.ConversionError, match="non-finite"):
        conv.convert_tensors(bad, cfg)


_WRITER = r'''
import sys, numpy as np, h5py
src, dst, libver = sys.argv[1:4]
z = np.load(src)
groups = {}
for key in z.files:                      # "<layer>/<weight>"
    layer = key.split("/")[0]
    top = "rpn_model" if layer.startswith("rpn_") else layer
    groups.setdefault(top, []).append(key)
with h5py.File(dst, "w", libver=libver) as f:
    f.attrs["layer_names"] = [n.encode() for n in groups]
    f.attrs["backend"] = b"tensorflow"; f.attrs["keras_version"] = b"2.1.6"
    for top, keys in groups.items():
        g = f.create_group(top)
        g.attrs["weight_names"] = [(k + ":0").encode() for k in keys]
        for k in keys:
            d = g.create_dataset(k + ":0", z[k].shape, dtype=z[k].dtype); d[:] = z[k]
'''


def _have_conda_h5py():
    try:
        return subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True, timeout=120).returncode == 0
    except Exception:
        return False


@pytest.mark.skipif(not _have_conda_h5py(), reason="needs the image's conda python with h5py to write a checkpoint")
def test_convert_command_end_to_end(pkg, weights_mod, anchors_mod, conv, hdf5, tmp_path):
    cfg_d = {"architecture": "resnet50", "input_image_shape": [256, 256, 3], "num_classes": 3,
             "pre_nms_max_proposals": 600, "max_proposals": 100}
    cfg = pkg.ModelConfig.from_dict(cfg_d)
    models = weights_mod.synthetic_models(cfg, seed=5)
    model_dir = tmp_path / "model"; model_dir.mkdir()
    (model_dir / "config.json").write_text(json.dumps(cfg_d))
    np.savez(tmp_path / "k.npz", **_keras_checkpoint(conv, models))
    script = tmp_path / "w.py"; script.write_text(_WRITER)
    subprocess.run([CONDA_PY, str(script), str(tmp_path / "k.npz"), str(model_dir / "weights.h5"), "earliest"], check=True, timeout=600)
    paths = conv.convert(str(model_dir / "config.json"), str(model_dir / "weights.h5"), str(tmp_path / "products"), verbose=False)
    assert sorted(os.path.basename(p) for p in paths.values()) == ["Classifier.mrcw", "Mask.mrcw", "MaskRCNN.mrcw", "anchors.bin"]
    for kind in ("MaskRCNN", "Classifier", "Mask"):
        meta, tensors = weights_mod.read_mrcw(paths[kind])
        assert meta == models[kind][0]
        assert meta["ProposalLayer.preNMSMaxProposals"] == 600 if kind == "MaskRCNN" else True
        for n, a in models[kind][1].items():
            np.testing.assert_array_equal(tensors[n], a, err_msg=n)
    anchors = np.fromfile(paths["anchors"], dtype="<f4").reshape(-1, 4)
    np.testing.assert_array_equal(anchors, anchors_mod.generate_anchors(cfg))
    # the CLI spelling of ConvertCommand.swift:9-11
    assert conv.main(["--config", str(model_dir / "config.json"), "--weights", str(model_dir / "weights.h5"),
                      "--output_dir", str(tmp_path / "p2")]) == 0
    assert open(tmp_path / "p2" / "Mask.mrcw", "rb").read() == open(paths["Mask"], "rb").read()
    # a libver='latest' checkpoint is refused with a message, never half-read
    subprocess.run([CONDA_PY, str(script), str(tmp_path / "k.npz"), str(tmp_path / "latest.h5"), "latest"], check=True, timeout=600)
    with pytest.raises(hdf5.HDF5FormatError, match="not supported"):
        conv.convert(str(model_dir / "config.json"), str(tmp_path / "latest.h5"), str(tmp_path / "p3"), verbose=False)


def test_stored_split_exponents_ride_in_the_artefact_metadata(weights_mod, conv, tmp_path):
    """`convert --calibrate` (round 5): the exponent vector of the scale-aware split goes into MaskRCNN.mrcw as `split_exp.<group>` integer
    metadata — what mrcnn_model_load applies (engine.hip) — leaving every tensor and every other key untouched; storing again replaces the old
    vector instead of accumulating keys; the image loader of the flag takes uint8 (N,H,W,3) .npy files and refuses anything else."""
    d = tmp_path / "products"
    d.mkdir()
    t = {"conv1/kernel": np.arange(24, dtype="<f2").reshape(2, 3, 2, 2), "bn_conv1/gamma": np.ones(2, "<f2")}
    meta = {"kind": "MaskRCNN", "num_classes": 81, "bn_eps": 1e-3}
    weights_mod.write_mrcw(str(d / "MaskRCNN.mrcw"), meta, t)
    conv.store_split_exponents(str(d), ["C1", "res2a_branch2a", "P2"], [5, -3, 8])
    m2, t2 = weights_mod.read_mrcw(str(d / "MaskRCNN.mrcw"))
    assert m2["split_exp.C1"] == 5 and m2["split_exp.res2a_branch2a"] == -3 and m2["split_exp.P2"] == 8
    assert m2["kind"] == "MaskRCNN" and m2["num_classes"] == 81 and m2["bn_eps"] == pytest.approx(1e-3)
    assert set(t2) == set(t) and all(np.array_equal(t2[k], t[k]) and t2[k].dtype == t[k].dtype for k in t)
    conv.store_split_exponents(str(d), ["C1"], [7])
    m3, _ = weights_mod.read_mrcw(str(d / "MaskRCNN.mrcw"))
    assert m3["split_exp.C1"] == 7 and "split_exp.P2" not in m3 and sum(k.startswith("split_exp.") for k in m3) == 1
    imgs = np.random.default_rng(0).integers(0, 256, (3, 8, 8, 3), dtype=np.uint8)
    np.save(str(tmp_path / "cal.npy"), imgs)
    got = conv.load_calibration_images(str(tmp_path / "cal.npy"), 8, 8, limit=2)
    assert got.shape == (2, 8, 8, 3) and np.array_equal(got, imgs[:2])
    with pytest.raises(conv.ConversionError):
        conv.load_calibration_images(str(tmp_path / "cal.npy"), 16, 16)
    np.save(str(tmp_path / "bad.npy"), imgs.astype(np.float32))
    with pytest.raises(conv.ConversionError):
        conv.load_calibration_images(str(tmp_path / "bad.npy"), 8, 8)
    with pytest.raises(conv.ConversionError):
        conv.load_calibration_images(str(tmp_path / "nothing-here"), 8, 8)
