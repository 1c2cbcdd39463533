import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The bit-identity tests switch between forms of the kernels through mrcnn_debug_set: those knobs (and the MRCNN_* overrides of their
# defaults) exist only in a process started with MRCNN_TEST_KNOBS=1 (csrc/common.h) — set before the library is loaded.  A production
# host never sets it: tests/test_host.py::test_the_knobs_are_out_of_a_production_hosts_reach pins that.
os.environ.setdefault("MRCNN_TEST_KNOBS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("mask-rcnn-coreml_amd")


@pytest.fixture(scope="session")
def weights_mod():
    return importlib.import_module("mask-rcnn-coreml_amd.weights")


@pytest.fixture(scope="session")
def anchors_mod():
    return importlib.import_module("mask-rcnn-coreml_amd.anchors")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


def make_model_dir(tmp_path_factory, pkg, weights_mod, name, **cfg_kwargs):
    cfg = pkg.ModelConfig(**cfg_kwargs)
    d = str(tmp_path_factory.mktemp(name))
    weights_mod.save_synthetic_models(d, cfg, seed=0)
    return d, cfg


@pytest.fixture(scope="session")
def small_model(tmp_path_factory, pkg, weights_mod):
    """ResNet-50, 128×128, 20 classes: every conv shape family at a size the CPU oracle runs in ~1 s."""
    return make_model_dir(tmp_path_factory, pkg, weights_mod, "small", architecture="resnet50",
                          input_image_shape=(128, 128, 3), num_classes=21, pre_nms_max_proposals=300,
                          max_proposals=64, max_detections=16)


def rand_images(b, h, w, seed=1):
    return np.random.default_rng(seed).integers(0, 256, (b, h, w, 3), dtype=np.uint8)
