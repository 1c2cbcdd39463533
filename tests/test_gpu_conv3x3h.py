"""The 3x3 stride-1 layers of the fp16 mode on the halo-tile / fragment-streaming kernel (kernels_conv3x3_h.hip; the RPN's shared
layer and the FPN's output layers, Sources/maskrcnn/Python/Conversion/task.py:69-92).

Its K order (64-channel block, tap, 16-wide group) is its own: against the 128-row / ping-pong kernels (tap-major) the results differ
by fp32 summation noise — at most one fp16 step of the stored output — and both sit inside the fp16-operand tolerance of an fp64
evaluation.  Whether a layer runs on it is a property of the layer alone, so per-image results must not depend on the batch; edges
(sizes that are no multiple of the 16 x 16 tile, one-tile images, both output widths) are covered.
"""
import contextlib
import importlib

import numpy as np
import pytest

from test_gpu_conv_kernels import conv, torch_ref

pytestmark = pytest.mark.gpu
L = importlib.import_module("mask-rcnn-coreml_amd._lib")


@pytest.fixture(autouse=True)
def every_eligible_layer_on_the_kernel():
    """By default only the RPN's shared layer with its heads fused runs here ("conv_c3h" 1); these tests put every eligible layer on it."""
    L.check(L.lib().mrcnn_debug_set(b"conv_c3h", 2))
    yield
    L.check(L.lib().mrcnn_debug_set(b"conv_c3h", 1))


@contextlib.contextmanager
def knob(key, value, restore):
    L.check(L.lib().mrcnn_debug_set(key, value))
    try:
        yield
    finally:
        L.check(L.lib().mrcnn_debug_set(key, restore))


def make(B, H, W, Ci, Co, seed, relu_in=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Ci)).astype(np.float32)
    if relu_in:
        x = np.maximum(x, 0)
    w = (rng.standard_normal((Co, 3, 3, Ci)) * np.sqrt(2.0 / (9 * Ci))).astype(np.float32)
    sc = (1.0 + 0.1 * rng.standard_normal(Co)).astype(np.float32)
    sh = (0.1 * rng.standard_normal(Co)).astype(np.float32)
    return x, w, sc, sh


SHAPES = [(1, 16, 16, 256, 256), (2, 32, 48, 256, 512), (1, 24, 40, 256, 256), (3, 14, 14, 256, 256), (1, 64, 64, 128, 256), (2, 20, 36, 64, 512),
          (1, 8, 8, 512, 256), (1, 33, 17, 256, 256)]


@pytest.mark.parametrize("B,H,W,Ci,Co", SHAPES)
@pytest.mark.parametrize("act", [0, 1])
def test_c3h_against_fp64_and_the_tap_major_kernels(B, H, W, Ci, Co, act):
    x, w, sc, sh = make(B, H, W, Ci, Co, seed=H * 7 + W + Ci)
    got = conv(x, w, 3, 1, sc, sh, None, act=act, dtype="f16")
    with knob(b"conv_c3h", 0, 2):
        old = conv(x, w, 3, 1, sc, sh, None, act=act, dtype="f16")
    ref = torch_ref(x, w, 3, 1, sc, sh, None, act, dtype="f16")
    scale = max(1.0, float(np.abs(ref).max()))
    # fp16 operands, fp32 accumulation, ONE rounding of the output to fp16 (2^-11 relative)
    assert np.abs(got - ref).max() <= 1.5e-3 * scale, float(np.abs(got - ref).max())
    assert np.abs(old - ref).max() <= 1.5e-3 * scale
    # against the tap-major kernels: fp32 summation noise only — one step of the fp16 output where that step is larger than the noise
    # itself (outputs near zero carry the absolute noise of their O(1) terms: a few 1e-6 of the layer's range)
    step = np.maximum(np.abs(old) * 2.0 ** -10, 2e-5 * scale)
    assert (np.abs(got - old) <= step).all(), float((np.abs(got - old) / step).max())
    assert (got != old).mean() < 0.25


def test_c3h_images_do_not_depend_on_the_batch():
    x, w, sc, sh = make(3, 40, 56, 256, 512, seed=5)
    whole = conv(x, w, 3, 1, sc, sh, None, act=1, dtype="f16")
    for i in range(3):
        one = conv(x[i:i + 1], w, 3, 1, sc, sh, None, act=1, dtype="f16")
        assert np.array_equal(one[0].view(np.uint32), whole[i].view(np.uint32))


def test_c3h_zero_padding_and_tile_edges_match_the_reference_pixel_by_pixel():
    """Border pixels (padding taps), pixels on tile seams and the clipped last tile of a 37 x 21 image."""
    x, w, sc, sh = make(1, 37, 21, 256, 256, seed=11, relu_in=False)
    got = conv(x, w, 3, 1, sc, sh, None, act=0, dtype="f16")
    ref = torch_ref(x, w, 3, 1, sc, sh, None, 0, dtype="f16")
    err = np.abs(got - ref)
    scale = max(1.0, float(np.abs(ref).max()))
    for ys, xs in ((slice(0, 1), slice(None)), (slice(-1, None), slice(None)), (slice(None), slice(0, 1)), (slice(None), slice(-1, None)),
                   (slice(15, 17), slice(None)), (slice(None), slice(15, 17)), (slice(32, None), slice(16, None))):
        assert err[0, ys, xs].max() <= 1.5e-3 * scale
