"""The fused identity bottleneck of the fp16 mode (kernels_bneck.hip) against the three launches it replaces.

Reference layers: res<stage><block>_branch2a / 2b / 2c + BatchNorm + ReLU + shortcut of every non-first ResNet block
(Sources/maskrcnn/Python/Conversion/task.py:69-92).  The fused launch keeps both mid tensors on chip; it must agree with
conv_forward x 3 BIT FOR BIT (same K order, same fp16 roundings, same epilogue arithmetic) — which form runs depends on the
layer's geometry and a debug knob, per-image results must not — and with an fp64 evaluation of the same fp16 operands within
the fp16 output step.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
L = importlib.import_module("mask-rcnn-coreml_amd._lib")


def bneck(x, w1, w2, w3, bn, fused, iters=0):
    B, H, W, C4 = x.shape
    C = C4 // 4
    out = np.empty((B, H, W, C4), np.float32)
    ms = np.zeros(1, np.float32)
    keep = [np.ascontiguousarray(a, np.float32) for a in (x, w1, w2, w3) + tuple(bn)]
    L.check(L.lib().mrcnn_bottleneck_nhwc(keep[0].ctypes.data, B, H, W, C, *[k.ctypes.data for k in keep[1:]], int(fused), iters,
                                           out.ctypes.data, ms.ctypes.data))
    return out, float(ms[0])


def make(C, B, H, W, seed=0):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((B, H, W, 4 * C)), 0).astype(np.float32)          # a post-ReLU block input
    w1 = (rng.standard_normal((C, 4 * C)) * np.sqrt(2.0 / (4 * C))).astype(np.float32)
    w2 = (rng.standard_normal((C, 3, 3, C)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    w3 = (rng.standard_normal((4 * C, C)) * np.sqrt(2.0 / C)).astype(np.float32)
    bn = []
    for n in (C, C, 4 * C):
        bn.append((1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32))
        bn.append((0.1 * rng.standard_normal(n)).astype(np.float32))
    return x, w1, w2, w3, bn


def ref64(x, w1, w2, w3, bn):
    """fp64 evaluation with the fp16 roundings of the fp16 mode (operands, both mid tensors, the output)."""
    import torch
    import torch.nn.functional as F
    h = lambda a: torch.from_numpy(np.asarray(a, np.float32).astype(np.float16).astype(np.float64))
    r16 = lambda t: torch.from_numpy(t.numpy().astype(np.float32).astype(np.float16).astype(np.float64))
    v = lambda a: torch.from_numpy(np.asarray(a, np.float64))[None, :, None, None]
    xx = h(x).permute(0, 3, 1, 2)
    C = w1.shape[0]
    t1 = r16(F.relu(F.conv2d(xx, h(w1).reshape(C, 4 * C, 1, 1)) * v(bn[0]) + v(bn[1])))
    t2 = r16(F.relu(F.conv2d(t1, h(w2).permute(0, 3, 1, 2), padding=1) * v(bn[2]) + v(bn[3])))
    y = F.relu(F.conv2d(t2, h(w3).reshape(4 * C, C, 1, 1)) * v(bn[4]) + v(bn[5]) + xx)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


SHAPES = [(256, 1, 8, 16), (256, 2, 16, 32), (256, 1, 64, 64), (128, 1, 16, 16), (128, 2, 32, 48), (64, 1, 16, 16), (64, 1, 48, 32), (64, 2, 32, 32)]


@pytest.mark.parametrize("C,B,H,W", SHAPES)
def test_fused_bottleneck_equals_the_three_launches_bitwise(C, B, H, W):
    x, w1, w2, w3, bn = make(C, B, H, W, seed=C + H)
    fused, _ = bneck(x, w1, w2, w3, bn, True)
    three, _ = bneck(x, w1, w2, w3, bn, False)
    if C == 256:          # the form with every operand through the LDS ring (the default streams filter fragments into registers)
        ring, _ = bneck(x, w1, w2, w3, bn, 2)
        assert np.array_equal(ring.view(np.uint32), three.view(np.uint32))
    assert np.isfinite(fused).all()
    assert fused.shape == three.shape
    nz = np.flatnonzero(fused.view(np.uint32) != three.view(np.uint32))
    assert nz.size == 0, f"{nz.size} of {fused.size} outputs differ, first at {np.unravel_index(nz[0], fused.shape)}: {fused.flat[nz[0]]} vs {three.flat[nz[0]]}"


@pytest.mark.parametrize("C,B,H,W", [(256, 1, 16, 32), (128, 1, 32, 32), (64, 1, 32, 32)])
def test_fused_bottleneck_against_an_fp64_evaluation(C, B, H, W):
    x, w1, w2, w3, bn = make(C, B, H, W, seed=7)
    fused, _ = bneck(x, w1, w2, w3, bn, True)
    want = ref64(x, w1, w2, w3, bn)
    # one fp16 rounding of the output (2^-11 relative) + the flips of the mid tensors' roundings under fp32 summation noise
    tol = 2e-3 * np.maximum(1.0, np.abs(want))
    bad = np.abs(fused - want) > tol
    assert bad.mean() < 1e-3, f"{bad.sum()} of {bad.size} outputs beyond 2e-3"
    assert np.abs(fused - want).max() < 0.05 * max(1.0, np.abs(want).max())


def test_images_of_a_batch_do_not_depend_on_the_batch():
    C, H, W = 256, 16, 32
    x, w1, w2, w3, bn = make(C, 3, H, W, seed=3)
    whole, _ = bneck(x, w1, w2, w3, bn, True)
    for i in range(3):
        one, _ = bneck(x[i:i + 1], w1, w2, w3, bn, True)
        assert np.array_equal(one[0].view(np.uint32), whole[i].view(np.uint32))


def test_unsupported_geometry_is_refused_loudly():
    x, w1, w2, w3, bn = make(64, 1, 24, 24)          # W % 16 != 0
    with pytest.raises(L.MrcnnError):
        bneck(x, w1, w2, w3, bn, True)
    three, _ = bneck(x, w1, w2, w3, bn, False)        # the three launches take any geometry
    assert np.isfinite(three).all()


def bneck_first(x, w1, w2, w3, ws, bn8, fused, iters=0):
    import ctypes as C
    B, H, W, Cc = x.shape
    out = np.empty((B, H, W, 4 * Cc), np.float32)
    ms = np.zeros(1, np.float32)
    keep = [np.ascontiguousarray(a, np.float32) for a in (x, w1, w2, w3, ws) + tuple(bn8)]
    arr = (C.c_void_p * 8)(*[k.ctypes.data for k in keep[5:]])
    L.check(L.lib().mrcnn_bottleneck_first_nhwc(keep[0].ctypes.data, B, H, W, Cc, *[k.ctypes.data for k in keep[1:5]], arr, int(fused), iters,
                                                 out.ctypes.data, ms.ctypes.data))
    return out, float(ms[0])


def make_first(C, B, H, W, seed=0):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((B, H, W, C)), 0).astype(np.float32)               # the max-pooled stem output
    w1 = (rng.standard_normal((C, C)) * np.sqrt(2.0 / C)).astype(np.float32)
    w2 = (rng.standard_normal((C, 3, 3, C)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    w3 = (rng.standard_normal((4 * C, C)) * np.sqrt(2.0 / C)).astype(np.float32)
    ws = (rng.standard_normal((4 * C, C)) * np.sqrt(2.0 / C)).astype(np.float32)
    bn = []
    for n in (C, C, 4 * C, 4 * C):
        bn.append((1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32))
        bn.append((0.1 * rng.standard_normal(n)).astype(np.float32))
    return x, w1, w2, w3, ws, bn


@pytest.mark.parametrize("B,H,W", [(1, 16, 16), (2, 32, 48), (1, 64, 64), (3, 16, 32)])
def test_fused_stage_entry_block_equals_the_four_launches_bitwise(B, H, W):
    """res2a of the fp16 mode (stride 1, 64 channels in, shortcut = the 1x1 convolution branch1 of the input): one launch against
    branch2a, branch1, branch2b, branch2c as four — the shortcut is rounded to fp16 where its tensor would have been."""
    x, w1, w2, w3, ws, bn = make_first(64, B, H, W, seed=H + W)
    fused, _ = bneck_first(x, w1, w2, w3, ws, bn, True)
    four, _ = bneck_first(x, w1, w2, w3, ws, bn, False)
    assert np.isfinite(fused).all()
    nz = np.flatnonzero(fused.view(np.uint32) != four.view(np.uint32))
    assert nz.size == 0, f"{nz.size} of {fused.size} outputs differ, first at {np.unravel_index(nz[0], fused.shape)}: {fused.flat[nz[0]]} vs {four.flat[nz[0]]}"
    # and against fp64 with the fp16 roundings of the mode
    import torch
    import torch.nn.functional as F
    h = lambda a: torch.from_numpy(np.asarray(a, np.float32).astype(np.float16).astype(np.float64))
    r16 = lambda t: torch.from_numpy(t.numpy().astype(np.float32).astype(np.float16).astype(np.float64))
    v = lambda a: torch.from_numpy(np.asarray(a, np.float64))[None, :, None, None]
    xx = h(x).permute(0, 3, 1, 2)
    t1 = r16(F.relu(F.conv2d(xx, h(w1).reshape(64, 64, 1, 1)) * v(bn[0]) + v(bn[1])))
    t2 = r16(F.relu(F.conv2d(t1, h(w2).permute(0, 3, 1, 2), padding=1) * v(bn[2]) + v(bn[3])))
    sc = r16(F.conv2d(xx, h(ws).reshape(256, 64, 1, 1)) * v(bn[6]) + v(bn[7]))
    y = F.relu(F.conv2d(t2, h(w3).reshape(256, 64, 1, 1)) * v(bn[4]) + v(bn[5]) + sc).permute(0, 2, 3, 1).numpy()
    bad = np.abs(fused - y) > 2e-3 * np.maximum(1.0, np.abs(y))
    assert bad.mean() < 1e-3


# ---- round 6: the consecutive identity blocks of a C = 256 stage as ONE launch (STAGE form) ----------------------------------------------
def bneck_stage(x, w1, w2, w3, bn6, form, iters=0):
    import ctypes as C
    B, H, W, C4 = x.shape
    n = w1.shape[0]
    out = np.empty((B, H, W, C4), np.float32)
    ms = np.zeros(1, np.float32)
    flag = np.zeros(1, np.int32)
    keep = [np.ascontiguousarray(a, np.float32) for a in (x, w1, w2, w3) + tuple(bn6)]
    arr = (C.c_void_p * 6)(*[k.ctypes.data for k in keep[4:]])
    L.check(L.lib().mrcnn_bottleneck_stage_nhwc(keep[0].ctypes.data, B, H, W, n, *[k.ctypes.data for k in keep[1:4]], arr, int(form), iters,
                                                 out.ctypes.data, ms.ctypes.data, flag.ctypes.data))
    return out, float(ms[0]), int(flag[0])


def make_stage(n, B, H, W, seed=0):
    """n blocks whose residual chain stays inside the fp16 range (a damped branch: BatchNorm scale 0.25 on branch2c)."""
    rng = np.random.default_rng(seed)
    C = 256
    x = np.maximum(rng.standard_normal((B, H, W, 4 * C)), 0).astype(np.float32)
    w1 = (rng.standard_normal((n, C, 4 * C)) * np.sqrt(2.0 / (4 * C))).astype(np.float32)
    w2 = (rng.standard_normal((n, C, 3, 3, C)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    w3 = (rng.standard_normal((n, 4 * C, C)) * np.sqrt(2.0 / C)).astype(np.float32)
    bn = []
    for width, gain in ((C, 1.0), (C, 1.0), (4 * C, 0.25)):
        bn.append((gain * (1.0 + 0.1 * rng.standard_normal((n, width)))).astype(np.float32))
        bn.append((0.05 * rng.standard_normal((n, width))).astype(np.float32))
    return x, w1, w2, w3, bn


@pytest.mark.parametrize("n,B,H,W", [(2, 1, 8, 16), (3, 1, 16, 32), (5, 2, 64, 64), (22, 8, 64, 64), (4, 9, 64, 64), (3, 1, 24, 48)])
def test_whole_stage_launch_equals_one_launch_per_block_bitwise(n, B, H, W):
    """res4b..res4w of the fp16 mode (Conversion/task.py:69-92) as ONE launch whose tiles wait for their neighbours' previous block
    (per-tile counters, device-scope stores / loads) against 22 fused launches: same tile arithmetic, so every bit agrees; (4, 9, 64, 64):
    288 tiles on 256 blocks (a block owns more than one tile); (3, 1, 24, 48): a 3 x 3 tile grid (a centre tile with all eight neighbours)."""
    x, w1, w2, w3, bn = make_stage(n, B, H, W, seed=n + H)
    one, _, flag1 = bneck_stage(x, w1, w2, w3, bn, 1)
    per, _, flag0 = bneck_stage(x, w1, w2, w3, bn, 0)
    assert flag1 == 0 and flag0 == 0, f"flag words {flag1} / {flag0} (bit 0: fp16 range, bit 1: no progress)"
    assert np.isfinite(one).all() and one.max() > 0
    nz = np.flatnonzero(one.view(np.uint32) != per.view(np.uint32))
    assert nz.size == 0, f"{nz.size} of {one.size} outputs differ, first at {np.unravel_index(nz[0], one.shape)}: {one.flat[nz[0]]} vs {per.flat[nz[0]]}"


def test_whole_stage_launch_is_repeatable_and_chains_the_single_block_form():
    """200 launches of a 6-block stage give the same bits (a missed neighbour wait or a stale line would show as a changed halo row),
    and the result equals chaining the single-block entry the other tests pin against the three launches and fp64."""
    n, B, H, W = 6, 2, 64, 64
    x, w1, w2, w3, bn = make_stage(n, B, H, W, seed=11)
    ref, _, _ = bneck_stage(x, w1, w2, w3, bn, 1)
    cur = x
    for l in range(n):
        cur, _ = bneck(cur, w1[l], w2[l], w3[l], [b[l] for b in bn], True)
    assert np.array_equal(cur.view(np.uint32), ref.view(np.uint32))
    for _ in range(4):
        again, _, flag = bneck_stage(x, w1, w2, w3, bn, 1, iters=50)
        assert flag == 0 and np.array_equal(again.view(np.uint32), ref.view(np.uint32))
