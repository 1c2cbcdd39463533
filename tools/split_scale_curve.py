#!/usr/bin/env python
"""Accuracy of one convolution per compute mode when the SAME activation tensor is scaled by 2^e (e = 0 ... -20), against an
fp64 torch convolution: max |err| / max |ref|.  The split modes carry an activation exactly for |a| >= 0.5 and to 2^-24
absolute below (fp16 subnormal step of the last part), so they are not scale-invariant the way fp32 is; this prints the
curve next to the documented bound (tests/test_gpu_conv_kernels.py asserts it).  usage: split_scale_curve.py > profiles/rNN_split_scale_curve.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_conv_kernels as t  # noqa: E402

print("conv 3x3 256->128 on a post-ReLU tensor (values O(1-10)) scaled by 2^e; max|err| / max|ref| vs fp64")
print(f"{'e':>4s} " + " ".join(f"{m:>12s}" for m in ("f32", "f32x3", "f32s")) + f" {'bound(x3)':>12s}")
cur = {m: t.split_scale_curve(m) for m in ("f32", "f32x3", "f32s")}
for i, e in enumerate(t.SCALES):
    print(f"{e:4d} " + " ".join(f"{cur[m][i][1]:12.3e}" for m in ("f32", "f32x3", "f32s")) + f" {cur['f32x3'][i][2]:12.3e}")

print()
print("the same tensor with the engine's calibration (stored as 2^p * value, max |a| * 2^p in [2^11, 2^12); the consumer's scale carries 2^-p):")
print(f"{'e':>4s} {'p':>4s} " + " ".join(f"{m:>12s}" for m in ("f32x3", "f32s")))
cal = {m: t.split_scale_curve_calibrated(m) for m in ("f32x3", "f32s")}
for i, e in enumerate(t.SCALES):
    print(f"{e:4d} {cal['f32x3'][i][2]:4d} " + " ".join(f"{cal[m][i][1]:12.3e}" for m in ("f32x3", "f32s")))
