import ctypes as C, importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
L = importlib.import_module("mask-rcnn-coreml_amd._lib"); lib = L.lib()
from test_gpu_conv_kernels import conv
Ci = Co = 256
w = np.zeros((Co, 1, 1, Ci), np.float32)
for n in range(Co): w[n, 0, 0, n] = 1.0
B, H, W = 1, 32, 32   # M = 1024 = 4 tiles
x = np.zeros((B, H, W, Ci), np.float32); x[..., :] = np.arange(Ci)[None, None, None, :]
for k in (b"conv_pp_min_tiles", b"conv_pp_min_kt"): L.check(lib.mrcnn_debug_set(k, 1))
L.check(lib.mrcnn_debug_set(b"conv_pp", 1))
y = conv(x, w, 1, 1, None, None, None, 0, "f16")
print("channel map of pixel 0 (should be 0..255):"); print(y[0, 0, 0].astype(int).tolist())
bad = np.argwhere(y != x)
print("mismatches", len(bad), "of", y.size)
x2 = np.zeros_like(x); x2[0] = (np.arange(H * W).reshape(H, W, 1) % 1024)
y2 = conv(x2, w, 1, 1, None, None, None, 0, "f16")
print("pixel map, channel 0, first 70 pixels:", y2[0].reshape(-1, Ci)[:70, 0].astype(int).tolist())
print("pixel mismatches:", int((y2 != x2).sum()))
