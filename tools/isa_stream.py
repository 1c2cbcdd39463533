#!/usr/bin/env python
"""Instruction-order digest of the conv kernels' K loop (M = MFMA, R = ds_read, G = global→LDS DMA, W = s_waitcnt):
   tools/isa_stream.py   — compiles kernels_conv.hip to gfx950 assembly and prints the stream after the first barrier."""
import os as _os; _os.environ.setdefault("MRCNN_TEST_KNOBS", "1")      # arm the test / measurement knobs (csrc/common.h) before the library loads
import os
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mask-rcnn-coreml_amd", "csrc", "kernels_conv.hip")
out = os.path.join(tempfile.mkdtemp(), "kc.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-inline-asm",
                "-S", "--cuda-device-only", "-o", out, src] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read().splitlines()
for i, line in enumerate(txt):
    if not line.startswith("_ZN5mrcnn16k_conv_mfma_glds"):
        continue
    j = next(k for k in range(i, len(txt)) if txt[k].strip() == "s_endpgm")
    ops = []
    for l in txt[i:j]:
        t = l.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        o = t.split()[0]
        if o.startswith("v_mfma"): ops.append("M")
        elif o.startswith("ds_read"): ops.append("R")
        elif o.startswith("s_waitcnt"): ops.append("W(" + t.split(None, 1)[1].replace(" ", "") + ")")
        elif o == "s_barrier": ops.append("BAR")
        elif o.startswith("global_load_lds"): ops.append("G")
        elif o.startswith("scratch_"): ops.append("SCRATCH!")
    st = " ".join(ops)
    k = st.find("BAR")
    print(line.split(":")[0][28:62], st[k:k + 300], "\n")
