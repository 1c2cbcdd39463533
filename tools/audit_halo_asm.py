#!/usr/bin/env python3
"""Audit of the halo kernel's hand-counted `s_waitcnt vmcnt(N)` regions.

The kernel (mask-rcnn-coreml_amd/csrc/kernels_conv_halo.hip) counts its own VMEM instructions between a load and the wait that
retires it.  A register spill the compiler places INSIDE such a region is a scratch (VMEM) access the count does not know about and
would make the wait return early.  This tool compiles the file to gfx950 assembly (no GPU needed) and checks, per instantiation,
that the steady-state loop (the innermost backward branch spanning the 36 unrolled steps' MFMAs) contains no `scratch_` access.

    python tools/audit_halo_asm.py            # prints one line per instantiation, exit 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mask-rcnn-coreml_amd", "csrc", "kernels_conv_halo.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def assembly():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "halo.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-inline-asm",
                        "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", SRC, "-o", out], check=True)
        with open(out) as f:
            return f.read().split("\n")


def audit(lines):
    """-> [(instantiation, vgprs, spilled vgprs, mfmas in the steady-state loop, scratch accesses inside it)]"""
    rows = []
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN5mrcnn11k_conv_halo\w+:", l)]
    for i in starts:
        end = next(j for j in range(i, len(lines)) if ".end_amdhsa_kernel" in lines[j])
        body = lines[i:end]
        name = re.search(r"k_conv_haloI(\w+?)EEv", lines[i]).group(1)
        name = "<" + ",".join(p[1:] if p[0] == "L" else p for p in re.findall(r"L[ib]\d+", name)) + ">"
        scratch = [k for k, l in enumerate(body) if "scratch_" in l]
        mfma = [k for k, l in enumerate(body) if "v_mfma" in l]
        labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        # every backward branch spanning >= 72 MFMAs is a candidate; the hand-counted loop is the 36-step loop of the 3x3 phase — the
        # innermost candidate with the MOST MFMAs (the fused-tail instantiations also hold the 16-step 1x1 loop, whose loads the
        # compiler counts itself: spills there cost time, not correctness, and are reported separately)
        cands = []
        for k, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not m or labels.get(m.group(1), k) >= k:
                continue
            t = labels[m.group(1)]
            nm = sum(1 for x in mfma if t <= x <= k)
            if nm >= 2 * 36:
                cands.append((t, k, nm))
        inner = [c for c in cands if not any(o is not c and c[0] <= o[0] and o[1] <= c[1] for o in cands)]
        if not inner:   # no loop found: report it as a violation rather than pass silently
            rows.append((name, len(scratch), 0, -1))
            continue
        best = max(inner, key=lambda c: c[2])
        inside = sum(1 for x in scratch if best[0] <= x <= best[1])
        rows.append((name, len(scratch), best[2], inside))
    return rows


def main():
    bad = 0
    for row in audit(assembly()):
        name, nscratch, nm, inside = row[0], row[1], row[2], row[3]
        print(f"k_conv_halo{name:28s} scratch accesses {nscratch!s:>4s}   steady-state loop: {nm} MFMAs, {inside} scratch accesses inside")
        bad += inside != 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
