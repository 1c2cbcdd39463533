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


# ---- round 6: the fp16 mode's hand-scheduled kernels ----------------------------------------------------------------
# k_bneck_h<*> (kernels_bneck.hip) and k_conv3x3_h<*> (kernels_conv3x3_h.hip) issue LDS-DMAs from inline asm and retire them with hand-placed
# waits; k_conv_pp (kernels_conv_pp.hip) counts its waits like the halo kernel.  What can go wrong differs by kernel:
#   * a COUNTED wait (`vmcnt(N)`, N > 0, written by hand) is wrong as soon as the compiler puts a VMEM access of its own — a spill — inside
#     the counted region: k_conv_pp's steady-state loop must hold no scratch access (as the halo kernel's);
#   * k_bneck_h / k_conv3x3_h only ever wait with `vmcnt(0)` by hand (checked on the SOURCE: no other hand-written count), which no spill can
#     make return early — uncounted VMEM accesses issued by inline asm make the compiler's own counted waits more conservative, never less
#     (vmcnt retires in order) — so there a spill inside an MFMA loop is a performance defect, not a correctness one: counted and reported,
#     and required to be zero for the loops that carry the fp16 mode's time.
FP16_KERNELS = [("kernels_bneck.hip", r"k_bneck_h", 16), ("kernels_conv3x3_h.hip", r"k_conv3x3_h", 16), ("kernels_conv_pp.hip", r"k_conv_pp", 16)]


def assembly_of(src_name):
    src = os.path.join(ROOT, "mask-rcnn-coreml_amd", "csrc", src_name)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-inline-asm",
                        "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", src, "-o", out], check=True)
        with open(out) as f:
            return f.read().split("\n")


def hand_written_waits(src_name):
    """The `vmcnt(...)` arguments that appear in the SOURCE's inline asm (macros included), as strings."""
    with open(os.path.join(ROOT, "mask-rcnn-coreml_amd", "csrc", src_name)) as f:
        text = f.read()
    return re.findall(r'asm volatile\("[^"]*vmcnt\(([^)]*)\)', text)


def audit_loops(lines, kernel, min_mfma):
    """-> [(instantiation, vgprs, scratch bytes per lane, scratch accesses, sgpr spill lanes, [(mfmas, scratch inside)] per innermost MFMA loop)]"""
    rows = []
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN5mrcnn\d+" + kernel + r"\w*:", l)]
    for i in starts:
        end = next(j for j in range(i, len(lines)) if ".end_amdhsa_kernel" in lines[j])
        body = lines[i:end]
        sym = lines[i].split(":")[0]
        name = "<" + ",".join(p[1:] if p[0] == "L" else p for p in re.findall(r"L[ib]\d+", sym)) + ">"
        meta = "\n".join(body)
        vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta)
        sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta)
        scratch = [k for k, l in enumerate(body) if re.search(r"\bscratch_(load|store)", l)]
        lanes = sum(1 for l in body if "v_writelane_b32" in l)
        mfma = [k for k, l in enumerate(body) if "v_mfma" in l]
        labels = {l.split(":")[0]: k for k, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        cands = []
        for k, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not m or labels.get(m.group(1), k) >= k:
                continue
            t = labels[m.group(1)]
            nm = sum(1 for x in mfma if t <= x <= k)
            if nm >= min_mfma:
                cands.append((t, k, nm))
        inner = [c for c in cands if not any(o is not c and c[0] <= o[0] and o[1] <= c[1] for o in cands)]
        loops = [(c[2], sum(1 for x in scratch if c[0] <= x <= c[1])) for c in inner]
        rows.append((name, int(vg.group(1)) if vg else -1, int(sc.group(1)) if sc else -1, len(scratch), lanes, loops))
    return rows


def audit_fp16():
    """-> {kernel: (hand-written vmcnt arguments in its source, rows of audit_loops)}"""
    return {k: (hand_written_waits(src), audit_loops(assembly_of(src), k, mm)) for src, k, mm in FP16_KERNELS}


def main():
    bad = 0
    for kernel, (waits, rows) in audit_fp16().items():
        print(f"{kernel}: hand-written waits vmcnt({', '.join(sorted(set(waits))) or '-'})")
        for name, vg, sbytes, nscratch, lanes, loops in rows:
            inside = sum(x[1] for x in loops)
            print(f"  {kernel}{name:22s} {vg:3d} VGPRs, scratch {sbytes:3d} B/lane ({nscratch} accesses), {lanes} SGPRs spilled to lanes; "
                  f"MFMA loops (MFMAs, scratch inside): {loops}")
            bad += inside != 0
        bad += any(w.strip() != "0" for w in waits) and kernel != "k_conv_pp"      # only the ping-pong kernel counts its waits by hand
    for row in audit(assembly()):
        name, nscratch, nm, inside = row[0], row[1], row[2], row[3]
        print(f"k_conv_halo{name:28s} scratch accesses {nscratch!s:>4s}   steady-state loop: {nm} MFMAs, {inside} scratch accesses inside")
        bad += inside != 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
