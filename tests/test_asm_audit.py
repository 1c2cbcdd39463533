"""The halo kernel counts its own VMEM instructions between a load and the `s_waitcnt vmcnt(N)` that retires it; a register spill the
compiler put inside such a region would be an uncounted VMEM access.  hipcc cross-compiles here, so the audit runs without a GPU."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_scratch_access_inside_the_hand_counted_wait_regions():
    spec = importlib.util.spec_from_file_location("audit_halo_asm", os.path.join(ROOT, "tools", "audit_halo_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.audit(mod.assembly())
    assert len(rows) >= 8, rows                       # every instantiation halo_launch can pick
    for name, _, mfmas, inside in rows:
        assert mfmas >= 72, (name, mfmas)             # the 36-step loop was found
        assert inside == 0, (name, inside)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_fp16_kernels_keep_their_mfma_loops_free_of_spills_and_count_no_wait_by_hand():
    """Round 6: k_bneck_h<*> (incl. the whole-stage form), k_conv3x3_h<*> and k_conv_pp<*>.  The bottleneck and 3x3 kernels
    retire their inline-asm LDS-DMAs with `vmcnt(0)` only — a wait no compiler-placed spill can make return early (checked on the source) —
    the ping-pong kernel counts its waits, so its loops must hold no scratch access at all; and no MFMA loop of any of them carries a spill
    (a scratch round trip per K step is what cost the fused RPN heads 60 us before their sums were parked in memory)."""
    spec = importlib.util.spec_from_file_location("audit_halo_asm", os.path.join(ROOT, "tools", "audit_halo_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.audit_fp16()
    assert set(res) == {"k_bneck_h", "k_conv3x3_h", "k_conv_pp"}
    assert len(res["k_bneck_h"][1]) >= 6 and len(res["k_conv3x3_h"][1]) == 2 and len(res["k_conv_pp"][1]) >= 3
    for kernel, (waits, rows) in res.items():
        if kernel != "k_conv_pp":
            assert all(w.strip() == "0" for w in waits), (kernel, waits)
        for name, vgprs, sbytes, nscratch, lanes, loops in rows:
            assert loops, (kernel, name)                                   # the K loops were found
            assert all(inside == 0 for _, inside in loops), (kernel, name, loops)
            assert 0 < vgprs <= 256 and sbytes <= 384, (kernel, name, vgprs, sbytes)     # the spill budget of the phase borders (k_conv3x3_h<HEAD>: 324 B)
