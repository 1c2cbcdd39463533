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
