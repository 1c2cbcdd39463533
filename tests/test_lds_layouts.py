"""The LDS layouts of the convolution kernels against the bank model of MI355X_MICROARCH.md (tools/lds_bank_model.py: the REAL lane
groups of ds_read_b128 / ds_write_b128, not contiguous sixteens).  Round 3 shipped two layouts that were conflict-free only under the
wrong grouping — the padded epilogue tile (2-way on every read-back) and multi-row halo regions of pitch W + 2 (2-way on every fragment
read of the mask head, 0.43 measured) — which the counters then showed; these pins keep the shipped layouts conflict-free
in the model, next to the counter evidence in profiles/r04_pmc_halo_geo_f32x3.txt / r04_pmc_kernels_f32x3_head.txt.  No GPU needed."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lds_bank_model", os.path.join(ROOT, "tools", "lds_bank_model.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)


def test_wave_private_epilogue_tiles_are_conflict_free_both_ways():
    assert M.epilogue_tile(32, True) == (16, 32)            # conv_epilogue_wave: 4 read-backs x 4 cycles, 4 transposing writes x 8
    assert M.epilogue_tile_h(64, True) == (32, 64)          # conv_epilogue_wave_h
    assert M.epilogue_tile(36, False) == (32, 32)           # what rounds 2-3 shipped: writes free, every read-back 2-way
    assert M.head_partial_write(True) == 8 and M.head_partial_write(False) == 64


def test_halo_planes_fragment_reads():
    # 32 consecutive slots: free at every shift (one-row, two-row and 64-wide regions)
    assert all(M.plane_fragment_read([s + p for p in range(32)]) == 4 for s in range(64))
    # regions whose rows are shorter than a wave's 32 pixels: pitch W + 16 keeps consecutive pixels consecutive mod 16
    for W in (14, 16, 24, 40, 47, 56):
        assert M.mean_fragment_cycles(W, W + 16) == 4, W
        assert M.mean_fragment_cycles(W, W + 2) > 5, W
    # ... and across the two zero rows between images, with the skew halo_geometry computes: (16 - 2 W mod 16) mod 16
    for W in (14, 12, 10):
        skew = (16 - (2 * W) % 16) % 16
        assert M.mean_fragment_cycles(W, W + 16, skew, W) == 4, W
    assert M.mean_fragment_cycles(14, 30, 0, 14) > 4
