#!/usr/bin/env python
"""HBM traffic of the dominant kernel from two rocprofv3 PMC passes (MI355X_MICROARCH.md §HBM recipe).

usage: tools/pmc_traffic.py <dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass> <out.json> [kernel substring]

Each pass is `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -d <dir> -o p -- python bench.py
--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events` (counters in their own runs, no other trace domains).
FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide coalesced reads, so FETCH_SIZE is doubled.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter, match):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    tot, n, name, whole = 0.0, 0, None, 0.0
    for f in files:
        for row in csv.DictReader(open(f, newline="")):
            if row["Counter_Name"] != counter:
                continue
            v = float(row["Counter_Value"])
            whole += v
            if match in row["Kernel_Name"]:
                tot += v
                n += 1
                name = row["Kernel_Name"]
    if not n:
        raise SystemExit(f"no dispatch of a kernel matching {match!r} in {d}")
    return tot / n, n, name, whole


def main():
    fd, wd, out = sys.argv[1:4]
    match = sys.argv[4] if len(sys.argv) > 4 else "k_conv_mfma_glds<float, float, 128"
    f_avg, n, name, f_whole = per_kernel(fd, "FETCH_SIZE", match)
    w_avg, n2, _, w_whole = per_kernel(wd, "WRITE_SIZE", match)
    json.dump({"kernel": name, "launches_sampled": n,
               "FETCH_SIZE_avg_KB_raw": f_avg, "WRITE_SIZE_avg_KB": w_avg,
               "hbm_bytes_per_launch_corrected": (2.0 * f_avg + w_avg) * 1024.0,
               "whole_run_GB": {"fetch_x2": 2.0 * f_whole * 1024 / 1e9, "write": w_whole * 1024 / 1e9},
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports 1/2 of wide coalesced reads); "
                       "WRITE_SIZE uncalibrated; separate --pmc passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                       "--no-kernel-events`; per-launch average over every launch of the dominant kernel in the run"},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
