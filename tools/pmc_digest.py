#!/usr/bin/env python
"""Per-kernel digest of rocprofv3 PMC passes (MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots recipe: counters in their own
runs, kernel-trace only).

usage: tools/pmc_digest.py <out.json> <stats.csv of the un-profiled run> <pmc dir> [<pmc dir> ...]

Each <pmc dir> is the output of one `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -o p -- python
bench.py ...`.  For every kernel: launches, the per-launch average of every counter found, and derived figures —
  hbm_bytes      = 2*FETCH_SIZE + WRITE_SIZE (KB → B; gfx950 reports half of wide coalesced reads, WRITE_SIZE uncalibrated)
  hbm_GBps       = hbm_bytes / avg duration (duration from the kernel-trace --stats run: a profiled pass runs slower)
  hbm_frac       = hbm_GBps / 6290 (measured streaming peak), and / 8000 (spec)
  clock_GHz      = GRBM_GUI_ACTIVE / 8 XCDs / duration of the SAME pass (the counter is summed over the XCDs)
  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs × GRBM_GUI_ACTIVE / 8)
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import json
import os
import sys


import re


def _demangle_mrcnn(name):
    """Minimal Itanium demangling of this library's kernels (GNU c++filt does not know DF16_ = _Float16; no llvm-cxxfilt here):
    _ZN5mrcnn<len><ident>I<template args>E... with args f / DF16_ / Li<n>E / Lb<0|1>E."""
    m = re.match(r"_ZN5mrcnn(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    ident = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        i = 1
        while i < len(rest) and rest[i] != "E":
            if rest.startswith("DF16_", i):
                args.append("_Float16"); i += 5
            elif rest[i] == "f":
                args.append("float"); i += 1
            elif rest[i] == "L":
                j = rest.index("E", i)
                args.append(rest[i + 2:j].replace("n", "-")); i = j + 1
            else:
                break
    return ident + ("<" + ", ".join(args) + ">" if args else "")


def short(name):
    """rocprofv3 hands out some kernel names mangled; drop the argument list and the namespace."""
    if name.startswith("_Z"):
        return _demangle_mrcnn(name)
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):            # the argument list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].replace("void ", "").replace("mrcnn::", "").strip()


def main():
    out, stats = sys.argv[1:3]
    dirs = sys.argv[3:]
    dur = {}
    for r in csv.DictReader(open(stats, newline="")):
        dur[short(r["kernel"])] = (int(r["calls"]), float(r["avg_us"]), float(r["percent"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    pass_dur = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        names = set()
        for f in files:
            for r in csv.DictReader(open(f, newline="")):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Counter_Name"])
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                k = short(r["Kernel_Name"])
                for c in names:
                    pass_dur[k][c].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {}
    for k, counters in acc.items():
        e = {c: sum(v) / len(v) for c, v in counters.items()}
        n = max(len(v) for v in counters.values())
        rec = {"launches_sampled": n, "counters_avg_per_launch": {c: round(v, 1) for c, v in e.items()}}
        if k in dur:
            rec["calls_unprofiled_run"], rec["avg_us_unprofiled"], rec["percent_of_gpu_time"] = dur[k]
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            b = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
            rec["hbm_bytes_per_launch"] = round(b)
            if k in dur and dur[k][1] > 0:
                gbps = b / (dur[k][1] * 1e-6) / 1e9
                rec["hbm_GBps"] = round(gbps, 1)
                rec["hbm_frac_of_6290"] = round(gbps / 6290.0, 4)
                rec["hbm_frac_of_8000"] = round(gbps / 8000.0, 4)
        if "GRBM_GUI_ACTIVE" in e:
            pd = pass_dur[k].get("GRBM_GUI_ACTIVE")
            if pd:
                rec["clock_GHz"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / (sum(pd) / len(pd)) / 1e3, 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e["GRBM_GUI_ACTIVE"] > 0:
                rec["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0), 4)
        if e.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_per_active"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 5)
        res[k] = rec
    order = sorted(res, key=lambda k: -(res[k].get("percent_of_gpu_time") or 0))
    json.dump({"kernels": {k: res[k] for k in order},
               "note": "per-launch averages; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM, WRITE_SIZE uncalibrated; durations from the "
                       "un-profiled --kernel-trace --stats run of the same command; clock from the profiled pass itself"},
              open(out, "w"), indent=1)
    for k in order[:14]:
        r = res[k]
        print(f"{k[:58]:58s} {r.get('avg_us_unprofiled', 0):9.1f} us  hbm {r.get('hbm_GBps', 0):7.1f} GB/s  clk {r.get('clock_GHz', 0):5.2f}  mfma {r.get('mfma_util', 0):5.3f}")


if __name__ == "__main__":
    main()
