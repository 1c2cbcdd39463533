#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) kernel trace: per-kernel calls / total / average.
usage: tools/rocpd_summary.py results.db [out.csv]"""
import csv
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_digest import short  # noqa: E402  (demangles this library's kernel names)

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
for name, calls, total, avg, pct in rows:
    w.writerow([short(name), calls, round(total, 3), round(avg, 3), round(pct, 4)])
