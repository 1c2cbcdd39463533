#!/usr/bin/env python
"""Where a predict's wall time goes between kernels: from a rocprofv3 --kernel-trace CSV, the kernels of the LAST predict
(a run of launches delimited by k_preprocess), their summed duration, the idle gaps between consecutive kernels and the
distribution of kernel durations.  usage: trace_gaps.py <dir with *kernel_trace.csv> [--list]      --list: every kernel of that predict in launch order"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
starts = [i for i, r in enumerate(rows) if "k_preprocess" in r[2]]
a, b = starts[-2], starts[-1]                     # the last complete predict
ks = rows[a:b]
wall = ks[-1][1] - ks[0][0]
busy = sum(e - s for s, e, _ in ks)
gaps = [max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)]
dur = sorted(e - s for s, e, _ in ks)
print(f"kernels {len(ks)}  wall {wall / 1e3:.1f} us  sum of kernel durations {busy / 1e3:.1f} us ({100.0 * busy / wall:.1f} %)  "
      f"idle between kernels {sum(gaps) / 1e3:.1f} us ({100.0 * sum(gaps) / wall:.1f} %), median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us")
print("kernel durations (us): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % tuple(dur[int(q * (len(dur) - 1))] / 1e3 for q in (0, 0.1, 0.5, 0.9, 1)))
short = [d for d in dur if d < 20000]
print(f"kernels shorter than 20 us: {len(short)} ({sum(short) / 1e3:.1f} us together)")
if "--list" in sys.argv:
    import re
    t0 = ks[0][0]
    for s_, e_, n_ in ks:
        print(f"{(s_ - t0) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f} us  {re.sub(r'^void |mrcnn::', '', n_)[:100]}")
