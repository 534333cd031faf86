#!/usr/bin/env python3
"""One steady-state forward out of a rocprofv3 --kernel-trace CSV: kernels between the last two launches of a marker kernel
(default sweep_corr), grouped by name.  tools/frame_kernels.py <kernel_trace.csv> [marker substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "sweep_corr_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
frame = rows[a:b]
span = (int(frame[-1]["End_Timestamp"]) - int(frame[0]["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(lambda: [0, 0.0])
for r in frame:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[r["Kernel_Name"]][0] += 1
    acc[r["Kernel_Name"]][1] += d
busy = sum(v[1] for v in acc.values())
print(f"one forward: {len(frame)} launches, span {span:.1f} us, kernel time {busy:.1f} us")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-110s x%3d %9.1f us" % (k[:110], n, t))
