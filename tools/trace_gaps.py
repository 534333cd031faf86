#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv and prints, for the steady-state part of a bench run, the busy time (sum of
kernel durations), the span and the idle gaps per step:  python tools/trace_gaps.py <kernel_trace.csv> <steps> [marker]
A step is delimited by the marker kernel (default: softmax_regress, the last kernel of a Path B forward)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else "softmax_regress"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
ends = ends[-(steps + 1):]
sel = rows[ends[0] + 1: ends[-1] + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print(f"{steps} steps: span {(t1 - t0) / steps / 1e6:.3f} ms/step, kernels busy {busy / steps / 1e6:.3f} ms/step, "
      f"idle {(t1 - t0 - busy) / steps / 1e6:.3f} ms/step, {len(sel) / steps:.1f} kernels/step")
per = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = r["Kernel_Name"][:90]
    per[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); per[k][1] += 1
for k, (ns, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {ns / steps / 1e6:7.3f} ms/step  x{n / steps:4.1f}  {k}")
# largest gaps
gaps = []
for a, b in zip(sel, sel[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]))
gaps.sort(reverse=True)
print("largest gaps (us):")
for g, a, b in gaps[:12]:
    print(f"  {g / 1e3:8.1f}  after {a}  before {b}")
