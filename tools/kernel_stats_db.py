#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results.db (rocprofv3 --kernel-trace --stats writes sqlite by default on ROCm 7.2).
usage: tools/kernel_stats_db.py <results.db> [skip_first_n_dispatches_per_kernel]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
agg = {}
for name, s, e in rows:
    agg.setdefault(name, []).append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'%':>6s}")
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:90]:90s} {len(v):6d} {sum(v):10.1f} {sum(v)/len(v):9.2f} {min(v):9.2f} {100*sum(v)/tot:6.2f}")
