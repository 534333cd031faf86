#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (one dir per pass) into per-kernel averages per dispatch."""
import csv, glob, sys, collections, re
root = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and not pat.search(k):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k[:110])
    for c in sorted(cs):
        v = cs[c]
        print(f"   {c:40s} avg/dispatch {sum(v)/len(v):.4g}   (n={len(v)})")
