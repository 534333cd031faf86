#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 --kernel-trace --stats kernel_stats.csv: tools/kernel_stats_table.py <csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print("%-100s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       100 * float(r["TotalDurationNs"]) / tot))
