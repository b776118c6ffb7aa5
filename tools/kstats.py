#!/usr/bin/env python3
"""Prints the top kernels of the NEWEST rocprofv3 kernel_stats.csv under a directory. Usage: kstats.py DIR [per-step divisor | cell:<kernel substring>]"""
import csv, glob, os, sys
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
f = max(files, key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
div = 1.0
if len(sys.argv) > 2:
    a = sys.argv[2]
    if a.startswith("cell:"):
        div = [int(r["Calls"]) for r in rows if a[5:] in r["Name"]][0] / 57.0
    else:
        div = float(a)
print(f, f"total {tot / 1e6 / div:.2f} ms per unit")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{100 * float(r['TotalDurationNs']) / tot:6.2f}% {float(r['TotalDurationNs']) / 1e6 / div:8.2f} ms {int(r['Calls']) / div:8.1f} calls {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:88]}")
