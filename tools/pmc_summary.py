#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean of each counter over
the full-size dispatches (largest grid of that kernel)."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(lambda: collections.defaultdict(list))
grid = collections.defaultdict(int)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    g = int(r["Grid_Size"])
    grid[k] = max(grid[k], g)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if int(r["Grid_Size"]) != grid[k]:
        continue
    by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(by.items()):
    if not any(x in k for x in ("k_joint_fwd", "k_dhidden", "k_dw", "k_lattice", "k_make_hidden")):
        continue
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:32s} {sum(v)/len(v):.4g}  (n={len(v)})")
