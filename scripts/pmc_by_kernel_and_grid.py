#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass: mean counter value per (kernel, launch grid) -- the multigrid kernels run on every level, and
the levels differ only by their grid.  usage: pmc_by_kernel_and_grid.py <counter_collection.csv> <COUNTER> <kernel name part>..."""
import collections
import csv
import sys

f, counter, parts = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != counter:
        continue
    name = r["Kernel_Name"]
    if parts and not any(p in name for p in parts):
        continue
    grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    acc[(name[:80], grid)].append(float(r["Counter_Value"]))
for (name, grid), v in sorted(acc.items()):
    print(counter, name, "grid", grid, "launches", len(v), "mean_KB", round(sum(v) / len(v), 1))
