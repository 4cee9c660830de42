#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not any(x in k for x in ("sf", "rle", "hist")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        big = sorted(v)[len(v) // 2:]          # ignore tiny warm-up dispatches
        print(f"   {c:28s} n={len(v):3d} mean_upper_half={sum(big) / len(big):16.1f}")
