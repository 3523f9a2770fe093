#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output: mean counter value per dispatch, per kernel."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0].replace("void okge::", "").replace("okge::", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if not any(x in k for x in ("fused", "dq", "prefix", "adagrad", "encode", "eval_")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
