#!/usr/bin/env python3
"""Summarise a tools/collect_profiles.sh run: kernel stats, per-kernel PMC means, and HBM traffic per launch
(bytes = 2 * FETCH_SIZE KB * 1024 [gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM]
 + WRITE_SIZE KB * 1024)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void okge::", "").replace("okge::", "")


def pmc(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


fetch, write = pmc("pmc_fetch"), pmc("pmc_write")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("_"):
        continue
    if not any(x in k for x in ("fused", "dq", "prefix", "adagrad", "encode", "ranks", "eval_", "pool", "bn_", "col_", "dc_reduce")):
        continue
    fk, wk = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic[k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": 2.0 * fk * 1024 + wk * 1024}
traffic["_command"] = "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- " + os.environ.get(
    "OKGE_PROFILE_COMMAND", "python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline (OKGE_BENCH_OLP=0)")
traffic["_commit"] = os.environ.get("OKGE_COMMIT")          # stamped by tools/publish_profiles.py in the build container
json.dump(traffic, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(root, "pmc_summary.txt"), "w") as out:
    for sub in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
        out.write(f"== {sub}\n")
        for k, d in sorted(pmc(sub).items()):
            if any(x in k for x in ("fused", "dq", "prefix", "adagrad", "encode", "eval_", "pool", "bn_", "col_", "dc_reduce")):
                out.write(k + "\n")
                for c, v in sorted(d.items()):
                    out.write(f"    {c:36s} mean/launch = {v:16.1f}\n")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read()[:1500])
print(json.dumps(traffic, indent=1))
