#!/usr/bin/env python3
"""BUILD CONTAINER: copy the summaries of a tools/collect_profiles.sh run (gpurun_out/<tag>/) into profiles/ under a
round prefix, stamping the commit the library was built from; profiles/pmc_traffic.json is what bench.py reads for
`roofline.traffic` (with `_commit` / `_command` as its provenance).

    python tools/publish_profiles.py r2prof round2_final"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
os.environ["OKGE_COMMIT"] = commit
dst = os.path.join(ROOT, "profiles")
if os.path.isdir(os.path.join(src, "pmc_fetch")):          # raw CSVs still there: (re)summarise with the commit stamp
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), src], stdout=subprocess.DEVNULL)
else:                                                       # the box kept only the summaries: stamp the commit here
    tj = json.load(open(os.path.join(src, "pmc_traffic.json")))
    tj["_commit"] = commit
    json.dump(tj, open(os.path.join(src, "pmc_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True) + glob.glob(os.path.join(src, "kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, prefix + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, prefix + "_pmc_summary.txt"))
shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, prefix + "_pmc_traffic.json"))
if len(sys.argv) < 4 or sys.argv[3] != "--no-bench-traffic":       # (the other configurations' profiles: bench.py does not quote them)
    shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, "pmc_traffic.json"))
if os.path.exists(os.path.join(src, "bench_trace.json")):
    shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, prefix + "_bench_under_rocprof.json"))
print("published", prefix, "at", commit, json.load(open(os.path.join(dst, "pmc_traffic.json")))["_commit"])
