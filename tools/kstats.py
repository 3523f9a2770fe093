#!/usr/bin/env python3
"""print a rocprofv3 kernel_stats.csv compactly: kernel, calls, avg / min / max us"""
import csv, re, sys
for f in sys.argv[1:]:
    print("==", f)
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "at::native" in n:
            continue
        m = re.search(r"(\w+(<[^>]*>)?)\(", n.replace("(anonymous namespace)::", ""))
        short = m.group(1) if m else n[:40]
        print(f"{short:42s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
