#!/usr/bin/env python3
"""Registers / spills / LDS / occupancy of every gfx950 kernel of the library, from hipcc's -Rpass-analysis=kernel-resource-usage
(no GPU needed).  `python tools/resource_usage.py > profiles/roundN_kernel_resources.txt`: tracked round to round."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "open_knowledge_graph_embeddings_amd", "csrc")
rows = []
for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
    out = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "Function Name":
            cur = {"file": src, "name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    name = re.sub(r"\(.*", "", r["name"].replace("(anonymous namespace)::", "")).replace("okge::", "").replace("void ", "")
    print(f"{(r['file'][5:-4] + ': ' + name)[:100]:100s} {r.get('VGPRs', ''):>5s} {r.get('AGPRs', ''):>5s} {r.get('VGPRs Spill', ''):>6s} "
          f"{r.get('SGPRs Spill', ''):>6s} {r.get('ScratchSize [bytes/lane]', ''):>7s} {r.get('Occupancy [waves/SIMD]', ''):>4s} {r.get('LDS Size [bytes/block]', ''):>7s}")
