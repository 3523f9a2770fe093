#!/bin/bash
# round-4 helper (GPU box): the whole GPU suite, then the bench line as the driver runs it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/full.log 2>&1; echo "tests rc=$?" >> $O/full.log
tail -12 $O/full.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r4/bench.json")))
print("S-FB ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "dropin", d["dropin"]["ms_per_step"], "olp", d["olp"]["ms_per_step"])
for k, v in d["configs"].items():
    print(k, v["ms_per_step"], "step_frac %.3f" % v["step_frac"], "tile %.1f us frac %.3f" % (v.get("tile_us", 0), v.get("tile_frac", 0)), v["kernels_us"])
PY
