#!/usr/bin/env python3
"""Per-kernel timing of one evaluation batch (S-FB shape) on the fused and the materialising path."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from open_knowledge_graph_embeddings_amd import hotpath as H, synthetic  # noqa: E402
from open_knowledge_graph_embeddings_amd.dataset import CollatedBatch  # noqa: E402
from open_knowledge_graph_embeddings_amd.evaluate import FusedEvaluator, PipelinedEvaluator  # noqa: E402

import dataclasses  # noqa: E402
w = synthetic.WORKLOADS["S-FB"]
if os.environ.get("OKGE_EVAL_D"):            # e.g. OKGE_EVAL_D=512 OKGE_EVAL_SCORER=distmult: configs[2]'s table shape
    w = dataclasses.replace(w, d=int(os.environ["OKGE_EVAL_D"]), scorer=os.environ.get("OKGE_EVAL_SCORER", w.scorer))
dev = torch.device("cuda:0")
E, R = synthetic.make_tables(w)
Et, Rt = torch.from_numpy(E).to(dev), torch.from_numpy(R).to(dev)
hb = synthetic.make_eval_batch(w, seed=777)
eb = bench.to_dev_batch(hb, w, dev)
t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
csr = [t(hb[k]) for k in ("filt_ptr", "filt_col", "row_ptr", "grp_ptr", "ids")]
cb = CollatedBatch(eb, float(w.B * w.N), float(hb["n_pos"]), w.N, row_ptr=csr[2], grp_ptr=csr[3], ids=csr[4], filt_ptr=csr[0], filt_col=csr[1])
eng = H.HotPath(dev)
import numpy as np  # noqa: E402
# floor of the sweep: one group in the whole batch (the counting loop runs for one row only)
one = CollatedBatch(eb, 1.0, 1.0, w.N, row_ptr=t(np.concatenate([[0], np.ones(w.B, np.int64)])), grp_ptr=t(np.asarray([0, 1], np.int64)),
                    ids=t(np.asarray([5], np.int32)), filt_ptr=t(np.zeros(w.B + 1, np.int64)), filt_col=t(np.zeros(0, np.int32)))
for name, cls in (("fused", FusedEvaluator), ("fused-one-group", FusedEvaluator), ("pipelined", PipelinedEvaluator)):
    if name == "fused-one-group":
        cb_keep, cb = cb, one
    elif name == "pipelined":
        cb = cb_keep
    ev = cls(Et, Rt, w.scorer, engine=eng)
    ev.run([cb] * 192)          # fresh streams are slow until the runtime's per-queue pools have grown to a full run's depth
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev.run([cb] * 40)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 40
    t0 = time.perf_counter()
    ev.run([cb] * 640)               # steady state: start-up (first launches, the final host read) amortised
    torch.cuda.synchronize()
    ms_long = 1e3 * (time.perf_counter() - t0) / 640
    eng.timing(True)
    ev.run([cb] * 10)
    torch.cuda.synchronize()
    per = {k: round(v[0] / v[1] * 1e3, 2) for k, v in eng.timing_collect().items()}
    eng.timing(False)
    print(json.dumps({"path": name, "ms_per_batch": round(ms, 4), "ms_per_batch_640": round(ms_long, 4), "kernels_us": per}))
