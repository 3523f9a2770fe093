#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of fused_tile_kernel<train> (needs the -DOKGE_STAMPS build:
tools/build_stamps.sh).  Never quote this build's run time; read the SHARES."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from open_knowledge_graph_embeddings_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "libokge_hip_stamps.so")
from open_knowledge_graph_embeddings_amd import synthetic
from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
sys.path.insert(0, ROOT)
import bench

w = synthetic.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "S-FB"]
dev = torch.device("cuda:0")
E, R = synthetic.make_tables(w)
step = FusedTrainStep(torch.from_numpy(E).to(dev), torch.from_numpy(R).to(dev), w.scorer, lr=w.lr,
                      input_dropout=w.input_dropout, seed=1)
batches = [bench.to_dev_batch(synthetic.make_batch(w, seed=i), w, dev) for i in range(4)]
tiles = (w.N + 63) // 64
buf = torch.zeros(tiles * 8 * 4 * 10, dtype=torch.int64, device=dev)
os.environ["OKGE_STAMPS_PTR"] = hex(buf.data_ptr())
for i in range(5):
    step.step(batches[i % 4])
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 10)
a = a[a[:, 8] > 0]
names = ["prologue(cand tile)", "phaseA(Q->LDS,bits,barrier)", "score product", "loss epilogue", "barrier",
         "G tile -> HBM", "dC product", "dC epilogue"]
tot = a[:, 8].astype(np.float64)
print(f"waves {len(a)}  mean total cycles {tot.mean():.0f}  max {tot.max():.0f}  min {tot.min():.0f}")
for i, n in enumerate(names):
    print(f"  {n:32s} {a[:, i].mean():10.0f} cyc  {100 * a[:, i].mean() / tot.mean():5.1f} %")
print(f"  {'unaccounted':32s} {(tot - a[:, :8].sum(1)).mean():10.0f} cyc")
span = (a[:, 9] + a[:, 8]).max() - a[:, 9].min()
print(f"kernel span (first wave start -> last wave end): {span} shader cycles (= {span / 2400:.1f} us at 2.4 GHz)")
