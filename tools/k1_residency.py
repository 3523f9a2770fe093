#!/usr/bin/env python3
"""Diagnostic (-DOKGE_STAMPS build): where and when each fused_tile32 workgroup ran."""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from open_knowledge_graph_embeddings_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "libokge_hip_stamps.so")
from open_knowledge_graph_embeddings_amd import synthetic
from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
import bench

w = synthetic.WORKLOADS["S-FB"]
dev = torch.device("cuda:0")
E, R = synthetic.make_tables(w)
step = FusedTrainStep(torch.from_numpy(E).to(dev), torch.from_numpy(R).to(dev), w.scorer, lr=w.lr,
                      input_dropout=w.input_dropout, seed=1)
batches = [bench.to_dev_batch(synthetic.make_batch(w, seed=i), w, dev) for i in range(4)]
tile_w = int(os.environ.get("OKGE_TILE_W", "64"))
tiles = (64 // tile_w) * ((w.N + 63) // 64)
buf = torch.zeros(tiles * 4, dtype=torch.int64, device=dev)
os.environ["OKGE_STAMPS_PTR"] = hex(buf.data_ptr())
for i in range(5):
    step.step(batches[i % 4])
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 4)
t0, t1, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
cu = (xcc & 7) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15)
base = t0.min()
print("workgroups", len(a), "distinct CUs", len(set(cu.tolist())), "kernel span cycles", int(t1.max() - base))
print("mean WG lifetime cycles", float((t1 - t0).mean()))
per = defaultdict(list)
for i in range(len(a)):
    per[int(cu[i])].append((int(t0[i] - base), int(t1[i] - base), i))
cnt = defaultdict(int)
overlap_frac = []
for k, v in per.items():
    cnt[len(v)] += 1
    if len(v) == 2:
        (s0, e0, _), (s1, e1, _) = sorted(v)
        ov = max(0, min(e0, e1) - max(s0, s1))
        overlap_frac.append(ov / max(1, max(e0, e1) - min(s0, s1)))
print("CUs by number of workgroups hosted:", dict(cnt))
if overlap_frac:
    print("pairs: mean overlap fraction of the pair's span %.3f (min %.3f)" % (np.mean(overlap_frac), np.min(overlap_frac)))
for k in list(per)[:6]:
    print(" CU", hex(k), [(s, e, i) for s, e, i in sorted(per[k])])
