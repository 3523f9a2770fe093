#!/usr/bin/env python3
"""Diagnostic (-DOKGE_STAMPS build): phase timeline of two fused_tile32 workgroups sharing a CU."""
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
import bench

w = synthetic.WORKLOADS["S-FB"]
if os.environ.get("OKGE_TL_D"):                      # another slot size at the S-FB shape (e.g. 256: the single-buffer instance)
    import dataclasses
    w = dataclasses.replace(w, d=int(os.environ["OKGE_TL_D"]))
dev = torch.device("cuda:0")
E, R = synthetic.make_tables(w)
step = FusedTrainStep(torch.from_numpy(E).to(dev), torch.from_numpy(R).to(dev), w.scorer, lr=w.lr,
                      input_dropout=w.input_dropout, seed=1)
batches = [bench.to_dev_batch(synthetic.make_batch(w, seed=i), w, dev) for i in range(4)]
tile_w = int(os.environ.get("OKGE_TILE_W", "64"))
tiles = (64 // tile_w) * ((w.N + 63) // 64)
buf = torch.zeros(tiles * 4 + tiles * 80, dtype=torch.int64, device=dev)
os.environ["OKGE_STAMPS_PTR"] = hex(buf.data_ptr())
for i in range(3):
    step.step(batches[i % 4])
torch.cuda.synchronize()
a = buf.cpu().numpy()
head = a[: tiles * 4].reshape(-1, 4)
tl = a[tiles * 4:].reshape(-1, 80)
if tile_w == 64:
    for wg in (int(sys.argv[1]) if len(sys.argv) > 1 else 5,):
        base = head[wg, 0]
        print(f"WG {wg}: start 0 end {head[wg,1]-base}")
        print("chunk | staged  barrier->P1start  P1end  P2end | P1 dur  epi+P2 dur  wait@barrier  period")
        prev = None
        for ch in range(8):
            r = tl[wg, 4 * ch: 4 * ch + 4] - base
            print(f"{ch:3d}  {r[0]:7d} {r[1]:7d} {r[2]:7d} {r[3]:7d} | {r[2]-r[1]:6d} {r[3]-r[2]:6d} {r[1]-r[0]:6d} "
                  f"{(r[1]-prev) if prev is not None else 0:6d}")
            prev = r[1]
        print("staging of each chunk: entered (closing barrier passed) -> queries parked -> label bits -> next chunk requested -> staged (masked rows out)")
        for ch in range(8):
            r = [tl[wg, 48 + ch], tl[wg, 56 + ch], tl[wg, 64 + ch], tl[wg, 72 + ch], tl[wg, 4 * ch]]
            if min(r) > 0:
                r = [x - base for x in r]
                print(f"{ch:3d}  {r[0]:7d} +{r[1]-r[0]:5d} +{r[2]-r[1]:5d} +{r[3]-r[2]:5d} +{r[4]-r[3]:5d}")
        print("loop done at", tl[wg, 32] - base)
        pe = tl[wg, 40:45] - base
        print(f"prologue: gather loads issued {pe[0]}, tile parked in LDS {pe[1]}, barrier passed {pe[2]}")
        print(f"epilogue: partial sums combined {pe[3]}, loss partial out {pe[4]}, workgroup end {head[wg,1]-base}")
    sys.exit(0)
for wg in (int(sys.argv[1]) if len(sys.argv) > 1 else 5,):
    A, B = wg, wg + 256
    base = min(head[A, 0], head[B, 0])
    print(f"WG {A}: start {head[A,0]-base} end {head[A,1]-base} | WG {B}: start {head[B,0]-base} end {head[B,1]-base}")
    print("chunk |  A: P1start P1end P2start P2end (dur P1, P2) |  B: ...")
    for ch in range(16):
        ra = tl[A, 4 * ch: 4 * ch + 4] - base
        rb = tl[B, 4 * ch: 4 * ch + 4] - base
        print(f"{ch:3d}  A {ra[0]:7d} {ra[1]:7d} {ra[2]:7d} {ra[3]:7d} ({ra[1]-ra[0]:5d},{ra[3]-ra[2]:5d})   "
              f"B {rb[0]:7d} {rb[1]:7d} {rb[2]:7d} {rb[3]:7d} ({rb[1]-rb[0]:5d},{rb[3]-rb[2]:5d})")
