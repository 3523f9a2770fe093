#!/usr/bin/env python3
"""GPU box: okge_adagrad_lazy alone at configs[4]'s table shapes -- the rotating window sweep and the stamped rows apart,
cold (accumulators ~1e-22: every replayed step moves the row) and warm (a replayed step returns its input bits)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_knowledge_graph_embeddings_amd import hotpath as H  # noqa: E402

dev = torch.device("cuda:0")
eng = H.HotPath(dev)
shapes = [(200_000, 256), (50_000, 256)]


def run(window, frac, warm, reps=20):
    P = [torch.randn(s, device=dev) * 0.1 for s in shapes]
    G = [torch.zeros(s, device=dev) for s in shapes]
    S = [torch.full(s, 1e-4 if warm else 0.0, device=dev) for s in shapes]
    steps = [torch.zeros(s[0], dtype=torch.int32, device=dev) for s in shapes]
    maps = [torch.zeros(s[0], dtype=torch.uint8, device=dev) for s in shapes]
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    sel = [torch.rand(s[0], device=dev) < frac for s in shapes]
    tens = [(p, g, s_, st, m, 1) for p, g, s_, st, m in zip(P, G, S, steps, maps)]
    for _ in range(min(window, 64) + 2):                        # steady state: every row has been through the window once
        eng.adagrad_lazy(tens, cnt, window, False, 0.1)
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(reps):
        for m, g, se in zip(maps, G, sel):
            m[se] = 1
            g[se] = 1e-3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.adagrad_lazy(tens, cnt, window, False, 0.1)
        torch.cuda.synchronize()
        t += time.perf_counter() - t0
    return round(1e6 * t / reps, 1)


for warm in (False, True):
    for window, frac in ((8, 0.0), (8, 0.16), (4, 0.16), (16, 0.16), (1, 0.16)):
        print(json.dumps({"warm": warm, "window": window, "stamped_frac": frac, "us_incl_launch_and_sync": run(window, frac, warm)}), flush=True)
