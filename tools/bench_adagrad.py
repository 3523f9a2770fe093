#!/usr/bin/env python3
"""Adagrad sweep variants on the token tables of BASELINE configs[4] (200 k + 50 k rows of 256 floats), 15 % of the rows with a
gradient: okge_adagrad_step2 (dense), okge_adagrad_multi without / with the touched-row map.  HIP-event times per launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_knowledge_graph_embeddings_amd import hotpath as H  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    eng = H.HotPath(dev)
    d = 256
    tabs = []
    for rows in (200_000, 50_000):
        W = torch.randn((rows, d), device=dev) * 0.1
        tabs.append(dict(W=W, g=torch.zeros_like(W), s=torch.zeros_like(W), map=torch.zeros(rows, dtype=torch.uint8, device=dev),
                         rows=torch.randperm(rows, device=dev)[:int(0.15 * rows)]))

    def fill(stamp):
        for t in tabs:
            t["g"].index_fill_(0, t["rows"], 1e-3)
            t["map"].index_fill_(0, t["rows"], stamp)

    def timed(name, fn, warm):
        for t in tabs:
            t["s"].fill_(1e-4 if warm else 0.0)
        ts = []
        for it in range(12):
            fill(it % 255 + 1)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(it % 255 + 1)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts = sorted(ts[2:])
        print(f"{name:34s} {'warm' if warm else 'cold'}  median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}", flush=True)

    e, r = tabs
    for warm in (False, True):
        timed("adagrad_step2 (dense)", lambda st: eng.adagrad2(e["W"], e["g"], e["s"], r["W"], r["g"], r["s"], 0.1), warm)
        timed("adagrad_multi, no map", lambda st: eng.adagrad_multi([(e["W"], e["g"], e["s"]), (r["W"], r["g"], r["s"])], 0.1), warm)
        timed(f"adagrad_multi, map (U={os.environ.get('OKGE_ADAGRAD_U', '4')})",
              lambda st: eng.adagrad_multi([(e["W"], e["g"], e["s"], e["map"], st), (r["W"], r["g"], r["s"], r["map"], st)], 0.1), warm)


if __name__ == "__main__":
    main()
