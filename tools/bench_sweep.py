#!/usr/bin/env python3
"""GPU box: the score sweep (okge_score_prefixes = encode_queries + fused_tile_kernel<score>) alone, HIP-event timed.
Round 3 used it to A/B a register-operand / double-buffered variant of the sweep kernel (as fused_tile64_kernel has): 44.7 vs
44.4 us at S-FB, 1211 vs 1187 us at d=256 B=4096 N=65536 (the sweep already runs at 0.72 of peak there), 18.3 -> 23.7 us at
d=64 B=128 and the three-chain evaluation 0.0427 -> 0.0496 ms/batch -- not adopted."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_knowledge_graph_embeddings_amd import hotpath as H  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hp = H.HotPath(dev)
    rng = np.random.default_rng(0)
    for n_ent, d, B in ((14543, 200, 512), (14543, 64, 128), (65538, 256, 4096), (10002, 128, 512)):
        E = torch.randn((n_ent, d), device=dev) * 0.1
        R = torch.randn((300, d), device=dev) * 0.1
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        b = H.PrefixBatch(po_rel=t(rng.integers(2, 300, B // 2).astype(np.int32)), po_obj=t(rng.integers(2, n_ent, B // 2).astype(np.int32)),
                          sp_subj=t(rng.integers(2, n_ent, B // 2).astype(np.int32)), sp_rel=t(rng.integers(2, 300, B // 2).astype(np.int32)),
                          cand_first=2, n_cand=n_ent - 2)
        out = hp.score(E, R, "complex", b)
        for _ in range(5):
            hp.score(E, R, "complex", b, out=out)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        ev0.record()
        for _ in range(n):
            hp.score(E, R, "complex", b, out=out)
        ev1.record()
        torch.cuda.synchronize()
        us = 1e3 * ev0.elapsed_time(ev1) / n
        print(f"score d={d} B={B} N={n_ent - 2}: {us:.1f} us per call, {2.0 * B * (n_ent - 2) * d / us / 1e6:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
