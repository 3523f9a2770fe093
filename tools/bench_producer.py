#!/usr/bin/env python3
"""End-to-end feed rate: OneToNBatchProducer (host collate thread + one pinned-arena H2D copy per batch) driving
FusedTrainStep at the FB15k-237 shape, against the same steps on batches already resident in HBM."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from open_knowledge_graph_embeddings_amd.dataset import OneToNBatchProducer, pack_groups  # noqa: E402
from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep  # noqa: E402


def table(rng, n_ent, n_rel, n_prefix):
    seen, rows = [], []
    for _ in range(n_prefix):
        groups = [[int(rng.integers(2, n_ent))] for _ in range(int(min(rng.geometric(0.55), 64)))]
        packed = pack_groups(groups).tolist()
        slot, rel, ent = int(rng.choice([0, 2])), int(rng.integers(2, n_rel)), int(rng.integers(2, n_ent))
        rows.append([rel if slot == 0 else ent, ent if slot == 0 else rel, len(seen), len(seen) + len(packed), 0, 0, slot])
        seen += packed
    return np.asarray(rows, np.int32), np.asarray(seen, np.int32)


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n_ent, n_rel, d = 14543, 239, 200
    pref, seen = table(rng, n_ent, n_rel, 512 * 120)
    E = torch.randn((n_ent, d), device=dev) * 0.1
    R = torch.randn((n_rel, d), device=dev) * 0.1
    step = FusedTrainStep(E, R, "complex", lr=0.3, input_dropout=0.4, seed=1)
    prod = OneToNBatchProducer(pref, seen, None, n_ent, batch_size=512, is_training_data=True, shuffle=True, device=dev, prefetch=4)
    resident = [cb for cb, _ in zip(prod, range(16))]
    for cb in resident:
        step.step(cb.batch, normalizer=cb.normalizer_loss)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(240):
        cb = resident[i % 16]
        step.step(cb.batch, normalizer=cb.normalizer_loss)
    torch.cuda.synchronize()
    ms_res = 1e3 * (time.perf_counter() - t0) / 240
    n = 0
    t0 = time.perf_counter()
    for _ in range(2):
        for cb in prod:
            step.step(cb.batch, normalizer=cb.normalizer_loss)
            n += 1
    torch.cuda.synchronize()
    ms_feed = 1e3 * (time.perf_counter() - t0) / n
    print(json.dumps({"ms_per_step_resident": round(ms_res, 4), "ms_per_step_fed_by_producer": round(ms_feed, 4), "batches": n}))


if __name__ == "__main__":
    main()
