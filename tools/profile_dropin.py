#!/usr/bin/env python3
"""GPU box: where the host time of the drop-in step goes (cProfile over bench.run_dropin's step, S-FB).
    python tools/profile_dropin.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_knowledge_graph_embeddings_amd import synthetic  # noqa: E402
from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta  # noqa: E402
from open_knowledge_graph_embeddings_amd.model import Models  # noqa: E402
from open_knowledge_graph_embeddings_amd.optim import OkgeAdagrad  # noqa: E402
from open_knowledge_graph_embeddings_amd.trainer import AddLossModule  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    if os.environ.get("OKGE_PROFILE_SINGLE_THREAD") == "1":      # backward on the calling thread (no hand-off to the engine's device thread)
        torch.autograd.set_multithreading_enabled(False)
    dev = torch.device("cuda:0")
    w = synthetic.WORKLOADS["S-FB"]
    host_batches = [synthetic.make_batch(w, seed=1234 + i) for i in range(4)]
    torch.manual_seed(0)
    m = Models.LookupComplexRelationModel(entity_slot_size=w.d, input_dropout=w.input_dropout, init_std=w.init_std, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=w.n_ent, relations_size=w.n_rel)).to(dev)
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0, training_outputs=False)
    opt = OkgeAdagrad(m.parameters(), lr=w.lr, weight_decay=1e-10, eps=1e-8)
    cand = torch.arange(w.n_ent, device=dev)[2:].int().unsqueeze(1)
    t = lambda a: torch.from_numpy(a).to(dev).unsqueeze(1)  # noqa: E731
    batches = [([(t(hb["po_rel"]), t(hb["po_obj"])), (t(hb["sp_subj"]), t(hb["sp_rel"]))],
                (torch.from_numpy(hb["pos_row"]).to(dev), torch.from_numpy(hb["pos_col"]).to(dev))) for hb in host_batches]
    norm = float(w.B * w.N)

    def step(i):
        inputs, coords = batches[i % len(batches)]
        opt.zero_grad()
        loss, _, _ = mod(inputs=inputs, labels=coords, use_batch_shared_entities=False, batch_shared_entities=cand, epoch=1,
                         input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / norm).backward()
        opt.step()
    for i in range(50):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t_host = time.perf_counter() - t0               # host time to ISSUE the steps (the device may lag behind)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"issue {1e3 * t_host / steps:.4f} ms/step, complete {1e3 * t_all / steps:.4f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(steps):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
