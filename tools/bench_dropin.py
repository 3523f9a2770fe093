#!/usr/bin/env python3
"""ms/step of the DROP-IN path at S-FB: the reference Trainer's own step sequence
(trainer.py:206-244: zero_grad, AddLossModule forward, (loss/normalizer).backward(), optimizer.step()) on our Models /
AddLossModule, in three configurations:
   api        AddLossModule as the reference calls it (all_outputs materialised) + torch.optim.Adagrad
   api+opt    ... + optim.OkgeAdagrad
   fast       AddLossModule(training_outputs=False) + OkgeAdagrad, labels as column-sorted coordinates
and FusedTrainStep for comparison."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from open_knowledge_graph_embeddings_amd import synthetic  # noqa: E402
from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta  # noqa: E402
from open_knowledge_graph_embeddings_amd.model import Models  # noqa: E402
from open_knowledge_graph_embeddings_amd.optim import OkgeAdagrad  # noqa: E402
from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep  # noqa: E402
from open_knowledge_graph_embeddings_amd.trainer import AddLossModule  # noqa: E402

w = synthetic.WORKLOADS["S-FB"]
dev = torch.device("cuda:0")
host = [synthetic.make_batch(w, seed=1234 + i) for i in range(8)]
cand = torch.arange(w.n_ent, device=dev)[2:].int().unsqueeze(1)
batches = []
for hb in host:
    t = lambda a: torch.from_numpy(a).to(dev).unsqueeze(1)  # noqa: E731
    y = torch.from_numpy(synthetic.dense_labels(hb, w.B, w.N)).to(dev)
    coords = (torch.from_numpy(hb["pos_row"]).to(dev), torch.from_numpy(hb["pos_col"]).to(dev))
    batches.append(([(t(hb["po_rel"]), t(hb["po_obj"])), (t(hb["sp_subj"]), t(hb["sp_rel"]))], y, coords))


def run(name, outputs, okge_opt, coord_labels, steps=300, warm=30):
    torch.manual_seed(0)
    m = Models.LookupComplexRelationModel(entity_slot_size=w.d, input_dropout=w.input_dropout, init_std=w.init_std, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=w.n_ent, relations_size=w.n_rel)).cuda()
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0, training_outputs=outputs)
    opt = (OkgeAdagrad if okge_opt else torch.optim.Adagrad)(m.parameters(), lr=w.lr, weight_decay=1e-10, eps=1e-8)
    norm = float(w.B * w.N)

    def step(i):
        inputs, y, coords = batches[i % 8]
        opt.zero_grad()
        loss, hook, _ = mod(inputs=inputs, labels=coords if coord_labels else y, use_batch_shared_entities=False,
                            batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / norm).backward()
        opt.step()
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    print(json.dumps({"path": name, "ms_per_step": round(1e3 * (time.perf_counter() - t0) / steps, 4)}), flush=True)


run("api: all_outputs + torch.optim.Adagrad, dense labels", True, False, False)
run("api+opt: all_outputs + OkgeAdagrad, dense labels", True, True, False)
run("fast: no training outputs + OkgeAdagrad, dense labels", False, True, False)
run("fast: no training outputs + OkgeAdagrad, coordinate labels", False, True, True)
E, R = synthetic.make_tables(w)
st = FusedTrainStep(torch.from_numpy(E).to(dev), torch.from_numpy(R).to(dev), w.scorer, lr=w.lr, input_dropout=w.input_dropout, seed=1)
fb = [bench.to_dev_batch(hb, w, dev) for hb in host]
for i in range(30):
    st.step(fb[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300):
    st.step(fb[i % 8])
torch.cuda.synchronize()
print(json.dumps({"path": "FusedTrainStep", "ms_per_step": round(1e3 * (time.perf_counter() - t0) / 300, 4)}))

if os.environ.get("OKGE_PROFILE_DROPIN") == "1":
    import cProfile
    import pstats
    torch.manual_seed(0)
    m = Models.LookupComplexRelationModel(entity_slot_size=w.d, input_dropout=w.input_dropout, init_std=w.init_std, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=w.n_ent, relations_size=w.n_rel)).cuda()
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0, training_outputs=False)
    opt = OkgeAdagrad(m.parameters(), lr=w.lr, weight_decay=1e-10, eps=1e-8)

    def step(i):
        inputs, y, coords = batches[i % 8]
        opt.zero_grad()
        loss, hook, _ = mod(inputs=inputs, labels=coords, use_batch_shared_entities=False, batch_shared_entities=cand, epoch=1,
                            input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / float(w.B * w.N)).backward()
        opt.step()
    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    pstats.Stats(pr).sort_stats("tottime").print_stats(24)
