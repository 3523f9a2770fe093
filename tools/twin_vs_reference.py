#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (imports /root/reference): times the imported reference classes (Models.* + AddLossModule +
OptimRegime, as Trainer.compute_one_batch drives them, openkge/trainer.py:181-257) against oracle/torch_twin.py -- the
CPU baseline bench.py reports -- on the same S-FB batches and thread count, and checks that they compute the same loss.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 -B /root/repo/tools/twin_vs_reference.py [threads]

Prints one JSON line (committed as profiles/round2_twin_vs_reference.json).  The twin is the reference's ATen op
sequence without its Python-side extras, so the reported baseline errs in the CPU's favour."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.path.isdir("/root/reference"), "needs the reference checkout (build container)"
if "/root/reference" not in sys.path:
    sys.path.insert(0, "/root/reference")

from openkge.dataset import EntityRelationDatasetMeta  # noqa: E402
from openkge.model import Models  # noqa: E402
from openkge.trainer import AddLossModule  # noqa: E402
from utils.optim import OptimRegime  # noqa: E402

from open_knowledge_graph_embeddings_amd import synthetic  # noqa: E402
from oracle import torch_twin  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
w = synthetic.WORKLOADS["S-FB"]
host = [synthetic.make_batch(w, seed=1234 + i) for i in range(4)]
cand = torch.arange(w.n_ent)[2:].int().unsqueeze(1)
prepared = []
for hb in host:
    y = torch.from_numpy(synthetic.dense_labels(hb, w.B, w.N))
    po = (torch.from_numpy(hb["po_rel"]).unsqueeze(1), torch.from_numpy(hb["po_obj"]).unsqueeze(1))
    sp = (torch.from_numpy(hb["sp_subj"]).unsqueeze(1), torch.from_numpy(hb["sp_rel"]).unsqueeze(1))
    prepared.append((po, sp, y))


def timed(fn, warm=3, steps=30):
    for i in range(warm):
        fn(i)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    return 1e3 * (time.perf_counter() - t0) / steps


# ---- the reference ----------------------------------------------------------------------------------------------
meta = EntityRelationDatasetMeta(entity_id_count_map={}, relation_id_count_map={}, entity_token_id_count_map={},
                                 relation_token_id_count_map={}, entity_id_to_tokens_map={}, relation_id_to_tokens_map={},
                                 entities_size=w.n_ent, relations_size=w.n_rel, min_entities_size=2, min_relations_size=2,
                                 entity_tokens_size=4, relation_tokens_size=4, max_length=1)
torch.manual_seed(1234)
ref = Models.LookupComplexRelationModel(entity_slot_size=w.d, input_dropout=w.input_dropout, init_std=w.init_std, sparse=False,
                                        train_data=meta)
ref.train()
opts = OptimRegime.setup_optimizer_regime(args={"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": w.lr,
                                                                         "weight_decay": 1.0e-10},
                                                "lr_scheduler_config": None}, model=ref)
mod = AddLossModule(ref, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
mod.train()
ref_losses = []


def ref_step(i):
    po, sp, y = prepared[i % 4]
    for o in opts:
        o.update(1, i + 1)
        o.zero_grad()
    loss, _, _ = mod(inputs=[po, sp], labels=y.clone(), use_batch_shared_entities=False, batch_shared_entities=cand, epoch=1,
                     input_style_triple_or_prefix="right_and_left_prefix")
    (loss.sum() / float(y.numel())).backward()
    for o in opts:
        o.step()
    ref_losses.append(float(loss))


ref_ms = timed(ref_step)

# ---- the twin ---------------------------------------------------------------------------------------------------
torch.manual_seed(1234)
twin = torch_twin.TwinModel(w.scorer, w.n_ent, w.n_rel, w.d, input_dropout=w.input_dropout, init_std=w.init_std)
twin.train()
opt = torch_twin.make_adagrad(twin, lr=w.lr)
twin_losses = []


def twin_step(i):
    po, sp, y = prepared[i % 4]
    twin_losses.append(float(torch_twin.train_step(twin, opt, po, sp, cand, y)))


twin_ms = timed(twin_step)
# same seed, same construction order, same op sequence incl. the dropout draws: the first losses agree
first = abs(ref_losses[0] - twin_losses[0]) / abs(ref_losses[0])
print(json.dumps({"workload": "S-FB (B=512, N=14541, d=200, input_dropout 0.4, bce, Adagrad lr 0.3)", "threads": threads,
                  "reference_ms_per_step": round(ref_ms, 2), "twin_ms_per_step": round(twin_ms, 2),
                  "twin_over_reference": round(twin_ms / ref_ms, 3), "first_step_loss_rel_diff": first,
                  "torch": torch.__version__, "numpy": np.__version__}))
