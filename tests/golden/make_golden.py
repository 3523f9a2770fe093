#!/usr/bin/env python3
"""Generate golden vectors for the hot path by running the REFERENCE itself (CPU, this container only).

Run (never on the GPU box -- /root/reference does not exist there):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 -B /root/repo/tests/golden/make_golden.py

The script imports the unmodified reference (openkge.model / openkge.trainer / openkge.dataset /
utils.optim), drives the classes on the hot path with small seeded inputs and stores inputs and
expected outputs as .npz fixtures next to this file.  Fixtures are DATA only (ids, tables, masks,
scores, losses, gradients, updated weights, ranks); no reference source is copied.

Vectors:
  g1_scores_*     sp_prefix_score / po_prefix_score, ComplEx + DistMult, dropout 0
  g1_triples_*    forward(subj, rel, obj) = triple_score of the encoded rows (Hadamard form)
  g2_loss_*       AddLossModule forward + (loss/normalizer).backward(): loss, all_outputs, dE, dR
                  (bce, bce + label smoothing, kl, one direction None, batch-shared candidates,
                  input_dropout 0.4 with the Bernoulli masks captured)
  g3_adagrad_*    three optimisation steps through utils.optim.OptimRegime (Adagrad, leaked eps)
  g4_collate_*    OneToNMentionRelationDataset_collate_func on packed toy prefix tables: 1-vs-all and batch-shared,
                  training and evaluation, with and without numpy-sampled fill-up negatives
  g6_dataset_*    OneToNMentionRelationDataset on tests/golden/toy_kg (our own synthetic 5-column TSV + id maps):
                  seen_prefixes (P,7) / packed seen_entities / all_splits_entities for train, valid, test; plus
                  shapes + sha256 of the same tensors for the reference's FB15k-237 valid/test files
  g9_unigram_*    UnigramPoolingComplexRelationModel (token-pooled embedder, sum/mean/max pooling, batch-norm):
                  AddLossModule forward + backward -> loss, outputs, token-table / batch-norm gradients, running
                  stats; eval-mode precompute_embeddings_from_tokens.  The reference class lacks the attribute
                  `entity_projection` it reads (model.py:789); the harness sets it to None on the instance.
  g10_fb15k237    the FIRST 512-prefix evaluation batch of FB15k-237 valid.txt as the reference's dataset + collate
                  produce it, scored / trained / ranked by the reference at the full BASELINE size (|E|=14543, d=200):
                  ids, label / filter coordinates and answer groups, a score slice, per-row score sums, loss,
                  gradient checksums, ranks.  The tables are regenerated from the seed by the test
                  (torch.manual_seed + the same constructor order), guarded by checksums.
  g5_ranks_*      OneToNMentionRelationDataset.compute_metrics (known answer + ties + mention groups)
  g8_checkpoint   the checkpoint dict Trainer.save writes (state_dict + OptimRegime.state_dict()) after two steps,
                  stored with torch.save (tensors / containers only), plus the third step's batch and result
  g7_traj_*       20 training steps, fixed batches -> loss curve and final tables, then compute_metrics on an
                  evaluation slice of the trained tables (scores, per-group ranks, MRR / MR / Hits)
  g12_variant_*   LookupComplexRelationModel with batch_norm / project_entity / normalize='norm' / l2_reg on
                  (model.py:463-479): loss, hook loss, outputs, every parameter's gradient, running stats, eval scores
  g14_distmult    configs[2] at its size: the reference's LookupDistmultRelationModel d = 512 on a real FB15k-237 batch that its
                  collate built with batch-shared sampled candidates (N = 10 000): loss, score slice, gradient checksums
  g13_valid_pass  reference-trained tables (rounded to bf16, stored 16-bit) + the reference's evaluation over ALL of
                  FB15k-237 valid.txt (its loader, collate, eval-mode AddLossModule, compute_metrics): per-group ranks, meters
  g15_epochs      three whole training passes of the reference over its own loader (test.txt as the training split, shuffle off,
                  dropout 0): per-step loss / normalizer, rows and positives per batch; then its evaluation over all of
                  valid.txt on the trained fp32 tables: meters
  g16_unigram_adagrad  the token-pooled model (sum pooling + batch-norm) through TWELVE optimisation steps of the reference's own
                  OptimRegime Adagrad (weight_decay 1e-10, leaked eps) on different batches: per-step loss, final token tables,
                  accumulators and batch-norm parameters -- in particular the rows NO batch names, which the reference's dense
                  optimizer moves every step by their weight-decay term alone (what okge_adagrad_lazy defers and replays)
  g11_traj_*      30 training steps at the BASELINE size on real FB15k-237 batches (reference dataset + collate), then
                  filtered ranks / MRR of the first valid.txt batch on the trained tables
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
assert os.path.isdir("/root/reference"), "golden vectors can only be generated where the reference is mounted"
if "/root/reference" not in sys.path:
    sys.path.insert(0, "/root/reference")

from openkge.dataset import EntityRelationDatasetMeta, OneToNMentionRelationDataset  # noqa: E402
from openkge.model import Models  # noqa: E402
from openkge.trainer import AddLossModule  # noqa: E402
from utils.optim import OptimRegime  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def meta(n_ent, n_rel):
    return EntityRelationDatasetMeta(
        entity_id_count_map={}, relation_id_count_map={}, entity_token_id_count_map={},
        relation_token_id_count_map={}, entity_id_to_tokens_map={}, relation_id_to_tokens_map={},
        entities_size=n_ent, relations_size=n_rel, min_entities_size=2, min_relations_size=2,
        entity_tokens_size=4, relation_tokens_size=4, max_length=1,
    )


def make_model(name, n_ent, n_rel, d, seed, input_dropout=0.0, init_std=0.1):
    torch.manual_seed(seed)
    m = getattr(Models, name)(entity_slot_size=d, input_dropout=input_dropout, init_std=init_std,
                              sparse=False, train_data=meta(n_ent, n_rel))
    return m


def rand_ids(rng, lo, hi, n):
    return torch.from_numpy(rng.integers(lo, hi, size=(n, 1)).astype(np.int32))


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in kw.items()})


# ----------------------------------------------------------------------------------------------
# G1: prefix scores
# ----------------------------------------------------------------------------------------------
def g1():
    for mname, tag in (("LookupComplexRelationModel", "complex"), ("LookupDistmultRelationModel", "distmult")):
        for (n_ent, n_rel, d, b, case) in ((66, 10, 16, 8, "tiny"), (301, 17, 200, 24, "d200"), (130, 9, 36, 5, "odd")):
            rng = np.random.default_rng(1000 + d)
            m = make_model(mname, n_ent, n_rel, d, seed=11 + d)
            m.eval()
            subj, rel_sp = rand_ids(rng, 2, n_ent, b), rand_ids(rng, 2, n_rel, b)
            rel_po, obj = rand_ids(rng, 2, n_rel, b), rand_ids(rng, 2, n_ent, b)
            cand = torch.from_numpy(rng.permutation(np.arange(2, n_ent))[: max(7, (n_ent - 2) // 2)].astype(np.int32))
            with torch.no_grad():
                sp_all = m.sp_prefix_score(subj, rel_sp)                 # all objects (ids 2..)
                po_all = m.po_prefix_score(rel_po, obj)                  # all subjects
                pre = m.precompute_batch_shared_inputs(cand.view(-1))
                sp_c = m.sp_prefix_score(subj, rel_sp, pre)
                po_c = m.po_prefix_score(rel_po, obj, pre)
            save(f"g1_scores_{tag}_{case}",
                 E=npy(m.entity_embedding.weight), R=npy(m.relation_embedding.weight),
                 subj=npy(subj), rel_sp=npy(rel_sp), rel_po=npy(rel_po), obj=npy(obj), cand=npy(cand),
                 sp_all=npy(sp_all), po_all=npy(po_all), sp_cand=npy(sp_c), po_cand=npy(po_c))


def g1_triples():
    for mname, tag in (("LookupComplexRelationModel", "complex"), ("LookupDistmultRelationModel", "distmult")):
        for (n_ent, n_rel, d, b, case) in ((66, 10, 16, 9, "tiny"), (301, 17, 200, 33, "d200")):
            rng = np.random.default_rng(4000 + d)
            m = make_model(mname, n_ent, n_rel, d, seed=41 + d)
            m.eval()
            subj, rel, obj = rand_ids(rng, 2, n_ent, b), rand_ids(rng, 2, n_rel, b), rand_ids(rng, 2, n_ent, b)
            with torch.no_grad():
                out = m(subj, rel, obj)
            save(f"g1_triples_{tag}_{case}", E=npy(m.entity_embedding.weight), R=npy(m.relation_embedding.weight),
                 subj=npy(subj), rel=npy(rel), obj=npy(obj), scores=npy(out))


# ----------------------------------------------------------------------------------------------
# G2: loss + gradients through AddLossModule
# ----------------------------------------------------------------------------------------------
def dense_labels(rng, B, N, max_pos=4):
    y = np.zeros((B, N), dtype=np.float32)
    for b in range(B):
        k = int(rng.integers(1, max_pos + 1))
        y[b, rng.choice(N, size=k, replace=False)] = 1.0
    return y


def run_loss(mname, n_ent, n_rel, d, b_po, b_sp, loss_kind, smoothing, cand_mode, seed, input_dropout=0.0):
    rng = np.random.default_rng(seed)
    m = make_model(mname, n_ent, n_rel, d, seed=seed, input_dropout=input_dropout)
    m.train()
    if cand_mode == "all":
        cand = torch.arange(n_ent)[2:].int().unsqueeze(1)
    else:
        ids = rng.permutation(np.arange(2, n_ent))[: int(cand_mode)]
        cand = torch.from_numpy(ids.astype(np.int32)).unsqueeze(1)
    N = cand.shape[0]
    inputs = []
    po = sp = None
    if b_po > 0:
        po = (rand_ids(rng, 2, n_rel, b_po), rand_ids(rng, 2, n_ent, b_po))
    if b_sp > 0:
        sp = (rand_ids(rng, 2, n_ent, b_sp), rand_ids(rng, 2, n_rel, b_sp))
    inputs = [po, sp]
    B = b_po + b_sp
    y = dense_labels(rng, B, N)
    labels = torch.from_numpy(y.copy())
    if loss_kind == "bce":
        loss = torch.nn.BCEWithLogitsLoss(reduction="sum")
    else:
        loss = torch.nn.KLDivLoss(reduction="sum")
    mod = AddLossModule(m, loss, bce_label_smoothing=smoothing)
    mod.train()
    masks = {}
    if input_dropout > 0:
        # Capture the Bernoulli keep-masks the reference is about to draw: same generator state, same op
        # sequence (candidates, then obj rows of po, then subj rows of sp; relation dropout is 0).
        st = torch.get_rng_state()
        torch.manual_seed(seed + 77)
        masks["mask_cand"] = (torch.nn.functional.dropout(torch.ones(N, d), p=input_dropout, training=True) > 0)
        if b_po > 0:
            masks["mask_po_ent"] = (torch.nn.functional.dropout(torch.ones(b_po, d), p=input_dropout, training=True) > 0)
        if b_sp > 0:
            masks["mask_sp_ent"] = (torch.nn.functional.dropout(torch.ones(b_sp, d), p=input_dropout, training=True) > 0)
        torch.set_rng_state(st)
        torch.manual_seed(seed + 77)
    out_loss, hook, outputs = mod(inputs=inputs, labels=labels, use_batch_shared_entities=(cand_mode != "all"),
                                  batch_shared_entities=cand, epoch=1,
                                  input_style_triple_or_prefix="right_and_left_prefix")
    assert hook is None
    normalizer = float(B * N)
    (out_loss.sum() / normalizer).backward()
    kw = dict(E=npy(m.entity_embedding.weight), R=npy(m.relation_embedding.weight),
              cand=npy(cand), labels=y, loss=np.float64(out_loss.item()), outputs=npy(outputs),
              dE=npy(m.entity_embedding.weight.grad), dR=npy(m.relation_embedding.weight.grad),
              normalizer=np.float64(normalizer), smoothing=np.float64(smoothing),
              input_dropout=np.float64(input_dropout))
    if po is not None:
        kw.update(po_rel=npy(po[0]), po_obj=npy(po[1]))
    if sp is not None:
        kw.update(sp_subj=npy(sp[0]), sp_rel=npy(sp[1]))
    for k, v in masks.items():
        kw[k] = npy(v).astype(np.uint8)
    return kw


def g2():
    C, D = "LookupComplexRelationModel", "LookupDistmultRelationModel"
    cases = [
        # name, model, n_ent, n_rel, d, b_po, b_sp, loss, smoothing, cand, seed, dropout
        ("complex_bce_all", C, 66, 10, 16, 6, 7, "bce", 0.0, "all", 21, 0.0),
        ("complex_bce_smooth_all", C, 66, 10, 16, 6, 7, "bce", 0.1, "all", 22, 0.0),
        ("complex_kl_all", C, 66, 10, 16, 6, 7, "kl", 0.0, "all", 23, 0.0),
        ("complex_bce_po_only", C, 66, 10, 16, 9, 0, "bce", 0.0, "all", 24, 0.0),
        ("complex_bce_sp_only", C, 66, 10, 16, 0, 9, "bce", 0.0, "all", 25, 0.0),
        ("complex_bce_shared", C, 150, 12, 24, 10, 11, "bce", 0.0, "70", 26, 0.0),
        ("complex_kl_shared", C, 150, 12, 24, 10, 11, "kl", 0.0, "70", 27, 0.0),
        ("complex_bce_dropout_all", C, 66, 10, 16, 6, 7, "bce", 0.0, "all", 28, 0.4),
        ("complex_bce_d200", C, 400, 20, 200, 40, 33, "bce", 0.0, "all", 29, 0.0),
        ("complex_bce_d200_dropout", C, 400, 20, 200, 40, 33, "bce", 0.0, "all", 30, 0.4),
        ("distmult_bce_all", D, 66, 10, 16, 6, 7, "bce", 0.0, "all", 31, 0.0),
        ("distmult_kl_all", D, 66, 10, 16, 6, 7, "kl", 0.0, "all", 32, 0.0),
        ("distmult_bce_shared_d64", D, 300, 12, 64, 20, 21, "bce", 0.0, "128", 33, 0.0),
        ("distmult_bce_dropout_shared", D, 300, 12, 64, 20, 21, "bce", 0.0, "128", 34, 0.4),
    ]
    for (name, mname, n_ent, n_rel, d, b_po, b_sp, lk, sm, cm, seed, dp) in cases:
        kw = run_loss(mname, n_ent, n_rel, d, b_po, b_sp, lk, sm, cm, seed, dp)
        kw["loss_kind"] = np.array(lk)
        kw["model"] = np.array("complex" if mname == C else "distmult")
        save("g2_loss_" + name, **kw)


# ----------------------------------------------------------------------------------------------
# G3: Adagrad through OptimRegime (eps leaked from the Adam shell)
# ----------------------------------------------------------------------------------------------
def g3():
    for mname, tag in (("LookupComplexRelationModel", "complex"), ("LookupDistmultRelationModel", "distmult")):
        n_ent, n_rel, d, b = 80, 9, 20, 12
        seed = 41
        rng = np.random.default_rng(seed)
        m = make_model(mname, n_ent, n_rel, d, seed=seed)
        m.train()
        args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": 0.3, "weight_decay": 1.0e-10},
                "lr_scheduler_config": None}
        opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
        mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
        mod.train()
        E0, R0 = npy(m.entity_embedding.weight).copy(), npy(m.relation_embedding.weight).copy()
        cand = torch.arange(n_ent)[2:].int().unsqueeze(1)
        N = cand.shape[0]
        steps = []
        group_info = None
        for step in range(3):
            po = (rand_ids(rng, 2, n_rel, b), rand_ids(rng, 2, n_ent, b))
            sp = (rand_ids(rng, 2, n_ent, b), rand_ids(rng, 2, n_rel, b))
            y = dense_labels(rng, 2 * b, N)
            for o in opts:
                o.update(1, step + 1)
            for o in opts:
                o.zero_grad()
            loss, _, _ = mod(inputs=[po, sp], labels=torch.from_numpy(y.copy()), use_batch_shared_entities=False,
                             batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
            (loss.sum() / float(2 * b * N)).backward()
            for o in opts:
                o.step()
            g = opts[0].optimizer.param_groups[0]
            group_info = {k: g[k] for k in ("lr", "eps", "weight_decay", "lr_decay", "initial_accumulator_value")}
            st = opts[0].optimizer.state
            steps.append(dict(po_rel=npy(po[0]), po_obj=npy(po[1]), sp_subj=npy(sp[0]), sp_rel=npy(sp[1]), labels=y,
                              loss=np.float64(loss.item()),
                              E=npy(m.entity_embedding.weight).copy(), R=npy(m.relation_embedding.weight).copy(),
                              sumE=npy(st[m.entity_embedding.weight]["sum"]).copy(),
                              sumR=npy(st[m.relation_embedding.weight]["sum"]).copy()))
        print("effective Adagrad group:", group_info, type(opts[0].optimizer).__name__)
        kw = dict(E0=E0, R0=R0, cand=npy(cand), nsteps=np.int64(3))
        for k, v in group_info.items():
            kw["opt_" + k] = np.float64(v)
        for i, s in enumerate(steps):
            for k, v in s.items():
                kw[f"s{i}_{k}"] = v
        save(f"g3_adagrad_{tag}", **kw)


# ----------------------------------------------------------------------------------------------
# G5: filtered ranks
# ----------------------------------------------------------------------------------------------
def pack_groups(label_ids):
    """list[B] of list[G_b] of int tensors -> (row_ptr[B+1], grp_ptr[G+1], ids[M])"""
    row_ptr, grp_ptr, ids = [0], [0], []
    for groups in label_ids:
        for g in groups:
            ids.extend(int(x) for x in g.tolist())
            grp_ptr.append(len(ids))
        row_ptr.append(len(grp_ptr) - 1)
    return np.asarray(row_ptr, np.int64), np.asarray(grp_ptr, np.int64), np.asarray(ids, np.int32)


def per_group_ranks(filter_mask, label_ids, predictions):
    """The rank rule of compute_metrics, evaluated group by group with the reference's own tensor ops
    (masked_fill_, <, ==, sum, //) so the integer ranks themselves can be stored (the reference only
    returns their meters)."""
    ranks = []
    for f, groups, p in zip(filter_mask, label_ids, predictions):
        true = torch.Tensor([p[g.long()].max(0)[0] for g in groups])
        rep = p.unsqueeze(0).repeat(len(groups), 1)
        rep.masked_fill_(f.unsqueeze(0).repeat(len(groups), 1), -1e8)
        fp = (true.view(len(groups), -1) < rep).long().sum(1)
        eq = (true.view(len(groups), -1) == rep).long().sum(1)
        ranks.extend((fp + eq // 2).tolist())
    return np.asarray(ranks, np.int64)


def g5():
    # known answer
    pred = torch.tensor([[0.5, 0.9, 0.9, 0.2, 0.3, 0.9, 0.7, 5.0]])
    filt = torch.zeros(1, 8, dtype=torch.bool)
    filt[0, [1, 4, 6, 7]] = True
    lids = [[torch.IntTensor([1]), torch.IntTensor([4, 6])]]
    res = OneToNMentionRelationDataset.compute_metrics(filt, lids, pred)
    rp, gp, ids = pack_groups(lids)
    save("g5_ranks_known", pred=npy(pred), filt=npy(filt).astype(np.uint8), row_ptr=rp, grp_ptr=gp, ids=ids,
         ranks=per_group_ranks(filt, lids, pred),
         **{"m_" + k: np.float64(v.avg) for k, v in res.items()},
         **{"c_" + k: np.float64(v.count) for k, v in res.items()})
    # random with forced ties and multi-mention groups
    for case, (B, N, seed) in {"rand_small": (9, 50, 5), "rand_wide": (6, 1000, 6), "rand_ties": (12, 200, 7)}.items():
        rng = np.random.default_rng(seed)
        p = rng.standard_normal((B, N)).astype(np.float32)
        if case == "rand_ties":
            p = np.round(p * 2) / 2  # many exact ties
        filt = np.zeros((B, N), dtype=bool)
        lids = []
        for b in range(B):
            ng = int(rng.integers(1, 5))
            groups = []
            for _ in range(ng):
                sz = int(rng.integers(1, 4))
                g = rng.choice(N, size=sz, replace=False)
                groups.append(torch.IntTensor(g.astype(np.int32)))
                filt[b, g] = True
            extra = rng.choice(N, size=int(rng.integers(0, 6)), replace=False)  # answers from other splits
            filt[b, extra] = True
            lids.append(groups)
        pt, ft = torch.from_numpy(p), torch.from_numpy(filt)
        res = OneToNMentionRelationDataset.compute_metrics(ft, lids, pt)
        rp, gp, ids = pack_groups(lids)
        save("g5_ranks_" + case, pred=p, filt=filt.astype(np.uint8), row_ptr=rp, grp_ptr=gp, ids=ids,
             ranks=per_group_ranks(ft, lids, pt),
             **{"m_" + k: np.float64(v.avg) for k, v in res.items()},
             **{"c_" + k: np.float64(v.count) for k, v in res.items()})


# ----------------------------------------------------------------------------------------------
# G7: short trajectory
# ----------------------------------------------------------------------------------------------
def g7():
    mname = "LookupComplexRelationModel"
    n_ent, n_rel, d, b, nsteps = 120, 11, 32, 16, 20
    seed = 51
    rng = np.random.default_rng(seed)
    m = make_model(mname, n_ent, n_rel, d, seed=seed)
    m.train()
    args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": 0.3, "weight_decay": 1.0e-10},
            "lr_scheduler_config": None}
    opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    mod.train()
    E0, R0 = npy(m.entity_embedding.weight).copy(), npy(m.relation_embedding.weight).copy()
    cand = torch.arange(n_ent)[2:].int().unsqueeze(1)
    N = cand.shape[0]
    # four fixed batches cycled
    batches = []
    for _ in range(4):
        batches.append((rand_ids(rng, 2, n_rel, b), rand_ids(rng, 2, n_ent, b), rand_ids(rng, 2, n_ent, b),
                        rand_ids(rng, 2, n_rel, b), dense_labels(rng, 2 * b, N)))
    losses = []
    for step in range(nsteps):
        po_rel, po_obj, sp_subj, sp_rel, y = batches[step % 4]
        for o in opts:
            o.update(1, step + 1)
            o.zero_grad()
        loss, _, _ = mod(inputs=[(po_rel, po_obj), (sp_subj, sp_rel)], labels=torch.from_numpy(y.copy()),
                         use_batch_shared_entities=False, batch_shared_entities=cand, epoch=1,
                         input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / float(2 * b * N)).backward()
        for o in opts:
            o.step()
        losses.append(loss.item() / float(2 * b * N))
    kw = dict(E0=E0, R0=R0, cand=npy(cand), losses=np.asarray(losses, np.float64),
              E=npy(m.entity_embedding.weight), R=npy(m.relation_embedding.weight), nsteps=np.int64(nsteps))
    # evaluation slice on the TRAINED tables (trainer.py:258-272 -> dataset.py:423-453): batch 0's prefixes, their
    # training answers as answer groups (some two-id mention groups), a filter that also hides a few other columns
    m.eval()
    po_rel, po_obj, sp_subj, sp_rel, y = batches[0]
    with torch.no_grad():
        scores = torch.cat([m.po_prefix_score(po_rel, po_obj), m.sp_prefix_score(sp_subj, sp_rel)], 0)
    filt = torch.from_numpy(y > 0)
    label_ids = []
    for r in range(y.shape[0]):
        pos = np.flatnonzero(y[r]).astype(np.int32)
        extra = rng.choice(N, size=3, replace=False)
        filt[r, torch.from_numpy(extra).long()] = True
        groups = [torch.IntTensor(pos[i:i + 2]) if (i % 3 == 0 and i + 1 < len(pos)) else torch.IntTensor(pos[i:i + 1])
                  for i in range(len(pos))]
        label_ids.append(groups)
    res = OneToNMentionRelationDataset.compute_metrics(filt, label_ids, scores.clone())
    rp, gp, gids = pack_groups(label_ids)
    kw.update(eval_scores=npy(scores), eval_filter=npy(filt).astype(np.uint8), eval_row_ptr=rp, eval_grp_ptr=gp, eval_ids=gids,
              eval_ranks=per_group_ranks(filt, label_ids, scores.clone()),
              **{"eval_m_" + k: np.float64(v.avg) for k, v in res.items()})
    for i, (a, b_, c, e, y) in enumerate(batches):
        kw.update({f"b{i}_po_rel": npy(a), f"b{i}_po_obj": npy(b_), f"b{i}_sp_subj": npy(c), f"b{i}_sp_rel": npy(e),
                   f"b{i}_labels": y})
    save("g7_traj_complex", **kw)


# ----------------------------------------------------------------------------------------------
# G4: the batch producer (collate) on packed prefix tables
# ----------------------------------------------------------------------------------------------
def g4():
    import numpy
    from openkge.dataset import OneToNMentionRelationDataset_collate_func as collate
    from utils.misc import pack_list_of_lists

    rng = np.random.default_rng(44)
    n_ent, off = 60, 2
    seen, allsp, rows = [], [], []
    for i in range(40):
        slot = 0 if rng.random() < 0.5 else 2
        k = int(rng.integers(1, 5))
        groups = [rng.integers(off, n_ent, size=int(rng.integers(1, 4))).tolist() for _ in range(k)]   # mentions; repeats allowed
        packed = pack_list_of_lists(groups)
        flat = sorted({e for g in groups for e in g})
        extra = rng.integers(off, n_ent, size=int(rng.integers(0, 4))).tolist()
        everything = list(dict.fromkeys(flat + extra))                       # unique, as the merged splits are (a set)
        rng.shuffle(everything)
        rows.append([int(rng.integers(2, 9)), int(rng.integers(2, n_ent)), len(seen), len(seen) + len(packed),
                     len(allsp), len(allsp) + len(everything), slot])
        seen += packed
        allsp += everything
    seen_t, all_t, rows_t = torch.IntTensor(seen), torch.IntTensor(allsp), torch.IntTensor(rows)
    kw = dict(seen=np.asarray(seen, np.int32), all_splits=np.asarray(allsp, np.int32), prefixes=np.asarray(rows, np.int32),
              n_ent=np.int64(n_ent), offset=np.int64(off))
    batches = [list(range(0, 12)), list(range(12, 40, 3)), [5], [i for i in range(40) if rows[i][6] == 0][:6]]
    ncase = 0
    for shared, min_size in ((False, 0), (True, 0), (True, 48), (True, -1)):
        for training in (True, False):
            for bi, idx in enumerate(batches):
                numpy.random.seed(7 + bi)
                out = collate(use_batch_shared_entities=shared, sp_po__batch=[rows_t[i] for i in idx],
                              entity_vocab_size=n_ent, entity_vocab_offset=off, is_training_data=training,
                              this_split_entities_list=seen_t, all_splits_entities_tensor=all_t,
                              min_size_batch_labels=min_size)
                inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = out
                tag = f"c{ncase}_"
                kw[tag + "rows"] = np.asarray(idx, np.int64)
                kw[tag + "cfg"] = np.asarray([int(shared), min_size, int(training)], np.int64)
                for name, part in zip(("po", "sp"), inputs):
                    kw[tag + name] = (np.zeros((0, 2), np.int32) if part is None
                                      else np.concatenate([npy(part[0]), npy(part[1])], axis=1))
                kw[tag + "labels"] = npy(labels.nonzero()).astype(np.int32)                 # (nnz, 2) row, col
                kw[tag + "shape"] = np.asarray(labels.shape, np.int64)
                kw[tag + "norm"] = np.asarray([norm_loss, norm_metric], np.float64)
                kw[tag + "cand"] = npy(cand).reshape(-1)
                if not training:
                    kw[tag + "filter"] = npy(filt.nonzero()).astype(np.int32)
                    gp, ids, rp = [0], [], [0]
                    for row_groups in label_ids:
                        for g in row_groups:
                            ids += npy(g).reshape(-1).tolist()
                            gp.append(len(ids))
                        rp.append(len(gp) - 1)
                    kw[tag + "row_ptr"], kw[tag + "grp_ptr"] = np.asarray(rp, np.int64), np.asarray(gp, np.int64)
                    kw[tag + "ids"] = np.asarray(ids, np.int32)
                ncase += 1
    kw["n_cases"] = np.int64(ncase)
    save("g4_collate_toy", **kw)


# ----------------------------------------------------------------------------------------------
# G6: on-disk formats -> dataset tensors
# ----------------------------------------------------------------------------------------------
def build_datasets(src_dir, files, max_size_prefix_label=-1):
    import hashlib
    import shutil
    import tempfile
    scratch = tempfile.mkdtemp(prefix="okge_g6_")
    for f in os.listdir(src_dir):
        if f.endswith(".txt"):
            shutil.copy(os.path.join(src_dir, f), scratch)
    out = {}
    try:
        ds = {}
        for split, fname in files.items():
            ds[split] = OneToNMentionRelationDataset(
                dataset_dir=scratch, input_file=fname, is_training_data=(split == "train"), batch_size=8,
                copy_data_to_dev_shm=False, max_size_prefix_label=max_size_prefix_label,
                entity_id_tokens_ids_map_file="absent.txt", relation_id_tokens_ids_map_file="absent.txt")
        ds["valid"].merge_all_splits_triples(dataset_dir=scratch, train_input_file=files["train"],
                                             valid_input_file=files["valid"], test_input_file=files["test"])
        for split in ds:
            ds[split].create_data_tensors(dataset_dir=scratch, train_input_file=files["train"],
                                          valid_input_file=files["valid"], test_input_file=files["test"])
            out[split] = (npy(ds[split].seen_prefixes_tensor), npy(ds[split].seen_entities_tensor),
                          npy(ds[split].all_splits_entities_tensor), ds[split].entity_vocab_size)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return out


def g6():
    import hashlib
    toy = os.path.join(OUT, "toy_kg")
    files = {"train": "train.txt", "valid": "valid.txt", "test": "test.txt"}
    kw = {}
    for split, (pref, seen, allsp, n_ent) in build_datasets(toy, files).items():
        kw.update({f"{split}_prefixes": pref, f"{split}_seen": seen, f"{split}_all": allsp})
        kw["entity_vocab_size"] = np.int64(n_ent)
    # training split with max_size_prefix_label = 3: long answer lists are cut into several prefix rows.  The
    # reference over-allocates these tensors (its counting loop runs one chunk too far, dataset.py:628-640) and
    # leaves the tail uninitialised; consumers of this fixture compare only the rows / ids they produce themselves.
    pref, seen, _, _ = build_datasets(toy, files, max_size_prefix_label=3)["train"]
    kw.update(train3_prefixes=pref, train3_seen=seen)
    save("g6_dataset_toy", **kw)
    # FB15k-237 as shipped with the reference (train split absent: test.txt stands in for it); hashes only
    fb = "/root/reference/data/fb15k237/mapped_to_ids"
    kw = {}
    for split, (pref, seen, allsp, n_ent) in build_datasets(fb, {"train": "test.txt", "valid": "valid.txt", "test": "test.txt"}).items():
        if split != "valid":       # "test" shares its tensor cache file with the stand-in training split: skip
            continue
        sort_slices = np.concatenate([np.sort(allsp[a:b]) for a, b in sorted({(int(r[4]), int(r[5])) for r in pref})]) if len(pref) else allsp
        kw.update({f"{split}_shapes": np.asarray([pref.shape[0], seen.shape[0], allsp.shape[0]], np.int64),
                   f"{split}_sha_prefixes": hashlib.sha256(np.ascontiguousarray(pref).tobytes()).hexdigest(),
                   f"{split}_sha_seen": hashlib.sha256(np.ascontiguousarray(seen).tobytes()).hexdigest(),
                   # the order inside one prefix's all-splits slice is a Python set's iteration order: hash the
                   # slices sorted (each distinct slice once, in table order)
                   f"{split}_sha_all_sorted": hashlib.sha256(np.ascontiguousarray(sort_slices).tobytes()).hexdigest()})
        kw["entity_vocab_size"] = np.int64(n_ent)
    save("g6_dataset_fb15k237_hashes", **kw)


# ----------------------------------------------------------------------------------------------
# G9: token-pooled embedder
# ----------------------------------------------------------------------------------------------
def g9():
    cases = [
        # name, pool, normalize, n_ent, n_rel, d, b_po, b_sp, n_cand ("all" or int), dropout-mask capture
        ("sum_bn_all", "sum", "batchnorm", 70, 9, 16, 6, 7, "all"),
        ("sum_bn_shared", "sum", "batchnorm", 160, 12, 24, 10, 11, 64),
        ("mean_bn_shared", "mean", "batchnorm", 160, 12, 24, 10, 11, 64),
        ("max_none_shared", "max", None, 160, 12, 24, 10, 11, 64),
        ("sum_none_all", "sum", None, 70, 9, 16, 6, 0, "all"),
    ]
    for ci, (name, pool, normalize, n_ent, n_rel, d, b_po, b_sp, n_cand) in enumerate(cases):
        rng = np.random.default_rng(900 + ci)
        L, vt_e, vt_r = 6, 40, 15
        def token_map(n, vocab):
            out = [[1], [1]]                                 # reserved ids 0, 1 (dataset.py:200-201)
            for _ in range(2, n):
                k = int(rng.integers(1, L + 3))              # some longer than max_length: the tail is kept
                out.append([2] + rng.integers(4, vocab, size=k).tolist() + [3])
            return tuple(out)
        md = meta(n_ent, n_rel)
        md.entity_id_to_tokens_map, md.relation_id_to_tokens_map = token_map(n_ent, vt_e), token_map(n_rel, vt_r)
        md.entity_tokens_size, md.relation_tokens_size, md.max_length = vt_e, vt_r, (L, L)
        torch.manual_seed(900 + ci)
        m = Models.UnigramPoolingComplexRelationModel(entity_slot_size=d, relation_slot_size=d, train_data=md, pool=pool,
                                                      normalize=normalize, dropout=0.0, sparse=False, init_std=0.3)
        m.entity_projection = None                           # harness shim, see the module docstring
        m.train()
        if n_cand == "all":
            cand = torch.arange(n_ent)[2:].int().unsqueeze(1)
        else:
            cand = torch.from_numpy(rng.permutation(np.arange(2, n_ent))[:n_cand].astype(np.int32)).unsqueeze(1)
        N = cand.shape[0]
        po = (rand_ids(rng, 2, n_rel, b_po), rand_ids(rng, 2, n_ent, b_po)) if b_po else None
        sp = (rand_ids(rng, 2, n_ent, b_sp), rand_ids(rng, 2, n_rel, b_sp)) if b_sp else None
        B = b_po + b_sp
        y = dense_labels(rng, B, N)
        kw = dict(We=npy(m.entity_embedding.weight).copy(), Wr=npy(m.relation_embedding.weight).copy(),
                  ent_tokens=npy(m.entity_token_ids).astype(np.int32), rel_tokens=npy(m.relation_token_ids).astype(np.int32),
                  cand=npy(cand), labels=y, pool=pool, normalize=str(normalize))
        if normalize == "batchnorm":
            kw.update(bn_e_w=npy(m.entity_batchnorm.weight).copy(), bn_e_b=npy(m.entity_batchnorm.bias).copy(),
                      bn_r_w=npy(m.relation_batchnorm.weight).copy(), bn_r_b=npy(m.relation_batchnorm.bias).copy())
        mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), bce_label_smoothing=0.0)
        mod.train()
        loss, hook, outputs = mod(inputs=[po, sp], labels=torch.from_numpy(y.copy()),
                                  use_batch_shared_entities=(n_cand != "all"), batch_shared_entities=cand, epoch=1,
                                  input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / float(B * N)).backward()
        kw.update(loss=np.float64(loss.item()), outputs=npy(outputs), normalizer=np.float64(B * N),
                  dWe=npy(m.entity_embedding.weight.grad), dWr=npy(m.relation_embedding.weight.grad))
        if po is not None:
            kw.update(po_rel=npy(po[0]), po_obj=npy(po[1]))
        if sp is not None:
            kw.update(sp_subj=npy(sp[0]), sp_rel=npy(sp[1]))
        if normalize == "batchnorm":
            kw.update(d_bn_e_w=npy(m.entity_batchnorm.weight.grad), d_bn_e_b=npy(m.entity_batchnorm.bias.grad),
                      d_bn_r_w=npy(m.relation_batchnorm.weight.grad), d_bn_r_b=npy(m.relation_batchnorm.bias.grad),
                      run_e_mean=npy(m.entity_batchnorm.running_mean), run_e_var=npy(m.entity_batchnorm.running_var),
                      run_r_mean=npy(m.relation_batchnorm.running_mean), run_r_var=npy(m.relation_batchnorm.running_var))
        # evaluation: full tables from tokens with the running statistics (model.py:670-712), then prefix scores
        m.eval()
        with torch.no_grad():
            m.precompute_embeddings_from_tokens()
            kw.update(E_eval=npy(m.entity_embedding_from_tokens), R_eval=npy(m.relations_embedding_from_tokens))
            if sp is not None:
                kw["sp_all_eval"] = npy(m.sp_prefix_score(sp[0], sp[1]))
            if po is not None:
                kw["po_all_eval"] = npy(m.po_prefix_score(po[0], po[1]))
        save(f"g9_unigram_{name}", **kw)


# ----------------------------------------------------------------------------------------------
# G10: full-size FB15k-237 batch through the reference
# ----------------------------------------------------------------------------------------------
def g10():
    import shutil
    import tempfile
    from openkge.dataset import OneToNMentionRelationDataset_collate_func as collate
    fb = "/root/reference/data/fb15k237/mapped_to_ids"
    scratch = tempfile.mkdtemp(prefix="okge_g10_")
    try:
        for f in os.listdir(fb):
            shutil.copy(os.path.join(fb, f), scratch)
        files = {"train": "test.txt", "valid": "valid.txt", "test": "test.txt"}     # train split absent upstream
        ds = {}
        for split in ("train", "valid"):
            ds[split] = OneToNMentionRelationDataset(dataset_dir=scratch, input_file=files[split],
                                                     is_training_data=(split == "train"), batch_size=512, copy_data_to_dev_shm=False)
        ds["valid"].merge_all_splits_triples(dataset_dir=scratch, train_input_file=files["train"],
                                             valid_input_file=files["valid"], test_input_file=files["test"])
        ds["valid"].create_data_tensors(dataset_dir=scratch, train_input_file=files["train"],
                                        valid_input_file=files["valid"], test_input_file=files["test"])
        v = ds["valid"]
        n_ent, n_rel = v.entity_vocab_size, v.relations_size
        # 256 po rows (slot 0) and 256 sp rows (slot 2), taken from the start of each block of the prefix table
        pref = v.seen_prefixes_tensor
        slot = pref[:, 6]
        rows = torch.cat([torch.nonzero(slot == 0).view(-1)[:256], torch.nonzero(slot == 2).view(-1)[:256]])
        out = collate(use_batch_shared_entities=False, sp_po__batch=[pref[i] for i in rows.tolist()],
                      entity_vocab_size=n_ent, entity_vocab_offset=2, is_training_data=False,
                      this_split_entities_list=v.seen_entities_tensor, all_splits_entities_tensor=v.all_splits_entities_tensor,
                      min_size_batch_labels=0)
        inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = out
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    seed, d = 2024, 200
    m = make_model("LookupComplexRelationModel", n_ent, n_rel, d, seed=seed, init_std=0.1)
    E, R = npy(m.entity_embedding.weight), npy(m.relation_embedding.weight)
    m.eval()
    with torch.no_grad():
        po = m.po_prefix_score(inputs[0][0], inputs[0][1])
        sp = m.sp_prefix_score(inputs[1][0], inputs[1][1])
        scores = torch.cat([po, sp], 0)
    metrics = OneToNMentionRelationDataset.compute_metrics(filter_mask=filt, label_ids=label_ids, predictions=scores.clone())
    # per-group ranks with the reference's own rule on its own scores (dataset.py:436-446)
    ranks = []
    masked_all = scores.clone()
    for b in range(scores.shape[0]):
        masked = scores[b].clone()
        masked[filt[b]] = -1e8
        for gidx in label_ids[b]:
            true = scores[b][gidx.long()].max()
            ranks.append(int((masked > true).sum()) + int((masked == true).sum()) // 2)
    del masked_all
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    mod.train()
    loss, _, outputs = mod(inputs=list(inputs), labels=labels.clone(), use_batch_shared_entities=False,
                           batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
    (loss.sum() / float(norm_loss)).backward()
    dE, dR = npy(m.entity_embedding.weight.grad), npy(m.relation_embedding.weight.grad)
    rp, gp, ids = pack_groups(label_ids)
    fnz = npy(filt.nonzero()).astype(np.int32)
    lnz = npy(labels.nonzero()).astype(np.int32)
    save("g10_fb15k237_batch",
         seed=np.int64(seed), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), d=np.int64(d),
         table_check=np.asarray([E.sum(dtype=np.float64), np.abs(E).sum(dtype=np.float64), R.sum(dtype=np.float64),
                                 float(E[5, 7]), float(E[-1, -1]), float(R[3, 4])], np.float64),
         po_rel=npy(inputs[0][0]), po_obj=npy(inputs[0][1]), sp_subj=npy(inputs[1][0]), sp_rel=npy(inputs[1][1]),
         labels=lnz, filter=fnz, row_ptr=rp, grp_ptr=gp, ids=ids, prefix_rows=npy(rows).astype(np.int64),
         score_slice=npy(scores[192:320, 1000:1128]), score_row_sum=npy(scores.double().sum(1)),
         score_row_absmax=npy(scores.abs().max(1).values), ranks=np.asarray(ranks, np.int64),
         mrr=np.float64(metrics["mrr"].avg) if "mrr" in metrics else np.float64(-1),
         loss=np.float64(loss.item()), normalizer=np.float64(norm_loss), n_labels=np.float64(norm_metric),
         dE_row_sum=dE.astype(np.float64).sum(1), dE_abs_sum=np.float64(np.abs(dE).sum(dtype=np.float64)),
         dE_slice=dE[2:66, :16].copy(), dR=dR)


# ----------------------------------------------------------------------------------------------
# G11: training trajectory at the BASELINE size on real FB15k-237 batches, then filtered MRR on the trained tables
# ----------------------------------------------------------------------------------------------
def g11():
    """30 optimisation steps of the reference (LookupComplexRelationModel d=200, AddLossModule bce, OptimRegime Adagrad
    lr 0.3 wd 1e-10, dropout 0) on 512-prefix batches that its own dataset class + collate produce from test.txt (the
    stand-in training split; train.txt is absent upstream), then Trainer.evaluate's arithmetic (eval-mode prefix scores
    -> compute_metrics) on the first 512-prefix batch of valid.txt: loss curve, table checksums + slices, per-group
    ranks, MRR / MR / Hits."""
    import shutil
    import tempfile
    from openkge.dataset import OneToNMentionRelationDataset_collate_func as collate
    fb = "/root/reference/data/fb15k237/mapped_to_ids"
    scratch = tempfile.mkdtemp(prefix="okge_g11_")
    nsteps, half = 30, 256
    try:
        for f in os.listdir(fb):
            shutil.copy(os.path.join(fb, f), scratch)
        files = {"train": "test.txt", "valid": "valid.txt", "test": "test.txt"}
        ds = {}
        for split in ("train", "valid"):
            ds[split] = OneToNMentionRelationDataset(dataset_dir=scratch, input_file=files[split],
                                                     is_training_data=(split == "train"), batch_size=512, copy_data_to_dev_shm=False)
        for split in ("train", "valid"):
            ds[split].merge_all_splits_triples(dataset_dir=scratch, train_input_file=files["train"],
                                               valid_input_file=files["valid"], test_input_file=files["test"])
            ds[split].create_data_tensors(dataset_dir=scratch, train_input_file=files["train"],
                                          valid_input_file=files["valid"], test_input_file=files["test"])
        tr, v = ds["train"], ds["valid"]
        n_ent, n_rel = v.entity_vocab_size, v.relations_size

        def batch_of(dset, step, training):
            pref = dset.seen_prefixes_tensor
            slot = pref[:, 6]
            po_rows, sp_rows = torch.nonzero(slot == 0).view(-1), torch.nonzero(slot == 2).view(-1)
            rows = torch.cat([po_rows[step * half:(step + 1) * half], sp_rows[step * half:(step + 1) * half]])
            assert len(rows) == 2 * half
            return collate(use_batch_shared_entities=False, sp_po__batch=[pref[i] for i in rows.tolist()],
                           entity_vocab_size=n_ent, entity_vocab_offset=2, is_training_data=training,
                           this_split_entities_list=dset.seen_entities_tensor,
                           all_splits_entities_tensor=dset.all_splits_entities_tensor, min_size_batch_labels=0)

        train_batches = [batch_of(tr, t, True) for t in range(nsteps)]
        eval_batch = batch_of(v, 0, False)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    seed, d = 2025, 200
    m = make_model("LookupComplexRelationModel", n_ent, n_rel, d, seed=seed, init_std=0.1)
    E0, R0 = npy(m.entity_embedding.weight).copy(), npy(m.relation_embedding.weight).copy()
    m.train()
    args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": 0.3, "weight_decay": 1.0e-10},
            "lr_scheduler_config": None}
    opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    mod.train()
    kw, losses = {}, []
    for step, out in enumerate(train_batches):
        inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = out
        for o in opts:
            o.update(1, step + 1)
            o.zero_grad()
        loss, _, _ = mod(inputs=list(inputs), labels=labels.clone(), use_batch_shared_entities=False,
                         batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / float(norm_loss)).backward()
        for o in opts:
            o.step()
        losses.append(loss.item())
        kw.update({f"s{step}_po_rel": npy(inputs[0][0]).reshape(-1), f"s{step}_po_obj": npy(inputs[0][1]).reshape(-1),
                   f"s{step}_sp_subj": npy(inputs[1][0]).reshape(-1), f"s{step}_sp_rel": npy(inputs[1][1]).reshape(-1),
                   f"s{step}_labels": npy(labels.nonzero()).astype(np.int32), f"s{step}_normalizer": np.float64(norm_loss)})
    E, R = npy(m.entity_embedding.weight), npy(m.relation_embedding.weight)
    sumE = npy(opts[0].optimizer.state[m.entity_embedding.weight]["sum"])
    m.eval()
    inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = eval_batch
    with torch.no_grad():
        scores = torch.cat([m.po_prefix_score(inputs[0][0], inputs[0][1]), m.sp_prefix_score(inputs[1][0], inputs[1][1])], 0)
    res = OneToNMentionRelationDataset.compute_metrics(filter_mask=filt, label_ids=label_ids, predictions=scores.clone())
    rp, gp, gids = pack_groups(label_ids)
    chk = lambda a: np.asarray([a.sum(dtype=np.float64), np.abs(a).sum(dtype=np.float64), (a.astype(np.float64) ** 2).sum()])  # noqa: E731
    # the same evaluation against a 2048-candidate SUBSET (every answer and filter column of the batch + random fill),
    # with the trained rows it needs stored: ranks on IDENTICAL trained tables can then be compared without the 11.6 MB
    # table (many_obj / many_subj path of the prefix scorers, model.py:52-74)
    rng = np.random.default_rng(seed)
    need = set(int(x) for g in label_ids for grp in g for x in grp.tolist()) | set(npy(filt.nonzero())[:, 1].tolist())
    fill = [int(x) for x in rng.permutation(n_ent - 2) if int(x) not in need][:2048 - len(need)]
    sub_cols = np.sort(np.asarray(sorted(need) + fill, np.int64))              # candidate-relative (entity id - 2)
    assert len(sub_cols) == 2048
    col_to_sub = -np.ones(n_ent - 2, np.int64)
    col_to_sub[sub_cols] = np.arange(len(sub_cols))
    sub_ids = torch.from_numpy((sub_cols + 2).astype(np.int32)).unsqueeze(1)
    with torch.no_grad():
        enc = m.precompute_batch_shared_inputs(sub_ids.view(-1))              # as AddLossModule does, trainer.py:80-82
        sub_scores = torch.cat([m.po_prefix_score(inputs[0][0], inputs[0][1], enc),
                                m.sp_prefix_score(inputs[1][0], inputs[1][1], enc)], 0)
    sub_filt = filt[:, torch.from_numpy(sub_cols)]
    sub_label_ids = [[torch.from_numpy(col_to_sub[grp.long().numpy()]).int() for grp in g] for g in label_ids]
    sub_res = OneToNMentionRelationDataset.compute_metrics(filter_mask=sub_filt, label_ids=sub_label_ids, predictions=sub_scores.clone())
    srp, sgp, sgids = pack_groups(sub_label_ids)
    row_ids = np.unique(np.concatenate([sub_cols + 2, npy(inputs[0][1]).reshape(-1), npy(inputs[1][0]).reshape(-1)])).astype(np.int64)
    kw.update(sub_cand_ids=(sub_cols + 2).astype(np.int32), sub_filter=npy(sub_filt.nonzero()).astype(np.int32),
              sub_row_ptr=srp, sub_grp_ptr=sgp, sub_ids=sgids, sub_ranks=per_group_ranks(sub_filt, sub_label_ids, sub_scores.clone()),
              sub_scores_slice=npy(sub_scores[192:320, 1000:1128]), sub_m_mrr=np.float64(sub_res["mrr"].avg),
              trained_row_ids=row_ids, trained_rows=E[row_ids].copy())
    save("g11_traj_fb15k237", seed=np.int64(seed), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), d=np.int64(d),
         nsteps=np.int64(nsteps), lr=np.float64(0.3),
         table_check=np.asarray([E0.sum(dtype=np.float64), np.abs(E0).sum(dtype=np.float64), R0.sum(dtype=np.float64),
                                 float(E0[5, 7]), float(E0[-1, -1]), float(R0[3, 4])], np.float64),
         losses=np.asarray(losses, np.float64), E_check=chk(E), R_check=chk(R), sumE_check=chk(sumE),
         E_rows=E[2:66].copy(), E_row_ids=np.arange(2, 66), R_final=R.copy(),
         E_col_sum=E.astype(np.float64).sum(0), E_row_sum=E.astype(np.float64).sum(1),
         eval_po_rel=npy(inputs[0][0]).reshape(-1), eval_po_obj=npy(inputs[0][1]).reshape(-1),
         eval_sp_subj=npy(inputs[1][0]).reshape(-1), eval_sp_rel=npy(inputs[1][1]).reshape(-1),
         eval_filter=npy(filt.nonzero()).astype(np.int32), eval_row_ptr=rp, eval_grp_ptr=gp, eval_ids=gids,
         eval_ranks=per_group_ranks(filt, label_ids, scores.clone()),
         eval_score_slice=npy(scores[192:320, 1000:1128]), eval_score_row_absmax=npy(scores.abs().max(1).values),
         **{"eval_m_" + k: np.float64(v.avg) for k, v in res.items()},
         **{"eval_c_" + k: np.float64(v.count) for k, v in res.items()}, **kw)


# ----------------------------------------------------------------------------------------------
# G13: a FULL validation pass of the reference over FB15k-237 valid.txt (Trainer.evaluate's arithmetic, every batch)
# ----------------------------------------------------------------------------------------------
def g13():
    """The reference trains LookupComplexRelationModel d=200 (AddLossModule bce, OptimRegime Adagrad lr 0.3 wd 1e-10,
    input_dropout 0.4) for three passes over its own loader on test.txt (the stand-in training split: train.txt is absent
    upstream).  The trained tables are then ROUNDED TO bf16-REPRESENTABLE VALUES -- so that both sides can hold exactly
    the same tables in a 16-bit fixture -- loaded back into the model, and the reference's evaluation runs over ALL of
    valid.txt exactly as scripts/train.py + Trainer.evaluate do it: dataset.get_loader(shuffle=False, drop_last=False)
    -> collate -> AddLossModule in eval mode (trainer.py:258-272) -> compute_metrics (dataset.py:423-453), MetricResult
    summed over the batches.  Stored: the 16-bit tables, the per-group ranks of every batch (reference tensor ops), the
    per-batch group counts and the final meters."""
    import shutil
    import tempfile
    fb = "/root/reference/data/fb15k237/mapped_to_ids"
    scratch = tempfile.mkdtemp(prefix="okge_g13_")
    try:
        for f in os.listdir(fb):
            shutil.copy(os.path.join(fb, f), scratch)
        files = {"train": "test.txt", "valid": "valid.txt", "test": "test.txt"}
        ds = {}
        for split in ("train", "valid"):
            ds[split] = OneToNMentionRelationDataset(dataset_dir=scratch, input_file=files[split],
                                                     is_training_data=(split == "train"), batch_size=512, copy_data_to_dev_shm=False)
        for split in ("train", "valid"):
            ds[split].merge_all_splits_triples(dataset_dir=scratch, train_input_file=files["train"],
                                               valid_input_file=files["valid"], test_input_file=files["test"])
            ds[split].create_data_tensors(dataset_dir=scratch, train_input_file=files["train"],
                                          valid_input_file=files["valid"], test_input_file=files["test"])
        tr, v = ds["train"], ds["valid"]
        n_ent, n_rel = v.entity_vocab_size, v.relations_size
        seed, d = 2026, 200
        m = make_model("LookupComplexRelationModel", n_ent, n_rel, d, seed=seed, input_dropout=0.4, init_std=0.1)
        m.train()
        args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": 0.3, "weight_decay": 1.0e-10},
                "lr_scheduler_config": None}
        opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
        mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
        mod.train()
        step, losses = 0, []
        torch.manual_seed(seed + 1)
        for epoch in range(3):
            for batch in tr.get_loader(shuffle=True, num_workers=0, drop_last=True):
                inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = batch
                step += 1
                for o in opts:
                    o.update(epoch + 1, step)
                    o.zero_grad()
                loss, _, _ = mod(inputs=list(inputs), labels=labels, use_batch_shared_entities=False,
                                 batch_shared_entities=cand, epoch=epoch + 1, input_style_triple_or_prefix="right_and_left_prefix")
                (loss.sum() / float(norm_loss)).backward()
                for o in opts:
                    o.step()
                losses.append(loss.item() / float(norm_loss))
        print("g13: trained", step, "steps; loss/normalizer", losses[0], "->", losses[-1])
        # tables both sides can hold exactly: round to bf16, keep the 16 bits
        with torch.no_grad():
            Eb, Rb = m.entity_embedding.weight.bfloat16(), m.relation_embedding.weight.bfloat16()
            m.entity_embedding.weight.copy_(Eb.float())
            m.relation_embedding.weight.copy_(Rb.float())
        mod.eval()
        ranks, groups_per_batch, rows_per_batch, total, loss_sum = [], [], [], None, 0.0
        with torch.no_grad():
            for batch in v.get_loader(shuffle=False, num_workers=0, drop_last=False):
                inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = batch
                loss, _, outputs = mod(inputs=list(inputs), labels=labels, use_batch_shared_entities=False,
                                       batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
                res = OneToNMentionRelationDataset.compute_metrics(filter_mask=filt, label_ids=label_ids, predictions=outputs.clone())
                total = res if total is None else total + res                 # Trainer.compute_one_epoch, trainer.py:310
                r = per_group_ranks(filt, label_ids, outputs.clone())
                ranks.append(r)
                groups_per_batch.append(len(r))
                rows_per_batch.append(outputs.shape[0])
                loss_sum += float(loss.item())
        ranks = np.concatenate(ranks)
        print("g13: batches", len(groups_per_batch), "groups", len(ranks), "mrr", total["mrr"].avg, "h10", total["h10"].avg)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    save("g13_valid_pass_fb15k237", seed=np.int64(seed), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), d=np.int64(d),
         E_bf16=Eb.view(torch.int16).numpy().view(np.uint16).copy(), R_bf16=Rb.view(torch.int16).numpy().view(np.uint16).copy(),
         batch_size=np.int64(512), ranks=ranks.astype(np.int32), groups_per_batch=np.asarray(groups_per_batch, np.int64),
         rows_per_batch=np.asarray(rows_per_batch, np.int64), eval_loss_sum=np.float64(loss_sum),
         train_steps=np.int64(step), train_loss_first=np.float64(losses[0]), train_loss_last=np.float64(losses[-1]),
         **{"m_" + k: np.float64(val.avg) for k, val in total.items() if k != "loss"},
         **{"c_" + k: np.float64(val.count) for k, val in total.items() if k != "loss"})


# ----------------------------------------------------------------------------------------------
# G15: epoch-scale training parity -- whole passes of Trainer.train's arithmetic over the reference's own loader
# ----------------------------------------------------------------------------------------------
def g15():
    """The reference trains LookupComplexRelationModel d=200 (AddLossModule bce, OptimRegime Adagrad lr 0.3 wd 1e-10,
    input_dropout 0 so that the run is a function of the data alone) for THREE passes over its own loader on test.txt (the
    stand-in training split), shuffle off, drop_last on -- the loop of Trainer.compute_one_epoch (trainer.py:274-354,
    :217-257) statement for statement.  Stored per step: loss / normalizer, rows (po, sp) and positives of the batch (so the
    test can check it feeds the same batches); after the last pass the reference's evaluation over ALL of valid.txt on the
    trained fp32 tables (as in g13): meters.  The initial tables are regenerated from the seed by the test (checksums)."""
    import shutil
    import tempfile
    fb = "/root/reference/data/fb15k237/mapped_to_ids"
    scratch = tempfile.mkdtemp(prefix="okge_g15_")
    try:
        for f in os.listdir(fb):
            shutil.copy(os.path.join(fb, f), scratch)
        files = {"train": "test.txt", "valid": "valid.txt", "test": "test.txt"}
        ds = {}
        for split in ("train", "valid"):
            ds[split] = OneToNMentionRelationDataset(dataset_dir=scratch, input_file=files[split],
                                                     is_training_data=(split == "train"), batch_size=512, copy_data_to_dev_shm=False)
        for split in ("train", "valid"):
            ds[split].merge_all_splits_triples(dataset_dir=scratch, train_input_file=files["train"],
                                               valid_input_file=files["valid"], test_input_file=files["test"])
            ds[split].create_data_tensors(dataset_dir=scratch, train_input_file=files["train"],
                                          valid_input_file=files["valid"], test_input_file=files["test"])
        tr, v = ds["train"], ds["valid"]
        n_ent, n_rel = v.entity_vocab_size, v.relations_size
        seed, d, lr, n_epochs = 2027, 200, 0.3, 3
        m = make_model("LookupComplexRelationModel", n_ent, n_rel, d, seed=seed, input_dropout=0.0, init_std=0.1)
        E0, R0 = m.entity_embedding.weight.detach().numpy().copy(), m.relation_embedding.weight.detach().numpy().copy()
        table_check = np.asarray([E0.sum(dtype=np.float64), np.abs(E0).sum(dtype=np.float64), R0.sum(dtype=np.float64),
                                  float(E0[5, 7]), float(E0[-1, -1]), float(R0[3, 4])])
        m.train()
        args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": lr, "weight_decay": 1.0e-10},
                "lr_scheduler_config": None}
        opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
        mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
        mod.train()
        step, losses, shape = 0, [], []
        for epoch in range(n_epochs):
            for batch in tr.get_loader(shuffle=False, num_workers=0, drop_last=True):
                inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = batch
                step += 1
                for o in opts:
                    o.update(epoch + 1, step)
                    o.zero_grad()
                loss, _, _ = mod(inputs=list(inputs), labels=labels, use_batch_shared_entities=False,
                                 batch_shared_entities=cand, epoch=epoch + 1, input_style_triple_or_prefix="right_and_left_prefix")
                (loss.sum() / float(norm_loss)).backward()
                for o in opts:
                    o.step()
                losses.append(loss.item() / float(norm_loss))
                n_po = 0 if inputs[0] is None else int(inputs[0][0].shape[0])
                n_sp = 0 if inputs[1] is None else int(inputs[1][0].shape[0])
                shape.append((n_po, n_sp, int(labels.sum().item()), float(norm_loss)))
        print("g15: trained", step, "steps in", n_epochs, "passes; loss/normalizer", losses[0], "->", losses[-1])
        mod.eval()
        total = None
        with torch.no_grad():
            for batch in v.get_loader(shuffle=False, num_workers=0, drop_last=False):
                inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = batch
                loss, _, outputs = mod(inputs=list(inputs), labels=labels, use_batch_shared_entities=False,
                                       batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
                res = OneToNMentionRelationDataset.compute_metrics(filter_mask=filt, label_ids=label_ids, predictions=outputs.clone())
                total = res if total is None else total + res
        print("g15: valid mrr", total["mrr"].avg, "h10", total["h10"].avg)
        E1, R1 = m.entity_embedding.weight.detach().numpy(), m.relation_embedding.weight.detach().numpy()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    save("g15_epochs_fb15k237", seed=np.int64(seed), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), d=np.int64(d), lr=np.float64(lr),
         n_epochs=np.int64(n_epochs), batch_size=np.int64(512), table_check=table_check, losses=np.asarray(losses, np.float64),
         batch_shape=np.asarray(shape, np.float64), final_check=np.asarray([E1.sum(dtype=np.float64), np.abs(E1).sum(dtype=np.float64),
                                                                           R1.sum(dtype=np.float64), np.abs(R1).sum(dtype=np.float64)]),
         E_rows=E1[[2, 100, 5000, 14000]].copy(), R_rows=R1[[2, 50, 200]].copy(),
         **{"m_" + k: np.float64(val.avg) for k, val in total.items() if k != "loss"},
         **{"c_" + k: np.float64(val.count) for k, val in total.items() if k != "loss"})


# ----------------------------------------------------------------------------------------------
# G14: BASELINE configs[2] at its size through the reference: LookupDistmultRelationModel d = 512, batch-shared sampled candidates
# ----------------------------------------------------------------------------------------------
def g14():
    """512 prefixes of FB15k-237 (test.txt standing in for the absent train split) collated by the reference with
    use_batch_shared_entities=True, min_size_batch_labels=10000 (numpy-sampled fill-up negatives, dataset.py:853-860), then
    AddLossModule (bce) forward + (loss / normalizer).backward() of the reference's LookupDistmultRelationModel d = 512.
    Stored: prefix ids, the 10 000 candidate ids, label coordinates, loss, a score slice + per-row score sums, gradient
    checksums + slices.  The tables are regenerated from the seed by the test (checksums guard them)."""
    import shutil
    import tempfile
    from openkge.dataset import OneToNMentionRelationDataset_collate_func as collate
    fb = "/root/reference/data/fb15k237/mapped_to_ids"
    scratch = tempfile.mkdtemp(prefix="okge_g14_")
    try:
        for f in os.listdir(fb):
            shutil.copy(os.path.join(fb, f), scratch)
        files = {"train": "test.txt", "valid": "valid.txt", "test": "test.txt"}
        tr = OneToNMentionRelationDataset(dataset_dir=scratch, input_file=files["train"], is_training_data=True, batch_size=512,
                                          copy_data_to_dev_shm=False)
        OneToNMentionRelationDataset(dataset_dir=scratch, input_file=files["valid"], is_training_data=False, batch_size=512,
                                     copy_data_to_dev_shm=False)          # (writes the valid split's prefix files the merge reads)
        tr.merge_all_splits_triples(dataset_dir=scratch, train_input_file=files["train"], valid_input_file=files["valid"],
                                    test_input_file=files["test"])
        tr.create_data_tensors(dataset_dir=scratch, train_input_file=files["train"], valid_input_file=files["valid"],
                               test_input_file=files["test"])
        n_ent, n_rel = tr.entity_vocab_size, tr.relations_size
        pref = tr.seen_prefixes_tensor
        slot = pref[:, 6]
        rows = torch.cat([torch.nonzero(slot == 0).view(-1)[1000:1256], torch.nonzero(slot == 2).view(-1)[2000:2256]])
        np.random.seed(1414)
        out = collate(use_batch_shared_entities=True, sp_po__batch=[pref[i] for i in rows.tolist()], entity_vocab_size=n_ent,
                      entity_vocab_offset=2, is_training_data=True, this_split_entities_list=tr.seen_entities_tensor,
                      all_splits_entities_tensor=tr.all_splits_entities_tensor, min_size_batch_labels=10000)
        inputs, norm_loss, norm_metric, labels, label_ids, filt, cand = out
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    assert cand.numel() == 10000 and labels.shape == (512, 10000)
    seed, d = 2027, 512
    m = make_model("LookupDistmultRelationModel", n_ent, n_rel, d, seed=seed, init_std=0.1)
    E, R = npy(m.entity_embedding.weight).copy(), npy(m.relation_embedding.weight).copy()
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    mod.train()
    loss, _, outputs = mod(inputs=list(inputs), labels=labels.clone(), use_batch_shared_entities=True,
                           batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
    (loss.sum() / float(norm_loss)).backward()
    dE, dR = npy(m.entity_embedding.weight.grad), npy(m.relation_embedding.weight.grad)
    x = outputs.detach()
    save("g14_distmult_d512_sampled", seed=np.int64(seed), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), d=np.int64(d),
         table_check=np.asarray([E.sum(dtype=np.float64), np.abs(E).sum(dtype=np.float64), R.sum(dtype=np.float64),
                                 float(E[5, 7]), float(E[-1, -1]), float(R[3, 4])], np.float64),
         po_rel=npy(inputs[0][0]).reshape(-1), po_obj=npy(inputs[0][1]).reshape(-1), sp_subj=npy(inputs[1][0]).reshape(-1),
         sp_rel=npy(inputs[1][1]).reshape(-1), cand=npy(cand).reshape(-1).astype(np.int32),
         labels=npy(labels.nonzero()).astype(np.int32), loss=np.float64(loss.item()), normalizer=np.float64(norm_loss),
         n_labels=np.float64(norm_metric), score_slice=npy(x[192:320, 4000:4128]), score_row_sum=npy(x.double().sum(1)),
         score_row_absmax=npy(x.abs().max(1).values), dE_row_sum=dE.astype(np.float64).sum(1),
         dE_abs_sum=np.float64(np.abs(dE).sum(dtype=np.float64)), dE_slice=dE[npy(cand).reshape(-1)[:64].astype(np.int64), :16].copy(),
         dR_row_sum=dR.astype(np.float64).sum(1), dR_slice=dR[:32, :64].copy())


# ----------------------------------------------------------------------------------------------
# G12: embedder variants of the lookup models (batch-norm, entity projection, normalisation, l2_reg hook)
# ----------------------------------------------------------------------------------------------
def g12(only_cases=()):
    """LookupComplexRelationModel with the _encode variants of model.py:463-479 switched on (dropout 0: torch's CPU
    Bernoulli stream is not reproducible elsewhere): AddLossModule forward, Trainer's backward_loss = (loss + hook) /
    normalizer (trainer.py:217-222), gradients of EVERY parameter, batch-norm running statistics after the step, and an
    eval-mode forward afterwards."""
    cases = {"bn": dict(batch_norm=True), "proj": dict(project_entity=True), "norm": dict(normalize="norm"),
             "l2": dict(l2_reg=0.01), "all": dict(batch_norm=True, project_entity=True, normalize="norm", l2_reg=0.01),
             # batch_shared_entities=None (trainer.py:86-87): each prefix scorer encodes its own candidate block
             # (get_all_subj for the po rows, get_all_obj for the sp rows, model.py:60-61 / :71-72)
             "all_noshare": dict(batch_norm=True, project_entity=True, normalize="norm", l2_reg=0.01)}
    n_ent, n_rel, d, b = 60, 8, 16, 6
    for name, kw in cases.items():
        if only_cases and name not in only_cases:
            continue
        no_shared = name.endswith("_noshare")
        seed = 120 + len(name)
        rng = np.random.default_rng(seed)
        torch.manual_seed(seed)
        m = Models.LookupComplexRelationModel(entity_slot_size=d, input_dropout=0.0, init_std=0.3, sparse=False,
                                              train_data=meta(n_ent, n_rel), **kw)
        m.train()
        state0 = {k: npy(v).copy() for k, v in m.state_dict().items()}
        mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
        mod.train()
        cand = torch.arange(n_ent)[2:].int().unsqueeze(1)
        N = cand.shape[0]
        po_rel, po_obj = rand_ids(rng, 2, n_rel, b), rand_ids(rng, 2, n_ent, b)
        sp_subj, sp_rel = rand_ids(rng, 2, n_ent, b), rand_ids(rng, 2, n_rel, b)
        y = dense_labels(rng, 2 * b, N)
        loss, hook, outputs = mod(inputs=[(po_rel, po_obj), (sp_subj, sp_rel)], labels=torch.from_numpy(y.copy()),
                                  use_batch_shared_entities=False, batch_shared_entities=None if no_shared else cand, epoch=1,
                                  input_style_triple_or_prefix="right_and_left_prefix")
        normalizer = float(2 * b * N)
        backward_loss = loss.sum()
        if hook is not None:
            backward_loss = backward_loss + hook
        (backward_loss / normalizer).backward()
        out = dict(case=np.str_(name), kw_keys=np.asarray(sorted(kw)), po_rel=npy(po_rel), po_obj=npy(po_obj), sp_subj=npy(sp_subj),
                   sp_rel=npy(sp_rel), cand=npy(cand), labels=y, loss=np.float64(loss.item()),
                   hook=np.float64(hook.item() if hook is not None else 0.0), has_hook=np.bool_(hook is not None),
                   outputs=npy(outputs), normalizer=np.float64(normalizer), no_shared=np.bool_(no_shared),
                   batch_norm=np.bool_(kw.get("batch_norm", False)), project_entity=np.bool_(kw.get("project_entity", False)),
                   normalize=np.str_(kw.get("normalize", "")), l2_reg=np.float64(kw.get("l2_reg", 0)))
        for k, v in state0.items():
            out["p0_" + k] = v
        for k, prm in m.named_parameters():
            out["g_" + k] = npy(prm.grad) if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32)
        for k, v in m.state_dict().items():
            if "running" in k:
                out["after_" + k] = npy(v)
        m.eval()
        with torch.no_grad():
            ev = torch.cat([m.po_prefix_score(po_rel, po_obj), m.sp_prefix_score(sp_subj, sp_rel)], 0)
        out["eval_outputs"] = npy(ev)
        save("g12_variant_" + name, **out)


# ----------------------------------------------------------------------------------------------
# G8: checkpoint interop (Trainer.save layout, trainer.py:608-618)
# ----------------------------------------------------------------------------------------------
def g8():
    n_ent, n_rel, d, b = 90, 11, 24, 10
    rng = np.random.default_rng(88)
    m = make_model("LookupComplexRelationModel", n_ent, n_rel, d, seed=88)
    m.train()
    args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": 0.3, "weight_decay": 1.0e-10},
            "lr_scheduler_config": None}
    opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    mod.train()
    cand = torch.arange(n_ent)[2:].int().unsqueeze(1)
    N = cand.shape[0]
    kw = {}
    for step in range(3):
        po = (rand_ids(rng, 2, n_rel, b), rand_ids(rng, 2, n_ent, b))
        sp = (rand_ids(rng, 2, n_ent, b), rand_ids(rng, 2, n_rel, b))
        y = dense_labels(rng, 2 * b, N)
        if step == 2:
            # exactly what Trainer.save stores (minus the ResultsLog object, which is not tensor data)
            state = {"epoch": 1, "training_steps": 2, "state_dict": m.state_dict(),
                     "optimizer_state_dict": [o.state_dict() for o in opts], "validation_results": None}
            torch.save(state, os.path.join(OUT, "g8_checkpoint.pt"))
            kw.update(po_rel=npy(po[0]), po_obj=npy(po[1]), sp_subj=npy(sp[0]), sp_rel=npy(sp[1]), labels=y)
        for o in opts:
            o.update(1, step + 1)
            o.zero_grad()
        loss, _, _ = mod(inputs=[po, sp], labels=torch.from_numpy(y.copy()), use_batch_shared_entities=False,
                         batch_shared_entities=cand, epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / float(2 * b * N)).backward()
        for o in opts:
            o.step()
    st = opts[0].optimizer.state
    kw.update(loss=np.float64(loss.item()), E=npy(m.entity_embedding.weight), R=npy(m.relation_embedding.weight),
              sumE=npy(st[m.entity_embedding.weight]["sum"]), sumR=npy(st[m.relation_embedding.weight]["sum"]))
    save("g8_checkpoint_step3", **kw)
    # the state after step 3, as the reference would save it: target layout for OUR writer
    state = {"epoch": 1, "training_steps": 3, "state_dict": m.state_dict(),
             "optimizer_state_dict": [o.state_dict() for o in opts], "validation_results": None}
    torch.save(state, os.path.join(OUT, "g8_checkpoint_after.pt"))


def g16():
    rng = np.random.default_rng(1600)
    n_ent, n_rel, d, b_po, b_sp, n_cand, L, vt_e, vt_r, nsteps = 220, 14, 24, 9, 10, 48, 6, 300, 40, 12

    def token_map(n, vocab, hi):                             # body tokens from [4, hi): ids >= hi are never named by anything
        out = [[1], [1]]
        for _ in range(2, n):
            k = int(rng.integers(1, L))
            out.append([2] + rng.integers(4, hi, size=k).tolist() + [3])
        return tuple(out)
    md = meta(n_ent, n_rel)
    md.entity_id_to_tokens_map, md.relation_id_to_tokens_map = token_map(n_ent, vt_e, 200), token_map(n_rel, vt_r, 30)
    md.entity_tokens_size, md.relation_tokens_size, md.max_length = vt_e, vt_r, (L, L)
    torch.manual_seed(1600)
    m = Models.UnigramPoolingComplexRelationModel(entity_slot_size=d, relation_slot_size=d, train_data=md, pool="sum",
                                                  normalize="batchnorm", dropout=0.0, sparse=False, init_std=0.3)
    m.entity_projection = None                               # harness shim, see the module docstring
    m.train()
    args = {"optimization_config": {"optimizer": "Adagrad", "epoch": 0, "lr": 0.1, "weight_decay": 1.0e-10}, "lr_scheduler_config": None}
    opts = OptimRegime.setup_optimizer_regime(args=args, model=m)
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    mod.train()
    kw = dict(We=npy(m.entity_embedding.weight).copy(), Wr=npy(m.relation_embedding.weight).copy(),
              ent_tokens=npy(m.entity_token_ids).astype(np.int32), rel_tokens=npy(m.relation_token_ids).astype(np.int32),
              bn_e_w=npy(m.entity_batchnorm.weight).copy(), bn_e_b=npy(m.entity_batchnorm.bias).copy(),
              bn_r_w=npy(m.relation_batchnorm.weight).copy(), bn_r_b=npy(m.relation_batchnorm.bias).copy(), nsteps=np.int64(nsteps))
    B = b_po + b_sp
    for step in range(nsteps):
        lo = 2 + (step % 4) * 50                              # the batches move through the entity ids: rows go cold and come back
        cand = torch.from_numpy((lo + rng.permutation(60)[:n_cand]).astype(np.int32)).unsqueeze(1)
        po = (rand_ids(rng, 2, n_rel, b_po), rand_ids(rng, lo, lo + 60, b_po))
        sp = (rand_ids(rng, lo, lo + 60, b_sp), rand_ids(rng, 2, n_rel, b_sp))
        y = dense_labels(rng, B, n_cand)
        for o in opts:
            o.update(1, step + 1)
            o.zero_grad()
        loss, _, _ = mod(inputs=[po, sp], labels=torch.from_numpy(y.copy()), use_batch_shared_entities=True, batch_shared_entities=cand,
                         epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
        (loss.sum() / float(B * n_cand)).backward()
        for o in opts:
            o.step()
        kw.update({f"s{step}_cand": npy(cand), f"s{step}_po_rel": npy(po[0]), f"s{step}_po_obj": npy(po[1]), f"s{step}_sp_subj": npy(sp[0]),
                   f"s{step}_sp_rel": npy(sp[1]), f"s{step}_labels": y, f"s{step}_loss": np.float64(loss.item())})
    g = opts[0].optimizer.param_groups[0]
    st = opts[0].optimizer.state
    kw.update({"opt_" + k: np.float64(g[k]) for k in ("lr", "eps", "weight_decay")})
    kw.update(We_end=npy(m.entity_embedding.weight).copy(), Wr_end=npy(m.relation_embedding.weight).copy(),
              sumWe_end=npy(st[m.entity_embedding.weight]["sum"]).copy(), sumWr_end=npy(st[m.relation_embedding.weight]["sum"]).copy(),
              bn_e_w_end=npy(m.entity_batchnorm.weight).copy(), bn_e_b_end=npy(m.entity_batchnorm.bias).copy(),
              run_e_mean_end=npy(m.entity_batchnorm.running_mean).copy(), run_e_var_end=npy(m.entity_batchnorm.running_var).copy())
    named = np.zeros(vt_e, bool)
    named[np.unique(npy(m.entity_token_ids))] = True
    print("g16: entity token rows never named:", int((~named).sum()), "of", vt_e, "| moved by the optimizer:",
          float(np.abs(kw["We_end"][~named] - kw["We"][~named]).max()))
    save("g16_unigram_adagrad", **kw)


if __name__ == "__main__":
    only = sys.argv[1:]                      # e.g. `make_golden.py g1_triples` regenerates one family
    for fn in (g1, g1_triples, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, g12, g13, g14, g15, g16):
        if fn is g12 and any(a.startswith("g12:") for a in only):    # `make_golden.py g12:all_noshare`: one case of the family
            fn(tuple(a[4:] for a in only if a.startswith("g12:")))
        elif not only or fn.__name__ in only:
            fn()
    print("torch", torch.__version__, "numpy", np.__version__)
