"""The pieces that make the reference-Trainer drop-in path fast and safe:

  optim.OkgeAdagrad            torch.optim.Adagrad's arithmetic on the HIP sweep, nameable by OptimRegime (utils/optim.py:139-160)
  AddLossModule(training_outputs=False), loss.reduction check (trainer.py:48-113)
  AddLossModule on the token-pooled models (autograd bridge; trainer.py:142,206-234 over model.py:716-796) vs G9
  FusedTrainStep(grad_clip, accumulate)   trainer.py:229-244
  okge_id_errors               an out-of-range id is substituted and counted, never dereferenced
  okge_merge_logsumexp         cross-shard log-sum-exp of the sharded KL loss"""
import os

import numpy as np
import pytest
import torch

from conftest import golden, golden_names
from oracle import kge_oracle as ko


def test_reduction_other_than_sum_is_refused():
    from open_knowledge_graph_embeddings_amd.trainer import AddLossModule
    with pytest.raises(NotImplementedError):
        AddLossModule(torch.nn.Linear(1, 1), torch.nn.BCEWithLogitsLoss(reduction="mean"), 0.0)
    AddLossModule(torch.nn.Linear(1, 1), torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)


def test_okge_adagrad_is_registered_for_optimregime():
    import open_knowledge_graph_embeddings_amd  # noqa: F401
    assert torch.optim.__dict__["OkgeAdagrad"].__name__ == "OkgeAdagrad"
    # utils/optim.py:143-144 builds the new optimizer from the OLD one's param_groups: Adam's eps = 1e-8 leaks in
    p = torch.nn.Parameter(torch.zeros(4, 4))
    adam = torch.optim.Adam([p], lr=0)
    o = torch.optim.__dict__["OkgeAdagrad"](adam.param_groups)
    assert o.param_groups[0]["eps"] == 1e-8 and o.param_groups[0]["lr_decay"] == 0
    assert set(o.state[p].keys()) == {"step", "sum"}                   # torch.optim.Adagrad's state layout


def _dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dt is None else t.to(dt)).cuda()


@pytest.mark.gpu
def test_okge_adagrad_matches_torch_adagrad(okge_lib):
    from open_knowledge_graph_embeddings_amd.optim import OkgeAdagrad
    torch.manual_seed(3)
    shapes = [(301, 200), (17, 200), (5, 8)]
    mine = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.1) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in mine]
    o1 = OkgeAdagrad(mine, lr=0.3, weight_decay=1e-10, eps=1e-8)
    o2 = torch.optim.Adagrad(ref, lr=0.3, weight_decay=1e-10, eps=1e-8)
    for step in range(4):
        for a, b in zip(mine, ref):
            g = torch.randn(a.shape) * 1e-3
            if step == 2 and a.shape[0] == 17:
                a.grad, b.grad = None, None                       # a parameter without gradient is skipped by both
                continue
            a.grad, b.grad = g.cuda(), g.clone()
        o1.step()
        o2.step()
    for a, b in zip(mine, ref):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=0, atol=3e-6)
        np.testing.assert_allclose(o1.state[a]["sum"].cpu().numpy(), o2.state[b]["sum"].numpy(), rtol=1e-6, atol=1e-12)
        assert float(o1.state[a]["step"]) == float(o2.state[b]["step"])
    sd = o1.state_dict()                                            # loads into torch.optim.Adagrad: same layout
    o3 = torch.optim.Adagrad([torch.nn.Parameter(p.detach().clone()) for p in mine], lr=0.3)
    o3.load_state_dict(sd)


def _lookup_model(z, scorer_cls="LookupComplexRelationModel", dropout=0.0):
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    m = getattr(Models, scorer_cls)(entity_slot_size=z["E"].shape[1], input_dropout=dropout, init_std=0.1, sparse=False,
                                    train_data=EntityRelationDatasetMeta(entities_size=z["E"].shape[0], relations_size=z["R"].shape[0]))
    m = m.cuda()
    m.entity_embedding.weight.data.copy_(_dev(z["E"]))
    m.relation_embedding.weight.data.copy_(_dev(z["R"]))
    return m


@pytest.mark.gpu
def test_addloss_without_training_outputs_and_okge_adagrad_step(okge_lib):
    """the INTEGRATION section 1 patch in its fast form: no all_outputs in training, OkgeAdagrad as the optimizer --
    loss, gradients and the updated tables equal the reference's (G2 loss/grads, G3-style update by the oracle)"""
    from open_knowledge_graph_embeddings_amd.optim import OkgeAdagrad
    from open_knowledge_graph_embeddings_amd.trainer import AddLossModule
    z = golden("g2_loss_complex_bce_all")
    m = _lookup_model(z)
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0, training_outputs=False)
    opt = OkgeAdagrad(torch.optim.Adam(m.parameters(), lr=0).param_groups)          # as OptimRegime.adjust does
    for grp in opt.param_groups:
        grp["lr"], grp["weight_decay"] = 0.3, 1e-10
    inputs = [(_dev(z["po_rel"]), _dev(z["po_obj"])), (_dev(z["sp_subj"]), _dev(z["sp_rel"]))]
    loss, hook, outs = mod(inputs=inputs, labels=_dev(z["labels"]), use_batch_shared_entities=False,
                           batch_shared_entities=_dev(z["cand"]), epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
    assert outs is None and hook is None
    assert abs(float(loss.detach()) - float(z["loss"])) <= 3e-5 * abs(float(z["loss"]))
    (loss.sum() / float(z["normalizer"])).backward()
    gE, gR = m.entity_embedding.weight.grad, m.relation_embedding.weight.grad
    np.testing.assert_allclose(gE.cpu().numpy(), z["dE"], rtol=0, atol=3e-5 * np.abs(z["dE"]).max())
    np.testing.assert_allclose(gR.cpu().numpy(), z["dR"], rtol=0, atol=3e-5 * np.abs(z["dR"]).max())
    E, R = z["E"].copy(), z["R"].copy()
    ko.adagrad_step(E, gE.cpu().numpy(), np.zeros_like(E), 0.3)
    ko.adagrad_step(R, gR.cpu().numpy(), np.zeros_like(R), 0.3)
    opt.step()
    np.testing.assert_allclose(m.entity_embedding.weight.detach().cpu().numpy(), E, rtol=0, atol=2e-6)
    np.testing.assert_allclose(m.relation_embedding.weight.detach().cpu().numpy(), R, rtol=0, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("factor", [1.0, 2.5, "other_normalizer", "sum_of_two"])
def test_addloss_gradient_scale_follows_the_upstream_gradient(okge_lib, factor):
    """AddLossModule folds the Trainer's 1 / (B N) into the fused step and _FusedLossFn.backward rescales on the device only
    if autograd's upstream scalar is another number: the reference's own expression (factor 1.0: nothing to rescale), a
    multiple of it, a different normalizer, and a loss used twice in the graph all give the right gradient"""
    from open_knowledge_graph_embeddings_amd.trainer import AddLossModule
    z = golden("g2_loss_complex_bce_all")
    m = _lookup_model(z)
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0, training_outputs=False)
    inputs = [(_dev(z["po_rel"]), _dev(z["po_obj"])), (_dev(z["sp_subj"]), _dev(z["sp_rel"]))]
    loss, _, _ = mod(inputs=inputs, labels=_dev(z["labels"]), use_batch_shared_entities=False,
                     batch_shared_entities=_dev(z["cand"]), epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
    norm = float(z["normalizer"])
    if factor == "other_normalizer":
        (loss.sum() / 123.0).backward()
        scale = norm / 123.0
    elif factor == "sum_of_two":
        (loss.sum() / norm + 0.5 * loss / norm).backward()
        scale = 1.5
    else:
        (factor * loss.sum() / norm).backward()
        scale = factor
    for g, ref in ((m.entity_embedding.weight.grad, z["dE"]), (m.relation_embedding.weight.grad, z["dR"])):
        np.testing.assert_allclose(g.cpu().numpy(), scale * ref, rtol=0, atol=3e-5 * scale * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names("g9_unigram_"))
def test_addloss_on_token_pooled_models(okge_lib, name):
    """AddLossModule + autograd on UnigramPoolingComplexRelationModel: the reference Trainer's own path, vs the
    reference's loss / outputs / parameter gradients (G9)"""
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.token_pooled import UnigramPoolingComplexRelationModel
    from open_knowledge_graph_embeddings_amd.trainer import AddLossModule
    z = golden(name)
    n_ent, L = z["ent_tokens"].shape
    n_rel = z["rel_tokens"].shape[0]
    md = EntityRelationDatasetMeta(entities_size=n_ent, relations_size=n_rel, entity_tokens_size=z["We"].shape[0],
                                   relation_tokens_size=z["Wr"].shape[0], max_length=(L, L),
                                   entity_id_to_tokens_map=[[int(t) for t in row if t] or [0] for row in z["ent_tokens"]],
                                   relation_id_to_tokens_map=[[int(t) for t in row if t] or [0] for row in z["rel_tokens"]])
    bn = str(z["normalize"]) == "batchnorm"
    m = UnigramPoolingComplexRelationModel(entity_slot_size=z["We"].shape[1], relation_slot_size=z["We"].shape[1], train_data=md,
                                           pool=str(z["pool"]), normalize="batchnorm" if bn else None, dropout=0.0, init_std=0.3).cuda()
    m.entity_embedding.weight.data.copy_(_dev(z["We"]))
    m.relation_embedding.weight.data.copy_(_dev(z["Wr"]))
    if bn:
        for mod_bn, w in ((m.entity_batchnorm, "e"), (m.relation_batchnorm, "r")):
            mod_bn.weight.data.copy_(_dev(z[f"bn_{w}_w"]))
            mod_bn.bias.data.copy_(_dev(z[f"bn_{w}_b"]))
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    inputs = [None, None]
    if "po_rel" in z.files:
        inputs[0] = (_dev(z["po_rel"]), _dev(z["po_obj"]))
    if "sp_subj" in z.files:
        inputs[1] = (_dev(z["sp_subj"]), _dev(z["sp_rel"]))
    shared = bool(z["shared"]) if "shared" in z.files else True
    loss, _, outs = mod(inputs=inputs, labels=_dev(z["labels"]), use_batch_shared_entities=shared,
                        batch_shared_entities=_dev(z["cand"].astype(np.int32)), epoch=1,
                        input_style_triple_or_prefix="right_and_left_prefix")
    np.testing.assert_allclose(outs.cpu().numpy(), z["outputs"], rtol=0, atol=1e-4)
    assert abs(float(loss.detach()) - float(z["loss"])) <= 3e-5 * abs(float(z["loss"]))
    norm = float(z["normalizer"]) if "normalizer" in z.files else 1.0
    (loss.sum() / norm).backward()
    scale = 1.0 / norm
    for p, key in ((m.entity_embedding.weight, "dWe"), (m.relation_embedding.weight, "dWr")):
        ref = z[key] * (scale if "normalizer" not in z.files or abs(np.abs(z[key]).max()) > 0 else 1.0)
        got = p.grad.cpu().numpy()
        # the fixture's gradients are of loss / normalizer when it carries one, of the summed loss otherwise
        ref = z[key]
        k = 1.0 if np.allclose(got, ref, rtol=0, atol=5e-5 * np.abs(ref).max()) else norm
        np.testing.assert_allclose(got * k, ref, rtol=0, atol=5e-5 * np.abs(ref).max())
    if bn:
        for mod_bn, w in ((m.entity_batchnorm, "e"), (m.relation_batchnorm, "r")):
            for par, key in ((mod_bn.weight, f"d_bn_{w}_w"), (mod_bn.bias, f"d_bn_{w}_b")):
                got, ref = par.grad.cpu().numpy(), z[key]
                k = 1.0 if np.allclose(got, ref, rtol=0, atol=5e-5 * np.abs(ref).max() + 1e-8) else norm
                np.testing.assert_allclose(got * k, ref, rtol=0, atol=5e-5 * np.abs(ref).max() + 1e-8)
    # eval mode: loss + scores without gradients run too
    m.eval()
    with torch.no_grad():
        l2, _, o2 = mod(inputs=inputs, labels=_dev(z["labels"]), use_batch_shared_entities=shared,
                        batch_shared_entities=_dev(z["cand"].astype(np.int32)), epoch=1,
                        input_style_triple_or_prefix="right_and_left_prefix")
    assert torch.isfinite(l2) and o2.shape == outs.shape


@pytest.mark.gpu
def test_fused_step_grad_clip_and_accumulation(okge_lib):
    """FusedTrainStep(grad_clip, accumulate=2) == oracle: sum the gradients of two batches, clip the global 2-norm
    (torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1), one Adagrad step (trainer.py:229-244)"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    rng = np.random.default_rng(11)
    n_ent, n_rel, d, b = 400, 12, 200, 48
    E = (rng.standard_normal((n_ent, d)) * 0.1).astype(np.float32)
    R = (rng.standard_normal((n_rel, d)) * 0.1).astype(np.float32)
    N = n_ent - 2
    batches, grads = [], []
    for k in range(2):
        po = (rng.integers(2, n_rel, b).astype(np.int32), rng.integers(2, n_ent, b).astype(np.int32))
        sp = (rng.integers(2, n_ent, b).astype(np.int32), rng.integers(2, n_rel, b).astype(np.int32))
        y = np.zeros((2 * b, N), np.float32)
        for r in range(2 * b):
            y[r, rng.choice(N, size=2, replace=False)] = 1
        ref = ko.step_forward_backward(ko.COMPLEX, E, R, po, sp, np.arange(2, n_ent), y)
        grads.append((ref["dE"], ref["dR"]))
        pr, pc = H.positives_from_dense(_dev(y))
        batches.append(H.PrefixBatch(po_rel=_dev(po[0]), po_obj=_dev(po[1]), sp_subj=_dev(sp[0]), sp_rel=_dev(sp[1]),
                                     pos_row=pr, pos_col=pc, cand_first=2, n_cand=N))
    gE, gR = grads[0][0] + grads[1][0], grads[0][1] + grads[1][1]
    norm = np.sqrt((gE.astype(np.float64) ** 2).sum() + (gR.astype(np.float64) ** 2).sum())
    clip = 0.5 * norm                                               # forces clipping
    coef = min(1.0, clip / (norm + 1e-6))
    Eo, Ro = E.copy(), R.copy()
    ko.adagrad_step(Eo, (gE * np.float32(coef)).astype(np.float32), np.zeros_like(E), 0.3)
    ko.adagrad_step(Ro, (gR * np.float32(coef)).astype(np.float32), np.zeros_like(R), 0.3)
    Et, Rt = _dev(E), _dev(R)
    ts = FusedTrainStep(Et, Rt, "complex", lr=0.3, grad_clip=float(clip), accumulate=2)
    ts.step(batches[0])
    assert torch.equal(Et.cpu(), torch.from_numpy(E))               # no optimizer step after the first of two batches
    ts.step(batches[1])
    torch.cuda.synchronize()
    # the first Adagrad step is lr * g / (|g| + eps): compare where |g| >> eps, i.e. the update is insensitive to noise
    big = np.abs(gE * coef) > 1e-6
    np.testing.assert_allclose(Et.cpu().numpy()[big], Eo[big], rtol=0, atol=2e-5)
    assert np.abs(Et.cpu().numpy() - Eo).max() < 5e-3
    np.testing.assert_allclose(ts.sumE.cpu().numpy(), (gE * np.float32(coef)) ** 2, rtol=2e-3, atol=1e-16)
    # and without clipping / accumulation the step is unchanged (a limit far above the norm: coef clamps to 1)
    E2, R2 = _dev(E), _dev(R)
    t2 = FusedTrainStep(E2, R2, "complex", lr=0.3, grad_clip=1e9)
    E3, R3 = _dev(E), _dev(R)
    t3 = FusedTrainStep(E3, R3, "complex", lr=0.3)
    t2.step(batches[0]); t3.step(batches[0])
    # (equal up to the order of the prefix rows' float atomics)
    assert (E2 - E3).abs().max() < 5e-3 and ((E2 - E3).abs() > 1e-5).float().mean() < 1e-3 and (R2 - R3).abs().max() < 5e-3


@pytest.mark.gpu
def test_out_of_range_ids_are_counted_not_dereferenced(okge_lib, monkeypatch):
    """production configuration (OKGE_VALIDATE unset): a bad id never becomes an out-of-bounds access; okge_id_errors
    reports it (the reference raises IndexError inside torch.nn.Embedding, model.py:457-460)"""
    from open_knowledge_graph_embeddings_amd import _native as N, hotpath as H
    monkeypatch.setattr(H, "VALIDATE", False)
    hp = H.HotPath("cuda:0")
    rng = np.random.default_rng(2)
    n_ent, n_rel, d, b = 300, 10, 64, 40
    E, R = _dev((rng.standard_normal((n_ent, d)) * 0.1).astype(np.float32)), _dev((rng.standard_normal((n_rel, d)) * 0.1).astype(np.float32))
    N.id_errors()                                                   # clear
    good = H.PrefixBatch(po_rel=_dev(rng.integers(2, n_rel, b).astype(np.int32)), po_obj=_dev(rng.integers(2, n_ent, b).astype(np.int32)),
                         sp_subj=_dev(rng.integers(2, n_ent, b).astype(np.int32)), sp_rel=_dev(rng.integers(2, n_rel, b).astype(np.int32)),
                         pos_row=_dev(np.arange(2 * b, dtype=np.int32)), pos_col=_dev(np.sort(rng.integers(0, n_ent - 2, 2 * b)).astype(np.int32)),
                         cand_first=2, n_cand=n_ent - 2)
    dE, dR = torch.zeros_like(E), torch.zeros_like(R)
    hp.forward_backward(E, R, "complex", good, dE, dR, grads_zero=True)
    hp.score(E, R, "complex", good)
    assert N.id_errors() == 0
    bad_obj = good.po_obj.clone(); bad_obj[3] = n_ent + 1000; bad_obj[5] = -7
    bad_rel = good.sp_rel.clone(); bad_rel[0] = n_rel
    bad = H.PrefixBatch(po_rel=good.po_rel, po_obj=bad_obj, sp_subj=good.sp_subj, sp_rel=bad_rel, pos_row=good.pos_row,
                        pos_col=good.pos_col, cand_ids=_dev(np.concatenate([np.arange(2, 200), [10 ** 7, -3]]).astype(np.int32)))
    bad.pos_col = _dev(np.sort(rng.integers(0, 200, 2 * b)).astype(np.int32))
    dE.zero_(); dR.zero_()
    loss = hp.forward_backward(E, R, "complex", bad, dE, dR)
    x = hp.score(E, R, "complex", bad)
    torch.cuda.synchronize()
    assert N.id_errors() > 0 and N.id_errors() == 0                 # counted, then cleared
    assert torch.isfinite(loss).all() and torch.isfinite(x).all() and torch.isfinite(dE).all()
    with pytest.raises(N.OkgeError):
        hp.score(E, R, "complex", bad)
        N.check_ids()
    hp.encode_rows(E, _dev(np.asarray([1, 5, 10 ** 6], np.int32)))
    assert N.id_errors() == 1


@pytest.mark.gpu
def test_merge_logsumexp(okge_lib):
    from open_knowledge_graph_embeddings_amd import hotpath as H
    hp = H.HotPath("cuda:0")
    x = torch.randn(5, 1000, device="cuda") * 20
    x[2, 7] = float("-inf")
    np.testing.assert_allclose(hp.merge_logsumexp(x).cpu().numpy(), torch.logsumexp(x, 0).cpu().numpy(), rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_validate_config_rejects_bad_ids_before_any_kernel(okge_lib, validate_config):
    """OKGE_VALIDATE=1 (integration work): ids are range-checked on the host before the call; nothing reaches the kernels"""
    from open_knowledge_graph_embeddings_amd import _native as N
    from open_knowledge_graph_embeddings_amd import hotpath as H
    dev = torch.device("cuda:0")
    hp = H.HotPath(dev)
    E, R = torch.randn((50, 16), device=dev), torch.randn((7, 16), device=dev)
    t = lambda *a: torch.tensor(a, dtype=torch.int32, device=dev)      # noqa: E731
    bad = H.PrefixBatch(sp_subj=t(3, 50), sp_rel=t(2, 3), pos_row=t(0), pos_col=t(1), cand_first=2, n_cand=48)
    with pytest.raises(N.OkgeError):
        hp.score(E, R, "complex", bad)
    assert N.id_errors() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["all_entities_bce", "id_list_bce_smoothing_outputs", "all_entities_kl"])
def test_fast_training_call_is_the_general_path(okge_lib, monkeypatch, case):
    """AddLossModule's persistent-descriptor call for (lookup model, training mode, coordinate labels, int32 device ids) against the
    general path on the same inputs and dropout counters: loss and all_outputs bit for bit, gradients equal up to the order of the
    prefix backward's float atomics, over two steps"""
    from open_knowledge_graph_embeddings_amd import trainer as T
    from open_knowledge_graph_embeddings_amd.hotpath import positives_from_dense
    z = golden("g2_loss_complex_bce_all")
    n_ent = z["E"].shape[0]
    inputs = [(_dev(z["po_rel"]), _dev(z["po_obj"])), (_dev(z["sp_subj"]), _dev(z["sp_rel"]))]
    ids_case = case.startswith("id_list")
    cand = _dev(np.arange(2, n_ent, dtype=np.int32)[::2].copy().reshape(-1, 1)) if ids_case else _dev(z["cand"])
    n = cand.numel()
    labels = _dev(z["labels"])[:, :n].contiguous()
    coords = positives_from_dense(labels)
    loss_mod = torch.nn.KLDivLoss(reduction="sum") if case.endswith("kl") else torch.nn.BCEWithLogitsLoss(reduction="sum")
    out = {}
    for fast in (True, False):
        monkeypatch.setattr(T, "FAST_CALL", fast)
        m = _lookup_model(z, dropout=0.3)
        m.train()
        mod = T.AddLossModule(m, loss_mod, 0.1 if "smoothing" in case else 0.0, training_outputs="outputs" in case)
        seen = []
        for step in range(2):
            m.zero_grad(set_to_none=True)
            loss, hook, outs = mod(inputs=inputs, labels=coords, use_batch_shared_entities=ids_case, batch_shared_entities=cand, epoch=1,
                                   input_style_triple_or_prefix="right_and_left_prefix")
            (loss.sum() / float(labels.numel())).backward()
            seen.append((loss.detach().clone(), None if outs is None else outs.clone(), m.entity_embedding.weight.grad.clone(),
                         m.relation_embedding.weight.grad.clone()))
        assert (getattr(mod, "_fd", None) is not None) == fast           # the fast call ran / did not run
        out[fast] = seen
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a[0], b[0])
        assert (a[1] is None and b[1] is None) or torch.equal(a[1], b[1])
        # gradient rows that several batch rows add into with float atomics (every relation row; the few shared prefix entities)
        # differ in the last bits from run to run of EITHER path; everything else is bit-equal
        for x, y in ((a[2], b[2]), (a[3], b[3])):
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=2e-6 * float(y.abs().max()))
        assert float((a[2] != b[2]).float().mean()) < 0.02
    assert not torch.equal(out[True][0][2], out[True][1][2])             # (the dropout counter moved on between the steps)
