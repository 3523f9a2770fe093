"""The plugin methods called WITH gradients enabled, outside AddLossModule (a user's own loss on `sp_prefix_score` /
`po_prefix_score` / `forward` / `_score`): HIP forward + the GEMM / chain-rule backward of autograd_score.py, against the
reference's op sequence differentiated by ATen -- oracle/torch_twin.TwinModel (openkge/model.py:198-240, :268-278,
:455-480; pinned by tests/test_oracle_golden.py) in FLOAT64 on the CPU.  Tolerances: scores 2e-5 absolute (fp32 sums of
d <= 200 terms of size 0.1), gradients 1e-4 of the largest gradient entry."""
import numpy as np
import pytest
import torch

from oracle import torch_twin

pytestmark = pytest.mark.gpu


def _pair(kind, n_ent, n_rel, d, seed, dropout=0.0):
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    rng = np.random.default_rng(seed)
    E = (rng.standard_normal((n_ent, d)) * 0.3).astype(np.float32)
    R = (rng.standard_normal((n_rel, d)) * 0.3).astype(np.float32)
    name = "LookupComplexRelationModel" if kind == "complex" else "LookupDistmultRelationModel"
    m = getattr(Models, name)(entity_slot_size=d, init_std=0.1, sparse=False, input_dropout=dropout,
                              train_data=EntityRelationDatasetMeta(entities_size=n_ent, relations_size=n_rel))
    twin = torch_twin.TwinModel(kind, n_ent, n_rel, d).double()
    with torch.no_grad():
        m.entity_embedding.weight.copy_(torch.from_numpy(E))
        m.relation_embedding.weight.copy_(torch.from_numpy(R))
        twin.entity_embedding.weight.copy_(torch.from_numpy(E).double())
        twin.relation_embedding.weight.copy_(torch.from_numpy(R).double())
    return m.cuda(), twin, rng


def _ids(rng, lo, hi, n):
    return torch.from_numpy(rng.integers(lo, hi, (n, 1)).astype(np.int32))


def _compare_grads(m, twin):
    for name in ("entity_embedding", "relation_embedding"):
        got, ref = getattr(m, name).weight.grad, getattr(twin, name).weight.grad
        assert got is not None, name
        ref = ref.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-4 * np.abs(ref).max() + 1e-12, err_msg=name)


@pytest.mark.parametrize("kind,d", [("complex", 16), ("distmult", 24), ("complex", 200)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_prefix_scores_carry_a_graph(okge_lib, kind, d, mode):
    """own loss on sp_prefix_score (all objects) + po_prefix_score (all subjects): scores and both tables' gradients"""
    m, twin, rng = _pair(kind, 300, 12, d, seed=d)
    getattr(m, mode)()
    b = 9
    subj, rel_s, rel_o, obj = _ids(rng, 2, 300, b), _ids(rng, 2, 12, b), _ids(rng, 2, 12, b), _ids(rng, 2, 300, b)
    W1 = torch.from_numpy(rng.standard_normal((b, 298)))
    W2 = torch.from_numpy(rng.standard_normal((b, 298)))
    own = lambda x1, x2, w1, w2: (torch.tanh(x1) * w1).sum() + (x2 * x2 * w2).sum()       # noqa: E731
    x_sp = m.sp_prefix_score(subj.cuda(), rel_s.cuda())
    x_po = m.po_prefix_score(rel_o.cuda(), obj.cuda())
    assert x_sp.requires_grad and x_po.requires_grad and x_sp.shape == (b, 298)
    own(x_sp, x_po, W1.float().cuda(), W2.float().cuda()).backward()
    cand = twin.enc_ent(torch.arange(2, 300))
    r_sp = twin.score(twin.enc_ent(subj), twin.enc_rel(rel_s), cand, sp=True)
    r_po = twin.score(twin.enc_ent(obj), twin.enc_rel(rel_o), cand, sp=False)
    own(r_sp, r_po, W1, W2).backward()
    np.testing.assert_allclose(x_sp.detach().cpu().numpy(), r_sp.detach().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(x_po.detach().cpu().numpy(), r_po.detach().numpy(), rtol=0, atol=2e-5)
    _compare_grads(m, twin)
    # and without a graph the fused id path gives the same scores
    with torch.no_grad():
        y = m.sp_prefix_score(subj.cuda(), rel_s.cuda())
    assert not y.requires_grad
    np.testing.assert_allclose(y.cpu().numpy(), x_sp.detach().cpu().numpy(), rtol=0, atol=2e-6)


@pytest.mark.parametrize("kind", ["complex", "distmult"])
def test_shared_candidates_and_triples_carry_a_graph(okge_lib, kind):
    """precompute_batch_shared_inputs (a candidate id list WITH repeats: their gradients add up) -> *_prefix_score(many_*),
    `_score` on encoded rows, and forward(subj, rel, obj) (Hadamard triple scores)"""
    m, twin, rng = _pair(kind, 120, 9, 32, seed=5)
    m.train()
    b = 7
    subj, rel, obj = _ids(rng, 2, 120, b), _ids(rng, 2, 9, b), _ids(rng, 2, 120, b)
    cand_ids = torch.from_numpy(np.concatenate([rng.integers(2, 120, 40), [5, 5, 7]]).astype(np.int32))
    W = torch.from_numpy(rng.standard_normal((b, 43)))
    Wt = torch.from_numpy(rng.standard_normal((b, 1)))
    shared = m.precompute_batch_shared_inputs(cand_ids.cuda())
    assert shared.requires_grad
    x = m.sp_prefix_score(subj.cuda(), rel.cuda(), shared) + m.po_prefix_score(rel.cuda(), obj.cuda(), shared)
    x2 = m._score(m.encode_subj(subj.cuda()), m.encode_rel(rel.cuda()), shared, prefix=True, sp=True, po=False)
    t = m(subj.cuda(), rel.cuda(), obj.cuda())
    assert t.shape == (b, 1) and t.requires_grad
    ((x * W.float().cuda()).sum() + (x2 * x2).sum() + (t * Wt.float().cuda()).sum()).backward()
    c = twin.enc_ent(cand_ids)
    s, r, o = twin.enc_ent(subj), twin.enc_rel(rel), twin.enc_ent(obj)
    rx = twin.score(s, r, c, sp=True) + twin.score(o, r, c, sp=False)
    rx2 = twin.score(s, r, c, sp=True)
    if kind == "distmult":
        rt = (s * o * r).sum(1, keepdim=True)
    else:                                                                   # model.py:231-238
        r1, r2 = r.chunk(2, 1)
        o1, o2 = o.chunk(2, 1)
        rt = (torch.cat((s, s), 1) * torch.cat((o, o2, o1), 1) * torch.cat((r1, r, -r2), 1)).sum(1, keepdim=True)
    ((rx * W).sum() + (rx2 * rx2).sum() + (rt * Wt).sum()).backward()
    np.testing.assert_allclose(x.detach().cpu().numpy(), rx.detach().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(t.detach().cpu().numpy(), rt.detach().numpy(), rtol=0, atol=2e-5)
    _compare_grads(m, twin)
    with pytest.raises(Exception):
        m._score(shared, shared, shared, prefix=True)                        # neither sp nor po (model.py:217-218)


def test_dropout_masks_of_forward_and_backward_agree(okge_lib):
    """training mode with input dropout: the gradient must be the derivative of THE scores that were returned (same Philox
    masks in the backward).  The scores are linear in the entity table for fixed masks, so a finite difference along a
    random direction is exact up to rounding."""
    m, _, rng = _pair("complex", 200, 8, 64, seed=11, dropout=0.4)
    m.train()
    b = 16
    subj, rel = _ids(rng, 2, 200, b).cuda(), _ids(rng, 2, 8, b).cuda()
    W = torch.from_numpy(rng.standard_normal((b, 198)).astype(np.float32)).cuda()
    D = torch.from_numpy(rng.standard_normal((200, 64)).astype(np.float32)).cuda()

    def run():
        m.dropout_step = 41                                                  # the call advances it: same masks every time
        return m.sp_prefix_score(subj, rel)
    x = run()
    (x * W).sum().backward()
    g = m.entity_embedding.weight.grad.clone()
    # dropped components: rows of the gradient have exact zeros where the mask dropped the candidate's column
    assert 0.2 < float((g[2:] == 0).float().mean()) < 0.6
    # the scores are BILINEAR in the entity table (prefix entity x candidate): central difference of the quadratic is exact
    # (run() stays outside no_grad: the graph path and the fused id path number their dropout steps differently)
    eps = 1e-2
    E0 = m.entity_embedding.weight.detach().clone()
    m.entity_embedding.weight.data.copy_(E0 + eps * D)
    xp = run().detach().double()
    m.entity_embedding.weight.data.copy_(E0 - eps * D)
    xm = run().detach().double()
    m.entity_embedding.weight.data.copy_(E0)
    assert torch.equal(run().detach(), x.detach())                          # same masks: the same scores again
    fd = float(((xp - xm) * W.double()).sum() / (2 * eps))
    an = float((g.double() * D.double()).sum())
    assert abs(fd - an) <= 2e-4 * max(1.0, abs(an)), (fd, an)


@pytest.mark.parametrize("pool,bn", [("sum", True), ("mean", False), ("max", True)])
def test_token_pooled_methods_carry_a_graph(okge_lib, pool, bn):
    """UnigramPooling* models called with gradients enabled: pooling + batch-norm in torch ops (the reference's sequence,
    model.py:762-786), HIP scorer with the GEMM backward; against the same sequence in float64 on the CPU -- scores, the
    token tables' and the batch-norm parameters' gradients, and the running statistics the training-mode call leaves"""
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.token_pooled import UnigramPoolingComplexRelationModel
    rng = np.random.default_rng(len(pool))
    n_ent, n_rel, vt, vr, L, d, b = 60, 9, 40, 15, 4, 16, 6
    toks = lambda n, v: [[int(t) for t in rng.integers(4, v, int(rng.integers(1, L + 1)))] for _ in range(n)]   # noqa: E731
    et, rt = toks(n_ent, vt), toks(n_rel, vr)
    md = EntityRelationDatasetMeta(entities_size=n_ent, relations_size=n_rel, entity_tokens_size=vt, relation_tokens_size=vr,
                                   max_length=(L, L), entity_id_to_tokens_map=et, relation_id_to_tokens_map=rt)
    m = UnigramPoolingComplexRelationModel(entity_slot_size=d, relation_slot_size=d, train_data=md, pool=pool,
                                           normalize="batchnorm" if bn else None, dropout=0.0, init_std=0.3).cuda()
    m.train()
    subj, rel = _ids(rng, 2, n_ent, b), _ids(rng, 2, n_rel, b)
    cand_ids = torch.from_numpy(rng.integers(2, n_ent, 20).astype(np.int32))
    W = torch.from_numpy(rng.standard_normal((b, 20)))
    shared = m.precompute_batch_shared_inputs(cand_ids.cuda())
    x = m.sp_prefix_score(subj.cuda(), rel.cuda(), shared)
    assert x.requires_grad and x.shape == (b, 20)
    (torch.tanh(x) * W.float().cuda()).sum().backward()

    # the reference's op sequence, float64
    We = m.entity_embedding.weight.detach().cpu().double().requires_grad_()
    Wr = m.relation_embedding.weight.detach().cpu().double().requires_grad_()
    bns = {}
    for key, mod in (("e", m.entity_batchnorm), ("r", m.relation_batchnorm)):
        if mod is not None:
            ref = torch.nn.BatchNorm1d(d, momentum=mod.momentum, eps=mod.eps).double()
            ref.weight.data.copy_(mod.weight.detach().cpu().double())
            ref.bias.data.copy_(mod.bias.detach().cpu().double())
            bns[key] = ref.train()

    def enc(ids, table, tok, key):
        t = tok.cpu()[ids.reshape(-1).long()].long()
        e = torch.nn.functional.embedding(t, table, padding_idx=0)           # (model.py:588-589: the pad row gets no gradient)
        if pool == "max":
            e = e.max(dim=1)[0]
        elif pool == "mean":
            e = e.sum(1) / ((t > 0).double().sum(1, keepdim=True) + 1e-12)
        else:
            e = e.sum(1)
        return bns[key](e) if key in bns else e
    c = enc(cand_ids, We, m.entity_token_ids, "e")                           # the module's call order: candidates, subj, rel
    s = enc(subj, We, m.entity_token_ids, "e")
    r = enc(rel, Wr, m.relation_token_ids, "r")
    twin = torch_twin.TwinModel("complex", 4, 4, d)
    rx = twin.score(s, r, c, sp=True)
    (torch.tanh(rx) * W).sum().backward()
    np.testing.assert_allclose(x.detach().cpu().numpy(), rx.detach().numpy(), rtol=0, atol=5e-5)
    for got, ref, what in ((m.entity_embedding.weight.grad, We.grad, "We"), (m.relation_embedding.weight.grad, Wr.grad, "Wr")):
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4 * float(ref.abs().max()) + 1e-12, err_msg=what)
    for key, mod in (("e", m.entity_batchnorm), ("r", m.relation_batchnorm)):
        if mod is not None:
            np.testing.assert_allclose(mod.weight.grad.cpu().numpy(), bns[key].weight.grad.numpy(), rtol=0,
                                       atol=2e-4 * float(bns[key].weight.grad.abs().max()))
            np.testing.assert_allclose(mod.running_mean.cpu().numpy(), bns[key].running_mean.numpy(), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(mod.running_var.cpu().numpy(), bns[key].running_var.numpy(), rtol=1e-5, atol=1e-6)
    # no graph under no_grad, and the same scores from the HIP pooling kernels in eval mode vs the torch sequence in eval mode
    m.eval()
    with torch.no_grad():
        y0 = m.sp_prefix_score(subj.cuda(), rel.cuda(), m.precompute_batch_shared_inputs(cand_ids.cuda()))
    y1 = m.sp_prefix_score(subj.cuda(), rel.cuda(), m.precompute_batch_shared_inputs(cand_ids.cuda()))
    assert not y0.requires_grad and y1.requires_grad
    np.testing.assert_allclose(y1.detach().cpu().numpy(), y0.cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_direct_encode_calls_draw_fresh_masks(okge_lib):
    """encode_* / get_all_* called directly in training mode with gradients on: every call its own Philox mask, as each of
    the reference's _encode calls draws its own Bernoulli mask (model.py:455-470); in eval mode no dropout at all"""
    m, _, rng = _pair("complex", 200, 8, 64, seed=12, dropout=0.4)
    m.train()
    obj = _ids(rng, 2, 200, 32).cuda()
    a, b = m.encode_obj(obj), m.encode_obj(obj)
    za, zb = (a == 0), (b == 0)
    assert 0.25 < float(za.float().mean()) < 0.55 and bool((za != zb).any())          # dropped components differ between the calls
    c, d_ = m.get_all_obj(), m.get_all_obj()
    assert bool(((c == 0) != (d_ == 0)).any())
    (a.sum() + b.sum()).backward()                                                      # each backward replays ITS call's mask
    g = m.entity_embedding.weight.grad
    assert float(g.abs().sum()) > 0
    m.eval()
    assert torch.equal(m.encode_obj(obj), m.encode_obj(obj))
