"""Entity-sharded training step (open_knowledge_graph_embeddings_amd/sharded.py).

* CPU: two gloo ranks run the REAL exchange protocol (shard ranges, all-reduces, candidate column offsets, dropout
  keys) with the oracle standing in for the HIP kernels (tests/shard_engine_cpu.py); the concatenated result must
  equal the single-process oracle step.
* GPU: the three C-ABI phases (okge_encode_queries / okge_train_tiles / okge_prefix_backward) are driven for 2 and 3
  emulated shards on one device and compared with the fused single call and the oracle; plus ShardedTrainStep with a
  one-rank process group against FusedTrainStep.
"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import kge_oracle as ko

SCORER, N_ENT, N_REL, D, N_PO, N_SP = "complex", 301, 9, 24, 21, 20
P_DROP, SEED, LR = 0.4, 1234567, 0.3


def problem(step):
    rng = np.random.default_rng(100 + step)
    b = {"po_rel": rng.integers(2, N_REL, N_PO).astype(np.int32), "po_obj": rng.integers(2, N_ENT, N_PO).astype(np.int32),
         "sp_subj": rng.integers(2, N_ENT, N_SP).astype(np.int32), "sp_rel": rng.integers(2, N_REL, N_SP).astype(np.int32)}
    nc = N_ENT - 2
    y = np.zeros((N_PO + N_SP, nc), np.float32)
    for r in range(N_PO + N_SP):
        y[r, rng.choice(nc, size=int(rng.integers(1, 4)), replace=False)] = 1
    col, row = np.nonzero(y.T)
    b["pos_col"], b["pos_row"], b["labels"] = col.astype(np.int32), row.astype(np.int32), y
    return b


def tables():
    rng = np.random.default_rng(7)
    return (rng.standard_normal((N_ENT, D)) * 0.3).astype(np.float32), (rng.standard_normal((N_REL, D)) * 0.3).astype(np.float32)


def eval_problem(seed=5):
    """answer groups (some with several mentions), all-splits filter incl. the answers; global candidate columns"""
    rng = np.random.default_rng(seed)
    b = problem(seed)
    B, nc = N_PO + N_SP, N_ENT - 2
    row_ptr, grp_ptr, ids, filt_ptr, filt_col = [0], [0], [], [0], []
    for r in range(B):
        for _ in range(int(rng.integers(1, 4))):
            ids += rng.choice(nc, size=int(rng.integers(1, 4)), replace=False).tolist()
            grp_ptr.append(len(ids))
        row_ptr.append(len(grp_ptr) - 1)
        mine = ids[grp_ptr[row_ptr[r]]:]
        filt_col += sorted(set(mine) | set(rng.choice(nc, size=4, replace=False).tolist()))
        filt_ptr.append(len(filt_col))
    b.update(row_ptr=np.asarray(row_ptr, np.int64), grp_ptr=np.asarray(grp_ptr, np.int64), ids=np.asarray(ids, np.int32),
             filt_ptr=np.asarray(filt_ptr, np.int64), filt_col=np.asarray(filt_col, np.int32))
    return b


def oracle_ranks(b):
    E, R = tables()
    kind, C = ko.KIND_NAMES[SCORER], None
    C = E[2:]
    x = np.concatenate([ko.score_prefix(kind, ko.DIR_PO, E[b["po_obj"]], R[b["po_rel"]], C),
                        ko.score_prefix(kind, ko.DIR_SP, E[b["sp_subj"]], R[b["sp_rel"]], C)])
    filt = np.zeros(x.shape, bool)
    for r in range(x.shape[0]):
        filt[r, b["filt_col"][b["filt_ptr"][r]:b["filt_ptr"][r + 1]]] = True
    return ko.filtered_ranks(x, filt, b["row_ptr"], b["grp_ptr"], b["ids"])


def oracle_reference(nsteps, p=P_DROP, loss=ko.LOSS_BCE):
    from open_knowledge_graph_embeddings_amd import hotpath as H
    E, R = tables()
    sE, sR = np.zeros_like(E), np.zeros_like(R)
    losses = []
    for step in range(1, nsteps + 1):
        b = problem(step)
        kw = {}
        if p > 0:
            kw = dict(p_ent=p, keep_cand=ko.dropout_keep_mask(SEED, H.STREAM_CAND, step, N_ENT - 2, D, p),
                      keep_po_ent=ko.dropout_keep_mask(SEED, H.STREAM_PO_ENT, step, N_PO, D, p),
                      keep_sp_ent=ko.dropout_keep_mask(SEED, H.STREAM_SP_ENT, step, N_SP, D, p))
        out = ko.step_forward_backward(ko.KIND_NAMES[SCORER], E, R, (b["po_rel"], b["po_obj"]), (b["sp_subj"], b["sp_rel"]),
                                       np.arange(2, N_ENT), b["labels"], loss_kind=loss, **kw)
        ko.adagrad_step(E, out["dE"], sE, LR)
        ko.adagrad_step(R, out["dR"], sR, LR)
        losses.append(out["loss"])
    return E, R, losses


def to_batch(b, dev):
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    t = lambda a: torch.from_numpy(a).to(dev)      # noqa: E731
    return PrefixBatch(po_rel=t(b["po_rel"]), po_obj=t(b["po_obj"]), sp_subj=t(b["sp_subj"]), sp_rel=t(b["sp_rel"]),
                       pos_row=t(b["pos_row"]), pos_col=t(b["pos_col"]), cand_first=2, n_cand=N_ENT - 2)


# ------------------------------------------------------------------------------------------- CPU, gloo, 2 ranks
def _worker(rank, world, port, outdir, nsteps, loss, use_plan=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ShardedEvaluator, ShardedTrainStep, shard_range
    from shard_engine_cpu import OracleShardEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(minutes=5))
    E, R = tables()
    lo, hi = shard_range(N_ENT, world, rank)
    # evaluation on the initial tables: ranks must come out identical on every rank
    ev = ShardedEvaluator(torch.from_numpy(E[lo:hi].copy()), torch.from_numpy(R.copy()), SCORER, N_ENT,
                          engine=OracleShardEngine())
    eb = eval_problem()
    t = torch.from_numpy
    from open_knowledge_graph_embeddings_amd.sharded import make_exchange_plan
    eplan = make_exchange_plan(eb["po_obj"], eb["sp_subj"], N_ENT, world, "cpu") if use_plan else None
    ranks = ev.ranks(to_batch(eb, "cpu"), t(eb["filt_ptr"]), t(eb["filt_col"]), t(eb["row_ptr"]), t(eb["grp_ptr"]),
                     t(eb["ids"]), plan=eplan)                       # fused protocol: points / MAX / sweep / counts / SUM
    ranks_m = ev.ranks_materialised(to_batch(eb, "cpu"), t(eb["filt_ptr"]), t(eb["filt_col"]), t(eb["row_ptr"]),
                                    t(eb["grp_ptr"]), t(eb["ids"]))
    assert torch.equal(ranks, ranks_m)
    st = ShardedTrainStep(torch.from_numpy(E[lo:hi].copy()), torch.from_numpy(R.copy()), SCORER, N_ENT, lr=LR,
                          input_dropout=P_DROP, seed=SEED, engine=OracleShardEngine(), loss=loss)
    losses = []
    for step in range(1, nsteps + 1):
        pb = problem(step)
        plan = make_exchange_plan(pb["po_obj"], pb["sp_subj"], N_ENT, world, "cpu") if use_plan else None
        assert not use_plan or plan is not None
        from open_knowledge_graph_embeddings_amd.sharded import make_row_segments
        segs = make_row_segments(pb["po_rel"], pb["po_obj"], pb["sp_subj"], pb["sp_rel"], "cpu", min_rows=1,
                                 min_rows_per_relation=1.0) if use_plan else None
        st.step(to_batch(pb, "cpu"), plan=plan, rel_segments=segs)
        losses.append(float(st.reduce_loss()[0]))
    # checkpoint interop: shards gathered into the reference's state-dict layout, then scattered back
    from open_knowledge_graph_embeddings_amd.checkpoint import load_reference_checkpoint, save_checkpoint
    ck = save_checkpoint(os.path.join(outdir, "ckpt.pt"), st, epoch=1)
    st2 = ShardedTrainStep(torch.zeros(hi - lo, D), torch.zeros(N_REL, D), SCORER, N_ENT, engine=OracleShardEngine(), loss=loss)
    dist.barrier()
    load_reference_checkpoint(st2, os.path.join(outdir, "ckpt.pt"))
    assert torch.equal(st2.E, st.E) and torch.equal(st2.sumE, st.sumE) and torch.equal(st2.sumR, st.sumR)
    assert st2.steps == nsteps and st2.lr == LR
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), E=st.E.numpy(), R=st.R.numpy(), lo=lo, hi=hi,
             losses=np.asarray(losses), ranks=ranks.numpy(), E_full=ck["state_dict"]["entity_embedding.weight"].numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,loss,use_plan", [(2, "bce", False), (3, "bce", False), (2, "kl", False), (3, "bce", True),
                                                 (2, "kl", True)])
def test_sharded_exchange_protocol_gloo(world, loss, use_plan):
    """use_plan: exchange 1 as an all-gather of the rows each rank owns (host-built plan) instead of an all-reduce"""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    nsteps = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, port, outdir, nsteps, loss, use_plan), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]
    E_ref, R_ref, losses_ref = oracle_reference(nsteps, loss=ko.LOSS_KL if loss == "kl" else ko.LOSS_BCE)
    ranks_ref = oracle_ranks(eval_problem())
    for p in parts:                                       # sharded evaluation: exact integer ranks on every rank
        np.testing.assert_array_equal(p["ranks"], ranks_ref)
    E = np.concatenate([p["E"] for p in parts])
    for p in parts:
        np.testing.assert_array_equal(p["E_full"], E)                  # gathered checkpoint tables
    assert [int(p["lo"]) for p in parts] == sorted(int(p["lo"]) for p in parts) and E.shape == E_ref.shape
    close = np.isclose(E, E_ref, rtol=2e-4, atol=2e-5)
    assert close.mean() > 0.999 and np.abs(E - E_ref).max() < 5e-3      # see the note on Adagrad conditioning below
    for p in parts:                                       # relation table: replicated and identical everywhere
        np.testing.assert_allclose(p["R"], R_ref, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(p["losses"], losses_ref, rtol=1e-5)
    np.testing.assert_array_equal(parts[0]["R"], parts[1]["R"])


def replica_problem(rank, step):
    """per-rank batch with its own sampled candidate list (batch-shared 1-vs-N, duplicates allowed)"""
    rng = np.random.default_rng(1000 * rank + step)
    b = problem(10 * rank + step)
    cand = rng.integers(2, N_ENT, 96).astype(np.int32)
    y = np.zeros((N_PO + N_SP, len(cand)), np.float32)
    for r in range(N_PO + N_SP):
        y[r, rng.choice(len(cand), size=int(rng.integers(1, 4)), replace=False)] = 1
    col, row = np.nonzero(y.T)
    b.update(cand=cand, labels=y, pos_col=col.astype(np.int32), pos_row=row.astype(np.int32))
    return b


def _replica_worker(rank, world, port, outdir, nsteps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaTrainStep
    from shard_engine_cpu import OracleShardEngine
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(minutes=5))
    E, R = tables()
    st = ReplicaTrainStep(torch.from_numpy(E.copy()), torch.from_numpy(R.copy()), SCORER, lr=LR, engine=OracleShardEngine())
    t = torch.from_numpy
    losses = []
    for step in range(1, nsteps + 1):
        b = replica_problem(rank, step)
        batch = PrefixBatch(po_rel=t(b["po_rel"]), po_obj=t(b["po_obj"]), sp_subj=t(b["sp_subj"]), sp_rel=t(b["sp_rel"]),
                            pos_row=t(b["pos_row"]), pos_col=t(b["pos_col"]), cand_ids=t(b["cand"]))
        losses.append(float(st.step(batch)[0]))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), E=st.E.numpy(), R=st.R.numpy(), losses=np.asarray(losses))
    dist.destroy_process_group()


def test_replica_step_gloo():
    """Two replicas with different batches == one process averaging the two oracle gradients, then Adagrad."""
    import torch.multiprocessing as mp
    world, nsteps = 2, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_replica_worker, args=(world, port, outdir, nsteps), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]
    E, R = tables()
    sE, sR = np.zeros_like(E), np.zeros_like(R)
    losses = []
    for step in range(1, nsteps + 1):
        dE, dR, ls = np.zeros_like(E), np.zeros_like(R), 0.0
        for rank in range(world):
            b = replica_problem(rank, step)
            out = ko.step_forward_backward(ko.KIND_NAMES[SCORER], E, R, (b["po_rel"], b["po_obj"]), (b["sp_subj"], b["sp_rel"]),
                                           b["cand"], b["labels"], normalizer=float(world * b["labels"].size))
            dE += out["dE"]
            dR += out["dR"]
            ls += out["loss"]
        ko.adagrad_step(E, dE, sE, LR)
        ko.adagrad_step(R, dR, sR, LR)
        losses.append(ls)
    np.testing.assert_array_equal(parts[0]["E"], parts[1]["E"])          # replicas never drift
    np.testing.assert_array_equal(parts[0]["R"], parts[1]["R"])
    np.testing.assert_allclose(parts[0]["losses"], losses, rtol=1e-6)
    close = np.isclose(parts[0]["E"], E, rtol=2e-4, atol=2e-5)
    assert close.mean() > 0.999 and np.abs(parts[0]["E"] - E).max() < 5e-3
    np.testing.assert_allclose(parts[0]["R"], R, rtol=2e-4, atol=2e-5)


class _ToyInner:
    """the smallest object with the ReplicaStep protocol: gradient = batch-dependent tensor, one running statistic"""

    def __init__(self, rank):
        self.w = torch.zeros(5)
        self.g, self.g2 = torch.zeros(5), torch.zeros(3)
        self.stat = torch.full((2,), float(rank))
        self.seed = 7

    def grad_tensors(self):
        return [self.g, self.g2]

    def stat_tensors(self):
        return [self.stat]

    def rebind(self, grads, stats):
        self.g, self.g2 = grads
        (self.stat,) = stats

    def forward_backward(self, batch, normalizer):
        self.g.copy_(batch["x"] / normalizer)
        self.g2.fill_(1.0 / normalizer)
        self.stat += 1.0
        return torch.tensor([float(batch["x"].sum()) / normalizer], dtype=torch.float64)

    def optimizer_step(self):
        self.w -= self.g
        self.g.zero_()
        self.g2.zero_()


def _replica_step_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaStep
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(minutes=5))
    inner = _ToyInner(rank)
    st = ReplicaStep(inner)
    assert inner.seed == 7 + 1000003 * rank

    class B:                                                    # noqa: D401  (duck-typed batch)
        B, n_candidates = 2, 5

        def __getitem__(self, k):
            return torch.arange(5, dtype=torch.float32) * (rank + 1)
    loss = st.step(B())
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=inner.w.numpy(), stat=inner.stat.numpy(), loss=loss.numpy())
    dist.destroy_process_group()


class _ToySparseInner(_ToyInner):
    """+ one row-sparse table gradient: each batch touches the rows it names"""

    def __init__(self, rank):
        super().__init__(rank)
        self.table = torch.zeros((50, 4))
        self.gt = torch.zeros((50, 4))

    def grad_tensors(self):
        return [self.g, self.gt, self.g2]

    def sparse_grad_indices(self):
        return [1]

    def sparse_grad_rows(self, batch):
        return [(1, batch.rows)]

    def rebind(self, grads, stats):
        self.g, self.gt, self.g2 = grads
        (self.stat,) = stats

    def forward_backward(self, batch, normalizer):
        for r in batch.rows.reshape(-1).tolist():
            self.gt[r] += (r + 1.0) / normalizer
        return super().forward_backward(batch, normalizer)

    def optimizer_step(self):
        self.table -= self.gt
        self.gt.zero_()
        super().optimizer_step()


def _replica_sparse_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaStep
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(minutes=5))
    inner = _ToySparseInner(rank)
    st = ReplicaStep(inner)
    sent = []
    for step in range(3):
        class B:                                                    # noqa: D401  (duck-typed batch)
            B, n_candidates = 2, 5
            rows = torch.tensor([[0, 3 + rank, 7, 7], [40 + step, 3 + rank, 0, 11 * rank]])      # repeats, overlap, own rows

            def __getitem__(self, k):
                return torch.arange(5, dtype=torch.float32) * (rank + 1)
        st.step(B())
        sent.append(st.last_exchanged_elements)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), w=inner.w.numpy(), stat=inner.stat.numpy(), table=inner.table.numpy(),
             sent=np.asarray(sent), gt=inner.gt.numpy())
    dist.destroy_process_group()


def test_replica_step_sparse_rows_gloo():
    """touched-row exchange: only the union of the replicas' touched table rows travels (plus the small dense head);
    tables stay identical and equal to the dense sum"""
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_replica_sparse_worker, args=(world, port, outdir), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]
    np.testing.assert_array_equal(parts[0]["table"], parts[1]["table"])
    np.testing.assert_array_equal(parts[0]["w"], parts[1]["w"])
    np.testing.assert_array_equal(parts[0]["stat"], parts[1]["stat"])
    norm = 2 * 5 * world
    want = np.zeros((50, 4), np.float32)
    for step in range(3):
        for rank in range(world):
            for r in [0, 3 + rank, 7, 7, 40 + step, 3 + rank, 0, 11 * rank]:
                want[r] -= (r + 1.0) / norm
    np.testing.assert_allclose(parts[0]["table"], want, rtol=1e-6)
    assert not parts[0]["gt"].any()
    # union per step: rows {0, 3, 4, 7, 11, 40 + step} = 6 rows x 4 columns, + the dense head (5 + 3 -> 8 + 4, stat 2 -> 4)
    assert parts[0]["sent"].tolist() == [16 + 24] * 3 and 16 + 24 < 50 * 4


def test_replica_step_protocol_gloo():
    """ReplicaStep (used for the token-pooled models): summed gradients of the mean loss, averaged running statistics,
    identical parameters on every rank"""
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_replica_step_worker, args=(world, port, outdir), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]
    x = np.arange(5, dtype=np.float32)
    norm = 2 * 5 * world
    np.testing.assert_allclose(parts[0]["w"], -(x * 1 + x * 2) / norm, rtol=1e-6)
    np.testing.assert_array_equal(parts[0]["w"], parts[1]["w"])
    np.testing.assert_allclose(parts[0]["stat"], [(0 + 1 + 1 + 1) / 2.0] * 2)        # mean of (rank + 1) over the ranks
    np.testing.assert_array_equal(parts[0]["stat"], parts[1]["stat"])
    np.testing.assert_allclose(parts[0]["loss"], [(x.sum() * 1 + x.sum() * 2) / norm])


def test_row_segments_plan():
    from open_knowledge_graph_embeddings_amd.sharded import make_row_segments
    po_rel, sp_rel = np.asarray([5, 3, 5, 9]), np.asarray([3, 3, 7, 5, 3])
    po_obj, sp_subj = np.asarray([40, 41, 40, 42]), np.asarray([43, 41, 44, 45, 40])
    sg = make_row_segments(po_rel, po_obj, sp_subj, sp_rel, "cpu", min_rows=1, min_rows_per_relation=1.0)
    order, seg = sg.rel
    rel = np.concatenate([po_rel, sp_rel])
    assert order.dtype == torch.int32 and seg.tolist() == [0, 4, 7, 8, 9]
    assert order.tolist() == [1, 4, 5, 8, 0, 2, 7, 6, 3]                      # stable: rows ascending inside a relation
    assert [sorted(set(rel[order[a:b].numpy()].tolist())) for a, b in zip(seg[:-1].tolist(), seg[1:].tolist())] == [[3], [5], [7], [9]]
    e_order, e_seg = sg.ent
    assert e_order.tolist() == [0, 2, 8, 1, 5, 3, 4, 6, 7] and e_seg.tolist() == [0, 3, 5, 6, 7, 8, 9]
    # relations that hardly repeat keep their atomics, the entity plan stays; small batches get no plan at all
    sg2 = make_row_segments(po_rel, po_obj, sp_subj, sp_rel, "cpu", min_rows=1, min_rows_per_relation=4.0)
    assert sg2.rel is None and sg2.ent is not None
    assert make_row_segments(po_rel, po_obj, sp_subj, sp_rel, "cpu") is None
    assert make_row_segments(np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0), "cpu", min_rows=0) is None


@pytest.mark.gpu
@pytest.mark.parametrize("scorer,d,n_rel,which", [("complex", 200, 12, "both"), ("distmult", 100, 7, "both"), ("complex", 256, 40, "rel"),
                                                  ("complex", 64, 9, "ent")])
def test_prefix_backward_segmented_equals_atomics(okge_lib, scorer, d, n_rel, which):
    """okge_prefix_backward_segmented (gradient rows stored, one workgroup per relation / entity adds them up in the plan's
    order, one read-modify-write per table row) == okge_prefix_backward (float atomics): same dE, dR to rounding; bit-reproducible;
    entity rows of ANOTHER shard are left alone"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.sharded import RowSegments, make_row_segments
    hp = H.HotPath("cuda:0")
    rng = np.random.default_rng(d + n_rel)
    n_ent, n_po, n_sp = 500, 190, 150
    lo, hi = 100, 400                                                       # this "rank" owns entity ids [100, 400)
    Efull = (rng.standard_normal((n_ent, d)) * 0.3).astype(np.float32)
    E = torch.from_numpy(Efull[lo:hi].copy()).cuda()
    R = torch.from_numpy((rng.standard_normal((n_rel, d)) * 0.3).astype(np.float32)).cuda()
    ids = dict(po_rel=rng.integers(2, n_rel, n_po).astype(np.int32), po_obj=rng.integers(2, 60, n_po).astype(np.int32) * 8,
               sp_subj=rng.integers(2, 60, n_sp).astype(np.int32) * 8, sp_rel=rng.integers(2, n_rel, n_sp).astype(np.int32))
    t = lambda a: torch.from_numpy(a).cuda()      # noqa: E731
    batch = H.PrefixBatch(po_rel=t(ids["po_rel"]), po_obj=t(ids["po_obj"]), sp_subj=t(ids["sp_subj"]), sp_rel=t(ids["sp_rel"]))
    batch.drop_po_ent, batch.drop_sp_ent = H.DropoutSpec(0.3, 5, H.STREAM_PO_ENT, 2), H.DropoutSpec(0.3, 5, H.STREAM_SP_ENT, 2)
    batch.drop_po_rel, batch.drop_sp_rel = H.DropoutSpec(0.2, 5, H.STREAM_PO_REL, 2), H.DropoutSpec(0.2, 5, H.STREAM_SP_REL, 2)
    shard = H.Shard(lo, hi, 0)
    # the masked entity rows of ALL prefixes (what exchange 1 delivers): from a whole-table encode
    whole = H.Shard(0, n_ent, 0)
    er = hp.encode_entity_rows(torch.from_numpy(Efull).cuda(), R, scorer, batch, whole)
    dq = torch.from_numpy(rng.standard_normal(tuple(er.shape)).astype(np.float32)).cuda()
    sg = make_row_segments(ids["po_rel"], ids["po_obj"], ids["sp_subj"], ids["sp_rel"], "cuda:0", min_rows=1)
    assert sg.rel is not None and sg.ent is not None
    plan = {"both": sg, "rel": RowSegments(rel=sg.rel), "ent": RowSegments(ent=sg.ent)}[which]
    out = []
    for use in (None, plan, plan):
        dE, dR = torch.full_like(E, -0.5), torch.full_like(R, 0.25)           # (both accumulate onto what is there)
        hp.prefix_backward(E, R, scorer, batch, shard, dq, er, dE, dR, rel_segments=use)
        torch.cuda.synchronize()
        out.append((dE.cpu().numpy(), dR.cpu().numpy()))
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=2e-6 * np.abs(out[0][1]).max())
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=2e-6 * np.abs(out[0][0]).max())
    if which != "ent":
        np.testing.assert_array_equal(out[1][1], out[2][1])                  # fixed summation order: bit-reproducible
    if which != "rel":
        np.testing.assert_array_equal(out[1][0], out[2][0])
    assert np.abs(out[0][1] - 0.25).max() > 1e-3 and (out[0][1][:2] == 0.25).all()      # reserved relation rows untouched
    touched = np.unique(np.concatenate([ids["po_obj"], ids["sp_subj"]]))
    owned = touched[(touched >= lo) & (touched < hi)] - lo
    assert len(owned) > 5 and len(owned) < len(touched)                       # some prefixes live on other "ranks"
    mask = np.zeros(hi - lo, bool)
    mask[owned] = True
    assert (out[1][0][~mask] == -0.5).all() and (np.abs(out[1][0][mask] + 0.5).max(axis=1) > 0).all()


def test_exchange_plan():
    from open_knowledge_graph_embeddings_amd.sharded import make_exchange_plan, shard_range
    rng = np.random.default_rng(0)
    n_ent, world = 1000, 4
    po, sp = rng.integers(2, n_ent, 40), rng.integers(2, n_ent, 24)
    plan = make_exchange_plan(po, sp, n_ent, world, "cpu")
    ent = np.concatenate([po, sp])
    assert plan.cap * world < 2 * len(ent)
    seen = np.zeros(len(ent), bool)
    for r in range(world):
        lo, hi = shard_range(n_ent, world, r)
        own = plan.owned[r].numpy()
        n_own = int(((ent >= lo) & (ent < hi)).sum())
        assert ((ent[own[:n_own]] >= lo) & (ent[own[:n_own]] < hi)).all()            # its own rows first ...
        assert not ((ent[own[n_own:]] >= lo) & (ent[own[n_own:]] < hi)).any()        # ... padded with rows it does not own
        for k, b in enumerate(own[:n_own]):
            assert plan.slot[b] == r * plan.cap + k
            seen[b] = True
    assert seen.all()
    # all prefixes on one shard: the all-gather would move world x B rows -> fall back to the all-reduce
    assert make_exchange_plan(np.full(40, 5), np.full(24, 7), n_ent, world, "cpu") is None


def test_shard_ranges_cover_table():
    from open_knowledge_graph_embeddings_amd.sharded import shard_range
    for n, w in ((14543, 8), (301, 3), (10, 4), (2_500_000, 8)):
        r = [shard_range(n, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("world,loss", [(2, "bce"), (3, "bce"), (2, "kl"), (3, "kl")])
def test_three_phase_abi_emulated_shards(world, loss, okge_lib):
    """One device plays every rank in turn; the all-reduces are plain sums of the per-rank buffers."""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.sharded import shard_range
    hp = H.HotPath("cuda:0")
    E, R = tables()
    b = problem(1)
    step, p = 1, P_DROP
    kw = dict(p_ent=p, keep_cand=ko.dropout_keep_mask(SEED, H.STREAM_CAND, step, N_ENT - 2, D, p),
              keep_po_ent=ko.dropout_keep_mask(SEED, H.STREAM_PO_ENT, step, N_PO, D, p),
              keep_sp_ent=ko.dropout_keep_mask(SEED, H.STREAM_SP_ENT, step, N_SP, D, p))
    ref = ko.step_forward_backward(ko.KIND_NAMES[SCORER], E, R, (b["po_rel"], b["po_obj"]), (b["sp_subj"], b["sp_rel"]),
                                   np.arange(2, N_ENT), b["labels"],
                                   loss_kind=ko.LOSS_KL if loss == "kl" else ko.LOSS_BCE, **kw)
    batch = to_batch(b, "cuda:0")
    batch.drop_cand = H.DropoutSpec(p, SEED, H.STREAM_CAND, step)
    batch.drop_po_ent = H.DropoutSpec(p, SEED, H.STREAM_PO_ENT, step)
    batch.drop_sp_ent = H.DropoutSpec(p, SEED, H.STREAM_SP_ENT, step)
    Rt = torch.from_numpy(R).cuda()
    shards, Es = [], []
    for r in range(world):
        lo, hi = shard_range(N_ENT, world, r)
        c_lo = max(lo, 2)
        shards.append((H.Shard(lo, hi, c_lo - 2), c_lo - lo, hi - c_lo))
        Es.append(torch.from_numpy(E[lo:hi].copy()).cuda())
    qe = sum(hp.encode_queries(Es[r], Rt, SCORER, batch, shards[r][0]) for r in range(world))     # "all-reduce"
    # the production exchange: only the masked entity rows travel, every rank folds the queries itself -- bit-identical
    er = sum(hp.encode_entity_rows(Es[r], Rt, SCORER, batch, shards[r][0]) for r in range(world))
    assert torch.equal(er, qe[1]) and torch.equal(hp.fold_queries(Es[0], Rt, SCORER, batch, er), qe[0])
    dEs, dqs, losses = [], [], []

    def local_batch(r):
        return H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                             pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=shards[r][1], n_cand=shards[r][2],
                             drop_cand=batch.drop_cand)

    row_lse = None
    if loss == "kl":                                                 # "all-gather" + log-sum-exp over ranks
        row_lse = torch.logsumexp(torch.stack([hp.row_logsumexp(Es[r], Rt, SCORER, qe[0], batch.B, local_batch(r),
                                                                shards[r][0]) for r in range(world)]), dim=0).contiguous()
        np.testing.assert_allclose(row_lse.cpu().numpy(), ko.log_softmax(ref["outputs"])[:, 0] * -1 + ref["outputs"][:, 0],
                                   rtol=2e-6, atol=2e-6)
    for r in range(world):
        sh, first, n = shards[r]
        local = local_batch(r)
        dE, dq = torch.zeros_like(Es[r]), torch.empty_like(qe[0])
        losses.append(hp.train_tiles(Es[r], Rt, SCORER, qe[0], local, sh, dE, dq, N_ENT - 2, loss=loss,
                                     normalizer=float(batch.B) * (N_ENT - 2), grads_zero=True,
                                     row_lse=row_lse).clone())
        dEs.append(dE)
        dqs.append(dq)
    dq = sum(dqs)
    dRs = []
    for r in range(world):
        dR = torch.zeros_like(Rt)
        hp.prefix_backward(Es[r], Rt, SCORER, batch, shards[r][0], dq, qe[1], dEs[r], dR)
        dRs.append(dR)
    torch.cuda.synchronize()
    assert abs(sum(float(x[0]) for x in losses) - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    dE = torch.cat(dEs).cpu().numpy()
    np.testing.assert_allclose(dE, ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
    for dR in dRs:
        np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("world,loss,d", [(2, "bce", 320), (3, "kl", 512)])
def test_three_phase_abi_emulated_shards_wide_slots(world, loss, d, okge_lib, monkeypatch):
    """the same three phases at slot sizes above 256: the register-tile kernel (stream-K launch, shard-relative candidate
    columns for the Philox keys and the positives) and the 32-candidate-chunk dQ kernel inside okge_train_tiles"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "D", d)
    test_three_phase_abi_emulated_shards(world, loss, okge_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ranks_emulated_shards(world, okge_lib):
    """score_queries + group_true_scores ("all-reduce max") + rank_counts ("all-reduce sum") over emulated shards
    == the single-device score + filtered_ranks, bit for bit; and == the oracle up to float near-ties."""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.sharded import shard_range
    hp = H.HotPath("cuda:0")
    E, R = tables()
    eb = eval_problem()
    batch = to_batch(eb, "cuda:0")
    Et, Rt = torch.from_numpy(E).cuda(), torch.from_numpy(R).cuda()
    dev = lambda k: torch.from_numpy(eb[k]).cuda()      # noqa: E731
    fp, fc, rp, gp, ids = dev("filt_ptr"), dev("filt_col"), dev("row_ptr"), dev("grp_ptr"), dev("ids")
    whole = hp.filtered_ranks(hp.score(Et, Rt, SCORER, batch), fp, fc, rp, gp, ids).cpu().numpy()
    shards, xs = [], []
    for r in range(world):
        lo, hi = shard_range(N_ENT, world, r)
        c_lo = max(lo, 2)
        sh = H.Shard(lo, hi, c_lo - 2)
        shards.append(sh)
        xs.append((Et[lo:hi].contiguous(), c_lo - lo, hi - c_lo))
    qe = sum(hp.encode_queries(xs[r][0], Rt, SCORER, batch, shards[r]) for r in range(world))
    scores = []
    for r in range(world):
        local = H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                              cand_first=xs[r][1], n_cand=xs[r][2])
        scores.append(hp.score_queries(xs[r][0], Rt, SCORER, qe[0], batch.B, local, shards[r]))
    true = torch.stack([hp.group_true_scores(scores[r], shards[r].cand_col0, rp, gp, ids) for r in range(world)]).max(0).values
    counts = sum(hp.rank_counts(scores[r], shards[r].cand_col0, fp, fc, rp, true) for r in range(world))
    ranks = (counts[:, 0] + counts[:, 1] // 2).cpu().numpy()
    np.testing.assert_array_equal(ranks, whole)
    ref = oracle_ranks(eb)
    assert (ranks != ref).mean() < 0.01 and np.abs(ranks - ref).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_fused_ranks_emulated_shards(world, okge_lib):
    """okge_evaluate_fused_shard over emulated shards -- points -> "all-reduce max" -> sweep -> counts -> "all-reduce sum" -- gives
    the ranks of the single-device fused evaluation AND of score + filtered_ranks, bit for bit: no (B, N / world) score block"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.sharded import shard_range
    hp = H.HotPath("cuda:0")
    E, R = tables()
    eb = eval_problem()
    batch = to_batch(eb, "cuda:0")
    Et, Rt = torch.from_numpy(E).cuda(), torch.from_numpy(R).cuda()
    dev = lambda k: torch.from_numpy(eb[k]).cuda()      # noqa: E731
    fp, fc, rp, gp, ids = dev("filt_ptr"), dev("filt_col"), dev("row_ptr"), dev("grp_ptr"), dev("ids")
    whole = hp.filtered_ranks(hp.score(Et, Rt, SCORER, batch), fp, fc, rp, gp, ids).cpu().numpy()
    fused, _ = hp.evaluate_fused(Et, Rt, SCORER, batch, fp, fc, rp, gp, ids)
    np.testing.assert_array_equal(fused.cpu().numpy(), whole)
    n_groups = int(gp.numel()) - 1
    engines, shards, locals_, tables_ = [], [], [], []
    for r in range(world):
        lo, hi = shard_range(N_ENT, world, r)
        c_lo = max(lo, 2)
        shards.append(H.Shard(lo, hi, c_lo - 2))
        tables_.append(Et[lo:hi].contiguous())
        locals_.append(H.PrefixBatch(cand_first=c_lo - lo, n_cand=hi - c_lo))
        engines.append(H.HotPath("cuda:0"))                       # one workspace per "rank": it lives across the phases
    er = sum(hp.encode_entity_rows(tables_[r], Rt, SCORER, batch, shards[r]) for r in range(world))      # "exchange 1"
    Q = hp.fold_queries(tables_[0], Rt, SCORER, batch, er)
    trues = [torch.full((n_groups,), float("-inf"), device="cuda:0") for _ in range(world)]
    counts = [torch.zeros((n_groups, 2), dtype=torch.int64, device="cuda:0") for _ in range(world)]
    call = lambda ph, r, tr: engines[r].evaluate_fused_shard(ph, tables_[r], Rt, SCORER, Q, batch.B, locals_[r], shards[r],   # noqa: E731
                                                             N_ENT - 2, fp, fc, rp, gp, ids, tr, counts[r])
    for r in range(world):
        call(1, r, trues[r])
    true = torch.stack(trues).max(0).values                          # "all-reduce(MAX)"
    assert bool(torch.isfinite(true).all())
    for r in range(world):
        g = true.clone()
        call(2, r, g)
        call(4, r, g)
    total = sum(counts)                                              # "all-reduce(SUM)"
    ranks = (total[:, 0] + total[:, 1] // 2).cpu().numpy()
    np.testing.assert_array_equal(ranks, whole)
    # every shard saw only its own columns: no shard alone has all the true scores (the maxima really were exchanged)
    assert any(bool(torch.isinf(t_).any()) for t_ in trues) or world == 1


@pytest.mark.gpu
@pytest.mark.parametrize("world,d", [(2, 320), (3, 512)])
def test_sharded_fused_ranks_emulated_shards_wide_slots(world, d, okge_lib, monkeypatch):
    """the same over slot sizes 257 .. 512 (round 4: the counting mode of the register-tile kernel; until then these sizes fell
    back to a (B, N / world) score block)"""
    monkeypatch.setattr(sys.modules[__name__], "D", d)
    test_sharded_fused_ranks_emulated_shards(world, okge_lib)


@pytest.mark.gpu
def test_sharded_step_one_rank_equals_fused_step(okge_lib):
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ShardedTrainStep
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        E, R = tables()
        a = ShardedTrainStep(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), SCORER, N_ENT, lr=LR,
                             input_dropout=P_DROP, seed=SEED)
        f = FusedTrainStep(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), SCORER, lr=LR,
                           input_dropout=P_DROP, seed=SEED)
        for step in range(1, 4):
            la = float(a.step(to_batch(problem(step), "cuda:0"))[0])
            lf = float(f.step(to_batch(problem(step), "cuda:0"))[0])
            assert abs(la - lf) <= 1e-6 * abs(lf)
        np.testing.assert_allclose(a.E.cpu().numpy(), f.E.cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(a.R.cpu().numpy(), f.R.cpu().numpy(), rtol=1e-4, atol=1e-5)
        E_ref, R_ref, _ = oracle_reference(3)
        # trajectory vs the oracle: Adagrad's first step amplifies 1e-12 gradient noise where |g| ~ 1e-9
        # (tests/test_oracle_golden.py::adagrad_tol), so a handful of elements may sit further out
        close = np.isclose(a.E.cpu().numpy(), E_ref, rtol=1e-3, atol=1e-4)
        assert close.mean() > 0.999 and np.abs(a.E.cpu().numpy() - E_ref).max() < 5e-3
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_replica_step_one_rank_equals_fused_step(okge_lib):
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaTrainStep
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        E, R = tables()
        mk = lambda cls, **kw: cls(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), SCORER,   # noqa: E731
                                   lr=LR, input_dropout=P_DROP, seed=SEED, **kw)
        a, f = mk(ReplicaTrainStep), mk(FusedTrainStep)
        t = lambda x: torch.from_numpy(x).cuda()      # noqa: E731
        for step in range(1, 4):
            b = replica_problem(0, step)
            for st in (a, f):
                batch = PrefixBatch(po_rel=t(b["po_rel"]), po_obj=t(b["po_obj"]), sp_subj=t(b["sp_subj"]), sp_rel=t(b["sp_rel"]),
                                    pos_row=t(b["pos_row"]), pos_col=t(b["pos_col"]), cand_ids=t(b["cand"]))
                st.step(batch)
        torch.cuda.synchronize()
        # duplicate candidate ids accumulate with atomics: same sums, run-dependent order
        assert abs(float(a.loss_out[0]) - float(f.loss_out[0])) <= 1e-6 * abs(float(f.loss_out[0]))
        np.testing.assert_allclose(a.E.cpu().numpy(), f.E.cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(a.R.cpu().numpy(), f.R.cpu().numpy(), rtol=1e-4, atol=1e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_replica_step_token_pooled_one_rank(okge_lib):
    """sharded.ReplicaStep around TokenPooledTrainStep (BASELINE configs[4]'s multi-GPU mode) with a one-rank group ==
    the plain step: the gradients and running statistics live in the flat exchange buffer"""
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaStep
    from open_knowledge_graph_embeddings_amd.token_pooled import TokenPooledTrainStep, TokenSlot
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(3)
        vocab, n_ids, L, d = 90, 150, 5, 64
        We = (rng.standard_normal((vocab, d)) * 0.3).astype(np.float32)
        Wr = (rng.standard_normal((40, d)) * 0.3).astype(np.float32)
        te = rng.integers(0, vocab, (n_ids, L)).astype(np.int32)
        tr = rng.integers(0, 40, (30, L)).astype(np.int32)
        t = lambda x: torch.from_numpy(x).cuda()      # noqa: E731

        def make():
            e = TokenSlot(t(We.copy()), t(te), "sum", True, torch.linspace(0.5, 1.5, d).cuda(), torch.zeros(d).cuda())
            r = TokenSlot(t(Wr.copy()), t(tr), "sum", True, torch.linspace(0.5, 1.5, d).cuda(), torch.zeros(d).cuda())
            return TokenPooledTrainStep(e, r, "complex", lr=0.1, dropout=0.0, seed=5)
        plain, inner = make(), make()
        rep = ReplicaStep(inner)
        inner.seed = plain.seed                                   # rank 0 offsets the seed by 0 anyway
        for step in range(3):
            r2 = np.random.default_rng(50 + step)
            b, nc = 24, 64
            cand = r2.permutation(np.arange(2, n_ids))[:nc].astype(np.int32)
            rows = np.arange(2 * b, dtype=np.int32)
            cols = np.sort(r2.integers(0, nc, 2 * b)).astype(np.int32)
            mk = lambda: PrefixBatch(po_rel=t(r2.integers(2, 30, b).astype(np.int32)), po_obj=t(r2.integers(2, n_ids, b).astype(np.int32)),  # noqa: E731
                                     sp_subj=t(r2.integers(2, n_ids, b).astype(np.int32)), sp_rel=t(r2.integers(2, 30, b).astype(np.int32)),
                                     pos_row=t(rows), pos_col=t(cols), cand_ids=t(cand))
            batch = mk()
            la = float(plain.step(batch)[0])
            lb = float(rep.step(batch)[0])
            assert abs(la - lb) <= 1e-6 * abs(la)
        plain.flush()
        rep.flush()
        torch.cuda.synchronize()
        for a, b_ in ((plain.entity, inner.entity), (plain.relation, inner.relation)):
            np.testing.assert_allclose(a.W.cpu().numpy(), b_.W.cpu().numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(a.bn.cpu().numpy(), b_.bn.cpu().numpy(), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(a.running_mean.cpu().numpy(), b_.running_mean.cpu().numpy(), rtol=1e-5, atol=1e-6)
        lo, hi = rep.flat.data_ptr(), rep.flat.data_ptr() + 4 * rep.flat.numel()
        assert lo <= inner.entity.d_bn.data_ptr() < hi and lo <= inner.relation.running_var.data_ptr() < hi
        assert not (lo <= inner.entity.dW.data_ptr() < hi)          # token tables: row-sparse exchange, not in the flat head
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_replica_step_keeps_module_running_statistics(okge_lib):
    """ReplicaStep around model.train_step(): rebind moves the step's running statistics into the exchange buffer; the
    MODULE's BatchNorm buffers (what eval-mode encode, state_dict and checkpoints read) must keep following them"""
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaStep
    from open_knowledge_graph_embeddings_amd.token_pooled import UnigramPoolingComplexRelationModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(11)
        n_ent, n_rel, vocab, L, d = 120, 20, 60, 4, 64
        md = EntityRelationDatasetMeta(entities_size=n_ent, relations_size=n_rel, entity_tokens_size=vocab, relation_tokens_size=vocab,
                                       max_length=(L, L),
                                       entity_id_to_tokens_map=[rng.integers(1, vocab, L).tolist() for _ in range(n_ent)],
                                       relation_id_to_tokens_map=[rng.integers(1, vocab, L).tolist() for _ in range(n_rel)])
        m = UnigramPoolingComplexRelationModel(entity_slot_size=d, relation_slot_size=d, train_data=md, pool="sum",
                                               normalize="batchnorm", dropout=0.0, init_std=0.3).cuda()
        rep = ReplicaStep(m.train_step(lr=0.1))
        t = lambda x: torch.from_numpy(x).cuda()      # noqa: E731
        for step in range(3):
            b, nc = 16, 64
            rows = np.arange(2 * b, dtype=np.int32)
            cols = np.sort(rng.integers(0, nc, 2 * b)).astype(np.int32)
            rep.step(PrefixBatch(po_rel=t(rng.integers(2, n_rel, b).astype(np.int32)), po_obj=t(rng.integers(2, n_ent, b).astype(np.int32)),
                                 sp_subj=t(rng.integers(2, n_ent, b).astype(np.int32)), sp_rel=t(rng.integers(2, n_rel, b).astype(np.int32)),
                                 pos_row=t(rows), pos_col=t(cols), cand_ids=t(rng.permutation(np.arange(2, n_ent))[:nc].astype(np.int32))))
        torch.cuda.synchronize()
        inner = rep.inner
        for sl, bn in ((inner.entity, m.entity_batchnorm), (inner.relation, m.relation_batchnorm)):
            assert float((bn.running_mean - 0).abs().max()) > 0 and float((bn.running_var - 1).abs().max()) > 0
            np.testing.assert_array_equal(bn.running_mean.cpu().numpy(), sl.running_mean.cpu().numpy())
            np.testing.assert_array_equal(bn.running_var.cpu().numpy(), sl.running_var.cpu().numpy())
            np.testing.assert_array_equal(bn.weight.data.cpu().numpy(), sl.bn_weight.cpu().numpy())
        assert "entity_batchnorm.running_mean" in m.state_dict()
    finally:
        dist.destroy_process_group()


def _nccl_worker(rank, world, port, outdir, nsteps, loss):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ShardedEvaluator, ShardedTrainStep, make_exchange_plan, shard_range
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    # two devices are visible: from here on ANY failure -- bring-up included -- fails the test (a dead peer ends in the
    # collective timeout, not in a hang)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev,
                            timeout=datetime.timedelta(minutes=5))
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    assert float(probe[0]) == world
    E, R = tables()
    lo, hi = shard_range(N_ENT, world, rank)
    t = lambda a: torch.from_numpy(a).to(dev)      # noqa: E731
    ev = ShardedEvaluator(t(E[lo:hi].copy()), t(R.copy()), SCORER, N_ENT)
    eb = eval_problem()
    csr = (t(eb["filt_ptr"]), t(eb["filt_col"]), t(eb["row_ptr"]), t(eb["grp_ptr"]), t(eb["ids"]))
    ranks = ev.ranks(to_batch(eb, dev), *csr)
    # ... and with exchange 1 as the all-gather of owned rows (the host-built plan)
    ranks_plan = ev.ranks(to_batch(eb, dev), *csr, plan=make_exchange_plan(eb["po_obj"], eb["sp_subj"], N_ENT, world, dev))
    st = ShardedTrainStep(t(E[lo:hi].copy()), t(R.copy()), SCORER, N_ENT, lr=LR, input_dropout=P_DROP, seed=SEED, loss=loss)
    losses = []
    for step in range(1, nsteps + 1):
        pb = problem(step)
        plan = make_exchange_plan(pb["po_obj"], pb["sp_subj"], N_ENT, world, dev) if step % 2 == 0 else None
        st.step(to_batch(pb, dev), plan=plan)
        losses.append(float(st.reduce_loss()[0]))
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), E=st.E.cpu().numpy(), R=st.R.cpu().numpy(), lo=lo, hi=hi,
             losses=np.asarray(losses), ranks=ranks.cpu().numpy(), ranks_plan=ranks_plan.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["bce", "kl"])
def test_sharded_step_two_ranks_rccl(okge_lib, loss):
    """The real thing: two processes, two GPUs, RCCL -- entity table row-sharded, both exchanges (all-gather plan on even
    steps, all-reduce on odd ones), sharded evaluation with and without the plan.  Skipped ONLY on one-GPU boxes: once two
    devices are visible, an RCCL bring-up or collective error is a failure, not a skip."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    world, nsteps = 2, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_nccl_worker, args=(world, port, outdir, nsteps, loss), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]
    E_ref, R_ref, losses_ref = oracle_reference(nsteps, loss=ko.LOSS_KL if loss == "kl" else ko.LOSS_BCE)
    ranks_ref = oracle_ranks(eval_problem())
    for p in parts:
        assert (p["ranks"] != ranks_ref).mean() < 0.01 and np.abs(p["ranks"] - ranks_ref).max() <= 1
        np.testing.assert_array_equal(p["ranks"], p["ranks_plan"])       # the plan moves the same rows: same bits
    np.testing.assert_array_equal(parts[0]["ranks"], parts[1]["ranks"])
    E = np.concatenate([p["E"] for p in parts])
    close = np.isclose(E, E_ref, rtol=1e-3, atol=1e-4)
    assert close.mean() > 0.999 and np.abs(E - E_ref).max() < 5e-3
    for p in parts:
        np.testing.assert_allclose(p["R"], R_ref, rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(p["losses"], losses_ref, rtol=3e-5)


def _token_pooled_replica_problem(rank, step, n_ids=300, n_rel=30):
    r2 = np.random.default_rng(900 + 10 * step + rank)
    b, nc = 24, 64
    return dict(cand=r2.permutation(np.arange(2, n_ids))[:nc].astype(np.int32), rows=np.arange(2 * b, dtype=np.int32),
                cols=np.sort(r2.integers(0, nc, 2 * b)).astype(np.int32), po_rel=r2.integers(2, n_rel, b).astype(np.int32),
                po_obj=r2.integers(2, n_ids, b).astype(np.int32), sp_subj=r2.integers(2, n_ids, b).astype(np.int32),
                sp_rel=r2.integers(2, n_rel, b).astype(np.int32))


def _token_pooled_tables(n_ids=300, n_rel=30, vocab=80, L=4, d=64):
    rng = np.random.default_rng(77)
    tok = lambda n: np.concatenate([rng.integers(1, vocab, (n, L - 1)), np.zeros((n, 1), np.int64)], axis=1).astype(np.int32)   # noqa: E731
    return dict(We=(rng.standard_normal((vocab, d)) * 0.3).astype(np.float32), Wr=(rng.standard_normal((vocab, d)) * 0.3).astype(np.float32),
                te=tok(n_ids), tr=tok(n_rel), bw=rng.random(d).astype(np.float32), bb=(rng.standard_normal(d) * 0.1).astype(np.float32))


def _nccl_replica_worker(rank, world, port, outdir, nsteps, backend="nccl"):
    """ReplicaStep(sparse=True) around the token-pooled step on real streams: side-stream mask all-reduce, packed touched-row
    exchange, the touched-row map of the optimizer sweep stamped for the OTHER replica's rows"""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch
    from open_knowledge_graph_embeddings_amd.sharded import ReplicaStep
    from open_knowledge_graph_embeddings_amd.token_pooled import TokenPooledTrainStep, TokenSlot
    dev = torch.device("cuda", rank if backend == "nccl" else 0)          # gloo: both ranks share the one GPU of the box
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev,
                                timeout=datetime.timedelta(minutes=5))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                timeout=datetime.timedelta(minutes=5))
    z = _token_pooled_tables()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    e = TokenSlot(t(z["We"]), t(z["te"]), "sum", True, t(z["bw"]), t(z["bb"]))
    r = TokenSlot(t(z["Wr"]), t(z["tr"]), "sum", True, t(z["bw"]), t(z["bb"]))
    rep = ReplicaStep(TokenPooledTrainStep(e, r, "complex", lr=0.1, dropout=0.0, seed=5), sparse=True)
    sent = []
    for step in range(nsteps):
        p = _token_pooled_replica_problem(rank, step)
        rep.step(PrefixBatch(po_rel=t(p["po_rel"]), po_obj=t(p["po_obj"]), sp_subj=t(p["sp_subj"]), sp_rel=t(p["sp_rel"]),
                             pos_row=t(p["rows"]), pos_col=t(p["cols"]), cand_ids=t(p["cand"])))
        sent.append(rep.last_exchanged_elements)
    rep.flush()
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), We=e.W.cpu().numpy(), Wr=r.W.cpu().numpy(), bn=e.bn.cpu().numpy(),
             rm=e.running_mean.cpu().numpy(), sent=np.asarray(sent), dense=e.W.numel() + r.W.numel())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_replica_step_sparse_two_ranks_rccl(okge_lib):
    """token-pooled replicas over RCCL: identical tables on both ranks after three steps, equal to ONE replica stepping on
    the two batches' summed gradients (float64 oracle, loose: Adagrad conditioning), fewer floats on the wire than the tables"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    world, nsteps = 2, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_nccl_replica_worker, args=(world, port, outdir, nsteps), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, f"rank{r}.npz")) for r in range(world)]
    for k in ("We", "Wr", "bn", "rm"):
        np.testing.assert_array_equal(parts[0][k], parts[1][k])         # the same exchanged sums, the same sweep: no drift
    assert (parts[0]["sent"] < parts[0]["dense"]).all() and (parts[0]["sent"] == parts[1]["sent"]).all()
    z = _token_pooled_tables()
    assert np.abs(parts[0]["We"] - z["We"]).max() > 1e-3                # the tables moved


@pytest.mark.gpu
def test_replica_step_sparse_two_ranks_one_gpu_deferred_decay(okge_lib, monkeypatch):
    """the same two-rank token-pooled replica run over gloo with both ranks on ONE GPU (runs on every box), seven steps, once
    with the deferred decay (default window 8) and once with every row every step: a replica catches up only the rows of ITS
    batch before the forward; the other replica's rows arrive stamped through the exchange and take what they owe inside the
    update -- both ranks must hold identical tables, and the deferred run must equal the eager one bit for bit after flush()"""
    import torch.multiprocessing as mp
    world, nsteps = 2, 7
    runs = {}
    for window in ("8", "1"):
        monkeypatch.setenv("OKGE_LAZY_DECAY", window)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        with tempfile.TemporaryDirectory() as outdir:
            mp.spawn(_nccl_replica_worker, args=(world, port, outdir, nsteps, "gloo"), nprocs=world, join=True)
            parts = [dict(np.load(os.path.join(outdir, f"rank{r}.npz"))) for r in range(world)]
        for k in ("We", "Wr", "bn", "rm"):
            np.testing.assert_array_equal(parts[0][k], parts[1][k])
        runs[window] = parts[0]
    for k in ("We", "Wr", "bn", "rm"):
        np.testing.assert_array_equal(runs["8"][k], runs["1"][k])
    z = _token_pooled_tables()
    assert np.abs(runs["8"]["We"] - z["We"]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["bce", "kl"])
def test_sharded_step_rccl_one_rank_rehearsal(okge_lib, loss, monkeypatch):
    """the exchange path (all-gather plan / all-reduce, cross-rank log-sum-exp, dQ all-reduce) through the RCCL backend
    with a one-rank group: every collective call the multi-GPU run makes is issued and must leave the step equal to the
    fused single-device step"""
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ShardedTrainStep, make_exchange_plan
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    monkeypatch.setenv("OKGE_SHARDED_FORCE_EXCHANGE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        E, R = tables()
        a = ShardedTrainStep(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), SCORER, N_ENT, lr=LR,
                             input_dropout=P_DROP, seed=SEED, loss=loss)
        assert a.force_exchange
        f = FusedTrainStep(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), SCORER, lr=LR,
                           input_dropout=P_DROP, seed=SEED, loss=loss)
        for step in range(1, 4):
            pb = problem(step)
            plan = make_exchange_plan(pb["po_obj"], pb["sp_subj"], N_ENT, 1, dev) if step % 2 == 0 else None
            la = float(a.step(to_batch(pb, "cuda:0"), plan=plan)[0])
            lf = float(f.step(to_batch(pb, "cuda:0"))[0])
            assert abs(la - lf) <= 2e-6 * abs(lf)
        np.testing.assert_allclose(a.E.cpu().numpy(), f.E.cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(a.R.cpu().numpy(), f.R.cpu().numpy(), rtol=1e-4, atol=1e-5)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["bce", "kl"])
def test_sharded_step_captured_in_a_hip_graph_rccl_one_rank(okge_lib, loss, monkeypatch):
    """the SHARDED step -- RCCL collectives included (exchange 1 as all-reduce, the KL all-gather, the dQ all-reduce) -- captured
    once in a HIP graph (train_step.GraphedTrainStep) and replayed: same losses and tables as the same step launched from
    Python, fresh dropout masks per replay (device-side step counter).  One-rank group with the exchange path forced: every
    collective call of a multi-GPU run is in the graph.  Also prints replay against launch time."""
    import time
    import torch.distributed as dist
    from open_knowledge_graph_embeddings_amd.sharded import ShardedTrainStep
    from open_knowledge_graph_embeddings_amd.train_step import GraphedTrainStep
    monkeypatch.setenv("OKGE_SHARDED_FORCE_EXCHANGE", "1")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        E, R = tables()
        mk = lambda: ShardedTrainStep(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), SCORER, N_ENT, lr=LR,   # noqa: E731
                                      input_dropout=P_DROP, seed=SEED, loss=loss)
        plain, inner = mk(), mk()
        assert plain.force_exchange and inner.force_exchange
        batches = [to_batch(problem(step), "cuda:0") for step in range(1, 5)]
        graphed = GraphedTrainStep(inner, batches[0], pos_capacity=max(b.nnz for b in batches) + 5)
        np.testing.assert_array_equal(inner.E.cpu().numpy(), E)            # the capture warm-up left no trace
        for i in range(6):
            b = batches[i % 4]
            lp, lg = float(plain.step(b)[0]), float(graphed.step(b)[0])
            assert abs(lp - lg) <= 2e-6 * abs(lp), (i, lp, lg)
        for a, b2 in ((inner.E, plain.E), (inner.R, plain.R)):           # (float atomics in the prefix scatter: order noise)
            a, b2 = a.cpu().numpy(), b2.cpu().numpy()
            assert np.isclose(a, b2, rtol=1e-4, atol=1e-5).mean() > 0.9999 and np.abs(a - b2).max() < 1e-3
        times = {}
        for name, fn in (("launched", lambda b: plain.step(b)), ("replayed", lambda b: graphed.step(b))):
            for b in batches:
                fn(b)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                fn(batches[i % 4])
            torch.cuda.synchronize()
            times[name] = 1e3 * (time.perf_counter() - t0) / 200
        print(f"[hip sharded graph, {loss}] ms/step launched {times['launched']:.4f}, replayed {times['replayed']:.4f}")
    finally:
        dist.destroy_process_group()
