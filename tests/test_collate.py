"""Batch producer: the host-side C-ABI collator (csrc/okge_collate.cpp) through
open_knowledge_graph_embeddings_amd.dataset.OneToNBatchProducer against the reference's own collate outputs
(tests/golden/g4_collate_toy.npz, 32 cases) and against the oracle restatement on random tables.  Host code only:
runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import golden
from oracle import kge_oracle as ko


def producer(z, training, shared, min_size, **kw):
    from open_knowledge_graph_embeddings_amd.dataset import OneToNBatchProducer
    return OneToNBatchProducer(z["prefixes"], z["seen"], z["all_splits"], int(z["n_ent"]), int(z["offset"]),
                               is_training_data=training, use_batch_shared_entities=shared,
                               min_size_batch_labels=min_size, device="cpu", **kw)


def cases():
    z = golden("g4_collate_toy")
    for c in range(int(z["n_cases"])):
        t = f"c{c}_"
        shared, min_size, training = (int(x) for x in z[t + "cfg"])
        yield z, t, bool(shared), min_size, bool(training)


def coords(b):
    """(row, col) label pairs in row-major order, like label_tensor.nonzero()"""
    rc = np.stack([b.pos_row.numpy(), b.pos_col.numpy()], axis=1)
    return rc[np.lexsort((rc[:, 1], rc[:, 0]))]


@pytest.mark.both
def test_collate_matches_reference_cases(okge_lib):
    n = 0
    for z, t, shared, min_size, training in cases():
        out = producer(z, training, shared, min_size).collate(z[t + "rows"], seed=3)
        b = out.batch
        ref_cand = z[t + "cand"]
        if shared:
            n_seen = len(ko.collate_batch(z["prefixes"][z[t + "rows"]], z["seen"], z["all_splits"], int(z["n_ent"]),
                                          int(z["offset"]), training, True, 0)["cand"])
            got = b.cand_ids.numpy()
            assert len(got) == len(ref_cand)
            np.testing.assert_array_equal(got[:n_seen], ref_cand[:n_seen])       # first-seen order of the answers
            fill = got[n_seen:]                                                  # sampled fill-up: order is ours
            assert len(set(fill.tolist())) == len(fill) and not set(fill.tolist()) & set(got[:n_seen].tolist())
            assert fill.size == 0 or (fill.min() >= int(z["offset"]) and fill.max() < int(z["n_ent"]))
        else:
            assert b.cand_ids is None and b.cand_first == int(z["offset"]) and out.n_cand == len(ref_cand)
        for name, cols in (("po", (b.po_rel, b.po_obj)), ("sp", (b.sp_subj, b.sp_rel))):
            got = np.zeros((0, 2), np.int32) if cols[0] is None else np.stack([c.numpy() for c in cols], axis=1)
            np.testing.assert_array_equal(got, z[t + name])
        np.testing.assert_array_equal(coords(b), z[t + "labels"])
        col = b.pos_col.numpy()
        assert (np.diff(col) >= 0).all()                                         # okge_positives: sorted by column
        assert [out.normalizer_loss, out.normalizer_metric] == z[t + "norm"].tolist()
        assert (b.B, out.n_cand) == tuple(z[t + "shape"])
        np.testing.assert_array_equal(out.dense_labels().nonzero().numpy(), z[t + "labels"])
        if training:
            assert out.row_ptr is None and out.filt_ptr is None
        else:
            np.testing.assert_array_equal(out.row_ptr.numpy(), z[t + "row_ptr"])
            np.testing.assert_array_equal(out.grp_ptr.numpy(), z[t + "grp_ptr"])
            np.testing.assert_array_equal(out.ids.numpy(), z[t + "ids"])
            fp, fc = out.filt_ptr.numpy(), out.filt_col.numpy()
            filt = np.stack([np.repeat(np.arange(b.B), np.diff(fp)), fc], axis=1).astype(np.int32)
            np.testing.assert_array_equal(filt, z[t + "filter"])
        n += 1
    assert n == 32


def random_tables(rng, n_ent, n_prefix, max_groups=6, max_mentions=3):
    from open_knowledge_graph_embeddings_amd.dataset import pack_groups
    seen, allsp, rows = [], [], []
    for _ in range(n_prefix):
        groups = [rng.integers(2, n_ent, size=int(rng.integers(1, max_mentions + 1))).tolist()
                  for _ in range(int(rng.integers(1, max_groups + 1)))]
        packed = pack_groups(groups).tolist()
        assert packed == ko.pack_groups(groups)
        everything = list(dict.fromkeys([e for g in groups for e in g] + rng.integers(2, n_ent, size=5).tolist()))
        slot, rel, ent = int(rng.choice([0, 2])), int(rng.integers(2, 50)), int(rng.integers(2, n_ent))
        rows.append([rel if slot == 0 else ent, ent if slot == 0 else rel, len(seen), len(seen) + len(packed),
                     len(allsp), len(allsp) + len(everything), slot])          # slot 0: (rel, obj), slot 2: (subj, rel)
        seen += packed
        allsp += everything
    return dict(prefixes=np.asarray(rows, np.int32), seen=np.asarray(seen, np.int32),
                all_splits=np.asarray(allsp, np.int32), n_ent=n_ent, offset=2)


@pytest.mark.parametrize("shared,training", [(False, True), (False, False), (True, True), (True, False)])
def test_collate_matches_oracle_random(okge_lib, shared, training):
    rng = np.random.default_rng(5 + 2 * shared + training)
    z = random_tables(rng, 5000, 700)
    p = producer(z, training, shared, 0)
    for _ in range(5):
        rows = rng.choice(700, size=int(rng.integers(1, 257)), replace=False)
        out = p.collate(rows)
        ref = ko.collate_batch(z["prefixes"][rows], z["seen"], z["all_splits"], 5000, 2, training, shared, 0)
        np.testing.assert_array_equal(coords(out.batch), np.asarray(ref["labels"], np.int32).reshape(-1, 2))
        if shared:
            np.testing.assert_array_equal(out.batch.cand_ids.numpy(), ref["cand"])
        if not training:
            ids = [i for row in ref["groups"] for g in row for i in g]
            np.testing.assert_array_equal(out.ids.numpy(), np.asarray(ids, np.int32))
            np.testing.assert_array_equal(out.filt_col.numpy(), np.asarray([c for f in ref["filters"] for c in f], np.int32))
            np.testing.assert_array_equal(np.diff(out.filt_ptr.numpy()), [len(f) for f in ref["filters"]])
            np.testing.assert_array_equal(np.diff(out.row_ptr.numpy()), [len(r) for r in ref["groups"]])


def test_fill_up_negatives(okge_lib):
    z = random_tables(np.random.default_rng(9), 3000, 64)
    p = producer(z, True, True, 1024)
    a, b2, c = p.collate(np.arange(32), seed=1), p.collate(np.arange(32), seed=1), p.collate(np.arange(32), seed=2)
    ids = a.batch.cand_ids.numpy()
    assert len(ids) == 1024 == len(set(ids.tolist())) and ids.min() >= 2 and ids.max() < 3000
    np.testing.assert_array_equal(ids, b2.batch.cand_ids.numpy())                # same seed, same list
    assert (ids != c.batch.cand_ids.numpy()).any()
    assert a.normalizer_loss == 32 * 1024


def test_epoch_iteration_and_errors(okge_lib):
    from open_knowledge_graph_embeddings_amd import OkgeError
    z = random_tables(np.random.default_rng(11), 400, 103)
    p = producer(z, True, False, 0, batch_size=16, shuffle=True, seed=5)
    seen_rows = 0
    batches = list(p)
    assert len(batches) == len(p) == 103 // 16
    for cb in batches:
        seen_rows += cb.batch.B
        assert cb.normalizer_loss == 16 * 398 and cb.batch.nnz == cb.normalizer_metric
    assert seen_rows == 96
    first = [b.batch.po_rel.tolist() if b.batch.po_rel is not None else [] for b in batches]
    again = [b.batch.po_rel.tolist() if b.batch.po_rel is not None else [] for b in p]          # next epoch reshuffles
    assert first != again
    assert len(list(producer(z, True, False, 0, batch_size=16, drop_last=False))) == 7
    with pytest.raises(OkgeError):
        p.collate(np.asarray([1000]))                                             # row outside the table
    bad = dict(z)
    bad["seen"] = z["seen"].copy()
    bad["seen"][z["prefixes"][0, 2]] += 1                                         # corrupt a packed header
    with pytest.raises(OkgeError):
        producer(bad, True, False, 0).collate(np.asarray([0]))


# --------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("shared", [False, True])
def test_producer_feeds_train_step_and_ranks(okge_lib, shared):
    """prefix table -> producer (pinned arena, one H2D copy) -> fused step / score + filtered ranks, against the
    oracle fed with the dense tensors the reference's collate would have built."""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    rng = np.random.default_rng(21)
    n_ent, n_rel, d = 900, 50, 32
    z = random_tables(rng, n_ent, 200)
    E = (rng.standard_normal((n_ent, d)) * 0.3).astype(np.float32)
    R = (rng.standard_normal((n_rel, d)) * 0.3).astype(np.float32)
    # training batches
    p = producer(z, True, shared, 256 if shared else 0, batch_size=64)
    p.device = torch.device("cuda:0")
    st = FusedTrainStep(torch.from_numpy(E.copy()).cuda(), torch.from_numpy(R.copy()).cuda(), "complex", lr=0.3)
    Eo, Ro, sE, sR = E.copy(), R.copy(), np.zeros_like(E), np.zeros_like(R)
    for cb in p:
        b = cb.batch
        loss = float(st.step(b, normalizer=cb.normalizer_loss)[0])
        cand = b.cand_ids.cpu().numpy() if shared else np.arange(2, n_ent)
        out = ko.step_forward_backward(ko.COMPLEX, Eo, Ro, (b.po_rel.cpu().numpy(), b.po_obj.cpu().numpy()),
                                       (b.sp_subj.cpu().numpy(), b.sp_rel.cpu().numpy()), cand,
                                       cb.dense_labels().cpu().numpy(), normalizer=cb.normalizer_loss)
        ko.adagrad_step(Eo, out["dE"], sE, 0.3)
        ko.adagrad_step(Ro, out["dR"], sR, 0.3)
        assert abs(loss - out["loss"]) <= 3e-5 * abs(out["loss"])
    close = np.isclose(st.E.cpu().numpy(), Eo, rtol=1e-3, atol=1e-4)
    assert close.mean() > 0.999
    # evaluation batch
    pe = producer(z, False, shared, 0, batch_size=48)
    pe.device = torch.device("cuda:0")
    cb = next(iter(pe))
    b = cb.batch
    hp = H.HotPath("cuda:0")
    Et, Rt = torch.from_numpy(E).cuda(), torch.from_numpy(R).cuda()
    x = hp.score(Et, Rt, "complex", b)
    ranks = hp.filtered_ranks(x, cb.filt_ptr, cb.filt_col, cb.row_ptr, cb.grp_ptr, cb.ids).cpu().numpy()
    filt = np.zeros((b.B, cb.n_cand), bool)
    fp, fc = cb.filt_ptr.cpu().numpy(), cb.filt_col.cpu().numpy()
    filt[np.repeat(np.arange(b.B), np.diff(fp)), fc] = True
    ref = ko.filtered_ranks(x.cpu().numpy(), filt, cb.row_ptr.cpu().numpy(), cb.grp_ptr.cpu().numpy(), cb.ids.cpu().numpy())
    np.testing.assert_array_equal(ranks, ref)


# --------------------------------------------------------------------------------- on-disk format -> tensors (f3)
@pytest.mark.both
def test_dataset_loader_matches_reference_tensors(okge_lib):
    """csrc/okge_dataset.cpp on tests/golden/toy_kg against the tensors the reference's dataset class built from the
    same files (tests/golden/g6_dataset_toy.npz), and against the oracle restatement."""
    import os
    from conftest import GOLDEN
    from open_knowledge_graph_embeddings_amd.dataset import dataset_meta, load_dataset_tensors
    from test_oracle_golden import assert_all_splits_equal, toy_records
    z = golden("g6_dataset_toy")
    toy = os.path.join(GOLDEN, "toy_kg")
    out, all_splits, max_ids = load_dataset_tensors(toy)
    for split in ("train", "valid", "test"):
        np.testing.assert_array_equal(out[split][0], z[split + "_prefixes"])
        np.testing.assert_array_equal(out[split][1], z[split + "_seen"])
    assert_all_splits_equal(z["valid_prefixes"], all_splits, z["valid_all"])
    meta = dataset_meta(toy)
    assert meta.entities_size == int(z["entity_vocab_size"]) == 45 and meta.relations_size == 7
    assert max_ids[0] < meta.entities_size and max_ids[1] < meta.relations_size
    # long answer lists cut into several training rows
    out3, _, _ = load_dataset_tensors(toy, max_size_prefix_label=3)
    rec, merged = toy_records()
    pref, seen, allsp = ko.dataset_tensors(rec["train"], merged, True, max_size_prefix_label=3)
    np.testing.assert_array_equal(out3["train"][0], pref)
    np.testing.assert_array_equal(out3["train"][1], seen)
    np.testing.assert_array_equal(all_splits, allsp)                         # same (ascending) order as the oracle
    np.testing.assert_array_equal(pref, z["train3_prefixes"][:len(pref)])    # reference: plus an uninitialised tail
    np.testing.assert_array_equal(out3["valid"][0], out["valid"][0])


@pytest.mark.both
def test_dataset_loader_fb15k237_hashes(okge_lib, tmp_path):
    """The reference's own FB15k-237 id files (data fixtures under tests/golden/fb15k237): the loader's tensors are
    sha256-identical to what the reference's dataset class built from them (G6)."""
    import hashlib
    from conftest import fb15k237_dir
    from open_knowledge_graph_embeddings_amd.dataset import load_dataset_tensors
    fb = fb15k237_dir(tmp_path)
    z = golden("g6_dataset_fb15k237_hashes")
    out, all_splits, _ = load_dataset_tensors(fb, train_input_file="test.txt")       # train split absent upstream
    pref, seen = out["valid"]
    assert [pref.shape[0], seen.shape[0], all_splits.shape[0]] == z["valid_shapes"].tolist() == [20110, 110356, 75998]
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()    # noqa: E731
    assert sha(pref) == str(z["valid_sha_prefixes"]) and sha(seen) == str(z["valid_sha_seen"])
    slices = np.concatenate([np.sort(all_splits[a:b]) for a, b in sorted({(int(r[4]), int(r[5])) for r in pref})])
    assert sha(slices) == str(z["valid_sha_all_sorted"])


def test_dataset_loader_errors(okge_lib, tmp_path):
    from open_knowledge_graph_embeddings_amd import OkgeError
    from open_knowledge_graph_embeddings_amd.dataset import load_dataset_tensors
    with pytest.raises(OkgeError):
        load_dataset_tensors(str(tmp_path))                                            # files missing
    for name in ("train.txt", "valid.txt", "test.txt"):
        (tmp_path / name).write_text("2\t3\t4\t2\t4\n5\t3\n")
    with pytest.raises(OkgeError):
        load_dataset_tensors(str(tmp_path))                                            # a short line


@pytest.mark.both
def test_files_to_batches_end_to_end(okge_lib):
    """toy_kg text files -> loader -> producer -> batches, against the oracle fed with the reference-built tensors"""
    import os
    from conftest import GOLDEN
    from open_knowledge_graph_embeddings_amd.dataset import OneToNBatchProducer, dataset_meta, load_dataset_tensors
    toy = os.path.join(GOLDEN, "toy_kg")
    z = golden("g6_dataset_toy")
    out, all_splits, _ = load_dataset_tensors(toy)
    meta = dataset_meta(toy)
    p = OneToNBatchProducer(out["valid"][0], out["valid"][1], all_splits, meta.entities_size, batch_size=16,
                            is_training_data=False, drop_last=False)
    n = 0
    for cb, rows in zip(p, p.batch_rows()):
        ref = ko.collate_batch(z["valid_prefixes"][rows], z["valid_seen"], z["valid_all"], 45, 2, False, False)
        np.testing.assert_array_equal(coords(cb.batch), np.asarray(ref["labels"], np.int32).reshape(-1, 2))
        np.testing.assert_array_equal(cb.filt_col.numpy(), np.asarray([c for f in ref["filters"] for c in f], np.int32))
        n += cb.batch.B
    assert n == 91


@pytest.mark.gpu
def test_toy_kg_end_to_end_training_improves_mrr(okge_lib):
    """files -> loader -> producer -> fused train steps -> score + filtered ranks: the model must learn the toy KG
    (training split as evaluation data: filtered MRR goes from chance to near-perfect memorisation)."""
    from conftest import GOLDEN
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.dataset import OneToNBatchProducer, dataset_meta, load_dataset_tensors
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    toy = os.path.join(GOLDEN, "toy_kg")
    # evaluate on the training triples: use train.txt as the "valid" split too, so its rows carry filter slices
    out, all_splits, _ = load_dataset_tensors(toy, "train.txt", "train.txt", "test.txt")
    meta = dataset_meta(toy)
    d = 32
    g = torch.Generator().manual_seed(0)
    E = (torch.randn((meta.entities_size, d), generator=g) * 0.1).cuda()
    R = (torch.randn((meta.relations_size, d), generator=g) * 0.1).cuda()
    train = OneToNBatchProducer(*out["train"], all_splits, meta.entities_size, batch_size=32, is_training_data=True,
                                shuffle=True, seed=1, device="cuda:0")
    valid = OneToNBatchProducer(*out["valid"], all_splits, meta.entities_size, batch_size=64, is_training_data=False,
                                drop_last=False, device="cuda:0")
    step = FusedTrainStep(E, R, "complex", lr=0.3, input_dropout=0.1, seed=3)
    hp = step.engine

    def mrr():
        rr, n = 0.0, 0
        for cb in valid:
            x = hp.score(E, R, "complex", cb.batch)
            ranks = hp.filtered_ranks(x, cb.filt_ptr, cb.filt_col, cb.row_ptr, cb.grp_ptr, cb.ids)
            rr += float((1.0 / (ranks.double() + 1.0)).sum())
            n += ranks.numel()
        return rr / n

    before = mrr()
    losses = []
    for _ in range(60):
        for cb in train:
            losses.append(float(step.step(cb.batch, normalizer=cb.normalizer_loss)[0]))      # loss_out is reused: read now
    after = mrr()
    first, last = losses[0], losses[-1]
    assert before < 0.25 and after > 0.8 and last < 0.5 * first, (before, after, first, last)
    # the two-stream evaluator (score on one stream, ranks + device-side meters on another) agrees
    from open_knowledge_graph_embeddings_amd.evaluate import PipelinedEvaluator
    res, n_groups = PipelinedEvaluator(E, R, "complex", engine=hp).run(valid)
    assert abs(res["mrr"].avg - after) < 1e-6 and n_groups == res["mrr"].count > 0
    assert 0.0 <= res["h1"].avg <= res["h3"].avg <= res["h10"].avg <= res["h50"].avg <= 1.0 and res["mr"].avg >= 0


@pytest.mark.parametrize("shared,training", [(False, True), (False, False), (True, True), (True, False)])
def test_grouped_collate_equals_single_batches(okge_lib, shared, training):
    """okge_collate_batches (K batches, one arena) == K calls of okge_collate_batch, array by array"""
    rng = np.random.default_rng(31 + 2 * shared + training)
    z = random_tables(rng, 3000, 400)
    p = producer(z, training, shared, 0, batch_size=32)
    rows_list = [rng.choice(400, size=32, replace=False) for _ in range(5)]
    group = p.group_to_device(*p.collate_group_host(rows_list, seed=9))
    assert len(group) == 5
    for rows, g in zip(rows_list, group):
        one = p.collate(rows)
        for name in ("po_rel", "po_obj", "sp_subj", "sp_rel", "pos_row", "pos_col", "cand_ids"):
            a, b2 = getattr(g.batch, name), getattr(one.batch, name)
            assert (a is None) == (b2 is None)
            if a is not None:
                np.testing.assert_array_equal(a.numpy(), b2.numpy(), err_msg=name)
        assert (g.normalizer_loss, g.normalizer_metric, g.n_cand) == (one.normalizer_loss, one.normalizer_metric, one.n_cand)
        if not training:
            for name in ("row_ptr", "grp_ptr", "ids", "filt_ptr", "filt_col"):
                np.testing.assert_array_equal(getattr(g, name).numpy(), getattr(one, name).numpy(), err_msg=name)
