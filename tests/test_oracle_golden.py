"""Pin the CPU oracle (oracle/) against golden vectors generated from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden, golden_names
import oracle
from oracle import kge_oracle as ko
from oracle import torch_twin


def _kind(z):
    return ko.KIND_NAMES[str(z["model"])] if "model" in z.files else None


# ---------------------------------------------------------------------------------------------- G1
@pytest.mark.parametrize("name", golden_names("g1_scores_"))
def test_g1_scores(name):
    z = golden(name)
    kind = ko.COMPLEX if "complex" in name else ko.DISTMULT
    E, R = z["E"], z["R"]
    for cand_ids, sp_key, po_key in ((np.arange(2, E.shape[0]), "sp_all", "po_all"), (z["cand"], "sp_cand", "po_cand")):
        C = ko.encode(E, cand_ids)
        sp = ko.score_prefix(kind, ko.DIR_SP, ko.encode(E, z["subj"]), ko.encode(R, z["rel_sp"]), C)
        po = ko.score_prefix(kind, ko.DIR_PO, ko.encode(E, z["obj"]), ko.encode(R, z["rel_po"]), C)
        np.testing.assert_allclose(sp, z[sp_key], rtol=0, atol=2e-6)
        np.testing.assert_allclose(po, z[po_key], rtol=0, atol=2e-6)
        # the literal 4-product form agrees with the folded single-product form
        sp4 = ko.score_prefix_4mm(kind, ko.DIR_SP, ko.encode(E, z["subj"]), ko.encode(R, z["rel_sp"]), C)
        po4 = ko.score_prefix_4mm(kind, ko.DIR_PO, ko.encode(E, z["obj"]), ko.encode(R, z["rel_po"]), C)
        np.testing.assert_allclose(sp4, z[sp_key], rtol=0, atol=2e-6)
        np.testing.assert_allclose(po4, z[po_key], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", golden_names("g1_triples_"))
def test_g1_triples(name):
    z = golden(name)
    kind = ko.COMPLEX if "complex" in name else ko.DISTMULT
    out = ko.score_triples(kind, ko.encode(z["E"], z["subj"]), ko.encode(z["R"], z["rel"]), ko.encode(z["E"], z["obj"]))
    np.testing.assert_allclose(out[:, None], z["scores"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["g1_scores_complex_tiny", "g1_scores_distmult_tiny", "g1_scores_complex_odd"])
def test_g1_scores_c_oracle(name):
    z = golden(name)
    lib = oracle.load_c()
    kind = ko.COMPLEX if "complex" in name else ko.DISTMULT
    E, R = z["E"], z["R"]
    C = np.ascontiguousarray(ko.encode(E, z["cand"]))
    for direction, ent_ids, rel_ids, key in ((ko.DIR_SP, z["subj"], z["rel_sp"], "sp_cand"),
                                             (ko.DIR_PO, z["obj"], z["rel_po"], "po_cand")):
        ent = np.ascontiguousarray(ko.encode(E, ent_ids))
        rel = np.ascontiguousarray(ko.encode(R, rel_ids))
        out = np.zeros((ent.shape[0], C.shape[0]), np.float32)
        lib.okge_oracle_score_prefix(kind, direction, ent.ctypes.data, rel.ctypes.data, C.ctypes.data,
                                     ent.shape[0], C.shape[0], C.shape[1], out.ctypes.data)
        np.testing.assert_allclose(out, z[key], rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------------- G2
def _step_from_golden(z, dtype=np.float32):
    kind = ko.KIND_NAMES[str(z["model"])]
    po = (z["po_rel"], z["po_obj"]) if "po_rel" in z.files else None
    sp = (z["sp_subj"], z["sp_rel"]) if "sp_subj" in z.files else None
    p = float(z["input_dropout"])
    kw = {}
    if p > 0:
        kw = dict(p_ent=p, keep_cand=z["mask_cand"], keep_po_ent=z["mask_po_ent"] if po else None,
                  keep_sp_ent=z["mask_sp_ent"] if sp else None)
    return ko.step_forward_backward(kind, z["E"].astype(dtype), z["R"].astype(dtype), po, sp, z["cand"],
                                    z["labels"], ko.LOSS_NAMES[str(z["loss_kind"])], float(z["smoothing"]),
                                    float(z["normalizer"]), **kw)


@pytest.mark.parametrize("name", golden_names("g2_loss_"))
def test_g2_loss_and_grads(name):
    z = golden(name)
    out = _step_from_golden(z)
    np.testing.assert_allclose(out["outputs"], z["outputs"], rtol=0, atol=5e-6)
    assert abs(out["loss"] - float(z["loss"])) <= 2e-5 * max(1.0, abs(float(z["loss"])))
    scale = max(np.abs(z["dE"]).max(), 1e-12)
    np.testing.assert_allclose(out["dE"], z["dE"], rtol=0, atol=2e-5 * scale + 1e-9)
    scale = max(np.abs(z["dR"]).max(), 1e-12)
    np.testing.assert_allclose(out["dR"], z["dR"], rtol=0, atol=2e-5 * scale + 1e-9)
    # rows 0/1 (PAD/UNK) never receive gradient
    assert not out["dE"][:2].any()


# ---------------------------------------------------------------------------------------------- G3
def adagrad_tol(sum_ref, sum_prev, lr, eps, rel_dg=1e-6):
    """|dp| that a gradient perturbation of rel_dg * max|g| can cause in ONE step: Adagrad's
    p -= lr * g / (sqrt(sum) + eps) amplifies absolute gradient noise by lr / (sqrt(sum) + eps)
    (the first step is p -= lr * g / (|g| + 1e-8): elements with |g| ~ 1e-9 move by O(lr))."""
    dg = rel_dg * np.sqrt(np.maximum(sum_ref - sum_prev, 0).max())
    return 2e-6 + lr * dg / (np.sqrt(sum_ref) + eps)


@pytest.mark.parametrize("name", golden_names("g3_adagrad_"))
def test_g3_adagrad(name):
    """Each optimisation step is checked on its own, restarted from the reference's state before it
    (a trajectory amplifies 1e-12 gradient noise chaotically through p -= lr*g/(|g|+1e-8); the
    trajectory itself is G7)."""
    z = golden(name)
    kind = ko.COMPLEX if "complex" in name else ko.DISTMULT
    assert float(z["opt_eps"]) == 1e-8 and float(z["opt_lr_decay"]) == 0 and float(z["opt_initial_accumulator_value"]) == 0
    lr, wd, eps = float(z["opt_lr"]), float(z["opt_weight_decay"]), float(z["opt_eps"])
    for i in range(int(z["nsteps"])):
        if i == 0:
            E, R = z["E0"].copy(), z["R0"].copy()
            sE, sR = np.zeros_like(E), np.zeros_like(R)
        else:
            E, R = z[f"s{i-1}_E"].copy(), z[f"s{i-1}_R"].copy()
            sE, sR = z[f"s{i-1}_sumE"].copy(), z[f"s{i-1}_sumR"].copy()
        pE, pR = sE.copy(), sR.copy()
        out = ko.step_forward_backward(kind, E, R, (z[f"s{i}_po_rel"], z[f"s{i}_po_obj"]),
                                       (z[f"s{i}_sp_subj"], z[f"s{i}_sp_rel"]), z["cand"], z[f"s{i}_labels"])
        assert abs(out["loss"] - float(z[f"s{i}_loss"])) <= 2e-5 * abs(float(z[f"s{i}_loss"]))
        ko.adagrad_step(E, out["dE"], sE, lr, wd, eps)
        ko.adagrad_step(R, out["dR"], sR, lr, wd, eps)
        for mine, ref, s_mine, s_ref, s_prev in ((E, z[f"s{i}_E"], sE, z[f"s{i}_sumE"], pE),
                                                 (R, z[f"s{i}_R"], sR, z[f"s{i}_sumR"], pR)):
            tol = adagrad_tol(s_ref, s_prev, lr, eps)
            assert np.all(np.abs(mine - ref) <= tol), float((np.abs(mine - ref) / tol).max())
            assert np.mean(np.abs(mine - ref) <= 2e-6) > 0.97          # and almost everything is tight
            np.testing.assert_allclose(np.sqrt(s_mine), np.sqrt(s_ref), rtol=1e-4, atol=1e-6 * np.sqrt(s_ref.max()))
    # untouched rows 0/1 still move (weight decay + leaked eps): reproduced, not skipped
    assert np.abs(z["s2_E"][:2] - z["E0"][:2]).max() > 0
    np.testing.assert_allclose(E[:2], z["s2_E"][:2], rtol=2e-5, atol=2e-6)


def test_adagrad_arithmetic_isolated():
    """Same gradient in -> same update out (the ill-conditioned part of G3 removed): oracle vs the
    optimizer class the reference instantiates (torch.optim.Adagrad with the leaked eps)."""
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal((50, 24)).astype(np.float32) * 0.1
    p_t = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adagrad([p_t], lr=0.3, weight_decay=1e-10, eps=1e-8)
    p, s = p0.copy(), np.zeros_like(p0)
    for step in range(4):
        g = (rng.standard_normal(p0.shape) * 10.0 ** rng.integers(-9, -2, size=p0.shape)).astype(np.float32)
        g[:3] = 0.0                                              # rows that only see weight decay
        p_t.grad = torch.from_numpy(g.copy())
        opt.step()
        ko.adagrad_step(p, g, s, 0.3, 1e-10, 1e-8)
        np.testing.assert_allclose(p, p_t.detach().numpy(), rtol=3e-7, atol=2e-7)   # a few ulp at the magnitude of the lr-sized update
        np.testing.assert_allclose(s, opt.state[p_t]["sum"].numpy(), rtol=3e-7, atol=0)


# ---------------------------------------------------------------------------------------------- G5
@pytest.mark.parametrize("name", golden_names("g5_ranks_"))
def test_g5_ranks(name):
    z = golden(name)
    ranks = ko.filtered_ranks(z["pred"], z["filt"], z["row_ptr"], z["grp_ptr"], z["ids"])
    assert ranks.dtype == np.int64
    np.testing.assert_array_equal(ranks, z["ranks"])          # bit-exact
    m, n = ko.metrics_from_ranks(ranks, z["row_ptr"])
    assert n == int(z["c_mrr"])
    for k in ("mrr", "mr", "h1", "h3", "h10", "h50"):
        assert abs(m[k] - float(z["m_" + k])) <= 1e-6 * max(1.0, abs(float(z["m_" + k]))), k
    # plain-C oracle agrees bit for bit
    lib = oracle.load_c()
    pred = np.ascontiguousarray(z["pred"], np.float32)
    filt = np.ascontiguousarray(z["filt"], np.uint8)
    out = np.zeros(len(z["grp_ptr"]) - 1, np.int64)
    lib.okge_oracle_filtered_ranks(pred.ctypes.data, filt.ctypes.data, pred.shape[0], pred.shape[1],
                                   np.ascontiguousarray(z["row_ptr"]).ctypes.data,
                                   np.ascontiguousarray(z["grp_ptr"]).ctypes.data,
                                   np.ascontiguousarray(z["ids"]).ctypes.data, out.ctypes.data)
    np.testing.assert_array_equal(out, z["ranks"])


def test_g5_known_answer_values():
    z = golden("g5_ranks_known")
    np.testing.assert_array_equal(z["ranks"], [1, 2])
    assert abs(float(z["m_mrr"]) - 0.4166667) < 1e-6 and float(z["m_mr"]) == 1.5
    assert float(z["m_h1"]) == 0.0 and float(z["m_h3"]) == 1.0


# ---------------------------------------------------------------------------------------------- G7
def test_g7_trajectory():
    z = golden("g7_traj_complex")
    E, R = z["E0"].copy(), z["R0"].copy()
    sE, sR = np.zeros_like(E), np.zeros_like(R)
    for step in range(int(z["nsteps"])):
        i = step % 4
        out = ko.step_forward_backward(ko.COMPLEX, E, R, (z[f"b{i}_po_rel"], z[f"b{i}_po_obj"]),
                                       (z[f"b{i}_sp_subj"], z[f"b{i}_sp_rel"]), z["cand"], z[f"b{i}_labels"])
        n = z[f"b{i}_labels"].size
        assert abs(out["loss"] / n - z["losses"][step]) <= 5e-5 * abs(z["losses"][step])
        ko.adagrad_step(E, out["dE"], sE, 0.3)
        ko.adagrad_step(R, out["dR"], sR, 0.3)
    np.testing.assert_allclose(E, z["E"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(R, z["R"], rtol=1e-3, atol=2e-4)


# ------------------------------------------------------------------------------- torch twin (cpu_baseline)
@pytest.mark.parametrize("name", ["g2_loss_complex_bce_all", "g2_loss_distmult_kl_all", "g2_loss_complex_bce_smooth_all"])
def test_torch_twin_matches_golden(name):
    z = golden(name)
    m = torch_twin.TwinModel(str(z["model"]), z["E"].shape[0], z["R"].shape[0], z["E"].shape[1])
    with torch.no_grad():
        m.entity_embedding.weight.copy_(torch.from_numpy(z["E"]))
        m.relation_embedding.weight.copy_(torch.from_numpy(z["R"]))
    m.train()
    po = (torch.from_numpy(z["po_rel"]), torch.from_numpy(z["po_obj"]))
    sp = (torch.from_numpy(z["sp_subj"]), torch.from_numpy(z["sp_rel"]))
    loss, x = m.forward_loss(po, sp, torch.from_numpy(z["cand"]), torch.from_numpy(z["labels"].copy()),
                             str(z["loss_kind"]), float(z["smoothing"]))
    (loss / float(z["normalizer"])).backward()
    np.testing.assert_allclose(x.detach().numpy(), z["outputs"], atol=1e-6)
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    np.testing.assert_allclose(m.entity_embedding.weight.grad.numpy(), z["dE"], atol=1e-8)


def test_torch_twin_adagrad_matches_golden():
    z = golden("g3_adagrad_complex")
    m = torch_twin.TwinModel("complex", z["E0"].shape[0], z["R0"].shape[0], z["E0"].shape[1])
    with torch.no_grad():
        m.entity_embedding.weight.copy_(torch.from_numpy(z["E0"]))
        m.relation_embedding.weight.copy_(torch.from_numpy(z["R0"]))
    m.train()
    opt = torch_twin.make_adagrad(m, lr=0.3)
    for i in range(3):
        torch_twin.train_step(m, opt, (torch.from_numpy(z[f"s{i}_po_rel"]), torch.from_numpy(z[f"s{i}_po_obj"])),
                              (torch.from_numpy(z[f"s{i}_sp_subj"]), torch.from_numpy(z[f"s{i}_sp_rel"])),
                              torch.from_numpy(z["cand"]), torch.from_numpy(z[f"s{i}_labels"].copy()))
        np.testing.assert_allclose(m.entity_embedding.weight.detach().numpy(), z[f"s{i}_E"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------- Philox mask: numpy == C
@pytest.mark.parametrize("nrows,d,p", [(5, 16, 0.4), (33, 200, 0.4), (7, 37, 0.25), (3, 8, 0.0), (4, 12, 0.999)])
def test_philox_mask_numpy_equals_c(nrows, d, p):
    lib = oracle.load_c()
    a = ko.dropout_keep_mask(0x1234_5678_9ABC_DEF0, 3, 17, nrows, d, p)
    out = np.zeros((nrows, d), np.uint8)
    lib.okge_oracle_philox_keep(ctypes.c_uint64(0x1234_5678_9ABC_DEF0), 3, 17, nrows, d, p, None, out.ctypes.data)
    np.testing.assert_array_equal(a.astype(np.uint8), out)
    keys = (np.arange(nrows, dtype=np.uint32) * 7919 + 11).astype(np.uint32)
    a = ko.dropout_keep_mask(99, 1, 2, nrows, d, p, row_keys=keys)
    lib.okge_oracle_philox_keep(ctypes.c_uint64(99), 1, 2, nrows, d, p, keys.ctypes.data, out.ctypes.data)
    np.testing.assert_array_equal(a.astype(np.uint8), out)
    if p == 0.0:
        assert a.all()


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    c = ko.philox4x32_10(np.uint32(0), np.uint32(0), np.uint32(0), np.uint32(0), 0, 0)
    assert [int(x) for x in c] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = np.uint32(0xFFFFFFFF)
    c = ko.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x) for x in c] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    c = ko.philox4x32_10(np.uint32(0x243F6A88), np.uint32(0x85A308D3), np.uint32(0x13198A2E), np.uint32(0x03707344),
                         0xA4093822, 0x299F31D0)
    assert [int(x) for x in c] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_keep_rate():
    m = ko.dropout_keep_mask(7, 0, 0, 2000, 200, 0.4)
    assert abs(m.mean() - 0.6) < 0.005


# ---------------------------------------------------------------------------------------------- G4
def collate_cases():
    z = golden("g4_collate_toy")
    for c in range(int(z["n_cases"])):
        t = f"c{c}_"
        shared, min_size, training = (int(x) for x in z[t + "cfg"])
        yield z, t, bool(shared), min_size, bool(training)


def test_g4_pack_roundtrip():
    groups = [[7], [3, 4, 3], [9, 9]]
    packed = ko.pack_groups(groups)
    assert packed == [5, 6, 9, 11, 0, 7, 3, 4, 3, 9, 9]           # L = k+2 = 5 header slots
    assert ko.pack_groups([[1], [2], [3]]) == [5, 6, 7, 8, 0, 1, 2, 3]
    assert ko.unpack_groups(packed) == (groups, [7, 3, 4, 3, 9, 9])


def test_g4_collate_oracle_matches_reference():
    n = 0
    for z, t, shared, min_size, training in collate_cases():
        rows = z["prefixes"][z[t + "rows"]]
        n_idx = None
        if shared:       # replay the reference's fill-up order (a Python set's iteration order, dataset.py:853-860)
            probe = ko.collate_batch(rows, z["seen"], z["all_splits"], int(z["n_ent"]), int(z["offset"]), training, True, 0)
            n_idx = len(probe["cand"])
        out = ko.collate_batch(rows, z["seen"], z["all_splits"], int(z["n_ent"]), int(z["offset"]), training, shared,
                               min_size, negatives=None if n_idx is None else z[t + "cand"][n_idx:])
        np.testing.assert_array_equal(out["cand"], z[t + "cand"])
        for name in ("po", "sp"):
            got = np.zeros((0, 2), np.int32) if out[name] is None else np.stack(out[name], axis=1)
            np.testing.assert_array_equal(got, z[t + name])
        np.testing.assert_array_equal(np.asarray(out["labels"], np.int32).reshape(-1, 2), z[t + "labels"])
        assert [out["normalizer_loss"], out["normalizer_metric"]] == z[t + "norm"].tolist()
        assert (len(rows), len(out["cand"])) == tuple(z[t + "shape"])
        if not training:
            filt = [(r, c) for r, cols in enumerate(out["filters"]) for c in cols]
            np.testing.assert_array_equal(np.asarray(filt, np.int32).reshape(-1, 2), z[t + "filter"])
            gp, ids, rp = [0], [], [0]
            for row_groups in out["groups"]:
                for g in row_groups:
                    ids += g
                    gp.append(len(ids))
                rp.append(len(gp) - 1)
            np.testing.assert_array_equal(rp, z[t + "row_ptr"])
            np.testing.assert_array_equal(gp, z[t + "grp_ptr"])
            np.testing.assert_array_equal(np.asarray(ids, np.int32), z[t + "ids"])
        n += 1
    assert n == 32


# ---------------------------------------------------------------------------------------------- G6
def toy_records():
    import os
    from conftest import GOLDEN
    lines = {sp: open(os.path.join(GOLDEN, "toy_kg", sp + ".txt")).readlines() for sp in ("train", "valid", "test")}
    rec = {sp: {d: ko.collect_prefix_groups(lines[sp], d) for d in ko.DIRECTIONS} for sp in lines}
    merged = {d: ko.merge_all_splits(rec["train"][d], rec["valid"][d], rec["test"][d]) for d in ko.DIRECTIONS}
    return rec, merged


def assert_all_splits_equal(pref, mine, ref):
    """same slices; inside a slice the reference keeps a Python set's iteration order"""
    assert mine.shape == ref.shape
    for a, b in {(int(r[4]), int(r[5])) for r in pref}:
        np.testing.assert_array_equal(np.sort(mine[a:b]), np.sort(ref[a:b]))


def test_g6_dataset_tensors_oracle_matches_reference():
    z = golden("g6_dataset_toy")
    rec, merged = toy_records()
    for split in ("train", "valid", "test"):
        pref, seen, allsp = ko.dataset_tensors(rec[split], merged, split == "train")
        np.testing.assert_array_equal(pref, z[split + "_prefixes"])
        np.testing.assert_array_equal(seen, z[split + "_seen"])
        assert_all_splits_equal(z["valid_prefixes"], allsp, z[split + "_all"])
    # max_size_prefix_label = 3: the reference's tensors carry an uninitialised tail (over-counted allocation)
    pref, seen, _ = ko.dataset_tensors(rec["train"], merged, True, max_size_prefix_label=3)
    assert 0 < len(pref) <= len(z["train3_prefixes"]) and len(seen) <= len(z["train3_seen"])
    np.testing.assert_array_equal(pref, z["train3_prefixes"][:len(pref)])
    np.testing.assert_array_equal(seen, z["train3_seen"][:len(seen)])


# ---------------------------------------------------------------------------------------------- G9
def unigram_bn(z, which):
    if str(z["normalize"]) != "batchnorm":
        return None
    d = z["We"].shape[1]
    return dict(weight=z[f"bn_{which}_w"].copy(), bias=z[f"bn_{which}_b"].copy(),
                running_mean=np.zeros(d, np.float32), running_var=np.ones(d, np.float32))


@pytest.mark.parametrize("name", golden_names("g9_unigram_"))
def test_g9_unigram_oracle_matches_reference(name):
    z = golden(name)
    po = (z["po_rel"], z["po_obj"]) if "po_rel" in z.files else None
    sp = (z["sp_subj"], z["sp_rel"]) if "sp_subj" in z.files else None
    bn_e, bn_r = unigram_bn(z, "e"), unigram_bn(z, "r")
    out = ko.unigram_step_forward_backward(ko.COMPLEX, z["We"], z["Wr"], z["ent_tokens"], z["rel_tokens"], po, sp,
                                           z["cand"], z["labels"], pool=str(z["pool"]), bn_ent=bn_e, bn_rel=bn_r)
    np.testing.assert_allclose(out["outputs"], z["outputs"], rtol=0, atol=2e-5)
    assert abs(out["loss"] - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    for mine, key in ((out["dWe"], "dWe"), (out["dWr"], "dWr")):
        np.testing.assert_allclose(mine, z[key], rtol=0, atol=2e-5 * np.abs(z[key]).max())
    if bn_e is not None:
        for (dw, db), w in ((out["d_bn_ent"], "e"), (out["d_bn_rel"], "r")):
            np.testing.assert_allclose(dw, z[f"d_bn_{w}_w"], rtol=0, atol=2e-5 * np.abs(z[f"d_bn_{w}_w"]).max())
            np.testing.assert_allclose(db, z[f"d_bn_{w}_b"], rtol=0, atol=2e-5 * np.abs(z[f"d_bn_{w}_b"]).max() + 1e-9)
        np.testing.assert_allclose(bn_e["running_mean"], z["run_e_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn_e["running_var"], z["run_e_var"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn_r["running_mean"], z["run_r_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn_r["running_var"], z["run_r_var"], rtol=1e-5, atol=1e-6)
    # evaluation tables: every id encoded from its tokens with the running statistics
    def table(W, tokens, bn):
        x, _ = ko.token_pool(W, tokens, np.arange(tokens.shape[0]), str(z["pool"]))
        return x if bn is None else ko.batchnorm_eval(x, bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"])
    np.testing.assert_allclose(table(z["We"], z["ent_tokens"], bn_e), z["E_eval"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(table(z["Wr"], z["rel_tokens"], bn_r), z["R_eval"], rtol=1e-5, atol=2e-6)


def test_g16_rows_no_batch_names_move_under_the_reference_optimizer():
    """G16 (token-pooled model, twelve steps of the reference's OptimRegime Adagrad): 107 of the 300 entity token rows are named by
    no entity.  The reference's dense optimizer moves them all the same -- by their weight-decay term, up to 0.013 over the run --
    and the oracle's Adagrad, fed zero gradients for those rows, lands on the reference's values and accumulators: the update
    okge_adagrad_lazy defers and replays is the reference's, not an artefact of this build."""
    z = golden("g16_unigram_adagrad")
    named = np.zeros(z["We"].shape[0], bool)
    named[np.unique(z["ent_tokens"])] = True
    assert (~named).sum() == 107
    moved = np.abs(z["We_end"][~named] - z["We"][~named])
    assert moved.max() > 0.01 and (moved > 0).mean() > 0.99 and float(z["opt_eps"]) == 1e-8 and float(z["opt_weight_decay"]) == 1e-10
    p, s = z["We"][~named].copy(), np.zeros_like(z["We"][~named])
    for _ in range(int(z["nsteps"])):
        ko.adagrad_step(p, np.zeros_like(p), s, float(z["opt_lr"]), float(z["opt_weight_decay"]), float(z["opt_eps"]))
    np.testing.assert_allclose(p, z["We_end"][~named], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(s, z["sumWe_end"][~named], rtol=2e-6, atol=1e-32)
