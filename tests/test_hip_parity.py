"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors generated from the
reference and against the CPU oracle on seeded inputs.  Run with `pytest -m gpu` on an MI355X.

Tolerances (fp32): scores |diff| <= 1e-4 (BASELINE.json north_star); we assert tighter where the
arithmetic allows.  Ranks: bit-exact (int64)."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names
from oracle import kge_oracle as ko

pytestmark = pytest.mark.gpu

SCORE_ATOL = 1e-4


@pytest.fixture(scope="module")
def hp(okge_lib):
    from open_knowledge_graph_embeddings_amd.hotpath import HotPath
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return HotPath("cuda:0")


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to("cuda:0")


def make_batch(z_or_dict, cand=None, n_ent=None, masks=None, p=0.0, labels=None):
    from open_knowledge_graph_embeddings_amd.hotpath import DropoutSpec, PrefixBatch, positives_from_dense
    z = z_or_dict
    has = (lambda k: k in z.files) if hasattr(z, "files") else (lambda k: k in z)
    b = PrefixBatch()
    if has("po_rel"):
        b.po_rel, b.po_obj = dev(z["po_rel"].reshape(-1)), dev(z["po_obj"].reshape(-1))
    if has("sp_subj"):
        b.sp_subj, b.sp_rel = dev(z["sp_subj"].reshape(-1)), dev(z["sp_rel"].reshape(-1))
    cand = np.asarray(cand).reshape(-1)
    if n_ent is not None and len(cand) == n_ent - 2 and np.array_equal(cand, np.arange(2, n_ent)):
        b.cand_first, b.n_cand = 2, len(cand)
    else:
        b.cand_ids, b.n_cand = dev(cand.astype(np.int32)), len(cand)
    if masks is not None and p > 0:
        b.drop_cand = DropoutSpec(p=p, keep=dev(masks["mask_cand"].astype(np.uint8)))
        if has("po_rel"):
            b.drop_po_ent = DropoutSpec(p=p, keep=dev(masks["mask_po_ent"].astype(np.uint8)))
        if has("sp_subj"):
            b.drop_sp_ent = DropoutSpec(p=p, keep=dev(masks["mask_sp_ent"].astype(np.uint8)))
    if labels is not None:
        b.pos_row, b.pos_col = positives_from_dense(dev(labels))
    return b


# ---------------------------------------------------------------------------------------------- G1
@pytest.mark.parametrize("name", golden_names("g1_scores_"))
def test_g1_scores(hp, name):
    z = golden(name)
    scorer = "complex" if "complex" in name else "distmult"
    E, R = dev(z["E"]), dev(z["R"])
    n_ent = z["E"].shape[0]
    for cand, sp_key, po_key in ((np.arange(2, n_ent), "sp_all", "po_all"), (z["cand"], "sp_cand", "po_cand")):
        both = {"po_rel": z["rel_po"], "po_obj": z["obj"], "sp_subj": z["subj"], "sp_rel": z["rel_sp"]}
        out = hp.score(E, R, scorer, make_batch(both, cand, n_ent)).cpu().numpy()
        n_po = len(z["obj"])
        np.testing.assert_allclose(out[:n_po], z[po_key], rtol=0, atol=2e-6)
        np.testing.assert_allclose(out[n_po:], z[sp_key], rtol=0, atol=2e-6)
        # one direction only (the other is None in the reference, trainer.py:73)
        sp_only = {"sp_subj": z["subj"], "sp_rel": z["rel_sp"]}
        out = hp.score(E, R, scorer, make_batch(sp_only, cand, n_ent)).cpu().numpy()
        np.testing.assert_allclose(out, z[sp_key], rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------------- G2
@pytest.mark.parametrize("name", golden_names("g2_loss_"))
def test_g2_loss_and_grads(hp, name):
    z = golden(name)
    scorer, loss_kind = str(z["model"]), str(z["loss_kind"])
    p = float(z["input_dropout"])
    E, R = dev(z["E"]), dev(z["R"])
    batch = make_batch(z, z["cand"], z["E"].shape[0], masks=z if p > 0 else None, p=p, labels=z["labels"])
    dE, dR = torch.zeros_like(E), torch.zeros_like(R)
    B, Nc = z["labels"].shape
    scores = torch.empty((B, (Nc + 3) // 4 * 4), device="cuda:0")[:, :Nc]
    loss = hp.forward_backward(E, R, scorer, batch, dE, dR, loss=loss_kind, label_smoothing=float(z["smoothing"]),
                               normalizer=float(z["normalizer"]), scores=scores)
    torch.cuda.synchronize()
    np.testing.assert_allclose(scores.cpu().numpy(), z["outputs"], rtol=0, atol=5e-6)
    assert abs(loss.item() - float(z["loss"])) <= 2e-5 * max(1.0, abs(float(z["loss"])))
    for mine, ref in ((dE, z["dE"]), (dR, z["dR"])):
        scale = max(np.abs(ref).max(), 1e-12)
        np.testing.assert_allclose(mine.cpu().numpy(), ref, rtol=0, atol=2e-5 * scale + 1e-9)
    assert not dE[:2].any().item()          # PAD/UNK rows never receive gradient


def test_gradients_accumulate(hp):
    """dE/dR are accumulated into (autograd .grad semantics): two calls == twice the gradient."""
    z = golden("g2_loss_complex_bce_all")
    E, R = dev(z["E"]), dev(z["R"])
    batch = make_batch(z, z["cand"], z["E"].shape[0], labels=z["labels"])
    dE, dR = torch.zeros_like(E), torch.zeros_like(R)
    for _ in range(2):
        hp.forward_backward(E, R, "complex", batch, dE, dR, normalizer=float(z["normalizer"]))
    np.testing.assert_allclose(dE.cpu().numpy(), 2 * z["dE"], rtol=0, atol=4e-5 * np.abs(z["dE"]).max())
    np.testing.assert_allclose(dR.cpu().numpy(), 2 * z["dR"], rtol=0, atol=4e-5 * np.abs(z["dR"]).max())


# ---------------------------------------------------------------------------------------------- G3
@pytest.mark.parametrize("name", golden_names("g3_adagrad_"))
def test_g3_adagrad_steps(hp, name):
    """Each step restarted from the reference's state (see tests/test_oracle_golden.py for why)."""
    from test_oracle_golden import adagrad_tol
    z = golden(name)
    scorer = "complex" if "complex" in name else "distmult"
    lr, wd, eps = float(z["opt_lr"]), float(z["opt_weight_decay"]), float(z["opt_eps"])
    for i in range(int(z["nsteps"])):
        if i == 0:
            E, R = dev(z["E0"]), dev(z["R0"])
            sE, sR = torch.zeros_like(E), torch.zeros_like(R)
        else:
            E, R = dev(z[f"s{i-1}_E"]), dev(z[f"s{i-1}_R"])
            sE, sR = dev(z[f"s{i-1}_sumE"]), dev(z[f"s{i-1}_sumR"])
        pE, pR = sE.cpu().numpy().copy(), sR.cpu().numpy().copy()
        step = {"po_rel": z[f"s{i}_po_rel"], "po_obj": z[f"s{i}_po_obj"], "sp_subj": z[f"s{i}_sp_subj"],
                "sp_rel": z[f"s{i}_sp_rel"]}
        batch = make_batch(step, z["cand"], z["E0"].shape[0], labels=z[f"s{i}_labels"])
        dE, dR = torch.zeros_like(E), torch.zeros_like(R)
        loss = hp.forward_backward(E, R, scorer, batch, dE, dR)
        hp.adagrad(E, dE, sE, lr, wd, eps, zero_grad=True)
        hp.adagrad(R, dR, sR, lr, wd, eps, zero_grad=True)
        torch.cuda.synchronize()
        assert abs(loss.item() - float(z[f"s{i}_loss"])) <= 2e-5 * abs(float(z[f"s{i}_loss"]))
        assert not dE.any().item() and not dR.any().item()      # zero_grad fused into the sweep
        for mine, ref, s_mine, s_ref, s_prev in ((E, z[f"s{i}_E"], sE, z[f"s{i}_sumE"], pE),
                                                 (R, z[f"s{i}_R"], sR, z[f"s{i}_sumR"], pR)):
            tol = adagrad_tol(s_ref, s_prev, lr, eps)
            diff = np.abs(mine.cpu().numpy() - ref)
            assert np.all(diff <= tol), float((diff / tol).max())
            assert np.mean(diff <= 2e-6) > 0.97
            np.testing.assert_allclose(np.sqrt(s_mine.cpu().numpy()), np.sqrt(s_ref), rtol=1e-4,
                                       atol=1e-6 * np.sqrt(s_ref.max()))


def test_adagrad_arithmetic_isolated(hp):
    rng = np.random.default_rng(3)
    for shape in ((50, 24), (7, 37), (1, 3)):                   # incl. sizes that are not multiples of 4
        p0 = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        p, s = p0.copy(), np.zeros_like(p0)
        pt, st = dev(p0), torch.zeros(shape, device="cuda:0")
        for _ in range(4):
            g = (rng.standard_normal(shape) * 10.0 ** rng.integers(-9, -2, size=shape)).astype(np.float32)
            g[:1] = 0.0
            gt = dev(g)
            hp.adagrad(pt, gt, st, 0.3, 1e-10, 1e-8, zero_grad=False)
            ko.adagrad_step(p, g, s, 0.3, 1e-10, 1e-8)
            np.testing.assert_allclose(pt.cpu().numpy(), p, rtol=3e-7, atol=2e-7)
            np.testing.assert_allclose(st.cpu().numpy(), s, rtol=3e-7, atol=0)
            np.testing.assert_array_equal(gt.cpu().numpy(), g)  # zero_grad=False leaves the gradient


# ---------------------------------------------------------------------------------------------- G5
def ranks_args(z):
    filt = z["filt"].astype(bool)
    fptr = np.concatenate([[0], np.cumsum(filt.sum(1))]).astype(np.int64)
    fcol = np.concatenate([np.nonzero(r)[0] for r in filt]).astype(np.int32)
    return (dev(z["pred"]), dev(fptr), dev(fcol) if len(fcol) else torch.empty(0, dtype=torch.int32, device="cuda:0"),
            dev(z["row_ptr"]), dev(z["grp_ptr"]), dev(z["ids"]))


@pytest.mark.parametrize("name", golden_names("g5_ranks_"))
def test_g5_ranks_bit_exact(hp, name):
    z = golden(name)
    ranks = hp.filtered_ranks(*ranks_args(z)).cpu().numpy()
    assert ranks.dtype == np.int64
    np.testing.assert_array_equal(ranks, z["ranks"])
    m, n = ko.metrics_from_ranks(ranks, z["row_ptr"])
    for k in ("mrr", "mr", "h1", "h3", "h10", "h50"):
        assert abs(m[k] - float(z["m_" + k])) <= 1e-6 * max(1.0, abs(float(z["m_" + k])))


def test_ranks_many_groups_and_ties(hp):
    """More groups per row than the kernel's chunk of 8, heavy ties, empty filter rows."""
    rng = np.random.default_rng(11)
    B, Nc = 5, 3000
    pred = (np.round(rng.standard_normal((B, Nc)) * 3) / 3).astype(np.float32)
    filt = np.zeros((B, Nc), bool)
    row_ptr, grp_ptr, ids = [0], [0], []
    for b in range(B):
        for _ in range(int(rng.integers(9, 30))):
            g = rng.choice(Nc, size=int(rng.integers(1, 4)), replace=False)
            ids.extend(g.tolist())
            grp_ptr.append(len(ids))
            if b != 2:
                filt[b, g] = True
        row_ptr.append(len(grp_ptr) - 1)
    z = {"pred": pred, "filt": filt.astype(np.uint8), "row_ptr": np.asarray(row_ptr, np.int64),
         "grp_ptr": np.asarray(grp_ptr, np.int64), "ids": np.asarray(ids, np.int32)}
    ref = ko.filtered_ranks(pred, filt, z["row_ptr"], z["grp_ptr"], z["ids"])
    got = hp.filtered_ranks(*ranks_args(z)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


# ------------------------------------------------------------------------------- seeded random cases vs oracle
def random_problem(seed, n_ent, n_rel, d, n_po, n_sp, n_cand=None, max_pos=5):
    rng = np.random.default_rng(seed)
    E = (rng.standard_normal((n_ent, d)) * 0.3).astype(np.float32)
    R = (rng.standard_normal((n_rel, d)) * 0.3).astype(np.float32)
    z = {}
    if n_po:
        z["po_rel"], z["po_obj"] = rng.integers(2, n_rel, n_po).astype(np.int32), rng.integers(2, n_ent, n_po).astype(np.int32)
    if n_sp:
        z["sp_subj"], z["sp_rel"] = rng.integers(2, n_ent, n_sp).astype(np.int32), rng.integers(2, n_rel, n_sp).astype(np.int32)
    cand = np.arange(2, n_ent) if n_cand is None else rng.permutation(np.arange(2, n_ent))[:n_cand]
    Nc = len(cand)
    y = np.zeros((n_po + n_sp, Nc), np.float32)
    for b in range(n_po + n_sp):
        y[b, rng.choice(Nc, size=int(rng.integers(1, min(max_pos, Nc) + 1)), replace=False)] = 1
    return E, R, z, cand.astype(np.int32), y


def oracle_step(scorer, E, R, z, cand, y, loss_kind="bce", smoothing=0.0, **kw):
    po = (z["po_rel"], z["po_obj"]) if "po_rel" in z else None
    sp = (z["sp_subj"], z["sp_rel"]) if "sp_subj" in z else None
    return ko.step_forward_backward(ko.KIND_NAMES[scorer], E, R, po, sp, cand, y, ko.LOSS_NAMES[loss_kind], smoothing, **kw)


CASES = [
    # scorer, n_ent, n_rel, d, n_po, n_sp, n_cand, loss, smoothing   (ragged sizes on purpose)
    ("complex", 1000, 30, 200, 70, 61, None, "bce", 0.0),
    ("complex", 777, 19, 64, 1, 130, None, "bce", 0.1),
    ("complex", 400, 9, 256, 33, 0, 257, "kl", 0.0),
    ("distmult", 900, 25, 100, 64, 64, 512, "bce", 0.0),
    ("distmult", 333, 7, 36, 5, 3, None, "kl", 0.0),
    ("complex", 5000, 50, 128, 100, 156, None, "bce", 0.0),
    ("distmult", 200, 5, 17, 3, 2, None, "bce", 0.0),          # odd slot size: scalar (non-float4) paths
    ("complex", 150, 5, 6, 2, 2, 1, "bce", 0.0),               # a single candidate
    ("distmult", 2000, 30, 512, 70, 61, 700, "bce", 0.0),      # slot sizes above 256: the KB=32 instantiations
    ("complex", 900, 20, 320, 33, 40, None, "kl", 0.0),
    ("complex", 700, 11, 512, 40, 0, None, "bce", 0.1),
    # few candidate tiles, many batch rows: the batch is split over blockIdx.y (partial dE slabs + dc_reduce) -- the shape of
    # a multi-GPU shard -- on the register-operand tile kernel and the double-buffered dQ kernel (slot sizes up to 208)
    ("complex", 1502, 20, 200, 256, 256, None, "bce", 0.0),
    ("complex", 3000, 20, 200, 250, 200, 700, "kl", 0.0),
    ("distmult", 1200, 15, 128, 300, 221, None, "bce", 0.1),
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-d{c[3]}-b{c[4]}+{c[5]}-{c[7]}" for c in CASES])
def test_random_vs_oracle(hp, case):
    scorer, n_ent, n_rel, d, n_po, n_sp, n_cand, loss_kind, smoothing = case
    E, R, z, cand, y = random_problem(100 + CASES.index(case), n_ent, n_rel, d, n_po, n_sp, n_cand)
    ref = oracle_step(scorer, E, R, z, cand, y, loss_kind, smoothing)
    Et, Rt = dev(E), dev(R)
    batch = make_batch(z, cand, n_ent, labels=y)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, scorer, batch, dE, dR, loss=loss_kind, label_smoothing=smoothing)
    out = hp.score(Et, Rt, scorer, batch).cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out, ref["outputs"], rtol=0, atol=SCORE_ATOL)
    assert np.abs(out - ref["outputs"]).max() <= 3e-6 * max(1.0, np.abs(ref["outputs"]).max())
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    for mine, r in ((dE, ref["dE"]), (dR, ref["dR"])):
        np.testing.assert_allclose(mine.cpu().numpy(), r, rtol=0, atol=3e-5 * np.abs(r).max() + 1e-12)


@pytest.mark.parametrize("grads_zero", [False, True])
def test_repeated_candidate_ids_accumulate(hp, grads_zero):
    """An explicit candidate list may name an entity several times (precompute_batch_shared_inputs takes any id
    list, model.py:76-77): autograd's embedding backward sums the rows' gradients, so must the tile write-back --
    across workgroups and inside one tile, with and without the zeroed-gradients fast path."""
    E, R, z, _, _ = random_problem(77, 300, 9, 200, 40, 33)
    rng = np.random.default_rng(78)
    cand = rng.integers(2, 60, 700).astype(np.int32)                 # ~12 copies of each of 58 entities
    y = np.zeros((73, 700), np.float32)
    for b in range(73):
        y[b, rng.choice(700, size=3, replace=False)] = 1
    ref = oracle_step("complex", E, R, z, cand, y)
    Et, Rt = dev(E), dev(R)
    batch = make_batch(z, cand, 300, labels=y)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, "complex", batch, dE, dR, grads_zero=grads_zero)
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    for mine, r in ((dE, ref["dE"]), (dR, ref["dR"])):
        np.testing.assert_allclose(mine.cpu().numpy(), r, rtol=0, atol=3e-5 * np.abs(r).max() + 1e-12)
    # a list without repeats may be declared so (the collator's lists): plain stores, same numbers
    E2, R2, z2, cand2, y2 = random_problem(79, 300, 9, 200, 40, 33, n_cand=250)
    ref2 = oracle_step("complex", E2, R2, z2, cand2, y2)
    batch2 = make_batch(z2, cand2, 300, labels=y2)
    batch2.cand_unique = True
    dE2, dR2 = torch.zeros_like(Et), torch.zeros_like(Rt)
    hp.forward_backward(dev(E2), dev(R2), "complex", batch2, dE2, dR2, grads_zero=grads_zero)
    np.testing.assert_allclose(dE2.cpu().numpy(), ref2["dE"], rtol=0, atol=3e-5 * np.abs(ref2["dE"]).max())


@pytest.mark.parametrize("scorer,d,lo,hi", [("complex", 200, 2, 300), ("distmult", 37, 17, 251), ("complex", 64, 0, 129)])
def test_clear_grads_flag(hp, scorer, d, lo, hi):
    """OKGE_TRAIN_CLEAR_GRADS: gradient buffers full of garbage, a contiguous candidate range anywhere in the table -- the
    call stores the candidate rows and clears everything else it accumulates into (all of dR, the rows of dE in front of and
    behind the candidates) inside its first launch: the same gradients as zeroed buffers + OKGE_TRAIN_GRADS_ZERO, exact
    zeros where nothing was added, and equal to the oracle.  An id list is refused."""
    from open_knowledge_graph_embeddings_amd._native import OkgeError
    n_ent, n_rel = 300, 9
    E, R, z, _, _ = random_problem(d + lo, n_ent, n_rel, d, 40, 33)
    rng = np.random.default_rng(hi)
    cand = np.arange(lo, hi).astype(np.int32)
    y = np.zeros((73, len(cand)), np.float32)
    for b in range(73):
        y[b, rng.choice(len(cand), size=3, replace=False)] = 1
    ref = oracle_step(scorer, E, R, z, cand, y)
    Et, Rt = dev(E), dev(R)
    batch = make_batch(z, cand, None, labels=y)
    batch.cand_ids, batch.cand_first, batch.n_cand = None, lo, hi - lo
    dE0, dR0 = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss0 = hp.forward_backward(Et, Rt, scorer, batch, dE0, dR0, grads_zero=True).clone()
    dE1, dR1 = torch.full_like(Et, float("nan")), torch.full_like(Rt, 123.0)
    loss1 = hp.forward_backward(Et, Rt, scorer, batch, dE1, dR1, clear_grads=True)
    assert float(loss0) == float(loss1)
    # (not bit for bit: the prefix rows and the relation rows are float atomics in either call; a NaN left behind would show)
    for a0, a1 in ((dE0, dE1), (dR0, dR1)):
        np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=0, atol=2e-6 * float(a0.abs().max()))
    untouched = np.setdiff1d(np.arange(n_ent), np.concatenate([cand, z["po_obj"], z["sp_subj"]]))
    assert float(dE1[torch.from_numpy(untouched).cuda()].abs().max()) == 0.0 if len(untouched) else True
    for mine, r in ((dE1, ref["dE"]), (dR1, ref["dR"])):
        np.testing.assert_allclose(mine.cpu().numpy(), r, rtol=0, atol=3e-5 * np.abs(r).max() + 1e-12)
    listed = make_batch(z, cand, None, labels=y)
    with pytest.raises(OkgeError):
        hp.forward_backward(Et, Rt, scorer, listed, dE1, dR1, clear_grads=True)


def test_b_split_path(hp, monkeypatch):
    """Few candidate tiles -> the batch is split across blockIdx.y; partial dC rows go to slabs summed by dc_reduce."""
    E, R, z, cand, y = random_problem(5, 300, 11, 64, 200, 184, 100)
    ref = oracle_step("complex", E, R, z, cand, y)
    Et, Rt = dev(E), dev(R)
    batch = make_batch(z, cand, 300, labels=y)
    for split in ("1", "3", "6"):
        monkeypatch.setenv("OKGE_B_SPLIT", split)
        dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
        loss = hp.forward_backward(Et, Rt, "complex", batch, dE, dR)
        torch.cuda.synchronize()
        assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
        np.testing.assert_allclose(dE.cpu().numpy(), ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
        np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())


# ------------------------------------------------------------------------------- slot sizes above 256
WIDE = [
    # scorer, d, n_ent, n_po, n_sp, n_cand, loss, p_ent, split    (fused_tile64k_kernel: candidate tile in registers)
    ("distmult", 512, 700, 70, 61, None, "bce", 0.3, "0"),
    ("complex", 512, 300, 200, 184, 130, "bce", 0.5, "3"),        # few tiles, batch split over blockIdx.y: slabs + dc_reduce
    ("complex", 320, 900, 33, 40, None, "kl", 0.4, "0"),          # 16-column blocks past the slot size: zero operands
    ("distmult", 257, 450, 17, 90, 200, "bce", 0.4, "2"),         # odd slot size: scalar gather, partial last octet
    ("complex", 258, 333, 64, 0, None, "bce", 0.0, "0"),          # one direction, no dropout, d % 4 != 0
    ("distmult", 400, 2100, 100, 156, None, "kl", 0.2, "0"),      # 33 tiles, 8 chunks of 32 rows
]


@pytest.mark.parametrize("tile_w", ["64", "32"])
@pytest.mark.parametrize("case", WIDE, ids=[f"{c[0]}-d{c[1]}-b{c[3]}+{c[4]}-{c[6]}-p{c[7]}-s{c[8]}" for c in WIDE])
def test_wide_slots(hp, monkeypatch, case, tile_w):
    """slot sizes 257..512 on the register-tile kernel (tile_w 64) and on the round-1/2 32 x 32 cut it replaced (OKGE_TILE_W=32):
    scores, loss, gradients against the oracle with Philox dropout on every stream"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    scorer, d, n_ent, n_po, n_sp, n_cand, loss_kind, p, split = case
    E, R, z, cand, y = random_problem(900 + WIDE.index(case), n_ent, 13, d, n_po, n_sp, n_cand)
    kw, batch = {}, make_batch(z, cand, n_ent, labels=y)
    if p > 0:
        sd, step = 0xFACE0FF + d, 11
        km = lambda stream, n, pp: ko.dropout_keep_mask(sd, stream, step, n, d, pp) if n else None     # noqa: E731
        kw = dict(p_ent=p, p_rel=0.25, keep_cand=km(H.STREAM_CAND, len(cand), p), keep_po_ent=km(H.STREAM_PO_ENT, n_po, p),
                  keep_sp_ent=km(H.STREAM_SP_ENT, n_sp, p), keep_po_rel=km(H.STREAM_PO_REL, n_po, 0.25),
                  keep_sp_rel=km(H.STREAM_SP_REL, n_sp, 0.25))
        batch.drop_cand = H.DropoutSpec(p, sd, H.STREAM_CAND, step)
        batch.drop_po_ent, batch.drop_sp_ent = H.DropoutSpec(p, sd, H.STREAM_PO_ENT, step), H.DropoutSpec(p, sd, H.STREAM_SP_ENT, step)
        batch.drop_po_rel, batch.drop_sp_rel = H.DropoutSpec(0.25, sd, H.STREAM_PO_REL, step), H.DropoutSpec(0.25, sd, H.STREAM_SP_REL, step)
    ref = oracle_step(scorer, E, R, z, cand, y, loss_kind, 0.0, **kw)
    monkeypatch.setenv("OKGE_TILE_W", tile_w)
    if split != "0":
        monkeypatch.setenv("OKGE_B_SPLIT", split)
    Et, Rt = dev(E), dev(R)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, scorer, batch, dE, dR, loss=loss_kind)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    for mine, r in ((dE, ref["dE"]), (dR, ref["dR"])):
        np.testing.assert_allclose(mine.cpu().numpy(), r, rtol=0, atol=3e-5 * np.abs(r).max() + 1e-12)
    # gradients ACCUMULATE into non-zero buffers too (grads_zero off): twice the same step = twice the gradient
    loss2 = hp.forward_backward(Et, Rt, scorer, batch, dE, dR, loss=loss_kind)
    torch.cuda.synchronize()
    assert abs(loss2.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    np.testing.assert_allclose(dE.cpu().numpy(), 2 * ref["dE"], rtol=0, atol=6e-5 * np.abs(ref["dE"]).max() + 1e-12)


def test_wide_slots_replayed_reference_masks(hp):
    """d = 512 with EXPLICIT keep masks (the path that replays masks captured from the reference, okge_dropout.keep)"""
    from open_knowledge_graph_embeddings_amd.hotpath import DropoutSpec
    n_ent, d, n_po, n_sp, p = 400, 512, 40, 24, 0.4
    E, R, z, cand, y = random_problem(77, n_ent, 9, d, n_po, n_sp)
    rng = np.random.default_rng(78)
    masks = {k: rng.random((n, d)) >= p for k, n in (("cand", len(cand)), ("po", n_po), ("sp", n_sp))}
    ref = oracle_step("complex", E, R, z, cand, y, p_ent=p, keep_cand=masks["cand"], keep_po_ent=masks["po"], keep_sp_ent=masks["sp"])
    batch = make_batch(z, cand, n_ent, labels=y)
    batch.drop_cand = DropoutSpec(p=p, keep=dev(masks["cand"].astype(np.uint8)))
    batch.drop_po_ent = DropoutSpec(p=p, keep=dev(masks["po"].astype(np.uint8)))
    batch.drop_sp_ent = DropoutSpec(p=p, keep=dev(masks["sp"].astype(np.uint8)))
    Et, Rt = dev(E), dev(R)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, "complex", batch, dE, dR)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    np.testing.assert_allclose(dE.cpu().numpy(), ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
    np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())


# ------------------------------------------------------------------------------- counter-based dropout
@pytest.mark.parametrize("scorer", ["complex", "distmult"])
def test_philox_dropout_matches_oracle_masks(hp, scorer):
    """The kernels' Philox masks are integer work: identical to the oracle's, so the whole step matches."""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    n_ent, n_rel, d, n_po, n_sp = 500, 12, 200, 40, 37
    E, R, z, cand, y = random_problem(21, n_ent, n_rel, d, n_po, n_sp)
    p, p_rel, seed, step = 0.4, 0.25, 0xDEADBEEFCAFE, 7
    Nc = len(cand)
    keep = {
        "keep_cand": ko.dropout_keep_mask(seed, H.STREAM_CAND, step, Nc, d, p),
        "keep_po_ent": ko.dropout_keep_mask(seed, H.STREAM_PO_ENT, step, n_po, d, p),
        "keep_sp_ent": ko.dropout_keep_mask(seed, H.STREAM_SP_ENT, step, n_sp, d, p),
        "keep_po_rel": ko.dropout_keep_mask(seed, H.STREAM_PO_REL, step, n_po, d, p_rel),
        "keep_sp_rel": ko.dropout_keep_mask(seed, H.STREAM_SP_REL, step, n_sp, d, p_rel),
    }
    ref = oracle_step(scorer, E, R, z, cand, y, p_ent=p, p_rel=p_rel, **keep)
    batch = make_batch(z, cand, n_ent, labels=y)
    batch.drop_cand = H.DropoutSpec(p, seed, H.STREAM_CAND, step)
    batch.drop_po_ent = H.DropoutSpec(p, seed, H.STREAM_PO_ENT, step)
    batch.drop_sp_ent = H.DropoutSpec(p, seed, H.STREAM_SP_ENT, step)
    batch.drop_po_rel = H.DropoutSpec(p_rel, seed, H.STREAM_PO_REL, step)
    batch.drop_sp_rel = H.DropoutSpec(p_rel, seed, H.STREAM_SP_REL, step)
    Et, Rt = dev(E), dev(R)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, scorer, batch, dE, dR)
    out = hp.score(Et, Rt, scorer, batch).cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out, ref["outputs"], rtol=0, atol=1e-5)
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    np.testing.assert_allclose(dE.cpu().numpy(), ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
    np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())


@pytest.mark.usefixtures("production_config")
def test_full_size_distmult_d512_sampled(hp):
    """configs[2]: FB15k-237 LookupDistmultRelationModel d=512, B=512, batch-shared sampled candidates N=10 000."""
    n_ent, n_rel, d = 14543, 239, 512
    E, R, z, cand, y = random_problem(4321, n_ent, n_rel, d, 256, 256, 10000, max_pos=6)
    E *= 0.3
    ref = oracle_step("distmult", E, R, z, cand, y)
    Et, Rt = dev(E), dev(R)
    batch = make_batch(z, cand, n_ent, labels=y)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, "distmult", batch, dE, dR)
    out = hp.score(Et, Rt, "distmult", batch).cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out, ref["outputs"], rtol=0, atol=SCORE_ATOL)
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    np.testing.assert_allclose(dE.cpu().numpy(), ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
    np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())
    # rows outside the sampled candidate set and outside the prefixes get no gradient
    touched = np.zeros(n_ent, bool)
    touched[cand] = True
    touched[z["po_obj"]] = True
    touched[z["sp_subj"]] = True
    assert not dE.cpu().numpy()[~touched].any()


# ------------------------------------------------------------------------------- BASELINE.json full size
@pytest.mark.usefixtures("production_config")
def test_full_size_fb15k237_shape(hp):
    """configs[1]: |E|=14543, |R|=239, d=200, B=512 (256 po + 256 sp), N=14541, 1-vs-all, BCE."""
    n_ent, n_rel, d = 14543, 239, 200
    E, R, z, cand, y = random_problem(1234, n_ent, n_rel, d, 256, 256, None, max_pos=8)
    E *= 1 / 3.0
    ref = oracle_step("complex", E, R, z, cand, y)
    Et, Rt = dev(E), dev(R)
    batch = make_batch(z, cand, n_ent, labels=y)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, "complex", batch, dE, dR)
    out = hp.score(Et, Rt, "complex", batch).cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out, ref["outputs"], rtol=0, atol=SCORE_ATOL)
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    np.testing.assert_allclose(dE.cpu().numpy(), ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
    np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())
    # size-independent properties: (1) the gradient of the summed loss w.r.t. scores sums to
    # sum(sigmoid(x)) - nnz, which the entity gradient inherits through linearity in Q;
    # (2) scoring a candidate subset equals the matching columns of the full score matrix
    sub = np.sort(np.random.default_rng(0).choice(len(cand), 1000, replace=False))
    b2 = make_batch(z, cand[sub], n_ent)
    out_sub = hp.score(Et, Rt, "complex", b2).cpu().numpy()
    np.testing.assert_array_equal(out_sub, out[:, sub])


def _full_size_case(hp, scorer, n_ent, n_rel, d, n_po, n_sp, loss_kind, smoothing, p_ent, scale, seed):
    """one BASELINE-sized step against the oracle: scores, loss, full dE / dR; Philox dropout on when p_ent > 0"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    E, R, z, cand, y = random_problem(seed, n_ent, n_rel, d, n_po, n_sp, None, max_pos=8)
    E *= scale
    kw, batch = {}, make_batch(z, cand, n_ent, labels=y)
    if p_ent > 0:
        sd, step = 0x5EED5EED, 3
        kw = dict(p_ent=p_ent, p_rel=0.0, keep_cand=ko.dropout_keep_mask(sd, H.STREAM_CAND, step, len(cand), d, p_ent),
                  keep_po_ent=ko.dropout_keep_mask(sd, H.STREAM_PO_ENT, step, n_po, d, p_ent),
                  keep_sp_ent=ko.dropout_keep_mask(sd, H.STREAM_SP_ENT, step, n_sp, d, p_ent))
        batch.drop_cand = H.DropoutSpec(p_ent, sd, H.STREAM_CAND, step)
        batch.drop_po_ent = H.DropoutSpec(p_ent, sd, H.STREAM_PO_ENT, step)
        batch.drop_sp_ent = H.DropoutSpec(p_ent, sd, H.STREAM_SP_ENT, step)
    ref = oracle_step(scorer, E, R, z, cand, y, loss_kind, smoothing, **kw)
    Et, Rt = dev(E), dev(R)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, scorer, batch, dE, dR, loss=loss_kind, label_smoothing=smoothing)
    out = hp.score(Et, Rt, scorer, batch).cpu().numpy()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out, ref["outputs"], rtol=0, atol=SCORE_ATOL)
    assert abs(loss.item() - ref["loss"]) <= 3e-5 * abs(ref["loss"])
    np.testing.assert_allclose(dE.cpu().numpy(), ref["dE"], rtol=0, atol=3e-5 * np.abs(ref["dE"]).max())
    np.testing.assert_allclose(dR.cpu().numpy(), ref["dR"], rtol=0, atol=3e-5 * np.abs(ref["dR"]).max())


@pytest.mark.usefixtures("production_config")
def test_full_size_fb15k237_plumbing_shape(hp):
    """configs[0]: |E|=14543, |R|=239, d=64, B=128 (64 po + 64 sp), N=14541, 1-vs-all, BCE, input_dropout 0.4
    (config/fb15k237/fb15k237-complex-kge.yaml with the README's d=64 / batch 128 overrides)."""
    _full_size_case(hp, "complex", 14543, 239, 64, 64, 64, "bce", 0.0, 0.4, 0.5, 2468)


@pytest.mark.usefixtures("production_config")
def test_full_size_fb15k237_shape_kl(hp):
    """configs[1] with the softmax / KL loss (trainer.py:99-101): |E|=14543, d=200, B=512, N=14541, dropout 0.4 --
    the row log-sum-exp over all 14 541 candidates (stats pass + merge) at the full size."""
    _full_size_case(hp, "complex", 14543, 239, 200, 256, 256, "kl", 0.0, 0.4, 1 / 3.0, 1357)


@pytest.mark.usefixtures("production_config")
def test_full_size_fb15k237_shape_smoothing_dropout(hp):
    """configs[1] as the YAML trains it: BCE, input_dropout 0.4, plus label smoothing 0.1 (trainer.py:103-105)"""
    _full_size_case(hp, "complex", 14543, 239, 200, 256, 256, "bce", 0.1, 0.4, 1 / 3.0, 97531)


# --------------------------------------------------------------------------------------- HIP-graph replay
def test_graphed_step_equals_plain_step(hp):
    """GraphedTrainStep (captured once, replayed with padded positives and a device-side dropout counter) walks the
    same trajectory as FusedTrainStep called from Python: same masks, same losses, same tables."""
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep, GraphedTrainStep
    n_ent, n_rel, d, n_po, n_sp = 1200, 20, 200, 70, 58
    problems = [random_problem(500 + i, n_ent, n_rel, d, n_po, n_sp, max_pos=2 + 2 * i) for i in range(4)]
    E, R = problems[0][0], problems[0][1]
    batches = [make_batch(z, cand, n_ent, labels=y) for _, _, z, cand, y in problems]
    assert len({b.nnz for b in batches}) > 1                       # the positives' count really varies
    plain = FusedTrainStep(dev(E.copy()), dev(R.copy()), "complex", lr=0.3, input_dropout=0.4, seed=99)
    inner = FusedTrainStep(dev(E.copy()), dev(R.copy()), "complex", lr=0.3, input_dropout=0.4, seed=99)
    graphed = GraphedTrainStep(inner, batches[0], pos_capacity=max(b.nnz for b in batches) + 13)
    np.testing.assert_array_equal(inner.E.cpu().numpy(), E)        # the capture warm-up left no trace
    for i in range(6):
        b = batches[i % 4]
        lp = float(plain.step(b)[0])
        lg = float(graphed.step(b)[0])
        # the prefix rows' gradients are scattered with float atomics (order varies run to run): equal to ~1e-7
        assert abs(lp - lg) <= 1e-7 * abs(lp), (i, lp, lg)
    for a, b2 in ((inner.E, plain.E), (inner.R, plain.R)):       # Adagrad amplifies that noise on a few tiny gradients
        a, b2 = a.cpu().numpy(), b2.cpu().numpy()
        assert np.isclose(a, b2, rtol=1e-4, atol=1e-5).mean() > 0.9999 and np.abs(a - b2).max() < 1e-3
    with pytest.raises(ValueError):
        graphed.step(make_batch(*[random_problem(9, n_ent, n_rel, d, n_po + 1, n_sp)[i] for i in (2, 3)], n_ent,
                                labels=random_problem(9, n_ent, n_rel, d, n_po + 1, n_sp)[4]))


def test_train_step_lazy_gradient_clear(hp):
    """FusedTrainStep leaves dE uncleared after a 1-vs-all step (the next one overwrites it); a sampled candidate
    list next must see zeros in the rows it does not touch.  1-vs-all, sampled, 1-vs-all against the oracle."""
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    n_ent, n_rel, d = 600, 12, 64
    probs = [random_problem(900, n_ent, n_rel, d, 20, 20), random_problem(901, n_ent, n_rel, d, 20, 20, n_cand=150),
             random_problem(902, n_ent, n_rel, d, 20, 20)]
    E, R = probs[0][0].copy(), probs[0][1].copy()
    st = FusedTrainStep(dev(E.copy()), dev(R.copy()), "complex", lr=0.3)
    sE, sR = np.zeros_like(E), np.zeros_like(R)
    for _, _, z, cand, y in probs:
        loss = float(st.step(make_batch(z, cand, n_ent, labels=y))[0])
        ref = oracle_step("complex", E, R, z, cand, y)
        assert abs(loss - ref["loss"]) <= 3e-5 * abs(ref["loss"])
        ko.adagrad_step(E, ref["dE"], sE, 0.3)
        ko.adagrad_step(R, ref["dR"], sR, 0.3)
    np.testing.assert_allclose(st.sumE.cpu().numpy(), sE, rtol=2e-4, atol=1e-12)     # accumulators see every gradient
    assert np.isclose(st.E.cpu().numpy(), E, rtol=1e-3, atol=1e-4).mean() > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["fb_1vsall_dropout", "sampled_list_distmult", "kl"])
def test_train_step_with_fused_update_is_bit_identical(okge_lib, monkeypatch, case):
    """okge_train_step (the Adagrad update inside the step's last launches) against okge_train_forward_backward followed by
    okge_adagrad_step2: the SAME arithmetic element for element -- tables and accumulators bit-equal after several steps,
    except where float atomics order the prefix gradients (rows that several batch rows name: compared to 1e-6)"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    rng = np.random.default_rng(77)
    if case == "sampled_list_distmult":
        n_ent, n_rel, d, B, N, scorer, loss, p = 3000, 40, 64, 256, 900, "distmult", "bce", 0.0
    elif case == "kl":
        n_ent, n_rel, d, B, N, scorer, loss, p = 2000, 30, 200, 192, 1998, "complex", "kl", 0.0
    else:
        n_ent, n_rel, d, B, N, scorer, loss, p = 14543, 239, 200, 512, 14541, "complex", "bce", 0.4
    E0 = (rng.standard_normal((n_ent, d)) * 0.1).astype(np.float32)
    R0 = (rng.standard_normal((n_rel, d)) * 0.1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731

    def batches():
        r2 = np.random.default_rng(5)
        out = []
        for _ in range(4):
            rows = np.repeat(np.arange(B), 2)
            cols = r2.integers(0, N, rows.size)
            key = np.unique(cols.astype(np.int64) * B + rows)
            kw = dict(cand_first=2, n_cand=N) if N == n_ent - 2 else dict(cand_ids=t(r2.permutation(np.arange(2, n_ent))[:N].astype(np.int32)), cand_unique=True)
            out.append(H.PrefixBatch(po_rel=t(r2.integers(2, n_rel, B // 2).astype(np.int32)), po_obj=t(r2.integers(2, n_ent, B // 2).astype(np.int32)),
                                     sp_subj=t(r2.integers(2, n_ent, B // 2).astype(np.int32)), sp_rel=t(r2.integers(2, n_rel, B // 2).astype(np.int32)),
                                     pos_row=t((key % B).astype(np.int32)), pos_col=t((key // B).astype(np.int32)), **kw))
        return out
    runs = {}
    bs = batches()
    for fused in ("1", "0"):
        monkeypatch.setenv("OKGE_FUSED_UPDATE", fused)
        st = FusedTrainStep(t(E0), t(R0), scorer, loss=loss, lr=0.3, input_dropout=p, seed=3)
        assert st.fuse_update == (fused == "1")
        first = float(st.step(bs[0])[0])
        torch.cuda.synchronize()
        after_one = (st.E.clone(), st.sumE.clone(), st.R.clone(), st.sumR.clone())
        losses = [first] + [float(st.step(b)[0]) for b in bs[1:] + bs]
        torch.cuda.synchronize()
        runs[fused] = (after_one, (st.E.clone(), st.R.clone(), st.sumE.clone(), st.sumR.clone()), st.dR.clone(), losses)
        if fused == "1":
            assert int(st._prefix_flags.abs().sum()) == 0            # the flags are back to zero after every step
    a, b = runs["1"], runs["0"]
    # ONE step from the same state: every entity row no prefix names is bit-equal (same arithmetic, another launch); the <= B
    # prefix rows and the relation table take float atomics in either run (their order is not fixed): 1e-6
    prefix_rows = torch.cat([bs[0].po_obj, bs[0].sp_subj]).long().unique()
    mask = torch.ones(n_ent, dtype=torch.bool, device="cuda")
    mask[prefix_rows] = False
    for x, y in zip(a[0][:2], b[0][:2]):
        assert torch.equal(x[mask], y[mask])
        np.testing.assert_allclose(x[~mask].cpu().numpy(), y[~mask].cpu().numpy(), rtol=0, atol=1e-6)
    for x, y in zip(a[0][2:], b[0][2:]):                             # (the relation table: ~2 batch rows add into every row)
        np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=5e-6)
    # eight steps: the runs stay together (the atomics' order noise of the prefix rows feeds the next step's queries)
    np.testing.assert_allclose(a[3], b[3], rtol=2e-6)
    for x, y in zip(a[1], b[1]):                                     # (Adagrad amplifies that noise on a few tiny gradients)
        x, y = x.cpu().numpy(), y.cpu().numpy()
        assert np.isclose(x, y, rtol=1e-4, atol=1e-5).mean() > 0.999 and np.abs(x - y).max() < 1e-3
    assert float(a[2].abs().max()) == 0 and float(b[2].abs().max()) == 0
