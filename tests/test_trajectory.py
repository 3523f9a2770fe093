"""Training trajectories of the reference, replayed: loss curve, trained tables, and the filtered ranks / MRR the
reference's compute_metrics reports on those TRAINED tables (north_star: "MRR parity").

  g7_traj_complex      20 steps, toy size (|E| = 120, d = 32)              trainer.py:181-257 -> dataset.py:423-453
  g11_traj_fb15k237    30 steps at the BASELINE size on real FB15k-237 batches produced by the reference's dataset class
                       and collate function, evaluated on the first 512-prefix batch of valid.txt

CPU: the NumPy oracle walks them (pins the oracle).  GPU: FusedTrainStep (fused forward + loss + backward + Adagrad,
dropout 0) walks them through the C ABI and the evaluation goes through okge_evaluate-style scoring + okge_filtered_ranks.
Tolerances: loss curve 5e-5 relative per step; final MRR 1e-3 absolute (observed: see the asserts); ranks against the
reference's own ranks on the trained tables: the mismatch RATE is asserted and printed (fp32 summation order differs, so
a handful of near-ties may move by one place; the rank RULE on identical scores is bit-exact, G5).
Two comparisons are made: (1) on tables produced by OUR OWN 30-step trajectory (scores drift ~1e-3 from the reference's
after 30 steps of different summation order, and with a mean rank of ~6400 among 14 541 dense scores that moves ~5 % of
the groups by a few places -- asserted <= 10 %, |delta| <= 8 of ~6400, MRR within 1e-3); (2) on the reference's OWN trained
rows (stored for a 2048-candidate subset): identical tables, so only the fp32 summation order of one score differs --
asserted <= 0.5 % (observed 0)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import kge_oracle as ko


def dense(coords, shape, dtype=np.float32):
    y = np.zeros(shape, dtype)
    y[coords[:, 0], coords[:, 1]] = 1
    return y


def fb_tables(z):
    """initial tables of G11: regenerated from the seed through the same constructor order as the reference's model
    (identical torch CPU RNG stream), guarded by the stored checksums"""
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    torch.manual_seed(int(z["seed"]))
    m = Models.LookupComplexRelationModel(entity_slot_size=int(z["d"]), input_dropout=0.0, init_std=0.1, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=int(z["n_ent"]),
                                                                              relations_size=int(z["n_rel"])))
    E, R = m.entity_embedding.weight.detach().numpy().copy(), m.relation_embedding.weight.detach().numpy().copy()
    chk = [E.sum(dtype=np.float64), np.abs(E).sum(dtype=np.float64), R.sum(dtype=np.float64), float(E[5, 7]),
           float(E[-1, -1]), float(R[3, 4])]
    np.testing.assert_array_equal(np.asarray(chk), z["table_check"])
    return E, R


def checks(a):
    return np.asarray([a.sum(dtype=np.float64), np.abs(a).sum(dtype=np.float64), (a.astype(np.float64) ** 2).sum()])


def mrr_of(ranks):
    return float((1.0 / (ranks.astype(np.float64) + 1.0)).mean())


def compare_ranks(ranks, ref, max_rate, what, max_delta=2):
    """rank agreement with the reference on trained tables; returns the mismatch rate"""
    assert ranks.shape == ref.shape
    rate = float((ranks != ref).mean())
    print(f"[{what}] rank mismatches vs reference: {int((ranks != ref).sum())} of {len(ref)} groups (rate {rate:.4f}), "
          f"max |delta| {int(np.abs(ranks - ref).max())}, MRR {mrr_of(ranks):.6f} vs {mrr_of(ref):.6f}")
    assert rate <= max_rate and np.abs(ranks - ref).max() <= max_delta
    assert abs(mrr_of(ranks) - mrr_of(ref)) < 1e-3
    return rate


def close_tables(a, ref, what):
    """Trained weights after 30 Adagrad steps.  The first update of an element is lr * g / (|g| + 1e-8): where a
    gradient element is itself ~1e-9 (a cancellation), its 1e-10 summation noise moves the weight by ~lr * 1e-2 -- a
    handful of the 3 M elements.  So: 99.9 % of the elements within 2e-3 of the table's scale, every element within 2e-2."""
    scale = np.abs(ref).max()
    err = np.abs(a - ref)
    assert (err <= 2e-3 * scale).mean() >= 0.999, (what, float((err <= 2e-3 * scale).mean()))
    assert err.max() <= 2e-2 * scale, (what, float(err.max()), float(scale))


def check_final_tables(E, R, z, rtol):
    np.testing.assert_allclose(checks(E), z["E_check"], rtol=rtol)
    np.testing.assert_allclose(checks(R), z["R_check"], rtol=rtol)
    close_tables(E[2:66], z["E_rows"], "E rows 2..65")
    close_tables(R, z["R_final"], "R")


# ------------------------------------------------------------------------------------------------------ oracle (CPU)
def test_oracle_g7_eval_on_trained_tables():
    z = golden("g7_traj_complex")
    filt = z["eval_filter"].astype(bool)
    # the rule on the reference's own scores: bit-exact
    np.testing.assert_array_equal(ko.filtered_ranks(z["eval_scores"], filt, z["eval_row_ptr"], z["eval_grp_ptr"], z["eval_ids"]),
                                  z["eval_ranks"])
    assert abs(mrr_of(z["eval_ranks"]) - float(z["eval_m_mrr"])) < 1e-6
    # the oracle's own trajectory -> its scores -> ranks
    E, R = z["E0"].copy(), z["R0"].copy()
    sE, sR = np.zeros_like(E), np.zeros_like(R)
    for step in range(int(z["nsteps"])):
        i = step % 4
        out = ko.step_forward_backward(ko.COMPLEX, E, R, (z[f"b{i}_po_rel"], z[f"b{i}_po_obj"]),
                                       (z[f"b{i}_sp_subj"], z[f"b{i}_sp_rel"]), z["cand"], z[f"b{i}_labels"])
        ko.adagrad_step(E, out["dE"], sE, 0.3)
        ko.adagrad_step(R, out["dR"], sR, 0.3)
    x = ko.step_forward_backward(ko.COMPLEX, E, R, (z["b0_po_rel"], z["b0_po_obj"]), (z["b0_sp_subj"], z["b0_sp_rel"]),
                                 z["cand"], z["b0_labels"], want_grads=False)["outputs"]
    assert np.abs(x - z["eval_scores"]).max() < 2e-3        # 20 steps of accumulated fp32 differences
    compare_ranks(ko.filtered_ranks(x, filt, z["eval_row_ptr"], z["eval_grp_ptr"], z["eval_ids"]), z["eval_ranks"], 0.03, "oracle g7")


def test_oracle_g11_trajectory_and_mrr():
    z = golden("g11_traj_fb15k237")
    E, R = fb_tables(z)
    N = E.shape[0] - 2
    sE, sR = np.zeros_like(E), np.zeros_like(R)
    cand = np.arange(2, E.shape[0])
    for step in range(int(z["nsteps"])):
        y = dense(z[f"s{step}_labels"], (512, N))
        out = ko.step_forward_backward(ko.COMPLEX, E, R, (z[f"s{step}_po_rel"], z[f"s{step}_po_obj"]),
                                       (z[f"s{step}_sp_subj"], z[f"s{step}_sp_rel"]), cand, y,
                                       normalizer=float(z[f"s{step}_normalizer"]))
        assert abs(out["loss"] - z["losses"][step]) <= 5e-5 * z["losses"][step], (step, out["loss"], z["losses"][step])
        ko.adagrad_step(E, out["dE"], sE, float(z["lr"]))
        ko.adagrad_step(R, out["dR"], sR, float(z["lr"]))
    check_final_tables(E, R, z, rtol=1e-4)
    x = ko.step_forward_backward(ko.COMPLEX, E, R, (z["eval_po_rel"], z["eval_po_obj"]), (z["eval_sp_subj"], z["eval_sp_rel"]),
                                 cand, np.zeros((512, N), np.float32), want_grads=False)["outputs"]
    assert np.abs(x[192:320, 1000:1128] - z["eval_score_slice"]).max() < 2e-2
    ranks = ko.filtered_ranks(x, dense(z["eval_filter"], (512, N), bool), z["eval_row_ptr"], z["eval_grp_ptr"], z["eval_ids"])
    compare_ranks(ranks, z["eval_ranks"], 0.10, "oracle g11 (own 30-step trajectory)", max_delta=8)
    m, _ = ko.metrics_from_ranks(ranks, z["eval_row_ptr"])
    assert abs(m["mrr"] - float(z["eval_m_mrr"])) < 1e-3


def _sub_tables(z):
    """tables holding the reference's TRAINED rows where the candidate-subset evaluation needs them (zeros elsewhere)"""
    E = np.zeros((int(z["n_ent"]), int(z["d"])), np.float32)
    E[z["trained_row_ids"]] = z["trained_rows"]
    return E, z["R_final"].copy()


def test_oracle_ranks_on_reference_trained_tables():
    """identical (reference-trained) tables, 2048-candidate subset: only the fp32 summation order of the scores differs"""
    z = golden("g11_traj_fb15k237")
    E, R = _sub_tables(z)
    cand = z["sub_cand_ids"]
    x = ko.step_forward_backward(ko.COMPLEX, E, R, (z["eval_po_rel"], z["eval_po_obj"]), (z["eval_sp_subj"], z["eval_sp_rel"]),
                                 cand, np.zeros((512, len(cand)), np.float32), want_grads=False)["outputs"]
    assert np.abs(x[192:320, 1000:1128] - z["sub_scores_slice"]).max() < 1e-4
    ranks = ko.filtered_ranks(x, dense(z["sub_filter"], x.shape, bool), z["sub_row_ptr"], z["sub_grp_ptr"], z["sub_ids"])
    rate = compare_ranks(ranks, z["sub_ranks"], 0.005, "oracle, reference-trained tables")
    assert rate <= 0.005


# ------------------------------------------------------------------------------------------------------ HIP (GPU)
def _dev(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dt is None else t.to(dt)).cuda()


def _batch(H, z, pre, N):
    lab = z[pre + "labels"]
    order = np.lexsort((lab[:, 0], lab[:, 1]))                     # by column, then row
    return H.PrefixBatch(po_rel=_dev(z[pre + "po_rel"].reshape(-1)), po_obj=_dev(z[pre + "po_obj"].reshape(-1)),
                         sp_subj=_dev(z[pre + "sp_subj"].reshape(-1)), sp_rel=_dev(z[pre + "sp_rel"].reshape(-1)),
                         pos_row=_dev(lab[order, 0].astype(np.int32)), pos_col=_dev(lab[order, 1].astype(np.int32)),
                         cand_first=2, n_cand=N)


@pytest.mark.gpu
@pytest.mark.usefixtures("production_config")
def test_hip_g7_trajectory_and_ranks(okge_lib):
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    z = golden("g7_traj_complex")
    E, R = _dev(z["E0"]), _dev(z["R0"])
    N = E.shape[0] - 2
    ts = FusedTrainStep(E, R, "complex", loss="bce", lr=0.3)
    batches = []
    for i in range(4):
        y = z[f"b{i}_labels"]
        coords = np.argwhere(y > 0).astype(np.int32)
        zz = {"x_po_rel": z[f"b{i}_po_rel"], "x_po_obj": z[f"b{i}_po_obj"], "x_sp_subj": z[f"b{i}_sp_subj"],
              "x_sp_rel": z[f"b{i}_sp_rel"], "x_labels": coords}
        batches.append((_batch(H, zz, "x_", N), y.size))
    for step in range(int(z["nsteps"])):
        b, n = batches[step % 4]
        loss = float(ts.step(b)[0]) / n
        assert abs(loss - z["losses"][step]) <= 5e-5 * abs(z["losses"][step]), (step, loss, z["losses"][step])
    np.testing.assert_allclose(E.cpu().numpy(), z["E"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(R.cpu().numpy(), z["R"], rtol=1e-3, atol=2e-4)
    hp = ts.engine
    x = hp.score(E, R, "complex", batches[0][0])
    assert np.abs(x.cpu().numpy() - z["eval_scores"]).max() < 2e-3
    f = np.argwhere(z["eval_filter"] > 0)
    fptr = np.concatenate([[0], np.cumsum(np.bincount(f[:, 0], minlength=x.shape[0]))]).astype(np.int64)
    ranks = hp.filtered_ranks(x.contiguous(), _dev(fptr), _dev(f[:, 1].astype(np.int32)), _dev(z["eval_row_ptr"]),
                              _dev(z["eval_grp_ptr"]), _dev(z["eval_ids"])).cpu().numpy()
    compare_ranks(ranks, z["eval_ranks"], 0.03, "hip g7")


@pytest.mark.gpu
@pytest.mark.usefixtures("production_config")
def test_hip_g11_trajectory_and_mrr(okge_lib):
    """the north-star 'MRR parity' on the HIP path: 30 reference steps at the BASELINE size, then filtered MRR"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    z = golden("g11_traj_fb15k237")
    E0, R0 = fb_tables(z)
    E, R = _dev(E0), _dev(R0)
    N = E.shape[0] - 2
    ts = FusedTrainStep(E, R, "complex", loss="bce", lr=float(z["lr"]))
    worst = 0.0
    for step in range(int(z["nsteps"])):
        loss = float(ts.step(_batch(H, z, f"s{step}_", N), normalizer=float(z[f"s{step}_normalizer"]))[0])
        rel = abs(loss - z["losses"][step]) / z["losses"][step]
        worst = max(worst, rel)
        assert rel <= 5e-5, (step, loss, z["losses"][step])
    print(f"[hip g11] worst relative loss deviation over {int(z['nsteps'])} steps: {worst:.2e}")
    En, Rn = E.cpu().numpy(), R.cpu().numpy()
    check_final_tables(En, Rn, z, rtol=1e-4)
    np.testing.assert_allclose(checks(ts.sumE.cpu().numpy()), z["sumE_check"], rtol=1e-4)
    # evaluation on the trained tables (eval mode: no dropout), first 512-prefix batch of valid.txt
    hp = ts.engine
    eb = H.PrefixBatch(po_rel=_dev(z["eval_po_rel"]), po_obj=_dev(z["eval_po_obj"]), sp_subj=_dev(z["eval_sp_subj"]),
                       sp_rel=_dev(z["eval_sp_rel"]), cand_first=2, n_cand=N)
    x = hp.score(E, R, "complex", eb)
    xs = x[192:320, 1000:1128].cpu().numpy()
    print(f"[hip g11] max |score - reference| on the stored slice after training: {np.abs(xs - z['eval_score_slice']).max():.2e}")
    # |x| reaches 20 here; after 30 optimisation steps in a different summation order a few weights differ by up to
    # ~5e-3 (close_tables), which shows as <= 2e-2 on a score (the 1e-4 score bound holds on identical tables: next test)
    assert np.abs(xs - z["eval_score_slice"]).max() < 2e-2
    f = z["eval_filter"]
    fptr = np.concatenate([[0], np.cumsum(np.bincount(f[:, 0], minlength=512))]).astype(np.int64)
    ranks = hp.filtered_ranks(x.contiguous(), _dev(fptr), _dev(f[:, 1].astype(np.int32)), _dev(z["eval_row_ptr"]),
                              _dev(z["eval_grp_ptr"]), _dev(z["eval_ids"])).cpu().numpy()
    # observed: 46 of 723 groups (6.4 %) move, none by more than 4 places of ~6400 (30 steps of another summation order;
    # on IDENTICAL tables -- the next test and G13 -- it is 1 of 723 by one place); bound = observed + 50 %
    compare_ranks(ranks, z["eval_ranks"], 0.096, "hip g11 (own 30-step trajectory)", max_delta=4)
    m, _ = ko.metrics_from_ranks(ranks, z["eval_row_ptr"])
    assert abs(m["mrr"] - float(z["eval_m_mrr"])) < 1e-3
    for k in ("h1", "h3", "h10", "h50"):
        assert abs(m[k] - float(z["eval_m_" + k])) < 5e-3


@pytest.mark.gpu
@pytest.mark.usefixtures("production_config")
def test_hip_ranks_on_reference_trained_tables(okge_lib):
    """rank mismatch rate of the HIP path against the reference on IDENTICAL trained tables (expected 0; a score pair
    closer than fp32 summation noise may swap)"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    z = golden("g11_traj_fb15k237")
    En, Rn = _sub_tables(z)
    E, R = _dev(En), _dev(Rn)
    hp = H.HotPath("cuda:0")
    eb = H.PrefixBatch(po_rel=_dev(z["eval_po_rel"]), po_obj=_dev(z["eval_po_obj"]), sp_subj=_dev(z["eval_sp_subj"]),
                       sp_rel=_dev(z["eval_sp_rel"]), cand_ids=_dev(z["sub_cand_ids"]))
    x = hp.score(E, R, "complex", eb)
    assert np.abs(x[192:320, 1000:1128].cpu().numpy() - z["sub_scores_slice"]).max() < 1e-4      # north-star bound
    f = z["sub_filter"]
    fptr = np.concatenate([[0], np.cumsum(np.bincount(f[:, 0], minlength=512))]).astype(np.int64)
    ranks = hp.filtered_ranks(x.contiguous(), _dev(fptr), _dev(f[:, 1].astype(np.int32)), _dev(z["sub_row_ptr"]),
                              _dev(z["sub_grp_ptr"]), _dev(z["sub_ids"])).cpu().numpy()
    rate = compare_ranks(ranks, z["sub_ranks"], 0.005, "hip, reference-trained tables")
    assert rate <= 0.005
    np.testing.assert_array_equal(ranks, ko.filtered_ranks(x.cpu().numpy(), dense(f, tuple(x.shape), bool), z["sub_row_ptr"],
                                                           z["sub_grp_ptr"], z["sub_ids"]))
