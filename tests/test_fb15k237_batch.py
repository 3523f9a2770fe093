"""Full BASELINE-size parity on a real FB15k-237 batch (tests/golden/g10_fb15k237_batch.npz): the first 256 po + 256 sp
prefixes of valid.txt as the reference's dataset class and collate function produce them, scored, ranked and
back-propagated by the reference's LookupComplexRelationModel (d=200, |E|=14543).  The embedding tables are not stored:
they are regenerated from the seed (same constructor order as the reference => identical torch CPU RNG stream), which
the stored checksums verify before anything else is compared.

CPU: the oracle.  GPU: the HIP path through the C ABI (scores within the north-star 1e-4, in fact ~1e-6)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import kge_oracle as ko


def tables(z):
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    torch.manual_seed(int(z["seed"]))
    m = Models.LookupComplexRelationModel(entity_slot_size=int(z["d"]), input_dropout=0.0, init_std=0.1, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=int(z["n_ent"]),
                                                                              relations_size=int(z["n_rel"])))
    E, R = m.entity_embedding.weight.detach().numpy().copy(), m.relation_embedding.weight.detach().numpy().copy()
    chk = [E.sum(dtype=np.float64), np.abs(E).sum(dtype=np.float64), R.sum(dtype=np.float64), float(E[5, 7]),
           float(E[-1, -1]), float(R[3, 4])]
    np.testing.assert_array_equal(np.asarray(chk), z["table_check"])          # same tables as the reference built
    return m, E, R


def check_scores(x, z):
    np.testing.assert_allclose(x[192:320, 1000:1128], z["score_slice"], rtol=0, atol=1e-4)
    assert np.abs(x[192:320, 1000:1128] - z["score_slice"]).max() < 5e-6
    np.testing.assert_allclose(x.astype(np.float64).sum(1), z["score_row_sum"], rtol=0, atol=2e-3)   # 14541 terms per row
    np.testing.assert_allclose(np.abs(x).max(1), z["score_row_absmax"], rtol=0, atol=1e-5)


def check_ranks(ranks, z):
    """the rank RULE is exact; against the reference's own float scores a near-tie may flip a neighbour"""
    ref = z["ranks"]
    assert ranks.shape == ref.shape
    assert (ranks != ref).mean() < 0.02 and np.abs(ranks - ref).max() <= 2
    mrr = float((1.0 / (ranks + 1.0)).mean())
    assert abs(mrr - float((1.0 / (ref + 1.0)).mean())) < 1e-5


def dense(coords, shape, dtype):
    y = np.zeros(shape, dtype)
    y[coords[:, 0], coords[:, 1]] = 1
    return y


def test_oracle_on_fb15k237_batch():
    z = golden("g10_fb15k237_batch")
    _, E, R = tables(z)
    N = E.shape[0] - 2
    po, sp = (z["po_rel"], z["po_obj"]), (z["sp_subj"], z["sp_rel"])
    y = dense(z["labels"], (512, N), np.float32)
    out = ko.step_forward_backward(ko.COMPLEX, E, R, po, sp, np.arange(2, E.shape[0]), y, normalizer=float(z["normalizer"]))
    check_scores(out["outputs"], z)
    assert abs(out["loss"] - float(z["loss"])) <= 2e-6 * float(z["loss"]) and float(z["n_labels"]) == len(z["labels"])
    np.testing.assert_allclose(out["dE"].astype(np.float64).sum(1), z["dE_row_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["dE"][2:66, :16], z["dE_slice"], rtol=0, atol=2e-5 * np.abs(z["dE_slice"]).max())
    np.testing.assert_allclose(out["dR"], z["dR"], rtol=0, atol=2e-5 * np.abs(z["dR"]).max())
    ranks = ko.filtered_ranks(out["outputs"], dense(z["filter"], (512, N), bool), z["row_ptr"], z["grp_ptr"], z["ids"])
    check_ranks(ranks, z)


@pytest.mark.gpu
@pytest.mark.usefixtures("production_config")
def test_hip_on_fb15k237_batch(okge_lib):
    from open_knowledge_graph_embeddings_amd import hotpath as H
    z = golden("g10_fb15k237_batch")
    m, E, R = tables(z)
    dev = lambda a, dt=None: (torch.from_numpy(np.ascontiguousarray(a)) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dt)).cuda()  # noqa: E731
    m = m.cuda().eval()
    po_rel, po_obj, sp_subj, sp_rel = (dev(z[k]) for k in ("po_rel", "po_obj", "sp_subj", "sp_rel"))
    # 1. the model API, as the reference's evaluation calls it
    with torch.no_grad():                                                   # (trainer.py:366)
        x = torch.cat([m.po_prefix_score(po_rel, po_obj), m.sp_prefix_score(sp_subj, sp_rel)], 0)
    check_scores(x.cpu().numpy(), z)
    # 2. filtered ranks from CSR answer groups / filter
    hp = H.HotPath("cuda:0")
    f = z["filter"]
    fp = np.concatenate([[0], np.cumsum(np.bincount(f[:, 0], minlength=512))]).astype(np.int64)
    ranks = hp.filtered_ranks(x.contiguous(), dev(fp), dev(f[:, 1].astype(np.int32)), dev(z["row_ptr"]), dev(z["grp_ptr"]),
                              dev(z["ids"])).cpu().numpy()
    check_ranks(ranks, z)
    # the rule itself is bit-exact on identical scores
    N = E.shape[0] - 2
    np.testing.assert_array_equal(ranks, ko.filtered_ranks(x.cpu().numpy(), dense(f, (512, N), bool), z["row_ptr"],
                                                           z["grp_ptr"], z["ids"]))
    # 3. the fused training step's forward + backward
    lab = z["labels"]
    order = np.lexsort((lab[:, 0], lab[:, 1]))                     # by column, then row
    batch = H.PrefixBatch(po_rel=po_rel.view(-1), po_obj=po_obj.view(-1), sp_subj=sp_subj.view(-1), sp_rel=sp_rel.view(-1),
                          pos_row=dev(lab[order, 0].astype(np.int32)), pos_col=dev(lab[order, 1].astype(np.int32)),
                          cand_first=2, n_cand=N)
    Et, Rt = dev(E), dev(R)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    loss = hp.forward_backward(Et, Rt, "complex", batch, dE, dR, normalizer=float(z["normalizer"]), grads_zero=True)
    assert abs(float(loss[0]) - float(z["loss"])) <= 3e-6 * float(z["loss"])
    dEn, dRn = dE.cpu().numpy(), dR.cpu().numpy()
    np.testing.assert_allclose(dEn.astype(np.float64).sum(1), z["dE_row_sum"], rtol=0, atol=2e-9)
    np.testing.assert_allclose(dEn[2:66, :16], z["dE_slice"], rtol=0, atol=3e-5 * np.abs(z["dE_slice"]).max())
    np.testing.assert_allclose(dRn, z["dR"], rtol=0, atol=3e-5 * np.abs(z["dR"]).max())
    assert abs(np.abs(dEn).sum(dtype=np.float64) - float(z["dE_abs_sum"])) <= 1e-5 * float(z["dE_abs_sum"])


# ------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] at its size, through the reference (tests/golden/g14_distmult_d512_sampled.npz): LookupDistmultRelationModel
# d = 512 on 512 real FB15k-237 prefixes, candidates = the batch-shared list the reference's collate built (answers + numpy-sampled
# fill-up negatives, N = 10 000), AddLossModule bce forward + (loss / normalizer).backward()
# ------------------------------------------------------------------------------------------------------------------------
def tables_distmult(z):
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    torch.manual_seed(int(z["seed"]))
    m = Models.LookupDistmultRelationModel(entity_slot_size=int(z["d"]), input_dropout=0.0, init_std=0.1, sparse=False,
                                           train_data=EntityRelationDatasetMeta(entities_size=int(z["n_ent"]),
                                                                               relations_size=int(z["n_rel"])))
    E, R = m.entity_embedding.weight.detach().numpy().copy(), m.relation_embedding.weight.detach().numpy().copy()
    chk = [E.sum(dtype=np.float64), np.abs(E).sum(dtype=np.float64), R.sum(dtype=np.float64), float(E[5, 7]),
           float(E[-1, -1]), float(R[3, 4])]
    np.testing.assert_array_equal(np.asarray(chk), z["table_check"])          # same tables as the reference built
    return m, E, R


def check_g14(x, loss, dE, dR, z):
    np.testing.assert_allclose(x[192:320, 4000:4128], z["score_slice"], rtol=0, atol=1e-4)
    assert np.abs(x[192:320, 4000:4128] - z["score_slice"]).max() < 1e-5
    np.testing.assert_allclose(x.astype(np.float64).sum(1), z["score_row_sum"], rtol=0, atol=2e-3)      # 10 000 terms per row
    np.testing.assert_allclose(np.abs(x).max(1), z["score_row_absmax"], rtol=0, atol=1e-5)
    assert abs(loss - float(z["loss"])) <= 3e-6 * float(z["loss"])
    np.testing.assert_allclose(dE.astype(np.float64).sum(1), z["dE_row_sum"], rtol=0, atol=2e-9)
    assert abs(np.abs(dE).sum(dtype=np.float64) - float(z["dE_abs_sum"])) <= 1e-5 * float(z["dE_abs_sum"])
    rows = z["cand"][:64].astype(np.int64)
    np.testing.assert_allclose(dE[rows, :16], z["dE_slice"], rtol=0, atol=3e-5 * np.abs(z["dE_slice"]).max())
    np.testing.assert_allclose(dR.astype(np.float64).sum(1), z["dR_row_sum"], rtol=0, atol=2e-9)
    np.testing.assert_allclose(dR[:32, :64], z["dR_slice"], rtol=0, atol=3e-5 * np.abs(z["dR_slice"]).max())
    touched = np.zeros(dE.shape[0], bool)
    touched[z["cand"]] = True
    touched[z["po_obj"]] = True
    touched[z["sp_subj"]] = True
    assert not dE[~touched].any() and len(np.unique(z["cand"])) == 10000


def test_oracle_on_reference_distmult_d512_batch():
    z = golden("g14_distmult_d512_sampled")
    _, E, R = tables_distmult(z)
    y = dense(z["labels"], (512, 10000), np.float32)
    out = ko.step_forward_backward(ko.DISTMULT, E, R, (z["po_rel"], z["po_obj"]), (z["sp_subj"], z["sp_rel"]), z["cand"], y,
                                   normalizer=float(z["normalizer"]))
    assert float(z["n_labels"]) == len(z["labels"]) and float(z["normalizer"]) == 512 * 10000
    check_g14(out["outputs"], out["loss"], out["dE"], out["dR"], z)


@pytest.mark.gpu
@pytest.mark.usefixtures("production_config")
def test_hip_on_reference_distmult_d512_batch(okge_lib):
    """the register-tile kernel (stream-K launch), the 32-candidate-chunk dQ kernel and the score sweep of slot sizes above 256
    against what the REFERENCE computed on this batch"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    z = golden("g14_distmult_d512_sampled")
    _, E, R = tables_distmult(z)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    lab = z["labels"]
    order = np.lexsort((lab[:, 0], lab[:, 1]))                     # by column, then row
    batch = H.PrefixBatch(po_rel=dev(z["po_rel"]), po_obj=dev(z["po_obj"]), sp_subj=dev(z["sp_subj"]), sp_rel=dev(z["sp_rel"]),
                          pos_row=dev(lab[order, 0].astype(np.int32)), pos_col=dev(lab[order, 1].astype(np.int32)),
                          cand_ids=dev(z["cand"]), cand_unique=True)
    hp = H.HotPath("cuda:0")
    Et, Rt = dev(E), dev(R)
    dE, dR = torch.zeros_like(Et), torch.zeros_like(Rt)
    x = torch.empty((512, 10000), device="cuda:0")
    loss = hp.forward_backward(Et, Rt, "distmult", batch, dE, dR, normalizer=float(z["normalizer"]), grads_zero=True, scores=x)
    torch.cuda.synchronize()
    check_g14(x.cpu().numpy(), float(loss[0]), dE.cpu().numpy(), dR.cpu().numpy(), z)
