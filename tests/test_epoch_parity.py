"""Epoch-scale training parity (north_star: results match the reference on identical FB15k-237 inputs): the reference trains
THREE whole passes over its own loader -- test.txt as the training split (train.txt is absent upstream), shuffle off, dropout 0,
AddLossModule bce + OptimRegime Adagrad lr 0.3 wd 1e-10, the statements of Trainer.compute_one_epoch (openkge/trainer.py:274-354,
:217-257) -- and evaluates over all of valid.txt (tests/golden/g15_epochs_fb15k237.npz: per-step loss / normalizer, the shape
of every batch, the final meters).  Here the whole chain of THIS build walks the same passes on the same files:
    tests/golden/fb15k237/*.gz -> load_dataset_tensors (row f3) -> OneToNBatchProducer(shuffle=False) (row f1)
    -> FusedTrainStep (rows a1-a8) x 132 steps -> FusedEvaluator over valid.txt (rows a1-a5, a9)
CPU: the producer feeds exactly the reference loader's batches (rows per direction, positives, normalizer, step for step) and
the NumPy oracle walks the first pass.  GPU: loss curve and final MRR / Hits against the reference's."""
import numpy as np
import pytest
import torch

from conftest import fb15k237_dir, golden
from oracle import kge_oracle as ko


def _tables(z):
    """initial tables regenerated from the seed through the same constructor order as the reference's model (identical
    torch CPU RNG stream), guarded by the stored checksums"""
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


def _producers(tmp_path, z, device):
    from open_knowledge_graph_embeddings_amd.dataset import OneToNBatchProducer, load_dataset_tensors
    out, all_splits, _ = load_dataset_tensors(fb15k237_dir(tmp_path), train_input_file="test.txt")
    kw = dict(batch_size=int(z["batch_size"]), device=device, shuffle=False)
    train = OneToNBatchProducer(*out["train"], all_splits, int(z["n_ent"]), is_training_data=True, drop_last=True, **kw)
    valid = OneToNBatchProducer(*out["valid"], all_splits, int(z["n_ent"]), is_training_data=False, drop_last=False, **kw)
    return train, valid


@pytest.mark.both
def test_producer_feeds_the_reference_loaders_batches_and_oracle_first_pass(okge_lib, tmp_path):
    z = golden("g15_epochs_fb15k237")
    train, _ = _producers(tmp_path, z, "cpu")
    steps_per_pass = len(z["losses"]) // int(z["n_epochs"])
    assert len(train) == steps_per_pass == 44
    E, R = _tables(z)
    sumE, sumR = np.zeros_like(E), np.zeros_like(R)
    N = E.shape[0] - 2
    cand = np.arange(2, E.shape[0])
    worst = 0.0
    for step, cb in enumerate(train):
        b = cb.batch
        # the same batch as the reference's loader produced at this step: rows per direction, positives, normalizer
        np.testing.assert_array_equal([b.n_po, b.n_sp, b.nnz, cb.normalizer_loss], z["batch_shape"][step])
        if step >= 12:                                     # (the oracle walks the first steps at full size: ~0.3 s each)
            continue
        y = np.zeros((b.B, N), np.float32)
        y[b.pos_row.numpy(), b.pos_col.numpy()] = 1
        po = (b.po_rel.numpy(), b.po_obj.numpy()) if b.n_po else None
        sp = (b.sp_subj.numpy(), b.sp_rel.numpy()) if b.n_sp else None
        out = ko.step_forward_backward(ko.COMPLEX, E, R, po, sp, cand, y, normalizer=cb.normalizer_loss)
        rel = abs(out["loss"] / cb.normalizer_loss - z["losses"][step]) / z["losses"][step]
        worst = max(worst, rel)
        assert rel <= 5e-5, (step, rel)
        ko.adagrad_step(E, out["dE"], sumE, float(z["lr"]))
        ko.adagrad_step(R, out["dR"], sumR, float(z["lr"]))
    print(f"[oracle g15] worst relative loss deviation over the first 12 steps: {worst:.2e}")


@pytest.mark.gpu
def test_hip_three_training_passes_and_validation_mrr(okge_lib, tmp_path):
    from open_knowledge_graph_embeddings_amd.evaluate import FusedEvaluator
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    z = golden("g15_epochs_fb15k237")
    E0, R0 = _tables(z)
    E, R = torch.from_numpy(E0).cuda(), torch.from_numpy(R0).cuda()
    train, valid = _producers(tmp_path, z, "cuda:0")
    ts = FusedTrainStep(E, R, "complex", loss="bce", lr=float(z["lr"]))
    n_epochs, steps_per_pass = int(z["n_epochs"]), len(train)
    step, dev = 0, []
    for epoch in range(n_epochs):
        for cb in train:
            np.testing.assert_array_equal([cb.batch.n_po, cb.batch.n_sp, cb.batch.nnz, cb.normalizer_loss], z["batch_shape"][step])
            loss = float(ts.step(cb.batch, normalizer=cb.normalizer_loss)[0]) / cb.normalizer_loss     # the summed loss / normalizer, as the reference logs it
            dev.append(abs(loss - z["losses"][step]) / z["losses"][step])
            step += 1
    assert step == len(z["losses"]) == 132
    dev = np.asarray(dev)
    print(f"[hip g15] relative loss deviation: pass 1 max {dev[:steps_per_pass].max():.2e}, pass 2 max "
          f"{dev[steps_per_pass:2 * steps_per_pass].max():.2e}, pass 3 max {dev[2 * steps_per_pass:].max():.2e}; "
          f"loss {z['losses'][0]:.4f} -> {z['losses'][-1]:.2e}")
    # the whole curve: within 5e-5 relative while the trajectories are close -- the first pass (observed 2.1e-5; G11 pins 30
    # steps at 8e-6); passes two and three are 88 / 132 Adagrad steps of another summation order away and the loss itself has
    # fallen by three orders of magnitude: observed 4.2e-5 / 6.2e-5, bound 2e-4
    assert dev[:steps_per_pass].max() <= 5e-5
    assert dev.max() <= 2e-4
    # trained tables: stored sample rows, and the checksums of the whole tables
    Ef, Rf = E.cpu().numpy(), R.cpu().numpy()
    np.testing.assert_allclose(Ef[[2, 100, 5000, 14000]], z["E_rows"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(Rf[[2, 50, 200]], z["R_rows"], rtol=0, atol=5e-3)
    assert abs(np.abs(Ef).sum(dtype=np.float64) - z["final_check"][1]) <= 1e-3 * z["final_check"][1]
    assert abs(np.abs(Rf).sum(dtype=np.float64) - z["final_check"][3]) <= 1e-3 * z["final_check"][3]
    # the validation pass on the tables THIS run trained
    res, n_groups = FusedEvaluator(E, R, "complex").run(iter(valid))
    assert n_groups == int(z["c_mrr"])
    print(f"[hip g15] valid MRR {res['mrr'].avg:.6f} (reference {float(z['m_mrr']):.6f}), h10 {res['h10'].avg:.5f} ({float(z['m_h10']):.5f})")
    assert abs(res["mrr"].avg - float(z["m_mrr"])) <= 1e-3
    for k in ("h1", "h3", "h10", "h50"):
        assert abs(res[k].avg - float(z["m_" + k])) <= 5e-3, k
    assert abs(res["mr"].avg - float(z["m_mr"])) <= 2e-2 * float(z["m_mr"])
