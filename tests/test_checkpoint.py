"""Checkpoint interop against a checkpoint written by the reference's own classes
(tests/golden/g8_checkpoint*.pt: model.state_dict() + OptimRegime.state_dict() after two Adagrad steps)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR, golden


class HostStep:
    """the attributes of a train step that the checkpoint code touches (no GPU needed for the layout tests)"""
    def __init__(self, n_ent=90, n_rel=11, d=24):
        self.E, self.R = torch.zeros(n_ent, d), torch.zeros(n_rel, d)
        self.sumE, self.sumR, self.dE, self.dR = (torch.zeros_like(t) for t in (self.E, self.R, self.E, self.R))
        self.steps, self.lr, self.weight_decay, self.eps = 0, 0.1, 0.0, 1e-10


def structure(x):
    if isinstance(x, dict):
        return {k: structure(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [structure(v) for v in x]
    if torch.is_tensor(x):
        return (tuple(x.shape), str(x.dtype))
    return x


@pytest.mark.both
def test_roundtrip_layout_equals_reference_checkpoint():
    from open_knowledge_graph_embeddings_amd.checkpoint import load_reference_checkpoint, to_reference_checkpoint
    path = os.path.join(GOLDEN_DIR, "g8_checkpoint_after.pt")
    ref = torch.load(path, map_location="cpu", weights_only=True)
    st = HostStep()
    load_reference_checkpoint(st, path)
    assert st.steps == 3 and (st.lr, st.weight_decay, st.eps) == (0.3, 1e-10, 1e-8)
    mine = to_reference_checkpoint(st, epoch=1)
    assert structure(mine) == structure(ref)                       # same keys, shapes, dtypes, scalars
    for k in ("entity_embedding.weight", "relation_embedding.weight"):
        assert torch.equal(mine["state_dict"][k], ref["state_dict"][k])
    for i in (0, 1):
        a, b = (c["optimizer_state_dict"][0]["optimizer_state"]["state"][i] for c in (mine, ref))
        assert torch.equal(a["sum"], b["sum"]) and float(a["step"]) == float(b["step"])
    st2 = HostStep()
    load_reference_checkpoint(st2, mine, reset_optimizer=True)     # Trainer.load(reset_optimizer=True)
    assert st2.steps == 0 and float(st2.sumE.abs().sum()) == 0 and torch.equal(st2.E, st.E)
    with pytest.raises(ValueError):
        load_reference_checkpoint(HostStep(n_ent=91), mine)


@pytest.mark.both
def test_reference_can_load_our_checkpoint_structure(tmp_path):
    """torch.optim.Adagrad.load_state_dict accepts the optimizer_state we write (it is what OptimRegime hands it,
    utils/optim.py:193) and model.load_state_dict-style key lookup finds both tables."""
    from open_knowledge_graph_embeddings_amd.checkpoint import load_reference_checkpoint, save_checkpoint
    st = HostStep()
    load_reference_checkpoint(st, os.path.join(GOLDEN_DIR, "g8_checkpoint_after.pt"))
    path = str(tmp_path / "ckpt.pth.tar")
    save_checkpoint(path, st, epoch=1)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    p = [torch.nn.Parameter(ck["state_dict"]["entity_embedding.weight"].clone()),
         torch.nn.Parameter(ck["state_dict"]["relation_embedding.weight"].clone())]
    opt = torch.optim.Adagrad(p, lr=0.0)
    opt.load_state_dict(ck["optimizer_state_dict"][0]["optimizer_state"])
    assert torch.equal(opt.state[p[0]]["sum"], st.sumE) and opt.param_groups[0]["lr"] == 0.3


@pytest.mark.gpu
def test_resume_from_reference_checkpoint_matches_reference_step(okge_lib):
    """load the reference's checkpoint after step 2, run step 3 on the GPU, compare with the reference's step 3"""
    from open_knowledge_graph_embeddings_amd.checkpoint import load_reference_checkpoint
    from open_knowledge_graph_embeddings_amd.hotpath import PrefixBatch, positives_from_dense
    from open_knowledge_graph_embeddings_amd.train_step import FusedTrainStep
    z = golden("g8_checkpoint_step3")
    st = FusedTrainStep(torch.zeros(90, 24, device="cuda:0"), torch.zeros(11, 24, device="cuda:0"), "complex", lr=0.0)
    load_reference_checkpoint(st, os.path.join(GOLDEN_DIR, "g8_checkpoint.pt"))
    assert st.steps == 2 and st.lr == 0.3
    dev = lambda a: torch.from_numpy(a.reshape(-1)).cuda()       # noqa: E731
    prow, pcol = positives_from_dense(torch.from_numpy(z["labels"]).cuda())
    loss = st.step(PrefixBatch(po_rel=dev(z["po_rel"]), po_obj=dev(z["po_obj"]), sp_subj=dev(z["sp_subj"]),
                               sp_rel=dev(z["sp_rel"]), pos_row=prow, pos_col=pcol, cand_first=2, n_cand=88))
    assert abs(float(loss[0]) - float(z["loss"])) <= 3e-5 * abs(float(z["loss"]))
    np.testing.assert_allclose(st.sumE.cpu().numpy(), z["sumE"], rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(st.sumR.cpu().numpy(), z["sumR"], rtol=1e-4, atol=1e-12)
    # third step: the accumulators are well conditioned by now
    np.testing.assert_allclose(st.E.cpu().numpy(), z["E"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(st.R.cpu().numpy(), z["R"], rtol=2e-4, atol=2e-5)
