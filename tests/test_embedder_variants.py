"""Embedder variants of the lookup models (openkge/model.py:463-479: batch_norm, project_entity, normalize='norm', l2_reg
hook): the encode falls through to torch (the reference's op sequence), the HIP scorer / loss / backward runs on the
encoded rows (trainer._VirtualTablesLossFn).  Against G12 = the reference's LookupComplexRelationModel + AddLossModule:
loss, hook loss, outputs, the gradient of EVERY parameter of (loss + hook) / normalizer, running statistics, eval scores."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_names


def _model(z, cuda):
    from open_knowledge_graph_embeddings_amd.dataset import EntityRelationDatasetMeta
    from open_knowledge_graph_embeddings_amd.model import Models
    kw = {}
    if bool(z["batch_norm"]):
        kw["batch_norm"] = True
    if bool(z["project_entity"]):
        kw["project_entity"] = True
    if str(z["normalize"]):
        kw["normalize"] = str(z["normalize"])
    if float(z["l2_reg"]) > 0:
        kw["l2_reg"] = float(z["l2_reg"])
    n_ent, d = z["p0_entity_embedding.weight"].shape
    m = Models.LookupComplexRelationModel(entity_slot_size=d, input_dropout=0.0, init_std=0.3, sparse=False,
                                          train_data=EntityRelationDatasetMeta(entities_size=n_ent, relations_size=z["p0_relation_embedding.weight"].shape[0]),
                                          **kw)
    ref_keys = sorted(k[3:] for k in z.files if k.startswith("p0_"))
    assert sorted(m.state_dict().keys()) == ref_keys                      # the reference's state-dict layout, key for key
    m.load_state_dict({k: torch.from_numpy(z["p0_" + k]) for k in ref_keys})
    return m.cuda() if cuda else m


@pytest.mark.parametrize("name", golden_names("g12_variant_"))
def test_state_dict_layout_matches_reference(name):
    m = _model(golden(name), cuda=False)
    assert m.encode_in_torch


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names("g12_variant_"))
def test_variant_forward_backward_matches_reference(okge_lib, name):
    from open_knowledge_graph_embeddings_amd.trainer import AddLossModule
    z = golden(name)
    m = _model(z, cuda=True)
    m.train()
    mod = AddLossModule(m, torch.nn.BCEWithLogitsLoss(reduction="sum"), 0.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    inputs = [(t(z["po_rel"]), t(z["po_obj"])), (t(z["sp_subj"]), t(z["sp_rel"]))]
    # *_noshare: batch_shared_entities=None -- every prefix scorer encodes its own candidate block (trainer.py:86-87)
    no_shared = "no_shared" in z.files and bool(z["no_shared"])
    shared = None if no_shared else t(z["cand"])
    loss, hook, outs = mod(inputs=inputs, labels=t(z["labels"]), use_batch_shared_entities=False, batch_shared_entities=shared,
                           epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
    np.testing.assert_allclose(outs.cpu().numpy(), z["outputs"], rtol=0, atol=1e-4)
    assert abs(float(loss.detach()) - float(z["loss"])) <= 3e-5 * abs(float(z["loss"]))
    assert (hook is not None) == bool(z["has_hook"])
    backward_loss = loss.sum()
    if hook is not None:
        assert abs(float(hook.detach()) - float(z["hook"])) <= 1e-5 * abs(float(z["hook"]))
        backward_loss = backward_loss + hook
    (backward_loss / float(z["normalizer"])).backward()                   # trainer.py:217-222
    for k, p in m.named_parameters():
        ref = z["g_" + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-5 * np.abs(ref).max() + 1e-10, err_msg=k)
    for k in z.files:
        if k.startswith("after_"):
            np.testing.assert_allclose(m.state_dict()[k[6:]].cpu().numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
    m.eval()
    with torch.no_grad():
        ev = torch.cat([m.po_prefix_score(*inputs[0]), m.sp_prefix_score(*inputs[1])], 0)
        l2, h2, o2 = mod(inputs=inputs, labels=t(z["labels"]), use_batch_shared_entities=False, batch_shared_entities=shared,
                         epoch=1, input_style_triple_or_prefix="right_and_left_prefix")
    np.testing.assert_allclose(ev.cpu().numpy(), z["eval_outputs"], rtol=0, atol=1e-4)
    # AddLossModule in eval mode scores BOTH directions against get_all_obj() (trainer.py:77-78); with an entity
    # projection the po rows therefore differ from po_prefix_score's (get_all_subj): compare what must agree
    n_po = z["po_rel"].shape[0]
    np.testing.assert_allclose(o2[n_po:].cpu().numpy(), z["eval_outputs"][n_po:], rtol=0, atol=1e-4)
    if no_shared or not bool(z["project_entity"]):
        np.testing.assert_allclose(o2[:n_po].cpu().numpy(), z["eval_outputs"][:n_po], rtol=0, atol=1e-4)
    assert h2 is None and torch.isfinite(l2)
