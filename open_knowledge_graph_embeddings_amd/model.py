"""Plugin surface of the hot path: the reference's RelationScorer x RelationEmbedder composition and the
``Models`` registry (openkge/model.py:31-139, :176-278, :353-558, :1006-1066), backed by the HIP kernels.

Same class names, constructor keywords, method names and tensor conventions as the reference:
ids are int32 ``(b, 1)`` (or ``(N,)`` for candidates), scores come back ``(b, N)`` fp32, state-dict keys are
``entity_embedding.weight`` and ``relation_embedding.weight``.  What differs:

* every forward runs in libokge_hip.so (no ATen math).  Training gradients are produced by ``trainer.AddLossModule``
  (fused forward + loss + backward), which is how the reference's Trainer consumes the model
  (openkge/trainer.py:142,206-234).  Called directly WITH gradients enabled (a user's own loss on ``sp_prefix_score`` /
  ``po_prefix_score`` / ``forward`` / ``_score`` / ``encode_*``) the methods return tensors with an autograd graph
  (``autograd_score``: HIP forward, GEMM + chain-rule backward); under ``torch.no_grad()`` they take the fused id path;
* the fused path (gather + dropout inside the tile kernels) covers the lookup embedder with batch_norm / projection /
  normalize / l2_reg off -- every BASELINE config.  With one of those `_encode` variants on (model.py:463-479) the
  embedder FALLS THROUGH to torch for the encode (the reference's own op sequence, differentiable) and the encoded rows
  go through the HIP scorer / loss / backward as two small virtual tables (trainer.AddLossModule); sparse gradients and
  the relation projection of the non-"Simple" embedder raise NotImplementedError at construction;
* the two dropouts the reference applies in sequence (input_dropout, dropout; model.py:461-470) draw from a
  counter-based Philox stream instead of torch's global generator.
"""
from __future__ import annotations

import torch

from . import autograd_score as AG
from . import hotpath as H
from ._native import OkgeError


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)

PAD = 0   # openkge/index_mapper.py:14  (ids 0, 1 are reserved; real ids start at 2)


class RelationModel(torch.nn.Module):
    """openkge/model.py:14-28"""
    is_cuda = False

    def cuda(self, device=None):
        super().cuda(device=device)
        self.is_cuda = True
        return self

    def cpu(self):
        raise OkgeError("this model runs on an MI355X; there is no CPU path")


class RelationScorer(RelationModel):
    """openkge/model.py:31-77"""
    scorer_name = None

    def forward(self, subj, rel, obj, **kwargs):
        return self.triple_score(self.encode_subj(subj), self.encode_rel(rel), self.encode_obj(obj), **kwargs)

    def triple_score(self, subj, rel, obj, **kwargs):
        """(b, 1) scores of encoded triples, Hadamard form (model.py:178-179, :231-238, :276); with gradients enabled and
        rows that carry a graph, the same form in differentiable torch ops"""
        if _wants_grad(subj, rel, obj):
            return AG.triple_score(self.scorer_name, subj, rel, obj)
        return self.engine().score_triples(self.scorer_name, subj, rel, obj)

    def sp_prefix_score(self, subj=None, rel=None, many_obj=None):
        """scores (b, N) of (subj, rel, ?) against all objects, or against pre-encoded rows `many_obj`."""
        return self._prefix_score(H.PrefixBatch(sp_subj=subj, sp_rel=rel), many_obj)

    def po_prefix_score(self, rel=None, obj=None, many_subj=None):
        """scores (b, N) of (?, rel, obj) against all subjects, or against pre-encoded rows `many_subj`."""
        return self._prefix_score(H.PrefixBatch(po_rel=rel, po_obj=obj), many_subj)

    def precompute_batch_shared_inputs(self, entity_ids):
        return self.encode_obj(entity_ids)

    def _score(self, subj, rel, obj, prefix=False, sp=None, po=None):
        """openkge/model.py:181-240 / :248-278 on ALREADY ENCODED rows: (b,d),(b,d),(N,d) -> (b,N)."""
        if not prefix:
            return self.triple_score(subj, rel, obj)
        eng = self.engine()
        if not sp and not po:
            raise Exception      # model.py:217-218, :273-274
        if _wants_grad(subj, rel, obj):     # a graph for the caller's own loss (AddLossModule never comes through here)
            flat = lambda t: t.reshape(-1, t.shape[-1])      # noqa: E731
            ent, cand = (flat(subj), flat(obj)) if sp else (flat(obj), flat(subj))
            return AG.PrefixScoreFn.apply(ent, flat(rel), cand, eng, self.scorer_name, bool(sp))
        subj, rel, obj = subj.detach(), rel.detach(), obj.detach()
        rel = rel.reshape(-1, rel.shape[-1]).contiguous()
        b = rel.shape[0]
        ar = torch.arange(b, dtype=torch.int32, device=rel.device)
        if sp:
            ent, cand = subj.reshape(-1, subj.shape[-1]).contiguous(), obj.reshape(-1, obj.shape[-1]).contiguous()
            batch = H.PrefixBatch(sp_subj=ar, sp_rel=ar)
        elif po:
            ent, cand = obj.reshape(-1, obj.shape[-1]).contiguous(), subj.reshape(-1, subj.shape[-1]).contiguous()
            batch = H.PrefixBatch(po_rel=ar, po_obj=ar)
        else:
            raise Exception
        batch.cand_table, batch.cand_first, batch.n_cand = cand, 0, cand.shape[0]
        return eng.score(ent, rel, self.scorer_name, batch)


class ComplexRelationScorer(RelationScorer):
    scorer_name = "complex"


class DistmultRelationScorer(RelationScorer):
    scorer_name = "distmult"


class RelationEmbedder(RelationModel):
    """openkge/model.py:80-139 (abstract protocol)"""

    def precompute_embeddings_from_tokens(self):
        raise NotImplementedError


class LookupBaseRelationEmbedder(RelationEmbedder):
    """openkge/model.py:353-542, restricted to what the fused path implements."""

    def __init__(self, entity_slot_size, relation_slot_size, train_data, entity_embedding_size=None,
                 relation_embedding_size=None, normalize='', dropout=0.0, input_dropout=0.0, relation_dropout=0.0,
                 relation_input_dropout=0.0, project_entity=False, project_entity_activation='ReLU',
                 project_relation=True, project_relation_activation=None, sparse=False, init_std=0.01,
                 batch_norm=False, l2_reg=0, seed=0):
        super().__init__()
        if sparse or project_relation:
            raise NotImplementedError("outside the lookup path: sparse gradients / relation projection (openkge/model.py:391-409)")
        if relation_slot_size is None or relation_slot_size <= 0:
            relation_slot_size = entity_slot_size
        e_size = entity_slot_size if entity_embedding_size is None else entity_embedding_size
        r_size = relation_slot_size if relation_embedding_size is None else relation_embedding_size
        if r_size != entity_slot_size or (e_size != entity_slot_size and not project_entity):
            raise NotImplementedError("embedding sizes other than the slot size need the matching projection")
        self.train_data = train_data
        self.slot_size = entity_slot_size
        # module construction order = the reference's (model.py:389-440): identical parameters from identical seeds
        self.entity_embedding = torch.nn.Embedding(train_data.entities_size, e_size, padding_idx=PAD)
        self.relation_embedding = torch.nn.Embedding(train_data.relations_size, r_size, padding_idx=PAD)
        self.project_entity, self.project_relation = bool(project_entity), False
        if project_entity:
            act = lambda: [getattr(torch.nn, project_entity_activation)()] if project_entity_activation else []   # noqa: E731
            subj_layer = torch.nn.Linear(entity_slot_size, entity_slot_size, bias=False)
            obj_layer = torch.nn.Linear(entity_slot_size, entity_slot_size, bias=False)
            torch.nn.init.xavier_normal_(subj_layer.weight.data)
            torch.nn.init.xavier_normal_(obj_layer.weight.data)
            self.subj_projection = torch.nn.Sequential(subj_layer, *act())
            self.obj_projection = torch.nn.Sequential(obj_layer, *act())
        torch.nn.init.normal_(self.entity_embedding.weight.data, std=init_std)       # model.py:429-430
        torch.nn.init.normal_(self.relation_embedding.weight.data, std=init_std)
        self.dropout = dropout
        self.input_dropout = input_dropout
        self.relation_dropout = dropout if relation_dropout is None else relation_dropout          # model.py:434-435
        self.relation_input_dropout = input_dropout if relation_input_dropout is None else relation_input_dropout
        self.batch_norm = bool(batch_norm)
        if self.batch_norm:
            self.bn_e = torch.nn.BatchNorm1d(e_size)
            self.bn_r = torch.nn.BatchNorm1d(r_size)
        self.normalize, self.l2_reg = normalize or '', l2_reg
        self._l2_reg_hook = None
        # any _encode variant on: the embedder falls through to torch (module docstring)
        self.encode_in_torch = bool(self.batch_norm or self.project_entity or self.normalize or self.l2_reg)
        if self.normalize not in ('', 'norm'):
            raise NotImplementedError(f"normalize={normalize!r}")
        self.dropout_seed = seed
        self.dropout_step = 0          # advanced once per training batch by AddLossModule
        self._engine = None

    # -- plumbing ------------------------------------------------------------------------------------
    def engine(self) -> H.HotPath:
        dev = self.entity_embedding.weight.device
        if self._engine is None or self._engine.device != dev:
            self._engine = H.HotPath(dev)
        return self._engine

    @property
    def E(self):
        return self.entity_embedding.weight.data

    @property
    def R(self):
        return self.relation_embedding.weight.data

    def keep_prob_dropout(self, p_in, p_out):
        """Two dropouts in sequence (model.py:461-462, :469-470) == one with keep prob (1-p_in)(1-p_out)."""
        if not self.training:
            return 0.0
        return 1.0 - (1.0 - p_in) * (1.0 - p_out)

    def dropout_spec(self, stream, relation=False):
        p = self.keep_prob_dropout(self.relation_input_dropout, self.relation_dropout) if relation else \
            self.keep_prob_dropout(self.input_dropout, self.dropout)
        return H.DropoutSpec(p, self.dropout_seed, stream, self.dropout_step)

    def after_batch_loss_hook(self, epoch):
        """model.py:447-453"""
        if self.training and self.l2_reg > 0:
            result, self._l2_reg_hook = self._l2_reg_hook, None
            return result
        return None

    # -- RelationEmbedder protocol ------------------------------------------------------------------
    def _encode_torch(self, slot_item, embedding, project, input_dropout, dropout, batch_norm, lookup=True):
        """model.py:455-480, op for op (ATen on the GPU, differentiable): the fall-through for embedder variants"""
        F = torch.nn.functional
        if lookup:
            rows = embedding(slot_item.reshape(-1).long())
        else:
            rows = slot_item
        if input_dropout > 0:
            rows = F.dropout(rows, p=input_dropout, training=self.training)
        if self.batch_norm:
            rows = batch_norm(rows)
        if project is not None:
            rows = project(rows)
        if self.normalize == 'norm':
            rows = F.normalize(rows)
        if dropout > 0:
            rows = F.dropout(rows, p=dropout, training=self.training)
        if self.training and self.l2_reg > 0:
            hook = rows
            if self.dropout > 0:
                hook = hook / self.dropout
            hook = self.l2_reg * hook.abs().pow(3).sum()
            self._l2_reg_hook = hook if self._l2_reg_hook is None else self._l2_reg_hook + hook
        return rows

    def _encode(self, table, slot_item, stream, relation, lookup=True, which=None):
        if self.encode_in_torch:
            if relation:
                return self._encode_torch(slot_item, self.relation_embedding, None, self.relation_input_dropout,
                                          self.relation_dropout, self.bn_r if self.batch_norm else None, lookup)
            proj = (self.subj_projection if which == "subj" else self.obj_projection) if self.project_entity else None
            return self._encode_torch(slot_item, self.entity_embedding, proj, self.input_dropout, self.dropout,
                                      self.bn_e if self.batch_norm else None, lookup)
        eng = self.engine()
        weight = (self.relation_embedding if relation else self.entity_embedding).weight
        self._fresh_mask_for_direct_call(slot_item if not lookup else weight)
        if not lookup:                    # model.py:459-460: already rows, only dropout applies
            if _wants_grad(slot_item):
                return AG.MaskRowsFn.apply(slot_item, eng, self.dropout_spec(stream, relation))
            rows = slot_item.contiguous()
            return eng.encode_rows(rows, None, 0, rows.shape[0], self.dropout_spec(stream, relation))
        if _wants_grad(weight):
            ids = slot_item.reshape(-1)
            return AG.EncodeRowsFn.apply(weight, ids, 0, int(ids.numel()), eng, self.dropout_spec(stream, relation))
        return eng.encode_rows(table, slot_item.reshape(-1), drop=self.dropout_spec(stream, relation))

    def _fresh_mask_for_direct_call(self, source):
        """encode_* / get_all_* called DIRECTLY in training mode with gradients enabled (a caller's own loss built from
        encoded rows): every such call draws a fresh Philox mask, as each of the reference's _encode calls draws a fresh
        Bernoulli mask (model.py:455-470).  Inside _prefix_score / AddLossModule the step counter is advanced once per
        call there (the five encode calls of a step are told apart by their stream ids)."""
        if self.training and not getattr(self, "_in_prefix_score", False) and _wants_grad(source):
            self.dropout_step += 1

    def encode_subj(self, subj, lookup=True):
        return self._encode(self.E, subj, H.STREAM_SP_ENT, False, lookup, which="subj")

    def encode_obj(self, obj, lookup=True):
        return self._encode(self.E, obj, H.STREAM_PO_ENT, False, lookup, which="obj")

    def encode_rel(self, rel, lookup=True):
        return self._encode(self.R, rel, H.STREAM_SP_REL, True, lookup)

    def _get_all(self, table, min_size, stream, relation, which=None):
        if self.encode_in_torch:           # model.py:512-514: encode(weight[min_size:], lookup=False)
            w = (self.relation_embedding if relation else self.entity_embedding).weight[min_size:].contiguous()
            return self._encode(None, w, stream, relation, lookup=False, which=which)
        weight = (self.relation_embedding if relation else self.entity_embedding).weight
        self._fresh_mask_for_direct_call(weight)
        if _wants_grad(weight):
            return AG.EncodeRowsFn.apply(weight, None, min_size, table.shape[0] - min_size, self.engine(),
                                         self.dropout_spec(stream, relation))
        return self.engine().encode_rows(table, None, min_size, table.shape[0] - min_size,
                                         self.dropout_spec(stream, relation))

    def get_all_subj(self):
        return self._get_all(self.E, self.train_data.min_entities_size, H.STREAM_CAND, False, "subj")

    def get_all_obj(self):
        return self._get_all(self.E, self.train_data.min_entities_size, H.STREAM_CAND, False, "obj")

    def get_all_rel(self):
        return self._get_all(self.R, self.train_data.min_relations_size, H.STREAM_SP_REL, True)

    def _get(self, encode, id):
        return encode(torch.tensor([id], dtype=torch.int32, device=self.E.device))

    def get_subj(self, subj):
        return self._get(self.encode_subj, subj)

    def get_rel(self, rel):
        return self._get(self.encode_rel, rel)

    def get_obj(self, obj):
        return self._get(self.encode_obj, obj)

    def get_slot_size(self):
        return self.slot_size

    # -- fused prefix scoring (ids in, scores out) -----------------------------------------------------
    def _prefix_score(self, batch: H.PrefixBatch, many=None):
        eng = self.engine()
        with_graph = not self.encode_in_torch and _wants_grad(self.entity_embedding.weight, self.relation_embedding.weight, many)
        if with_graph and self.training:
            self.dropout_step += 1         # fresh masks per call, as each of the reference's encode calls draws them
        if self.encode_in_torch or with_graph:   # torch encode of the variants / a graph for the caller's loss: the
            self._in_prefix_score = True         # reference's own call order (model.py:52-74), HIP scorer
            try:
                if batch.sp_subj is not None:
                    subj, rel = self.encode_subj(batch.sp_subj), self.encode_rel(batch.sp_rel)
                    return self._score(subj, rel, self.get_all_obj() if many is None else many, prefix=True, sp=True, po=False)
                if many is None:
                    many = self.get_all_subj()
                rel, obj = self.encode_rel(batch.po_rel), self.encode_obj(batch.po_obj)
                return self._score(many, rel, obj, prefix=True, sp=False, po=True)
            finally:
                self._in_prefix_score = False
        if many is None:                   # all entities with id >= min_entities_size (model.py:512-523)
            batch.cand_first = self.train_data.min_entities_size
            batch.n_cand = self.E.shape[0] - batch.cand_first
            batch.drop_cand = self.dropout_spec(H.STREAM_CAND)
        else:                              # rows already encoded by precompute_batch_shared_inputs / get_all_obj
            many = many.reshape(-1, many.shape[-1]).contiguous()
            batch.cand_table, batch.cand_first, batch.n_cand = many, 0, many.shape[0]
        batch.drop_po_ent, batch.drop_sp_ent = self.dropout_spec(H.STREAM_PO_ENT), self.dropout_spec(H.STREAM_SP_ENT)
        batch.drop_po_rel = self.dropout_spec(H.STREAM_PO_REL, True)
        batch.drop_sp_rel = self.dropout_spec(H.STREAM_SP_REL, True)
        return eng.score(self.E, self.R, self.scorer_name, batch)


class LookupSimpleRelationEmbedder(LookupBaseRelationEmbedder):
    """openkge/model.py:545-558: relation slot = entity slot, no relation projection."""

    def __init__(self, entity_slot_size, **kwargs):
        kwargs.pop('relation_slot_size', None)
        kwargs.pop('project_relation', None)
        super().__init__(entity_slot_size=entity_slot_size, relation_slot_size=entity_slot_size,
                         project_relation=False, **kwargs)
        self.relation_projection = None


class LookupComplexRelationModel(ComplexRelationScorer, LookupSimpleRelationEmbedder):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)


class LookupDistmultRelationModel(DistmultRelationScorer, LookupSimpleRelationEmbedder):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)


class Models:
    """openkge/model.py:1052-1066: looked up with getattr(Models, args['model']) (scripts/train.py:88).
    The lookup models are registered here, the token-pooled ones by token_pooled.py (imported by the package);
    the others stay with the reference."""
    LookupDistmultRelationModel = LookupDistmultRelationModel
    LookupComplexRelationModel = LookupComplexRelationModel
