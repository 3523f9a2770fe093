"""One training step of the hot path, as `Trainer.compute_one_batch(training=True)` drives it in the
reference (openkge/trainer.py:181-257): forward + loss + backward + dense Adagrad on both tables.

All arithmetic is in libokge_hip.so; this class only owns the buffers (tables, dense gradients, Adagrad
accumulators) and sequences the C-ABI calls on the current stream.  No host synchronisation per step.
"""
from __future__ import annotations

import torch

from . import hotpath as H


class FusedTrainStep:
    def __init__(self, E: torch.Tensor, R: torch.Tensor, scorer: str, loss: str = "bce", lr: float = 0.3,
                 weight_decay: float = 1e-10, eps: float = 1e-8, label_smoothing: float = 0.0,
                 input_dropout: float = 0.0, relation_input_dropout: float = 0.0, seed: int = 0, engine=None):
        """Defaults follow config/fb15k237/fb15k237-complex-kge.yaml and the optimizer OptimRegime actually
        builds (utils/optim.py:29,139-160): Adagrad(lr, weight_decay=1e-10, eps=1e-8 leaked from Adam)."""
        self.E, self.R = E, R
        self.scorer, self.loss = scorer, loss
        self.lr, self.weight_decay, self.eps = lr, weight_decay, eps
        self.label_smoothing = label_smoothing
        self.input_dropout, self.relation_input_dropout = input_dropout, relation_input_dropout
        self.seed = seed
        self.engine = engine or H.HotPath(E.device)
        self.dE, self.dR = torch.zeros_like(E), torch.zeros_like(R)       # dense .grad (model_config.sparse False)
        self.sumE, self.sumR = torch.zeros_like(E), torch.zeros_like(R)   # Adagrad state 'sum' (init 0)
        self.steps = 0
        self._grads_zero = True           # fresh buffers; kept true by the zero_grad fused into Adagrad
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=E.device)

    def _set_dropout(self, batch: H.PrefixBatch, training=True):
        pe = self.input_dropout if training else 0.0
        pr = self.relation_input_dropout if training else 0.0
        s, t = self.seed, self.steps
        batch.drop_cand = H.DropoutSpec(pe, s, H.STREAM_CAND, t)
        batch.drop_po_ent = H.DropoutSpec(pe, s, H.STREAM_PO_ENT, t)
        batch.drop_sp_ent = H.DropoutSpec(pe, s, H.STREAM_SP_ENT, t)
        batch.drop_po_rel = H.DropoutSpec(pr, s, H.STREAM_PO_REL, t)
        batch.drop_sp_rel = H.DropoutSpec(pr, s, H.STREAM_SP_REL, t)

    def forward_backward(self, batch: H.PrefixBatch, normalizer=None):
        """trainer.py:206-234.  Returns the summed loss (device double[1], valid after stream sync)."""
        self._set_dropout(batch)
        return self.engine.forward_backward(self.E, self.R, self.scorer, batch, self.dE, self.dR, loss=self.loss,
                                            label_smoothing=self.label_smoothing, normalizer=normalizer,
                                            loss_out=self.loss_out, grads_zero=self._grads_zero)

    def optimizer_step(self):
        """trainer.py:240-244: optimizer.step() then zero_grad() -- one sweep per table."""
        self.engine.adagrad2(self.E, self.dE, self.sumE, self.R, self.dR, self.sumR, self.lr, self.weight_decay,
                             self.eps, zero_grad=True)

    def step(self, batch: H.PrefixBatch, normalizer=None):
        self.steps += 1
        loss = self.forward_backward(batch, normalizer)
        self._grads_zero = False
        self.optimizer_step()
        self._grads_zero = True
        return loss
