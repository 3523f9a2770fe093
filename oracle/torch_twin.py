"""torch-CPU twin of the reference's training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Same ATen op sequence as the reference on CPU (so it has the reference's performance character and
is what ``bench.py``'s ``cpu_baseline`` leg times, ``kind: "port"``):

  index_select gathers + F.dropout          openkge/model.py:455-480
  6 chunk().contiguous() + 8 mul + 4 mm per direction (ComplEx) / 1 mm (DistMult)
                                            openkge/model.py:200-216, :269-272
  cat, BCEWithLogitsLoss(sum) | log_softmax + KLDivLoss(sum), label smoothing
                                            openkge/trainer.py:91-106
  (loss / (B*N)).backward()                 openkge/trainer.py:217-234
  dense torch.optim.Adagrad(lr, weight_decay, eps=1e-8)   utils/optim.py:29,139-160

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: PINNED against the same golden vectors as oracle/kge_oracle.py
(tests/test_oracle_golden.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class TwinModel(torch.nn.Module):
    def __init__(self, kind, n_ent, n_rel, d, input_dropout=0.0, init_std=0.1):
        super().__init__()
        self.kind = kind
        self.entity_embedding = torch.nn.Embedding(n_ent, d, padding_idx=0)
        self.relation_embedding = torch.nn.Embedding(n_rel, d, padding_idx=0)
        torch.nn.init.normal_(self.entity_embedding.weight.data, std=init_std)
        torch.nn.init.normal_(self.relation_embedding.weight.data, std=init_std)
        self.input_dropout = input_dropout

    def enc_ent(self, ids):
        x = self.entity_embedding(ids.squeeze().long())
        if self.input_dropout > 0:
            x = F.dropout(x, p=self.input_dropout, training=self.training)
        return x

    def enc_rel(self, ids):
        return self.relation_embedding(ids.squeeze().long())

    def score(self, a, r, cand, sp):
        a = a.view(-1, a.size(-1))
        r = r.view(-1, r.size(-1))
        if self.kind == "distmult":
            return (a * r).mm(cand.transpose(0, 1))
        r1, r2 = (t.contiguous() for t in r.chunk(2, dim=1))
        a1, a2 = (t.contiguous() for t in a.chunk(2, dim=1))
        c1, c2 = (t.contiguous() for t in cand.chunk(2, dim=1))
        if sp:
            return (a1 * r1).mm(c1.t()) + (a2 * r1).mm(c2.t()) + (a1 * r2).mm(c2.t()) - (a2 * r2).mm(c1.t())
        return (a1 * r1).mm(c1.t()) + (a2 * r1).mm(c2.t()) + (a2 * r2).mm(c1.t()) - (a1 * r2).mm(c2.t())

    def forward_loss(self, po, sp, cand_ids, labels, loss_kind="bce", smoothing=0.0):
        cand = self.enc_ent(cand_ids.view(-1))
        outs = []
        if po is not None:
            outs.append(self.score(self.enc_ent(po[1]), self.enc_rel(po[0]), cand, sp=False))
        if sp is not None:
            outs.append(self.score(self.enc_ent(sp[0]), self.enc_rel(sp[1]), cand, sp=True))
        x = torch.cat(outs)
        if loss_kind == "kl":
            loss = F.kl_div(F.log_softmax(x, dim=1).view(-1), labels.view(-1), reduction="sum")
        else:
            if smoothing > 0:
                labels = (labels + 1.0 / labels.size(-1)) * (1.0 - smoothing)
            loss = F.binary_cross_entropy_with_logits(x.view(-1), labels.view(-1), reduction="sum")
        return loss, x


def make_adagrad(model, lr, weight_decay=1e-10, eps=1e-8):
    return torch.optim.Adagrad(model.parameters(), lr=lr, weight_decay=weight_decay, eps=eps,
                               lr_decay=0, initial_accumulator_value=0)


def train_step(model, opt, po, sp, cand_ids, labels, loss_kind="bce", smoothing=0.0):
    """One step exactly as Trainer.compute_one_batch drives it (trainer.py:217-246)."""
    opt.zero_grad()
    loss, x = model.forward_loss(po, sp, cand_ids, labels, loss_kind, smoothing)
    (loss.sum() / float(labels.numel())).backward()
    opt.step()
    return loss.detach()
