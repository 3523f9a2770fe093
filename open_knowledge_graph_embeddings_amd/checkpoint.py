"""Checkpoint interop with the reference: `Trainer.save` / `Trainer.load`
(openkge/trainer.py:561-638) store

    {"epoch", "training_steps", "state_dict": model.state_dict(),
     "optimizer_state_dict": [OptimRegime.state_dict(), ...], "validation_results", "results"}

with model keys `entity_embedding.weight` / `relation_embedding.weight` and, per OptimRegime
(utils/optim.py:170-193), {"optimizer_state": torch.optim.Adagrad.state_dict(), "regime": [config...]}: Adagrad
state `{param index: {"step": float tensor, "sum": tensor}}`, parameter 0 = entity table, 1 = relation table, and one
param group that still carries the keys of the Adam shell the regime was born as (betas, amsgrad, eps=1e-8 ...).

These functions move that layout to and from the fused train steps (tables, Adagrad accumulators, step counter), for
one device or for the row-sharded step (shards are gathered to / scattered from full tables).  Files written here
contain tensors and plain containers only, so they load with `torch.load(..., weights_only=True)`; the reference's own
`results` entry (a pickled ResultsLog object) is never written and never read.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

ENTITY_KEY, RELATION_KEY = "entity_embedding.weight", "relation_embedding.weight"


def _gather_rows(local, n_rows, group):
    """all ranks' row blocks -> full (n_rows, d) tensor on every rank (blocks padded to the largest)."""
    world = dist.get_world_size(group)
    per = (n_rows + world - 1) // world
    pad = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), pad.view(-1), group=group)
    return out[:n_rows]


def _full_tables(step):
    """(E, R, sumE, sumR) as full tables, whatever the step class."""
    if hasattr(step, "ent_lo"):                                   # ShardedTrainStep: gather the row shards
        return (_gather_rows(step.E, step.n_ent, step.group), step.R, _gather_rows(step.sumE, step.n_ent, step.group),
                step.sumR)
    return step.E, step.R, step.sumE, step.sumR


def to_reference_checkpoint(step, epoch=0, training_steps=None):
    """The dict `Trainer.save` would write for this training state (tensors on the CPU)."""
    E, R, sumE, sumR = (t.detach().cpu().clone() for t in _full_tables(step))
    n_steps = float(step.steps)
    group = {"lr": step.lr, "betas": (0.9, 0.999), "eps": step.eps, "weight_decay": step.weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "decoupled_weight_decay": False, "lr_decay": 0, "initial_accumulator_value": 0, "params": [0, 1]}
    optimizer_state = {"state": {0: {"step": torch.tensor(n_steps), "sum": sumE},
                                 1: {"step": torch.tensor(n_steps), "sum": sumR}},
                       "param_groups": [group]}
    regime = [{"optimizer": "Adagrad", "epoch": 0, "lr": step.lr, "weight_decay": step.weight_decay}]
    return {"epoch": int(epoch), "training_steps": int(step.steps if training_steps is None else training_steps),
            "state_dict": {ENTITY_KEY: E, RELATION_KEY: R},
            "optimizer_state_dict": [{"optimizer_state": optimizer_state, "regime": regime}],
            "validation_results": None}


def save_checkpoint(path, step, epoch=0, training_steps=None):
    """Collective for the sharded step (every rank calls it); rank 0 writes."""
    ckpt = to_reference_checkpoint(step, epoch, training_steps)
    if not hasattr(step, "ent_lo") or step.rank == 0:
        torch.save(ckpt, path)
    return ckpt


def load_reference_checkpoint(step, ckpt_or_path, reset_optimizer=False):
    """Load tables (and, unless `reset_optimizer`, the Adagrad accumulators and step count -- the reference's
    `Trainer.load(reset_optimizer=...)` switch, trainer.py:561-605) into a fused / sharded train step."""
    ckpt = ckpt_or_path
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt_or_path, map_location="cpu", weights_only=True)
    sd = ckpt["state_dict"]
    E, R = sd[ENTITY_KEY], sd[RELATION_KEY]
    lo, hi = (step.ent_lo, step.ent_hi) if hasattr(step, "ent_lo") else (0, E.shape[0])
    if E[lo:hi].shape != step.E.shape or R.shape != step.R.shape:
        raise ValueError(f"checkpoint tables {tuple(E.shape)}, {tuple(R.shape)} do not fit this model")
    step.E.copy_(E[lo:hi])
    step.R.copy_(R)
    if reset_optimizer:
        step.sumE.zero_()
        step.sumR.zero_()
        step.steps = 0
    else:
        st = ckpt["optimizer_state_dict"][0]["optimizer_state"]
        group, state = st["param_groups"][0], st["state"]
        step.sumE.copy_(state[0]["sum"][lo:hi])
        step.sumR.copy_(state[1]["sum"])
        step.steps = int(float(state[0]["step"]))
        step.lr, step.weight_decay, step.eps = float(group["lr"]), float(group["weight_decay"]), float(group["eps"])
        if float(group.get("lr_decay", 0)) != 0 or float(group.get("initial_accumulator_value", 0)) != 0:
            raise NotImplementedError("Adagrad lr_decay / initial_accumulator_value are not used by the reference configs")
    step.dE.zero_()
    step.dR.zero_()
    return ckpt
