"""`OkgeAdagrad`: torch.optim.Adagrad's dense update (the optimizer OptimRegime builds for every reference config,
utils/optim.py:139-160) on the HIP sweep kernel -- one launch per PAIR of parameters instead of ATen's foreach chain.

The reference names its optimizer in YAML (`optimization_config.optimizer`) and looks the class up in
`torch.optim.__dict__` (utils/optim.py:143-144), constructing it from the PREVIOUS optimizer's param_groups -- which is
how Adam's eps = 1e-8 leaks into Adagrad.  Importing this module registers the class there, so
`optimizer: OkgeAdagrad` in a config selects it and the leaked keys are honoured the same way (torch's Optimizer keeps
keys a group already has).  State layout = torch.optim.Adagrad's ({'step', 'sum'} per parameter), so checkpoints move
between the two."""
from __future__ import annotations

import torch

from . import hotpath as H


class OkgeAdagrad(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, lr_decay=0, weight_decay=0, initial_accumulator_value=0, eps=1e-10):
        defaults = dict(lr=lr, lr_decay=lr_decay, eps=eps, weight_decay=weight_decay,
                        initial_accumulator_value=initial_accumulator_value)
        super().__init__(params, defaults)
        self._engines = {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state[p]
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["sum"] = torch.full_like(p, float(group["initial_accumulator_value"]), memory_format=torch.preserve_format)

    def _engine(self, dev):
        if dev not in self._engines:
            self._engines[dev] = H.HotPath(dev)
        return self._engines[dev]

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad(set_to_none=True) without its profiler scope and foreach bookkeeping (~15 us of
        host time per step on a 130 us device step); anything else goes to the base class"""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def step(self, closure=None):
        """One Adagrad update.  torch wraps every optimizer's `step` in a profiler scope + hook dispatch
        (Optimizer.profile_hook_step, ~30 us of host time per call); this class opts out of the wrapper (`hooked` below)
        and pays for it only when step hooks are actually registered."""
        if self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _global_step_hooks():
            return torch.optim.Optimizer.profile_hook_step(OkgeAdagrad._step)(self, closure)
        return self._step(closure)

    @torch.no_grad()
    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse or p.dtype != torch.float32 or g.dtype != torch.float32 or p.device.type != "cuda" or \
                        not p.is_contiguous():
                    raise RuntimeError("OkgeAdagrad updates dense contiguous fp32 parameters on the GPU (model_config.sparse: false)")
                st = self.state[p]
                if st["sum"].device != p.device:
                    st["sum"] = st["sum"].to(p.device)
                st["step"] += 1
                todo.append((p, g if g.is_contiguous() else g.contiguous(), st))
            if not todo:
                continue
            if group["lr_decay"] != 0:
                clrs = [group["lr"] / (1 + (float(st["step"]) - 1) * group["lr_decay"]) for _, _, st in todo]
            else:
                clrs = [group["lr"]] * len(todo)
            eng = self._engine(todo[0][0].device)
            i = 0
            while i < len(todo):
                if i + 1 < len(todo) and clrs[i] == clrs[i + 1]:
                    (p0, g0, s0), (p1, g1, s1) = todo[i], todo[i + 1]
                    eng.adagrad2(p0.data, g0, s0["sum"], p1.data, g1, s1["sum"], clrs[i], group["weight_decay"], group["eps"],
                                 zero_grad=False)
                    i += 2
                else:
                    p0, g0, s0 = todo[i]
                    eng.adagrad(p0.data, g0, s0["sum"], clrs[i], group["weight_decay"], group["eps"], zero_grad=False)
                    i += 1
        return loss


import sys as _sys

_optim_mod = _sys.modules[torch.optim.Optimizer.__module__]       # torch.optim.optimizer (the submodule, not the re-export)


def _global_step_hooks():
    return bool(getattr(_optim_mod, "_global_optimizer_pre_hooks", None)) or \
        bool(getattr(_optim_mod, "_global_optimizer_post_hooks", None))


OkgeAdagrad.step.hooked = True          # Optimizer._patch_step_function leaves a step marked like this alone


# nameable from the reference's YAML: `optimizer: OkgeAdagrad` -> torch.optim.__dict__["OkgeAdagrad"] (utils/optim.py:143)
torch.optim.OkgeAdagrad = OkgeAdagrad
