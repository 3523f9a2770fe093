"""OkgeAdagrad opts out of torch's per-call `profile_hook_step` wrapper (host time on a 130 us device step) -- and must still
behave like a torch.optim.Optimizer: step hooks run when registered, LR schedulers can wrap it, zero_grad keeps both modes,
the state layout stays torch.optim.Adagrad's.  CPU: parameters without gradients (the sweep itself needs the GPU)."""
import torch

from open_knowledge_graph_embeddings_amd.optim import OkgeAdagrad


def _opt():
    p = torch.nn.Parameter(torch.ones(4, 3))
    q = torch.nn.Parameter(torch.ones(2, 3))
    return OkgeAdagrad([p, q], lr=0.3, weight_decay=1e-10, eps=1e-8), p, q


def test_step_is_not_wrapped_and_hooks_still_run():
    opt, p, q = _opt()
    assert getattr(OkgeAdagrad.step, "hooked", False) and OkgeAdagrad.step.__name__ == "step"
    assert opt.step() is None                                  # no gradients: nothing to do, no device needed
    calls = []
    h1 = opt.register_step_pre_hook(lambda o, a, k: calls.append("pre"))
    h2 = opt.register_step_post_hook(lambda o, a, k: calls.append("post"))
    opt.step()
    assert calls == ["pre", "post"]
    h1.remove(), h2.remove()
    opt.step()
    assert calls == ["pre", "post"]
    assert opt.step(closure=lambda: torch.tensor(2.5)) == torch.tensor(2.5)


def test_zero_grad_modes_and_state_layout():
    opt, p, q = _opt()
    p.grad, q.grad = torch.ones_like(p), torch.ones_like(q)
    opt.zero_grad(set_to_none=False)
    assert p.grad is not None and float(p.grad.abs().sum()) == 0.0
    opt.zero_grad()
    assert p.grad is None and q.grad is None
    ref = torch.optim.Adagrad([torch.nn.Parameter(torch.ones(4, 3))], lr=0.3)
    assert set(opt.state[p].keys()) == set(ref.state[ref.param_groups[0]["params"][0]].keys()) == {"step", "sum"}
    sd = opt.state_dict()
    assert sorted(sd["param_groups"][0].keys()) == sorted(ref.state_dict()["param_groups"][0].keys() & sd["param_groups"][0].keys())


def test_lr_scheduler_can_wrap_it():
    opt, p, q = _opt()
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    opt.step()
    sched.step()
    assert abs(opt.param_groups[0]["lr"] - 0.15) < 1e-12
