"""Fused multi-tensor Adam on the HIP kernel `icg_adam_multi`.

Drop-in for ``torch.optim.Adam(params, lr, betas, weight_decay=0, eps)`` as the reference builds it
(BigGAN_PyTorch/trainer.py:158-171, BigGAN.py:299-320): same update rule, and the same ``state_dict()``
structure (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``) so ``G_optim.pth`` / ``D_optim.pth``
written by either implementation load into the other.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import ops


class FusedAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam: weight_decay=0, amsgrad=False (the only setting the reference uses)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            by_step = {}
            live = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                live.append((p, st))
            # the per-parameter `step` counters (host tensors, kept for torch.optim.Adam's state_dict layout) advance in one call
            counters = [st["step"] for _, st in live if torch.is_tensor(st["step"])]
            if counters:
                torch._foreach_add_(counters, 1.0)
            for p, st in live:
                if not torch.is_tensor(st["step"]):
                    st["step"] += 1
                k = int(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(k, ([], [], [], []))
                lst = by_step[k]
                lst[0].append(p); lst[1].append(g); lst[2].append(st["exp_avg"]); lst[3].append(st["exp_avg_sq"])
            for k, (ps, gs, ms, vs) in by_step.items():
                ops.adam_multi(ps, gs, ms, vs, group["lr"], beta1, beta2, group["eps"], k)
        return loss
