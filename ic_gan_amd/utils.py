"""Step-level utilities the reference keeps in BigGAN_PyTorch/utils.py and that sit on the hot path or on
its checkpoint boundary: EMA, ortho regularisation, toggle_grad, seed_rng, save/load weights."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops


def seed_rng(seed):
    """utils.py:1019-1022 (torch + cuda + numpy)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)


def toggle_grad(model, on_or_off):
    """utils.py:1103-1105."""
    for param in model.parameters():
        param.requires_grad = on_or_off


class ema(object):
    """EMA of every state_dict entry (parameters AND buffers) — utils.py:1039-1067 — as ONE multi-tensor kernel."""

    def __init__(self, source, target, decay=0.9999, start_itr=0):
        self.source, self.target, self.decay, self.start_itr = source, target, decay, start_itr
        self.source_dict = self.source.state_dict()
        self.target_dict = self.target.state_dict()
        print("Initializing EMA parameters to be source parameters...")
        with torch.no_grad():
            for key in self.source_dict:
                self.target_dict[key].data.copy_(self.source_dict[key].data)
        ops.bump_version(*self.target_dict.values())         # `.data` writes do not bump autograd's version counters

    def update(self, itr=None):
        decay = 0.0 if (itr and itr < self.start_itr) else self.decay
        with torch.no_grad():
            tg, sr = [], []
            for key in self.source_dict:
                t, s = self.target_dict[key].data, self.source_dict[key].data
                if t.dtype != torch.float32 or s.dtype != torch.float32 or not t.is_contiguous() or not s.is_contiguous():
                    # entries the multi-tensor kernel does not take (non-fp32 / strided): the reference's expression
                    t.copy_(t * decay + s * (1 - decay))
                    continue
                tg.append(t)
                sr.append(s)
            ops.ema_multi(tg, sr, decay)
            ops.bump_version(*self.target_dict.values())     # (the kernel wrote through `.data` aliases)


def ortho(model, strength=1e-4, blacklist=[]):
    """Modified orthogonal regularisation, direct gradient form (utils.py:1073-1083):
    grad += strength * 2 * ((W W^T) * (1 - I)) W .  Off (strength 0) in every shipped config."""
    with torch.no_grad():
        for param in model.parameters():
            if len(param.shape) < 2 or any(param is item for item in blacklist):
                continue
            w = param.view(param.shape[0], -1).contiguous()
            m, k = w.shape
            gram = torch.empty(m, m, device=w.device, dtype=torch.float32)
            ops.gemm(w, w, gram, m, m, k, False, True)
            gram.fill_diagonal_(0.0)
            g = torch.empty(m, k, device=w.device, dtype=torch.float32)
            ops.gemm(gram, w, g, m, k, m, False, False)
            param.grad.data += (2.0 * strength) * g.view(param.shape)


def join_strings(base_string, strings):
    return base_string.join([item for item in strings if item])


def save_weights(G, D, state_dict, weights_root, experiment_name, name_suffix=None, G_ema=None,
                 embedded_optimizers=True, G_optim=None, D_optim=None):
    """Same files as the reference (utils.py:1116-1167): {G,D,G_ema,G_optim,D_optim,state_dict}[_suffix].pth"""
    root = "/".join([weights_root, experiment_name])
    os.makedirs(root, exist_ok=True)
    print("Saving weights to %s%s..." % (root, ("/" + name_suffix) if name_suffix else ""))
    # No collective in here: the reference writes checkpoints from rank 0 only (trainer.py:520 `... and rank == 0`,
    # train_fns.py:330/338), and rank 0's buffers are the canonical ones (DistributedDataParallel broadcasts FROM rank 0), so a
    # rank-0 checkpoint is right as it stands, with or without train_fns.COMM_SAVINGS.  A caller that writes from another rank, or
    # evaluates on every rank, calls sync_buffers(module) on ALL ranks first.

    def path(stem):
        return "%s/%s.pth" % (root, join_strings("_", [stem, name_suffix]))

    torch.save(G.state_dict(), path("G"))
    torch.save(D.state_dict(), path("D"))
    torch.save(state_dict, path("state_dict"))
    torch.save((G.optim if embedded_optimizers else G_optim).state_dict(), path("G_optim"))
    torch.save((D.optim if embedded_optimizers else D_optim).state_dict(), path("D_optim"))
    if G_ema is not None:
        torch.save(G_ema.state_dict(), path("G_ema"))


def load_weights(G, D, state_dict, weights_root, experiment_name, name_suffix=None, G_ema=None, strict=True,
                 load_optim=True, eval=False, map_location=None, embedded_optimizers=True, G_optim=None, D_optim=None):
    """Mirror of utils.py:1171-1265."""
    root = "/".join([weights_root, experiment_name])
    if not os.path.exists(root):
        print("Not loading data, experiment folder does not exist yet!")
        print(root)
        if eval:
            raise ValueError("Make sure foder exists")
        return

    def load(stem):
        # G / D / G_ema / optimizer files hold tensors and plain containers only; state_dict.pth pickles the whole config
        # (incl. nn.ReLU objects) and needs the full unpickler
        return torch.load("%s/%s.pth" % (root, join_strings("_", [stem, name_suffix])), map_location=map_location,
                          weights_only=(stem != "state_dict"))

    print("Loading %sweights from %s..." % ((name_suffix + " ") if name_suffix else "", root))
    if G is not None:
        G.load_state_dict(load("G"), strict=strict)
        if load_optim:
            (G.optim if embedded_optimizers else G_optim).load_state_dict(load("G_optim"))
    if D is not None:
        D.load_state_dict(load("D"), strict=strict)
        if load_optim:
            (D.optim if embedded_optimizers else D_optim).load_state_dict(load("D_optim"))
    try:
        saved = load("state_dict")
        for item in state_dict:
            if item in saved:
                state_dict[item] = saved[item]
    except Exception:
        print("No values to load")
    if G_ema is not None:
        G_ema.load_state_dict(load("G_ema"), strict=strict)


def sync_buffers(module, src: int = 0):
    """Broadcast `src`'s buffers (BN running statistics, spectral-norm u / sv) to every rank -- what DistributedDataParallel does
    before a synchronising forward (trainer.py:196-210 wraps with broadcast_buffers on).  For use before evaluation or a
    checkpoint on ranks other than `src` when the step ran with train_fns.COMM_SAVINGS (forwards under no_sync() skip DDP's own
    broadcast).  No-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    m = module.module if hasattr(module, "module") and hasattr(module, "no_sync") else module
    for b in m.buffers():
        dist.broadcast(b, src)
