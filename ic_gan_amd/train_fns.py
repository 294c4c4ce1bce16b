"""The IC-GAN G+D training step closure — same entry point and schedule as the reference's
``train_fns.GAN_training_function`` (BigGAN_PyTorch/train_fns.py:28-193):

    D phase: for each D step: zero D grads; for each accumulation: sample conditionings on the host,
             G forward without grad, D on fake++real, hinge loss / n_acc, backward; Adam(D)
    G phase: zero G grads; for each accumulation: sample, G forward, D forward, -mean(D_fake)/n_acc, backward
             (requires_grad of D is off, so only data-gradients flow through D); Adam(G); EMA

Every forward/backward op, both Adam steps and the EMA run in the HIP kernels; this file is host control
flow only.  Returns the same three floats (three device->host reads per step, as in the reference).

Data-parallel traffic (G and D wrapped in DistributedDataParallel as trainer.py:196-210 does).  The DEFAULT is the reference's
pattern (every backward all-reduces, every forward broadcasts rank 0's buffers: SURVEY F3 / F11), because a drop-in must not change
when replicas exchange state behind the caller's back.  `COMM_SAVINGS = True` (or ICG_COMM_SAVINGS=1 in the environment; bench.py
switches it on and says so in its JSON line) trims it:
  (1) accumulation rounds before the last run under DistributedDataParallel.no_sync() and accumulate locally; the last
      round's all-reduce carries the sum (mean of sums == sum of means up to the fp32 rounding of the all-reduce itself).  The
      reference all-reduces every round: num_*_accumulations x the bytes (4x for the shipped 16 x 4 schedule at 256x256);
  (2) D runs under no_sync() in the G phase while toggle_grads has it frozen.  With find_unused_parameters=True a reducer
      could ship the 100 M unused parameters (400 MB per step at cfg3); torch 2.10's does not (no autograd hook of the wrapper
      fires, so nothing is marked ready) -- measured by tests/test_ddp_gloo_cpu.py with a counting communication hook -- and the
      explicit no_sync() keeps it that way independently of the reducer's internals.
Neither changes a loss or a parameter: same losses, parameters of the replicas bit-identical, bucket traffic counted in the tests.
Side effect on BUFFERS: DistributedDataParallel broadcasts rank 0's buffers (BN running statistics, the spectral-norm `u` /
`sv`) at the start of a forward only when the previous forward was a synchronising one (`require_forward_param_sync`), so under
(1) and (2) the replicas' buffers are re-aligned less often than in the reference: between two such broadcasts every rank
advances its own running statistics / power iteration from its own shard.  Training does not read another rank's buffers; before
anything that does -- evaluation or a checkpoint written from a rank other than 0 -- call `utils.sync_buffers(module)` (an
explicit rank-0 broadcast; tests/test_ddp_gloo_cpu.py::test_ddp_buffers_after_accumulation).
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import losses, utils

# The START of the next step -- its first host-side conditioning draw and the generator forward of its first D accumulation, which need
# nothing from the next batch -- issued at the END of the current one, behind the G update and the EMA, before the three `.item()` reads
# block the host.  Without it the device idles ~3.7 ms at the start of every cfg3 step (profiles/r06_step_trace.txt): the host comes back
# from the loss read-back with an empty queue and needs ~12 ms to launch a generator forward whose first hundred kernels (4x4 ... 16x16
# blocks) take microseconds each.  Same operations on the same values in the same order -- G is not touched between the end of a step and
# its forward in the next one; the parameter / buffer version counters are re-checked when the stash is used and a stale one is dropped.
# OPT-IN (ICG_PREFETCH_NEXT_STEP=1; bench.py switches it on and says so): a caller whose data loader shares the sampler's global numpy /
# torch RNG sees one draw of the sampler move ahead of its next batch fetch -- a different (equally valid) stream than the reference's.
PREFETCH_NEXT_STEP = os.environ.get("ICG_PREFETCH_NEXT_STEP", "0") == "1"
COMM_SAVINGS = os.environ.get("ICG_COMM_SAVINGS", "0") == "1"      # opt-in: the default is the reference's traffic pattern


def _no_sync(module, on):
    """module.no_sync() when `module` is a DistributedDataParallel wrapper and `on`, else a null context."""
    if on and COMM_SAVINGS and hasattr(module, "no_sync") and hasattr(module, "module"):
        return module.no_sync()
    return contextlib.nullcontext()


def _frozen(module):
    return not any(p.requires_grad for p in module.parameters())


def dummy_training_function():
    def train(x, y):
        return {}

    return train


def GAN_training_function(G, D, GD, ema, state_dict, config, sample_conditionings, embedded_optimizers=True,
                          device="cuda", batch_size=0):
    def optimizers():
        if embedded_optimizers:
            return G.optim, D.optim
        return GD.optimizer_G, GD.optimizer_D

    ahead = {}            # the next step's first draw and generator output, issued by the previous call (PREFETCH_NEXT_STEP)
    try:                  # a G_D wrapper without the `G_z` keyword (the reference's own, a user's): no opening is issued ahead
        import inspect
        fwd = inspect.signature(GD.forward).parameters
        gd_takes_gz = callable(getattr(GD, "generate", None)) and ("G_z" in fwd or any(p.kind == p.VAR_KEYWORD for p in fwd.values()))
    except (TypeError, ValueError):
        gd_takes_gz = False
    ahead_host = {}       # pinned landing buffer of the three losses

    def g_versions():
        m = G.module if hasattr(G, "module") else G
        return tuple(t._version for t in m.parameters()) + tuple(t._version for t in m.buffers()) + (m.training,)

    def draw(features, y, truncate):
        """Host-side conditioning draw -> device tensors (train_fns.py:70-85 / 135-149)."""
        cond = sample_conditionings()
        labels_g = f_g = None
        if features is not None and y is not None:
            z_, labels_g, f_g = cond
        elif y is not None:
            z_, labels_g = cond
        elif features is not None:
            z_, f_g = cond
        else:
            z_ = cond
        if truncate:
            z_ = z_[:batch_size]
            labels_g = labels_g[:batch_size] if labels_g is not None else None
            f_g = f_g[:batch_size] if f_g is not None else None
        # a `Distribution` (Tensor subclass) would otherwise propagate its type through every op of the step
        z_ = z_.to(device, non_blocking=True).as_subclass(torch.Tensor)
        if labels_g is not None:
            labels_g = labels_g.to(device, non_blocking=True).long()
        if f_g is not None:
            f_g = f_g.to(device, non_blocking=True)
        return z_, labels_g, f_g

    def train(x, y=None, features=None):
        opt_G, opt_D = optimizers()
        opt_G.zero_grad()
        opt_D.zero_grad()
        x = torch.split(x, batch_size)
        y = torch.split(y, batch_size) if y is not None else None
        f_ = torch.split(features, batch_size) if features is not None else None
        counter = 0
        if config["toggle_grads"]:
            utils.toggle_grad(D, True)
            utils.toggle_grad(G, False)
        for _ in range(config["num_D_steps"]):
            opt_D.zero_grad()
            for acc in range(config["num_D_accumulations"]):
                G_z = None
                if ahead:
                    st = dict(ahead)
                    ahead.clear()
                    if st["key"] == (features is not None, y is not None) and st["versions"] == g_versions():
                        (z_, labels_g, f_g), G_z = st["cond"], st["G_z"]
                    # (else: the caller changed its conditioning pattern or touched G between two steps; the stash is dropped)
                if G_z is None:
                    z_, labels_g, f_g = draw(features, y, truncate=True)
                with _no_sync(D, acc + 1 < config["num_D_accumulations"]):     # earlier rounds accumulate locally
                    D_fake, D_real = GD(z_, labels_g, f_g, x[counter], y[counter] if y is not None else None,
                                        f_[counter] if f_ is not None else None, train_G=False,
                                        split_D=config["split_D"], policy=config["DiffAugment"], DA=config["DA"],
                                        **({"G_z": G_z} if G_z is not None else {}))
                    D_loss_real, D_loss_fake = losses.discriminator_loss(D_fake, D_real)
                    D_loss = (D_loss_real + D_loss_fake) / float(config["num_D_accumulations"])
                    D_loss.backward()
                counter += 1
            if config["D_ortho"] > 0.0:
                print("using modified ortho reg in D")
                utils.ortho(D, config["D_ortho"])
            opt_D.step()
        if config["toggle_grads"]:
            utils.toggle_grad(D, False)
            utils.toggle_grad(G, True)
        opt_G.zero_grad()
        d_frozen = _frozen(D)             # toggle_grads: D takes no gradient in this phase -> nothing of D's to all-reduce
        for acc in range(config["num_G_accumulations"]):
            z_, labels_g, f_g = draw(features, y, truncate=False)
            with _no_sync(D, d_frozen), _no_sync(G, acc + 1 < config["num_G_accumulations"]):
                D_fake = GD(z_, labels_g, f_g, train_G=True, split_D=config["split_D"], policy=config["DiffAugment"],
                            DA=config["DA"])
                G_loss = losses.generator_loss(D_fake) / float(config["num_G_accumulations"])
                G_loss.backward()
        if config["G_ortho"] > 0.0:
            print("using modified ortho reg in G")
            module = G.module if hasattr(G, "module") else G
            utils.ortho(G, config["G_ortho"], blacklist=[param for param in module.shared.parameters()])
        opt_G.step()
        if config["ema"]:
            ema.update(state_dict["itr"])
        if PREFETCH_NEXT_STEP and gd_takes_gz and config["num_D_steps"] > 0 and config["num_D_accumulations"] > 0 and not config["DA"]:
            # the three losses start their way to the host FIRST (one asynchronous copy into pinned memory + an event), then the next
            # step's opening is queued, then the host waits for the event only: it comes back while the device still has the generator
            # forward to run.  (`.item()` after the queued forward would wait for the forward too and give the idle gap back.)
            read = None
            if G_loss.is_cuda:
                if "pinned" not in ahead_host:
                    ahead_host["pinned"] = torch.empty(3, dtype=torch.float32, pin_memory=True)
                vals = torch.stack([G_loss.detach().reshape(()).float(), D_loss_real.detach().reshape(()).float(),
                                    D_loss_fake.detach().reshape(()).float()])
                ahead_host["pinned"].copy_(vals, non_blocking=True)
                read = torch.cuda.Event()
                read.record()
            ahead.clear()
            cond = draw(features, y, truncate=True)
            with torch.no_grad():             # exactly what G_D.forward(train_G=False) does first
                G_z = GD.generate(*cond)
            ahead.update(key=(features is not None, y is not None), cond=cond, G_z=G_z, versions=g_versions())
            if read is not None:
                read.synchronize()
                g, dr, df = (float(v) for v in ahead_host["pinned"].tolist())
                return {"G_loss": g, "D_loss_real": dr, "D_loss_fake": df}
        return {"G_loss": float(G_loss.item()), "D_loss_real": float(D_loss_real.item()),
                "D_loss_fake": float(D_loss_fake.item())}

    return train
