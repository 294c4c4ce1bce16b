"""One iteration of the StyleGAN2 training loop (stylegan2_ada_pytorch/training/training_loop.py:428-539 with the
phase set-up of 313-346): lazy-regularisation phases (Gmain, Greg every `G_reg_interval`, Dmain, Dreg every
`D_reg_interval`), gradient accumulation over `batch_size // (batch_gpu * num_gpus)` rounds, NaN scrubbing, Adam, and the
G_ema update.  Data loading, ADA, metrics, snapshots and logging stay with the caller.

Optimisers are `ic_gan_amd.optim.FusedAdam` (one multi-tensor HIP launch per step; state_dict compatible with the
`torch.optim.Adam` the reference constructs from `G_opt_kwargs` / `D_opt_kwargs`, train.py:357-362)."""
from types import SimpleNamespace

import torch

from ..optim import FusedAdam
from .loss import StyleGAN2Loss


def make_phases(G, D, G_opt_kwargs, D_opt_kwargs, G_reg_interval=4, D_reg_interval=16, opt_class=FusedAdam):
    """training_loop.py:313-346.  opt kwargs: dict(lr=, betas=, eps=)."""
    phases = []
    for name, module, kw, interval in (("G", G, G_opt_kwargs, G_reg_interval), ("D", D, D_opt_kwargs, D_reg_interval)):
        kw = {k: v for k, v in dict(kw).items() if k != "class_name"}
        if interval is None:
            opt = opt_class(module.parameters(), **kw)
            phases.append(SimpleNamespace(name=name + "both", module=module, opt=opt, interval=1))
        else:     # lazy regularisation: one optimiser shared by the main and the regulariser phase
            mb_ratio = interval / (interval + 1)
            kw["lr"] = kw["lr"] * mb_ratio
            kw["betas"] = [beta ** mb_ratio for beta in kw["betas"]]
            opt = opt_class(module.parameters(), **kw)
            phases.append(SimpleNamespace(name=name + "main", module=module, opt=opt, interval=1))
            phases.append(SimpleNamespace(name=name + "reg", module=module, opt=opt, interval=interval))
    return phases


class TrainingStep:
    def __init__(self, G, D, G_ema, device, batch_size, batch_gpu, num_gpus=1, loss_kwargs=None, G_opt_kwargs=None,
                 D_opt_kwargs=None, G_reg_interval=4, D_reg_interval=16, ema_kimg=10, ema_rampup=None, ddp_modules=None):
        self.G, self.D, self.G_ema, self.device = G, D, G_ema, device
        self.batch_size, self.batch_gpu, self.num_gpus = batch_size, batch_gpu, num_gpus
        self.ema_kimg, self.ema_rampup = ema_kimg, ema_rampup
        mods = ddp_modules or dict(G_mapping=G.mapping, G_synthesis=G.synthesis, D=D)
        self.loss = StyleGAN2Loss(device=device, **mods, **(loss_kwargs or {}))
        adam = dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)
        self.phases = make_phases(G, D, G_opt_kwargs or adam, D_opt_kwargs or adam, G_reg_interval, D_reg_interval)
        self.cur_nimg = 0
        self.batch_idx = 0
        for m in (G, D, G_ema):
            m.requires_grad_(False)

    def __call__(self, real_img, real_c, real_h, all_gen_z, all_gen_c, all_gen_h):
        """real_img [batch_size/num_gpus, C, H, W] already scaled to [-1, 1]; real_c / real_h per sample;
        all_gen_* [len(phases) * batch_size/num_gpus, ...] — one batch of generator inputs per phase
        (training_loop.py:447-484).  Returns the names of the phases that ran."""
        n = len(self.phases)
        per = self.batch_size // self.num_gpus
        assert all_gen_z.shape[0] == n * per, (all_gen_z.shape, n, per)
        real_img, real_c, real_h = (t.split(self.batch_gpu) for t in (real_img, real_c, real_h))
        gen = [[t.split(self.batch_gpu) for t in full.split(per)] for full in (all_gen_z, all_gen_c, all_gen_h)]
        rounds = self.batch_size // (self.batch_gpu * self.num_gpus)
        ran = []
        for i, phase in enumerate(self.phases):
            if self.batch_idx % phase.interval != 0:
                continue
            phase.opt.zero_grad(set_to_none=True)
            phase.module.requires_grad_(True)
            for r, (x, c, h, gz, gc, gh) in enumerate(zip(real_img, real_c, real_h, gen[0][i], gen[1][i], gen[2][i])):
                self.loss.accumulate_gradients(phase=phase.name, real_img=x, real_c=c, real_h=h, gen_z=gz, gen_c=gc,
                                               gen_h=gh, sync=(r == rounds - 1), gain=phase.interval)
            phase.module.requires_grad_(False)
            # training_loop.py:511-515 (torch.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad) per parameter):
            # every fp32 gradient in one multi-tensor launch -- the loop is bound by the host, 144 launches per iteration at cfg4
            grads = [p.grad for p in phase.module.parameters() if p.grad is not None]
            from .. import ops as _ops
            _ops.nan_to_num_multi([g for g in grads if g.dtype == torch.float32 and g.is_contiguous()], nan=0, posinf=1e5, neginf=-1e5)
            for g in grads:
                if not (g.dtype == torch.float32 and g.is_contiguous()):
                    torch.nan_to_num(g, nan=0, posinf=1e5, neginf=-1e5, out=g)
            phase.opt.step()
            ran.append(phase.name)
        self.update_ema()
        self.cur_nimg += self.batch_size
        self.batch_idx += 1
        return ran

    @torch.no_grad()
    def update_ema(self):
        """training_loop.py:522-531."""
        ema_nimg = self.ema_kimg * 1000
        if self.ema_rampup is not None:
            ema_nimg = min(ema_nimg, self.cur_nimg * self.ema_rampup)
        beta = 0.5 ** (self.batch_size / max(ema_nimg, 1e-8))
        from .. import ops
        tg = [p for p in self.G_ema.parameters()]
        sr = [p.detach() for p in self.G.parameters()]
        ops.ema_multi([t.data for t in tg], sr, beta)          # p_ema <- lerp(p, p_ema, beta), one multi-tensor launch
        ops.bump_version(*tg)
        pairs = [(b_ema, b) for b_ema, b in zip(self.G_ema.buffers(), self.G.buffers())]
        same = [(t, s) for t, s in pairs if t.shape == s.shape and t.dtype == s.dtype and t.device == s.device]
        if same:                                            # one multi-tensor copy instead of a launch per buffer (34 at cfg4)
            torch._foreach_copy_([t for t, _ in same], [s for _, s in same])
        for t, s in pairs:
            if not (t.shape == s.shape and t.dtype == s.dtype and t.device == s.device):
                t.copy_(s)
