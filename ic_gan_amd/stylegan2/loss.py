"""Non-saturating logistic loss with lazy R1 and path-length regularisation
(stylegan2_ada_pytorch/training/loss.py:30-197; IC-GAN's variant threads the instance features `h` through G and D).

`accumulate_gradients(phase, ...)` runs the forward passes of one phase and back-propagates into `.grad`:
    Gmain  softplus(-D(G(z)))                                   (94-110)
    Greg   path length: |d(G(z)·noise)/dw| vs its running mean   (112-146)  — second-order through the synthesis net
    Dmain  softplus(D(G(z))) + softplus(-D(x))                   (148-178)
    Dreg   R1: (gamma/2) |dD(x)/dx|^2                            (180-194)  — second-order through the discriminator
`Gboth` / `Dboth` combine main and regulariser.  ADA augmentation (`augment_pipe`) is not part of this engine.
"""
import contextlib

import numpy as np
import torch

from ..stylegan_ops import conv2d_gradfix, fused_layers
from . import networks


def _mergeable(gen_z, real_img, gen_c, real_c, gen_h, real_h):
    """can D(generated) and D(real) run as ONE pass?  equal batch sizes and conditioning tensors that concatenate (same trailing
    shape, dtype and device); anything else keeps the reference's two passes"""
    if gen_z.shape[0] != real_img.shape[0]:
        return False
    for a, b in ((gen_c, real_c), (gen_h, real_h)):
        if (a is None) != (b is None):
            return False
        if a is not None and (a.shape != b.shape or a.dtype != b.dtype or a.device != b.device):
            return False
    return True


MERGE_D_PASSES = True      # Dmain: D(generated) and D(real) as one pass over the concatenated batch (False: two passes, as the reference runs them)


def _randn_like(t):
    """single entry point for the loss's random numbers (tests substitute a host-seeded generator)."""
    return torch.randn_like(t)


def _ddp_sync(module, sync):
    if sync or not isinstance(module, torch.nn.parallel.DistributedDataParallel):
        return contextlib.nullcontext()
    return module.no_sync()


class StyleGAN2Loss:
    def __init__(self, device, G_mapping, G_synthesis, D, augment_pipe=None, style_mixing_prob=0.9, r1_gamma=10,
                 pl_batch_shrink=2, pl_decay=0.01, pl_weight=2):
        if augment_pipe is not None:
            raise NotImplementedError("ADA augmentation is outside the hot path of this engine (pass augment_pipe=None)")
        self.device = device
        self.G_mapping, self.G_synthesis, self.D = G_mapping, G_synthesis, D
        self.style_mixing_prob, self.r1_gamma = style_mixing_prob, r1_gamma
        self.pl_batch_shrink, self.pl_decay, self.pl_weight = pl_batch_shrink, pl_decay, pl_weight
        self.pl_mean = torch.zeros([], device=device)
        self.stats = {}            # last reported values (the reference streams them to training_stats)

    def run_G(self, z, c, h, sync):
        with _ddp_sync(self.G_mapping, sync):
            ws = self.G_mapping(z, c, h)
            if self.style_mixing_prob > 0:
                cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
                cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff,
                                     torch.full_like(cutoff, ws.shape[1]))
                # ws[:, cutoff:] = mixed[:, cutoff:] (loss.py:52) as a select: slicing with a device scalar reads it back to the
                # host -- a device synchronisation per generator pass, and not capturable in a HIP graph; same values
                mixed = self.G_mapping(_randn_like(z), c, h, skip_w_avg_update=True)
                layer = torch.arange(ws.shape[1], device=ws.device).reshape(1, -1, 1)
                ws = torch.where(layer >= cutoff, mixed, ws)
        with _ddp_sync(self.G_synthesis, sync):
            img = self.G_synthesis(ws)
        return img, ws

    def run_D(self, img, c, h, sync):
        with _ddp_sync(self.D, sync):
            return self.D(img, c, h)

    def accumulate_gradients(self, phase, real_img, real_c, real_h, gen_z, gen_c, gen_h, sync, gain):
        assert phase in ["Gmain", "Greg", "Gboth", "Dmain", "Dreg", "Dboth"]
        do_Gmain = phase in ["Gmain", "Gboth"]
        do_Dmain = phase in ["Dmain", "Dboth"]
        do_Gpl = phase in ["Greg", "Gboth"] and self.pl_weight != 0
        do_Dr1 = phase in ["Dreg", "Dboth"] and self.r1_gamma != 0
        softplus = torch.nn.functional.softplus
        # Gmain / Dmain differentiate once: every layer is ONE autograd node there (stylegan_ops/fused_layers.py); the regularisers
        # differentiate twice and keep the composed operators.  The weights both networks' layers read were prepared for the
        # previous optimiser step: re-prepare all of them in one launch pair
        fused_layers.refresh(self.G_mapping, self.G_synthesis, self.D)

        if do_Gmain:
            with fused_layers.first_order():
                gen_img, _ = self.run_G(gen_z, gen_c, gen_h, sync=(sync and not do_Gpl))
                gen_logits = self.run_D(gen_img, gen_c, gen_h, sync=False)
                loss_Gmain = softplus(-gen_logits)
                self.stats["Loss/G/loss"] = loss_Gmain.detach()
                loss_Gmain.mean().mul(gain).backward()

        if do_Gpl:
            # the path-length penalty differentiates the synthesis network's backward: its modulated-convolution and toRGB layers run as
            # nodes whose backward is a node with a hand-written adjoint (stylegan_ops/fused_layers.py: second_order)
            with fused_layers.second_order():
                n = gen_z.shape[0] // self.pl_batch_shrink
                gen_img, gen_ws = self.run_G(gen_z[:n], gen_c[:n], gen_h[:n], sync=sync)
                pl_noise = _randn_like(gen_img) / np.sqrt(gen_img.shape[2] * gen_img.shape[3])
                with conv2d_gradfix.no_weight_gradients():
                    (pl_grads,) = torch.autograd.grad(outputs=[(gen_img * pl_noise).sum()], inputs=[gen_ws],
                                                      create_graph=True, only_inputs=True)
                pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
                pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
                self.pl_mean.copy_(pl_mean.detach())
                pl_penalty = (pl_lengths - pl_mean).square()
                loss_Gpl = pl_penalty * self.pl_weight
                self.stats["Loss/pl_penalty"] = pl_penalty.detach()
                (gen_img[:, 0, 0, 0] * 0 + loss_Gpl).mean().mul(gain).backward()

        loss_Dgen = 0
        if do_Dmain and not do_Dr1 and MERGE_D_PASSES and _mergeable(gen_z, real_img, gen_c, real_c, gen_h, real_h):
            # Dmain alone (every iteration): the generated and the real batch go through D in ONE pass of 2 B images and one backward
            # -- the same gradients as the reference's two passes (loss.py:148-178: every layer of D acts per sample; the minibatch-
            # standard-deviation layer is told to keep the two halves apart), half the launches and one all-reduce under DDP
            with fused_layers.first_order():
                gen_img, _ = self.run_G(gen_z, gen_c, gen_h, sync=False)
                b = gen_img.shape[0]
                with networks.separate_batches(2):
                    logits = self.run_D(torch.cat([gen_img.detach(), real_img.detach().to(gen_img.dtype)]), torch.cat([gen_c, real_c]),
                                        torch.cat([gen_h, real_h]), sync=sync)
                loss_Dgen, loss_Dreal = softplus(logits[:b]), softplus(-logits[b:])
                self.stats["Loss/D/loss"] = (loss_Dgen + loss_Dreal).detach()
                (loss_Dgen.mean() + loss_Dreal.mean()).mul(gain).backward()
            return
        if do_Dmain:
            with fused_layers.first_order():
                gen_img, _ = self.run_G(gen_z, gen_c, gen_h, sync=False)
                gen_logits = self.run_D(gen_img, gen_c, gen_h, sync=False)
                loss_Dgen = softplus(gen_logits)
                loss_Dgen.mean().mul(gain).backward()

        if do_Dmain or do_Dr1:
            real_img_tmp = real_img.detach().requires_grad_(do_Dr1)
            with (fused_layers.first_order() if not do_Dr1 else contextlib.nullcontext()):
                real_logits = self.run_D(real_img_tmp, real_c, real_h, sync=sync)
            loss_Dreal = 0
            if do_Dmain:
                loss_Dreal = softplus(-real_logits)
                self.stats["Loss/D/loss"] = (loss_Dgen + loss_Dreal).detach()
            loss_Dr1 = 0
            if do_Dr1:
                with conv2d_gradfix.no_weight_gradients():
                    (r1_grads,) = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp],
                                                      create_graph=True, only_inputs=True)
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                self.stats["Loss/r1_penalty"] = r1_penalty.detach()
            (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()
