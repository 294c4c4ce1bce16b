#!/usr/bin/env python
"""Is a StyleGAN2 training iteration reproducible from process to process?  cfg4 real network (fp16 blocks, the golden's batch): the four
phases Gmain / Greg / Dmain / Dreg of training/loss.py on seeded weights and inputs, an md5 per parameter gradient and phase.  Run it in
several processes and diff (tools/gpu_determinism.sh sg2).
    python tools/determinism_sg2.py [name]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from ic_gan_amd.stylegan2 import loss as LL, networks as N
from ic_gan_amd.stylegan2.loss import StyleGAN2Loss
from tests.stylegan_cases import SG2_LOSS, SG2_REAL_NETS, sg2_inputs, sg2_state

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg4_r256_fp16"
dev = "cuda:0"
cfg = SG2_REAL_NETS[name]
N._randn = lambda shape, device: torch.randn(shape).to(device)
LL._randn_like = lambda t: torch.randn(t.shape).to(t.device)
torch.manual_seed(11)
G = N.Generator(**cfg["G"]).train().requires_grad_(False).to(dev)
D = N.Discriminator(**cfg["D"]).train().requires_grad_(False).to(dev)
for m, seed in ((G, 1), (D, 2)):
    sd = sg2_state([[k, list(v.shape)] for k, v in m.state_dict().items()], seed)
    cur = m.state_dict()
    m.load_state_dict({k: (cur[k] if v is None else v.to(dev)) for k, v in sd.items()})
b = cfg["batch"]
z, gc, gh, img, rc, rh = (t.to(dev) for t in sg2_inputs(cfg, 7, 4))
print("init", hashlib.md5(b"".join(p.detach().cpu().numpy().tobytes() for p in list(G.parameters()) + list(D.parameters()) + list(G.buffers()))).hexdigest()[:12])
for pi, phase in enumerate(["Gmain", "Greg", "Dmain", "Dreg"]):
    L = StyleGAN2Loss(device=dev, G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, **SG2_LOSS)
    mod = G if phase[0] == "G" else D
    mod.requires_grad_(True)
    for p in mod.parameters():
        p.grad = None
    torch.manual_seed(100 + pi)
    L.accumulate_gradients(phase=phase, real_img=img, real_c=rc, real_h=rh, gen_z=z[:b], gen_c=gc[:b], gen_h=gh[:b], sync=True,
                           gain=cfg.get("phase_gain", 1))
    torch.cuda.synchronize()
    mod.requires_grad_(False)
    for n, p in mod.named_parameters():
        if p.grad is not None:
            print("grad", phase, n, hashlib.md5(p.grad.detach().cpu().numpy().tobytes()).hexdigest()[:10], flush=True)
