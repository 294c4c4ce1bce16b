#!/usr/bin/env python
"""cfg4 real network (fp16 blocks, batch 2 as the golden): the Greg phase's gradients with the second-order layer nodes, with the composed double
backward, and the reference's golden -- per tensor max |difference| / rms (the statistic of tests/test_stylegan2.py) and relative L2 (tools only, GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ic_gan_amd.stylegan2 import loss as LL, networks as N
from ic_gan_amd.stylegan2.loss import StyleGAN2Loss
from ic_gan_amd.stylegan_ops import fused_layers as FL
from tests.stylegan_cases import SG2_LOSS, SG2_REAL_NETS, sg2_inputs, sg2_state

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_r256_fp16"
dev = "cuda:0"
cfg = SG2_REAL_NETS[name]
gold = np.load(os.path.join(ROOT, "tests", "golden", "stylegan2_%s.npz" % name))
N._randn = lambda shape, device: torch.randn(shape).to(device)
LL._randn_like = lambda t: torch.randn(t.shape).to(t.device)
G = N.Generator(**cfg["G"]).train().requires_grad_(False).to(dev)
D = N.Discriminator(**cfg["D"]).train().requires_grad_(False).to(dev)
for m, seed in ((G, 1), (D, 2)):
    sd = sg2_state([[k, list(v.shape)] for k, v in m.state_dict().items()], seed)
    cur = m.state_dict()
    m.load_state_dict({k: (cur[k] if v is None else v.to(dev)) for k, v in sd.items()})
b = cfg["batch"]
z, gc, gh, img, rc, rh = (t.to(dev) for t in sg2_inputs(cfg, 7, 4))
res = {}
for mode in (True, False):
    FL.SECOND_ORDER_ENABLED = mode
    L = StyleGAN2Loss(device=dev, G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, **SG2_LOSS)
    G.requires_grad_(True)
    for p in G.parameters():
        p.grad = None
    torch.manual_seed(101)
    L.accumulate_gradients(phase="Greg", real_img=img, real_c=rc, real_h=rh, gen_z=z[:b], gen_c=gc[:b], gen_h=gh[:b], sync=True,
                           gain=cfg.get("phase_gain", 1))
    res[mode] = {n: (p.grad.detach().float().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for n, p in G.named_parameters()}
    G.requires_grad_(False)
names = json.loads(str(gold["grad/Greg/names"]))
rows = []
for i, n in enumerate(names):
    gs = gold["grad/Greg/samp"][i]
    ns = gs.shape[0]
    numel = res[True][n].numel()
    rms = float(np.sqrt(gold["grad/Greg/sq"][i] / max(numel, 1)))
    def samp(t):
        t = t.double().flatten()
        s = t[:: max(t.numel() // ns, 1)][:ns].numpy()
        out = np.zeros(ns); out[: s.size] = s
        return out
    a, c = samp(res[True][n]), samp(res[False][n])
    rows.append((n, rms, np.abs(a - gs).max() / max(rms, 1e-30), np.abs(c - gs).max() / max(rms, 1e-30), np.abs(a - c).max() / max(rms, 1e-30),
                 float((res[True][n] - res[False][n]).norm() / (res[False][n].norm() + 1e-30))))
rows.sort(key=lambda r: -r[2])
print("tensor, golden rms | max/rms: fused2 vs golden, composed vs golden, fused2 vs composed | rel L2 fused2 vs composed")
for r in rows[:14]:
    print("  %-44s %.2e | %.3f %.3f %.3f | %.3e" % r)
print("median over %d tensors: fused2-golden %.3f, composed-golden %.3f, fused2-composed %.3f" % (
    len(rows), np.median([r[2] for r in rows]), np.median([r[3] for r in rows]), np.median([r[4] for r in rows])))
