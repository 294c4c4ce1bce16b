#!/usr/bin/env python
"""One Attention block (layers.Attention, reference layers.py:203-244) forward + backward at the cfg3 shapes, for rocprofv3
--kernel-trace --stats (tools only): which kernels its 11.6 / 13.2 ms of the block profile are.

    rocprofv3 --kernel-trace --stats ... -- python tools/attn_block_prof.py [G|D]"""
import functools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from ic_gan_amd import layers

which = sys.argv[1] if len(sys.argv) > 1 else "G"
ch, B = (384, 64) if which == "G" else (192, 128)
dev = "cuda:0"
torch.manual_seed(0)
conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, num_svs=1, num_itrs=1, eps=1e-6)
att = layers.Attention(ch, conv).to(dev)
with torch.no_grad():
    att.gamma.fill_(0.3)
x = torch.randn(B, ch, 64, 64, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
dy = torch.randn(B, ch, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
att.train()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for it in range(6):
    if it == 1:
        torch.cuda.synchronize(); t_f = t_b = 0.0
    ev[0].record()
    y = att(x)
    ev[1].record()
    y.backward(dy)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 1:
        t_f += ev[0].elapsed_time(ev[1]); t_b += ev[1].elapsed_time(ev[2])
    x.grad = None
    for p in att.parameters():
        p.grad = None
print("%s attention block [%d, %d, 64, 64]: forward %.3f ms, backward %.3f ms (5 iterations)" % (which, B, ch, t_f / 5, t_b / 5))
