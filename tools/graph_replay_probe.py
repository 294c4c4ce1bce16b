"""HIP-graph replay probe for the StyleGAN2 synthesis layers (tools only).

A piece's forward + backward is run eagerly twice, captured in a HIP graph, and replayed three times; per replay the parameters
whose gradient is not bit-identical to the eager one are listed.  Finding of round 4 (MI355X, ROCm 7.0, torch 2.10): every single
operation replays exactly, and so do modulated convolutions without a noise operand; modulated_conv2d with demodulation AND a noise
operand (+ bias_act) -- i.e. a SynthesisLayer -- is exact in replay 0 and differs from replay 1 on, in fp16 and in fp32 storage,
in gradients that vary with the memory layout -- also with the convolution replaced by an ATen einsum (case 3c) and with the WHOLE
composition written in ATen operations (3e / 3f: no kernel of this repository in the graph), not with the convolution removed (3b).  Replay 0 runs on fresh (zero) pool memory, later replays on
the previous replay's leftovers: something in that composition reads memory it did not write in the same replay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ic_gan_amd.stylegan_ops import conv2d_gradfix as CG, conv2d_resample as CR, bias_act as BA, modconv as MC
from ic_gan_amd.stylegan2 import networks as N

def pn(shape, salt, dtype=torch.float32, scale=1.0):
    n = int(np.prod(shape))
    t = torch.arange(n, device="cuda", dtype=torch.float32)
    return ((torch.sin(t * 12.9898 + salt) * 43758.5453).frac().mul(2).sub(1) * scale).reshape(shape).to(dtype)

class Box(torch.nn.Module):
    def __init__(self, **t):
        super().__init__()
        for k, v in t.items():
            setattr(self, k, torch.nn.Parameter(v))

def check(tag, fwd, **leaves):
    m = Box(**leaves).cuda()
    named = list(m.named_parameters()); params = [p for _, p in named]
    def run():
        m.requires_grad_(True)
        out = fwd(m)
        (out.float() * pn(out.shape, 6.5)).sum().backward()
        m.requires_grad_(False)
    m.requires_grad_(False)
    side = torch.cuda.Stream()
    for _ in range(2):
        for p in params: p.grad = None
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = [p.grad.detach().clone() for p in params]
    g = torch.cuda.CUDAGraph()
    for p in params: p.grad = None
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        run()
    held = [p.grad for p in params]
    res = []
    bits = lambda t: t.contiguous().view(torch.int32)
    for rep in range(3):
        g.replay(); torch.cuda.synchronize()
        res.append([(n if r.numel() > 1 else "%s: eager %.6g graph %.6g" % (n, float(r), float(h))) for (n, _), r, h in zip(named, ref, held)
                    if not torch.equal(bits(r), bits(h))])
    print("%-64s differing: %s" % (tag, res), flush=True)

B, C, H = 2, 512, 32
x0 = pn((B, C, H, H), 5.5)
nz = pn((B, 1, H, H), 10.5)
d0 = pn((B, C), 9.5) + 1.0
h = lambda t: t.to(torch.float16)
xcl = lambda m: h(m.x).contiguous(memory_format=torch.channels_last)
dd = lambda m: h(m.d).reshape(B, -1, 1, 1)
check("a  x + (nz * ns).half()            (fp16 channels-last)", lambda m: xcl(m) + h(nz * m.ns), x=x0, ns=torch.full([], 0.3))
check("a2 x + (nz * ns)                   (fp32 channels-last)", lambda m: m.x.contiguous(memory_format=torch.channels_last) + nz * m.ns, x=x0, ns=torch.full([], 0.3))
check("a3 x + (nz * ns)                   (fp32 NCHW)", lambda m: m.x + nz * m.ns, x=x0, ns=torch.full([], 0.3))
check("b  addcmul((nz * ns).half(), x, d) (fp16 channels-last)", lambda m: torch.addcmul(h(nz * m.ns), xcl(m), dd(m)), x=x0, d=d0, ns=torch.full([], 0.3))
check("c  x * d + (nz * ns).half()", lambda m: xcl(m) * dd(m) + h(nz * m.ns), x=x0, d=d0, ns=torch.full([], 0.3))
check("d  addcmul(noise_leaf.half(), x, d)", lambda m: torch.addcmul(h(m.nl), xcl(m), dd(m)), x=x0, d=d0, nl=nz.clone())
check("e  x + noise_leaf                  (fp32 NCHW, leaf [B,1,H,W])", lambda m: m.x + m.nl, x=x0, nl=nz.clone())
check("f  nz * ns only -> [B,1,H,W]", lambda m: nz * m.ns, ns=torch.full([], 0.3))
check("g  x * ns (0-dim leaf times a big tensor)", lambda m: m.x * m.ns, x=x0, ns=torch.full([], 0.3))
check("h  x + ns1 (1-element [1] leaf broadcast)", lambda m: m.x + m.n1, x=x0, n1=torch.full([1], 0.3))


# ---- compositions: a SynthesisLayer built up piece by piece (512 -> 512 @32, batch 2)
w0 = pn((C, C, 3, 3), 1.5)
aw, ab, wl = pn((C, 512), 12.5), pn((C,), 13.5) + 1.0, pn((B, 512), 11.5)
sty = lambda m: CG.linear_nt(m.wl, m.aw, alpha=1 / np.sqrt(512)) + m.ab.unsqueeze(0)
L = dict(x=x0, w=w0, wl=wl, aw=aw, ab=ab)
check("1 modconv(demod=False, noise=None), styles = affine(w)", lambda m: MC.modulated_conv2d(xcl(m), m.w, sty(m), noise=None, padding=1, demodulate=False), **L)
check("2 modconv(demod=True, noise=None)", lambda m: MC.modulated_conv2d(xcl(m), m.w, sty(m), noise=None, padding=1, demodulate=True), **L)
check("3 modconv(demod=True, noise = nz * strength)", lambda m: MC.modulated_conv2d(xcl(m), m.w, sty(m), noise=nz * m.ns, padding=1, demodulate=True), ns=torch.full([], 0.3), **L)
check("4 3 + bias_act(lrelu, clamp)", lambda m: BA.bias_act(MC.modulated_conv2d(xcl(m), m.w, sty(m), noise=nz * m.ns, padding=1, demodulate=True), m.bb.to(torch.float16), act="lrelu", gain=1.4, clamp=256), ns=torch.full([], 0.3), bb=pn((C,), 8.5), **L)
check("5 fp32 storage: modconv(demod=True, noise) + bias_act", lambda m: BA.bias_act(MC.modulated_conv2d(m.x.contiguous(memory_format=torch.channels_last), m.w, sty(m), noise=nz * m.ns, padding=1, demodulate=True), m.bb, act="lrelu", gain=1.4), ns=torch.full([], 0.3), bb=pn((C,), 8.5), **L)

# ---- case 3 with the convolution replaced / removed
def mod_noconv(m, conv):
    x, wgt, styles = xcl(m), m.w, sty(m)
    wgt = wgt * (1 / np.sqrt(C * 9) / wgt.norm(float("inf"), dim=[1, 2, 3], keepdim=True))
    styles = styles / styles.norm(float("inf"), dim=1, keepdim=True)
    wsq = wgt.square().sum(dim=[2, 3])
    dco = (CG.linear_nt(styles.square().float(), wsq.float()) + 1e-8).rsqrt()
    y = x * styles.to(x.dtype).reshape(B, -1, 1, 1)
    y = conv(y, wgt)
    return torch.addcmul((nz * m.ns).to(y.dtype), y, dco.to(y.dtype).reshape(B, -1, 1, 1))
check("3a case 3 spelled out (our conv)", lambda m: mod_noconv(m, lambda y, wgt: CG.conv2d(y, wgt.to(y.dtype), padding=1)), ns=torch.full([], 0.3), **L)
check("3b no convolution at all (identity)", lambda m: mod_noconv(m, lambda y, wgt: y * 1.0 + 0 * wgt.sum().to(y.dtype)), ns=torch.full([], 0.3), **L)
check("3c conv as an fp32 einsum over the centre tap", lambda m: mod_noconv(m, lambda y, wgt: torch.einsum("nchw,oc->nohw", y.float(), wgt[:, :, 1, 1]).to(y.dtype).contiguous(memory_format=torch.channels_last)), ns=torch.full([], 0.3), **L)
check("3d our conv, noise without the strength parameter", lambda m: mod_noconv(m, lambda y, wgt: CG.conv2d(y, wgt.to(y.dtype), padding=1)) + 0 * m.ns.to(torch.float16), ns=torch.full([], 0.3), **L)

# ---- 3c with every operation in ATen (no kernel of this repository at all): F.linear for the affine layer and the demodulation
def mod_aten(m):
    x = xcl(m)
    styles = torch.nn.functional.linear(m.wl, m.aw * (1 / np.sqrt(512)), m.ab)
    wgt = m.w * (1 / np.sqrt(C * 9) / m.w.norm(float("inf"), dim=[1, 2, 3], keepdim=True))
    styles = styles / styles.norm(float("inf"), dim=1, keepdim=True)
    dco = (torch.nn.functional.linear(styles.square(), wgt.square().sum(dim=[2, 3])) + 1e-8).rsqrt()
    y = x * styles.to(x.dtype).reshape(B, -1, 1, 1)
    y = torch.einsum("nchw,oc->nohw", y.float(), wgt[:, :, 1, 1]).to(x.dtype).contiguous(memory_format=torch.channels_last)
    return torch.addcmul((nz * m.ns).to(y.dtype), y, dco.to(y.dtype).reshape(B, -1, 1, 1))
check("3e the same composition in ATen only", mod_aten, ns=torch.full([], 0.3), **L)
check("3f ATen only, fp32 storage", lambda m: mod_aten(type("M", (), dict(x=m.x, w=m.w, wl=m.wl, aw=m.aw, ab=m.ab, ns=m.ns))) if False else mod_aten(m).float(), ns=torch.full([], 0.3), **L)
