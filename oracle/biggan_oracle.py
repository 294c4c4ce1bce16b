"""CPU oracle (TEST INFRASTRUCTURE, not product): functional restatement of the
IC-GAN BigGAN G+D training step in plain fp32 PyTorch ops.

Every function names the reference lines it follows (paths relative to
/root/reference/BigGAN_PyTorch).  The oracle works on a *flat state dict*
(``name -> tensor``) with exactly the reference's ``state_dict()`` key names,
so the same synthetic weights can be loaded into the reference (when
generating goldens), into this oracle, and into ``ic_gan_amd`` modules.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were
produced by the unmodified reference (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------
# architecture tables  (BigGAN.py:32-85 G_arch, BigGAN.py:390-432 D_arch)
# --------------------------------------------------------------------------
_G_MULT = {
    512: ([16, 16, 8, 8, 4, 2, 1], [16, 8, 8, 4, 2, 1, 1]),
    256: ([16, 16, 8, 8, 4, 2], [16, 8, 8, 4, 2, 1]),
    128: ([16, 16, 8, 4, 2], [16, 8, 4, 2, 1]),
    64: ([16, 16, 8, 4], [16, 8, 4, 2]),
    32: ([4, 4, 4], [4, 4, 4]),
}
_D_MULT = {
    256: ([1, 2, 4, 8, 8, 16], [1, 2, 4, 8, 8, 16, 16], [128, 64, 32, 16, 8, 4, 4]),
    128: ([1, 2, 4, 8, 16], [1, 2, 4, 8, 16, 16], [64, 32, 16, 8, 4, 4]),
    64: ([1, 2, 4, 8], [1, 2, 4, 8, 16], [32, 16, 8, 4, 4]),
    32: ([4, 4, 4], [4, 4, 4, 4], [16, 16, 16, 16]),
}


def _attn_set(spec: str) -> set:
    return {int(t) for t in str(spec).split("_") if t}


def g_arch(ch: int, attention: str, resolution: int) -> dict:
    """BigGAN.py:32-85."""
    cin, cout = _G_MULT[resolution]
    n = len(cout)
    res = [8 * 2 ** i for i in range(n)]
    att = _attn_set(attention)
    return {
        "in_channels": [ch * m for m in cin],
        "out_channels": [ch * m for m in cout],
        "resolution": res,
        "attention": [r in att for r in res],
    }


def d_arch(ch: int, attention: str, resolution: int) -> dict:
    """BigGAN.py:390-432."""
    cin, cout, res = _D_MULT[resolution]
    att = _attn_set(attention)
    down = [True] * (len(cout) - 1) + [False]
    if resolution == 32:
        down = [True, True, False, False]
    return {
        "in_channels": [3] + [ch * m for m in cin],
        "out_channels": [ch * m for m in cout],
        "downsample": down,
        "resolution": res,
        # the reference only tabulates 2**i for i in range(2, 8) (or 7 / 6)
        "attention": [(r in att) for r in res],
    }


def g_dims(cfg: dict) -> dict:
    """dim_z / chunk bookkeeping, BigGAN.py:171-226."""
    arch = g_arch(cfg["G_ch"], cfg["G_attn"], cfg["resolution"])
    dim_z = cfg.get("dim_z", 128)
    shared_dim = cfg.get("shared_dim", 0) or dim_z
    if cfg.get("hier", False):
        slots = len(arch["in_channels"]) + 1
        chunk = dim_z // slots
        dim_z = chunk * slots
    else:
        slots, chunk = 1, 0
    cc, ic = cfg.get("class_cond", True), cfg.get("instance_cond", False)
    bn_in = chunk if (cc or ic) else cfg.get("n_classes", 1000)
    if cc:
        bn_in += shared_dim
    if ic:
        bn_in += cfg.get("shared_dim_feat", 2048)
    return dict(arch=arch, dim_z=dim_z, slots=slots, chunk=chunk, bn_in=bn_in)


# --------------------------------------------------------------------------
# spectral norm  (layers.py:39-61 power_iteration, layers.py:98-112 SN.W_)
# --------------------------------------------------------------------------
def sn_weight(sd: State, prefix: str, training: bool, eps: float) -> Tensor:
    w = sd[prefix + ".weight"]
    wm = w.reshape(w.shape[0], -1)
    u = sd[prefix + ".u0"]
    with torch.no_grad():
        v = F.normalize(torch.matmul(u, wm), eps=eps)
        u_new = F.normalize(torch.matmul(v, wm.t()), eps=eps)
        if training:
            u.copy_(u_new)
    sigma = torch.squeeze(torch.matmul(torch.matmul(v, wm.t()), u_new.t()))
    if training:
        with torch.no_grad():
            sd[prefix + ".sv0"][:] = sigma
    return w / sigma


def sn_linear(sd, prefix, x, training, eps):
    """layers.py:157-165."""
    return F.linear(x, sn_weight(sd, prefix, training, eps), sd.get(prefix + ".bias"))


def sn_conv(sd, prefix, x, training, eps, padding):
    """layers.py:116-153."""
    return F.conv2d(x, sn_weight(sd, prefix, training, eps), sd.get(prefix + ".bias"), 1, padding)


# --------------------------------------------------------------------------
# normalisation  (layers.py:359-442 ccbn, layers.py:446-503 bn)
# --------------------------------------------------------------------------
def ccbn(sd, prefix, x, y, training, bn_eps, sn_eps):
    gain = 1.0 + sn_linear(sd, prefix + ".gain", y, training, sn_eps)
    bias = sn_linear(sd, prefix + ".bias", y, training, sn_eps)
    out = F.batch_norm(x, sd[prefix + ".stored_mean"], sd[prefix + ".stored_var"],
                       None, None, training, 0.1, bn_eps)
    return out * gain.view(y.shape[0], -1, 1, 1) + bias.view(y.shape[0], -1, 1, 1)


def plain_bn(sd, prefix, x, training, bn_eps, momentum=0.1):
    return F.batch_norm(x, sd[prefix + ".stored_mean"], sd[prefix + ".stored_var"],
                        sd[prefix + ".gain"], sd[prefix + ".bias"], training, momentum, bn_eps)


# --------------------------------------------------------------------------
# blocks  (layers.py:206-244 Attention, 512-552 GBlock, 556-613 DBlock)
# --------------------------------------------------------------------------
def attention(sd, prefix, x, training, sn_eps):
    b, c, h, w = x.shape
    theta = sn_conv(sd, prefix + ".theta", x, training, sn_eps, 0)
    phi = F.max_pool2d(sn_conv(sd, prefix + ".phi", x, training, sn_eps, 0), [2, 2])
    g = F.max_pool2d(sn_conv(sd, prefix + ".g", x, training, sn_eps, 0), [2, 2])
    theta = theta.view(-1, c // 8, h * w)
    phi = phi.view(-1, c // 8, h * w // 4)
    g = g.view(-1, c // 2, h * w // 4)
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
    o = torch.bmm(g, beta.transpose(1, 2)).view(-1, c // 2, h, w)
    o = sn_conv(sd, prefix + ".o", o, training, sn_eps, 0)
    return sd[prefix + ".gamma"] * o + x


def gblock(sd, prefix, x, y, training, bn_eps, sn_eps):
    h = F.relu(ccbn(sd, prefix + ".bn1", x, y, training, bn_eps, sn_eps))
    h = F.interpolate(h, scale_factor=2)          # nearest, BigGAN.py:260
    x = F.interpolate(x, scale_factor=2)
    h = sn_conv(sd, prefix + ".conv1", h, training, sn_eps, 1)
    h = F.relu(ccbn(sd, prefix + ".bn2", h, y, training, bn_eps, sn_eps))
    h = sn_conv(sd, prefix + ".conv2", h, training, sn_eps, 1)
    x = sn_conv(sd, prefix + ".conv_sc", x, training, sn_eps, 0)   # learnable_sc always (upsample)
    return h + x


def dblock(sd, prefix, x, training, sn_eps, preact: bool, down: bool, learnable_sc: bool):
    h = F.relu(x) if preact else x
    h = sn_conv(sd, prefix + ".conv1", h, training, sn_eps, 1)
    h = sn_conv(sd, prefix + ".conv2", F.relu(h), training, sn_eps, 1)
    if down:
        h = F.avg_pool2d(h, 2)
    s = x
    if preact:
        if learnable_sc:
            s = sn_conv(sd, prefix + ".conv_sc", s, training, sn_eps, 0)
        if down:
            s = F.avg_pool2d(s, 2)
    else:
        if down:
            s = F.avg_pool2d(s, 2)
        if learnable_sc:
            s = sn_conv(sd, prefix + ".conv_sc", s, training, sn_eps, 0)
    return h + s


# --------------------------------------------------------------------------
# networks  (BigGAN.py:350-386 Generator.forward, 617-642 Discriminator.forward,
#            655-711 G_D.forward)
# --------------------------------------------------------------------------
def generator_forward(sd: State, cfg: dict, z, label=None, feats=None, training=True,
                      taps: Optional[dict] = None):
    d = g_dims(cfg)
    arch = d["arch"]
    bn_eps, sn_eps = cfg.get("BN_eps", 1e-5), cfg.get("SN_eps", 1e-12)
    emb = []
    if label is not None:
        emb.append(F.embedding(label, sd["shared.weight"]))                   # nn.Embedding (no SN)
    if feats is not None:
        emb.append(sn_linear(sd, "shared_feat", feats, training, sn_eps))
    y = torch.cat(emb, -1) if emb else None
    nb = len(arch["out_channels"])
    if cfg.get("hier", False):
        zs = torch.split(z, d["chunk"], 1)
        z = zs[0]
        ys = [torch.cat([y, zi], 1) for zi in zs[1:]]
    else:
        ys = [y] * nb
    h = sn_linear(sd, "linear", z, training, sn_eps)
    bw = cfg.get("bottom_width", 4)
    h = h.view(h.size(0), -1, bw, bw)
    for i in range(nb):
        h = gblock(sd, f"blocks.{i}.0", h, ys[i], training, bn_eps, sn_eps)
        if taps is not None:
            taps[f"g.block{i}"] = h.detach().clone()
        if arch["attention"][i]:
            h = attention(sd, f"blocks.{i}.1", h, training, sn_eps)
            if taps is not None:
                taps[f"g.attn{i}"] = h.detach().clone()
    h = F.relu(plain_bn(sd, "output_layer.0", h, training, bn_eps))
    h = sn_conv(sd, "output_layer.2", h, training, sn_eps, 1)
    return torch.tanh(h)


def discriminator_forward(sd: State, cfg: dict, x, y=None, feat=None, training=True,
                          taps: Optional[dict] = None):
    arch = d_arch(cfg["D_ch"], cfg["D_attn"], cfg["resolution"])
    sn_eps = cfg.get("SN_eps", 1e-12)
    wide = cfg.get("D_wide", True)
    assert wide, "only wide D is restated (every shipped config uses D_wide=True)"
    h = x
    for i in range(len(arch["out_channels"])):
        cin, cout, down = arch["in_channels"][i], arch["out_channels"][i], arch["downsample"][i]
        h = dblock(sd, f"blocks.{i}.0", h, training, sn_eps, preact=(i > 0), down=down,
                   learnable_sc=(cin != cout) or down)
        if taps is not None:
            taps[f"d.block{i}"] = h.detach().clone()
        if arch["attention"][i]:
            h = attention(sd, f"blocks.{i}.1", h, training, sn_eps)
    h = torch.sum(F.relu(h), [2, 3])
    out = sn_linear(sd, "linear", h, training, sn_eps)
    if y is not None and feat is not None:
        e = F.embedding(y, sn_weight(sd, "embed", training, sn_eps))
        proj = torch.cat([e, sn_linear(sd, "linear_feat", feat, training, sn_eps)], -1)
        out = out + torch.sum(proj * h, 1, keepdim=True)
    elif y is not None:
        e = F.embedding(y, sn_weight(sd, "embed", training, sn_eps))
        out = out + torch.sum(e * h, 1, keepdim=True)
    elif feat is not None:
        out = out + torch.sum(sn_linear(sd, "linear_feat", feat, training, sn_eps) * h, 1, keepdim=True)
    return out


def gd_forward(gsd, dsd, cfg, z, gy, feats_g=None, x=None, dy=None, feats=None, train_G=False,
               training=True):
    """G_D.forward, non-split, no DiffAugment (BigGAN.py:670-711)."""
    with torch.set_grad_enabled(train_G):
        g_z = generator_forward(gsd, cfg, z, gy, feats_g, training)
    d_in = torch.cat([g_z, x], 0) if x is not None else g_z
    d_cls = torch.cat([gy, dy], 0) if dy is not None else gy
    if feats_g is not None:
        d_f = torch.cat([feats_g, feats], 0) if feats is not None else feats_g
    else:
        d_f = None
    d_out = discriminator_forward(dsd, cfg, d_in, d_cls, d_f, training)
    if x is not None:
        return torch.split(d_out, [g_z.shape[0], x.shape[0]])
    return d_out


# --------------------------------------------------------------------------
# losses / optimiser / EMA
# --------------------------------------------------------------------------
def loss_hinge_dis(d_fake, d_real):
    """losses.py:24-27."""
    return torch.mean(F.relu(1.0 - d_real)), torch.mean(F.relu(1.0 + d_fake))


def loss_hinge_gen(d_fake):
    """losses.py:36-38."""
    return -torch.mean(d_fake)


class AdamState:
    """torch.optim.Adam(betas, eps, weight_decay=0) restated (trainer.py:158-171).

    Formula of torch 2.x single-tensor Adam (amsgrad=False, maximize=False):
      m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
      p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
    """

    def __init__(self, names: List[str], lr, b1, b2, eps):
        self.names, self.lr, self.b1, self.b2, self.eps = names, lr, b1, b2, eps
        self.t = 0
        self.m: Dict[str, Tensor] = {}
        self.v: Dict[str, Tensor] = {}

    @torch.no_grad()
    def step(self, sd: State, grads: Dict[str, Optional[Tensor]]):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        for n in self.names:
            g = grads.get(n)
            if g is None:
                continue
            p = sd[n]
            if n not in self.m:
                self.m[n] = torch.zeros_like(p)
                self.v[n] = torch.zeros_like(p)
            m, v = self.m[n], self.v[n]
            m.lerp_(g, 1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


@torch.no_grad()
def ema_update(src: State, tgt: State, itr, decay, start_itr):
    """utils.py:1055-1067 — applies to every state_dict key incl. buffers."""
    d = 0.0 if (itr and itr < start_itr) else decay
    for k in src:
        tgt[k].copy_(tgt[k] * d + src[k] * (1 - d))


PARAM_SUFFIXES = (".weight", ".bias", ".gain", ".gamma")


def param_names(sd: State) -> List[str]:
    """Names that are nn.Parameters in the reference (everything except u0/sv0/stored_*)."""
    out = []
    for k in sd:
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("u0", "sv0", "stored_mean", "stored_var"):
            continue
        out.append(k)
    return out


def train_step(gsd: State, dsd: State, ema_sd: Optional[State], cfg: dict, opt_g: AdamState,
               opt_d: AdamState, x, y, feats, sample_conditionings, itr: int, g_batch: int):
    """One call of train_fns.GAN_training_function.train (train_fns.py:40-191),
    toggle_grads=True, split_D=False, DA=False, ortho=0."""
    gp, dp = param_names(gsd), param_names(dsd)
    xs = torch.split(x, g_batch)
    ys = torch.split(y, g_batch) if y is not None else None
    fs = torch.split(feats, g_batch) if feats is not None else None

    def unpack(c):
        lab = fg = None
        if feats is not None and y is not None:
            z_, lab, fg = c
        elif y is not None:
            z_, lab = c
        elif feats is not None:
            z_, fg = c
        else:
            z_ = c
        return z_, lab, fg

    counter = 0
    for _ in range(cfg.get("num_D_steps", 1)):
        for n in dp:
            dsd[n].requires_grad_(True)
            dsd[n].grad = None
        for n in gp:
            gsd[n].requires_grad_(False)
        nacc = cfg.get("num_D_accumulations", 1)
        for _ in range(nacc):
            z_, lab, fg = unpack(sample_conditionings())
            z_ = z_[:g_batch]
            lab = lab[:g_batch].long() if lab is not None else None
            fg = fg[:g_batch] if fg is not None else None
            d_fake, d_real = gd_forward(gsd, dsd, cfg, z_, lab, fg, xs[counter],
                                        ys[counter] if ys is not None else None,
                                        fs[counter] if fs is not None else None, train_G=False)
            l_real, l_fake = loss_hinge_dis(d_fake, d_real)
            ((l_real + l_fake) / float(nacc)).backward()
            counter += 1
        opt_d.step(dsd, {n: dsd[n].grad for n in dp})
    for n in dp:
        dsd[n].requires_grad_(False)
    for n in gp:
        gsd[n].requires_grad_(True)
        gsd[n].grad = None
    nacc = cfg.get("num_G_accumulations", 1)
    for _ in range(nacc):
        z_, lab, fg = unpack(sample_conditionings())
        lab = lab.long() if lab is not None else None
        d_fake = gd_forward(gsd, dsd, cfg, z_, lab, fg, train_G=True)
        g_loss = loss_hinge_gen(d_fake) / float(nacc)
        g_loss.backward()
    g_grads = {n: gsd[n].grad for n in gp}
    opt_g.step(gsd, g_grads)
    for n in gp:
        gsd[n].requires_grad_(False)
    if ema_sd is not None:
        ema_update(gsd, ema_sd, itr, cfg.get("ema_decay", 0.9999), cfg.get("ema_start", 0))
    return {"G_loss": float(g_loss.item()), "D_loss_real": float(l_real.item()),
            "D_loss_fake": float(l_fake.item())}, g_grads, {n: dsd[n].grad for n in dp}
