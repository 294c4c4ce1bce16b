"""Replay of the fp64 forward's DISCRETE decisions in the HIP train step (test infrastructure, never the product).

An fp32 forward that differs from the exact one by rounding only still flips the sign of the few ReLU inputs that lie within
rounding distance of zero, and the winner of the few 2 x 2 max-pool windows whose two largest entries lie that close together;
each flip switches one path of the backward pass on or off and shows up as an isolated gradient sample far outside the
rounding-noise distribution of its tensor (what the dense four-part rule of tests/helpers.py allows for).  This module

  make_masks(case, out_dir)   (CPU) runs the fp64 oracle's train step on the golden's seeded inputs (the oracle's fp64 step IS the
                              reference's: the gradients are compared with tests/golden/biggan_<case>_f64.npz and must be equal)
                              and stores every 4-D ReLU input's sign pattern and every max-pool window's winner, in call order;
  hip_step(case, Nudger)      (GPU) runs the HIP train step with every ReLU-prologue input / max-pool input nudged IN PLACE, before
                              the kernel that reads it is launched (the backward kernels recompute their masks from the same saved
                              tensor), so that sign(x*scale+shift) and the window winners equal the fp64 pattern.

Fixture format (v2, tests/golden/decisions_<case>.npz): a decision can only differ between an fp32 and the fp64 forward where the
ReLU input lies within rounding distance of zero, so per ReLU input only the elements with |a| < BAND x rms(a) are stored (flat
NCHW index + sign, ~2e-4 of the elements) together with two digests of the FULL fp64 sign pattern (number of positive elements and
the sum of their flat indices): the replay imposes the stored signs and then proves with the digests that every other element agrees
too.  (v1 stored every sign bit: 4 MB per toy case, ~300 MB at the headline configuration.)  Max-pool winners (attention only, small)
are stored in full.

The nudge moves an element by <= 4e-6 of its own magnitude (it was within rounding distance of the decision boundary to begin
with), so the two runs differ by the flipped paths and nothing else.  tests/test_decision_replay_gpu.py holds the HIP gradients of
the replayed step to the fp64 reference two orders of magnitude tighter than GRAD_RTOL; tools/mask_attribution.py prints the
attribution table (profiles/r04_parity_report.txt)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
MASK_DIR = os.path.join(ROOT, "tools", "_masks")      # scratch (git-ignored); committed files: tests/golden/decisions_<case>.npz


def decisions_path(case):
    p = os.path.join(GOLDEN_DIR, "decisions_%s.npz" % case)
    return p if os.path.exists(p) else os.path.join(MASK_DIR, case + ".npz")


BAND = 2e-4          # |a| < BAND * rms(a): the elements whose sign an fp32 forward may take differently (measured deviations: ~1e-5)


def _sparse_decisions(a):
    """(shape, flat indices of the near-zero elements, their signs, #positive, sum of the flat indices of the positive elements)"""
    flat = a.reshape(-1)
    pos = flat > 0
    rms = float(flat.double().square().mean().sqrt())
    idx = torch.nonzero(flat.abs() < BAND * rms).reshape(-1)
    allpos = torch.nonzero(pos).reshape(-1)
    return (tuple(a.shape), idx.numpy().astype(np.int64), pos[idx].numpy(), int(allpos.numel()), int(allpos.sum()))


def _sparse_pool_decisions(win):
    """win [B, C, H/2, W/2, 4] -> (shape, flat indices of the windows whose two largest entries lie within BAND x rms of each other,
    their winners, digests of ALL winners: sum of the winners and sum of window index x winner)"""
    top2 = win.topk(2, dim=-1).values
    rms = float(win.double().square().mean().sqrt())
    winner = win.argmax(-1).reshape(-1)
    idx = torch.nonzero((top2[..., 0] - top2[..., 1]).reshape(-1) < BAND * rms).reshape(-1)
    ar = torch.arange(winner.numel(), dtype=torch.int64)
    return (tuple(win.shape[:4]), idx.numpy().astype(np.int64), winner[idx].to(torch.uint8).numpy(), int(winner.sum()), int((ar * winner).sum()))


def make_masks(case, out_path=None):
    import tests.helpers as H
    from oracle import biggan_oracle as O
    from oracle import synth
    g = H.load_golden(case)
    cfg = g["cfg"]
    gb, steps = int(g["g_batch"]), 1
    c64 = lambda t: t.double() if t.is_floating_point() else t
    gsd = {k: c64(v) for k, v in synth.synth_state(g["gspec"], 11).items()}
    dsd = {k: c64(v) for k, v in synth.synth_state(g["dspec"], 22).items()}
    dim_z = O.g_dims(cfg)["dim_z"]          # the Generator rounds dim_z down to a multiple of its hierarchy slots (BigGAN.py:174-177)
    samp32 = synth.CondSampler(cfg, dim_z, gb, seed=7)

    def samp():
        c = samp32()
        return tuple(c64(t) for t in c) if isinstance(c, tuple) else c64(c)

    masks, pools = [], []

    class FProxy:
        def __getattr__(self, name):
            return getattr(torch.nn.functional, name)

        @staticmethod
        def relu(x, *a, **k):
            if x.dim() == 4:
                masks.append(_sparse_decisions(x.detach()))
            return torch.nn.functional.relu(x, *a, **k)

        @staticmethod
        def max_pool2d(x, k, *a, **kw):
            assert list(k) == [2, 2] and not a and not kw
            B, C, H, W = x.shape
            win = x.detach().reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
            pools.append(_sparse_pool_decisions(win))                    # winner of a 2 x 2 window: 2 dy + dx
            return torch.nn.functional.max_pool2d(x, k)

    saved_F, O.F = O.F, FProxy()
    try:
        opt_g = O.AdamState(O.param_names(gsd), cfg["G_lr"], cfg["G_B1"], cfg["G_B2"], cfg["adam_eps"])
        opt_d = O.AdamState(O.param_names(dsd), cfg["D_lr"], cfg["D_B1"], cfg["D_B2"], cfg["adam_eps"])
        dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
        x, y, f = synth.synth_batch(cfg, dbatch, seed=100)
        losses, g_grads, d_grads = O.train_step(gsd, dsd, None, cfg, opt_g, opt_d, c64(x), y, c64(f) if f is not None else None, samp, 1, gb)
    finally:
        O.F = saved_F
    print(case, "oracle fp64 step:", losses, "golden losses", g["losses"][0])
    g64 = np.load(os.path.join(H.GOLDEN_DIR, "biggan_%s_f64.npz" % case), allow_pickle=False)
    worst, where = 0.0, ""      # the masks are the fp64 REFERENCE's only if the oracle's fp64 step is that reference's step
    top = max(float((v.double() ** 2).mean().sqrt()) for v in list(g_grads.values()) + list(d_grads.values()) if v is not None)
    for prefix, grads in (("step1/G_grad/", g_grads), ("step1/D_grad/", d_grads)):
        names = json.loads(str(g64[prefix + "names"]))
        ns = g64[prefix + "samp"].shape[1]
        for i, n in enumerate(names):
            if grads.get(n) is not None:
                s = H.fingerprint(grads[n], ns)[2]
                # (tensors whose whole gradient is rounding noise -- a bias feeding a BatchNorm -- are measured against 1e-9 of the largest rms)
                r = float(np.abs(s - g64[prefix + "samp"][i]).max() / max(np.sqrt((s ** 2).mean()), 1e-9 * top))
                if r > worst:
                    worst, where = r, prefix + n
    print(case, "oracle fp64 gradients vs the reference's fp64 goldens: largest |difference| / tensor rms = %.1e (%s)" % (worst, where))
    assert worst == 0.0, "the oracle's fp64 step is not the reference's fp64 step"
    out_path = out_path or os.path.join(MASK_DIR, case + ".npz")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    np.savez_compressed(out_path, version=2, band=BAND, n=len(masks), npool=len(pools),
                        shapes=json.dumps([m[0] for m in masks]),
                        digests=np.array([[m[3], m[4]] for m in masks], dtype=np.int64),
                        **{"i%d" % i: (m[1].astype(np.uint32) if int(np.prod(m[0])) < 2 ** 32 else m[1]) for i, m in enumerate(masks)},
                        **{"s%d" % i: np.packbits(m[2]) for i, m in enumerate(masks)},
                        pshapes=json.dumps([q[0] for q in pools]),
                        pdigests=np.array([[q[3], q[4]] for q in pools], dtype=np.int64).reshape(-1, 2),
                        **{"pi%d" % i: q[1].astype(np.uint32) for i, q in enumerate(pools)},
                        **{"pw%d" % i: q[2] for i, q in enumerate(pools)})
    print(case, len(masks), "ReLU inputs,", sum(int(np.prod(m[0])) for m in masks), "elements,", sum(m[1].size for m in masks),
          "of them inside the band;", len(pools), "max-pool inputs,", sum(int(np.prod(q[0])) for q in pools), "windows,",
          sum(q[1].size for q in pools), "inside the band")


# -------------------------------------------------------------------------------------------------------------------- run (GPU)
# forward entries with an ICG_PRE_RELU prologue: name -> positions of (x, scale, shift, ssb, flags) in the argument list
ENTRIES = {
    "icg_conv2d_fprop": (0, 5, 6, 7, 14), "icg_conv2d_fprop_ws": (0, 5, 6, 7, 14),
    "icg_conv2d_wino_fprop": (0, 5, 6, 7, 13), "icg_conv2d_wino4_fprop": (0, 5, 6, 7, 13),
    "icg_conv2d_up_fprop": (0, 4, 5, 6, 12), "icg_conv2d_up_wino_fprop": (0, 4, 5, 6, 12),
    "icg_conv2d_down_fprop": (0, None, None, None, 10), "icg_conv2d_down_wino_fprop": (0, None, None, None, 10),
    "icg_bn_apply": (0, 1, 2, 3, 7),
}


class Nudger:
    def __init__(self, case, L):
        z = np.load(decisions_path(case))
        assert int(z["version"]) == 2, "decision fixture of the old dense format: regenerate (tools/mask_attribution.py masks --golden)"
        self.z, self.shapes, self.n, self.digests = z, json.loads(str(z["shapes"])), int(z["n"]), z["digests"]
        self.i, self.ip, self.npool, self.L, self.orig = 0, 0, int(z["npool"]), L, L.call
        self.census = []                # (entry, shape, flips, residual mismatches)
        self.pool_census = []           # (shape, windows whose winner differs, left after the nudge)

    def decisions(self, shape):
        """-> (flat NCHW indices inside the band, the fp64 signs there, (#positive, index sum) of the full fp64 pattern)"""
        assert self.i < self.n, "more ReLU prologues in the HIP step than ReLUs in the oracle step"
        want = tuple(self.shapes[self.i])
        assert want == tuple(shape), ("ReLU #%d" % self.i, want, tuple(shape))
        idx = torch.from_numpy(self.z["i%d" % self.i].astype(np.int64)).cuda()
        sign = torch.from_numpy(np.unpackbits(self.z["s%d" % self.i])[: idx.numel()].astype(bool)).cuda()
        dig = self.digests[self.i]
        self.i += 1
        return idx, sign, (int(dig[0]), int(dig[1]))

    @staticmethod
    def _digest(a):
        pos = torch.nonzero(a.reshape(-1) > 0).reshape(-1)
        return int(pos.numel()), int(pos.sum())

    def nudge(self, name, x, scale, shift, ssb):
        idx, sign, digest = self.decisions(x.shape)
        B, C = x.shape[:2]
        xd = x.data
        if scale is not None:
            rows = B if ssb else 1                      # per-sample rows (conditional BN: ssb = C) or one shared row
            assert ssb in (0, C)
            sc, sh = scale.reshape(rows, C, 1, 1), shift.reshape(rows, C, 1, 1)
            a = xd * sc + sh
        else:
            sc = sh = None
            a = xd
        mism = (a.reshape(-1)[idx] > 0) != sign
        flips = int(mism.sum())
        if flips:
            bad = torch.zeros(a.numel(), dtype=torch.bool, device=a.device)
            bad[idx[mism]] = True
            bad = bad.view(a.shape)
            want = torch.zeros(a.numel(), dtype=torch.bool, device=a.device)
            want[idx] = sign
            want = want.view(a.shape)
            mag = (xd * sc).abs() + sh.abs() if sc is not None else xd.abs()
            target = torch.where(want, 1.0, -1.0) * (4e-6 * mag + 1e-30)
            xn = (target - sh) / torch.where(sc == 0, torch.ones_like(sc), sc) if sc is not None else target
            xd.copy_(torch.where(bad, xn, xd))
            a = xd * sc + sh if sc is not None else xd
        # residual: stored signs still unmatched, plus the full pattern's digests (elements outside the band)
        left = int(((a.reshape(-1)[idx] > 0) != sign).sum())
        if self._digest(a) != digest:
            left += 1
        self.census.append((name, tuple(x.shape), flips, left))

    def nudge_pool(self, x):
        """2 x 2 max-pool input: where the window's winner differs from the fp64 forward's, lift the fp64 winner just above the
        window maximum (it was within rounding distance of it)."""
        assert self.ip < self.npool
        idx = torch.from_numpy(self.z["pi%d" % self.ip].astype(np.int64)).cuda()
        wsel = torch.from_numpy(self.z["pw%d" % self.ip].astype(np.int64)).cuda()
        digest = tuple(int(v) for v in self.z["pdigests"][self.ip])
        pshape = tuple(json.loads(str(self.z["pshapes"]))[self.ip])
        self.ip += 1
        B, C, H, W = x.shape
        assert pshape == (B, C, H // 2, W // 2), (pshape, tuple(x.shape))
        win = x.data.unfold(2, 2, 2).unfold(3, 2, 2)                       # view [B, C, H/2, W/2, 2, 2] of the storage
        flat = win.reshape(B, C, H // 2, W // 2, 4)                         # (copy)
        mx, have = flat.max(-1)
        want = have.clone().reshape(-1)
        want[idx] = wsel                                                   # outside the band the fp32 winner is the fp64 one (digest below)
        want = want.view_as(have)
        bad = have != want
        n = int(bad.sum())
        if n:
            lifted = mx + 4e-6 * mx.abs() + 1e-30
            sel = torch.nn.functional.one_hot(want, 4).bool().reshape(B, C, H // 2, W // 2, 2, 2) & bad[..., None, None]
            win[sel] = lifted[..., None, None].expand_as(win)[sel]
            have = x.data.unfold(2, 2, 2).unfold(3, 2, 2).reshape(B, C, H // 2, W // 2, 4).argmax(-1)
        left = int((have != want).sum())
        hv = have.reshape(-1)
        if (int(hv.sum()), int((torch.arange(hv.numel(), device=hv.device) * hv).sum())) != digest:
            left += 1
        self.pool_census.append((tuple(x.shape), n, left))

    def call(self, name, *args):
        if name in ENTRIES:
            ix, isc, ish, iss, ifl = ENTRIES[name]
            if int(args[ifl]) & self.L.ICG_PRE_RELU:
                aff = isc is not None and (int(args[ifl]) & self.L.ICG_PRE_AFFINE)
                self.nudge(name, args[ix], args[isc] if aff else None, args[ish] if aff else None, int(args[iss]) if aff else 0)
        elif name == "icg_relu_sumpool_fwd":
            self.nudge(name, args[0], None, None, 0)
        elif name == "icg_maxpool2_fwd":
            self.nudge_pool(args[0])
        elif name == "icg_attn_split_pool":
            # the stacked projections of the attention block (ops.AttnProjFn): y [B, 2 d + dv, H, W], phi and g are pooled in that order
            y, d, dv = args[0], int(args[7]), int(args[8])
            self.nudge_pool(y[:, d:2 * d])
            self.nudge_pool(y[:, 2 * d:2 * d + dv])
        return self.orig(name, *args)


def hip_step(case, nudger_cls=None):
    """One golden train step on the GPU (tests/test_parity_gpu.py::_train_steps_case without the assertions); returns the 4096
    strided samples of every gradient tensor as {("G"|"D", name): samples} and the mask census."""
    import tests.helpers as H
    from tests import test_parity_gpu as T
    from oracle import synth
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    import ic_gan_amd._lib as L
    g = H.load_golden(case)
    cfg = g["cfg"]
    M, G, D, _, _ = T._build(cfg, g["gspec"], g["dspec"])
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True}).to("cuda")
    ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), eps=cfg["adam_eps"])
    opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), eps=cfg["adam_eps"])
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 1}
    gb = int(g["g_batch"])
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, synth.CondSampler(cfg, G.dim_z, gb, seed=7),
                                            embedded_optimizers=False, device="cuda", batch_size=gb)
    x, y, f = synth.synth_batch(cfg, gb * cfg["num_D_accumulations"] * cfg["num_D_steps"], seed=100)
    G.train(); D.train(); G_ema.train()
    nd = None
    if nudger_cls is not None:
        nd = nudger_cls(case, L)
        L.call = nd.call
    try:
        m = train(T._d(x), T._d(y), T._d(f))
    finally:
        if nd is not None:
            L.call = nd.orig
    if nd is not None:
        assert nd.i == nd.n and nd.ip == nd.npool, ("ReLUs / max-pools consumed", nd.i, "of", nd.n, nd.ip, "of", nd.npool)
    ns = g["step1/G_grad/samp"].shape[1]
    out = {}
    for tag, net in (("G", G), ("D", D)):
        for n, p in net.named_parameters():
            if p.grad is not None:
                out[(tag, n)] = H.fingerprint(p.grad, ns)[2]
    return out, (nd.census if nd else None), (nd.pool_census if nd else None)




def gradient_errors(case, samples):
    """{(net, name): (rms error, max error)} of hip_step()'s gradient samples against the fp64 reference's
    (tests/golden/biggan_<case>_f64.npz), both in units of the tensor's fp64 rms; tensors whose whole gradient is rounding noise
    (mathematically zero: a bias feeding a BatchNorm; rms < 1e-6 of the largest tensor rms) are left out."""
    import tests.helpers as H
    g64 = np.load(os.path.join(H.GOLDEN_DIR, "biggan_%s_f64.npz" % case), allow_pickle=False)
    rms_of = lambda r: np.sqrt((r ** 2).sum() / max(int(np.count_nonzero(r)), 1))
    top = max(rms_of(r) for pf in ("step1/G_grad/", "step1/D_grad/") for r in g64[pf + "samp"])
    out = {}
    for tag in "GD":
        prefix = "step1/%s_grad/" % tag
        for i, n in enumerate(json.loads(str(g64[prefix + "names"]))):
            r64 = g64[prefix + "samp"][i]
            rms = rms_of(r64)
            if (tag, n) not in samples or rms < 1e-6 * top:
                continue
            e = samples[(tag, n)] - r64
            out[(tag, n)] = (float(np.sqrt((e ** 2).sum() / max(int(np.count_nonzero(r64)), 1)) / rms), float(np.abs(e).max() / rms))
    return out
