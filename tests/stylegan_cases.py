"""Shared case tables for the StyleGAN2 custom-op goldens (generator script + tests)."""
import numpy as np
import torch

ACTS = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]
UPFIR = [  # N, C, H, W, filter taps (1-D -> outer product unless sep), up, down, padding, flip, gain
    (2, 3, 8, 8, [1, 3, 3, 1], 2, 1, [2, 1, 2, 1], False, 4.0),
    (2, 5, 9, 7, [1, 3, 3, 1], 1, 2, [1, 1, 1, 1], False, 1.0),
    (1, 4, 16, 16, [1, 2, 3, 4], 1, 1, [1, 2, 2, 1], True, 1.0),
    (2, 2, 6, 6, [1, 4, 6, 4, 1], 2, 2, [0, 3, -1, 2], False, 0.7),
    (1, 8, 17, 17, [1, 3, 3, 1], 1, 1, [-1, -1, 0, 0], False, 1.0),
    (2, 3, 8, 8, [1, 2, 3, 4, 4, 3, 2, 1], 2, 1, [4, 3, 4, 3], True, 2.0),      # 8 taps -> separable in the reference
]


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)




# ---- a19: conv2d_gradfix / conv2d_resample / modulated_conv2d ------------------------------------------------------
CONV = [  # N, Cin, H, W, Cout, R, stride, padding, transpose, output_padding
    (2, 8, 9, 9, 12, 3, 1, 1, False, 0),
    (2, 16, 8, 8, 16, 3, 2, 0, False, 0),
    (1, 4, 11, 7, 6, 1, 1, 0, False, 0),
    (2, 8, 10, 10, 4, 3, 2, 1, False, 0),        # (H + 2p - R) % stride != 0: the data gradient needs output_padding
    (2, 5, 6, 6, 3, 3, 1, 2, False, 0),          # "full" padding
    (2, 8, 5, 5, 6, 3, 2, 0, True, 0),
    (2, 16, 4, 4, 16, 3, 2, 1, True, 1),
    (1, 4, 6, 6, 4, 1, 1, 0, True, 0),
    (2, 8, 4, 4, 8, 4, 2, 1, True, 0),
    (2, 32, 16, 16, 32, 3, 1, 1, False, 0),      # Cin % 16 == 0, pow2 grid: the fast loader paths
    (2, 32, 16, 16, 32, 3, 2, 1, True, 1),
]
RESAMPLE = [  # N, Cin, H, W, Cout, k, up, down, padding, flip_weight, flip_filter, filter taps
    (2, 8, 8, 8, 6, 3, 2, 1, 1, False, False, [1, 3, 3, 1]),     # SynthesisLayer conv0 (up)
    (2, 8, 8, 8, 6, 3, 1, 2, 1, True, False, [1, 3, 3, 1]),      # DiscriminatorBlock conv1 (down)
    (2, 8, 8, 8, 6, 1, 1, 2, 0, True, False, [1, 3, 3, 1]),      # DiscriminatorBlock skip (1x1, down)
    (2, 8, 8, 8, 6, 1, 2, 1, 0, True, False, [1, 3, 3, 1]),      # 1x1 + up
    (2, 8, 8, 8, 6, 3, 1, 1, 1, True, False, None),              # plain 3x3
    (2, 8, 8, 8, 3, 1, 1, 1, 0, True, False, None),              # ToRGB
    (1, 4, 6, 6, 4, 3, 2, 2, 1, True, True, [1, 2, 3, 1]),       # up and down, asymmetric filter, flipped
    (1, 4, 7, 7, 4, 3, 1, 1, [2, 0, 1, 1], False, False, None),  # asymmetric padding -> generic fallback
]
MODCONV = [  # N, Cin, H, W, Cout, k, up, demodulate, noise, fused_modconv
    (3, 8, 8, 8, 6, 3, 1, True, True, False),
    (3, 8, 8, 8, 6, 3, 2, True, True, False),
    (2, 8, 8, 8, 3, 1, 1, False, False, False),      # ToRGB
    (2, 8, 4, 4, 8, 3, 1, True, False, True),        # eval-mode fused request
    (2, 8, 4, 4, 8, 3, 2, True, True, True),
]


# ---- N1: StyleGAN2 networks / loss / training iteration -------------------------------------------------------------
SG2_NETS = {
    # name: dict(G kwargs, D kwargs, batch)
    "ic_r32": dict(
        G=dict(z_dim=32, c_dim=0, h_dim=24, w_dim=32, img_resolution=32, img_channels=3,
               mapping_kwargs=dict(num_layers=2), synthesis_kwargs=dict(channel_base=512, channel_max=64)),
        D=dict(c_dim=0, h_dim=24, img_resolution=32, img_channels=3, channel_base=512, channel_max=64,
               mapping_kwargs=dict(num_layers=2), epilogue_kwargs=dict(mbstd_group_size=2)),
        batch=4),
    "cc_ic_r16_resnetG": dict(
        G=dict(z_dim=16, c_dim=5, h_dim=12, w_dim=24, img_resolution=16, img_channels=3,
               mapping_kwargs=dict(num_layers=3), synthesis_kwargs=dict(channel_base=256, channel_max=32,
                                                                       architecture="resnet", conv_clamp=256)),
        D=dict(c_dim=5, h_dim=12, img_resolution=16, img_channels=3, channel_base=256, channel_max=32,
               architecture="skip", conv_clamp=256, mapping_kwargs=dict(num_layers=2),
               epilogue_kwargs=dict(mbstd_group_size=4)),
        batch=4),
}
# the reference's cfg=auto block precision (train.py:297-310: num_fp16_res=4, conv_clamp=256) scaled to a 32x32 net: the 16x16 and
# 32x32 blocks of G and D store fp16
SG2_NETS["ic_r32_fp16"] = dict(
    G=dict(z_dim=32, c_dim=0, h_dim=24, w_dim=32, img_resolution=32, img_channels=3, mapping_kwargs=dict(num_layers=2),
           synthesis_kwargs=dict(channel_base=512, channel_max=64, num_fp16_res=2, conv_clamp=256)),
    D=dict(c_dim=0, h_dim=24, img_resolution=32, img_channels=3, channel_base=512, channel_max=64, num_fp16_res=2,
           conv_clamp=256, mapping_kwargs=dict(num_layers=2), epilogue_kwargs=dict(mbstd_group_size=2)),
    # loss gain of the per-phase gradient comparison: at gain 1 the gradients inside the fp16 blocks of this synthetic net are
    # ~1e-6, i.e. at fp16's subnormal step (6e-8), and both implementations return quantisation noise there
    batch=4, phase_gain=1024.0)
# BASELINE.json configs[3] at its REAL network: IC-GAN StyleGAN2 256x256, `cfg=auto` on one GPU (stylegan2_ada_pytorch/train.py:291-372:
# fmaps 0.5 -> channel_base 16384, channel_max 512, 2 mapping layers, mbstd group 4, h_dim 2048), batch cut to 2 for the CPU run of
# the reference; fp32 throughout (`cfg4_r256`) and with the reference's block precision (`cfg4_r256_fp16`: num_fp16_res=4, conv_clamp=256)
_CFG4_COMMON = dict(channel_base=16384, channel_max=512)
SG2_REAL_NETS = {
    "cfg4_r256": dict(
        G=dict(z_dim=512, c_dim=0, h_dim=2048, w_dim=512, img_resolution=256, img_channels=3, mapping_kwargs=dict(num_layers=2),
               synthesis_kwargs=dict(**_CFG4_COMMON)),
        D=dict(c_dim=0, h_dim=2048, img_resolution=256, img_channels=3, mapping_kwargs=dict(num_layers=2),
               epilogue_kwargs=dict(mbstd_group_size=4), **_CFG4_COMMON),
        batch=2),
    "cfg4_r256_fp16": dict(
        G=dict(z_dim=512, c_dim=0, h_dim=2048, w_dim=512, img_resolution=256, img_channels=3, mapping_kwargs=dict(num_layers=2),
               synthesis_kwargs=dict(num_fp16_res=4, conv_clamp=256, **_CFG4_COMMON)),
        D=dict(c_dim=0, h_dim=2048, img_resolution=256, img_channels=3, mapping_kwargs=dict(num_layers=2), num_fp16_res=4,
               conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=4), **_CFG4_COMMON),
        batch=2, phase_gain=1024.0),
}
SG2_LOSS = dict(style_mixing_prob=0, r1_gamma=1.0, pl_batch_shrink=2, pl_decay=0.01, pl_weight=2)
SG2_OPT = dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)


def sg2_state(spec, seed):
    """Deterministic values for every state_dict entry (name -> tensor): N(0,1) weights like the reference's init,
    non-trivial biases / noise strengths / running averages so that every term of the networks is exercised."""
    out = {}
    for i, (name, shape) in enumerate(spec):
        rs = np.random.RandomState(seed * 100003 + i)
        if name.endswith("resample_filter"):
            out[name] = None                      # keep the constructor's value
            continue
        a = rs.standard_normal(shape).astype(np.float32)
        if name.endswith(".bias"):
            a = 0.2 * a + (1.0 if ".affine." in name else 0.0)
        elif name.endswith("noise_strength"):
            a = np.float32(0.1) * a
        elif name.endswith("w_avg"):
            a = 0.1 * a
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return out


def sg2_inputs(cfg, seed, n_sets):
    """n_sets batches of (z, c, h) plus one real batch (img in [-1,1], c, h)."""
    g, b = cfg["G"], cfg["batch"]
    rs = np.random.RandomState(seed)

    def cond(n):
        c = np.zeros((n, g["c_dim"]), dtype=np.float32)
        if g["c_dim"]:
            c[np.arange(n), rs.randint(0, g["c_dim"], size=n)] = 1
        h = rs.standard_normal((n, g["h_dim"])).astype(np.float32)
        h /= np.linalg.norm(h, axis=1, keepdims=True)
        return torch.from_numpy(c), torch.from_numpy(h)

    z = torch.from_numpy(rs.standard_normal((n_sets * b, g["z_dim"])).astype(np.float32))
    gc, gh = cond(n_sets * b)
    img = torch.from_numpy((rs.randint(0, 256, size=(b, 3, g["img_resolution"], g["img_resolution"])) / 127.5 - 1).astype(np.float32))
    rc, rh = cond(b)
    return z, gc, gh, img, rc, rh


# ---- N1: one synthesis / toRGB layer differentiated TWICE (what the path-length regulariser does, training/loss.py:120-139) ----------
# kind, Cin, Cout, w_dim, resolution, up, noise_mode, conv_clamp, N   (widths the fused nodes of stylegan_ops/fused_layers.py serve)
SG2_LAYER2 = [
    ("synthesis", 64, 64, 64, 16, 1, "const", 256, 3),
    ("synthesis", 64, 32, 64, 16, 2, "const", 256, 3),
    ("synthesis", 32, 64, 48, 8, 2, "none", None, 2),
    ("synthesis", 128, 128, 64, 8, 1, "const", 0.9, 2),            # a clamp that bites
    ("torgb", 64, 3, 64, 16, 1, None, 256, 3),
    ("torgb", 128, 3, 48, 8, 1, None, 0.5, 2),
]


def sg2_layer2_state(layer, seed):
    """seeded values for every parameter AND the noise_const buffer of a layer (same rule on the reference's and on this repository's
    modules: their state_dict keys agree)"""
    out = {}
    for i, (k, v) in enumerate(layer.state_dict().items()):
        if k.endswith("resample_filter"):
            continue
        t = rnd(tuple(v.shape), seed * 100 + i) if v.dim() else rnd((1,), seed * 100 + i)[0]
        if k.endswith("bias"):
            t = 0.3 * t + (1.0 if "affine" in k else 0.0)
        if k.endswith("noise_strength"):
            t = 0.3 * t
        out[k] = t
    return out


SG2_LAYER2_SALT = [4, 0, 0, 0, 0, 0]      # per case: added to the seed of x until no activation sits within 5e-6 of a kink (make_golden_sg2_layers2.py)


def sg2_layer2_inputs(case, idx):
    kind, cin, cout, wd, res, up, noise_mode, clamp, n = case
    x = rnd((n, cin, res // up, res // up), 2000 + 10 * idx + 1000 * SG2_LAYER2_SALT[idx])
    w = rnd((n, wd), 2001 + 10 * idx)
    img = rnd((n, 3, res, res), 2002 + 10 * idx) if kind == "torgb" else None
    return x, w, img


def sg2_layer2_probes(idx, y_shape, g_shape, gx_shape):
    """cotangent r, the extra differentiable factor rs (its gradient is the gradient w.r.t. the incoming cotangent), and the weights of
    the scalar built from the first-order results"""
    return dict(r=rnd(y_shape, 2003 + 10 * idx), rs=rnd(y_shape, 2004 + 10 * idx), q=rnd(g_shape, 2005 + 10 * idx),
                qx=rnd(gx_shape, 2006 + 10 * idx), qy=rnd(y_shape, 2007 + 10 * idx))


def sg2_layer2_run(idx, case, make_layer, dtype=torch.float64, device="cpu", act_dtype=None, torgb_call=None, context=None):
    """the second-order pattern of training/loss.py:120-139 on ONE layer -> dict of float64 numpy arrays.  Shared by
    tests/golden/make_golden_sg2_layers2.py (the reference's layer classes, float64, CPU) and by the tests (this repository's classes,
    fp32 parameters, fp32 / fp16 activations, CPU-emulated or HIP kernels).  `torgb_call(layer, x, w, img)`: how the image accumulation
    is expressed (the reference adds outside the layer, SynthesisBlock.forward 618-622); `context`: a context-manager factory entered
    around the whole computation (fused_layers.second_order)."""
    import contextlib
    kind, cin, cout, wd, res, up, noise_mode, clamp, n = case
    layer = make_layer(case)
    sd = sg2_layer2_state(layer, 7 + idx)
    cur = layer.state_dict()
    layer.load_state_dict({k: sd.get(k, cur[k]) for k in cur})
    layer = layer.to(dtype).to(device)
    if hasattr(layer, "resample_filter"):                   # (conv2d_resample.py:109 insists on a float32 filter; upfirdn2d casts it to x's type)
        layer.resample_filter = layer.resample_filter.float()
    x, w, img = sg2_layer2_inputs(case, idx)
    ins = [x.to(act_dtype or dtype).to(device).requires_grad_(True), w.to(dtype).to(device).requires_grad_(True)]
    if img is not None:
        ins.append(img.to(dtype).to(device).requires_grad_(True))
    names = [n_ for n_, _ in layer.named_parameters()]
    params = [p for _, p in layer.named_parameters()]
    with (context() if context is not None else contextlib.nullcontext()):
        if kind == "synthesis":
            y = layer(ins[0], ins[1], noise_mode=noise_mode, fused_modconv=False)
        elif torgb_call is not None:
            y = torgb_call(layer, *ins)
        else:
            y = ins[2] + layer(ins[0], ins[1], fused_modconv=False).to(dtype)
        y = y.to(dtype)
        pr = {k: v.to(dtype).to(device) for k, v in sg2_layer2_probes(idx, tuple(y.shape), (n, wd), tuple(x.shape)).items()}
        rs = pr["rs"].requires_grad_(True)
        g, gx = torch.autograd.grad([(y * (pr["r"] * rs)).sum()], [ins[1], ins[0]], create_graph=True, only_inputs=True)
        scalar = g.square().sum() * 0.5 + (g * pr["q"]).sum() + (gx.to(dtype) * pr["qx"]).sum() + (y * pr["qy"]).sum() * 0.1
        grads = torch.autograd.grad(scalar, ins + [rs] + params, allow_unused=True)
    out = {"y": y, "g": g, "gx": gx, "dd_x": grads[0], "dd_w": grads[1], "dd_rs": grads[len(ins)]}
    if img is not None:
        out["dd_img"] = grads[2]
    for n_, gr in zip(names, grads[len(ins) + 1:]):
        if gr is not None:
            out["dd_p/" + n_] = gr
    return {k: v.detach().double().cpu().numpy() for k, v in out.items()}
