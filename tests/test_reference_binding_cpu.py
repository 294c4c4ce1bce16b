"""The reference-side binding of INTEGRATION.md, EXECUTED: the reference's own host code drives this
repository's modules through the seams it would be bound at.  CPU only, kernels emulated by oracle/kernel_ref.py, skipped when
/root/reference is absent (the GPU box).

  (a) BigGAN seam: the UNMODIFIED reference `train_fns.GAN_training_function` (train_fns.py:28-193),
      `utils.ema` (utils.py:1039-1067), `utils.toggle_grad` and `torch.optim.Adam` step `ic_gan_amd.BigGAN.{Generator,
      Discriminator, G_D}` -- what `model = __import__(config["model"])` (trainer.py:122) gives the trainer when config["model"]
      names this package's module -- and reproduce the golden the reference produced with its own modules.
  (b) StyleGAN2 plugin seam (section 3): the reference's `_bias_act_cuda` / `_upfirdn2d_cuda` autograd classes
      (torch_utils/ops/bias_act.py:174-317, upfirdn2d.py:196-349), initialised through the reference's `_init()` with
      `custom_ops.get_plugin` (custom_ops.py:52-148) replaced by `ic_gan_amd.stylegan_ops.plugin.get_plugin`, reproduce the
      goldens of the reference's `impl='ref'` path, first and second derivatives included.
"""
import contextlib
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import kernel_ref, synth
from tests.helpers import GRAD_RTOL, STATE_RTOL, adam_slack, check_group, load_golden
from tests.stylegan_cases import ACTS, UPFIR, rnd

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")


@contextlib.contextmanager
def _reference_modules(paths, names):
    """import the reference's top-level modules from `paths` (torchvision stubbed as in tests/golden/make_golden.py) and remove
    every trace afterwards: `utils`, `layers`, `losses` ... are generic names other tests must not inherit"""
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    saved_flag = sys.dont_write_bytecode
    sys.dont_write_bytecode = True                # never write __pycache__ into /root/reference
    for m in ("torchvision", "torchvision.transforms", "torchvision.utils"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path[:0] = paths
    try:
        yield [importlib.import_module(n) for n in names]
    finally:
        sys.path[:] = saved_path
        for k, mod in list(sys.modules.items()):          # only what came from the reference tree (+ the stubs); torch's own lazy
            if k not in saved_mods and (k.startswith("torchvision") or str(getattr(mod, "__file__", "") or "").startswith(REF)):
                del sys.modules[k]                        # imports must stay (re-importing them re-registers operators)
        sys.dont_write_bytecode = saved_flag


@pytest.mark.parametrize("case", ["cc_ic_r64", "ic_r64_acc2"])
def test_reference_train_fns_drive_this_repositorys_modules(case, monkeypatch):
    kernel_ref.install(monkeypatch)
    import ic_gan_amd.BigGAN as M                                    # config["model"] -> this module (trainer.py:122)
    g = load_golden(case)
    cfg = g["cfg"]
    with _reference_modules([REF, os.path.join(REF, "BigGAN_PyTorch")], ["train_fns", "utils"]) as (ref_train_fns, ref_utils):
        assert ref_train_fns.__file__.startswith(REF) and ref_utils.__file__.startswith(REF)
        G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
        D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
        G.load_state_dict(synth.synth_state(g["gspec"], 11))
        D.load_state_dict(synth.synth_state(g["dspec"], 22))
        G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
        ema = ref_utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])                     # the reference's EMA
        opt_d = torch.optim.Adam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), weight_decay=0,
                                 eps=cfg["adam_eps"])                                         # trainer.py:158-171
        opt_g = torch.optim.Adam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), weight_decay=0,
                                 eps=cfg["adam_eps"])
        GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
        state = {"itr": 0}
        gb, steps = int(g["g_batch"]), int(g["steps"])
        samp = synth.CondSampler(cfg, G.dim_z, gb, seed=7)
        train = ref_train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp, embedded_optimizers=False, device="cpu",
                                                    batch_size=gb)                            # the reference's step function
        dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
        for s in range(steps):
            x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
            state["itr"] += 1
            G.train(); D.train(); G_ema.train()
            m = train(x, y, f)
            np.testing.assert_allclose([m["G_loss"], m["D_loss_real"], m["D_loss_fake"]], g["losses"][s], rtol=5e-4, atol=5e-4)
            if s == 0:
                check_group(g, "step1/G_grad/", {n: p.grad for n, p in G.named_parameters() if p.grad is not None},
                            GRAD_RTOL, 1e-6, "G grad ")
                check_group(g, "step1/D_grad/", {n: p.grad for n, p in D.named_parameters() if p.grad is not None},
                            GRAD_RTOL, 1e-6, "D grad ")
            gx = adam_slack(g, "step1/G_grad/", cfg["G_lr"], s + 1, G.state_dict().keys())
            dx = adam_slack(g, "step1/D_grad/", cfg["D_lr"], s + 1, D.state_dict().keys())
            check_group(g, f"step{s + 1}/G_state/", G.state_dict(), STATE_RTOL, 2e-6, "G ", extra_atol=gx)
            check_group(g, f"step{s + 1}/D_state/", D.state_dict(), STATE_RTOL, 2e-6, "D ", extra_atol=dx)
            check_group(g, f"step{s + 1}/EMA_state/", G_ema.state_dict(), STATE_RTOL, 2e-6, "EMA ", extra_atol=gx)
    assert "train_fns" not in sys.modules and "utils" not in sys.modules


@pytest.fixture
def bound_ops(monkeypatch):
    """the reference's torch_utils.ops with `custom_ops.get_plugin` bound to this repository's plugin module (the one-line change
    of INTEGRATION.md section 3), kernels emulated, the plugin's device check lifted for the CPU run"""
    kernel_ref.install(monkeypatch)
    from ic_gan_amd.stylegan_ops import plugin
    monkeypatch.setattr(plugin, "_on_device", lambda t: True)
    monkeypatch.setattr(plugin, "_device_of", lambda t: contextlib.nullcontext())
    with _reference_modules([os.path.join(REF, "stylegan2_ada_pytorch")],
                            ["torch_utils.custom_ops", "torch_utils.ops.bias_act", "torch_utils.ops.upfirdn2d"]) as (co, ba, up):
        calls = []

        def get_plugin(module_name, sources=None, **kw):
            calls.append(module_name)
            return plugin.get_plugin(module_name, sources=sources, **kw)

        monkeypatch.setattr(co, "get_plugin", get_plugin)
        assert ba._init() and up._init()                             # the reference's own initialisation (bias_act.py:108-125)
        assert calls == ["bias_act_plugin", "upfirdn2d_plugin"] and ba._plugin is plugin.get_plugin("bias_act_plugin")
        yield ba, up


@pytest.mark.parametrize("act", ACTS)
def test_reference_bias_act_cuda_path_over_this_plugin(act, bound_ops):
    ba, _ = bound_ops
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stylegan_ops.npz"))
    ai = ACTS.index(act)
    for ci, clamp in enumerate([None, 0.7]):
        x = rnd((3, 6, 5, 5), 10 + ai, 1.5).requires_grad_(True)
        b = rnd((6,), 20 + ai, 0.5).requires_grad_(True)
        dy = rnd((3, 6, 5, 5), 30 + ai).requires_grad_(True)
        d2 = rnd((3, 6, 5, 5), 40 + ai)
        # what bias_act(..., impl='cuda') runs on a CUDA tensor (bias_act.py:165-171): the cached autograd class over `_plugin`
        y = ba._bias_act_cuda(dim=1, act=act, alpha=None, gain=None, clamp=clamp).apply(x, b)
        if act == "linear" and ci == 1:
            # the CUDA path does not mask the gradient of a clamped linear op (bias_act.py:262-266: nothing is saved for it);
            # the `ref` golden does -- a quirk of the reference itself, so only the forward value is comparable
            np.testing.assert_allclose(y.detach().numpy(), gold[f"ba/{act}/{ci}/y"], rtol=2e-5, atol=2e-5)
            continue
        dx, db = torch.autograd.grad(y, (x, b), dy, create_graph=True)
        ddx, ddy = torch.autograd.grad((dx * d2).sum(), (x, dy), allow_unused=True)
        k = f"ba/{act}/{ci}/"
        for name, got in (("y", y), ("dx", dx), ("db", db), ("ddx", ddx if ddx is not None else torch.zeros_like(x)), ("ddy", ddy)):
            ref = gold[k + name]
            np.testing.assert_allclose(got.detach().numpy(), ref, rtol=2e-5, atol=2e-5 * (np.abs(ref).max() + 1e-30),
                                       err_msg=f"{k}{name}")


@pytest.mark.parametrize("i", range(len(UPFIR)))
def test_reference_upfirdn2d_cuda_path_over_this_plugin(i, bound_ops):
    _, up = bound_ops
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stylegan_ops.npz"))
    n, c, h, w, taps, u, d, pad, flip, gain = UPFIR[i]
    x = rnd((n, c, h, w), 50 + i).requires_grad_(True)
    f = up.setup_filter(taps, flip_filter=False)
    upx, upy = up._parse_scaling(u)
    downx, downy = up._parse_scaling(d)
    padx0, padx1, pady0, pady1 = up._parse_padding(pad)
    # upfirdn2d(..., impl='cuda') on a CUDA tensor (upfirdn2d.py:187-193)
    y = up._upfirdn2d_cuda(up=u, down=d, padding=pad, flip_filter=flip, gain=gain).apply(x, f)
    dy = rnd(tuple(y.shape), 60 + i)
    (dx,) = torch.autograd.grad(y, x, dy)
    for name, got in (("y", y), ("dx", dx)):
        ref = gold[f"up/{i}/{name}"]
        np.testing.assert_allclose(got.detach().numpy(), ref, rtol=2e-5, atol=2e-5 * (np.abs(ref).max() + 1e-30), err_msg=f"up/{i}/{name}")
