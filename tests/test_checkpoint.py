"""Checkpoint consumers/producers against a checkpoint WRITTEN BY THE REFERENCE
(tests/golden/ckpt_cc_ic_r64/, made by make_golden_checkpoint.py with the reference's utils.save_weights after two
reference training steps):

  * load -> eval-mode samples of G_ema / G match the reference's within 1e-3 relative L2 (north_star's bar)
  * load (weights + Adam moments + counters) -> one more training step reproduces the reference's step 3
  * save -> files carry the reference's key names / shapes / optimizer-state structure, values round-trip
  * load_model_inference picks the best-FID checkpoint and adopts its config

CPU variants route the C-ABI to oracle/kernel_ref.py (test-only); GPU variants run the HIP library.
"""
import os
import shutil

import numpy as np
import pytest
import torch

from oracle import kernel_ref, synth
from tests.helpers import GOLDEN_DIR, STATE_RTOL, check_group

NAME = "ckpt_cc_ic_r64"
GOLD = np.load(os.path.join(GOLDEN_DIR, NAME + ".npz"), allow_pickle=False)
import json
CFG = json.loads(str(GOLD["cfg"]))


@pytest.fixture
def emu(monkeypatch):
    kernel_ref.install(monkeypatch)


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _build(device):
    import ic_gan_amd.BigGAN as M
    from ic_gan_amd import utils
    from ic_gan_amd.optim import FusedAdam
    G = M.Generator(**{**CFG, "skip_init": True, "embedded_optimizers": False}).to(device)
    D = M.Discriminator(**{**CFG, "skip_init": True, "embedded_optimizers": False}).to(device)
    G_ema = M.Generator(**{**CFG, "skip_init": True, "no_optim": True}).to(device)
    og = FusedAdam(G.parameters(), lr=CFG["G_lr"], betas=(CFG["G_B1"], CFG["G_B2"]), eps=CFG["adam_eps"])
    od = FusedAdam(D.parameters(), lr=CFG["D_lr"], betas=(CFG["D_B1"], CFG["D_B2"]), eps=CFG["adam_eps"])
    state = {"itr": 0, "epoch": 0, "save_num": 0, "save_best_num": 0, "best_IS": 0, "best_FID": 999999, "config": {}}
    ema = utils.ema(G, G_ema, CFG["ema_decay"], CFG["ema_start"])     # trainer.py order: ema() first, then load_weights
    utils.load_weights(G, D, state, GOLDEN_DIR, NAME, None, G_ema, strict=True, load_optim=True,
                       map_location=device, embedded_optimizers=False, G_optim=og, D_optim=od)
    return M, G, D, G_ema, og, od, state, ema


def _check_samples(device):
    M, G, D, G_ema, og, od, state, ema = _build(device)
    assert state["itr"] == 2 and state["config"]["resolution"] == 64
    z, y, f = (torch.from_numpy(GOLD["sample/" + k]).to(device) for k in ("z", "y", "feats"))
    from ic_gan_amd import inference
    G_ema.eval(); G.eval(); D.eval()
    img_ema, y_out, f_out = inference.sample(G_ema, lambda: (z, y, f), CFG, class_cond=True, instance_cond=True, device=device)
    img_g, _, _ = inference.sample(G, lambda: (z, y, f), CFG, class_cond=True, instance_cond=True, device=device)
    assert img_ema.shape == (6, 3, 64, 64) and torch.equal(y_out, y)
    assert _rel_l2(img_ema.cpu().numpy(), GOLD["sample/G_ema"]) < 1e-3       # measured ~1e-6
    assert _rel_l2(img_g.cpu().numpy(), GOLD["sample/G"]) < 1e-3
    np.testing.assert_allclose(img_ema.cpu().numpy(), GOLD["sample/G_ema"], rtol=0, atol=2e-5)
    with torch.no_grad():
        logit = D(torch.from_numpy(GOLD["sample/G_ema"]).to(device), y, f)
    np.testing.assert_allclose(logit.cpu().numpy(), GOLD["sample/D_logit"], rtol=2e-4, atol=2e-4)
    # eval mode must not have touched the SN vectors / BN statistics
    ref = torch.load(os.path.join(GOLDEN_DIR, NAME, "G_ema.pth"), map_location="cpu", weights_only=False)
    for k, v in G_ema.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k


def _check_resume(device, slack):
    M, G, D, G_ema, og, od, state, ema = _build(device)
    from ic_gan_amd import train_fns, utils
    GD = M.G_D(G, D, optimizer_G=og, optimizer_D=od)
    gb = int(GOLD["g_batch"])
    samp = synth.CondSampler(CFG, G.dim_z, gb, seed=7)
    samp(), samp(), samp(), samp()                # the reference drew 2 (D, G) x 2 steps before the checkpoint
    train = train_fns.GAN_training_function(G, D, GD, ema, state, CFG, samp, embedded_optimizers=False,
                                            device=device, batch_size=gb)
    x, y, f = synth.synth_batch(CFG, gb, seed=102)
    state["itr"] += 1
    G.train(); D.train(); G_ema.train()
    m = train(x.to(device), y.to(device), f.to(device))
    got = np.array([m["G_loss"], m["D_loss_real"], m["D_loss_fake"]])
    np.testing.assert_allclose(got, GOLD["losses_after"][0], rtol=2e-4, atol=2e-4)
    g = {k: GOLD[k] for k in GOLD.files}
    # Adam with beta1 = 0 moves a parameter whose gradient is rounding noise by +-lr (tests/helpers.adam_slack)
    # (the CPU emulator needs none; the GPU's different fp32 summation order flips the sign of such noise gradients:
    # 2.2 lr for biases (a conv bias that feeds a BatchNorm has zero true gradient), 0.5 lr elsewhere — the policy of
    # tests/test_parity_gpu.py)
    slack_g = {n: slack * (2.2 if n.endswith(".bias") else 0.5) * CFG["G_lr"] for n in G.state_dict()}
    slack_d = {n: slack * (2.2 if n.endswith(".bias") else 0.5) * CFG["D_lr"] for n in D.state_dict()}
    check_group(g, "after/G_state/", G.state_dict(), rtol=STATE_RTOL, atol=1e-6, what="G ", extra_atol=slack_g)
    check_group(g, "after/D_state/", D.state_dict(), rtol=STATE_RTOL, atol=1e-6, what="D ", extra_atol=slack_d)
    check_group(g, "after/EMA_state/", G_ema.state_dict(), rtol=STATE_RTOL, atol=1e-6, what="EMA ", extra_atol=slack_g)


def test_samples_from_reference_checkpoint_cpu(emu):
    _check_samples("cpu")


def test_resume_from_reference_checkpoint_cpu(emu):
    _check_resume("cpu", 0.0)


@pytest.mark.gpu
def test_samples_from_reference_checkpoint_gpu():
    _check_samples("cuda:0")


@pytest.mark.gpu
def test_resume_from_reference_checkpoint_gpu():
    _check_resume("cuda:0", 1.0)


def test_save_writes_reference_layout(tmp_path):
    """our save_weights -> same files, key names, shapes, dtypes and Adam-state structure as the reference's."""
    M, G, D, G_ema, og, od, state, ema = _build("cpu")
    from ic_gan_amd import utils
    utils.save_weights(G, D, state, str(tmp_path), "exp", "copy3", G_ema, embedded_optimizers=False, G_optim=og, D_optim=od)
    ref_dir = os.path.join(GOLDEN_DIR, NAME)
    for stem in ("G", "D", "G_ema", "G_optim", "D_optim", "state_dict"):
        ours = torch.load(str(tmp_path / "exp" / ("%s_copy3.pth" % stem)), map_location="cpu", weights_only=False)
        ref = torch.load(os.path.join(ref_dir, stem + ".pth"), map_location="cpu", weights_only=False)
        if stem.endswith("optim"):
            assert ours["param_groups"][0]["params"] == ref["param_groups"][0]["params"]
            for key in ("lr", "betas", "eps", "weight_decay"):
                assert ours["param_groups"][0][key] == ref["param_groups"][0][key], key
            assert sorted(ours["state"]) == sorted(ref["state"])
            for i, st in ref["state"].items():
                assert set(st) <= set(ours["state"][i]), (i, set(st), set(ours["state"][i]))
                assert float(ours["state"][i]["step"]) == float(st["step"])
                for key in ("exp_avg", "exp_avg_sq"):
                    assert torch.equal(ours["state"][i][key], st[key]), (i, key)
            # and plain torch.optim.Adam accepts it (the reference's optimizer class, trainer.py:158-171)
            params = [torch.nn.Parameter(torch.zeros_like(p)) for p in (G if stem[0] == "G" else D).parameters()]
            torch.optim.Adam(params, lr=1e-3).load_state_dict(ours)
        elif stem == "state_dict":
            assert ours["itr"] == ref["itr"] == 2 and ours["config"] == ref["config"]
        else:
            assert list(ours) == list(ref)
            for k in ref:
                assert ours[k].dtype == ref[k].dtype and torch.equal(ours[k], ref[k]), k


def test_ddp_prefixed_checkpoint_needs_prefix(tmp_path):
    """a checkpoint saved from DDP-wrapped modules has `module.`-prefixed keys; it loads into a wrapped
    model and is rejected (strict) by a bare one — same behaviour as the reference."""
    M, G, D, G_ema, og, od, state, ema = _build("cpu")
    sd = {"module." + k: v for k, v in G.state_dict().items()}
    with pytest.raises(RuntimeError):
        G.load_state_dict(sd, strict=True)
    holder = torch.nn.Module()
    holder.module = G
    holder.load_state_dict(sd, strict=True)


def test_load_model_inference_picks_best_fid(tmp_path, emu):
    from ic_gan_amd import inference
    src = os.path.join(GOLDEN_DIR, NAME)
    exp = tmp_path / "exp"
    exp.mkdir()
    for suffix, fid in (("best0", 31.0), ("best1", 12.5)):
        for stem in ("G", "D", "G_ema", "G_optim", "D_optim"):
            shutil.copy(os.path.join(src, stem + ".pth"), str(exp / ("%s_%s.pth" % (stem, suffix))))
        sd = torch.load(os.path.join(src, "state_dict.pth"), weights_only=False)
        sd["best_FID"] = fid
        sd["config"] = dict(sd["config"], skip_init=True, no_optim=True)
        torch.save(sd, str(exp / ("state_dict_%s.pth" % suffix)))
    # make best1's EMA weights recognisable
    w = torch.load(str(exp / "G_ema_best1.pth"), weights_only=False)
    w["shared.weight"] = w["shared.weight"] + 1.0
    torch.save(w, str(exp / "G_ema_best1.pth"))
    config = dict(weights_root=str(tmp_path), experiment_name="exp", model_backbone="biggan", use_ema=True, ema=True,
                  G_eval_mode=True, batch_size=3, seed=99, resolution=999)
    gen, cfg = inference.load_model_inference(config, device="cpu")
    assert cfg["load_weights"] == "best1" and cfg["resolution"] == 64 and cfg["batch_size"] == 3 and cfg["seed"] == 99
    assert not gen.training and not cfg["feature_augmentation"]
    assert torch.equal(gen.state_dict()["shared.weight"], w["shared.weight"])
    z, y, f = (torch.from_numpy(GOLD["sample/" + k]) for k in ("z", "y", "feats"))
    img, _, _ = inference.sample(gen, lambda: (z, y, f), cfg, class_cond=True, instance_cond=True, device="cpu")
    assert img.shape == (6, 3, 64, 64) and torch.isfinite(img).all()
    with pytest.raises(ValueError):
        inference.load_model_inference(dict(config, experiment_name="missing"), device="cpu")


@pytest.mark.gpu
def test_graphed_generator_matches_eager():
    """HIP-graph replay of the eval forward == eager eval forward, for several inputs and after a weight change."""
    from ic_gan_amd import inference
    M, G, D, G_ema, og, od, state, ema = _build("cuda:0")
    G_ema.eval()
    gg = inference.GraphedGenerator(G_ema, 6, class_cond=True, instance_cond=True, device="cuda:0", feature_dim=2048)
    z, y, f = (torch.from_numpy(GOLD["sample/" + k]).cuda() for k in ("z", "y", "feats"))
    for scale in (1.0, 0.5):
        with torch.no_grad():
            ref = G_ema(z * scale, y, f)
        got = gg(z * scale, y, f)
        assert torch.equal(got, ref)
    assert _rel_l2(gg(z, y, f).cpu().numpy(), GOLD["sample/G_ema"]) < 1e-3
    with torch.no_grad():
        G_ema.linear.weight.mul_(1.01)               # weights are read at replay time
        ref = G_ema(z, y, f)
    assert torch.equal(gg(z, y, f), ref)
    img, _, _ = inference.sample(gg, lambda: (z.cpu(), y.cpu(), f.cpu()), CFG, class_cond=True, instance_cond=True, device="cuda:0")
    assert torch.equal(img, ref)
    # static_weights: W/sigma baked in at capture; refresh() picks up new weights
    gs = inference.GraphedGenerator(G_ema, 6, class_cond=True, instance_cond=True, device="cuda:0", static_weights=True)
    assert torch.equal(gs(z, y, f), ref)
    with torch.no_grad():
        G_ema.linear.weight.mul_(0.99)
        ref2 = G_ema(z, y, f)
    assert torch.equal(gs(z, y, f), ref) and not torch.equal(ref, ref2)
    gs.refresh()
    assert torch.equal(gs(z, y, f), ref2)
