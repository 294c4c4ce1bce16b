"""CPU, world_size = 2 over gloo: the data-parallel wiring of the hot path.

The HIP kernels cannot run here, so every C-ABI call is routed to oracle/kernel_ref.py inside the spawned
workers (test-only); what is under test is the distributed *host* logic:
  * DDP gradient averaging + buffer broadcast keep G/D replicas bit-identical after a full G+D step
  * sync_bn=True: 2 ranks on half batches == 1 process on the concatenated batch (outputs, running statistics,
    DDP-averaged gradients) — the oracle for cross-replica BN named in SURVEY F2
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kernel_ref, synth

CFG = dict(dim_z=24, shared_dim=16, shared_dim_feat=32, G_shared=True, G_shared_feat=True, hier=True, n_classes=10,
           SN_eps=1e-6, BN_eps=1e-5, adam_eps=1e-6, G_ch=8, D_ch=8, G_attn="16", D_attn="16", resolution=32,
           class_cond=True, instance_cond=True, toggle_grads=True, num_D_steps=1, num_D_accumulations=1,
           num_G_accumulations=1, split_D=False, DiffAugment="", DA=False, D_ortho=0.0, G_ortho=0.0, ema=True,
           ema_decay=0.9, ema_start=0)


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kernel_ref.install(_Patch())
    torch.set_num_threads(2)


def _models(cfg):
    import ic_gan_amd.BigGAN as M
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    G.load_state_dict(synth.synth_state(synth.spec_of(G.state_dict()), 11))
    D.load_state_dict(synth.synth_state(synth.spec_of(D.state_dict()), 22))
    return M, G, D


class _Traffic:
    """state of a DDP communication hook that counts what a wrapper all-reduces (elements and buckets)"""

    def __init__(self):
        self.elements, self.buckets = 0, 0


def _traffic_hook(state, bucket):
    """... and otherwise does what the default hook does (average over the ranks)"""
    buf = bucket.buffer()
    state.elements += buf.numel()
    state.buckets += 1
    buf.div_(dist.get_world_size())
    return dist.all_reduce(buf, async_op=True).get_future().then(lambda f: f.value()[0])


def _worker_step(rank, world, port, out, savings=True, accumulations=1, steps=1, sn_group=None, prefetch=False):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from ic_gan_amd import ops, train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    _init(rank, world, port)
    train_fns.COMM_SAVINGS = savings
    train_fns.PREFETCH_NEXT_STEP = prefetch
    if sn_group is not None:
        ops.SN_BACKWARD_GROUP = sn_group
    groups = []
    many = ops.sn_backward_many
    ops.sn_backward_many = lambda items: (groups.append(len(items)), many(items))[1]
    CFG = dict(globals()["CFG"], num_D_accumulations=accumulations, num_G_accumulations=accumulations)
    M, G, D = _models(CFG)
    if rank == 1:                       # perturb rank 1's buffers: the per-forward broadcast must overwrite them (F3)
        with torch.no_grad():
            G.linear.u0.add_(1.0)
    G_ema = M.Generator(**{**CFG, "skip_init": True, "no_optim": True})
    ema = utils.ema(G, G_ema, 0.9, 0)
    opt_d = FusedAdam(D.parameters(), lr=2e-3, betas=(0.0, 0.999), eps=1e-6)
    opt_g = FusedAdam(G.parameters(), lr=1e-3, betas=(0.0, 0.999), eps=1e-6)
    Gd, Dd = DDP(G, find_unused_parameters=True), DDP(D, find_unused_parameters=True)
    tg, td = _Traffic(), _Traffic()
    Gd.register_comm_hook(tg, _traffic_hook)
    Dd.register_comm_hook(td, _traffic_hook)
    GD = M.G_D(Gd, Dd, optimizer_G=opt_g, optimizer_D=opt_d)
    gb = 2
    train = train_fns.GAN_training_function(Gd, Dd, GD, ema, {"itr": 1}, CFG, synth.CondSampler(CFG, G.dim_z, gb, 50 + rank),
                                            embedded_optimizers=False, device="cpu", batch_size=gb)
    x, y, f = synth.synth_batch(CFG, gb * accumulations, seed=70 + rank)
    Gd.train(); Dd.train()
    for _ in range(steps):
        m = train(x, y, f)
    flat = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    bufs = lambda: torch.cat([b.detach().reshape(-1).float() for b in list(G.buffers()) + list(D.buffers())])
    bg = [torch.empty_like(bufs()) for _ in range(world)]
    dist.all_gather(bg, bufs())
    # utils.save_weights holds no collective (the reference saves under `rank == 0` only, trainer.py:520): rank 0 writes FIRST, alone,
    # while rank 1 does not enter save_weights at all -- this would hang on gloo if a broadcast hid in there ...
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        if rank == 0:
            utils.save_weights(Gd, Dd, {"itr": 1}, tmp, "exp", "alone", None, embedded_optimizers=False, G_optim=opt_g, D_optim=opt_d)
            alone = torch.load("%s/exp/G_alone.pth" % tmp)
            out["rank0_checkpoint_is_rank0_buffers"] = all(bool(torch.equal(alone["module." + k], b)) for k, b in G.named_buffers())
        # ... and a checkpoint written from ANY rank carries rank 0's buffers once every rank has called utils.sync_buffers
        for wrapped in (Gd, Dd):
            utils.sync_buffers(wrapped)
        utils.save_weights(Gd, Dd, {"itr": 1}, tmp, "exp", "rank%d" % rank, None, embedded_optimizers=False, G_optim=opt_g, D_optim=opt_d)
        saved = torch.load("%s/exp/G_rank%d.pth" % (tmp, rank))
    ckpt = torch.cat([saved["module." + k].reshape(-1).float() for k, _ in G.named_buffers()])
    cg = [torch.empty_like(ckpt) for _ in range(world)]
    dist.all_gather(cg, ckpt)
    bs = [torch.empty_like(bufs()) for _ in range(world)]
    dist.all_gather(bs, bufs())
    if rank == 0:
        out["buffers_identical_after_step"] = all(bool(torch.equal(bg[0], g)) for g in bg[1:])
        out["buffers_identical_after_sync"] = all(bool(torch.equal(bs[0], g)) for g in bs[1:])
        out["buffers_rank0_unchanged_by_sync"] = bool(torch.equal(bs[0], bg[0]))
        out["checkpoints_carry_rank0_buffers"] = all(bool(torch.equal(cg[0], g)) for g in cg[1:])
        out["identical"] = all(bool(torch.equal(gathered[0], g)) for g in gathered[1:])
        out["sn_groups"] = list(groups)
        out["finite"] = bool(torch.isfinite(flat).all())
        out["loss"] = m
        out["flat"] = flat.clone()
        out["traffic"] = {"G": tg.elements, "D": td.elements, "G_params": sum(p.numel() for p in G.parameters()),
                          "D_params": sum(p.numel() for p in D.parameters())}
    dist.destroy_process_group()


def _worker_syncbn(rank, world, port, out, split=(2, 2), skew=False):
    from torch.nn.parallel import DistributedDataParallel as DDP
    _init(rank, world, port)
    cfg = dict(CFG, sync_bn=True)
    _, G, _ = _models(cfg)
    if skew and rank == 1:
        # replicas whose running means differ (broadcast_buffers=False, per-rank checkpoint loads: ADVICE r1): the packed
        # payload is taken to a common origin before the all-reduce, so the normalisation must not depend on them
        with torch.no_grad():
            for m in G.modules():
                if hasattr(m, "stored_mean"):
                    m.stored_mean.add_(0.37)
    Gd = DDP(G, broadcast_buffers=not skew)
    B = sum(split)
    c = synth.CondSampler(cfg, G.dim_z, B, 9)()
    z, lab, fg = c
    wts = torch.from_numpy(__import__("numpy").random.RandomState(3).standard_normal((B, 3, 32, 32))).float()
    lo = sum(split[:rank])
    sl = slice(lo, lo + split[rank])          # possibly UNEQUAL shards: the element count travels in the payload
    Gd.train()
    img = Gd(z[sl], lab[sl], fg[sl])
    (img * wts[sl]).sum().backward()
    grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()])
    full = torch.zeros(B, *img.shape[1:])
    full[sl] = img.detach()
    dist.all_reduce(full)
    if rank == 0:
        out["img"] = full
        out["grads"] = grads.clone()
        out["rm"] = G.blocks[0][0].bn1.stored_mean.clone()
        out["rv"] = G.output_layer[0].stored_var.clone()
    dist.destroy_process_group()


def _spawn(fn, *extra, world=2):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(fn, args=(world, _free_port(), out) + extra, nprocs=world, join=True)
    return dict(out)


@pytest.mark.timeout(600)
def test_ddp_step_keeps_replicas_identical():
    out = _spawn(_worker_step)
    assert out["finite"] and out["identical"], out


@pytest.mark.timeout(900)
def test_ddp_g_phase_does_not_allreduce_the_frozen_discriminator():
    """D's buckets must not be all-reduced in the G phase (toggle_grads froze D; 400 MB per step at cfg3 if they were): per step
    D's traffic is exactly ONE pass over its parameters and G's one pass over G's.  Measured here with a counting
    communication hook: torch 2.10's reducer already skips a wrapper none of whose parameters produced a gradient (no autograd
    hook fires, so the lazily marked 'unused' parameters are never shipped), i.e. the reference's wiring (COMM_SAVINGS = False)
    has the same traffic; train_fns additionally runs the frozen D under no_sync() so that this does not hinge on the reducer's
    internals.  Either way the step's result is bit-identical and the replicas stay identical."""
    lean, ref = _spawn(_worker_step, True), _spawn(_worker_step, False)
    assert lean["finite"] and lean["identical"] and ref["identical"]
    t, r = lean["traffic"], ref["traffic"]
    assert t["D"] == t["D_params"] and t["G"] == t["G_params"], t
    assert r["D"] == r["D_params"] and r["G"] == r["G_params"], r
    assert lean["loss"] == ref["loss"]
    assert torch.equal(lean["flat"], ref["flat"])


@pytest.mark.timeout(900)
def test_ddp_accumulation_allreduces_once_per_phase():
    """the shipped cfg3 schedule accumulates 4 x 16 images per step (cc_icgan_res256.json:22-24): with COMM_SAVINGS the first
    rounds accumulate locally (no_sync) and ONE all-reduce per phase carries the sum; the reference pattern all-reduces every
    round.  Same result up to the rounding of the averaged sum."""
    lean, ref = _spawn(_worker_step, True, 2), _spawn(_worker_step, False, 2)
    assert lean["finite"] and lean["identical"] and ref["identical"]
    t, r = lean["traffic"], ref["traffic"]
    assert t["D"] == t["D_params"] and t["G"] == t["G_params"], t
    assert r["D"] == 2 * r["D_params"] and r["G"] == 2 * r["G_params"], r   # reference pattern: one all-reduce per round
    for k in lean["loss"]:
        assert abs(lean["loss"][k] - ref["loss"][k]) <= 1e-5 * (1 + abs(ref["loss"][k])), (k, lean["loss"], ref["loss"])
    rel = float((lean["flat"] - ref["flat"]).norm() / ref["flat"].norm())
    assert rel < 1e-5, rel


@pytest.mark.timeout(900)
def test_ddp_buffers_after_accumulation():
    """COMM_SAVINGS runs D in the G phase and the early accumulation rounds under no_sync(): DDP then skips its rank-0 buffer
    broadcast at the next forward, so after a step the replicas' BUFFERS (running statistics, u / sv: each rank's own shard) may
    differ although every parameter is bit-identical; utils.sync_buffers re-aligns them on rank 0's values."""
    out = _spawn(_worker_step, True, 2)
    assert out["identical"] and out["finite"]
    assert out["buffers_identical_after_sync"] and out["buffers_rank0_unchanged_by_sync"] and out["checkpoints_carry_rank0_buffers"]
    assert out["rank0_checkpoint_is_rank0_buffers"]
    ref = _spawn(_worker_step, False, 2)           # the reference pattern: every forward synchronises
    assert ref["identical"] and ref["buffers_identical_after_sync"]


@pytest.mark.timeout(900)
def test_ddp_grouped_spectral_norm_backward():
    """From the second step on the spectral-norm backward of 8 consecutive layers runs in ONE autograd node (ops.SNGroupFn): under
    DistributedDataParallel the parameter gradients of a group then reach the reducer together.  Two steps on two ranks: the groups
    are really used, the replicas stay bit-identical, and the result equals the per-layer path (SN_BACKWARD_GROUP = 0) bit for bit."""
    grouped, single = _spawn(_worker_step, True, 1, 2, 8), _spawn(_worker_step, True, 1, 2, 0)
    assert grouped["finite"] and grouped["identical"] and single["identical"]
    assert grouped["sn_groups"] and max(grouped["sn_groups"]) <= 8 and not single["sn_groups"]
    assert grouped["loss"] == single["loss"]
    assert torch.equal(grouped["flat"], single["flat"])
    t = grouped["traffic"]
    assert t["D"] == 2 * t["D_params"] and t["G"] == 2 * t["G_params"], t          # one pass over each network's parameters per step


@pytest.mark.timeout(900)
def test_ddp_prefetched_next_step_opening_keeps_the_run():
    """train_fns.PREFETCH_NEXT_STEP under DistributedDataParallel: the generator forward of the next step's first D accumulation is
    issued at the end of the current step on every rank (the wrapper's rank-0 buffer broadcast with it, in the same order on all ranks).
    Two steps on two ranks: replicas bit-identical, and the same parameters as without the prefetch."""
    ahead = _spawn(_worker_step, True, 1, 2, None, True)
    plain = _spawn(_worker_step, True, 1, 2, None, False)
    assert ahead["finite"] and ahead["identical"] and plain["identical"]
    assert ahead["loss"] == plain["loss"]
    assert torch.equal(ahead["flat"], plain["flat"])


def test_ddp_step_world_size_4():
    out = _spawn(_worker_step, True, 1, world=4)
    assert out["finite"] and out["identical"], out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("split,skew", [((2, 2), False), ((3, 1), False), ((2, 2), True), ((1, 3, 2, 2), False)])
def test_syncbn_two_ranks_equal_one_process_on_full_batch(monkeypatch, split, skew):
    """(also world_size 4 with an uneven 1 + 3 + 2 + 2 split of an 8-image batch)"""
    out = _spawn(_worker_syncbn, split, skew, world=len(split))
    kernel_ref.install(monkeypatch)
    cfg = dict(CFG, sync_bn=False)
    _, G, _ = _models(cfg)
    B = sum(split)
    z, lab, fg = synth.CondSampler(cfg, G.dim_z, B, 9)()
    import numpy as np
    wts = torch.from_numpy(np.random.RandomState(3).standard_normal((B, 3, 32, 32))).float()
    G.train()
    img = G(z, lab, fg)
    ((img * wts).sum() / len(split)).backward()          # DDP averages the ranks' gradients
    grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()])
    assert torch.allclose(out["img"], img.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(out["rm"], G.blocks[0][0].bn1.stored_mean, rtol=1e-5, atol=1e-6)      # rank 0's buffers
    assert torch.allclose(out["rv"], G.output_layer[0].stored_var, rtol=1e-5, atol=1e-6)
    rel = float((out["grads"] - grads).norm() / grads.norm())
    assert rel < 1e-4, rel


def _worker_stylegan2(rank, world, port, out):
    """StyleGAN2 iteration under DDP (mapping / synthesis / D wrapped separately like training_loop.py:293-310), two
    accumulation rounds per phase so that the no_sync path (loss.py run_G / run_D `sync` argument) is exercised."""
    import copy
    from torch.nn.parallel import DistributedDataParallel as DDP
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan2.training_step import TrainingStep
    from tests.stylegan_cases import SG2_LOSS, SG2_NETS, SG2_OPT, sg2_inputs, sg2_state
    _init(rank, world, port)
    cfg = SG2_NETS["cc_ic_r16_resnetG"]

    def load(m, seed):
        spec = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        cur = m.state_dict()
        m.load_state_dict({k: (cur[k] if v is None else v) for k, v in sg2_state(spec, seed).items()})

    G = N.Generator(**cfg["G"]).train().requires_grad_(False)
    D = N.Discriminator(**cfg["D"]).train().requires_grad_(False)
    load(G, 1); load(D, 2)
    G_ema = copy.deepcopy(G).eval()
    mods = {}
    for name, m in (("G_mapping", G.mapping), ("G_synthesis", G.synthesis), ("D", D)):
        m.requires_grad_(True)
        mods[name] = DDP(m, broadcast_buffers=False)
        m.requires_grad_(False)
    step = TrainingStep(G, D, G_ema, "cpu", batch_size=8, batch_gpu=2, num_gpus=world, loss_kwargs=SG2_LOSS,
                        G_opt_kwargs=SG2_OPT, D_opt_kwargs=SG2_OPT, G_reg_interval=4, D_reg_interval=16, ema_kimg=0.02,
                        ddp_modules=mods)
    torch.manual_seed(1234 + rank)
    z, gc, gh, img, rc, rh = sg2_inputs({**cfg, "batch": 4}, 40 + rank, 4)
    ran = step(img, rc, rh, z, gc, gh)
    flat = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out["ran"] = ran
        out["identical"] = bool(torch.equal(gathered[0], gathered[1]))
        out["finite"] = bool(torch.isfinite(flat).all())
        start = torch.cat([v.reshape(-1) for k, v in sg2_state([[k, list(v.shape)] for k, v in G.state_dict().items()], 1).items()
                           if v is not None and k in dict(G.named_parameters())])
        out["moved"] = bool((flat[: start.numel()] - start).abs().max() > 0)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_stylegan2_ddp_iteration_keeps_replicas_identical():
    out = _spawn(_worker_stylegan2)
    assert out["ran"] == ["Gmain", "Greg", "Dmain", "Dreg"]
    assert out["finite"] and out["identical"] and out["moved"], out
