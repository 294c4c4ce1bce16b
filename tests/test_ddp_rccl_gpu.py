"""Multi-rank data parallelism ON THE HIP KERNELS over RCCL.

  * 1 rank, always runnable on the 1-GPU box: the cross-replica BN path (icg_bn_sync_pack, asynchronous all-reduce of the
    packed [sum x | sum x^2 | n] payload on RCCL's stream, device-side element count) forced on over a single-rank RCCL
    group must reproduce the local-statistics path.
  * min(2, device_count) ranks, skipped on a 1-GPU box: replicas stay BIT-IDENTICAL after two full G+D steps under the
    reference's DDP wiring (trainer.py:196-210), and N-rank SyncBN == 1 process on the concatenated batch (the
    oracle for cross-replica BN), both on the production kernels.  The driver's multi-GPU tier runs these."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_workers(mode, n, tmp_path, port, shared_gpu=False):
    out = str(tmp_path / f"{mode}.pt")
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONDONTWRITEBYTECODE": "1"}
    if shared_gpu:
        env.update(ICG_TEST_BACKEND="gloo", ICG_TEST_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_rccl_worker.py"), mode, out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return torch.load(out, weights_only=False)


def _nranks():
    return min(2, torch.cuda.device_count())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (the driver's multi-GPU tier)")
def test_ddp_rccl_replicas_bit_identical_after_steps(tmp_path):
    out = _run_workers("step", _nranks(), tmp_path, 29541)
    assert out["world"] >= 2 and out["finite"] and out["identical"], out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (the driver's multi-GPU tier)")
def test_syncbn_rccl_ranks_equal_one_process_on_concatenated_batch(tmp_path):
    from tests import ddp_rccl_worker as W
    out = _run_workers("syncbn", _nranks(), tmp_path, 29542)
    cfg = dict(W.CFG, sync_bn=False)
    _, G, _ = W.models(cfg, "cuda:0")
    z, lab, fg, wts = W.syncbn_inputs(cfg, G.dim_z)
    G.train()
    img = G(z.cuda(), lab.cuda(), fg.cuda())
    ((img * wts.cuda()).sum() / out["world"]).backward()          # DDP averages the ranks' gradients
    grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()]).cpu()
    assert torch.allclose(out["img"], img.detach().cpu(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(out["rm"], G.blocks[0][0].bn1.stored_mean.cpu(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(out["rv"], G.output_layer[0].stored_var.cpu(), rtol=1e-5, atol=1e-6)
    rel = float((out["grads"] - grads).norm() / grads.norm())
    assert rel < 1e-4, rel


def _check_syncbn(out):
    from tests import ddp_rccl_worker as W
    cfg = dict(W.CFG, sync_bn=False)
    _, G, _ = W.models(cfg, "cuda:0")
    z, lab, fg, wts = W.syncbn_inputs(cfg, G.dim_z)
    G.train()
    img = G(z.cuda(), lab.cuda(), fg.cuda())
    ((img * wts.cuda()).sum() / out["world"]).backward()          # DDP averages the ranks' gradients
    grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()]).cpu()
    assert torch.allclose(out["img"], img.detach().cpu(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(out["rm"], G.blocks[0][0].bn1.stored_mean.cpu(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(out["rv"], G.output_layer[0].stored_var.cpu(), rtol=1e-5, atol=1e-6)
    rel = float((out["grads"] - grads).norm() / grads.norm())
    assert rel < 1e-4, rel


def test_two_ranks_sharing_one_gpu_replicas_bit_identical(tmp_path):
    """2 processes, both on GPU 0, gloo transport (RCCL does not allow two ranks per device): the production kernels under the
    reference's DDP wiring with a real second rank -- buffer broadcast, bucketed gradient all-reduce, find_unused_parameters."""
    out = _run_workers("step", 2, tmp_path, 29544, shared_gpu=True)
    assert out["world"] == 2 and out["finite"] and out["identical"], out


def test_two_ranks_sharing_one_gpu_syncbn_equals_one_process(tmp_path):
    """... and cross-replica BN across the two processes (packed payload, asynchronous all-reduce, device-side count) against one
    process on the concatenated batch."""
    out = _run_workers("syncbn", 2, tmp_path, 29545, shared_gpu=True)
    assert out["world"] == 2
    _check_syncbn(out)


@pytest.mark.parametrize("extra", [[], ["--sync-bn"]])
def test_bench_script_two_ranks_sharing_one_gpu(extra):
    """the N > 1 path of bench.py (DDP wrapping, barrier + synchronize bracket, MAX over ranks, whole-job value, rank-0 JSON line) run
    as the driver launches it, with both ranks on GPU 0 over gloo (ICG_BENCH_SHARED_GPU: a test switch, not a measurement)."""
    import json
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONDONTWRITEBYTECODE": "1", "ICG_BENCH_SHARED_GPU": "1"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29546", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload",
           "cfg1", "--init", "N02"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert d["config"]["rccl_world_size"] == 2 and d["config"]["sync_bn"] == bool(extra)
    comm = d["config"]["comm"]
    assert comm["comm_savings"] is True and "exposed_ms_per_step" in comm and comm["G"]["buckets_per_step"] > 0
    assert ("sync_bn_leg" in comm) == (not extra)                        # the plain run reports a cross-replica-BN leg beside it
    if not extra:
        assert comm["sync_bn_leg"]["ms_per_step"] > 0 and comm["sync_bn_leg"]["bn_layers"] > 10
    host = d["config"]["rank_host_resources"]
    assert host is None or (host["cores"] >= 1 and host["intra_op_threads"] >= 1)
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-2 * d["value"]
    assert all(v == v for v in d["config"]["losses_last_step"].values())          # finite losses
    assert "cpu_baseline" not in d


def test_bench_script_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script starts its ranks itself, as the
    reference's trainer does with mp.spawn (BigGAN_PyTorch/trainer.py:70-75); one JSON line from rank 0, world size 2."""
    import json
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONDONTWRITEBYTECODE": "1", "ICG_BENCH_SHARED_GPU": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg1",
           "--init", "N02"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_world_size"] == 2 and d["config"]["global_batch"] == 16
    assert d["config"]["comm"] is not None and d["config"]["comm"]["G"]["allreduce_bytes_per_step"] > 0
    assert "cpu_baseline" not in d


def test_syncbn_path_single_rank_rccl_matches_local_statistics(monkeypatch):
    """world_size 1 over RCCL with the cross-replica path forced on: pack kernel, async all-reduce (a no-op sum), device-side
    count in finalize / backward coefficients -- against the plain path on the same inputs (fp64 payload algebra: equal to
    ~1e-6 relative, not bitwise: the sums travel un-shifted)."""
    import torch.distributed as dist
    from tests import ddp_rccl_worker as W
    import ic_gan_amd.ops as ops
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = dict(W.CFG)
        _, G0, _ = W.models(dict(cfg, sync_bn=False), "cuda:0")
        _, G1, _ = W.models(dict(cfg, sync_bn=True), "cuda:0")
        z, lab, fg, wts = W.syncbn_inputs(cfg, G0.dim_z)
        G0.train(); G1.train()
        a = G0(z.cuda(), lab.cuda(), fg.cuda())
        (a * wts.cuda()).sum().backward()
        calls = {"n": 0}
        real = dist.all_reduce

        def counting(*args, **kw):
            calls["n"] += 1
            return real(*args, **kw)

        monkeypatch.setattr(ops, "_sync_enabled", lambda bn: bn is not None and bn.sync_group is not None)
        monkeypatch.setattr(dist, "all_reduce", counting)
        b = G1(z.cuda(), lab.cuda(), fg.cuda())
        (b * wts.cuda()).sum().backward()
        torch.cuda.synchronize()
        n_bn = sum(1 for m in G1.modules() if hasattr(m, "stored_mean"))
        assert calls["n"] == 2 * n_bn, (calls, n_bn)                  # one packed message per BN layer and direction
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        for (n0, p0), (_, p1) in zip(G0.named_parameters(), G1.named_parameters()):
            rel = float((p0.grad - p1.grad).norm() / (p0.grad.norm() + 1e-20))
            assert rel < 1e-4, (n0, rel)
        for (k, v0), (_, v1) in zip(G0.state_dict().items(), G1.state_dict().items()):
            assert torch.allclose(v0, v1, rtol=1e-5, atol=1e-6), k
    finally:
        dist.destroy_process_group()
