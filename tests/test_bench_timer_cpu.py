"""bench.py's kernel timer (host logic, no GPU): stratified sampling -- every launch of an (entry point, shape) key is counted, every
`period`-th is bracketed, a key's time is (mean bracketed launch) x (launch count) -- and the roofline assembly on top of it."""
import importlib
import sys
import os

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Ev:
    """stand-in for a pair of torch.cuda.Event: elapsed_time in milliseconds"""

    def __init__(self, ms=0.0):
        self.ms = ms

    def elapsed_time(self, other):
        return other.ms - self.ms


@pytest.fixture
def bench():
    return importlib.import_module("bench")


def test_records_scale_the_mean_bracketed_launch_by_the_launch_count(bench):
    t = bench.KernelTimer(period=4)
    # key A: 8 launches, 2 bracketed (1.0 and 3.0 ms) -> 8 x 2.0 ms; key B: 1 launch, bracketed (5 ms)
    t.keys[("icg_conv2d_fprop", 1)] = ["kern A", 10.0, 4.0, 100.0, 8, [(_Ev(0), _Ev(1.0)), (_Ev(0), _Ev(3.0))]]
    t.keys[("icg_conv2d_fprop", 2)] = ["kern B", 7.0, 7.0, 50.0, 1, [(_Ev(0), _Ev(5.0))]]
    t.keys[("icg_conv2d_fprop", 3)] = ["kern C", 1.0, 1.0, 1.0, 3, []]            # counted, never bracketed: carries no time
    recs = t.records
    assert len(recs) == 9 and sum(1 for r in recs if r[0] == "kern A") == 8
    agg = t.summary()
    a, b = agg["kern A"], agg["kern B"]
    assert a[2] == 8 and abs(a[1] - 8 * 2.0e-3) < 1e-12 and a[0] == 80.0 and a[3] == 32.0 and a[4] == 800.0
    assert b[2] == 1 and abs(b[1] - 5.0e-3) < 1e-12
    assert "kern C" not in agg


def test_hbm_records_and_roofline(bench):
    t = bench.KernelTimer(period=2)
    t.hbm_keys[("icg_bias_act", 1)] = ["bias_act (icg_bias_act)", 8.0e9, 4, [(_Ev(0), _Ev(2.0))]]
    r = t.hbm_roofline()
    assert r["launches"] == 4 and abs(r["achieved"] - 8.0e9 / 2.0e-3 / 1e9) < 1e-6 and r["bound"] == "hbm"
    assert abs(r["frac"] - r["achieved"] / bench.PEAK_HBM_GBPS) < 1e-4


def test_period_one_is_exact_and_period_is_clamped(bench):
    assert bench.KernelTimer(period=0).period == 1
    t = bench.KernelTimer(period=1)
    t.keys[("k", 0)] = ["kern", 1.0, 1.0, 1.0, 3, [(_Ev(0), _Ev(1.0)), (_Ev(0), _Ev(2.0)), (_Ev(0), _Ev(6.0))]]
    assert abs(t.summary()["kern"][1] - 9.0e-3) < 1e-12          # mean 3 ms x 3 launches = the plain sum


def test_traffic_hash_ignores_sources_the_cfg3_step_does_not_launch(bench, tmp_path, monkeypatch):
    """csrc_sha256 (traffic_stale) covers the kernel sources of the cfg3 step only: touching the StyleGAN2 / kNN sources keeps it"""
    import shutil
    root = os.path.dirname(os.path.abspath(bench.__file__))
    dst = tmp_path / "repo"
    shutil.copytree(os.path.join(root, "ic_gan_amd", "csrc"), dst / "ic_gan_amd" / "csrc",
                    ignore=shutil.ignore_patterns("obj", "*.o"))
    monkeypatch.setattr(bench, "__file__", str(dst / "bench.py"))
    h0 = bench.csrc_sha256()
    with open(dst / "ic_gan_amd" / "csrc" / "hconv.hip", "a") as f:
        f.write("// touched\n")
    assert bench.csrc_sha256() == h0
    with open(dst / "ic_gan_amd" / "csrc" / "pgemm.hip", "a") as f:
        f.write("// touched\n")
    assert bench.csrc_sha256() != h0


def test_self_launch_command_is_the_drivers_torchrun_form(bench, monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE re-executes itself under torch.distributed.run (reference: mp.spawn in
    BigGAN_PyTorch/trainer.py:70-75): one node, N ranks, rendezvous on 127.0.0.1, the original arguments passed through."""
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "3", "--workload", "cfg1"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--workload", "cfg1"]
    assert 1024 < bench.free_port() < 65536
    # the branch is taken only without a launcher: with WORLD_SIZE set main() must not re-launch
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda n: called.append(n))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--workload", "cfg1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.main()
    assert called == [2]
