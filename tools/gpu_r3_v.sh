#!/bin/bash
# round 3, call V: BigGAN-deep toy-network step with / without the dense-layer routing onto the row-streaming GEMM
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_biggan_deep.py -m gpu -q -p no:cacheprovider > gpurun_out/r3v_on.log 2>&1; echo "on rc=$?"; grep -E "AssertionError|passed|failed" gpurun_out/r3v_on.log | tail -4 | cut -c1-250
ICG_SMALLM_DENSE=0 timeout 300 python -m pytest tests/test_biggan_deep.py -m gpu -q -p no:cacheprovider > gpurun_out/r3v_off.log 2>&1; echo "off rc=$?"; grep -E "AssertionError|passed|failed" gpurun_out/r3v_off.log | tail -4 | cut -c1-250
cat > /tmp/dense_probe.py <<'PY'
import torch, sys, os, itertools
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ic_gan_amd._lib as L
worst = 0
for M, K, N in itertools.product((1, 3, 4, 16, 17, 64, 100, 256), (16, 24, 40, 64, 72, 128, 657, 1000), (1, 3, 4, 6, 8, 24, 96, 130)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    out = torch.full((M, N), float("nan"), device="cuda")
    L.call("icg_conv2d_fprop", x, w, b, None, out, None, None, 0, M, 1, 1, K, N, 1, 0, 0.5)
    ref = 0.5 * (x.double() @ w.double().t()) + b.double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, err)
    if not (err < 1e-5): print("BAD", M, K, N, err)
print("worst rel err", worst)
PY
timeout 300 python /tmp/dense_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
