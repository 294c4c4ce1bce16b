"""Shared helpers for the parity tests (golden loading, fingerprints)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NS = 64
CASES = ["cc_ic_r64", "ic_r64_acc2", "cc_r32_flat", "cc_ic_r128", "cc_ic_r256"]
# BASELINE.json's configurations at their REAL widths (tests/golden/make_golden_real_widths.py): configs[0] exactly
# (64x64, ch 64, B 8, 2 steps), configs[1] / configs[2] at ch 96 with the batch cut to 2
REAL_CASES = ["cfg1_icgan_res64", "cfg2_w96_r128", "cfg3_w96_r256"]
# ... and configs[2] at batch sizes where the benchmark's routes are taken (BN statistics over >= 16 images, split-K slice
# counts, persistent tile runs, 0.25 - 1 GB attention maps): B = 16 with an fp64 twin, and the benchmark's own B = 64
BENCH_CASES = ["cfg3_w96_r256_b16", "cfg3_w96_r256_b64"]


def load_golden(case):
    z = np.load(os.path.join(GOLDEN_DIR, f"biggan_{case}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["cfg"] = json.loads(str(g["cfg"]))
    g["gspec"] = [(n, tuple(s)) for n, s in json.loads(str(g["gspec"]))]
    g["dspec"] = [(n, tuple(s)) for n, s in json.loads(str(g["dspec"]))]
    return g


def fingerprint(t, ns=NS):
    """(sum, sum of squares, `ns` strided samples) -- the generator scripts' fingerprint (tests/golden/make_golden.py); `ns` is
    read off the golden group being compared (64 for state groups, 4096 for the gradients of the real-width cases: every
    gradient of <= 4096 elements is then compared in full)."""
    t = t.detach().double().flatten().cpu()
    n = t.numel()
    stride = max(n // ns, 1)
    s = t[::stride][:ns]
    samp = np.zeros(ns)
    samp[: s.numel()] = s.numpy()
    return float(t.sum()), float((t * t).sum()), samp


def noise_grad_names(gold, prefix, rel=1e-3):
    """Parameters whose golden gradient is rounding noise (mathematically zero, e.g. a conv bias feeding a
    BatchNorm).  Adam with beta1 = 0 turns that noise into +-lr updates, so their post-step values are only
    defined up to ~lr per step; callers pass them to check_group(extra_atol=...)."""
    names = json.loads(str(gold[prefix + "names"]))
    sq, samp = gold[prefix + "sq"], gold[prefix + "samp"]
    rms = [float(np.sqrt(sq[i] / max(np.count_nonzero(samp[i]), 1))) for i in range(len(names))]
    top = max(rms) if rms else 0.0
    return {n for n, r in zip(names, rms) if r < rel * top}


SOFT_REPORT = None      # tools/parity_report.py sets this to a list: failures are recorded as (label, err/tol) instead of raised
STATS = None            # ... and this to a list: (group label, tensor name, golden rms, tolerance, samples) of every compared tensor
DENSE_REPORT = None     # ... and this: (tensor, sub-grid max / tol, max / tol, rms difference / tol, share of samples above tol)


def _fail(cond, msg, ratio):
    if cond:
        return
    if SOFT_REPORT is not None:
        SOFT_REPORT.append((msg, ratio))
    else:
        raise AssertionError(msg)


DENSE_EXCEED_SHARE = 1.0 / 64     # dense groups: share of a tensor's samples that may exceed the per-element tolerance ...
# round 5: GRAD_RTOL 1.5e-2 -> 1e-2 (below); the two dense OUTLIER bounds keep their absolute size (x 1.5 in units of the new tolerance):
# they bound isolated decision flips -- what an fp32 ReLU / max-pool network does to itself (profiles/r03_reference_sensitivity.txt) -- and the
# arithmetic behind them is pinned two orders of magnitude tighter with the fp64 forward's decisions imposed, at cfg1, cfg2 AND the
# headline cfg3 (tests/test_decision_replay_gpu.py: rms <= 1e-4 / 2e-4, max <= 2e-3 of the tensor rms)
DENSE_MAX_MULT = 4.875            # ... by at most this factor (= 3.25 x 1.5e-2 / 1e-2; measured worst: profiles/r05_parity_report.txt),
DENSE_RMS_FRAC = 0.45             # ... while the rms of the differences stays below this fraction of the tolerance (= 0.3 x 1.5)


def check_group(gold, prefix, tensors, rtol, atol, what="", noise_floor=2e-3, extra_atol=None):
    """Compare a dict name->tensor with the packed fingerprints stored under `prefix`.

    Per tensor the tolerance is  atol + rtol * max(rms(golden), noise_floor * max rms in the group):
    tensors that are pure rounding noise relative to their siblings (e.g. the gradient of a conv bias
    that feeds a BatchNorm, mathematically zero) are compared on the group's scale, not their own.

    64-sample groups (every state group; the gradients of the small goldens): max |difference| <= tolerance.
    DENSE groups (4096 samples per tensor: the gradients of the real-width and bench-configuration cases):
      (1) the 64-sample sub-grid (every 64th sample = the positions the 64-sample fingerprints hold) is held to the same
          max criterion as before, and over ALL samples
      (2) the rms of the differences <= DENSE_RMS_FRAC x tolerance,
      (3) at most DENSE_EXCEED_SHARE of the samples exceed the tolerance (the exceedance rate a 64-sample max test admits), and
      (4) none by more than DENSE_MAX_MULT x.
    Why not a plain max over 4096 samples: two fp32 evaluations of a ReLU / max-pool network differ in ISOLATED elements by
    far more than their rms difference -- units whose pre-activation lies within rounding of zero take the other branch, which
    moves single weight-gradient elements by a few per cent of the tensor rms when the gradient sums over few pixels.  The
    unmodified reference does this to itself: with its weights perturbed by 1e-6 (relative) its own cfg2 gradients move by
    2e-3 rms with maxima of 7e-2 of the tensor rms, max / rms = 34 (tools/ref_sensitivity.py ->
    profiles/r03_reference_sensitivity.txt); the HIP path against the goldens shows the same figures (2.4e-3 / 8.5e-2 / 35)."""
    names = json.loads(str(gold[prefix + "names"]))
    assert set(names) == set(tensors.keys()), (what, sorted(set(names) ^ set(tensors.keys()))[:8])
    scales = []
    for i, n in enumerate(names):
        scales.append(float(np.sqrt(gold[prefix + "sq"][i] / max(tensors[n].numel(), 1))))
    floor = noise_floor * (max(scales) if scales else 0.0)
    worst = (0.0, None)
    for i, n in enumerate(names):
        gs, gq, gsamp = gold[prefix + "sum"][i], gold[prefix + "sq"][i], gold[prefix + "samp"][i]
        ns = gsamp.shape[0]
        s, q, samp = fingerprint(tensors[n], ns)
        n_el = max(tensors[n].numel(), 1)
        scale = max(scales[i], floor)
        tol = atol + rtol * scale + (extra_atol or {}).get(n, 0.0)
        diff = np.abs(samp - gsamp)
        if STATS is not None:
            STATS.append((what, n, scales[i], tol, samp))
        if ns <= NS:
            err = float(diff.max())
            ratio = err / tol
            _fail(err <= tol, f"{what}{n}: samples differ, max abs {err:.3e} > tol {tol:.3e} (rms {scales[i]:.3e})", ratio)
        else:
            k = min(n_el, ns)                                   # samples that exist (short tensors are stored in full, zero-padded)
            sub = float(diff[:: ns // NS].max())                # (1) the positions of the 64-sample fingerprint
            rms_err = float(np.sqrt((diff[:k] ** 2).sum() / k))
            share = float((diff[:k] > tol).mean())
            err = float(diff.max())
            ratio = max(sub / tol, rms_err / (DENSE_RMS_FRAC * tol), share / DENSE_EXCEED_SHARE, err / (DENSE_MAX_MULT * tol))
            _fail(sub <= tol, f"{what}{n}: 64-sample sub-grid differs, max abs {sub:.3e} > tol {tol:.3e} (rms {scales[i]:.3e})", sub / tol)
            _fail(rms_err <= DENSE_RMS_FRAC * tol, f"{what}{n}: rms difference {rms_err:.3e} > {DENSE_RMS_FRAC} x tol {tol:.3e}",
                  rms_err / (DENSE_RMS_FRAC * tol))
            _fail(share <= DENSE_EXCEED_SHARE, f"{what}{n}: {share:.4f} of {k} samples exceed tol {tol:.3e}", share / DENSE_EXCEED_SHARE)
            _fail(err <= DENSE_MAX_MULT * tol, f"{what}{n}: max abs {err:.3e} > {DENSE_MAX_MULT} x tol {tol:.3e} (rms {scales[i]:.3e})",
                  err / (DENSE_MAX_MULT * tol))
            if DENSE_REPORT is not None:
                DENSE_REPORT.append((what + n, sub / tol, err / tol, rms_err / tol, share))
        if ratio > worst[0]:
            worst = (ratio, n)
        _fail(abs(np.sqrt(q / n_el) - np.sqrt(gq / n_el)) <= 4 * tol, f"{what}{n}: rms {q} vs {gq}",
              abs(np.sqrt(q / n_el) - np.sqrt(gq / n_el)) / (4 * tol))
        _fail(abs(s - gs) <= 32 * tol * n_el ** 0.5 + atol * n_el, f"{what}{n}: sum {s} vs {gs}",
              abs(s - gs) / (32 * tol * n_el ** 0.5 + atol * n_el))
    if SOFT_REPORT is not None and worst[1] is not None:
        SOFT_REPORT.append((f"{what}worst={worst[1]}", worst[0]))
    return worst


GRAD_RTOL = 1e-2       # of the tensor rms (round 5: 1.5e-2 -> 1e-2); the goldens' own fp32-vs-fp64 conditioning is <= 3.5e-3 (B = 4 cases)
STATE_RTOL = 5e-3
COND_K = 4.0           # real-width cases: + COND_K x |fp32 reference - fp64 reference| per tensor (see conditioning_slack)


def conditioning_slack(case, grad_prefix):
    """Per-tensor absolute slack = COND_K x the distance between the reference's fp32 gradient and the SAME reference run
    in fp64 (tests/golden/biggan_<case>_f64.npz, make_golden_real_widths.py::run_case_f64), maximum over the fingerprint
    samples.  At the real widths with batch 2 the fp32 reference itself is up to 0.8 x (GRAD_RTOL * rms) away from the
    exact gradient on the first generator layers (cfg3: blocks.0.0.conv1.weight 0.82, linear.bias 0.71); no fp32
    implementation with another summation order can be held to less than a small multiple of that.  {} when the case has
    no fp64 companion file (the small-width goldens: plain GRAD_RTOL)."""
    path = os.path.join(GOLDEN_DIR, f"biggan_{case}_f64.npz")
    if not os.path.exists(path):
        return {}
    g32, g64 = load_golden(case), np.load(path, allow_pickle=False)
    names = json.loads(str(g32[grad_prefix + "names"]))
    assert names == json.loads(str(g64[grad_prefix + "names"]))
    d = np.abs(g32[grad_prefix + "samp"] - g64[grad_prefix + "samp"]).max(axis=1)
    return {n: COND_K * float(d[i]) for i, n in enumerate(names)}


def adam_slack(gold, grad_prefix, lr, steps, names):
    """Per-name absolute slack for post-Adam values.  With beta1 = 0 the update is lr*g/(|g|+eps): elements whose
    gradient is within rounding noise of eps move by an O(lr) amount that is not reproducible across summation
    orders.  0.5*lr per step for ordinary tensors (isolated tiny-|g| elements), 2.2*lr for tensors whose whole
    gradient is noise (mathematically zero)."""
    noise = noise_grad_names(gold, grad_prefix)
    return {n: (2.2 if n in noise else 0.5) * lr * steps for n in names}
