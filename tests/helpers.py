"""Shared helpers for the parity tests (golden loading, fingerprints)."""
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NS = 64
CASES = ["cc_ic_r64", "ic_r64_acc2", "cc_r32_flat", "cc_ic_r128", "cc_ic_r256"]


def load_golden(case):
    z = np.load(os.path.join(GOLDEN_DIR, f"biggan_{case}.npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["cfg"] = json.loads(str(g["cfg"]))
    g["gspec"] = [(n, tuple(s)) for n, s in json.loads(str(g["gspec"]))]
    g["dspec"] = [(n, tuple(s)) for n, s in json.loads(str(g["dspec"]))]
    return g


def fingerprint(t):
    t = t.detach().double().flatten().cpu()
    n = t.numel()
    stride = max(n // NS, 1)
    s = t[::stride][:NS]
    samp = np.zeros(NS)
    samp[: s.numel()] = s.numpy()
    return float(t.sum()), float((t * t).sum()), samp


def check_group(gold, prefix, tensors, rtol, atol, what=""):
    """Compare a dict name->tensor with the packed fingerprints stored under `prefix`."""
    names = json.loads(str(gold[prefix + "names"]))
    assert set(names) == set(tensors.keys()), (what, sorted(set(names) ^ set(tensors.keys()))[:8])
    worst = (0.0, None)
    for i, n in enumerate(names):
        s, q, samp = fingerprint(tensors[n])
        gs, gq, gsamp = gold[prefix + "sum"][i], gold[prefix + "sq"][i], gold[prefix + "samp"][i]
        scale = float(np.sqrt(gq / max(tensors[n].numel(), 1))) + 1e-12     # rms of the golden tensor
        err = float(np.max(np.abs(samp - gsamp))) / (scale + atol / max(rtol, 1e-30))
        if err > worst[0]:
            worst = (err, n)
        assert np.allclose(samp, gsamp, rtol=rtol, atol=atol + rtol * scale), \
            f"{what}{n}: samples differ, max abs {np.max(np.abs(samp - gsamp)):.3e} (rms {scale:.3e})"
        assert abs(q - gq) <= 4 * rtol * abs(gq) + atol, f"{what}{n}: sumsq {q} vs {gq}"
        n_el = max(tensors[n].numel(), 1)
        assert abs(s - gs) <= 4 * rtol * scale * n_el ** 0.5 * 8 + atol * n_el, f"{what}{n}: sum {s} vs {gs}"
    return worst
