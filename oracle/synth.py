"""TEST INFRASTRUCTURE: deterministic synthetic weights and inputs.

numpy ``RandomState`` streams are platform-stable, so the same (name, shape)
list yields bit-identical tensors in the build container (where the reference
is run to produce goldens) and on the GPU box (where only the oracle and the
HIP path exist).  Input conventions follow BASELINE.md §3.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch


def _rs(seed: int, name: str) -> np.random.RandomState:
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    rs = _rs(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    shape = tuple(shape)
    if leaf == "sv0":
        a = np.ones(shape)
    elif leaf == "u0":
        a = rs.standard_normal(shape)
    elif leaf == "stored_mean":
        a = 0.1 * rs.standard_normal(shape)
    elif leaf == "stored_var":
        a = 1.0 + 0.2 * rs.uniform(size=shape)
    elif leaf == "gamma":
        a = np.full(shape, 0.6)           # reference inits 0.0, which would hide the attention path
    elif leaf == "gain" and len(shape) == 1:
        a = 1.0 + 0.1 * rs.standard_normal(shape)
    elif leaf == "bias":
        a = 0.1 * rs.standard_normal(shape)
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = rs.standard_normal(shape) / np.sqrt(max(fan_in, 1))
        if name.startswith("shared.") or name.startswith("embed."):
            a = rs.standard_normal(shape)
    else:
        raise KeyError(f"no synthesis rule for state key {name!r}")
    return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape).copy())


def synth_state(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    return {n: synth_tensor(n, tuple(s), seed) for n, s in spec}


def spec_of(state_dict) -> List[Tuple[str, Tuple[int, ...]]]:
    return [(k, tuple(v.shape)) for k, v in state_dict.items()]


def synth_batch(cfg: dict, batch: int, seed: int, n_classes: int = None):
    """Real-side batch: x uint8->[-1,1], labels, unit-norm 2048-d features."""
    rs = np.random.RandomState(seed)
    r = cfg["resolution"]
    n_classes = n_classes or cfg.get("n_classes", 1000)
    u8 = rs.randint(0, 256, size=(batch, 3, r, r)).astype(np.float32)
    x = torch.from_numpy(((u8 / 255.0) - 0.5) * 2.0).float()
    y = torch.from_numpy(rs.randint(0, n_classes, size=(batch,)).astype(np.int64))
    f = rs.standard_normal((batch, 2048))
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    return x, (y if cfg.get("class_cond", True) else None), \
        (torch.from_numpy(f).float() if cfg.get("instance_cond", False) else None)


class CondSampler:
    """Seeded stand-in for ``sample_conditionings()`` (train_fns.py:70): returns
    (z[, labels][, feats]) in the reference's tuple order (data_utils/utils.py:877-901)."""

    def __init__(self, cfg: dict, dim_z: int, batch: int, seed: int):
        self.cfg, self.dim_z, self.batch = cfg, dim_z, batch
        self.rs = np.random.RandomState(seed)

    def __call__(self):
        z = torch.from_numpy(self.rs.standard_normal((self.batch, self.dim_z)).astype(np.float32))
        cc, ic = self.cfg.get("class_cond", True), self.cfg.get("instance_cond", False)
        lab = torch.from_numpy(self.rs.randint(0, self.cfg.get("n_classes", 1000),
                                               size=(self.batch,)).astype(np.int64))
        f = self.rs.standard_normal((self.batch, 2048))
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        f = torch.from_numpy(f).float()
        if cc and ic:
            return z, lab, f
        if cc:
            return z, lab
        if ic:
            return z, f
        return z
