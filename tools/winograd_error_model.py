#!/usr/bin/env python
"""Where the fp32 error of Winograd F(4x4,3x3) comes from -- a CPU model in numpy (no GPU needed, ~1 min).

  1. interpolation points: Toom-Cook matrices in exact rationals for {0, 1, -1, a, b, inf} (the 25-plane form of the
     resample-fused layers needs -1 in the set), fp32 transforms + sequential fp32 accumulation over C channels, error vs fp64.
  2. precision of the stages: fp32 everywhere / fp64 transforms with fp32 storage / fp64 transforms AND exact accumulation.

Findings (profiles/r01_winograd_error_model.txt): the shipped points {0, +-1, +-2, inf} are within 1.5x of the best pair found
({-1/2, 2}); fp64 transform arithmetic buys 1.1-1.3x; exact accumulation of the U.V products brings the error down to the
direct convolution's (3.3e-7): the error is the fp32 accumulation of large, cancelling products in the plane GEMMs, so a
two-level (blocked) accumulation there is the lever (model: ~3x at 32-channel blocks)."""
import itertools
from fractions import Fraction as Fr

import numpy as np


def toom_cook(points, m=4, r=3):
    """F(m, r) matrices (A^T [m x n], G [n x r], B^T [n x n]) for n-1 finite points + infinity:  y = A^T [(G g) .* (B^T d)]"""
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1
    AT = [[pts[j] ** i for j in range(n - 1)] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    G = []
    for j in range(n - 1):
        N = Fr(1)
        for l in range(n - 1):
            if l != j:
                N *= pts[j] - pts[l]
        G.append([pts[j] ** k / N for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for k, y in enumerate(b):
                out[i + k] += x * y
        return out

    BT = []
    for j in range(n):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-pts[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    return tuple(np.array([[float(x) for x in row] for row in M]) for M in (AT, G, BT))


def identity_error(points):
    AT, G, BT = toom_cook(points)
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(3), rng.standard_normal(6)
    ref = np.array([sum(d[i + k] * g[k] for k in range(3)) for i in range(4)])
    return np.abs(AT @ ((G @ g) * (BT @ d)) - ref).max()


def tile_error(points, C, mode="f32", block=0, trials=6, seed=1):
    """relative L2 error of one 4x4 output tile summed over C channels.  mode: f32 | t64 (fp64 transforms, fp32 storage) |
    t64acc64 (+ exact accumulation); block > 0: fp32 accumulation in blocks of `block` channels, block sums added in fp32"""
    AT, G, BT = toom_cook(points)
    tdt = np.float32 if mode == "f32" else np.float64
    rng = np.random.default_rng(seed)
    es = []
    for _ in range(trials):
        g = (rng.standard_normal((C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
        d = np.maximum(rng.standard_normal((C, 6, 6)), 0).astype(np.float32)              # post-ReLU activations
        ref = np.array([[np.sum(d[:, i:i + 3, j:j + 3].astype(np.float64) * g.astype(np.float64)) for j in range(4)]
                        for i in range(4)])
        U = np.einsum("ir,crs,js->cij", G.astype(tdt), g.astype(tdt), G.astype(tdt)).astype(np.float32)
        V = np.einsum("ir,crs,js->cij", BT.astype(tdt), d.astype(tdt), BT.astype(tdt)).astype(np.float32)
        if mode == "t64acc64":
            M = np.einsum("cij,cij->ij", U.astype(np.float64), V.astype(np.float64))
        else:
            M = np.zeros((6, 6), np.float32)
            part = np.zeros((6, 6), np.float32)
            for c in range(C):
                part = (part + U[c] * V[c]).astype(np.float32)
                if block and (c + 1) % block == 0:
                    M = (M + part).astype(np.float32)
                    part = np.zeros((6, 6), np.float32)
            M = (M + part).astype(np.float32)
        Y = (AT.astype(tdt) @ M.astype(tdt) @ AT.T.astype(tdt)).astype(np.float32)
        es.append(np.linalg.norm(Y - ref) / np.linalg.norm(ref))
    return float(np.mean(es))


def direct_error(C, trials=6, seed=1):
    rng = np.random.default_rng(seed)
    es = []
    for _ in range(trials):
        g = (rng.standard_normal((C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
        d = np.maximum(rng.standard_normal((C, 6, 6)), 0).astype(np.float32)
        ref, y = np.zeros((4, 4)), np.zeros((4, 4), np.float32)
        for i in range(4):
            for j in range(4):
                ref[i, j] = np.sum(d[:, i:i + 3, j:j + 3].astype(np.float64) * g.astype(np.float64))
                acc = np.float32(0)
                for v in (d[:, i:i + 3, j:j + 3] * g).reshape(-1):
                    acc = np.float32(acc + v)
                y[i, j] = acc
        es.append(np.linalg.norm(y - ref) / np.linalg.norm(ref))
    return float(np.mean(es))


if __name__ == "__main__":
    base = [0, 1, -1, 2, -2]
    print("identity check of the shipped points:", identity_error(base))
    print("\n-- stages (shipped points), relative L2 error of a 4x4 tile")
    for C in (96, 384, 1536):
        print(f"C={C:5d}  direct fp32 {direct_error(C):.2e} | winograd fp32 {tile_error(base, C):.2e} | fp64 transforms "
              f"{tile_error(base, C, 't64'):.2e} | + exact accumulation {tile_error(base, C, 't64acc64'):.2e} | fp32 with "
              f"32-channel blocks {tile_error(base, C, block=32):.2e} | 64-channel blocks {tile_error(base, C, block=64):.2e}")
    print("\n-- interpolation points {0, 1, -1, a, b, inf}, C = 128")
    cands = [Fr(1, 2), Fr(-1, 2), Fr(2), Fr(-2), Fr(3), Fr(-3), Fr(1, 3), Fr(-1, 3), Fr(3, 2), Fr(-3, 2), Fr(2, 3), Fr(-2, 3),
             Fr(1, 4), Fr(-1, 4), Fr(4), Fr(-4), Fr(3, 4), Fr(-3, 4), Fr(4, 3), Fr(-4, 3)]
    res = []
    for a, b in itertools.combinations(cands, 2):
        pts = [0, 1, -1, a, b]
        if identity_error(pts) < 1e-9:
            res.append((tile_error(pts, 128, trials=4), str(a), str(b)))
    res.sort()
    print("shipped {2, -2}:", f"{tile_error(base, 128, trials=4):.2e}")
    for e, a, b in res[:8]:
        print(f"  {{{a}, {b}}}: {e:.2e}")
    print(f"  worst {{{res[-1][1]}, {res[-1][2]}}}: {res[-1][0]:.2e}")
