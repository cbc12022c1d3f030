#!/usr/bin/env python3
"""Cook-Toom / Winograd F(m, 3) matrices from a set of interpolation points, in exact rational arithmetic, and an fp32 error
study of the 2-D forms against the float64 direct sum.  CPU only (numpy); no device code depends on it.

  python tools/wino_matrices.py --print 4 0,1,-1,2,-1/2        # A^T, G, B^T of F(4,3) for those points (+ infinity)
  python tools/wino_matrices.py --study                         # the table of profiles/r02_study_winograd_f4_numerics.txt

Why: the trunk's F(3x3,3x3) (points 0, 1, -1, 2, inf; csrc/winograd.hip) spends 25 multiplies per 9 outputs.  F(4x4,3x3) spends 36
per 16 (-19 % GEMM FLOPs, -19 % V / M plane bytes) but is usually dismissed for fp32 because the textbook points (0, +-1, +-2)
treble the rounding error.  The study shows that the error depends on the POINT SET, not on the tile size: with the asymmetric set
{0, 1, -1, 2, -1/2} (or {0, 1, -1, 1/2, -2}) F(4x4,3x3) has the error of the F(3x3,3x3) in use today."""
import argparse
import sys
from fractions import Fraction as Fr

import numpy as np


def _polymul(a, b):
    out = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def matrices(m, points, r=3):
    """F(m, r) from len(points) == m + r - 2 finite points plus the point at infinity.  Returns (AT, G, BT) as lists of
    Fractions: y = AT @ ((G @ g) * (BT @ d)) for d of m + r - 1 samples and an r-tap filter g (correlation, as Caffe)."""
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1, "need m + r - 2 distinct finite points"
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * r for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    for i, p in enumerate(pts):
        Ni = Fr(1)
        poly = [Fr(1)]
        for j, q in enumerate(pts):
            if j != i:
                Ni *= (p - q)
                poly = _polymul(poly, [-q, Fr(1)])
        for k in range(m):
            AT[k][i] = p ** k
        for k in range(r):
            G[i][k] = p ** k / Ni
        for k in range(n - 1):
            BT[i][k] = poly[k]
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][r - 1] = Fr(1)
    poly = [Fr(1)]
    for q in pts:
        poly = _polymul(poly, [-q, Fr(1)])
    for k in range(n):
        BT[n - 1][k] = poly[k]
    return AT, G, BT


def as_float(M):
    return np.array([[float(x) for x in row] for row in M])


def identity_holds(m, points):
    """The 1-D identity in exact arithmetic on integer data."""
    AT, G, BT = matrices(m, points)
    n = m + 2
    d = [Fr(v) for v in (3, -1, 4, 1, -5, 9, 2, -6)[:n]]
    g = [Fr(2), Fr(-7), Fr(5)]
    U = [sum(G[i][k] * g[k] for k in range(3)) for i in range(n)]
    V = [sum(BT[i][k] * d[k] for k in range(n)) for i in range(n)]
    y = [sum(AT[o][i] * U[i] * V[i] for i in range(n)) for o in range(m)]
    return y == [sum(d[o + k] * g[k] for k in range(3)) for o in range(m)]


def fp32_error(m, points, Cin, Cout, tiles, dist, seed=1):
    """max |y - truth| / max(1, |truth|) of the 2-D form with every stage rounded to fp32 (filter transform in float64 at pack
    time, like csrc/winograd.hip) and of the fp32 direct sum, on `tiles` tiles of `Cin` channels."""
    AT, G, BT = (as_float(M) for M in matrices(m, points))
    n = m + 2
    rng = np.random.default_rng(seed)
    if dist == "relu":
        x = np.maximum(rng.standard_normal((Cin, tiles, n, n)), 0)
    elif dist == "lognormal":
        x = np.exp(rng.standard_normal((Cin, tiles, n, n)) * 1.5) * (rng.random((Cin, tiles, n, n)) < 0.5)
    elif dist == "dc":
        x = np.maximum(rng.standard_normal((Cin, tiles, n, n)) + 3.0, 0) * 10
    else:
        raise ValueError(dist)
    w = rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))
    x32, w32 = x.astype(np.float32), w.astype(np.float32)
    truth = np.zeros((Cout, tiles, m, m))
    dir32 = np.zeros((Cout, tiles, m, m), np.float32)
    for kh in range(3):
        for kw in range(3):
            truth += np.einsum("oc,ctij->otij", w32[:, :, kh, kw].astype(np.float64), x32[:, :, kh:kh + m, kw:kw + m].astype(np.float64))
            dir32 += np.einsum("oc,ctij->otij", w32[:, :, kh, kw], x32[:, :, kh:kh + m, kw:kw + m]).astype(np.float32)
    U = np.einsum("ik,ockl,jl->ocij", G, w32.astype(np.float64), G).astype(np.float32)
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    V = np.einsum("ik,ctkl->ctil", BT32, x32).astype(np.float32)
    V = np.einsum("ctil,jl->ctij", V, BT32).astype(np.float32)
    M = np.einsum("ocij,ctij->otij", U, V).astype(np.float32)
    Y = np.einsum("ik,otkl->otil", AT32, M).astype(np.float32)
    Y = np.einsum("otil,jl->otij", Y, AT32).astype(np.float32)
    rel = lambda a: float((np.abs(a - truth) / np.maximum(1, np.abs(truth))).max())      # noqa: E731
    return rel(Y), rel(dir32)


FORMS = [("F(3x3,3x3) {0,1,-1,2}  (in use)", 3, [0, 1, -1, 2]),
         ("F(4x4,3x3) {0,1,-1,2,-2}", 4, [0, 1, -1, 2, -2]),
         ("F(4x4,3x3) {0,1,-1,1/2,-1/2}", 4, [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
         ("F(4x4,3x3) {0,1,-1,2,-1/2}", 4, [0, 1, -1, 2, Fr(-1, 2)]),
         ("F(4x4,3x3) {0,1,-1,1/2,-2}", 4, [0, 1, -1, Fr(1, 2), -2])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--print", nargs=2, metavar=("M", "POINTS"))
    ap.add_argument("--study", action="store_true")
    a = ap.parse_args()
    if a.print:
        m, pts = int(a.print[0]), [Fr(p) for p in a.print[1].split(",")]
        AT, G, BT = matrices(m, pts)
        for name, M in (("A^T", AT), ("G", G), ("B^T", BT)):
            print(name)
            for row in M:
                print("  ", "  ".join(f"{str(v):>6s}" for v in row))
        print("identity holds:", identity_holds(m, pts))
    if a.study:
        print("# max |y - truth| / max(1, |truth|), every stage rounded to fp32, Cout 64, 200 tiles; truth = float64 direct sum")
        print(f"# {'distribution':10s} {'Cin':>4s} {'form':34s} {'Winograd':>10s} {'direct fp32':>12s}")
        for dist in ("relu", "lognormal", "dc"):
            for Cin in (256, 512):
                for name, m, pts in FORMS:
                    e, ed = fp32_error(m, pts, Cin, 64, 200, dist)
                    print(f"  {dist:10s} {Cin:4d} {name:34s} {e:10.2e} {ed:12.2e}", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
