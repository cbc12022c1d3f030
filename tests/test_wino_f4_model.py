"""CPU model behind the next Winograd step: F(4x4,3x3) spends 36 multiplies per 16 outputs where the trunk's
F(3x3,3x3) spends 25 per 9 (-19 % GEMM FLOPs and V / M plane bytes), and its fp32 rounding error is a property of the
interpolation POINTS, not of the tile size.  tools/wino_matrices.py derives the matrices in exact rational arithmetic; here:
(1) it reproduces the matrices the device code uses today (csrc/winograd.hip), (2) the Cook-Toom identity holds exactly for every
candidate set, (3) with the asymmetric sets {0, 1, -1, 2, -1/2} / {0, 1, -1, 1/2, -2} the fp32 error of F(4x4,3x3) is that of
today's F(3x3,3x3), while the textbook symmetric sets are 1.5-3x worse.  No device code depends on this yet."""
import importlib.util
import os
from fractions import Fraction as Fr

import numpy as np
import pytest

_spec = importlib.util.spec_from_file_location(
    "wino_matrices", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "wino_matrices.py"))
wm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(wm)


def test_generator_reproduces_the_device_matrices_of_f3x3():
    AT, G, BT = wm.matrices(3, [0, 1, -1, 2])
    assert BT == [[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]]          # winograd.hip bt5
    assert G == [[Fr(1, 2), 0, 0], [Fr(-1, 2)] * 3, [Fr(-1, 6), Fr(1, 6), Fr(-1, 6)], [Fr(1, 6), Fr(1, 3), Fr(2, 3)], [0, 0, 1]]
    assert AT == [[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]]


@pytest.mark.parametrize("name,m,pts", wm.FORMS)
def test_cook_toom_identity_is_exact(name, m, pts):
    assert wm.identity_holds(m, pts)
    AT, G, BT = wm.matrices(m, pts)
    assert len(AT) == m and len(G) == m + 2 and len(BT) == m + 2 and all(len(r) == m + 2 for r in BT)


def test_f4x4_with_asymmetric_points_has_the_error_of_todays_f3x3():
    err = {name: wm.fp32_error(m, pts, Cin=128, Cout=32, tiles=48, dist="relu")[0] for name, m, pts in wm.FORMS}
    direct = wm.fp32_error(3, [0, 1, -1, 2], Cin=128, Cout=32, tiles=48, dist="relu")[1]
    f3 = err["F(3x3,3x3) {0,1,-1,2}  (in use)"]
    assert direct < f3 < 1e-4                                          # Winograd pays ~10x the direct sum's error, inside the bar
    for good in ("F(4x4,3x3) {0,1,-1,2,-1/2}", "F(4x4,3x3) {0,1,-1,1/2,-2}"):
        assert err[good] <= 1.5 * f3, (good, err[good], f3)
    for textbook in ("F(4x4,3x3) {0,1,-1,2,-2}", "F(4x4,3x3) {0,1,-1,1/2,-1/2}"):
        assert err[textbook] >= 1.3 * min(err["F(4x4,3x3) {0,1,-1,2,-1/2}"], err["F(4x4,3x3) {0,1,-1,1/2,-2}"]), (textbook, err)


def test_device_transform_functions_match_the_exact_matrices(tmp_path):
    """csrc/wino_f4_math.h (bt6 / g6 / at6: the straight-line code the F(4x4,3x3) kernels of csrc/winograd.hip call) compiled
    for the host and probed with unit vectors: it must apply exactly 2 B^T, G / 2 and A^T of the generator, and a full 2-D tile
    through the three functions must equal the direct 3x3 correlation."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "probe.cpp"
    src.write_text(r'''
#include <cstdio>
#include "wino_f4_math.h"
int main() {
  using namespace wino_f4;
  for (int k = 0; k < 6; ++k) { float d[6] = {0, 0, 0, 0, 0, 0}, r[6]; d[k] = 1.f; bt6(d, r); for (int i = 0; i < 6; ++i) printf("%.9g ", r[i]); printf("\n"); }
  for (int k = 0; k < 3; ++k) { float g[3] = {0, 0, 0}, u[6]; g[k] = 1.f; g6(g, u); for (int i = 0; i < 6; ++i) printf("%.9g ", u[i]); printf("\n"); }
  for (int k = 0; k < 6; ++k) { float m[6] = {0, 0, 0, 0, 0, 0}, y[4]; m[k] = 1.f; at6(m, y); for (int i = 0; i < 4; ++i) printf("%.9g ", y[i]); printf("\n"); }
  // one 2-D tile: d 6x6, g 3x3 -> y 4x4
  float d[6][6], g[3][3], U[6][6], V[6][6], M[6][6], y[4][4], t[6][6];
  unsigned s = 7u;
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { s = s * 1664525u + 1013904223u; d[i][j] = (float)((int)(s >> 10) % 201 - 100) / 32.f; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { s = s * 1664525u + 1013904223u; g[i][j] = (float)((int)(s >> 10) % 201 - 100) / 64.f; }
  for (int j = 0; j < 6; ++j) { float c[6], r[6]; for (int i = 0; i < 6; ++i) c[i] = d[i][j]; bt6(c, r); for (int i = 0; i < 6; ++i) t[i][j] = r[i]; }
  for (int i = 0; i < 6; ++i) bt6(t[i], V[i]);
  float tg[6][3];
  for (int j = 0; j < 3; ++j) { float c[3] = {g[0][j], g[1][j], g[2][j]}, u[6]; g6(c, u); for (int i = 0; i < 6; ++i) tg[i][j] = u[i]; }
  for (int i = 0; i < 6; ++i) g6(tg[i], U[i]);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) M[i][j] = U[i][j] * V[i][j];
  float ta[4][6];
  for (int j = 0; j < 6; ++j) { float c[6], o[4]; for (int i = 0; i < 6; ++i) c[i] = M[i][j]; at6(c, o); for (int i = 0; i < 4; ++i) ta[i][j] = o[i]; }
  for (int i = 0; i < 4; ++i) at6(ta[i], y[i]);
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    double ref = 0; for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) ref += (double)d[i + a][j + b] * g[a][b];
    printf("%.9g %.9g\n", y[i][j], ref);
  }
  return 0;
}
''')
    exe = tmp_path / "probe"
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "mscnn_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [[float(v) for v in line.split()] for line in subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip().split("\n")]
    AT, G, BT = (wm.as_float(M) for M in wm.matrices(4, [0, 1, -1, 2, Fr(-1, 2)]))
    assert np.allclose(np.array(rows[0:6]).T, 2 * BT, rtol=0, atol=1e-7)          # probe row k = column k of the matrix
    assert np.allclose(np.array(rows[6:9]).T, G / 2, rtol=1e-6, atol=1e-8)
    assert np.allclose(np.array(rows[9:15]).T, AT, rtol=0, atol=1e-7)
    tile = np.array(rows[15:])
    assert tile.shape == (16, 2) and np.abs(tile[:, 0] - tile[:, 1]).max() < 2e-5 * max(1.0, np.abs(tile[:, 1]).max())
