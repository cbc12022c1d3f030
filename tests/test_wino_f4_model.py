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
