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


_LAYER_HARNESS = r'''
// Runs a whole 3x3 / stride 1 layer through the per-thread bodies of the F(4x4,3x3) kernels (wino_f4_math.h) on the host: the loops
// below stand in for the grid, the plane GEMMs read the packed weight layout the way the 1x1 igemm kernel addresses it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wino_f4_math.h"
int main(int argc, char** argv) {
  const int N = atoi(argv[1]), Cin = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), Cout = atoi(argv[5]), pad = atoi(argv[6]), relu = atoi(argv[7]);
  const int BM = 128, CK = 32, MT = (Cout + BM - 1) / BM, KI = (Cin + CK - 1) / CK;
  const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2, th = (Ho + 3) / 4, tw = (Wo + 3) / 4, T = N * th * tw, T_pad = (T + 127) / 128 * 128;
  std::vector<float> x((size_t)N * Cin * H * W), w((size_t)Cout * Cin * 9), bias(Cout);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) % 2001 - 1000) / 500.f; };
  for (auto& v : x) { v = rnd(); if (v < 0) v = 0; }
  const float ws = std::sqrt(2.f / (Cin * 9));
  for (auto& v : w) v = rnd() * ws;
  for (auto& v : bias) v = rnd();
  const long img_stride = (long)MT * KI * CK * BM;
  std::vector<float> wp((size_t)36 * img_stride, 1e30f), V((size_t)36 * Cin * T_pad, 1e30f), M((size_t)36 * Cout * T_pad, 0.f);
  for (long i = 0; i < (long)MT * BM * KI * CK; ++i) wino_f4::weight_pair(w.data(), wp.data(), i, Cout, Cin, BM, CK, MT, KI);
  for (int ci = 0; ci < Cin; ++ci) for (int t = 0; t < T_pad; ++t) wino_f4::input_tile(x.data(), V.data(), t, ci, Cin, H, W, pad, pad, th, tw, T, T_pad);
  for (int p = 0; p < 36; ++p)
    for (int co = 0; co < Cout; ++co) {
      const int mt = co / BM, m = co % BM;
      for (int t = 0; t < T_pad; ++t) {
        float acc = 0.f;
        for (int ci = 0; ci < KI * CK; ++ci) {      // padded K: the packed weights must be zero there
          const float u = wp[p * img_stride + (((long)mt * KI + ci / CK) * CK + ci % CK) * BM + m];
          const float v = ci < Cin ? V[((long)p * Cin + ci) * T_pad + t] : 0.f;
          acc += u * v;
        }
        M[((long)p * Cout + co) * T_pad + t] = acc;
      }
    }
  const int Hp = (Ho + 1) / 2, Wp = (Wo + 1) / 2;
  std::vector<float> y((size_t)N * Cout * Ho * Wo, -7.f), yp((size_t)N * Cout * Hp * Wp, -7.f);
  unsigned am = 0;
  for (int co = 0; co < Cout; ++co) for (int t = 0; t < T; ++t) {
    const unsigned a = wino_f4::output_tile(M.data(), bias.data(), y.data(), yp.data(), t, co, Cout, Ho, Wo, th, tw, T_pad, relu);
    am = a > am ? a : am;
  }
  double err = 0, ymax = 0; int pool_bad = 0, pad_bad = 0;
  for (int n = 0; n < N; ++n) for (int co = 0; co < Cout; ++co) for (int oh = 0; oh < Ho; ++oh) for (int ow = 0; ow < Wo; ++ow) {
    double ref = bias[co];
    for (int ci = 0; ci < Cin; ++ci) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
      const int h = oh + a - pad, ww = ow + b - pad;
      if (h >= 0 && h < H && ww >= 0 && ww < W) ref += (double)x[(((size_t)n * Cin + ci) * H + h) * W + ww] * w[(((size_t)co * Cin + ci) * 3 + a) * 3 + b];
    }
    if (relu && ref < 0) ref = 0;
    const double got = y[(((size_t)n * Cout + co) * Ho + oh) * Wo + ow];
    const double e = std::fabs(got - ref) / std::fmax(1.0, std::fabs(ref));
    if (e > err) err = e;
    if (std::fabs(got) > ymax) ymax = std::fabs(got);
  }
  for (int n = 0; n < N; ++n) for (int co = 0; co < Cout; ++co) for (int ph = 0; ph < Hp; ++ph) for (int pw = 0; pw < Wp; ++pw) {
    float m = -3.402823466e+38f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) {
      const int oh = 2 * ph + a, ow = 2 * pw + b;
      if (oh < Ho && ow < Wo) { const float v = y[(((size_t)n * Cout + co) * Ho + oh) * Wo + ow]; if (v > m) m = v; }
    }
    if (yp[(((size_t)n * Cout + co) * Hp + ph) * Wp + pw] != m) ++pool_bad;
  }
  for (int p = 0; p < 36; ++p) for (int ci = 0; ci < Cin; ++ci) for (int t = T; t < T_pad; ++t) if (V[((long)p * Cin + ci) * T_pad + t] != 0.f) ++pad_bad;
  union { float f; unsigned u; } c; c.f = (float)ymax;
  printf("%.6g %d %d %d\n", err, pool_bad, pad_bad, (int)(am == c.u));
  return 0;
}
'''


@pytest.mark.parametrize("case", [(1, 16, 8, 12, 24, 1, 0), (1, 20, 13, 21, 130, 1, 1), (2, 24, 10, 14, 32, 1, 1), (1, 8, 9, 16, 16, 0, 0),
                                  (1, 40, 7, 30, 12, 0, 1), (1, 64, 16, 16, 64, 1, 1)])
def test_kernel_bodies_run_whole_layers_on_the_host(tmp_path, case):
    """The code a GPU thread of wino44_{weight,input_plane,output}_kernel executes (wino_f4::weight_pair / input_tile /
    output_tile), driven over whole layers by host loops: packed-weight layout, tile decode, zero padding of the image border and
    of the GEMM's padding tiles, ragged last tiles, the fused 2x2 pooling (ceil mode, odd sizes) and the published max |y|.
    Odd sizes, pad 0 / 1, batch 2, Cout beyond one 128-row tile, Cin not a multiple of the 32-channel chunk."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "layer"
    src = tmp_path / "layer.cpp"
    src.write_text(_LAYER_HARNESS)
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "mscnn_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([str(exe)] + [str(v) for v in case], capture_output=True, text=True, timeout=300).stdout.split()
    err, pool_bad, pad_bad, amax_ok = float(out[0]), int(out[1]), int(out[2]), int(out[3])
    assert err < 2e-5, err                 # fp32 Winograd against the float64 direct sum at these channel counts
    assert pool_bad == 0 and pad_bad == 0 and amax_ok == 1


def test_f4_plan_plumbing_without_a_device():
    """MSCNN_CONV_ALGO_WINO_F4 through the plan API (host-only calls): 36 planes in the packed weights and the workspace,
    executed FLOPs 36 / 16 per output against F(3x3,3x3)'s 25 / 9, pooling supported, ROI maps stay on F(3x3,3x3)."""
    torch = pytest.importorskip("torch")  # noqa: F841  (ConvPlan holds its buffers as torch tensors)
    from mscnn_amd import hipapi as hip
    L = hip.lib()
    p3 = hip.ConvPlan(1, 512, 72, 240, 512, 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F3, device="cpu")
    p4 = hip.ConvPlan(1, 512, 72, 240, 512, 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F4, device="cpu")
    assert (p3.kernel, p4.kernel) == ("winograd_f3x3_3x3", "winograd_f4x4_3x3") and p4.dtype == "f32" and p4.can_pool
    assert L.mscnn_conv2d_packed_weight_bytes(p4._p) * 25 == L.mscnn_conv2d_packed_weight_bytes(p3._p) * 36
    tiles3, tiles4 = 24 * 80, 18 * 60
    assert p3.executed_flops == 2.0 * 25 * 512 * 512 * tiles3 and p4.executed_flops == 2.0 * 36 * 512 * 512 * tiles4
    assert p4.flops == p3.flops and p4.executed_flops < 0.82 * p3.executed_flops
    pad4 = (tiles4 + 31) // 32 * 32      # the tile axis is padded to the plane GEMM's column tile (96 ... 256, wgemm.hip's plan)
    assert L.mscnn_conv2d_workspace_bytes(p4._p) >= 36 * (512 + 512) * pad4 * 4
    roi = hip.ConvPlan(64, 1024, 7, 7, 512, 3, 3, (0, 0), relu=True, algo=hip.ALGO_WINO_F4, device="cpu")
    assert roi.kernel == "winograd_f3x3_3x3"
    # AUTO (round 3, after the first hardware runs): F(4x4,3x3) where the 4x4 tiles number >= 1000
    auto = lambda cin, h, w, cout, **kw: hip.ConvPlan(1, cin, h, w, cout, 3, 3, (1, 1), device="cpu", **kw).kernel      # noqa: E731
    assert auto(512, 72, 240, 512) == "winograd_f4x4_3x3" and auto(256, 144, 480, 256) == "winograd_f4x4_3x3"
    assert auto(128, 288, 960, 128) == "winograd_f4x4_3x3" and auto(64, 288, 960, 128) == "winograd_f4x4_3x3"        # conv2_2, conv2_1
    assert auto(128, 144, 480, 256) == "winograd_f4x4_3x3" and auto(256, 72, 240, 512) == "winograd_f4x4_3x3"        # conv3_1, conv4_1
    assert auto(512, 36, 120, 512) == "winograd_f3x3_3x3" and auto(512, 18, 60, 512) == "winograd_f3x3_3x3"          # conv5_x, conv6_1
    # conv1_2: the one-launch Winograd F(2x2,3x3) kernel (round 5); tune_flags bit 16 -> the direct ring kernel, + bit 15 -> the igemm kernel
    assert auto(64, 576, 1920, 64) == "winograd2x2_fused_k3x3_c64" and auto(64, 576, 1920, 64, tune_flags=65536) == "wconv_64x512_k3x3"
    assert auto(64, 576, 1920, 64, tune_flags=65536 | 32768).startswith("igemm_") and auto(64, 576, 1920, 64, algo=hip.ALGO_DIRECT) == "wconv_64x512_k3x3"
    assert auto(64, 288, 960, 64) == "winograd2x2_fused_k3x3_c64" and auto(64, 288, 960, 64, tune_flags=65536).startswith("igemm_")   # (not whole 4 x 128 tiles: no ring kernel)
    assert auto(64, 100, 960, 64).startswith("igemm_")                                                               # (not whole 8 x 32 blocks either)
    assert auto(512, 72, 240, 512, tune_flags=64) == "winograd_f3x3_3x3"                                             # A/B knob: the round-2 choice
    # conv1_1 (Cin = 3) has its own VALU kernel; tune_flags bit 11 keeps the MFMA igemm kernel
    assert auto(3, 576, 1920, 64) == "conv3x3_c3_valu_f32" and auto(3, 576, 1920, 64, tune_flags=2048).startswith("igemm_")
    assert auto(3, 20, 32, 16).startswith("igemm_")                                                                  # small maps stay on the igemm kernel
    # the plane GEMM's ROW tile is part of the packed-weight layout word: bits 8.. = 200 + BM / 128 (the 256-row shapes -- x 96, x 128,
    # x 160 columns -- share one packing, so a re-plan between them needs no re-pack); Cout = 128 layers take the 128 x 256 tile
    L.mscnn_conv2d_plan_weight_layout.restype = __import__("ctypes").c_ulonglong
    variant = lambda cin, h, w, cout: ((L.mscnn_conv2d_plan_weight_layout(hip.ConvPlan(1, cin, h, w, cout, 3, 3, (1, 1), device="cpu")._p) >> 8) & 0xffff) - 200      # noqa: E731
    assert variant(512, 36, 120, 512) == 2 and variant(512, 72, 240, 512) == 2 and variant(512, 18, 60, 512) == 1      # conv6_1: too few 256-row tiles for the chip -> 128 x 128
    assert variant(128, 288, 960, 128) == 1 and variant(512, 48, 160, 512) == 2


def test_proposal_head_split_rule_through_the_plan():
    """The split of the proposal-head kernels (headconv.hip: head_plan) as the C ABI shows it without a GPU: the packed-weight size is
    KI x quads x registers x 64 lanes (so it names the channel chunk), the workspace is G workgroups x 2 slabs.  Round 5: the 5-row
    kernels run chunks of 4 channels on every map, the 7-row ones chunks of 2 on maps of <= 16 tiles (16 x 32 pixels) and of 4 above;
    one unit per workgroup up to 256 units, then half as many workgroups as units, at most 512; tune_variant 500 / 501 force full /
    half chunks, tune_grid the workgroups (clamped to the units).  A batch change that moves a 7-row head across the 16-tile line
    changes the weight layout id (the binding re-packs)."""
    pytest.importorskip("torch")
    from mscnn_amd import hipapi as hip
    L = hip.lib()

    def plan(N, H, W, k, cout=9, **kw):
        p = hip.ConvPlan(N, 512, H, W, cout, k, k, (k // 2, k // 2), device="cpu", **kw)
        return p.kernel, L.mscnn_conv2d_packed_weight_bytes(p._p), L.mscnn_conv2d_workspace_bytes(p._p), L.mscnn_conv2d_plan_weight_layout(p._p)

    def packed(ck, taps, quads):
        return (512 // ck) * quads * ((ck * taps + 15) // 16) * 64 * 4

    def ws(G, quads):
        return G * 2 * (quads * 4 * 512) * 4

    for H, W, G in [(9, 30, 128), (18, 60, 256), (36, 120, 512), (72, 240, 512)]:        # 1 / 4 / 12 / 40 tiles, 128 units each
        name, pb, wb, _ = plan(1, H, W, 5)
        assert name == "head4x4_k5x5_m3x4" and pb == packed(4, 25, 3) and wb == ws(G, 3), (H, W)
    for N, H, W, ck in [(1, 18, 60, 2), (1, 36, 120, 2), (2, 36, 120, 4)]:                 # 4 / 12 / 24 tiles
        name, pb, wb, _ = plan(N, H, W, 7)
        assert name == "head4x4_k7x7_m3x4" and pb == packed(ck, 49, 3) and wb == ws(512, 3), (N, H, W)
    assert plan(1, 36, 120, 7)[3] != plan(2, 36, 120, 7)[3]
    assert plan(1, 72, 240, 7)[0] == "head_kwfold_shiftadd_f32"                            # KW x Cout = 63: the kw-folded GEMM form
    assert plan(1, 72, 240, 7, cout=7)[:3] == ("head4x4_k7x7_m2x4", packed(4, 49, 2), ws(512, 2))
    assert plan(1, 72, 240, 7, cout=7, tune_variant=501)[1] == packed(2, 49, 2)
    assert plan(1, 9, 30, 5, tune_variant=500)[1:3] == (packed(8, 25, 3), ws(32, 3))       # the round-4 split: >= 2 chunks per workgroup
    assert plan(1, 9, 30, 5, tune_variant=500, tune_grid=1 << 20)[2] == ws(64, 3)
    assert plan(1, 9, 30, 5, tune_grid=7)[2] == ws(7, 3)
