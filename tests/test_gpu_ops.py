"""Parity of every HIP op (through the C ABI of libmscnn_hip.so) against the CPU oracle on the same
seeded inputs.  Bars (north_star): bit-exact for index/selection and compare/select work (pool, ROI pool,
NMS keep sets, BoxOutput selection); fp32 values within 1e-4 relative (|a-b| <= 1e-4 * max(1,|b|))."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X: torch.cuda.is_available() is False")
    from mscnn_amd import hipapi
    hipapi.lib()          # raises if libmscnn_hip.so is missing: no silent fallback
    return hipapi


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol=1e-4):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert err.max() <= tol, f"max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


# ------------------------------------------------------------------ elementwise / pooling
def test_relu(hip, orc):
    rng = np.random.default_rng(1)
    for n in (1, 7, 1024, 4099):
        x = rng.standard_normal(n).astype(np.float32)
        assert np.array_equal(hip.relu(dev(x)).cpu().numpy(), orc.relu(x))
    x = rng.standard_normal(64).astype(np.float32)
    close(hip.relu(dev(x), 0.1).cpu().numpy(), orc.relu(x, 0.1), 1e-6)
    t = dev(x)
    hip.relu(t, inplace=True)
    assert np.array_equal(t.cpu().numpy(), orc.relu(x))


@pytest.mark.parametrize("shape,k,p,s,m", [
    ((1, 64, 32, 48), (2, 2), (0, 0), (2, 2), "MAX"),     # fast path
    ((2, 3, 9, 15), (2, 2), (0, 0), (2, 2), "MAX"),       # ceil-mode edges (caltech pool6)
    ((1, 2, 3, 5), (2, 2), (0, 0), (1, 1), "MAX"),
    ((1, 1, 3, 3), (3, 3), (2, 2), (2, 2), "MAX"),        # test_pooling_layer.cpp:478-522
    ((1, 4, 8, 8), (2, 2), (0, 0), (1, 1), "AVE"),        # widerface 2x2 s1 AVE after ROIAlign
    ((1, 2, 7, 7), (3, 3), (1, 1), (2, 2), "AVE"),
])
def test_pool(hip, orc, shape, k, p, s, m):
    x = np.random.default_rng(2).standard_normal(shape).astype(np.float32)
    y = hip.pool2d(dev(x), k, p, s, m).cpu().numpy()
    ref = orc.pool2d(x, k, p, s, m)
    if m == "MAX":
        assert np.array_equal(y, ref)
    else:
        close(y, ref, 1e-6)


def test_parity_metric_kernels_on_device(hip):
    """The device forms of the suite's own metric (the host runtime's Winograd-vs-direct checks run on them): mscnn_sum_squares_f32,
    mscnn_max_rel_diff_f32, and the strided form the numerics watch uses on a band of rows of an NCHW blob -- floor max(1, rms) taken on
    the device from the preceding reduction -- against numpy in float64; a NaN anywhere in the compared range gives +inf, a NaN outside
    it does not; mscnn_store_words_i32 writes 1 .. 4 words and nothing else."""
    rng = np.random.default_rng(11)
    C_, H, W = 5, 19, 33
    ref = (rng.standard_normal((C_, H, W)) * 7).astype(np.float32)
    a = (ref + rng.standard_normal(ref.shape).astype(np.float32) * 1e-3).astype(np.float32)
    ss = hip.sum_squares(dev(ref))
    assert abs(float(ss.cpu()[0]) - float((ref.astype(np.float64) ** 2).sum())) <= 1e-9 * float((ref.astype(np.float64) ** 2).sum())
    want = (np.abs(a.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))).max()
    assert abs(float(hip.max_rel_diff(dev(a), dev(ref)).cpu()[0]) - want) <= 1e-6 * want
    # rows [r0, r0 + rows) of every plane of `a` against a band that holds rows [r0 - 1, r0 + rows + 1): what BeginBandCheck compares
    r0, rows = 6, 4
    band = np.ascontiguousarray(ref[:, r0 - 1:r0 + rows + 1])
    rms = np.sqrt((band.astype(np.float64) ** 2).mean())
    got = hip.max_rel_diff_strided(dev(a).view(-1)[r0 * W:], H * W, dev(band).view(-1)[W:], (rows + 2) * W,
                                   C_, rows * W, hip.sum_squares(dev(band)), band.size)
    w2 = (np.abs(a[:, r0:r0 + rows].astype(np.float64) - ref[:, r0:r0 + rows]) / np.maximum(max(1.0, rms), np.abs(ref[:, r0:r0 + rows]))).max()
    assert abs(float(got.cpu()[0]) - w2) <= 1e-5 * w2, (float(got.cpu()[0]), w2)
    bad = a.copy(); bad[2, r0 + 1, 5] = np.nan
    assert np.isinf(float(hip.max_rel_diff_strided(dev(bad).view(-1)[r0 * W:], H * W, dev(band).view(-1)[W:], (rows + 2) * W, C_, rows * W,
                                                   hip.sum_squares(dev(band)), band.size).cpu()[0]))
    bad = a.copy(); bad[2, r0 - 1, 5] = np.nan; bad[1, r0 + rows, 0] = np.nan          # outside the compared rows: not seen
    assert np.isfinite(float(hip.max_rel_diff_strided(dev(bad).view(-1)[r0 * W:], H * W, dev(band).view(-1)[W:], (rows + 2) * W, C_, rows * W,
                                                      hip.sum_squares(dev(band)), band.size).cpu()[0]))
    buf = torch.full((8,), -7, dtype=torch.int32, device="cuda")
    hip.store_words(buf[2:], [11, -3, 2 ** 31 - 1])
    assert buf.cpu().tolist() == [-7, -7, 11, -3, 2 ** 31 - 1, -7, -7, -7]
    with pytest.raises(Exception):
        hip.store_words(buf, [1, 2, 3, 4, 5])


def test_pool_kat_on_device(hip):
    plane = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    y = hip.pool2d(dev(np.tile(plane, (2, 2, 1, 1))), (2, 2), (0, 0), (1, 1)).cpu().numpy()
    assert np.array_equal(y[1, 1], np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32))


def test_concat_softmax_deconv(hip, orc):
    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 3, 7, 7)).astype(np.float32); b = rng.standard_normal((5, 4, 7, 7)).astype(np.float32)
    assert np.array_equal(hip.concat_channels([dev(a), dev(b)]).cpu().numpy(), orc.concat_channels([a, b]))
    x = rng.standard_normal((6, 5)).astype(np.float32)
    close(hip.softmax(dev(x)).cpu().numpy(), orc.softmax(x), 1e-6)
    x = rng.standard_normal((1, 8, 9, 11)).astype(np.float32)
    w = orc.bilinear_filler((8, 1, 4, 4))
    close(hip.deconv_depthwise(dev(x), dev(w), None, (1, 1), (2, 2)).cpu().numpy(),
          orc.deconv2d(x, w, None, (1, 1), (2, 2), group=8), 1e-6)


# ------------------------------------------------------------------ convolution
CONV_CASES = [
    # (N, Cin, H, W, Cout, k, pad, stride, group)          kernel family expected
    (1, 3, 20, 33, 16, (3, 3), (1, 1), (1, 1), 1),          # conv1_1-like: Cin 3 zero-padded to one 8-channel chunk
    (1, 12, 11, 19, 40, (3, 3), (1, 1), (1, 1), 1),         # Cin not a multiple of 8 over two chunks
    (13, 16, 7, 7, 256, (3, 3), (0, 0), (1, 1), 1),         # roi_c1 (kitti car): ROI mode, 5 images per tile, ragged last tile
    (9, 8, 7, 5, 130, (3, 3), (0, 0), (1, 1), 1),           # roi_c1 ped/cyc 7x5 -> 5x3
    (6, 24, 8, 4, 64, (3, 3), (1, 1), (1, 1), 1),           # roi_c1 caltech 8x4 pad 1 -> 8x4
    (2, 8, 5, 5, 16, (3, 3), (1, 1), (2, 2), 1),            # stride 2 -> direct
    (2, 6, 9, 7, 4, (3, 3), (0, 0), (2, 2), 2),             # stride/group -> direct
    (1, 8, 16, 32, 128, (3, 3), (1, 1), (1, 1), 1),         # igemm 128x128, exact tiles
    (1, 16, 13, 37, 64, (3, 3), (1, 1), (1, 1), 1),         # ragged H/W, Cout 64
    (1, 24, 36, 120, 256, (3, 3), (1, 1), (1, 1), 1),       # conv5-shaped plane, 2 M tiles, stream-K splits
    (2, 8, 10, 20, 130, (3, 3), (1, 1), (1, 1), 1),         # batch 2, Cout not a multiple of 32
    (1, 32, 18, 60, 9, (5, 5), (2, 2), (1, 1), 1),          # proposal head 5x5
    (1, 16, 9, 30, 9, (7, 7), (3, 3), (1, 1), 1),           # proposal head 7x7
    (1, 16, 12, 20, 7, (5, 3), (2, 1), (1, 1), 1),          # ped/cyc head "3x5" (kernel_w 3, kernel_h 5)
    (1, 8, 12, 20, 7, (7, 5), (3, 2), (1, 1), 1),           # ped/cyc head "5x7" (kernel_w 5, kernel_h 7)
    (2, 20, 40, 70, 6, (5, 3), (2, 1), (1, 1), 1),          # caltech head: 6 channels, ragged Cin, batch 2, several tiles
    (1, 64, 36, 120, 9, (7, 7), (3, 3), (1, 1), 1),         # conv5_3-sized head plane: stream-K splits + fix-up
    (1, 12, 20, 40, 12, (5, 5), (2, 2), (1, 1), 1),         # 12 channels (3 quads all live)
    (1, 64, 10, 20, 96, (1, 1), (0, 0), (1, 1), 1),         # 1x1 (igemm_128x128_k1x1)
    (3, 16, 7, 7, 64, (3, 3), (0, 0), (1, 1), 1),           # roi_c1-shaped: R x (C,7,7) no pad -> 5x5
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_conv(hip, orc, case, relu):
    N, Cin, H, W, Cout, k, pad, stride, group = case
    rng = np.random.default_rng(1701)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin // group, *k)) * np.sqrt(2.0 / (Cin // group * k[0] * k[1]))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    y = hip.conv2d(dev(x), dev(w), dev(b), pad, stride, group, relu).cpu().numpy()
    ref = orc.conv2d(x, w, b, pad, stride, group)
    if relu:
        ref = orc.relu(ref)
    close(y, ref)


@pytest.mark.parametrize("case", [(1, 64, 96, 320, True, True), (2, 64, 67, 132, True, False), (1, 16, 130, 36, False, True), (1, 128, 64, 64, True, True),
                                  (1, 64, 576, 1920, True, True)])
def test_conv_c3_bit_identical_to_the_igemm_kernel(hip, orc, case):
    """conv1_1 (Cin = 3, K = 27) on its own VALU kernel (conv_c3.hip) against the MFMA igemm kernel it replaces (tune_flags bit 11):
    the same k order (taps outer, channels inner, bias last) => equal values everywhere (== treats the -0 a zero-padded MFMA k can
    leave as +0); ragged heights, widths that are not a multiple of the 128-column tile, batch 2, no bias / no ReLU; and the
    reference's tolerance against the oracle on the small cases."""
    N, Cout, H, W, relu, with_bias = case
    g = torch.Generator(device="cuda").manual_seed(H + W)
    x = torch.randn((N, 3, H, W), device="cuda", generator=g) * 60.0                  # BGR - mean scale
    w = torch.randn((Cout, 3, 3, 3), device="cuda", generator=g) * (2.0 / 27) ** 0.5
    b = torch.randn((Cout,), device="cuda", generator=g) if with_bias else None
    p = hip.ConvPlan(N, 3, H, W, Cout, 3, 3, (1, 1), relu=relu)
    assert p.kernel == "conv3x3_c3_valu_f32"
    p.pack(w)
    y = p.forward(x, b).clone()
    q = hip.ConvPlan(N, 3, H, W, Cout, 3, 3, (1, 1), relu=relu, tune_flags=2048)
    assert q.kernel.startswith("igemm_")
    q.pack(w)
    y_ig = q.forward(x, b)
    assert bool((y == y_ig).all())
    if H * W <= 50000:
        ref = orc.conv2d(x.cpu().numpy(), w.cpu().numpy(), None if b is None else b.cpu().numpy(), (1, 1))
        close(y.cpu().numpy(), orc.relu(ref) if relu else ref)


@pytest.mark.parametrize("case", [(1, 512, 72, 240, 9, (5, 5), (2, 2)), (1, 512, 36, 120, 9, (7, 7), (3, 3)), (1, 512, 18, 60, 9, (5, 5), (2, 2)),
                                  (1, 512, 9, 30, 9, (7, 7), (3, 3)), (2, 96, 40, 70, 6, (5, 3), (2, 1)), (3, 64, 20, 33, 12, (5, 5), (2, 2))])
@pytest.mark.parametrize("flags,name", [(0, "head4x4")])
@pytest.mark.parametrize("tune", [(0, 0), (500, 0), (501, 0), (500, 1 << 20), (501, 1 << 20)], ids=["auto", "full-chunks", "half-chunks", "full+one-per-wg", "half+one-per-wg"])
def test_conv_head_stream_k_is_deterministic(hip, orc, case, flags, name, tune):
    """(The packed-FMA variant of the same split, tools/micro/headvalu.hip, left the product library in round 5: `make witness`.)
    The M = 4 head kernel splits its few tiles stream-K style and a fix-up launch adds the partial sums in k order: against the
    oracle (reference tolerance 1e-4) at the full-size head shapes of the 7s nets, bit-identical over 20 back-to-back launches,
    and exactly doubled after re-packing doubled weights.  (Round 3 tried the combine inside the launch -- last arrival reduces --
    and measured it slower; this test is what it had to pass.)  `tune`: the split itself -- full / half-size channel chunks
    whatever the map (tune_variant 500 / 501; AUTO takes half chunks on maps of <= 16 tiles) and as many workgroups as there are
    (tile, chunk) units (tune_grid, clamped to them): up to 256 contributors per tile through the fix-up's list."""
    N, Cin, H, W, Cout, k, pad = case
    rng = np.random.default_rng(99)
    x = np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, *k)) * np.sqrt(2.0 / (Cin * k[0] * k[1]))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    p = hip.ConvPlan(N, Cin, H, W, Cout, k[0], k[1], pad, tune_flags=flags, tune_variant=tune[0], tune_grid=tune[1])
    assert p.kernel.startswith(name)
    p.pack(dev(w))
    xd, bd = dev(x), dev(b)
    y0 = p.forward(xd, bd).clone()
    close(y0.cpu().numpy(), orc.conv2d(x, w, b, pad))
    for _ in range(20):
        assert torch.equal(p.forward(xd, bd), y0)
    p.pack(dev(w * 2))      # re-packing zeroes the counters again and the next launch sees the new weights
    y2 = p.forward(xd, dev(b * 2))
    assert torch.equal(y2, y0 * 2)


@pytest.mark.parametrize("case", [(1, 64, 36, 120, 9, (5, 5)), (1, 128, 18, 60, 9, (7, 7)), (2, 64, 12, 20, 7, (5, 3)), (1, 96, 13, 21, 6, (7, 5)),
                                  (1, 512, 9, 30, 9, (5, 5)), (1, 32, 7, 9, 12, (3, 3))])
def test_conv_head_x3_gemm_shiftadd(hip, orc, case):
    """Proposal heads in the split-fp16 mode: ONE dense GEMM T[tap * Cout + co][pixel] over the taps (M = taps * Cout instead of Cout)
    + a shift-and-add with the zero padding of the convolution -- against the oracle's direct convolution (1e-4) and no worse than 3x
    the fp32 head kernel + 2e-6; borders smaller than the kernel, batch 2, ReLU, handed-over max |x|."""
    N, Cin, H, W, Cout, (kh, kw) = case
    rng = np.random.default_rng(43)
    x = np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32) * 2
    w = (rng.standard_normal((Cout, Cin, kh, kw)) * np.sqrt(2.0 / (Cin * kh * kw))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), algo=hip.ALGO_WINO_F3_X3)
    assert plan.kernel == "head_gemm_shiftadd_x3f16" and plan.dtype == "f16x3" and not plan.can_pool and not plan.publishes_amax
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.conv2d(x, w, b, (kh // 2, kw // 2))
    close(y, ref)
    p32 = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2))
    p32.pack(dev(w))
    y32 = p32.forward(dev(x), dev(b)).cpu().numpy()
    truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                       padding=(kh // 2, kw // 2)).numpy()
    m = lambda a: float((np.abs(a - truth) / np.maximum(1, np.abs(truth))).max())      # noqa: E731
    print(f"head x3 err {m(y):.2e}  fp32 head kernel err {m(y32):.2e}")
    assert m(y) <= 3 * m(y32) + 2e-6
    pr = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), relu=True, algo=hip.ALGO_WINO_F3_X3)
    pr.pack(dev(w))
    if N == 1:
        bound = torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda")
        bound[3] = torch.tensor(float(np.abs(x).max()) * 1.3, dtype=torch.float32).view(torch.int32)
        pr.set_amax_io(bound, None)
    close(pr.forward(dev(x), dev(b)).cpu().numpy(), np.maximum(ref, 0))


@pytest.mark.parametrize("case", [(1, 64, 36, 120, 9, (5, 5)), (1, 128, 18, 60, 9, (7, 7)), (2, 64, 12, 20, 7, (5, 3)), (1, 96, 13, 21, 6, (7, 5)),
                                  (1, 512, 9, 30, 9, (5, 5)), (1, 32, 7, 9, 12, (3, 3)), (1, 512, 72, 240, 9, (7, 7)), (1, 512, 36, 120, 9, (5, 5))])
def test_conv_head_gemm_shiftadd_f32(hip, orc, case):
    """Proposal heads on the fp32 MFMA as ONE dense GEMM over the taps (the nested 1x1 igemm plan: the vectorised Winograd GEMM
    kernel where H * W is a multiple of 128, the generic one else) + the shift-and-add: against the oracle's direct convolution
    (1e-4) and no worse than 3x the M = 4 head kernel + 2e-6 against float64; borders smaller than the kernel, batch 2, ReLU; the
    last two cases are LFCN_1_7x7 / LFCN_2_5x5 of mscnn-7s-576 at full size."""
    N, Cin, H, W, Cout, (kh, kw) = case
    rng = np.random.default_rng(44)
    x = np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32) * 2
    w = (rng.standard_normal((Cout, Cin, kh, kw)) * np.sqrt(2.0 / (Cin * kh * kw))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), tune_flags=16)
    assert plan.kernel == "head_gemm_shiftadd_f32" and plan.dtype == "f32" and not plan.can_pool and not plan.publishes_amax
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.conv2d(x, w, b, (kh // 2, kw // 2))
    close(y, ref)
    p32 = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), tune_flags=32)
    assert not p32.kernel.startswith("head_gemm")                # the M = 4 head kernel (3x3: the igemm tile)
    p32.pack(dev(w))
    y32 = p32.forward(dev(x), dev(b)).cpu().numpy()
    truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                       padding=(kh // 2, kw // 2)).numpy()
    m = lambda a: float((np.abs(a - truth) / np.maximum(1, np.abs(truth))).max())      # noqa: E731
    print(f"head gemm err {m(y):.2e}  M = 4 head kernel err {m(y32):.2e}")
    assert m(y) <= 3 * m(y32) + 2e-6
    pr = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), relu=True, tune_flags=16)
    pr.pack(dev(w))
    yr = pr.forward(dev(x), dev(b))
    close(yr.cpu().numpy(), np.maximum(ref, 0))
    assert np.array_equal(pr.forward(dev(x), dev(b)).cpu().numpy(), yr.cpu().numpy())      # deterministic (k-ordered fix-up, no atomics)


@pytest.mark.parametrize("case", [(1, 64, 36, 120, 9, (5, 5)), (1, 128, 34, 61, 9, (7, 7)), (2, 64, 33, 40, 9, (7, 7)), (1, 96, 40, 37, 8, (7, 5)),
                                  (1, 512, 72, 240, 9, (7, 7)), (1, 512, 72, 240, 9, (5, 5))])
def test_conv_head_kwfold(hip, orc, case):
    """Proposal heads with the kernel's COLUMNS folded into M (round 3): a KH x 1 implicit-GEMM convolution with KW * Cout <= 64 output
    channels on the 64-row MFMA tile + a shift-and-add over KW rows, against the oracle's direct convolution (1e-4) and no worse than
    3x the M = 4 head kernel + 2e-6 against float64; ragged widths, batch 2, ReLU, determinism; the last two cases are LFCN_1_7x7 (the
    shape AUTO takes it for) and LFCN_1_5x5 of mscnn-7s-576 at full size."""
    N, Cin, H, W, Cout, (kh, kw) = case
    rng = np.random.default_rng(45)
    x = np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32) * 2
    w = (rng.standard_normal((Cout, Cin, kh, kw)) * np.sqrt(2.0 / (Cin * kh * kw))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), tune_flags=1024)
    assert plan.kernel == "head_kwfold_shiftadd_f32" and plan.dtype == "f32" and not plan.can_pool
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.conv2d(x, w, b, (kh // 2, kw // 2))
    close(y, ref)
    p32 = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), tune_flags=512 | 32)
    assert p32.kernel.startswith("head4x4")
    p32.pack(dev(w))
    y32 = p32.forward(dev(x), dev(b)).cpu().numpy()
    truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                       padding=(kh // 2, kw // 2)).numpy()
    m = lambda a: float((np.abs(a - truth) / np.maximum(1, np.abs(truth))).max())      # noqa: E731
    print(f"kw-fold err {m(y):.2e}  M = 4 head kernel err {m(y32):.2e}")
    assert m(y) <= 3 * m(y32) + 2e-6
    pr = hip.ConvPlan(N, Cin, H, W, Cout, kh, kw, (kh // 2, kw // 2), relu=True, tune_flags=1024)
    pr.pack(dev(w))
    yr = pr.forward(dev(x), dev(b))
    close(yr.cpu().numpy(), np.maximum(ref, 0))
    assert np.array_equal(pr.forward(dev(x), dev(b)).cpu().numpy(), yr.cpu().numpy())      # deterministic (k-ordered fix-up, no atomics)


WINO_CASES = [   # N, Cin, H, W, Cout, pad
    (1, 16, 8, 12, 24, 1),        # exact 2x2 tiles
    (1, 40, 13, 21, 130, 1),      # odd H and W: partial tiles at the bottom / right edge, Cout ragged
    (2, 24, 10, 14, 32, 1),       # batch 2
    (1, 8, 9, 16, 16, 0),         # pad 0 (Ho = H - 2)
    (1, 320, 18, 60, 320, 1),     # chosen by the default heuristic (conv6_1-like plane), stream-K split of the 1x1 GEMM
    (2, 24, 20, 100, 32, 1),      # 34 tile columns, batch 2: the LDS-staged input transform (>= 32 tile columns), ragged block
    (1, 16, 11, 290, 24, 1),      # 97 tile columns = 1.5 blocks of 64, 4 tile rows (one block row), odd sizes
    (1, 8, 30, 96, 16, 0),        # pad 0, 32 tile columns exactly, 10 tile rows = 2.5 block rows
]


@pytest.mark.parametrize("case", WINO_CASES)
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("m", [2, 3])
def test_conv_winograd(hip, orc, case, relu, m):
    """Winograd paths on whole planes -- F(2x2,3x3) (16 planes) and F(3x3,3x3) (25 planes, the default): input transform ->
    batched 1x1 igemm GEMMs -> output transform, against the oracle's direct convolution: same 1e-4 bound as every other fp32
    layer."""
    N, Cin, H, W, Cout, pad = case
    algo = hip.ALGO_WINO_F2 if m == 2 else hip.ALGO_WINO_F3
    rng = np.random.default_rng(4242)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=relu, algo=algo)
    assert plan.kernel == f"winograd_f{m}x{m}_3x3"
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.conv2d(x, w, b, (pad, pad))
    if relu:
        ref = orc.relu(ref)
    close(y, ref)
    assert not hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), algo=hip.ALGO_DIRECT).kernel.startswith("winograd")


# F(4x4,3x3) (MSCNN_CONV_ALGO_WINO_F4): first run on hardware in round 3 (all 23 cases green on the first run); the plan's AUTO choice
# takes it for the large layers (conv.hip: wino_plan).
@pytest.mark.parametrize("case", WINO_CASES + [(1, 256, 36, 60, 256, 1), (1, 512, 24, 40, 512, 1)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv_winograd_f4x4(hip, orc, case, relu):
    """F(4x4,3x3) with the points {0, 1, -1, 2, -1/2, inf}: 36 plane GEMMs, against the oracle's direct convolution (1e-4) and no
    worse than 1.5x the F(3x3,3x3) form + 2e-6 against float64."""
    N, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(4243)
    x = np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=relu, algo=hip.ALGO_WINO_F4)
    assert plan.kernel == "winograd_f4x4_3x3" and plan.can_pool and plan.dtype == "f32"
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.conv2d(x, w, b, (pad, pad))
    if relu:
        ref = orc.relu(ref)
    close(y, ref)
    # the two FORMS against float64 with the same association of the K sum in their plane GEMMs: whole tiles (tune_variant 300 + 512:
    # one k-ordered fmaf chain per output; where the plan splits a tile stream-K style the cut points depend on the tile count,
    # which differs between 25 and 36 planes -- round 4's 128 x 128 tiles for small problems moved them)
    p4 = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=relu, algo=hip.ALGO_WINO_F4, tune_variant=300 + 512)
    p4.pack(dev(w))
    y4 = p4.forward(dev(x), dev(b)).cpu().numpy()
    close(y4, ref)
    p3 = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=relu, algo=hip.ALGO_WINO_F3, tune_variant=300 + 512)
    p3.pack(dev(w))
    y3 = p3.forward(dev(x), dev(b)).cpu().numpy()
    truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=pad).numpy()
    if relu:
        truth = np.maximum(truth, 0)
    m = lambda a: float((np.abs(a - truth) / np.maximum(1, np.abs(truth))).max())      # noqa: E731
    print(f"F(4x4,3x3) err {m(y4):.2e} (as planned: {m(y):.2e})  F(3x3,3x3) err {m(y3):.2e}")
    assert m(y4) <= 1.5 * m(y3) + 2e-6
    assert m(y) < 1e-4 / 2


@pytest.mark.parametrize("case", [(1, 32, 16, 24, 48, 1), (1, 24, 13, 21, 32, 1), (2, 16, 10, 14, 24, 1)])
def test_conv_winograd_f4x4_fused_pool(hip, orc, case):
    """The 4x4 tile holds 2x2 pooling windows: fused MAX 2x2 / stride 2 (ceil mode, odd sizes) bit-identical to pooling the
    layer's own output."""
    N, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=True, algo=hip.ALGO_WINO_F4)
    plan.pack(dev(w))
    yp = torch.empty((N, Cout, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device="cuda")
    y = plan.forward(dev(x), dev(b), pool_out=yp).cpu().numpy()
    close(y, orc.relu(orc.conv2d(x, w, b, (pad, pad))))
    assert np.array_equal(yp.cpu().numpy(), orc.pool2d(y))


# (the last case: 36 x 8 = 288 tiles of 8 chunks on 256 workgroups -- forced split: 32 tiles, each cut into EIGHT one-chunk parts, every
# finisher adds seven contributors' slabs)
@pytest.mark.parametrize("case", [(1, 64, 24, 40, 256, 3), (1, 96, 36, 60, 288, 4), (2, 32, 20, 28, 128, 4), (1, 512, 36, 120, 512, 3), (1, 128, 72, 240, 128, 4),
                                  (1, 256, 128, 128, 256, 4)])
def test_wgemm_plane_gemm_against_the_igemm_kernel(hip, case):
    """The plane GEMM of the Winograd layers (wgemm.hip, round 3) against the round-2 igemm kernel on the same planes.  With whole
    tiles both are k-ordered fmaf chains over the same operands, so the layer outputs must be BIT-IDENTICAL (tune_variant 300 + v +
    512 forces whole tiles, tune_flags bit 7 selects the igemm kernel).  With the stream-K split forced (+ 256) a tile's chunks are
    summed in two or three parts: equal within 4e-5 (the output transforms multiply a re-association difference in M by up to
    4 x 4 / 8 x 8; measured 1.5e-5 .. 2.9e-5 at Cin = 512; the reference's own tolerance is 1e-4), and bit-identical from run to run
    (fixed order, no atomics on data)."""
    N, Cin, H, W, Cout, m = case
    tol = 4e-5
    algo = hip.ALGO_WINO_F4 if m == 4 else hip.ALGO_WINO_F3
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.relu(torch.randn((N, Cin, H, W), device="cuda", generator=g))
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn((Cout,), device="cuda", generator=g)

    def run(tune_variant, tune_flags):
        p = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=algo, tune_variant=tune_variant, tune_flags=tune_flags)
        assert p.kernel.startswith(f"winograd_f{m}x{m}")
        p.pack(w)
        return p.forward(x, b).clone(), p
    y_ig, _ = run(0, 128)
    y_whole, _ = run(300 + 512, 0)
    rel = lambda a, r: ((a - r).abs() / torch.clamp(r.abs(), min=1.0)).max().item()      # noqa: E731
    if Cin >= 128:      # (shapes on which the igemm plan runs whole tiles too; on the small ones it splits K itself)
        assert torch.equal(y_whole, y_ig)
    else:
        assert rel(y_whole, y_ig) < tol
    y_split, p = run(300 + 256, 0)
    err = ((y_split - y_whole).abs() / torch.clamp(y_whole.abs(), min=1.0)).max().item()
    assert err < tol, err
    for _ in range(3):
        assert torch.equal(p.forward(x, b), y_split)
    y_auto, _ = run(0, 0)
    assert torch.equal(y_auto, y_whole) or ((y_auto - y_whole).abs() / torch.clamp(y_whole.abs(), min=1.0)).max().item() < tol
    # the 256 x 96 tile (variant 4: 12 LDS-DMA pieces of B on 8 waves, pieces that straddle rows): whole tiles are the same fmaf chains
    y_96, p96 = run(300 + 4 + 512, 0)
    assert p96.kernel.startswith(f"winograd_f{m}x{m}")
    if Cin >= 128:
        assert torch.equal(y_96, y_ig)
    else:
        assert rel(y_96, y_ig) < tol
    y_96s, p96s = run(300 + 4 + 256, 0)
    assert rel(y_96s, y_whole) < tol
    assert torch.equal(p96s.forward(x, b), y_96s)
    # the 256 x 160 tile (variant 5, round 4: 20 pieces of B on 8 waves, 8 x 1 waves of 32 x 160): same chains, same contract
    y_160, p160 = run(300 + 5 + 512, 0)
    assert p160.kernel.startswith(f"winograd_f{m}x{m}")
    if Cin >= 128:
        assert torch.equal(y_160, y_ig)
    else:
        assert rel(y_160, y_ig) < tol
    y_160s, p160s = run(300 + 5 + 256, 0)
    assert rel(y_160s, y_whole) < tol
    assert torch.equal(p160s.forward(x, b), y_160s)


@pytest.mark.parametrize("relu", [True, False])
def test_wgemm_handoff_timeout_is_reported_never_silent(hip, relu):
    """A stream-K hand-off that times out (fault injection: the contributors never publish, the finisher gives up after 64 polls)
    must be impossible to miss: (1) the launch's tag appears in the device's pinned status word (mscnn_wgemm_handoff_event), (2) the
    tiles it could not finish are NaN in the layer's output -- also behind the fused ReLU, which keeps a NaN a NaN like
    relu_layer.cpp:14-15's std::max -- and (3) after mscnn_wgemm_force_whole_tiles(1), the host's answer, the SAME plan (still
    planned for the split) produces the whole-tile result bit for bit and reports nothing."""
    N, Cin, H, W, Cout = 1, 256, 128, 128, 256
    g = torch.Generator(device="cuda").manual_seed(23)
    x = torch.relu(torch.randn((N, Cin, H, W), device="cuda", generator=g))
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn((Cout,), device="cuda", generator=g)

    def plan(variant):
        p = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=relu, algo=hip.ALGO_WINO_F4, tune_variant=variant)
        p.pack(w)
        return p
    p_whole, p_split = plan(300 + 512), plan(300 + 256)
    y_whole = p_whole.forward(x, b).clone()
    y_split = p_split.forward(x, b).clone()
    torch.cuda.synchronize()
    assert not torch.isnan(y_split).any()
    ev0 = hip.wgemm_handoff_event()
    assert not hip.wgemm_whole_tiles_forced()
    try:
        hip.debug_wgemm_handoff_fault(True, 64)
        y_bad = p_split.forward(x, b).clone()
        torch.cuda.synchronize()
        ev1 = hip.wgemm_handoff_event()
        assert ev1 != ev0 and ev1 != 0                      # (1) reported
        bad = torch.isnan(y_bad)
        assert bad.any()                                   # (2) poisoned, ReLU or not
        assert torch.equal(y_bad[~bad], y_split[~bad])     #     ... and only the split tiles
        p_whole.forward(x, b)                              # a launch that hands nothing over reports nothing, fault or not
        torch.cuda.synchronize()
        assert hip.wgemm_handoff_event() == ev1
        hip.wgemm_force_whole_tiles(True)                  # (3) the answer
        y_again = p_split.forward(x, b).clone()
        torch.cuda.synchronize()
        assert hip.wgemm_handoff_event() == ev1
        assert torch.equal(y_again, y_whole)
    finally:
        hip.debug_wgemm_handoff_fault(False, 0)
        hip.wgemm_force_whole_tiles(False)
    assert torch.equal(p_split.forward(x, b), y_split)


@pytest.mark.parametrize("case", [(1, 32, 24, 64, 48), (2, 16, 36, 260, 32), (1, 64, 72, 240, 64), (1, 8, 10, 512, 16), (1, 24, 13, 28, 40)])
def test_wino_f4_vector_transforms_bit_identical(hip, case):
    """The vectorised F(4x4,3x3) transforms (float4 rows + neighbour-lane halo, float4 / float2 stores) against the scalar kernels
    whose per-thread bodies the host model checks (tune_flags bit 8): same arithmetic, same order -- y and the fused pooled output
    must be bit-identical, over tile rows shorter / longer than a wave, several segments per row, odd tile-row counts and batch 2."""
    N, Cin, H, W, Cout = case
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((N, Cin, H, W), device="cuda", generator=g)
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn((Cout,), device="cuda", generator=g)
    outs = []
    for flags in (0, 256):
        plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F4, tune_flags=flags)
        plan.pack(w)
        yp = torch.full((N, Cout, (H + 1) // 2, (W + 1) // 2), float("nan"), device="cuda")
        y = plan.forward(x, b, pool_out=yp).clone()
        outs.append((y, yp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][1], hip.pool2d(outs[0][0], (2, 2), (0, 0), (2, 2)))


WINO33_CASES = [   # R, Cin, H, W, Cout, pad  (ROI-pooled maps -> roi_c1)
    (20, 64, 7, 7, 48, 0),        # kitti_car: 7x7 -> 5x5 (2x2 tiles of 3x3, last row / column dropped)
    (33, 40, 7, 5, 130, 0),       # ped/cyc: 7x5 -> 5x3
    (16, 24, 8, 4, 32, 1),        # caltech: 8x4 pad 1 -> 8x4
    (50, 1024, 7, 7, 512, 0),     # roi_c1 channel counts (default heuristic picks the path)
]


@pytest.mark.parametrize("case", WINO33_CASES)
def test_conv_winograd_f3x3(hip, orc, case):
    """Winograd F(3x3,3x3) on the small ROI maps against the oracle's direct convolution, 1e-4 bound, inputs post-ReLU
    like the ROI-pooled features."""
    R, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(99)
    x = np.maximum(rng.standard_normal((R, Cin, H, W)), 0).astype(np.float32) * 2.0
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(R, Cin, H, W, Cout, 3, 3, (pad, pad), relu=True, algo=hip.ALGO_WINO_F3)
    assert plan.kernel == "winograd_f3x3_3x3" and not plan.can_pool
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.relu(orc.conv2d(x, w, b, (pad, pad)))
    close(y, ref)
    plan.set_batch(R + 7)                                  # ROI count changes from frame to frame
    x2 = np.concatenate([x, x[:7]], 0)
    close(plan.forward(dev(x2), dev(b)).cpu().numpy(), np.concatenate([ref, ref[:7]], 0))


X3_CASES = [   # N, Cin, H, W, Cout, pad, tune_variant (0 default, 2: 256-row tiles)
    (1, 32, 13, 21, 130, 1, 0),      # odd H and W, Cout ragged (2 M tiles of 128, rows beyond 130 dropped)
    (2, 64, 10, 14, 32, 1, 0),       # batch 2, Cout < one MFMA block row
    (1, 32, 9, 16, 16, 0, 0),        # pad 0
    (1, 320, 18, 60, 320, 1, 0),     # 10 k-chunks, 3 M tiles
    (1, 96, 36, 60, 256, 1, 2),      # 256-row tiles (4 x 2 MFMA blocks per wave), 3 k-chunks
    (1, 128, 24, 50, 512, 1, 2),     # 256-row tiles, 2 M tiles, ragged tile columns
]


@pytest.mark.parametrize("case", X3_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_conv_winograd_x3(hip, orc, case, relu):
    """Split-fp16 F(3x3,3x3) (MSCNN_CONV_ALGO_WINO_F3_X3: operands split exactly into fp16 hi + lo, three fp16 MFMAs per
    product pair, fp32 accumulators) on whole planes against the oracle's direct fp32 convolution: the SAME 1e-4 bound as
    every fp32 layer -- this is not a reduced-precision mode."""
    N, Cin, H, W, Cout, pad, tv = case
    rng = np.random.default_rng(777)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=relu, algo=hip.ALGO_WINO_F3_X3, tune_variant=tv, tune_flags=4)
    assert plan.kernel == f"winograd_f3x3_3x3_x3f16_{256 if tv == 2 else 128}" and plan.dtype == "f16x3"
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.conv2d(x, w, b, (pad, pad))
    if relu:
        ref = orc.relu(ref)
    close(y, ref)
    # against the fp32 Winograd path on the same data: the split adds (far) less than the transforms' own rounding
    p32 = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=relu, algo=hip.ALGO_WINO_F3)
    p32.pack(dev(w))
    y32 = p32.forward(dev(x), dev(b)).cpu().numpy()
    e3 = float((np.abs(y - ref) / np.maximum(1, np.abs(ref))).max())
    e32 = float((np.abs(y32 - ref) / np.maximum(1, np.abs(ref))).max())
    print(f"x3 err {e3:.2e}  fp32 winograd err {e32:.2e}")
    assert e3 <= 2 * e32 + 2e-6
    # Cin not a multiple of 32: the plan keeps the fp32 F(3x3,3x3) kernels
    assert hip.ConvPlan(1, 40, 13, 21, 64, 3, 3, (1, 1), algo=hip.ALGO_WINO_F3_X3, tune_flags=4).kernel == "winograd_f3x3_3x3"
    # without the force flag the AUTO heuristic decides: a 32-channel layer is not a Winograd layer
    assert not hip.ConvPlan(1, 32, 64, 64, 32, 3, 3, (1, 1), algo=hip.ALGO_WINO_F3_X3).kernel.startswith("winograd")


X3_DIRECT_CASES = [   # N, Cin, H, W, Cout, tune_grid: layers the Winograd heuristic leaves direct (conv1_2 / conv2_1 class)
    (1, 64, 64, 96, 64, 0),        # one M tile, fused pooling
    (2, 32, 70, 130, 130, 0),      # batch 2, odd sizes, ragged Cout (3 M tiles of 64)
    (1, 48, 65, 97, 128, 0),       # Cin = 48: three 16-channel chunks
    (2, 64, 72, 200, 128, 100),    # grid of 100 over 252 tiles: 2 whole tiles per workgroup + a stream-K phase with fix-up
]


@pytest.mark.parametrize("case", X3_DIRECT_CASES)
def test_conv_direct_x3(hip, orc, case):
    """Split-fp16 direct 3x3 implicit GEMM (the igemm template with hi + lo operand tiles, variant 210) -- the form
    MSCNN_CONV_ALGO_WINO_F3_X3 selects where the Winograd heuristic stays direct: 1e-4 against the oracle, no worse than 3x the
    fp32 MFMA kernel's error + 2e-6, fused 2x2 pooling bit-identical to pooling y, and max |y| published exactly."""
    N, Cin, H, W, Cout, grid = case
    rng = np.random.default_rng(31)
    x = np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32) * 4.0
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F3_X3, tune_grid=grid)
    assert plan.kernel == "igemm16x3_64x256_k3x3_tw32" and plan.dtype == "f16x3" and plan.can_pool and plan.publishes_amax
    assert plan.executed_flops == 3 * plan.flops
    plan.pack(dev(w))
    slots = torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    plan.set_amax_io(None, slots)
    yp = torch.full((N, Cout, (H + 1) // 2, (W + 1) // 2), float("nan"), device="cuda")
    y = plan.forward(dev(x), dev(b), pool_out=yp)
    assert torch.equal(yp, hip.pool2d(y, (2, 2), (0, 0), (2, 2)))
    assert slots.max().item() == y.abs().max().view(torch.int32).item()              # published max |y|, bit pattern
    ref = orc.relu(orc.conv2d(x, w, b, (1, 1)))
    close(y.cpu().numpy(), ref)
    p32 = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT, tune_grid=grid)
    p32.pack(dev(w))
    s32 = torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    p32.set_amax_io(None, s32)
    y32 = p32.forward(dev(x), dev(b))
    assert s32.max().item() == y32.abs().max().view(torch.int32).item()              # the fp32 kernels publish too
    truth = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                                                  padding=1)).numpy()
    e3 = float((np.abs(y.cpu().numpy() - truth) / np.maximum(1, np.abs(truth))).max())
    e32 = float((np.abs(y32.cpu().numpy() - truth) / np.maximum(1, np.abs(truth))).max())
    print(f"direct x3 err {e3:.2e}  fp32 igemm err {e32:.2e}")
    assert e3 <= 3 * e32 + 2e-6
    # handed-over bound instead of the plan's own max |x| pass: any upper bound gives the same result up to the split's rounding
    bound = torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    bound[7] = torch.tensor(float(np.abs(x).max()) * 1.7, dtype=torch.float32).view(torch.int32)
    plan.set_amax_io(bound, None)
    close(plan.forward(dev(x), dev(b)).cpu().numpy(), ref)


def test_winograd_layers_publish_amax(hip):
    """The F(3x3,3x3) output transforms (plain and fused-pooling, fp32 and x3 GEMM) publish the exact max |y|."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((1, 64, 24, 48), device="cuda", generator=g)
    w = torch.randn((96, 64, 3, 3), device="cuda", generator=g) * 0.05
    b = torch.randn(96, device="cuda", generator=g)
    for algo, flags in ((hip.ALGO_WINO_F3, 0), (hip.ALGO_WINO_F3_X3, 4)):
        for pooled in (False, True):
            plan = hip.ConvPlan(1, 64, 24, 48, 96, 3, 3, (1, 1), relu=False, algo=algo, tune_flags=flags)
            assert plan.publishes_amax
            plan.pack(w)
            slots = torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda")
            plan.set_amax_io(None, slots)
            yp = torch.empty((1, 96, 12, 24), device="cuda") if pooled else None
            y = plan.forward(x, b, pool_out=yp)
            assert slots.max().item() == y.abs().max().view(torch.int32).item(), (algo, pooled)
    # the Cin = 3 VALU kernel (conv1_1) publishes too (conv1_2's split-fp16 plan no longer measures 283 MB itself)
    x3c = torch.randn((1, 3, 72, 136), device="cuda", generator=g) * 50
    w3c = torch.randn((64, 3, 3, 3), device="cuda", generator=g) * 0.05
    b3c = torch.randn(64, device="cuda", generator=g)
    for relu in (False, True):
        c3 = hip.ConvPlan(1, 3, 72, 136, 64, 3, 3, (1, 1), relu=relu)
        assert c3.kernel == "conv3x3_c3_valu_f32" and c3.publishes_amax
        c3.pack(w3c)
        y0 = c3.forward(x3c, b3c).clone()                       # (no slots: nothing published, same bytes)
        slots = torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda")
        c3.set_amax_io(None, slots)
        y = c3.forward(x3c, b3c)
        assert torch.equal(y, y0) and slots.max().item() == y.abs().max().view(torch.int32).item(), relu
    head = hip.ConvPlan(1, 512, 36, 60, 9, 5, 5, (2, 2))
    assert not head.publishes_amax
    with pytest.raises(hip.MscnnError):
        head.set_amax_io(None, torch.zeros(hip.AMAX_SLOTS, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("case", [(20, 64, 7, 7, 48, 0), (33, 32, 7, 5, 130, 0), (16, 32, 8, 4, 32, 1), (50, 1024, 7, 7, 512, 0)])
def test_conv_winograd_x3_roi_maps(hip, orc, case):
    """Split-fp16 F(3x3,3x3) on the ROI-pooled maps (roi_c1), incl. a changing ROI count; 1e-4 against the oracle."""
    R, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(99)
    x = np.maximum(rng.standard_normal((R, Cin, H, W)), 0).astype(np.float32) * 2.0
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(R, Cin, H, W, Cout, 3, 3, (pad, pad), relu=True, algo=hip.ALGO_WINO_F3_X3, tune_flags=4)
    assert plan.kernel.startswith("winograd_f3x3_3x3_x3f16") and not plan.can_pool
    plan.pack(dev(w))
    y = plan.forward(dev(x), dev(b)).cpu().numpy()
    ref = orc.relu(orc.conv2d(x, w, b, (pad, pad)))
    close(y, ref)
    plan.set_batch(R + 7)
    x2 = np.concatenate([x, x[:7]], 0)
    close(plan.forward(dev(x2), dev(b)).cpu().numpy(), np.concatenate([ref, ref[:7]], 0))


def test_conv_winograd_x3_fused_pool_and_dynamic_scale(hip, orc):
    """The x3 path shares the fp32 output transforms (fused 2x2 pooling included); the activation scale is measured on the
    device every forward, so frames whose magnitudes differ by 1e6 run through one plan without overflow or loss."""
    N, Cin, H, W, Cout = 1, 64, 12, 24, 130
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F3_X3, tune_flags=4)
    assert plan.can_pool
    plan.pack(dev(w))
    for scale in (1.0, 1e4, 1e-2, 3e4):
        x = (rng.standard_normal((N, Cin, H, W)) * scale).astype(np.float32)
        bb = (b * scale).astype(np.float32)
        yp = torch.full((N, Cout, H // 2, W // 2), float("nan"), device="cuda")
        y = plan.forward(dev(x), dev(bb), pool_out=yp)
        ref = orc.relu(orc.conv2d(x, w, bb, (1, 1)))
        err = float((np.abs(y.cpu().numpy() - ref) / np.maximum(scale, np.abs(ref))).max())    # bound relative to the frame's scale
        assert err < 1e-4, (scale, err)
        assert torch.equal(yp, hip.pool2d(y, (2, 2), (0, 0), (2, 2)))
    x0 = np.zeros((N, Cin, H, W), np.float32)                                           # all-zero frame: scale 1, y = relu(bias)
    y = plan.forward(dev(x0), dev(b)).cpu().numpy()
    assert np.array_equal(y, np.broadcast_to(np.maximum(b, 0)[None, :, None, None], y.shape))


@pytest.mark.parametrize("shape", [(1, 512, 72, 240, 512, 1), (1, 128, 288, 960, 128, 1), (700, 1024, 7, 7, 512, 0)])
def test_winograd_full_size_matches_direct(hip, shape):
    """BASELINE.json sizes (conv4_2, conv2_2, roi_c1 at R = 700), too large for the CPU oracle in a unit test: the Winograd
    F(3x3,3x3) path against the direct implicit-GEMM path (itself oracle-checked at small sizes) on the same device data,
    1e-4; plus linearity of the Winograd path, y(2a + b) = 2 y(a) + y(b), which needs no second implementation at all."""
    N, Cin, H, W, Cout, pad = shape
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.relu(torch.randn((N, Cin, H, W), device="cuda", generator=g))
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    outs = {}
    for mode in ("0", "1", "x3"):
        plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad),
                            algo={"0": hip.ALGO_DIRECT, "1": hip.ALGO_AUTO, "x3": hip.ALGO_WINO_F3_X3}[mode])
        # (the x3 mode runs conv2_2 on its direct split-fp16 kernel and the other two shapes as Winograd with the split GEMM)
        assert (plan.dtype == "f16x3") == (mode == "x3") and (mode == "x3" or plan.kernel.startswith("winograd_f") == (mode == "1"))
        plan.pack(w)
        outs[mode] = plan.forward(x).clone()
        if mode == "1":
            x2 = torch.relu(torch.randn((N, Cin, H, W), device="cuda", generator=g))
            lin = plan.forward(2 * x + x2).clone()
            y2 = plan.forward(x2).clone()
            err = ((lin - (2 * outs["1"] + y2)).abs() / torch.clamp((2 * outs["1"] + y2).abs(), min=1.0)).max().item()
            assert err < 3e-4, err          # three independent roundings, at up to 3x the input scale
            print(f"linearity err {err:.2e}")
        torch.cuda.synchronize()
    r = outs["0"].double()
    for mode in ("1", "x3"):
        err = ((outs[mode].double() - r).abs() / torch.clamp(r.abs(), min=1.0)).max().item()
        print(f"winograd[{mode}] vs direct err {err:.2e}, |y|max {r.abs().max().item():.1f}")
        assert err < 1e-4, (mode, err)


# ---- Winograd robustness over input / filter statistics --------------------------------------------------------------------
def _stat_inputs(kind, rng, shape):
    """Activation statistics beyond the unit-scale He-normal data of the other tests."""
    N, C, H, W = shape
    g = rng.standard_normal(shape)
    if kind == "relu_unit":
        x = np.maximum(g, 0)
    elif kind == "scale_1e-2":
        x = np.maximum(g, 0) * 1e-2
    elif kind == "scale_1e3":
        x = np.maximum(g, 0) * 1e3
    elif kind == "lognormal":                      # heavy-tailed: a few activations 50x the median
        x = np.exp(1.5 * g) * (rng.uniform(size=shape) < 0.5)
    elif kind == "dc_offsets":                     # per-channel DC of +-100 under unit noise (the mean-subtracted frame is +-128)
        x = g + rng.choice([-100.0, 100.0], size=(1, C, 1, 1))
    elif kind == "sparse_spikes":                  # mostly zero, isolated large values
        x = (rng.uniform(size=shape) < 0.02) * np.abs(g) * 30
    else:
        raise KeyError(kind)
    return x.astype(np.float32)


def _stat_filters(kind, rng, cout, cin):
    w = rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))
    if kind == "he":
        pass
    elif kind == "vgg_like":                       # smooth centre-weighted taps, log-normal per-filter gain, a few dead filters
        tap = np.array([[0.5, 1.0, 0.5], [1.0, 2.0, 1.0], [0.5, 1.0, 0.5]]) / 2.0
        w = w * tap * np.exp(0.8 * rng.standard_normal((cout, 1, 1, 1)))
        w[rng.uniform(size=cout) < 0.05] = 0
    elif kind == "zero_mean":                      # edge-like filters: every 3x3 kernel sums to zero (cancels the input DC)
        w = w - w.mean(axis=(2, 3), keepdims=True)
    else:
        raise KeyError(kind)
    return w.astype(np.float32)


ROBUST_SHAPES = [(1, 128, 36, 60, 128), (1, 512, 18, 30, 512)]          # conv2_2- and conv5-like channel counts
ROBUST_INPUTS = ["relu_unit", "scale_1e-2", "scale_1e3", "lognormal", "dc_offsets", "sparse_spikes"]
ROBUST_FILTERS = ["he", "vgg_like", "zero_mean"]


@pytest.mark.parametrize("shape", ROBUST_SHAPES)
@pytest.mark.parametrize("xkind", ROBUST_INPUTS)
@pytest.mark.parametrize("wkind", ROBUST_FILTERS)
def test_winograd_robustness_over_statistics(hip, shape, xkind, wkind):
    """F(3x3,3x3) has ~10x the rounding error of the direct sum; on unit-scale data that leaves 4x headroom to the 1e-4 bound,
    but the bound is absolute for |y| <= 1 and the error scales with the data.  Contract checked here, for every
    distribution: the algorithm the runtime ends up with after its calibration step (Winograd when it stays within 5e-5 of
    the direct kernel on the data, the direct kernel otherwise -- Net::CalibrateNumerics) is within 1e-4 of the float64
    truth whenever the direct fp32 sum itself is; i.e. Winograd is never the reason a layer misses the bound."""
    N, Cin, H, W, Cout = shape
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{xkind}|{wkind}|{Cin}".encode()))
    x = _stat_inputs(xkind, rng, (N, Cin, H, W))
    w = _stat_filters(wkind, rng, Cout, Cin)
    truth = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
    ys = {}
    for name, algo in (("direct", hip.ALGO_DIRECT), ("wino", hip.ALGO_WINO_F3), ("x3", hip.ALGO_WINO_F3_X3), ("wino4", hip.ALGO_WINO_F4)):
        plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), algo=algo, tune_flags=4 if name == "x3" else 0)
        assert plan.kernel.startswith("winograd_f4x4" if name == "wino4" else "winograd_f3x3") == (name != "direct")
        plan.pack(dev(w))
        ys[name] = plan.forward(dev(x)).cpu().numpy().astype(np.float64)
        torch.cuda.synchronize()
    metric = lambda a, b: float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max())     # noqa: E731  (the parity metric)
    e_direct, e_wino, e_cal = metric(ys["direct"], truth), metric(ys["wino"], truth), metric(ys["wino"], ys["direct"])
    e_x3, e_cal_x3 = metric(ys["x3"], truth), metric(ys["x3"], ys["direct"])
    chosen = "wino" if e_cal <= 5e-5 else "direct"
    e_chosen = e_wino if chosen == "wino" else e_direct
    print(f"\nROBUST Cin={Cin:4d} x={xkind:14s} w={wkind:9s} |y|max {np.abs(truth).max():10.3g}  direct {e_direct:.2e}  "
          f"winograd {e_wino:.2e}  wino-vs-direct {e_cal:.2e}  -> {chosen} ({e_chosen:.2e})")
    if e_direct < 1e-4:
        assert e_chosen < 1e-4, (chosen, e_chosen)
    else:       # fp32 itself cannot meet an ABSOLUTE 1e-4 on this data (scale 1e3): the chosen path must not be worse than 2x direct
        assert e_chosen <= 2 * e_direct + 1e-4
    # The split-fp16 GEMM under the same contract.  Its products carry 22-bit operands (fp32 MFMA: exact products, one rounding per
    # accumulate), so where a few huge terms dominate a sum (heavy tails, isolated spikes) it is up to ~4.5x the fp32 Winograd
    # error -- and the same calibration step sends exactly those layers back to the direct fp32 kernel.
    # F(4x4,3x3) (the AUTO choice of the large layers since round 3) under the same contract; with the points {0, 1, -1, 2, -1/2, inf}
    # its error is 0.8 .. 3.2x the F(3x3,3x3) form's over the 36 distributions of this table (median 1.2x; profiles/r03_robustness.txt)
    e_w4, e_cal4 = metric(ys["wino4"], truth), metric(ys["wino4"], ys["direct"])
    chosen4 = "wino4" if e_cal4 <= 5e-5 else "direct"
    e_chosen4 = e_w4 if chosen4 == "wino4" else e_direct
    print(f"       F(4x4,3x3) {e_w4:.2e} ({e_w4 / max(e_wino, 1e-12):.1f}x F(3x3,3x3))  vs-direct {e_cal4:.2e}  -> {chosen4} ({e_chosen4:.2e})")
    if e_direct < 1e-4:
        assert e_chosen4 < 1e-4, (chosen4, e_chosen4)
    else:
        assert e_chosen4 <= 2 * e_direct + 1e-4
    assert e_w4 <= 3.5 * e_wino + 2e-6, (e_w4, e_wino)
    chosen3 = "x3" if e_cal_x3 <= 5e-5 else "direct"
    e_chosen3 = e_x3 if chosen3 == "x3" else e_direct
    print(f"       x3 {e_x3:.2e} ({e_x3 / max(e_wino, 1e-12):.1f}x winograd)  x3-vs-direct {e_cal_x3:.2e}  -> {chosen3} ({e_chosen3:.2e})")
    assert e_x3 <= 6 * e_wino + 2e-6, (e_x3, e_wino)
    if e_direct < 1e-4:
        assert e_chosen3 < 1e-4, (chosen3, e_chosen3)
    else:
        assert e_chosen3 <= 2 * e_direct + 1e-4


POOL_CASES = [   # N, Cin, H, W, Cout, winograd (0: direct igemm, 2: F(2x2,3x3), 3: F(3x3,3x3))
    (1, 40, 12, 24, 130, 3),      # F(3x3,3x3): 4 x 8 tiles -> 2 x 4 groups of 6x6 outputs
    (2, 24, 18, 36, 32, 3),       # F(3x3,3x3), batch 2
    (1, 16, 16, 32, 128, 0),      # igemm 128x128 tw16: exact tiles
    (1, 8, 13, 37, 64, 0),        # igemm 64x256 tw32: odd H and W (clipped ceil-mode windows at the edges)
    (2, 24, 36, 120, 256, 0),     # conv5-shaped plane: stream-K split tiles go through the pooled fix-up kernel
    (1, 16, 11, 18, 130, 0),      # Cout ragged
    (1, 40, 13, 21, 130, 2),      # Winograd path, odd sizes
    (2, 24, 10, 14, 32, 2),       # Winograd path, batch 2
]


@pytest.mark.parametrize("case", POOL_CASES)
def test_conv_fused_pool(hip, orc, case):
    """Conv + ReLU with the following MAX 2x2/2 PoolingLayer fused into the epilogue: y unchanged, pooled output
    bit-identical to the stand-alone pooling kernel on y, and equal to the oracle's pooling of y."""
    N, Cin, H, W, Cout, wino = case
    algo = {0: hip.ALGO_DIRECT, 2: hip.ALGO_WINO_F2, 3: hip.ALGO_WINO_F3}[wino]
    rng = np.random.default_rng(77)
    x = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=algo)
    assert plan.can_pool and plan.kernel == {0: plan.kernel, 2: "winograd_f2x2_3x3", 3: "winograd_f3x3_3x3"}[wino]
    assert plan.kernel.startswith("winograd") == (wino != 0)
    plan.pack(dev(w))
    y0 = plan.forward(dev(x), dev(b)).clone()
    yp = torch.full((N, Cout, (H + 1) // 2, (W + 1) // 2), float("nan"), device="cuda")
    y1 = plan.forward(dev(x), dev(b), pool_out=yp)
    assert torch.equal(y0, y1)
    ref = hip.pool2d(y0, (2, 2), (0, 0), (2, 2))
    assert torch.equal(yp, ref)
    assert np.array_equal(yp.cpu().numpy(), orc.pool2d(y0.cpu().numpy(), (2, 2), (0, 0), (2, 2), "MAX"))
    close(y0.cpu().numpy(), orc.relu(orc.conv2d(x, w, b, (1, 1))))
    assert not hip.ConvPlan(1, 512, 72, 240, 9, 5, 5, (2, 2)).can_pool       # proposal-head kernel: no pooling epilogue
    p4 = hip.ConvPlan(1, 512, 72, 240, 512, 3, 3, (1, 1))
    assert p4.kernel == "winograd_f4x4_3x3" and p4.can_pool                    # AUTO: 18 x 60 tiles of 4x4 -> F(4x4,3x3), pooling inside a tile
    p3 = hip.ConvPlan(1, 512, 36, 120, 512, 3, 3, (1, 1))
    assert p3.kernel == "winograd_f3x3_3x3" and p3.can_pool                    # 12 x 40 tiles: even -> fused pooling
    assert not hip.ConvPlan(1, 512, 39, 120, 512, 3, 3, (1, 1)).can_pool       # 13 tile rows: the caller pools separately


@pytest.mark.parametrize("cin,grid,extra", [(64, 512, 1), (64, 768, 2), (32, 500, 1), (64, 300, 1)])
@pytest.mark.parametrize("pool", [False, True])
def test_conv_stream_k_remainder_spread_over_a_large_grid(hip, orc, cin, grid, extra, pool):
    """`extra` remainder tiles on a grid of `grid` workgroups -- fewer stream-K iterations than workgroups, so the
    contributors of a split tile are spread over a span of up to the whole grid with empty ranges in between; the fix-up's list used
    to be cut at the first 256 workgroups of the SPAN (4 of 8 slabs summed at KI = 8 on 512).  Both fix-up kernels (plain and with the
    fused pooling) against the oracle, tile count = grid + extra exactly, deterministic over repeats."""
    probe = hip.ConvPlan(1, cin, 64, 128, 64, 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT, tune_flags=32768)
    name = probe.kernel                                   # "igemm_<BM>x<BN>_k3x3_tw<TW>..." : the tile is BN / TW rows of TW pixels
    assert name.startswith("igemm_"), name
    bn = int(name.split("_")[1].split("x")[1]); tw = int(name.split("tw")[1].split("_")[0]); th = bn // tw
    tiles = grid + extra
    nth = next(d for d in range(int(tiles ** 0.5), 0, -1) if tiles % d == 0)
    H, W = nth * th, (tiles // nth) * tw
    rng = np.random.default_rng(grid + cin)
    x = rng.standard_normal((1, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((64, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    p = hip.ConvPlan(1, cin, H, W, 64, 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT, tune_flags=32768, tune_grid=grid)
    assert p.kernel == name
    p.pack(dev(w))
    xd, bd = dev(x), dev(b)
    yp = torch.full((1, 64, (H + 1) // 2, (W + 1) // 2), float("nan"), device="cuda") if pool else None
    y = p.forward(xd, bd, pool_out=yp).clone()
    ref = orc.relu(orc.conv2d(x, w, b, (1, 1)))
    close(y.cpu().numpy(), ref)
    if pool:
        assert np.array_equal(yp.cpu().numpy(), orc.pool2d(y.cpu().numpy(), (2, 2), (0, 0), (2, 2), "MAX"))
    for _ in range(3):
        assert torch.equal(p.forward(xd, bd, pool_out=yp), y)


def test_conv_no_bias_and_kernel_selection(hip, orc):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 8, 8, 16)).astype(np.float32)
    w = rng.standard_normal((32, 8, 3, 3)).astype(np.float32) * 0.1
    plan = hip.ConvPlan(1, 8, 8, 16, 32, 3, 3, (1, 1))
    assert plan.kernel.startswith("igemm_")
    plan.pack(dev(w))
    close(plan.forward(dev(x)).cpu().numpy(), orc.conv2d(x, w, None, (1, 1)))
    assert hip.ConvPlan(1, 3, 8, 16, 32, 3, 3, (1, 1)).kernel.startswith("igemm_")          # Cin 3 is zero-padded
    assert hip.ConvPlan(1, 8, 8, 16, 32, 3, 3, (1, 1), stride=(2, 2)).kernel == "direct_f32"
    assert hip.ConvPlan(50, 1024, 7, 7, 512, 3, 3).kernel == "winograd_f3x3_3x3"             # roi_c1: F(3x3,3x3)
    assert "roi7x7p0" in hip.ConvPlan(50, 1024, 7, 7, 512, 3, 3, algo=hip.ALGO_DIRECT).kernel   # direct ROI-mode igemm
    assert hip.ConvPlan(1, 512, 72, 240, 9, 7, 7, (3, 3)).kernel == "head_kwfold_shiftadd_f32"     # 63 rows x 17,280 pixels: columns folded into M
    assert hip.ConvPlan(1, 512, 36, 120, 9, 7, 7, (3, 3)).kernel == "head4x4_k7x7_m3x4"     # smaller maps / 5x5: M = 4 MFMA head kernel
    assert hip.ConvPlan(1, 512, 72, 240, 9, 5, 5, (2, 2)).kernel == "head4x4_k5x5_m3x4"
    assert hip.ConvPlan(1, 512, 72, 240, 6, 5, 3, (2, 1)).kernel == "head4x4_k5x3_m2x4"
    assert plan.flops == 2.0 * 32 * 8 * 16 * 8 * 9


def test_conv_identity_asymmetric(hip):
    # transpose-detecting check: delta weights must reproduce a shifted copy of the (asymmetric) input
    Cin = Cout = 32
    x = np.arange(Cin * 8 * 16, dtype=np.float32).reshape(1, Cin, 8, 16) / 100.0
    w = np.zeros((Cout, Cin, 3, 3), np.float32)
    for c in range(Cout):
        w[c, (c + 1) % Cin, 0, 2] = 1.0          # y[c][h][w] = x[c+1][h-1][w+1]
    y = hip.conv2d(dev(x), dev(w), None, (1, 1)).cpu().numpy()
    exp = np.zeros_like(y)
    xs = np.roll(x, -1, axis=1)
    exp[0, :, 1:, :-1] = xs[0, :, :-1, 1:]
    assert np.array_equal(y, exp)


# ---- fp16-operand kernels (MSCNN_CONV_ALGO_F16, BASELINE config 5; no reference counterpart) -----------------------------------
def _fp16_exact(a):
    """Values a fp16 MFMA operand represents exactly: with them the fp16 kernel computes the same products as the fp32 oracle
    (11 + 11 mantissa bits fit fp32), so any layout / indexing bug shows at the 1e-4 bar instead of hiding in rounding noise."""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


F16_CASES = [   # N, Cin, H, W, Cout, pad, pooled
    (1, 16, 8, 16, 128, 1, False),        # one exact 128x128 tile (8 x 16 patch), one k-step
    (1, 32, 12, 40, 130, 1, False),       # ragged tiles, Cout ragged, two chunks
    (1, 24, 13, 21, 64, 1, False),        # Cin % 16 != 0 (zero-filled channels), Cout = 64 -> 64x256 tiles
    (1, 3, 16, 48, 64, 1, True),          # conv1_1 shape: 3 channels in a 16-channel step, fused pooling
    (2, 48, 18, 36, 96, 1, True),         # batch 2, fused 2x2 pooling
    (1, 64, 36, 120, 256, 1, False),      # conv5-like plane: persistent grid with stream-K split tiles + fix-up
    (1, 40, 9, 16, 32, 0, False),         # pad 0
]


@pytest.mark.parametrize("tune_variant", [0, 202, 203])     # 202: the 3-workgroups-per-CU build, 203: 128 x 256 tiles
@pytest.mark.parametrize("case", F16_CASES)
def test_conv_f16_planes(hip, orc, case, tune_variant):
    N, Cin, H, W, Cout, pad, pooled = case
    rng = np.random.default_rng(31)
    x = _fp16_exact(rng.standard_normal((N, Cin, H, W)))
    w = _fp16_exact(rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9)))
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad), relu=True, algo=hip.ALGO_F16, tune_variant=tune_variant)
    assert plan.kernel.startswith("igemm16_") and plan.dtype == "f16", plan.kernel
    assert plan.kernel.endswith("_occ3") == (tune_variant == 202)          # (these shapes are far below the 2000-tile switch)
    assert ("128x256" in plan.kernel) == (tune_variant == 203)
    assert hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (pad, pad)).dtype == "f32"
    plan.pack(dev(w))
    Ho, Wo = plan.out_shape()[2:]
    yp = torch.full((N, Cout, (Ho + 1) // 2, (Wo + 1) // 2), float("nan"), device="cuda") if pooled else None
    y = plan.forward(dev(x), dev(b), pool_out=yp).cpu().numpy()
    ref = orc.relu(orc.conv2d(x, w, b, (pad, pad)))
    close(y, ref)
    if pooled:
        assert plan.can_pool and np.array_equal(yp.cpu().numpy(), orc.pool2d(y, (2, 2), (0, 0), (2, 2), "MAX"))
    # general fp32 data: only rounding of the operands to fp16 separates the two (relative 2^-11 per operand, averaged over K)
    x2 = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    w2 = (rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32)
    plan.pack(dev(w2))
    y2 = plan.forward(dev(x2), dev(b)).cpu().numpy()
    ref2 = orc.relu(orc.conv2d(x2, w2, b, (pad, pad)))
    scale = float(np.sqrt((ref2.astype(np.float64) ** 2).mean()))
    assert np.abs(y2 - ref2).max() / max(scale, 1e-6) < 5e-3


@pytest.mark.parametrize("case", [(33, 64, 7, 7, 48, 0), (20, 40, 7, 5, 130, 0), (16, 32, 8, 4, 64, 1), (150, 1024, 8, 4, 512, 1)])
def test_conv_f16_roi_maps(hip, orc, case):
    """roi_c1 in fp16 mode: ROI-mode tiles (several whole ROI maps per tile), ROI count changing between calls."""
    R, Cin, H, W, Cout, pad = case
    rng = np.random.default_rng(32)
    x = _fp16_exact(np.maximum(rng.standard_normal((R, Cin, H, W)), 0))
    w = _fp16_exact(rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (Cin * 9)))
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(R, Cin, H, W, Cout, 3, 3, (pad, pad), relu=True, algo=hip.ALGO_F16)
    assert plan.kernel.startswith("igemm16_") and "roi" in plan.kernel, plan.kernel
    plan.pack(dev(w))
    ref = orc.relu(orc.conv2d(x, w, b, (pad, pad)))
    close(plan.forward(dev(x), dev(b)).cpu().numpy(), ref)
    plan.set_batch(R - 5)
    close(plan.forward(dev(x[:R - 5]), dev(b)).cpu().numpy(), ref[:R - 5])


@pytest.mark.parametrize("M,N,K", [(1, 2048, 16384), (150, 2048, 16384), (700, 4096, 12800), (33, 64, 72), (257, 320, 1000)])
def test_inner_product_f16(hip, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn((M, K), device="cuda", generator=g).half().float()
    w = (torch.randn((N, K), device="cuda", generator=g) * (2.0 / K) ** 0.5).half().float()
    b = torch.randn(N, device="cuda", generator=g)
    y = hip.inner_product_f16(x, w, b, relu=True)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    err = ((y.double() - ref).abs() / torch.clamp(ref.abs(), min=1.0)).max().item()
    assert err < 1e-4, err                                   # fp16-exact operands: only the fp32 accumulation order differs
    x2 = torch.randn((M, K), device="cuda", generator=g)
    y2 = hip.inner_product_f16(x2, w, b)
    ref2 = x2.double() @ w.double().t() + b.double()
    assert ((y2.double() - ref2).abs().max() / ref2.pow(2).mean().sqrt()).item() < 5e-3
    with pytest.raises(hip.MscnnError):
        hip.inner_product_f16(x[:, :K - 4].contiguous(), w[:, :K - 4].contiguous())     # K % 8 != 0: the fp32 entry point's job


@pytest.mark.parametrize("M,N,K", [(1, 2048, 16384), (150, 2048, 16384), (700, 4096, 12800), (33, 128, 96), (257, 320, 1024), (129, 4096, 800)])
def test_inner_product_x3(hip, orc, M, N, K):
    """Split-fp16 InnerProduct against the float64 product: the fp32 bound (1e-4), and no worse than 3x the fp32 MFMA kernel's
    own error + 2e-6; ragged M (rows padded to 128) and N (320 = 2.5 tiles), k-split slabs summed in order."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.relu(torch.randn((M, K), device="cuda", generator=g)) * 3.0
    w = torch.randn((N, K), device="cuda", generator=g) * (2.0 / K) ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    y3 = hip.inner_product_x3(x, w, b, relu=True)
    y32 = hip.inner_product(x, w, b, relu=True)
    e3 = ((y3.double() - ref).abs() / torch.clamp(ref.abs(), min=1.0)).max().item()
    e32 = ((y32.double() - ref).abs() / torch.clamp(ref.abs(), min=1.0)).max().item()
    print(f"ip x3 err {e3:.2e}  fp32 err {e32:.2e}")
    assert e3 < 1e-4 and e3 <= 3 * e32 + 2e-6, (e3, e32)
    assert torch.equal(y3, hip.inner_product_x3(x, w, b, relu=True))           # deterministic (no atomics in the k-split)
    y0 = hip.inner_product_x3(torch.zeros_like(x), w, b)                         # all-zero input: scale 1, y = bias
    assert torch.equal(y0, b[None, :].expand(M, N))
    with pytest.raises(hip.MscnnError):
        hip.inner_product_x3(x[:, :K - 4].contiguous(), w[:, :K - 4].contiguous())


# ------------------------------------------------------------------ inner product
@pytest.mark.parametrize("M,N,K", [(1, 5, 64), (7, 20, 4096), (3, 128, 256), (130, 192, 1000), (257, 4096, 800), (1, 4096, 12800),
                                   (3, 10, 9000), (5, 70, 33), (700, 20, 4096), (150, 512, 260), (700, 320, 1000), (676, 2, 4096), (676, 8, 4096),
                                   (513, 5, 1000)])
def test_inner_product(hip, orc, M, N, K):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    close(hip.inner_product(dev(x), dev(w), dev(b)).cpu().numpy(), orc.inner_product(x, w, b))
    close(hip.inner_product(dev(x), dev(w), None, relu=True).cpu().numpy(), orc.relu(orc.inner_product(x, w, None)))


@pytest.mark.parametrize("M,N,K", [(676, 8, 4096), (676, 2, 4096), (515, 20, 1000), (9, 5, 4096)])
def test_inner_product_small_n_rows_per_workgroup(hip, M, N, K):
    """cls_pred / bbox_pred (N = classes / 4 x classes outputs over fc6's 4096): the kernel keeps 2 / 4 / 8 rows of x in registers per
    workgroup and does 4, 5 or 8 outputs per pass (mscnn_debug_inner_product_rows; the default is what measured fastest in round 5).
    A row's arithmetic does not depend on either: every form computes the same bits, ragged last workgroup included."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.relu(torch.randn((M, K), device="cuda", generator=g))
    w = torch.randn((N, K), device="cuda", generator=g) * (2.0 / K) ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    try:
        hip.debug_inner_product_rows(2)
        y2 = hip.inner_product(x, w, b).clone()
        hip.debug_inner_product_rows(4)
        y4 = hip.inner_product(x, w, b).clone()
        hip.debug_inner_product_rows(8)
        y8 = hip.inner_product(x, w, b).clone()
    finally:
        hip.debug_inner_product_rows(0)
    ya = hip.inner_product(x, w, b)
    assert torch.equal(y2, y4)
    assert torch.equal(y4, y8) and torch.equal(ya, y4)
    ref = (x.double() @ w.double().T + b.double()).float()
    assert float((ya - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,N,K", [(696, 4096, 12800), (193, 256, 64), (250, 384, 1024), (1000, 2048, 3200), (257, 4096, 800), (700, 256, 32)])
def test_inner_product_on_the_plane_gemm_kernel(hip, orc, M, N, K):
    """mscnn_inner_product_wg_*: fc6-class InnerProduct on wgemm.hip's kernel (x re-packed into the A layout, weights transposed once,
    bias / ReLU in the epilogue, the last row tile ragged: rows past M fall outside the output buffer).  Against the float64 product:
    the fp32 bound, and no worse than the stream-K kernel of gemm.hip (+ 2e-6); deterministic; below 192 rows / N % 128 != 0 refused."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.relu(torch.randn((M, K), device="cuda", generator=g)) * 2.0
    w = torch.randn((N, K), device="cuda", generator=g) * (2.0 / K) ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    assert hip.inner_product_wg_supported(M, N, K)
    assert not hip.inner_product_wg_supported(100, N, K) and not hip.inner_product_wg_supported(M, N + 64, K) and not hip.inner_product_wg_supported(M, N, K + 8)
    ref = x.double() @ w.double().t() + b.double()
    big = torch.full((M + 64, N), 777.0, device="cuda")            # rows past M must stay untouched (the kernel's tiles reach past them)
    _, wt = hip.inner_product_wg(x, w, b, relu=False, out=big)
    assert bool((big[M:] == 777.0).all())
    y = big[:M].clone()
    y_old = hip.inner_product(x, w, b)
    rel = lambda a, r: ((a.double() - r).abs() / torch.clamp(r.abs(), min=1.0)).max().item()      # noqa: E731
    e_new, e_old = rel(y, ref), rel(y_old, ref)
    print(f"ip wg err {e_new:.2e}  stream-K kernel err {e_old:.2e}")
    assert e_new < 1e-4 and e_new <= 1.5 * e_old + 2e-6, (e_new, e_old)
    yr, _ = hip.inner_product_wg(x, w, b, relu=True, wt=wt)
    assert torch.equal(yr, torch.relu(y))                               # the epilogue's ReLU on the very same sums
    y2, _ = hip.inner_product_wg(x, w, b, relu=False, wt=wt)
    assert torch.equal(y2, y)                                           # deterministic (partial sums meet in a fixed order)
    yn, _ = hip.inner_product_wg(x, w, None, relu=False, wt=wt)
    assert rel(yn, ref - b.double()) < 1e-4


@pytest.mark.parametrize("case", [(540, 64, 7, 7, 96, 0), (37, 32, 7, 5, 40, 0), (130, 32, 8, 4, 64, 1), (9, 48, 5, 5, 24, 1), (257, 1024, 7, 7, 512, 0)])
def test_wino_roi_output_transform_bit_identical(hip, case):
    """roi_c1's F(3x3,3x3) output transform through LDS (contiguous runs of y per ROI) against the generic per-tile kernel
    (tune_flags bit 8): the same expressions in the same order -- the same bits; ragged ROI / channel counts, 5x5, 5x3, 8x4 outputs."""
    R, Cin, H, W, Cout, pad = case
    g = torch.Generator(device="cuda").manual_seed(R)
    x = torch.relu(torch.randn((R, Cin, H, W), device="cuda", generator=g))
    w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) * (2.0 / (Cin * 9)) ** 0.5
    b = torch.randn((Cout,), device="cuda", generator=g)
    outs = []
    for flags in (0, 256):
        p = hip.ConvPlan(R, Cin, H, W, Cout, 3, 3, (pad, pad), relu=True, algo=hip.ALGO_WINO_F3, tune_flags=flags)
        assert p.kernel.startswith("winograd_f3x3")
        p.pack(w)
        outs.append(p.forward(x, b).clone())
    assert torch.equal(outs[0], outs[1])
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=pad))
    assert ((outs[0].double() - ref).abs() / torch.clamp(ref.abs(), min=1.0)).max().item() < 1e-4


def test_inner_product_stream_k_is_deterministic(hip):
    """The stream-K InnerProduct (partial tiles through slabs + a fix-up launch, k order): the same bits from launch to launch, with
    other shapes (other tile counts and slab sizes, fp16 weights) through the same library-owned slab buffer in between, and the
    float64 product within the reference's 1e-4."""
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(540, 4096, 12800), (257, 320, 1024), (700, 4096, 12800), (130, 2048, 4096)]
    data = []
    for M, N, K in shapes:
        x = torch.relu(torch.randn((M, K), device="cuda", generator=g))
        w = torch.randn((N, K), device="cuda", generator=g) * (2.0 / K) ** 0.5
        b = torch.randn(N, device="cuda", generator=g)
        y = hip.inner_product(x, w, b, relu=True).clone()
        ref = torch.relu(x.double() @ w.double().t() + b.double())
        assert ((y.double() - ref).abs() / torch.clamp(ref.abs(), min=1.0)).max().item() < 1e-4
        data.append((x, w, b, y))
    for rep in range(4):
        for x, w, b, y in data:
            assert torch.equal(hip.inner_product(x, w, b, relu=True), y)
        x, w, b, _ = data[rep % len(data)]
        hip.inner_product_f16(x.half().float(), w.half().float(), b)      # the fp16 kernel shares the buffer and the protocol


# ------------------------------------------------------------------ ROI pooling (bit-exact)
def _random_rois(rng, R, img_h, img_w, batch=1):
    x1 = rng.uniform(-40, img_w, R); y1 = rng.uniform(-40, img_h, R)
    w = rng.uniform(1, 500, R); h = rng.uniform(1, 400, R)
    return np.stack([rng.integers(0, batch, R), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


@pytest.mark.parametrize("ph,pw,scale,C,R", [(7, 7, 0.125, 24, 97), (7, 5, 0.25, 40, 33), (8, 4, 0.125, 128, 50), (5, 5, 0.25, 16, 200)])
def test_roipool_pair_bitexact(hip, orc, ph, pw, scale, C, R):
    """roi_pool_org + roi_pool_ctx in one launch (mscnn_roipool_pair_fwd_f32): both channel windows bit-identical to the oracle's
    two poolings (max is exact), incl. ROIs outside the map, the whole map, ROIs wider than the LDS column buffer, both pad orders."""
    rng = np.random.default_rng(11)
    feat = rng.standard_normal((2, C, 36, 150)).astype(np.float32)
    rois = _random_rois(rng, R, 36 / scale, 150 / scale, batch=2)
    rois[0] = [0, -500, -500, -300, -300]
    rois[1] = [1, 0, 0, 150 / scale - 1, 36 / scale - 1]
    rois[2] = [0, 20, 20, 20, 20]
    for pa, pb in ((0.0, 0.25), (0.25, 0.0)):
        y = hip.roipool_pair(dev(feat), dev(rois), ph, pw, scale, pa, pb).cpu().numpy()
        assert np.array_equal(y[:, :C], orc.roipool(feat, rois, ph, pw, scale, pa))
        assert np.array_equal(y[:, C:], orc.roipool(feat, rois, ph, pw, scale, pb))


@pytest.mark.parametrize("ph,pw,scale,pad", [(7, 7, 0.125, 0.0), (7, 7, 0.125, 0.25), (7, 5, 0.25, 0.25), (8, 4, 0.125, 0.0)])
def test_roipool_bitexact(hip, orc, ph, pw, scale, pad):
    rng = np.random.default_rng(7)
    feat = rng.standard_normal((2, 24, 36, 120)).astype(np.float32)
    rois = _random_rois(rng, 97, 36 / scale, 120 / scale, batch=2)
    rois[0] = [0, -500, -500, -300, -300]                      # fully outside -> zeros
    rois[1] = [1, 0, 0, 120 / scale - 1, 36 / scale - 1]       # whole map
    rois[2] = [0, 20, 20, 20, 20]                              # half-away-from-zero rounding case
    y = hip.roipool(dev(feat), dev(rois), ph, pw, scale, pad).cpu().numpy()
    assert np.array_equal(y, orc.roipool(feat, rois, ph, pw, scale, pad))


def test_roipool_wide_rois(hip, orc):
    """ROIs wider than one wave (64 columns) and wider than the kernel's LDS column buffer (512 columns)."""
    rng = np.random.default_rng(9)
    feat = np.maximum(rng.standard_normal((1, 16, 20, 700)), 0).astype(np.float32)
    rois = _random_rois(rng, 40, 80, 2800)
    rois[:, 3] = rois[:, 1] + rng.uniform(200, 2790, 40)       # 50 .. 700 feature columns at scale 0.25
    rois[0] = [0, 0, 0, 2799, 79]; rois[1] = [0, -300, 5, 2500, 60]
    for pad in (0.0, 0.25):
        y = hip.roipool(dev(feat), dev(rois), 7, 7, 0.25, pad).cpu().numpy()
        assert np.array_equal(y, orc.roipool(feat, rois, 7, 7, 0.25, pad))


@pytest.mark.parametrize("bias", [False, True])
def test_deconv_upsample2x_and_roipool_on_it(hip, orc, bias):
    """The "-2x" nets' detection sub-net input: the depthwise 4x4 / stride 2 / pad 1 Deconvolution (one thread per 2x2 output quad)
    against the oracle (1e-6) and bit-identical to the generic depthwise kernel's tap order, then ROI pooling on the up-sampled
    map incl. ROIs wider than the 128-column LDS buffer (pooled in column segments), bit-exact against the oracle."""
    rng = np.random.default_rng(23)
    N, C, H, W = 2, 32, 30, 90
    x = np.maximum(rng.standard_normal((N, C, H, W)), 0).astype(np.float32)
    w = (orc.bilinear_filler((C, 1, 4, 4)) * rng.uniform(0.5, 1.5, (C, 1, 1, 1))).astype(np.float32)      # per-channel gains
    b = rng.standard_normal(C).astype(np.float32) if bias else None
    up = hip.deconv_depthwise(dev(x), dev(w), None if b is None else dev(b), (1, 1), (2, 2))
    close(up.cpu().numpy(), orc.deconv2d(x, w, b, (1, 1), (2, 2), group=C), 1e-6)
    scale = 0.25
    rois = _random_rois(rng, 120, 2 * H / scale, 2 * W / scale, batch=N)
    rois[0] = [0, -900, -900, -700, -700]                                   # fully outside
    rois[1] = [1, 0, 0, 2 * W / scale - 1, 2 * H / scale - 1]               # the whole map: 180 columns = 2 segments
    rois[2] = [0, -40, -40, 200, 150]
    rois[3] = [1, 600, 100, 2 * W / scale + 80, 2 * H / scale + 50]
    for pad in (0.0, 0.25):
        y = hip.roipool(up, dev(rois), 7, 5, scale, pad)
        assert np.array_equal(y.cpu().numpy(), orc.roipool(up.cpu().numpy(), rois, 7, 5, scale, pad))


def test_roipool_concat_window(hip, orc):
    rng = np.random.default_rng(8)
    feat = rng.standard_normal((1, 16, 18, 60)).astype(np.float32)
    rois = _random_rois(rng, 10, 144, 480)
    out = torch.zeros((10, 32, 7, 7), dtype=torch.float32, device="cuda")
    hip.roipool(dev(feat), dev(rois), 7, 7, 0.125, 0.0, out=out, c_total=32, c_offset=0)
    hip.roipool(dev(feat), dev(rois), 7, 7, 0.125, 0.25, out=out, c_total=32, c_offset=16)
    ref = orc.concat_channels([orc.roipool(feat, rois, 7, 7, 0.125, 0.0), orc.roipool(feat, rois, 7, 7, 0.125, 0.25)])
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("ph,pw,scale,pad", [(7, 7, 0.125, 0.0), (7, 7, 0.125, 0.25), (4, 6, 0.25, 0.5)])
def test_roialign(hip, orc, ph, pw, scale, pad):
    rng = np.random.default_rng(17)
    feat = rng.standard_normal((2, 24, 36, 120)).astype(np.float32)
    rois = _random_rois(rng, 97, 36 / scale, 120 / scale, batch=2)
    rois[0] = [0, -500, -500, -300, -300]; rois[1, 3] = rois[1, 1] - 3
    y = hip.roialign(dev(feat), dev(rois), ph, pw, scale, pad).cpu().numpy()
    assert np.array_equal(y, orc.roialign(feat, rois, ph, pw, scale, pad))       # same op order, contraction off: bit-exact


def test_eltwise(hip, orc):
    rng = np.random.default_rng(18)
    xs = [rng.standard_normal((37, 5)).astype(np.float32) for _ in range(3)]
    for op, cf in (("SUM", [0.33333333] * 3), ("SUM", None), ("PROD", None), ("MAX", None)):
        assert np.array_equal(hip.eltwise([dev(x) for x in xs], op, cf).cpu().numpy(), orc.eltwise(xs, op, cf)), op


# ------------------------------------------------------------------ NMS / BoxOutput (index-exact)
def _clustered_boxes(rng, n):
    centers = rng.uniform(0, 1500, (max(1, n // 12), 2))
    c = centers[rng.integers(0, len(centers), n)] + rng.normal(0, 12, (n, 2))
    wh = rng.uniform(20, 200, (n, 2))
    return np.concatenate([c, wh], 1).astype(np.float32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 500, 2000, 3000])
@pytest.mark.parametrize("mode", ["IOU", "IOMU", "IOFU"])
def test_nms_keep_set_bitexact(hip, orc, n, mode):
    boxes = _clustered_boxes(np.random.default_rng(n), n)
    boxes[0, 2] = 0.0 if n > 3 else boxes[0, 2]               # degenerate box: IoU 0 with everything
    k = hip.nms_greedy(dev(boxes), 0.65, mode).cpu().numpy()
    assert np.array_equal(k, orc.nms_greedy(boxes, 0.65, mode))


def test_nms_threshold_strict_on_device(hip):
    boxes = np.array([[0, 0, 10, 10], [0, 0, 10, 5], [100, 100, 5, 5]], np.float32)
    assert hip.nms_greedy(dev(boxes), 0.5).cpu().tolist() == [True, True, True]
    assert hip.nms_greedy(dev(boxes), 0.4999).cpu().tolist() == [True, False, True]


KITTI_HEADS = dict(shapes=[(18, 60), (18, 60), (9, 30), (9, 30), (5, 15), (5, 15), (3, 8)],
                   field=[60, 84, 120, 168, 240, 336, 480], ds=[8, 8, 16, 16, 32, 32, 64])


def _heads(rng, shapes, num=1, cls=5, bg_bias=0.0, sigma=2.0):
    out = []
    for (h, w) in shapes:
        t = rng.standard_normal((num, cls + 4, h, w)).astype(np.float32)
        t[:, :cls] *= sigma
        t[:, 0] += bg_bias
        t[:, cls:] *= 0.5
        out.append(t)
    return out


@pytest.mark.parametrize("regime,bg_bias,max_nms", [("dense", -8.0, 2000), ("sparse", 6.0, 2000), ("trunc", -8.0, 300),
                                                    ("empty", 60.0, 2000), ("k1024", -8.0, 1024), ("k3000", -8.0, 3000), ("k4032", -8.0, 4032)])
def test_boxoutput_selection_bitexact(hip, orc, regime, bg_bias, max_nms):
    rng = np.random.default_rng(1701)
    heads = _heads(rng, KITTI_HEADS["shapes"], bg_bias=bg_bias)
    kw = dict(fg_thr=-5.0, iou_thr=0.65, max_nms_num=max_nms, min_size=15.0)
    rois_r, props_r, cidx, nreal_r, aids_r = orc.boxoutput(heads, KITTI_HEADS["field"], KITTI_HEADS["field"], KITTI_HEADS["ds"],
                                                          with_anchor_ids=True, **kw)
    d = hip.make_boxoutput_desc(KITTI_HEADS["shapes"], 1, 9, KITTI_HEADS["field"], KITTI_HEADS["field"], KITTI_HEADS["ds"], **kw)
    layer = hip.BoxOutput(d)
    rois, props, aids, nreal = layer.forward([dev(h) for h in heads])
    assert nreal == nreal_r and rois.shape[0] == rois_r.shape[0]
    assert np.array_equal(aids.cpu().numpy(), aids_r)                       # NMS index selection: bit-exact
    close(rois.cpu().numpy(), rois_r)
    close(props.cpu().numpy(), props_r)
    if regime != "empty":
        # the libm-expf restatement on the device makes the boxes themselves bit-identical too
        assert np.array_equal(rois.cpu().numpy(), rois_r)
        assert np.array_equal(props.cpu().numpy(), props_r)
    else:
        assert rois.cpu().tolist() == [[0, 1, 1, 10, 10]] and aids.cpu().tolist() == [-1]


def test_boxoutput_batch2_bboxnorm_ties(hip, orc):
    rng = np.random.default_rng(11)
    shapes = [(12, 20), (6, 10)]
    heads = _heads(rng, shapes, num=2, cls=3, bg_bias=-4.0)
    heads[0][1] = heads[0][0]                                   # image 1 == image 0 on head 0
    heads[0][:, 1:3, 4:8, :] = 1.5                              # plateaus of exactly equal scores (tie-break by index)
    kw = dict(fg_thr=-7.0, iou_thr=0.65, max_nms_num=100, min_size=5.0, bbox_mean=[0, 0, 0, 0], bbox_std=[0.1, 0.1, 0.2, 0.2])
    ref = orc.boxoutput(heads, [40, 80], [56, 112], [8, 16], with_anchor_ids=True, **kw)
    d = hip.make_boxoutput_desc(shapes, 2, 7, [40, 80], [56, 112], [8, 16], **kw)
    rois, props, aids, nreal = hip.BoxOutput(d).forward([dev(h) for h in heads])
    assert nreal == ref[3]
    assert np.array_equal(aids.cpu().numpy(), ref[4])
    assert np.array_equal(rois.cpu().numpy(), ref[0])
    assert set(rois[:, 0].cpu().tolist()) == {0.0, 1.0}


def test_boxoutput_full_size_properties(hip):
    """BASELINE config-2 size (45,630 anchors, dense): size-independent properties instead of the oracle."""
    shapes = [(72, 240), (72, 240), (36, 120), (36, 120), (18, 60), (18, 60), (9, 30)]
    rng = np.random.default_rng(3)
    heads = [dev(h) for h in _heads(rng, shapes, bg_bias=-8.0)]
    d = hip.make_boxoutput_desc(shapes, 1, 9, KITTI_HEADS["field"], KITTI_HEADS["field"], KITTI_HEADS["ds"])
    rois, props, aids, nreal = hip.BoxOutput(d).forward(heads)
    R = rois.shape[0]
    assert 1 <= R <= 2000 and nreal == R
    s = props[:, 5].cpu().numpy()
    assert np.all(np.diff(s) <= 0)                                              # score order
    r = rois.cpu().numpy()
    assert np.all(r[:, 1] >= 0) and np.all(r[:, 2] >= 0) and np.all(r[:, 3] <= 1920) and np.all(r[:, 4] <= 576)
    assert np.all(r[:, 3] - r[:, 1] >= 15 - 1e-3) and np.all(r[:, 4] - r[:, 2] >= 15 - 1e-3)
    xywh = torch.stack([rois[:, 1], rois[:, 2], rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]], 1).contiguous()
    assert bool(hip.nms_greedy(xywh, 0.6501).all())                              # idempotence: nothing left to suppress
    assert len(set(aids.cpu().tolist())) == R


# ---- more than 4032 boxes into NMS: the tiled path (csrc/nms_large.h) -- global bitonic sort + per-tile bit matrix + cross-tile
# suppression.  Same bar as the LDS-resident path: the reference's keep set / rows exactly.
@pytest.mark.parametrize("n,mode", [(4033, "IOU"), (4096, "IOMU"), (8064, "IOU"), (9001, "IOFU")])
def test_nms_keep_set_bitexact_tiled(hip, orc, n, mode):
    boxes = _clustered_boxes(np.random.default_rng(n), n)
    boxes[7, 2] = 0.0                                         # degenerate box: IoU 0 with everything
    k = hip.nms_greedy(dev(boxes), 0.65, mode).cpu().numpy()
    assert np.array_equal(k, orc.nms_greedy(boxes, 0.65, mode))
    assert k.sum() > 200 and k[4032:].any() and (n < 8000 or (~k[4032:]).any())      # kept / suppressed boxes beyond the first tile


BIG_HEADS = dict(shapes=[(36, 120), (36, 120), (18, 60)], field=[60, 84, 120], ds=[8, 8, 16])      # 9,720 anchors


@pytest.mark.parametrize("name,bg_bias,kw", [
    ("uncapped_dense", -8.0, dict(max_nms_num=0)),             # caffe.proto default: every candidate enters NMS
    ("uncapped_sparse", 9.0, dict(max_nms_num=0)),             # large path chosen by the anchor count, few candidates
    ("cap_6000", -8.0, dict(max_nms_num=6000)),                # top-K above 4032: cut after the global sort
    ("uncapped_post_500", -8.0, dict(max_nms_num=0, max_post_nms_num=500)),
    ("uncapped_iomu", -8.0, dict(max_nms_num=0, nms_type="IOMU")),
])
def test_boxoutput_more_than_4032_candidates_bitexact(hip, orc, name, bg_bias, kw):
    rng = np.random.default_rng(77)
    heads = _heads(rng, BIG_HEADS["shapes"], bg_bias=bg_bias)
    kw = dict(fg_thr=-5.0, iou_thr=0.65, min_size=15.0, **kw)
    ref = orc.boxoutput(heads, BIG_HEADS["field"], BIG_HEADS["field"], BIG_HEADS["ds"], with_anchor_ids=True, **kw)
    d = hip.make_boxoutput_desc(BIG_HEADS["shapes"], 1, 9, BIG_HEADS["field"], BIG_HEADS["field"], BIG_HEADS["ds"], **kw)
    rois, props, aids, nreal = hip.BoxOutput(d).forward([dev(h) for h in heads])
    assert nreal == ref[3] and rois.shape[0] == ref[0].shape[0]
    assert np.array_equal(aids.cpu().numpy(), ref[4])
    assert np.array_equal(rois.cpu().numpy(), ref[0]) and np.array_equal(props.cpu().numpy(), ref[1])
    if name == "uncapped_dense":
        assert rois.shape[0] > 4032                            # more rows out than one tile holds


def test_boxoutput_uncapped_batch2(hip, orc):
    rng = np.random.default_rng(78)
    heads = _heads(rng, BIG_HEADS["shapes"], num=2, bg_bias=-8.0)
    heads[1][1, 0] += 9.0                                      # image 1: a sparse head
    kw = dict(fg_thr=-5.0, iou_thr=0.65, max_nms_num=0, min_size=15.0)
    ref = orc.boxoutput(heads, BIG_HEADS["field"], BIG_HEADS["field"], BIG_HEADS["ds"], with_anchor_ids=True, **kw)
    d = hip.make_boxoutput_desc(BIG_HEADS["shapes"], 2, 9, BIG_HEADS["field"], BIG_HEADS["field"], BIG_HEADS["ds"], **kw)
    layer = hip.BoxOutput(d)
    for _ in range(2):                                         # second forward: counters / kept list start clean
        rois, props, aids, nreal = layer.forward([dev(h) for h in heads])
        assert nreal == ref[3] and np.array_equal(aids.cpu().numpy(), ref[4]) and np.array_equal(rois.cpu().numpy(), ref[0])


def test_boxoutput_uncapped_full_size(hip, orc):
    """BASELINE config-2 head sizes (45,630 anchors), max_nms_num 0, about a third of the anchors pass fg_thr."""
    shapes = [(72, 240), (72, 240), (36, 120), (36, 120), (18, 60), (18, 60), (9, 30)]
    rng = np.random.default_rng(5)
    heads = _heads(rng, shapes, bg_bias=-1.5)
    kw = dict(fg_thr=-5.0, iou_thr=0.65, max_nms_num=0, min_size=15.0)
    ref = orc.boxoutput(heads, KITTI_HEADS["field"], KITTI_HEADS["field"], KITTI_HEADS["ds"], with_anchor_ids=True, **kw)
    d = hip.make_boxoutput_desc(shapes, 1, 9, KITTI_HEADS["field"], KITTI_HEADS["field"], KITTI_HEADS["ds"], **kw)
    rois, props, aids, nreal = hip.BoxOutput(d).forward([dev(h) for h in heads])
    assert nreal == ref[3] and np.array_equal(aids.cpu().numpy(), ref[4])
    assert np.array_equal(props.cpu().numpy(), ref[1])


# ------------------------------------------------------------------ DecodeBBox / final detections
def test_decode_bbox(hip, orc):
    rng = np.random.default_rng(12)
    prior = _random_rois(rng, 200, 576, 1920)
    bbox = rng.standard_normal((200, 8)).astype(np.float32)
    out = hip.decode_bbox(dev(bbox), dev(prior), (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2)).cpu().numpy()
    assert np.array_equal(out, orc.decode_bbox(bbox, prior, (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2)))


@pytest.mark.parametrize("R", [1, 37, 1000, 2000])
def test_detections_stage(hip, orc, R):
    rng = np.random.default_rng(R)
    b = _clustered_boxes(rng, R)
    props = np.concatenate([np.zeros((R, 1), np.float32), b[:, :2], b[:, :2] + b[:, 2:], rng.normal(0, 4, (R, 1)).astype(np.float32)], 1)
    props[::17, 5] = -11.0                                        # below proposal_thr
    props[5::29, 3] = props[5::29, 1]                             # zero width
    bbox_pred = rng.standard_normal((R, 20)).astype(np.float32)
    cls_pred = (rng.standard_normal((R, 5)) * 2).astype(np.float32)
    cls_pred[3::7] = cls_pred[2::7][: len(cls_pred[3::7])]        # exact prob ties -> stable order
    kw = dict(cls_id=2, ratios=(576 / 375, 1920 / 1242), org_hw=(375, 1242))
    dets, ids = hip.detections(dev(bbox_pred), dev(cls_pred), dev(props), **kw)
    dref, iref = orc.detections(bbox_pred, cls_pred, props, **kw)
    assert np.array_equal(ids.cpu().numpy(), iref)               # selection + order: exact
    close(dets.cpu().numpy(), dref)


@pytest.mark.parametrize("R", [4033, 9000])
def test_detections_stage_more_than_4032_rows(hip, orc, R):
    """Final stage over more ROIs than the LDS-resident path holds: sort in HBM + tiled NMS, same rows and order."""
    rng = np.random.default_rng(R)
    b = _clustered_boxes(rng, R)
    props = np.concatenate([np.zeros((R, 1), np.float32), b[:, :2], b[:, :2] + b[:, 2:], rng.normal(0, 4, (R, 1)).astype(np.float32)], 1)
    props[::17, 5] = -11.0
    props[5::29, 3] = props[5::29, 1]
    bbox_pred = rng.standard_normal((R, 20)).astype(np.float32)
    cls_pred = (rng.standard_normal((R, 5)) * 2).astype(np.float32)
    cls_pred[3::7] = cls_pred[2::7][: len(cls_pred[3::7])]        # exact prob ties -> stable order (lower row first)
    kw = dict(cls_id=2, ratios=(576 / 375, 1920 / 1242), org_hw=(375, 1242))
    dets, ids = hip.detections(dev(bbox_pred), dev(cls_pred), dev(props), **kw)
    dref, iref = orc.detections(bbox_pred, cls_pred, props, **kw)
    assert np.array_equal(ids.cpu().numpy(), iref)
    close(dets.cpu().numpy(), dref)


def test_detections_cascade_stage_more_than_4032_rows(hip, orc):
    R = 6000
    rng = np.random.default_rng(6000)
    b = _clustered_boxes(rng, R)
    boxes = np.concatenate([np.zeros((R, 1), np.float32), b[:, :2] - 30, b[:, :2] + b[:, 2:] + 25], 1).astype(np.float32)
    props = np.concatenate([np.zeros((R, 1), np.float32), b[:, :2], b[:, :2] + b[:, 2:]], 1).astype(np.float32)
    props[3::31, 3] = props[3::31, 1] - 1
    prob = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    prob[3::7] = prob[2::7][: len(prob[3::7])]
    kw = dict(cls_id=2, det_thr=0.2, ratios=(576 / 375, 1920 / 1242), org_hw=(375, 1242))
    dets, ids = hip.detections_cascade(dev(boxes), dev(prob), dev(props), **kw)
    dref, iref = orc.detections_cascade(boxes, prob, props, **kw)
    assert np.array_equal(ids.cpu().numpy(), iref)
    close(dets.cpu().numpy(), dref)


@pytest.mark.parametrize("R,det_thr", [(1, 0.0), (41, 0.0), (1500, 0.0), (1500, 0.3)])
def test_detections_cascade_stage(hip, orc, R, det_thr):
    rng = np.random.default_rng(100 + R)
    b = _clustered_boxes(rng, R)
    boxes = np.concatenate([np.zeros((R, 1), np.float32), b[:, :2] - 30, b[:, :2] + b[:, 2:] + 25], 1).astype(np.float32)
    props = np.concatenate([np.zeros((R, 1), np.float32), b[:, :2], b[:, :2] + b[:, 2:]], 1).astype(np.float32)
    props[3::31, 3] = props[3::31, 1] - 1                         # degenerate proposal (width 0 in the +1 convention)
    prob = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    prob[3::7] = prob[2::7][: len(prob[3::7])]                    # exact ties -> stable order
    kw = dict(cls_id=2, det_thr=det_thr, ratios=(576 / 375, 1920 / 1242), org_hw=(375, 1242))
    dets, ids = hip.detections_cascade(dev(boxes), dev(prob), dev(props), **kw)
    dref, iref = orc.detections_cascade(boxes, prob, props, **kw)
    assert np.array_equal(ids.cpu().numpy(), iref)
    assert np.array_equal(dets.cpu().numpy(), dref)              # no transcendental in this stage: bit-exact


# ------------------------------------------------------------------ loud failure without a GPU tensor
def test_no_cpu_fallback(hip):
    with pytest.raises(hip.MscnnError):
        hip.relu(torch.zeros(4))


# ------------------------------------------------------------------ pre-processing (resize + BGR + mean; bit-exact vs the oracle)
@pytest.mark.parametrize("org,out", [((375, 1242), (576, 1920)), ((370, 1224), (384, 1280)), ((96, 130), (40, 57)),
                                     ((50, 200), (120, 210))])
def test_preprocess_bitexact(hip, orc, org, out):
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (*org, 3), dtype=np.uint8)
    y = hip.preprocess(torch.from_numpy(img).cuda(), out[0], out[1]).cpu().numpy()
    assert np.array_equal(y, orc.preprocess(img, out[0], out[1]))


def _kitti_like_rois(rng, R, H8, W8, batch=1):
    """Proposals as BoxOutput leaves them (image coordinates, stride-8 map of H8 x W8): log-uniform widths, some ROIs leaving the image,
    some degenerate (zero / negative size), some far larger than the 32-column LDS segment of the fused kernel."""
    w = np.exp(rng.uniform(np.log(6), np.log(8 * W8 * 0.9), R)); h = w * rng.uniform(0.3, 1.6, R)
    x1 = rng.uniform(-40, 8 * W8 - 10, R); y1 = rng.uniform(-30, 8 * H8 - 10, R)
    rois = np.stack([rng.integers(0, batch, R).astype(np.float64), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    rois[3, 3] = rois[3, 1] - 5.0                  # x2 < x1
    rois[5, 1:] = [8 * W8 + 50, 10, 8 * W8 + 90, 60]      # entirely right of the map: every bin empty
    rois[7, 1:] = [-300, -200, 8 * W8 + 300, 8 * H8 + 200]   # covers everything (several column segments)
    return rois


@pytest.mark.parametrize("case", [(37, 64, 24, 40, 96, 1), (133, 128, 36, 120, 64, 1), (700, 512, 72, 240, 512, 1), (21, 64, 20, 28, 64, 2)])
def test_conv_roipool_pair_fused_is_bit_identical(hip, orc, case):
    """mscnn_conv2d_fwd_roipool_pair_f32 (roipool_wino.hip: both ROI poolings + the F(3x3,3x3) input transform of roi_c1 in one pass
    over a channel-last copy of the map) against the unfused sequence mscnn_roipool_pair_fwd_f32 -> mscnn_conv2d_fwd_f32 with the
    same plan: the pooled values are exact and the transform uses the same expressions, so y must be BIT-IDENTICAL; the pooled
    blob of the unfused path is held to the oracle (roi_pooling_layer.cpp:48-139) bit for bit on a subset of ROIs."""
    R, Cc, H8, W8, Cout, batch = case
    rng = np.random.default_rng(R)
    feat = np.maximum(rng.standard_normal((batch, Cc, H8, W8)), 0).astype(np.float32) * 3.0
    rois = _kitti_like_rois(rng, R, H8, W8, batch)
    w = (rng.standard_normal((Cout, 2 * Cc, 3, 3)) * np.sqrt(2.0 / (2 * Cc * 9))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    plan = hip.ConvPlan(R, 2 * Cc, 7, 7, Cout, 3, 3, (0, 0), relu=True, algo=hip.ALGO_WINO_F3)
    assert plan.kernel == "winograd_f3x3_3x3" and plan.can_fuse_roipool(Cc, 7, 7)
    assert not plan.can_fuse_roipool(Cc, 7, 5) and not plan.can_fuse_roipool(Cc // 2, 7, 7)
    plan.pack(dev(w))
    fd, rd, bd = dev(feat), dev(rois), dev(b)
    pooled = hip.roipool_pair(fd, rd, 7, 7, 0.125, 0.0, 0.25)
    y_ref = plan.forward(pooled, bd).clone()
    y = plan.forward_roipool_pair(fd, rd, 0.125, 0.0, 0.25, bd)
    assert torch.equal(y, y_ref)
    # the maps built ahead of time on another stream (mscnn_roipool_maps_build_f32) and handed in: same bytes, no feature pointer needed
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    maps = hip.roipool_maps(fd, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(plan.forward_roipool_pair(fd, rd, 0.125, 0.0, 0.25, bd, maps=maps), y_ref)
    sub = np.r_[0:min(R, 12), R - 3:R]
    want = np.concatenate([orc.roipool(feat, rois[sub], 7, 7, 0.125, 0.0), orc.roipool(feat, rois[sub], 7, 7, 0.125, 0.25)], 1)
    assert np.array_equal(pooled[torch.as_tensor(sub, device="cuda")].cpu().numpy(), want)
    # a changing ROI count through the same plan (plan_set_batch): a last workgroup with fewer than its four ROIs
    plan.set_batch(R - 6)
    y3_ref = plan.forward(pooled[:R - 6].contiguous(), bd).clone()      # (another ROI count may plan another GEMM schedule: same plan, same batch)
    y3 = plan.forward_roipool_pair(fd, rd[:R - 6].contiguous(), 0.125, 0.0, 0.25, bd)
    assert torch.equal(y3, y3_ref)
    plan.set_batch(5)                               # below 8 ROIs the plan leaves Winograd: nothing to fuse into
    assert not plan.can_fuse_roipool(Cc, 7, 7)


@pytest.mark.parametrize("shape", [(1, 32, 64, 48, 72, 240), (1, 16, 32, 32, 144, 480), (2, 8, 16, 24, 40, 200), (1, 8, 8, 8, 288, 960), (1, 8, 16, 8, 44, 640),
                                   (2, 8, 8, 8, 36, 1280), (1, 8, 8, 8, 24, 1984)])
def test_conv_chain_f4_is_bit_identical(hip, shape):
    """mscnn_conv2d_fwd_chain_f32: three same-resolution F(4x4,3x3) layers with the activation between them never written -- the
    output transform of one layer writes the next one's input-transform planes (winograd.hip: wino44_outin_kernel; 1, 2, 3, 4, 6 and 8
    strips of tile columns, row chunks, a batch of 2).  Every result is bit-identical to the three separate forwards; y written on request
    is the separate forward's y; the tail takes the fused pooling."""
    N, C0, C1, C2, H, W = shape
    rng = np.random.default_rng(5)
    x = dev(rng.standard_normal((N, C0, H, W)).astype(np.float32))
    chans = [C0, C1, C2, C1]
    plans, ws, bs = [], [], []
    for i in range(3):
        p = hip.ConvPlan(N, chans[i], H, W, chans[i + 1], 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F4)
        assert p.kernel == "winograd_f4x4_3x3"
        w = dev((rng.standard_normal((chans[i + 1], chans[i], 3, 3)) * np.sqrt(2.0 / (9 * chans[i]))).astype(np.float32))
        p.pack(w)
        plans.append(p); ws.append(w); bs.append(dev(rng.standard_normal(chans[i + 1]).astype(np.float32)))
    a, b, c = plans
    assert a.can_chain(b) and b.can_chain(c)
    y1 = a.forward(x, bs[0]).clone()
    y2 = b.forward(y1, bs[1]).clone()
    pool_ref = torch.empty((N, chans[3], (H + 1) // 2, (W + 1) // 2), device="cuda")
    y3 = c.forward(y2, bs[2], pool_out=pool_ref).clone()
    # the chain: a (own input transform) -> b (prepared) -> c (prepared, ordinary output + pooling)
    a.forward_chain(x, b, bs[0], write_y=False)
    y2c = b.forward_chain(None, c, bs[1], write_y=True).clone()
    pool = torch.empty_like(pool_ref)
    y3c = c.forward_chain(None, None, bs[2], pool_out=pool)
    assert torch.equal(y2c, y2) and torch.equal(y3c, y3) and torch.equal(pool, pool_ref)
    # the tail with ONLY the pooled blob written (its own y has no other reader in the trunk: conv2_2, conv3_3)
    assert c.can_pool_only
    a.forward_chain(x, b, bs[0], write_y=False)
    b.forward_chain(None, c, bs[1], write_y=False)
    pool2 = torch.zeros_like(pool_ref)
    assert c.forward_chain(None, None, bs[2], pool_out=pool2, write_y=False) is None and torch.equal(pool2, pool_ref)
    with pytest.raises(hip.MscnnError, match="y may be NULL only"):
        c.forward_chain(y2, None, bs[2], write_y=False)
    # a direct-kernel plan does not chain, and the call says so
    d = hip.ConvPlan(N, chans[1], H, W, chans[2], 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT)
    assert not a.can_chain(d)
    with pytest.raises(hip.MscnnError, match="do not chain"):
        a.forward_chain(x, d, bs[0])


def test_conv_pool_only_direct_kernel(hip, orc):
    """mscnn_conv2d_plan_can_pool_only on the direct MFMA kernel (conv1_2's shape class): y == NULL writes the pooled blob only --
    the same bytes as the forward that writes both -- also where tiles are split across workgroups (fix-up launch)."""
    rng = np.random.default_rng(11)
    for (N, Cin, H, W, Cout) in [(1, 64, 96, 160, 64), (2, 32, 33, 47, 48), (1, 64, 576, 1920, 64)]:
        x = dev(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
        w = dev((rng.standard_normal((Cout, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
        b = dev(rng.standard_normal(Cout).astype(np.float32))
        p = hip.ConvPlan(N, Cin, H, W, Cout, 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT, tune_flags=32768)      # (bit 15: conv1_2's shape stays on this kernel)
        assert p.kernel.startswith("igemm_") and p.can_pool and p.can_pool_only, p.kernel
        p.pack(w)
        pool_ref = torch.empty((N, Cout, (H + 1) // 2, (W + 1) // 2), device="cuda")
        y = p.forward(x, b, pool_out=pool_ref)
        pool = torch.zeros_like(pool_ref)
        wsb = p.ws.numel() * 4 if p.ws is not None else 0
        hip._check(hip.lib().mscnn_conv2d_fwd_pool_f32(p._p, hip._dev(x), hip._dev(w), hip._dev(p.packed), hip._dev(b), None, hip._dev(pool),
                                                       hip._dev(p.ws), wsb, hip._stream()))
        assert torch.equal(pool, pool_ref)
        if H * W < 20000:
            ref = orc.relu(orc.conv2d(x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), (1, 1)))
            close(y.cpu().numpy(), ref)
    q = hip.ConvPlan(1, 512, 36, 120, 512, 3, 3, (1, 1), relu=True, algo=hip.ALGO_WINO_F3)
    assert q.can_pool and not q.can_pool_only


@pytest.mark.parametrize("shape", [(1, 8, 8, 32), (2, 16, 16, 96), (1, 64, 64, 256), (3, 40, 24, 64), (1, 64, 576, 1920)])
@pytest.mark.parametrize("relu", [True, False])
def test_wf2conv_one_launch_winograd_f2x2(hip, orc, shape, relu):
    """conv1_2's shape class as ONE-launch Winograd F(2x2,3x3) (wf2conv.hip: input transform, 16 plane products on
    v_mfma_f32_16x16x4_f32 and output transform inside one workgroup; AUTO's choice for full-resolution maps since round 5): against the
    oracle / the direct ring kernel within the fp32 bar, the fused MAX 2x2 pooling bit-identical to pooling the layer's own output, the
    pool-only forward bit-identical to both, image borders (every tile row / column position incl. single-block maps), batch, odd
    channel-chunk counts, with and without bias."""
    N, Cin, H, W = shape
    rng = np.random.default_rng(29)
    x = dev(np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32))
    w = dev((rng.standard_normal((64, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
    b = dev(rng.standard_normal(64).astype(np.float32))
    p = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=relu, tune_variant=403)      # (403: also maps with < 512 blocks)
    assert p.kernel == "winograd2x2_fused_k3x3_c64", p.kernel
    assert p.can_pool and p.can_pool_only and not p.publishes_amax
    assert abs(p.executed_flops / p.flops - 16.0 / 36.0) < 1e-9
    p.pack(w)
    pool = torch.full((N, 64, H // 2, W // 2), -7.0, device="cuda")
    y = p.forward(x, b, pool_out=pool).clone()
    assert torch.equal(pool, torch.nn.functional.max_pool2d(y, 2))
    pool2 = torch.full_like(pool, -9.0)
    hip._check(hip.lib().mscnn_conv2d_fwd_pool_f32(p._p, hip._dev(x), hip._dev(w), hip._dev(p.packed), hip._dev(b), None, hip._dev(pool2),
                                                   None, 0, hip._stream()))
    assert torch.equal(pool2, pool)
    for _ in range(2):
        assert torch.equal(p.forward(x, b), y)
    d = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=relu, algo=hip.ALGO_DIRECT)
    d.pack(w)
    yd = d.forward(x, b)
    err = ((y - yd).abs() / torch.clamp(yd.abs(), min=1.0)).max().item()
    assert err < 2e-5, err
    if H * W <= 20000:
        ref = orc.conv2d(x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), (1, 1))
        close(y.cpu().numpy(), orc.relu(ref) if relu else ref)
        y0 = p.forward(x)                                  # no bias
        ref0 = orc.conv2d(x.cpu().numpy(), w.cpu().numpy(), np.zeros(64, np.float32), (1, 1))
        close(y0.cpu().numpy(), orc.relu(ref0) if relu else ref0)


@pytest.mark.parametrize("shape", [(1, 8, 8, 128), (2, 16, 12, 256), (1, 64, 64, 1024), (3, 40, 16, 128), (1, 48, 8, 384), (1, 64, 576, 1920)])
def test_wconv_ring_kernel_against_the_igemm_kernel(hip, orc, shape):
    """conv1_2's shape class on the ring kernel of wconv.hip (one 8-wave workgroup per CU, LDS-DMA ring with dword patch pieces,
    ride-along epilogue; AUTO's choice for full-resolution maps) against the igemm kernel it replaces (tune_flags bit 15), on the SAME packed weights: bit-identical --
    y, the fused 2x2 pooling, the pool-only forward, with and without bias / ReLU --; small cases against the oracle."""
    N, Cin, H, W = shape
    rng = np.random.default_rng(17)
    x = dev(np.maximum(rng.standard_normal((N, Cin, H, W)), 0).astype(np.float32))
    w = dev((rng.standard_normal((64, Cin, 3, 3)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
    b = dev(rng.standard_normal(64).astype(np.float32))
    # (the igemm kernel on a grid that divides its 8 x 32-pixel tiles: whole tiles there too -- its stream-K split associates the
    # partial sums of a split tile differently)
    tiles_ig = N * ((H + 7) // 8) * ((W + 31) // 32)
    grid_ig = max(g for g in range(1, min(tiles_ig, 768) + 1) if tiles_ig % g == 0)
    p = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT, tune_variant=402)
    q = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=True, algo=hip.ALGO_DIRECT, tune_flags=32768, tune_grid=grid_ig)
    assert p.kernel == "wconv_64x512_k3x3" and q.kernel.startswith("igemm_64x256"), (p.kernel, q.kernel)
    # AUTO takes it where the map has at least two tiles per CU (conv1_2 of the deploy nets), never below
    # (round 5: where the map is whole 8 x 32 blocks, >= 512 of them, AUTO prefers the one-launch Winograd F(2x2,3x3) kernel of wf2conv.hip;
    # tune_flags bit 16 keeps the ring kernel)
    fused = H % 8 == 0 and W % 32 == 0 and N * (H // 8) * (W // 32) >= 512
    auto_k = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=True).kernel
    assert auto_k == ("winograd2x2_fused_k3x3_c64" if fused else "wconv_64x512_k3x3" if N * (H // 4) * (W // 128) >= 512 else auto_k)
    assert (hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=True, tune_flags=65536).kernel == "wconv_64x512_k3x3") == (N * (H // 4) * (W // 128) >= 512)
    assert p.can_pool and p.can_pool_only and not p.publishes_amax
    p.pack(w); q.pack(w)
    assert torch.equal(p.packed, q.packed)
    pool_p = torch.zeros((N, 64, H // 2, W // 2), device="cuda"); pool_q = torch.zeros_like(pool_p)
    yp = p.forward(x, b, pool_out=pool_p).clone()
    yq = q.forward(x, b, pool_out=pool_q)
    assert torch.equal(yp, yq) and torch.equal(pool_p, pool_q)
    for _ in range(3):
        assert torch.equal(p.forward(x, b), yp)
    pool2 = torch.zeros_like(pool_p)      # the pooled blob alone
    hip._check(hip.lib().mscnn_conv2d_fwd_pool_f32(p._p, hip._dev(x), hip._dev(w), hip._dev(p.packed), hip._dev(b), None, hip._dev(pool2),
                                                   None, 0, hip._stream()))
    assert torch.equal(pool2, pool_p)
    if H * W <= 20000:
        ref = orc.relu(orc.conv2d(x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), (1, 1)))
        close(yp.cpu().numpy(), ref)
    p2 = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=False, algo=hip.ALGO_DIRECT, tune_variant=402)
    q2 = hip.ConvPlan(N, Cin, H, W, 64, 3, 3, (1, 1), relu=False, algo=hip.ALGO_DIRECT, tune_flags=32768, tune_grid=grid_ig)
    p2.pack(w); q2.pack(w)
    assert torch.equal(p2.forward(x), q2.forward(x))
