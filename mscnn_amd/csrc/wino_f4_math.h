// F(4x4, 3x3) Winograd transforms for the interpolation points {0, 1, -1, 2, -1/2, inf}: the 1-D building blocks and the per-thread
// bodies of the three kernels, usable from host and device code (tests/test_wino_f4_model.py compiles this header with g++, checks the
// 1-D functions against the exact Cook-Toom matrices of tools/wino_matrices.py and runs whole layers through the kernel bodies).
//
// Why these points: 36 multiplies per 16 outputs instead of the 25 per 9 of F(3x3,3x3); the textbook set {0, +-1, +-2} trebles the
// fp32 rounding error, this asymmetric one keeps the error of the F(3x3,3x3) form in use (profiles/r02_study_winograd_f4_numerics.txt).
//
// 2 B^T = | 2  3 -4 -3  2  0 |    G = |   1      0      0   |    A^T = | 1  1  1  1    1   0 |
//         | 0 -2 -5 -1  2  0 |        | -1/3   -1/3   -1/3  |          | 0  1 -1  2  -1/2  0 |
//         | 0  2  1 -5  2  0 |        |  1/3   -1/3    1/3  |          | 0  1  1  4   1/4  0 |
//         | 0 -1 -2  1  2  0 |        |  1/15   2/15   4/15 |          | 0  1 -1  8  -1/8  1 |
//         | 0  4 -2 -4  2  0 |        | -16/15  8/15  -4/15 |
//         | 0  2  3 -4 -3  2 |        |   0      0      1   |
//
// The functions below apply 2 B^T (integers) and G / 2: the factors of two cancel in the element-wise product (exactly: powers of
// two), so y = A^T [ (G g G^T) . (B^T d B) ] A holds with bt6 / g6 / at6 as they are.
#pragma once

#if defined(__HIPCC__)
#define WINO_F4_FN __host__ __device__ __forceinline__
#else
#define WINO_F4_FN inline
#endif

namespace wino_f4 {

// r = (2 B^T) d : six input samples -> six transform-domain values
WINO_F4_FN void bt6(const float d[6], float r[6]) {
  r[0] = 2.f * d[0] + 3.f * d[1] - 4.f * d[2] - 3.f * d[3] + 2.f * d[4];
  r[1] = -2.f * d[1] - 5.f * d[2] - d[3] + 2.f * d[4];
  r[2] = 2.f * d[1] + d[2] - 5.f * d[3] + 2.f * d[4];
  r[3] = -d[1] - 2.f * d[2] + d[3] + 2.f * d[4];
  r[4] = 4.f * d[1] - 2.f * d[2] - 4.f * d[3] + 2.f * d[4];
  r[5] = 2.f * d[1] + 3.f * d[2] - 4.f * d[3] - 3.f * d[4] + 2.f * d[5];
}

// u = (G / 2) g : three filter taps -> six transform-domain values
WINO_F4_FN void g6(const float g[3], float u[6]) {
  u[0] = 0.5f * g[0];
  u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
  u[2] = (g[0] - g[1] + g[2]) * (1.f / 6.f);
  u[3] = g[0] * (1.f / 30.f) + g[1] * (1.f / 15.f) + g[2] * (2.f / 15.f);
  u[4] = g[0] * (-8.f / 15.f) + g[1] * (4.f / 15.f) + g[2] * (-2.f / 15.f);
  u[5] = 0.5f * g[2];
}

// y = A^T m : six transform-domain values -> four outputs
WINO_F4_FN void at6(const float m[6], float y[4]) {
  y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  y[1] = m[1] - m[2] + 2.f * m[3] - 0.5f * m[4];
  y[2] = m[1] + m[2] + 4.f * m[3] + 0.25f * m[4];
  y[3] = m[1] - m[2] + 8.f * m[3] - 0.125f * m[4] + m[5];
}

// ---- the per-thread bodies of the three F(4x4,3x3) kernels of winograd.hip, written against explicit indices so that the host
// can run exactly the code a GPU thread runs (tests/test_wino_f4_model.py drives them over whole layers, multiplies the 36 planes
// with numpy in between and compares with the direct convolution) ----------------------------------------------------------

WINO_F4_FN unsigned abs_bits(float v) {
  union { float f; unsigned u; } c;
  c.f = v;
  return c.u & 0x7fffffffu;
}

// weight transform of the padded (co, ci) pair number i into the igemm packed layout of a 1x1 convolution with one weight set
// per plane: wp[xinu][mt][kc][ck][BM], U = (G g G^T) / 4
WINO_F4_FN void weight_pair(const float* w, float* wp, long i, int Cout, int Cin, int BM, int CK, int MT, int KI) {
  const long img_stride = (long)MT * KI * CK * BM;
  const int m = (int)(i % BM);
  long r = i / BM;
  const int ck = (int)(r % CK); r /= CK;
  const int kc = (int)(r % KI);
  const int mt = (int)(r / KI);
  const int co = mt * BM + m, ci = kc * CK + ck;
  const bool live = co < Cout && ci < Cin;
  float t[6][3];   // G g (columns of g)
  for (int b = 0; b < 3; ++b) {
    float col[3], u[6];
    for (int a = 0; a < 3; ++a) col[a] = live ? w[((long)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
    g6(col, u);
    for (int a = 0; a < 6; ++a) t[a][b] = u[a];
  }
  float* dst = wp + (((long)mt * KI + kc) * CK + ck) * BM + m;
  for (int a = 0; a < 6; ++a) {
    float u[6];
    g6(t[a], u);     // (G g) G^T
    for (int b = 0; b < 6; ++b) dst[(a * 6 + b) * img_stride] = u[b];
  }
}

// input transform of tile t (< T_pad; tiles T .. T_pad are GEMM padding and written as zeros) of channel ci:
// V[xinu][ci][t] = 4 (B^T d B)[xi][nu], d = the 6x6 patch of output tile (n, ty, tx), zero outside the image
WINO_F4_FN void input_tile(const float* x, float* V, int t, int ci, int Cin, int H, int W, int pad_h, int pad_w, int tiles_h,
                           int tiles_w, int T, int T_pad) {
  const long plane_stride = (long)Cin * T_pad;
  float* dst = V + (long)ci * T_pad + t;
  float d[6][6];
  if (t < T) {
    const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
    const float* src = x + ((long)n * Cin + ci) * H * W;
    const int h0 = 4 * ty - pad_h, w0 = 4 * tx - pad_w;
    for (int i = 0; i < 6; ++i) {
      const int h = h0 + i;
      const bool hok = h >= 0 && h < H;
      for (int j = 0; j < 6; ++j) {
        const int wv = w0 + j;
        d[i][j] = (hok && wv >= 0 && wv < W) ? src[h * W + wv] : 0.f;
      }
    }
  } else {
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) d[i][j] = 0.f;
  }
  float r[6][6];   // B^T d (columns of d)
  for (int j = 0; j < 6; ++j) {
    const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
    float o[6];
    bt6(col, o);
    for (int i = 0; i < 6; ++i) r[i][j] = o[i];
  }
  for (int i = 0; i < 6; ++i) {
    float o[6];
    bt6(r[i], o);     // (B^T d) B
    for (int j = 0; j < 6; ++j) dst[(i * 6 + j) * plane_stride] = o[j];
  }
}

// output transform of tile t (< T) of channel co: y = A^T m A + bias, optional ReLU; yp != nullptr: the fused MAX 2x2 / stride 2
// pooling -- a 4x4 tile holds exactly 2x2 windows (ceil mode: windows cut by the bottom / right edge take the maximum over what
// is inside, pooling_layer.cpp:87-101).  Returns the bit pattern of max |y| over what it stored.
WINO_F4_FN unsigned output_tile(const float* M, const float* bias, float* y, float* yp, int t, int co, int Cout, int Ho, int Wo,
                                int tiles_h, int tiles_w, int T_pad, int relu) {
  const long plane_stride = (long)Cout * T_pad;
  const float* src = M + (long)co * T_pad + t;
  unsigned am = 0;
  float r[4][6];   // A^T m (columns of m)
  for (int j = 0; j < 6; ++j) {
    float col[6], o[4];
    for (int i = 0; i < 6; ++i) col[i] = src[(i * 6 + j) * plane_stride];
    at6(col, o);
    for (int i = 0; i < 4; ++i) r[i][j] = o[i];
  }
  const float b = bias ? bias[co] : 0.f;
  const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
  float* dst = y + ((long)n * Cout + co) * Ho * Wo;
  float out[4][4];
  for (int i = 0; i < 4; ++i) {
    float o[4];
    at6(r[i], o);     // (A^T m) A
    const int oh = 4 * ty + i;
    for (int j = 0; j < 4; ++j) {
      const int ow = 4 * tx + j;
      float u = o[j] + b;
      if (relu) u = u < 0.f ? 0.f : u;
      const bool in = oh < Ho && ow < Wo;
      if (in) {
        dst[oh * Wo + ow] = u;
        const unsigned a = abs_bits(u);
        am = a > am ? a : am;
      }
      out[i][j] = in ? u : -3.402823466e+38f;
    }
  }
  if (yp) {
    const int Hp = (Ho + 1) / 2, Wp = (Wo + 1) / 2;
    float* pd = yp + ((long)n * Cout + co) * Hp * Wp;
    for (int pi = 0; pi < 2; ++pi)
      for (int pj = 0; pj < 2; ++pj) {
        const int ph = 2 * ty + pi, pw = 2 * tx + pj;
        if (ph >= Hp || pw >= Wp) continue;
        float m = out[2 * pi][2 * pj];
        if (out[2 * pi][2 * pj + 1] > m) m = out[2 * pi][2 * pj + 1];
        if (out[2 * pi + 1][2 * pj] > m) m = out[2 * pi + 1][2 * pj];
        if (out[2 * pi + 1][2 * pj + 1] > m) m = out[2 * pi + 1][2 * pj + 1];
        pd[ph * Wp + pw] = m;
      }
  }
  return am;
}

}  // namespace wino_f4
