// F(4x4, 3x3) Winograd transforms for the interpolation points {0, 1, -1, 2, -1/2, inf} -- the 1-D building blocks, usable from
// host and device code (tests/test_wino_f4_model.py compiles this header with g++ and checks every function against the exact
// Cook-Toom matrices of tools/wino_matrices.py).
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

}  // namespace wino_f4
