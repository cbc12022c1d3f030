// Stand-alone check + timing of roipool_wino.hip (the fused ROIPooling x 2 -> F(3x3,3x3) input stage): V against a plain C++ restatement
// (reference pooling loop, roi_pooling_layer.cpp:48-139, then B^T d B in the operation order of wino33_device.h), bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mscnn_amd/csrc roipool_wino_check.hip ../../mscnn_amd/csrc/roipool_wino.hip ../../mscnn_amd/csrc/common.cpp -o roipool_wino_check
//   ./roipool_wino_check [R] [C] [H] [W] [iters]
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "roipool_wino.h"
extern "C" const char* mscnn_last_error(void);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static void bt5h(const float d[5], float r[5]) {
  r[0] = 2.f * d[0] - d[1] - 2.f * d[2] + d[3];
  r[1] = -2.f * d[1] - d[2] + d[3];
  r[2] = 2.f * d[1] - 3.f * d[2] + d[3];
  r[3] = d[3] - d[1];
  r[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
}

static void pool_ref(const float* feat, int C, int H, int W, const float* roi, float scale, float pad_ratio, int c, float out[49]) {
  const int b = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const float pad_w = (x2 - x1 + 1) * pad_ratio, pad_h = (y2 - y1 + 1) * pad_ratio;
  const int sw = (int)roundf((x1 - pad_w) * scale), sh = (int)roundf((y1 - pad_h) * scale);
  const int ew = (int)roundf((x2 + pad_w) * scale), eh = (int)roundf((y2 + pad_h) * scale);
  const int rw = std::max(ew - sw + 1, 1), rh = std::max(eh - sh + 1, 1);
  const float bh = (float)rh / 7.f, bw = (float)rw / 7.f;
  const float* plane = feat + ((size_t)b * C + c) * H * W;
  for (int ph = 0; ph < 7; ++ph)
    for (int pw = 0; pw < 7; ++pw) {
      int hs = (int)floorf((float)ph * bh), ws = (int)floorf((float)pw * bw), he = (int)ceilf((float)(ph + 1) * bh), we = (int)ceilf((float)(pw + 1) * bw);
      hs = std::min(std::max(hs + sh, 0), H); he = std::min(std::max(he + sh, 0), H);
      ws = std::min(std::max(ws + sw, 0), W); we = std::min(std::max(we + sw, 0), W);
      float m = (he <= hs || we <= ws) ? 0.f : -FLT_MAX;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) if (plane[h * W + w] > m) m = plane[h * W + w];
      out[ph * 7 + pw] = m;
    }
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 64, C = argc > 2 ? atoi(argv[2]) : 64, H = argc > 3 ? atoi(argv[3]) : 24, W = argc > 4 ? atoi(argv[4]) : 40;
  const int iters = argc > 5 ? atoi(argv[5]) : 20, N = 1;
  const int kitti = argc > 6 ? atoi(argv[6]) : 0;      // 1: proposals like the benchmark frame's (width 20 .. 400 px log-uniform, h = 0.5 .. 1.6 w, inside the image)
  const int T_pad = (4 * R + 127) / 128 * 128;
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f; };
  std::vector<float> feat((size_t)N * C * H * W), rois((size_t)R * 5);
  for (auto& v : feat) { v = rnd() * 6.f - 3.f; v = v > 0 ? v : 0.f; }
  for (int r = 0; r < R; ++r) {
    float w = expf(logf(6.f) + rnd() * (logf(8.f * W * 0.9f) - logf(6.f))), h = w * (0.3f + 1.3f * rnd());
    float x1 = -40.f + rnd() * (8.f * W + 30.f), y1 = -30.f + rnd() * (8.f * H + 20.f);
    if (kitti) { w = expf(logf(20.f) + rnd() * (logf(400.f) - logf(20.f))); h = w * (0.5f + 1.1f * rnd()); x1 = rnd() * (8.f * W - w); y1 = rnd() * fmaxf(8.f * H - h, 1.f); }
    rois[5 * r] = 0; rois[5 * r + 1] = x1; rois[5 * r + 2] = y1; rois[5 * r + 3] = x1 + w; rois[5 * r + 4] = y1 + h;
  }
  if (R > 8 && !kitti) { rois[5 * 3 + 3] = rois[5 * 3 + 1] - 5.f; rois[5 * 5 + 1] = 8.f * W + 50; rois[5 * 5 + 3] = 8.f * W + 90; rois[5 * 7 + 1] = -300; rois[5 * 7 + 2] = -200; rois[5 * 7 + 3] = 8.f * W + 300; rois[5 * 7 + 4] = 8.f * H + 200; }
  float *dF, *dT, *dR, *dV;
  const size_t vN = (size_t)25 * 2 * C * T_pad;
  CK(hipMalloc(&dF, feat.size() * 4)); CK(hipMalloc(&dT, mscnn::roipool_wino33_scratch_bytes(N, C, H, W))); CK(hipMalloc(&dR, rois.size() * 4)); CK(hipMalloc(&dV, vN * 4));
  CK(hipMemcpy(dF, feat.data(), feat.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, rois.data(), rois.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dV, 0xff, vN * 4));
  if (mscnn::roipool_wino33_build_maps(dF, dT, N, C, H, W, nullptr) || mscnn::roipool_wino33_forward(dT, dR, dV, R, N, C, H, W, T_pad, 0.125f, 0.f, 0.25f, nullptr)) { printf("launch failed: %s\n", mscnn_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> V(vN);
  CK(hipMemcpy(V.data(), dV, vN * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, checked = 0;
  const int rstep = R > 256 ? R / 64 : 1, cstep = C > 64 ? 7 : 1;
  for (int r = 0; r < R; r += rstep)
    for (int q = 0; q < 2; ++q)
      for (int c = 0; c < C; c += cstep) {
        float p[49];
        pool_ref(feat.data(), C, H, W, &rois[5 * r], 0.125f, q ? 0.25f : 0.f, c, p);
#ifdef RPW_DBG_POOLED
        for (int k = 0; k < 49; ++k) {
          const float got = V[((size_t)(k % 25) * 2 * C + q * C + c) * T_pad + 4 * r + k / 25];
          ++checked;
          if (memcmp(&got, &p[k], 4) != 0 && !(got == 0.f && p[k] == 0.f)) {
            if (bad < 24) printf("  pooled mismatch roi %d window %d channel %d bin (%d,%d): got %g want %g\n", r, q, c, k / 7, k % 7, got, p[k]);
            ++bad;
          }
        }
        continue;
#endif
        for (int t = 0; t < 4; ++t) {
          const int ty = t >> 1, tx = t & 1;
          float d[5][5], rr[5][5];
          for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) { const int h = 3 * ty + i, w = 3 * tx + j; d[i][j] = (h < 7 && w < 7) ? p[h * 7 + w] : 0.f; }
          for (int j = 0; j < 5; ++j) { const float col[5] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]}; float o[5]; bt5h(col, o); for (int i = 0; i < 5; ++i) rr[i][j] = o[i]; }
          for (int i = 0; i < 5; ++i) {
            float o[5]; bt5h(rr[i], o);
            for (int j = 0; j < 5; ++j) {
              const float got = V[((size_t)(i * 5 + j) * 2 * C + q * C + c) * T_pad + 4 * r + t];
              ++checked;
              if (memcmp(&got, &o[j], 4) != 0 && !(got == 0.f && o[j] == 0.f)) {
                if (bad < 12) printf("  mismatch roi %d window %d channel %d tile %d plane (%d,%d): got %g want %g\n", r, q, c, t, i, j, got, o[j]);
                ++bad;
              }
            }
          }
        }
      }
  printf("check R=%d C=%d map %dx%d: %zu of %zu values differ\n", R, C, H, W, bad, checked);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) { mscnn::roipool_wino33_build_maps(dF, dT, N, C, H, W, nullptr); mscnn::roipool_wino33_forward(dT, dR, dV, R, N, C, H, W, T_pad, 0.125f, 0.f, 0.25f, nullptr); }
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) { mscnn::roipool_wino33_build_maps(dF, dT, N, C, H, W, nullptr); mscnn::roipool_wino33_forward(dT, dR, dV, R, N, C, H, W, T_pad, 0.125f, 0.f, 0.25f, nullptr); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("transpose + pooling + transform: %.1f us per call\n", ms * 1e3 / iters);
  return bad ? 2 : 0;
}
