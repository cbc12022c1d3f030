// Stand-alone check + timing of the fused F(4x4,3x3) output -> input transform (winograd.hip: wino44_output_into_input) against the
// unfused pair (wino_output_transform then wino_input_transform): V and y bit for bit, then the time of both routes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../mscnn_amd/csrc wino_outin_check.hip ../../mscnn_amd/csrc/winograd.hip ../../mscnn_amd/csrc/common.cpp -o wino_outin_check
//   ./wino_outin_check C H W [iters] [strip_w] [chunk_rows] [T_pad_m] [T_pad_v]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "winograd.h"
extern "C" const char* mscnn_last_error(void);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RC(x) do { int r_ = (x); if (r_ != 0) { printf("error %d at %d: %s\n", r_, __LINE__, mscnn_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 512, H = argc > 2 ? atoi(argv[2]) : 72, W = argc > 3 ? atoi(argv[3]) : 240;
  const int iters = argc > 4 ? atoi(argv[4]) : 50, strip_w = argc > 5 ? atoi(argv[5]) : 0, chunk_rows = argc > 6 ? atoi(argv[6]) : 0;
  const int N = 1, th = H / 4, tw = W / 4, T = N * th * tw;
  const int Tm = argc > 7 ? atoi(argv[7]) : (T + 159) / 160 * 160, Tv = argc > 8 ? atoi(argv[8]) : (T + 127) / 128 * 128;
  const size_t nm = (size_t)36 * C * Tm, nv = (size_t)36 * C * Tv, ny = (size_t)N * C * H * W;
  std::vector<float> hm(nm), hb(C);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hm) v = rnd() * 4.f;
  for (auto& v : hb) v = rnd();
  float *dM, *dB, *dY, *dY2, *dV, *dV2;
  CK(hipMalloc(&dM, nm * 4)); CK(hipMalloc(&dB, C * 4)); CK(hipMalloc(&dY, ny * 4)); CK(hipMalloc(&dY2, ny * 4));
  CK(hipMalloc(&dV, nv * 4)); CK(hipMalloc(&dV2, nv * 4));
  CK(hipMemcpy(dM, hm.data(), nm * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dV, 0, nv * 4)); CK(hipMemset(dV2, 0, nv * 4)); CK(hipMemset(dY2, 0xff, ny * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  RC(mscnn::wino_output_transform(4, dM, dB, dY, nullptr, N, C, H, W, th, tw, Tm, 1, st));
  RC(mscnn::wino_input_transform(4, dY, dV, N, C, H, W, 1, 1, th, tw, Tv, st));
  RC(mscnn::wino44_output_into_input(dM, dB, dY2, dV2, N, C, H, W, th, tw, Tm, Tv, 1, st, nullptr, strip_w, chunk_rows));
  CK(hipStreamSynchronize(st));
  std::vector<float> v1(nv), v2(nv), y1(ny), y2(ny);
  CK(hipMemcpy(v1.data(), dV, nv * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(v2.data(), dV2, nv * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y1.data(), dY, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y2.data(), dY2, ny * 4, hipMemcpyDeviceToHost));
  size_t badv = 0, bady = 0, firstv = 0;
  for (size_t p = 0; p < 36; ++p)
    for (size_t c = 0; c < (size_t)C; ++c)
      for (size_t t = 0; t < (size_t)T; ++t) {
        const size_t i = (p * C + c) * Tv + t;
        if (memcmp(&v1[i], &v2[i], 4)) { if (!badv) firstv = i; ++badv; }
      }
  for (size_t i = 0; i < ny; ++i) bady += memcmp(&y1[i], &y2[i], 4) != 0;
  printf("C=%d H=%d W=%d tiles %dx%d T_pad_m=%d T_pad_v=%d: V mismatches %zu (first at %zu: %g vs %g), y mismatches %zu\n", C, H, W, th, tw, Tm, Tv,
         badv, firstv, badv ? v1[firstv] : 0.f, badv ? v2[firstv] : 0.f, bady);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) {
      RC(mscnn::wino_output_transform(4, dM, dB, dY, nullptr, N, C, H, W, th, tw, Tm, 1, st));
      RC(mscnn::wino_input_transform(4, dY, dV, N, C, H, W, 1, 1, th, tw, Tv, st));
    }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_pair = ms * 1e3 / iters, by_pair = 4.0 * 36 * C * T * 2 + 8.0 * ny;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) RC(mscnn::wino44_output_into_input(dM, dB, nullptr, dV2, N, C, H, W, th, tw, Tm, Tv, 1, st, nullptr, strip_w, chunk_rows));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_f = ms * 1e3 / iters, by_f = 4.0 * 36 * C * T * 2;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) RC(mscnn::wino44_output_into_input(dM, dB, dY2, dV2, N, C, H, W, th, tw, Tm, Tv, 1, st, nullptr, strip_w, chunk_rows));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_fy = ms * 1e3 / iters;
    if (rep) printf("   unfused pair %.1f us (%.2f TB/s of %.0f MB)   fused %.1f us (%.2f TB/s of %.0f MB)   fused + y %.1f us\n", us_pair, by_pair / us_pair * 1e-6,
                    by_pair * 1e-6, us_f, by_f / us_f * 1e-6, by_f * 1e-6, us_fy);
  }
  return badv || bady ? 2 : 0;
}
