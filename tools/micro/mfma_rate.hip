// Micro-benchmark: issue rate of the f32 MFMA forms on gfx950 as a function of waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  if constexpr (MODE == 3) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
          if constexpr (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 5, 0);
          if constexpr (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int MODE>
void run(const char* name, int per_iter, double flop_per) {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
    const int grid = 256 * wgs_per_cu;          // 256 CUs, 4 waves per workgroup = 1 wave per SIMD per workgroup
    k<MODE><<<grid, 256>>>(out, 100, 1.f, 2.f);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_per_simd = (double)iters * per_iter * wgs_per_cu;
    printf("%-22s waves/SIMD=%d  %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz)  %.1f TFLOP/s\n", name, wgs_per_cu,
           ms * 1e6 / n_per_simd, ms * 1e6 / n_per_simd * 2.4, flop_per * n_per_simd * 1024 / (ms * 1e-3) / 1e12);
  }
  hipFree(out);
}

int main() {
  run<0>("4x4x1_16b", 16, 512);
  run<1>("4x4x1_16b cbsz4", 16, 512);
  run<2>("16x16x4", 16, 2048);
  run<3>("32x32x2", 16, 4096);
  return 0;
}
