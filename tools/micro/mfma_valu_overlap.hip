// Micro-benchmark: can the fp32 MFMA pipe and the fp32 VALU (v_pk_fma_f32) of a gfx950 SIMD run at the same time?
// Both peak at 64 FLOP / clk / SIMD (157.3 TFLOP/s chip-wide).  A workgroup has 8 waves = 2 per SIMD: waves 0-3 issue
// MFMAs, waves 4-7 issue packed FMAs; each role can be switched off (its waves exit at once).
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters, float a0, float b0) {
  const int wave = threadIdx.x >> 6;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  } else {
    f32x2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
    const f32x2 av = {a, a * 0.5f}, bv = {b * 1e-3f, b * 2e-3f};
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(av, bv, acc[i]);      // v_pk_fma_f32
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out; hipMalloc(&out, 512 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int MI = 4000, VI = 4000;                 // per iteration: 16 MFMAs (64 cycles each) resp. 64 pk_fma (4 cycles each)
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = 256 * wgs_per_cu;
    for (int mode = 1; mode <= 3; ++mode) {        // 1: MFMA only, 2: VALU only, 3: both
      // balanced amounts: 16 MFMAs = 1024 cycles of matrix pipe; 64 pk_fma = 256 cycles of VALU -> 4x more VALU iterations
      const int mi = (mode & 1) ? MI : 0, vi = (mode & 2) ? VI * 4 : 0;
      k<<<grid, 512>>>(out, mi / 40, vi / 40, 1.f, 2.f);
      hipEventRecord(e0);
      k<<<grid, 512>>>(out, mi, vi, 1.f, 2.f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double waves = (double)grid * 4;       // waves per role
      const double mf = (double)mi * 16 * 4096 * waves, vf = (double)vi * 64 * 256 * waves;
      printf("wgs/CU=%d %-10s %8.3f ms   MFMA %6.1f TFLOP/s   VALU %6.1f TFLOP/s   sum %6.1f\n", wgs_per_cu,
             mode == 1 ? "mfma" : mode == 2 ? "valu" : "both", ms, mf / (ms * 1e-3) / 1e12, vf / (ms * 1e-3) / 1e12,
             (mf + vf) / (ms * 1e-3) / 1e12);
    }
  }
  hipFree(out);
  return 0;
}
