// Micro-benchmark: the fp32 MFMA issue rate and the shader clock it is sustained at.
// One wave per SIMD (256-thread workgroups, one per CU) issues v_mfma_f32_32x32x2_f32 back to back on 4 accumulators; wave 0 of
// every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop, so the run reports
//   cycles per MFMA (64.0 = the pipe is never idle)  and  the effective clock = d(s_memtime) / d(s_memrealtime)
// for zero operands, random operands, and for 1 / 2 / 3 workgroups per CU.  TFLOP/s = 1024 SIMDs x 4096 FLOP / (cycles x period).
//   hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock && ./mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int iters, const float* src) {
  float a = src[threadIdx.x], b = src[256 + threadIdx.x];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned long long c0 = 0, r0 = 0;
  if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - c0;
    stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float *out, *src; unsigned long long* st;
  hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&src, 512 * 4); hipMalloc(&st, 4096 * 4 * 8);
  std::vector<float> h(512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    for (int i = 0; i < 512; ++i) h[i] = data ? (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f : 0.f;
    hipMemcpy(src, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int wpc = 1; wpc <= 3; ++wpc)
      for (int iters : {2000, 20000}) {
        const int grid = 256 * wpc;
        k<<<grid, 256>>>(out, st, iters / 10, src);
        hipEventRecord(e0);
        k<<<grid, 256>>>(out, st, iters, src);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> s(grid * 4);
        hipMemcpy(s.data(), st, grid * 32, hipMemcpyDeviceToHost);
        std::vector<double> cyc(grid), mhz(grid);
        for (int g = 0; g < grid; ++g) { cyc[g] = (double)s[g * 4] / ((double)iters * 16 * wpc); mhz[g] = (double)s[g * 4] / ((double)s[g * 4 + 1] / 100.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
        const double tf = (double)grid * 4 * iters * 16 * 4096.0 / (ms * 1e-3) / 1e12;
        printf("%-6s wg/CU=%d iters=%5d  %8.3f ms  %6.1f TFLOP/s  cycles/MFMA/SIMD median %.2f (max %.2f)  clock median %.0f MHz (min %.0f)\n",
               data ? "random" : "zero", wpc, iters, ms, tf, cyc[grid / 2], cyc[grid - 1], mhz[grid / 2], mhz[0]);
      }
  }
  return 0;
}
