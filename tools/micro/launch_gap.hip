// What does one dependent launch of a wgemm-shaped grid cost with nothing to do?  256 workgroups x 512 threads, 144 KB of LDS each
// (one per CU), back to back on one stream: per-launch time by HIP events, and the spread of the workgroups' start times.
//   hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap && ./launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LDS_FLOATS, int THREADS>
__global__ __launch_bounds__(THREADS) void empty_kernel(unsigned long long* t, int touch) {
  __shared__ float lds[LDS_FLOATS];
  if (touch) lds[threadIdx.x] = 1.f;
  if (threadIdx.x == 0 && t) t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  if (touch > 1) t[0] = (unsigned long long)lds[0];
}

template <int LDS_FLOATS, int THREADS>
int run(const char* name, int grid) {
  unsigned long long* t; CK(hipMalloc(&t, grid * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) empty_kernel<LDS_FLOATS, THREADS><<<grid, THREADS>>>(t, 1);
  CK(hipEventRecord(e0));
  const int N = 500;
  for (int i = 0; i < N; ++i) empty_kernel<LDS_FLOATS, THREADS><<<grid, THREADS>>>(t, 1);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid); CK(hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost));
  printf("%-44s grid %4d: %.2f us per dependent launch; workgroup starts spread over %.2f us\n", name, grid, ms * 1e3 / N,
         (*std::max_element(h.begin(), h.end()) - *std::min_element(h.begin(), h.end())) / 100.0);
  CK(hipFree(t));
  return 0;
}

int main() {
  if (run<36864, 512>("512 threads, 144 KB LDS (the wgemm grid)", 256)) return 1;
  if (run<36864, 512>("512 threads, 144 KB LDS, 2 rounds", 512)) return 1;
  if (run<8192, 512>("512 threads, 32 KB LDS", 256)) return 1;
  if (run<1024, 256>("256 threads, 4 KB LDS", 256)) return 1;
  if (run<1024, 256>("256 threads, 4 KB LDS", 4096)) return 1;
  if (run<1024, 256>("256 threads, 4 KB LDS", 65536)) return 1;
  return 0;
}
