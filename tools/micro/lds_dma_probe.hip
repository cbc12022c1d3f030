// Probe: where does `buffer_load_dwordx4 ... offen lds` put each lane's 16 bytes, and how far does M0 reach (160 KB of LDS)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}
__global__ __launch_bounds__(512, 2) void k(const float* g, float* out, unsigned off0, unsigned soff) {
  __shared__ __attribute__((aligned(1024))) float lds[36864];    // 144 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 36864; i += 512) lds[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 1 << 20, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)lds;
  dma16(r, lane * 16u, soff + wave * 1024u, lds0 + off0 + wave * 1024u);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = tid; i < 36864; i += 512) out[i] = lds[i];
  if (tid == 0) out[36864] = __builtin_bit_cast(float, lds0);
}
int main() {
  float *g, *o; hipMalloc(&g, 1 << 20); hipMalloc(&o, 36865 * 4);
  std::vector<float> h(1 << 18); for (int i = 0; i < (1 << 18); ++i) h[i] = (float)i;
  hipMemcpy(g, h.data(), 1 << 20, hipMemcpyHostToDevice);
  for (unsigned off0 : {0u, 49152u, 65536u, 98304u, 139264u}) {
    k<<<1, 512>>>(g, o, off0, 4096u);
    std::vector<float> r(36865); hipMemcpy(r.data(), o, 36865 * 4, hipMemcpyDeviceToHost);
    int first = -1, cnt = 0, ok = 0;
    for (int i = 0; i < 36864; ++i) if (r[i] != -1.f) { if (first < 0) first = i; ++cnt; if (r[i] == (float)(1024 + (i - (int)off0 / 4))) ++ok; }
    unsigned lds0; memcpy(&lds0, &r[36864], 4);
    printf("off0=%6u (lds base %u): %d floats written, first at float %d (expected %u), %d match the linear image; r[first..+4] = %g %g %g %g %g\n", off0, lds0, cnt, first, off0 / 4,
           ok, first >= 0 ? r[first] : 0, first >= 0 ? r[first + 1] : 0, first >= 0 ? r[first + 2] : 0, first >= 0 ? r[first + 3] : 0, first >= 0 ? r[first + 4] : 0);
  }
  return 0;
}
