// InnerProduct forward for gfx950: y[M,N] = x[M,K] * w[N,K]^T + bias (+ReLU).
// Replaces InnerProductLayer<Dtype>::Forward_gpu (src/caffe/layers/inner_product_layer.cu:10-31:
// cublasSgemm / cublasSgemv + cublasSaxpy for the bias).
//
// fc6 (M = #ROIs, K = 12800, N = 4096): fp32 MFMA GEMM.  Both operands are K-contiguous in HBM, so tiles
// are staged [row][k] into LDS with float4 loads/stores (row stride BK+4 floats keeps 16-B alignment and
// limits the ds_read_b32 bank conflict to 4-way, invisible behind the 64-cycle v_mfma_f32_32x32x2_f32).
// The ROI rows are the MFMA "M" side and the output units the "N" side so that each accumulator register
// is a 128-B run of y[m][n..n+31] (coalesced stores).  Work split is stream-K over (tile, k-chunk), as in
// conv.hip: M is data dependent (1..2000 ROIs) so tile counts never match the 256 CUs; partial tiles go
// through fp32 slabs + a fix-up kernel that also applies bias/ReLU.  Tiles are ordered with the ROI-tile
// index fastest so that the workgroups sharing a 128-row slice of W run together and share it in L2
// (W for fc6 is 210 MB: it must stream from HBM once, not once per ROI tile).
// cls_pred / bbox_pred (N = 5 / 20): one workgroup per row, lanes split K, wave reductions.
#include "common.h"
#include "wgemm.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, LDK = BK + 4;     // tile shapes: 128 x 128 (2 x 2 waves) or 64 x 256 (1 x 4 waves); each wave 64 x 64

struct GemmArgs {
  const float* x; const float* w; const float* bias; float* y; float* ws;
  int M, N, K, MT, NT, KI, G, relu;
  long total_iters;
};

__device__ __forceinline__ void wg_range(long total, int G, int g, long& b, long& e) {
  b = total * g / G;
  e = total * (g + 1) / G;
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs a) {      // (min 3 workgroups / CU was tried: 168 VGPRs + 176 B scratch)
  static_assert(BM * BN == 128 * 128 && BM % 64 == 0 && BN % 64 == 0, "4 waves of 64 x 64");
  __shared__ __attribute__((aligned(16))) float ldsX[BM * LDK];
  __shared__ __attribute__((aligned(16))) float ldsW[BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  constexpr int WN = BN / 64;                   // waves along N
  const int wm = wave / WN, wn = wave % WN;     // 64 x 64 per wave
  constexpr int XV = BM * (BK / 4) / 256, WV = BN * (BK / 4) / 256;     // float4 per thread and operand

  long it, it_end;
  wg_range(a.total_iters, a.G, blockIdx.x, it, it_end);

  // staging map: rows x 8 float4 per operand; float4 number v = tid + i * 256 is row v / 8, k offset (v % 8) * 4
  const int s_row0 = tid >> 3, s_k4 = (tid & 7) * 4;      // row of pass i: s_row0 + 32 i

  while (it < it_end) {
    const int t = (int)(it / a.KI);
    const int k0 = (int)(it % a.KI);
    const int k1 = (int)min((long)a.KI, k0 + (it_end - it));
    const int nt = t / a.MT, mt = t % a.MT;      // ROI tile fastest
    const int m0 = mt * BM, n0 = nt * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 rx[XV], rw[WV];
    auto load_chunk = [&](int kc) {
      const int k = kc * BK + s_k4;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int m = m0 + s_row0 + 32 * i;
        rx[i] = (m < a.M && k < a.K) ? *reinterpret_cast<const float4*>(a.x + (long)m * a.K + k) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WV; ++i) {
        const int n = n0 + s_row0 + 32 * i;
        rw[i] = (n < a.N && k < a.K) ? *reinterpret_cast<const float4*>(a.w + (long)n * a.K + k) : make_float4(0, 0, 0, 0);
      }
    };
    auto store_chunk = [&]() {
#pragma unroll
      for (int i = 0; i < XV; ++i) *reinterpret_cast<float4*>(&ldsX[(s_row0 + 32 * i) * LDK + s_k4]) = rx[i];
#pragma unroll
      for (int i = 0; i < WV; ++i) *reinterpret_cast<float4*>(&ldsW[(s_row0 + 32 * i) * LDK + s_k4]) = rw[i];
    };

    load_chunk(k0);
    for (int kc = k0; kc < k1; ++kc) {
      __syncthreads();
      store_chunk();
      __syncthreads();
      if (kc + 1 < k1) load_chunk(kc + 1);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = ldsX[(wm * 64 + i * 32 + l31) * LDK + kk + khalf];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = ldsW[(wn * 64 + j * 32 + l31) * LDK + kk + khalf];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }

    const bool full = (k0 == 0 && k1 == a.KI);
    float* slab = a.ws + ((long)blockIdx.x * 2 + (k0 > 0 ? 0 : 1)) * (BM * BN);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = wn * 64 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (full) {
            const int m = m0 + ml, n = n0 + nl;
            if (m < a.M && n < a.N) {
              float v = acc[i][j][r];
              if (a.bias) v += a.bias[n];
              if (a.relu) v = v < 0.f ? 0.f : v;
              a.y[(long)m * a.N + n] = v;
            }
          } else {
            slab[ml * BN + nl] = acc[i][j][r];
          }
        }
      }
    it += (k1 - k0);
    __syncthreads();
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_fixup_kernel(GemmArgs a) {
  __shared__ const float* s_slab[64];
  __shared__ int s_gf, s_n;
  const int t = blockIdx.x >> 2, part = blockIdx.x & 3;    // 4 workgroups per 128x128 tile
  const long its = (long)t * a.KI, ite = its + a.KI;
  if (threadIdx.x == 0) {
    int gf = (int)(its * a.G / a.total_iters), gl = (int)((ite - 1) * a.G / a.total_iters);
    long b, e;
    wg_range(a.total_iters, a.G, gf, b, e);
    while (e <= its) { ++gf; wg_range(a.total_iters, a.G, gf, b, e); }
    while (b > its) { --gf; wg_range(a.total_iters, a.G, gf, b, e); }
    wg_range(a.total_iters, a.G, gl, b, e);
    while (e <= ite - 1) { ++gl; wg_range(a.total_iters, a.G, gl, b, e); }
    while (b > ite - 1) { --gl; wg_range(a.total_iters, a.G, gl, b, e); }
    s_gf = gf;
    s_n = gf == gl ? 0 : min(gl - gf + 1, 64);
  }
  __syncthreads();
  const int n = s_n;
  if (n == 0) return;
  if ((int)threadIdx.x < n) {      // the list is built by n threads at once: a range is two 64-bit divisions (r5)
    const int g = s_gf + threadIdx.x;
    long b, e;
    wg_range(a.total_iters, a.G, g, b, e);
    s_slab[threadIdx.x] = a.ws + ((long)g * 2 + (b > its ? 0 : 1)) * (BM * BN);
  }
  __syncthreads();
  const int nt = t / a.MT, mt = t % a.MT;
  const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = part * 4096 + (j * 256 + threadIdx.x) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < n; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(s_slab[s] + i);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int m = m0 + i / BN, nn = n0 + i % BN;
    if (m >= a.M) continue;
    const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (nn + q >= a.N) continue;
      float r = vals[q];
      if (a.bias) r += a.bias[nn + q];
      if (a.relu) r = r < 0.f ? 0.f : r;
      a.y[(long)m * a.N + nn + q] = r;
    }
  }
}

// ---- fp16-operand variant (MSCNN_CONV_ALGO_F16's counterpart for InnerProduct; no reference counterpart) --------------------
// y = x[M,K] (fp32, rounded to fp16 while it is staged) * w16[N,K]^T (fp16, converted once by mscnn_inner_product_pack_f16),
// v_mfma_f32_32x32x16_f16 with fp32 accumulators.  At small M (a few hundred ROIs) fc6 is bound by streaming its weights
// once: fp16 weights halve those bytes (caltech fc6: 134 -> 67 MB).  LDS rows are [row][BK + 8] halves: the 80-byte stride
// puts the 16-byte operand reads of 32 consecutive rows on distinct bank groups.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = BK + 8;

__global__ __launch_bounds__(256) void pack_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ w16, long count) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256) w16[i] = (_Float16)w[i];
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm16_tn_kernel(GemmArgs a) {
  static_assert(BM * BN == 128 * 128 && BM % 64 == 0 && BN % 64 == 0, "4 waves of 64 x 64");
  __shared__ __attribute__((aligned(16))) _Float16 ldsX[BM * LDH];
  __shared__ __attribute__((aligned(16))) _Float16 ldsW[BN * LDH];
  const _Float16* w16 = reinterpret_cast<const _Float16*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  constexpr int WN = BN / 64;
  const int wm = wave / WN, wn = wave % WN;
  constexpr int XV = BM * (BK / 4) / 256;       // float4 (4 k) per thread of the x tile
  constexpr int WV = BN * (BK / 8) / 256;       // half8 (8 k) per thread of the w tile

  long it, it_end;
  wg_range(a.total_iters, a.G, blockIdx.x, it, it_end);
  const int xs_row0 = tid >> 3, xs_k = (tid & 7) * 4;       // x: 8 float4 per row
  const int ws_row0 = tid >> 2, ws_k = (tid & 3) * 8;       // w: 4 half8 per row

  while (it < it_end) {
    const int t = (int)(it / a.KI);
    const int k0 = (int)(it % a.KI);
    const int k1 = (int)min((long)a.KI, k0 + (it_end - it));
    const int nt = t / a.MT, mt = t % a.MT;
    const int m0 = mt * BM, n0 = nt * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 rx[XV];
    f16x8 rw[WV];
    auto load_chunk = [&](int kc) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int m = m0 + xs_row0 + 32 * i, k = kc * BK + xs_k;
        rx[i] = (m < a.M && k < a.K) ? *reinterpret_cast<const float4*>(a.x + (long)m * a.K + k) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < WV; ++i) {
        const int n = n0 + ws_row0 + 64 * i, k = kc * BK + ws_k;
        f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        rw[i] = (n < a.N && k < a.K) ? *reinterpret_cast<const f16x8*>(w16 + (long)n * a.K + k) : z;
      }
    };
    auto store_chunk = [&]() {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        f16x4 h = {(_Float16)rx[i].x, (_Float16)rx[i].y, (_Float16)rx[i].z, (_Float16)rx[i].w};
        *reinterpret_cast<f16x4*>(&ldsX[(xs_row0 + 32 * i) * LDH + xs_k]) = h;
      }
#pragma unroll
      for (int i = 0; i < WV; ++i) *reinterpret_cast<f16x8*>(&ldsW[(ws_row0 + 64 * i) * LDH + ws_k]) = rw[i];
    };

    load_chunk(k0);
    for (int kc = k0; kc < k1; ++kc) {
      __syncthreads();
      store_chunk();
      __syncthreads();
      if (kc + 1 < k1) load_chunk(kc + 1);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        f16x8 av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const f16x8*>(&ldsX[(wm * 64 + i * 32 + l31) * LDH + ks * 16 + khalf * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const f16x8*>(&ldsW[(wn * 64 + j * 32 + l31) * LDH + ks * 16 + khalf * 8]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }

    const bool full = (k0 == 0 && k1 == a.KI);
    float* slab = a.ws + ((long)blockIdx.x * 2 + (k0 > 0 ? 0 : 1)) * (BM * BN);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = wn * 64 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (full) {
            const int m = m0 + ml, n = n0 + nl;
            if (m < a.M && n < a.N) {
              float v = acc[i][j][r];
              if (a.bias) v += a.bias[n];
              if (a.relu) v = v < 0.f ? 0.f : v;
              a.y[(long)m * a.N + n] = v;
            }
          } else {
            slab[ml * BN + nl] = acc[i][j][r];
          }
        }
      }
    it += (k1 - k0);
    __syncthreads();
  }
}

// Small-N path (cls_pred N = 5, bbox_pred N = 20, K = 4096): one workgroup per row m.  The row of x is held in registers
// (K / 256 values per thread), every output n is a register dot product + wave reduction, the 4 wave partials of all N
// outputs are combined after ONE barrier.  W (N x K) is re-read by every workgroup out of L2.
constexpr int kRowMaxN = 64, kRowMaxKPerThread = 16;
// NB = outputs handled per pass, a compile-time bound: the NB dot products and their wave reductions are unrolled, so the 6-step
// shuffle chains of different outputs interleave instead of running one after the other (bbox_pred, N = 20: 42 -> see
// profiles/r03_layers_*.txt); the order of the additions inside one output is unchanged (bit-identical results).
// kRowsPerBlock rows share every load of W.  More rows per workgroup = fewer line requests for W, and measured SLOWER (r5,
// profiles/r05_ab_ip_rows.txt: bbox_pred inside the net 19.3 / 24.0 / 35.2 us at 2 / 4 / 8 rows): the kernel is bound by the dependent
// chain of one workgroup, not by W.  mscnn_debug_inner_product_rows picks the instantiation (tests: all of them bit-identical).
template <int NB, int kRowsPerBlock>
__global__ __launch_bounds__(256) void ip_rowwise_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                         int K, int relu) {
  __shared__ float red[kRowsPerBlock][4][kRowMaxN];
  const int m0 = blockIdx.x * kRowsPerBlock;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float xv[kRowsPerBlock][kRowMaxKPerThread];
#pragma unroll
  for (int r = 0; r < kRowsPerBlock; ++r)
#pragma unroll
    for (int i = 0; i < kRowMaxKPerThread; ++i) {
      const int k = tid + i * 256;
      xv[r][i] = (k < K && m0 + r < M) ? x[(long)(m0 + r) * K + k] : 0.f;
    }
  for (int n0 = 0; n0 < N; n0 += NB) {
    float acc[NB][kRowsPerBlock];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = min(n0 + j, N - 1);                 // (surplus outputs of the last pass recompute the last one; not stored)
      const float* wr = w + (long)n * K;
#pragma unroll
      for (int r = 0; r < kRowsPerBlock; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int i = 0; i < kRowMaxKPerThread; ++i) {
        const int k = tid + i * 256;
        const float wv = k < K ? wr[k] : 0.f;
#pragma unroll
        for (int r = 0; r < kRowsPerBlock; ++r) acc[j][r] += xv[r][i] * wv;
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < kRowsPerBlock; ++r) acc[j][r] += __shfl_down(acc[j][r], d, 64);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (n0 + j < N) {
#pragma unroll
          for (int r = 0; r < kRowsPerBlock; ++r) red[r][wave][n0 + j] = acc[j][r];
        }
    }
  }
  __syncthreads();
  for (int i = tid; i < kRowsPerBlock * N; i += 256) {
    const int r = i / N, n = i % N;
    if (m0 + r >= M) continue;
    float v = (red[r][0][n] + red[r][1][n]) + (red[r][2][n] + red[r][3][n]);
    if (bias) v += bias[n];
    if (relu) v = v < 0.f ? 0.f : v;
    y[(long)(m0 + r) * N + n] = v;
  }
}

// Fully general fallback (any N, K): one output per workgroup pass.
__global__ __launch_bounds__(256) void ip_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M, int N,
                                                         int K, int relu) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  const float* xr = x + (long)m * K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = blockIdx.y; n < N; n += gridDim.y) {
    const float* wr = w + (long)n * K;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) acc += xr[k] * wr[k];
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float v = (red[0] + red[1]) + (red[2] + red[3]);
      if (bias) v += bias[n];
      if (relu) v = v < 0.f ? 0.f : v;
      y[(long)m * N + n] = v;
    }
    __syncthreads();
  }
}

}  // namespace

using namespace mscnn;

// Workspace for the stream-K slabs is owned by the library because the Caffe layer interface has no workspace argument for
// InnerProduct: one buffer per (host thread, device), grown on demand.  Per thread because the reference's execution model is one
// thread per GPU with a thread-local Caffe singleton (common.cpp:13-20; host/tools/detect_multi_gpu.cpp runs one replica per
// thread), per device because a thread may switch devices; a buffer shared across threads would have concurrent replicas
// writing their partial sums into each other's slabs.
namespace {
constexpr int kMaxDevices = 64;
struct SlabWs { float* p = nullptr; size_t bytes = 0; };
int reserve_slabs(size_t need, float** out) {
  static thread_local SlabWs ws[kMaxDevices];
  int dev = 0;
  MSCNN_HIP_TRY(hipGetDevice(&dev));
  MSCNN_REQUIRE(dev >= 0 && dev < kMaxDevices, "inner_product: device index %d out of range", dev);
  SlabWs& w = ws[dev];
  if (need > w.bytes) {
    if (w.p) MSCNN_HIP_TRY(hipFree(w.p));      // (synchronises the device: kernels still reading the old buffer have finished)
    w.p = nullptr; w.bytes = 0;
    MSCNN_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.p), need));
    w.bytes = need;
  }
  *out = w.p;
  return MSCNN_OK;
}
}  // namespace

extern "C" int mscnn_inner_product_f16_supported(int N, int K) { return N >= 64 && K % 8 == 0; }

extern "C" int mscnn_inner_product_pack_f16(const float* w, void* w16, int N, int K, void* stream) {
  MSCNN_REQUIRE(w && w16 && N > 0 && K > 0, "inner_product pack: bad argument");
  const long count = (long)N * K;
  long blocks = (count + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  pack_f16_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, static_cast<_Float16*>(w16), count);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

static int inner_product_gemm(const float* x, const void* w, bool w_is_f16, const float* bias, float* y, int M, int N, int K, int relu,
                              hipStream_t st);

extern "C" int mscnn_inner_product_fwd_f16(const float* x, const void* w16, const float* bias, float* y, int M, int N, int K, int relu,
                                           void* stream) {
  MSCNN_REQUIRE(M >= 0 && N > 0 && K > 0, "inner_product: bad shape M=%d N=%d K=%d", M, N, K);
  if (M == 0) return MSCNN_OK;
  MSCNN_REQUIRE(x && w16 && y, "inner_product: null pointer");
  MSCNN_REQUIRE(mscnn_inner_product_f16_supported(N, K) && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w16) % 16 == 0,
                "inner_product f16: needs N >= 64, K %% 8 == 0 and 16-byte aligned operands (N=%d K=%d)", N, K);
  return inner_product_gemm(x, w16, true, bias, y, M, N, K, relu, as_stream(stream));
}

static int g_ip_rows = 0;
// dev / test knob: rows per workgroup of the small-N kernel (2, 4 or 8; 0 = the default: 4 for N <= 5, else 2).  The per-row arithmetic does not depend on it.
extern "C" void mscnn_debug_inner_product_rows(int rows) { g_ip_rows = rows == 2 || rows == 4 || rows == 8 ? rows : 0; }

extern "C" int mscnn_inner_product_fwd_f32(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                                           int relu, void* stream) {
  MSCNN_REQUIRE(M >= 0 && N > 0 && K > 0, "inner_product: bad shape M=%d N=%d K=%d", M, N, K);
  if (M == 0) return MSCNN_OK;
  MSCNN_REQUIRE(x && w && y, "inner_product: null pointer");
  hipStream_t st = as_stream(stream);
  const bool aligned = (K % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(w) % 16 == 0);
  if (N < 64 || !aligned) {
    if (N <= kRowMaxN && K <= 256 * kRowMaxKPerThread) {
      // N <= 5 (cls_pred): 4 rows, one pass of 5 outputs.  N > 5 (bbox_pred): 2 rows and 8 outputs per pass -- the kernel is a
      // latency chain per workgroup (x loads -> per pass: W loads -> FMAs -> 6 shuffle steps), so shorter chains on more workgroups
      // win: bbox_pred inside the net 24.0 (4 rows) -> 19.3 us (2 rows), 35.2 at 8 rows (profiles/r05_ab_ip_rows.txt).
      const int rows = g_ip_rows ? g_ip_rows : (N <= 5 ? 4 : 2);
#define MSCNN_IP_ROWWISE(NB_, R_) ip_rowwise_kernel<NB_, R_><<<cdiv(M, R_), 256, 0, st>>>(x, w, bias, y, M, N, K, relu)
      if (N <= 5) { if (rows == 8) MSCNN_IP_ROWWISE(5, 8); else if (rows == 2) MSCNN_IP_ROWWISE(5, 2); else MSCNN_IP_ROWWISE(5, 4); }
      else { if (rows == 8) MSCNN_IP_ROWWISE(4, 8); else if (rows == 2) MSCNN_IP_ROWWISE(8, 2); else MSCNN_IP_ROWWISE(4, 4); }
#undef MSCNN_IP_ROWWISE
    } else {
      ip_generic_kernel<<<dim3(M, N < 64 ? N : 64), 256, 0, st>>>(x, w, bias, y, M, N, K, relu);
    }
    MSCNN_POST_LAUNCH();
    return MSCNN_OK;
  }
  return inner_product_gemm(x, w, false, bias, y, M, N, K, relu, st);
}

static int inner_product_gemm(const float* x, const void* w, bool w_is_f16, const float* bias, float* y, int M, int N, int K, int relu,
                              hipStream_t st) {
  GemmArgs a;
  a.x = x; a.w = static_cast<const float*>(w); a.bias = bias; a.y = y;
  a.M = M; a.N = N; a.K = K; a.relu = relu;
  // tile shape: 64-row tiles when they waste fewer padded ROI rows (M = 700: 704 instead of 768 rows, -8 % MFMA work)
  const bool m64 = cdiv(M, 64) * 64 < cdiv(M, 128) * 128 && N >= 256;
  const int BM = m64 ? 64 : 128, BN = m64 ? 256 : 128;
  a.MT = cdiv(M, BM); a.NT = cdiv(N, BN); a.KI = cdiv(K, BK);
  a.total_iters = (long)a.MT * a.NT * a.KI;
  long G = 512;
  if (a.total_iters / 8 < G) G = a.total_iters / 8;
  if (G < 1) G = 1;
  a.G = (int)G;
  const size_t need = (size_t)a.G * 2 * BM * BN * sizeof(float);
  {
    const int rc = reserve_slabs(need, &a.ws);
    if (rc != MSCNN_OK) return rc;
  }
  if (w_is_f16) {
    if (m64) gemm16_tn_kernel<64, 256><<<a.G, 256, 0, st>>>(a);
    else gemm16_tn_kernel<128, 128><<<a.G, 256, 0, st>>>(a);
  } else {
    if (m64) gemm_tn_kernel<64, 256><<<a.G, 256, 0, st>>>(a);
    else gemm_tn_kernel<128, 128><<<a.G, 256, 0, st>>>(a);
  }
  MSCNN_POST_LAUNCH();
  if (m64) gemm_fixup_kernel<64, 256><<<a.MT * a.NT * 4, 256, 0, st>>>(a);
  else gemm_fixup_kernel<128, 128><<<a.MT * a.NT * 4, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}


// ---- InnerProduct on the plane-GEMM kernel of wgemm.hip (round 4) --------------------------------------------------------------------
// fc6 is a plain [R x K] x [K x N] GEMM with K = 12800: gemm_tn_kernel above still has the round-1 structure (operands staged through
// registers, two barriers per chunk, 0.74 - 0.77 of the MFMA peak, 604 - 630 us); wgemm_kernel (LDS-DMA ring, one barrier per chunk,
// ride-along stores) runs the same shape at 0.84+.  Mapping: the ROI rows are the kernel's "Cout" side -- x is re-packed per forward
// into its A layout Up[mt][kc][32][256] (one transposing pass over 35 MB) --, the N outputs its tile columns -- the weights are kept
// transposed, Wt[K][N], packed once per weight change --, so M = y[R][N] comes out in the blob's own layout and the kernel's epilogue
// adds bias and ReLU.  96 tiles x 400 chunks are split stream-K over the 256 workgroups (150 chunks each; a tile's partial sums meet
// in the workgroup that holds its first chunks, in k order: deterministic).
namespace {

// Up[mt][kc][ck][256] <- x[r][k] (zero rows past M): 256 rows x 32 k per workgroup through LDS, 128-byte reads, 1 KB writes
__global__ __launch_bounds__(256) void ip_pack_rows_kernel(const float* __restrict__ x, float* __restrict__ up, int M, int K, int KI) {
  __shared__ float t[32][257];
  const int kc = blockIdx.x, mt = blockIdx.y, tid = threadIdx.x;
  const int r0 = mt * 256, k0 = kc * 32;
#pragma unroll
  for (int i = 0; i < 8; ++i) {      // 256 rows x 8 float4
    const int v = tid + i * 256, row = v >> 3, k4 = (v & 7) * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < M) q = *reinterpret_cast<const float4*>(x + (size_t)(r0 + row) * K + k0 + k4);
    t[k4][row] = q.x; t[k4 + 1][row] = q.y; t[k4 + 2][row] = q.z; t[k4 + 3][row] = q.w;
  }
  __syncthreads();
  float* dst = up + ((size_t)mt * KI + kc) * (32 * 256);
#pragma unroll
  for (int i = 0; i < 32; ++i) dst[i * 256 + tid] = t[i][tid];
}

// Wt[k][n] = w[n][k]: 32 x 32 tiles
__global__ __launch_bounds__(256) void ip_transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int K) {
  __shared__ float t[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, k = k0 + tx;
    t[ty + 8 * i][tx] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, n = n0 + tx;
    if (k < K && n < N) wt[(size_t)k * N + n] = t[tx][ty + 8 * i];
  }
}

bool ip_wg_plan(int M, int N, int K, mscnn::WgemmPlan* pl) {
  // rows in 32-row blocks; the tile columns must be the blob's own row stride (N % 128 == 0); below ~200 rows the 256-row tiles are
  // mostly padding and the stream-K kernel above is the better fit
  if (M < 192 || N % 128 != 0 || K % 32 != 0 || N < 256) return false;
  const int Mp = (M + 31) / 32 * 32;
  return mscnn::wgemm_plan(1, Mp, K, N, 1 + 64, pl) && pl->T_pad == N;
}

}  // namespace

extern "C" int mscnn_inner_product_wg_supported(int M, int N, int K) {
  mscnn::WgemmPlan pl;
  return ip_wg_plan(M, N, K, &pl) ? 1 : 0;
}
extern "C" size_t mscnn_inner_product_wg_packed_bytes(int N, int K) { return (size_t)N * K * sizeof(float); }
extern "C" size_t mscnn_inner_product_wg_workspace_bytes(int M, int N, int K) {
  mscnn::WgemmPlan pl;
  if (!ip_wg_plan(M, N, K, &pl)) return 0;
  return (pl.packed_bytes + 255) / 256 * 256 + pl.ws_bytes;
}
extern "C" int mscnn_inner_product_wg_pack(const float* w, float* wt, int N, int K, void* stream) {
  MSCNN_REQUIRE(w && wt && N > 0 && K > 0, "inner_product wg pack: bad argument");
  ip_transpose_w_kernel<<<dim3(cdiv(K, 32), cdiv(N, 32)), 256, 0, as_stream(stream)>>>(w, wt, N, K);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
extern "C" int mscnn_inner_product_wg_fwd(const float* x, const float* wt, const float* bias, float* y, int M, int N, int K, int relu,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(M >= 0 && N > 0 && K > 0, "inner_product: bad shape M=%d N=%d K=%d", M, N, K);
  if (M == 0) return MSCNN_OK;
  mscnn::WgemmPlan pl;
  MSCNN_REQUIRE(ip_wg_plan(M, N, K, &pl), "inner_product wg: shape M=%d N=%d K=%d is not one the plane-GEMM kernel takes", M, N, K);
  MSCNN_REQUIRE(x && wt && y && reinterpret_cast<uintptr_t>(x) % 16 == 0, "inner_product wg: null or unaligned pointer");
  const size_t need = mscnn_inner_product_wg_workspace_bytes(M, N, K);
  if (!workspace || workspace_bytes < need) {
    set_error("inner_product wg: workspace %zu < %zu", workspace_bytes, need);
    return MSCNN_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  float* up = static_cast<float*>(workspace);
  float* slabs = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + (pl.packed_bytes + 255) / 256 * 256);
  ip_pack_rows_kernel<<<dim3(pl.KI, pl.MT), 256, 0, st>>>(x, up, M, K, pl.KI);
  MSCNN_POST_LAUNCH();
  return mscnn::wgemm_launch(pl, up, wt, y, pl.ws_bytes ? slabs : nullptr, st, 0, nullptr, bias, relu, (size_t)M * N * sizeof(float));
}
