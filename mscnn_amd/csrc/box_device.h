// Device helpers shared by boxoutput.hip and detections.hip (gfx950).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#ifdef MSCNN_BO_TRACE
extern __device__ unsigned long long* g_bo_trace;      // boxoutput.hip, trace build only
#endif

namespace mscnn_dev {

typedef unsigned long long u64;
constexpr int kMaxK = 4032;               // 63 bitmap words: LDS bitonic (32 KB) + scan double buffer (63 KB) fit
constexpr int kSortThreads = 1024;
constexpr int kSortCap = 4096;            // bitonic network size (power of two >= kMaxK)

// ---- expf with glibc's algorithm (sysdeps/ieee754/flt-32/e_expf.c, Szabolcs Nagy's exp2f table
// method: N = 32 table, degree-3 polynomial in double).  The reference's `exp(Dtype)` resolves to
// libm expf on the host; device ocml expf is not bit-identical to it, this restatement is (checked
// exhaustively on the host against libm for |x| <= 80: 2 differing inputs out of 2.2e9, none in the
// clamp range [-ln 2, ln 2] used here).  Valid for |x| < 88 (the callers clamp).
__device__ __constant__ u64 kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

__device__ __forceinline__ float expf_libm(float x) {
  if (!(fabsf(x) < 87.f)) return expf(x);   // outside the table method's plain range: ocml (overflow/underflow/NaN)
  const double N = 32.0;
  const double InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  const double xd = (double)x;
  double z = InvLn2N * xd;
  double kd = z + SHIFT;
  const u64 ki = (u64)__double_as_longlong(kd);
  kd -= SHIFT;
  const double r = z - kd;
  u64 t = kExp2fTab[ki % 32];
  t += ki << 47;
  const double s = __longlong_as_double((long long)t);
  z = C0 * r + C1;
  const double r2 = r * r;
  double y = C2 * r + 1.0;
  y = z * r2 + y;
  y = y * s;
  return (float)y;
}

__device__ __forceinline__ unsigned orderable(float s) {
  if (s == 0.f) s = 0.f;                                // -0 and +0 compare equal in the reference's sort
  const unsigned b = __float_as_uint(s);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}

// OR of a 64-bit value over the 64 lanes (every lane gets the result).
__device__ __forceinline__ u64 wave_or64(u64 v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    lo |= (unsigned)__shfl_xor((int)lo, d, 64);
    hi |= (unsigned)__shfl_xor((int)hi, d, 64);
  }
  return ((u64)hi << 32) | lo;
}

// Greedy scan over the upper-triangular bit matrix (nmsMax, box_output_layer.cpp:38-63): boxes in chunks of 64, one wavefront,
// no LDS and no barrier inside the scan (round 3; rounds 1-2 staged every chunk's 64 mask rows through LDS with three producer
// waves and OR-ed the kept rows' words into a removed-bitmap: ~1.8 us per chunk, 57 us for 2000 boxes, paced by that OR loop and
// the chunk barrier -- tools/bo_trace.py).  What chunk c needs is ONE word: which of its 64 boxes are already suppressed,
//     removed(c) = OR over the kept rows i < 64 c of mask[i][c]      (column c of the bit matrix).
//   * diagonal pass, all scalar: the next surviving box is s_ff1 of the live word; its diagonal mask word comes from a
//     v_readlane with a scalar lane index -- one iteration per KEPT box, not per box;
//   * column c + 1 is gathered straight from L2 while that pass runs: lane l asks for mask[64 cc + l][c + 1] of every earlier
//     chunk cc whose box l was kept (a buffer load per earlier chunk, all in flight together; rows not kept are an out-of-range
//     offset and read as 0); chunk c's own rows come from a word prefetched one chunk ahead, as does its diagonal word;
//   * one 64-lane OR (12 lane exchanges) turns the gathered words into removed(c + 1).
// W, `buf`: unused since round 3 (kept for the callers' launch geometry: 256 threads, waves 1-3 idle).  Result: lane c of wave 0
// returns the keep-word of chunk c.  `removed_init` (optional, one word per chunk): boxes already suppressed from outside -- by the
// kept boxes of earlier tiles in the tiled path of nms_large.h; they are neither kept nor do they suppress anything.
// NB = static bound on the earlier chunks gathered per step (>= nchunks - 1): the loads of a step are a fixed number of
// instructions -- a chunk that does not exist yet, like a row that was not kept, is an out-of-range offset -- so the compiler can
// count them (s_waitcnt vmcnt(k)) and lets the diagonal pass run under them; with a run-time number it waited for all of them first.
template <int NB>
__device__ __forceinline__ u64 greedy_scan_wave(const u64* __restrict__ mask, int n, int wpr, const u64* __restrict__ removed_init) {
  const int lane = threadIdx.x & 63;
  const int nchunks = (n + 63) >> 6;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(mask), 0, n * wpr * 8, 0x00020000);
  const unsigned lane_row = (unsigned)(lane * wpr * 8);          // byte offset of row `lane` of chunk 0
  const unsigned chunk_step = (unsigned)(64 * wpr * 8);          // ... from one chunk's row to the next chunk's
  constexpr unsigned kOobOff = 0x80000000u;
  auto ld = [&](unsigned voff) -> u64 {
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(msrc, voff, 0, 0);
    return ((u64)w[1] << 32) | w[0];
  };
  u64 mykeep = 0;
  u64 dg = ld(lane_row);                                          // mask[lane][0]
  u64 n1 = ld(nchunks > 1 ? lane_row + 8u : kOobOff);             // mask[lane][1]
  u64 removed_c = removed_init ? removed_init[0] : 0ull;          // wave-uniform
  for (int c = 0; c < nchunks; ++c) {
#ifdef MSCNN_BO_TRACE
    if (threadIdx.x == 0 && g_bo_trace && c < 64) g_bo_trace[64 + c] = __builtin_amdgcn_s_memrealtime();
#endif
    const bool more = c + 1 < nchunks;
    const unsigned col = (unsigned)(c + 1) * 8u;
    u64 v[NB];
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) {
      const u64 kw = readlane64(mykeep, cc);                        // (0 for the chunks still to come)
      const bool want = more && cc < c && ((kw >> lane) & 1ull);
      v[cc] = ld(want ? lane_row + col + (unsigned)cc * chunk_step : kOobOff);
    }
    const unsigned nrow = lane_row + col + (unsigned)(c + 1) * chunk_step;
    const u64 dg_n = ld(more ? nrow : kOobOff);                    // mask[64 (c+1) + lane][c + 1]
    const u64 n1_n = ld(c + 2 < nchunks ? nrow + 8u : kOobOff);
    const int valid = min(64, n - c * 64);
    u64 live = ~removed_c;
    if (valid < 64) live &= (1ull << valid) - 1ull;
    u64 keep = 0;
    while (live) {                                     // wave-uniform: scalar loop over the KEPT boxes of the chunk
      const int i = __ffsll((long long)live) - 1;
      keep |= 1ull << i;
      live &= ~readlane64(dg, i);                      // boxes it suppresses
      live &= ~((2ull << i) - 1ull);                   // boxes up to and including i are decided
    }
    if (lane == c) mykeep = keep;
    u64 acc = ((keep >> lane) & 1ull) ? n1 : 0ull;
#pragma unroll
    for (int cc = 0; cc < NB; ++cc) acc |= v[cc];
    acc = wave_or64(acc);
    removed_c = ((u64)__builtin_amdgcn_readfirstlane((unsigned)(acc >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)acc);
    if (removed_init && more) removed_c |= removed_init[c + 1];
    dg = dg_n; n1 = n1_n;
  }
  return mykeep;
}

__device__ __forceinline__ u64 greedy_scan(const u64* __restrict__ mask, int n, int wpr, int W, u64* buf,
                                           const u64* __restrict__ removed_init = nullptr) {
  (void)W; (void)buf;
  u64 mykeep = 0;
  if (threadIdx.x < 64) {
    const int nchunks = (n + 63) >> 6;
    if (nchunks <= 9) mykeep = greedy_scan_wave<8>(mask, n, wpr, removed_init);
    else if (nchunks <= 17) mykeep = greedy_scan_wave<16>(mask, n, wpr, removed_init);
    else if (nchunks <= 33) mykeep = greedy_scan_wave<32>(mask, n, wpr, removed_init);
    else mykeep = greedy_scan_wave<63>(mask, n, wpr, removed_init);
  }
  __syncthreads();
  return mykeep;
}

// Bitonic sort (descending) of 1024 * EPT keys held in registers by a 1024-thread workgroup: element i lives in thread i % 1024,
// register i / 1024.  A compare-exchange at stride s pairs i with i ^ s: s < 64 is a lane exchange inside the wavefront
// (two 32-bit shuffles, no LDS, no barrier), s >= 1024 another register of the same thread, and only 64 <= s <= 512 goes through
// LDS with a barrier -- 10 (EPT 1) / 14 (EPT 2) / 18 (EPT 4) such steps instead of the 55 / 66 / 78 of the all-LDS network below,
// which paced select_sort_kernel (round 3).  Same network, same comparisons: the same order.  `sk` needs 1024 * EPT words.
template <int EPT>
__device__ __forceinline__ void bitonic_desc_regs(u64 (&r)[EPT], u64* sk, int tid) {
  constexpr int P = 1024 * EPT;
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 1024) {
        // (register pairs named by constants: a run-time register index would send r[] to scratch)
        auto cx = [&](auto jl_c, auto jh_c) {
          constexpr int jl = decltype(jl_c)::value, jh = decltype(jh_c)::value;
          const bool desc = ((tid + (jl << 10)) & size) == 0;
          const u64 a = r[jl], b = r[jh];                       // a = the lower index of the pair
          const bool swap = desc ? (a < b) : (a > b);
          if (swap) { r[jl] = b; r[jh] = a; }
        };
        typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
        typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3;
        if constexpr (EPT >= 2) {
          if (stride == 1024) { cx(I0{}, I1{}); if constexpr (EPT == 4) cx(I2{}, I3{}); }
        }
        if constexpr (EPT == 4) {
          if (stride == 2048) { cx(I0{}, I2{}); cx(I1{}, I3{}); }
        }
      } else if (stride < 64) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
          const int i = tid + (j << 10);
          const u64 a = r[j];
          const unsigned blo = (unsigned)__shfl_xor((int)(unsigned)a, stride, 64), bhi = (unsigned)__shfl_xor((int)(unsigned)(a >> 32), stride, 64);
          const u64 b = ((u64)bhi << 32) | blo;
          const bool want_max = (((i & stride) == 0) == ((i & size) == 0));
          r[j] = want_max ? (a > b ? a : b) : (a < b ? a : b);
        }
      } else {
        __syncthreads();                                        // (the previous LDS round has been read)
#pragma unroll
        for (int j = 0; j < EPT; ++j) sk[tid + (j << 10)] = r[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
          const int i = tid + (j << 10);
          const u64 a = r[j], b = sk[i ^ stride];
          const bool want_max = (((i & stride) == 0) == ((i & size) == 0));
          r[j] = want_max ? (a > b ? a : b) : (a < b ? a : b);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < EPT; ++j) sk[tid + (j << 10)] = r[j];
  __syncthreads();
}

// Bitonic sort (descending) of P (power of two, <= kMaxK) 64-bit keys in LDS by one workgroup.
__device__ __forceinline__ void bitonic_desc(u64* sk, int P, int tid, int nthreads) {
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += nthreads) {
        const int lo = (t / stride) * (stride << 1) + (t % stride);
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const u64 a = sk[lo], b = sk[hi];
        const bool swap = desc ? (a < b) : (a > b);
        if (swap) { sk[lo] = b; sk[hi] = a; }
      }
      __syncthreads();
    }
  }
}

}  // namespace mscnn_dev
