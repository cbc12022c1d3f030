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

// (Round 3 also tried a one-wavefront form without LDS or barriers -- chunk c + 1's removed word gathered as a COLUMN of the bit
// matrix from L2, kept rows only, while chunk c's diagonal pass runs: correct, but 2.8 us per chunk against 1.8 here, because the
// gather can only be issued once the previous chunk's kept set is known and the matrix, written by other XCDs, comes from memory.)
// Greedy scan over the upper-triangular bit matrix (nmsMax, box_output_layer.cpp:38-63) by one 256-thread
// workgroup.  Wave 0 owns the removed-bitmap (lane w = boxes [64w, 64w+64)) and walks the boxes in chunks of 64:
//   * diagonal pass, all scalar: the next surviving box is s_ff1 of the live word; its diagonal mask word comes from a
//     v_readlane with a scalar lane index -- one iteration per KEPT box, not per box;
//   * the kept rows' words (w >= c) are OR-ed into the removed-bitmap from LDS, four independent reads at a time.
// (r3: the OR is shared by all four waves, see the function body.)
// Waves 1-3 stream the NEXT chunk's 64 mask rows (words c+1.. only) from L2 into the other half of a double buffer in LDS
// meanwhile: lane = word, one row per load instruction (coalesced), all of a thread's rows in flight at once.
// W = words per row actually used (<= 64).  `buf` = dynamic LDS of 2 * 64 * W u64.  Result: lane c of wave 0 returns the
// keep-word of chunk c.  `removed_init` (optional, W words): boxes already suppressed from outside -- by the kept boxes of
// earlier tiles in the tiled path of nms_large.h; they are neither kept nor do they suppress anything.
__device__ __forceinline__ u64 greedy_scan(const u64* __restrict__ mask, int n, int wpr, int W, u64* buf,
                                           const u64* __restrict__ removed_init = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = (n + 63) >> 6;
  // rows r = r0, r0 + rstep, ... of chunk c; this thread moves word `lane` of each.  The producers are software-pipelined across
  // the chunk barrier: the rows of chunk c + 2 are requested while chunk c is scanned and parked in LDS one iteration later, so
  // the L2 round trip of a chunk's rows overlaps a whole scan step instead of being waited for inside it (round 3: 63 -> see DESIGN.md).
  constexpr int kBatch = 22;                             // ceil(64 / 3): every row of a producer wave in one batch
  // (buffer loads: a row past n or a word this lane does not need is an out-of-range offset and reads as 0 -- no branch and no
  // 64-bit address arithmetic per row; the per-row form with both was 15 instructions x 22 rows per chunk and paced the scan)
  const __amdgpu_buffer_rsrc_t msrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64*>(mask), 0, n * wpr * 8, 0x00020000);
  auto load_rows = [&](int c, int r0, int rstep, u64 (&v)[kBatch]) {
    const bool need = lane < W && lane >= c && lane < wpr;      // words left of the diagonal are never read
    const unsigned off0 = need ? (unsigned)(((c * 64 + r0) * wpr + lane) * 8) : 0x80000000u;
    const unsigned step = (unsigned)(rstep * wpr * 8);
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      // (rows r >= 64 of the batch belong to the next chunk or lie past n: they are loaded into registers nobody stores)
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(msrc, off0 + (unsigned)j * step, 0, 0);
      v[j] = ((u64)w[1] << 32) | w[0];
    }
  };
  auto store_rows = [&](int b, int r0, int rstep, const u64 (&v)[kBatch]) {
    if (lane >= W) return;
    u64* dst = buf + (size_t)b * 64 * W;
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int r = r0 + j * rstep;
      if (r < 64) dst[r * W + lane] = v[j];
    }
  };
  // The removed-bitmap lives in LDS (word w = boxes [64 w, 64 w + 64)).  Per chunk: wave 0 runs the scalar diagonal pass while
  // waves 1-3 move rows; barrier; then ALL FOUR waves OR the chunk's kept rows into the bitmap, 16 candidate rows each (LDS
  // atomic OR, lane = word), barrier.  (Rounds 1-2 and most of round 3: wave 0 did that OR alone, a chain of ~20 kept rows four
  // at a time = 0.8 of the 1.8 us a chunk took -- tools/bo_trace.py.)
  __shared__ u64 s_removed[64];
  __shared__ u64 s_keepw;
  auto uniform64 = [](u64 x) -> u64 {      // (readfirstlane returns int: cast back to unsigned before widening)
    return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32) |
           (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
  };
  if (tid < 64) s_removed[tid] = (removed_init != nullptr && tid < W) ? removed_init[tid] : 0ull;
  u64 v[kBatch];
  load_rows(0, wave, 4, v);                              // chunk 0: all four waves, 16 rows each
  store_rows(0, wave, 4, v);
  if (wave != 0 && nchunks > 1) load_rows(1, wave - 1, 3, v);      // in flight across the barrier
  __syncthreads();
  u64 mykeep = 0;
  for (int c = 0; c < nchunks; ++c) {
#ifdef MSCNN_BO_TRACE
    if (tid == 0 && g_bo_trace && c < 64) g_bo_trace[64 + c] = __builtin_amdgcn_s_memrealtime();
#endif
    const u64* cur_rows = buf + (size_t)(c & 1) * 64 * W;
    if (wave != 0) {
      if (c + 1 < nchunks) {
        store_rows((c + 1) & 1, wave - 1, 3, v);         // requested one step ago
        if (c + 2 < nchunks) load_rows(c + 2, wave - 1, 3, v);
      }
    } else {
      const int valid = min(64, n - c * 64);
      const u64 dg = cur_rows[min(lane, 63) * W + c];    // diagonal word of row (c*64 + lane): bits j > lane it suppresses
      u64 live = ~uniform64(s_removed[c]);
      if (valid < 64) live &= (1ull << valid) - 1ull;
      u64 keep = 0;
      while (live) {                                     // wave-uniform: scalar loop over the KEPT boxes of the chunk
        const int i = __ffsll((long long)live) - 1;
        keep |= 1ull << i;
        live &= ~readlane64(dg, i);                      // boxes it suppresses
        live &= ~((2ull << i) - 1ull);                   // boxes up to and including i are decided
      }
      if (lane == c) mykeep = keep;
      if (lane == 0) s_keepw = keep;
    }
    __syncthreads();
    if (c + 1 < nchunks) {                               // (after the last chunk nobody reads the bitmap)
      u64 kk = (uniform64(s_keepw) >> (16 * wave)) & 0xffffull;
      if (lane < W && lane > c && kk) {                  // words <= c are never read again
        const u64* rows = cur_rows + (size_t)(16 * wave) * W + lane;
        u64 acc = 0;
        while (kk) {                                     // uniform loop over this wave's kept rows, four LDS reads in flight
          const int i0 = __ffsll((long long)kk) - 1; kk &= kk - 1;
          const int i1 = kk ? __ffsll((long long)kk) - 1 : i0; kk &= kk - 1;     // (kk & (kk - 1) of 0 is 0)
          const int i2 = kk ? __ffsll((long long)kk) - 1 : i0; kk &= kk - 1;
          const int i3 = kk ? __ffsll((long long)kk) - 1 : i0; kk &= kk - 1;
          const u64 r0 = rows[i0 * W], r1 = rows[i1 * W], r2 = rows[i2 * W], r3 = rows[i3 * W];
          acc |= (r0 | r1) | (r2 | r3);
        }
        if (acc) atomicOr(reinterpret_cast<unsigned long long*>(&s_removed[lane]), acc);
      }
    }
    __syncthreads();
  }
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
