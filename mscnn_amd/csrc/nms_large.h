// More boxes than the LDS-resident fast path holds (K > kMaxK = 4032): a global-memory bitonic sort and a TILED greedy NMS.
// Shared by boxoutput.hip (BoxOutput with max_nms_num 0 = "no cap" -- the caffe.proto default -- or > 4032, and the stand-alone
// NMS entry) and detections.hip (final stage over more than 4032 ROIs).  The fast paths are untouched; these run only when the
// host-side bound on K exceeds kMaxK.
//
// Greedy NMS (nmsMax, box_output_layer.cpp:38-63; bbNms.m:112-126) walks the boxes in sorted order, but a box only ever
// depends on the KEPT boxes before it: j is dropped iff some kept i < j overlaps it.  So the sorted list is cut into tiles of
// kMaxK boxes and tile t
//   1. tests its boxes against the kept boxes of the tiles before it (final by then; one wavefront per 64 boxes, the kept box
//      is a wave-uniform load) -> the tile's initial removed-bitmap,
//   2. builds its own kMaxK x kMaxK upper-triangular bit matrix with the fast path's mask kernel,
//   3. runs the fast path's one-wavefront scan (greedy_scan) seeded with that bitmap and appends its kept boxes to the list.
// The result is the reference's keep set exactly (same predicate, same order); memory stays at one tile's bit matrix (2 MB)
// whatever K is.
#pragma once
#include "box_device.h"

namespace mscnn_dev {

constexpr int kBigSortLocal = 2048;       // keys one workgroup sorts / merges in LDS (16 KB)
constexpr int kTileWords = kMaxK / 64;    // 63 bitmap words per tile row

// ---- descending bitonic sort of P = 2^k >= kBigSortLocal 64-bit keys in global memory ---------------------------------------
// Network position = GLOBAL index, same compare rule as bitonic_desc.  Sizes size_lo..size_hi, strides <= 1024 only.
static __global__ __launch_bounds__(1024) void big_sort_local_kernel(u64* __restrict__ keys, int size_lo, int size_hi) {
  __shared__ u64 sk[kBigSortLocal];
  const int tid = threadIdx.x;
  const int gbase = blockIdx.x * kBigSortLocal;
  sk[tid] = keys[gbase + tid];
  sk[tid + 1024] = keys[gbase + tid + 1024];
  __syncthreads();
  for (int size = size_lo; size <= size_hi; size <<= 1) {
    int stride = size >> 1;
    if (stride > (kBigSortLocal >> 1)) stride = kBigSortLocal >> 1;
    for (; stride > 0; stride >>= 1) {
      const int lo = (tid / stride) * (stride << 1) + (tid % stride);
      const int hi = lo + stride;
      const bool desc = (((gbase + lo) & size) == 0);
      const u64 a = sk[lo], b = sk[hi];
      const bool swap = desc ? (a < b) : (a > b);
      if (swap) { sk[lo] = b; sk[hi] = a; }
      __syncthreads();
    }
  }
  keys[gbase + tid] = sk[tid];
  keys[gbase + tid + 1024] = sk[tid + 1024];
}

// one compare-exchange per thread for a stride that spans workgroups (stride >= kBigSortLocal)
static __global__ __launch_bounds__(256) void big_sort_step_kernel(u64* __restrict__ keys, int size, int stride, int half) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= half) return;
  const int lo = (t / stride) * (stride << 1) + (t % stride);
  const int hi = lo + stride;
  const bool desc = ((lo & size) == 0);
  const u64 a = keys[lo], b = keys[hi];
  const bool swap = desc ? (a < b) : (a > b);
  if (swap) { keys[lo] = b; keys[hi] = a; }
}

inline int big_sort_pow2(int n) {
  int P = kBigSortLocal;
  while (P < n) P <<= 1;
  return P;
}

// keys[0..P): real keys are unique and non-zero, the tail is zero padding (sorts last).  Returns a hipError_t.
inline hipError_t big_sort_desc(u64* keys, int P, hipStream_t st) {
  big_sort_local_kernel<<<P / kBigSortLocal, 1024, 0, st>>>(keys, 2, kBigSortLocal);
  for (int size = 2 * kBigSortLocal; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride >= kBigSortLocal; stride >>= 1)
      big_sort_step_kernel<<<(P / 2 + 255) / 256, 256, 0, st>>>(keys, size, stride, P / 2);
    big_sort_local_kernel<<<P / kBigSortLocal, 1024, 0, st>>>(keys, size, size);
  }
  return hipPeekAtLastError();
}

// ---- tiled greedy NMS -------------------------------------------------------------------------------------------------------
// Tr: struct with `Box`, `Params` and `static __device__ bool over(const Box& kept_earlier, const Box& later, const Params&)`
// -- the predicate the tile's mask kernel evaluates, argument order included (IOFU is not symmetric).
// state[0] = boxes in this tile, state[1] = kept so far (both device ints; state[1] must be 0 before the first tile).
enum { BIG_TILE_N = 0, BIG_NKEPT = 1, BIG_STATE_WORDS = 2 };

template <class Tr>
__global__ __launch_bounds__(64) void big_nms_cross_kernel(const typename Tr::Box* __restrict__ boxes,
                                                           const int* __restrict__ total_k, int n_fixed, int base,
                                                           const typename Tr::Box* __restrict__ kept_box,
                                                           typename Tr::Params p, u64* __restrict__ removed_init,
                                                           int* __restrict__ state) {
  const int K = total_k ? *total_k : n_fixed;
  int nt = K - base;
  nt = nt < 0 ? 0 : (nt > kMaxK ? kMaxK : nt);
  const int w = blockIdx.x, lane = threadIdx.x;
  if (w == 0 && lane == 0) state[BIG_TILE_N] = nt;
  const int j = w * 64 + lane;
  bool rem = false;
  if (j < nt) {
    const typename Tr::Box b = boxes[base + j];
    const int nk = state[BIG_NKEPT];
    for (int i = 0; i < nk; ++i) rem = rem || Tr::over(kept_box[i], b, p);
  }
  const u64 bits = __ballot(rem);
  if (lane == 0) removed_init[w] = bits;
}

template <class Tr>
__global__ __launch_bounds__(256) void big_nms_scan_kernel(const u64* __restrict__ mask,
                                                           const typename Tr::Box* __restrict__ boxes, int base,
                                                           const u64* __restrict__ removed_init, int* __restrict__ kept_idx,
                                                           typename Tr::Box* __restrict__ kept_box, int* __restrict__ state) {
  extern __shared__ __attribute__((aligned(16))) u64 dyn_lds[];
  __shared__ u64 keepw[64];
  __shared__ int pre[64];
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = state[BIG_TILE_N];
  if (n <= 0) return;
  const int nk0 = state[BIG_NKEPT];
  const u64 mykeep = greedy_scan(mask, n, kTileWords, kTileWords, dyn_lds, removed_init);
  if (wave == 0) {
    keepw[lane] = mykeep;
    const int mine = __popcll(mykeep);
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    pre[lane] = incl - mine;
    if (lane == 63) s_total = incl;
  }
  __syncthreads();
  for (int k = tid; k < n; k += 256) {
    const int c = k >> 6, l = k & 63;
    const u64 kw = keepw[c];
    if (!((kw >> l) & 1ull)) continue;
    const int pos = nk0 + pre[c] + __popcll(kw & ((1ull << l) - 1ull));
    kept_idx[pos] = base + k;
    kept_box[pos] = boxes[base + k];
  }
  __syncthreads();
  if (tid == 0) state[BIG_NKEPT] = nk0 + s_total;
}

// All tiles of a list of at most `kcap` sorted boxes (their number on the device: *total_k, or n_fixed when total_k is null).
// launch_mask(boxes_of_tile, count_ptr) launches the caller's bit-matrix kernel on one tile with kTileWords words per row.
// On return (stream order) kept_idx[0 .. state[BIG_NKEPT]) holds the kept boxes' sorted indices in ascending order.
template <class Tr, class MaskFn>
inline hipError_t big_nms_tiles(const typename Tr::Box* sbox, const int* total_k, int n_fixed, int kcap, typename Tr::Params p,
                                u64* mask, u64* removed_init, int* kept_idx, typename Tr::Box* kept_box, int* state,
                                MaskFn launch_mask, hipStream_t st) {
  const int tiles = (kcap + kMaxK - 1) / kMaxK;
  for (int t = 0; t < tiles; ++t) {
    const int base = t * kMaxK;
    big_nms_cross_kernel<Tr><<<kTileWords, 64, 0, st>>>(sbox, total_k, n_fixed, base, kept_box, p, removed_init, state);
    launch_mask(sbox + base, state + BIG_TILE_N);
    big_nms_scan_kernel<Tr><<<1, 256, (size_t)2 * 64 * kTileWords * sizeof(u64), st>>>(mask, sbox, base, removed_init, kept_idx,
                                                                                       kept_box, state);
  }
  return hipPeekAtLastError();
}

}  // namespace mscnn_dev
