// Device helpers shared by the split-fp16 ("x3") paths: exact fp16 hi + lo split, power-of-two scales from a tensor's
// max |x|, and the max |y| hand-over between layers (kAmaxSlots partial maxima per tensor; see mscnn_conv2d_plan_set_amax_io).
#pragma once
#include <hip/hip_runtime.h>

namespace mscnn {

constexpr int kAmaxSlots = 1024;   // == MSCNN_AMAX_SLOTS (mscnn_hip.h)

// s = 2^(14 - floor(log2(bound))): bound * s in [2^14, 2^15).  bound == 0 (all-zero tensor) -> 1.
__device__ __forceinline__ void pow2_scale(float bound, float* s, float* inv) {
  int e = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 127;
  if (bound == 0.f) e = 14;
  e = e < -100 ? -100 : (e > 110 ? 110 : e);
  *s = __uint_as_float((unsigned)(127 + 14 - e) << 23);
  *inv = __uint_as_float((unsigned)(127 - 14 + e) << 23);
}

// |v| < 2^15:  v = hi + lo up to max(2^-22 |v|, 2^-25)
__device__ __forceinline__ void split16(float v, _Float16* hi, _Float16* lo) {
  const _Float16 h = (_Float16)v;
  *hi = h;
  *lo = (_Float16)(v - (float)h);
}

// max over the kAmaxSlots published partial maxima: 4 loads per thread, served by the L2.  Every thread of the 256-thread
// workgroup must call this (it contains a barrier); returns the float whose bit pattern is the maximum.
// Two halves so that a kernel can put the slot loads in flight first and reduce them after its own global loads were issued.
__device__ __forceinline__ unsigned slots_partial(const unsigned* __restrict__ slots) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < kAmaxSlots / 256; ++i) m = max(m, slots[threadIdx.x + 256 * i]);
  return m;
}
__device__ __forceinline__ float slots_finish(unsigned m) {
  __shared__ unsigned s_b[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = m;
  __syncthreads();
  return __uint_as_float(max(max(s_b[0], s_b[1]), max(s_b[2], s_b[3])));
}
__device__ __forceinline__ float bound_from_slots(const unsigned* __restrict__ slots) {
  __shared__ unsigned s_b[4];
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < kAmaxSlots / 256; ++i) m = max(m, slots[threadIdx.x + 256 * i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = m;
  __syncthreads();
  return __uint_as_float(max(max(s_b[0], s_b[1]), max(s_b[2], s_b[3])));
}

// The workgroup's maximum of the bit patterns m goes, with ONE fire-and-forget atomic, into slot `slot` (any value; taken
// mod kAmaxSlots) -- thousands of workgroups hitting a single address cost 45 us per layer (measured), spread over 1024
// addresses nothing measurable.  Every thread of the 256-thread workgroup must call this (it contains a barrier).
__device__ __forceinline__ void publish_amax(unsigned m, unsigned* amax, unsigned slot) {
  __shared__ unsigned s_am[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) s_am[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax + (slot & (kAmaxSlots - 1)), max(max(s_am[0], s_am[1]), max(s_am[2], s_am[3])));
}

}  // namespace mscnn
