// Winograd F(3x3,3x3) with split-fp16 plane GEMMs for gfx950 (opt-in: MSCNN_CONV_ALGO_WINO_F3_X3).
//
// The fp32 MFMA pipe (v_mfma_f32_32x32x2_f32) runs at 1/16 of the fp16 pipe (v_mfma_f32_32x32x16_f16).  An fp32 number x
// with |x * s| < 2^15 (s a power of two) splits EXACTLY into two fp16 numbers up to 22 significant bits:
//     hi = fp16(x s),  lo = fp16(x s - hi),   |x s - hi - lo| <= max(2^-22 |x s|, 2^-25)
// (fp16 has an 11-bit significand; lo is the exact remainder rounded once more; below 2^-14 lo is a denormal with quantum
// 2^-24).  A product of two such numbers is then  a b = (ah bh + ah bl + al bh) + O(2^-22 |a b|): three fp16 MFMAs whose
// partial products are exact in the fp32 accumulator (11 x 11 bits), summed in fp32 exactly like the fp32 MFMA does.  The result
// carries a 2^-21 relative error per product against 2^-24 for fp32 operands -- far inside the 1e-4 parity bound and below the
// rounding of the Winograd transforms themselves (measured: tests/test_gpu_ops.py::test_conv_winograd_x3*) -- at
// 16/3 of the fp32 MFMA rate.
//
// Scales.  Weights: s_U from max |g| of the layer at pack time (|G g G^T| <= 2.25 max|g|), stored as 1/s_U in the header of
// the packed buffer.  Activations: s_V from max |x| of THIS forward's input, measured on the device by x3_amax (one streaming
// pass; |B^T d B| <= 36 max|d|) -- a frame can never overflow the fp16 range, whatever its statistics.  Both are powers of
// two, so scaling and un-scaling are exact.
//
// Layouts (16-byte units of 8 halves = one lane's MFMA operand, kg = 8-channel group):
//     U16[plane][part][kg][Cout_pad][8]     part 0 = hi, 1 = lo
//     V16[plane][part][kg][T_pad][8]        t = tile index (n, ty, tx), T_pad a multiple of 128
// so an A / B tile of the GEMM is, per (part, kg), one contiguous run of BM / 128 units: staged with b128 buffer loads and
// ds_write_b128, read back with one conflict-free ds_read_b128 per operand.  M comes out in the fp32 path's layout
// [plane][Cout][T_pad] so the output transforms (winograd.hip) are shared.
#include "wino_x3.h"
#include "winograd.h"
#include "x3_device.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr size_t kHdrBytes = 8192;     // packed weights: float[0] = 1 / s_U; bytes 4096.. = the kAmaxSlots partial maxima of |g|
constexpr unsigned kOob = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

using mscnn::pow2_scale; using mscnn::split16; using mscnn::bound_from_slots;

// ---- max |x| -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  __shared__ unsigned s_m[4];
  unsigned m = 0;
  const long n4 = n >> 2;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const uint4 v = x4[i];
    m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out + (blockIdx.x & (mscnn::kAmaxSlots - 1)), max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])));
}

// ---- weights: U = G g G^T, scaled, split ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_weight_kernel(const float* __restrict__ w, unsigned char* __restrict__ packed, int Cout,
                                                        int Cin, int Cout_pad, int KG) {
  const float amax = bound_from_slots(reinterpret_cast<const unsigned*>(packed + kHdrBytes / 2));
  float s, inv;
  pow2_scale(2.25f * amax, &s, &inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(packed)[0] = inv;
  _Float16* U = reinterpret_cast<_Float16*>(packed + kHdrBytes);
  const long part_stride = (long)KG * Cout_pad * 8, plane_stride = 2 * part_stride;
  const long total = (long)Cout_pad * KG * 8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int co = (int)(i % Cout_pad), ci = (int)(i / Cout_pad);
    const bool live = co < Cout && ci < Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = live ? w[((long)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
    float t[5][3];   // G g  (the same expressions as wino33_weight_kernel: identical U before the split)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = 0.5f * g[0][b];
      t[1][b] = -0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = (-g[0][b] + g[1][b] - g[2][b]) * (1.f / 6.f);
      t[3][b] = g[0][b] * (1.f / 6.f) + g[1][b] * (1.f / 3.f) + g[2][b] * (2.f / 3.f);
      t[4][b] = g[2][b];
    }
    _Float16* dst = U + ((long)(ci >> 3) * Cout_pad + co) * 8 + (ci & 7);
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const float x0 = t[a][0], x1 = t[a][1], x2 = t[a][2];
      const float u[5] = {0.5f * x0, -0.5f * (x0 + x1 + x2), (-x0 + x1 - x2) * (1.f / 6.f),
                          x0 * (1.f / 6.f) + x1 * (1.f / 3.f) + x2 * (2.f / 3.f), x2};
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        _Float16 hi, lo;
        split16(u[b] * s, &hi, &lo);
        dst[(a * 5 + b) * plane_stride] = hi;
        dst[(a * 5 + b) * plane_stride + part_stride] = lo;
      }
    }
  }
}

// ---- input transform -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bt5(const float d[5], float r[5]) {      // B^T of winograd.hip
  r[0] = 2.f * d[0] - d[1] - 2.f * d[2] + d[3];
  r[1] = -2.f * d[1] - d[2] + d[3];
  r[2] = 2.f * d[1] - 3.f * d[2] + d[3];
  r[3] = d[3] - d[1];
  r[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
}

// d (5x5 patch, registers) -> 25 scaled, split values in the LDS transpose buffer sh[part][plane][q][c] (halves)
__device__ __forceinline__ void transform_split_store(const float d[5][5], float s, _Float16* sh, int q, int c) {
  float r[5][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float col[5] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]};
    float o[5];
    bt5(col, o);
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float o[5];
    bt5(r[i], o);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      _Float16 hi, lo;
      split16(o[j] * s, &hi, &lo);
      sh[(((i * 5 + j)) * 32 + q) * 8 + c] = hi;
      sh[((25 + (i * 5 + j)) * 32 + q) * 8 + c] = lo;
    }
  }
}

// LDS transpose buffer -> V16: 50 runs (plane, part) of nq consecutive 16-byte units starting at tile t0
__device__ __forceinline__ void flush_units(const _Float16* sh, uint4* V16, int KG, long T_pad, int kg, long t0, int nq, int tid) {
  for (int u = tid; u < 50 * 32; u += 256) {
    const int q = u & 31, pp = u >> 5;          // pp = part * 25 + plane
    if (q >= nq) continue;
    const int part = pp / 25, plane = pp % 25;
    V16[((long)(plane * 2 + part) * KG + kg) * T_pad + t0 + q] = reinterpret_cast<const uint4*>(sh)[pp * 32 + q];
  }
}

// whole image planes: workgroup = 8 channels (one kg) x 32 consecutive tiles; a wave is 2 channels x 32 tiles
__global__ __launch_bounds__(256) void x3_input_plane_kernel(const float* __restrict__ x, uint4* __restrict__ V16,
                                                             const unsigned* __restrict__ scal, int N, int Cin, int H, int W,
                                                             int pad_h, int pad_w, int tiles_h, int tiles_w, int T, long T_pad,
                                                             int KG) {
  __shared__ __attribute__((aligned(16))) _Float16 sh[2 * 25 * 32 * 8];
  const int tid = threadIdx.x, q = tid & 31, c = tid >> 5;
  const int kg = blockIdx.y;
  const long t0 = (long)blockIdx.x * 32;
  const int t = (int)t0 + q, ci = kg * 8 + c;
  const unsigned slot_m = mscnn::slots_partial(scal);                  // in flight together with the patch loads below
  float d[5][5];
  if (t < T) {
    const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
    const float* src = x + ((long)n * Cin + ci) * H * W;
    const int h0 = 3 * ty - pad_h, w0 = 3 * tx - pad_w;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int h = h0 + i;
      const bool hok = h >= 0 && h < H;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int wv = w0 + j;
        d[i][j] = (hok && wv >= 0 && wv < W) ? src[h * W + wv] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) d[i][j] = 0.f;
  }
  float s, inv;
  pow2_scale(36.f * mscnn::slots_finish(slot_m), &s, &inv);
  transform_split_store(d, s, sh, q, c);
  __syncthreads();
  flush_units(sh, V16, KG, T_pad, kg, t0, 32, tid);
}

// ROI maps (H * W <= 64; roi_c1): workgroup = 8 channels x nr ROIs (nr * tiles-per-ROI <= 32); the nr x 8 maps are staged
// through LDS with coalesced loads (8 channels of one ROI are one contiguous run), then thread = (channel, roi, tile)
constexpr int kRoiMaxHW = 64, kRoiMax = 8;

__global__ __launch_bounds__(256) void x3_input_roi_kernel(const float* __restrict__ x, uint4* __restrict__ V16,
                                                           const unsigned* __restrict__ scal, int N, int Cin, int H, int W,
                                                           int pad_h, int pad_w, int tiles_h, int tiles_w, int T, long T_pad,
                                                           int KG, int nr) {
  __shared__ __attribute__((aligned(16))) _Float16 sh[2 * 25 * 32 * 8];
  extern __shared__ __attribute__((aligned(16))) float sm[];          // nr * 8 * H * W floats
  const int tid = threadIdx.x;
  const unsigned slot_m = mscnn::slots_partial(scal);                  // in flight while the maps are staged
  const int kg = blockIdx.y, c0 = kg * 8;
  const int r0 = blockIdx.x * nr;
  const int HW = H * W, tpr = tiles_h * tiles_w;
  const int run = 8 * HW;                        // contiguous floats per ROI (its 8 channels)
  const int live_r = min(nr, N - r0);
  if ((run & 3) == 0 && (((long)Cin * HW) & 3) == 0 && (((long)c0 * HW) & 3) == 0) {
    // one flat loop over (roi, float4) so that all 256 threads have loads in flight (98 float4 per ROI for 7x7 maps)
    const int per = run / 4;
    for (int i = tid; i < live_r * per; i += 256) {
      const int rl = i / per, e = i % per;
      reinterpret_cast<float4*>(sm + rl * run)[e] = reinterpret_cast<const float4*>(x + ((long)(r0 + rl) * Cin + c0) * HW)[e];
    }
  } else {
    for (int i = tid; i < live_r * run; i += 256) {
      const int rl = i / run, e = i % run;
      sm[rl * run + e] = x[((long)(r0 + rl) * Cin + c0) * HW + e];
    }
  }
  float s, inv;
  pow2_scale(36.f * mscnn::slots_finish(slot_m), &s, &inv);  // (its barrier also publishes the staged maps)
  const int nq = nr * tpr;                       // <= 32
  const int q = tid & 31, c = tid >> 5;
  if (q < nq) {
    const int rl = q / tpr, tl = q % tpr;
    float d[5][5];
    if (r0 + rl < N) {
      const int ty = tl / tiles_w, tx = tl % tiles_w;
      const float* map = sm + (rl * 8 + c) * HW;
      const int h0 = 3 * ty - pad_h, w0 = 3 * tx - pad_w;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int h = h0 + i;
        const bool hok = h >= 0 && h < H;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int wv = w0 + j;
          d[i][j] = (hok && wv >= 0 && wv < W) ? map[h * W + wv] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) d[i][j] = 0.f;
    }
    transform_split_store(d, s, sh, q, c);
  }
  __syncthreads();
  const long t0 = (long)r0 * tpr;
  int live = (int)min((long)nq, (long)T - t0);   // the last block's missing ROIs are not written here ...
  if (live < 0) live = 0;
  flush_units(sh, V16, KG, T_pad, kg, t0, live, tid);
  if (blockIdx.x == gridDim.x - 1) {             // ... the GEMM's padding columns T .. T_pad are zeros
    const int padn = (int)(T_pad - T);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (int u = tid; u < 50 * padn; u += 256) V16[((long)(u / padn) * KG + kg) * T_pad + T + u % padn] = z;
  }
}

// ---- the 25 plane GEMMs ------------------------------------------------------------------------------------------------------
struct X3Args {
  const void* U; const void* V; float* M; const unsigned* scal; const float* hdr;
  int Cout, Cout_pad, KG, KI, MT, NT, tiles, xcd_map;
  unsigned T_pad;
  int planes;        // 25 (Winograd), 1 (InnerProduct, proposal heads)
  float bound_mult;  // max |B operand| <= bound_mult * (max over the scal slots): 36 behind the Winograd transform, 1 otherwise
  int hw;            // BF32 kernels: V is the fp32 tensor x[K][hw] itself (split while it is staged), columns >= hw are zeros
  int ks;            // k-split: > 0 = every tile is cut into ks ranges of KI / ks chunks whose raw accumulators go to
  float* slabs;      //          slabs[((plane * NT + nt) * MT + mt) * ks + s][BM][128] (summed in s order by x3_fixup_kernel); 0 = off
};

// Workgroup = BM x 128 tile of one plane, 4 waves as 2 x 2, wave tile (BM/2) x 64 = MI x 2 MFMA blocks.  A k-chunk is 32
// channels (4 kg, two k = 16 MFMA steps): LDS A [part][kgl][BM] + B [part][kgl][128] units (BM 256: 48 KB, BM 128: 32 KB),
// next chunk prefetched into registers while the current one is multiplied.  Per k-step a wave reads 2 MI + 4 operands (1 KB
// each) for 6 MI MFMAs of 32 cycles: 64 B/clk (BM 256) / 85 B/clk (BM 128) per workgroup against the LDS's 256 B/clk.
// BF32: the B operand is staged from fp32 (8 coalesced channel-row loads per 16-byte unit, scaled and split in registers -- no
// separate split pass, no X16 in HBM; the proposal heads, whose B is the feature map itself)
template <int BM, bool BF32 = false>
__global__ __launch_bounds__(256, 2) void x3_gemm_kernel(X3Args a) {
  constexpr int MI = BM / 64, NI = 2, WM = BM / 2, AU = BM / 32, BU = 4, BFU = 2;
  __shared__ uint4 ldsA[2 * 4 * BM];
  __shared__ uint4 ldsB[2 * 4 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: XCD x (workgroups b == x mod 8) walks one contiguous range of (plane, nt, mt): the plane's U stays in
  // that XCD's L2 and the MT workgroups that share a V tile run next to each other
  int ti = (int)blockIdx.x;
  if (a.xcd_map) {
    const int xcd = ti % 8, gq = a.tiles / 8, gr = a.tiles % 8;
    ti = xcd * gq + min(xcd, gr) + ti / 8;
  }
  const int ks = a.ks > 0 ? a.ks : 1;
  const int per_plane = a.MT * a.NT * ks;
  const int plane = ti / per_plane, rem = ti % per_plane;
  const int nt = rem / (a.MT * ks), ksi = (rem / a.MT) % ks, mt = rem % a.MT;    // mt fastest: the workgroups sharing a B tile
  const int kc0 = ksi * (a.KI / ks), kc1 = kc0 + a.KI / ks;

  float sx = 1.f, inv_x = 1.f;
  if constexpr (BF32) pow2_scale(a.bound_mult * bound_from_slots(a.scal), &sx, &inv_x);
  const unsigned a_bytes = 2u * (unsigned)a.planes * (unsigned)a.KG * (unsigned)a.Cout_pad * 16u;
  const unsigned v_bytes = BF32 ? (unsigned)a.KG * 8u * (unsigned)a.hw * 4u : 2u * (unsigned)a.planes * (unsigned)a.KG * a.T_pad * 16u;
  const __amdgpu_buffer_rsrc_t asrc = make_rsrc(a.U, a_bytes), bsrc = make_rsrc(a.V, v_bytes);
  // unit u = tid + 256 i of a tile: (part, kgl, row) with row fastest
  unsigned a_off[AU], b_off[BU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = tid + 256 * i, part = u / (4 * BM), kgl = (u / BM) % 4, row = u % BM;
    a_off[i] = ((unsigned)((plane * 2 + part) * a.KG + kgl) * (unsigned)a.Cout_pad + (unsigned)(mt * BM + row)) * 16u;
  }
#pragma unroll
  for (int i = 0; i < BU; ++i) {
    const int u = tid + 256 * i, part = u / 512, kgl = (u / 128) % 4, col = u % 128;
    b_off[i] = ((unsigned)((plane * 2 + part) * a.KG + kgl) * a.T_pad + (unsigned)(nt * 128 + col)) * 16u;
  }
  const unsigned a_step = 4u * (unsigned)a.Cout_pad * 16u, b_step = BF32 ? 32u * (unsigned)a.hw * 4u : 4u * a.T_pad * 16u;   // one k-chunk further
  // BF32: unit u = tid + 256 i (i < 2) = (kgl = u / 128, column u % 128): 8 channels kgl * 8 .. + 7 of pixel nt * 128 + column
  unsigned f_off[BFU];
  if constexpr (BF32) {
#pragma unroll
    for (int i = 0; i < BFU; ++i) {
      const int u = tid + 256 * i, kgl = u / 128, t = nt * 128 + u % 128;
      f_off[i] = t < a.hw ? ((unsigned)(kgl * 8) * (unsigned)a.hw + (unsigned)t) * 4u : kOob;
    }
  }

  uint4 ra[AU], rb[BU];
  float rf[BFU][8];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < AU; ++i)
      ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_off[i], (unsigned)kc * a_step, 0));
    if constexpr (BF32) {
#pragma unroll
      for (int i = 0; i < BFU; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          rf[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bsrc, f_off[i], (unsigned)kc * b_step + (unsigned)j * (unsigned)a.hw * 4u, 0));
    } else {
#pragma unroll
      for (int i = 0; i < BU; ++i)
        rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, b_off[i], (unsigned)kc * b_step, 0));
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  load_chunk(kc0);
  for (int kc = kc0; kc < kc1; ++kc) {
#pragma unroll
    for (int i = 0; i < AU; ++i) ldsA[tid + 256 * i] = ra[i];
    if constexpr (BF32) {
#pragma unroll
      for (int i = 0; i < BFU; ++i) {
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          _Float16 hh, ll;
          split16(rf[i][j] * sx, &hh, &ll);
          h[j] = hh; l[j] = ll;
        }
        ldsB[tid + 256 * i] = __builtin_bit_cast(uint4, h);            // [part 0][kgl][column]
        ldsB[512 + tid + 256 * i] = __builtin_bit_cast(uint4, l);      // [part 1][kgl][column]
      }
    } else {
#pragma unroll
      for (int i = 0; i < BU; ++i) ldsB[tid + 256 * i] = rb[i];
    }
    __syncthreads();
    if (kc + 1 < kc1) load_chunk(kc + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kgl = 2 * s + khalf;
      f16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        ah[mi] = __builtin_bit_cast(f16x8, ldsA[kgl * BM + wm * WM + mi * 32 + l31]);
        al[mi] = __builtin_bit_cast(f16x8, ldsA[(4 + kgl) * BM + wm * WM + mi * 32 + l31]);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bh[ni] = __builtin_bit_cast(f16x8, ldsB[kgl * 128 + wn * 64 + ni * 32 + l31]);
        bl[ni] = __builtin_bit_cast(f16x8, ldsB[(4 + kgl) * 128 + wn * 64 + ni * 32 + l31]);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  if (a.ks > 0) {      // raw partial sums; x3_fixup_kernel adds the ks slabs of a tile in order, un-scales, adds the bias
    float* slab = a.slabs + ((size_t)((plane * a.NT + nt) * a.MT + mt) * ks + ksi) * (BM * 128);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float* sp = slab + (wm * WM + mi * 32 + 4 * khalf) * 128 + wn * 64 + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) sp[((r & 3) + 8 * (r >> 2)) * 128] = acc[mi][ni][r];
      }
    return;
  }
  // epilogue: un-scale (exact: powers of two), store M[plane][co][t]
  float sv, inv_v = inv_x;
  if constexpr (!BF32) pow2_scale(a.bound_mult * bound_from_slots(a.scal), &sv, &inv_v);
  const float inv = inv_v * a.hdr[0];
  const unsigned m_bytes = (unsigned)a.planes * (unsigned)a.Cout * a.T_pad * 4u;
  const __amdgpu_buffer_rsrc_t msrc = make_rsrc(a.M, m_bytes);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int co0 = mt * BM + wm * WM + mi * 32 + 4 * khalf;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const unsigned voff = ((unsigned)(plane * a.Cout + co0) * a.T_pad + (unsigned)(nt * 128 + wn * 64 + ni * 32 + l31)) * 4u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const unsigned vo = (co0 + row < a.Cout) ? voff : kOob;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[mi][ni][r] * inv), msrc, vo,
                                              (unsigned)row * a.T_pad * 4u, 0);
      }
    }
  }
}

// ---- InnerProduct: y[M][N] = x[M][K] w[N][K]^T + bias on the same GEMM kernel (A = the rows of x, B = the rows of w) ----------
// fp32 rows x[R][K] -> split units out[part][kg][R_pad][8], scaled by 2^e from the bound in `slots`; rows R .. R_pad are zeros.
// Thread = (row, kg): 32 contiguous bytes in, 2 x 16 bytes out; a wave covers 8 rows x 8 kg (256-byte runs in, 128-byte out).
__global__ __launch_bounds__(256) void x3_split_rows_kernel(const float* __restrict__ x, uint4* __restrict__ out,
                                                            const unsigned* __restrict__ slots, int R, int K, int R_pad, float* inv_out) {
  float s, inv;
  pow2_scale(bound_from_slots(slots), &s, &inv);
  if (inv_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *inv_out = inv;
  const int KG = K / 8;
  const int row = blockIdx.y * 32 + (threadIdx.x >> 3), kg = blockIdx.x * 8 + (threadIdx.x & 7);
  if (row >= R_pad || kg >= KG) return;
  f16x8 hi, lo;
  if (row < R) {
    const float4 a = *reinterpret_cast<const float4*>(x + (size_t)row * K + kg * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + (size_t)row * K + kg * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      _Float16 h, l;
      split16(v[i] * s, &h, &l);
      hi[i] = h; lo[i] = l;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) { hi[i] = (_Float16)0.f; lo[i] = (_Float16)0.f; }
  }
  out[(size_t)kg * R_pad + row] = __builtin_bit_cast(uint4, hi);
  out[((size_t)KG + kg) * R_pad + row] = __builtin_bit_cast(uint4, lo);
}

// y[row][col] = relu?((sum over the ks slabs of the tile, in order) * inv_x * inv_w + bias[col]); one workgroup per 128 x 128 tile
__global__ __launch_bounds__(256) void x3_fixup_kernel(const float* __restrict__ slabs, const float* __restrict__ inv_x,
                                                       const float* __restrict__ inv_w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int M, int N, int MT, int ks, int relu) {
  const int mt = blockIdx.x % MT, nt = blockIdx.x / MT;
  const float inv = inv_x[0] * inv_w[0];
  const float* base = slabs + (size_t)(nt * MT + mt) * ks * (128 * 128);
#pragma unroll 1
  for (int j = 0; j < 16; ++j) {
    const int i = (j * 256 + threadIdx.x) * 4;       // 4 consecutive columns of one row
    const int r = i / 128, c = i % 128;
    const int row = mt * 128 + r, col = nt * 128 + c;
    if (row >= M) continue;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < ks; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(base + (size_t)s * (128 * 128) + i);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    float o[4] = {v.x * inv, v.y * inv, v.z * inv, v.w * inv};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (col + q >= N) continue;
      float t = o[q] + (bias ? bias[col + q] : 0.f);
      if (relu) t = t < 0.f ? 0.f : t;
      y[(size_t)row * N + col + q] = t;
    }
  }
}

// ---- proposal heads as a GEMM + shift-and-add ---------------------------------------------------------------------------------
// A K x K convolution with a handful of output channels over 512 input channels (LFCN_*: Cout 6..9, 5x5 / 7x7 / 3x5 / 5x7) is a
// poor MFMA shape as a convolution (M = Cout), but  y[co][p] = sum_tap ( sum_c w[co][c][tap] x[c][p + off(tap)] )  is ONE dense
// GEMM  T[tap * Cout + co][p] = W'[tap * Cout + co][c] x[c][p]  (M = taps * Cout = 225 / 441 rows, K = Cin, N = pixels: no patch, no
// halo, x is its own B operand) followed by a shift-and-add over the taps -- the transposed view of im2col + GEMM (col2im after the
// GEMM instead of im2col before it).  Same multiplies as the direct form; T is taps * Cout * H * W floats (30 MB for the 7x7 head on
// conv4_3).  In the split-fp16 arithmetic: W' packed once, x split by the GEMM itself while it stages its B tile (BF32).
// w [Cout][Cin][taps] -> W16[part][kg][rows_pad][8], row = tap * Cout + co; scale from the slots behind the header
__global__ __launch_bounds__(256) void x3_head_weight_kernel(const float* __restrict__ w, unsigned char* __restrict__ packed, int Cout,
                                                             int Cin, int taps, int rows_pad) {
  float s, inv;
  pow2_scale(bound_from_slots(reinterpret_cast<const unsigned*>(packed + kHdrBytes / 2)), &s, &inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(packed)[0] = inv;
  _Float16* W16 = reinterpret_cast<_Float16*>(packed + kHdrBytes);
  const int KG = Cin / 8;
  const long part_stride = (long)KG * rows_pad * 8, total = (long)rows_pad * Cin;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i % rows_pad), ci = (int)(i / rows_pad);
    const int tap = row / Cout, co = row % Cout;
    const float v = row < taps * Cout ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
    _Float16 hi, lo;
    split16(v * s, &hi, &lo);
    const long o = ((long)(ci >> 3) * rows_pad + row) * 8 + (ci & 7);
    W16[o] = hi;
    W16[part_stride + o] = lo;
  }
}

// y[co][oy][ox] = bias[co] + sum over taps (kh, kw ascending) of T[tap * Cout + co][(oy + kh - ph) * W + ox + kw - pw], taps that fall
// outside the map skipped (zero padding); thread = (co, output pixel), lanes along ox: every T row is read as contiguous runs
__global__ __launch_bounds__(256) void x3_head_shift_add_kernel(const float* __restrict__ T, const float* __restrict__ bias,
                                                                float* __restrict__ y, int Cout, int H, int W, int Ho, int Wo, int KH,
                                                                int KW, int ph, int pw, unsigned T_pad, int relu) {
  const int p = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
  if (p >= Ho * Wo) return;
  const int oy = p / Wo, ox = p % Wo;
  float acc = bias ? bias[co] : 0.f;
  for (int kh = 0; kh < KH; ++kh) {
    const int iy = oy + kh - ph;
    if (iy < 0 || iy >= H) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int ix = ox + kw - pw;
      if (ix < 0 || ix >= W) continue;
      acc += T[(size_t)((kh * KW + kw) * Cout + co) * T_pad + (unsigned)(iy * W + ix)];
    }
  }
  if (relu) acc = acc < 0.f ? 0.f : acc;
  y[((size_t)co * Ho + oy) * Wo + ox] = acc;
}

}  // namespace

namespace mscnn {

bool x3_head_plan(int Cin, int Cout, int KH, int KW, long HW, X3HeadPlan* out) {
  if (Cin % 32 != 0 || Cin < 32 || Cout < 1 || Cout > 16 || KH * KW < 2 || KH * KW > 64 || HW < 1) return false;
  X3HeadPlan p;
  p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.KG = Cin / 8;
  p.rows = KH * KW * Cout;
  p.rows_pad = (p.rows + 127) / 128 * 128;
  p.T_pad = (HW + 127) / 128 * 128;
  const double w_bytes = 2.0 * p.KG * p.rows_pad * 16.0, x_bytes = (double)Cin * HW * 4.0, t_bytes = (double)p.rows * p.T_pad * 4.0;
  if (w_bytes >= 4.0e9 || x_bytes >= 4.0e9 || t_bytes >= 4.0e9) return false;
  p.packed_bytes = kHdrBytes + (size_t)w_bytes;
  p.t_bytes = (size_t)t_bytes;
  *out = p;
  return true;
}

int x3_head_pack(const X3HeadPlan& p, const float* w, void* packed, hipStream_t st) {
  unsigned* hdr = static_cast<unsigned*>(packed);
  const int rc = x3_amax(w, (long)p.Cout * p.Cin * p.KH * p.KW, hdr + kHdrBytes / 8, st);
  if (rc != MSCNN_OK) return rc;
  long blocks = ((long)p.rows_pad * p.Cin + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  x3_head_weight_kernel<<<(int)blocks, 256, 0, st>>>(w, static_cast<unsigned char*>(packed), p.Cout, p.Cin, p.KH * p.KW, p.rows_pad);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

// one image: x [Cin][H][W] -> y [Cout][Ho][Wo].  ws: [4 KB own slots][T]
int x3_head_forward(const X3HeadPlan& p, const float* x, const void* packed, const float* bias, float* y, int H, int W, int Ho, int Wo,
                    int pad_h, int pad_w, int relu, const unsigned* in_bound, void* ws, hipStream_t st) {
  unsigned char* wsb = static_cast<unsigned char*>(ws);
  unsigned* own = reinterpret_cast<unsigned*>(wsb);
  float* T = reinterpret_cast<float*>(wsb + 4096);
  const int HW = H * W;
  if (!in_bound) {
    const int rc = x3_amax(x, (long)p.Cin * HW, own, st);
    if (rc != MSCNN_OK) return rc;
  }
  const unsigned* slots = in_bound ? in_bound : own;
  const unsigned char* pk = static_cast<const unsigned char*>(packed);
  X3Args a;
  a.U = pk + kHdrBytes; a.V = x; a.hw = HW; a.M = T; a.scal = slots; a.hdr = reinterpret_cast<const float*>(pk);
  a.Cout = p.rows; a.Cout_pad = p.rows_pad; a.KG = p.KG; a.KI = p.KG / 4; a.MT = p.rows_pad / 128; a.NT = (int)(p.T_pad / 128);
  a.tiles = a.MT * a.NT; a.xcd_map = 1; a.T_pad = (unsigned)p.T_pad;
  a.planes = 1; a.ks = 0; a.slabs = nullptr; a.bound_mult = 1.f;
  x3_gemm_kernel<128, true><<<a.tiles, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return head_shift_add(T, bias, y, p.Cout, H, W, Ho, Wo, p.KH, p.KW, pad_h, pad_w, (unsigned)p.T_pad, relu, st);
}

// the shift-and-add of the GEMM-over-taps form, shared with the fp32 variant (conv.hip head_gemm_plan): T rows of T_pad floats
int head_shift_add(const float* T, const float* bias, float* y, int Cout, int H, int W, int Ho, int Wo, int KH, int KW, int pad_h,
                   int pad_w, unsigned T_pad, int relu, hipStream_t st) {
  x3_head_shift_add_kernel<<<dim3(cdiv(Ho * Wo, 256), Cout), 256, 0, st>>>(T, bias, y, Cout, H, W, Ho, Wo, KH, KW, pad_h, pad_w, T_pad, relu);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

bool x3_plan(int Cin, int Cout, long T_pad, int tune_variant, X3Plan* out) {
  if (Cin % 32 != 0 || Cin < 32 || T_pad % 128 != 0) return false;
  X3Plan p;
  p.Cin = Cin; p.Cout = Cout; p.KG = Cin / 8; p.T_pad = T_pad;
  p.NT = (int)(T_pad / 128);
  // 256-row tiles (tune_variant 2) halve the LDS and L2 traffic per MFMA but sit at 234 VGPRs; measured equal or slower on every
  // layer of the 7s-576 net (conv4_2 GEMM 87 vs 94 us, roi_c1 233 vs 244, conv3_2 109 vs 107): 128 rows is the default
  p.BM = (tune_variant == 2 && Cout % 256 == 0) ? 256 : 128;
  p.MT = cdiv(Cout, p.BM);
  p.Cout_pad = p.MT * p.BM;
  const double u_bytes = 50.0 * p.KG * p.Cout_pad * 16.0, v_bytes = 50.0 * p.KG * (double)T_pad * 16.0;
  const double m_bytes = 25.0 * Cout * (double)T_pad * 4.0;
  if (u_bytes >= 4.0e9 || v_bytes >= 4.0e9 || m_bytes >= 4.0e9) return false;     // 32-bit buffer windows
  p.packed_bytes = kHdrBytes + (size_t)u_bytes;
  p.v_bytes = (size_t)v_bytes;
  *out = p;
  return true;
}

int x3_amax(const float* x, long n, unsigned* scal, hipStream_t st) {
  MSCNN_HIP_TRY(hipMemsetAsync(scal, 0, sizeof(unsigned) * kAmaxSlots, st));
  long blocks = (n / 4 + 255) / 256;
  if (blocks > kAmaxSlots) blocks = kAmaxSlots;       // one atomic per slot
  if (blocks < 1) blocks = 1;
  amax_kernel<<<(int)blocks, 256, 0, st>>>(x, n, scal);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int x3_pack_weights(const X3Plan& p, const float* w, void* packed, hipStream_t st) {
  unsigned* hdr = static_cast<unsigned*>(packed);
  int rc = x3_amax(w, (long)p.Cout * p.Cin * 9, hdr + kHdrBytes / 8, st);
  if (rc != MSCNN_OK) return rc;
  const long total = (long)p.Cout_pad * p.KG * 8;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  x3_weight_kernel<<<(int)blocks, 256, 0, st>>>(w, static_cast<unsigned char*>(packed), p.Cout, p.Cin, p.Cout_pad, p.KG);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int x3_input_transform(const X3Plan& p, const float* x, void* V16, const unsigned* scal, int N, int H, int W, int pad_h,
                       int pad_w, int tiles_h, int tiles_w, hipStream_t st) {
  const int T = N * tiles_h * tiles_w;
  if (H * W <= kRoiMaxHW && tiles_h * tiles_w <= 32) {
    int nr = 32 / (tiles_h * tiles_w);
    if (nr > kRoiMax) nr = kRoiMax;
    dim3 grid(cdiv(N, nr), p.KG);
    x3_input_roi_kernel<<<grid, 256, sizeof(float) * nr * 8 * H * W, st>>>(x, static_cast<uint4*>(V16), scal, N, p.Cin, H, W, pad_h, pad_w, tiles_h, tiles_w, T,
                                              p.T_pad, p.KG, nr);
  } else {
    dim3 grid((unsigned)(p.T_pad / 32), p.KG);
    x3_input_plane_kernel<<<grid, 256, 0, st>>>(x, static_cast<uint4*>(V16), scal, N, p.Cin, H, W, pad_h, pad_w, tiles_h, tiles_w,
                                                T, p.T_pad, p.KG);
  }
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int x3_gemm(const X3Plan& p, const void* packed, const void* V16, float* M, const unsigned* scal, int xcd_map, hipStream_t st) {
  X3Args a;
  a.U = static_cast<const unsigned char*>(packed) + kHdrBytes;
  a.V = V16; a.M = M; a.scal = scal; a.hdr = static_cast<const float*>(packed);
  a.Cout = p.Cout; a.Cout_pad = p.Cout_pad; a.KG = p.KG; a.KI = p.KG / 4; a.MT = p.MT; a.NT = p.NT;
  a.tiles = 25 * p.MT * p.NT; a.xcd_map = xcd_map; a.T_pad = (unsigned)p.T_pad;
  a.planes = 25; a.ks = 0; a.slabs = nullptr; a.bound_mult = 36.f; a.hw = 0;
  if (p.BM == 256) x3_gemm_kernel<256><<<a.tiles, 256, 0, st>>>(a);
  else x3_gemm_kernel<128><<<a.tiles, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn

// ---- C ABI: InnerProduct on the split-fp16 GEMM -------------------------------------------------------------------------------
using namespace mscnn;

namespace {
struct IpShape { int KG, KI, N_pad, NT, M_pad, MT, ks; size_t x_bytes, slab_bytes; };
IpShape ip_shape(int M, int N, int K) {
  IpShape s;
  s.KG = K / 8; s.KI = K / 32;
  s.N_pad = (N + 127) / 128 * 128; s.NT = s.N_pad / 128;
  s.M_pad = (M + 127) / 128 * 128; s.MT = s.M_pad / 128;
  s.ks = 1;      // enough workgroups for 256 CUs x 3: cut K (the slabs are summed in order: deterministic)
  for (int k = 1; k <= 16; ++k)
    if (s.KI % k == 0) { s.ks = k; if ((long)s.MT * s.NT * k >= 768) break; }
  s.x_bytes = (size_t)2 * s.KG * s.M_pad * 16;
  s.slab_bytes = (size_t)s.MT * s.NT * s.ks * 128 * 128 * sizeof(float);
  return s;
}
}  // namespace

extern "C" int mscnn_inner_product_x3_supported(int N, int K) { return N >= 128 && K % 32 == 0 && K >= 32; }

extern "C" size_t mscnn_inner_product_x3_packed_bytes(int N, int K) {
  return kHdrBytes + (size_t)2 * (K / 8) * ((N + 127) / 128 * 128) * 16;
}

extern "C" size_t mscnn_inner_product_x3_workspace_bytes(int M, int N, int K) {
  const IpShape s = ip_shape(M, N, K);
  return 8192 + s.x_bytes + s.slab_bytes;
}

extern "C" int mscnn_inner_product_x3_pack(const float* w, void* packed, int N, int K, void* stream) {
  MSCNN_REQUIRE(w && packed && mscnn_inner_product_x3_supported(N, K), "inner_product x3 pack: bad argument (N=%d K=%d)", N, K);
  hipStream_t st = as_stream(stream);
  unsigned char* pk = static_cast<unsigned char*>(packed);
  unsigned* slots = reinterpret_cast<unsigned*>(pk + kHdrBytes / 2);
  int rc = x3_amax(w, (long)N * K, slots, st);
  if (rc != MSCNN_OK) return rc;
  const int N_pad = (N + 127) / 128 * 128;
  dim3 grid(cdiv(K / 8, 8), cdiv(N_pad, 32));
  x3_split_rows_kernel<<<grid, 256, 0, st>>>(w, reinterpret_cast<uint4*>(pk + kHdrBytes), slots, N, K, N_pad, reinterpret_cast<float*>(pk));
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_inner_product_x3_fwd(const float* x, const void* packed, const float* bias, float* y, int M, int N, int K,
                                          int relu, const uint32_t* in_bound, void* workspace, size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(M >= 0 && N > 0 && K > 0, "inner_product: bad shape M=%d N=%d K=%d", M, N, K);
  if (M == 0) return MSCNN_OK;
  MSCNN_REQUIRE(x && packed && y && mscnn_inner_product_x3_supported(N, K) && reinterpret_cast<uintptr_t>(x) % 16 == 0,
                "inner_product x3: needs N >= 128, K %% 32 == 0, 16-byte aligned x (N=%d K=%d)", N, K);
  const IpShape s = ip_shape(M, N, K);
  if (!workspace || workspace_bytes < 8192 + s.x_bytes + s.slab_bytes) {
    set_error("inner_product x3: workspace %zu < %zu", workspace_bytes, 8192 + s.x_bytes + s.slab_bytes);
    return MSCNN_ERR_WORKSPACE;
  }
  MSCNN_REQUIRE((double)s.x_bytes < 4.0e9 && 2.0 * s.KG * s.N_pad * 16.0 < 4.0e9, "inner_product x3: operand beyond the 32-bit window");
  hipStream_t st = as_stream(stream);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  unsigned* slots = reinterpret_cast<unsigned*>(ws);
  float* inv_x = reinterpret_cast<float*>(ws + 4096);
  uint4* X16 = reinterpret_cast<uint4*>(ws + 8192);
  float* slabs = reinterpret_cast<float*>(ws + 8192 + s.x_bytes);
  if (!in_bound) {
    const int rc = x3_amax(x, (long)M * K, slots, st);
    if (rc != MSCNN_OK) return rc;
  }
  dim3 grid(cdiv(s.KG, 8), cdiv(s.M_pad, 32));
  x3_split_rows_kernel<<<grid, 256, 0, st>>>(x, X16, in_bound ? in_bound : slots, M, K, s.M_pad, inv_x);
  MSCNN_POST_LAUNCH();
  const unsigned char* pk = static_cast<const unsigned char*>(packed);
  X3Args a;
  a.U = X16; a.V = pk + kHdrBytes; a.M = nullptr; a.scal = nullptr; a.hdr = nullptr;
  a.Cout = M; a.Cout_pad = s.M_pad; a.KG = s.KG; a.KI = s.KI; a.MT = s.MT; a.NT = s.NT;
  a.tiles = s.MT * s.NT * s.ks; a.xcd_map = 1; a.T_pad = (unsigned)s.N_pad;
  a.planes = 1; a.ks = s.ks; a.slabs = slabs; a.bound_mult = 1.f; a.hw = 0;
  x3_gemm_kernel<128><<<a.tiles, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  x3_fixup_kernel<<<s.MT * s.NT, 256, 0, st>>>(slabs, inv_x, reinterpret_cast<const float*>(pk), bias, y, M, N, s.MT, s.ks, relu);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
