// ROIPooling x 2 (ROI + context window) fused with the F(3x3,3x3) input transform of the convolution that consumes them (roi_c1)
// for gfx950.  Replaces, for the detection sub-net of the deploy files,
//     ROIPoolingLayer::Forward_gpu x 2  (src/caffe/layers/roi_pooling_layer.cu:19-104)  ->  ConcatLayer::Forward_gpu
//     (concat_layer.cu:28-46)  ->  the im2col of roi_c1 (conv_layer.cu:8-23)
// by ONE pass from the feature map to the transform planes V[25][2C][T] the plane GEMM (wgemm.hip) reads: the R x 2C x 7 x 7
// blob between them (140 MB per 7s-576 frame: written by the pooling kernel, read back by the input transform) is never
// materialised on this path (the Net writes it on demand with roipool.hip's kernel when somebody asks for the blob by name).
//
// Why another pooling kernel.  roipool.hip's row kernel gives a SLOT of lanes one channel and lets the lanes walk the ROI's columns of
// an NCHW plane: a wave load touches 64 / Wp different rows -- 32 .. 128-byte segments of as many cache lines -- and 37 % of the
// lanes idle on ROIs whose width is not a power of two; it moves ~4 GB through the L1s for the 2 x 0.28 GB it needs (281 us,
// profiles/r03_*).  And a ROI's cost is its area: 300 .. 24,000 loads per 64 channels on the benchmark frame.
// Here
//   * LANES OWN CHANNELS of a channel-last copy of the map (L0[n][h][w][c]): every load of a wave is one pixel x 64 channels = two full
//     cache lines whatever the ROI's shape, no lane idles, and a wave handles one ROI, so its control flow (bin edges, level choice,
//     loop bounds) is scalar;
//   * every bin costs the same few loads whatever its size: three more maps hold the SLIDING maxima over s x s squares,
//     Lk[y][x] = max L0[y .. y + s)[x .. x + s), s = 2, 4, 8 (each built from the one before by a 4-point maximum: 35 MB written per
//     level).  max is idempotent, so a bin [hs, he) x [ws, we) is EXACTLY the maximum of the squares of the largest s <= min(h, w)
//     anchored at hs, hs + s, .., he - s by ws, ws + s, .., we - s: 2 x 4 of them along the bin's longer side cover every bin up to an
//     aspect ratio of 2 .. 4 (longer ones finish in a short loop).  A 7 x 7 pooling of the biggest ROI of the frame reads ~800 values
//     per channel instead of ~24,000 -- and no wave runs 80x longer than its neighbours;
//   * the 8 loads of four bins are issued together (32 independent loads in flight per lane), no branch per load.
// max is exact and order independent on post-ReLU maps, the bin edges are the reference's float expressions
// (roi_pooling_layer.cu:33-59): the pooled values are the reference's, bit for bit; V = B^T d B uses the very expressions of
// winograd.hip's ROI input transform (wino33_device.h), so V -- and everything behind it -- is bit-identical to the unfused path.
#include "roipool_wino.h"
#include "wino33_device.h"
#include <cfloat>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kNB = 4;        // bins whose 8 squares are loaded together (32 loads in flight per lane)
constexpr int kLevels = 4;    // L0 (the map itself, channel-last) + sliding maxima over 2 x 2, 4 x 4, 8 x 8

struct RpwArgs {
  const float* maps;          // [kLevels][N][H][W][C]
  const float* rois;          // [R][5]
  float* V;                   // [25][2C][T_pad]
  int R, C, H, W, T_pad;
  unsigned level_bytes, maps_bytes, v_bytes;
  float spatial_scale, pad_a, pad_b;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// max of three / two floats in one instruction.  Against the reference's `if (v > m) m = v` chain (roi_pooling_layer.cu:66-71): NaNs are
// skipped by both; a tie between -0 and +0 may come out with the other sign (equal values; post-ReLU maps hold no -0 anyway).
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// hides a value's history from the optimiser (no instruction): keeps it from sharing sub-expressions across unrolled iterations
__device__ __forceinline__ void launder(float& v) { asm volatile("" : "+v"(v)); }

// One wave = one ROI x 64 channels of one window (blockIdx.z 0: pad_a -> channels [0, C) of the concatenated blob, 1: pad_b -> [C, 2C));
// one workgroup = kRois consecutive ROIs, so that the 16-byte units the waves produce (a lane's four tiles of one plane) leave through
// LDS as whole 128-byte lines of V: written straight from the lanes they are 64 partial lines per store instruction, 11 KB apart,
// and the kernel ran at 0.86 TB/s (330 us, tools/sessions/r04_s7.sh).
constexpr int kRois = 8;
constexpr int kLdsRow = kRois * 4 + 4;      // floats per (plane, channel) row in LDS: 32 + 4 of padding (lane stride 144 B: no bank conflicts)

template <int PH, int PW>
__global__ __launch_bounds__(kRois * 64) void roipool_wino33_kernel(RpwArgs a) {
  static_assert(PH == 7 && PW == 7, "tiles per ROI: 2 x 2 tiles of 3 x 3 outputs (a 16-byte unit of V per plane)");
  __shared__ __attribute__((aligned(16))) float s_out[5 * 64 * kLdsRow];
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int r0 = blockIdx.y * kRois;
  const int r = r0 + wave;
  const bool live = r < a.R;                                    // (a wave without a ROI still takes part in the barriers)
  const int c0 = blockIdx.x * 64;
  const int c = c0 + lane;
  const __amdgpu_buffer_rsrc_t rF = make_rsrc(a.maps, a.maps_bytes), rV = make_rsrc(a.V, a.v_bytes);
  const unsigned lane_off = (unsigned)c * 4u;

  const float* roi = a.rois + 5 * (size_t)(live ? r : 0);
  const int b = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const int H = a.H, W = a.W, C = a.C;
  const int row0 = b * H;
  // every address is wave-uniform but for the lane's channel: offsets are byte counts in SGPRs (a pixel is pxB bytes, a map row rowB)
  const unsigned pxB = (unsigned)C * 4u, rowB = (unsigned)W * pxB;
  auto load_at = [&](unsigned off) {
#ifdef RPW_NO_POOL
    return (float)(off + lane);
#endif
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rF, lane_off, off, 0));
  };

  {
    const int q = blockIdx.z;      // the window: a workgroup pools ONE of the two (1360 half-size workgroups spread over the CUs better than 680)
    // ---- window geometry and bin edges: roi_pooling_layer.cu:33-59, the same float expressions (context padding, no clipping of
    // the ROI itself, edges clipped to the map); every value is wave-uniform
    const float pad_ratio = q == 0 ? a.pad_a : a.pad_b;
    const float pad_w = (x2 - x1 + 1) * pad_ratio;
    const float pad_h = (y2 - y1 + 1) * pad_ratio;
    const int roi_start_w = (int)roundf((x1 - pad_w) * a.spatial_scale);
    const int roi_start_h = (int)roundf((y1 - pad_h) * a.spatial_scale);
    const int roi_end_w = (int)roundf((x2 + pad_w) * a.spatial_scale);
    const int roi_end_h = (int)roundf((y2 + pad_h) * a.spatial_scale);
    const int roi_width = max(roi_end_w - roi_start_w + 1, 1);
    const int roi_height = max(roi_end_h - roi_start_h + 1, 1);
    const float bin_size_h = (float)roi_height / (float)PH;
    const float bin_size_w = (float)roi_width / (float)PW;
    const bool wide = roi_width >= roi_height;               // the squares of a bin are laid 2 x 4 along the ROI's longer side
    // per bin column: width (0 = empty), and -- for the loads -- a clamped copy that always lies inside the map (an empty bin reads one
    // valid pixel and is overwritten with 0 below: no branch around the loads)
    int bw_[PW];
    unsigned wsB[PW], weB[PW];
#pragma unroll
    for (int pw = 0; pw < PW; ++pw) {
      const int ws = uni(min(max((int)floorf((float)pw * bin_size_w) + roi_start_w, 0), W));
      const int we = uni(min(max((int)ceilf((float)(pw + 1) * bin_size_w) + roi_start_w, 0), W));
      bw_[pw] = we - ws;
      wsB[pw] = (unsigned)min(ws, W - 1) * pxB;
      weB[pw] = (unsigned)max(we, min(ws, W - 1) + 1) * pxB;
    }

    float pooled[PH * PW];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      const int hs = uni(min(max((int)floorf((float)ph * bin_size_h) + roi_start_h, 0), H));
      const int he = uni(min(max((int)ceilf((float)(ph + 1) * bin_size_h) + roi_start_h, 0), H));
      const int h = he - hs;
      const unsigned hcl = (unsigned)max(h, 1);
      const unsigned hsB = (unsigned)(row0 + min(hs, H - 1)) * rowB;
      // A bin is the maximum of 2 x 4 squares of the largest level s <= min(h, w), four along the ROI's longer side: exact whenever the
      // bin is at most 2 s by 4 s (s > min(h, w) / 2: every bin up to an aspect ratio of 2 .. 4); longer ones finish in the loop below.
      // (The level is the BIN's: one level per bin row -- from the row's narrowest bin -- was measured 1.6x slower, a window clipped
      // by the map's border has one-column bins that drag every other bin of the row down to s = 1.)  Four bins x 8 squares are in
      // flight together (32 independent loads per lane), then the row's other three.
      const unsigned heB = hsB + hcl * rowB;
#pragma unroll
      for (int half = 0; half < (PW + kNB - 1) / kNB; ++half) {
        constexpr int NB = kNB;
        float v[NB][8];
        int lev[NB];
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
          const int pw = half * NB + bi;
          if (pw < PW) {
            const int k = min(3, 31 - __builtin_clz(min(hcl, (unsigned)max(bw_[pw], 1))));
            lev[bi] = k;
            const unsigned stepx = pxB << k, stepy = rowB << k;
            const unsigned xlastB = weB[pw] - stepx, ylastB = heB - stepy;
            const unsigned base = (unsigned)k * a.level_bytes;
            // along the longer side: start / last / step; across: start / last (byte offsets add up in any order)
            const unsigned al0 = wide ? wsB[pw] : hsB, alL = wide ? xlastB : ylastB, alS = wide ? stepx : stepy;
            const unsigned ac0 = base + (wide ? hsB : wsB[pw]), ac1 = base + (wide ? ylastB : xlastB);
            const unsigned a1 = min(al0 + alS, alL), a2 = min(al0 + 2 * alS, alL), a3 = min(al0 + 3 * alS, alL);
            v[bi][0] = load_at(ac0 + al0); v[bi][1] = load_at(ac0 + a1); v[bi][2] = load_at(ac0 + a2); v[bi][3] = load_at(ac0 + a3);
            v[bi][4] = load_at(ac1 + al0); v[bi][5] = load_at(ac1 + a1); v[bi][6] = load_at(ac1 + a2); v[bi][7] = load_at(ac1 + a3);
          }
        }
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
          const int pw = half * NB + bi;
          if (pw < PW) {
            const int w = bw_[pw];
            float m = max3(v[bi][0], v[bi][1], v[bi][2]);
            m = max3(m, v[bi][3], v[bi][4]);
            m = max3(m, v[bi][5], v[bi][6]);
            m = max3(m, v[bi][7], -FLT_MAX);
            const int s_ = 1 << lev[bi];
            const int across_len = wide ? h : w, along_len = wide ? w : h;
            if (across_len > 2 * s_ || along_len > 4 * s_) {           // (wave-uniform, rare) more squares than the batch covered
              const unsigned stepx = pxB << lev[bi], stepy = rowB << lev[bi];
              const unsigned xlastB = weB[pw] - stepx, ylastB = heB - stepy, base = (unsigned)lev[bi] * a.level_bytes;
              for (unsigned yB = hsB; yB < heB; yB += stepy)
                for (unsigned xB = wsB[pw]; xB < weB[pw]; xB += 4 * stepx) {
                  const unsigned rb = base + min(yB, ylastB);
                  const float u0 = load_at(rb + min(xB, xlastB)), u1 = load_at(rb + min(xB + stepx, xlastB));
                  const float u2 = load_at(rb + min(xB + 2 * stepx, xlastB)), u3 = load_at(rb + min(xB + 3 * stepx, xlastB));
                  m = max3(m, u0, u1);
                  m = max3(m, u2, u3);
                }
            }
            pooled[ph * PW + pw] = (h <= 0 || w <= 0) ? 0.f : m;      // empty bins are 0 (roi_pooling_layer.cu:56-58)
          }
        }
      }
    }

#ifdef RPW_DBG_POOLED      // development: the pooled values themselves, value k at plane k % 25, column 4 r + k / 25
    {
      const unsigned rb = ((unsigned)(q * C) * (unsigned)a.T_pad + (unsigned)r * 4u) * 4u, vl = (unsigned)c * (unsigned)a.T_pad * 4u;
      const unsigned pb = (unsigned)(2 * C) * (unsigned)a.T_pad * 4u;
      if (live)
#pragma unroll
        for (int k = 0; k < PH * PW; ++k)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pooled[k]), rV, vl, rb + (unsigned)(k % 25) * pb + (unsigned)(k / 25) * 4u, 0);
      return;
    }
#endif
    // ---- V = B^T d B of the ROI's 2 x 2 tiles (patch rows / columns 3 t .. 3 t + 4 of the 7 x 7 map, zero beyond it): the operation
    // order of wino33_input_kernel.  Plane row i at a time: every wave parks its five 16-byte units (plane (i, j), tiles 0 .. 3) in
    // LDS as s_out[j][channel][4 wave ..], then the workgroup writes the 5 x 64 rows of kRois x 4 columns as 128-byte runs of V.
    const unsigned plane_bytes = (unsigned)(2 * C) * (unsigned)a.T_pad * 4u;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      // (plane row i needs row i of every column transform: recomputed per i from the pooled values -- laundered, so that the
      // optimiser does not keep all 4 x 5 x 5 of them alive across the five iterations -- 100 registers for 60 flops saved)
#pragma unroll
      for (int k = 0; k < PH * PW; ++k) launder(pooled[k]);
      float out[5][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ty = t >> 1, tx = t & 1;
        float rr[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float d[5], o[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const int h = 3 * ty + k, w = 3 * tx + j;
            d[k] = (h < PH && w < PW) ? pooled[h * PW + w] : 0.f;
          }
          mscnn::bt5(d, o);
          rr[j] = o[i];
        }
        float o2[5];
        mscnn::bt5(rr, o2);
#pragma unroll
        for (int j = 0; j < 5; ++j) out[j][t] = o2[j];
      }
      __syncthreads();                                         // (the previous plane row has left s_out)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        *reinterpret_cast<float4*>(&s_out[(j * 64 + lane) * kLdsRow + wave * 4]) = make_float4(out[j][0], out[j][1], out[j][2], out[j][3]);
      __syncthreads();
      // 5 planes x 64 channels x kRois units; consecutive threads take consecutive ROIs of one (plane, channel) row
#pragma unroll
      for (int n = 0; n < 5 * 64 * kRois / (kRois * 64); ++n) {
        const int item = tid + n * (kRois * 64);
        const int rq = item % kRois, ch = (item / kRois) % 64, j = item / (kRois * 64);
        if (r0 + rq < a.R) {
          const float4 u = *reinterpret_cast<const float4*>(&s_out[(j * 64 + ch) * kLdsRow + rq * 4]);
          u32x4 pk;
          pk.x = __builtin_bit_cast(unsigned, u.x); pk.y = __builtin_bit_cast(unsigned, u.y);
          pk.z = __builtin_bit_cast(unsigned, u.z); pk.w = __builtin_bit_cast(unsigned, u.w);
          // (the whole offset in the VGPR, SOFFSET = 0: with a 16-byte store whose SOFFSET is an SGPR the compiler's hazard recogniser
          // assumes no store-data hazard and lets the next VALU instruction overwrite the data registers one cycle later -- on gfx950
          // that clobbered dword 1 of lanes 12 .. 15 of every 16 before the store had read them (tools/sessions/r04_s5.sh))
          const unsigned off = (unsigned)(i * 5 + j) * plane_bytes + ((unsigned)(q * C + c0 + ch) * (unsigned)a.T_pad + (unsigned)(r0 + rq) * 4u) * 4u;
#ifndef RPW_NO_STORE
          __builtin_amdgcn_raw_buffer_store_b128(pk, rV, off, 0u, 0);
#else
          if (a.T_pad < 0) __builtin_amdgcn_raw_buffer_store_b128(pk, rV, off, 0u, 0);
#endif
        }
      }
    }
  }
}

// L0[n][hw][c] = feat[n][c][hw]: 64 x 64 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
  __shared__ float t[64][65];
  const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = x + (size_t)n * C * HW;
  float* dst = y + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int cc = c0 + ty + 4 * i, pp = p0 + tx;
    t[ty + 4 * i][tx] = (cc < C && pp < HW) ? src[(size_t)cc * HW + pp] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pp = p0 + ty + 4 * i, cc = c0 + tx;
    if (pp < HW && cc < C) dst[(size_t)pp * C + cc] = t[tx][ty + 4 * i];
  }
}

// Lk[y][x] = max of Lk-1 at (y, x), (y, x + h), (y + h, x), (y + h, x + h), h = 2^(k-1): the sliding maximum over 2^k x 2^k squares.
// Positions whose square leaves the map are never queried (a bin's squares lie inside the bin, a bin inside the map); they are
// filled with the clamped value.  One thread = one pixel x 4 channels (float4): a wave moves 1 KB runs.
__global__ __launch_bounds__(256) void sliding_max_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C4, int half,
                                                          long total) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= total) return;
  const int c4 = (int)(i % C4);
  long p = i / C4;
  const int x = (int)(p % W); p /= W;
  const int y = (int)(p % H);
  const long n = p / H;
  const int y2 = min(y + half, H - 1), x2 = min(x + half, W - 1);
  const float4* s4 = reinterpret_cast<const float4*>(src) + n * H * W * C4;
  const float4 a = s4[((long)y * W + x) * C4 + c4], b = s4[((long)y * W + x2) * C4 + c4];
  const float4 c = s4[((long)y2 * W + x) * C4 + c4], d = s4[((long)y2 * W + x2) * C4 + c4];
  float4 m = a;      // (the order of the comparisons does not matter: maxima of the same values; ties are equal bit patterns on post-ReLU maps)
  m.x = b.x > m.x ? b.x : m.x; m.y = b.y > m.y ? b.y : m.y; m.z = b.z > m.z ? b.z : m.z; m.w = b.w > m.w ? b.w : m.w;
  m.x = c.x > m.x ? c.x : m.x; m.y = c.y > m.y ? c.y : m.y; m.z = c.z > m.z ? c.z : m.z; m.w = c.w > m.w ? c.w : m.w;
  m.x = d.x > m.x ? d.x : m.x; m.y = d.y > m.y ? d.y : m.y; m.z = d.z > m.z ? d.z : m.z; m.w = d.w > m.w ? d.w : m.w;
  reinterpret_cast<float4*>(dst)[i] = m;
}

}  // namespace

namespace mscnn {

bool roipool_wino33_supported(int C, int pooled_h, int pooled_w, int conv_pad_h, int conv_pad_w) {
  return pooled_h == 7 && pooled_w == 7 && conv_pad_h == 0 && conv_pad_w == 0 && C > 0 && C % 64 == 0;
}

size_t roipool_wino33_scratch_bytes(int N, int C, int H, int W) { return (size_t)kLevels * N * C * H * W * sizeof(float); }

int roipool_wino33_build_maps(const float* feat, float* maps, int N, int C, int H, int W, hipStream_t st) {
  MSCNN_REQUIRE(feat && maps, "roipool maps: null pointer");
  MSCNN_REQUIRE(N > 0 && C % 64 == 0 && H > 0 && W > 0, "roipool maps: bad shape");
  MSCNN_REQUIRE(kLevels * (double)N * C * H * W * 4.0 < 4.0e9, "roipool maps: beyond a 32-bit buffer window");
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(maps) % 16 == 0, "roipool maps: the maps must be 16-byte aligned");
  const int HW = H * W;
  const long per_level = (long)N * C * HW;
  nchw_to_nhwc_kernel<<<dim3(cdiv(HW, 64), cdiv(C, 64), N), 256, 0, st>>>(feat, maps, C, HW);
  MSCNN_POST_LAUNCH();
  for (int k = 1; k < kLevels; ++k) {
    const long total = per_level / 4;
    sliding_max_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(maps + (k - 1) * per_level, maps + k * per_level, H, W, C / 4, 1 << (k - 1), total);
    MSCNN_POST_LAUNCH();
  }
  return MSCNN_OK;
}

int roipool_wino33_forward(const float* maps, const float* rois, float* V, int R, int N, int C, int H, int W, int T_pad,
                           float spatial_scale, float pad_a, float pad_b, hipStream_t st) {
  MSCNN_REQUIRE(maps && rois && V, "roipool+transform: null pointer");
  MSCNN_REQUIRE(R > 0 && N > 0 && C % 64 == 0 && H > 0 && W > 0 && T_pad >= 4 * R && T_pad % 4 == 0, "roipool+transform: bad shape");
  const double lb = (double)N * C * H * W * 4.0, vb = 25.0 * 2.0 * C * (double)T_pad * 4.0;
  MSCNN_REQUIRE(kLevels * lb < 4.0e9 && vb < 4.0e9, "roipool+transform: feature maps or transform planes beyond a 32-bit buffer window");
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(V) % 16 == 0, "roipool+transform: V must be 16-byte aligned");
  RpwArgs a;
  a.maps = maps; a.rois = rois; a.V = V;
  a.R = R; a.C = C; a.H = H; a.W = W; a.T_pad = T_pad;
  a.level_bytes = (unsigned)lb; a.maps_bytes = (unsigned)(kLevels * lb); a.v_bytes = (unsigned)vb;
  a.spatial_scale = spatial_scale; a.pad_a = pad_a; a.pad_b = pad_b;
  // channel block on grid.x: workgroups go round-robin over the 8 XCDs by linear id, so XCD j only ever touches channel blocks
  // == j (mod 8) of the maps -- with C = 512 exactly one 64-channel slice (4 x 4.4 MB) per XCD's L2
  roipool_wino33_kernel<7, 7><<<dim3(C / 64, cdiv(R, kRois), 2), kRois * 64, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
