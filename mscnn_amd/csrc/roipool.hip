// ROIPooling with MS-CNN context padding for gfx950.
// Reference: ROIPoolingLayer<Dtype>::Forward_gpu, src/caffe/layers/roi_pooling_layer.cu:19-104
// (CPU twin roi_pooling_layer.cpp:48-139).  Pure compare/select arithmetic => bit-exact.
//
// Mapping: one workgroup per (roi, channel block).  The ROI geometry (4 roundf + bin sizes) is
// computed once per workgroup into SGPR-uniform values instead of once per output as the
// reference kernel does.  A wavefront owns channels; its 64 lanes cover the PH*PW bins of one or
// more channels, so the stores of a wave are one contiguous run of out[r][c..][:][:] (coalesced),
// and the feature-map reads of a wave stay inside a handful of rows of one channel plane
// (conv4_3 is 35 MB: L2 / Infinity-Cache resident across ROIs).
#include "common.h"
#include <cfloat>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void roipool_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                           float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                           float spatial_scale, float pad_ratio, int C_total, int c_offset,
                                                           int chan_per_block) {
  const int r = blockIdx.y;
  const int c_begin = blockIdx.x * chan_per_block;
  const int c_end = min(C, c_begin + chan_per_block);

  const float* roi = rois + 5 * (size_t)r;
  const int b = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  // roi_pooling_layer.cu:33-41 -- context padding, no clipping of the roi itself
  const float pad_w = (x2 - x1 + 1) * pad_ratio;
  const float pad_h = (y2 - y1 + 1) * pad_ratio;
  const int roi_start_w = (int)roundf((x1 - pad_w) * spatial_scale);
  const int roi_start_h = (int)roundf((y1 - pad_h) * spatial_scale);
  const int roi_end_w = (int)roundf((x2 + pad_w) * spatial_scale);
  const int roi_end_h = (int)roundf((y2 + pad_h) * spatial_scale);
  const int roi_width = max(roi_end_w - roi_start_w + 1, 1);
  const int roi_height = max(roi_end_h - roi_start_h + 1, 1);
  const float bin_size_h = (float)roi_height / (float)PH;
  const float bin_size_w = (float)roi_width / (float)PW;

  const int bins = PH * PW;
  const int work = (c_end - c_begin) * bins;
  const float* fbase = feat + (size_t)b * C * H * W;
  float* obase = out + ((size_t)r * C_total + c_offset) * bins;

  for (int i = threadIdx.x; i < work; i += kThreads) {
    const int c = c_begin + i / bins;
    const int bin = i % bins;
    const int ph = bin / PW, pw = bin % PW;
    int hstart = (int)floorf((float)ph * bin_size_h);
    int wstart = (int)floorf((float)pw * bin_size_w);
    int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
    int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
    hstart = min(max(hstart + roi_start_h, 0), H);
    hend = min(max(hend + roi_start_h, 0), H);
    wstart = min(max(wstart + roi_start_w, 0), W);
    wend = min(max(wend + roi_start_w, 0), W);
    const bool is_empty = (hend <= hstart) || (wend <= wstart);
    float maxval = is_empty ? 0.f : -FLT_MAX;
    const float* plane = fbase + (size_t)c * H * W;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) {
        const float v = plane[h * W + w];
        if (v > maxval) maxval = v;   // first max wins; value only (argmax is not needed at TEST)
      }
    obase[(size_t)c * bins + bin] = maxval;
  }
}


// Separable form, one wavefront per (roi, channel): a max over a rectangular bin is the max over its columns of the
// per-column maxima.  Phase 1: lane = feature column; it walks the rows of every bin row ph and keeps PH running maxima
// (every row is a coalesced read across lanes).  Phase 2: lane = output bin (ph, pw); it reduces its <= ceil(bin_w)+1
// columns out of LDS.  The ROI window is read ~once instead of once per overlapping bin by strided lanes, and the
// PH*PW outputs of a (roi, channel) are one contiguous store.  max() is order independent -> still bit-exact.
constexpr int kMaxPH = 8;

__global__ __launch_bounds__(256) void roipool_wave_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                           float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                           float spatial_scale, float pad_ratio, int C_total, int c_offset,
                                                           int chan_per_wave) {
  __shared__ float colmax[4][kMaxPH][64];
  const int r = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c_begin = (blockIdx.x * 4 + wave) * chan_per_wave;
  const int c_end = min(C, c_begin + chan_per_wave);

  const float* roi = rois + 5 * (size_t)r;
  const int b = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const float pad_w = (x2 - x1 + 1) * pad_ratio;
  const float pad_h = (y2 - y1 + 1) * pad_ratio;
  const int roi_start_w = (int)roundf((x1 - pad_w) * spatial_scale);
  const int roi_start_h = (int)roundf((y1 - pad_h) * spatial_scale);
  const int roi_end_w = (int)roundf((x2 + pad_w) * spatial_scale);
  const int roi_end_h = (int)roundf((y2 + pad_h) * spatial_scale);
  const int roi_width = max(roi_end_w - roi_start_w + 1, 1);
  const int roi_height = max(roi_end_h - roi_start_h + 1, 1);
  const float bin_size_h = (float)roi_height / (float)PH;
  const float bin_size_w = (float)roi_width / (float)PW;

  // row range of every bin row (uniform), column range of this lane's output bin
  int hs[kMaxPH], he[kMaxPH];
#pragma unroll
  for (int ph = 0; ph < kMaxPH; ++ph) {
    hs[ph] = min(max((int)floorf((float)ph * bin_size_h) + roi_start_h, 0), H);
    he[ph] = min(max((int)ceilf((float)(ph + 1) * bin_size_h) + roi_start_h, 0), H);
    if (ph >= PH) he[ph] = hs[ph] = 0;
  }
  const int bins = PH * PW;
  const int my_ph = lane / PW, my_pw = lane % PW;
  const int ws = min(max((int)floorf((float)my_pw * bin_size_w) + roi_start_w, 0), W);
  const int we = min(max((int)ceilf((float)(my_pw + 1) * bin_size_w) + roi_start_w, 0), W);
  // clamped column span of the whole roi
  const int w_lo = min(max(roi_start_w, 0), W);
  const int w_hi = min(max((int)ceilf((float)PW * bin_size_w) + roi_start_w, 0), W);

  const float* fbase = feat + (size_t)b * C * H * W;
  float* obase = out + ((size_t)r * C_total + c_offset) * bins;
  float (*cm)[64] = colmax[wave];

  for (int c = c_begin; c < c_end; ++c) {
    const float* plane = fbase + (size_t)c * H * W;
    float m = -FLT_MAX;
    for (int w0 = w_lo; w0 < w_hi; w0 += 64) {
      const int w = w0 + lane;
      const bool col_ok = w < w_hi;
#pragma unroll
      for (int ph = 0; ph < kMaxPH; ++ph) {
        float v = -FLT_MAX;
        if (col_ok)
          for (int h = hs[ph]; h < he[ph]; ++h) v = fmaxf(v, plane[h * W + w]);
        cm[ph][lane] = v;
      }
      // same wave wrote and reads: LDS ops of one wave are ordered, only the compiler needs to be told
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < bins) {
        const int lo = max(ws, w0), hi = min(we, w0 + 64);
        for (int ww = lo; ww < hi; ++ww) m = fmaxf(m, cm[my_ph][ww - w0]);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < bins) {
      const bool is_empty = (he[my_ph] <= hs[my_ph]) || (we <= ws);
      obase[(size_t)c * bins + lane] = is_empty ? 0.f : m;
    }
  }
}

}  // namespace

using namespace mscnn;

extern "C" int mscnn_roipool_fwd_f32(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W,
                                     int pooled_h, int pooled_w, float spatial_scale, float pad_ratio, int C_total,
                                     int c_offset, void* stream) {
  MSCNN_REQUIRE(feat && rois && out, "roipool: null pointer");
  MSCNN_REQUIRE(R >= 0 && N > 0 && C > 0 && H > 0 && W > 0, "roipool: bad shape");
  MSCNN_REQUIRE(pooled_h > 0 && pooled_w > 0, "roipool: pooled_h/pooled_w must be > 0");   // roi_pooling_layer.cpp:26-29
  MSCNN_REQUIRE(c_offset >= 0 && c_offset + C <= C_total, "roipool: channel window outside the output buffer");
  if (R == 0) return MSCNN_OK;
  // XCD-aware launch geometry: the feature map (conv4_3: 35 MB) does not fit one XCD's 4 MB L2, but 1/8 of its channels
  // does.  Workgroups are dispatched round-robin over the 8 XCDs by linear id, so with the CHANNEL GROUP on grid.x (a
  // multiple of 8 groups) XCD j only ever touches channel groups == j (mod 8): its L2 keeps those planes hot across all
  // ROIs instead of every XCD streaming the whole map from the Infinity Cache for every ROI.
  const int bins = pooled_h * pooled_w;
  static const bool use_wave = [] { const char* e = getenv("MSCNN_ROIPOOL_WAVE"); return e && *e == '1'; }();
  if (use_wave && bins <= 64 && pooled_h <= kMaxPH) {
    int chan_per_wave = 8;
    while (chan_per_wave > 1 && (C % (32 * chan_per_wave)) != 0) chan_per_wave >>= 1;
    dim3 grid(cdiv(C, 4 * chan_per_wave), R);
    roipool_wave_kernel<<<grid, 256, 0, as_stream(stream)>>>(feat, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale,
                                                             pad_ratio, C_total, c_offset, chan_per_wave);
    MSCNN_POST_LAUNCH();
    return MSCNN_OK;
  }
  int chan_per_block = max(1, (kThreads * 4) / bins);
  if (chan_per_block > C) chan_per_block = C;
  if (C % 128 == 0) chan_per_block = 16;            // C/16 channel groups: a multiple of 8
  dim3 grid(cdiv(C, chan_per_block), R);
  roipool_kernel<<<grid, kThreads, 0, as_stream(stream)>>>(feat, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale,
                                                           pad_ratio, C_total, c_offset, chan_per_block);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
