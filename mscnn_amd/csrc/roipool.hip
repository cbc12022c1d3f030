// ROIPooling with MS-CNN context padding for gfx950.
// Reference: ROIPoolingLayer<Dtype>::Forward_gpu, src/caffe/layers/roi_pooling_layer.cu:19-104
// (CPU twin roi_pooling_layer.cpp:48-139).  Pure compare/select arithmetic => bit-exact.
//
// Mapping: one workgroup per (channel group, roi).  The ROI geometry (4 roundf + bin sizes) is
// computed once per workgroup into SGPR-uniform values instead of once per output as the
// reference kernel does; consecutive lanes are consecutive bins then channels, so the stores of a
// wave are one contiguous run of out[r][c..][:][:].  The launch geometry is XCD-aware (see below):
// measured 1.19 -> 0.81 ms for both ROI poolings of a 700-ROI frame.  (A separable
// wave-per-(roi,channel) variant was measured slower, 1.08 ms: low lane utilisation on narrow ROIs.)
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


// ---- row-coalesced ROI max pooling -----------------------------------------------------------------------------------
// The per-output kernel above makes every lane of a wave walk its own window: 64 different cache lines per load
// instruction and one dependent load chain per lane.  Here a SLOT of Wp lanes (Wp = the ROI's clipped width rounded up to a
// power of two, 8..128) owns one channel of the workgroup's ROI: the lanes read the ROI's feature rows as contiguous row
// segments (64/Wp channels per wave instruction), up to 24 independent row loads in flight per lane (4 bin-rows x 6 rows, rows past a bin masked off),
// reduce them to per-column maxima of every bin-row ph, park those PH x Wp values in LDS, and then finish the PH x PW bins
// of the channel with a short max over each bin's columns, written as one contiguous run of out[r][c][:][:].
// max is exact and order independent (inputs are post-ReLU: no -0/+0 or NaN ordering question arises), so the result is
// bit-identical to the reference's scan order (roi_pooling_layer.cu:60-76).  Bin edges are computed with the reference's
// float expressions, once per workgroup.
constexpr int kMaxP = 16;        // pooled_h / pooled_w limit of this kernel
constexpr int kMaxSpan = 128;
constexpr int kRowsInFlight = 6;  // rows of a bin fetched in the first batch (4 bins x 6 rows = 24 loads in flight per lane; measured 4: 303 us, 6: 282 us, 8: 305 us for both poolings)    // widest clipped ROI (feature columns) it handles; wider ROIs take the per-bin loop

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// NP = 1: one pooling; NP = 2: the ROI is pooled twice with two context paddings into two channel windows of the same output
// (roi_pool_ctx + roi_pool_org of the deploy nets, whose Concat is fused away) in ONE launch -- for each group of channels the
// wider (context) window first, then the ROI's own: its rows were just read by the same workgroup.
struct RoiPoolPass { float pad_ratio; int c_offset; };

template <int NP>
__global__ __launch_bounds__(kThreads) void roipool_rows_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                                float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                                float spatial_scale, RoiPoolPass pass0, RoiPoolPass pass1, int C_total,
                                                                int chan_per_block) {
  __shared__ int s_h0[NP][kMaxP], s_h1[NP][kMaxP], s_w0[NP][kMaxP], s_w1[NP][kMaxP];
  __shared__ unsigned char s_ph[kMaxP * kMaxP], s_pw[kMaxP * kMaxP];
  extern __shared__ float s_col[];            // 512 * PH floats: [slot][ph][Wp]
  const int tid = threadIdx.x;
  const int r = blockIdx.y;
  const int c_begin = blockIdx.x * chan_per_block;
  const int nchan = min(C, c_begin + chan_per_block) - c_begin;
  const int bins = PH * PW;

  const float* roi = rois + 5 * (size_t)r;
  const int b = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const float pad_ratio = q == 0 ? pass0.pad_ratio : pass1.pad_ratio;
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
    if (tid < PH) {
      s_h0[q][tid] = min(max((int)floorf((float)tid * bin_size_h) + roi_start_h, 0), H);
      s_h1[q][tid] = min(max((int)ceilf((float)(tid + 1) * bin_size_h) + roi_start_h, 0), H);
    } else if (tid >= 64 && tid < 64 + PW) {
      const int pw = tid - 64;
      s_w0[q][pw] = min(max((int)floorf((float)pw * bin_size_w) + roi_start_w, 0), W);
      s_w1[q][pw] = min(max((int)ceilf((float)(pw + 1) * bin_size_w) + roi_start_w, 0), W);
    }
  }
  if (tid < bins) { s_ph[tid] = (unsigned char)(tid / PW); s_pw[tid] = (unsigned char)(tid % PW); }
  __syncthreads();
  const int HW = H * W;
  const float* fbase = feat + (size_t)b * C * HW + (size_t)c_begin * HW;

  // per pass: span of clipped columns and the slot geometry (both uniform over the workgroup)
  int x_lo_[NP], x_hi_[NP], Wp_[NP];
  int group = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    x_lo_[q] = s_w0[q][0]; x_hi_[q] = s_w1[q][PW - 1];      // both edge sequences are non-decreasing in pw
    int Wp = 8;
    while (Wp < min(x_hi_[q] - x_lo_[q], kMaxSpan)) Wp <<= 1;
    Wp_[q] = Wp;
    group = max(group, kThreads / min(Wp, 64));
  }

  // channels in groups of `group` (the narrower pass's slot count: that pass does a group in one round, the wider one in two or
  // more), context pass first.  The passes map slots to LDS differently, hence the workgroup barrier between them.
  for (int c0 = 0; c0 < nchan; c0 += group)
#pragma unroll
  for (int qq = 0; qq < NP; ++qq) {
    const int q = NP == 2 ? 1 - qq : 0;        // pass 1 = the padded (context) window runs first
    if (NP == 2) __syncthreads();
    const int c_offset = q == 0 ? pass0.c_offset : pass1.c_offset;
    float* obase = out + ((size_t)r * C_total + c_offset + c_begin) * bins;
    const int x_lo = x_lo_[q], x_hi = x_hi_[q];
    const int span = x_hi - x_lo;
    const int cend = min(nchan, c0 + group);
    if (span <= 0) {
      // nothing inside the map: every bin is empty -> 0 (the loop below finds no cell)
      for (int i = c0 * bins + tid; i < cend * bins; i += kThreads) {
        const int c = i / bins, bin = i % bins, ph = s_ph[bin], pw = s_pw[bin];
        const int hs = s_h0[q][ph], he = s_h1[q][ph], ws = s_w0[q][pw], we = s_w1[q][pw];
        float m = (he <= hs || we <= ws) ? 0.f : -FLT_MAX;
        const float* plane = fbase + c * HW;
        for (int h = hs; h < he; ++h)
          for (int w = ws; w < we; ++w) { const float v = plane[h * W + w]; if (v > m) m = v; }
        obase[i] = m;
      }
      continue;
    }
    // ROIs wider than the LDS column buffer (kMaxSpan columns) are pooled in column segments of kMaxSpan: a bin's value is the
    // maximum over the segments it touches, accumulated in the output itself by the lane that owns the bin (same thread, same
    // address: program order).  (Such ROIs used to take the per-bin loop above: roi_pool_ctx of the 7s-576 frame 270 -> 158 us.)
    const int Wp = Wp_[q];
    const int lanes = min(Wp, 64);              // lanes of one slot; a slot never straddles a wave
    const int xper = Wp / lanes;                // columns per lane (2 when the ROI is 65..128 columns wide)
    const int slots = kThreads / lanes;
    const int slot = tid / lanes, lx = tid % lanes;
    float* col = s_col + slot * (Wp * PH);
    for (int cb = c0; cb < cend; cb += slots)
    for (int seg_lo = x_lo; seg_lo < x_hi; seg_lo += kMaxSpan) {
      const int c = cb + slot;
      const bool live = c < cend;
      const int seg_hi = min(seg_lo + kMaxSpan, x_hi);
      if (live) {
        for (int xi = 0; xi < xper; ++xi) {
          const int xcol = lx + xi * 64;
          const float* p = fbase + c * HW + min(seg_lo + xcol, seg_hi - 1);   // re-reading the last column never changes a max
          for (int ph0 = 0; ph0 < PH; ph0 += 4) {
            float m[4];
            int hs[4], he[4];
            float v[4][kRowsInFlight];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ph = min(ph0 + j, PH - 1);
              hs[j] = s_h0[q][ph]; he[j] = s_h1[q][ph];
#pragma unroll
              for (int i = 0; i < kRowsInFlight; ++i)       // rows past the bin are not fetched at all (lane masked off)
                v[j][i] = (hs[j] + i < he[j]) ? p[(hs[j] + i) * W] : -FLT_MAX;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              m[j] = -FLT_MAX;
#pragma unroll
              for (int i = 0; i < kRowsInFlight; ++i) if (v[j][i] > m[j]) m[j] = v[j][i];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              for (int h = hs[j] + kRowsInFlight; h < he[j]; h += 4) {     // bins taller than kRowsInFlight rows
                const float u0 = p[h * W], u1 = p[min(h + 1, he[j] - 1) * W];
                const float u2 = p[min(h + 2, he[j] - 1) * W], u3 = p[min(h + 3, he[j] - 1) * W];
                if (u0 > m[j]) m[j] = u0;
                if (u1 > m[j]) m[j] = u1;
                if (u2 > m[j]) m[j] = u2;
                if (u3 > m[j]) m[j] = u3;
              }
              if (ph0 + j < PH) col[(ph0 + j) * Wp + xcol] = m[j];
            }
          }
        }
      }
      wave_lds_sync();
      if (live) {
        for (int o = lx; o < bins; o += lanes) {
          const int ph = s_ph[o], pw = s_pw[o];
          const bool empty = (s_h1[q][ph] <= s_h0[q][ph]) || (s_w1[q][pw] <= s_w0[q][pw]);
          const int ws = max(s_w0[q][pw], seg_lo) - seg_lo, we = min(s_w1[q][pw], seg_hi) - seg_lo;     // the bin's columns in this segment
          const float* cr = col + ph * Wp;
          const int last = max(we - 1, 0);
          const float a0 = cr[min(ws, last)], a1 = cr[min(ws + 1, last)], a2 = cr[min(ws + 2, last)], a3 = cr[min(ws + 3, last)];
          float m = -FLT_MAX;
          if (we > ws) {
            if (a0 > m) m = a0;
            if (a1 > m) m = a1;
            if (a2 > m) m = a2;
            if (a3 > m) m = a3;
            for (int x = ws + 4; x < we; ++x) { const float u = cr[x]; if (u > m) m = u; }
          }
          if (seg_lo > x_lo && !empty) { const float prev = obase[c * bins + o]; if (prev > m) m = prev; }
          obase[c * bins + o] = empty ? 0.f : m;
        }
      }
      wave_lds_sync();
    }
  }
}

// ROIAlign: (PH+1) x (PW+1) bilinear samples per (roi, channel) -- roi_align_layer.cu:21-98, same operation order.
__global__ __launch_bounds__(kThreads) void roialign_kernel(const float* __restrict__ feat, const float* __restrict__ rois,
                                                            float* __restrict__ out, int C, int H, int W, int PH, int PW,
                                                            float spatial_scale, float pad_ratio, int chan_per_block) {
  const int r = blockIdx.y;
  const int c_begin = blockIdx.x * chan_per_block;
  const int c_end = min(C, c_begin + chan_per_block);
  const int GH = PH + 1, GW = PW + 1, pts = GH * GW;
  const float* roi = rois + 5 * (size_t)r;
  const int b = (int)roi[0];
  const float pad_w = (roi[3] - roi[1] + 1) * pad_ratio, pad_h = (roi[4] - roi[2] + 1) * pad_ratio;
  float roi_start_w = (roi[1] - pad_w) * spatial_scale, roi_start_h = (roi[2] - pad_h) * spatial_scale;
  float roi_end_w = (roi[3] + pad_w) * spatial_scale, roi_end_h = (roi[4] + pad_h) * spatial_scale;
  roi_start_w -= 0.5f; roi_start_h -= 0.5f; roi_end_w -= 0.5f; roi_end_h -= 0.5f;
  const float roi_height = roi_end_h - roi_start_h, roi_width = roi_end_w - roi_start_w;
  const float bin_size_h = roi_height / (float)PH, bin_size_w = roi_width / (float)PW;
  const float* fbase = feat + (size_t)b * C * H * W;
  float* obase = out + (size_t)r * C * pts;
  const int work = (c_end - c_begin) * pts;
  for (int i = threadIdx.x; i < work; i += kThreads) {
    const int c = c_begin + i / pts, g = i % pts;
    const int ph = g / GW, pw = g % GW;
    float val = 0.f;
    if (!(roi_height <= 0 || roi_width <= 0)) {
      float hfloat = roi_start_h + (float)ph * bin_size_h, wfloat = roi_start_w + (float)pw * bin_size_w;
      if (!(hfloat < -0.5f || hfloat > (H - 0.5f) || wfloat < -0.5f || wfloat > (W - 0.5f))) {
        int hfloor = (int)floorf(hfloat), wfloor = (int)floorf(wfloat);
        int hceil = hfloor + 1, wceil = wfloor + 1;
        hfloat = fminf(fmaxf(hfloat, 0.f), (float)(H - 1)); wfloat = fminf(fmaxf(wfloat, 0.f), (float)(W - 1));
        hfloor = min(max(hfloor, 0), H - 1); wfloor = min(max(wfloor, 0), W - 1);
        hceil = min(max(hceil, 0), H - 1); wceil = min(max(wceil, 0), W - 1);
        const float lh = hfloat - hfloor, lw = wfloat - wfloor, hh = 1 - lh, hw = 1 - lw;
        const float w00 = hw * hh, w10 = lw * hh, w01 = hw * lh, w11 = lw * lh;
        const float* plane = fbase + (size_t)c * H * W;
        const float v00 = plane[hfloor * W + wfloor], v10 = plane[hfloor * W + wceil];
        const float v01 = plane[hceil * W + wfloor], v11 = plane[hceil * W + wceil];
        val = w00 * v00 + w10 * v10 + w01 * v01 + w11 * v11;
      }
    }
    obase[(size_t)c * pts + g] = val;
  }
}

}  // namespace

using namespace mscnn;

extern "C" int mscnn_roialign_fwd_f32(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W,
                                      int pooled_h, int pooled_w, float spatial_scale, float pad_ratio, void* stream) {
  MSCNN_REQUIRE(feat && rois && out, "roialign: null pointer");
  MSCNN_REQUIRE(R >= 0 && N > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0, "roialign: bad shape");
  if (R == 0) return MSCNN_OK;
  const int chan_per_block = (C % 128 == 0) ? 16 : max(1, min(C, 16));
  dim3 grid(cdiv(C, chan_per_block), R);   // channel group on grid.x: XCD-local feature planes, as in ROIPooling
  roialign_kernel<<<grid, kThreads, 0, as_stream(stream)>>>(feat, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale,
                                                            pad_ratio, chan_per_block);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}


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
  int chan_per_block = max(1, (kThreads * 4) / bins);
  if (chan_per_block > C) chan_per_block = C;
  if (C % 128 == 0) chan_per_block = 16;            // C/16 channel groups: a multiple of 8
  dim3 grid(cdiv(C, chan_per_block), R);
  const bool per_bin = tune_env("MSCNN_ROIPOOL_PERBIN", 0) != 0;      // A/B switches: debug builds only (MSCNN_TUNING_ENV)
  const int cpb = tune_env("MSCNN_ROIPOOL_CPB", 0);
  if (cpb > 0) { chan_per_block = cpb; grid.x = cdiv(C, cpb); }
  if (!per_bin && pooled_h <= kMaxP && pooled_w <= kMaxP && (size_t)C * H * W < (1u << 30))
    roipool_rows_kernel<1><<<grid, kThreads, 512 * pooled_h * sizeof(float), as_stream(stream)>>>(
        feat, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale, RoiPoolPass{pad_ratio, c_offset}, RoiPoolPass{0.f, 0}, C_total,
        chan_per_block);
  else
    roipool_kernel<<<grid, kThreads, 0, as_stream(stream)>>>(feat, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale,
                                                             pad_ratio, C_total, c_offset, chan_per_block);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

// Two poolings of the same ROIs over the same map with two context paddings, into two channel windows of one output: one launch
// (see roipool_rows_kernel<2>).  Shapes the row kernel does not take run as two single launches.
extern "C" int mscnn_roipool_pair_fwd_f32(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W,
                                          int pooled_h, int pooled_w, float spatial_scale, float pad_ratio_a, int c_offset_a,
                                          float pad_ratio_b, int c_offset_b, int C_total, void* stream) {
  MSCNN_REQUIRE(feat && rois && out, "roipool: null pointer");
  MSCNN_REQUIRE(R >= 0 && N > 0 && C > 0 && H > 0 && W > 0, "roipool: bad shape");
  MSCNN_REQUIRE(pooled_h > 0 && pooled_w > 0, "roipool: pooled_h/pooled_w must be > 0");
  MSCNN_REQUIRE(c_offset_a >= 0 && c_offset_a + C <= C_total && c_offset_b >= 0 && c_offset_b + C <= C_total &&
                    (c_offset_a + C <= c_offset_b || c_offset_b + C <= c_offset_a),
                "roipool pair: channel windows outside the output buffer or overlapping");
  if (R == 0) return MSCNN_OK;
  if (!(pooled_h <= kMaxP && pooled_w <= kMaxP && (size_t)C * H * W < (1u << 30))) {
    const int rc = mscnn_roipool_fwd_f32(feat, rois, out, R, N, C, H, W, pooled_h, pooled_w, spatial_scale, pad_ratio_a, C_total, c_offset_a, stream);
    if (rc != MSCNN_OK) return rc;
    return mscnn_roipool_fwd_f32(feat, rois, out, R, N, C, H, W, pooled_h, pooled_w, spatial_scale, pad_ratio_b, C_total, c_offset_b, stream);
  }
  const int bins = pooled_h * pooled_w;
  int chan_per_block = max(1, (kThreads * 4) / bins);
  if (chan_per_block > C) chan_per_block = C;
  if (C % 128 == 0) chan_per_block = 16;            // C/16 channel groups: a multiple of 8 (XCD-aware, as above)
  dim3 grid(cdiv(C, chan_per_block), R);
  // pass 1 runs first in the kernel: the wider window (larger padding)
  const bool a_wider = pad_ratio_a >= pad_ratio_b;
  const RoiPoolPass narrow{a_wider ? pad_ratio_b : pad_ratio_a, a_wider ? c_offset_b : c_offset_a};
  const RoiPoolPass wide{a_wider ? pad_ratio_a : pad_ratio_b, a_wider ? c_offset_a : c_offset_b};
  roipool_rows_kernel<2><<<grid, kThreads, 512 * pooled_h * sizeof(float), as_stream(stream)>>>(
      feat, rois, out, C, H, W, pooled_h, pooled_w, spatial_scale, narrow, wide, C_total, chan_per_block);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
