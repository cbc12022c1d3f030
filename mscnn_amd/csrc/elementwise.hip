// HBM-bound stock layers of the MS-CNN path for gfx950: ReLU, MAX/AVE pooling, channel concat,
// depthwise transposed convolution (bilinear 2x upsample), channel softmax.
// All of them are bandwidth kernels: one pass, coalesced (16 B per lane where the shape allows),
// grid sized to >= a few workgroups per CU with a grid-stride loop.
#include "common.h"
#include "up2_device.h"
#include <cfloat>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 256 * 8;   // 256 CUs x 8 workgroups

inline int grid_for(long work_items) {
  long b = (work_items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// ---- ReLU: relu_layer.cu:9-14  y = x > 0 ? x : x * slope --------------------------------------
__global__ __launch_bounds__(kThreads) void relu_kernel_v4(const float4* __restrict__ x, float4* __restrict__ y,
                                                           size_t n4, float slope) {
  for (size_t i = blockIdx.x * (size_t)kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
    float4 v = x[i];
    v.x = v.x > 0.f ? v.x : v.x * slope;
    v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope;
    v.w = v.w > 0.f ? v.w : v.w * slope;
    y[i] = v;
  }
}
__global__ __launch_bounds__(kThreads) void relu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        size_t n, float slope) {
  for (size_t i = blockIdx.x * (size_t)kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const float v = x[i];
    y[i] = v > 0.f ? v : v * slope;
  }
}

// ---- Pooling ------------------------------------------------------------------------------------
// Fast path: 2x2 stride 2 no pad, even W: each lane produces two horizontally adjacent outputs from
// two float4 reads (rows h, h+1), i.e. 32 B in, 8 B out per lane, fully coalesced.
// First-max-wins with strict '>' and init -FLT_MAX exactly as pooling_layer.cu:26-35.
__device__ __forceinline__ float max4_first(float a, float b, float c, float d) {
  float m = -FLT_MAX;
  if (a > m) m = a;
  if (b > m) m = b;
  if (c > m) m = c;
  if (d > m) m = d;
  return m;
}

__global__ __launch_bounds__(kThreads) void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              long planes, int H, int W, int Ho, int Wo) {
  // work item = (plane, ho, wo/2)
  const int wo2 = Wo >> 1;
  const long total = planes * Ho * wo2;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int w2 = (int)(i % wo2);
    const long r = i / wo2;
    const int ho = (int)(r % Ho);
    const long p = r / Ho;
    const float* src = x + (p * H + 2 * ho) * (long)W + 4 * w2;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + W);
    float2 o;
    o.x = max4_first(a.x, a.y, b.x, b.y);
    o.y = max4_first(a.z, a.w, b.z, b.w);
    *reinterpret_cast<float2*>(y + (p * Ho + ho) * (long)Wo + 2 * w2) = o;
  }
}

// General path (any kernel/pad/stride, ceil-mode edges): one output per lane.
__global__ __launch_bounds__(kThreads) void pool_general_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                long planes, int H, int W, int Ho, int Wo, int kh, int kw,
                                                                int ph, int pw, int sh, int sw, int method) {
  const long total = planes * Ho * Wo;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int wo = (int)(i % Wo);
    const long r = i / Wo;
    const int ho = (int)(r % Ho);
    const long p = r / Ho;
    const float* src = x + p * (long)H * W;
    int hs = ho * sh - ph, ws = wo * sw - pw;
    if (method == 0) {
      const int he = min(hs + kh, H), we = min(ws + kw, W);
      hs = max(hs, 0); ws = max(ws, 0);
      float m = -FLT_MAX;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) {
          const float v = src[h * W + w];
          if (v > m) m = v;
        }
      y[i] = m;
    } else {  // AVE: pooling_layer.cu:50-81 (pool_size counts the padded window)
      int he = min(hs + kh, H + ph), we = min(ws + kw, W + pw);
      const int pool_size = (he - hs) * (we - ws);
      hs = max(hs, 0); ws = max(ws, 0); he = min(he, H); we = min(we, W);
      float acc = 0.f;
      for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) acc += src[h * W + w];
      y[i] = acc / pool_size;
    }
  }
}

// ---- Concat: concat_layer.cu:9-25 ---------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void concat_kernel(const float* __restrict__ x, float* __restrict__ y, long per_n,
                                                          long total, long y_per_n, long y_off) {
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const long n = i / per_n, r = i - n * per_n;
    y[n * y_per_n + y_off + r] = x[i];
  }
}

// ---- Depthwise transposed convolution (gather form) ---------------------------------------------
// Reference: Deconvolution with group == channels is 512 K=1 GEMMs + col2im (deconv_layer.cu:8-23,
// im2col.cu:247-285).  col2im sums, for an output pixel, the contributing (ky,kx) taps in ascending
// (ky, kx) order (im2col.cu:262-278 loops h_col, w_col ascending which is ky descending...); we use
// the oracle's order: ky ascending, kx ascending.  <= 4 taps contribute for 4x4 s2.
__global__ __launch_bounds__(kThreads) void deconv_dw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int N,
                                                             int C, int H, int W, int Ho, int Wo, int Kh, int Kw, int ph,
                                                             int pw, int sh, int sw) {
  const long total = (long)N * C * Ho * Wo;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int ox = (int)(i % Wo);
    long r = i / Wo;
    const int oy = (int)(r % Ho);
    r /= Ho;
    const int c = (int)(r % C);
    const float* src = x + r * (long)H * W;
    const float* wk = w + (long)c * Kh * Kw;
    float acc = 0.f;
    for (int ky = 0; ky < Kh; ++ky) {
      const int ty = oy + ph - ky;
      if (ty < 0 || ty % sh) continue;
      const int iy = ty / sh;
      if (iy >= H) continue;
      for (int kx = 0; kx < Kw; ++kx) {
        const int tx = ox + pw - kx;
        if (tx < 0 || tx % sw) continue;
        const int ix = tx / sw;
        if (ix >= W) continue;
        acc += wk[ky * Kw + kx] * src[iy * W + ix];
      }
    }
    if (bias) acc += bias[c];
    y[i] = acc;
  }
}

// The 4x4 / stride 2 / pad 1 case (every "-2x" deploy net): one thread per 2x2 output quad = 9 coalesced input loads, two float2
// stores; 35 MB in, 141 MB out for conv4_3 of the 576 x 1920 frame (the per-output kernel above took 469 us for it).
__global__ __launch_bounds__(kThreads) void deconv_dw_up2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int C, int H,
                                                                 int W) {
  const int b = blockIdx.x * kThreads + threadIdx.x, a = blockIdx.y, nc = blockIdx.z;
  if (b >= W) return;
  const float* src = x + (long)nc * H * W;
  const float* w16 = w + (long)(nc % C) * 16;
  float v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int r = a - 1 + i, c = b - 1 + j;
      v[i][j] = (r >= 0 && r < H && c >= 0 && c < W) ? src[r * W + c] : 0.f;
    }
  const float bv = bias ? bias[nc % C] : 0.f;
  float* dst = y + ((long)nc * 2 * H + 2 * a) * (2 * W) + 2 * b;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    // rows: py 0 -> A = a (v[1]), B = a - 1 (v[0]);  py 1 -> A = a + 1 (v[2]), B = a (v[1]);  columns likewise
    const float* rA = py ? v[2] : v[1];
    const float* rB = py ? v[1] : v[0];
    float o0 = mscnn::up2_value(w16, py, 0, rA[1], rA[0], rB[1], rB[0]);
    float o1 = mscnn::up2_value(w16, py, 1, rA[2], rA[1], rB[2], rB[1]);
    if (bias) { o0 += bv; o1 += bv; }
    *reinterpret_cast<float2*>(dst + (long)py * 2 * W) = make_float2(o0, o1);
  }
}

// ---- Softmax over channels (C small: 2..5 in the deploy nets) -------------------------------------
__global__ __launch_bounds__(kThreads) void softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int outer,
                                                           int C, int inner) {
  const long total = (long)outer * inner;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const long o = i / inner, k = i - o * inner;
    const float* xp = x + o * (long)C * inner + k;
    float* yp = y + o * (long)C * inner + k;
    float mx = xp[0];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, xp[(long)j * inner]);
    float sum = 0.f;
    for (int j = 0; j < C; ++j) {
      const float e = expf(xp[(long)j * inner] - mx);
      yp[(long)j * inner] = e;
      sum += e;
    }
    for (int j = 0; j < C; ++j) yp[(long)j * inner] /= sum;
  }
}

// ---- Eltwise: eltwise_layer.cu (PROD / SUM with coefficients / MAX), up to 8 bottoms ----------------------------
struct EltArgs { const float* x[8]; float coeff[8]; int nb, op; };
__global__ __launch_bounds__(kThreads) void eltwise_kernel(EltArgs a, float* __restrict__ y, long count) {
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < count; i += (long)gridDim.x * kThreads) {
    float v;
    if (a.op == 0) {
      v = a.x[0][i] * a.x[1][i];
      for (int b = 2; b < a.nb; ++b) v = v * a.x[b][i];
    } else if (a.op == 1) {
      v = 0.f;
      for (int b = 0; b < a.nb; ++b) v = a.coeff[b] * a.x[b][i] + v;     // caffe_axpy per bottom, in order
    } else {
      v = a.x[0][i] > a.x[1][i] ? a.x[0][i] : a.x[1][i];
      for (int b = 2; b < a.nb; ++b) if (a.x[b][i] > v) v = a.x[b][i];
    }
    y[i] = v;
  }
}

// ---- generic transposed convolution (any group / stride / pad), one output element per thread: gathers the taps
// (kh, kw) whose input position is integral, sums the group's input channels per tap (deconv_layer.cpp:8-40 computes
// col = W^T x per image and scatters it with col2im; same sum, different order -> 1e-4 class like every conv).
__global__ __launch_bounds__(kThreads) void deconv_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int N,
                                                                  int Cin, int H, int W, int Cout, int Ho, int Wo, int Kh, int Kw,
                                                                  int ph, int pw, int sh, int sw, int group) {
  const long total = (long)N * Cout * Ho * Wo;
  const int cig = Cin / group, cog = Cout / group;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int ow = (int)(i % Wo);
    long r = i / Wo;
    const int oh = (int)(r % Ho); r /= Ho;
    const int co = (int)(r % Cout);
    const int n = (int)(r / Cout);
    const int g = co / cog, col = co % cog;
    float acc = 0.f;
    for (int kh = 0; kh < Kh; ++kh) {
      const int th = oh + ph - kh;
      if (th < 0 || th % sh != 0 || th / sh >= H) continue;
      for (int kw = 0; kw < Kw; ++kw) {
        const int tw = ow + pw - kw;
        if (tw < 0 || tw % sw != 0 || tw / sw >= W) continue;
        const float* xp = x + (((long)n * Cin + g * cig) * H + th / sh) * W + tw / sw;
        const float* wp = w + (((long)(g * cig) * cog + col) * Kh + kh) * Kw + kw;      // w[Cin][Cout/g][Kh][Kw]
        for (int c = 0; c < cig; ++c) acc += xp[(long)c * H * W] * wp[(long)c * cog * Kh * Kw];
      }
    }
    if (bias) acc += bias[co];
    y[i] = acc;
  }
}

// ---- max_i |a_i - ref_i| / max(floor, |ref_i|): the parity metric of the tests, on the device (Net::CalibrateNumerics).
// Non-negative floats order like their bit patterns, so the reduction is an integer atomicMax; NaN anywhere -> +inf.
__global__ __launch_bounds__(kThreads) void max_rel_diff_kernel(const float* __restrict__ a, const float* __restrict__ ref, long n,
                                                                float floor_, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const float r = ref[i], d = fabsf(a[i] - r) / fmaxf(floor_, fabsf(r));
    m = (d != d) ? __builtin_inff() : fmaxf(m, d);
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// the same over `planes` runs of `run` consecutive floats (a band of rows of an NCHW blob against a band computed elsewhere), the floor
// max(1, rms) taken from a sum of squares that is still on the device (no host round trip between the two reductions)
__global__ __launch_bounds__(kThreads) void max_rel_diff_strided_kernel(const float* __restrict__ a, long a_stride, const float* __restrict__ ref,
                                                                        long ref_stride, long planes, long run, const double* __restrict__ sumsq,
                                                                        double sumsq_count, unsigned* __restrict__ out) {
  const float floor_ = fmaxf(1.f, (float)sqrt(sumsq[0] / sumsq_count));
  float m = 0.f;
  const long n = planes * run;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const long p = i / run, j = i - p * run;
    const float r = ref[p * ref_stride + j], d = fabsf(a[p * a_stride + j] - r) / fmaxf(floor_, fabsf(r));
    m = (d != d) ? __builtin_inff() : fmaxf(m, d);
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// up to four 32-bit words in ONE launch (a pack header {R, cap, 0}: three 4-byte memsets were three blit launches)
__global__ void store_words_kernel(int* __restrict__ dst, int4 v, int n) {
  const int t = threadIdx.x;
  if (t < n) dst[t] = t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w;
}

// sum of squares in double (the scale of a blob for the calibration metric): per-wave partial sums, one f64 atomicAdd per wave
__global__ __launch_bounds__(kThreads) void sum_squares_kernel(const float* __restrict__ x, long n, double* __restrict__ out) {
  double s = 0.0;
  for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) s += (double)x[i] * (double)x[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

}  // namespace

using namespace mscnn;

extern "C" int mscnn_sum_squares_f32(const float* x, size_t count, double* out_dev, void* stream) {
  MSCNN_REQUIRE(out_dev && (count == 0 || x), "sum_squares: bad argument");
  MSCNN_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(double), as_stream(stream)));
  if (count == 0) return MSCNN_OK;
  sum_squares_kernel<<<grid_for((long)count), kThreads, 0, as_stream(stream)>>>(x, (long)count, out_dev);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_max_rel_diff_f32(const float* a, const float* ref, size_t count, float floor_, float* out_dev, void* stream) {
  MSCNN_REQUIRE(out_dev && (count == 0 || (a && ref)) && floor_ > 0.f, "max_rel_diff: bad argument");
  MSCNN_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), as_stream(stream)));
  if (count == 0) return MSCNN_OK;
  max_rel_diff_kernel<<<grid_for((long)count), kThreads, 0, as_stream(stream)>>>(a, ref, (long)count, floor_,
                                                                                 reinterpret_cast<unsigned*>(out_dev));
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_store_words_i32(int* dst_dev, const int* values_host, int n, void* stream) {
  MSCNN_REQUIRE(dst_dev && values_host && n >= 1 && n <= 4, "store_words: 1 .. 4 words");
  int4 v = make_int4(values_host[0], n > 1 ? values_host[1] : 0, n > 2 ? values_host[2] : 0, n > 3 ? values_host[3] : 0);
  store_words_kernel<<<1, 64, 0, as_stream(stream)>>>(dst_dev, v, n);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_max_rel_diff_strided_f32(const float* a, size_t a_stride, const float* ref, size_t ref_stride, size_t planes, size_t run,
                                              const double* sumsq_dev, double sumsq_count, float* out_dev, void* stream) {
  MSCNN_REQUIRE(out_dev && sumsq_dev && sumsq_count > 0 && (planes * run == 0 || (a && ref)), "max_rel_diff_strided: bad argument");
  MSCNN_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), as_stream(stream)));
  if (planes * run == 0) return MSCNN_OK;
  max_rel_diff_strided_kernel<<<grid_for((long)(planes * run)), kThreads, 0, as_stream(stream)>>>(
      a, (long)a_stride, ref, (long)ref_stride, (long)planes, (long)run, sumsq_dev, sumsq_count, reinterpret_cast<unsigned*>(out_dev));
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_deconv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                                      int Cout, int Kh, int Kw, int pad_h, int pad_w, int stride_h, int stride_w, int group,
                                      void* stream) {
  MSCNN_REQUIRE(x && w && y, "deconv: null pointer");
  MSCNN_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Kh > 0 && Kw > 0 && stride_h > 0 && stride_w > 0 && group > 0,
                "deconv: bad shape");
  MSCNN_REQUIRE(Cin % group == 0 && Cout % group == 0, "deconv: channels not divisible by group");
  if (group == Cin && Cout == Cin)      // depthwise (the bilinear 2x up-sampling of the "-2x" nets): its own kernel
    return mscnn_deconv_depthwise_fwd_f32(x, w, bias, y, N, Cin, H, W, Kh, Kw, pad_h, pad_w, stride_h, stride_w, stream);
  const int Ho = stride_h * (H - 1) + Kh - 2 * pad_h, Wo = stride_w * (W - 1) + Kw - 2 * pad_w;
  MSCNN_REQUIRE(Ho > 0 && Wo > 0, "deconv: empty output");
  deconv_generic_kernel<<<grid_for((long)N * Cout * Ho * Wo), kThreads, 0, as_stream(stream)>>>(
      x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, Kh, Kw, pad_h, pad_w, stride_h, stride_w, group);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_eltwise_fwd_f32(const float* const* bottoms_host, int num_bottoms, const float* coeffs_host, float* y,
                                     size_t count, int op, void* stream) {
  MSCNN_REQUIRE(bottoms_host && y && num_bottoms >= 2 && num_bottoms <= 8, "eltwise: 2..8 bottoms");
  MSCNN_REQUIRE(op >= 0 && op <= 2, "eltwise: op must be 0 PROD, 1 SUM, 2 MAX");
  if (count == 0) return MSCNN_OK;
  EltArgs a;
  a.nb = num_bottoms; a.op = op;
  for (int b = 0; b < num_bottoms; ++b) {
    MSCNN_REQUIRE(bottoms_host[b], "eltwise: null bottom");
    a.x[b] = bottoms_host[b];
    a.coeff[b] = coeffs_host ? coeffs_host[b] : 1.f;
  }
  eltwise_kernel<<<grid_for((long)count), kThreads, 0, as_stream(stream)>>>(a, y, (long)count);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}


extern "C" int mscnn_relu_fwd_f32(const float* x, float* y, size_t count, float negative_slope, void* stream) {
  if (count == 0) return MSCNN_OK;
  MSCNN_REQUIRE(x && y, "relu: null pointer");
  const bool vec = (count % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0);
  if (vec) {
    const size_t n4 = count / 4;
    relu_kernel_v4<<<grid_for((long)n4), kThreads, 0, as_stream(stream)>>>(reinterpret_cast<const float4*>(x),
                                                                            reinterpret_cast<float4*>(y), n4, negative_slope);
  } else {
    relu_kernel<<<grid_for((long)count), kThreads, 0, as_stream(stream)>>>(x, y, count, negative_slope);
  }
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_pool_out_dim(int in, int kernel, int pad, int stride) {
  // pooling_layer.cpp:90-107
  int o = (int)ceilf((float)(in + 2 * pad - kernel) / stride) + 1;
  if (pad && (o - 1) * stride >= in + pad) --o;
  return o;
}

extern "C" int mscnn_pool2d_fwd_f32(const float* x, float* y, int N, int C, int H, int W, int kernel_h, int kernel_w,
                                    int pad_h, int pad_w, int stride_h, int stride_w, int method, void* stream) {
  MSCNN_REQUIRE(x && y, "pool: null pointer");
  MSCNN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && kernel_h > 0 && kernel_w > 0 && stride_h > 0 && stride_w > 0,
                "pool: bad shape");
  MSCNN_REQUIRE(method == 0 || method == 1, "pool: method %d not supported (0 MAX, 1 AVE)", method);
  MSCNN_REQUIRE(pad_h < kernel_h && pad_w < kernel_w, "pool: pad must be smaller than kernel");
  const int Ho = mscnn_pool_out_dim(H, kernel_h, pad_h, stride_h), Wo = mscnn_pool_out_dim(W, kernel_w, pad_w, stride_w);
  const long planes = (long)N * C;
  const bool fast = method == 0 && kernel_h == 2 && kernel_w == 2 && stride_h == 2 && stride_w == 2 && pad_h == 0 &&
                    pad_w == 0 && (H % 2 == 0) && (W % 4 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                    (reinterpret_cast<uintptr_t>(y) % 8 == 0);
  if (fast) {
    maxpool2x2_kernel<<<grid_for(planes * Ho * (Wo / 2)), kThreads, 0, as_stream(stream)>>>(x, y, planes, H, W, Ho, Wo);
  } else {
    pool_general_kernel<<<grid_for(planes * Ho * Wo), kThreads, 0, as_stream(stream)>>>(
        x, y, planes, H, W, Ho, Wo, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, method);
  }
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_concat_channels_f32(const float* x, float* y, int N, int C, int inner, int C_total, int c_offset,
                                         void* stream) {
  MSCNN_REQUIRE(x && y, "concat: null pointer");
  MSCNN_REQUIRE(N >= 0 && C > 0 && inner > 0 && c_offset >= 0 && c_offset + C <= C_total, "concat: bad shape");
  const long per_n = (long)C * inner, total = per_n * N;
  if (total == 0) return MSCNN_OK;
  concat_kernel<<<grid_for(total), kThreads, 0, as_stream(stream)>>>(x, y, per_n, total, (long)C_total * inner,
                                                                      (long)c_offset * inner);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_deconv_depthwise_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int C,
                                              int H, int W, int Kh, int Kw, int pad_h, int pad_w, int stride_h,
                                              int stride_w, void* stream) {
  MSCNN_REQUIRE(x && w && y, "deconv: null pointer");
  MSCNN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Kh > 0 && Kw > 0 && stride_h > 0 && stride_w > 0, "deconv: bad shape");
  const int Ho = stride_h * (H - 1) + Kh - 2 * pad_h, Wo = stride_w * (W - 1) + Kw - 2 * pad_w;
  MSCNN_REQUIRE(Ho > 0 && Wo > 0, "deconv: empty output");
  if (Kh == 4 && Kw == 4 && stride_h == 2 && stride_w == 2 && pad_h == 1 && pad_w == 1 && (long)N * C <= 65535 && H <= 65535) {
    deconv_dw_up2_kernel<<<dim3((W + kThreads - 1) / kThreads, H, N * C), kThreads, 0, as_stream(stream)>>>(x, w, bias, y, C, H, W);
    MSCNN_POST_LAUNCH();
    return MSCNN_OK;
  }
  deconv_dw_kernel<<<grid_for((long)N * C * Ho * Wo), kThreads, 0, as_stream(stream)>>>(
      x, w, bias, y, N, C, H, W, Ho, Wo, Kh, Kw, pad_h, pad_w, stride_h, stride_w);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_softmax_fwd_f32(const float* x, float* y, int outer, int C, int inner, void* stream) {
  MSCNN_REQUIRE(x && y && outer >= 0 && C > 0 && inner > 0, "softmax: bad argument");
  if (outer == 0) return MSCNN_OK;
  softmax_kernel<<<grid_for((long)outer * inner), kThreads, 0, as_stream(stream)>>>(x, y, outer, C, inner);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
