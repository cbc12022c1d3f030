// Image pre-processing in front of net.forward, on the device.
// Replaces the MATLAB lines examples/kitti_car/run_mscnn_detection.m:64-69:
//     test_image = imresize(test_image,[imgH imgW]); test_image = single(test_image(:,:,[3 2 1]));
//     test_image = bsxfun(@minus,test_image,mu);    test_image = permute(test_image, [2 1 3]);
// imresize is restated from its published algorithm (bicubic a = -0.5, kernel stretched by 1/scale when shrinking,
// P = ceil(width) + 2 taps, normalised weights, symmetric border mirroring, smaller-scale dimension first, uint8 rounding
// after EACH 1-D pass) -- MATLAB itself is not available, so this stage is "parity unpinned" like the final detection stage;
// it is bit-identical to oracle/pyoracle.py::preprocess (same double-precision operation order; -ffp-contract=off).
// Two HBM-bound kernels: pass 1 resizes along the first dimension into a uint8 scratch image, pass 2 resizes along the
// other one and fuses the RGB->BGR swap, the mean subtraction and the HWC -> NCHW layout change.
#include "common.h"
#include <cstdint>

namespace {

__device__ __forceinline__ double cubic(double x) {
  const double ax = fabs(x), ax2 = ax * ax, ax3 = ax2 * ax;
  if (ax <= 1.0) return (1.5 * ax3 - 2.5 * ax2) + 1.0;
  if (ax <= 2.0) return ((-0.5 * ax3 + 2.5 * ax2) - 4.0 * ax) + 2.0;
  return 0.0;
}

// 1-D resize of output index `o` (0-based): calls f(k, weight, source index) for the P taps in order
struct Taps {
  double scale, kw, u, left, sum;
  int P, in_len;
  __device__ Taps(int in_len_, int out_len, int o) : in_len(in_len_) {
    scale = (double)out_len / (double)in_len_;
    kw = scale < 1.0 ? 4.0 / scale : 4.0;
    const double x = (double)(o + 1);
    u = x / scale + 0.5 * (1.0 - 1.0 / scale);
    left = floor(u - kw / 2.0);
    P = (int)ceil(kw) + 2;
    sum = 0.0;
    for (int k = 0; k < P; ++k) sum = sum + raw(k);
  }
  __device__ double raw(int k) const {
    const double d = u - (left + (double)k);
    return scale < 1.0 ? scale * cubic(scale * d) : cubic(d);
  }
  __device__ int index(int k) const {            // symmetric mirroring: aux = [1..n, n..1]
    const long period = 2L * in_len;
    long m = ((long)left + k - 1) % period;
    if (m < 0) m += period;
    return (int)(m < in_len ? m : period - 1 - m);
  }
};

__device__ __forceinline__ unsigned char to_u8(double v) {
  v = floor(v + 0.5);
  return (unsigned char)(v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));
}

// src [sh][sw][3] u8 -> dst [dh][sw][3] u8 (resize along H) or [sh][dw][3] (along W)
__global__ __launch_bounds__(256) void resize_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                        int sh, int sw, int dh, int dw, int along_w) {
  const long total = (long)dh * dw * 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 3), x = (int)((i / 3) % dw), y = (int)(i / (3L * dw));
    Taps t(along_w ? sw : sh, along_w ? dw : dh, along_w ? x : y);
    double acc = 0.0;
    for (int k = 0; k < t.P; ++k) {
      const int s = t.index(k);
      const double v = along_w ? (double)src[((long)y * sw + s) * 3 + c] : (double)src[((long)s * sw + x) * 3 + c];
      acc = acc + (t.raw(k) / t.sum) * v;
    }
    dst[i] = to_u8(acc);
  }
}

// last pass fused with BGR swap + mean subtraction + layout: src [sh][sw][3] u8 RGB -> out [3][dh][dw] f32 (BGR planes)
__global__ __launch_bounds__(256) void resize_finish_kernel(const unsigned char* __restrict__ src, float* __restrict__ out, int sh,
                                                            int sw, int dh, int dw, int along_w, float mb, float mg, float mr) {
  const long total = (long)dh * dw;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % dw), y = (int)(i / dw);
    Taps t(along_w ? sw : sh, along_w ? dw : dh, along_w ? x : y);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < t.P; ++k) {
      const int s = t.index(k);
      const unsigned char* p = along_w ? src + ((long)y * sw + s) * 3 : src + ((long)s * sw + x) * 3;
      const double w = t.raw(k) / t.sum;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = acc[c] + w * (double)p[c];
    }
    out[i] = (float)to_u8(acc[2]) - mb;                 // plane 0 = B
    out[total + i] = (float)to_u8(acc[1]) - mg;         // plane 1 = G
    out[2 * total + i] = (float)to_u8(acc[0]) - mr;     // plane 2 = R
  }
}

}  // namespace

using namespace mscnn;

extern "C" size_t mscnn_preprocess_workspace_bytes(int org_h, int org_w, int H, int W) {
  // the intermediate image after the first 1-D pass
  const double sh = (double)H / org_h, sw = (double)W / org_w;
  return sh <= sw ? (size_t)H * org_w * 3 : (size_t)org_h * W * 3;
}

extern "C" int mscnn_preprocess_u8_f32(const unsigned char* img_rgb, int org_h, int org_w, float* out, int H, int W,
                                       const float* mean_bgr, void* workspace, size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(img_rgb && out && mean_bgr, "preprocess: null pointer");
  MSCNN_REQUIRE(org_h > 0 && org_w > 0 && H > 0 && W > 0, "preprocess: bad shape");
  MSCNN_REQUIRE(workspace && workspace_bytes >= mscnn_preprocess_workspace_bytes(org_h, org_w, H, W), "preprocess: workspace too small");
  hipStream_t st = as_stream(stream);
  const double sh = (double)H / org_h, sw = (double)W / org_w;
  unsigned char* tmp = static_cast<unsigned char*>(workspace);
  const bool h_first = sh <= sw;                           // [~, order] = sort(scale)
  const int mh = h_first ? H : org_h, mw = h_first ? org_w : W;
  long blocks = ((long)mh * mw * 3 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  resize_u8_kernel<<<(int)blocks, 256, 0, st>>>(img_rgb, tmp, org_h, org_w, mh, mw, h_first ? 0 : 1);
  MSCNN_POST_LAUNCH();
  blocks = ((long)H * W + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  resize_finish_kernel<<<(int)blocks, 256, 0, st>>>(tmp, out, mh, mw, H, W, h_first ? 1 : 0, mean_bgr[0], mean_bgr[1], mean_bgr[2]);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
