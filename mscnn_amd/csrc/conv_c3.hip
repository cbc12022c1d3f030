// conv1_1 of the VGG trunk for gfx950: 3 input channels, K = 27.
// Replaces ConvolutionLayer<Dtype>::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23: im2col + SGEMM) for the one layer whose work
// is its OUTPUT: 1 x 64 x 576 x 1920 floats = 283 MB written against 13 MB read and 3.8 GFLOP -- an HBM-store-bound layer.  On the
// trunk's MFMA igemm kernel (K padded to 32: one chunk, so every tile is all prologue and epilogue) it took 102 us = 2.9 TB/s.
// Here no matrix core and no LDS staging of the input: a thread owns 4 consecutive pixels of one row, holds their 3 x 3 x 6 input
// window in registers (27 loads, the tile's rows come from L1 / L2), and walks the output channels: per channel the 27 weights are
// broadcast reads of LDS (the whole filter bank, 6.9 KB, is staged once per workgroup), 108 FMAs, bias, ReLU, one 16-byte store --
// a wave writes two 512-byte runs of the output per channel.  1.9 G FMAs = 27 us of VALU time on the chip, under the ~57 us the
// stores need.
// Summation order: taps outer, input channels inner, accumulator starts at 0, bias added last -- the k order of the igemm kernel
// it replaces (whose fp32 MFMA is an fmaf chain in k), so the two are bit-identical (test_conv_c3_bit_identical_to_the_igemm_kernel).
#include "conv_c3.h"
#include "x3_device.h"

namespace {

constexpr int kPx = 4;                 // pixels per thread (one float4)
constexpr int kTileW = 32 * kPx;       // 128 columns: lanes 0..31 of a wave
constexpr int kTileH = 8;              // 4 waves x 2 rows (lanes 32..63: the wave's second row)
constexpr int kMaxCout = 128;
constexpr int kWStride = 28;           // 27 weights padded to a multiple of 4 floats (16-byte LDS reads)

struct C3Args {
  const float* x; const float* w; const float* bias; float* y;
  unsigned* amax_out;      // max |y| slots of the split-fp16 hand-over (x3_device.h), nullptr = off
  int N, H, W, Cout, relu, tiles_w, tiles_h;
};

__global__ __launch_bounds__(256) void conv3x3_c3_kernel(C3Args a) {
  __shared__ __attribute__((aligned(16))) float sw[kMaxCout * kWStride];
  __shared__ float sb[kMaxCout];
  const int tid = threadIdx.x;
  // stage the filter bank: sw[co][tap * 3 + ci] (taps outer, channels inner: the order of the sum), Caffe layout is [co][ci][tap]
  for (int i = tid; i < a.Cout * kWStride; i += 256) {
    const int co = i / kWStride, k = i % kWStride;
    const int tap = k / 3, ci = k % 3;
    sw[i] = k < 27 ? a.w[co * 27 + ci * 9 + tap] : 0.f;
  }
  for (int i = tid; i < a.Cout; i += 256) sb[i] = a.bias ? a.bias[i] : 0.f;
  __syncthreads();

  const int t = blockIdx.x;
  const int tw = t % a.tiles_w, th = (t / a.tiles_w) % a.tiles_h, n = t / (a.tiles_w * a.tiles_h);
  const int lane = tid & 63, wave = tid >> 6;
  const int yrow = th * kTileH + wave * 2 + (lane >> 5);
  const int x0 = tw * kTileW + (lane & 31) * kPx;
  unsigned am = 0;                           // bit pattern of this thread's max |y| (published per workgroup when asked for)
  if (yrow < a.H && x0 < a.W) {              // (W % 4 == 0: a thread's four pixels are all inside or all outside)

  // the 3 x 3 x 6 window: rows yrow-1 .. yrow+1, columns x0-1 .. x0+4, zero outside the image
  const long plane = (long)a.H * a.W;
  const float* xin = a.x + (long)n * 3 * plane;
  float in[3][3][6];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = yrow - 1 + r;
      const bool rok = yy >= 0 && yy < a.H;
      const float* row = xin + ci * plane + (long)(rok ? yy : 0) * a.W;
      const float4 mid = rok ? *reinterpret_cast<const float4*>(row + x0) : make_float4(0.f, 0.f, 0.f, 0.f);
      in[ci][r][0] = (rok && x0 > 0) ? row[x0 - 1] : 0.f;
      in[ci][r][1] = mid.x; in[ci][r][2] = mid.y; in[ci][r][3] = mid.z; in[ci][r][4] = mid.w;
      in[ci][r][5] = (rok && x0 + 4 < a.W) ? row[x0 + 4] : 0.f;
    }

  float* yout = a.y + (long)n * a.Cout * plane + (long)yrow * a.W + x0;
#pragma unroll 2
  for (int co = 0; co < a.Cout; ++co) {
    const float4* wv = reinterpret_cast<const float4*>(sw + co * kWStride);
    float wk[kWStride];
#pragma unroll
    for (int q = 0; q < kWStride / 4; ++q) { const float4 v = wv[q]; wk[4 * q] = v.x; wk[4 * q + 1] = v.y; wk[4 * q + 2] = v.z; wk[4 * q + 3] = v.w; }
    float acc[kPx] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          const float wgt = wk[(kh * 3 + kw) * 3 + ci];
#pragma unroll
          for (int p = 0; p < kPx; ++p) acc[p] = fmaf(wgt, in[ci][kh][kw + p], acc[p]);
        }
    const float b = sb[co];
    float4 o = make_float4(acc[0] + b, acc[1] + b, acc[2] + b, acc[3] + b);
    if (a.relu) { o.x = o.x < 0.f ? 0.f : o.x; o.y = o.y < 0.f ? 0.f : o.y; o.z = o.z < 0.f ? 0.f : o.z; o.w = o.w < 0.f ? 0.f : o.w; }
    *reinterpret_cast<float4*>(yout + (long)co * plane) = o;
    if (a.amax_out) am = max(am, __float_as_uint(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)))));
  }
  }
  if (a.amax_out) mscnn::publish_amax(am, a.amax_out, blockIdx.x);      // (every thread: it contains a barrier)
}

}  // namespace

namespace mscnn {

bool c3_plan(const mscnn_conv_desc& d, int Ho, int Wo) {
  return d.Cin == 3 && d.Kh == 3 && d.Kw == 3 && d.stride_h == 1 && d.stride_w == 1 && d.pad_h == 1 && d.pad_w == 1 && d.group == 1 &&
         d.N > 0 && d.Cout % 4 == 0 && d.Cout >= 16 && d.Cout <= kMaxCout && d.W % 4 == 0 && Ho == d.H && Wo == d.W &&
         (double)d.Cout * d.H * d.W * 4.0 < 2.0e9 && (long)d.H * d.W >= 4096;
}

const char* c3_kernel_name() { return "conv3x3_c3_valu_f32"; }

int c3_forward(const mscnn_conv_desc& d, const float* x, const float* w, const float* bias, float* y, unsigned* amax_out, hipStream_t st) {
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0, "conv(c3): x and y must be 16-byte aligned");
  C3Args a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.amax_out = amax_out;
  a.N = d.N; a.H = d.H; a.W = d.W; a.Cout = d.Cout; a.relu = d.relu;
  a.tiles_w = cdiv(d.W, kTileW); a.tiles_h = cdiv(d.H, kTileH);
  conv3x3_c3_kernel<<<(unsigned)((long)d.N * a.tiles_w * a.tiles_h), 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
