// Convolution forward for gfx950 (MI355X): im2col-free implicit GEMM on the fp32 MFMA pipe.
//
// Replaces ConvolutionLayer<Dtype>::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23 ->
// base_conv_layer.cpp:325-349: im2col_gpu + cublasSgemm + K=1 bias GEMM), CuDNNConvolutionLayer
// (cudnn_conv_layer.cu:11-46) and the in-place ReLU that follows every trunk conv (relu_layer.cu:9-26).
//
// GEMM view per image:  y[Cout][Ho*Wo] = W[Cout][Cin*Kh*Kw] x patches[Cin*Kh*Kw][Ho*Wo]
//   M = output channels, N = output pixels, K = (channel chunk, tap, channel-in-chunk).
//
// Kernel design (one 256-thread workgroup = 4 wavefronts, one per SIMD):
//   * tile  BM output channels x BN pixels (a TH x TW patch of the output plane); each wave owns a
//     (BM/WGM) x (BN/WGN) sub-tile as MI x NI blocks of v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//   * K loop in chunks of CK input channels.  Per chunk the workgroup stages into LDS
//       A: the packed weight slab [tap][ck][BM]        (contiguous in HBM -> float4 loads, no index math)
//       B: the input patch       [ck][TH+Kh-1][TW+Kw-1] with zero fill at the image border -- loaded ONCE and
//          reused by all Kh*Kw taps: this is what replaces the reference's 9x im2col read amplification.
//     An MFMA B operand for tap (kh,kw) is one ds_read_b32 at  lane_base + const((ck,kh,kw)) : lanes 0-31 read 32
//     consecutive pixels of channel ck, lanes 32-63 the same pixels of channel ck+1 (the 32x32x2 k-pair).
//   * register prefetch: chunk t+1 travels HBM/L2 -> VGPRs while chunk t is multiplied out of LDS.
//   * stream-K work split: the flattened (tile, chunk) iteration space is cut into G equal ranges, one per
//     workgroup, so that all 256 CUs stay busy at batch 1 even when tiles/256 is far from an integer
//     (conv5: 144 tiles).  A range that covers a tile only partially writes a raw fp32 partial slab; a small
//     fix-up kernel adds the (at most few) slabs of such tiles in k order (deterministic), then bias + ReLU.
//   * epilogue fuses bias and ReLU; stores are 128-B runs along W.
// Shapes the MFMA path does not cover (stride > 1, groups, Cin < 8) use direct_conv_kernel.
#include "common.h"
#include <new>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmArgs {
  const float* x; const float* wp; const float* bias; float* y; float* ws;
  int Cin, H, W, Cout, Ho, Wo, pad_h, pad_w;
  int MT, NTH, NTW, NT, KI, G, relu;
  long total_iters;
};

template <int BM_, int BN_, int WGM_, int WGN_, int KH_, int KW_, int CK_, int TW_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, KH = KH_, KW = KW_, CK = CK_, TW = TW_;
  static constexpr int TH = BN / TW;
  static constexpr int PH = TH + KH - 1, PW = TW + KW - 1, PS = PW;
  static constexpr int TAPS = KH * KW;
  static constexpr int A_ELEMS = TAPS * CK * BM;
  static constexpr int A_VEC4 = A_ELEMS / 4;
  static constexpr int A_PER_T = (A_VEC4 + 255) / 256;
  static constexpr int B_ELEMS = CK * PH * PW;
  static constexpr int B_PER_T = (B_ELEMS + 255) / 256;
  static constexpr int B_LDS = CK * PH * PS;
  static constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 32, NI = WN / 32;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  static_assert(WM % 32 == 0 && WN % 32 == 0 && BN % TW == 0 && CK % 2 == 0 && A_ELEMS % 4 == 0, "tile shape");
};

__device__ __forceinline__ void wg_range(long total, int G, int g, long& b, long& e) {
  b = total * g / G;
  e = total * (g + 1) / G;
}

// Weight packing: w[Cout][Cin][KH][KW] -> wp[mt][kc][tap][ck][BM], zero padded in both Cout and Cin.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                           int Cin, int taps, int BM, int CK, int MT, int KI) {
  const long total = (long)MT * KI * taps * CK * BM;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i;
    const int m = (int)(r % BM); r /= BM;
    const int ck = (int)(r % CK); r /= CK;
    const int tap = (int)(r % taps); r /= taps;
    const int kc = (int)(r % KI); r /= KI;
    const int mt = (int)r;
    const int co = mt * BM + m, ci = kc * CK + ck;
    wp[i] = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
  }
}

// Buffer resources (SGPR descriptors): every global access of the kernel is a raw buffer op with a 32-bit
// per-lane offset, so (a) no 64-bit address VGPRs, (b) out-of-range offsets read 0 / drop the store -- that is
// how the zero padding of the input patch, the ragged last tile and Cout < BM are handled without branches.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), /*stride*/ 0, (int)bytes, /*flags*/ 0x00020000);
}
constexpr unsigned kOob = 0x80000000u;   // offset that is out of range for every tensor here (< 2 GiB each)

template <class C>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmArgs a) {
  __shared__ __attribute__((aligned(16))) float ldsA[C::A_ELEMS];
  __shared__ float ldsB[C::B_LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  long it, it_end;
  wg_range(a.total_iters, a.G, blockIdx.x, it, it_end);

  // per-lane LDS read bases (floats)
  const float* aRd = ldsA + khalf * C::BM + wm * C::WM + l31;
  const float* bRd[C::NI];
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni) {
    const int p = wn * C::WN + ni * 32 + l31;
    bRd[ni] = ldsB + (khalf * C::PH + p / C::TW) * C::PS + (p % C::TW);
  }
  float4* aWr = reinterpret_cast<float4*>(ldsA) + tid;

  const int plane = a.H * a.W;
  const unsigned x_bytes = (unsigned)a.Cin * plane * 4u, y_bytes = (unsigned)a.Cout * a.Ho * a.Wo * 4u;
  const __amdgpu_buffer_rsrc_t wsrc = make_rsrc(a.wp, (unsigned)((long)a.MT * a.KI * C::A_ELEMS * 4));
  const __amdgpu_buffer_rsrc_t bias_rsrc = make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);

  while (it < it_end) {
    const int t = (int)(it / a.KI);
    const int k0 = (int)(it % a.KI);
    const int k1 = (int)min((long)a.KI, k0 + (it_end - it));
    // tile decode: t = mt * NT + nt ; nt = (img * NTH + th) * NTW + tw
    const int mt = t / a.NT, nt = t % a.NT;
    const int tw = nt % a.NTW, th = (nt / a.NTW) % a.NTH, img = nt / (a.NTW * a.NTH);
    const int h0 = th * C::TH, w0 = tw * C::TW;
    const __amdgpu_buffer_rsrc_t xsrc = make_rsrc(a.x + (long)img * a.Cin * plane, x_bytes);

    // byte offsets of this thread's patch elements inside the image (channel chunk term is the scalar offset)
    unsigned g_off[C::B_PER_T];
#pragma unroll
    for (int i = 0; i < C::B_PER_T; ++i) {
      const int idx = tid + i * 256;
      const int ck = idx / (C::PH * C::PW), rem = idx % (C::PH * C::PW);
      const int ih = h0 - a.pad_h + rem / C::PW, iw = w0 - a.pad_w + rem % C::PW;
      const bool ok = (idx < C::B_ELEMS) && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      g_off[i] = ok ? (unsigned)(ck * plane + ih * a.W + iw) * 4u : kOob;
    }

    f32x16 acc[C::MI][C::NI];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    float4 ra[C::A_PER_T];
    float rb[C::B_PER_T];
    const unsigned a_voff = (unsigned)tid * 16u;
    const unsigned a_tile = (unsigned)(mt * a.KI) * (C::A_ELEMS * 4u);

#define MSCNN_LOAD_CHUNK(kc)                                                                                        \
    {                                                                                                               \
      const unsigned a_soff = a_tile + (unsigned)(kc) * (C::A_ELEMS * 4u);                                          \
      _Pragma("unroll") for (int i = 0; i < C::A_PER_T; ++i) {                                                      \
        const unsigned vo = (C::A_VEC4 % 256 == 0 || tid + i * 256 < C::A_VEC4) ? a_voff : kOob;                    \
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, vo, a_soff + i * 4096u, 0)); \
      }                                                                                                             \
      const unsigned b_soff = (unsigned)(kc) * (unsigned)(C::CK * 4) * (unsigned)plane;                             \
      _Pragma("unroll") for (int i = 0; i < C::B_PER_T; ++i)                                                        \
        rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, g_off[i], b_soff, 0));         \
    }

    MSCNN_LOAD_CHUNK(k0);
    for (int kc = k0; kc < k1; ++kc) {
      __syncthreads();                 // everyone finished reading the previous chunk
#pragma unroll
      for (int i = 0; i < C::A_PER_T; ++i)
        if (C::A_VEC4 % 256 == 0 || tid + i * 256 < C::A_VEC4) aWr[i * 256] = ra[i];
#pragma unroll
      for (int i = 0; i < C::B_PER_T; ++i) {
        const int idx = tid + i * 256;
        const int ck = idx / (C::PH * C::PW), rem = idx % (C::PH * C::PW);
        if (C::B_ELEMS % 256 == 0 || idx < C::B_ELEMS) ldsB[(ck * C::PH + rem / C::PW) * C::PS + rem % C::PW] = rb[i];
      }
      __syncthreads();
      if (kc + 1 < k1) MSCNN_LOAD_CHUNK(kc + 1);   // in flight while this chunk is multiplied
#pragma unroll
      for (int kh = 0; kh < C::KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < C::KW; ++kw)
#pragma unroll
          for (int cp = 0; cp < C::CK / 2; ++cp) {
            float av[C::MI], bv[C::NI];
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi) av[mi] = aRd[((kh * C::KW + kw) * C::CK + cp * 2) * C::BM + mi * 32];
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni) bv[ni] = bRd[ni][(cp * 2 * C::PH + kh) * C::PS + kw];
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < C::NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
            // keep the scheduler from hoisting every ds_read of the chunk to the top (register blow-up)
            if (cp % 2 == 1) asm volatile("" ::: "memory");
          }
    }
#undef MSCNN_LOAD_CHUNK

    const bool full = (k0 == 0 && k1 == a.KI);
    if (full) {
      const __amdgpu_buffer_rsrc_t ysrc = make_rsrc(a.y + (long)img * a.Cout * a.Ho * a.Wo, y_bytes);
      const int HW = a.Ho * a.Wo;
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) {
        const int co0 = mt * C::BM + wm * C::WM + mi * 32 + 4 * khalf;
        float bvals[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          bvals[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                   bias_rsrc, (unsigned)co0 * 4u, ((r & 3) + 8 * (r >> 2)) * 4u, 0));
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) {
          const int p = wn * C::WN + ni * 32 + l31;
          const int oh = h0 + p / C::TW, ow = w0 + p % C::TW;
          const unsigned voff = (oh < a.Ho && ow < a.Wo) ? (unsigned)(co0 * HW + oh * a.Wo + ow) * 4u : kOob;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mi][ni][r] + bvals[r];
            if (a.relu) v = v > 0.f ? v : 0.f;
            const unsigned vo = (co0 + (r & 3) + 8 * (r >> 2) < a.Cout) ? voff : kOob;   // Cout < BM (proposal heads)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc, vo,
                                                  (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)HW * 4u, 0);
          }
        }
      }
    } else {
      float* slab = a.ws + ((long)blockIdx.x * 2 + (k0 > 0 ? 0 : 1)) * (C::BM * C::BN);
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) {
          const int p = wn * C::WN + ni * 32 + l31;
          float* sp = slab + (wm * C::WM + mi * 32 + 4 * khalf) * C::BN + p;
#pragma unroll
          for (int r = 0; r < 16; ++r) sp[((r & 3) + 8 * (r >> 2)) * C::BN] = acc[mi][ni][r];
        }
    }
    it += (k1 - k0);
    __syncthreads();   // LDS is re-used by the next segment's first stores
  }
}

// Sums the partial slabs of every tile that was split across workgroups, in k order, + bias + ReLU.
template <class C>
__global__ __launch_bounds__(256) void igemm_fixup_kernel(IgemmArgs a) {
  const int t = blockIdx.x;
  const long its = (long)t * a.KI, ite = its + a.KI;
  int gf = (int)(its * a.G / a.total_iters), gl = (int)((ite - 1) * a.G / a.total_iters);
  long b, e;
  wg_range(a.total_iters, a.G, gf, b, e);
  while (e <= its) { ++gf; wg_range(a.total_iters, a.G, gf, b, e); }
  while (b > its) { --gf; wg_range(a.total_iters, a.G, gf, b, e); }
  wg_range(a.total_iters, a.G, gl, b, e);
  while (e <= ite - 1) { ++gl; wg_range(a.total_iters, a.G, gl, b, e); }
  while (b > ite - 1) { --gl; wg_range(a.total_iters, a.G, gl, b, e); }
  if (gf == gl) return;   // the tile was computed whole by one workgroup

  const int mt = t / a.NT, nt = t % a.NT;
  const int tw = nt % a.NTW, th = (nt / a.NTW) % a.NTH, img = nt / (a.NTW * a.NTH);
  const int h0 = th * C::TH, w0 = tw * C::TW;
  float* yimg = a.y + (long)img * a.Cout * a.Ho * a.Wo;
  for (int i = threadIdx.x; i < C::BM * C::BN; i += 256) {
    const int m = i / C::BN, p = i % C::BN;
    float v = 0.f;
    for (int g = gf; g <= gl; ++g) {
      wg_range(a.total_iters, a.G, g, b, e);
      const float* slab = a.ws + ((long)g * 2 + (b > its ? 0 : 1)) * (C::BM * C::BN);
      v += slab[i];
    }
    const int co = mt * C::BM + m, oh = h0 + p / C::TW, ow = w0 + p % C::TW;
    if (co < a.Cout && oh < a.Ho && ow < a.Wo) {
      if (a.bias) v += a.bias[co];
      if (a.relu) v = v > 0.f ? v : 0.f;
      yimg[((long)co * a.Ho + oh) * a.Wo + ow] = v;
    }
  }
}

// Generic fallback: one output element per lane; any stride / pad / group.  Same (c, kh, kw) summation
// order as the definitional loop (test_convolution_layer.cpp:84-107).
__global__ __launch_bounds__(256) void direct_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int N,
                                                          int Cin, int H, int W, int Cout, int Kh, int Kw, int ph, int pw,
                                                          int sh, int sw, int group, int Ho, int Wo, int relu) {
  const long total = (long)N * Cout * Ho * Wo;
  const int cig = Cin / group, cog = Cout / group;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    long r = i / Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const int oc = (int)(r % Cout);
    const int n = (int)(r / Cout);
    const int g = oc / cog;
    const float* wk = w + (long)oc * cig * Kh * Kw;
    const float* xb = x + ((long)n * Cin + g * cig) * H * W;
    float acc = 0.f;
    for (int c = 0; c < cig; ++c)
      for (int ky = 0; ky < Kh; ++ky) {
        const int iy = oy * sh - ph + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < Kw; ++kx) {
          const int ix = ox * sw - pw + kx;
          if (ix < 0 || ix >= W) continue;
          acc += xb[((long)c * H + iy) * W + ix] * wk[(c * Kh + ky) * Kw + kx];
        }
      }
    if (bias) acc += bias[oc];
    if (relu) acc = acc > 0.f ? acc : 0.f;
    y[i] = acc;
  }
}

// ---- kernel table -------------------------------------------------------------------------------------
typedef void (*IgemmFn)(IgemmArgs);
struct KernelEntry {
  const char* name;
  int BM, BN, KH, KW, CK, TW, TH;
  IgemmFn main_fn, fix_fn;
};

#define ENTRY(BM, BN, WGM, WGN, KH, KW, CK, TW)                                                        \
  {"igemm_" #BM "x" #BN "_k" #KH "x" #KW "_tw" #TW, BM, BN, KH, KW, CK, TW, BN / TW,                  \
   igemm_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW>>, igemm_fixup_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW>>}

const KernelEntry kTable[] = {
    // trunk 3x3
    ENTRY(128, 128, 2, 2, 3, 3, 8, 16),
    ENTRY(128, 128, 2, 2, 3, 3, 8, 32),
    ENTRY(64, 256, 1, 4, 3, 3, 8, 32),
    // proposal heads (Cout = 4 + classes <= 32): kitti_car 5x5 / 7x7, ped-cyc + caltech 3x5 / 5x7
    ENTRY(32, 128, 1, 4, 5, 5, 8, 16),
    ENTRY(32, 128, 1, 4, 7, 7, 8, 16),
    ENTRY(32, 128, 1, 4, 5, 3, 8, 16),   // "3x5" heads are kernel_w 3 x kernel_h 5
    ENTRY(32, 128, 1, 4, 7, 5, 8, 16),   // "5x7": kernel_w 5 x kernel_h 7
};
constexpr int kTableN = sizeof(kTable) / sizeof(kTable[0]);

}  // namespace

struct mscnn_conv_plan {
  mscnn_conv_desc d;
  int Ho, Wo;
  int entry;          // -1: direct
  int MT, NTH, NTW, NT, KI, G;
  long total_iters;
  size_t packed_bytes, ws_bytes;
};

using namespace mscnn;

static void plan_shape(mscnn_conv_plan* p) {
  const mscnn_conv_desc& d = p->d;
  p->Ho = (d.H + 2 * d.pad_h - d.Kh) / d.stride_h + 1;     // conv_layer.cpp:8-22
  p->Wo = (d.W + 2 * d.pad_w - d.Kw) / d.stride_w + 1;
  p->entry = -1;
  p->packed_bytes = 0;
  p->ws_bytes = 0;
  if (d.stride_h != 1 || d.stride_w != 1 || d.group != 1 || d.Cin % 8 != 0) return;   // CK = 8 channels per chunk
  // choose the table entry with the least padded work
  double best = 1e300;
  for (int i = 0; i < kTableN; ++i) {
    const KernelEntry& k = kTable[i];
    if (k.KH != d.Kh || k.KW != d.Kw) continue;
    const long mt = cdiv(d.Cout, k.BM), nth = cdiv(p->Ho, k.TH), ntw = cdiv(p->Wo, k.TW);
    double cost = (double)mt * k.BM * nth * k.TH * ntw * k.TW;
    // prefer larger M tiles at equal cost (less re-staging of the input patch)
    cost *= (1.0 + 0.001 * (128.0 / k.BM));
    if (cost < best) { best = cost; p->entry = i; }
  }
  if (p->entry < 0) return;
  const KernelEntry& k = kTable[p->entry];
  p->MT = cdiv(d.Cout, k.BM);
  p->NTH = cdiv(p->Ho, k.TH);
  p->NTW = cdiv(p->Wo, k.TW);
  p->NT = d.N * p->NTH * p->NTW;
  p->KI = cdiv(d.Cin, k.CK);
  p->total_iters = (long)p->MT * p->NT * p->KI;
  // stream-K grid: two workgroups per CU, but never less than ~4 chunks per workgroup
  long G = 512;
  if (p->total_iters / 4 < G) G = p->total_iters / 4;
  if (G < 1) G = 1;
  // when the tile count already fills whole rounds of 512 slots, plain data-parallel is exact
  const long tiles = (long)p->MT * p->NT;
  if (tiles % 512 == 0) G = tiles;
  p->G = (int)G;
  p->packed_bytes = (size_t)p->MT * p->KI * k.KH * k.KW * k.CK * k.BM * sizeof(float);
  p->ws_bytes = (size_t)p->G * 2 * k.BM * k.BN * sizeof(float);
}

extern "C" int mscnn_conv2d_plan_create(const mscnn_conv_desc* desc, mscnn_conv_plan** plan_out) {
  MSCNN_REQUIRE(desc && plan_out, "conv plan: null pointer");
  const mscnn_conv_desc& d = *desc;
  MSCNN_REQUIRE(d.N >= 0 && d.Cin > 0 && d.H > 0 && d.W > 0 && d.Cout > 0 && d.Kh > 0 && d.Kw > 0, "conv plan: bad shape");
  MSCNN_REQUIRE(d.stride_h > 0 && d.stride_w > 0 && d.pad_h >= 0 && d.pad_w >= 0 && d.group > 0, "conv plan: bad params");
  MSCNN_REQUIRE(d.Cin % d.group == 0 && d.Cout % d.group == 0, "conv plan: channels not divisible by group");
  MSCNN_REQUIRE(d.H + 2 * d.pad_h >= d.Kh && d.W + 2 * d.pad_w >= d.Kw, "conv plan: kernel larger than padded input");
  mscnn_conv_plan* p = new (std::nothrow) mscnn_conv_plan();
  MSCNN_REQUIRE(p, "conv plan: out of memory");
  p->d = d;
  plan_shape(p);
  *plan_out = p;
  return MSCNN_OK;
}

extern "C" void mscnn_conv2d_plan_destroy(mscnn_conv_plan* plan) { delete plan; }
extern "C" size_t mscnn_conv2d_packed_weight_bytes(const mscnn_conv_plan* p) { return p ? p->packed_bytes : 0; }
extern "C" size_t mscnn_conv2d_workspace_bytes(const mscnn_conv_plan* p) { return p ? p->ws_bytes : 0; }
extern "C" const char* mscnn_conv2d_plan_kernel(const mscnn_conv_plan* p) {
  if (!p) return "";
  return p->entry < 0 ? "direct_f32" : kTable[p->entry].name;
}
extern "C" double mscnn_conv2d_plan_flops(const mscnn_conv_plan* p) {
  if (!p) return 0;
  const mscnn_conv_desc& d = p->d;
  return 2.0 * d.N * d.Cout * p->Ho * p->Wo * (double)(d.Cin / d.group) * d.Kh * d.Kw;
}
extern "C" int mscnn_conv2d_plan_set_batch(mscnn_conv_plan* p, int N) {
  MSCNN_REQUIRE(p && N >= 0, "conv plan: bad batch");
  p->d.N = N;
  plan_shape(p);
  return MSCNN_OK;
}

extern "C" int mscnn_conv2d_pack_weights(const mscnn_conv_plan* p, const float* w, float* packed, void* stream) {
  MSCNN_REQUIRE(p, "conv pack: null plan");
  if (p->entry < 0) return MSCNN_OK;   // direct kernel reads the Caffe layout
  MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
  const KernelEntry& k = kTable[p->entry];
  const long total = (long)p->MT * p->KI * k.KH * k.KW * k.CK * k.BM;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  pack_weights_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, packed, p->d.Cout, p->d.Cin, k.KH * k.KW, k.BM, k.CK,
                                                                 p->MT, p->KI);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_conv2d_fwd_f32(const mscnn_conv_plan* p, const float* x, const float* w, const float* packed,
                                    const float* bias, float* y, void* workspace, size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(p, "conv: null plan");
  const mscnn_conv_desc& d = p->d;
  if (d.N == 0) return MSCNN_OK;
  MSCNN_REQUIRE(x && y, "conv: null pointer");
  hipStream_t st = as_stream(stream);
  if (p->entry < 0) {
    MSCNN_REQUIRE(w, "conv: direct kernel needs the Caffe-layout weights");
    const long total = (long)d.N * d.Cout * p->Ho * p->Wo;
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    direct_conv_kernel<<<(int)blocks, 256, 0, st>>>(x, w, bias, y, d.N, d.Cin, d.H, d.W, d.Cout, d.Kh, d.Kw, d.pad_h, d.pad_w,
                                                    d.stride_h, d.stride_w, d.group, p->Ho, p->Wo, d.relu);
    MSCNN_POST_LAUNCH();
    return MSCNN_OK;
  }
  MSCNN_REQUIRE(packed, "conv: igemm kernel needs packed weights (mscnn_conv2d_pack_weights)");
  const KernelEntry& k = kTable[p->entry];
  const long tiles = (long)p->MT * p->NT;
  const bool split = (p->total_iters % p->G != 0) || ((p->total_iters / p->G) % p->KI != 0);
  if (split) {
    if (!workspace || workspace_bytes < p->ws_bytes) {
      set_error("conv: workspace %zu < %zu", workspace_bytes, p->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
  }
  IgemmArgs a;
  a.x = x; a.wp = packed; a.bias = bias; a.y = y; a.ws = static_cast<float*>(workspace);
  a.Cin = d.Cin; a.H = d.H; a.W = d.W; a.Cout = d.Cout; a.Ho = p->Ho; a.Wo = p->Wo; a.pad_h = d.pad_h; a.pad_w = d.pad_w;
  a.MT = p->MT; a.NTH = p->NTH; a.NTW = p->NTW; a.NT = p->NT; a.KI = p->KI; a.G = p->G; a.relu = d.relu;
  a.total_iters = p->total_iters;
  k.main_fn<<<p->G, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  if (split) {
    k.fix_fn<<<(int)tiles, 256, 0, st>>>(a);
    MSCNN_POST_LAUNCH();
  }
  return MSCNN_OK;
}
