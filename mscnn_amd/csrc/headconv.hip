// Proposal-head convolutions (LFCN_*: Cout = 4 + classes <= 12, kernels 5x5 / 7x7 / 3x5 / 5x7 over 512 channels) for gfx950.
//
// Replaces ConvolutionLayer<Dtype>::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23) for the seven / eight LFCN heads of
// the deploy nets (examples/kitti_car/mscnn-7s-576/mscnn_deploy.prototxt, "LFCN_*" layers).
//
// Why a separate kernel: with 6..9 output channels the 32-row tile of the trunk igemm kernel multiplies 72-81 % zeros
// (measured 31-36 TFLOP/s on the two conv4_3 heads).  v_mfma_f32_4x4x1_16b_f32 has the same 64 FLOP/clk/SIMD rate but
// an M of only 4, and its A-broadcast mode (CBSZ = 4, ABID = b) feeds all 16 blocks with the A rows held by lanes
// 4b..4b+3 -- so ONE VGPR holds the weights of 4 output channels for 16 consecutive k, selected by an immediate, and the
// weights of a whole channel chunk live in registers for the chunk (no A traffic in the inner loop at all):
//
//   D[4 channels][64 pixels] += A(abid = k % 16)[4 channels] (x) B[64 pixels]      8 cycles, one k
//
//   * lane l <-> output pixel l of the wave's 64-pixel group (2 rows x 32 columns); B operand = one ds_read_b32 of the
//     LDS input patch at  lane_base + const(channel, kh, kw)  -- im2col-free, like the trunk kernel;
//   * a workgroup (4 waves) owns a 16 x 32 pixel tile: each wave 2 groups x NQ channel quads = 2*NQ f32x4 accumulators;
//   * K loop in chunks of CK input channels (8; 4 for the 7-row kernels to keep the weights at <= 39 registers); per chunk
//     the patch [CK][16+KH-1][32+KW-1] and the chunk's packed weights wp[chunk][quad][reg][lane] are staged through
//     registers into LDS (prefetched one chunk ahead); each wave then copies the weights LDS -> VGPRs once per chunk;
//   * tiles are few (72x240 -> 40 tiles), so the (tile, chunk) space is cut stream-K style into G equal ranges;
//     partial sums go to fp32 slabs and a fix-up kernel adds them in a fixed order (deterministic) + bias.  (r5) The fix-up is
//     wide -- a workgroup per (tile quarter, channel), 8 slab groups adding in parallel, the contributor list built by the whole
//     workgroup: 4 us instead of 11 - 18 --, so the split is fine: the 5-row kernels run chunks of 4 channels (the 7-row ones of 2
//     on maps of <= 16 tiles), see head_plan.
// Useful-work fraction: Cout / (4 * NQ) = 75 % for the 9-channel KITTI heads (vs 28 % with M = 32).
// Measured (conv4_3 heads 5x5 / 7x7, 1 x 512 x 72 x 240): 81 / 140 us against 136 / 231 us for the 32-row igemm tile.
// PMC (rocprofv3, 5x5 head): 12.29 M MFMAs, SQ_VALU_MFMA_BUSY_CYCLES = 8 cycles each, 57 % of the kernel's cycles at
// 2.32 GHz -- this MFMA form is issue-limited: tools/micro/mfma_rate.hip gives 9-10.5 cycles per 4x4x1 MFMA in isolation
// (32.2 / 64.5 for 16x16x4 / 32x32x2), and ~14 inside this loop.  Variants that were built and measured slower or equal:
// 16x16x4 with the same structure (84 / 142 us main kernel vs 72 / 134), CK = 4 with 3 workgroups per CU (77 / 140 us
// total), and a v_pk_fma_f32 version with wave-uniform weights in SGPRs (the 64 weight SGPRs pushed the buffer
// descriptors into VGPRs -> waterfall loops).
#include "common.h"
#include "headconv.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadArgs {
  const float* x; const float* wp; const float* bias; float* y; float* ws;
  int N, Cin, H, W, Cout, Ho, Wo, pad_h, pad_w;
  int NTH, NTW, KI, G, relu;
  long total_iters;
};

template <int KH_, int KW_, int NQ_, int CK_>
struct HCfg {
  static constexpr int KH = KH_, KW = KW_, NQ = NQ_, CK = CK_, TH = 16, TW = 32, BN = TH * TW;
  static constexpr int PH = TH + KH - 1, PW = TW + KW - 1, CH_STRIDE = PH * PW, TAPS = KH * KW, KC = CK * TAPS;
  static constexpr int AREGS = (KC + 15) / 16;       // weight registers per channel quad: 16 k per register
  static constexpr int A_ELEMS = NQ * AREGS * 64, A_VEC4 = A_ELEMS / 4, A_PER_T = (A_VEC4 + 255) / 256;
  static constexpr int B_ELEMS = CK * CH_STRIDE, B_PER_T = (B_ELEMS + 255) / 256;
  static constexpr int SLAB = NQ * 4 * BN;
  static constexpr int GK = 4;                       // k values per software-pipeline group
  static constexpr int NG = (KC + GK - 1) / GK;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void wg_range(long total, int G, int g, long& b, long& e) {
  b = total * g / G;
  e = total * (g + 1) / G;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOob = 0x80000000u;

// w[Cout][Cin][KH][KW] -> wp[chunk][quad][reg][lane]:  lane l of register r holds channel 4*quad + l%4 at
// k = 16*r + l/4 of the chunk, k = ck * TAPS + tap (taps fastest, so neighbouring k read neighbouring LDS words).
__global__ __launch_bounds__(256) void head_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                        int taps, int NQ, int AREGS, int CK, int KI) {
  const long total = (long)KI * NQ * AREGS * 64;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int lane = (int)(i % 64);
    long r = i / 64;
    const int reg = (int)(r % AREGS); r /= AREGS;
    const int q = (int)(r % NQ);
    const int kc = (int)(r / NQ);
    const int k = reg * 16 + lane / 4, co = q * 4 + lane % 4;
    const int ck = k / taps, tap = k % taps, ci = kc * CK + ck;
    wp[i] = (ck < CK && co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
  }
}

// B operands of pipeline group G: GK consecutive k, both 64-pixel groups of the wave
template <class C, int G>
__device__ __forceinline__ void lds_group(const float* b0, const float* b1, float (&bv)[C::GK][2]) {
#pragma unroll
  for (int j = 0; j < C::GK; ++j) {
    constexpr int kbase = G * C::GK;
    const int k = kbase + j;
    if (k < C::KC) {
      const int ck = k / C::TAPS, tap = k % C::TAPS;
      const int off = ck * C::CH_STRIDE + (tap / C::KW) * C::PW + tap % C::KW;
      bv[j][0] = b0[off];
      bv[j][1] = b1[off];
    }
  }
}

template <class C>
__global__ __launch_bounds__(256, 2) void head_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float ldsA[C::A_ELEMS];
  __shared__ float ldsB[C::B_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  long it, it_end;
  wg_range(a.total_iters, a.G, wg, it, it_end);

  // pixel of this lane inside the 16 x 32 tile, for the wave's two 64-pixel groups (2 rows x 32 columns each)
  const int prow = wave * 4 + (lane >> 5), pcol = lane & 31;
  const float* bRd0 = ldsB + prow * C::PW + pcol;
  const float* bRd1 = bRd0 + 2 * C::PW;
  float4* aWr = reinterpret_cast<float4*>(ldsA) + tid;

  const int plane = a.H * a.W;
  const bool ragged_c = (a.Cin % C::CK) != 0;
  const __amdgpu_buffer_rsrc_t wsrc = make_rsrc(a.wp, (unsigned)((long)a.KI * C::A_ELEMS * 4));
  const __amdgpu_buffer_rsrc_t bias_rsrc = make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
  const int co_stride = a.Ho * a.Wo;

  while (it < it_end) {
    const int t = (int)(it / a.KI);
    const int k0 = (int)(it % a.KI);
    const int k1 = (int)min((long)a.KI, k0 + (it_end - it));
    it += (k1 - k0);
    const int tw = t % a.NTW, th = (t / a.NTW) % a.NTH, img = t / (a.NTW * a.NTH);
    const int h0 = th * C::TH, w0 = tw * C::TW;
    const __amdgpu_buffer_rsrc_t xsrc = make_rsrc(a.x + (long)img * a.Cin * plane, (unsigned)a.Cin * (unsigned)plane * 4u);

    unsigned g_off[C::B_PER_T];
#pragma unroll
    for (int i = 0; i < C::B_PER_T; ++i) {
      const int idx = tid + i * 256;
      const int ck = idx / C::CH_STRIDE, rem = idx % C::CH_STRIDE;
      const int ih = h0 - a.pad_h + rem / C::PW, iw = w0 - a.pad_w + rem % C::PW;
      const bool ok = idx < C::B_ELEMS && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      g_off[i] = ok ? (unsigned)(ck * plane + ih * a.W + iw) * 4u : kOob;
    }

    f32x4 acc[2][C::NQ];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < C::NQ; ++q) acc[g][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    float rb[C::B_PER_T];
    float4 ra[C::A_PER_T];
    float areg[C::NQ][C::AREGS];

#define HEAD_LOAD_CHUNK(kc)                                                                            \
    {                                                                                                  \
      const unsigned a_soff = (unsigned)(kc) * (C::A_ELEMS * 4u);                                      \
      _Pragma("unroll") for (int i = 0; i < C::A_PER_T; ++i) {                                         \
        const unsigned vo = (C::A_VEC4 % 256 == 0 || tid + i * 256 < C::A_VEC4) ? (unsigned)tid * 16u : kOob; \
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, vo, a_soff + i * 4096u, 0)); \
      }                                                                                                \
      const unsigned b_soff = (unsigned)(kc) * (unsigned)(C::CK * 4) * (unsigned)plane;                \
      const int c_left = a.Cin - (kc) * C::CK;                                                         \
      _Pragma("unroll") for (int i = 0; i < C::B_PER_T; ++i) {                                         \
        unsigned vo = g_off[i];                                                                        \
        if (ragged_c && (tid + i * 256) / C::CH_STRIDE >= c_left) vo = kOob;                           \
        rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, vo, b_soff, 0));  \
      }                                                                                                \
    }

    HEAD_LOAD_CHUNK(k0);
    for (int kc = k0; kc < k1; ++kc) {
      __syncthreads();                 // everyone finished reading the previous chunk
#pragma unroll
      for (int i = 0; i < C::A_PER_T; ++i)
        if (C::A_VEC4 % 256 == 0 || tid + i * 256 < C::A_VEC4) aWr[i * 256] = ra[i];
#pragma unroll
      for (int i = 0; i < C::B_PER_T; ++i)
        if (C::B_ELEMS % 256 == 0 || tid + i * 256 < C::B_ELEMS) ldsB[tid + i * 256] = rb[i];
      __syncthreads();
      if (kc + 1 < k1) HEAD_LOAD_CHUNK(kc + 1);      // in flight while this chunk is multiplied
      // the chunk's weights: LDS -> registers, once per wave
#pragma unroll
      for (int q = 0; q < C::NQ; ++q)
#pragma unroll
        for (int r = 0; r < C::AREGS; ++r) areg[q][r] = ldsA[(q * C::AREGS + r) * 64 + lane];

      float bv[2][C::GK][2];
      lds_group<C, 0>(bRd0, bRd1, bv[0]);
      static_for<0, C::NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + 1 < C::NG) lds_group<C, g + 1>(bRd0, bRd1, bv[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);           // next group's ds_reads are issued before this group's MFMAs
        static_for<0, C::GK>([&](auto jc) {
          constexpr int j = decltype(jc)::value, k = g * C::GK + j;      // register k / 16, broadcast block k % 16
          if constexpr (k < C::KC) {
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) {
              acc[0][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(areg[q][k / 16], bv[g & 1][j][0], acc[0][q], 4, k % 16, 0);
              acc[1][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(areg[q][k / 16], bv[g & 1][j][1], acc[1][q], 4, k % 16, 0);
            }
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
#undef HEAD_LOAD_CHUNK

    // D layout of the 4x4x1 MFMA: lane l, element i  =  channel (4q + i) of pixel l
    const bool full = (k0 == 0 && k1 == a.KI);
    if (full) {
      const __amdgpu_buffer_rsrc_t ysrc = make_rsrc(a.y + (long)img * a.Cout * co_stride, (unsigned)a.Cout * (unsigned)co_stride * 4u);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int oh = h0 + prow + 2 * g, ow = w0 + pcol;
        const unsigned voff = (oh < a.Ho && ow < a.Wo) ? (unsigned)(oh * a.Wo + ow) * 4u : kOob;
#pragma unroll
        for (int q = 0; q < C::NQ; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = q * 4 + i;
            float v = acc[g][q][i] + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bias_rsrc, 0, (unsigned)co * 4u, 0));
            if (a.relu) v = v < 0.f ? 0.f : v;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc, co < a.Cout ? voff : kOob,
                                                  (unsigned)co * (unsigned)co_stride * 4u, 0);
          }
      }
    } else {
      float* slab = a.ws + ((long)wg * 2 + (k0 > 0 ? 0 : 1)) * C::SLAB;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int p = (prow + 2 * g) * C::TW + pcol;
#pragma unroll
        for (int q = 0; q < C::NQ; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) slab[(q * 4 + i) * C::BN + p] = acc[g][q][i];
      }
    }
    __syncthreads();   // LDS is re-used by the next segment's first stores
  }
}

// Adds the partial slabs of every tile that was split across workgroups + bias (+ ReLU).  One workgroup per (tile, output channel,
// quarter of the tile = 4 rows x 32 columns); its 256 threads are 8 slab groups x 32 float4 lanes: group s adds the s-th eighth of the
// tile's contributor list in k order, eight loads in flight, and the eight partial sums are added in group order through LDS -- a
// fixed order, so the result is reproducible bit for bit, and a list of 256 contributors is four rounds of loads instead of sixty-four.
// The contributor list is built by the whole workgroup (thread i <-> workgroup gf + i: its range is two 64-bit divisions, which ONE
// thread looping over 32 ... 256 contributors had turned into 5 ... 60 us of serial latency in the round-4 kernel).
template <class C>
__global__ __launch_bounds__(256) void head_fixup_kernel(HeadArgs a) {
  __shared__ const float* s_slab[256];
  __shared__ __attribute__((aligned(16))) float s_part[8][128];
  __shared__ int s_gf, s_n;
  const int quarter = blockIdx.x & 3, tc = blockIdx.x >> 2;
  const int t = tc / a.Cout, co = tc % a.Cout;
  const int tw = t % a.NTW, th = (t / a.NTW) % a.NTH, img = t / (a.NTW * a.NTH);
  if (th * C::TH + quarter * 4 >= a.Ho) return;              // the whole quarter lies below the map
  const long its = (long)t * a.KI, ite = its + a.KI;
  if (threadIdx.x == 0) {
    int gf = (int)(its * a.G / a.total_iters), gl = (int)((ite - 1) * a.G / a.total_iters);
    long b, e;
    wg_range(a.total_iters, a.G, gf, b, e);
    while (e <= its) { ++gf; wg_range(a.total_iters, a.G, gf, b, e); }
    while (b > its) { --gf; wg_range(a.total_iters, a.G, gf, b, e); }
    wg_range(a.total_iters, a.G, gl, b, e);
    while (e <= ite - 1) { ++gl; wg_range(a.total_iters, a.G, gl, b, e); }
    while (b > ite - 1) { --gl; wg_range(a.total_iters, a.G, gl, b, e); }
    s_gf = gf;
    s_n = gf == gl ? 0 : min(gl - gf + 1, 256);           // gf == gl: computed whole by one workgroup, already in y
  }
  __syncthreads();
  const int n = s_n;
  if (n == 0) return;
  if ((int)threadIdx.x < n) {
    const int g = s_gf + threadIdx.x;
    long b, e;
    wg_range(a.total_iters, a.G, g, b, e);
    s_slab[threadIdx.x] = e <= b ? nullptr : a.ws + ((long)g * 2 + (b > its ? 0 : 1)) * C::SLAB + co * C::BN;   // (an empty range: G > iterations)
  }
  __syncthreads();
  const int sg = threadIdx.x >> 5, l = threadIdx.x & 31;
  const int p = quarter * 128 + l * 4;
  const int s0 = sg * n / 8, s1 = (sg + 1) * n / 8;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int D = 8;
  for (int s = s0; s < s1; s += D) {
    float4 u[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float* sl = s + j < s1 ? s_slab[s + j] : nullptr;
      u[j] = sl ? *reinterpret_cast<const float4*>(sl + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (s + j < s1 && s_slab[s + j]) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
  }
  *reinterpret_cast<float4*>(&s_part[sg][l * 4]) = v;
  __syncthreads();
  if (threadIdx.x >= 128) return;
  const int px = threadIdx.x;
  float r = s_part[0][px];
#pragma unroll
  for (int j = 1; j < 8; ++j) r += s_part[j][px];
  r += a.bias ? a.bias[co] : 0.f;
  if (a.relu) r = r < 0.f ? 0.f : r;
  const int oh = th * C::TH + quarter * 4 + px / C::TW, ow = tw * C::TW + px % C::TW;
  if (oh < a.Ho && ow < a.Wo) a.y[((long)img * a.Cout + co) * a.Ho * a.Wo + oh * a.Wo + ow] = r;
}

typedef void (*HeadFn)(HeadArgs);
struct HeadEntry {
  const char* name;
  int KH, KW, NQ, CK, AREGS, SLAB;
  HeadFn main_fn, fix_fn;
};
#define HENTRY(KH, KW, NQ, CK)                                                                                        \
  {"head4x4_k" #KH "x" #KW "_m" #NQ "x4", KH, KW, NQ, CK, HCfg<KH, KW, NQ, CK>::AREGS, HCfg<KH, KW, NQ, CK>::SLAB,    \
   head_kernel<HCfg<KH, KW, NQ, CK>>, head_fixup_kernel<HCfg<KH, KW, NQ, CK>>}
const HeadEntry kHeads[] = {
    HENTRY(5, 5, 3, 8), HENTRY(7, 7, 3, 4),      // kitti_car: 9 channels
    HENTRY(5, 3, 2, 8), HENTRY(7, 5, 2, 4),      // ped/cyc (7) and caltech (6): "3x5" = kernel_w 3 x kernel_h 5
    HENTRY(5, 5, 2, 8), HENTRY(7, 7, 2, 4),
    HENTRY(5, 3, 3, 8), HENTRY(7, 5, 3, 4),
    // the same eight with half the channels per chunk (entry + kHalf): twice the (tile, chunk) units on maps of a few tiles
    HENTRY(5, 5, 3, 4), HENTRY(7, 7, 3, 2),
    HENTRY(5, 3, 2, 4), HENTRY(7, 5, 2, 2),
    HENTRY(5, 5, 2, 4), HENTRY(7, 7, 2, 2),
    HENTRY(5, 3, 3, 4), HENTRY(7, 5, 3, 2),
};
constexpr int kHalf = 8;

}  // namespace

namespace mscnn {

bool head_plan(const mscnn_conv_desc& d, int Ho, int Wo, HeadPlan* hp) {
  const bool off = (tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 2) != 0;   // A/B switch: heads on the 32-row igemm tile
  hp->entry = -1;
  hp->valu = -1;
#ifdef MSCNN_HEAD_VALU_WITNESS
  // (witness build only -- `make witness`, tools/micro/libmscnn_hip_witness.so: tune_flags bit 13 puts the heads on the packed-FMA
  // kernel of tools/micro/headvalu.hip, a second implementation of the same split that measured 12-45 % slower on every head of the
  // 7s nets, profiles/r04_ab_heads_valu.txt; retired from the product library in round 5)
  if (!off && (tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 8192) && headv_plan(d, Ho, Wo, hp)) { hp->entry = 0; return true; }
#endif
  if (off || d.stride_h != 1 || d.stride_w != 1 || d.group != 1 || d.N == 0 || d.Cout > 12 || d.Cin > 1024) return false;
  if ((double)d.Cin * d.H * d.W * 4.0 >= 2.0e9 || (double)d.Cout * Ho * Wo * 4.0 >= 2.0e9) return false;
  const int nq = d.Cout <= 8 ? 2 : 3;
  for (int i = 0; i < kHalf; ++i)
    if (kHeads[i].KH == d.Kh && kHeads[i].KW == d.Kw && kHeads[i].NQ == nq) hp->entry = i;
  if (hp->entry < 0) return false;
  hp->NTH = cdiv(Ho, 16);
  hp->NTW = cdiv(Wo, 32);
  const long tiles = (long)d.N * hp->NTH * hp->NTW;
  // Half chunks double the (tile, chunk) units the split can hand out (a unit is 3.7 us of MFMAs instead of 7.3):
  //  * maps of a few tiles (the 36 x 120 / 18 x 60 / 9 x 30 levels of the 7s nets: 12 / 4 / 1 tiles) are latency chains, not
  //    throughput: one unit per workgroup while that still is one workgroup per CU.  Measured per level (r5,
  //    tools/sessions/r05_s40.sh / r05_s41.sh): -3 ... -4 us on the 5x5 heads, +- 0 on the 7x7 ones;
  //  * the 8-channel-chunk kernels (5x5, 5x3) also on large maps: LFCN_1_5x5 (40 tiles) 84.6 - 86.1 -> 80.7 - 81.9 us (r05_s46.sh).
  // tune_variant 500 / 501: full / half chunks whatever the map.
  const int variant = tune_env("MSCNN_TUNE_VARIANT", d.tune_variant);
  const bool half = (variant == 501 || (variant != 500 && (tiles <= 16 || kHeads[hp->entry].CK == 8))) &&
                    cdiv(d.Cin, kHeads[hp->entry + kHalf].CK) <= 256;
  if (half) hp->entry += kHalf;
  const HeadEntry& k = kHeads[hp->entry];
  hp->KI = cdiv(d.Cin, k.CK);       // <= 256 contributors per tile (fix-up slab list)
  hp->total_iters = tiles * hp->KI;
  const int genv = tune_env("MSCNN_TUNE_GRID", d.tune_grid);   // tuning knob
  long G;
  if (genv > 0) G = genv < hp->total_iters ? genv : hp->total_iters;
  else if (half) G = hp->total_iters <= 256 ? hp->total_iters : (hp->total_iters / 2 < 256 ? 256 : hp->total_iters / 2 > 512 ? 512 : hp->total_iters / 2);
  else G = hp->total_iters / 2 < 512 ? hp->total_iters / 2 : 512;      // at least ~2 chunks per workgroup
  if (G < 1) G = 1;
  hp->G = (int)G;
  hp->tiles = (int)tiles;
  hp->packed_bytes = (size_t)hp->KI * k.NQ * k.AREGS * 64 * sizeof(float);
  hp->ws_bytes = (size_t)hp->G * 2 * k.SLAB * sizeof(float);
  return true;
}

const char* head_kernel_name(const HeadPlan& hp) {
#ifdef MSCNN_HEAD_VALU_WITNESS
  if (hp.valu >= 0) return headv_kernel_name(hp);
#endif
  return kHeads[hp.entry].name;
}

int head_pack(const mscnn_conv_desc& d, const HeadPlan& hp, const float* w, float* packed, hipStream_t st) {
#ifdef MSCNN_HEAD_VALU_WITNESS
  if (hp.valu >= 0) return headv_pack(d, hp, w, packed, st);
#endif
  const HeadEntry& k = kHeads[hp.entry];
  const long total = (long)hp.KI * k.NQ * k.AREGS * 64;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  head_pack_kernel<<<(int)blocks, 256, 0, st>>>(w, packed, d.Cout, d.Cin, k.KH * k.KW, k.NQ, k.AREGS, k.CK, hp.KI);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int head_forward(const mscnn_conv_desc& d, const HeadPlan& hp, int Ho, int Wo, const float* x, const float* packed,
                 const float* bias, float* y, void* workspace, size_t workspace_bytes, hipStream_t st) {
#ifdef MSCNN_HEAD_VALU_WITNESS
  if (hp.valu >= 0) return headv_forward(d, hp, Ho, Wo, x, packed, bias, y, workspace, workspace_bytes, st);
#endif
  const HeadEntry& k = kHeads[hp.entry];
  if (!workspace || workspace_bytes < hp.ws_bytes) {
    set_error("conv(head): workspace %zu < %zu", workspace_bytes, hp.ws_bytes);
    return MSCNN_ERR_WORKSPACE;
  }
  HeadArgs a;
  a.x = x; a.wp = packed; a.bias = bias; a.y = y; a.ws = static_cast<float*>(workspace);
  a.N = d.N; a.Cin = d.Cin; a.H = d.H; a.W = d.W; a.Cout = d.Cout; a.Ho = Ho; a.Wo = Wo; a.pad_h = d.pad_h; a.pad_w = d.pad_w;
  a.NTH = hp.NTH; a.NTW = hp.NTW; a.KI = hp.KI; a.G = hp.G; a.relu = d.relu; a.total_iters = hp.total_iters;
  k.main_fn<<<hp.G, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  k.fix_fn<<<hp.tiles * d.Cout * 4, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
