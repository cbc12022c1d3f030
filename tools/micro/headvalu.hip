// Proposal-head convolutions on the VECTOR ALU (round 4): packed fp32 FMAs instead of the M = 4 MFMA of headconv.hip.
//
// Replaces ConvolutionLayer<Dtype>::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23) for the LFCN_* heads of the deploy nets
// (Cout = 4 + classes <= 12 over 512 channels, kernels 5x5 / 7x7 / 3x5 / 5x7, "same" padding, stride 1).
//
// Why.  On gfx950 v_pk_fma_f32 has the fp32 MFMA's peak (64 lanes x 2 x 2 FLOP per 4 cycles = 256 FLOP / clk / CU): for a GEMM whose
// M side is 6 .. 9 output channels the matrix core has nothing over the vector ALU.  v_mfma_f32_4x4x1_16b_f32 -- the only MFMA whose
// M is that small -- is issue-limited at ~14 cycles instead of 8 and multiplies 9 of 12 rows: 43 % of the peak in headconv.hip's
// kernel (52 - 80 TFLOP/s on the conv4_3 heads, 2 - 30 on the small maps, 0.38 ms per 7s-576 frame for 17.7 GFLOP).  Here:
//   * a lane owns EIGHT consecutive output pixels of one row and all output channels, as CO2 = ceil(Cout / 2) channel PAIRS: one
//     v_pk_fma_f32 = (acc[p][2c], acc[p][2c+1]) += (w[2c], w[2c+1]) * x[p + kw] -- two weight pairs are one wave-uniform 16-byte
//     LDS broadcast read, the x value one of the lane's registers selected by op_sel; 9 of 10 multiplier lanes do useful work for
//     Cout = 9, and the instruction issues back to back every 4 cycles;
//   * per (channel, kernel row) the lane reads its 16-float row window with four aligned ds_read_b128 and reuses it for all KW taps:
//     KW x CO2 x 8 packed FMAs per 4 + KW x ceil(CO2 / 2) LDS reads (280 : 25 for a 7 x 7 head).  (A first version with four pixels
//     per lane and 8-byte weight reads -- 140 : 38 -- was LDS-bound: the CU's one LDS pipe serves four SIMDs; 159 us on LFCN_1_7x7.)
//   * a wave = 64 lanes = 512 output columns: two rows of a conv4_3 head (2 x 240 columns), four rows of a conv5_3 head, more on the
//     small maps; a workgroup = 4 waves; the x patch [CK][rows + KH - 1][row + 8] and the chunk's weight pairs are staged in LDS
//     per CK = 4 input channels;
//   * tiles are few (18 on 72 x 240), so the (tile, channel chunk) space is cut stream-K style into G equal ranges exactly as in
//     headconv.hip: whole tiles are stored with bias / ReLU, partial sums go to fp32 slabs that a fix-up kernel adds in k order
//     (deterministic: fixed order, no atomics).
// Every product is one fused multiply-add of an fp32 chain in (channel, kh, kw) order per range -- the same arithmetic class as the
// MFMA kernels, held to the same 1e-4 bar against the oracle.
#include "common.h"
#include "headconv.h"
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct HvArgs {
  const float* x; const float* wp; const float* bias; float* y; float* ws;
  int N, Cin, H, W, Cout, Ho, Wo;
  int LPR, RPW, TR, PR, RL;          // lanes per row, rows per wave, rows per tile, patch rows (TR + KH - 1), LDS row length (8 LPR + 8)
  int NTH, KI, G, relu;
  long total_iters;
};

template <int KH_, int KW_, int CO2_, int CK_>
struct VCfg {
  static constexpr int KH = KH_, KW = KW_, CO2 = CO2_, CK = CK_, TAPS = KH * KW;
  static constexpr int PH = (KH - 1) / 2, PW = (KW - 1) / 2;      // "same" padding
  static constexpr int PX = 8;                                    // output pixels per lane
  static constexpr int CO2P = (CO2 + 1) / 2 * 2;                  // pairs per tap in LDS / the packed weights: even, so that two pairs are one 16-byte read
  static constexpr int W_FLOATS = CK * TAPS * CO2P * 2;           // one chunk of packed weight pairs
  static constexpr int SLAB = 2 * CO2 * 256 * PX;                 // floats: [channel][thread][8 pixels]
  static_assert(PW <= 4 && W_FLOATS % 4 == 0, "row window / weight staging");
};

__device__ __forceinline__ void wg_range(long total, int G, int g, long& b, long& e) {
  b = total * g / G;
  e = total * (g + 1) / G;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOob = 0x80000000u;

// w[Cout][Cin][KH][KW] -> wp[ci][kh][kw][pair < CO2P][2] (zero for channels past Cout)
__global__ __launch_bounds__(256) void headv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int taps,
                                                         int CO2P) {
  const long total = (long)Cin * taps * CO2P * 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i % 2);
    long r = i / 2;
    const int cp = (int)(r % CO2P); r /= CO2P;
    const int tap = (int)(r % taps);
    const int ci = (int)(r / taps);
    const int co = 2 * cp + e;
    wp[i] = co < Cout ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
  }
}

template <class C>
__global__ __launch_bounds__(256) void headv_kernel(HvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsW = lds;                                   // [CK][KH][KW][CO2] pairs
  float* ldsX = lds + C::W_FLOATS;                     // [CK][PR][RL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long it, it_end;
  wg_range(a.total_iters, a.G, blockIdx.x, it, it_end);

  const int rw = lane / a.LPR, lx = lane % a.LPR;      // lanes >= RPW * LPR idle along (their row index runs past the tile)
  const int trow = wave * a.RPW + rw;                  // output row of this lane inside the tile
  const int plane = a.H * a.W, co_stride = a.Ho * a.Wo;
  const int row_units = a.RL / 4, patch_units = C::CK * a.PR * row_units;
  const bool vec_rows = (a.W % 4) == 0;                // 16-byte units of an input row are then all inside or all outside the image
  const float* xrow0 = ldsX + (min(trow, a.TR - 1)) * a.RL + C::PX * lx;      // + (c PR + kh) RL per (channel, kernel row)
  const __amdgpu_buffer_rsrc_t wsrc = make_rsrc(a.wp, (unsigned)((long)a.KI * C::W_FLOATS * 4));

  while (it < it_end) {
    const int t = (int)(it / a.KI);
    const int k0 = (int)(it % a.KI);
    const int k1 = (int)min((long)a.KI, k0 + (it_end - it));
    it += (k1 - k0);
    const int th = t % a.NTH, img = t / a.NTH;
    const int h0 = th * a.TR;
    const __amdgpu_buffer_rsrc_t xsrc = make_rsrc(a.x + (long)img * a.Cin * plane, (unsigned)a.Cin * (unsigned)plane * 4u);

    f32x2 acc[C::PX][C::CO2];
#pragma unroll
    for (int p = 0; p < C::PX; ++p)
#pragma unroll
      for (int cp = 0; cp < C::CO2; ++cp) acc[p][cp] = f32x2{0.f, 0.f};

    for (int kc = k0; kc < k1; ++kc) {
      __syncthreads();                 // everyone finished reading the previous chunk
      // ---- stage the chunk: weight pairs (contiguous), then the x patch: LDS column j <-> input column j - 4, row r <-> h0 - PH + r
      for (int u = tid; u < C::W_FLOATS / 4; u += 256)
        reinterpret_cast<float4*>(ldsW)[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, (unsigned)u * 16u, (unsigned)kc * (C::W_FLOATS * 4u), 0));
      const unsigned c_soff = (unsigned)(kc * C::CK) * (unsigned)plane * 4u;
      if (vec_rows) {
        for (int u = tid; u < patch_units; u += 256) {
          const int q = u % row_units, rr = (u / row_units) % a.PR, c = u / (row_units * a.PR);
          const int ih = h0 - C::PH + rr, iw = 4 * q - 4;
          const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W && kc * C::CK + c < a.Cin;
          reinterpret_cast<float4*>(ldsX)[u] = __builtin_bit_cast(
              float4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, ok ? (unsigned)(c * plane + ih * a.W + iw) * 4u : kOob, c_soff, 0));
        }
      } else {
        for (int u = tid; u < patch_units * 4; u += 256) {
          const int j = u % a.RL, rr = (u / a.RL) % a.PR, c = u / (a.RL * a.PR);
          const int ih = h0 - C::PH + rr, iw = j - 4;
          const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W && kc * C::CK + c < a.Cin;
          ldsX[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, ok ? (unsigned)(c * plane + ih * a.W + iw) * 4u : kOob, c_soff, 0));
        }
      }
      __syncthreads();
      // ---- multiply: channel by channel (dynamic loop), kernel rows and taps unrolled
      for (int c = 0; c < C::CK; ++c) {
        const float* xr = xrow0 + c * a.PR * a.RL;
        const float4* wr = reinterpret_cast<const float4*>(ldsW) + c * C::TAPS * (C::CO2P / 2);
#pragma unroll
        for (int kh = 0; kh < C::KH; ++kh) {
          float xs[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 xq = *reinterpret_cast<const float4*>(xr + kh * a.RL + 4 * q);
            xs[4 * q] = xq.x; xs[4 * q + 1] = xq.y; xs[4 * q + 2] = xq.z; xs[4 * q + 3] = xq.w;
          }
#pragma unroll
          for (int kw = 0; kw < C::KW; ++kw) {
            f32x2 wv[C::CO2P];
#pragma unroll
            for (int c2 = 0; c2 < C::CO2P / 2; ++c2) {                                           // wave-uniform: LDS broadcast, two pairs per read
              const float4 w4 = wr[(kh * C::KW + kw) * (C::CO2P / 2) + c2];
              wv[2 * c2] = f32x2{w4.x, w4.y};
              wv[2 * c2 + 1] = f32x2{w4.z, w4.w};
            }
#pragma unroll
            for (int p = 0; p < C::PX; ++p) {
              const float xv = xs[p + kw - C::PW + 4];                                            // input column 8 lx + p + kw - PW
#pragma unroll
              for (int cp = 0; cp < C::CO2; ++cp) acc[p][cp] = __builtin_elementwise_fma(wv[cp], f32x2{xv, xv}, acc[p][cp]);
            }
          }
        }
      }
    }

    const int oh = h0 + trow, ow = C::PX * lx;
    const bool live = trow < a.TR && rw < a.RPW && oh < a.Ho;
    if (k0 == 0 && k1 == a.KI) {       // the whole K range: y = sum + bias (+ ReLU)
      if (live) {
        float* yp = a.y + (long)img * a.Cout * co_stride + (long)oh * a.Wo + ow;
#pragma unroll
        for (int cp = 0; cp < C::CO2; ++cp)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int co = 2 * cp + e;
            if (co < a.Cout) {
              const float bv = a.bias ? a.bias[co] : 0.f;
              float v[C::PX];
#pragma unroll
              for (int p = 0; p < C::PX; ++p) {
                v[p] = acc[p][cp][e] + bv;
                if (a.relu) v[p] = v[p] < 0.f ? 0.f : v[p];
              }
              float* dst = yp + (long)co * co_stride;
              if ((a.Wo % 4) == 0) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                if (ow + 4 < a.Wo) *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
              } else {
#pragma unroll
                for (int p = 0; p < C::PX; ++p) if (ow + p < a.Wo) dst[p] = v[p];
              }
            }
          }
      }
    } else {                            // a partial sum: this workgroup's slab (two per workgroup: the range may end one tile and begin the next)
      float4* slab = reinterpret_cast<float4*>(a.ws + ((long)blockIdx.x * 2 + (k0 > 0 ? 0 : 1)) * C::SLAB);
#pragma unroll
      for (int cp = 0; cp < C::CO2; ++cp)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          slab[((2 * cp + e) * 2 + 0) * 256 + tid] = make_float4(acc[0][cp][e], acc[1][cp][e], acc[2][cp][e], acc[3][cp][e]);
          slab[((2 * cp + e) * 2 + 1) * 256 + tid] = make_float4(acc[4][cp][e], acc[5][cp][e], acc[6][cp][e], acc[7][cp][e]);
        }
    }
  }
}

// Adds the partial slabs of every tile that was split across workgroups, in k order, + bias (+ ReLU).  One workgroup per
// (tile, output channel, half): a thread = four of the eight pixels of the main kernel's thread; the slab loads go out eight deep
// (a tile of a small map has up to 128 contributors: the chain of dependent batches is what this kernel's time is).
template <class C>
__global__ __launch_bounds__(256) void headv_fixup_kernel(HvArgs a) {
  __shared__ const float* s_slab[512];
  __shared__ int s_n;
  const int half = blockIdx.x % 2, co = (blockIdx.x / 2) % a.Cout, t = blockIdx.x / (2 * a.Cout);
  const int tid = threadIdx.x;
  if (tid == 0) {
    const long its = (long)t * a.KI, ite = its + a.KI;
    int gf = (int)(its * a.G / a.total_iters), gl = (int)((ite - 1) * a.G / a.total_iters);
    long b, e;
    wg_range(a.total_iters, a.G, gf, b, e);
    while (e <= its) { ++gf; wg_range(a.total_iters, a.G, gf, b, e); }
    while (b > its) { --gf; wg_range(a.total_iters, a.G, gf, b, e); }
    wg_range(a.total_iters, a.G, gl, b, e);
    while (e <= ite - 1) { ++gl; wg_range(a.total_iters, a.G, gl, b, e); }
    while (b > ite - 1) { --gl; wg_range(a.total_iters, a.G, gl, b, e); }
    int n = 0;
    if (gf != gl) {           // gf == gl: computed whole by one workgroup, already in y
      for (int g = gf; g <= gl && n < 512; ++g) {
        wg_range(a.total_iters, a.G, g, b, e);
        if (e <= b) continue;
        s_slab[n++] = a.ws + ((long)g * 2 + (b > its ? 0 : 1)) * C::SLAB + (long)(co * 2 + half) * 1024;
      }
    }
    s_n = n;
  }
  __syncthreads();
  const int n = s_n;
  if (n == 0) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 8 <= n; s += 8) {
    float4 u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = reinterpret_cast<const float4*>(s_slab[s + i])[tid];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v.x += u[i].x; v.y += u[i].y; v.z += u[i].z; v.w += u[i].w; }
  }
  for (; s < n; ++s) {
    const float4 u = reinterpret_cast<const float4*>(s_slab[s])[tid];
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  const int lane = tid & 63, wave = tid >> 6;
  const int rw = lane / a.LPR, lx = lane % a.LPR, trow = wave * a.RPW + rw;
  const int th = t % a.NTH, img = t / a.NTH;
  const int oh = th * a.TR + trow, ow = C::PX * lx + 4 * half;
  if (!(trow < a.TR && rw < a.RPW && oh < a.Ho) || ow >= a.Wo) return;
  const float bv = a.bias ? a.bias[co] : 0.f;
  float r[4] = {v.x + bv, v.y + bv, v.z + bv, v.w + bv};
  if (a.relu)
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = r[p] < 0.f ? 0.f : r[p];
  float* dst = a.y + ((long)img * a.Cout + co) * a.Ho * a.Wo + (long)oh * a.Wo + ow;
  if ((a.Wo % 4) == 0) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
  else
#pragma unroll
    for (int p = 0; p < 4; ++p) if (ow + p < a.Wo) dst[p] = r[p];
}

typedef void (*HvFn)(HvArgs);
struct HvEntry {
  const char* name;
  int KH, KW, CO2, CK, W_FLOATS, SLAB;
  HvFn main_fn, fix_fn;
};
#define HVENTRY(KH, KW, CO2, CK)                                                                                              \
  {"headvalu_k" #KH "x" #KW "_c" #CO2 "x2", KH, KW, CO2, CK, VCfg<KH, KW, CO2, CK>::W_FLOATS, VCfg<KH, KW, CO2, CK>::SLAB,    \
   headv_kernel<VCfg<KH, KW, CO2, CK>>, headv_fixup_kernel<VCfg<KH, KW, CO2, CK>>}
const HvEntry kHv[] = {
    HVENTRY(5, 5, 5, 4), HVENTRY(7, 7, 5, 4),      // kitti_car: 9 channels = 5 pairs
    HVENTRY(5, 3, 4, 4), HVENTRY(7, 5, 4, 4),      // ped/cyc (7 channels) and caltech (6: 3 pairs): "3x5" = kernel_w 3 x kernel_h 5
    HVENTRY(5, 3, 3, 4), HVENTRY(7, 5, 3, 4),
    HVENTRY(5, 5, 4, 4), HVENTRY(7, 7, 4, 4),
    HVENTRY(5, 5, 3, 4), HVENTRY(7, 7, 3, 4),
    HVENTRY(5, 3, 5, 4), HVENTRY(7, 5, 5, 4),
    HVENTRY(5, 5, 6, 4), HVENTRY(7, 7, 6, 4),
};
constexpr int kHvN = sizeof(kHv) / sizeof(kHv[0]);

}  // namespace

namespace mscnn {

bool headv_plan(const mscnn_conv_desc& d, int Ho, int Wo, HeadPlan* hp) {
  hp->valu = -1;
  if (d.stride_h != 1 || d.stride_w != 1 || d.group != 1 || d.N == 0 || d.Cout > 12 || d.Cout < 2 || d.Cin > 2048 || d.Cin % 4 != 0) return false;
  if (d.pad_h != (d.Kh - 1) / 2 || d.pad_w != (d.Kw - 1) / 2 || Ho != d.H || Wo != d.W || Wo > 512) return false;      // "same" padding, a row = at most one wave
  if ((double)d.Cin * d.H * d.W * 4.0 >= 2.0e9 || (double)d.Cout * Ho * Wo * 4.0 >= 2.0e9) return false;
  const int co2 = (d.Cout + 1) / 2;
  for (int i = 0; i < kHvN; ++i)
    if (kHv[i].KH == d.Kh && kHv[i].KW == d.Kw && kHv[i].CO2 == co2) hp->valu = i;
  if (hp->valu < 0) return false;
  const HvEntry& k = kHv[hp->valu];
  hp->LPR = cdiv(Wo, 8);
  hp->RPW = 64 / hp->LPR;
  hp->TR = 4 * hp->RPW;
  if (hp->TR > Ho) {                 // a map lower than the tile: fewer rows per wave (the other lanes idle), fewer patch rows to stage
    hp->RPW = cdiv(Ho, 4);
    hp->TR = 4 * hp->RPW;
  }
  hp->PR = hp->TR + k.KH - 1;
  hp->RL = 8 * hp->LPR + 8;
  hp->NTH = cdiv(Ho, hp->TR);
  hp->NTW = 1;
  hp->KI = d.Cin / k.CK;
  hp->lds_bytes = (size_t)(k.W_FLOATS + k.CK * hp->PR * hp->RL) * sizeof(float);
  if (hp->lds_bytes > 64 * 1024) return false;
  const long tiles = (long)d.N * hp->NTH;
  hp->total_iters = tiles * hp->KI;
  const int genv = tune_env("MSCNN_TUNE_GRID", d.tune_grid);   // tuning knob
  long G = genv > 0 ? genv : 768;                               // three workgroups per CU
  if (hp->total_iters / 2 < G) G = hp->total_iters / 2;         // at least ~2 chunks per workgroup
  if (G < 1) G = 1;
  hp->G = (int)G;
  hp->tiles = (int)tiles;
  hp->packed_bytes = (size_t)hp->KI * k.W_FLOATS * sizeof(float);
  hp->ws_bytes = (size_t)hp->G * 2 * k.SLAB * sizeof(float);
  return true;
}

const char* headv_kernel_name(const HeadPlan& hp) { return kHv[hp.valu].name; }

int headv_pack(const mscnn_conv_desc& d, const HeadPlan& hp, const float* w, float* packed, hipStream_t st) {
  const HvEntry& k = kHv[hp.valu];
  const int co2p = (k.CO2 + 1) / 2 * 2;
  const long total = (long)d.Cin * k.KH * k.KW * co2p * 2;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  headv_pack_kernel<<<(int)blocks, 256, 0, st>>>(w, packed, d.Cout, d.Cin, k.KH * k.KW, co2p);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int headv_forward(const mscnn_conv_desc& d, const HeadPlan& hp, int Ho, int Wo, const float* x, const float* packed, const float* bias,
                  float* y, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const HvEntry& k = kHv[hp.valu];
  if (!workspace || workspace_bytes < hp.ws_bytes) {
    set_error("conv(head, valu): workspace %zu < %zu", workspace_bytes, hp.ws_bytes);
    return MSCNN_ERR_WORKSPACE;
  }
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 && reinterpret_cast<uintptr_t>(packed) % 16 == 0,
                "conv(head, valu): x, y and the packed weights must be 16-byte aligned");
  HvArgs a;
  a.x = x; a.wp = packed; a.bias = bias; a.y = y; a.ws = static_cast<float*>(workspace);
  a.N = d.N; a.Cin = d.Cin; a.H = d.H; a.W = d.W; a.Cout = d.Cout; a.Ho = Ho; a.Wo = Wo;
  a.LPR = hp.LPR; a.RPW = hp.RPW; a.TR = hp.TR; a.PR = hp.PR; a.RL = hp.RL;
  a.NTH = hp.NTH; a.KI = hp.KI; a.G = hp.G; a.relu = d.relu; a.total_iters = hp.total_iters;
  k.main_fn<<<hp.G, 256, hp.lds_bytes, st>>>(a);
  MSCNN_POST_LAUNCH();
  k.fix_fn<<<hp.tiles * d.Cout * 2, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
