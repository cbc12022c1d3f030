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
//
// The same template has two more operand types (Cfg::F16 / Cfg::X3): fp16 operands rounded while they are staged (the
// reduced-precision mode, MSCNN_CONV_ALGO_F16), and the split-fp16 form (MSCNN_CONV_ALGO_WINO_F3_X3 on layers that stay direct:
// every fp32 operand as fp16 hi + lo, three MFMAs per pair, fp32-grade; x3_device.h, wino_x3.hip).  Every kernel of the family
// can publish max |y| of what it stores for a split-fp16 consumer (IgemmArgs::amax_out).
#include "common.h"
#include "headconv.h"
#include "winograd.h"
#include "wino_x3.h"
#include "wgemm.h"
#include "roipool_wino.h"
#include "conv_c3.h"
#include "wconv.h"
#include "wf2conv.h"
#include "x3_device.h"
#include <new>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct IgemmArgs {
  const float* x; const float* wp; const float* bias; float* y; float* ws;
  int N, Cin, H, W, Cout, Ho, Wo, pad_h, pad_w;
  int MT, NTH, NTW, NT, KI, G, relu;
  int xcd_map;         // 1: XCD-aware workgroup -> range mapping
  int epi_prio;        // 1: the epilogue of a tile runs at wave priority 3 (tune_flags bit 14, A/B)
  float* yp;           // != nullptr: also write the 2x2 / stride-2 max-pooled output [N][Cout][Hp][Wp] (fused PoolingLayer)
  int Hp, Wp;
  int nt_major;        // 1: tile index t = nt * MT + mt (the M tiles of one pixel tile run together; Winograd GEMM)
  unsigned w_img_bytes;  // per-image weight stride in bytes (0: one weight set; Winograd GEMM: one U matrix per "image")
  int full_q;          // whole tiles per workgroup in the data-parallel phase (tile t = g + j * G, j < full_q)
  long total_iters;    // iterations (tile, chunk) of the stream-K phase: the remaining tiles [full_q * G, MT * NT)
  // split-fp16 (X3) kernels: kAmaxSlots partial maxima bounding |x| (scale of the activations), 1 / s_w of the packed weights
  const unsigned* amax_in; const float* w_inv;
  unsigned* amax_out;  // != nullptr: every kernel of the family publishes max |y| into these slots (x3_device.h)
#ifdef MSCNN_WG_TRACE
  unsigned long long* trace;   // debug builds only (make EXTRA=-DMSCNN_WG_TRACE): 16 words per workgroup, see tools/wg_trace.py
#endif
};
#ifdef MSCNN_WG_TRACE
static unsigned long long* g_wg_trace = nullptr;
extern "C" __attribute__((visibility("default"))) void mscnn_debug_set_wg_trace(void* p) { g_wg_trace = static_cast<unsigned long long*>(p); }
#define MSCNN_TRACE_STAMP(slot)                                                                                  \
  if (a.trace && tid == 0 && (slot) < 16) a.trace[(long)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define MSCNN_TRACE_STAMP(slot)
#endif
constexpr int kSlabsPerWg = 2;   // stream-K tail partial, stream-K head partial

// Tile configuration.  Two geometries share one kernel:
//   plane mode (RH == 0): the N side of a tile is a TH x TW patch of one image's output plane (trunk, heads);
//   ROI mode   (RH  > 0): the images are tiny (RH x RW, e.g. the 7x7 ROI-pooled maps of roi_c1) and a tile packs
//                         IPT whole images: N side = IPT * OH * OW output pixels.  The reference's CAFFE engine runs
//                         one im2col+GEMM with N = 25 per ROI here (conv_layer.cu:14-21).
template <int BM_, int BN_, int WGM_, int WGN_, int KH_, int KW_, int CK_, int TW_, int RH_ = 0, int RW_ = 0, int RP_ = 0, int PF_ = 0, int VEC_ = 0, int NOPN_ = 0, int F16_ = 0>
struct Cfg {
  // F16: operands rounded to fp16 while they are staged into LDS, v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate), fp32
  // accumulators; blobs stay fp32 in HBM.  A k-step is 16 input channels of one tap: a lane holds 8 consecutive channels, so
  // both LDS tiles are channel-innermost 16-byte units: A [tap][kg][BM][8], B [kg][patch pixel][8]  (kg = 8-channel group).
  static constexpr bool F16 = F16_ != 0;
  // X3 (F16_ == 2): the F16 kernel with every fp32 operand split exactly into fp16 hi + lo (x3_device.h) and three MFMAs per
  // pair -- fp32-grade results.  Both LDS tiles and the packed weights carry a second, "lo" copy behind the "hi" one.
  static constexpr bool X3 = F16_ == 2;
  static constexpr int PARTS = X3 ? 2 : 1;
  // kernels whose output may feed a split-fp16 layer publish max |y| (IgemmArgs::amax_out): the fp32 and X3 3x3 kernels; not
  // the Winograd GEMM (its output transform does) nor the reduced-precision fp16 mode (registers: the occupancy-bound builds
  // would spill)
  static constexpr bool PUBLISH = VEC_ == 0 && F16_ != 1;
  // PF_ == 3: as PF_ == 1, compiled for 4 workgroups per CU (<= 128 VGPRs; the 1x1 VEC kernel fits: 127, no scratch)
  // PF_ == 4 (fp16 kernels, whose inner loop does not use PF): compiled for 3 workgroups per CU (168 VGPRs, 16 B of scratch)
  // PF_ == 5 (fp32 3x3 kernels): the PF_ == 0 inner loop compiled for 3 workgroups per CU (A/B variant 4 on conv1_2's 64x256 tile)
  static constexpr int MIN_WG_PER_CU = PF_ == 3 ? 4 : (PF_ == 4 || PF_ == 5) ? 3 : 2;
  static constexpr int KG = CK_ / 8;
  // VEC: 1x1 kernel over planes of exactly 128-pixel rows (the Winograd GEMM operands): the B tile is CK contiguous
  // 512-byte rows, staged with b128 loads / ds_write_b128 and no per-element offset table
  static constexpr int VEC = VEC_;
  static constexpr bool PREFETCH_NEXT = VEC_ != 0 && NOPN_ == 0;      // (also tried on the 64x256 conv1_x tiles: 796 vs 775 us, no gain)
  static constexpr int F4_PER_CH = BN_ / 4;                 // float4s per channel row of the B tile (tile = BN contiguous pixels)
  static constexpr int CH_PER_PASS = 256 / F4_PER_CH;       // channels staged per pass of the 256 threads
  static constexpr int BV_PER_T = CK_ * F4_PER_CH / 256;
  static constexpr int PF = PF_;   // 1: LDS operand reads software-pipelined one MFMA group ahead
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, KH = KH_, KW = KW_, CK = CK_, TW = TW_;
  static constexpr bool ROI = RH_ > 0;
  static constexpr int RH = RH_, RW = RW_, RP = RP_;
  static constexpr int TH = BN / TW;
  // plane mode patch
  static constexpr int PH = TH + KH - 1, PW = TW + KW - 1;
  // ROI mode: padded image, output size, images per tile
  static constexpr int IPH = RH + 2 * RP, IPW = RW + 2 * RP;
  static constexpr int OH = ROI ? IPH - KH + 1 : 1, OW = ROI ? IPW - KW + 1 : 1, OPX = OH * OW;
  static constexpr int IPT = ROI ? BN / OPX : 1;
  static constexpr int ROWS = ROI ? IPW : PW;                           // LDS row stride of the patch
  static constexpr int CH_STRIDE = ROI ? IPT * IPH * IPW : PH * PW;     // LDS floats per channel
  static constexpr int TAPS = KH * KW;
  static constexpr int A_ELEMS = TAPS * CK * BM;
  static constexpr int A_VEC4 = (F16_ ? A_ELEMS / 2 * PARTS : A_ELEMS) / 4;      // float4s of one chunk's packed weight slab
  static constexpr int A_PER_T = (A_VEC4 + 255) / 256;
  static constexpr int B_ELEMS = CK * CH_STRIDE;
  static constexpr int B_PER_T = F16_ ? 1 : (B_ELEMS + 255) / 256;
  static constexpr int B_UNITS = (CK_ / 8) * CH_STRIDE;                  // F16: 16-byte units (8 channels of one patch pixel)
  static constexpr int BU_PER_T = F16_ ? (B_UNITS + 255) / 256 : 1;
  static constexpr int A_LDS_FLOATS = F16_ ? A_ELEMS / 2 * PARTS : A_ELEMS, B_LDS_FLOATS = F16_ ? B_ELEMS / 2 * PARTS : B_ELEMS;
  static constexpr int A_UNITS = TAPS * (CK_ / 8) * BM_;                 // F16: 16-byte units of one part of the A tile
  static constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 32, NI = WN / 32;
  static constexpr int FIX_SPLIT = (BM * BN) / 4096;                    // fix-up workgroups per tile
  // fused 2x2 max pooling in the epilogue: a 32-pixel MFMA block is two 16-pixel rows (TW 16) or one row whose partner
  // row is the next block of the same lane (TW 32); tiles start on even rows / columns
  static constexpr bool CAN_POOL = !ROI && (BN / TW) % 2 == 0 && (TW_ == 16 || (TW_ == 32 && (BN / WGN / 32) % 2 == 0));
  static_assert(!VEC_ || (KH_ == 1 && KW_ == 1 && TW_ == 128 && (BN_ == 128 || BN_ == 256) && RH_ == 0 && CK_ % 8 == 0),
                "VEC: 1x1, tiles of one or two whole 128-pixel rows");
  static_assert(!F16_ || (CK_ % 16 == 0 && VEC_ == 0), "F16: k-steps of 16 channels");
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  static_assert(WM % 32 == 0 && WN % 32 == 0 && BN % TW == 0 && CK % 2 == 0 && A_ELEMS % 4 == 0, "tile shape");
  static_assert((BM * BN) % 4096 == 0, "fix-up split");
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

// F16 kernels: w[Cout][Cin][KH][KW] (fp32) -> halves wp[mt][kc][tap][kg][BM][8]: the 16-byte unit (tap, kg, m) holds input
// channels kc * CK + kg * 8 .. + 7 of output channel mt * BM + m -- exactly one lane's MFMA A operand.  Round to nearest even.
__global__ __launch_bounds__(256) void pack_weights_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout,
                                                               int Cin, int taps, int BM, int CK, int MT, int KI) {
  const long total = (long)MT * KI * taps * CK * BM;
  const int KG = CK / 8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i;
    const int e = (int)(r % 8); r /= 8;
    const int m = (int)(r % BM); r /= BM;
    const int kg = (int)(r % KG); r /= KG;
    const int tap = (int)(r % taps); r /= taps;
    const int kc = (int)(r % KI); r /= KI;
    const int mt = (int)r;
    const int co = mt * BM + m, ci = kc * CK + kg * 8 + e;
    wp[i] = (_Float16)((co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * taps + tap] : 0.f);
  }
}

// X3 kernels: the F16 layout twice -- wp[mt][kc][part][tap][kg][BM][8], part 0 = hi, 1 = lo of w * s_w (exact split,
// x3_device.h); s_w = the power of two from max |w| found in the slots at hdr + 4096 bytes, 1 / s_w is left in hdr[0].
__global__ __launch_bounds__(256) void pack_weights_x3_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, float* __restrict__ hdr,
                                                              int Cout, int Cin, int taps, int BM, int CK, int MT, int KI) {
  float s, inv;
  mscnn::pow2_scale(mscnn::bound_from_slots(reinterpret_cast<const unsigned*>(hdr) + 1024), &s, &inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = inv;
  const long total = (long)MT * KI * taps * CK * BM;
  const int KG = CK / 8;
  const long part_stride = (long)taps * CK * BM;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i;
    const int e = (int)(r % 8); r /= 8;
    const int m = (int)(r % BM); r /= BM;
    const int kg = (int)(r % KG); r /= KG;
    const int tap = (int)(r % taps); r /= taps;
    const int kc = (int)(r % KI); r /= KI;
    const int mt = (int)r;
    const int co = mt * BM + m, ci = kc * CK + kg * 8 + e;
    const float v = (co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * taps + tap] : 0.f;
    _Float16 hi, lo;
    mscnn::split16(v * s, &hi, &lo);
    const long chunk = i / part_stride, within = i % part_stride;      // chunk = (mt, kc)
    wp[chunk * 2 * part_stride + within] = hi;
    wp[chunk * 2 * part_stride + part_stride + within] = lo;
  }
}

// Buffer resources (SGPR descriptors): every global access of the kernel is a raw buffer op with a 32-bit
// per-lane offset, so (a) no 64-bit address VGPRs, (b) out-of-range offsets read 0 / drop the store -- that is
// how the zero padding of the input patch, the ragged last tile and Cout < BM are handled without branches.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), /*stride*/ 0, (int)bytes, /*flags*/ 0x00020000);
}
constexpr unsigned kOob = 0x80000000u;   // offset that is out of range for every tensor here (< 2 GiB each)

// Where a tile sits: decoded once per stream-K segment (main kernel) / once per workgroup (fix-up kernel).
template <class C>
struct TileGeo {
  int h0, w0, img;   // plane mode
  int r0;            // ROI mode: first image of the tile

  __device__ __forceinline__ void decode(const IgemmArgs& a, int nt) {
    if (C::ROI) {
      r0 = nt * C::IPT; h0 = w0 = img = 0;
    } else {
      const int tw = nt % a.NTW, th = (nt / a.NTW) % a.NTH;
      img = nt / (a.NTW * a.NTH); h0 = th * C::TH; w0 = tw * C::TW; r0 = 0;
    }
  }
  // x / y buffer windows the 32-bit offsets are relative to
  __device__ __forceinline__ const float* x_base(const IgemmArgs& a) const { return C::ROI ? a.x : a.x + (long)img * a.Cin * a.H * a.W; }
  __device__ __forceinline__ unsigned x_bytes(const IgemmArgs& a) const {
    return (unsigned)(C::ROI ? a.N : 1) * (unsigned)a.Cin * (unsigned)(a.H * a.W) * 4u;
  }
  __device__ __forceinline__ float* y_base(const IgemmArgs& a) const { return C::ROI ? a.y : a.y + (long)img * a.Cout * a.Ho * a.Wo; }
  __device__ __forceinline__ unsigned y_bytes(const IgemmArgs& a) const {
    return (unsigned)(C::ROI ? a.N : 1) * (unsigned)a.Cout * (unsigned)(a.Ho * a.Wo) * 4u;
  }
  // float offset (inside x_base, channel-chunk term excluded) of staging element idx, or -1 for zero fill
  __device__ __forceinline__ int in_off(const IgemmArgs& a, int idx) const {
    const int ck = idx / C::CH_STRIDE, rem = idx % C::CH_STRIDE;
    if (idx >= C::B_ELEMS) return -1;
    if (C::ROI) {
      const int rl = rem / (C::IPH * C::IPW), e = rem % (C::IPH * C::IPW);
      const int ih = e / C::IPW - C::RP, iw = e % C::IPW - C::RP, r = r0 + rl;
      if (r >= a.N || ih < 0 || ih >= C::RH || iw < 0 || iw >= C::RW) return -1;
      return (r * a.Cin + ck) * (C::RH * C::RW) + ih * C::RW + iw;
    }
    const int ih = h0 - a.pad_h + rem / C::PW, iw = w0 - a.pad_w + rem % C::PW;
    if (ih < 0 || ih >= a.H || iw < 0 || iw >= a.W) return -1;
    return ck * (a.H * a.W) + ih * a.W + iw;
  }
  // float offset (inside y_base) of output pixel p of the tile for channel 0, or -1 when p is padding
  __device__ __forceinline__ int out_off(const IgemmArgs& a, int p) const {
    if (C::ROI) {
      const int rl = p / C::OPX, q = p % C::OPX, r = r0 + rl;
      if (rl >= C::IPT || r >= a.N) return -1;
      return r * a.Cout * C::OPX + q;
    }
    const int oh = h0 + p / C::TW, ow = w0 + p % C::TW;
    if (oh >= a.Ho || ow >= a.Wo) return -1;
    return oh * a.Wo + ow;
  }
};

// LDS float offset of output pixel p's top-left input sample inside one channel of the staged patch
template <class C>
__device__ __forceinline__ int lane_patch_off(int p) {
  if (C::ROI) {
    int rl = p / C::OPX;
    const int q = p % C::OPX;
    if (rl >= C::IPT) rl = 0;          // padding lanes read something valid; their results are dropped
    return rl * (C::IPH * C::IPW) + (q / C::OW) * C::IPW + q % C::OW;
  }
  return (p / C::TW) * C::PW + p % C::TW;
}

__device__ __forceinline__ float lane_xor1(float v) {      // DPP quad_perm [1,0,3,2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_xor16(float v) {     // ds_swizzle bit mode: and 0x1f, or 0, xor 0x10
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float max2(float a, float b) { return b > a ? b : a; }
constexpr float kNegMax = -3.402823466e+38f;

template <class C>
__global__ __launch_bounds__(256, C::MIN_WG_PER_CU) void igemm_kernel(IgemmArgs a) {
  __shared__ __attribute__((aligned(16))) float ldsA[C::A_LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float ldsB[C::B_LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;
#ifdef MSCNN_WG_TRACE
  int trace_slot = 2;
  if (a.trace && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.trace[(long)blockIdx.x * 16 + 0] = ((unsigned long long)xcc << 32) | hw;
    a.trace[(long)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime();     // shader clock at entry / exit: the effective clock
  }
  MSCNN_TRACE_STAMP(1);
#endif

  // Hybrid schedule: first full_q whole tiles per workgroup (no partial sums at all), then the remaining tiles are
  // cut stream-K style into G equal (tile, chunk) ranges so that every CU finishes at the same time.
  // XCD-aware workgroup -> range mapping: the dispatcher places workgroup b on XCD b % 8; giving XCD x the contiguous
  // ranges [x * G/8, (x+1) * G/8) keeps one M-tile's packed weights (2.4 MB for conv4) resident in that XCD's 4 MB L2
  // instead of every XCD cycling through all of them (speed only: any mapping is correct).
  // (XCD x holds the workgroups b == x mod 8: G/8 of them, one more for x < G % 8)
  const int xcd = (int)(blockIdx.x % 8), gq = a.G / 8, gr = a.G % 8;
  const int wg = a.xcd_map ? xcd * gq + min(xcd, gr) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  long it, it_end;
  wg_range(a.total_iters, a.G, wg, it, it_end);
  int full_j = 0;
  const int rem_tile0 = a.full_q * a.G;

  // per-lane LDS read bases (floats)
  const float* aRd = ldsA + khalf * C::BM + wm * C::WM + l31;
  const float* bRd[C::NI];
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni) bRd[ni] = ldsB + khalf * C::CH_STRIDE + lane_patch_off<C>(wn * C::WN + ni * 32 + l31);
  float4* aWr = reinterpret_cast<float4*>(ldsA) + tid;

  const int plane = a.H * a.W;
  float x3_s = 1.f, x3_inv = 1.f;
  if constexpr (C::X3) {
    float inv_x;
    mscnn::pow2_scale(mscnn::bound_from_slots(a.amax_in), &x3_s, &inv_x);
    x3_inv = inv_x * a.w_inv[0];
  }
  unsigned am = 0;                                // max |y| (bit pattern) of everything this workgroup stores
  const bool ragged_c = (a.Cin % C::CK) != 0;    // last chunk has fewer than CK real channels (conv1_1: Cin = 3)
  const __amdgpu_buffer_rsrc_t wsrc =
      make_rsrc(a.wp, (unsigned)((long)a.MT * a.KI * C::A_LDS_FLOATS * 4) + (a.w_img_bytes ? (unsigned)(a.N - 1) * a.w_img_bytes : 0u));
  const __amdgpu_buffer_rsrc_t bias_rsrc = make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
  const int co_stride = a.Ho * a.Wo;

  // ---- segment state: the segment whose first chunk is in flight / being multiplied --------------------------------------
  int t = 0, k0 = 0, k1 = 0, mt = 0;
  TileGeo<C> geo;
  __amdgpu_buffer_rsrc_t xsrc = wsrc;
  unsigned g_off[C::B_PER_T];      // byte offsets of this thread's patch elements (channel-chunk term = scalar offset of the load)
  unsigned bv_voff = 0;            // VEC: float4 number tid + i*256 of the tile = channel tid / F4_PER_CH + CH_PER_PASS * i,
                                   // pixels 4 (tid % F4_PER_CH) .. +3 of the tile's BN contiguous pixels
  unsigned a_tile = 0;
  float4 ra[C::A_PER_T];
  float rb[C::B_PER_T];
  float rh[C::BU_PER_T][8];        // F16: the 8 channels of this thread's units (fp32 until they are stored into LDS)
  unsigned u_off[C::BU_PER_T];     // F16: byte offset of unit i's pixel in channel (8 kg) of the chunk, or kOob
  float4 rbv[C::VEC ? C::BV_PER_T : 1];
  const unsigned a_voff = (unsigned)tid * 16u;

  auto next_segment = [&]() -> bool {
    if (full_j < a.full_q) {
      t = wg + full_j * a.G; k0 = 0; k1 = a.KI;
      ++full_j;
    } else if (it < it_end) {
      t = rem_tile0 + (int)(it / a.KI);
      k0 = (int)(it % a.KI);
      k1 = (int)min((long)a.KI, k0 + (it_end - it));
      it += (k1 - k0);
    } else {
      return false;
    }
    mt = a.nt_major ? t % a.MT : t / a.NT;
    const int nt = a.nt_major ? t / a.MT : t % a.NT;
    geo.decode(a, nt);
    xsrc = make_rsrc(geo.x_base(a), geo.x_bytes(a));
    if constexpr (C::F16) {
#pragma unroll
      for (int i = 0; i < C::BU_PER_T; ++i) {
        const int u = tid + i * 256;
        // staging element (kg * 8, pixel): in_off's index is channel * CH_STRIDE + pixel
        const int o = u < C::B_UNITS ? geo.in_off(a, (u / C::CH_STRIDE) * 8 * C::CH_STRIDE + u % C::CH_STRIDE) : -1;
        u_off[i] = o >= 0 ? (unsigned)o * 4u : kOob;
      }
    } else if constexpr (!C::VEC) {
#pragma unroll
      for (int i = 0; i < C::B_PER_T; ++i) {
        const int o = geo.in_off(a, tid + i * 256);
        g_off[i] = o >= 0 ? (unsigned)o * 4u : kOob;
      }
    }
    bv_voff = ((unsigned)(tid / C::F4_PER_CH) * (unsigned)plane + (unsigned)geo.h0 * 128u + (unsigned)(tid % C::F4_PER_CH) * 4u) * 4u;
    a_tile = (unsigned)(mt * a.KI) * (C::A_LDS_FLOATS * 4u) + (unsigned)geo.img * a.w_img_bytes;
    return true;
  };

#define MSCNN_LOAD_CHUNK(kc)                                                                                        \
    {                                                                                                               \
      const unsigned a_soff = a_tile + (unsigned)(kc) * (C::A_LDS_FLOATS * 4u);                                     \
      _Pragma("unroll") for (int i = 0; i < C::A_PER_T; ++i) {                                                      \
        const unsigned vo = (C::A_VEC4 % 256 == 0 || tid + i * 256 < C::A_VEC4) ? a_voff : kOob;                    \
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, vo, a_soff + i * 4096u, 0)); \
      }                                                                                                             \
      const unsigned b_soff = (unsigned)(kc) * (unsigned)(C::CK * 4) * (unsigned)plane;                             \
      const int c_left = a.Cin - (kc) * C::CK;                                                                      \
      if constexpr (C::F16) {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < C::BU_PER_T; ++i)                                                     \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                           \
            unsigned vo = u_off[i];                                                                                 \
            if (ragged_c && ((tid + i * 256) / C::CH_STRIDE) * 8 + j >= c_left) vo = kOob;                          \
            rh[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(                              \
                                               xsrc, vo, b_soff + (unsigned)j * (unsigned)plane * 4u, 0));           \
          }                                                                                                         \
      } else if constexpr (C::VEC) {                                                                                \
        _Pragma("unroll") for (int i = 0; i < C::BV_PER_T; ++i) {                                                   \
          const unsigned vo = (ragged_c && tid / C::F4_PER_CH + C::CH_PER_PASS * i >= c_left) ? kOob : bv_voff;     \
          rbv[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(                                \
                                         xsrc, vo, b_soff + (unsigned)(C::CH_PER_PASS * i) * (unsigned)plane * 4u, 0)); \
        }                                                                                                           \
      } else {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < C::B_PER_T; ++i) {                                                    \
          unsigned vo = g_off[i];                                                                                   \
          if (ragged_c && (tid + i * 256) / C::CH_STRIDE >= c_left) vo = kOob;                                      \
          rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, vo, b_soff, 0));             \
        }                                                                                                           \
      }                                                                                                             \
    }

  bool more = false;
  if constexpr (C::PREFETCH_NEXT) {      // every later segment's first chunk is fetched during the previous segment's last chunk
    more = next_segment();
    if (more) MSCNN_LOAD_CHUNK(k0);
  }
  for (;;) {
    if constexpr (C::PREFETCH_NEXT) {
      if (!more) break;
    } else {
      if (!next_segment()) break;
      MSCNN_LOAD_CHUNK(k0);
    }
    f32x16 acc[C::MI][C::NI];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // This segment's identity (the epilogue needs it after the segment state has moved on).  PREFETCH_NEXT kernels (the 1x1
    // Winograd GEMMs: K = Cin only, so a tile is short and its prologue shows): while the LAST chunk is multiplied, the NEXT
    // segment's first chunk is already put in flight -- the registers ra / rb are idle then -- so its HBM / L2 latency hides
    // behind that chunk's MFMAs and the epilogue's stores.  (Costs registers in the epilogue: for the 3x3 kernels it would
    // take one workgroup of occupancy -- measured slower -- so they fetch it after the epilogue.)
    const TileGeo<C> geo_e = geo;
    const int t_e = t, k0_e = k0, k1_e = k1, mt_e = mt;
    more = false;
#ifdef MSCNN_WG_TRACE
    MSCNN_TRACE_STAMP(trace_slot); ++trace_slot;          // segment start (accumulators zeroed)
#endif
    for (int kc = k0_e; kc < k1_e; ++kc) {
      __syncthreads();                 // everyone finished reading the previous chunk
#pragma unroll
      for (int i = 0; i < C::A_PER_T; ++i)
        if (C::A_VEC4 % 256 == 0 || tid + i * 256 < C::A_VEC4) aWr[i * 256] = ra[i];
      if constexpr (C::F16) {
#pragma unroll
        for (int i = 0; i < C::BU_PER_T; ++i) {
          f16x8 h, l;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if constexpr (C::X3) {
              _Float16 hh, ll;
              mscnn::split16(rh[i][j] * x3_s, &hh, &ll);
              h[j] = hh; l[j] = ll;
            } else {
              h[j] = (_Float16)rh[i][j];            // round to nearest even
            }
          }
          if (C::B_UNITS % 256 == 0 || tid + i * 256 < C::B_UNITS) {
            reinterpret_cast<f16x8*>(ldsB)[tid + i * 256] = h;
            if constexpr (C::X3) reinterpret_cast<f16x8*>(ldsB)[C::B_UNITS + tid + i * 256] = l;
          }
        }
      } else if constexpr (C::VEC) {
#pragma unroll
        for (int i = 0; i < C::BV_PER_T; ++i) reinterpret_cast<float4*>(ldsB)[tid + i * 256] = rbv[i];
      } else {
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i)
          if (C::B_ELEMS % 256 == 0 || tid + i * 256 < C::B_ELEMS) ldsB[tid + i * 256] = rb[i];
      }
      __syncthreads();
      if (kc + 1 < k1_e) {
        MSCNN_LOAD_CHUNK(kc + 1);   // in flight while this chunk is multiplied
      } else if constexpr (C::PREFETCH_NEXT) {
        more = next_segment();                                     // (overwrites t, k0, k1, mt, geo, xsrc, g_off, ...)
        if (more) MSCNN_LOAD_CHUNK(k0);
      }
      if constexpr (C::F16) {
        // unit (16 B) indices: A [tap][kg][BM], B [kg][CH_STRIDE]; this lane's k-group inside a 16-channel step = khalf
        const f16x8* a16 = reinterpret_cast<const f16x8*>(ldsA) + khalf * C::BM + wm * C::WM + l31;
        const f16x8* b16 = reinterpret_cast<const f16x8*>(ldsB) + khalf * C::CH_STRIDE;
        int poff[C::NI];
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) poff[ni] = lane_patch_off<C>(wn * C::WN + ni * 32 + l31);
#pragma unroll
        for (int kh = 0; kh < C::KH; ++kh)
#pragma unroll
          for (int kw = 0; kw < C::KW; ++kw)
#pragma unroll
            for (int ks = 0; ks < C::CK / 16; ++ks) {
              f16x8 av[C::MI], bv[C::NI];
#pragma unroll
              for (int mi = 0; mi < C::MI; ++mi) av[mi] = a16[((kh * C::KW + kw) * C::KG + 2 * ks) * C::BM + mi * 32];
#pragma unroll
              for (int ni = 0; ni < C::NI; ++ni) bv[ni] = b16[2 * ks * C::CH_STRIDE + poff[ni] + kh * C::ROWS + kw];
              if constexpr (C::X3) {
                f16x8 al[C::MI], bl[C::NI];
#pragma unroll
                for (int mi = 0; mi < C::MI; ++mi) al[mi] = a16[C::A_UNITS + ((kh * C::KW + kw) * C::KG + 2 * ks) * C::BM + mi * 32];
#pragma unroll
                for (int ni = 0; ni < C::NI; ++ni) bl[ni] = b16[C::B_UNITS + 2 * ks * C::CH_STRIDE + poff[ni] + kh * C::ROWS + kw];
#pragma unroll
                for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                  for (int ni = 0; ni < C::NI; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bv[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                  }
              }
#pragma unroll
              for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
            }
      } else if constexpr (C::PF == 0 || C::PF == 5) {
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
              for (int ni = 0; ni < C::NI; ++ni) bv[ni] = bRd[ni][cp * 2 * C::CH_STRIDE + kh * C::ROWS + kw];
#pragma unroll
              for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
              // keep the scheduler from hoisting every ds_read of the chunk to the top (register blow-up)
              if (cp % 2 == 1) asm volatile("" ::: "memory");
            }
      } else {
        // software-pipelined operand reads: the ds_reads of MFMA group s+1 are issued before the MFMAs of group s,
        // so a wave never sits on lgkmcnt(0) with an idle matrix pipe
        constexpr int S = C::TAPS * (C::CK / 2);
        float av[2][C::MI], bv[2][C::NI];
#define MSCNN_LDS_GROUP(s, buf)                                                                                  \
        {                                                                                                        \
          constexpr int tap_ = (s) / (C::CK / 2), cp_ = (s) % (C::CK / 2), kh_ = tap_ / C::KW, kw_ = tap_ % C::KW; \
          _Pragma("unroll") for (int mi = 0; mi < C::MI; ++mi) av[buf][mi] = aRd[(tap_ * C::CK + cp_ * 2) * C::BM + mi * 32]; \
          _Pragma("unroll") for (int ni = 0; ni < C::NI; ++ni) bv[buf][ni] = bRd[ni][cp_ * 2 * C::CH_STRIDE + kh_ * C::ROWS + kw_]; \
        }
        MSCNN_LDS_GROUP(0, 0);
        static_for<0, S>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if constexpr (s + 1 < S) MSCNN_LDS_GROUP(s + 1, (s + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);      // pin: next group's ds_reads are issued BEFORE this group's MFMAs
#pragma unroll
          for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][mi], bv[s & 1][ni], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
#undef MSCNN_LDS_GROUP
      }
    }

#ifdef MSCNN_WG_TRACE
    MSCNN_TRACE_STAMP(trace_slot); ++trace_slot;          // K loop done
#endif
    if constexpr (C::X3) {      // un-scale (exact: a power of two) before the epilogue / the partial-sum slab
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= x3_inv;
    }
    {
    if (a.epi_prio) __builtin_amdgcn_s_setprio(3);
    const TileGeo<C>& geo = geo_e;
    const int t = t_e, k0 = k0_e, k1 = k1_e, mt = mt_e;
    (void)t;
    const bool full = (k0 == 0 && k1 == a.KI);
    if (full) {
      const __amdgpu_buffer_rsrc_t ysrc = make_rsrc(geo.y_base(a), geo.y_bytes(a));
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) {
        const int co0 = mt * C::BM + wm * C::WM + mi * 32 + 4 * khalf;
        float bvals[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#ifdef MSCNN_EPI_NOBIAS      // (dev builds: epilogue ablations, tools/sessions/r04_s30.sh)
          bvals[r] = 0.f;
#else
          bvals[r] = C::VEC ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                  bias_rsrc, (unsigned)co0 * 4u, ((r & 3) + 8 * (r >> 2)) * 4u, 0));
#endif
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) {
          const int o = geo.out_off(a, wn * C::WN + ni * 32 + l31);
          // (a.y == nullptr: only the pooled output is wanted -- mscnn_conv2d_plan_can_pool_only; the stores go out of bounds = dropped)
          const unsigned voff = o >= 0 && (!C::CAN_POOL || a.y) ? (unsigned)(co0 * co_stride + o) * 4u : kOob;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = C::VEC ? acc[mi][ni][r] : acc[mi][ni][r] + bvals[r];     // (VEC = Winograd GEMM: no bias, no ReLU)
            if (!C::VEC && a.relu) v = v < 0.f ? 0.f : v;
            const unsigned vo = (co0 + (r & 3) + 8 * (r >> 2) < a.Cout) ? voff : kOob;   // Cout < BM (proposal heads)
            if (!C::CAN_POOL || a.y)      // (pool-only forwards: not even issued -- the epilogue is issue-bound next to the other workgroup's MFMAs)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ysrc, vo,
                                                    (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)co_stride * 4u, 0);
            if constexpr (C::PUBLISH) { if (vo != kOob || (C::CAN_POOL && !a.y && o >= 0)) am = max(am, __builtin_bit_cast(unsigned, v) & 0x7fffffffu); }
            if constexpr (C::CAN_POOL) acc[mi][ni][r] = o >= 0 ? v : kNegMax;     // kept for the pooling pass below
          }
        }
      }
      if constexpr (C::CAN_POOL) {
#ifdef MSCNN_EPI_NOPOOL
        if (false) {
#else
        if (a.yp) {
#endif
          // fused PoolingLayer (MAX, 2x2, stride 2, pooling_layer.cu:11-47): window = rows {2i, 2i+1} x cols {2j, 2j+1};
          // pixels outside the plane were set to -FLT_MAX above (ceil-mode windows at odd edges are clipped)
          const int pstride = a.Hp * a.Wp;
          const __amdgpu_buffer_rsrc_t psrc =
              make_rsrc(a.yp + (long)geo.img * a.Cout * pstride, (unsigned)a.Cout * (unsigned)pstride * 4u);
#pragma unroll
          for (int mi = 0; mi < C::MI; ++mi) {
            const int co0 = mt * C::BM + wm * C::WM + mi * 32 + 4 * khalf;
#pragma unroll
            for (int ni = 0; ni < C::NI; ni += (C::TW == 32 ? 2 : 1)) {
              const int p = wn * C::WN + ni * 32 + l31;
              const int oh = geo.h0 + p / C::TW, ow = geo.w0 + p % C::TW;
              const bool owner = C::TW == 32 ? (l31 & 1) == 0 : (l31 & 17) == 0;
              const unsigned voff = (owner && oh < a.Ho && ow < a.Wo) ? (unsigned)(co0 * pstride + (oh >> 1) * a.Wp + (ow >> 1)) * 4u : kOob;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                float m;
                if constexpr (C::TW == 32) m = max2(acc[mi][ni][r], acc[mi][ni + 1][r]);
                else m = max2(acc[mi][ni][r], lane_xor16(acc[mi][ni][r]));
                m = max2(m, lane_xor1(m));
                const unsigned vo = (co0 + (r & 3) + 8 * (r >> 2) < a.Cout) ? voff : kOob;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), psrc, vo,
                                                      (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)pstride * 4u, 0);
              }
            }
          }
        }
      }
    } else {
      float* slab = a.ws + ((long)wg * kSlabsPerWg + (k0 > 0 ? 0 : 1)) * (C::BM * C::BN);
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
    }
    __syncthreads();   // LDS is re-used by the next segment's first stores
    if (a.epi_prio) __builtin_amdgcn_s_setprio(0);
#ifdef MSCNN_WG_TRACE
    MSCNN_TRACE_STAMP(trace_slot); ++trace_slot;          // epilogue done
#endif
  }
#undef MSCNN_LOAD_CHUNK
#ifdef MSCNN_WG_TRACE
  if (a.trace && tid == 0) a.trace[(long)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime();
#endif
  if constexpr (C::PUBLISH) { if (a.amax_out) mscnn::publish_amax(am, a.amax_out, blockIdx.x); }
}

// The contributors of stream-K tile `tr` (k order), resolved by the whole workgroup: thread 0 finds the first and the last
// contributing workgroup, then the span gf .. gl is walked 256 workgroups at a time, thread i looking at workgroup gf + r + i -- two
// 64-bit divisions each, which one thread looping over the list had turned into microseconds of serial latency (r5).  Workgroups with an
// EMPTY range (fewer stream-K iterations than workgroups: a one- or two-tile remainder on a 512 / 768 grid spreads its KI contributors
// over the whole grid) are compacted away, so the list holds exactly the non-empty contributors, of which there are at most KI <= 256
// (plan_shape) however long the span is (r6: the r5 form cut the SPAN at 256 and lost contributors).
// Returns the length of the list; 0: one workgroup computed the whole tile.  s_hdr: eight ints of LDS.
__device__ __forceinline__ int slab_list(const IgemmArgs& a, int tr, int slab_elems, const float** s_slab, int* s_hdr) {
  const long its = (long)tr * a.KI, ite = its + a.KI;
  if (threadIdx.x == 0) {
    int gf = (int)(its * a.G / a.total_iters), gl = (int)((ite - 1) * a.G / a.total_iters);
    long b, e;
    wg_range(a.total_iters, a.G, gf, b, e);
    while (e <= its) { ++gf; wg_range(a.total_iters, a.G, gf, b, e); }
    while (b > its) { --gf; wg_range(a.total_iters, a.G, gf, b, e); }
    wg_range(a.total_iters, a.G, gl, b, e);
    while (e <= ite - 1) { ++gl; wg_range(a.total_iters, a.G, gl, b, e); }
    while (b > ite - 1) { --gl; wg_range(a.total_iters, a.G, gl, b, e); }
    s_hdr[0] = gf;
    s_hdr[1] = gf == gl ? 0 : gl - gf + 1;
  }
  __syncthreads();
  const int gf = s_hdr[0], span = s_hdr[1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int n = 0;
  for (int r = 0; r < span; r += 256) {
    const float* ptr = nullptr;
    if (r + (int)threadIdx.x < span) {
      const int g = gf + r + threadIdx.x;
      long b, e;
      wg_range(a.total_iters, a.G, g, b, e);
      if (e > b) ptr = a.ws + ((long)g * kSlabsPerWg + (b > its ? 0 : 1)) * slab_elems;
    }
    const unsigned long long m = __ballot(ptr != nullptr);
    if (lane == 0) s_hdr[2 + wave] = __popcll(m);
    __syncthreads();
    int off = n;
    for (int w = 0; w < wave; ++w) off += s_hdr[2 + w];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (ptr && off < 256) s_slab[off] = ptr;
    n += s_hdr[2] + s_hdr[3] + s_hdr[4] + s_hdr[5];
    __syncthreads();
  }
  return min(n, 256);
}

// Sums the partial slabs of every tile that was split across workgroups, in k order (deterministic), + bias + ReLU.
// FIX_SPLIT workgroups per tile, each owning 4096 consecutive slab elements (16 per thread, float4 loads); the list of
// contributing slabs is resolved once per workgroup (the 64-bit range arithmetic is kept out of the element loop).
template <class C>
__global__ __launch_bounds__(256) void igemm_fixup_kernel(IgemmArgs a) {
  __shared__ const float* s_slab[256];   // a tile has at most KI contributors; the plan refuses KI > 256 (plan_shape)
  __shared__ int s_n[8];
  const int tr = blockIdx.x / C::FIX_SPLIT, part = blockIdx.x % C::FIX_SPLIT;   // tr: index among the stream-K tiles
  const int t = a.full_q * a.G + tr;
  const int n = slab_list(a, tr, C::BM * C::BN, s_slab, s_n);
  if (n == 0) return;       // the tile was computed whole by one workgroup and is already in y
  const int mt = a.nt_major ? t % a.MT : t / a.NT;
  const int nt = a.nt_major ? t / a.MT : t % a.NT;
  TileGeo<C> geo;
  geo.decode(a, nt);
  float* ybase = geo.y_base(a);
  const int co_stride = a.Ho * a.Wo;
  unsigned am = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = part * 4096 + (j * 256 + threadIdx.x) * 4;     // 4 consecutive pixels of one output channel row
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < n; s += 4) {                       // four loads in flight, the additions in k order
      float4 u[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const float* sl = s + d < n ? s_slab[s + d] : nullptr;
        u[d] = sl ? *reinterpret_cast<const float4*>(sl + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int d = 0; d < 4; ++d)
        if (s + d < n && s_slab[s + d]) { v.x += u[d].x; v.y += u[d].y; v.z += u[d].z; v.w += u[d].w; }
    }
    const int m = i / C::BN, p = i % C::BN;
    const int co = mt * C::BM + m;
    if (co >= a.Cout) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
    const float vals[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = geo.out_off(a, p + q);
      if (o < 0) continue;
      float r = vals[q] + bv;
      if (a.relu) r = r < 0.f ? 0.f : r;
      ybase[(long)co * co_stride + o] = r;
      am = max(am, __builtin_bit_cast(unsigned, r) & 0x7fffffffu);
    }
  }
  if (a.amax_out) mscnn::publish_amax(am, a.amax_out, 512u + blockIdx.x);
}

// Fix-up with fused 2x2 max pooling: a thread owns one pooling window (4 pixels of one channel) of a split tile.
template <class C>
__global__ __launch_bounds__(256) void igemm_fixup_pool_kernel(IgemmArgs a) {
  __shared__ const float* s_slab[256];
  __shared__ int s_n[8];
  const int tr = blockIdx.x / C::FIX_SPLIT, part = blockIdx.x % C::FIX_SPLIT;
  const int t = a.full_q * a.G + tr;
  const int n = slab_list(a, tr, C::BM * C::BN, s_slab, s_n);
  if (n == 0) return;      // computed whole by one workgroup: y and the pooled output are already written
  if constexpr (C::CAN_POOL) {
    const int mt = a.nt_major ? t % a.MT : t / a.NT;
    const int nt = a.nt_major ? t / a.MT : t % a.NT;
    TileGeo<C> geo;
    geo.decode(a, nt);
    float* ybase = geo.y_base(a);
    const int co_stride = a.Ho * a.Wo, pstride = a.Hp * a.Wp;
    float* pbase = a.yp + (long)geo.img * a.Cout * pstride;
    constexpr int WPT = C::BN / 4;                        // pooling windows per channel row of the tile
    unsigned am = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = j * 256 + threadIdx.x;                // window index inside this workgroup's 4096-element part
      const int m = part * (4096 / C::BN) + q / WPT, pp = q % WPT;
      const int prow = pp / (C::TW / 2), pcol = pp % (C::TW / 2);
      const int i0 = m * C::BN + (2 * prow) * C::TW + 2 * pcol;
      float2 u0 = make_float2(0.f, 0.f), u1 = make_float2(0.f, 0.f);
      for (int s = 0; s < n; ++s) {
        if (!s_slab[s]) continue;
        const float2 w0 = *reinterpret_cast<const float2*>(s_slab[s] + i0);
        const float2 w1 = *reinterpret_cast<const float2*>(s_slab[s] + i0 + C::TW);
        u0.x += w0.x; u0.y += w0.y; u1.x += w1.x; u1.y += w1.y;
      }
      const int co = mt * C::BM + m;
      if (co >= a.Cout) continue;
      const float bv = a.bias ? a.bias[co] : 0.f;
      const int oh = geo.h0 + 2 * prow, ow = geo.w0 + 2 * pcol;
      float vals[4] = {u0.x + bv, u0.y + bv, u1.x + bv, u1.y + bv};
      float mx = kNegMax;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int h = oh + (e >> 1), w = ow + (e & 1);
        if (h >= a.Ho || w >= a.Wo) continue;
        float r = vals[e];
        if (a.relu) r = r < 0.f ? 0.f : r;
        if (a.y) ybase[(long)co * co_stride + h * a.Wo + w] = r;
        mx = max2(mx, r);
        am = max(am, __builtin_bit_cast(unsigned, r) & 0x7fffffffu);
      }
      if (oh < a.Ho && ow < a.Wo) pbase[(long)co * pstride + (oh >> 1) * a.Wp + (ow >> 1)] = mx;
    }
    if (a.amax_out) mscnn::publish_amax(am, a.amax_out, 512u + blockIdx.x);
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
    if (relu) acc = acc < 0.f ? 0.f : acc;
    y[i] = acc;
  }
}

// proposal heads as GEMM + shift-and-add (head_gemm_plan): w [Cout][Cin][taps] -> W' [tap * Cout + co][Cin]
constexpr bool kHeadGemmDefault = false;      // A/B pending: opt-in through tune_flags bit 4
constexpr long kHeadGemmMinPixels = 4096;
__global__ __launch_bounds__(256) void head_gemm_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                               int taps) {
  const long total = (long)taps * Cout * Cin;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cin), row = (int)(i / Cin);
    const int tap = row / Cout, co = row % Cout;
    wp[i] = w[((long)co * Cin + c) * taps + tap];
  }
}

// "kw" form: w [Cout][Cin][KH][KW] -> w' [kw * Cout + co][Cin][KH][1]  (a KH x 1 convolution with KW * Cout output channels)
__global__ __launch_bounds__(256) void head_kw_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                             int KH, int KW) {
  const long total = (long)KW * Cout * Cin * KH;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int kh = (int)(i % KH);
    long r = i / KH;
    const int c = (int)(r % Cin); r /= Cin;
    const int co = (int)(r % Cout), kw = (int)(r / Cout);
    wp[i] = w[(((long)co * Cin + c) * KH + kh) * KW + kw];
  }
}

// ---- kernel table -------------------------------------------------------------------------------------
typedef void (*IgemmFn)(IgemmArgs);
struct KernelEntry {
  const char* name;
  int BM, BN, KH, KW, CK, TW, TH;
  int RH, RW, RP, IPT, fix_split, variant;
  IgemmFn main_fn, fix_fn;
  IgemmFn fix_pool_fn;     // nullptr: this tile geometry has no fused 2x2 max-pooling epilogue
};

#define ENTRY(BM, BN, WGM, WGN, KH, KW, CK, TW)                                                                     \
  {"igemm_" #BM "x" #BN "_k" #KH "x" #KW "_tw" #TW, BM, BN, KH, KW, CK, TW, BN / TW, 0, 0, 0, 1, (BM * BN) / 4096, 0,  \
   igemm_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW>>, igemm_fixup_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW>>,                  \
   Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW>::CAN_POOL ? igemm_fixup_pool_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW>> : nullptr}
#define ENTRY_PF(BM, BN, WGM, WGN, KH, KW, CK, TW)                                                                  \
  {"igemm_" #BM "x" #BN "_k" #KH "x" #KW "_tw" #TW "_pf", BM, BN, KH, KW, CK, TW, BN / TW, 0, 0, 0, 1, (BM * BN) / 4096, 1, \
   igemm_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW, 0, 0, 0, 1>>, igemm_fixup_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW, 0, 0, 0, 1>>, \
   Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW, 0, 0, 0, 1>::CAN_POOL ? igemm_fixup_pool_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, TW, 0, 0, 0, 1>> : nullptr}
// ROI mode entries: images of RH x RW with symmetric pad RP, IPT images per tile
#define ROI_ENTRY(BM, BN, WGM, WGN, KH, KW, CK, RH, RW, RP)                                                             \
  {"igemm_" #BM "x" #BN "_k" #KH "x" #KW "_roi" #RH "x" #RW "p" #RP, BM, BN, KH, KW, CK, 0, 0, RH, RW, RP,              \
   Cfg<BM, BN, WGM, WGN, KH, KW, CK, 32, RH, RW, RP>::IPT, (BM * BN) / 4096, 0,                                         \
   igemm_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, 32, RH, RW, RP>>,                                                     \
   igemm_fixup_kernel<Cfg<BM, BN, WGM, WGN, KH, KW, CK, 32, RH, RW, RP>>}

const KernelEntry kTable[] = {
    // trunk 3x3
    ENTRY(128, 128, 2, 2, 3, 3, 8, 16),
    ENTRY(128, 128, 2, 2, 3, 3, 8, 32),
    ENTRY(64, 256, 1, 4, 3, 3, 8, 32),
    // conv1_1 (Cin = 3): channel chunk of 4 instead of 8 halves the zero-padded K (36 instead of 72 for 27 real taps)
    {"igemm_64x256_k3x3_tw32_ck4", 64, 256, 3, 3, 4, 32, 8, 0, 0, 0, 1, 4, 50, igemm_kernel<Cfg<64, 256, 1, 4, 3, 3, 4, 32>>,
     igemm_fixup_kernel<Cfg<64, 256, 1, 4, 3, 3, 4, 32>>, igemm_fixup_pool_kernel<Cfg<64, 256, 1, 4, 3, 3, 4, 32>>},
    ENTRY_PF(128, 128, 2, 2, 3, 3, 8, 16),
    ENTRY_PF(128, 128, 2, 2, 3, 3, 8, 32),
    ENTRY_PF(64, 256, 1, 4, 3, 3, 8, 32),
    // variant 4 (tune_variant 5): the 64x256 tile compiled for three workgroups per CU (<= 168 VGPRs, 188 B of scratch), grid 768:
    // measured SLOWER on conv1_2 (723 vs 688 us, profiles/r04_ab_conv1_2_occ3.txt) -- A/B variant only
    {"igemm_64x256_k3x3_tw32_occ3", 64, 256, 3, 3, 8, 32, 8, 0, 0, 0, 1, 4, 4, igemm_kernel<Cfg<64, 256, 1, 4, 3, 3, 8, 32, 0, 0, 0, 5>>,
     igemm_fixup_kernel<Cfg<64, 256, 1, 4, 3, 3, 8, 32, 0, 0, 0, 5>>, igemm_fixup_pool_kernel<Cfg<64, 256, 1, 4, 3, 3, 8, 32, 0, 0, 0, 5>>},
    // 1x1 (the 16 batched GEMMs of the Winograd path; plane = [rows][128] so a tile is one 128-pixel row)
    {"igemm_128x128_k1x1_ck32_pf", 128, 128, 1, 1, 32, 128, 1, 0, 0, 0, 1, 4, 0, igemm_kernel<Cfg<128, 128, 2, 2, 1, 1, 32, 128, 0, 0, 0, 1>>,
     igemm_fixup_kernel<Cfg<128, 128, 2, 2, 1, 1, 32, 128, 0, 0, 0, 1>>},
    // variants 101 / 102: vectorised staging, planes of exactly 128-pixel rows only (Winograd GEMM)
    {"igemm_128x128_k1x1_ck32_vec", 128, 128, 1, 1, 32, 128, 1, 0, 0, 0, 1, 4, 101, igemm_kernel<Cfg<128, 128, 2, 2, 1, 1, 32, 128, 0, 0, 0, 1, 1>>,
     igemm_fixup_kernel<Cfg<128, 128, 2, 2, 1, 1, 32, 128, 0, 0, 0, 1, 1>>},
    {"igemm_128x128_k1x1_ck32_vec_occ4", 128, 128, 1, 1, 32, 128, 1, 0, 0, 0, 1, 4, 106, igemm_kernel<Cfg<128, 128, 2, 2, 1, 1, 32, 128, 0, 0, 0, 3, 1>>,
     igemm_fixup_kernel<Cfg<128, 128, 2, 2, 1, 1, 32, 128, 0, 0, 0, 3, 1>>},
    {"igemm_128x128_k1x1_ck64_vec", 128, 128, 1, 1, 64, 128, 1, 0, 0, 0, 1, 4, 102, igemm_kernel<Cfg<128, 128, 2, 2, 1, 1, 64, 128, 0, 0, 0, 1, 1>>,
     igemm_fixup_kernel<Cfg<128, 128, 2, 2, 1, 1, 64, 128, 0, 0, 0, 1, 1>>},
    {"igemm_128x256_k1x1_ck32_vec", 128, 256, 1, 1, 32, 128, 2, 0, 0, 0, 1, 8, 103, igemm_kernel<Cfg<128, 256, 2, 2, 1, 1, 32, 128, 0, 0, 0, 1, 1>>,
     igemm_fixup_kernel<Cfg<128, 256, 2, 2, 1, 1, 32, 128, 0, 0, 0, 1, 1>>},
    ENTRY(128, 256, 2, 2, 3, 3, 8, 32),     // variant 2 (selected with MSCNN_IGEMM_VARIANT=2): 64x128 wave tiles
    // fp16-operand variants (variant 200, mscnn_conv_desc::algo == MSCNN_CONV_ALGO_F16): trunk 3x3 planes and the ROI maps of roi_c1
#define ENTRY16(BM, BN, WGM, WGN, TW)                                                                                        \
  {"igemm16_" #BM "x" #BN "_k3x3_tw" #TW, BM, BN, 3, 3, 16, TW, BN / TW, 0, 0, 0, 1, (BM * BN) / 4096, 200,                    \
   igemm_kernel<Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 0, 0, 0, 1>>, igemm_fixup_kernel<Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 0, 0, 0, 1>>, \
   Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 0, 0, 0, 1>::CAN_POOL ? igemm_fixup_pool_kernel<Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 0, 0, 0, 1>> : nullptr}
#define ROI_ENTRY16(RH, RW, RP)                                                                                              \
  {"igemm16_128x128_k3x3_roi" #RH "x" #RW "p" #RP, 128, 128, 3, 3, 16, 0, 0, RH, RW, RP,                                       \
   Cfg<128, 128, 2, 2, 3, 3, 16, 32, RH, RW, RP, 0, 0, 0, 1>::IPT, 4, 200, igemm_kernel<Cfg<128, 128, 2, 2, 3, 3, 16, 32, RH, RW, RP, 0, 0, 0, 1>>, \
   igemm_fixup_kernel<Cfg<128, 128, 2, 2, 3, 3, 16, 32, RH, RW, RP, 0, 0, 0, 1>>}
    ENTRY16(128, 128, 2, 2, 16),
    ENTRY16(128, 128, 2, 2, 32),
    ENTRY16(64, 256, 1, 4, 32),
#define ENTRY16_OCC3(BM, BN, WGM, WGN, TW)                                                                                   \
  {"igemm16_" #BM "x" #BN "_k3x3_tw" #TW "_occ3", BM, BN, 3, 3, 16, TW, BN / TW, 0, 0, 0, 1, (BM * BN) / 4096, 201,            \
   igemm_kernel<Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 4, 0, 0, 1>>, igemm_fixup_kernel<Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 4, 0, 0, 1>>, \
   Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 4, 0, 0, 1>::CAN_POOL ? igemm_fixup_pool_kernel<Cfg<BM, BN, WGM, WGN, 3, 3, 16, TW, 0, 0, 0, 4, 0, 0, 1>> : nullptr}
    // 128 x 256 tiles = 64 x 128 per wave: 6 operand fragments per 8 MFMAs instead of 4 per 4 (the kernel is LDS-fed)
    {"igemm16_128x256_k3x3_tw32", 128, 256, 3, 3, 16, 32, 8, 0, 0, 0, 1, 8, 202, igemm_kernel<Cfg<128, 256, 2, 2, 3, 3, 16, 32, 0, 0, 0, 0, 0, 0, 1>>,
     igemm_fixup_kernel<Cfg<128, 256, 2, 2, 3, 3, 16, 32, 0, 0, 0, 0, 0, 0, 1>>, igemm_fixup_pool_kernel<Cfg<128, 256, 2, 2, 3, 3, 16, 32, 0, 0, 0, 0, 0, 0, 1>>},
    // split-fp16 (X3, fp32-grade) direct 3x3 kernel for the layers the Winograd heuristic leaves direct (conv1_2, conv2_1):
    // variant 210.  64 x 256 tiles: 59 KB of LDS (hi + lo of both operands), two workgroups per CU
    {"igemm16x3_64x256_k3x3_tw32", 64, 256, 3, 3, 16, 32, 8, 0, 0, 0, 1, 4, 210, igemm_kernel<Cfg<64, 256, 1, 4, 3, 3, 16, 32, 0, 0, 0, 0, 0, 0, 2>>,
     igemm_fixup_kernel<Cfg<64, 256, 1, 4, 3, 3, 16, 32, 0, 0, 0, 0, 0, 0, 2>>, igemm_fixup_pool_kernel<Cfg<64, 256, 1, 4, 3, 3, 16, 32, 0, 0, 0, 0, 0, 0, 2>>},
    ENTRY16_OCC3(128, 128, 2, 2, 16),
    ENTRY16_OCC3(128, 128, 2, 2, 32),
    ENTRY16_OCC3(64, 256, 1, 4, 32),
    ROI_ENTRY16(7, 7, 0),
    ROI_ENTRY16(7, 5, 0),
    ROI_ENTRY16(8, 4, 1),
    // proposal heads (Cout = 4 + classes <= 32): kitti_car 5x5 / 7x7, ped-cyc + caltech 3x5 / 5x7
    ENTRY(32, 128, 1, 4, 5, 5, 8, 16),
    ENTRY(32, 128, 1, 4, 7, 7, 8, 16),
    ENTRY(32, 128, 1, 4, 5, 3, 8, 16),   // "3x5" heads are kernel_w 3 x kernel_h 5
    ENTRY(32, 128, 1, 4, 7, 5, 8, 16),   // "5x7": kernel_w 5 x kernel_h 7
    // proposal heads with the kernel's columns folded into M (head_gemm_plan, "kw" form): KH x 1 taps, Cout' = KW * Cout <= 64 rows
    ENTRY(64, 256, 1, 4, 5, 1, 8, 32),
    ENTRY(64, 256, 1, 4, 7, 1, 8, 32),
    // detection sub-net roi_c1 (3x3 over the ROI-pooled maps): kitti_car 7x7 pad 0, ped/cyc 7x5 pad 0, caltech 8x4 pad 1
    ROI_ENTRY(128, 128, 2, 2, 3, 3, 8, 7, 7, 0),
    ROI_ENTRY(128, 128, 2, 2, 3, 3, 8, 7, 5, 0),
    ROI_ENTRY(128, 128, 2, 2, 3, 3, 8, 8, 4, 1),
};
constexpr int kTableN = sizeof(kTable) / sizeof(kTable[0]);

}  // namespace

struct mscnn_conv_plan {
  mscnn_conv_desc d;
  int Ho, Wo;
  int entry;          // -1: direct
  int MT, NTH, NTW, NT, KI, G, full_q;
  long total_iters;      // stream-K phase iterations
  size_t packed_bytes, ws_bytes;
  mscnn::HeadPlan head;  // head.entry >= 0: the small-Cout kernel family of headconv.hip runs this layer
  // Winograd F(2x2, 3x3) path (wino != nullptr): input transform -> 16 batched 1x1 GEMMs (the nested igemm plan) ->
  // output transform.  Workspace layout: [V: 16 x Cin x T_pad][M: 16 x Cout x T_pad][nested plan's stream-K slabs].
  mscnn_conv_plan* wino = nullptr;
  // use_wg: the plane GEMMs run on the LDS-DMA ring kernel of wgemm.hip instead of the nested igemm plan (which stays as the
  // description of the fall-back; tune_flags bit 7 selects it for A/B runs).  Workspace: [V][M][wgemm's stream-K slabs]
  bool use_wg = false;
  mscnn::WgemmPlan wg;
  bool c3 = false;       // conv1_1: the Cin = 3 VALU kernel of conv_c3.hip (reads the Caffe-layout weights; entry stays -1)
  bool wc_use = false;   // conv1_2's shape class: the ring kernel of wconv.hip runs the layer; `entry` (64 x 256 igemm, SAME weight packing) stays planned beside it
  mscnn::WconvPlan wc;
  // (round 5) the same shape class as ONE-launch Winograd F(2x2,3x3) (wf2conv.hip): its own packed filters (packed_bytes), no workspace;
  // AUTO's choice where it covers the map; tune_flags bit 16 keeps the ring kernel (A/B, the direct witness)
  bool wf_use = false;
  mscnn::Wf2Plan wf;
  int wino_m = 2;        // output tile edge: 2 = F(2x2,3x3) (16 planes), 3 = F(3x3,3x3) (25 planes; small ROI maps)
  int tiles_h = 0, tiles_w = 0, T_pad = 0;
  // split-fp16 form of the F(3x3,3x3) path (MSCNN_CONV_ALGO_WINO_F3_X3, wino_x3.hip): x3.BM > 0, wino == nullptr.
  // Workspace layout: [4 KB: max |x| slots, used when nobody hands the bound over][V16][M: 25 x Cout x T_pad floats]
  mscnn::X3Plan x3;
  mscnn::X3HeadPlan x3h;       // x3h.rows > 0: proposal head as one split-fp16 GEMM + shift-and-add (wino_x3.hip); per image
  // fp32 form of the same idea (hg != nullptr): T[tap * Cout + co][pixel] = W'[tap * Cout + co][c] x[c][pixel] on the nested 1x1
  // igemm plan (one image: the map viewed as HW / 128 rows of 128 pixels where that divides -- the vectorised Winograd GEMM kernel --
  // else as one row of HW pixels), then the shift-and-add.  Packed buffer: [nested pack][W': rows x Cin floats]; workspace: [T][nested]
  mscnn_conv_plan* hg = nullptr;
  int hg_rows = 0;
  int hg_kw = 0;           // > 0: the "kw" form (nested plan = KH x 1 convolution with hg_kw * Cout channels; T = [rows][H][W])
  size_t hg_t_bytes = 0;
  size_t x3d_hdr_off = 0, x3d_slots_off = 0;   // X3 direct kernel: header behind the packed weights, own-amax slots behind the slabs
  const unsigned* amax_in = nullptr;   // mscnn_conv2d_plan_set_amax_io (kept across re-planning)
  unsigned* amax_out = nullptr;
  // roofline accounting (mscnn_conv2d_plan_set_profiling): events around {input transform | GEMM | output transform}
  bool profiling = false;
  mutable hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  mutable bool ev_valid = false;
  ~mscnn_conv_plan() {
    delete hg;
    delete wino;
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
  }
};

using namespace mscnn;

static void plan_shape(mscnn_conv_plan* p);

// Proposal heads (Cout <= 12, K x K over 512 channels) on the fp32 MFMA as ONE dense GEMM over the taps + a shift-and-add: the form
// wino_x3.hip introduced for the split-fp16 mode (M = taps * Cout = 225 / 441 rows instead of 9), here on the nested 1x1 igemm plan.
// The M = 4 head kernel is bound by a per-chunk latency chain (47-65 TFLOP/s on the conv4_3-sized heads and 25 % slower on a box with
// slower clocks); the GEMM runs at the Winograd GEMM's rate and the shift-and-add streams T once.  tune_flags bit 4 forces the form
// wherever it is legal (tests, A/B), bit 5 disables it.
static bool head_gemm_plan(mscnn_conv_plan* p) {
  const mscnn_conv_desc& d = p->d;
  const int flags = tune_env("MSCNN_TUNE_FLAGS", d.tune_flags);
  const long HW = (long)d.H * d.W;
  if (tune_env("MSCNN_CONV_ALGO", d.algo) == MSCNN_CONV_ALGO_WINO_F3_X3 || (flags & 32) || (flags & 2)) return false;
  if (d.stride_h != 1 || d.stride_w != 1 || d.group != 1 || d.Cout > 12 || d.Kh * d.Kw < 2 || d.Kh * d.Kw > 64 || d.N < 1) return false;
  if (d.Cin % 32 != 0 || d.Cin < 32 || p->Ho < 1 || p->Wo < 1) return false;
  // (r3) "kw" form: fold the kernel's COLUMNS into M instead of all its taps.  T[(kw, co)][y][x'] = sum_{c, kh} w[co][c][kh][kw]
  // x[c][y + kh - pad_h][x'] is a KH x 1 convolution with KW * Cout = 45 / 63 output channels -- one 64-row MFMA tile at 70 / 98 %
  // row utilisation, the patch staged once in LDS and reused by the KH taps, K = Cin * KH inside the accumulators -- and
  // y[co][y][x] = bias + sum_kw T[(kw, co)][y][x + kw - pad_w] is a shift-and-add over KW rows of a T that is only 3 - 4 MB (the
  // taps form's T has KH * KW * Cout rows).  Measured against the M = 4 head kernel (profiles/r03_ab_heads_kwfold.txt, us): LFCN_1_7x7
  // (63 rows, 17,280 pixels) 108 vs 130; LFCN_1_5x5 (45 rows) 92 vs 90; on the smaller maps every form sits on the same ~50 us of
  // launch + K-chain latency (LFCN_2_7x7 72 vs 72, LFCN_3_5x5 52 vs 46) -> taken for >= 56 rows on maps of >= 8192 pixels.
  // tune_flags bit 9 disables it, bit 10 forces it wherever it is legal (tests, A/B).
  const bool kw_legal = d.Kw * d.Cout >= 40 && d.Kw * d.Cout <= 64 && d.pad_w * 2 + 1 == d.Kw && p->Ho == d.H && p->Wo == d.W &&
                        (d.Kh == 5 || d.Kh == 7) && HW >= 1024;
  if (!(flags & 16) && !(flags & 512) && kw_legal && ((flags & 1024) || (d.Kw * d.Cout >= 56 && HW >= 8192))) {
    mscnn_conv_plan* g = new (std::nothrow) mscnn_conv_plan();
    if (!g) return false;
    g->d = d;
    g->d.N = 1; g->d.Cout = d.Kw * d.Cout; g->d.Kw = 1; g->d.pad_w = 0; g->d.relu = 0;
    g->d.algo = MSCNN_CONV_ALGO_DIRECT;
    g->d.tune_variant = 0; g->d.tune_grid = 0; g->d.tune_flags = (flags & 1) | 32 | 2;      // (no nested head forms)
    plan_shape(g);
    if (g->entry >= 0 && !g->wino && !g->hg && g->head.entry < 0 && g->Ho == d.H && g->Wo == d.W && kTable[g->entry].KW == 1 &&
        kTable[g->entry].BM == 64) {
      p->hg = g;
      p->hg_kw = d.Kw;
      p->hg_rows = d.Kw * d.Cout;
      p->hg_t_bytes = ((size_t)p->hg_rows * HW * sizeof(float) + 255) / 256 * 256;
      p->packed_bytes = g->packed_bytes + (size_t)p->hg_rows * d.Cin * d.Kh * sizeof(float);
      p->ws_bytes = p->hg_t_bytes + g->ws_bytes;
      return true;
    }
    delete g;
  }
  if (!(flags & 16) && !kHeadGemmDefault) return false;
  if (!(flags & 16) && HW < kHeadGemmMinPixels) return false;    // small maps: the GEMM has too few tiles, the M = 4 kernel is as fast
  const int rows = d.Kh * d.Kw * d.Cout;
  if ((double)rows * HW * 4.0 >= 2.0e9) return false;
  mscnn_conv_plan* g = new (std::nothrow) mscnn_conv_plan();
  if (!g) return false;
  g->d = d;
  g->d.N = 1; g->d.Cout = rows; g->d.Kh = g->d.Kw = 1; g->d.pad_h = g->d.pad_w = 0; g->d.relu = 0;
  g->d.algo = MSCNN_CONV_ALGO_DIRECT;
  g->d.tune_variant = 0; g->d.tune_grid = 0; g->d.tune_flags = flags & 1;
  if (HW % 128 == 0) { g->d.H = (int)(HW / 128); g->d.W = 128; }   // rows of exactly 128 pixels: the vectorised 1x1 kernel
  else { g->d.H = 1; g->d.W = (int)HW; }
  plan_shape(g);
  if (g->entry < 0 || g->wino || g->hg || g->head.entry >= 0 || g->Ho * (long)g->Wo != HW) { delete g; return false; }
  p->hg = g;
  p->hg_kw = 0;
  p->hg_rows = rows;
  p->hg_t_bytes = ((size_t)rows * HW * sizeof(float) + 255) / 256 * 256;
  p->packed_bytes = g->packed_bytes + (size_t)rows * d.Cin * sizeof(float);
  p->ws_bytes = p->hg_t_bytes + g->ws_bytes;
  return true;
}

// Winograd is chosen where the cut in multiplies (2.25x for F(2x2,3x3), 3.24x for F(3x3,3x3)) outweighs the extra HBM traffic
// of the transforms (V and M are 4x / 2.78x the input / output and are written and read once each): GEMM FLOPs per transform
// byte grow with Cin * Cout / (Cin + Cout).  F(2x2,3x3) numbers:  Measured on MI355X: conv2_2 (64) 706 vs 640 us direct, conv3_1 (85) 329 vs 332, conv3_2 (128) 503 vs 614, conv4_1 (171) 250 vs 346,
// conv4_2 (256) 407 vs 617 -> threshold 100.
// desc.algo: DIRECT disables the path, WINO_F2 / WINO_F3 force that form wherever it is legal (tests, A/B runs, and the
// per-layer numerical fall-back of the host runtime: Net::CalibrateNumerics).
static bool wino_plan(mscnn_conv_plan* p) {
  const mscnn_conv_desc& d = p->d;
  const int algo = tune_env("MSCNN_CONV_ALGO", d.algo);
  if (algo == MSCNN_CONV_ALGO_DIRECT || d.Kh != 3 || d.Kw != 3 || d.stride_h != 1 || d.stride_w != 1 || d.group != 1) return false;
  if (p->Ho < 2 || p->Wo < 2) return false;
  const bool want_x3 = algo == MSCNN_CONV_ALGO_WINO_F3_X3;
  if (want_x3 && (d.tune_flags & 8)) return false;        // A/B: the split-fp16 direct kernel instead
  // WINO_F3_X3 follows the AUTO heuristic (it replaces the GEMM of the layers that run F(3x3,3x3) anyway); tune_flags bit 2
  // forces the form wherever it is legal, like WINO_F3 (tests)
  const bool force = algo == MSCNN_CONV_ALGO_WINO_F2 || algo == MSCNN_CONV_ALGO_WINO_F3 || algo == MSCNN_CONV_ALGO_WINO_F4 ||
                     (want_x3 && (d.tune_flags & 4));
  const double intensity = (double)d.Cin * d.Cout / (d.Cin + d.Cout);
  // small maps (the ROI-pooled 7x7 / 7x5 / 8x4 inputs of roi_c1): F(3x3,3x3) -- a 5x5 output is 2x2 tiles x 25 multiplies
  // instead of 225 (measured 1293 -> see DESIGN.md); larger planes: F(2x2,3x3)
  const bool roi_map = d.H <= 8 && d.W <= 8;
  // whole planes: F(3x3,3x3) as well (3.24x fewer multiplies, planes 2.78x the tensor instead of 4x; measured conv3_2 380 vs
  // 481 us with F(2x2,3x3), conv4_2 304 vs 389, conv5_1 107 vs 125, and the same end-to-end error, 3e-5).
  // Threshold for F(3x3,3x3): conv2_2 (intensity 64) 523 vs 642 us direct, conv2_1 (43) 375 vs 383 (tie -> direct), conv1_2
  // (32) 1175 vs 762.  WINO_F2 selects F(2x2,3x3) on planes for A/B runs and tests.
  // F(4x4,3x3) on whole planes (36 planes; wino_f4_math.h; points {0, 1, -1, 2, -1/2, inf}: the fp32 error of the F(3x3,3x3) form within
  // 0.8 .. 3.2x, profiles/r03_robustness.txt): 19 % fewer GEMM FLOPs and plane bytes.  First run on hardware in round 3.  With the
  // round-2 GEMM kernel it only paid where the tile count happened to divide the grid (profiles/r03_ab_wino_f4.txt); with the
  // wgemm kernel's stream-K schedule and the vectorised transforms (profiles/r03_ab_f4_more.txt, 7s-576 layers, us, F(3x3,3x3) /
  // F(4x4,3x3)): conv3_1 214 / 175, conv4_1 161 / 152, conv2_1 330 / 268 (direct: 435), conv2_2 469 / 374, conv3_2 336 / 273,
  // conv4_2 286 / 243 -- but conv5_1 103 / 105 (270 tiles of 4x4 pad to 384: the 128-tile padding eats the gain).  AUTO therefore
  // takes F(4x4,3x3) where the 4x4 tiles number >= 1000; intensity threshold 40 (conv2_1: 43) instead of 60.
  const long T4 = (long)d.N * cdiv(p->Ho, 4) * cdiv(p->Wo, 4);
  const bool auto_f4 = algo == MSCNN_CONV_ALGO_AUTO && !roi_map && T4 >= 1000 && !(d.tune_flags & 64);   // (bit 6: A/B runs keep F(3x3,3x3))
  const int m = roi_map ? 3 : (algo == MSCNN_CONV_ALGO_WINO_F2 ? 2 : (algo == MSCNN_CONV_ALGO_WINO_F4 || auto_f4) ? 4 : 3);
  const int planes = (m + 2) * (m + 2);
  // (split-fp16: the direct kernel runs at ~800 TFLOP/s executed, so Winograd -- HBM-bound on its V / M planes -- only pays from
  // conv3_1 up: measured conv2_2 (64) 415 vs 311 us direct, conv3_1 (85) 159 vs 175, conv3_2 (128) 225 vs 269)
  const double wino_min = want_x3 && d.Cin % 16 == 0 ? 80.0 : (m == 4 ? 40.0 : m == 3 ? 60.0 : 100.0);
  if (!force && (intensity < wino_min || (!roi_map && d.H * d.W < 256))) return false;
  if (roi_map && d.N < 8) return false;
  const int th = cdiv(p->Ho, m), tw = cdiv(p->Wo, m);
  const long T = (long)d.N * th * tw;
  const long T_pad = (T + 127) / 128 * 128;
  if ((double)T_pad * (d.Cin > d.Cout ? d.Cin : d.Cout) * 4.0 >= 2.0e9) return false;   // one transform plane per 32-bit window
  // the split-fp16 GEMM where the caller asked for it and the shape is one the fp32 heuristic sends to F(3x3,3x3) anyway
  if (want_x3 && m == 3 && x3_plan(d.Cin, d.Cout, T_pad, d.tune_variant, &p->x3)) {
    p->wino_m = 3;
    p->tiles_h = th; p->tiles_w = tw; p->T_pad = (int)T_pad;
    p->packed_bytes = p->x3.packed_bytes;
    p->ws_bytes = p->x3.scal_bytes + p->x3.v_bytes + (size_t)25 * d.Cout * T_pad * sizeof(float);
    return true;
  }
  mscnn_conv_plan* g = new (std::nothrow) mscnn_conv_plan();
  if (!g) return false;
  g->d = d;
  if (g->d.tune_variant >= 300) g->d.tune_variant = 0;      // (300 + v addresses the wgemm kernel, not the nested igemm plan)
  g->d.N = planes; g->d.H = (int)(T_pad / 128); g->d.W = 128; g->d.Kh = g->d.Kw = 1; g->d.pad_h = g->d.pad_w = 0; g->d.relu = 0;
  g->d.algo = MSCNN_CONV_ALGO_DIRECT;
  plan_shape(g);
  // the 4-workgroups-per-CU build of the GEMM (127 VGPRs) on a grid of 1000: measured +4..9 % where the 25 planes have >= 3000
  // tiles (conv2_2, conv3_x), -2 % on the 1500-tile layers and roi_c1 (profiles/r02_ab_gemm_occupancy4.txt)
  if (g->entry >= 0 && g->d.tune_variant == 0 && (long)g->MT * g->NT >= 3000 && ((long)g->MT * g->NT) % 1000 == 0) {
    g->d.tune_variant = 107;
    g->d.tune_grid = 1000;
    plan_shape(g);
  }
  if (g->entry < 0 || g->wino || g->head.entry >= 0) { delete g; return false; }
  p->wino_m = m;
  p->wino = g;
  p->tiles_h = th; p->tiles_w = tw; p->T_pad = (int)T_pad;
  p->packed_bytes = (size_t)planes * g->packed_bytes;
  p->ws_bytes = (size_t)planes * ((size_t)d.Cin + d.Cout) * T_pad * sizeof(float) + g->ws_bytes;
  // the plane GEMMs on wgemm.hip's kernel where it covers the shape (F(3x3,3x3) / F(4x4,3x3); Cin % 32 == 0, Cout % 32 == 0)
  p->use_wg = false;
  if (m >= 3 && !(d.tune_flags & 128) && mscnn::wgemm_plan(planes, d.Cout, d.Cin, (int)T, (d.tune_variant >= 300 && d.tune_variant < 1100) ? d.tune_variant - 300 : 0, &p->wg)) {
    p->use_wg = true;
    p->T_pad = p->wg.T_pad;
    p->packed_bytes = p->wg.packed_bytes;
    p->ws_bytes = (size_t)planes * ((size_t)d.Cin + d.Cout) * p->wg.T_pad * sizeof(float) + p->wg.ws_bytes;
  }
  return true;
}

static void plan_shape(mscnn_conv_plan* p) {
  const mscnn_conv_desc& d = p->d;
  p->Ho = (d.H + 2 * d.pad_h - d.Kh) / d.stride_h + 1;     // conv_layer.cpp:8-22
  p->Wo = (d.W + 2 * d.pad_w - d.Kw) / d.stride_w + 1;
  p->entry = -1;
  p->c3 = false;
  p->wc_use = false;
  p->wf_use = false;
  p->packed_bytes = 0;
  p->ws_bytes = 0;
  p->head.entry = -1;
  p->x3h = mscnn::X3HeadPlan();
  delete p->hg;
  p->hg = nullptr;
  p->hg_rows = 0;
  p->hg_kw = 0;
  p->hg_t_bytes = 0;
  // split-fp16 mode: a small-Cout K x K head is ONE dense GEMM over the taps + a shift-and-add (M = taps * Cout instead of Cout)
  if (tune_env("MSCNN_CONV_ALGO", d.algo) == MSCNN_CONV_ALGO_WINO_F3_X3 && d.stride_h == 1 && d.stride_w == 1 && d.group == 1 &&
      d.Cout <= 12 && d.Kh * d.Kw > 1 && d.N > 0 && !(d.tune_flags & 2) &&
      x3_head_plan(d.Cin, d.Cout, d.Kh, d.Kw, (long)d.H * d.W, &p->x3h)) {
    p->packed_bytes = p->x3h.packed_bytes;
    p->ws_bytes = 4096 + p->x3h.t_bytes;
    return;
  }
  if (head_gemm_plan(p)) return;
  if (head_plan(d, p->Ho, p->Wo, &p->head)) {
    p->packed_bytes = p->head.packed_bytes;
    p->ws_bytes = p->head.ws_bytes;
    return;
  }
  delete p->wino;
  p->wino = nullptr;
  p->x3 = mscnn::X3Plan();
  {
    // conv1_1 (Cin = 3): an output-store-bound layer with K = 27 -- its own VALU kernel (conv_c3.hip) unless the caller asked for a
    // Winograd form by name; tune_flags bit 11 keeps the MFMA igemm kernel (A/B runs, the bit-identity test)
    const int algo = tune_env("MSCNN_CONV_ALGO", d.algo);
    const bool named_wino = algo == MSCNN_CONV_ALGO_WINO_F2 || algo == MSCNN_CONV_ALGO_WINO_F3 || algo == MSCNN_CONV_ALGO_WINO_F4;
    if (!named_wino && !(d.tune_flags & 2048) && mscnn::c3_plan(d, p->Ho, p->Wo)) { p->c3 = true; return; }
  }
  if (d.stride_h != 1 || d.stride_w != 1 || d.group != 1 || d.N == 0 || d.Cin > 2048) return;   // KI <= 256
  const bool want16 = tune_env("MSCNN_CONV_ALGO", d.algo) == MSCNN_CONV_ALGO_F16 && d.Kh == 3 && d.Kw == 3;
  if (!want16 && wino_plan(p)) return;
  // WINO_F3_X3 on a layer the Winograd heuristic leaves direct: the split-fp16 direct kernel (conv1_2, conv2_1)
  const bool wantx3 = tune_env("MSCNN_CONV_ALGO", d.algo) == MSCNN_CONV_ALGO_WINO_F3_X3 && d.Kh == 3 && d.Kw == 3 && d.Cin % 16 == 0 &&
                      (long)d.H * d.W >= 4096;
  // 32-bit buffer offsets: every tensor window the kernel addresses must stay below 2 GiB
  const double win_x = (double)d.Cin * d.H * d.W * 4.0, win_y = (double)d.Cout * p->Ho * p->Wo * 4.0;
  // choose the table entry with the least padded work; an ROI-mode entry wins whenever it matches the image shape
  double best = 1e300;
  // tune_variant (value + 1): 0 baseline, 1 pipelined LDS reads, 2 128x256 tiles; 1x1 kernels: 0 generic, 101.. vectorised
  const int tv0 = tune_env("MSCNN_TUNE_VARIANT", d.tune_variant);
  const int tv = tv0 >= 400 ? 0 : tv0;      // (402 addresses wconv.hip's plan, below)
  const bool venv = tv > 0 && d.Kh == 3 && d.Kw == 3;
  // default: pipelined LDS reads for the 128-row tiles (+1..3 % measured on conv2_2..conv4_3), baseline for Cout = 64
  const int want = venv ? tv - 1 : (d.Cout >= 128 ? 1 : 0);
  for (int i = 0; i < kTableN; ++i) {
    const KernelEntry& k = kTable[i];
    if (k.KH != d.Kh || k.KW != d.Kw) continue;
    const bool is16 = k.variant >= 200 && k.variant <= 202;
    if (is16 != want16) continue;
    if ((k.variant == 210) != wantx3) continue;
    if (is16 && k.RH == 0) {
      // 2 workgroups / CU (172 VGPRs, grid 512) or the 3-per-CU build (168 VGPRs + 16 B scratch, grid 768): measured on the
      // 7s-576 trunk (profiles/r02_ab_f16_occ3.txt) the latter wins where there are thousands of tiles (conv1_2 208 -> 195 us,
      // conv2_1 104 -> 83, conv2_2 165 -> 147) and loses on the 540 / 1080-tile layers (conv4_2 134 -> 161).  tune_variant
      // 201 / 202 force one of them.
      const long t16 = (long)cdiv(d.Cout, k.BM) * d.N * cdiv(p->Ho, k.TH) * cdiv(p->Wo, k.TW);
      const int pick = (tv - 1 >= 200 && tv - 1 <= 202) ? tv - 1 : (t16 >= 2000 ? 201 : 200);
      if (k.variant != pick) continue;
    }
    const bool is256 = (k.BM == 128 && k.BN == 256);
    if (k.KH == 3 && k.KW == 3 && k.RH == 0 && !is16 && k.variant != 210) {
      if (is256 && k.variant == 0) continue;                  // (kept for reference; superseded by variant 3)
      if (k.variant != ((d.Cin <= 4 && d.Cout <= 64 && !venv) ? 50 : want)) continue;
    }
    if (k.KH == 1 && k.KW == 1) {
      const bool rows128 = d.W == 128 && d.pad_h == 0 && d.pad_w == 0;        // Winograd GEMM operand planes
      // measured on the F(3x3,3x3) GEMMs of mscnn-7s-576 (25 planes; conv3_2 / conv4_2 / conv5_1, us per layer, G = 512):
      // 128x128 CK 32 (101): 362 / 292 / 100;  128x128 CK 64 (102): 377 / 307 / 107;  128x256 CK 32 (103): 381 / 324 / 105
      const int want1 = !rows128 ? 0 : (tv > 0 ? tv - 1 : 101);
      if (k.variant != want1) continue;
    }
    double cost;
    if (k.RH > 0) {
      if (d.H != k.RH || d.W != k.RW || d.pad_h != k.RP || d.pad_w != k.RP) continue;
      if (win_x * d.N >= 2.0e9 || win_y * d.N >= 2.0e9) continue;
      cost = 0.0;
    } else {
      if (win_x >= 2.0e9 || win_y >= 2.0e9) continue;
      const long mt = cdiv(d.Cout, k.BM), nth = cdiv(p->Ho, k.TH), ntw = cdiv(p->Wo, k.TW);
      cost = (double)mt * k.BM * nth * k.TH * ntw * k.TW;
      cost *= (1.0 + 0.001 * (128.0 / k.BM));   // prefer larger M tiles at equal cost (less re-staging of the patch)
    }
    if (cost < best) { best = cost; p->entry = i; }
  }
  if (p->entry < 0) return;
  const KernelEntry& k = kTable[p->entry];
  p->MT = cdiv(d.Cout, k.BM);
  if (k.RH > 0) {
    p->NTH = p->NTW = 1;
    p->NT = cdiv(d.N, k.IPT);
  } else {
    p->NTH = cdiv(p->Ho, k.TH);
    p->NTW = cdiv(p->Wo, k.TW);
    p->NT = d.N * p->NTH * p->NTW;
  }
  p->KI = cdiv(d.Cin, k.CK);
  if (p->KI > 256) { p->entry = -1; return; }   // the fix-up kernels list at most 256 contributing slabs per tile: use the direct kernel
  const long tiles = (long)p->MT * p->NT;
  // grid: G workgroups (default 2 per CU; 3 for the 128x128 tiles whose 43 KB of LDS and 168 VGPRs allow it)
  const int genv = tune_env("MSCNN_TUNE_GRID", d.tune_grid);   // tuning knob
  long G = genv > 0 ? genv : ((k.BM == 128 && k.BN == 128 && k.KH == 3 && k.variant != 200) || k.variant == 201 || k.variant == 4 ? 768 : 512);   // 1x1 GEMMs: 512 measured best (768: +4 %)
  if (tiles * p->KI / 4 < G) G = tiles * p->KI / 4;         // never less than ~4 chunks per workgroup
  if (G < 1) G = 1;
  // A slightly smaller grid that divides the tile count exactly needs no stream-K phase and no fix-up launch at all (the 25
  // plane GEMMs of conv2_2..conv4_3 have 1500 / 3000 / 6000 tiles: G = 500 instead of 512 saves the 18 us fix-up and the
  // slab traffic for 2 % idle workgroup slots).
  if (!genv && tiles >= 2 * G) {
    // the 1x1 GEMM kernel (32 KB LDS, 166 VGPRs) also fits 3 per CU: try that range first (measured G = 750 vs 500 on the
    // 25-plane GEMMs: conv4_2 259 vs 266 us, conv3_2 332 vs 339, conv2_2 496 vs 510)
    const long tops[3] = {k.variant == 106 ? 1024 : 0, (k.KH == 1 && k.BN == 128 && k.CK < 64) ? 768 : 0, G};
    bool found = false;
    for (int c = 0; c < 3 && !found; ++c)
      // (how many idle slots an exact divisor may cost: 1/16 for the short 1x1 GEMMs, whose fix-up launch is 18 us of a ~200 us
      // layer; 1/50 for the 3x3 kernels -- conv1_2's 4320 tiles: G = 480 measured 698 us, G = 512 + stream-K remainder 680)
      for (long g2 = tops[c]; g2 > 0 && g2 >= tops[c] - tops[c] / (k.KH == 3 ? 50 : 16) && g2 * 2 <= tiles; --g2)
        if (tiles % g2 == 0) { G = g2; found = true; break; }
  }
  // the 25-plane GEMMs of the small layers (conv5_x of 7s-576: 400 tiles on 768 slots): one whole tile per workgroup beats
  // the stream-K split + fix-up launch (78 vs 91 us; profiles/r02_ab_nosplit.txt).  Not so below 256 tiles (conv6_1: 100
  // tiles, 45 vs 40 us) nor for the 3x3 / fp16 kernels (measured slower).
  if (!genv && k.KH == 1 && k.variant >= 101 && tiles >= 256 && tiles <= 768) G = tiles;
  p->G = (int)G;
  p->full_q = (int)(tiles / G);                               // data-parallel phase
  p->total_iters = (tiles - (long)p->full_q * G) * p->KI;     // stream-K phase over the remainder tiles
  p->packed_bytes = (size_t)p->MT * p->KI * k.KH * k.KW * k.CK * k.BM * ((k.variant >= 200 && k.variant <= 202) ? sizeof(_Float16) : sizeof(float));
  p->ws_bytes = (size_t)p->G * kSlabsPerWg * k.BM * k.BN * sizeof(float);
  if (k.variant == 210) {      // hi + lo halves, then [8 KB header: 1 / s_w, slots of max |w|]; workspace tail: slots of max |x|
    p->x3d_hdr_off = (size_t)p->MT * p->KI * k.KH * k.KW * k.CK * k.BM * 2 * sizeof(_Float16);
    p->packed_bytes = p->x3d_hdr_off + 8192;
    p->x3d_slots_off = p->ws_bytes;
    p->ws_bytes += 4096;
  }
  // conv1_2's shape class (3x3 / pad 1, 64 output channels, whole 4 x 128 tiles, at least two of them per CU): the ring kernel of
  // wconv.hip on the SAME packed weights (BM 64, CK 8, one M tile): -50 us per 7s-576 forward against the igemm kernel, alternating
  // in one process (profiles/r04_ab_conv1_2_ring.txt); tune_flags bit 15 keeps the igemm kernel (A/B, second witness), tune_variant
  // 402 also plans maps with fewer than two tiles per CU (tests)
  if (k.KH == 3 && k.KW == 3 && k.BM == 64 && k.CK == 8 && k.RH == 0 && (k.variant == 0 || k.variant == 1) && p->MT == 1 && d.pad_h == 1 &&
      d.pad_w == 1 && !(tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 32768)) {
    if (mscnn::wconv_plan(d.N, d.Cin, d.H, d.W, d.Cout, d.tune_variant == 402, &p->wc) && p->wc.packed_bytes == p->packed_bytes) p->wc_use = true;
  }
  // (round 5) ... and, for AUTO, the one-launch Winograd F(2x2,3x3) kernel of wf2conv.hip where the map is whole 8 x 32 blocks with at
  // least two of them per CU: 2.25x fewer multiplies with nothing but x and the pooled map / y in HBM.  Not under DIRECT (the
  // numerical fall-back and the direct witness of the layers' first-forward check) nor under the fp16 / split-fp16 algorithms.
  if (p->entry >= 0 && k.KH == 3 && k.KW == 3 && k.BM == 64 && k.RH == 0 && (k.variant == 0 || k.variant == 1) && p->MT == 1 && d.pad_h == 1 && d.pad_w == 1 &&
      d.stride_h == 1 && d.stride_w == 1 && d.group == 1 && tune_env("MSCNN_CONV_ALGO", d.algo) == MSCNN_CONV_ALGO_AUTO &&
      !(tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 65536) && mscnn::wf2_plan(d.N, d.Cin, d.H, d.W, d.Cout, &p->wf) &&
      (p->wf.tiles >= 512 || d.tune_variant == 403)) {
    p->wf_use = true;
    p->wc_use = false;
    p->packed_bytes = p->wf.packed_bytes;
    p->ws_bytes = 0;
  }
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
  if (p->x3h.rows) return "head_gemm_shiftadd_x3f16";
  if (p->hg) return p->hg_kw ? "head_kwfold_shiftadd_f32" : "head_gemm_shiftadd_f32";
  if (p->head.entry >= 0) return head_kernel_name(p->head);
  if (p->x3.BM) return p->x3.BM == 256 ? "winograd_f3x3_3x3_x3f16_256" : "winograd_f3x3_3x3_x3f16_128";
  if (p->wino) return p->wino_m == 4 ? "winograd_f4x4_3x3" : p->wino_m == 3 ? "winograd_f3x3_3x3" : "winograd_f2x2_3x3";
  if (p->c3) return mscnn::c3_kernel_name();
  if (p->wf_use) return mscnn::wf2_kernel_name();
  if (p->wc_use) return mscnn::wconv_kernel_name();
  return p->entry < 0 ? "direct_f32" : kTable[p->entry].name;
}
extern "C" unsigned long long mscnn_conv2d_plan_weight_layout(const mscnn_conv_plan* p) {
  if (!p) return 0;
  unsigned long long kind, e, mt, ki;
  if (p->x3h.rows) { kind = 7; e = (unsigned)p->x3h.rows_pad; mt = 0; ki = (unsigned)p->x3h.KG; }
  else if (p->hg) { kind = p->hg_kw ? 10 : 8; e = (unsigned)p->hg->entry; mt = (unsigned)p->hg->MT; ki = (unsigned)p->hg->KI; }
  else if (p->head.entry >= 0) { kind = 2; e = p->head.valu >= 0 ? 100u + (unsigned)p->head.valu : (unsigned)p->head.entry; mt = 0; ki = (unsigned)p->head.KI; }
  else if (p->x3.BM) { kind = 6; e = (unsigned)p->x3.BM; mt = (unsigned)p->x3.MT; ki = (unsigned)p->x3.KG; }
  else if (p->wino && p->use_wg) { kind = p->wino_m == 4 ? 9u : 2 + (unsigned)p->wino_m; e = 200u + (unsigned)(p->wg.BM / 128);      /* the packing depends on BM (and CK = 32) only: the 256-row tile shapes share it */ mt = (unsigned)p->wg.MT; ki = (unsigned)p->wg.KI; }
  else if (p->wino) { kind = p->wino_m == 4 ? 9u : 2 + (unsigned)p->wino_m; e = (unsigned)p->wino->entry; mt = (unsigned)p->wino->MT; ki = (unsigned)p->wino->KI; }
  else if (p->wf_use) { kind = 11; e = 0; mt = 1; ki = (unsigned)p->wf.KI; }
  else if (p->entry >= 0) { kind = 1; e = (unsigned)p->entry; mt = (unsigned)p->MT; ki = (unsigned)p->KI; }   // (entry distinguishes fp16 packs)
  else return 0;   // direct kernel: reads the Caffe layout
  return kind | (e << 8) | (mt << 24) | (ki << 44);
}
extern "C" double mscnn_conv2d_plan_flops(const mscnn_conv_plan* p) {
  if (!p) return 0;
  const mscnn_conv_desc& d = p->d;
  return 2.0 * d.N * d.Cout * p->Ho * p->Wo * (double)(d.Cin / d.group) * d.Kh * d.Kw;
}
extern "C" const char* mscnn_conv2d_plan_dtype(const mscnn_conv_plan* p) {
  if (p && p->x3h.rows) return "f16x3";
  if (p && (p->x3.BM || (!p->wino && p->head.entry < 0 && p->entry >= 0 && kTable[p->entry].variant == 210))) return "f16x3";
  return (p && !p->wino && p->head.entry < 0 && p->entry >= 0 && kTable[p->entry].variant >= 200 && kTable[p->entry].variant <= 202) ? "f16" : "f32";
}
extern "C" double mscnn_conv2d_plan_executed_flops(const mscnn_conv_plan* p) {
  if (!p) return 0;
  if (p->x3h.rows) return 3.0 * mscnn_conv2d_plan_flops(p);
  if (p->wf_use) return mscnn_conv2d_plan_flops(p) * (16.0 / 36.0);      // F(2x2,3x3): 16 multiplies per 4 outputs instead of 36
  if (!p->wino && !p->x3.BM)
    return mscnn_conv2d_plan_flops(p) * ((p->head.entry < 0 && p->entry >= 0 && kTable[p->entry].variant == 210) ? 3.0 : 1.0);
  const mscnn_conv_desc& d = p->d;
  const double planes = (double)((p->wino_m + 2) * (p->wino_m + 2)) * (p->x3.BM ? 3.0 : 1.0);    // x3: three fp16 MFMA products per pair
  return 2.0 * planes * d.Cout * d.Cin * ((double)d.N * p->tiles_h * p->tiles_w);
}
extern "C" int mscnn_conv2d_plan_set_profiling(mscnn_conv_plan* p, int on) {
  MSCNN_REQUIRE(p, "conv plan: null");
  p->profiling = on != 0;
  p->ev_valid = false;
  if (p->profiling)
    for (hipEvent_t& e : p->ev)
      if (!e) MSCNN_HIP_TRY(hipEventCreate(&e));
  return MSCNN_OK;
}
extern "C" int mscnn_conv2d_plan_stage_ms(const mscnn_conv_plan* p, float ms_out[3]) {
  MSCNN_REQUIRE(p && ms_out, "conv plan: null");
  ms_out[0] = ms_out[1] = ms_out[2] = 0.f;
  MSCNN_REQUIRE(p->profiling && p->ev_valid, "conv plan: no profiled forward yet (mscnn_conv2d_plan_set_profiling)");
  MSCNN_HIP_TRY(hipEventSynchronize(p->ev[3]));
  for (int i = 0; i < 3; ++i) MSCNN_HIP_TRY(hipEventElapsedTime(&ms_out[i], p->ev[i], p->ev[i + 1]));
  return MSCNN_OK;
}
extern "C" int mscnn_conv2d_plan_publishes_amax(const mscnn_conv_plan* p) {
  // the F(3x3,3x3) output transforms, every kernel of the implicit-GEMM family (main + fix-up) and the Cin = 3 VALU kernel publish
  // (not the ring kernel of wconv.hip: a split-fp16 consumer of conv1_2 measures its bottom itself)
  return p && !p->x3h.rows && (p->x3.BM || (p->wino && p->wino_m >= 3) || p->c3 ||
               (!p->wino && p->head.entry < 0 && p->entry >= 0 && !p->wc_use && !p->wf_use && !(kTable[p->entry].variant >= 200 && kTable[p->entry].variant <= 202))) ? 1 : 0;
}
extern "C" int mscnn_conv2d_plan_set_amax_io(mscnn_conv_plan* p, const uint32_t* in_bound, uint32_t* out_amax) {
  MSCNN_REQUIRE(p, "conv plan: null");
  if (out_amax && !mscnn_conv2d_plan_publishes_amax(p)) {
    set_error("conv plan: kernel %s does not publish max |y|", mscnn_conv2d_plan_kernel(p));
    return MSCNN_ERR_UNSUPPORTED;
  }
  p->amax_in = in_bound;
  p->amax_out = out_amax;
  return MSCNN_OK;
}
extern "C" int mscnn_conv2d_plan_set_batch(mscnn_conv_plan* p, int N) {
  MSCNN_REQUIRE(p && N >= 0, "conv plan: bad batch");
  p->d.N = N;
  plan_shape(p);
  return MSCNN_OK;
}

extern "C" int mscnn_conv2d_pack_weights(const mscnn_conv_plan* p, const float* w, float* packed, void* stream) {
  MSCNN_REQUIRE(p, "conv pack: null plan");
  if (p->x3h.rows) {
    MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
    return x3_head_pack(p->x3h, w, packed, as_stream(stream));
  }
  if (p->hg) {
    MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
    float* wprime = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(packed) + p->hg->packed_bytes);
    if (p->hg_kw) {
      const long tot = (long)p->hg_rows * p->d.Cin * p->d.Kh;
      head_kw_weight_kernel<<<(int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256), 256, 0, as_stream(stream)>>>(
          w, wprime, p->d.Cout, p->d.Cin, p->d.Kh, p->d.Kw);
      MSCNN_POST_LAUNCH();
      return mscnn_conv2d_pack_weights(p->hg, wprime, packed, stream);
    }
    const long total = (long)p->hg_rows * p->d.Cin;
    head_gemm_weight_kernel<<<(int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256), 256, 0, as_stream(stream)>>>(
        w, wprime, p->d.Cout, p->d.Cin, p->d.Kh * p->d.Kw);
    MSCNN_POST_LAUNCH();
    return mscnn_conv2d_pack_weights(p->hg, wprime, packed, stream);
  }
  if (p->head.entry >= 0) {
    MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
    return head_pack(p->d, p->head, w, packed, as_stream(stream));
  }
  if (p->x3.BM) {
    MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
    return x3_pack_weights(p->x3, w, packed, as_stream(stream));
  }
  if (p->wino) {
    MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
    if (p->use_wg) return wino_pack_weights(p->wino_m, w, packed, p->d.Cout, p->d.Cin, p->wg.BM, p->wg.CK, p->wg.MT, p->wg.KI, as_stream(stream));
    const KernelEntry& k = kTable[p->wino->entry];
    return wino_pack_weights(p->wino_m, w, packed, p->d.Cout, p->d.Cin, k.BM, k.CK, p->wino->MT, p->wino->KI, as_stream(stream));
  }
  if (p->entry < 0) return MSCNN_OK;   // direct kernel reads the Caffe layout
  MSCNN_REQUIRE(w && packed, "conv pack: null pointer");
  if (p->wf_use) return mscnn::wf2_pack(p->wf, w, packed, as_stream(stream));
  const KernelEntry& k = kTable[p->entry];
  const long total = (long)p->MT * p->KI * k.KH * k.KW * k.CK * k.BM;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (k.variant == 210) {
    unsigned char* pk = reinterpret_cast<unsigned char*>(packed);
    float* hdr = reinterpret_cast<float*>(pk + p->x3d_hdr_off);
    const int rc = x3_amax(w, (long)p->d.Cout * p->d.Cin * k.KH * k.KW, reinterpret_cast<unsigned*>(hdr) + 1024, as_stream(stream));
    if (rc != MSCNN_OK) return rc;
    pack_weights_x3_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, reinterpret_cast<_Float16*>(packed), hdr, p->d.Cout, p->d.Cin,
                                                                      k.KH * k.KW, k.BM, k.CK, p->MT, p->KI);
  } else if (k.variant >= 200 && k.variant <= 202)
    pack_weights_f16_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, reinterpret_cast<_Float16*>(packed), p->d.Cout, p->d.Cin,
                                                                       k.KH * k.KW, k.BM, k.CK, p->MT, p->KI);
  else
    pack_weights_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, packed, p->d.Cout, p->d.Cin, k.KH * k.KW, k.BM, k.CK,
                                                                   p->MT, p->KI);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

// igemm launch (main kernel + fix-up).  w_img_bytes / nt_major: see IgemmArgs (non-zero only for the Winograd GEMM).
static int launch_igemm(const mscnn_conv_plan* p, const float* x, const float* packed, const float* bias, float* y,
                        float* y_pool, void* workspace, size_t workspace_bytes, hipStream_t st, unsigned w_img_bytes,
                        int nt_major) {
  const mscnn_conv_desc& d = p->d;
  const KernelEntry& k = kTable[p->entry];
  const long rem_tiles = (long)p->MT * p->NT - (long)p->full_q * p->G;
  const bool split = rem_tiles > 0;
  if (split) {
    if (!workspace || workspace_bytes < p->ws_bytes) {
      set_error("conv: workspace %zu < %zu", workspace_bytes, p->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
  }
  IgemmArgs a;
  a.x = x; a.wp = packed; a.bias = bias; a.y = y; a.ws = static_cast<float*>(workspace);
  a.N = d.N; a.Cin = d.Cin; a.H = d.H; a.W = d.W; a.Cout = d.Cout; a.Ho = p->Ho; a.Wo = p->Wo; a.pad_h = d.pad_h; a.pad_w = d.pad_w;
  a.MT = p->MT; a.NTH = p->NTH; a.NTW = p->NTW; a.NT = p->NT; a.KI = p->KI; a.G = p->G; a.relu = d.relu;
  a.total_iters = p->total_iters; a.full_q = p->full_q;
  a.w_img_bytes = w_img_bytes; a.nt_major = nt_major;
  a.yp = y_pool; a.Hp = (p->Ho + 1) / 2; a.Wp = (p->Wo + 1) / 2;
  if (y_pool && !k.fix_pool_fn) {
    set_error("conv: kernel %s has no fused pooling epilogue", k.name);
    return MSCNN_ERR_BAD_ARG;
  }
  a.xcd_map = (tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 1) ? 0 : 1;
  // the epilogue of a tile at wave priority 3: it is ~600 VALU / store instructions that otherwise take turns with the co-resident
  // workgroup's MFMAs (10 - 22 us per tile in the phase timeline of conv1_2, during which that other workgroup alone feeds the matrix
  // pipe at ~58 %).  Measured (profiles/r04_ab_conv1_2_epilogue.txt): conv1_2 693 -> 674 us; neutral on the k7x1 head GEMM, + 2.5 % on
  // the 128 x 128 tiles -> on for the 64 x 256 3x3 kernel only; tune_flags bit 14 inverts the choice (A/B)
  a.epi_prio = ((k.BM == 64 && k.BN == 256 && k.KH == 3 && k.KW == 3) ? 1 : 0) ^ ((tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 16384) ? 1 : 0);
#ifdef MSCNN_WG_TRACE
  a.trace = g_wg_trace;
#endif
  a.amax_in = nullptr; a.w_inv = nullptr;
  a.amax_out = nt_major ? nullptr : p->amax_out;          // (nt_major: the nested Winograd GEMM -- its output transform publishes)
  if (k.variant == 210) {
    if (!workspace || workspace_bytes < p->ws_bytes) {
      set_error("conv(x3): workspace %zu < %zu", workspace_bytes, p->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
    a.w_inv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(packed) + p->x3d_hdr_off);
    a.amax_in = p->amax_in;
    if (!a.amax_in) {      // nobody handed max |x| over: one streaming pass
      unsigned* slots = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(workspace) + p->x3d_slots_off);
      const int rc = x3_amax(x, (long)d.N * d.Cin * d.H * d.W, slots, st);
      if (rc != MSCNN_OK) return rc;
      a.amax_in = slots;
    }
  }
  k.main_fn<<<p->G, 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  if (split) {
    (y_pool ? k.fix_pool_fn : k.fix_fn)<<<(int)rem_tiles * k.fix_split, 256, 0, st>>>(a);
    MSCNN_POST_LAUNCH();
  }
  return MSCNN_OK;
}

extern "C" int mscnn_conv2d_plan_can_pool(const mscnn_conv_plan* p) {
  if (!p || p->head.entry >= 0 || p->x3h.rows || p->hg) return 0;
  if (p->wino || p->x3.BM) return p->wino_m == 2 || p->wino_m == 4 || (p->tiles_h % 2 == 0 && p->tiles_w % 2 == 0 && p->d.H > 8);
  return p->entry >= 0 && kTable[p->entry].fix_pool_fn != nullptr;
}

static int conv_forward_single(const mscnn_conv_plan* p, const float* x, const float* w, const float* packed, const float* bias,
                               float* y, float* y_pool, void* workspace, size_t workspace_bytes, hipStream_t st);

extern "C" int mscnn_conv2d_fwd_f32(const mscnn_conv_plan* p, const float* x, const float* w, const float* packed,
                                    const float* bias, float* y, void* workspace, size_t workspace_bytes, void* stream) {
  return mscnn_conv2d_fwd_pool_f32(p, x, w, packed, bias, y, nullptr, workspace, workspace_bytes, stream);
}

// The fp32 Winograd path: {input stage -> V | plane GEMM V -> M | output stage M -> y}; the input stage is the plan's own
// transform of x, the fused ROI pooling (mscnn_conv2d_fwd_roipool_pair_f32) or nothing (planes prepared by the previous layer of a
// chain); the output stage the plan's own transform or the one that writes the next layer's planes (mscnn_conv2d_fwd_chain_f32).
template <class InputStage, class OutputStage>
static int wino_forward_stages(const mscnn_conv_plan* p, const float* packed, void* workspace, size_t workspace_bytes, hipStream_t st,
                               InputStage&& input_stage, OutputStage&& output_stage) {
  const mscnn_conv_desc& d = p->d;
#define MSCNN_STAGE_EVENT(i) do { if (p->profiling) MSCNN_HIP_TRY(hipEventRecord(p->ev[i], st)); } while (0)
  MSCNN_REQUIRE(packed, "conv: Winograd path needs packed weights (mscnn_conv2d_pack_weights)");
  if (!workspace || workspace_bytes < p->ws_bytes) {
    set_error("conv(winograd): workspace %zu < %zu", workspace_bytes, p->ws_bytes);
    return MSCNN_ERR_WORKSPACE;
  }
  const mscnn_conv_plan* g = p->wino;
  const size_t planes = (size_t)(p->wino_m + 2) * (p->wino_m + 2);
  float* V = static_cast<float*>(workspace);
  float* M = V + planes * d.Cin * p->T_pad;
  float* gws = M + planes * d.Cout * p->T_pad;
  MSCNN_STAGE_EVENT(0);
  int rc = input_stage(V);
  if (rc != MSCNN_OK) return rc;
  MSCNN_STAGE_EVENT(1);
  if (p->use_wg) rc = mscnn::wgemm_launch(p->wg, packed, V, M, gws, st);
  else rc = launch_igemm(g, V, packed, nullptr, M, nullptr, gws, g->ws_bytes, st, (unsigned)g->packed_bytes, 1);
  if (rc != MSCNN_OK) return rc;
  MSCNN_STAGE_EVENT(2);
  rc = output_stage(M);
  if (rc != MSCNN_OK) return rc;
  MSCNN_STAGE_EVENT(3);
  p->ev_valid = p->profiling;
  return MSCNN_OK;
#undef MSCNN_STAGE_EVENT
}

template <class InputStage>
static int wino_forward(const mscnn_conv_plan* p, const float* packed, const float* bias, float* y, float* y_pool, void* workspace,
                        size_t workspace_bytes, hipStream_t st, InputStage&& input_stage) {
  const mscnn_conv_desc& d = p->d;
  return wino_forward_stages(p, packed, workspace, workspace_bytes, st, input_stage, [&](const float* M) {
    return wino_output_transform(p->wino_m, M, bias, y, y_pool, d.N, d.Cout, p->Ho, p->Wo, p->tiles_h, p->tiles_w, p->T_pad, d.relu, st,
                                 p->wino_m >= 3 ? p->amax_out : nullptr, (d.tune_flags & 256) != 0);
  });
}

extern "C" int mscnn_conv2d_fwd_pool_f32(const mscnn_conv_plan* p, const float* x, const float* w, const float* packed,
                                         const float* bias, float* y, float* y_pool, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  MSCNN_REQUIRE(p, "conv: null plan");
  MSCNN_REQUIRE(!y_pool || mscnn_conv2d_plan_can_pool(p), "conv: this plan has no fused 2x2 max-pooling epilogue");
  const mscnn_conv_desc& d = p->d;
  if (d.N == 0) return MSCNN_OK;
  MSCNN_REQUIRE(x && (y || (y_pool && mscnn_conv2d_plan_can_pool_only(p))),
                "conv: null pointer (y may be NULL only with y_pool on a plan where mscnn_conv2d_plan_can_pool_only)");
  hipStream_t st = as_stream(stream);
#define MSCNN_STAGE_EVENT(i) do { if (p->profiling) MSCNN_HIP_TRY(hipEventRecord(p->ev[i], st)); } while (0)
  if (p->x3.BM) {        // split-fp16 Winograd: {amax + input transform | GEMM | output transform}
    MSCNN_REQUIRE(packed, "conv: Winograd path needs packed weights (mscnn_conv2d_pack_weights)");
    if (!workspace || workspace_bytes < p->ws_bytes) {
      set_error("conv(winograd x3): workspace %zu < %zu", workspace_bytes, p->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
    unsigned char* wsb = static_cast<unsigned char*>(workspace);
    const unsigned* scal = p->amax_in ? p->amax_in : reinterpret_cast<unsigned*>(wsb);
    void* V16 = wsb + p->x3.scal_bytes;
    float* M = reinterpret_cast<float*>(wsb + p->x3.scal_bytes + p->x3.v_bytes);
    MSCNN_STAGE_EVENT(0);
    int rc = p->amax_in ? MSCNN_OK : x3_amax(x, (long)d.N * d.Cin * d.H * d.W, reinterpret_cast<unsigned*>(wsb), st);
    if (rc != MSCNN_OK) return rc;
    rc = x3_input_transform(p->x3, x, V16, scal, d.N, d.H, d.W, d.pad_h, d.pad_w, p->tiles_h, p->tiles_w, st);
    if (rc != MSCNN_OK) return rc;
    MSCNN_STAGE_EVENT(1);
    rc = x3_gemm(p->x3, packed, V16, M, scal, (tune_env("MSCNN_TUNE_FLAGS", d.tune_flags) & 1) ? 0 : 1, st);
    if (rc != MSCNN_OK) return rc;
    MSCNN_STAGE_EVENT(2);
    rc = wino_output_transform(3, M, bias, y, y_pool, d.N, d.Cout, p->Ho, p->Wo, p->tiles_h, p->tiles_w, p->T_pad, d.relu, st,
                               p->amax_out);
    if (rc != MSCNN_OK) return rc;
    MSCNN_STAGE_EVENT(3);
    p->ev_valid = p->profiling;
    return MSCNN_OK;
  }
  if (!p->wino) {        // one-stage kernels: {0, total, 0}
    MSCNN_STAGE_EVENT(0); MSCNN_STAGE_EVENT(1);
    const int rc = conv_forward_single(p, x, w, packed, bias, y, y_pool, workspace, workspace_bytes, st);
    if (rc != MSCNN_OK) return rc;
    MSCNN_STAGE_EVENT(2); MSCNN_STAGE_EVENT(3);
    p->ev_valid = p->profiling;
    return MSCNN_OK;
  }
  return wino_forward(p, packed, bias, y, y_pool, workspace, workspace_bytes, st, [&](float* V) {
    return wino_input_transform(p->wino_m, x, V, d.N, d.Cin, d.H, d.W, d.pad_h, d.pad_w, p->tiles_h, p->tiles_w, p->T_pad, st,
                                (d.tune_flags & 256) != 0, (d.tune_flags & 4096) != 0);
  });
#undef MSCNN_STAGE_EVENT
}


// ---- chains of same-resolution F(4x4,3x3) layers ---------------------------------------------------------------------------------
static bool is_f4_plane_path(const mscnn_conv_plan* p) { return p && p->wino && p->wino_m == 4 && !p->x3.BM && p->d.N > 0; }
extern "C" int mscnn_conv2d_plan_can_chain(const mscnn_conv_plan* p, const mscnn_conv_plan* next) {
  if (!is_f4_plane_path(p) || !is_f4_plane_path(next)) return 0;
  const mscnn_conv_desc &a = p->d, &b = next->d;
  return a.N == b.N && a.Cout == b.Cin && p->Ho == b.H && p->Wo == b.W && b.Kh == 3 && b.Kw == 3 && b.pad_h == 1 && b.pad_w == 1 &&
                 b.stride_h == 1 && b.stride_w == 1 && p->tiles_h == next->tiles_h && p->tiles_w == next->tiles_w &&
                 !((a.tune_flags | b.tune_flags) & 256) && mscnn::wino44_outin_supported(p->Ho, p->Wo, p->tiles_h, p->tiles_w)
             ? 1 : 0;
}
extern "C" int mscnn_conv2d_plan_can_pool_only(const mscnn_conv_plan* p) {
  if (!p || !mscnn_conv2d_plan_can_pool(p)) return 0;
  if (is_f4_plane_path(p)) return p->Wo % 4 == 0 && p->tiles_w * 4 == p->Wo && !(p->d.tune_flags & 256) ? 1 : 0;
  // the direct MFMA kernels with the pooling epilogue (conv1_2): y stores are predicated off
  return !p->wino && !p->x3.BM && !p->x3h.rows && !p->hg && p->head.entry < 0 && !p->c3 && p->entry >= 0 ? 1 : 0;
}
extern "C" int mscnn_conv2d_fwd_chain_f32(const mscnn_conv_plan* p, const mscnn_conv_plan* next, const float* x, const float* packed,
                                          const float* bias, float* y, float* y_pool, void* workspace, size_t workspace_bytes,
                                          void* next_workspace, size_t next_workspace_bytes, void* stream) {
  MSCNN_REQUIRE(p, "conv: null plan");
  MSCNN_REQUIRE(is_f4_plane_path(p), "conv(chain): the plan does not take the fp32 F(4x4,3x3) path");
  MSCNN_REQUIRE(!next || mscnn_conv2d_plan_can_chain(p, next), "conv(chain): these two plans do not chain (mscnn_conv2d_plan_can_chain)");
  MSCNN_REQUIRE(next ? !y_pool : (y != nullptr || (y_pool && mscnn_conv2d_plan_can_pool_only(p))),
                "conv(chain): y_pool only without a next plan; y may be NULL only with one, or with y_pool where mscnn_conv2d_plan_can_pool_only");
  MSCNN_REQUIRE(!y_pool || mscnn_conv2d_plan_can_pool(p), "conv: this plan has no fused 2x2 max-pooling epilogue");
  const mscnn_conv_desc& d = p->d;
  hipStream_t st = as_stream(stream);
  if (next) {
    const size_t vbytes = sizeof(float) * 36 * (size_t)next->d.Cin * next->T_pad;
    if (!next_workspace || next_workspace_bytes < next->ws_bytes) {
      set_error("conv(chain): next workspace %zu < %zu", next_workspace_bytes, next->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
    // the planes written for `next` must not touch this plan's own workspace (V is dead by then, M is being read)
    const unsigned char *a0 = static_cast<const unsigned char*>(workspace), *b0 = static_cast<const unsigned char*>(next_workspace);
    MSCNN_REQUIRE(b0 + vbytes <= a0 || a0 + p->ws_bytes <= b0, "conv(chain): the next plan's planes overlap this plan's workspace");
  }
  auto input_stage = [&](float* V) {
    if (!x) return (int)MSCNN_OK;      // prepared by the previous layer of the chain
    return wino_input_transform(p->wino_m, x, V, d.N, d.Cin, d.H, d.W, d.pad_h, d.pad_w, p->tiles_h, p->tiles_w, p->T_pad, st,
                                (d.tune_flags & 256) != 0, (d.tune_flags & 4096) != 0);
  };
  if (!next) return wino_forward(p, packed, bias, y, y_pool, workspace, workspace_bytes, st, input_stage);
  return wino_forward_stages(p, packed, workspace, workspace_bytes, st, input_stage, [&](const float* M) {
    return mscnn::wino44_output_into_input(M, bias, y, static_cast<float*>(next_workspace), d.N, d.Cout, p->Ho, p->Wo, p->tiles_h, p->tiles_w,
                                           p->T_pad, next->T_pad, d.relu, st, p->amax_out);
  });
}

extern "C" int mscnn_conv2d_plan_can_fuse_roipool(const mscnn_conv_plan* p, int C, int pooled_h, int pooled_w) {
  return p && p->wino && p->wino_m == 3 && !p->x3.BM && p->d.N > 0 && p->d.Cin == 2 * C && p->d.H == pooled_h && p->d.W == pooled_w &&
                 p->tiles_h == 2 && p->tiles_w == 2 && p->d.Kh == 3 && p->d.Kw == 3 &&
                 mscnn::roipool_wino33_supported(C, pooled_h, pooled_w, p->d.pad_h, p->d.pad_w)
             ? 1 : 0;
}
static size_t roipool_scratch_offset(const mscnn_conv_plan* p) { return (p->ws_bytes + 255) / 256 * 256; }
extern "C" size_t mscnn_conv2d_roipool_workspace_bytes(const mscnn_conv_plan* p, int N, int C, int H, int W) {
  return p ? roipool_scratch_offset(p) + mscnn::roipool_wino33_scratch_bytes(N, C, H, W) : 0;
}
extern "C" size_t mscnn_roipool_maps_bytes(int N, int C, int H, int W) { return mscnn::roipool_wino33_scratch_bytes(N, C, H, W); }
extern "C" int mscnn_roipool_maps_build_f32(const float* feat, float* maps, int N, int C, int H, int W, void* stream) {
  return mscnn::roipool_wino33_build_maps(feat, maps, N, C, H, W, as_stream(stream));
}
extern "C" int mscnn_conv2d_fwd_roipool_pair_f32(const mscnn_conv_plan* p, const float* feat, const float* prepared_maps, int N, int C,
                                                 int H, int W, const float* rois, float spatial_scale, float pad_ratio_a, float pad_ratio_b,
                                                 const float* packed, const float* bias, float* y, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(p, "conv: null plan");
  if (p->d.N == 0) return MSCNN_OK;
  MSCNN_REQUIRE((feat || prepared_maps) && rois && y && N > 0 && H > 0 && W > 0, "conv(roipool): bad argument");
  MSCNN_REQUIRE(mscnn_conv2d_plan_can_fuse_roipool(p, C, p->d.H, p->d.W), "conv(roipool): this plan does not take the fused ROI-pooling input stage");
  const size_t need = prepared_maps ? p->ws_bytes : mscnn_conv2d_roipool_workspace_bytes(p, N, C, H, W);
  if (!workspace || workspace_bytes < need) {
    set_error("conv(roipool): workspace %zu < %zu", workspace_bytes, need);
    return MSCNN_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  float* own_maps = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + roipool_scratch_offset(p));
  return wino_forward(p, packed, bias, y, nullptr, workspace, workspace_bytes, st, [&](float* V) {
    if (!prepared_maps) {
      const int rc = mscnn::roipool_wino33_build_maps(feat, own_maps, N, C, H, W, st);
      if (rc != MSCNN_OK) return rc;
    }
    return mscnn::roipool_wino33_forward(prepared_maps ? prepared_maps : own_maps, rois, V, p->d.N, N, C, H, W, p->T_pad, spatial_scale,
                                         pad_ratio_a, pad_ratio_b, st);
  });
}

// Head / direct / igemm forward (everything except the three-stage Winograd path).
static int conv_forward_single(const mscnn_conv_plan* p, const float* x, const float* w, const float* packed, const float* bias,
                               float* y, float* y_pool, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const mscnn_conv_desc& d = p->d;
  if (p->x3h.rows) {
    MSCNN_REQUIRE(packed, "conv: head kernel needs packed weights (mscnn_conv2d_pack_weights)");
    if (!workspace || workspace_bytes < p->ws_bytes) {
      set_error("conv(head x3): workspace %zu < %zu", workspace_bytes, p->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
    for (int n = 0; n < d.N; ++n) {      // (the deploy nets run batch 1; images share the workspace one after the other)
      const int rc = x3_head_forward(p->x3h, x + (size_t)n * d.Cin * d.H * d.W, packed, bias, y + (size_t)n * d.Cout * p->Ho * p->Wo, d.H,
                                     d.W, p->Ho, p->Wo, d.pad_h, d.pad_w, d.relu, d.N == 1 ? p->amax_in : nullptr, workspace, st);
      if (rc != MSCNN_OK) return rc;
    }
    return MSCNN_OK;
  }
  if (p->hg) {
    MSCNN_REQUIRE(packed, "conv: head kernel needs packed weights (mscnn_conv2d_pack_weights)");
    if (!workspace || workspace_bytes < p->ws_bytes) {
      set_error("conv(head gemm): workspace %zu < %zu", workspace_bytes, p->ws_bytes);
      return MSCNN_ERR_WORKSPACE;
    }
    float* T = static_cast<float*>(workspace);
    void* nested_ws = static_cast<unsigned char*>(workspace) + p->hg_t_bytes;
    for (int n = 0; n < d.N; ++n) {      // (the deploy nets run batch 1; images share T one after the other)
      // (tile order: the M tiles of one pixel tile together -- they share the B tile)
      int rc = launch_igemm(p->hg, x + (size_t)n * d.Cin * d.H * d.W, packed, nullptr, T, nullptr, nested_ws, p->hg->ws_bytes, st, 0u,
                            p->hg_kw ? 0 : 1);
      if (rc != MSCNN_OK) return rc;
      if (p->hg_kw)      // the rows already hold the sums over kh: shift-and-add over the kernel's columns only
        rc = head_shift_add(T, bias, y + (size_t)n * d.Cout * p->Ho * p->Wo, d.Cout, d.H, d.W, p->Ho, p->Wo, 1, d.Kw, 0, d.pad_w,
                            (unsigned)((long)d.H * d.W), d.relu, st);
      else
        rc = head_shift_add(T, bias, y + (size_t)n * d.Cout * p->Ho * p->Wo, d.Cout, d.H, d.W, p->Ho, p->Wo, d.Kh, d.Kw, d.pad_h, d.pad_w,
                            (unsigned)((long)d.H * d.W), d.relu, st);
      if (rc != MSCNN_OK) return rc;
    }
    return MSCNN_OK;
  }
  if (p->head.entry >= 0) {
    MSCNN_REQUIRE(packed, "conv: head kernel needs packed weights (mscnn_conv2d_pack_weights)");
    return head_forward(d, p->head, p->Ho, p->Wo, x, packed, bias, y, workspace, workspace_bytes, st);
  }
  if (p->c3) {
    MSCNN_REQUIRE(w, "conv: the Cin = 3 kernel needs the Caffe-layout weights");
    return mscnn::c3_forward(d, x, w, bias, y, p->amax_out, st);
  }
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
  if (p->wf_use) return mscnn::wf2_launch(p->wf, x, packed, bias, y, y_pool, d.relu, st);
  if (p->wc_use) return mscnn::wconv_launch(p->wc, x, packed, bias, y, y_pool, d.relu, st);
  return launch_igemm(p, x, packed, bias, y, y_pool, workspace, workspace_bytes, st, 0u, 0);
}
