// conv1_2 of the VGG trunk on gfx950: 3x3 / pad 1 / stride 1, 64 output channels, Cin a multiple of 8, on the full-resolution map
// (1 x 64 x 576 x 1920 -> 64: 81.5 GFLOP, the largest single kernel of a 7s-576 frame).  Replaces ConvolutionLayer::Forward_gpu
// (src/caffe/layers/conv_layer.cu:8-23: im2col + SGEMM per image) + the in-place ReLU (relu_layer.cu:17-26) + the MAX 2x2 pooling
// that follows (pooling_layer.cu:11-47) for that one shape class; every other direct convolution stays on conv.hip's igemm kernel.
//
// Why a second direct kernel.  The igemm kernel runs this layer as two 4-wave workgroups per CU (64 x 256 tiles, operands staged
// global -> VGPR -> LDS, two barriers per 8-channel chunk).  Its phase timeline (profiles/r04_ab_conv1_2_epilogue.txt) shows the
// matrix pipe saturated only while BOTH workgroups are inside their K loops (65 - 70 us per tile for 32.6 us of MFMA time each);
// for the 14 - 17 us per tile that one of them spends in its epilogue / prologue / closing barrier the other feeds the pipe alone
// at ~58 %: 0.77 of the fp32 MFMA peak.  This kernel is wgemm.hip's structure applied to the implicit GEMM
//     y[co][pixel] = sum over (chunk of 8 channels, tap, channel) w[co][c][tap] * x[c][pixel + tap]:
//   * ONE 512-thread workgroup per CU (8 waves, two per SIMD, in lockstep) owns a tile of 64 channels x (4 rows x 128 columns);
//     wave (wr, wc) = rows 2 wr, 2 wr + 1 x columns 32 wc .. + 31 = 2 x 2 MFMA blocks (both rows of a pooling window in one wave);
//   * operands by LDS-DMA into a 3-stage ring, two chunks ahead: the chunk's packed weight slab (9 taps x 8 channels x 64 = 18 KB,
//     dwordx4 pieces) and its input patch (8 channels x 6 rows x 130 columns = 24.4 KB) -- the patch rows start at column w0 - 1,
//     i.e. at 4-byte alignment, so they travel as 256-byte DWORD pieces whose lane offsets carry the zero padding (a lane outside
//     the map reads out of range = 0);
//   * one s_barrier per chunk; the taps are LDS read offsets (immediates), not data movement;
//   * two accumulator sets: bias + ReLU + the y stores + the 2x2-pooled store of a finished tile ride along the MFMAs of the
//     next tile's first chunk; whole tiles only (slot s of the persistent grid takes tiles s, s + G, ...).
// Summation order: chunk, tap (kh, kw), channel pair -- the k order of the igemm kernel, bias last: bit-identical to it
// (tests/test_gpu_ops.py::test_wconv_ring_kernel_against_the_igemm_kernel).
// STATUS (round 4): AUTO's choice for this shape class (tune_flags bit 15 keeps the igemm kernel).  Stand-alone conv1_2 runs in
// 654 - 660 us against the igemm kernel's 681 - 688 (0.785 vs 0.75 of the fp32 MFMA peak); in the net, same process, alternating:
// 4.718 vs 4.768 ms per forward (tools/ab_net_layer.py, profiles/r04_ab_conv1_2_ring.txt).  A stream-K tail for the 2160 tiles
// (8.44 rounds) was tried and dropped: it produced wrong sums once more than one tile was split.
#include "wconv.h"
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WcArgs {
  const float* x; const float* wp; const float* bias; float* y; float* yp;
  int N, Cin, H, W, NTH, NTW, KI, tiles, G, relu;
  unsigned x_bytes, w_bytes, y_bytes, p_bytes;
  unsigned long long* dbg;      // development: per workgroup {shader cycles, 100 MHz ticks}, then {start, end}
};

struct KC {
  static constexpr int BM = 64, TR = 4, TC = 128, CK = 8, TAPS = 9, ST = 3, NW = 8, THREADS = NW * 64;      // tile: 64 channels x (4 rows x 128 columns)
  static constexpr int A_BYTES = TAPS * CK * BM * 4;                       // 18432: one chunk of packed weights [tap][ck][64]
  static constexpr int PR = TR + 2, PC = TC + 2, CH_STRIDE = PR * PC;      // patch 6 x 130 per channel
  static constexpr int B_FLOATS = CK * CH_STRIDE;                          // 6240 floats = 24960 bytes
  static constexpr int PA = A_BYTES / 1024, PA_W = (PA + NW - 1) / NW;     // 18 dwordx4 pieces, 3 slots per wave
  static constexpr int PB = (B_FLOATS + 63) / 64, PB_W = (PB + NW - 1) / NW;   // 98 dword pieces, 13 slots per wave
  static constexpr int NP = PA_W + PB_W;                                   // LDS-DMA instructions per wave and chunk
  static constexpr int STAGE_BYTES = A_BYTES + PB * 256;                   // the last B piece is half used: 43520
  static constexpr int SPARE_A = ST * STAGE_BYTES, SPARE_B = SPARE_A + 1024;   // where the waves without a real piece in a slot aim
  static constexpr int LDS_BYTES = SPARE_B + 256;
  static constexpr int STEPS = TAPS * (CK / 2);                            // 36 MFMA groups (k pairs) per chunk
  static constexpr int MI = 2, NI = 2, NMF = MI * NI, NR = MI + NI;
  static constexpr int UNITS = MI * 16;                                    // a unit = (mi, r): y of both rows + the pooled value
  static constexpr int SK = STEPS - UNITS;                                 // first group that carries a unit of the finished tile
  static_assert(STAGE_BYTES % 256 == 0 && LDS_BYTES <= 160 * 1024 && NP <= STEPS && SK >= 0, "geometry");
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOob = 0x80000000u;

// LDS-DMA: lane l copies 16 (4) bytes from rsrc + voff(l) + soff to LDS lds_addr + 16 l (4 l).  Hidden from the compiler's waitcnt
// bookkeeping by design (counted by hand below); s_nop 4: SGPR written by v_readfirstlane -> VMEM read; s_nop 0: M0 write -> LDS-DMA.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  soff = __builtin_amdgcn_readfirstlane(soff);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  soff = __builtin_amdgcn_readfirstlane(soff);
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ float lds_rd(unsigned addr) {
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float lane_xor1(float v) {      // DPP quad_perm [1, 0, 3, 2]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float max2(float a, float b) { return b > a ? b : a; }

// A workgroup's work: the whole tiles slot, slot + G, ... (every tile = KI chunks); the producer (LDS-DMA) and the consumer (MFMA)
// walk the same list.
struct TileCursor {
  int t, G, tiles;
  __device__ __forceinline__ void init(const WcArgs& a, int slot) { t = slot; G = a.G; tiles = a.tiles; }
  __device__ __forceinline__ int count() const { return t < tiles ? (tiles - t + G - 1) / G : 0; }
  __device__ __forceinline__ bool next(int& tile) {
    if (t >= tiles) return false;
    tile = t; t += G;
    return true;
  }
};

__global__ __launch_bounds__(KC::THREADS, 2) void wconv_kernel(WcArgs a) {
  typedef KC C;
  __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[C::LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;      // rows 2 wr, 2 wr + 1; columns 32 wc .. + 31

  const int xcd = (int)(blockIdx.x % 8), gq = a.G / 8, gr = a.G % 8;
  const int slot = xcd * gq + min(xcd, gr) + (int)(blockIdx.x / 8);
  TileCursor pcur, ccur;
  pcur.init(a, slot); ccur.init(a, slot);
  const int nunits = pcur.count() * a.KI;
  if (nunits == 0) return;
  unsigned long long dbg_c = 0, dbg_r = 0;
  if (a.dbg && tid == 0) { dbg_c = __builtin_amdgcn_s_memtime(); dbg_r = __builtin_amdgcn_s_memrealtime(); }

  const __amdgpu_buffer_rsrc_t rA = make_rsrc(a.wp, a.w_bytes), rX = make_rsrc(a.x, a.x_bytes), rY = make_rsrc(a.y, a.y ? a.y_bytes : 0u),
                               rP = make_rsrc(a.yp, a.yp ? a.p_bytes : 0u), rBias = make_rsrc(a.bias, a.bias ? 64u * 4u : 0u);
  const int plane = a.H * a.W, hp = a.H / 2, wp2 = a.W / 2;
  const unsigned plane_bytes = (unsigned)plane * 4u, pplane_bytes = (unsigned)(hp * wp2) * 4u;
  const int tiles_per_img = a.NTH * a.NTW;
  const unsigned lds0 = (unsigned)(size_t)lds_raw;

  // this lane's patch elements in its wave's B slots: element e = slot * 64 + lane of [ch][6][130]
  unsigned pk[C::PB_W], vB[C::PB_W];
#pragma unroll
  for (int i = 0; i < C::PB_W; ++i) {
    const int sl = wave * C::PB_W + i, e = sl * 64 + lane;
    const int ch = e / C::CH_STRIDE, rem = e % C::CH_STRIDE;
    pk[i] = (sl < C::PB && e < C::B_FLOATS) ? ((unsigned)ch << 16) | ((unsigned)(rem / C::PC) << 8) | (unsigned)(rem % C::PC) : 0xFFFFFFFFu;
    vB[i] = kOob;
  }
  const unsigned vA = (unsigned)lane * 16u;
  // per-lane LDS read bases inside a stage (bytes)
  const unsigned a_lane = (unsigned)(khalf * C::BM + l31) * 4u;
  const unsigned b_lane = (unsigned)C::A_BYTES + (unsigned)(khalf * C::CH_STRIDE + 2 * wr * C::PC + wc * 32 + l31) * 4u;

  // bias of this lane's 32 output channels (registers for the whole kernel: Cout = 64 = one M tile)
  float bias_r[C::MI][16];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bias_r[mi][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, (unsigned)(mi * 32 + 4 * khalf) * 4u, (unsigned)((r & 3) + 8 * (r >> 2)) * 4u, 0));

  // ---- producer cursor ------------------------------------------------------------------------------------------------------
  int pu = 0, p_kc = 0, p_k1 = 0, p_stage = 0;
  unsigned p_b = 0;                     // byte offset of the tile's image in x
  unsigned vA_eff = vA;
  auto p_begin = [&]() {
    if (pu >= nunits) {                 // past the last unit: out-of-range pieces into a stage nobody reads any more
      vA_eff = kOob;
#pragma unroll
      for (int i = 0; i < C::PB_W; ++i) vB[i] = kOob;
    }
    if (p_kc == p_k1 && pu < nunits) {  // next tile: its lane offsets (zero padding = out of range)
      int t = 0;
      pcur.next(t);
      p_kc = 0; p_k1 = a.KI;
      const int n = t / tiles_per_img, tt = t % tiles_per_img;
      const int h0 = (tt / a.NTW) * C::TR, w0 = (tt % a.NTW) * C::TC;
      p_b = __builtin_amdgcn_readfirstlane((unsigned)(n * a.Cin) * plane_bytes);
#pragma unroll
      for (int i = 0; i < C::PB_W; ++i) {
        const int ch = (int)(pk[i] >> 16), hh = h0 - 1 + (int)((pk[i] >> 8) & 255u), ww = w0 - 1 + (int)(pk[i] & 255u);
        const bool ok = pk[i] != 0xFFFFFFFFu && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        vB[i] = ok ? (unsigned)(ch * plane + hh * a.W + ww) * 4u : kOob;
      }
    }
  };
  auto p_piece = [&](auto ic) {         // DMA instruction i of the unit at the cursor: weight pieces first, then patch pieces
    constexpr int i = decltype(ic)::value;
    const unsigned ls = lds0 + (unsigned)p_stage * C::STAGE_BYTES;
    if constexpr (i < C::PA_W) {
      const int sl = wave * C::PA_W + i;
      const unsigned sa = (unsigned)p_kc * C::A_BYTES + (unsigned)sl * 1024u;
      dma16(rA, sl < C::PA ? vA_eff : kOob, sa, sl < C::PA ? ls + (unsigned)sl * 1024u : lds0 + (unsigned)C::SPARE_A);
    } else {
      constexpr int j = i - C::PA_W;
      const int sl = wave * C::PB_W + j;
      const unsigned sb = p_b + (unsigned)(p_kc * C::CK) * plane_bytes;
      dma4(rX, vB[j], sb, sl < C::PB ? ls + (unsigned)C::A_BYTES + (unsigned)sl * 256u : lds0 + (unsigned)C::SPARE_B);
    }
  };
  auto p_end = [&]() {
    ++pu;
    ++p_kc;
    if (++p_stage == C::ST) p_stage = 0;
  };

  // ---- ring protocol: wgemm.hip's (three stages, unit u in stage u % 3, pieces of unit u + 2 issued during chunk u) ------------
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (pu < nunits) {
      p_begin();
      static_for<0, C::NP>([&](auto ic) { p_piece(ic); });
      p_end();
    }
  wait_vm_barrier<0>();

  f32x16 acc[2][C::MI][C::NI];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][mi][ni][r] = 0.f;
  // where the finished tile goes: y and the pooled map
  unsigned old_voff[C::MI][C::NI], old_poff[C::MI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi) {
    old_poff[mi] = kOob;
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) old_voff[mi][ni] = kOob;
  }

  float av[2][C::MI], bv[2][C::NI];
  // k pair s of a chunk = (tap s / 4, channels 2 (s % 4) and + 1): A [tap][ck][64], B [ck][6][130] at (row + kh, column + kw)
#define WC_A_IMM(s) ((((s) / 4) * C::CK + 2 * ((s) % 4)) * C::BM * 4)
#define WC_B_IMM(s) ((2 * ((s) % 4) * C::CH_STRIDE + (((s) / 4) / 3) * C::PC + (((s) / 4) % 3)) * 4)
#define WC_READ(buf, stage_addr, s)                                                                                        \
  {                                                                                                                         \
    static_for<0, C::MI>([&](auto m_) { av[buf][decltype(m_)::value] = lds_rd<WC_A_IMM(s) + decltype(m_)::value * 128>((stage_addr) + a_lane); }); \
    static_for<0, C::NI>([&](auto n_) { bv[buf][decltype(n_)::value] = lds_rd<WC_B_IMM(s) + decltype(n_)::value * C::PC * 4>((stage_addr) + b_lane); }); \
  }
  // stores issued behind a chunk's last DMA instruction (instruction NP - 1 follows the first store of group NP - 1)
  constexpr int AFTER_DMA = (C::NP - 1 >= C::SK ? 2 : 0) + 3 * (C::STEPS - (C::NP > C::SK ? C::NP : C::SK));
  static_assert(AFTER_DMA >= 0 && AFTER_DMA <= 63, "vmcnt field");

  unsigned c_addr = lds0;
  WC_READ(0, c_addr, 0);

  // part k of unit u = (mi, r) of the finished segment in accumulator set Q: k = 0 / 1: y of row 2 wr + k; k = 2: the pooled value
  auto store_part = [&](auto qc, auto uc, auto kc_) {
    constexpr int Q = decltype(qc)::value, u = decltype(uc)::value, k = decltype(kc_)::value;
    constexpr int mi = u / 16, r = u % 16, dr = (r & 3) + 8 * (r >> 2);
    {
      float v0 = acc[Q][mi][0][r] + bias_r[mi][r], v1 = acc[Q][mi][1][r] + bias_r[mi][r];
      if (a.relu) { v0 = v0 < 0.f ? 0.f : v0; v1 = v1 < 0.f ? 0.f : v1; }
      if constexpr (k == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rY, old_voff[mi][0], (unsigned)dr * plane_bytes, 0);
      else if constexpr (k == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rY, old_voff[mi][1], (unsigned)dr * plane_bytes, 0);
      else {
        float m = max2(v0, v1);
        m = max2(m, lane_xor1(m));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), rP, old_poff[mi], (unsigned)dr * pplane_bytes, 0);
      }
    }
  };

  auto chunk = [&](auto par_c, auto flush_c, unsigned next_addr) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FLUSH = decltype(flush_c)::value;
    p_begin();
    static_for<0, C::STEPS>([&](auto sc) {
      constexpr int s = decltype(sc)::value, cur = s & 1, nxt = cur ^ 1;
      if constexpr (s + 1 < C::STEPS) WC_READ(nxt, c_addr, s + 1)
      else WC_READ(nxt, next_addr, 0)
      lds_wait<C::NR>();
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) lds_pin(av[cur][mi]);
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) lds_pin(bv[cur][ni]);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, C::NMF>([&](auto jc) {
        constexpr int j = decltype(jc)::value, mi = j / C::NI, ni = j % C::NI;
        acc[PAR][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi], bv[cur][ni], acc[PAR][mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FLUSH && s >= C::SK && j < 3) {
          store_part(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, s - C::SK>{}, jc);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (j == 0 && s < C::NP) {
          p_piece(std::integral_constant<int, s + 0 * j>{});      // (dependent on jc, so that the discarded branch is not instantiated)
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
    p_end();
    if constexpr (FLUSH) wait_vm_barrier<AFTER_DMA>();
    else wait_vm_barrier<0>();
    c_addr = next_addr;
  };

  int c_stage = 0;
  auto next_stage_addr = [&]() {
    if (++c_stage == C::ST) c_stage = 0;
    return lds0 + (unsigned)c_stage * C::STAGE_BYTES;
  };
  int t = 0;
  // one tile on accumulator set PAR: its first chunk carries the previous tile's stores (tile 0: that set is empty and its offsets are
  // out of range); afterwards the flushed set is cleared for the tile after this one and this tile becomes the pending flush
  auto tile_pass = [&](auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    chunk(par_c, std::true_type{}, next_stage_addr());
    for (int kc = 1; kc < a.KI; ++kc) chunk(par_c, std::false_type{}, next_stage_addr());
    const int n = t / tiles_per_img, tt = t % tiles_per_img;
    const int h0 = (tt / a.NTW) * C::TR, w0 = (tt % a.NTW) * C::TC;
    const int col = w0 + wc * 32 + l31;
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
      const int co0 = mi * 32 + 4 * khalf;
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const int row = h0 + 2 * wr + ni;
        old_voff[mi][ni] = a.y ? (unsigned)((n * C::BM + co0) * plane + row * a.W + col) * 4u : kOob;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[PAR ^ 1][mi][ni][r] = 0.f;      // (its stores were issued during this tile's first chunk)
      }
      old_poff[mi] = (a.yp && !(lane & 1)) ? (unsigned)((n * C::BM + co0) * (hp * wp2) + ((h0 >> 1) + wr) * wp2 + (col >> 1)) * 4u : kOob;
    }
  };
  int par = 0;
  while (true) {
    if (!ccur.next(t)) break;
    tile_pass(std::integral_constant<int, 0>{});
    par = 1;
    if (!ccur.next(t)) break;
    tile_pass(std::integral_constant<int, 1>{});
    par = 0;
  }
  auto finish = [&](auto qc) {      // the last tile's stores
    static_for<0, C::UNITS>([&](auto uc) {
      store_part(qc, uc, std::integral_constant<int, 0>{});
      store_part(qc, uc, std::integral_constant<int, 1>{});
      store_part(qc, uc, std::integral_constant<int, 2>{});
    });
  };
  if (par == 1) finish(std::integral_constant<int, 0>{});
  else finish(std::integral_constant<int, 1>{});
  if (a.dbg && tid == 0) {
    const unsigned long long end_r = __builtin_amdgcn_s_memrealtime();
    a.dbg[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - dbg_c;
    a.dbg[blockIdx.x * 2 + 1] = end_r - dbg_r;
    a.dbg[2 * a.G + blockIdx.x * 2] = dbg_r;
    a.dbg[2 * a.G + blockIdx.x * 2 + 1] = end_r;
  }
#undef WC_READ
#undef WC_A_IMM
#undef WC_B_IMM
}

}  // namespace

namespace mscnn {

static int device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  return n;
}

const char* wconv_kernel_name() { return "wconv_64x512_k3x3"; }

bool wconv_plan(int N, int Cin, int H, int W, int Cout, bool force_small, WconvPlan* o) {
  typedef KC C;
  if (Cout != C::BM || Cin < C::CK || Cin % C::CK != 0 || N < 1 || H % C::TR != 0 || W % C::TC != 0) return false;
  o->N = N; o->Cin = Cin; o->H = H; o->W = W;
  o->NTH = H / C::TR; o->NTW = W / C::TC; o->KI = Cin / C::CK;
  const long tiles = (long)N * o->NTH * o->NTW;
  o->G = device_cus();
  if (tiles < 2L * o->G && !force_small) return false;      // (small maps: the igemm kernel's 64 x 256 tiles fill the chip better)
  o->tiles = (int)tiles;
  o->packed_bytes = (size_t)o->KI * C::A_BYTES;
  o->ws_bytes = 0;
  const double lim = 2147483648.0 - 65536.0;      // 32-bit buffer offsets; 0x80000000 is the out-of-range sentinel
  if ((double)N * Cin * H * W * 4 >= lim || (double)N * C::BM * H * W * 4 >= lim) return false;
  return true;
}

int wconv_launch(const WconvPlan& p, const float* x, const float* packed, const float* bias, float* y, float* y_pool, int relu, hipStream_t st,
                 unsigned long long* dbg) {
  typedef KC C;
  WcArgs a;
  a.x = x; a.wp = packed; a.bias = bias; a.y = y; a.yp = y_pool;
  a.N = p.N; a.Cin = p.Cin; a.H = p.H; a.W = p.W; a.NTH = p.NTH; a.NTW = p.NTW; a.KI = p.KI; a.tiles = p.tiles; a.G = p.G;
  a.relu = relu; a.dbg = dbg;
  a.x_bytes = (unsigned)((size_t)p.N * p.Cin * p.H * p.W * 4); a.w_bytes = (unsigned)p.packed_bytes;
  a.y_bytes = (unsigned)((size_t)p.N * C::BM * p.H * p.W * 4); a.p_bytes = (unsigned)((size_t)p.N * C::BM * (p.H / 2) * (p.W / 2) * 4);
  wconv_kernel<<<p.G, C::THREADS, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
