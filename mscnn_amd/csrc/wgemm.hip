// The batched transform-domain GEMM of the Winograd convolution paths on gfx950:
//     M[p][Cout][T_pad] = U[p][Cout][Cin] x V[p][Cin][T_pad],   p < P = 25 (F(3x3,3x3)) or 36 (F(4x4,3x3)) planes,
// exact fp32 on v_mfma_f32_32x32x2_f32.  This is where cudnnConvolutionForward's time goes in the reference
// (cudnn_conv_layer.cu:11-46) and the dominant kernel of this library (13 launches per 7s-576 frame).
//
// Why a second GEMM kernel.  Rounds 1-2 ran these GEMMs on the 1x1 instance of the implicit-GEMM convolution kernel
// (conv.hip: 128 x 128 tiles, 4 waves, operands staged global -> VGPR -> ds_write_b128 -> LDS, two barriers per 32-channel
// chunk, 2-4 workgroups per CU).  The per-workgroup timeline of round 3 (tools/wg_trace.py, profiles/r03_wg_trace.txt) showed
// where its 0.72-0.76 of the MFMA peak goes: the K loop itself runs at 63 % (one workgroup per CU) to 80 % (two) of the
// MFMA-bound time -- every chunk pays the barrier / ds_write / barrier / load-issue / first-ds_read sequence with the matrix
// pipe idle, and more co-resident workgroups stop helping because they fetch 8 bytes per MFMA clock and CU from L2 --
// and every tile then pays a 3-9 us store epilogue during which the workgroup issues no MFMA at all.
//
// This kernel is built around those three findings:
//   * bigger tiles, fewer bytes: one 512-thread workgroup (8 waves, two per SIMD) per CU owns a 256 x 128 (or 128 x 256)
//     tile; 48 KB of operands per 32-channel chunk instead of 2 x 32 KB for the same MFMA work (5.9 instead of 8 B/clk/CU);
//   * no staging through registers: the operands travel HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per
//     wave instruction) into a 3-stage ring, issued two chunks ahead; the loads stay in flight across the ONE s_barrier per
//     chunk (counted s_waitcnt vmcnt(N), never 0 in the steady state), and the ring runs across tile boundaries, so a tile has
//     no prologue;
//   * the epilogue overlaps the next tile: a finished tile's accumulators are moved aside and their stores are issued a few
//     at a time between the MFMAs of the next tile's first chunks.
// U is packed once per weight change as Up[p][mt][kc][ck][BM] (the igemm packing, so one chunk of a tile's A operand is
// one contiguous slab); V and M are the plain planes the transform kernels of winograd.hip write / read.
//
// Schedule: persistent grid, one workgroup per CU; tile t = ((p * NT + nt) * MT + mt); slot s of the grid takes tiles
// s, s + G, ...; XCD x (workgroups b = x mod 8) owns the contiguous slots [x G/8, (x+1) G/8): in every round an XCD works on
// 32 consecutive tiles = one or two planes, so that plane's U (<= 1 MB) and the V tile shared by the MT tiles above it stay in
// the XCD's own 4 MB L2.  Where the tile count does not divide the grid the remainder is split stream-K style (whole
// K chunks): a workgroup that holds later chunks of a split tile publishes its partial sum as a write-through fp32 slab + flag,
// the workgroup that holds the tile's FIRST chunks (it gets to them last in time) adds the slabs in k order and stores the tile
// (deterministic: fixed order, no atomics on data; hand-off = the release / acquire recipe of the CDNA programming guide, G16).
#include "wgemm.h"
#include "common.h"
#include <atomic>
#include <chrono>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgemmArgs {
  const float* Up; const float* V; float* M; float* ws;
  int P, MT, NT, KI, Cin, Cout, T_pad;
  int tiles, G, abl;
  int full_q;                   // whole tiles per workgroup (tile slot + r G, r < full_q); the other tiles are split stream-K style
  unsigned ws_bytes;
  unsigned epoch;               // this launch's tag for the partial-sum hand-off flags (never 0, unique per launch in the process)
  unsigned long long* dbg;      // development: [G] {shader cycles, 100 MHz ticks} over the kernel, then [G] {start, end} in 100 MHz ticks (nullptr in the product)
  unsigned up_bytes, v_bytes, m_bytes;
  const float* bias; int relu;  // EPI kernels only: y = acc + bias[column] (nullptr: none), then max(y, 0) when relu -- applied to what goes to M, never to a partial slab
  // hand-off health: a finisher that gives up on a contributor after spin_limit polls stores this launch's epoch into *status -- one
  // word of pinned host memory per device that the host reads at its next synchronisation point (mscnn_wgemm_handoff_event)
  unsigned long long* status;
  unsigned spin_limit;
  int drop_publish;             // fault injection (mscnn_debug_wgemm_handoff_fault): contributors never set their flag
};

template <int BM_, int BN_, int WGM_, int WGN_, int CK_, int ST_, int DPG_ = 1, int SK_ = 0>
struct WCfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, CK = CK_, ST = ST_;
  // schedule knobs: LDS-DMA pieces issued per MFMA group (from group 0 on), first group that carries ride-along stores
  static constexpr int DPG = DPG_, SK = SK_;
  static constexpr int NW = WGM * WGN, THREADS = NW * 64;
  static constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 32, NI = WN / 32;
  static constexpr int A_BYTES = CK * BM * 4, B_BYTES = CK * BN * 4, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int STAGE_FLOATS = STAGE_BYTES / 4;
  static constexpr int PA = A_BYTES / 1024, PB = B_BYTES / 1024;      // 1 KB LDS-DMA pieces per chunk
  static constexpr int PA_W = PA / NW, PB_W = (PB + NW - 1) / NW;     // ... per wave
  static constexpr int NP = PA_W + PB_W;
  static constexpr int F4_PER_ROW = BN / 4;                           // 16-byte units per B row
  // B pieces: when a 1 KB piece is a whole number of B rows and the pieces divide evenly over the waves (BN 128 / 256), every lane's
  // source offset is the same for all pieces; otherwise (BN 96: 24 units per row, 12 pieces on 8 waves) each piece has its own lane
  // offsets, and the waves without a piece in the last slot issue an out-of-range piece into a spare KB behind the ring
  static constexpr bool B_EVEN = (64 % F4_PER_ROW == 0) && (PB % NW == 0);
  static constexpr int ROWS_PER_PIECE = B_EVEN ? 64 / F4_PER_ROW : 0; // BN 128: 2 rows, BN 256: 1 row
  static constexpr int LDS_FLOATS = ST * STAGE_FLOATS + (B_EVEN ? 0 : 256);
  static constexpr int STEPS = CK / 2;                                // MFMA k-pairs per chunk
  static_assert(PA % NW == 0 && B_BYTES % 1024 == 0, "pieces per wave");
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS ring");
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile");
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

// One LDS-DMA instruction: lane l copies 16 bytes from  rsrc + voff(l) + soff  to LDS  lds_addr + 16 l  (1 KB per wave).
// Invisible to the compiler's s_waitcnt bookkeeping by design: completion is counted by hand (wait_vm below).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  soff = __builtin_amdgcn_readfirstlane(soff);            // wave-uniform by construction; makes it provably so for the "s" constraints
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  // (s_nop 4: an SGPR written by v_readfirstlane needs 5 wait states before a VMEM instruction reads it -- the compiler does not pad
  // inline asm; s_nop 0: M0 write -> LDS-DMA)
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// LDS operand read with an immediate offset (no address arithmetic on the vector ALU, which the fp32 MFMA shares) -- hidden from the
// compiler's lgkmcnt bookkeeping, so every use is preceded by lds_wait<N>() + lds_pin().
template <int OFF>
__device__ __forceinline__ float lds_rd(unsigned addr) {
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_pin(float& v) { asm volatile("" : "+v"(v)); }

// (tile, chunk) units of the stream-K remainder [0, total) owned by workgroup g: total < 2^31 / G is checked by the plan
__device__ __host__ __forceinline__ void wg_range(int total, int G, int g, int& b, int& e) {
  b = (int)((unsigned)total * (unsigned)g / (unsigned)G);
  e = (int)((unsigned)total * (unsigned)(g + 1) / (unsigned)G);
}

// A workgroup's work as a list of segments (tile, chunks [k0, k1)): first its full_q whole tiles, then its share of the remainder
// tiles' (tile, chunk) space.  part: -1 = the segment is a whole tile (stored to M), 0 / 1 = partial (stored to that slab).
struct SegCursor {
  int slot, G, KI, full_q, rem_tile0, r;
  int it, re;
  int nparts;
  __device__ __forceinline__ void init(const WgemmArgs& a, int slot_) {
    slot = slot_; G = a.G; KI = a.KI; rem_tile0 = a.full_q * a.G; r = 0; nparts = 0;
    // whole tiles of this slot: rounds r < full_q whose tile exists (with whole-tile scheduling the last round is not full)
    full_q = slot_ < a.tiles ? min(a.full_q, (a.tiles - slot_ + a.G - 1) / a.G) : 0;
    if (a.tiles > rem_tile0) wg_range((a.tiles - rem_tile0) * a.KI, a.G, slot_, it, re);
    else it = re = 0;
  }
  __device__ __forceinline__ int units() const { return full_q * KI + (re - it); }
  __device__ __forceinline__ bool next(int& t, int& k0, int& k1, int& part) {
    if (r < full_q) { t = slot + r * G; k0 = 0; k1 = KI; part = -1; ++r; return true; }
    if (it >= re) return false;
    t = rem_tile0 + it / KI;
    k0 = it % KI;
    const int left = re - it;
    k1 = left < KI - k0 ? k0 + left : KI;
    it += k1 - k0;
    part = (k0 == 0 && k1 == KI) ? -1 : nparts++;
    return true;
  }
};

template <class C, int ABL, int EPI = 0>
__global__ __launch_bounds__(C::THREADS, C::NW / 4) void wgemm_kernel(WgemmArgs a) {
  __shared__ __attribute__((aligned(1024))) float lds[C::LDS_FLOATS];
  __shared__ unsigned s_handoff_timeout;
  static_assert(C::ST == 3, "the ring protocol below is written for three stages");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int wm = wave / C::WGN, wn = wave % C::WGN;

  // slot of this workgroup (XCD-contiguous, see header)
  const int xcd = (int)(blockIdx.x % 8), gq = a.G / 8, gr = a.G % 8;
  const int slot = xcd * gq + min(xcd, gr) + (int)(blockIdx.x / 8);
  SegCursor pcur, ccur;                      // producer (LDS-DMA) and consumer (MFMA) walk the same segment list
  pcur.init(a, slot); ccur.init(a, slot);
  const int nunits = pcur.units();
  if (nunits == 0) return;
  unsigned long long dbg_c = 0, dbg_r = 0;
  if (a.dbg && tid == 0) { dbg_c = __builtin_amdgcn_s_memtime(); dbg_r = __builtin_amdgcn_s_memrealtime(); }

  const __amdgpu_buffer_rsrc_t rA = make_rsrc(a.Up, a.up_bytes), rB = make_rsrc(a.V, a.v_bytes), rM = make_rsrc(a.M, a.m_bytes),
                               rW = make_rsrc(a.ws, a.ws_bytes);
  const unsigned row_bytes = (unsigned)a.T_pad * 4u;
  const unsigned vA = (unsigned)lane * 16u;
  const unsigned vB = (unsigned)(lane / C::F4_PER_ROW) * row_bytes + (unsigned)(lane % C::F4_PER_ROW) * 16u;
  unsigned vBj[C::PB_W];                 // (uneven B pieces only) lane offsets of this wave's pieces: 16-byte unit j * 64 + lane of the stage
#pragma unroll
  for (int i = 0; i < C::PB_W; ++i) {
    const int u = (wave * C::PB_W + i) * 64 + lane;
    vBj[i] = wave * C::PB_W + i < C::PB ? (unsigned)(u / C::F4_PER_ROW) * row_bytes + (unsigned)(u % C::F4_PER_ROW) * 16u : 0x80000000u;
  }
  const unsigned lds0 = (unsigned)(size_t)lds;
  // per-lane LDS read bases inside a stage (bytes)
  const unsigned a_lane = (unsigned)(khalf * C::BM + wm * C::WM + l31) * 4u;
  const unsigned b_lane = (unsigned)C::A_BYTES + (unsigned)(khalf * C::BN + wn * C::WN + l31) * 4u;

  // ---- producer cursor: the next (tile, chunk) unit to put in flight; one LDS-DMA piece per call -----------------------------
  int pu = 0, p_kc = 0, p_k1 = 0, p_stage = 0;
  unsigned p_a = 0, p_b = 0;            // byte offsets of the unit's A slab / first B row
  unsigned vA_eff = vA, vB_eff = vB;
  auto p_begin = [&]() {                // called before piece 0 of a unit
    if (pu >= nunits) {
      vA_eff = 0x80000000u; vB_eff = 0x80000000u;
#pragma unroll
      for (int i = 0; i < C::PB_W; ++i) vBj[i] = 0x80000000u;
    }
    if (p_kc == p_k1 && pu < nunits) {  // next segment
      int t, part;
      pcur.next(t, p_kc, p_k1, part);
      const int mt = t % a.MT, nt = (t / a.MT) % a.NT, p = t / (a.MT * a.NT);
      // (the compiler does the divisions on the vector ALU: pin the results back to scalars -- the LDS-DMA statement needs SGPR operands)
      p_kc = __builtin_amdgcn_readfirstlane(p_kc); p_k1 = __builtin_amdgcn_readfirstlane(p_k1);
      p_a = __builtin_amdgcn_readfirstlane((unsigned)((p * a.MT + mt) * a.KI) * (unsigned)C::A_BYTES);
      p_b = __builtin_amdgcn_readfirstlane(((unsigned)(p * a.Cin) * (unsigned)a.T_pad + (unsigned)(nt * C::BN)) * 4u);
    }
  };
  // (past the last unit the cursor keeps running with out-of-range lane offsets: those pieces fetch nothing and write zeros into
  // a stage nobody reads any more -- the steady-state code path has no "is there a next unit" branch)
  auto p_piece = [&](auto ic) {         // piece i of the unit at the cursor (i < NP): A pieces first, then B pieces
    constexpr int i = decltype(ic)::value;
    const unsigned ls = lds0 + (unsigned)p_stage * C::STAGE_BYTES;
    if constexpr (i < C::PA_W) {
      const unsigned sa = p_a + (unsigned)p_kc * C::A_BYTES + (unsigned)(wave * C::PA_W + i) * 1024u;
      dma16(rA, vA_eff, sa, ls + (unsigned)(wave * C::PA_W + i) * 1024u);
    } else if constexpr (C::B_EVEN) {
      const int j = wave * C::PB_W + (i - C::PA_W);
      const unsigned sb = p_b + (unsigned)(p_kc * C::CK + j * C::ROWS_PER_PIECE) * row_bytes;
      dma16(rB, vB_eff, sb, ls + C::A_BYTES + (unsigned)j * 1024u);
    } else {
      const int j = wave * C::PB_W + (i - C::PA_W);
      const unsigned sb = p_b + (unsigned)(p_kc * C::CK) * row_bytes;
      dma16(rB, vBj[i - C::PA_W], sb, j < C::PB ? ls + C::A_BYTES + (unsigned)j * 1024u : lds0 + (unsigned)(C::ST * C::STAGE_BYTES));
    }
  };
  auto p_end = [&]() {
    ++pu;
    ++p_kc;
    if (++p_stage == C::ST) p_stage = 0;
  };

  // ---- ring protocol (three stages; unit u lives in stage u % 3) ------------------------------------------------------------
  //   chunk u:  its LDS-DMA pieces of unit u + 2 go out one per MFMA group during the first NP groups, into the stage that was
  //             read during chunk u - 1 (everybody is past that chunk's barrier);
  //             at its end every wave waits for ITS pieces of unit u + 2 (vmcnt), then the one s_barrier of the chunk.
  //   => unit u + 1 was complete and visible a whole chunk earlier, so the operands of the next chunk's first MFMA group are
  //      read BEFORE the barrier and the matrix pipe restarts right behind it.
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (pu < nunits) {
      p_begin();
      static_for<0, C::NP>([&](auto ic) { p_piece(ic); });
      p_end();
    }
  wait_vm_barrier<0>();

  // Two accumulator sets (round 4).  Segment n accumulates into acc[n & 1] while the stores of segment n - 1 are issued straight from
  // acc[(n - 1) & 1] between its MFMAs -- rounds 1-3 moved a finished tile aside first (64 v_mov + 64 zeroing moves per tile on the
  // vector ALU the fp32 MFMA shares, and 64 more live registers in every loop that touched both copies).  All register indices stay
  // compile-time: the segment body is instantiated once per parity.
  f32x16 acc[2][C::MI][C::NI];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][mi][ni][r] = 0.f;
  float old_bias[C::NI];                // (EPI) bias of this lane's column in each block column of the finished tile
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni) old_bias[ni] = 0.f;
  unsigned old_voff[C::MI][C::NI];      // byte offset of the finished tile's 32 x 32 blocks in M (row 0 of the block, this lane's column)
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) old_voff[mi][ni] = 0x80000000u;

  float av[2][C::MI], bv[2][C::NI];
#define WG_READ(buf, stage_addr, s)                                                                                         \
  {                                                                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < C::MI; ++mi) av[buf][mi] = 0.f;                                                  \
    static_for<0, C::MI>([&](auto m_) { av[buf][decltype(m_)::value] = lds_rd<(s) * 2 * C::BM * 4 + decltype(m_)::value * 128>((stage_addr) + a_lane); }); \
    static_for<0, C::NI>([&](auto n_) { bv[buf][decltype(n_)::value] = lds_rd<(s) * 2 * C::BN * 4 + decltype(n_)::value * 128>((stage_addr) + b_lane); }); \
  }
  constexpr int NR = C::MI + C::NI;                       // LDS reads per MFMA group
  constexpr int NMF = C::MI * C::NI;                      // MFMAs per group
  // ride-along stores of a finished tile: TOTAL of them over the groups SK .. STEPS-1, spread over a group's MFMAs
  constexpr int TOTAL = NMF * 16;
  constexpr int SPG = (TOTAL + (C::STEPS - C::SK) - 1) / (C::STEPS - C::SK);     // per group
  constexpr int SPM = (SPG + NMF - 1) / NMF;                                     // per MFMA
  constexpr int DMA_GROUPS = (C::NP + C::DPG - 1) / C::DPG;                      // groups 0 .. DMA_GROUPS-1 carry the unit's pieces
  static_assert(C::DPG <= NMF && DMA_GROUPS <= C::STEPS && C::SK < C::STEPS, "schedule");
  // stores issued behind a chunk's last piece (piece NP-1 sits behind MFMA (NP-1) % DPG of group DMA_GROUPS-1, after that slot's stores)
  constexpr int AFTER_DMA = [] {
    int n = 0;
    for (int g = 0; g < C::STEPS; ++g)
      for (int j = 0; j < NMF; ++j) {
        if (g < C::SK) continue;
        const int lo = (g - C::SK) * SPG + j * SPM, hi_g = (g - C::SK + 1) * SPG;
        int cnt = 0;
        for (int k = lo; k < lo + SPM && k < hi_g && k < TOTAL; ++k) ++cnt;
        const bool after = g > DMA_GROUPS - 1 || (g == DMA_GROUPS - 1 && j > (C::NP - 1) % C::DPG);
        if (after) n += cnt;
      }
    return n;
  }();

  __amdgpu_buffer_rsrc_t old_rsrc = rM;                  // where the finished segment goes: M (whole tile) or a partial-sum slab
  unsigned old_rowb = row_bytes;                          // ... its row stride in bytes
  bool old_pub = false;                                   // ... and whether it is a partial sum to publish (flag after its stores)
  unsigned c_addr = lds0;                                 // LDS address of the stage being multiplied
  WG_READ(0, c_addr, 0);

  // element e of the finished segment (accumulator set Q): block (e / 16), register e % 16
  auto store_one = [&](auto qc, auto ec) {
    constexpr int Q = decltype(qc)::value, e = decltype(ec)::value;
    constexpr int blk = e / 16, r = e % 16, mi = blk / C::NI, ni = blk % C::NI, dr = (r & 3) + 8 * (r >> 2);
    float v = acc[Q][mi][ni][r];
    // (a partial sum bound for another workgroup is stored write-through -- sc1 -- so that the flag that follows needs no L2
    // write-back; the branch is wave-uniform)
    if (old_pub) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), old_rsrc, old_voff[mi][ni], (unsigned)dr * old_rowb, 16);
    else {
      if constexpr (EPI) {
        v += old_bias[ni];
        if (a.relu) v = v < 0.f ? 0.f : v;      // std::max(v, 0) of relu_layer.cpp:14-15: a NaN stays a NaN
      }
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), old_rsrc, old_voff[mi][ni], (unsigned)dr * old_rowb, 0);
    }
  };

  // one chunk = STEPS MFMA groups accumulating into set PAR.  FLUSH: the previous segment's stores (set PAR ^ 1) ride along (SPM per
  // MFMA); DMA: this chunk carries the NP pieces of the unit two ahead.
  auto chunk = [&](auto par_c, auto flush_c, unsigned next_addr) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool FLUSH = decltype(flush_c)::value;
    p_begin();
    static_for<0, C::STEPS>([&](auto sc) {
      constexpr int s = decltype(sc)::value, cur = s & 1, nxt = cur ^ 1;
      if constexpr (!(ABL & 8)) {
        if constexpr (s + 1 < C::STEPS) WG_READ(nxt, c_addr, s + 1)
        else WG_READ(nxt, next_addr, 0)
        lds_wait<NR>();
      }
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) lds_pin(av[cur][mi]);
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) lds_pin(bv[cur][ni]);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NMF>([&](auto jc) {
        constexpr int j = decltype(jc)::value, mi = j / C::NI, ni = j % C::NI;
        if constexpr (!(ABL & 4)) acc[PAR][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi], bv[cur][ni], acc[PAR][mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FLUSH && s >= C::SK) {
          if constexpr (!(ABL & 2))
            static_for<0, SPM>([&](auto kc_) {
              constexpr int e = (s - C::SK) * SPG + j * SPM + decltype(kc_)::value;
              if constexpr (e < (s - C::SK + 1) * SPG && e < TOTAL) store_one(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, e>{});
            });
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (j < C::DPG && s * C::DPG + j < C::NP) {
          if constexpr (!(ABL & 1)) p_piece(std::integral_constant<int, s * C::DPG + j>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
    p_end();
    // this wave's pieces of the unit two ahead have landed; the stores of a flushed tile issued behind the last piece may still
    // be in flight (the hardware's vmcnt field holds 63)
    if constexpr (ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (FLUSH) wait_vm_barrier<(AFTER_DMA < 63 ? AFTER_DMA : 63)>();
    else wait_vm_barrier<0>();
    c_addr = next_addr;
  };

  int c_stage = 0;
  auto next_stage_addr = [&]() {
    if (++c_stage == C::ST) c_stage = 0;
    return lds0 + (unsigned)c_stage * C::STAGE_BYTES;
  };
  int t, k0, k1, part;
  int fin_tile = -1;
  typedef __attribute__((address_space(1))) unsigned long long gu64;
  gu64* flags = (gu64*)(a.ws + (size_t)a.G * (C::BM * C::BN));      // [G] hand-off flags + 1 status word, behind the G slabs
  const unsigned long long want_flag = ((unsigned long long)a.epoch << 32) | (unsigned)~a.epoch;
  auto publish_flag = [&]() {      // R1 of the guide's hand-off recipe: every storing wave drains, one barrier, ONE lane sets the flag
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0 && !a.drop_publish) __hip_atomic_store(flags + slot, want_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // one segment on accumulator set PAR: its first chunk carries the previous segment's stores (segment 0: that set is empty and its
  // offsets are out of range); afterwards the flushed set is cleared for the segment after this one, and this segment's destination --
  // whole tile -> M; partial without the tile's first chunk -> this workgroup's slab, published for the workgroup that has it;
  // partial WITH the first chunk (always the last segment) -> M, after the other contributors' slabs have been added (below) --
  // becomes the pending flush.
  auto segment = [&](auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    chunk(par_c, std::true_type{}, next_stage_addr());
    if (old_pub) { publish_flag(); old_pub = false; }
    for (int kc = k0 + 1; kc < k1; ++kc) chunk(par_c, std::false_type{}, next_stage_addr());
    const int mt = t % a.MT, nt = (t / a.MT) % a.NT, p = t / (a.MT * a.NT);
    const bool pub = part >= 0 && k0 > 0;
    if (part >= 0 && k0 == 0) fin_tile = t;
    old_pub = pub;
    if (pub) { old_rsrc = rW; old_rowb = (unsigned)C::BN * 4u; }
    else { old_rsrc = rM; old_rowb = row_bytes; }
    const unsigned slab = (unsigned)slot * (unsigned)(C::BM * C::BN * 4);
    if constexpr (EPI) {
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const int col = nt * C::BN + wn * C::WN + ni * 32 + l31;
        old_bias[ni] = (a.bias && col < a.T_pad) ? a.bias[col] : 0.f;
      }
    }
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const int row = wm * C::WM + mi * 32, col = wn * C::WN + ni * 32 + l31;
        const int co0 = mt * C::BM + row;
        const unsigned to_m = co0 + 32 <= a.Cout ? ((unsigned)(p * a.Cout + co0 + 4 * khalf) * (unsigned)a.T_pad + (unsigned)(nt * C::BN + col)) * 4u : 0x80000000u;
        const unsigned to_ws = slab + (unsigned)((row + 4 * khalf) * C::BN + col) * 4u;
        old_voff[mi][ni] = pub ? to_ws : to_m;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[PAR ^ 1][mi][ni][r] = 0.f;      // (its stores were issued during this segment's first chunk)
      }
  };
  int par = 0;                           // accumulator set of the NEXT segment (the last one computed sits in par ^ 1)
  while (true) {
    if (!ccur.next(t, k0, k1, part)) break;
    segment(std::integral_constant<int, 0>{});
    par = 1;
    if (!ccur.next(t, k0, k1, part)) break;
    segment(std::integral_constant<int, 1>{});
    par = 0;
  }
  // the last segment sits in set par ^ 1
  auto finish = [&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    if (fin_tile >= 0) {
      // ---- finish the split tile whose first chunks this workgroup computed: add the other contributors' slabs in k order (the
      // workgroups slot + 1, slot + 2, ... until the tile's chunks are covered; each published its part as ITS first remainder
      // segment, i.e. earlier in time).  Deterministic: the order of the additions is fixed. ----
      const int rem_tile0 = a.full_q * a.G, RU = (a.tiles - rem_tile0) * a.KI;
      const int tile_end = (fin_tile - rem_tile0 + 1) * a.KI;
      for (int s2 = slot + 1; s2 < a.G; ++s2) {
        int b, e;
        wg_range(RU, a.G, s2, b, e);
        if (b >= tile_end) break;
        if (e <= b) continue;
        if (tid == 0) {
          unsigned spins = 0;
          while (__hip_atomic_load(flags + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want_flag && ++spins < a.spin_limit) __builtin_amdgcn_s_sleep(4);
          // hand-off timed out (the contributor is not co-resident: CU masking, a shared GPU, another stream's kernels holding its CU):
          // never a silently wrong tile.  (1) this launch's epoch goes into the device's pinned status word, which the host reads at its
          // next synchronisation point (Net: behind BoxOutput's and the final stage's sync) and answers by forcing whole-tile
          // scheduling and running the frame again; (2) the tile is stored as NaN, and every ReLU behind M keeps a NaN a NaN.
          const bool timed_out = spins >= a.spin_limit;
          s_handoff_timeout = timed_out ? 1u : 0u;
          if (timed_out) {
            __hip_atomic_store(flags + a.G, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.status) __hip_atomic_store(a.status, (unsigned long long)a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const bool poisoned = s_handoff_timeout != 0;
        const unsigned sl2 = (unsigned)s2 * (unsigned)(C::BM * C::BN * 4);
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::NI; ++ni) {
            const unsigned vo = sl2 + (unsigned)((wm * C::WM + mi * 32 + 4 * khalf) * C::BN + wn * C::WN + ni * 32 + l31) * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[Q][mi][ni][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW, vo, (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)(C::BN * 4), 0));
            if (poisoned)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[Q][mi][ni][r] = __builtin_nanf("");
          }
        __syncthreads();                                     // (s_handoff_timeout is rewritten for the next contributor)
      }
    }
    // the last tile's stores
    if (!(ABL & 2) || a.tiles < 0) static_for<0, NMF * 16>([&](auto ec) { store_one(qc, ec); });      // (ablation builds keep the MFMAs alive)
  };
  if (par == 1) finish(std::integral_constant<int, 0>{});
  else finish(std::integral_constant<int, 1>{});
  if (old_pub) publish_flag();
  if (a.dbg && tid == 0) {
    const unsigned long long end_r = __builtin_amdgcn_s_memrealtime();
    a.dbg[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - dbg_c;
    a.dbg[blockIdx.x * 2 + 1] = end_r - dbg_r;
    a.dbg[2 * a.G + blockIdx.x * 2] = dbg_r;              // absolute start / end (100 MHz): launch stagger and drain of the grid
    a.dbg[2 * a.G + blockIdx.x * 2 + 1] = end_r;
  }
}

typedef void (*WgemmFn)(WgemmArgs);
struct WEntry { const char* name; int variant, abl, BM, BN, CK, threads; WgemmFn fn; };
#define WG_ENTRY(name, v, abl, BM, BN, WGM, WGN, CK) {name, v, abl, BM, BN, CK, WGM * WGN * 64, wgemm_kernel<WCfg<BM, BN, WGM, WGN, CK, 3>, abl>}
// variant + 64: the same tile with the bias / ReLU epilogue (abl field 64: never matched by the ablation look-up): InnerProduct on this kernel
#define WG_ENTRY_EPI(name, v, BM, BN, WGM, WGN, CK) {name, v + 64, 0, BM, BN, CK, WGM * WGN * 64, wgemm_kernel<WCfg<BM, BN, WGM, WGN, CK, 3>, 0, 1>}
#define WG_ENTRY_S(name, v, BM, BN, WGM, WGN, CK, DPG, SK) {name, v, 0, BM, BN, CK, WGM * WGN * 64, wgemm_kernel<WCfg<BM, BN, WGM, WGN, CK, 3, DPG, SK>, 0>}
const WEntry kW[] = {
    WG_ENTRY("wgemm_256x128_ck32", 1, 0, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_128x256_ck32", 2, 0, 128, 256, 2, 4, 32),
    WG_ENTRY("wgemm_128x128_ck32", 3, 0, 128, 128, 2, 4, 32),
    WG_ENTRY("wgemm_256x96_ck32", 4, 0, 256, 96, 8, 1, 32),       // conv5_x: 480 tile columns = 5 x 96, 250 tiles in one full round
    WG_ENTRY("wgemm_256x160_ck32", 5, 0, 256, 160, 8, 1, 32),
    WG_ENTRY_EPI("wgemm_256x128_ck32_epi", 1, 256, 128, 4, 2, 32),     // conv4_x F(4x4,3x3): 1080 columns = 6.75 x 160 -> 504 tiles = 1.97 rounds (r4)
#ifdef MSCNN_WGEMM_DEV      // schedule A/B (pieces per group, first store group)
    WG_ENTRY_S("wgemm_256x128_ck32_d2", 25, 256, 128, 4, 2, 32, 2, 0),
    WG_ENTRY_S("wgemm_256x128_ck32_d2_s3", 26, 256, 128, 4, 2, 32, 2, 3),
    WG_ENTRY_S("wgemm_256x128_ck32_d3_s2", 27, 256, 128, 4, 2, 32, 3, 2),
    WG_ENTRY_S("wgemm_256x128_ck32_d1_s6", 28, 256, 128, 4, 2, 32, 1, 6),
#endif
#ifdef MSCNN_WGEMM_DEV      // development ablations (tools/micro/wgemm_bench.hip): bit 0 no loads, 1 no stores, 2 no MFMAs, 3 no LDS reads, 4 no barriers
    WG_ENTRY("wgemm_256x128_ck32", 1, 1, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_256x128_ck32", 1, 2, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_256x128_ck32", 1, 3, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_256x128_ck32", 1, 4, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_256x128_ck32", 1, 11, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_256x128_ck32", 1, 19, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_256x128_ck32", 1, 27, 256, 128, 4, 2, 32),
    WG_ENTRY("wgemm_128x256_ck32", 2, 1, 128, 256, 2, 4, 32),
    WG_ENTRY("wgemm_128x256_ck32", 2, 2, 128, 256, 2, 4, 32),
    WG_ENTRY("wgemm_128x256_ck32", 2, 3, 128, 256, 2, 4, 32),
#endif
};

}  // namespace

namespace mscnn {

// The persistent grid is one workgroup per CU and the stream-K hand-off needs every workgroup co-resident: take the CU count of the
// current device (256 on an MI355X in SPX mode; fewer in a partition mode), 256 when no device is visible (planning on a CPU box).
static int device_cus(int* dev_out) {
  int dev = 0, n = 0;
  *dev_out = -1;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  *dev_out = dev;
  return n;
}

// ---- hand-off health (process-wide state; see the kernel's time-out branch) ---------------------------------------------------
namespace {
std::atomic<int> g_whole_tiles{0};                 // mscnn_wgemm_force_whole_tiles: no launch splits a tile any more
std::atomic<unsigned> g_spin_limit{1u << 22};      // polls of a contributor's flag before the finisher gives up (~0.5 s)
std::atomic<int> g_drop_publish{0};                // fault injection
std::atomic<unsigned long long*> g_status[64];     // one pinned host word per device, allocated by the first split launch on it
}  // namespace

static unsigned long long* status_word(int dev) {
  if (dev < 0 || dev >= 64) return nullptr;
  unsigned long long* w = g_status[dev].load(std::memory_order_acquire);
  if (w) return w;
  void* p = nullptr;
  if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || !p) {
    (void)hipGetLastError();
    return nullptr;
  }
  *static_cast<volatile unsigned long long*>(p) = 0ull;
  unsigned long long* expected = nullptr;
  if (!g_status[dev].compare_exchange_strong(expected, static_cast<unsigned long long*>(p))) {
    (void)hipHostFree(p);
    return expected;
  }
  return static_cast<unsigned long long*>(p);
}

unsigned long long wgemm_handoff_event() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (dev < 0 || dev >= 64) return 0;
  const unsigned long long* w = g_status[dev].load(std::memory_order_acquire);
  return w ? *static_cast<const volatile unsigned long long*>(w) : 0ull;
}
void wgemm_force_whole_tiles(int on) { g_whole_tiles.store(on ? 1 : 0); }
int wgemm_whole_tiles_forced() { return g_whole_tiles.load(); }
void wgemm_debug_handoff_fault(int drop_publish, unsigned spin_limit) {
  g_drop_publish.store(drop_publish ? 1 : 0);
  g_spin_limit.store(spin_limit ? spin_limit : (1u << 22));
}

bool wgemm_plan(int P, int Cout, int Cin, int T, int variant, WgemmPlan* o) {
  const int variant_flags = variant >> 8;
  variant &= 255;
  if (variant == 0) {
    variant = Cout >= 256 ? 1 : 2;
    // the other 256-row tile shapes where the model below says their tile count fits the CUs better than 256 x 128 (>= 5 %):
    //   256 x 96  -- conv5_x of the 7s-576 net: 480 columns = 250 tiles in one full round instead of 200 tiles of 512 padded columns (r3);
    //   256 x 160 -- conv4_x / loss1_conv1 in the F(4x4,3x3) form: 1080 columns = 7 x 160 (3.7 % padding) -> 504 tiles = two nearly full
    //                rounds of whole tiles, instead of 648 tiles of 128 (6.7 % padding) = 2.53 rounds with a stream-K hand-off (r4).
    // (the three share BM = 256, i.e. one packed-weight layout: a per-frame ROI count may flip the choice without a re-pack)
    if (variant == 1 && !variant_flags) {
      WgemmPlan p1, pc;
      if (wgemm_plan(P, Cout, Cin, T, 1, &p1)) {
        double best = 0.95 * p1.model_us;
        for (int cand : {4, 5})
          if (wgemm_plan(P, Cout, Cin, T, cand, &pc) && pc.model_us < best) { best = pc.model_us; variant = cand; }
        // a problem that does not even give every other CU a 256 x 128 tile (conv6_1: 50 tiles): 128 x 128 tiles, split stream-K style,
        // put four times as many workgroups on useful chunks -- 39.7 -> 27.4 us stand-alone (profiles/r04_ab_wgemm_tiles.txt).  Only
        // there: per FLOP the small tile moves twice the operands, which the model's MFMA-bound chunk time does not see.
        if ((long)P * p1.MT * p1.NT * 2 < p1.G && wgemm_plan(P, Cout, Cin, T, 3, &pc) && pc.model_us < best) { best = pc.model_us; variant = 3; }
      }
    }
  }
  const WEntry* e = nullptr;
  for (const WEntry& w : kW) if (w.variant == variant && w.abl == 0) e = &w;
  if (!e || Cin % e->CK != 0 || Cout % 32 != 0) return false;
  o->P = P; o->Cout = Cout; o->Cin = Cin; o->T = T;
  o->BM = e->BM; o->BN = e->BN; o->CK = e->CK;
  o->T_pad = (T + e->BN - 1) / e->BN * e->BN;
  o->MT = (Cout + e->BM - 1) / e->BM; o->NT = o->T_pad / e->BN; o->KI = Cin / e->CK;
  o->G = device_cus(&o->device);          // one 512-thread workgroup per CU (of the device that is current NOW: the launch checks it)
  const long tiles = (long)P * o->MT * o->NT;
  // Whole tiles (ceil(tiles / G) rounds) or the hybrid stream-K split of the last partial round?  Fitted to the measurements of
  // profiles/r03_wgemm.txt (256 x 128 x 32 chunks: 3.7 us each at the sustained clock, ~6 us per tile for its ride-along stores,
  // ~18 us for a workgroup's slab hand-off): the split wins for conv4_2 / conv4_3 / loss1 in the F(4x4,3x3) form (648 tiles = 2.53
  // rounds: 173 vs 187 us), roi_c1 (1100 tiles: 548 vs 605) and conv6_1 (50 tiles: 38 vs 71); whole tiles win where the last round is
  // nearly full (conv4_2 F(3x3,3x3): 750 tiles = 2.93 rounds, 192 vs 196) -- the model reproduces each of those choices.
  const long rem = tiles % o->G;
  const double chunk_us = (double)e->BM * e->BN * e->CK / 128.0 / 2214.0;                       // MFMA-bound time of one K chunk
  const double whole = (double)((tiles + o->G - 1) / o->G) * (o->KI * chunk_us + 6.0);
  const double handoff_us = 18.0 * (double)(e->BM * e->BN) / (256.0 * 128.0);      // a slab hand-off moves BM x BN floats (128 KB: 18 us measured)
  const double split = (double)tiles * o->KI / o->G * chunk_us + (double)tiles / o->G * 6.0 + handoff_us;
  o->full_q = (int)(tiles / o->G);
  if (rem > 0 && !(split < 0.97 * whole)) o->full_q += 1;       // whole tiles: the last round is simply not full
  o->model_us = rem > 0 ? (split < 0.97 * whole ? split : whole) : whole;
  if (variant_flags & 1) o->full_q = (int)(tiles / o->G);       // development: force the split
  if (variant_flags & 2) o->full_q = (int)((tiles + o->G - 1) / o->G);   // ... or whole tiles
  o->variant = variant; o->name = e->name;
  o->packed_bytes = (size_t)P * o->MT * o->KI * e->CK * e->BM * 4;
  o->ws_bytes = tiles > (long)o->full_q * o->G ? (size_t)o->G * e->BM * e->BN * 4 + ((size_t)o->G + 1) * 8 : 0;     // slabs, flags, status
  // every buffer strictly below 2^31 bytes: the kernel's out-of-range sentinel for lane offsets is 0x80000000 (the empty flush of
  // segment 0, rows past Cout, the producer's past-the-end pieces) and must lie outside num_records of V, M, Up and the slabs --
  // larger problems (batch >= 7 at 576 x 1920 in the F(4x4,3x3) form) stay on the per-plane igemm GEMM
  const double lim = 2147483648.0 - 65536.0;
  if ((double)tiles * o->KI * o->G >= 2.0e9 || o->KI < 1) return false;      // wg_range's 32-bit product
  if ((double)P * Cin * o->T_pad * 4 >= lim || (double)P * Cout * o->T_pad * 4 >= lim || (double)o->packed_bytes >= lim ||
      (double)o->ws_bytes >= lim) return false;
  return true;
}

int wgemm_launch(const WgemmPlan& p, const float* Up, const float* V, float* M, float* ws, hipStream_t st, int abl, unsigned long long* dbg,
                 const float* bias, int relu, size_t m_valid_bytes) {
  const WEntry* e = nullptr;
  for (const WEntry& w : kW) if (w.variant == p.variant && w.abl == abl) e = &w;
  if (!e) return MSCNN_ERR_BAD_ARG;
  // The persistent grid (and the co-residency the stream-K hand-off relies on) was sized for the device that was current when the
  // plan was made: a plan is not portable between devices / partitions.
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  if (p.device >= 0 && dev != p.device) {
    set_error("wgemm: plan made on device %d launched on device %d (re-plan on the device that runs it)", p.device, dev);
    return MSCNN_ERR_BAD_ARG;
  }
  WgemmArgs a;
  a.Up = Up; a.V = V; a.M = M; a.ws = ws;
  a.P = p.P; a.MT = p.MT; a.NT = p.NT; a.KI = p.KI; a.Cin = p.Cin; a.Cout = p.Cout; a.T_pad = p.T_pad;
  a.tiles = p.P * p.MT * p.NT; a.G = p.G; a.abl = abl; a.dbg = dbg;
  a.full_q = p.full_q; a.ws_bytes = (unsigned)p.ws_bytes;
  // (after a reported hand-off time-out the host forces whole tiles: the last round is then simply not full and nothing is handed over)
  if (g_whole_tiles.load(std::memory_order_relaxed)) a.full_q = (a.tiles + p.G - 1) / p.G;
  const int rem = a.tiles - a.full_q * p.G;
  if (rem > 0 && !ws) { set_error("wgemm: workspace missing"); return MSCNN_ERR_WORKSPACE; }
  a.status = rem > 0 ? status_word(dev) : nullptr;
  a.spin_limit = g_spin_limit.load(std::memory_order_relaxed);
  a.drop_publish = g_drop_publish.load(std::memory_order_relaxed);
  a.up_bytes = (unsigned)p.packed_bytes; a.v_bytes = (unsigned)((size_t)p.P * p.Cin * p.T_pad * 4); a.m_bytes = (unsigned)((size_t)p.P * p.Cout * p.T_pad * 4);
  if (m_valid_bytes) a.m_bytes = (unsigned)m_valid_bytes;      // (rows beyond the caller's buffer: their stores fall outside num_records and are dropped)
  a.bias = bias; a.relu = relu;
  // hand-off tag of this launch: unique within the process (monotonic), seeded per process so that flags a previous process left
  // in recycled device memory cannot match either; the workspace needs no clearing between launches
  static std::atomic<unsigned> g_epoch{(unsigned)std::chrono::steady_clock::now().time_since_epoch().count() * 2654435761u};
  unsigned ep = ++g_epoch;
  if (ep == 0) ep = ++g_epoch;
  a.epoch = ep;
  e->fn<<<p.G, e->threads, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn

extern "C" unsigned long long mscnn_wgemm_handoff_event(void) { return mscnn::wgemm_handoff_event(); }
extern "C" void mscnn_wgemm_force_whole_tiles(int on) { mscnn::wgemm_force_whole_tiles(on); }
extern "C" int mscnn_wgemm_whole_tiles_forced(void) { return mscnn::wgemm_whole_tiles_forced(); }
extern "C" void mscnn_debug_wgemm_handoff_fault(int drop_publish, unsigned spin_limit) { mscnn::wgemm_debug_handoff_fault(drop_publish, spin_limit); }
