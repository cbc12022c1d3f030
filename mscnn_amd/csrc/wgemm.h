// Internal interface of wgemm.hip: the batched transform-domain GEMM of the Winograd convolution paths,
//   M[p][Cout][T_pad] = U[p][Cout][Cin] x V[p][Cin][T_pad]     for the P = (m+2)^2 transform planes of F(m x m, 3x3).
// It replaces the 1x1 igemm kernel that ran these GEMMs in rounds 1-2 (conv.hip) -- see the header of wgemm.hip for why.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace mscnn {

struct WgemmPlan {
  int P, Cout, Cin, T, T_pad;       // T_pad: row stride of V and M (multiple of the N tile)
  int BM, BN, CK, MT, NT, KI;       // tile shape, tiles per plane, K chunks
  int G;                            // persistent grid (workgroups)
  int device;                       // the device G was taken from (-1: planned without one); wgemm_launch refuses another
  int full_q;                       // whole tiles per workgroup; the remaining tiles are split stream-K style
  int variant;
  double model_us;                  // the plan's own time estimate (picks between tile shapes)
  const char* name;
  size_t packed_bytes;              // U in the packed layout Up[p][mt][kc][ck][BM]
  size_t ws_bytes;                  // partial-tile slabs of the stream-K phase
};

// variant 0: pick per shape (bits 8 / 9 of variant: development -- force the stream-K split / whole tiles).  Returns false when no
// kernel covers the shape (the caller keeps the igemm path).
bool wgemm_plan(int P, int Cout, int Cin, int T, int variant, WgemmPlan* out);

// abl: development ablations (0 in the product): bit 0 no loads after the prologue, bit 1 no stores, bit 2 no MFMAs, bit 3 no
// LDS operand reads, bit 4 no barriers; dbg: per-workgroup {shader cycles, 100 MHz ticks} (tools/micro/wgemm_bench.hip)
// bias / relu: only for a plan of an epilogue variant (variant + 64: InnerProduct on this kernel; bias indexed by column);
// m_valid_bytes != 0: the bytes of M that exist (Cout was rounded up to the tile grid's 32-row blocks, the caller's buffer was not)
int wgemm_launch(const WgemmPlan& p, const float* Up, const float* V, float* M, float* ws, hipStream_t st, int abl = 0,
                 unsigned long long* dbg = nullptr, const float* bias = nullptr, int relu = 0, size_t m_valid_bytes = 0);

// Hand-off health (mscnn_hip.h: mscnn_wgemm_handoff_event & co).  event: the epoch of the last launch on the current device whose
// finisher gave up on a contributor (0: none yet) -- read it after synchronising the stream.
unsigned long long wgemm_handoff_event();
void wgemm_force_whole_tiles(int on);
int wgemm_whole_tiles_forced();
void wgemm_debug_handoff_fault(int drop_publish, unsigned spin_limit);

}  // namespace mscnn
