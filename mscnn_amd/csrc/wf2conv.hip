// conv1_2 of the VGG trunk (3x3 / pad 1 / stride 1, 64 -> 64 channels on the full-resolution map: 81.5 GFLOP as a direct convolution,
// the largest single kernel of a 7s-576 frame) as a ONE-LAUNCH Winograd F(2x2,3x3) on gfx950.  Replaces ConvolutionLayer::Forward_gpu
// (src/caffe/layers/conv_layer.cu:8-23) + the in-place ReLU (relu_layer.cu:17-26) + the MAX 2x2 pooling that follows
// (pooling_layer.cu:11-47) for that shape class.
//
// Why.  The three-kernel Winograd pipeline of the other trunk layers (input transform -> plane GEMMs -> output transform) does not pay
// here: with 64 channels on a 576 x 1920 map the V / M planes are 0.64 GB each way and the layer becomes HBM-bound at about the time
// of the direct kernel (wino_plan's table in conv.hip: 1175 vs 762 us in round 2).  But 64 x 64 channels is also the one shape where
// EVERYTHING a workgroup needs fits on a CU: the transformed filters of an 8-channel chunk are 32 KB, and the 16 plane products of a
// 64-channel x 64-tile block are 256 KB of fp32 accumulators = half the CU's vector registers.  So one workgroup does the whole
// algorithm for an 8 x 32 block of output pixels:
//     per 8-channel chunk:  x patch (10 x 34 per channel, fetched as 10 x 40: 16-byte LDS-DMA lanes) -> LDS;  V = B^T d B per (channel, tile) on the vector ALU -> LDS;
//                           M[p] += U[p] (64 x 8) x V[p] (8 x 64) for the 16 planes p on v_mfma_f32_16x16x4_f32
//     after the last chunk: y = A^T M A + bias, ReLU, the 2x2 tile IS one pooling window: max -> the pooled map
// and HBM sees x once (x 1.33 for the halo), the filters from L2, and y / the pooled map once.  Executed MFMA work: 2 x 16 x 64 x 64
// per 4 output pixels = 36.2 GFLOP per 7s-576 frame instead of 81.5 (2.25x fewer multiplies).
//
// Layout of a workgroup (512 threads = 8 waves, two per SIMD):
//   * tile = 4 x 16 Winograd tiles (n = row * 16 + column), 64 output channels; wave (ch, tr) owns channels 32 ch .. + 31 (two 16-row
//     MFMA blocks) x the 16 tiles of tile row tr, for all 16 planes: 16 x 2 accumulators of 4 registers = 128 VGPRs per lane;
//   * LDS: two stages of {U chunk [k 8][xi 4][cout 64][nu 4], V chunk [k 8][xi 4][tile 64][nu 4]} (32 KB each) + two patch buffers
//     (8 channels x 10 rows x 40 columns -- the 34 needed widened to aligned float4s -- in whole 1 KB LDS-DMA pieces: 13 KB each) + the
//     DMA dump and the biases: ~156 KB of the CU's 160 (the static_asserts below).  The four nu of one (k, xi, row) are ONE
//     ds_read_b128: 3 reads feed 8 MFMAs;
//   * per chunk c, between two barriers: global loads of U chunk c + 1 and patch c + 2 are issued, patch c + 1 is transformed into
//     the other V stage, the 64 MFMAs per wave of chunk c run, the loaded registers are written to LDS.  One barrier per chunk.
// Numerics: F(2x2,3x3) with the standard points {0, 1, -1, inf} (B^T, G, A^T below): the mildest of the Winograd forms used here
// (~2x the rounding error of the direct sum); the layer's own first-forward check against the direct kernel applies as to the others.
#include "wf2conv.h"
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Wf2Args {
  const float* x; const float* up; const float* bias; float* y; float* yp;
  int N, Cin, H, W, NTH, NTW, KI, relu, tiles;
  int dG_w, dG_h, dG_n;         // the grid size G as a step of the (column, row, image) tile cursor
  unsigned x_bytes;
};

#ifndef WF2_ABL
#define WF2_ABL 0      // development ablations (make wf2abl_<n>): bit 0 no MFMAs, 1 no LDS operand reads, 2 no input transform, 3 no global loads, 4 no barrier in the loop
#endif
#ifdef WF2_TRACE      // development (make wf2trace): shader-clock stamps of workgroup 0, waves 0 and 4, units 8 .. 39
__device__ unsigned long long g_wf2_trace[2 * 32 * 8];
#define WF2_STAMP(k) do { if (slot == 0 && (wave & 3) == 0 && lane == 0 && u >= 8 && u < 40) g_wf2_trace[((wave >> 2) * 32 + (u - 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WF2_STAMP(k) do { } while (0)
#endif
constexpr int TR = 4, TC = 16, NT = TR * TC;            // Winograd tiles per workgroup tile: 4 rows x 16 columns = 8 x 32 output pixels
constexpr int PR = 2 * TR + 2, PC = 2 * TC + 8;        // input patch per channel: 10 rows x 40 columns = the 34 needed (w0 - 1 .. w0 + 32) widened to
                                                       // whole aligned float4s (w0 - 4 .. w0 + 35): the patch travels as 16-byte LDS-DMA lanes
constexpr int PC4 = PC / 4, PX = 3;                    // float4s per row; column of w0 - 1 inside a row
constexpr int CK = 8;                                  // channels per chunk
constexpr int U_FLOATS = CK * 16 * 64, V_FLOATS = CK * 16 * NT, P_FLOATS = CK * PR * PC;      // 8192, 8192, 3200
constexpr int P_STRIDE = (P_FLOATS + 255) / 256 * 256;   // a patch buffer holds whole 1 KB LDS-DMA pieces: 13 x 256 = 3328 floats (the last piece's tail is never read)
constexpr int LDS_FLOATS = 2 * U_FLOATS + 2 * V_FLOATS + 2 * P_STRIDE;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");

// U[cin][xi][cout][nu] = (G g G^T)[xi][nu],  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
__global__ __launch_bounds__(256) void wf2_weight_kernel(const float* __restrict__ w, float* __restrict__ up, int Cin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * Cin) return;
  const int co = i % 64, ci = i / 64;
  const float* g = w + ((long)co * Cin + ci) * 9;
  float t[4][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
    t[0][b] = g0; t[1][b] = 0.5f * (g0 + g1 + g2); t[2][b] = 0.5f * (g0 - g1 + g2); t[3][b] = g2;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
    *reinterpret_cast<float4*>(up + (((long)ci * 4 + a) * 64 + co) * 4) = make_float4(u0, u1, u2, u3);
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOob = 0x80000000u;
// LDS-DMA: lane l copies 16 (4) bytes from rsrc + voff(l) + soff to LDS lds_addr + 16 l (4 l); a lane whose offset is out of range reads
// zero (= the convolution's zero padding).  Hidden from the compiler's waitcnt bookkeeping by design: waited for by hand (dma_wait) in
// front of the unit's barrier.  s_nop 4: SGPR written by v_readfirstlane -> VMEM read; s_nop 0: M0 write -> LDS-DMA.
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
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int P_PIECES = (P_FLOATS + 255) / 256;       // 1 KB LDS-DMA pieces of a patch (64 lanes x 16 bytes): 13
constexpr int P_SLOTS = (P_PIECES + 7) / 8;            // per wave: 2
constexpr int LDS_SPARE = LDS_FLOATS * 4;              // 1 KB behind everything: where a wave without a real piece in a slot aims
static_assert(LDS_SPARE + 1024 + 256 <= 160 * 1024, "LDS");

__global__ __launch_bounds__(512, 2) void wf2conv_kernel(Wf2Args a) {
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS + 256 + 64];      // + the DMA dump + the 64 biases
  float* const Us = lds;                               // [2][U_FLOATS]
  float* const Vs = lds + 2 * U_FLOATS;                // [2][V_FLOATS]
  float* const Ps = lds + 2 * U_FLOATS + 2 * V_FLOATS; // [2][P_STRIDE]
  const unsigned lds0 = (unsigned)(size_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lq = lane >> 4;
  const int ch = wave & 1, tr = wave >> 1;             // MFMA phase: channel half, tile row
  // The two waves of a SIMD (w and w + 4) walk a unit in opposite order (see the loop below)
  const bool mfma_first = wave >= 4;
  const int HW = a.H * a.W;
  const int G = (int)gridDim.x, slot = (int)blockIdx.x;
  const int my_tiles = slot < a.tiles ? (a.tiles - slot + G - 1) / G : 0;
  const int total = my_tiles * a.KI;                   // (tile, chunk) units of this workgroup: one stream, no per-tile prologue
  if (total == 0) return;
  const __amdgpu_buffer_rsrc_t rU = make_rsrc(a.up, (unsigned)(a.KI * U_FLOATS * 4)), rX = make_rsrc(a.x, a.x_bytes);

  // ---- producer: LDS-DMA of the transformed filters (4 pieces of 1 KB per wave and unit) and of the input patch (13 pieces of 1 KB =
  // 64 lanes x 16 bytes each, two slots per wave: wave w moves pieces w and w + 8, the slot of a piece >= 13 is aimed at the dump)
  // this lane's float4 in slot j: number f = (wave + 8 j) * 64 + lane of the patch [k 8][r 10][c4 10]; vP[j] = its byte offset in x
  unsigned vP[P_SLOTS];
#pragma unroll
  for (int j = 0; j < P_SLOTS; ++j) vP[j] = kOob;
  // tile cursor of the producer (the tile of unit pu), advanced by G tiles at a time without divisions
  int p_tw = slot % a.NTW, p_th = (slot / a.NTW) % a.NTH, p_n = slot / (a.NTW * a.NTH);
  unsigned p_img = 0;            // byte offset of the tile's image in x
  int pu = 0;                    // next unit whose patch is put in flight
  auto enter_tile = [&]() {      // lane offsets of the patch elements for the tile at the cursor; then the cursor moves on
    const int h0 = p_th * (2 * TR), w0 = p_tw * (2 * TC);
    p_img = __builtin_amdgcn_readfirstlane((unsigned)(p_n * a.Cin) * (unsigned)HW * 4u);
#pragma unroll
    for (int j = 0; j < P_SLOTS; ++j) {      // (the float4's (k, r, c4) is recomputed per tile -- divisions by constants -- rather than held in registers)
      const int f = (wave + 8 * j) * 64 + lane;      // float4 number f of [k][10][10]
      const int k = f / (PR * PC4), rem = f - k * (PR * PC4), r = rem / PC4, c4 = rem - r * PC4;
      const int hh = h0 - 1 + r, ww = w0 - 4 + 4 * c4;      // W is a multiple of 32 and w0 of 32: a float4 is inside the row or outside, never across
      const bool ok = f < P_FLOATS / 4 && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
      vP[j] = ok ? (unsigned)(k * HW + hh * a.W + ww) * 4u : kOob;
    }
    p_tw += a.dG_w; if (p_tw >= a.NTW) { p_tw -= a.NTW; ++p_th; }
    p_th += a.dG_h; if (p_th >= a.NTH) { p_th -= a.NTH; ++p_n; }
    p_n += a.dG_n;
  };
  auto dma_patch = [&](int buf) {                      // the patch of unit pu -> patch buffer buf (nothing past the last unit)
    if (pu < total) {
      const int kc = pu % a.KI;
      if (kc == 0) enter_tile();
      const unsigned sb = p_img + (unsigned)(kc * CK) * (unsigned)HW * 4u;
      const unsigned pb = lds0 + (unsigned)((2 * U_FLOATS + 2 * V_FLOATS + buf * P_STRIDE) * 4);
#pragma unroll
      for (int j = 0; j < P_SLOTS; ++j) {
        const int piece = wave + 8 * j;
        dma16(rX, vP[j], sb, piece < P_PIECES ? pb + (unsigned)piece * 1024u : lds0 + (unsigned)LDS_SPARE);
      }
    }
    ++pu;
  };
  auto dma_u = [&](int kc, int stage) {                // U chunk kc -> U stage
    const unsigned ub = lds0 + (unsigned)(stage * U_FLOATS * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned piece = (unsigned)(wave * 4 + j);
      dma16(rU, (unsigned)lane * 16u, (unsigned)kc * (U_FLOATS * 4) + piece * 1024u, ub + piece * 1024u);
    }
  };
  // V = B^T d B of (channel k = wave, tile = lane) of patch buffer `buf` into V stage `stage`;  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  auto transform = [&](int buf, int stage) {
    const int trn = lane >> 4, tcn = lane & 15;
    const float* p = Ps + buf * P_STRIDE + wave * (PR * PC) + (2 * trn) * PC + PX + 2 * tcn;      // (odd column: four dword reads per row)
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = p[i * PC + j];
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
    }
    float4* V = reinterpret_cast<float4*>(Vs + stage * V_FLOATS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      V[(wave * 4 + i) * NT + lane] = make_float4(t[i][0] - t[i][2], t[i][1] + t[i][2], t[i][2] - t[i][1], t[i][1] - t[i][3]);
  };

  f32x4 acc[16][2];

  // the 64 MFMAs of a chunk in 8 groups (k-step ks, transform row i) of 8; the three operand reads of group g + 1 are issued BEFORE the
  // MFMAs of group g (two register sets)
  auto mfmas = [&](int st, int c, auto first_c) {      // first_c: the tile's first chunk -- the products START the sums (C = 0), no clearing pass
    constexpr bool FIRST = decltype(first_c)::value;
    const f32x4* U = reinterpret_cast<const f32x4*>(Us + st * U_FLOATS) + lq * 4 * 64 + ch * 32 + l16;
    const f32x4* V = reinterpret_cast<const f32x4*>(Vs + st * V_FLOATS) + lq * 4 * NT + tr * 16 + l16;
    f32x4 a0[2], a1[2], bb[2];
    auto rd = [&](int g, int buf) {                   // group g = ks * 4 + i: rows (ks * 4 + lq) * 4 + i of U / V
      const int off = ((g >> 2) * 16 + (g & 3)) * 64;
      if (WF2_ABL & 2) { a0[buf] = (f32x4){1.f, 2.f, 3.f, (float)c}; a1[buf] = a0[buf]; bb[buf] = a0[buf]; }
      else { a0[buf] = U[off]; a1[buf] = U[off + 16]; bb[buf] = V[off]; }
    };
    rd(0, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cur = g & 1, i = g & 3;
      if (g + 1 < 8) rd(g + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (WF2_ABL & 1) { acc[i * 4][0][0] = (FIRST && g == 0 ? 0.f : acc[i * 4][0][0]) + a0[cur][0] + a1[cur][1] + bb[cur][2]; continue; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool fresh = FIRST && g < 4;      // (groups 0-3 = k-step 0 touch every accumulator once)
        acc[i * 4 + q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[cur][q], bb[cur][q], fresh ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i * 4 + q][0], 0, 0, 0);
        acc[i * 4 + q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[cur][q], bb[cur][q], fresh ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i * 4 + q][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // output transform of a finished tile:  y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1];  lane holds (cout = 32 ch + 16 mb + 4 lq + r,
  // tile column l16).  (The accumulators need no clearing: the next tile's first chunk starts its sums with C = 0.)  The consumer's tile
  // cursor moves like the producer's.
  const int Hp = a.H / 2, Wp = a.W / 2;
  int c_tw = slot % a.NTW, c_th = (slot / a.NTW) % a.NTH, c_n = slot / (a.NTW * a.NTH);
  float* const bias_s = lds + LDS_FLOATS + 256;      // the 64 biases in LDS (a global load per tile would expose its latency; registers are short)
  if (tid < 64) bias_s[tid] = a.bias ? a.bias[tid] : 0.f;
  auto epilogue = [&]() {
    const int oh = c_th * (2 * TR) + 2 * tr, ow = c_tw * (2 * TC) + 2 * l16;
    const long co0 = (long)c_n * 64 + ch * 32 + lq * 4;
    float* ydst = a.y ? a.y + (co0 * a.H + oh) * a.W + ow : nullptr;
    float* pdst = a.yp ? a.yp + (co0 * Hp + (oh >> 1)) * Wp + (ow >> 1) : nullptr;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = acc[0 + j][mb][r], m1 = acc[4 + j][mb][r], m2 = acc[8 + j][mb][r], m3 = acc[12 + j][mb][r];
          s0[j] = m0 + m1 + m2;
          s1[j] = m1 - m2 - m3;
        }
        const float bv = bias_s[ch * 32 + mb * 16 + lq * 4 + r];
        const long cofs = (long)(mb * 16 + r);
        if (!ydst) {
          // only the pooled map is wanted: max first, bias and ReLU once -- the same bits as pooling the four finished outputs
          // (x -> fl(x + b) and ReLU are monotone, so they commute with max exactly; a NaN wins only from the first position, as below)
          const float q00 = s0[0] + s0[1] + s0[2], q01 = s0[1] - s0[2] - s0[3], q10 = s1[0] + s1[1] + s1[2], q11 = s1[1] - s1[2] - s1[3];
          float m = q00;
          if (q01 > m) m = q01;
          if (q10 > m) m = q10;
          if (q11 > m) m = q11;
          m += bv;
          if (a.relu) m = m < 0.f ? 0.f : m;
          pdst[cofs * (Hp * Wp)] = m;
          continue;
        }
        float y00 = s0[0] + s0[1] + s0[2] + bv, y01 = s0[1] - s0[2] - s0[3] + bv;
        float y10 = s1[0] + s1[1] + s1[2] + bv, y11 = s1[1] - s1[2] - s1[3] + bv;
        if (a.relu) {
          y00 = y00 < 0.f ? 0.f : y00; y01 = y01 < 0.f ? 0.f : y01; y10 = y10 < 0.f ? 0.f : y10; y11 = y11 < 0.f ? 0.f : y11;
        }
        if (ydst) {
          float* dst = ydst + cofs * HW;
          *reinterpret_cast<float2*>(dst) = make_float2(y00, y01);
          *reinterpret_cast<float2*>(dst + a.W) = make_float2(y10, y11);
        }
        if (pdst) {      // the tile is one MAX 2x2 / stride 2 window (pooling_layer.cpp:87-101: first maximum wins, NaN never does)
          float m = y00;
          if (y01 > m) m = y01;
          if (y10 > m) m = y10;
          if (y11 > m) m = y11;
          pdst[cofs * (Hp * Wp)] = m;
        }
      }
    c_tw += a.dG_w; if (c_tw >= a.NTW) { c_tw -= a.NTW; ++c_th; }
    c_th += a.dG_h; if (c_th >= a.NTH) { c_th -= a.NTH; ++c_n; }
    c_n += a.dG_n;
  };

  // ---- prologue (once per workgroup): U 0 and the patches of units 0 and 1 in LDS, V of unit 0 transformed
  dma_u(0, 0);
  dma_patch(0);
  dma_patch(1);
  dma_wait();
  __syncthreads();
  transform(0, 0);
  __syncthreads();

  // ---- one stream of (tile, chunk) units.  Unit u multiplies U / V of stage u & 1.  During it, by LDS-DMA (no registers, no
  // ds_write): U of unit u + 1 -> stage (u + 1) & 1 and the patch of unit u + 2 -> patch buffer u & 1 -- both were last read during
  // unit u - 1 --, waited for in front of the unit's barrier; and the patch of unit u + 1 (buffer (u + 1) & 1) is transformed into
  // the other V stage.  The fp32 MFMA and the vector ALU do not overlap on this part (a wave's VALU / LDS / VMEM instructions crawl
  // while its SIMD partner streams MFMAs: profiles/r05_wf2_trace.txt), so everything beside the MFMAs is kept to as few instructions
  // as possible, and the two waves of a SIMD walk a unit in opposite order so that the pipe always has a wave to take MFMAs from:
  //   waves 0-3:  DMA issue -> transform -> MFMAs -> [epilogue]
  //   waves 4-7:  MFMAs -> [epilogue] -> DMA issue -> transform
  for (int u = 0; u < total; ++u) {
    const int st = u & 1, c = u % a.KI;
    const bool more1 = u + 1 < total;
    WF2_STAMP(0);
    // (next to a partner that streams fp32 MFMAs a wave's other instructions take ~27 cycles each instead of ~12; raising the wave
    // priority around them changed nothing -- it is not arbitration --, tools/sessions/r05_s17.sh)
    if (!mfma_first) {
      if (!(WF2_ABL & 8)) {
        dma_u(c + 1 < a.KI ? c + 1 : 0, st ^ 1);       // U of unit u + 1
        dma_patch(st);                                 // patch of unit u + 2
      }
      if (more1 && !(WF2_ABL & 4)) transform(st ^ 1, st ^ 1);
    }
    WF2_STAMP(1);
    if (c == 0) mfmas(st, c, std::true_type{});
    else mfmas(st, c, std::false_type{});
    WF2_STAMP(2);
    if (c == a.KI - 1) epilogue();
    WF2_STAMP(3);
    if (mfma_first) {
      if (!(WF2_ABL & 8)) {
        dma_u(c + 1 < a.KI ? c + 1 : 0, st ^ 1);
        dma_patch(st);
      }
      if (more1 && !(WF2_ABL & 4)) transform(st ^ 1, st ^ 1);
    }
    WF2_STAMP(4);
    dma_wait();
    if (!(WF2_ABL & 16)) __syncthreads();
    WF2_STAMP(5);
  }
}

}  // namespace

namespace mscnn {

bool wf2_plan(int N, int Cin, int H, int W, int Cout, Wf2Plan* o) {
  if (Cout != 64 || Cin % CK != 0 || Cin < CK || N < 1 || H % (2 * TR) != 0 || W % (2 * TC) != 0) return false;
  if ((double)N * (Cin > 64 ? Cin : 64) * H * W * 4.0 >= 2147483648.0 - 65536.0) return false;      // x inside one buffer descriptor below the out-of-range sentinel
  o->N = N; o->Cin = Cin; o->H = H; o->W = W;
  o->NTH = H / (2 * TR); o->NTW = W / (2 * TC); o->KI = Cin / CK;
  o->tiles = N * o->NTH * o->NTW;
  o->packed_bytes = (size_t)16 * 64 * Cin * sizeof(float);
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    cus = 256;      // (planning on a box without a device)
  }
  o->cus = cus;
  return true;
}

const char* wf2_kernel_name() { return "winograd2x2_fused_k3x3_c64"; }

int wf2_pack(const Wf2Plan& p, const float* w, float* packed, hipStream_t st) {
  wf2_weight_kernel<<<cdiv(64 * p.Cin, 256), 256, 0, st>>>(w, packed, p.Cin);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int wf2_launch(const Wf2Plan& p, const float* x, const float* packed, const float* bias, float* y, float* y_pool, int relu, hipStream_t st) {
  MSCNN_REQUIRE(x && packed && (y || y_pool), "conv(winograd2x2 fused): null pointer");
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(packed) % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && (!y || reinterpret_cast<uintptr_t>(y) % 8 == 0),
                "conv(winograd2x2 fused): x and the packed weights must be 16-byte, y 8-byte aligned");
  Wf2Args a;
  a.x = x; a.up = packed; a.bias = bias; a.y = y; a.yp = y_pool;
  a.N = p.N; a.Cin = p.Cin; a.H = p.H; a.W = p.W; a.NTH = p.NTH; a.NTW = p.NTW; a.KI = p.KI; a.relu = relu; a.tiles = p.tiles;
  // persistent grid: one workgroup per CU (155 KB of LDS each), slot s takes the tiles s, s + G, ... (no workgroup waits for another:
  // the grid size is a performance choice, not a co-residency requirement)
  const int cus = p.cus;
  const int G = p.tiles < cus ? p.tiles : cus;
  a.dG_w = G % p.NTW; a.dG_h = (G / p.NTW) % p.NTH; a.dG_n = G / (p.NTW * p.NTH);
  a.x_bytes = (unsigned)((size_t)p.N * p.Cin * p.H * p.W * sizeof(float));
  wf2conv_kernel<<<G, 512, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn

#ifdef WF2_TRACE
extern "C" __attribute__((visibility("default"))) int mscnn_debug_wf2_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wf2_trace), sizeof(unsigned long long) * 2 * 32 * 8);
}
#endif
