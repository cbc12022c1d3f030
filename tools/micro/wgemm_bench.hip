// Stand-alone bench + check of the Winograd-domain batched GEMM  M[p] = U[p] x V[p]  (P planes, [Cout x Cin] x [Cin x T_pad]):
// the development harness of mscnn_amd/csrc/wgemm.h (the kernel is included from there, so what is measured here is what ships).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../mscnn_amd/csrc wgemm_bench.hip -o wgemm_bench
//   ./wgemm_bench P Cout Cin T [variant] [iters] [check]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "wgemm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void ref_gemm(const float* U, const float* V, float* M, int Cout, int Cin, int T_pad) {
  // U plain [p][Cout][Cin]; one thread per (p, co, t)
  const long i = blockIdx.x * 256L + threadIdx.x;
  const int t = (int)(i % T_pad); long r = i / T_pad;
  const int co = (int)(r % Cout); const int p = (int)(r / Cout);
  const float* u = U + ((long)p * Cout + co) * Cin;
  const float* v = V + (long)p * Cin * T_pad + t;
  float acc = 0.f;
  for (int k = 0; k < Cin; ++k) acc = fmaf(u[k], v[(long)k * T_pad], acc);
  M[i] = acc;
}

__global__ void pack_u(const float* U, float* Up, int P, int Cout, int Cin, int BM, int CKc) {
  // Up[p][mt][kc][ck][BM]
  const int MT = (Cout + BM - 1) / BM, KI = (Cin + CKc - 1) / CKc;
  const long total = (long)P * MT * KI * CKc * BM;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i;
    const int m = (int)(r % BM); r /= BM;
    const int ck = (int)(r % CKc); r /= CKc;
    const int kc = (int)(r % KI); r /= KI;
    const int mt = (int)(r % MT); const int p = (int)(r / MT);
    const int co = mt * BM + m, ci = kc * CKc + ck;
    Up[i] = (co < Cout && ci < Cin) ? U[((long)p * Cout + co) * Cin + ci] : 0.f;
  }
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 25, Cout = argc > 2 ? atoi(argv[2]) : 512, Cin = argc > 3 ? atoi(argv[3]) : 512;
  const int T = argc > 4 ? atoi(argv[4]) : 1920;
  const int variant = argc > 5 ? atoi(argv[5]) : 0, iters = argc > 6 ? atoi(argv[6]) : 20, check = argc > 7 ? atoi(argv[7]) : 1;
  const int abl = argc > 8 ? atoi(argv[8]) : 0;
  mscnn::WgemmPlan pl;
  if (!mscnn::wgemm_plan(P, Cout, Cin, T, variant, &pl)) { printf("no plan for variant %d\n", variant); return 1; }
  const int T_pad = pl.T_pad;
  const size_t nU = (size_t)P * Cout * Cin, nV = (size_t)P * Cin * T_pad, nM = (size_t)P * Cout * T_pad;
  float *U, *Up, *V, *M, *Mr, *ws = nullptr;
  CK(hipMalloc(&U, nU * 4)); CK(hipMalloc(&Up, pl.packed_bytes)); CK(hipMalloc(&V, nV * 4)); CK(hipMalloc(&M, nM * 4)); CK(hipMalloc(&Mr, nM * 4));
  if (pl.ws_bytes) CK(hipMalloc(&ws, pl.ws_bytes));
  std::vector<float> h(std::max(nU, nV));
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (size_t i = 0; i < nU; ++i) h[i] = rnd();
  CK(hipMemcpy(U, h.data(), nU * 4, hipMemcpyHostToDevice));
  for (size_t i = 0; i < nV; ++i) h[i] = ((i % T_pad) < (size_t)T) ? rnd() : 0.f;
  CK(hipMemcpy(V, h.data(), nV * 4, hipMemcpyHostToDevice));
  pack_u<<<2048, 256>>>(U, Up, P, Cout, Cin, pl.BM, pl.CK);
  CK(hipMemset(M, 0xff, nM * 4));
  CK(hipDeviceSynchronize());
  int rc = mscnn::wgemm_launch(pl, Up, V, M, ws, nullptr, abl);
  if (rc) { printf("launch failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  if (check && !abl) {
    ref_gemm<<<(unsigned)(nM / 256), 256>>>(U, V, Mr, Cout, Cin, T_pad);
    CK(hipDeviceSynchronize());
    std::vector<float> a(nM), b(nM);
    CK(hipMemcpy(a.data(), M, nM * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), Mr, nM * 4, hipMemcpyDeviceToHost));
    double worst = 0; size_t bad = 0, wi = 0;
    for (size_t i = 0; i < nM; ++i) {
      const double d = std::fabs((double)a[i] - b[i]) / std::max(1.0, std::fabs((double)b[i]));
      if (!(d <= 1e-4)) ++bad;
      if (!(d <= worst)) { worst = d; wi = i; }
    }
    printf("check: worst rel err %.3g at %zu (got %g want %g), %zu bad of %zu\n", worst, wi, a[wi], b[wi], bad, nM);
    if (bad) {
      printf("ok map of plane 0, rows 0..39 (one char per row: number of correct columns among the first 64, in hex/4):\n");
      for (int r = 0; r < 40; ++r) { int okc = 0; for (int c = 0; c < 64; ++c) { const size_t i = (size_t)r * T_pad + c; okc += std::fabs((double)a[i] - b[i]) <= 1e-4 * std::max(1.0, std::fabs((double)b[i])); } printf("row %2d: %2d/64 ok   got[0..3] %9.4f %9.4f %9.4f %9.4f  want %9.4f %9.4f %9.4f %9.4f\n", r, okc, a[(size_t)r*T_pad], a[(size_t)r*T_pad+1], a[(size_t)r*T_pad+2], a[(size_t)r*T_pad+3], b[(size_t)r*T_pad], b[(size_t)r*T_pad+1], b[(size_t)r*T_pad+2], b[(size_t)r*T_pad+3]); }
      return 2;
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) mscnn::wgemm_launch(pl, Up, V, M, ws, nullptr, abl);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) mscnn::wgemm_launch(pl, Up, V, M, ws, nullptr, abl);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* dbg; CK(hipMalloc(&dbg, 4096 * 16)); CK(hipMemset(dbg, 0, 4096 * 16));
  mscnn::wgemm_launch(pl, Up, V, M, ws, nullptr, abl);
  mscnn::wgemm_launch(pl, Up, V, M, ws, nullptr, abl);
  CK(hipEventRecord(e0));
  mscnn::wgemm_launch(pl, Up, V, M, ws, nullptr, abl, dbg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> hd(pl.G * 4); CK(hipMemcpy(hd.data(), dbg, pl.G * 32, hipMemcpyDeviceToHost));
  double cyc = 0, tick = 0; for (int g = 0; g < pl.G; ++g) { cyc = std::max(cyc, (double)hd[2 * g]); tick = std::max(tick, (double)hd[2 * g + 1]); }
  printf("   longest workgroup: %.0f shader cycles in %.1f us = %.0f MHz\n", cyc, tick / 100.0, cyc / (tick / 100.0));
  {
    unsigned long long s0 = ~0ull, s1 = 0, e0_ = ~0ull, e1_ = 0;
    for (int g = 0; g < pl.G; ++g) { const unsigned long long a_ = hd[2 * pl.G + 2 * g], b_ = hd[2 * pl.G + 2 * g + 1]; if (!a_) continue; s0 = std::min(s0, a_); s1 = std::max(s1, a_); e0_ = std::min(e0_, b_); e1_ = std::max(e1_, b_); }
    printf("   grid: starts spread over %.1f us, ends over %.1f us, first start -> last end %.1f us; this launch by events %.1f us\n", (s1 - s0) / 100.0, (e1_ - e0_) / 100.0, (e1_ - s0) / 100.0, ms1 * 1e3);
  }
  const double us = ms * 1e3 / iters;
  const double fl = 2.0 * P * Cout * (double)Cin * T, flp = 2.0 * P * pl.MT * pl.BM * (double)Cin * pl.NT * pl.BN;
  printf("P=%d Cout=%d Cin=%d T=%d(T_pad %d) variant=%d %s abl=%d: tiles %d grid %d  %.1f us  %.1f TFLOP/s real (%.3f of 157.3), %.1f incl. padding\n", P, Cout, Cin,
         T, T_pad, variant, pl.name, abl, pl.P * pl.MT * pl.NT, pl.G, us, fl / us / 1e6, fl / us / 1e6 / 157.3, flp / us / 1e6);
  return 0;
}
