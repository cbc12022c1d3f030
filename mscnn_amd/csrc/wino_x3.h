// Internal interface of wino_x3.hip: Winograd F(3x3,3x3) whose 25 plane GEMMs run on the fp16 MFMA pipe with every fp32
// operand split exactly into two fp16 halves (x * s = hi + lo, s a power of two chosen from the tensor's max |x|) and
// three products per pair (hi*hi + hi*lo + lo*hi, fp32 accumulators): 22-bit significands, fp32-grade results at 16/3 of the
// fp32 MFMA rate.  Opt-in (MSCNN_CONV_ALGO_WINO_F3_X3); see wino_x3.hip for layouts and the error analysis.
#pragma once
#include "common.h"

namespace mscnn {

struct X3Plan {
  int Cin = 0, Cout = 0, KG = 0, Cout_pad = 0, BM = 0, MT = 0, NT = 0;
  long T_pad = 0;
  size_t packed_bytes = 0;   // header + U16
  size_t v_bytes = 0;        // V16 planes
  size_t scal_bytes = 4096;  // kAmaxSlots partial maxima of |x| in front of the workspace (when the plan measures them itself)
};

// Proposal heads (small Cout, K x K, stride 1) as ONE GEMM T[tap * Cout + co][pixel] + a shift-and-add over the taps (wino_x3.hip)
struct X3HeadPlan {
  int Cin = 0, Cout = 0, KH = 0, KW = 0, KG = 0, rows = 0, rows_pad = 0;
  long T_pad = 0;
  size_t packed_bytes = 0, t_bytes = 0;     // workspace = 4096 (own max |x| slots) + t_bytes
};
bool x3_head_plan(int Cin, int Cout, int KH, int KW, long HW, X3HeadPlan* out);
int x3_head_pack(const X3HeadPlan& p, const float* w, void* packed, hipStream_t st);
int x3_head_forward(const X3HeadPlan& p, const float* x, const void* packed, const float* bias, float* y, int H, int W, int Ho, int Wo,
                    int pad_h, int pad_w, int relu, const unsigned* in_bound, void* ws, hipStream_t st);
// y[co] = bias[co] + sum over the taps of T[tap * Cout + co] shifted by the tap's offset (zero padding), rows of T_pad floats
int head_shift_add(const float* T, const float* bias, float* y, int Cout, int H, int W, int Ho, int Wo, int KH, int KW, int pad_h,
                   int pad_w, unsigned T_pad, int relu, hipStream_t st);

// false when the shape is not covered (Cin not a multiple of 32, planes beyond the 32-bit buffer window)
bool x3_plan(int Cin, int Cout, long T_pad, int tune_variant, X3Plan* out);

int x3_pack_weights(const X3Plan& p, const float* w, void* packed, hipStream_t st);

// scal[0 .. kAmaxSlots) <- partial maxima (bit patterns) of |x| over the n floats of x; their maximum is max |x|
int x3_amax(const float* x, long n, unsigned* scal, hipStream_t st);

// V16[plane][part][kg][t][8] = split(s * (B^T d B)); H * W <= 64: the ROI-map form (tiles of one ROI are consecutive t)
int x3_input_transform(const X3Plan& p, const float* x, void* V16, const unsigned* scal, int N, int H, int W, int pad_h,
                       int pad_w, int tiles_h, int tiles_w, hipStream_t st);

// M[plane][co][t] (fp32, the layout of the fp32 Winograd path) = (U V)(plane) / (s_U s_V)
int x3_gemm(const X3Plan& p, const void* packed, const void* V16, float* M, const unsigned* scal, int xcd_map, hipStream_t st);

}  // namespace mscnn
