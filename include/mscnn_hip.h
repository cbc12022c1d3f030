/*
 * mscnn_hip.h -- C ABI of libmscnn_hip.so: the MI355X (gfx950) device ops of the MS-CNN
 * inference hot path.  This is the drop-in boundary: every entry point is
 * what a caffe::Layer<float>::Forward_gpu of the reference would bind for that layer.
 * Plain pointers and sizes only; all tensors are fp32, NCHW, contiguous (blob.hpp:153-164);
 * every pointer is a DEVICE pointer unless the parameter name ends in _host.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream, which is what every
 * reference layer uses, device_alternate.hpp:84-90).
 *
 * Return value: 0 = ok, non-zero = error (see mscnn_status).  Nothing here aborts; the C++
 * layer shim (mscnn_amd/host) turns a non-zero status into a CHECK-style fatal to mimic the
 * reference's glog convention (device_alternate.hpp:48-67).
 *
 * There is NO CPU fallback behind any of these functions.
 */
#ifndef MSCNN_HIP_H_
#define MSCNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define MSCNN_API __attribute__((visibility("default")))
#else
#define MSCNN_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MSCNN_OK = 0,
  MSCNN_ERR_BAD_ARG = 1,      /* shape / parameter the kernel cannot honour */
  MSCNN_ERR_HIP = 2,          /* a HIP runtime call or launch failed; see mscnn_last_error() */
  MSCNN_ERR_WORKSPACE = 3,    /* workspace too small; query the *_workspace_bytes function */
  MSCNN_ERR_UNSUPPORTED = 4
} mscnn_status;

/* Text of the last error on this host thread (static storage, never NULL). */
MSCNN_API const char* mscnn_last_error(void);
/* Library / device identification: "mscnn_hip <ver> gfx950". */
MSCNN_API const char* mscnn_version(void);

/* ------------------------------------------------------------------------------------------
 * Convolution  -- replaces ConvolutionLayer<Dtype>::Forward_gpu (src/caffe/layers/conv_layer.cu:8-23
 * -> base_conv_layer.cpp:325-349: im2col_gpu + cublasSgemm + bias GEMM) and
 * CuDNNConvolutionLayer::Forward_gpu (cudnn_conv_layer.cu:11-46), fused with the in-place
 * ReLULayer::Forward_gpu that follows it in every deploy net (relu_layer.cu:17-26) when relu != 0.
 * Cross-correlation, weights w[Cout][Cin/group][Kh][Kw] (base_conv_layer.cpp:135-140), bias may be NULL.
 *
 * The hot shapes (stride 1, group 1) run on an im2col-free implicit-GEMM MFMA kernel that reads a
 * pre-packed copy of the weights: create a plan once per layer (weights are constants at inference),
 * call mscnn_conv2d_pack_weights whenever the Caffe weight blob changes, then mscnn_conv2d_fwd_f32.
 * ------------------------------------------------------------------------------------------ */
typedef struct mscnn_conv_plan mscnn_conv_plan;   /* opaque */

typedef enum {
  /* The default.  3x3 / stride 1 / group 1 layers with enough arithmetic intensity run a Winograd form: F(4x4,3x3) (36 planes, points
   * {0, 1, -1, 2, -1/2, inf}) where the layer has >= 1000 tiles of 4x4 (conv2_1 .. conv4_3, loss1_conv1 of the 7s-576 net), F(3x3,3x3)
   * (25 planes) below that and on 7x7 ROI maps (conv5_x, conv6_1, roi_c1); the plane GEMMs run on wgemm.hip's kernel.  Cin = 3 runs a
   * VALU kernel, Cout <= 16 with 5x5 / 7x7 / 3x5 / 5x7 kernels the M = 4 head kernels (or the kw-folded GEMM), a 3x3 / pad 1 layer with
   * 64 output channels on a full-resolution map (conv1_2: whole 4 x 128 tiles, >= 2 per CU) the ring kernel of wconv.hip, everything
   * else the direct implicit-GEMM kernel.  The Winograd forms carry ~10x the rounding error of the direct sum: callers that go through the
   * C++ layer (libmscnn_caffe.so) get a per-layer check against DIRECT on the first forward after every weight change, and a
   * fall-back, without asking (include/mscnn_net.h: mscnn_net_set_auto_calibrate); callers of THIS ABI own that decision and can
   * make it with mscnn_max_rel_diff_f32 on a DIRECT plan's output. */
  MSCNN_CONV_ALGO_AUTO = 0,
  MSCNN_CONV_ALGO_DIRECT = 1,   /* never Winograd: the k-ordered implicit-GEMM sum (per-layer numerical fall-back) */
  MSCNN_CONV_ALGO_WINO_F2 = 2,  /* F(2x2,3x3) on whole planes wherever it is legal (small ROI maps: F(3x3,3x3)) */
  MSCNN_CONV_ALGO_WINO_F3 = 3,  /* F(3x3,3x3) wherever it is legal */
  /* Reduced-precision mode (no reference counterpart): operands rounded to fp16 on their way into LDS, v_mfma_f32_32x32x16_f16
   * with fp32 accumulators, direct 3x3 implicit GEMM (no Winograd).  Blobs stay fp32.  Shapes without an fp16 kernel (stride,
   * groups, heads) run their fp32 kernel.  Tolerance policy: DESIGN.md (per-layer 5e-3 of the layer's scale; detections IoU >=
   * 0.95 and |dscore| <= 5e-3 against the fp32 path). */
  MSCNN_CONV_ALGO_F16 = 4,
  /* F(3x3,3x3) whose 25 plane GEMMs run on the fp16 MFMA pipe with every fp32 operand split exactly into two fp16 halves
   * (x s = hi + lo, s a power of two taken from the tensor's max |x|, measured on the device each forward) and three products
   * per pair, fp32 accumulators: 22-bit significands -- fp32-grade results (same 1e-4 parity bar as the fp32 kernels) at 16/3
   * of the fp32 MFMA rate.  Opt-in; it replaces the plane GEMMs of the layers the AUTO heuristic runs as F(3x3,3x3); every other
   * layer (and Cin not a multiple of 32) keeps its fp32 kernel.  mscnn_conv2d_plan_dtype() reports "f16x3". */
  MSCNN_CONV_ALGO_WINO_F3_X3 = 5,
  /* F(4x4,3x3) with the interpolation points {0, 1, -1, 2, -1/2, inf} wherever it is legal (whole planes; ROI maps keep
   * F(3x3,3x3)): 36 multiplies per 16 outputs, fp32 error 0.8 .. 3.2x (median 1.2x) the F(3x3,3x3) form's over 36 input / filter
   * distributions (profiles/r03_robustness.txt).  In production since round 3: AUTO selects it for the nine largest 3x3 layers of
   * the 7s-576 net (A/B per layer: profiles/r03_ab_wino_f4.txt); this value forces it also where AUTO would keep F(3x3,3x3). */
  MSCNN_CONV_ALGO_WINO_F4 = 6
} mscnn_conv_algo;

typedef struct {
  int N, Cin, H, W;          /* bottom shape */
  int Cout, Kh, Kw;
  int pad_h, pad_w, stride_h, stride_w, group;
  int relu;                  /* 1: apply max(x,0) in the epilogue (negative_slope 0) */
  int algo;                  /* mscnn_conv_algo; shapes an algorithm does not cover fall back to the direct kernels */
  /* Tuning knobs for A/B measurements (tools/bench_layers.py); 0 = the measured default.  None changes results beyond
   * fp32 rounding.  tune_variant: igemm tile variant (value + 1, so that 0 keeps the default); tune_grid: workgroups of the
   * persistent igemm / head kernels; tune_flags: bit 0 = no XCD-aware workgroup map, bit 1 = proposal heads on the 32-row
   * igemm tile instead of the M = 4 head kernel, bit 2 = MSCNN_CONV_ALGO_WINO_F3_X3 wherever it is legal (default: only where
   * the AUTO heuristic picks F(3x3,3x3)), bit 4 = fp32 proposal heads as one GEMM over the taps + shift-and-add (measured equal or
   * slower than the head kernel: opt-in), bit 5 = never that form, bit 6 = AUTO keeps F(3x3,3x3) where it would take F(4x4,3x3),
   * bit 7 = the Winograd plane GEMMs on the round-2 igemm kernel instead of wgemm.hip's, bit 8 = the scalar (host-checked)
   * F(4x4,3x3) transform kernels instead of the vectorised ones and the generic F(3x3,3x3) output transform on ROI maps instead of the LDS-staged one, bit 9 = never / bit 10 = wherever legal: proposal heads with the
   * kernel's columns folded into M (KH x 1 convolution with KW * Cout channels + shift-and-add), bit 11 = a Cin = 3 layer (conv1_1) on the MFMA igemm kernel instead of its VALU kernel, bit 12 = F(4x4,3x3) input transform with one tile per lane instead of two, bit 15 = a 3x3 / pad 1 layer with 64 output channels on whole 4 x 128 tiles (conv1_2) stays on the igemm kernel instead of the ring kernel of wconv.hip (bit-identical for whole tiles; A/B, second witness), bit 16 = (round 5) a 3x3 / pad 1 layer with 64 output channels on a full-resolution map (conv1_2: whole 8 x 32 blocks, >= 512 of them) stays on the direct ring kernel of wconv.hip instead of the one-launch Winograd F(2x2,3x3) kernel of wf2conv.hip that AUTO takes there (A/B, the direct witness; tune_variant 403 also plans smaller maps on it: tests), bit 14 = inverts the wave priority of the direct MFMA kernel's tile epilogue (raised by default on the 64 x 256 3x3 tile = conv1_2 only), bit 13 = (witness build only, `make -C mscnn_amd/csrc witness`; ignored by the product library) proposal heads on the packed-FMA kernel of tools/micro/headvalu.hip instead of the M = 4 MFMA kernel (headconv.hip); tune_variant with WINO_F3_X3: 1 = 128-row, 2 = 256-row GEMM tiles;
   * tune_variant 300 + v with a Winograd algo: wgemm tile variant v (1: 256 x 128, 2: 128 x 256, 3: 128 x 128, 4: 256 x 96, 5: 256 x 160; + 256 forces the
   * stream-K split, + 512 whole tiles).  Proposal heads (headconv.hip, round 5): tune_variant 500 / 501 = full / half channel chunks
   * whatever the map (AUTO: half chunks on maps of <= 16 tiles of 16 x 32 pixels and for the 5-row kernels everywhere); tune_grid > 0 = that many workgroups, clamped to the
   * number of (tile, chunk) units. */
  int tune_variant, tune_grid, tune_flags;
} mscnn_conv_desc;

MSCNN_API int mscnn_conv2d_plan_create(const mscnn_conv_desc* desc, mscnn_conv_plan** plan_out);
MSCNN_API void mscnn_conv2d_plan_destroy(mscnn_conv_plan* plan);
/* Bytes of device memory the plan needs for packed weights and for split-K partial tiles. */
MSCNN_API size_t mscnn_conv2d_packed_weight_bytes(const mscnn_conv_plan* plan);
MSCNN_API size_t mscnn_conv2d_workspace_bytes(const mscnn_conv_plan* plan);
/* Which kernel family the plan selected (static string):
 *   "winograd_f4x4_3x3" | "winograd_f3x3_3x3" | "winograd_f2x2_3x3"        input transform -> wgemm plane GEMM -> output transform
 *   "winograd_f3x3_3x3_x3f16_128" | "..._256"                               the same with split-fp16 plane GEMMs (WINO_F3_X3)
 *   "igemm_<BM>x<BN>_k3x3_..." | "igemm_<BM>x<BN>_k1x1_..."                 direct implicit GEMM on v_mfma_f32_32x32x2_f32
 *   "igemm16_..." | "igemm16x3_..."                                          the same on fp16 / split-fp16 operands
 *   "conv3x3_c3_valu_f32"                                                    Cin = 3 (conv1_1)
 *   "winograd2x2_fused_k3x3_c64" (conv1_2 on full-resolution maps: ONE launch -- input transform, 16 plane products on v_mfma_f32_16x16x4_f32,
 *                               output transform, ReLU and the 2x2 pooling inside a workgroup; executed FLOPs = 16 / 36 of the direct count)
 *   "wconv_64x512_k3x3" (the same shape class as a direct convolution: DIRECT, or tune_flags bit 16) | "head4x4_k<Kh>x<Kw>_m<..>" | "head_kwfold_shiftadd_f32" | "head_gemm_shiftadd_f32" | "head_gemm_shiftadd_x3f16"   proposal heads
 *   "direct_f32"                                                             any stride / group / kernel size (generic kernel)
 * A name that starts with "winograd" is what the numerical checks of the C++ layer key on. */
MSCNN_API const char* mscnn_conv2d_plan_kernel(const mscnn_conv_plan* plan);
/* Algorithmic FLOPs (2*MACs of the direct convolution the reference computes) of one forward call. */
MSCNN_API double mscnn_conv2d_plan_flops(const mscnn_conv_plan* plan);
/* "f32" | "f16" | "f16x3": the arithmetic type of the plan's MFMA operands (accumulation is fp32 in all of them). */
MSCNN_API const char* mscnn_conv2d_plan_dtype(const mscnn_conv_plan* plan);
/* FLOPs the MFMA pipe really executes for the real (unpadded) problem: equal to the algorithmic count for the direct
 * kernels, 2 * planes * Cout * Cin * tiles for the Winograd forms (25/81 resp. 16/36 of it on exactly tiled planes). */
MSCNN_API double mscnn_conv2d_plan_executed_flops(const mscnn_conv_plan* plan);
/* Roofline accounting: with profiling on, every forward brackets the plan's stages with HIP events on the call's stream;
 * mscnn_conv2d_plan_stage_ms waits for the last forward and returns {input transform, MFMA GEMM kernel(s) incl. the
 * stream-K fix-up, output transform} in milliseconds (direct / head kernels: {0, total, 0}). */
MSCNN_API int mscnn_conv2d_plan_set_profiling(mscnn_conv_plan* plan, int on);
MSCNN_API int mscnn_conv2d_plan_stage_ms(const mscnn_conv_plan* plan, float ms_out[3]);
/* max |x| hand-over between the layers of a chain (split-fp16 plans, MSCNN_CONV_ALGO_WINO_F3_X3).  Such a plan needs an upper
 * bound of max |x| of its input; by default it measures it itself (one streaming pass over x per forward).
 * Both are DEVICE arrays of MSCNN_AMAX_SLOTS uint32 holding float bit patterns; the value they stand for is their maximum
 * (thousands of workgroups publishing into one address would serialise, so each takes one of the slots):
 *   in_bound  (max over the slots >= max |x|; NULL = measure): read by this plan's forward instead;
 *   out_amax  (NULL = off): the plan's forward does atomicMax(bits of a partial max |y|) into the slots -- the caller zeroes
 *             them before the forward.  Only the F(3x3,3x3) forms publish (mscnn_conv2d_plan_publishes_amax() == 1); for other plans a
 *             non-NULL out_amax is MSCNN_ERR_UNSUPPORTED.
 * A max-pooled or ROI-pooled copy of y is bounded by the same value. */
#define MSCNN_AMAX_SLOTS 1024
MSCNN_API int mscnn_conv2d_plan_publishes_amax(const mscnn_conv_plan* plan);
MSCNN_API int mscnn_conv2d_plan_set_amax_io(mscnn_conv_plan* plan, const uint32_t* in_bound, uint32_t* out_amax);
/* Re-shape a plan for a new batch size N (ROI count changes per image, layer.hpp:451-456). */
MSCNN_API int mscnn_conv2d_plan_set_batch(mscnn_conv_plan* plan, int N);
/* Identifies the packed-weight layout the plan currently expects (0: none, the kernel reads the Caffe layout).
 * mscnn_conv2d_plan_set_batch may select a different kernel family for the new batch (e.g. ROI count crossing the
 * Winograd threshold): when this value changes, call mscnn_conv2d_pack_weights again before the next forward. */
MSCNN_API unsigned long long mscnn_conv2d_plan_weight_layout(const mscnn_conv_plan* plan);
MSCNN_API int mscnn_conv2d_pack_weights(const mscnn_conv_plan* plan, const float* w, float* packed, void* stream);
MSCNN_API int mscnn_conv2d_fwd_f32(const mscnn_conv_plan* plan, const float* x, const float* w, const float* packed,
                         const float* bias, float* y, void* workspace, size_t workspace_bytes, void* stream);
/* Convolution (+ the plan's ReLU) with the following PoolingLayer (MAX, kernel 2, stride 2, pad 0, ceil mode:
 * pooling_layer.cu:11-47, pooling_layer.cpp:90-107) fused into the epilogue: y as above AND
 * y_pool[N][Cout][ceil(Ho/2)][ceil(Wo/2)].  Only for plans where mscnn_conv2d_plan_can_pool() is 1 (the trunk
 * kernels); bit-identical to mscnn_conv2d_fwd_f32 followed by mscnn_pool2d_fwd_f32.  y_pool == NULL: plain forward. */
MSCNN_API int mscnn_conv2d_plan_can_pool(const mscnn_conv_plan* plan);
MSCNN_API int mscnn_conv2d_fwd_pool_f32(const mscnn_conv_plan* plan, const float* x, const float* w, const float* packed,
                              const float* bias, float* y, float* y_pool, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ReLU -- ReLULayer::Forward_gpu (relu_layer.cu:9-26); in place allowed (y == x). */
MSCNN_API int mscnn_relu_fwd_f32(const float* x, float* y, size_t count, float negative_slope, void* stream);

/* Pooling -- PoolingLayer::Forward_gpu (pooling_layer.cu:11-47 MAX, :50-81 AVE; output size
 * pooling_layer.cpp:90-107, ceil mode).  method: 0 MAX, 1 AVE.  The argmax mask the reference
 * also writes is not produced (unused at TEST). */
MSCNN_API int mscnn_pool_out_dim(int in, int kernel, int pad, int stride);
MSCNN_API int mscnn_pool2d_fwd_f32(const float* x, float* y, int N, int C, int H, int W, int kernel_h, int kernel_w,
                         int pad_h, int pad_w, int stride_h, int stride_w, int method, void* stream);

/* InnerProduct -- InnerProductLayer::Forward_gpu (inner_product_layer.cu:10-31):
 * y[M,N] = x[M,K] * w[N,K]^T + bias[N]  (transpose_ = false), optional fused ReLU. */
MSCNN_API int mscnn_inner_product_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                                int M, int N, int K, int relu, void* stream);
/* dev / test knob of the small-N kernel behind it (cls_pred / bbox_pred): rows of x per workgroup, 2, 4 or 8 (0 = the default: 4 for
 * N <= 5, else 2 -- fewer rows measured faster, profiles/r05_ab_ip_rows.txt).  The per-row arithmetic is the same in every form. */
MSCNN_API void mscnn_debug_inner_product_rows(int rows);

/* The same InnerProduct (fp32, exact MFMA fmaf chains) on the plane-GEMM kernel of the Winograd layers (wgemm.hip: LDS-DMA operand ring,
 * one barrier per K chunk) -- fc6-class shapes: M >= 192 rows, N % 128 == 0, K % 32 == 0 (mscnn_inner_product_wg_supported).  wt = the
 * weights transposed once to [K][N] (mscnn_inner_product_wg_pack, N * K * 4 bytes); x is re-packed into the kernel's A layout every
 * forward inside the workspace; bias and ReLU are applied in the kernel's epilogue.  Results differ from mscnn_inner_product_fwd_f32
 * only in where the K sum is cut (both are k-ordered chains summed in a fixed order: deterministic, within the 1e-4 bar). */
MSCNN_API int mscnn_inner_product_wg_supported(int M, int N, int K);
MSCNN_API size_t mscnn_inner_product_wg_packed_bytes(int N, int K);
MSCNN_API size_t mscnn_inner_product_wg_workspace_bytes(int M, int N, int K);
MSCNN_API int mscnn_inner_product_wg_pack(const float* w, float* wt, int N, int K, void* stream);
MSCNN_API int mscnn_inner_product_wg_fwd(const float* x, const float* wt, const float* bias, float* y, int M, int N, int K, int relu,
                                         void* workspace, size_t workspace_bytes, void* stream);
/* Health of the plane-GEMM kernel's stream-K hand-off (the Winograd layers and the _wg InnerProduct above).  Where a launch splits
 * its last tiles over workgroups, the workgroup that finishes a tile waits for the others' partial sums; all workgroups of the
 * persistent grid must be co-resident for that (one per CU).  If a contributor never shows up (CU masking, a partition mode the plan
 * was not made for, another stream's kernels holding a CU for > ~0.5 s) the finisher gives up: the tile is stored as NaN (every
 * ReLU behind it keeps a NaN a NaN, as relu_layer.cpp:14-15's std::max does) AND the launch's tag is written to one word of pinned
 * host memory per device.
 *   mscnn_wgemm_handoff_event(): that word for the current device -- 0 = no hand-off has ever timed out; any change since the last
 *     look = at least one launch in between produced a poisoned tile.  Valid once the stream has been synchronised (a plain load).
 *   mscnn_wgemm_force_whole_tiles(1): from now on no launch of this process splits a tile (the last round of tiles is simply not
 *     full: a few % slower on the layers that used the split, never waits for anybody).  The answer to a reported event: force,
 *     then run the frame again -- caffe::Net does exactly that at its synchronisation points (mscnn_net_handoff_state).
 *   mscnn_debug_wgemm_handoff_fault(drop_publish, spin_limit): fault injection for the tests of the above -- contributors never
 *     publish (drop_publish != 0), finishers give up after spin_limit polls (0 = the default 2^22).  Never set in production. */
MSCNN_API unsigned long long mscnn_wgemm_handoff_event(void);
MSCNN_API void mscnn_wgemm_force_whole_tiles(int on);
MSCNN_API int mscnn_wgemm_whole_tiles_forced(void);
MSCNN_API void mscnn_debug_wgemm_handoff_fault(int drop_publish, unsigned spin_limit);
/* fp16-operand InnerProduct (the counterpart of MSCNN_CONV_ALGO_F16; no reference counterpart): w16 = the weights converted
 * once to fp16 [N][K] (mscnn_inner_product_pack_f16, N * K * 2 bytes), x rounded to fp16 on its way into LDS, fp32 accumulate.
 * Needs N >= 64 and K % 8 == 0 (mscnn_inner_product_f16_supported); smaller layers stay on the fp32 entry point. */
/* Split-fp16 InnerProduct (fp32-grade: see MSCNN_CONV_ALGO_WINO_F3_X3): w packed once into exactly split fp16 hi + lo units
 * (mscnn_inner_product_x3_pack, mscnn_inner_product_x3_packed_bytes), x split on the device every forward with the scale taken
 * from max |x| -- handed over in in_bound (MSCNN_AMAX_SLOTS uint32, see mscnn_conv2d_plan_set_amax_io) or, when NULL, measured.
 * Three fp16 MFMAs per operand pair, fp32 accumulate, K cut into ranges whose partial sums are added in a fixed order.
 * Needs N >= 128 and K % 32 == 0. */
MSCNN_API int mscnn_inner_product_x3_supported(int N, int K);
MSCNN_API size_t mscnn_inner_product_x3_packed_bytes(int N, int K);
MSCNN_API size_t mscnn_inner_product_x3_workspace_bytes(int M, int N, int K);
MSCNN_API int mscnn_inner_product_x3_pack(const float* w, void* packed, int N, int K, void* stream);
MSCNN_API int mscnn_inner_product_x3_fwd(const float* x, const void* packed, const float* bias, float* y, int M, int N, int K, int relu,
                               const uint32_t* in_bound, void* workspace, size_t workspace_bytes, void* stream);
MSCNN_API int mscnn_inner_product_f16_supported(int N, int K);
MSCNN_API int mscnn_inner_product_pack_f16(const float* w, void* w16, int N, int K, void* stream);
MSCNN_API int mscnn_inner_product_fwd_f16(const float* x, const void* w16, const float* bias, float* y, int M, int N, int K, int relu,
                                          void* stream);

/* Concat along channels -- ConcatLayer::Forward_gpu (concat_layer.cu:9-46).
 * Copies x[N, C, inner] into y[N, C_total, inner] at channel offset c_offset. */
MSCNN_API int mscnn_concat_channels_f32(const float* x, float* y, int N, int C, int inner, int C_total, int c_offset, void* stream);

/* Deconvolution -- DeconvolutionLayer::Forward_gpu (deconv_layer.cu:8-23), depthwise case used by the
 * "-2x" deploy nets (group == Cin == Cout, w[C][1][Kh][Kw], no bias or bias[C]). */
MSCNN_API int mscnn_deconv_depthwise_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                                   int N, int C, int H, int W, int Kh, int Kw, int pad_h, int pad_w,
                                   int stride_h, int stride_w, void* stream);

/* Deconvolution, general case (any group / stride / pad; w[Cin][Cout/group][Kh][Kw], base_conv_layer.cpp:135-140 with
 * reverse_dimensions()); routes the depthwise case to the kernel above. */
MSCNN_API int mscnn_deconv2d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                                     int Cout, int Kh, int Kw, int pad_h, int pad_w, int stride_h, int stride_w, int group,
                                     void* stream);

/* out_dev[0] = max_i |a[i] - ref[i]| / max(floor, |ref[i]|) (+inf if any NaN): the parity metric of the test-suite on the
 * device; used by the host runtime's per-layer numerical calibration (Winograd against the direct sum). */
MSCNN_API int mscnn_max_rel_diff_f32(const float* a, const float* ref, size_t count, float floor, float* out_dev, void* stream);
/* The same metric over `planes` runs of `run` consecutive floats -- plane p of a at a + p * a_stride, of ref at ref + p * ref_stride (a
 * band of rows of an NCHW blob against a band computed elsewhere) -- with floor = max(1, sqrt(sumsq_dev[0] / sumsq_count)) read ON THE
 * DEVICE: the host runtime's numerics watch chains mscnn_sum_squares_f32 -> this call without a host round trip in between. */
MSCNN_API int mscnn_max_rel_diff_strided_f32(const float* a, size_t a_stride, const float* ref, size_t ref_stride, size_t planes, size_t run,
                                             const double* sumsq_dev, double sumsq_count, float* out_dev, void* stream);
/* dst_dev[0 .. n) = values_host[0 .. n), n <= 4, in ONE launch (values travel as kernel arguments): the header words of a detection pack. */
MSCNN_API int mscnn_store_words_i32(int* dst_dev, const int* values_host, int n, void* stream);
/* out_dev[0] = sum_i x[i]^2 in double: the scale (rms) of a blob, the floor of the calibration metric on hot activations. */
MSCNN_API int mscnn_sum_squares_f32(const float* x, size_t count, double* out_dev, void* stream);

/* Softmax over axis 1 of x[outer][C][inner] -- SoftmaxLayer::Forward_gpu (softmax_layer.cu:83-120). */
MSCNN_API int mscnn_softmax_fwd_f32(const float* x, float* y, int outer, int C, int inner, void* stream);

/* ------------------------------------------------------------------------------------------
 * ROIPooling -- ROIPoolingLayer<Dtype>::Forward_gpu (roi_pooling_layer.cu:19-104; CPU twin
 * roi_pooling_layer.cpp:48-139) with the MS-CNN pad_ratio context padding.  rois[R][5] =
 * [batch x1 y1 x2 y2].  Writes out[R][C_total][PH][PW] at channel offset c_offset so that two
 * calls fill the buffer the following Concat layer would produce (C_total == C, c_offset == 0
 * for the stand-alone layer).  A roi's batch index outside [0, N) is NOT checked on the device
 * (the reference CHECKs it on the CPU path only, roi_pooling_layer.cpp:64-65).
 * ------------------------------------------------------------------------------------------ */
MSCNN_API int mscnn_roipool_fwd_f32(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W,
                          int pooled_h, int pooled_w, float spatial_scale, float pad_ratio,
                          int C_total, int c_offset, void* stream);
/* The same ROIs pooled twice over the same map with two context paddings (roi_pool_org + roi_pool_ctx of the deploy nets,
 * roi_pooling_layer.cu:19-104 twice) into two disjoint channel windows [c_offset_x, c_offset_x + C) of one [R][C_total][ph][pw]
 * output, in one launch. */
MSCNN_API int mscnn_roipool_pair_fwd_f32(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W,
                                         int pooled_h, int pooled_w, float spatial_scale, float pad_ratio_a, int c_offset_a,
                                         float pad_ratio_b, int c_offset_b, int C_total, void* stream);

/* Chains of same-resolution 3x3 / pad 1 / stride 1 layers on the fp32 F(4x4,3x3) path (conv2_1 -> conv2_2, conv3_1 -> 3_2 -> 3_3,
 * conv4_1 -> 4_2 -> 4_3 of the VGG trunk; the reference runs each as its own cuDNN / im2col call, cudnn_conv_layer.cu:11-46,
 * conv_layer.cu:8-23): the output transform of `plan` writes the input-transform planes of `next` directly -- the activation between the
 * two layers is neither written (unless y != NULL) nor read; 4.5 instead of 6.5 activation-sized HBM passes per pair.  Bit-identical to
 * the two separate mscnn_conv2d_fwd_f32 calls.
 *   _can_chain: 1 when both plans take that path, plan's output is next's input (N, Cout == Cin, H, W) and the map is whole 4x4 tiles
 *     in 1 - 4, 6 or 8 strips of <= 62 tile columns (W <= 1984, W / 4 not in (248, 310] or (372, 434]);
 *   _fwd_chain: x == NULL: the planes of plan's input are already at the start of `workspace` (plan was the `next` of the previous
 *     call on that memory); next == NULL: ordinary output (y required, y_pool optional as in mscnn_conv2d_fwd_pool_f32) -- the tail of
 *     a chain; next != NULL: next's planes go to the start of next_workspace (>= next's mscnn_conv2d_workspace_bytes, disjoint from
 *     `workspace`), y may be NULL, y_pool must be;
 *   _can_pool_only: 1 when a forward with the fused pooling may leave y itself unwritten (y == NULL, y_pool != NULL, here and in
 *     mscnn_conv2d_fwd_pool_f32): the F(4x4,3x3) path and the direct MFMA kernels with the pooling epilogue -- conv1_2, conv2_2,
 *     conv3_3 of the trunk, whose only reader is the pooling layer (283 + 142 + 71 MB per 7s-576 frame not written). */
MSCNN_API int mscnn_conv2d_plan_can_chain(const mscnn_conv_plan* plan, const mscnn_conv_plan* next);
MSCNN_API int mscnn_conv2d_plan_can_pool_only(const mscnn_conv_plan* plan);
MSCNN_API int mscnn_conv2d_fwd_chain_f32(const mscnn_conv_plan* plan, const mscnn_conv_plan* next, const float* x, const float* packed,
                               const float* bias, float* y, float* y_pool, void* workspace, size_t workspace_bytes,
                               void* next_workspace, size_t next_workspace_bytes, void* stream);

/* The detection sub-net's entry fused: ROIPooling x 2 (roi_pooling_layer.cu:19-104, pad_ratio_a -> channels [0, C), pad_ratio_b ->
 * channels [C, 2C) of the concatenated blob, concat_layer.cu:28-46) + the 3x3 convolution that consumes it (roi_c1; conv_layer.cu:8-23)
 * in its Winograd F(3x3,3x3) form: the pooled values go straight into the transform planes of the plane GEMM; the R x 2C x 7 x 7 blob
 * between the layers (140 MB per 7s-576 frame) is neither written nor read.  `plan` = the convolution's plan (N = R ROIs, Cin = 2C,
 * H x W = pooled_h x pooled_w); y = its output, bit-identical to mscnn_roipool_pair_fwd_f32 followed by mscnn_conv2d_fwd_f32.
 *   _can_fuse_roipool: 1 when the plan's kernel family and shape take this path (fp32 F(3x3,3x3), 7 x 7 bins, pad 0, C % 64 == 0);
 *   _roipool_workspace_bytes: workspace of the fused call when it builds its maps itself (the plan's own + mscnn_roipool_maps_bytes);
 *   the pooling reads four "maps" of the feature blob -- a channel-last copy and its sliding maxima over 2 x 2, 4 x 4, 8 x 8 squares
 *   (mscnn_roipool_maps_bytes, mscnn_roipool_maps_build_f32).  They depend on the feature map only: a caller may build them early, on
 *   another stream, while the proposals are still being selected, and hand them in as prepared_maps (then feat may be NULL and the
 *   workspace is the plan's own mscnn_conv2d_workspace_bytes); prepared_maps = NULL builds them inside the call. */
MSCNN_API int mscnn_conv2d_plan_can_fuse_roipool(const mscnn_conv_plan* plan, int C, int pooled_h, int pooled_w);
MSCNN_API size_t mscnn_conv2d_roipool_workspace_bytes(const mscnn_conv_plan* plan, int N, int C, int H, int W);
MSCNN_API size_t mscnn_roipool_maps_bytes(int N, int C, int H, int W);
MSCNN_API int mscnn_roipool_maps_build_f32(const float* feat, float* maps, int N, int C, int H, int W, void* stream);
MSCNN_API int mscnn_conv2d_fwd_roipool_pair_f32(const mscnn_conv_plan* plan, const float* feat, const float* prepared_maps, int N, int C,
                                                int H, int W, const float* rois, float spatial_scale, float pad_ratio_a,
                                                float pad_ratio_b, const float* packed_w, const float* bias, float* y, void* workspace,
                                                size_t workspace_bytes, void* stream);

/* ROIAlign -- ROIAlignLayer<Dtype>::Forward_gpu (roi_align_layer.cu:21-112): out[R][C][pooled_h+1][pooled_w+1] bilinear
 * samples on the grid of the (context-padded) roi; the WiderFace cascade deploy follows it with a 2x2 stride-1 AVE Pooling. */
MSCNN_API int mscnn_roialign_fwd_f32(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W,
                                     int pooled_h, int pooled_w, float spatial_scale, float pad_ratio, void* stream);

/* Eltwise -- EltwiseLayer<Dtype>::Forward_gpu (eltwise_layer.cu): op 0 PROD, 1 SUM (coeffs_host[num_bottoms], NULL = all 1),
 * 2 MAX.  bottoms_host: host array of num_bottoms (2..8) device pointers of `count` floats each. */
MSCNN_API int mscnn_eltwise_fwd_f32(const float* const* bottoms_host, int num_bottoms, const float* coeffs_host, float* y,
                                    size_t count, int op, void* stream);

/* ------------------------------------------------------------------------------------------
 * BoxOutput -- BoxOutputLayer<Dtype>::Forward_cpu (box_output_layer.cpp:66-234; the reference has
 * no GPU version, box_output_layer.hpp:39-40): per-anchor decode + filter, descending sort on
 * (score, candidate index), top max_nms_num, greedy NMS (nmsMax :38-63, BoxIOU
 * math_functions.cpp:12-35), emit rois [b x1 y1 x2 y2] and proposals+score [b x1 y1 x2 y2 s].
 * ------------------------------------------------------------------------------------------ */
#define MSCNN_BOXOUT_MAX_HEADS 16
typedef struct {
  int num_heads;                              /* number of bottoms */
  int num;                                    /* batch size (images) */
  int channels;                               /* cls_num + 4 */
  int head_h[MSCNN_BOXOUT_MAX_HEADS], head_w[MSCNN_BOXOUT_MAX_HEADS];
  float field_w[MSCNN_BOXOUT_MAX_HEADS], field_h[MSCNN_BOXOUT_MAX_HEADS], downsample_rate[MSCNN_BOXOUT_MAX_HEADS];
  float fg_thr, iou_thr;
  int nms_mode;                               /* 0 "IOU", 1 "IOMU", 2 "IOFU" */
  float field_whr, field_xyr;
  int max_nms_num, max_post_nms_num;
  float min_size;
  int do_bbox_norm;                           /* bbox_reg_param present with 4 means and 4 stds */
  float bbox_mean[4], bbox_std[4];
} mscnn_boxoutput_desc;

MSCNN_API size_t mscnn_boxoutput_workspace_bytes(const mscnn_boxoutput_desc* desc);
/* Row capacity needed for rois_out / props_out / anchor_ids_out. */
MSCNN_API int mscnn_boxoutput_max_rows(const mscnn_boxoutput_desc* desc);
/*
 * heads_host: host array of num_heads DEVICE pointers.  rois_out[cap][5], props_out[cap][6]
 * (NULL allowed), anchor_ids_out[cap] (NULL allowed; global anchor id = head offset + h*w index of
 * each emitted row, -1 for the dummy row), count_out_dev: device int[2] = {R, num_real}
 * (R includes the dummy row [0 1 1 10 10] emitted when nothing survives, :195-199).
 * Asynchronous on `stream`; the caller copies count_out_dev back when it needs R on the host.
 */
MSCNN_API int mscnn_boxoutput_fwd_f32(const mscnn_boxoutput_desc* desc, const float* const* heads_host,
                            float* rois_out, float* props_out, int* anchor_ids_out, int cap,
                            int* count_out_dev, void* workspace, size_t workspace_bytes, void* stream);

/* Greedy NMS on already score-sorted boxes [n][4] = x y w h (nmsMax, greedy = true):
 * keep_out[n] bytes 0/1.  Exposed for the index-exactness parity tests. */
MSCNN_API size_t mscnn_nms_workspace_bytes(int n);
MSCNN_API int mscnn_nms_greedy_f32(const float* boxes_xywh, int n, float iou_thr, int nms_mode, unsigned char* keep_out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* DecodeBBox (TEST phase) -- DecodeBBoxLayer<Dtype>::Forward_cpu (decode_bbox_layer.cpp:54-123,
 * DecodeBBoxesWithPrior math_functions.cpp:46-75); bbox[R][8] (columns 4..7 used), prior[R][5],
 * mean/std host arrays of 4. */
MSCNN_API int mscnn_decodebbox_fwd_f32(const float* bbox, const float* prior, float* out, int R, int bbox_dim,
                             const float* mean_host, const float* std_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Final detection stage -- the MATLAB post-processing the reference runs on the net outputs
 * (examples/kitti_car/run_mscnn_detection.m:75-120, utils/bbNms.m:112-126): proposal filter,
 * bbox de-normalisation + transform, softmax prob of class cls_id (1-based), rescale to the
 * original image, clip, greedy NMS (stable sort, union, double precision).
 * dets_out[cap][5] doubles [x y w h prob], ids_out[cap] input row of each detection,
 * count_out_dev: device int[1] = D.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int ncls, cls_id;
  float bbox_mean[4], bbox_std[4];
  float proposal_thr;
  double ratio_h, ratio_w, org_h, org_w, nms_overlap;
} mscnn_detections_desc;
MSCNN_API size_t mscnn_detections_workspace_bytes(int R);
MSCNN_API int mscnn_detections_fwd(const mscnn_detections_desc* desc, const float* bbox_pred, const float* cls_pred,
                         const float* props, int R, double* dets_out, int* ids_out, int* count_out_dev,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Final stage of the cascade drivers (examples/kitti_car/run_cascademscnn.m:84-117): `boxes` = the decoded box blob of a
 * cascade stage [R][5] = [img x1 y1 x2 y2] (DecodeBBox output), `cls_prob` = that stage's in-net probabilities [R][ncls]
 * (Softmax / Eltwise blob), `props` = its proposal blob [R][5].  Rescale, clip, w = x2 - x1 + 1, drop rows whose proposal
 * has zero width or height, optional det_thr (> 0), then the same NMS.  desc: ncls, cls_id, ratio_*, org_*, nms_overlap. */
MSCNN_API int mscnn_detections_cascade_fwd(const mscnn_detections_desc* desc, float det_thr, const float* boxes,
                                 const float* cls_prob, const float* props, int R, double* dets_out, int* ids_out,
                                 int* count_out_dev, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Image pre-processing in front of net.forward -- MATLAB `run_mscnn_detection.m:64-69`:
 * imresize(uint8 image, [H W]) (bicubic, uint8 after each 1-D pass), RGB -> BGR, single, subtract the
 * per-channel mean, hand to the net as its (1,3,H,W) input blob.
 * img_rgb: device uint8 [org_h][org_w][3] (row-major HWC, as any image decoder delivers it);
 * out: device float [3][H][W] (planes B, G, R); mean_bgr: host float[3] ({104,117,123} in the reference).
 * ------------------------------------------------------------------------------------------ */
MSCNN_API size_t mscnn_preprocess_workspace_bytes(int org_h, int org_w, int H, int W);
MSCNN_API int mscnn_preprocess_u8_f32(const unsigned char* img_rgb, int org_h, int org_w, float* out, int H, int W,
                            const float* mean_bgr, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* MSCNN_HIP_H_ */
