// Layer classes of the MS-CNN deploy nets, same class names / prototxt type strings as the reference
// (include/caffe/layers/*.hpp).  Every Forward_gpu is a call into the C ABI of libmscnn_hip.so
// (include/mscnn_hip.h); there is no Forward_cpu implementation.
#ifndef MSCNN_CAFFE_LAYERS_HPP_
#define MSCNN_CAFFE_LAYERS_HPP_

#include <functional>
#include <string>
#include <vector>

#include "caffe/blob.hpp"
#include "caffe/layer.hpp"

struct mscnn_conv_plan;

namespace caffe {

// Device scratch owned by a layer (hipMalloc/hipFree), grown on demand.
class DeviceBuffer {
 public:
  DeviceBuffer() : ptr_(nullptr), bytes_(0) {}
  ~DeviceBuffer();
  void* Reserve(size_t bytes);
  void Release();                          // hipFree now (synchronises the device); the next Reserve allocates again
  void* get() const { return ptr_; }
  size_t bytes() const { return bytes_; }
 private:
  void* ptr_;
  size_t bytes_;
  DISABLE_COPY_AND_ASSIGN(DeviceBuffer);
};

// include/caffe/layers/input_layer.hpp
template <typename Dtype>
class InputLayer : public Layer<Dtype> {
 public:
  explicit InputLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual inline const char* type() const { return "Input"; }
  virtual inline int ExactNumBottomBlobs() const { return 0; }
  virtual inline int MinTopBlobs() const { return 1; }
 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
};

// include/caffe/layers/split_layer.hpp -- zero copy fan-out (split_layer.cpp:26-31)
template <typename Dtype>
class SplitLayer : public Layer<Dtype> {
 public:
  explicit SplitLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Split"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int MinTopBlobs() const { return 1; }
 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { Forward_cpu(bottom, top); }
};

template <typename Dtype> class ROIPoolingLayer;

// include/caffe/layers/base_conv_layer.hpp + conv_layer.hpp (2-D, dilation 1)
template <typename Dtype>
class ConvolutionLayer : public Layer<Dtype> {
 public:
  explicit ConvolutionLayer(const LayerParameter& param)
      : Layer<Dtype>(param), plan_(nullptr), relu_(false), weights_dirty_(true), planned_n_(-1), algo_(0), profiling_(false) {
    tune_[0] = tune_[1] = tune_[2] = 0;
  }
  virtual ~ConvolutionLayer();
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Convolution"; }
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int MinTopBlobs() const { return 1; }
  virtual void OnWeightsChanged() { weights_dirty_ = true; selfcheck_pending_ = true; wino_checked_ = false; }
  virtual bool FuseReLU(Dtype negative_slope);
  virtual bool FusePool2x2(Blob<Dtype>* pooled_top);
  // Net-level fusion (round 4): this layer's bottom is the concatenation of a deferred ROIPooling pair (`first` does both windows,
  // channels [0, C) and [C, 2C) in the order of their Concat offsets).  While the pair is pending and the planned kernel can pool in
  // its input stage (mscnn_conv2d_plan_can_fuse_roipool) Forward reads the feature map itself; otherwise it asks for the blob.
  void FuseRoiPoolInput(ROIPoolingLayer<Dtype>* first) { roi_src_ = first; }
  // (round 6) the fused ROI pooling's maps (a channel-last copy of the feature blob + its sliding maxima: they depend on the feature
  // blob only) built EARLY by the Net -- under BoxOutput's host round trip (BoxOutputLayer::set_before_sync) -- when the last Forward
  // pooled in the input stage and nothing argues against doing so again; the next Forward uses them if its feature blob is this one,
  // InvalidateRoiMaps (every Net::ForwardFromTo starts with it) drops them.
  void PrebuildRoiMaps(const Blob<Dtype>* feat);
  void InvalidateRoiMaps() { roi_maps_feat_ = nullptr; }
  // Net-level fusion (round 4): `next` is the ONLY reader of this layer's top and a same-resolution 3x3 / pad 1 convolution.  While
  // the Net marks the pair live (both run in the same ForwardFromTo call) and both planned kernels are the fp32 F(4x4,3x3) form
  // (mscnn_conv2d_plan_can_chain), this layer's output stage writes next's input-transform planes and NOT its top blob
  // (top_stale() until someone asks: Net::MaterializeBlob re-runs the layer unchained -- bit-identical kernels).
  void ChainTo(ConvolutionLayer* next) { chain_next_ = next; }
  ConvolutionLayer* chain_next() const { return chain_next_; }
  void set_chain_live(bool on) { chain_live_ = on; if (!on) prepared_ = false; }
  // The top's only reader is the fused-away 2x2 pooling (conv2_2, conv3_3): while live, an F(4x4,3x3) Forward writes the pooled blob
  // only (mscnn_conv2d_plan_can_pool_only); the top is top_stale() and re-created on demand like a chain's
  void set_pool_only_live(bool on) { pool_only_live_ = on; }
  // (round 6) this Forward writes the top blob even as a chain producer (the output stage then writes the next layer's planes AND y) or
  // where the pooling is its only reader: the numerics watch needs the blobs of the layer it looks at, without leaving the chain
  void set_keep_top(bool on) { keep_top_ = on; }
  bool top_stale() const { return top_stale_; }
  // Forward with the chain (and a prepared input) ignored: writes the top blob from the bottom blob
  void ForwardUnchained(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    const bool live = chain_live_, po = pool_only_live_;
    chain_live_ = false; prepared_ = false; pool_only_live_ = false;
    this->Forward(bottom, top);
    chain_live_ = live; pool_only_live_ = po;
  }
  virtual double ForwardFlops() const;
  const char* kernel_name() const;
  const char* dtype() const;              // "f32" | "f16": MFMA operand type of the planned kernel
  // --- extensions of this build (no reference counterpart) ---
  // Algorithm of this layer (mscnn_conv_algo in include/mscnn_hip.h: 0 auto, 1 direct, 2 / 3 Winograd F(2x2) / F(3x3));
  // Net::CalibrateNumerics sets 1 on layers whose Winograd result strays from the direct sum on representative data.
  void set_algo(int algo);
  int algo() const { return algo_; }
  // Net::CalibrateNumerics found this layer's Winograd form off the direct sum on the user's data: it stays on the direct fp32
  // kernel across precision switches (mscnn_net_set_precision) until an explicit mscnn_net_set_conv_algo clears the mark.
  void set_calibrated_direct(bool on) { calibrated_direct_ = on; }
  bool calibrated_direct() const { return calibrated_direct_; }
  // Safe by default: the FIRST Forward after construction, after a weight change (OnWeightsChanged) or after a switch to another
  // non-direct algorithm checks a Winograd result against the direct kernel on the very bottom it was given (ErrorAgainstDirect) and,
  // when it is off by more than the tolerance, puts the layer on the direct kernel for good and recomputes the tops before Forward
  // returns -- no caller ever sees an unchecked Winograd result.  kDefaultSelfcheckTol unless set_selfcheck says otherwise; 0 = off.
  static constexpr double kDefaultSelfcheckTol = 5e-5;
  void set_selfcheck(double tol) { selfcheck_tol_ = tol; selfcheck_pending_ = true; wino_checked_ = false; }
  double selfcheck_tol() const { return selfcheck_tol_; }
  // what the last self-check measured; `take` clears the "a check ran since the last take" mark (Net bookkeeping)
  bool take_selfcheck(double* err, bool* fell_back) {
    if (!selfcheck_ran_) return false;
    selfcheck_ran_ = false; *err = selfcheck_err_; *fell_back = selfcheck_fell_back_;
    selfcheck_fell_back_ = false;
    return true;
  }
  void set_tuning(int variant, int grid, int flags);      // A/B measurement knobs (mscnn_conv_desc::tune_*)
  // FLOPs the MFMA pipe executes (Winograd forms: fewer than ForwardFlops) and per-stage HIP-event times of the last
  // Forward {input transform, MFMA GEMM, output transform} -- roofline accounting (bench.py).
  double ExecutedFlops() const;
  void set_profiling(bool on);
  bool StageMs(float ms[3]) const;
  // max |y - y_direct| / max(1, |y_direct|) of the current algorithm against the direct kernel on the given bottom
  // (device scratch only; the layer's tops are not touched).  0 when the layer already runs a direct kernel.
  double ErrorAgainstDirect(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  // The numerics watch's form of the same comparison (round 6): ONE BAND of the layer -- a few rows of the map with a one-row halo
  // (3x3 / pad 1 trunk layers), or a few of the images / ROIs (small maps, roi_c1): ~30 us of direct-kernel work, at most an eighth of the layer -- is recomputed with
  // the direct kernel and compared with the same rows of the top blob this Forward wrote, everything ENQUEUED behind the frame on the
  // layer's stream: no host synchronisation, no second full-size convolution.  The verdict is read later (PollBandCheck), so a frame's
  // latency never contains a check; it applies from the frame after it is known.
  //   BeginBandCheck: false when there is nothing to check (a direct kernel, an empty or unwritten top, a check still in flight).
  //   PollBandCheck: 0 nothing in flight, 1 still running (wait = false), 2 done: *err = max |dy| / max(1, |y|, rms(y)) over the band.
  bool BeginBandCheck(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top, int band_seq);
  int PollBandCheck(double* err, bool wait);
  // the last ErrorAgainstDirect compared two results of an all-zero bottom (a zero warm-up frame): it says nothing about the layer's
  // numerics on data -- the first-forward check stays armed for the next bottom
  bool last_check_vacuous() const { return last_check_vacuous_; }
  // The checks' device scratch (the direct plan's packed weights + workspace + a copy of the largest checked top: up to ~0.6 GB for a
  // 7s-576 net, kept per host thread and device so that a check on a live stream pays no hipMalloc / synchronising hipFree) is
  // released by this call; Net::SetNumericsWatch(0, .) / SetAutoCalibrate(0) call it when both checks are switched off.
  static void ReleaseCheckScratch();
  // max |x| hand-over for the split-fp16 algorithm (mscnn_conv2d_plan_set_amax_io), wired by the Net: `out` is this layer's
  // slot (written when some consumer asked for it: set_amax_wanted), `src` / `in` the layer whose output bounds this layer's
  // bottom and its slot.  A hand-over is used only for forwards the Net marks trusted (the producer ran in the same call).
  void set_amax_io(const ConvolutionLayer* src, const unsigned* in, unsigned* out) { amax_src_ = src; amax_in_ = in; amax_out_ = out; }
  const ConvolutionLayer* amax_src() const { return amax_src_; }
  void set_amax_wanted(bool on) { amax_wanted_ = on; }
  void set_amax_trusted(bool on) { amax_trusted_ = on; }
  bool publishes_amax() const;
 protected:
  MSCNN_NO_CPU_PATH("Convolution")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  void Plan(int n, int h, int w);
  int num_output_, channels_, group_, kernel_h_, kernel_w_, pad_h_, pad_w_, stride_h_, stride_w_;
  bool bias_term_;
  mscnn_conv_plan* plan_;
  bool relu_, weights_dirty_;
  int planned_n_, planned_h_, planned_w_;
  Blob<Dtype>* pooled_top_ = nullptr;     // fused Pooling layer's top (FusePool2x2)
  DeviceBuffer packed_, workspace_;
  int algo_, tune_[3];
  ROIPoolingLayer<Dtype>* roi_src_ = nullptr;
  bool last_fused_roipool_ = false;       // the last Forward pooled inside its input stage (kernel_name() says so)
  DeviceBuffer roi_maps_;                 // PrebuildRoiMaps: the maps, the feature data they were built from, its shape
  const Dtype* roi_maps_feat_ = nullptr;
  int roi_maps_shape_[4] = {0, 0, 0, 0};
  ConvolutionLayer* chain_next_ = nullptr;
  bool chain_live_ = false, pool_only_live_ = false, top_stale_ = false, last_chained_ = false, keep_top_ = false;
  bool prepared_ = false;                 // the previous layer of the chain wrote this layer's planes at ws_off_ (this Forward only)
  size_t ws_off_ = 0, next_off_ = 0;      // this layer's / the next layer's region of the shared workspace while a chain runs
  bool fuse_next_now_ = false;            // decided by the head of the chain for this Forward
  bool ChainableNow(int n, int h, int w);
  bool calibrated_direct_ = false;
  double selfcheck_tol_ = kDefaultSelfcheckTol, selfcheck_err_ = 0.0;
  bool selfcheck_pending_ = true, selfcheck_ran_ = false, selfcheck_fell_back_ = false, last_check_vacuous_ = false;
  bool wino_checked_ = false;             // a Winograd result of the current weights / algorithm has been compared with the direct kernel
  struct BandTicket {                     // a band check in flight: the direct plan it runs on, its completion event, the pinned verdict
    mscnn_conv_plan* plan = nullptr;
    void* done = nullptr;                 // hipEvent_t
    float* host = nullptr;                // this layer's word of the (thread, device)'s pinned verdict array: the metric
    bool inflight = false;
  } band_;
  bool profiling_;
  const ConvolutionLayer* amax_src_ = nullptr;
  const unsigned* amax_in_ = nullptr;
  unsigned* amax_out_ = nullptr;
  bool amax_wanted_ = false, amax_trusted_ = false;
};

// include/caffe/layers/deconv_layer.hpp -- transposed conv; the depthwise case of the "-2x" nets has its own kernel
template <typename Dtype>
class DeconvolutionLayer : public Layer<Dtype> {
 public:
  explicit DeconvolutionLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Deconvolution"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("Deconvolution")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int num_output_, channels_, group_, kernel_h_, kernel_w_, pad_h_, pad_w_, stride_h_, stride_w_;
  bool bias_term_;
};

// include/caffe/layers/pooling_layer.hpp
template <typename Dtype>
class PoolingLayer : public Layer<Dtype> {
 public:
  explicit PoolingLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Pooling"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }   // the optional argmax-mask top is not produced
  virtual bool IsMaxPool2x2() const {
    return method_ == 0 && !global_pooling_ && kernel_h_ == 2 && kernel_w_ == 2 && stride_h_ == 2 && stride_w_ == 2 && pad_h_ == 0 && pad_w_ == 0;
  }
 protected:
  MSCNN_NO_CPU_PATH("Pooling")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int kernel_h_, kernel_w_, stride_h_, stride_w_, pad_h_, pad_w_;
  int channels_, height_, width_, pooled_height_, pooled_width_;
  bool global_pooling_;
  int method_;
};

// include/caffe/layers/relu_layer.hpp
template <typename Dtype>
class ReLULayer : public Layer<Dtype> {
 public:
  explicit ReLULayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { top[0]->ReshapeLike(*bottom[0]); }
  virtual inline const char* type() const { return "ReLU"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("ReLU")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
};

// include/caffe/layers/inner_product_layer.hpp
template <typename Dtype>
class InnerProductLayer : public Layer<Dtype> {
 public:
  explicit InnerProductLayer(const LayerParameter& param) : Layer<Dtype>(param), relu_(false), f16_(false), w16_dirty_(true) {}
  // fp16-operand mode (no reference counterpart): weights kept as an fp16 copy, fp32 accumulate; layers the fp16 kernel does
  // not cover (N < 64) keep running fp32.  dtype() says what the last Forward really used.
  void set_f16(bool on) { f16_ = on; w16_dirty_ = true; }
  // split-fp16 mode ("f16x3": fp32-grade, mscnn_inner_product_x3_*); max |x| comes from the convolution that produced the
  // bottom when the Net wired one (set_amax_in) and ran it in the same forward (set_amax_trusted), else it is measured
  void set_x3(bool on) { x3_ = on; w16_dirty_ = true; }
  bool x3() const { return x3_; }
  void set_amax_in(const ConvolutionLayer<Dtype>* src, const unsigned* in) { amax_src_ = src; amax_in_ = in; }
  void set_amax_trusted(bool on) { amax_trusted_ = on; }
  const char* dtype() const { return used_x3_ ? "f16x3" : used_f16_ ? "f16" : "f32"; }
  // fp32 kernel choice: 0 = auto (fc6-class shapes -- M >= 192 rows, N % 128 == 0 -- on the plane-GEMM kernel of wgemm.hip with the
  // weights kept transposed, everything else on gemm.hip's stream-K kernel), 1 = always gemm.hip's kernels (A/B runs, second witness)
  void set_algo(int algo) { algo_ = algo; }
  const char* kernel_name() const { return used_x3_ ? "x3_gemm" : used_f16_ ? "gemm16_tn" : used_wg_ ? "wgemm_256x128_ck32_epi" : "gemm_tn"; }
  virtual void OnWeightsChanged() { w16_dirty_ = true; wt_dirty_ = true; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "InnerProduct"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
  virtual bool FuseReLU(Dtype negative_slope) { if (negative_slope != 0) return false; relu_ = true; return true; }
  virtual double ForwardFlops() const { return 2.0 * M_ * N_ * K_; }
 protected:
  MSCNN_NO_CPU_PATH("InnerProduct")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int M_, K_, N_;
  bool bias_term_, relu_;
  bool f16_, w16_dirty_, used_f16_ = false;
  bool x3_ = false, used_x3_ = false, amax_trusted_ = false;
  const ConvolutionLayer<Dtype>* amax_src_ = nullptr;
  const unsigned* amax_in_ = nullptr;
  DeviceBuffer w16_, x3_ws_;
  int algo_ = 0;
  bool wt_dirty_ = true, used_wg_ = false;
  DeviceBuffer wt_, wg_ws_;               // weights transposed to [K][N]; packed x + partial-sum slabs
};

// include/caffe/layers/concat_layer.hpp (channel axis)
template <typename Dtype>
class ConcatLayer : public Layer<Dtype> {
 public:
  explicit ConcatLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Concat"; }
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("Concat")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int concat_axis_, num_concats_, concat_input_size_;
};

// include/caffe/layers/dropout_layer.hpp -- TEST phase: identity (dropout_layer.cpp:43-45)
template <typename Dtype>
class DropoutLayer : public Layer<Dtype> {
 public:
  explicit DropoutLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { top[0]->ReshapeLike(*bottom[0]); }
  virtual inline const char* type() const { return "Dropout"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("Dropout")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
};

// include/caffe/layers/softmax_layer.hpp
template <typename Dtype>
class SoftmaxLayer : public Layer<Dtype> {
 public:
  explicit SoftmaxLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Softmax"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("Softmax")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int outer_num_, inner_num_, softmax_axis_;
};

// include/caffe/layers/roi_pooling_layer.hpp
template <typename Dtype>
class ROIPoolingLayer : public Layer<Dtype> {
 public:
  explicit ROIPoolingLayer(const LayerParameter& param)
      : Layer<Dtype>(param), window_(nullptr), window_c_total_(0), window_c_offset_(0), partner_(nullptr), skip_(false) {}
  // Net-level fusion: two ROIPooling layers over the same map and ROIs whose tops feed one fused-away Concat (roi_pool_org +
  // roi_pool_ctx) run as ONE launch -- the first of the pair does both windows and tells the second to skip its next Forward
  // (Net clears that at the start of every ForwardFromTo, so a range that starts between the two still computes the second).
  bool PairWith(ROIPoolingLayer* second);
  void set_skip(bool s) { skip_ = s; }
  // Deferred pooling (round 4): when the pair's concatenated blob is read by ONE Convolution that can pool inside its own input
  // stage (ConvolutionLayer::FuseRoiPoolInput), the first layer's Forward launches nothing: it shapes the blob, remembers its
  // bottoms and marks the blob pending.  The consumer then computes straight from the feature map; anybody else who wants the
  // blob's bytes (Net::blob_by_name, a numerical check, a consumer whose plan cannot fuse) calls Materialize(), which runs the
  // pair kernel on the remembered bottoms.  The blob is never wrong, only late.
  void set_deferred(bool on) { deferred_ = on; if (!on) pending_ = false; }
  bool deferred() const { return deferred_; }
  bool pending() const { return pending_; }
  void Materialize();
  const vector<Blob<Dtype>*>& pending_bottoms() const { return pending_bottom_; }
  ROIPoolingLayer* partner() const { return partner_; }
  int channels() const { return channels_; }
  int pooled_height() const { return pooled_height_; }
  int pooled_width() const { return pooled_width_; }
  Dtype spatial_scale() const { return spatial_scale_; }
  Dtype pad_ratio() const { return pad_ratio_; }
  int window_c_offset() const { return window_c_offset_; }
  const Blob<Dtype>* window() const { return window_; }
  virtual bool SetOutputWindow(Blob<Dtype>* target, int c_total, int c_offset) {
    window_ = target; window_c_total_ = c_total; window_c_offset_ = c_offset;
    return true;
  }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "ROIPooling"; }
  virtual inline int MinBottomBlobs() const { return 2; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int MinTopBlobs() const { return 1; }
  virtual inline int MaxTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("ROIPooling")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int channels_, height_, width_, pooled_height_, pooled_width_;
  Dtype spatial_scale_, pad_ratio_;
  Blob<Dtype>* window_;
  int window_c_total_, window_c_offset_;
  ROIPoolingLayer* partner_;
  bool skip_;
  bool deferred_ = false, pending_ = false;
  vector<Blob<Dtype>*> pending_bottom_;
  void LaunchPair(const vector<Blob<Dtype>*>& bottom);
};

// include/caffe/layers/roi_align_layer.hpp -- tops are (R, C, pooled_h + 1, pooled_w + 1) grid samples
template <typename Dtype>
class ROIAlignLayer : public Layer<Dtype> {
 public:
  explicit ROIAlignLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "ROIAlign"; }
  virtual inline int MinBottomBlobs() const { return 2; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("ROIAlign")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int channels_, height_, width_, pooled_height_, pooled_width_;
  Dtype spatial_scale_, pad_ratio_;
};

// include/caffe/layers/eltwise_layer.hpp
template <typename Dtype>
class EltwiseLayer : public Layer<Dtype> {
 public:
  explicit EltwiseLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "Eltwise"; }
  virtual inline int MinBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("Eltwise")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int op_;
  vector<float> coeffs_;
};

// include/caffe/layers/box_output_layer.hpp -- GPU implementation (the reference's is CPU only)
template <typename Dtype>
class BoxOutputLayer : public Layer<Dtype> {
 public:
  explicit BoxOutputLayer(const LayerParameter& param) : Layer<Dtype>(param), forwarded_(false) {}
  virtual ~BoxOutputLayer();
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "BoxOutput"; }
  virtual inline int MinBottomBlobs() const { return 1; }
  virtual inline int MinTopBlobs() const { return 1; }
  virtual inline int MaxTopBlobs() const { return 2; }
  int last_num_rois() const { return last_rows_; }
  // (round 6) Work that does not depend on the row count, enqueued between BoxOutput's kernels and the host's wait for {R, real rows}:
  // the device runs it while the host takes R, reshapes the tops and launches what follows -- the frame's one host round trip no
  // longer idles the GPU.  The Net puts the sliding-maximum maps of the fused ROI pooling there (they depend on conv4_3 only).
  void set_before_sync(std::function<void()> f) { before_sync_ = std::move(f); }
 protected:
  MSCNN_NO_CPU_PATH("BoxOutput")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  float fg_thr_, iou_thr_;
  string nms_type_;
  bool output_proposal_with_score_;
  DeviceBuffer workspace_;
  int cap_, last_rows_;
  bool forwarded_;
  std::function<void()> before_sync_;
  void* count_ready_ = nullptr;      // hipEvent_t behind BoxOutput's kernels
  int* host_count_ = nullptr;        // host-coherent pinned landing place of {R, real rows}, and the address the device writes it through
  int* host_count_dev_ = nullptr;
};

// include/caffe/layers/decode_bbox_layer.hpp (TEST phase)
template <typename Dtype>
class DecodeBBoxLayer : public Layer<Dtype> {
 public:
  explicit DecodeBBoxLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  virtual inline const char* type() const { return "DecodeBBox"; }
  virtual inline int MinBottomBlobs() const { return 2; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }
 protected:
  MSCNN_NO_CPU_PATH("DecodeBBox")
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  float bbox_mean_[4], bbox_std_[4];
};

}  // namespace caffe
#endif
