// Layer implementations: parameter parsing + shape logic restated from the reference's layer sources,
// Forward_gpu = one call into libmscnn_hip.so (include/mscnn_hip.h).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>
#include <random>

#include "../../../include/mscnn_hip.h"
#include "caffe/layer_factory.hpp"
#include "caffe/layers/mscnn_layers.hpp"

namespace caffe {

namespace {
// Fillers that occur in the deploy files (include/caffe/filler.hpp): constant, gaussian, bilinear.
// Deploy nets carry no fillers for conv/IP weights => constant 0 (caffe.proto:45-46); real weights come from
// a .caffemodel or are injected through layer->blobs().
void Fill(const FillerParameter& fp, Blob<float>* blob, unsigned seed) {
  float* d = blob->mutable_cpu_data();
  const string t = fp.type();
  if (t == "constant") {
    if (fp.value() != 0.f)                      // freshly allocated host memory is already zero (SyncedMemory::to_cpu)
      for (int i = 0; i < blob->count(); ++i) d[i] = fp.value();
  } else if (t == "gaussian") {
    std::mt19937 gen(seed);
    std::normal_distribution<float> dist(fp.mean(), fp.std());
    for (int i = 0; i < blob->count(); ++i) d[i] = dist(gen);
  } else if (t == "bilinear") {   // filler.hpp:244-262
    CHECK_EQ(blob->num_axes(), 4) << "Blob must be 4 dim.";
    CHECK_EQ(blob->width(), blob->height()) << "Filter must be square";
    const int f = (int)std::ceil(blob->width() / 2.);
    const float c = (2 * f - 1 - f % 2) / (2. * f);
    for (int i = 0; i < blob->count(); ++i) {
      const float x = i % blob->width();
      const float y = (i / blob->width()) % blob->height();
      d[i] = (1 - std::fabs(x / f - c)) * (1 - std::fabs(y / f - c));
    }
  } else {
    LOG(FATAL) << "Unknown filler name: " << t;
  }
}

// base_conv_layer.cpp:12-183, 2-D / dilation-1 subset
void ParseConvParam(const ConvolutionParameter& p, int* kh, int* kw, int* ph, int* pw, int* sh, int* sw) {
  if (p.has_kernel_h() || p.has_kernel_w()) {
    CHECK_EQ(0, p.kernel_size_size()) << "Either kernel_size or kernel_h/w should be specified; not both.";
    *kh = p.kernel_h(); *kw = p.kernel_w();
  } else {
    const int n = p.kernel_size_size();
    CHECK(n == 1 || n == 2) << "kernel_size must be specified once, or once per spatial dimension (kernel_size specified " << n << " times)";
    *kh = p.kernel_size(0); *kw = p.kernel_size(n == 1 ? 0 : 1);
  }
  CHECK_GT(*kh, 0) << "Filter dimensions must be nonzero.";
  CHECK_GT(*kw, 0) << "Filter dimensions must be nonzero.";
  if (p.has_stride_h() || p.has_stride_w()) {
    CHECK_EQ(0, p.stride_size()) << "Either stride or stride_h/w should be specified; not both.";
    *sh = p.stride_h(); *sw = p.stride_w();
  } else {
    const int n = p.stride_size();
    CHECK(n <= 2) << "stride must be specified at most once per spatial dimension";
    *sh = n ? p.stride(0) : 1; *sw = n ? p.stride(n == 1 ? 0 : 1) : 1;
  }
  CHECK_GT(*sh, 0); CHECK_GT(*sw, 0);
  if (p.has_pad_h() || p.has_pad_w()) {
    CHECK_EQ(0, p.pad_size()) << "Either pad or pad_h/w should be specified; not both.";
    *ph = p.pad_h(); *pw = p.pad_w();
  } else {
    const int n = p.pad_size();
    CHECK(n <= 2) << "pad must be specified at most once per spatial dimension";
    *ph = n ? p.pad(0) : 0; *pw = n ? p.pad(n == 1 ? 0 : 1) : 0;
  }
  for (int i = 0; i < p.dilation_size(); ++i) CHECK_EQ(p.dilation(i), 1u) << "dilated convolution is not part of the MS-CNN path";
}

inline void* S() { return Caffe::stream(); }
}  // namespace

// ------------------------------------------------------------------------------------------------ Input
template <typename Dtype>
void InputLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const InputParameter p = this->layer_param_.input_param();
  const int num_shape = p.shape_size();
  CHECK(num_shape == 0 || num_shape == 1 || num_shape == (int)top.size())
      << "Must specify 'shape' once, once per top blob, or not at all: " << top.size() << " tops vs. " << num_shape << " shapes.";
  for (size_t i = 0; i < top.size() && num_shape > 0; ++i) {
    const BlobShape s = p.shape(num_shape == 1 ? 0 : (int)i);
    vector<int> shape;
    for (int d = 0; d < s.dim_size(); ++d) shape.push_back((int)s.dim(d));
    top[i]->Reshape(shape);
  }
}

// ------------------------------------------------------------------------------------------------ Split
template <typename Dtype>
void SplitLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  for (size_t i = 0; i < top.size(); ++i) {
    CHECK_NE(top[i], bottom[0]) << this->type() << " Layer does not allow in-place computation.";
    top[i]->ReshapeLike(*bottom[0]);
    top[i]->ShareData(*bottom[0]);
  }
}
template <typename Dtype>
void SplitLayer<Dtype>::Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  for (size_t i = 0; i < top.size(); ++i) top[i]->ShareData(*bottom[0]);
}

// ------------------------------------------------------------------------------------------------ Convolution
template <typename Dtype>
ConvolutionLayer<Dtype>::~ConvolutionLayer() {
  if (band_.inflight && band_.done) (void)hipEventSynchronize((hipEvent_t)band_.done);      // its kernels read this layer's weights
  if (band_.plan) mscnn_conv2d_plan_destroy(band_.plan);
  if (band_.done) (void)hipEventDestroy((hipEvent_t)band_.done);
  if (plan_) mscnn_conv2d_plan_destroy(plan_);
}

template <typename Dtype>
void ConvolutionLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const ConvolutionParameter p = this->layer_param_.convolution_param();
  CHECK_EQ(bottom[0]->num_axes(), 4) << "2-D convolution expects (N, C, H, W) bottoms";
  CHECK_EQ(p.axis(), 1);
  ParseConvParam(p, &kernel_h_, &kernel_w_, &pad_h_, &pad_w_, &stride_h_, &stride_w_);
  channels_ = bottom[0]->channels();
  num_output_ = p.num_output();
  CHECK_GT(num_output_, 0);
  group_ = p.group();
  CHECK_EQ(channels_ % group_, 0);
  CHECK_EQ(num_output_ % group_, 0) << "Number of output should be multiples of group.";
  bias_term_ = p.bias_term();
  // base_conv_layer.cpp:130-160: blobs_[0] = [Cout][Cin/g][Kh][Kw], blobs_[1] = [Cout]
  if (this->blobs_.size() > 0) {
    CHECK_EQ((size_t)(1 + bias_term_), this->blobs_.size()) << "Incorrect number of weight blobs.";
  } else {
    this->blobs_.resize(bias_term_ ? 2 : 1);
    this->blobs_[0].reset(new Blob<Dtype>(num_output_, channels_ / group_, kernel_h_, kernel_w_));
    Fill(p.weight_filler(), this->blobs_[0].get(), 1701);
    if (bias_term_) {
      this->blobs_[1].reset(new Blob<Dtype>(vector<int>(1, num_output_)));
      Fill(p.bias_filler(), this->blobs_[1].get(), 1702);
    }
  }
  weights_dirty_ = true;
}

template <typename Dtype>
void ConvolutionLayer<Dtype>::Plan(int n, int h, int w) {
  if (plan_ && planned_h_ == h && planned_w_ == w) {
    if (planned_n_ != n) {
      const unsigned long long before = mscnn_conv2d_plan_weight_layout(plan_);
      const size_t bytes_before = mscnn_conv2d_packed_weight_bytes(plan_);
      MSCNN_CHECK(mscnn_conv2d_plan_set_batch(plan_, n));
      planned_n_ = n;
      // another kernel family, or a packed buffer of another size (the head kernels keep per-tile counters behind the weights): re-pack
      if (mscnn_conv2d_plan_weight_layout(plan_) != before || mscnn_conv2d_packed_weight_bytes(plan_) != bytes_before) weights_dirty_ = true;
    }
    return;
  }
  if (plan_) mscnn_conv2d_plan_destroy(plan_);
  plan_ = nullptr;
  mscnn_conv_desc d;
  d.N = n; d.Cin = channels_; d.H = h; d.W = w; d.Cout = num_output_; d.Kh = kernel_h_; d.Kw = kernel_w_;
  d.pad_h = pad_h_; d.pad_w = pad_w_; d.stride_h = stride_h_; d.stride_w = stride_w_; d.group = group_; d.relu = relu_ ? 1 : 0;
  d.algo = algo_; d.tune_variant = tune_[0]; d.tune_grid = tune_[1]; d.tune_flags = tune_[2];
  MSCNN_CHECK(mscnn_conv2d_plan_create(&d, &plan_));
  if (profiling_) MSCNN_CHECK(mscnn_conv2d_plan_set_profiling(plan_, 1));
  planned_n_ = n; planned_h_ = h; planned_w_ = w;
  weights_dirty_ = true;
}

template <typename Dtype>
void ConvolutionLayer<Dtype>::set_algo(int algo) {
  CHECK(algo >= 0 && algo <= 6) << "unknown mscnn_conv_algo " << algo;
  if (algo == algo_) return;
  algo_ = algo;
  if (algo != MSCNN_CONV_ALGO_DIRECT) { selfcheck_pending_ = true; wino_checked_ = false; }      // another arithmetic: its first result is checked again
  if (plan_) { mscnn_conv2d_plan_destroy(plan_); plan_ = nullptr; }
}
template <typename Dtype>
bool ConvolutionLayer<Dtype>::publishes_amax() const {
  return plan_ && amax_wanted_ && amax_out_ && mscnn_conv2d_plan_publishes_amax(plan_);
}
template <typename Dtype>
void ConvolutionLayer<Dtype>::set_tuning(int variant, int grid, int flags) {
  tune_[0] = variant; tune_[1] = grid; tune_[2] = flags;
  if (plan_) { mscnn_conv2d_plan_destroy(plan_); plan_ = nullptr; }
}
template <typename Dtype>
double ConvolutionLayer<Dtype>::ExecutedFlops() const { return plan_ ? mscnn_conv2d_plan_executed_flops(plan_) : 0; }
template <typename Dtype>
void ConvolutionLayer<Dtype>::set_profiling(bool on) {
  profiling_ = on;
  if (plan_) MSCNN_CHECK(mscnn_conv2d_plan_set_profiling(plan_, on ? 1 : 0));
}
template <typename Dtype>
bool ConvolutionLayer<Dtype>::StageMs(float ms[3]) const {
  ms[0] = ms[1] = ms[2] = 0.f;
  return plan_ && profiling_ && mscnn_conv2d_plan_stage_ms(plan_, ms) == 0;
}

namespace {
// scratch shared by every check of this host thread on this device (kept: a check on a live stream -- the numerics watch -- must
// not pay a hipMalloc + a synchronising hipFree of 100s of MB; ConvolutionLayer::ReleaseCheckScratch frees it, otherwise it is leaked
// on thread exit like the shared conv workspace)
struct CheckScratch { DeviceBuffer packed, ws, y, err; };
thread_local CheckScratch* g_check_scratch[64] = {nullptr};
}  // namespace

template <typename Dtype>
void ConvolutionLayer<Dtype>::ReleaseCheckScratch() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_check_scratch[dev]) { (void)hipGetLastError(); return; }
  delete g_check_scratch[dev];      // (DeviceBuffer's destructor frees)
  g_check_scratch[dev] = nullptr;
}

template <typename Dtype>
double ConvolutionLayer<Dtype>::ErrorAgainstDirect(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  last_check_vacuous_ = false;
  Plan(bottom[0]->num(), bottom[0]->height(), bottom[0]->width());
  const std::string kname = mscnn_conv2d_plan_kernel(plan_);
  if (kname.compare(0, 8, "winograd") != 0 || bottom[0]->count() == 0) return 0.0;
  if (roi_src_ && roi_src_->pending() && bottom[0] == roi_src_->window()) roi_src_->Materialize();      // the check reads the blob
  // a second, direct plan of the same layer with its own packed weights and workspace; everything is released on return
  mscnn_conv_desc d;
  d.N = bottom[0]->num(); d.Cin = channels_; d.H = bottom[0]->height(); d.W = bottom[0]->width(); d.Cout = num_output_;
  d.Kh = kernel_h_; d.Kw = kernel_w_; d.pad_h = pad_h_; d.pad_w = pad_w_; d.stride_h = stride_h_; d.stride_w = stride_w_;
  d.group = group_; d.relu = relu_ ? 1 : 0; d.algo = MSCNN_CONV_ALGO_DIRECT; d.tune_variant = d.tune_grid = d.tune_flags = 0;
  mscnn_conv_plan* dp = nullptr;
  MSCNN_CHECK(mscnn_conv2d_plan_create(&d, &dp));
  int sdev = 0;
  HIP_CHECK(hipGetDevice(&sdev));
  CHECK(sdev >= 0 && sdev < 64);
  if (!g_check_scratch[sdev]) g_check_scratch[sdev] = new CheckScratch();
  DeviceBuffer &packed = g_check_scratch[sdev]->packed, &ws = g_check_scratch[sdev]->ws, &y = g_check_scratch[sdev]->y, &err = g_check_scratch[sdev]->err;
  const size_t pb = mscnn_conv2d_packed_weight_bytes(dp), wb = mscnn_conv2d_workspace_bytes(dp);
  float* pk = pb ? static_cast<float*>(packed.Reserve(pb)) : nullptr;
  const float* w = this->blobs_[0]->gpu_data();
  MSCNN_CHECK(mscnn_conv2d_pack_weights(dp, w, pk, S()));
  float* yd = static_cast<float*>(y.Reserve(sizeof(float) * top[0]->count()));
  float* ed = static_cast<float*>(err.Reserve(sizeof(double) * 2));
  double* sd = reinterpret_cast<double*>(ed) + 1;
  MSCNN_CHECK(mscnn_conv2d_fwd_f32(dp, bottom[0]->gpu_data(), w, pk, bias_term_ ? this->blobs_[1]->gpu_data() : nullptr, yd,
                                   wb ? ws.Reserve(wb) : nullptr, wb, S()));
  // The metric: max |dy| / max(1, |y|, rms(y)).  On unit-scale activations this is the parity metric of the tests (rms ~ 1 .. 3); on
  // hot ones (rms 10 .. 100, trained nets) an element near zero is the difference of partial sums far larger than itself, where
  // two fp32 summation orders of the DIRECT form already differ by more than 1e-4 of 1 -- the floor follows the blob's scale.
  MSCNN_CHECK(mscnn_sum_squares_f32(bottom[0]->gpu_data(), (size_t)bottom[0]->count(), sd, S()));
  double ssx = 0.0;
  HIP_CHECK(hipMemcpyAsync(&ssx, sd, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)S()));
  MSCNN_CHECK(mscnn_sum_squares_f32(yd, (size_t)top[0]->count(), sd, S()));
  double ss = 0.0;
  HIP_CHECK(hipMemcpyAsync(&ss, sd, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)S()));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)S()));
  last_check_vacuous_ = !(ssx > 0.0);      // an all-zero bottom: both forms return the bias, the comparison is empty
  const float rms = (float)std::sqrt(ss / (double)top[0]->count());
  MSCNN_CHECK(mscnn_max_rel_diff_f32(top[0]->gpu_data(), yd, (size_t)top[0]->count(), rms > 1.0f ? rms : 1.0f, ed, S()));
  float e = 0.f;
  HIP_CHECK(hipMemcpyAsync(&e, ed, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)S()));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)S()));
  mscnn_conv2d_plan_destroy(dp);
  return e;
}

namespace {
// the band checks' own scratch (separate from CheckScratch: a band check may still be running on the device when a first-forward
// check of another layer reserves -- and possibly re-allocates -- its buffers)
struct BandScratch {
  DeviceBuffer x, packed, ws, y, scal;
  float* host = nullptr;      // pinned: one verdict word per layer that ever ran a band check on this (thread, device)
  int host_used = 0;
  static constexpr int kHostWords = 1024;
};
thread_local BandScratch* g_band_scratch[64] = {nullptr};
}  // namespace

template <typename Dtype>
bool ConvolutionLayer<Dtype>::BeginBandCheck(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top, int band_seq) {
  if (band_.inflight || bottom[0]->count() == 0 || top[0]->count() == 0 || top_stale_ || !plan_) return false;
  if (std::strncmp(mscnn_conv2d_plan_kernel(plan_), "winograd", 8) != 0) return false;
  const int N = bottom[0]->num(), H = bottom[0]->height(), W = bottom[0]->width();
  const int Ho = top[0]->height(), Wo = top[0]->width();
  const bool roi_pending = roi_src_ && roi_src_->pending() && bottom[0] == roi_src_->window();
  // the band: rows [r0, r0 + rows) of every image with a one-row halo (3x3 / pad 1 / stride 1 on a map of >= 32 rows), else
  // images [n0, n0 + bn) whole
  const bool by_rows = kernel_h_ == 3 && kernel_w_ == 3 && pad_h_ == 1 && pad_w_ == 1 && stride_h_ == 1 && stride_w_ == 1 && H >= 32 && !roi_pending;
  // its size: what the direct kernel recomputes in ~30 us (kBandFlops of the layer's direct-convolution FLOPs; the halo rows count),
  // never more than an eighth of the layer -- conv2_2: 6 of 288 rows, conv4_x: 2 of 72, roi_c1: 10 ROIs
  const double kBandFlops = 2.5e9, flops = std::max(1.0, mscnn_conv2d_plan_flops(plan_));
  int bn = N, n0 = 0, r0 = 0, rows = Ho, in0 = 0, bh = H;
  if (by_rows) {
    const int hb = std::max(2, std::min((H + 7) / 8, (int)(H * kBandFlops / flops) - 2)), nb = (H + hb - 1) / hb;
    r0 = (band_seq % nb) * hb;
    rows = std::min(hb, H - r0);
    in0 = std::max(r0 - 1, 0);
    bh = std::min(r0 + rows + 1, H) - in0;
  } else {
    const int per = std::max(1, std::min((N + 7) / 8, (int)(N * kBandFlops / flops))), nb = (N + per - 1) / per;
    n0 = (band_seq % nb) * per;
    bn = std::min(per, N - n0);
  }
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  CHECK(dev >= 0 && dev < 64);
  if (!g_band_scratch[dev]) {
    // everything a band check of ANY layer of these nets needs, once (a watch frame must not pay an allocation -- and the device
    // synchronisation of the hipFree behind a growing buffer -- the first time each layer's turn comes: that was + 0.6 ms on four frames
    // of the first trip round the layers, tools/sessions/r06_s13.sh): a 512 -> 512 / 1024 -> 512 direct plan's packed weights, the
    // direct kernel's stream-K slabs, the band and its output
    g_band_scratch[dev] = new BandScratch();
    BandScratch& s0 = *g_band_scratch[dev];
    s0.packed.Reserve((size_t)24 << 20); s0.ws.Reserve((size_t)96 << 20); s0.x.Reserve((size_t)8 << 20); s0.y.Reserve((size_t)8 << 20);
    s0.scal.Reserve(32);
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s0.host), sizeof(float) * BandScratch::kHostWords, hipHostMallocDefault));
  }
  BandScratch& sc = *g_band_scratch[dev];
  hipStream_t st = (hipStream_t)S();
  // the band of the bottom, contiguous
  const size_t plane_in = (size_t)H * W, band_in = (size_t)bh * W;
  const float* xb = nullptr;
  if (by_rows) {
    float* xd = static_cast<float*>(sc.x.Reserve(sizeof(float) * (size_t)N * channels_ * band_in));
    HIP_CHECK(hipMemcpy2DAsync(xd, sizeof(float) * band_in, bottom[0]->gpu_data() + (size_t)in0 * W, sizeof(float) * plane_in,
                               sizeof(float) * band_in, (size_t)N * channels_, hipMemcpyDeviceToDevice, st));
    xb = xd;
  } else if (roi_pending) {      // the R x 2C x ph x pw blob was never written (pooled inside roi_c1's input stage): pool the band's ROIs
    ROIPoolingLayer<Dtype>* a = roi_src_;
    ROIPoolingLayer<Dtype>* b = roi_src_->partner();
    const vector<Blob<Dtype>*>& pb = a->pending_bottoms();
    if (!b || pb.size() != 2 || pb[1]->num() != N || a->channels() + b->channels() != channels_) return false;
    float* xd = static_cast<float*>(sc.x.Reserve(sizeof(float) * (size_t)bn * channels_ * plane_in));
    ROIPoolingLayer<Dtype>* both[2] = {a, b};
    for (ROIPoolingLayer<Dtype>* r : both)
      MSCNN_CHECK(mscnn_roipool_fwd_f32(pb[0]->gpu_data(), pb[1]->gpu_data() + (size_t)n0 * 5, xd, bn, pb[0]->num(), r->channels(), pb[0]->height(),
                                        pb[0]->width(), r->pooled_height(), r->pooled_width(), r->spatial_scale(), r->pad_ratio(), channels_,
                                        r->window_c_offset(), st));
    xb = xd;
  } else {
    xb = bottom[0]->gpu_data() + (size_t)n0 * channels_ * plane_in;
  }
  // a direct plan of the band's shape with its own packed weights and workspace
  mscnn_conv_desc d;
  d.N = bn; d.Cin = channels_; d.H = bh; d.W = W; d.Cout = num_output_;
  d.Kh = kernel_h_; d.Kw = kernel_w_; d.pad_h = pad_h_; d.pad_w = pad_w_; d.stride_h = stride_h_; d.stride_w = stride_w_;
  d.group = group_; d.relu = relu_ ? 1 : 0; d.algo = MSCNN_CONV_ALGO_DIRECT; d.tune_variant = d.tune_grid = d.tune_flags = 0;
  if (band_.plan) { mscnn_conv2d_plan_destroy(band_.plan); band_.plan = nullptr; }
  MSCNN_CHECK(mscnn_conv2d_plan_create(&d, &band_.plan));
  const size_t pbytes = mscnn_conv2d_packed_weight_bytes(band_.plan), wbytes = mscnn_conv2d_workspace_bytes(band_.plan);
  float* pk = pbytes ? static_cast<float*>(sc.packed.Reserve(pbytes)) : nullptr;
  const float* w = this->blobs_[0]->gpu_data();
  MSCNN_CHECK(mscnn_conv2d_pack_weights(band_.plan, w, pk, S()));
  const int bho = by_rows ? bh : Ho;                                   // rows of the direct band's output
  const size_t ycount = (size_t)bn * num_output_ * bho * Wo;
  float* yd = static_cast<float*>(sc.y.Reserve(sizeof(float) * ycount));
  unsigned char* scal = static_cast<unsigned char*>(sc.scal.Reserve(32));
  double* ss = reinterpret_cast<double*>(scal);
  float* ed = reinterpret_cast<float*>(scal + 16);
  MSCNN_CHECK(mscnn_conv2d_fwd_f32(band_.plan, xb, w, pk, bias_term_ ? this->blobs_[1]->gpu_data() : nullptr, yd, wbytes ? sc.ws.Reserve(wbytes) : nullptr,
                                   wbytes, S()));
  MSCNN_CHECK(mscnn_sum_squares_f32(yd, ycount, ss, S()));
  // rows [r0 - in0, + rows) of the direct band are the ones whose halo is real (rows beside them saw the band's artificial zero padding)
  const float* tb = top[0]->gpu_data() + (by_rows ? (size_t)r0 * Wo : (size_t)n0 * num_output_ * Ho * Wo);
  if (by_rows)
    MSCNN_CHECK(mscnn_max_rel_diff_strided_f32(tb, (size_t)Ho * Wo, yd + (size_t)(r0 - in0) * Wo, (size_t)bho * Wo, (size_t)N * num_output_,
                                               (size_t)rows * Wo, ss, (double)ycount, ed, S()));
  else
    MSCNN_CHECK(mscnn_max_rel_diff_strided_f32(tb, ycount, yd, ycount, 1, ycount, ss, (double)ycount, ed, S()));
  if (!band_.host) {      // this layer's word of the (thread, device)'s pinned verdict array
    CHECK_LT(sc.host_used, BandScratch::kHostWords);
    band_.host = sc.host + sc.host_used++;
  }
  if (!band_.done) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); band_.done = e; }
  HIP_CHECK(hipMemcpyAsync(band_.host, ed, sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipEventRecord((hipEvent_t)band_.done, st));
  band_.inflight = true;
  return true;
}

template <typename Dtype>
int ConvolutionLayer<Dtype>::PollBandCheck(double* err, bool wait) {
  if (!band_.inflight) return 0;
  if (wait) HIP_CHECK(hipEventSynchronize((hipEvent_t)band_.done));
  else {
    const hipError_t q = hipEventQuery((hipEvent_t)band_.done);
    if (q == hipErrorNotReady) return 1;
    HIP_CHECK(q);
  }
  band_.inflight = false;
  *err = (double)band_.host[0];
  mscnn_conv2d_plan_destroy(band_.plan);
  band_.plan = nullptr;
  return 2;
}

template <typename Dtype>
bool ConvolutionLayer<Dtype>::FuseReLU(Dtype negative_slope) {
  if (negative_slope != 0) return false;
  relu_ = true;
  if (plan_) { mscnn_conv2d_plan_destroy(plan_); plan_ = nullptr; }
  return true;
}

template <typename Dtype>
void ConvolutionLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  CHECK_EQ(bottom[0]->channels(), channels_) << "Input size incompatible with convolution kernel.";
  CHECK_EQ(bottom.size(), 1u) << "one bottom per Convolution layer in this build";
  const int h = bottom[0]->height(), w = bottom[0]->width();
  // conv_layer.cpp:8-22
  const int oh = (h + 2 * pad_h_ - kernel_h_) / stride_h_ + 1, ow = (w + 2 * pad_w_ - kernel_w_) / stride_w_ + 1;
  top[0]->Reshape(bottom[0]->num(), num_output_, oh, ow);
}

template <typename Dtype>
double ConvolutionLayer<Dtype>::ForwardFlops() const { return plan_ ? mscnn_conv2d_plan_flops(plan_) : 0; }
template <typename Dtype>
const char* ConvolutionLayer<Dtype>::kernel_name() const {
  if (plan_ && last_fused_roipool_ && std::strcmp(mscnn_conv2d_plan_kernel(plan_), "winograd_f3x3_3x3") == 0) return "winograd_f3x3_3x3+roipool_pair";
  if (plan_ && last_chained_ && std::strcmp(mscnn_conv2d_plan_kernel(plan_), "winograd_f4x4_3x3") == 0) return "winograd_f4x4_3x3+into_next";
  return plan_ ? mscnn_conv2d_plan_kernel(plan_) : "";
}
template <typename Dtype>
const char* ConvolutionLayer<Dtype>::dtype() const { return plan_ ? mscnn_conv2d_plan_dtype(plan_) : "f32"; }

// (called by the head of a chain on each later member) plans the layer for the chain's shape and says whether it may take prepared
// planes this Forward: no first-forward self-check pending, packed weights in place or packable, fp32 F(4x4,3x3) is checked by the caller
template <typename Dtype>
bool ConvolutionLayer<Dtype>::ChainableNow(int n, int h, int w) {
  if (selfcheck_pending_ || roi_src_ || kernel_h_ != 3 || kernel_w_ != 3) return false;
  Plan(n, h, w);
  return plan_ != nullptr;
}

template <typename Dtype>
void ConvolutionLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  Plan(bottom[0]->num(), bottom[0]->height(), bottom[0]->width());
  // AUTO plans per shape: a layer whose first bottoms ran a direct kernel (roi_c1 with a handful of ROIs) may get a Winograd form for a
  // later one -- the first Winograd result on these weights is checked whenever it comes, not only in the first Forward
  if (!selfcheck_pending_ && !wino_checked_ && selfcheck_tol_ > 0 && algo_ != MSCNN_CONV_ALGO_DIRECT && algo_ != MSCNN_CONV_ALGO_F16 &&
      std::strncmp(mscnn_conv2d_plan_kernel(plan_), "winograd", 8) == 0)
    selfcheck_pending_ = true;
  const float* w = this->blobs_[0]->gpu_data();
  const size_t pbytes = mscnn_conv2d_packed_weight_bytes(plan_);
  float* packed = pbytes ? static_cast<float*>(packed_.Reserve(pbytes)) : nullptr;
  if (weights_dirty_) {
    MSCNN_CHECK(mscnn_conv2d_pack_weights(plan_, w, packed, S()));
    weights_dirty_ = false;
  }
  // Transient workspace (stream-K slabs, Winograd V / M planes: up to 0.8 GB for conv2_2): the layers of a net run one after
  // the other on one stream, so all conv layers of a host THREAD share ONE buffer per device instead of 3 GB of per-layer
  // buffers.  Thread-local like the Caffe singleton itself (a thread is a device context, common.cpp:13-20): two threads may
  // drive nets on the same device without sharing scratch.  (Leaked on thread exit on purpose: a destructor could run after
  // the HIP runtime has been torn down.)
  static thread_local DeviceBuffer* shared_ws[64] = {nullptr};
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  CHECK(dev >= 0 && dev < 64);
  if (!shared_ws[dev]) shared_ws[dev] = new DeviceBuffer();
  const size_t wbytes = mscnn_conv2d_workspace_bytes(plan_);
  // A chain of same-resolution F(4x4,3x3) layers (ChainTo): the head -- the first member whose planes nobody prepared -- plans every
  // member, decides how far the chain goes this Forward and gives each member its own region of the shared buffer (a member's output
  // stage writes the next member's planes while its own M is still being read), reserving the whole extent ONCE: a Reserve that
  // grows the buffer later would lose the planes already written.
  const bool was_prepared = prepared_;
  prepared_ = false;
  if (!was_prepared) {
    ws_off_ = 0;
    fuse_next_now_ = false;
    size_t extent = wbytes;
    if (chain_live_ && chain_next_ && !pooled_top_ && !selfcheck_pending_ && !(roi_src_ && roi_src_->pending())) {
      const int n = bottom[0]->num(), h = bottom[0]->height(), wd = bottom[0]->width();
      for (ConvolutionLayer* c = this; c->chain_next_ && c->chain_live_ && !c->pooled_top_ && c->chain_next_->ChainableNow(n, h, wd) &&
                                       mscnn_conv2d_plan_can_chain(c->plan_, c->chain_next_->plan_); c = c->chain_next_) {
        ConvolutionLayer* nx = c->chain_next_;
        const size_t cb = (mscnn_conv2d_workspace_bytes(c->plan_) + 255) / 256 * 256, nb = mscnn_conv2d_workspace_bytes(nx->plan_);
        c->fuse_next_now_ = true;
        c->next_off_ = c->ws_off_ >= nb ? 0 : c->ws_off_ + cb;      // ping-pong: back to the front when the front region is large enough
        nx->ws_off_ = c->next_off_;
        nx->fuse_next_now_ = false;
        extent = std::max(extent, std::max(c->ws_off_ + cb, nx->ws_off_ + nb));
      }
    }
    if (extent) shared_ws[dev]->Reserve(extent);
  }
  void* ws = wbytes ? static_cast<unsigned char*>(shared_ws[dev]->Reserve(ws_off_ + wbytes)) + ws_off_ : nullptr;
  const float* bias = bias_term_ ? this->blobs_[1]->gpu_data() : nullptr;
  {
    const bool pub = amax_wanted_ && amax_out_ && mscnn_conv2d_plan_publishes_amax(plan_);
    const bool handed = amax_trusted_ && amax_src_ && amax_in_ && amax_src_->publishes_amax();
    MSCNN_CHECK(mscnn_conv2d_plan_set_amax_io(plan_, handed ? amax_in_ : nullptr, pub ? amax_out_ : nullptr));
  }
  // The bottom is the concatenation of a deferred ROIPooling pair that has not been written: pool inside the input stage when the
  // planned kernel can (fp32 F(3x3,3x3) on 7 x 7 maps) -- the R x 2C x 7 x 7 blob is then neither written nor read -- else ask for it.
  if (roi_src_ && roi_src_->pending() && bottom[0] == roi_src_->window()) {
    ROIPoolingLayer<Dtype>* a = roi_src_;
    ROIPoolingLayer<Dtype>* b = roi_src_->partner();
    ROIPoolingLayer<Dtype>* lo = a->window_c_offset() == 0 ? a : b;
    ROIPoolingLayer<Dtype>* hi = lo == a ? b : a;
    const int C = a->channels();
    const vector<Blob<Dtype>*>& pb = a->pending_bottoms();
    if (!selfcheck_pending_ && !pooled_top_ && b && lo->window_c_offset() == 0 && hi->window_c_offset() == C && pb.size() == 2 &&
        mscnn_conv2d_plan_can_fuse_roipool(plan_, C, a->pooled_height(), a->pooled_width())) {
      // maps the Net had built under BoxOutput's host round trip (PrebuildRoiMaps), if they are of this very feature blob
      const float* maps = roi_maps_feat_ && roi_maps_feat_ == pb[0]->gpu_data() && roi_maps_shape_[0] == pb[0]->num() && roi_maps_shape_[1] == C &&
                                  roi_maps_shape_[2] == pb[0]->height() && roi_maps_shape_[3] == pb[0]->width()
                              ? static_cast<const float*>(roi_maps_.get()) : nullptr;
      roi_maps_feat_ = nullptr;
      const size_t fbytes = maps ? wbytes : mscnn_conv2d_roipool_workspace_bytes(plan_, pb[0]->num(), C, pb[0]->height(), pb[0]->width());
      void* fws = shared_ws[dev]->Reserve(fbytes);
      MSCNN_CHECK(mscnn_conv2d_fwd_roipool_pair_f32(plan_, pb[0]->gpu_data(), maps, pb[0]->num(), C, pb[0]->height(), pb[0]->width(),
                                                    pb[1]->gpu_data(), a->spatial_scale(), lo->pad_ratio(), hi->pad_ratio(), packed, bias,
                                                    top[0]->mutable_gpu_data(), fws, fbytes, S()));
      last_fused_roipool_ = true;
      return;
    }
    a->Materialize();
  }
  last_fused_roipool_ = false;
  float* pooled = nullptr;
  if (pooled_top_) {
    // the fused-away Pooling layer's Reshape would run AFTER this Forward (layer.hpp:451-456): shape its top here, so that a
    // bottom reshaped between two Forward calls (legal without Net::Reshape) propagates as in the reference
    pooled_top_->Reshape(top[0]->num(), top[0]->channels(), (top[0]->height() + 1) / 2, (top[0]->width() + 1) / 2);
    if (mscnn_conv2d_plan_can_pool(plan_)) pooled = pooled_top_->mutable_gpu_data();
  }
  last_chained_ = false;
  const bool pool_only = pool_only_live_ && !keep_top_ && pooled && !selfcheck_pending_ && mscnn_conv2d_plan_can_pool_only(plan_);
  if (was_prepared || fuse_next_now_) {
    // a member of a running chain: planes prepared by the previous member (x = NULL) and / or written for the next one (its top blob
    // is then NOT written: top_stale_)
    ConvolutionLayer* nx = fuse_next_now_ ? chain_next_ : nullptr;
    const size_t nb = nx ? mscnn_conv2d_workspace_bytes(nx->plan_) : 0;
    void* nws = nx ? static_cast<unsigned char*>(shared_ws[dev]->Reserve(next_off_ + nb)) + next_off_ : nullptr;
    MSCNN_CHECK(mscnn_conv2d_fwd_chain_f32(plan_, nx ? nx->plan_ : nullptr, was_prepared ? nullptr : bottom[0]->gpu_data(), packed, bias,
                                           (nx && !keep_top_) || pool_only ? nullptr : top[0]->mutable_gpu_data(), nx ? nullptr : pooled, ws, wbytes, nws,
                                           nb, S()));
    if (nx) { nx->prepared_ = true; top_stale_ = !keep_top_; last_chained_ = true; }
    else top_stale_ = pool_only;
    fuse_next_now_ = false;
  } else {
    MSCNN_CHECK(mscnn_conv2d_fwd_pool_f32(plan_, bottom[0]->gpu_data(), w, packed, bias, pool_only ? nullptr : top[0]->mutable_gpu_data(),
                                          pooled, ws, wbytes, S()));
    top_stale_ = pool_only;
  }
  if (pooled_top_ && !pooled) {
    // the planned kernel has no pooling epilogue (e.g. a direct-kernel shape): run the pooling the fused-away layer would have
    MSCNN_CHECK(mscnn_pool2d_fwd_f32(top[0]->gpu_data(), pooled_top_->mutable_gpu_data(), top[0]->num(), top[0]->channels(),
                                     top[0]->height(), top[0]->width(), 2, 2, 0, 0, 2, 2, 0, S()));
  }
  // Safe by default (header): a Winograd result is never handed out unchecked on new weights -- compare it with the direct kernel
  // on this very bottom once; off by more than the tolerance => direct kernel from now on, tops recomputed before returning.
  if (selfcheck_pending_ && bottom[0]->count() > 0) {
    if (!(selfcheck_tol_ > 0 && algo_ != MSCNN_CONV_ALGO_DIRECT && algo_ != MSCNN_CONV_ALGO_F16)) {
      selfcheck_pending_ = false;      // opted out / an algorithm with nothing to check
    } else if (std::strncmp(mscnn_conv2d_plan_kernel(plan_), "winograd", 8) == 0) {
      selfcheck_pending_ = false;
      selfcheck_err_ = ErrorAgainstDirect(bottom, top);
      if (last_check_vacuous_) selfcheck_pending_ = true;      // a zero warm-up frame checks nothing: the next bottom is checked again
      else selfcheck_ran_ = wino_checked_ = true;
      if (!last_check_vacuous_ && !(selfcheck_err_ <= selfcheck_tol_)) {      // (NaN counts as a failure)
        LOG(WARNING) << "layer " << this->layer_param_.name() << ": Winograd result off the direct sum by " << selfcheck_err_ << " > "
                     << selfcheck_tol_ << " on the first input after a weight change: using the direct kernel";
        set_algo(MSCNN_CONV_ALGO_DIRECT);
        calibrated_direct_ = true;
        selfcheck_fell_back_ = true;
        Forward_gpu(bottom, top);
      }
    }
    else selfcheck_pending_ = false;      // a direct kernel for THIS shape (see the re-arming at the top of Forward_gpu)
  }
}

template <typename Dtype>
void ConvolutionLayer<Dtype>::PrebuildRoiMaps(const Blob<Dtype>* feat) {
  roi_maps_feat_ = nullptr;
  // only where the last Forward took the fused path and the coming one has no reason not to (a pending first-forward check makes it
  // ask for the blob instead); a wrong guess costs the maps' 70 us, never a wrong result (Forward checks the pointer and the shape)
  ROIPoolingLayer<Dtype>* a = roi_src_;
  if (!a || !last_fused_roipool_ || selfcheck_pending_ || pooled_top_ || feat->count() == 0 || feat->channels() != a->channels()) return;
  const int N = feat->num(), C = feat->channels(), H = feat->height(), W = feat->width();
  float* maps = static_cast<float*>(roi_maps_.Reserve(mscnn_roipool_maps_bytes(N, C, H, W)));
  MSCNN_CHECK(mscnn_roipool_maps_build_f32(feat->gpu_data(), maps, N, C, H, W, S()));
  roi_maps_feat_ = feat->gpu_data();
  roi_maps_shape_[0] = N; roi_maps_shape_[1] = C; roi_maps_shape_[2] = H; roi_maps_shape_[3] = W;
}

template <typename Dtype>
bool ConvolutionLayer<Dtype>::FusePool2x2(Blob<Dtype>* pooled_top) {
  pooled_top_ = pooled_top;
  return true;
}

// ------------------------------------------------------------------------------------------------ Deconvolution
template <typename Dtype>
void DeconvolutionLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const ConvolutionParameter p = this->layer_param_.convolution_param();
  ParseConvParam(p, &kernel_h_, &kernel_w_, &pad_h_, &pad_w_, &stride_h_, &stride_w_);
  channels_ = bottom[0]->channels();
  num_output_ = p.num_output();
  group_ = p.group();
  CHECK_GT(num_output_, 0);
  CHECK_EQ(channels_ % group_, 0);
  CHECK_EQ(num_output_ % group_, 0) << "Number of output should be multiples of group.";
  bias_term_ = p.bias_term();
  if (this->blobs_.size() == 0) {
    this->blobs_.resize(bias_term_ ? 2 : 1);
    // reversed dims for deconv: [Cin][Cout/g][Kh][Kw] (base_conv_layer.cpp:135-140 with reverse_dimensions())
    this->blobs_[0].reset(new Blob<Dtype>(channels_, num_output_ / group_, kernel_h_, kernel_w_));
    Fill(p.weight_filler(), this->blobs_[0].get(), 1703);
    if (bias_term_) {
      this->blobs_[1].reset(new Blob<Dtype>(vector<int>(1, num_output_)));
      Fill(p.bias_filler(), this->blobs_[1].get(), 1704);
    }
  }
}
template <typename Dtype>
void DeconvolutionLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  // deconv_layer.cpp:8-23
  const int oh = stride_h_ * (bottom[0]->height() - 1) + kernel_h_ - 2 * pad_h_;
  const int ow = stride_w_ * (bottom[0]->width() - 1) + kernel_w_ - 2 * pad_w_;
  top[0]->Reshape(bottom[0]->num(), num_output_, oh, ow);
}
template <typename Dtype>
void DeconvolutionLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  MSCNN_CHECK(mscnn_deconv2d_fwd_f32(bottom[0]->gpu_data(), this->blobs_[0]->gpu_data(),
                                     bias_term_ ? this->blobs_[1]->gpu_data() : nullptr, top[0]->mutable_gpu_data(),
                                     bottom[0]->num(), channels_, bottom[0]->height(), bottom[0]->width(), num_output_, kernel_h_,
                                     kernel_w_, pad_h_, pad_w_, stride_h_, stride_w_, group_, S()));
}

// ------------------------------------------------------------------------------------------------ Pooling
template <typename Dtype>
void PoolingLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const PoolingParameter p = this->layer_param_.pooling_param();   // pooling_layer.cpp:17-76
  global_pooling_ = p.global_pooling();
  if (global_pooling_) {
    CHECK(!(p.has_kernel_size() || p.has_kernel_h() || p.has_kernel_w())) << "With Global_pooling: true Filter size cannot specified";
    kernel_h_ = bottom[0]->height(); kernel_w_ = bottom[0]->width();
  } else {
    CHECK(!p.has_kernel_size() != !(p.has_kernel_h() && p.has_kernel_w())) << "Filter size is kernel_size OR kernel_h and kernel_w; not both";
    if (p.has_kernel_size()) kernel_h_ = kernel_w_ = p.kernel_size();
    else { kernel_h_ = p.kernel_h(); kernel_w_ = p.kernel_w(); }
  }
  CHECK_GT(kernel_h_, 0); CHECK_GT(kernel_w_, 0);
  if (!p.has_pad_h()) pad_h_ = pad_w_ = p.pad(); else { pad_h_ = p.pad_h(); pad_w_ = p.pad_w(); }
  if (!p.has_stride_h()) stride_h_ = stride_w_ = p.stride(); else { stride_h_ = p.stride_h(); stride_w_ = p.stride_w(); }
  if (pad_h_ != 0 || pad_w_ != 0) {
    CHECK(p.pool() == PoolingParameter_PoolMethod_AVE || p.pool() == PoolingParameter_PoolMethod_MAX) << "Padding implemented only for average and max pooling.";
    CHECK_LT(pad_h_, kernel_h_); CHECK_LT(pad_w_, kernel_w_);
  }
  CHECK(p.pool() != PoolingParameter_PoolMethod_STOCHASTIC) << "stochastic pooling is training-only";
  method_ = p.pool() == PoolingParameter_PoolMethod_MAX ? 0 : 1;
}
template <typename Dtype>
void PoolingLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  CHECK_EQ(4, bottom[0]->num_axes()) << "Input must have 4 axes, corresponding to (num, channels, height, width)";
  channels_ = bottom[0]->channels(); height_ = bottom[0]->height(); width_ = bottom[0]->width();
  if (global_pooling_) { kernel_h_ = height_; kernel_w_ = width_; }
  pooled_height_ = mscnn_pool_out_dim(height_, kernel_h_, pad_h_, stride_h_);   // pooling_layer.cpp:90-107 (ceil)
  pooled_width_ = mscnn_pool_out_dim(width_, kernel_w_, pad_w_, stride_w_);
  top[0]->Reshape(bottom[0]->num(), channels_, pooled_height_, pooled_width_);
}
template <typename Dtype>
void PoolingLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  MSCNN_CHECK(mscnn_pool2d_fwd_f32(bottom[0]->gpu_data(), top[0]->mutable_gpu_data(), bottom[0]->num(), channels_, height_, width_,
                                   kernel_h_, kernel_w_, pad_h_, pad_w_, stride_h_, stride_w_, method_, S()));
}

// ------------------------------------------------------------------------------------------------ ReLU
template <typename Dtype>
void ReLULayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  MSCNN_CHECK(mscnn_relu_fwd_f32(bottom[0]->gpu_data(), top[0]->mutable_gpu_data(), (size_t)bottom[0]->count(),
                                 this->layer_param_.relu_param().negative_slope(), S()));
}

// ------------------------------------------------------------------------------------------------ InnerProduct
template <typename Dtype>
void InnerProductLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const InnerProductParameter p = this->layer_param_.inner_product_param();   // inner_product_layer.cpp:10-55
  N_ = p.num_output();
  bias_term_ = p.bias_term();
  CHECK(!p.transpose()) << "transpose: true is not used by the MS-CNN nets";
  const int axis = bottom[0]->CanonicalAxisIndex(p.axis());
  K_ = bottom[0]->count(axis);
  if (this->blobs_.size() == 0) {
    this->blobs_.resize(bias_term_ ? 2 : 1);
    vector<int> ws(2); ws[0] = N_; ws[1] = K_;
    this->blobs_[0].reset(new Blob<Dtype>(ws));
    Fill(p.weight_filler(), this->blobs_[0].get(), 1705);
    if (bias_term_) {
      this->blobs_[1].reset(new Blob<Dtype>(vector<int>(1, N_)));
      Fill(p.bias_filler(), this->blobs_[1].get(), 1706);
    }
  }
}
template <typename Dtype>
void InnerProductLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const int axis = bottom[0]->CanonicalAxisIndex(this->layer_param_.inner_product_param().axis());
  const int new_K = bottom[0]->count(axis);
  CHECK_EQ(K_, new_K) << "Input size incompatible with inner product parameters.";
  M_ = bottom[0]->count(0, axis);
  vector<int> top_shape = bottom[0]->shape();
  top_shape.resize(axis + 1);
  top_shape[axis] = N_;
  top[0]->Reshape(top_shape);
}
template <typename Dtype>
void InnerProductLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  used_wg_ = false;
  used_x3_ = x3_ && mscnn_inner_product_x3_supported(N_, K_);
  if (used_x3_) {
    void* pk = w16_.Reserve(mscnn_inner_product_x3_packed_bytes(N_, K_));
    if (w16_dirty_) {
      MSCNN_CHECK(mscnn_inner_product_x3_pack(this->blobs_[0]->gpu_data(), pk, N_, K_, S()));
      w16_dirty_ = false;
    }
    const size_t wb = mscnn_inner_product_x3_workspace_bytes(M_, N_, K_);
    const bool handed = amax_trusted_ && amax_src_ && amax_in_ && amax_src_->publishes_amax();
    MSCNN_CHECK(mscnn_inner_product_x3_fwd(bottom[0]->gpu_data(), pk, bias_term_ ? this->blobs_[1]->gpu_data() : nullptr,
                                           top[0]->mutable_gpu_data(), M_, N_, K_, relu_ ? 1 : 0, handed ? amax_in_ : nullptr,
                                           x3_ws_.Reserve(wb), wb, S()));
    used_f16_ = false;
    return;
  }
  used_f16_ = f16_ && mscnn_inner_product_f16_supported(N_, K_);
  if (used_f16_) {
    void* w16 = w16_.Reserve((size_t)N_ * K_ * 2);
    if (w16_dirty_) {
      MSCNN_CHECK(mscnn_inner_product_pack_f16(this->blobs_[0]->gpu_data(), w16, N_, K_, S()));
      w16_dirty_ = false;
    }
    MSCNN_CHECK(mscnn_inner_product_fwd_f16(bottom[0]->gpu_data(), w16, bias_term_ ? this->blobs_[1]->gpu_data() : nullptr,
                                            top[0]->mutable_gpu_data(), M_, N_, K_, relu_ ? 1 : 0, S()));
    return;
  }
  // (after a hand-off time-out the process runs whole tiles only: fc6's 96 tiles would then occupy 96 of the 256 CUs -- 0.6 -> 1.6 ms,
  // 20 % of a 7s-576 frame, profiles/r06_ab_whole_tiles.txt -- so it goes back to the register-staged GEMM, whose k-split is summed
  // by a fix-up launch and never waits for another workgroup: + 30 us instead of + 1000)
  used_wg_ = algo_ == 0 && !mscnn_wgemm_whole_tiles_forced() && mscnn_inner_product_wg_supported(M_, N_, K_);
  if (used_wg_) {      // fc6-class: the plane-GEMM kernel (wgemm.hip) with the weights kept transposed
    float* wt = static_cast<float*>(wt_.Reserve(mscnn_inner_product_wg_packed_bytes(N_, K_)));
    if (wt_dirty_) {
      MSCNN_CHECK(mscnn_inner_product_wg_pack(this->blobs_[0]->gpu_data(), wt, N_, K_, S()));
      wt_dirty_ = false;
    }
    const size_t wb = mscnn_inner_product_wg_workspace_bytes(M_, N_, K_);
    MSCNN_CHECK(mscnn_inner_product_wg_fwd(bottom[0]->gpu_data(), wt, bias_term_ ? this->blobs_[1]->gpu_data() : nullptr,
                                           top[0]->mutable_gpu_data(), M_, N_, K_, relu_ ? 1 : 0, wg_ws_.Reserve(wb), wb, S()));
    return;
  }
  MSCNN_CHECK(mscnn_inner_product_fwd_f32(bottom[0]->gpu_data(), this->blobs_[0]->gpu_data(),
                                          bias_term_ ? this->blobs_[1]->gpu_data() : nullptr, top[0]->mutable_gpu_data(), M_, N_, K_,
                                          relu_ ? 1 : 0, S()));
}

// ------------------------------------------------------------------------------------------------ Concat
template <typename Dtype>
void ConcatLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const ConcatParameter p = this->layer_param_.concat_param();
  CHECK(!(p.raw().has("axis") && p.has_concat_dim())) << "Either axis or concat_dim should be specified; not both.";
}
template <typename Dtype>
void ConcatLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const ConcatParameter p = this->layer_param_.concat_param();   // concat_layer.cpp:17-55
  const int num_axes = bottom[0]->num_axes();
  concat_axis_ = p.has_concat_dim() ? (int)p.concat_dim() : bottom[0]->CanonicalAxisIndex(p.axis());
  CHECK_LT(concat_axis_, num_axes) << "concat axis out of range.";
  vector<int> top_shape = bottom[0]->shape();
  num_concats_ = bottom[0]->count(0, concat_axis_);
  concat_input_size_ = bottom[0]->count(concat_axis_ + 1);
  for (size_t i = 1; i < bottom.size(); ++i) {
    CHECK_EQ(num_axes, bottom[i]->num_axes()) << "All inputs must have the same #axes.";
    for (int j = 0; j < num_axes; ++j) {
      if (j == concat_axis_) continue;
      CHECK_EQ(top_shape[j], bottom[i]->shape(j)) << "All inputs must have the same shape, except at concat_axis.";
    }
    top_shape[concat_axis_] += bottom[i]->shape(concat_axis_);
  }
  top[0]->Reshape(top_shape);
  if (bottom.size() == 1) top[0]->ShareData(*bottom[0]);
}
template <typename Dtype>
void ConcatLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  if (bottom.size() == 1) return;
  Dtype* top_data = top[0]->mutable_gpu_data();
  const int top_concat_axis = top[0]->shape(concat_axis_);
  int offset = 0;
  for (size_t i = 0; i < bottom.size(); ++i) {
    const int c = bottom[i]->shape(concat_axis_);
    MSCNN_CHECK(mscnn_concat_channels_f32(bottom[i]->gpu_data(), top_data, num_concats_, c, concat_input_size_, top_concat_axis, offset, S()));
    offset += c;
  }
}

// ------------------------------------------------------------------------------------------------ Dropout (TEST)
template <typename Dtype>
void DropoutLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  CHECK(this->phase_ == TEST) << "Dropout: TRAIN phase is outside the inference path";
  if (top[0] == bottom[0]) return;                                  // in place: identity (dropout_layer.cpp:43-45)
  const Dtype* src = bottom[0]->gpu_data();
  Dtype* dst = top[0]->mutable_gpu_data();
  if (src != dst)
    HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(Dtype) * bottom[0]->count(), hipMemcpyDeviceToDevice, (hipStream_t)S()));
}

// ------------------------------------------------------------------------------------------------ Softmax
template <typename Dtype>
void SoftmaxLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  softmax_axis_ = bottom[0]->CanonicalAxisIndex(this->layer_param_.softmax_param().axis());
  top[0]->ReshapeLike(*bottom[0]);
  outer_num_ = bottom[0]->count(0, softmax_axis_);
  inner_num_ = bottom[0]->count(softmax_axis_ + 1);
}
template <typename Dtype>
void SoftmaxLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  MSCNN_CHECK(mscnn_softmax_fwd_f32(bottom[0]->gpu_data(), top[0]->mutable_gpu_data(), outer_num_, bottom[0]->shape(softmax_axis_), inner_num_, S()));
}

// ------------------------------------------------------------------------------------------------ ROIPooling
template <typename Dtype>
void ROIPoolingLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const ROIPoolingParameter p = this->layer_param_.roi_pooling_param();   // roi_pooling_layer.cpp:22-35
  CHECK_GT(p.pooled_h(), 0u) << "pooled_h must be > 0";
  CHECK_GT(p.pooled_w(), 0u) << "pooled_w must be > 0";
  pooled_height_ = p.pooled_h();
  pooled_width_ = p.pooled_w();
  spatial_scale_ = p.spatial_scale();
  pad_ratio_ = p.pad_ratio();
  LOG(INFO) << "Spatial scale: " << spatial_scale_;
}
template <typename Dtype>
void ROIPoolingLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  channels_ = bottom[0]->channels(); height_ = bottom[0]->height(); width_ = bottom[0]->width();
  top[0]->Reshape(bottom[1]->num(), channels_, pooled_height_, pooled_width_);   // :37-46
}
template <typename Dtype>
bool ROIPoolingLayer<Dtype>::PairWith(ROIPoolingLayer* b) {
  if (!b || b == this || !window_ || b->window_ != window_ || b->window_c_total_ != window_c_total_) return false;
  const ROIPoolingParameter pa = this->layer_param_.roi_pooling_param(), pb = b->layer_param_.roi_pooling_param();
  if (pa.pooled_h() != pb.pooled_h() || pa.pooled_w() != pb.pooled_w() || pa.spatial_scale() != pb.spatial_scale()) return false;
  partner_ = b;
  return true;
}

template <typename Dtype>
void ROIPoolingLayer<Dtype>::LaunchPair(const vector<Blob<Dtype>*>& bottom) {
  MSCNN_CHECK(mscnn_roipool_pair_fwd_f32(bottom[0]->gpu_data(), bottom[1]->gpu_data(), window_->mutable_gpu_data(), bottom[1]->num(),
                                         bottom[0]->num(), channels_, height_, width_, pooled_height_, pooled_width_, spatial_scale_,
                                         pad_ratio_, window_c_offset_, partner_->pad_ratio_, partner_->window_c_offset_, window_c_total_, S()));
}

template <typename Dtype>
void ROIPoolingLayer<Dtype>::Materialize() {
  if (!pending_) return;
  pending_ = false;
  if (pending_bottom_.size() == 2 && pending_bottom_[1]->num() > 0) LaunchPair(pending_bottom_);
}

template <typename Dtype>
void ROIPoolingLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  if (skip_) { skip_ = false; return; }      // the pair's first layer has written this layer's window in its launch
  Dtype* out = window_ ? nullptr : top[0]->mutable_gpu_data();
  int c_total = channels_, c_offset = 0;
  if (window_) {   // Net fused the following Concat away: write this layer's channels of the concatenated blob directly
    window_->Reshape(bottom[1]->num(), window_c_total_, pooled_height_, pooled_width_);
    c_total = window_c_total_; c_offset = window_c_offset_;
  }
  if (partner_ && window_ && partner_->channels_ == channels_) {
    if (deferred_) {      // the consumer pools inside its own input stage; the blob is written only when somebody asks (Materialize)
      pending_bottom_ = bottom;
      pending_ = true;
    } else {
      LaunchPair(bottom);
    }
    partner_->set_skip(true);
    return;
  }
  if (window_) out = window_->mutable_gpu_data();
  MSCNN_CHECK(mscnn_roipool_fwd_f32(bottom[0]->gpu_data(), bottom[1]->gpu_data(), out, bottom[1]->num(), bottom[0]->num(), channels_,
                                    height_, width_, pooled_height_, pooled_width_, spatial_scale_, pad_ratio_, c_total, c_offset, S()));
}

// ------------------------------------------------------------------------------------------------ ROIAlign
template <typename Dtype>
void ROIAlignLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const ROIPoolingParameter p = this->layer_param_.roi_pooling_param();   // roi_align_layer.cpp:22-37 (same message)
  CHECK_GT(p.pooled_h(), 0u) << "pooled_h must be > 0";
  CHECK_GT(p.pooled_w(), 0u) << "pooled_w must be > 0";
  pooled_height_ = p.pooled_h(); pooled_width_ = p.pooled_w();
  spatial_scale_ = p.spatial_scale(); pad_ratio_ = p.pad_ratio();
}
template <typename Dtype>
void ROIAlignLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  channels_ = bottom[0]->channels(); height_ = bottom[0]->height(); width_ = bottom[0]->width();
  top[0]->Reshape(bottom[1]->num(), channels_, pooled_height_ + 1, pooled_width_ + 1);   // :40-46 grid_height_/grid_width_
}
template <typename Dtype>
void ROIAlignLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  MSCNN_CHECK(mscnn_roialign_fwd_f32(bottom[0]->gpu_data(), bottom[1]->gpu_data(), top[0]->mutable_gpu_data(), bottom[1]->num(),
                                     bottom[0]->num(), channels_, height_, width_, pooled_height_, pooled_width_, spatial_scale_,
                                     pad_ratio_, S()));
}

// ------------------------------------------------------------------------------------------------ Eltwise
template <typename Dtype>
void EltwiseLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const EltwiseParameter p = this->layer_param_.eltwise_param();   // eltwise_layer.cpp:9-27
  CHECK(p.coeff_size() == 0 || p.coeff_size() == (int)bottom.size()) << "Eltwise Layer takes one coefficient per bottom blob.";
  CHECK(!(p.operation() == EltwiseParameter_EltwiseOp_PROD && p.coeff_size())) << "Eltwise layer only takes coefficients for summation.";
  op_ = (int)p.operation();
  coeffs_.assign(bottom.size(), 1.f);
  for (int i = 0; i < p.coeff_size(); ++i) coeffs_[i] = p.coeff(i);
  CHECK_LE(bottom.size(), 8u) << "Eltwise: at most 8 bottoms in this build";
}
template <typename Dtype>
void EltwiseLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  for (size_t i = 1; i < bottom.size(); ++i) CHECK(bottom[i]->shape() == bottom[0]->shape());
  top[0]->ReshapeLike(*bottom[0]);
}
template <typename Dtype>
void EltwiseLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const float* ptrs[8];
  for (size_t i = 0; i < bottom.size(); ++i) ptrs[i] = bottom[i]->gpu_data();
  MSCNN_CHECK(mscnn_eltwise_fwd_f32(ptrs, (int)bottom.size(), coeffs_.data(), top[0]->mutable_gpu_data(), (size_t)top[0]->count(), op_, S()));
}

// ------------------------------------------------------------------------------------------------ BoxOutput
template <typename Dtype>
void BoxOutputLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const BoxOutputParameter p = this->layer_param_.box_output_param();   // box_output_layer.cpp:19-26
  fg_thr_ = p.fg_thr();
  iou_thr_ = p.iou_thr();
  nms_type_ = p.nms_type();
  output_proposal_with_score_ = (top.size() == 2);
  cap_ = 0;
  last_rows_ = 1;
  // Parameter errors are reported at set-up rather than at the first Forward.  max_nms_num 0 (the caffe.proto default: no cap)
  // or above the 4032 boxes the LDS-resident sort / NMS holds is served by the tiled path of csrc/nms_large.h.
  mscnn_boxoutput_desc d;
  FillBoxOutputDesc(this->layer_param_, bottom, fg_thr_, iou_thr_, nms_type_, &d);
  CHECK_GT(mscnn_boxoutput_workspace_bytes(&d), 0u) << "BoxOutput layer '" << this->layer_param_.name() << "': " << mscnn_last_error();
}
template <typename Dtype>
void BoxOutputLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  // The reference does a dummy (1,5)/(1,6) reshape here on EVERY Forward (box_output_layer.cpp:29-36) and the real
  // one inside Forward_cpu.  Layer::Forward calls Reshape first (layer.hpp:451-456), so the observable shapes after
  // Forward are identical; between Net::Reshape() and Forward the tops are (1,5)/(1,6) as in the reference.
  top[0]->Reshape(1, 5, 1, 1);
  if (output_proposal_with_score_) top[1]->Reshape(1, 6, 1, 1);
}
// The layer's parameters + bottom shapes as the C ABI's descriptor (box_output_layer.cpp:80-103).
template <typename Dtype>
static void FillBoxOutputDesc(const LayerParameter& lp, const vector<Blob<Dtype>*>& bottom, float fg_thr, float iou_thr,
                              const string& nms_type, mscnn_boxoutput_desc* dp) {
  mscnn_boxoutput_desc& d = *dp;
  const BoxOutputParameter p = lp.box_output_param();
  const int n = (int)bottom.size();
  CHECK_EQ(n, p.field_h_size());        // :80-82
  CHECK_EQ(n, p.field_w_size());
  CHECK_EQ(n, p.downsample_rate_size());
  std::memset(&d, 0, sizeof(d));
  CHECK_LE(n, MSCNN_BOXOUT_MAX_HEADS);
  d.num_heads = n;
  d.num = bottom[0]->num();
  d.channels = bottom[0]->channels();
  for (int j = 0; j < n; ++j) {
    CHECK_EQ(bottom[j]->num(), d.num);
    CHECK_EQ(bottom[j]->channels(), d.channels);
    d.head_h[j] = bottom[j]->height(); d.head_w[j] = bottom[j]->width();
    d.field_w[j] = (float)p.field_w(j); d.field_h[j] = (float)p.field_h(j); d.downsample_rate[j] = (float)p.downsample_rate(j);
  }
  d.fg_thr = fg_thr; d.iou_thr = iou_thr;
  d.nms_mode = nms_type == "IOMU" ? 1 : nms_type == "IOFU" ? 2 : 0;   // BoxIOU: anything else is IOU (math_functions.cpp:26-32)
  d.field_whr = p.field_whr(); d.field_xyr = p.field_xyr();
  d.max_nms_num = (int)p.max_nms_num(); d.max_post_nms_num = (int)p.max_post_nms_num();
  d.min_size = p.min_size();
  const BBoxRegParameter br = lp.bbox_reg_param();
  if (br.bbox_mean_size() > 0 && br.bbox_std_size() > 0) {   // :92-103
    CHECK_EQ(br.bbox_mean_size(), 4); CHECK_EQ(br.bbox_std_size(), 4);
    d.do_bbox_norm = 1;
    for (int k = 0; k < 4; ++k) { d.bbox_mean[k] = br.bbox_mean(k); d.bbox_std[k] = br.bbox_std(k); }
  }
}

template <typename Dtype>
BoxOutputLayer<Dtype>::~BoxOutputLayer() {
  if (count_ready_) (void)hipEventDestroy((hipEvent_t)count_ready_);
  if (host_count_) (void)hipHostFree(host_count_);
}

template <typename Dtype>
void BoxOutputLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const int n = (int)bottom.size();
  mscnn_boxoutput_desc d;
  FillBoxOutputDesc(this->layer_param_, bottom, fg_thr_, iou_thr_, nms_type_, &d);
  const float* heads[MSCNN_BOXOUT_MAX_HEADS];
  for (int j = 0; j < n; ++j) heads[j] = bottom[j]->gpu_data();
  const size_t wbytes = mscnn_boxoutput_workspace_bytes(&d);
  CHECK_GT(wbytes, 0u) << mscnn_last_error();
  void* ws = workspace_.Reserve(wbytes);
  cap_ = mscnn_boxoutput_max_rows(&d);
  // the kernels write the tops themselves: the blobs are sized for the layer's row bound first (Blob::Reshape keeps the larger
  // allocation) and cut to R rows below -- the two D2D copies behind the host round trip are gone (r5)
  top[0]->Reshape(cap_, 5, 1, 1);
  float* rois = top[0]->mutable_gpu_data();
  float* props = nullptr;
  if (output_proposal_with_score_) {
    top[1]->Reshape(cap_, 6, 1, 1);
    props = top[1]->mutable_gpu_data();
  }
  // {R, real rows} land in host-coherent pinned memory straight from the last kernel's two stores (it only ever writes them): no D2H
  // copy sits in the stream between BoxOutput and what follows -- an 8-byte hipMemcpyAsync held the next kernel back for ~24 us of
  // copy-engine latency (profiles/r06_kernel_gaps.txt).  The host waits for an event behind the kernels, then reads the two words.
  if (!host_count_) {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host_count_), 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    void* dp = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dp, host_count_, 0));
    host_count_dev_ = static_cast<int*>(dp);
  }
  static_cast<volatile int*>(host_count_)[0] = -1;
  MSCNN_CHECK(mscnn_boxoutput_fwd_f32(&d, heads, rois, props, nullptr, cap_, host_count_dev_, ws, wbytes, S()));
  // The only host round trip of the layer: R (4 bytes) is needed to Reshape the tops (layer.hpp:451-456 propagates it
  // to ROIPooling and the detection sub-net).  The reference moves all 7 head blobs D2H and the ROIs H2D here.
  if (!count_ready_) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); count_ready_ = e; }
  HIP_CHECK(hipEventRecord((hipEvent_t)count_ready_, (hipStream_t)S()));
  // what the hook enqueues runs on the device while the host takes R and goes on (the wait is for BoxOutput's kernels, not for the stream)
  if (before_sync_) before_sync_();
  HIP_CHECK(hipEventSynchronize((hipEvent_t)count_ready_));
  const volatile int* host_count = host_count_;
  const int R = host_count[0];
  CHECK_GE(R, 1); CHECK_LE(R, cap_);
  last_rows_ = R;
  top[0]->Reshape(R, 5, 1, 1);
  if (output_proposal_with_score_) top[1]->Reshape(R, 6, 1, 1);
}

// ------------------------------------------------------------------------------------------------ DecodeBBox
template <typename Dtype>
void DecodeBBoxLayer<Dtype>::LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  const BBoxRegParameter p = this->layer_param_.bbox_reg_param();   // decode_bbox_layer.cpp:18-36
  if (p.bbox_mean_size() > 0 && p.bbox_std_size() > 0) {
    CHECK_EQ(p.bbox_mean_size(), 4); CHECK_EQ(p.bbox_std_size(), 4);
    for (int i = 0; i < 4; ++i) {
      bbox_mean_[i] = p.bbox_mean(i); bbox_std_[i] = p.bbox_std(i);
      CHECK_GT(bbox_std_[i], 0);
    }
  } else {
    for (int i = 0; i < 4; ++i) { bbox_mean_[i] = 0; bbox_std_[i] = 1; }
  }
}
template <typename Dtype>
void DecodeBBoxLayer<Dtype>::Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  CHECK_EQ(bottom[0]->num(), bottom[1]->num());   // :39-52
  CHECK(this->phase_ == TEST) << "DecodeBBox: TRAIN-phase filtering is outside the inference path";
  CHECK_EQ(bottom[0]->channels(), 8);
  CHECK_EQ(bottom[1]->channels(), 5);
  top[0]->ReshapeLike(*bottom[1]);
}
template <typename Dtype>
void DecodeBBoxLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
  MSCNN_CHECK(mscnn_decodebbox_fwd_f32(bottom[0]->gpu_data(), bottom[1]->gpu_data(), top[0]->mutable_gpu_data(), bottom[0]->num(),
                                       bottom[0]->channels(), bbox_mean_, bbox_std_, S()));
}

INSTANTIATE_CLASS(InputLayer);
INSTANTIATE_CLASS(SplitLayer);
INSTANTIATE_CLASS(ConvolutionLayer);
INSTANTIATE_CLASS(DeconvolutionLayer);
INSTANTIATE_CLASS(PoolingLayer);
INSTANTIATE_CLASS(ReLULayer);
INSTANTIATE_CLASS(InnerProductLayer);
INSTANTIATE_CLASS(ConcatLayer);
INSTANTIATE_CLASS(DropoutLayer);
INSTANTIATE_CLASS(SoftmaxLayer);
INSTANTIATE_CLASS(ROIPoolingLayer);
INSTANTIATE_CLASS(ROIAlignLayer);
INSTANTIATE_CLASS(EltwiseLayer);
INSTANTIATE_CLASS(BoxOutputLayer);
INSTANTIATE_CLASS(DecodeBBoxLayer);

REGISTER_LAYER_CLASS(Input);
REGISTER_LAYER_CLASS(Split);
REGISTER_LAYER_CLASS(Convolution);
REGISTER_LAYER_CLASS(Deconvolution);
REGISTER_LAYER_CLASS(Pooling);
REGISTER_LAYER_CLASS(ReLU);
REGISTER_LAYER_CLASS(InnerProduct);
REGISTER_LAYER_CLASS(Concat);
REGISTER_LAYER_CLASS(Dropout);
REGISTER_LAYER_CLASS(Softmax);
REGISTER_LAYER_CLASS(ROIPooling);
REGISTER_LAYER_CLASS(ROIAlign);
REGISTER_LAYER_CLASS(Eltwise);
REGISTER_LAYER_CLASS(BoxOutput);
REGISTER_LAYER_CLASS(DecodeBBox);

}  // namespace caffe
