// Layer<Dtype>: the drop-in boundary (mirror of include/caffe/layer.hpp:32-487).
// A layer subclasses it and overrides LayerSetUp / Reshape / Forward_cpu / Forward_gpu exactly as in
// the reference; Net (or a test, or `caffe time`) calls SetUp once and the non-virtual Forward, which
// Reshapes first (layer.hpp:451-456) -- that is how BoxOutput's data-dependent ROI count propagates.
// Differences, all on the inference path's side: no Backward, no loss weights, no layer mutex
// (layers are never shared across solver threads here); Forward_cpu of every shipped layer is fatal.
#ifndef MSCNN_CAFFE_LAYER_HPP_
#define MSCNN_CAFFE_LAYER_HPP_

#include <string>
#include <vector>

#include "caffe/blob.hpp"
#include "caffe/common.hpp"
#include "caffe/proto/caffe_param.hpp"

namespace caffe {

template <typename Dtype>
class Layer {
 public:
  explicit Layer(const LayerParameter& param) : layer_param_(param) {
    phase_ = param.has_phase() ? param.phase() : TEST;
  }
  virtual ~Layer() {}

  void SetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    CheckBlobCounts(bottom, top);
    LayerSetUp(bottom, top);
    Reshape(bottom, top);
  }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;

  inline Dtype Forward(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    Reshape(bottom, top);
    switch (Caffe::mode()) {
      case Caffe::CPU: Forward_cpu(bottom, top); break;
      case Caffe::GPU: Forward_gpu(bottom, top); break;
      default: LOG(FATAL) << "Unknown caffe mode.";
    }
    return Dtype(0);
  }

  vector<shared_ptr<Blob<Dtype> > >& blobs() { return blobs_; }
  const LayerParameter& layer_param() const { return layer_param_; }
  virtual inline const char* type() const { return ""; }
  virtual inline int ExactNumBottomBlobs() const { return -1; }
  virtual inline int MinBottomBlobs() const { return -1; }
  virtual inline int MaxBottomBlobs() const { return -1; }
  virtual inline int ExactNumTopBlobs() const { return -1; }
  virtual inline int MinTopBlobs() const { return -1; }
  virtual inline int MaxTopBlobs() const { return -1; }
  // Called by Net when the parameter blobs were (re)written on the host (weight injection,
  // CopyTrainedLayersFrom): layers that keep a device-side re-packed copy refresh it here.
  virtual void OnWeightsChanged() {}

  // Net-level operator fusion hooks (a fused layer must produce exactly what the pair produced).
  virtual bool FuseReLU(Dtype negative_slope) { return false; }
  // A producer that can also emit the output of the MAX 2x2 / stride 2 / pad 0 Pooling layer that consumes its top writes
  // it into `pooled_top` (already shaped by that Pooling layer) from then on and returns true.
  virtual bool FusePool2x2(Blob<Dtype>* pooled_top) { return false; }
  // Pooling layers only: true when the layer is exactly MAX, kernel 2, stride 2, pad 0.
  virtual bool IsMaxPool2x2() const { return false; }
  // A producer that can write its top straight into channels [c_offset, c_offset + C) of a wider blob (the top of the
  // Concat that would otherwise copy it) returns true and does so from then on; its own top blob is then not written.
  virtual bool SetOutputWindow(Blob<Dtype>* target, int c_total, int c_offset) { return false; }
  // Algorithmic FLOPs of the last Forward (0 for bandwidth layers), for roofline accounting.
  virtual double ForwardFlops() const { return 0; }

 protected:
  LayerParameter layer_param_;
  Phase phase_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;

  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) = 0;
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { Forward_cpu(bottom, top); }

  virtual void CheckBlobCounts(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    if (ExactNumBottomBlobs() >= 0) CHECK_EQ(ExactNumBottomBlobs(), (int)bottom.size()) << type() << " Layer takes " << ExactNumBottomBlobs() << " bottom blob(s) as input.";
    if (MinBottomBlobs() >= 0) CHECK_LE(MinBottomBlobs(), (int)bottom.size()) << type() << " Layer takes at least " << MinBottomBlobs() << " bottom blob(s) as input.";
    if (MaxBottomBlobs() >= 0) CHECK_GE(MaxBottomBlobs(), (int)bottom.size()) << type() << " Layer takes at most " << MaxBottomBlobs() << " bottom blob(s) as input.";
    if (ExactNumTopBlobs() >= 0) CHECK_EQ(ExactNumTopBlobs(), (int)top.size()) << type() << " Layer produces " << ExactNumTopBlobs() << " top blob(s) as output.";
    if (MinTopBlobs() >= 0) CHECK_LE(MinTopBlobs(), (int)top.size()) << type() << " Layer produces at least " << MinTopBlobs() << " top blob(s) as output.";
    if (MaxTopBlobs() >= 0) CHECK_GE(MaxTopBlobs(), (int)top.size()) << type() << " Layer produces at most " << MaxTopBlobs() << " top blob(s) as output.";
  }
  DISABLE_COPY_AND_ASSIGN(Layer);
};

// The product has no CPU compute path (the oracle under oracle/ is test infrastructure only).
#define MSCNN_NO_CPU_PATH(name)                                                                                    \
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {                   \
    LOG(FATAL) << name "Layer::Forward_cpu: this build is GPU (MI355X) only; Caffe::set_mode(Caffe::GPU) is required"; \
  }

}  // namespace caffe
#endif
