// Compile-only boundary check (tests/test_cabi.py): code a maintainer of the reference would write against ITS headers must
// compile unchanged against mscnn_amd/host/include -- (1) the INTEGRATION.md section 1 binding of one reference layer to the C
// ABI, (2) a user-defined layer registered with REGISTER_LAYER_CLASS (include/caffe/layer_factory.hpp:116-137), (3) the
// section 2 driver.  Mirrors the shape of src/caffe/layers/roi_pooling_layer.cu:92-104.
#include <vector>

#include "caffe/caffe.hpp"
#include "mscnn_hip.h"

namespace caffe {

// (1) a reference-style layer whose Forward_gpu calls the C ABI where the reference launches its CUDA kernel
template <typename Dtype>
class UserROIPoolingLayer : public Layer<Dtype> {
 public:
  explicit UserROIPoolingLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    ROIPoolingParameter p = this->layer_param_.roi_pooling_param();
    CHECK_GT(p.pooled_h(), 0u) << "pooled_h must be > 0";
    pooled_height_ = p.pooled_h(); pooled_width_ = p.pooled_w();
    spatial_scale_ = p.spatial_scale(); pad_ratio_ = p.pad_ratio();
  }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    channels_ = bottom[0]->channels(); height_ = bottom[0]->height(); width_ = bottom[0]->width();
    top[0]->Reshape(bottom[1]->num(), channels_, pooled_height_, pooled_width_);
  }
  virtual inline const char* type() const { return "UserROIPooling"; }
  virtual inline int MinBottomBlobs() const { return 2; }
  virtual inline int MaxBottomBlobs() const { return 2; }
  virtual inline int ExactNumTopBlobs() const { return 1; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { NOT_IMPLEMENTED; }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);
  int channels_, height_, width_, pooled_height_, pooled_width_;
  Dtype spatial_scale_, pad_ratio_;
};

template <>
void UserROIPoolingLayer<float>::Forward_gpu(const vector<Blob<float>*>& bottom, const vector<Blob<float>*>& top) {
  const int rc = mscnn_roipool_fwd_f32(bottom[0]->gpu_data(), bottom[1]->gpu_data(), top[0]->mutable_gpu_data(),
                                       bottom[1]->num(), bottom[0]->num(), channels_, height_, width_, pooled_height_,
                                       pooled_width_, spatial_scale_, pad_ratio_, /*C_total=*/channels_, /*c_offset=*/0,
                                       /*stream=*/nullptr);
  CHECK_EQ(rc, 0) << mscnn_last_error();            // glog fatal, the reference's error convention
}

// (2) a user layer written purely against the Layer interface
template <typename Dtype>
class ScaleByTwoLayer : public Layer<Dtype> {
 public:
  explicit ScaleByTwoLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { top[0]->ReshapeLike(*bottom[0]); }
  virtual inline const char* type() const { return "ScaleByTwo"; }
  virtual inline int ExactNumBottomBlobs() const { return 1; }
  virtual inline int ExactNumTopBlobs() const { return 1; }

 protected:
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) {
    const Dtype* x = bottom[0]->cpu_data();
    Dtype* y = top[0]->mutable_cpu_data();
    for (int i = 0; i < bottom[0]->count(); ++i) y[i] = Dtype(2) * x[i];
  }
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top) { Forward_cpu(bottom, top); }
};

INSTANTIATE_CLASS(UserROIPoolingLayer);
REGISTER_LAYER_CLASS(UserROIPooling);
INSTANTIATE_CLASS(ScaleByTwoLayer);
REGISTER_LAYER_CLASS(ScaleByTwo);

}  // namespace caffe

// (3) the INTEGRATION.md section 2 driver
int run_reference_style_driver(const char* prototxt, const char* caffemodel) {
  using namespace caffe;
  Caffe::set_mode(Caffe::GPU);
  Caffe::SetDevice(0);
  Net<float> net(prototxt, TEST);
  if (caffemodel) net.CopyTrainedLayersFrom(caffemodel);
  float* in = net.input_blobs()[0]->mutable_cpu_data();
  in[0] = 0.f;
  const std::vector<Blob<float>*>& out = net.Forward();
  const shared_ptr<Blob<float> > rois = net.blob_by_name("proposals");
  return (int)out.size() + (rois ? rois->num() : 0) + (net.has_layer("conv1_1") ? 1 : 0);
}
