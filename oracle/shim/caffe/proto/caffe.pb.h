// oracle/shim: hand-written stand-in for the protoc output of src/caffe/proto/caffe.proto, limited to the messages
// the reference's hot-path sources touch, with protobuf's accessor names and caffe.proto's defaults.
// TEST INFRASTRUCTURE (lets the reference's own .cpp files compile here); never part of the product.
#pragma once
#include <string>
#include <vector>

namespace caffe {

#define PB_OPT(type, name, dflt)                                             \
 private:                                                                    \
  type name##_ = dflt; bool has_##name##_ = false;                           \
 public:                                                                     \
  const type& name() const { return name##_; }                               \
  void set_##name(const type& v) { name##_ = v; has_##name##_ = true; }      \
  bool has_##name() const { return has_##name##_; }                          \
  void clear_##name() { name##_ = dflt; has_##name##_ = false; }

#define PB_REP(type, name)                                                   \
 private:                                                                    \
  std::vector<type> name##_;                                                 \
 public:                                                                     \
  int name##_size() const { return (int)name##_.size(); }                    \
  const type& name(int i) const { return name##_[i]; }                       \
  const std::vector<type>& name() const { return name##_; }                  \
  void add_##name(const type& v) { name##_.push_back(v); }                   \
  type* add_##name() { name##_.emplace_back(); return &name##_.back(); }     \
  type* mutable_##name(int i) { return &name##_[i]; }                        \
  void clear_##name() { name##_.clear(); }

#define PB_MSG(Type, name)                                                   \
 private:                                                                    \
  Type name##_; bool has_##name##_ = false;                                  \
 public:                                                                     \
  const Type& name() const { return name##_; }                               \
  Type* mutable_##name() { has_##name##_ = true; return &name##_; }          \
  bool has_##name() const { return has_##name##_; }                          \
  void clear_##name() { name##_ = Type(); has_##name##_ = false; }

enum Phase { TRAIN = 0, TEST = 1 };

class BlobShape {
  PB_REP(long long, dim)
};

class BlobProto {
  PB_MSG(BlobShape, shape)
  PB_REP(float, data)
  PB_REP(float, diff)
  PB_REP(double, double_data)
  PB_REP(double, double_diff)
  PB_OPT(int, num, 0)
  PB_OPT(int, channels, 0)
  PB_OPT(int, height, 0)
  PB_OPT(int, width, 0)
 public:
  void Clear() { *this = BlobProto(); }
};

enum FillerParameter_VarianceNorm {
  FillerParameter_VarianceNorm_FAN_IN = 0, FillerParameter_VarianceNorm_FAN_OUT = 1, FillerParameter_VarianceNorm_AVERAGE = 2
};

class FillerParameter {
  PB_OPT(std::string, type, "constant")
  PB_OPT(float, value, 0.f)
  PB_OPT(float, min, 0.f)
  PB_OPT(float, max, 1.f)
  PB_OPT(float, mean, 0.f)
  PB_OPT(float, std, 1.f)
  PB_OPT(int, sparse, -1)
  PB_OPT(FillerParameter_VarianceNorm, variance_norm, FillerParameter_VarianceNorm_FAN_IN)
};

class ConvolutionParameter {
  PB_OPT(unsigned, num_output, 0)
  PB_OPT(bool, bias_term, true)
  PB_REP(unsigned, pad)
  PB_REP(unsigned, kernel_size)
  PB_REP(unsigned, stride)
  PB_REP(unsigned, dilation)
  PB_OPT(unsigned, pad_h, 0)
  PB_OPT(unsigned, pad_w, 0)
  PB_OPT(unsigned, kernel_h, 0)
  PB_OPT(unsigned, kernel_w, 0)
  PB_OPT(unsigned, stride_h, 0)
  PB_OPT(unsigned, stride_w, 0)
  PB_OPT(unsigned, group, 1)
  PB_MSG(FillerParameter, weight_filler)
  PB_MSG(FillerParameter, bias_filler)
  PB_OPT(int, axis, 1)
  PB_OPT(bool, force_nd_im2col, false)
};

enum PoolingParameter_PoolMethod { PoolingParameter_PoolMethod_MAX = 0, PoolingParameter_PoolMethod_AVE = 1, PoolingParameter_PoolMethod_STOCHASTIC = 2 };

class PoolingParameter {
  PB_OPT(PoolingParameter_PoolMethod, pool, PoolingParameter_PoolMethod_MAX)
  PB_OPT(unsigned, pad, 0)
  PB_OPT(unsigned, pad_h, 0)
  PB_OPT(unsigned, pad_w, 0)
  PB_OPT(unsigned, kernel_size, 0)
  PB_OPT(unsigned, kernel_h, 0)
  PB_OPT(unsigned, kernel_w, 0)
  PB_OPT(unsigned, stride, 1)
  PB_OPT(unsigned, stride_h, 0)
  PB_OPT(unsigned, stride_w, 0)
  PB_OPT(bool, global_pooling, false)
};

class InnerProductParameter {
  PB_OPT(unsigned, num_output, 0)
  PB_OPT(bool, bias_term, true)
  PB_MSG(FillerParameter, weight_filler)
  PB_MSG(FillerParameter, bias_filler)
  PB_OPT(int, axis, 1)
  PB_OPT(bool, transpose, false)
};

class ReLUParameter { PB_OPT(float, negative_slope, 0.f) };
class SoftmaxParameter { PB_OPT(int, axis, 1) };
class ConcatParameter {
  PB_OPT(int, axis, 1)
  PB_OPT(unsigned, concat_dim, 1)
};
class DropoutParameter { PB_OPT(float, dropout_ratio, 0.5f) };
enum EltwiseParameter_EltwiseOp { EltwiseParameter_EltwiseOp_PROD = 0, EltwiseParameter_EltwiseOp_SUM = 1, EltwiseParameter_EltwiseOp_MAX = 2 };
class EltwiseParameter {
  PB_OPT(EltwiseParameter_EltwiseOp, operation, EltwiseParameter_EltwiseOp_SUM)
  PB_REP(float, coeff)
  PB_OPT(bool, stable_prod_grad, true)
};

class ROIPoolingParameter {      // caffe.proto:1257-1266
  PB_OPT(unsigned, pooled_h, 0)
  PB_OPT(unsigned, pooled_w, 0)
  PB_OPT(float, spatial_scale, 1.f)
  PB_OPT(float, pad_ratio, 0.f)
};

class BoxOutputParameter {       // caffe.proto:1315-1329
  PB_OPT(float, fg_thr, 0.f)
  PB_OPT(float, iou_thr, 0.5f)
  PB_OPT(std::string, nms_type, "IOU")
  PB_REP(unsigned, field_h)
  PB_REP(unsigned, field_w)
  PB_REP(unsigned, downsample_rate)
  PB_OPT(float, field_whr, 2.f)
  PB_OPT(float, field_xyr, 2.f)
  PB_OPT(unsigned, max_nms_num, 0)
  PB_OPT(unsigned, max_post_nms_num, 0)
  PB_OPT(float, min_size, 15.f)
};

class BBoxRegParameter {         // caffe.proto:1344-1348
  PB_REP(float, bbox_mean)
  PB_REP(float, bbox_std)
  PB_OPT(bool, cls_aware, true)
};

class DecodeBBoxParameter { PB_OPT(float, gt_iou_thr, 0.95f) };

class ParamSpec {
  PB_OPT(std::string, name, "")
  PB_OPT(float, lr_mult, 1.f)
  PB_OPT(float, decay_mult, 1.f)
};

class LayerParameter {
  PB_OPT(std::string, name, "")
  PB_OPT(std::string, type, "")
  PB_REP(std::string, bottom)
  PB_REP(std::string, top)
  PB_OPT(Phase, phase, TEST)
  PB_REP(float, loss_weight)
  PB_REP(ParamSpec, param)
  PB_REP(BlobProto, blobs)
  PB_REP(char, propagate_down)   // vector<bool> has no addressable elements
  PB_MSG(ConvolutionParameter, convolution_param)
  PB_MSG(PoolingParameter, pooling_param)
  PB_MSG(InnerProductParameter, inner_product_param)
  PB_MSG(ReLUParameter, relu_param)
  PB_MSG(SoftmaxParameter, softmax_param)
  PB_MSG(ConcatParameter, concat_param)
  PB_MSG(DropoutParameter, dropout_param)
  PB_MSG(EltwiseParameter, eltwise_param)
  PB_MSG(ROIPoolingParameter, roi_pooling_param)
  PB_MSG(BoxOutputParameter, box_output_param)
  PB_MSG(BBoxRegParameter, bbox_reg_param)
  PB_MSG(DecodeBBoxParameter, decode_bbox_param)
 public:
  void Clear() { *this = LayerParameter(); }
  void CopyFrom(const LayerParameter& o) { *this = o; }
};

}  // namespace caffe
