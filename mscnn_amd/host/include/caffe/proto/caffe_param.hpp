// Hand-written stand-in for the protobuf-generated caffe.pb.h (no protoc/libprotobuf in the image).
// Covers the schema subset the MS-CNN deploy nets use (src/caffe/proto/caffe.proto: NetParameter :65-102,
// LayerParameter :311-414, ConvolutionParameter, PoolingParameter, InnerProductParameter, ROIPoolingParameter
// :1257-1266, BoxOutputParameter :1315-1329, BBoxRegParameter :1344-1348, DecodeBBoxParameter :1351-1353 ...)
// with the accessor names the reference's layers call, and the field defaults of caffe.proto.
//
// Storage is a generic ordered field tree (TextMessage) filled by a protobuf *text-format* parser
// (comments, `name { }` and `name: { }` nesting, several fields per line, repeated scalars, quoted
// strings, bare enums, bools, negative/decimal numbers).
#ifndef MSCNN_CAFFE_PROTO_PARAM_HPP_
#define MSCNN_CAFFE_PROTO_PARAM_HPP_

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "caffe/util/logging.hpp"

namespace caffe {

class TextMessage;
typedef std::shared_ptr<TextMessage> TextMessagePtr;

struct TextField {
  std::string name;
  std::string scalar;      // valid when !msg
  TextMessagePtr msg;      // nested message
};

class TextMessage {
 public:
  std::vector<TextField> fields;

  int count(const std::string& name) const;
  bool has(const std::string& name) const { return count(name) > 0; }
  const std::string& str(const std::string& name, int i = 0) const;
  const TextMessage& sub(const std::string& name, int i = 0) const;   // empty message if absent
  TextMessage* mutable_sub(const std::string& name, int i = 0);
  double num(const std::string& name, int i, double dflt) const;
  bool boolean(const std::string& name, bool dflt) const;
  std::string str_or(const std::string& name, const std::string& dflt) const { return has(name) ? str(name) : dflt; }
  void add_scalar(const std::string& name, const std::string& v);
  TextMessage* add_message(const std::string& name);
  void clear(const std::string& name);
  void set_scalar(const std::string& name, int i, const std::string& v);
  std::string DebugString(int indent = 0) const;
  static const TextMessage& Empty();
};

// Parses protobuf text format; throws FatalError with line number on a syntax error.
TextMessagePtr ParseTextFormat(const std::string& text, const std::string& origin = "<string>");
bool ReadFileToString(const std::string& path, std::string* out);

enum Phase { TRAIN = 0, TEST = 1 };

#define MSCNN_PARAM_CLASS(Name)                      \
 public:                                             \
  Name() : m_(&TextMessage::Empty()) {}              \
  explicit Name(const TextMessage& m) : m_(&m) {}    \
  const TextMessage& raw() const { return *m_; }     \
 private:                                            \
  const TextMessage* m_;                             \
 public:

class FillerParameter {
  MSCNN_PARAM_CLASS(FillerParameter)
  std::string type() const { return m_->str_or("type", "constant"); }
  float value() const { return (float)m_->num("value", 0, 0); }
  float std() const { return (float)m_->num("std", 0, 1); }
  float mean() const { return (float)m_->num("mean", 0, 0); }
};

class ConvolutionParameter {
  MSCNN_PARAM_CLASS(ConvolutionParameter)
  unsigned num_output() const { return (unsigned)m_->num("num_output", 0, 0); }
  bool bias_term() const { return m_->boolean("bias_term", true); }
  int pad_size() const { return m_->count("pad"); }
  unsigned pad(int i) const { return (unsigned)m_->num("pad", i, 0); }
  int kernel_size_size() const { return m_->count("kernel_size"); }
  unsigned kernel_size(int i) const { return (unsigned)m_->num("kernel_size", i, 0); }
  int stride_size() const { return m_->count("stride"); }
  unsigned stride(int i) const { return (unsigned)m_->num("stride", i, 1); }
  int dilation_size() const { return m_->count("dilation"); }
  unsigned dilation(int i) const { return (unsigned)m_->num("dilation", i, 1); }
  bool has_pad_h() const { return m_->has("pad_h"); }
  bool has_pad_w() const { return m_->has("pad_w"); }
  unsigned pad_h() const { return (unsigned)m_->num("pad_h", 0, 0); }
  unsigned pad_w() const { return (unsigned)m_->num("pad_w", 0, 0); }
  bool has_kernel_h() const { return m_->has("kernel_h"); }
  bool has_kernel_w() const { return m_->has("kernel_w"); }
  unsigned kernel_h() const { return (unsigned)m_->num("kernel_h", 0, 0); }
  unsigned kernel_w() const { return (unsigned)m_->num("kernel_w", 0, 0); }
  bool has_stride_h() const { return m_->has("stride_h"); }
  bool has_stride_w() const { return m_->has("stride_w"); }
  unsigned stride_h() const { return (unsigned)m_->num("stride_h", 0, 1); }
  unsigned stride_w() const { return (unsigned)m_->num("stride_w", 0, 1); }
  unsigned group() const { return (unsigned)m_->num("group", 0, 1); }
  int axis() const { return (int)m_->num("axis", 0, 1); }
  FillerParameter weight_filler() const { return FillerParameter(m_->sub("weight_filler")); }
  FillerParameter bias_filler() const { return FillerParameter(m_->sub("bias_filler")); }
};

enum PoolingParameter_PoolMethod {
  PoolingParameter_PoolMethod_MAX = 0,
  PoolingParameter_PoolMethod_AVE = 1,
  PoolingParameter_PoolMethod_STOCHASTIC = 2
};

class PoolingParameter {
  MSCNN_PARAM_CLASS(PoolingParameter)
  PoolingParameter_PoolMethod pool() const {
    const std::string p = m_->str_or("pool", "MAX");
    if (p == "MAX" || p == "0") return PoolingParameter_PoolMethod_MAX;
    if (p == "AVE" || p == "1") return PoolingParameter_PoolMethod_AVE;
    if (p == "STOCHASTIC" || p == "2") return PoolingParameter_PoolMethod_STOCHASTIC;
    LOG(FATAL) << "Unknown pooling method " << p;
  }
  bool has_kernel_size() const { return m_->has("kernel_size"); }
  unsigned kernel_size() const { return (unsigned)m_->num("kernel_size", 0, 0); }
  bool has_kernel_h() const { return m_->has("kernel_h"); }
  bool has_kernel_w() const { return m_->has("kernel_w"); }
  unsigned kernel_h() const { return (unsigned)m_->num("kernel_h", 0, 0); }
  unsigned kernel_w() const { return (unsigned)m_->num("kernel_w", 0, 0); }
  bool has_pad_h() const { return m_->has("pad_h"); }
  bool has_pad_w() const { return m_->has("pad_w"); }
  unsigned pad() const { return (unsigned)m_->num("pad", 0, 0); }
  unsigned pad_h() const { return (unsigned)m_->num("pad_h", 0, 0); }
  unsigned pad_w() const { return (unsigned)m_->num("pad_w", 0, 0); }
  bool has_stride_h() const { return m_->has("stride_h"); }
  bool has_stride_w() const { return m_->has("stride_w"); }
  unsigned stride() const { return (unsigned)m_->num("stride", 0, 1); }
  unsigned stride_h() const { return (unsigned)m_->num("stride_h", 0, 1); }
  unsigned stride_w() const { return (unsigned)m_->num("stride_w", 0, 1); }
  bool global_pooling() const { return m_->boolean("global_pooling", false); }
};

class InnerProductParameter {
  MSCNN_PARAM_CLASS(InnerProductParameter)
  unsigned num_output() const { return (unsigned)m_->num("num_output", 0, 0); }
  bool bias_term() const { return m_->boolean("bias_term", true); }
  int axis() const { return (int)m_->num("axis", 0, 1); }
  bool transpose() const { return m_->boolean("transpose", false); }
  FillerParameter weight_filler() const { return FillerParameter(m_->sub("weight_filler")); }
  FillerParameter bias_filler() const { return FillerParameter(m_->sub("bias_filler")); }
};

class ReLUParameter {
  MSCNN_PARAM_CLASS(ReLUParameter)
  float negative_slope() const { return (float)m_->num("negative_slope", 0, 0); }
};

class DropoutParameter {
  MSCNN_PARAM_CLASS(DropoutParameter)
  float dropout_ratio() const { return (float)m_->num("dropout_ratio", 0, 0.5); }
};

class ConcatParameter {
  MSCNN_PARAM_CLASS(ConcatParameter)
  int axis() const { return (int)m_->num("axis", 0, 1); }
  bool has_concat_dim() const { return m_->has("concat_dim"); }
  unsigned concat_dim() const { return (unsigned)m_->num("concat_dim", 0, 1); }
};

class SoftmaxParameter {
  MSCNN_PARAM_CLASS(SoftmaxParameter)
  int axis() const { return (int)m_->num("axis", 0, 1); }
};

enum EltwiseParameter_EltwiseOp { EltwiseParameter_EltwiseOp_PROD = 0, EltwiseParameter_EltwiseOp_SUM = 1, EltwiseParameter_EltwiseOp_MAX = 2 };

class EltwiseParameter {
  MSCNN_PARAM_CLASS(EltwiseParameter)
  EltwiseParameter_EltwiseOp operation() const {
    const std::string o = m_->str_or("operation", "SUM");
    if (o == "PROD" || o == "0") return EltwiseParameter_EltwiseOp_PROD;
    if (o == "SUM" || o == "1") return EltwiseParameter_EltwiseOp_SUM;
    if (o == "MAX" || o == "2") return EltwiseParameter_EltwiseOp_MAX;
    LOG(FATAL) << "Unknown elementwise operation " << o;
  }
  int coeff_size() const { return m_->count("coeff"); }
  float coeff(int i) const { return (float)m_->num("coeff", i, 1); }
};

class ROIPoolingParameter {   // caffe.proto:1257-1266
  MSCNN_PARAM_CLASS(ROIPoolingParameter)
  unsigned pooled_h() const { return (unsigned)m_->num("pooled_h", 0, 0); }
  unsigned pooled_w() const { return (unsigned)m_->num("pooled_w", 0, 0); }
  float spatial_scale() const { return (float)m_->num("spatial_scale", 0, 1); }
  float pad_ratio() const { return (float)m_->num("pad_ratio", 0, 0); }
};

class BoxOutputParameter {    // caffe.proto:1315-1329
  MSCNN_PARAM_CLASS(BoxOutputParameter)
  float fg_thr() const { return (float)m_->num("fg_thr", 0, 0); }
  float iou_thr() const { return (float)m_->num("iou_thr", 0, 0.5); }
  std::string nms_type() const { return m_->str_or("nms_type", "IOU"); }
  int field_h_size() const { return m_->count("field_h"); }
  int field_w_size() const { return m_->count("field_w"); }
  int downsample_rate_size() const { return m_->count("downsample_rate"); }
  unsigned field_h(int i) const { return (unsigned)m_->num("field_h", i, 0); }
  unsigned field_w(int i) const { return (unsigned)m_->num("field_w", i, 0); }
  unsigned downsample_rate(int i) const { return (unsigned)m_->num("downsample_rate", i, 0); }
  float field_whr() const { return (float)m_->num("field_whr", 0, 2); }
  float field_xyr() const { return (float)m_->num("field_xyr", 0, 2); }
  unsigned max_nms_num() const { return (unsigned)m_->num("max_nms_num", 0, 0); }
  unsigned max_post_nms_num() const { return (unsigned)m_->num("max_post_nms_num", 0, 0); }
  float min_size() const { return (float)m_->num("min_size", 0, 15); }
};

class BBoxRegParameter {      // caffe.proto:1344-1348
  MSCNN_PARAM_CLASS(BBoxRegParameter)
  int bbox_mean_size() const { return m_->count("bbox_mean"); }
  int bbox_std_size() const { return m_->count("bbox_std"); }
  float bbox_mean(int i) const { return (float)m_->num("bbox_mean", i, 0); }
  float bbox_std(int i) const { return (float)m_->num("bbox_std", i, 1); }
  bool cls_aware() const { return m_->boolean("cls_aware", true); }
};

class DecodeBBoxParameter {
  MSCNN_PARAM_CLASS(DecodeBBoxParameter)
  float gt_iou_thr() const { return (float)m_->num("gt_iou_thr", 0, 0.95); }
};

class BlobShape {
  MSCNN_PARAM_CLASS(BlobShape)
  int dim_size() const { return m_->count("dim"); }
  long dim(int i) const { return (long)m_->num("dim", i, 0); }
};

class InputParameter {
  MSCNN_PARAM_CLASS(InputParameter)
  int shape_size() const { return m_->count("shape"); }
  BlobShape shape(int i) const { return BlobShape(m_->sub("shape", i)); }
};

class NetStateRule {
  MSCNN_PARAM_CLASS(NetStateRule)
  bool has_phase() const { return m_->has("phase"); }
  Phase phase() const { return m_->str("phase") == "TRAIN" || m_->str("phase") == "0" ? TRAIN : TEST; }
};

// LayerParameter owns (shares) its message so that Net may rewrite bottoms/tops (split insertion).
class LayerParameter {
 public:
  LayerParameter() : m_(std::make_shared<TextMessage>()) {}
  explicit LayerParameter(TextMessagePtr m) : m_(std::move(m)) {}
  LayerParameter Clone() const { return LayerParameter(std::make_shared<TextMessage>(*m_)); }
  const TextMessage& raw() const { return *m_; }
  TextMessage* mutable_raw() { return m_.get(); }

  std::string name() const { return m_->str_or("name", ""); }
  std::string type() const { return m_->str_or("type", ""); }
  void set_name(const std::string& v) { m_->clear("name"); m_->add_scalar("name", v); }
  void set_type(const std::string& v) { m_->clear("type"); m_->add_scalar("type", v); }
  int bottom_size() const { return m_->count("bottom"); }
  int top_size() const { return m_->count("top"); }
  const std::string& bottom(int i) const { return m_->str("bottom", i); }
  const std::string& top(int i) const { return m_->str("top", i); }
  void set_bottom(int i, const std::string& v) { m_->set_scalar("bottom", i, v); }
  void add_bottom(const std::string& v) { m_->add_scalar("bottom", v); }
  void add_top(const std::string& v) { m_->add_scalar("top", v); }
  bool has_phase() const { return m_->has("phase"); }
  Phase phase() const { return m_->str_or("phase", "TEST") == "TRAIN" ? TRAIN : TEST; }
  void set_phase(Phase p) { m_->clear("phase"); m_->add_scalar("phase", p == TRAIN ? "TRAIN" : "TEST"); }
  int loss_weight_size() const { return m_->count("loss_weight"); }
  float loss_weight(int i) const { return (float)m_->num("loss_weight", i, 0); }
  int include_size() const { return m_->count("include"); }
  int exclude_size() const { return m_->count("exclude"); }
  NetStateRule include(int i) const { return NetStateRule(m_->sub("include", i)); }
  NetStateRule exclude(int i) const { return NetStateRule(m_->sub("exclude", i)); }

  ConvolutionParameter convolution_param() const { return ConvolutionParameter(m_->sub("convolution_param")); }
  PoolingParameter pooling_param() const { return PoolingParameter(m_->sub("pooling_param")); }
  InnerProductParameter inner_product_param() const { return InnerProductParameter(m_->sub("inner_product_param")); }
  ReLUParameter relu_param() const { return ReLUParameter(m_->sub("relu_param")); }
  DropoutParameter dropout_param() const { return DropoutParameter(m_->sub("dropout_param")); }
  ConcatParameter concat_param() const { return ConcatParameter(m_->sub("concat_param")); }
  SoftmaxParameter softmax_param() const { return SoftmaxParameter(m_->sub("softmax_param")); }
  EltwiseParameter eltwise_param() const { return EltwiseParameter(m_->sub("eltwise_param")); }
  ROIPoolingParameter roi_pooling_param() const { return ROIPoolingParameter(m_->sub("roi_pooling_param")); }
  BoxOutputParameter box_output_param() const { return BoxOutputParameter(m_->sub("box_output_param")); }
  BBoxRegParameter bbox_reg_param() const { return BBoxRegParameter(m_->sub("bbox_reg_param")); }
  DecodeBBoxParameter decode_bbox_param() const { return DecodeBBoxParameter(m_->sub("decode_bbox_param")); }
  InputParameter input_param() const { return InputParameter(m_->sub("input_param")); }

 private:
  TextMessagePtr m_;
};

// NetParameter: name + layers (+ legacy input/input_dim/input_shape, upgraded by Net::Init like
// UpgradeNetInput, src/caffe/util/upgrade_proto.cpp:966-1003).
class NetParameter {
 public:
  NetParameter() : m_(std::make_shared<TextMessage>()) {}
  explicit NetParameter(TextMessagePtr m) : m_(std::move(m)) {}
  const TextMessage& raw() const { return *m_; }
  std::string name() const { return m_->str_or("name", ""); }
  int layer_size() const { return m_->count("layer"); }
  LayerParameter layer(int i) const;
  int input_size() const { return m_->count("input"); }
  const std::string& input(int i) const { return m_->str("input", i); }
  int input_dim_size() const { return m_->count("input_dim"); }
  int input_dim(int i) const { return (int)m_->num("input_dim", i, 0); }
  int input_shape_size() const { return m_->count("input_shape"); }
  BlobShape input_shape(int i) const { return BlobShape(m_->sub("input_shape", i)); }
  bool has_legacy_layers() const { return m_->has("layers"); }

 private:
  TextMessagePtr m_;
};

bool ReadProtoFromTextFile(const std::string& filename, NetParameter* param);
void ReadNetParamsFromTextFileOrDie(const std::string& filename, NetParameter* param);
NetParameter NetParameterFromString(const std::string& text);

}  // namespace caffe
#endif
