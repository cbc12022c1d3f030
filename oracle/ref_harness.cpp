// oracle/ref_harness.cpp -- C entry points that drive the REFERENCE's own layer classes (compiled from
// /root/reference/src/caffe/... by oracle/ref.mk) on caller-provided arrays.  TEST INFRASTRUCTURE ONLY: this is how the
// restatement in mscnn_oracle.c and the golden fixtures under tests/golden/ are pinned to the real reference code.
#include <cstring>
#include <string>
#include <vector>

#include "caffe/blob.hpp"
#include "caffe/filler.hpp"
#include "caffe/layers/box_output_layer.hpp"
#include "caffe/layers/concat_layer.hpp"
#include "caffe/layers/conv_layer.hpp"
#include "caffe/layers/decode_bbox_layer.hpp"
#include "caffe/layers/deconv_layer.hpp"
#include "caffe/layers/inner_product_layer.hpp"
#include "caffe/layers/pooling_layer.hpp"
#include "caffe/layers/relu_layer.hpp"
#include "caffe/layers/eltwise_layer.hpp"
#include "caffe/layers/roi_align_layer.hpp"
#include "caffe/layers/roi_pooling_layer.hpp"
#include "caffe/layers/softmax_layer.hpp"
#include "caffe/util/math_functions.hpp"

using namespace caffe;  // NOLINT
typedef std::vector<Blob<float>*> BV;

static thread_local std::string g_err;
#define REF_API extern "C" __attribute__((visibility("default")))
#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; }

REF_API const char* ref_last_error() { return g_err.c_str(); }

static void fill(Blob<float>* b, const float* src) { std::memcpy(b->mutable_cpu_data(), src, sizeof(float) * b->count()); }
static void take(const Blob<float>& b, float* dst) { std::memcpy(dst, b.cpu_data(), sizeof(float) * b.count()); }

static void conv_param(LayerParameter* lp, int Cout, int Kh, int Kw, int ph, int pw, int sh, int sw, int group, bool bias) {
  ConvolutionParameter* c = lp->mutable_convolution_param();
  c->set_num_output(Cout); c->set_kernel_h(Kh); c->set_kernel_w(Kw); c->set_pad_h(ph); c->set_pad_w(pw);
  c->set_stride_h(sh); c->set_stride_w(sw); c->set_group(group); c->set_bias_term(bias);
}

REF_API int ref_conv2d(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W, int Cout, int Kh,
                       int Kw, int ph, int pw, int sh, int sw, int group) {
  GUARD({
    LayerParameter lp; conv_param(&lp, Cout, Kh, Kw, ph, pw, sh, sw, group, b != nullptr);
    ConvolutionLayer<float> layer(lp);
    Blob<float> bottom(N, Cin, H, W), top;
    fill(&bottom, x);
    BV bv(1, &bottom), tv(1, &top);
    layer.SetUp(bv, tv);
    fill(layer.blobs()[0].get(), w);
    if (b) fill(layer.blobs()[1].get(), b);
    layer.Forward(bv, tv);
    take(top, y);
  })
}

REF_API int ref_deconv2d(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W, int Cout, int Kh,
                         int Kw, int ph, int pw, int sh, int sw, int group, int bilinear_filler) {
  GUARD({
    LayerParameter lp; conv_param(&lp, Cout, Kh, Kw, ph, pw, sh, sw, group, b != nullptr);
    if (bilinear_filler) lp.mutable_convolution_param()->mutable_weight_filler()->set_type("bilinear");
    DeconvolutionLayer<float> layer(lp);
    Blob<float> bottom(N, Cin, H, W), top;
    fill(&bottom, x);
    BV bv(1, &bottom), tv(1, &top);
    layer.SetUp(bv, tv);
    if (w) fill(layer.blobs()[0].get(), w);
    if (b) fill(layer.blobs()[1].get(), b);
    layer.Forward(bv, tv);
    take(top, y);
  })
}

REF_API int ref_pool2d(const float* x, float* y, int N, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw,
                       int method, int* out_h, int* out_w) {
  GUARD({
    LayerParameter lp;
    PoolingParameter* p = lp.mutable_pooling_param();
    p->set_kernel_h(kh); p->set_kernel_w(kw); p->set_pad_h(ph); p->set_pad_w(pw); p->set_stride_h(sh); p->set_stride_w(sw);
    p->set_pool(method == 0 ? PoolingParameter_PoolMethod_MAX : PoolingParameter_PoolMethod_AVE);
    PoolingLayer<float> layer(lp);
    Blob<float> bottom(N, C, H, W), top;
    fill(&bottom, x);
    BV bv(1, &bottom), tv(1, &top);
    layer.SetUp(bv, tv);
    if (out_h) *out_h = top.height();
    if (out_w) *out_w = top.width();
    if (y) { layer.Forward(bv, tv); take(top, y); }
  })
}

REF_API int ref_relu(const float* x, float* y, int n, float slope) {
  GUARD({
    LayerParameter lp; lp.mutable_relu_param()->set_negative_slope(slope);
    ReLULayer<float> layer(lp);
    Blob<float> bottom(1, 1, 1, n), top;
    fill(&bottom, x);
    BV bv(1, &bottom), tv(1, &top);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    take(top, y);
  })
}

REF_API int ref_inner_product(const float* x, const float* w, const float* b, float* y, int M, int N, int K) {
  GUARD({
    LayerParameter lp;
    lp.mutable_inner_product_param()->set_num_output(N);
    lp.mutable_inner_product_param()->set_bias_term(b != nullptr);
    InnerProductLayer<float> layer(lp);
    Blob<float> bottom(M, K, 1, 1), top;
    fill(&bottom, x);
    BV bv(1, &bottom), tv(1, &top);
    layer.SetUp(bv, tv);
    fill(layer.blobs()[0].get(), w);
    if (b) fill(layer.blobs()[1].get(), b);
    layer.Forward(bv, tv);
    take(top, y);
  })
}

REF_API int ref_softmax(const float* x, float* y, int outer, int C, int inner) {
  GUARD({
    LayerParameter lp;
    SoftmaxLayer<float> layer(lp);
    Blob<float> bottom(outer, C, inner, 1), top;
    fill(&bottom, x);
    BV bv(1, &bottom), tv(1, &top);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    take(top, y);
  })
}

REF_API int ref_roipool(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W, int PH, int PW,
                        float spatial_scale, float pad_ratio) {
  GUARD({
    LayerParameter lp;
    ROIPoolingParameter* p = lp.mutable_roi_pooling_param();
    p->set_pooled_h(PH); p->set_pooled_w(PW); p->set_spatial_scale(spatial_scale); p->set_pad_ratio(pad_ratio);
    ROIPoolingLayer<float> layer(lp);
    Blob<float> bottom(N, C, H, W), broi(R, 5, 1, 1), top;
    fill(&bottom, feat); fill(&broi, rois);
    BV bv; bv.push_back(&bottom); bv.push_back(&broi);
    BV tv(1, &top);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    take(top, out);
  })
}

REF_API int ref_roialign(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W, int PH, int PW,
                         float spatial_scale, float pad_ratio) {
  GUARD({
    LayerParameter lp;
    ROIPoolingParameter* p = lp.mutable_roi_pooling_param();
    p->set_pooled_h(PH); p->set_pooled_w(PW); p->set_spatial_scale(spatial_scale); p->set_pad_ratio(pad_ratio);
    ROIAlignLayer<float> layer(lp);
    Blob<float> bottom(N, C, H, W), broi(R, 5, 1, 1), top;
    fill(&bottom, feat); fill(&broi, rois);
    BV bv; bv.push_back(&bottom); bv.push_back(&broi);
    BV tv(1, &top);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    take(top, out);
  })
}

REF_API int ref_eltwise(const float* const* xs, int nb, const float* coeffs, float* y, int count, int op) {
  GUARD({
    LayerParameter lp;
    lp.mutable_eltwise_param()->set_operation((EltwiseParameter_EltwiseOp)op);
    if (coeffs && op == 1) for (int b = 0; b < nb; ++b) lp.mutable_eltwise_param()->add_coeff(coeffs[b]);
    EltwiseLayer<float> layer(lp);
    std::vector<shared_ptr<Blob<float> > > hold;
    BV bv;
    for (int b = 0; b < nb; ++b) {
      hold.push_back(shared_ptr<Blob<float> >(new Blob<float>(1, 1, 1, count)));
      fill(hold.back().get(), xs[b]);
      bv.push_back(hold.back().get());
    }
    Blob<float> top;
    BV tv(1, &top);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    take(top, y);
  })
}

struct ref_boxoutput_params {
  float fg_thr, iou_thr;
  int nms_mode;
  float field_whr, field_xyr;
  int max_nms_num, max_post_nms_num;
  float min_size;
  int do_bbox_norm;
  float bbox_mean[4], bbox_std[4];
};

// Same calling convention as orc_boxoutput (minus the index bookkeeping the reference does not expose).
REF_API int ref_boxoutput(const float* const* heads, const int* hs, const int* ws, int nheads, int num, int channels,
                          const float* field_w, const float* field_h, const float* downsample, const ref_boxoutput_params* p,
                          float* rois_out, float* props_out, int cap) {
  try {
    LayerParameter lp;
    BoxOutputParameter* bo = lp.mutable_box_output_param();
    bo->set_fg_thr(p->fg_thr); bo->set_iou_thr(p->iou_thr);
    bo->set_nms_type(p->nms_mode == 1 ? "IOMU" : p->nms_mode == 2 ? "IOFU" : "IOU");
    bo->set_field_whr(p->field_whr); bo->set_field_xyr(p->field_xyr);
    bo->set_max_nms_num(p->max_nms_num); bo->set_max_post_nms_num(p->max_post_nms_num); bo->set_min_size(p->min_size);
    for (int j = 0; j < nheads; ++j) {
      bo->add_field_w((unsigned)field_w[j]); bo->add_field_h((unsigned)field_h[j]); bo->add_downsample_rate((unsigned)downsample[j]);
    }
    if (p->do_bbox_norm)
      for (int k = 0; k < 4; ++k) { lp.mutable_bbox_reg_param()->add_bbox_mean(p->bbox_mean[k]); lp.mutable_bbox_reg_param()->add_bbox_std(p->bbox_std[k]); }
    BoxOutputLayer<float> layer(lp);
    std::vector<shared_ptr<Blob<float> > > hold;
    BV bv;
    for (int j = 0; j < nheads; ++j) {
      hold.push_back(shared_ptr<Blob<float> >(new Blob<float>(num, channels, hs[j], ws[j])));
      fill(hold.back().get(), heads[j]);
      bv.push_back(hold.back().get());
    }
    Blob<float> t0, t1;
    BV tv; tv.push_back(&t0); tv.push_back(&t1);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    const int R = t0.num();
    if (R > cap) { g_err = "cap too small"; return -3; }
    take(t0, rois_out);
    if (props_out) take(t1, props_out);
    return R;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

REF_API int ref_decode_bbox(const float* bbox, const float* prior, float* out, int R, int bbox_dim, const float* mean,
                            const float* stdv) {
  GUARD({
    LayerParameter lp;
    lp.set_phase(TEST);
    if (mean && stdv)
      for (int k = 0; k < 4; ++k) { lp.mutable_bbox_reg_param()->add_bbox_mean(mean[k]); lp.mutable_bbox_reg_param()->add_bbox_std(stdv[k]); }
    DecodeBBoxLayer<float> layer(lp);
    Blob<float> b0(R, bbox_dim, 1, 1), b1(R, 5, 1, 1), top;
    fill(&b0, bbox); fill(&b1, prior);
    BV bv; bv.push_back(&b0); bv.push_back(&b1);
    BV tv(1, &top);
    layer.SetUp(bv, tv);
    layer.Forward(bv, tv);
    take(top, out);
  })
}

REF_API float ref_box_iou(float x1, float y1, float w1, float h1, float x2, float y2, float w2, float h2, int mode) {
  return BoxIOU<float>(x1, y1, w1, h1, x2, y2, w2, h2, mode == 1 ? "IOMU" : mode == 2 ? "IOFU" : "IOU");
}
