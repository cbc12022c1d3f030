// C ABI over caffe::Net<float> (include/mscnn_net.h).  Every entry point catches caffe::FatalError (the
// CHECK-failure exception of this build) and returns it as an error string instead of aborting the host process.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../../include/mscnn_hip.h"
#include "../../../include/mscnn_net.h"
#include "caffe/caffe.hpp"

using caffe::Blob;
using caffe::Caffe;
using caffe::Net;

struct mscnn_net {
  std::unique_ptr<Net<float> > net;
  int device;
  caffe::DeviceBuffer det_ws;
  caffe::DeviceBuffer img_in, img_ws;      // set_image: the uint8 frame and the resize scratch
  caffe::DeviceBuffer det_pack;            // detect: [count | dets | ids] in one allocation -> ONE D2H copy, one sync
  void* det_host = nullptr;                // host-coherent pinned memory: the blocking detect's pack (written by the kernels themselves)
  void* det_host_dev = nullptr;            // ... as the device addresses it
  size_t det_host_bytes = 0;
  std::vector<int> row_end;                // detect_image: end row of every image in the ROI blobs, of forward number rows_forward
  long rows_forward = -1;
  // detect_begin / detect_end: two pinned slots, each behind an event on the net's stream
  struct Slot { caffe::DeviceBuffer pack; void* host = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; int cap = 0, handoff_errors = 0; } slot[2];
  int slot_head = 0, slot_tail = 0, slots_inflight = 0;
  ~mscnn_net() {
    if (det_host) (void)hipHostFree(det_host);
    for (Slot& sl : slot) {
      if (sl.done) { (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
      if (sl.host) (void)hipHostFree(sl.host);
    }
  }
};

namespace {
thread_local std::string g_err;

template <class F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

void fill_shape(const Blob<float>& b, int* dims8, int* ndim) {
  *ndim = b.num_axes();
  for (int i = 0; i < 8; ++i) dims8[i] = i < b.num_axes() ? b.shape(i) : 1;
}

int create(caffe::NetParameter param, int device, mscnn_net** out, unsigned flags = 0, bool use_default_fusion = true) {
  return guarded([&] {
    CHECK(out != nullptr);
    if (device >= 0) Caffe::SetDevice(device);   // device < 0: graph construction only (no HIP device touched)
    Caffe::set_mode(Caffe::GPU);
    std::unique_ptr<mscnn_net> h(new mscnn_net());
    h->device = device;
    if (use_default_fusion) h->net.reset(new Net<float>(param, caffe::TEST));
    else h->net.reset(new Net<float>(param, caffe::TEST, !(flags & MSCNN_NET_NO_FUSION)));
    *out = h.release();
  });
}
}  // namespace

extern "C" {

const char* mscnn_net_last_error(void) { return g_err.c_str(); }

int mscnn_net_create_from_file(const char* path, int device, mscnn_net** out) {
  caffe::NetParameter p;
  int rc = guarded([&] { caffe::ReadNetParamsFromTextFileOrDie(path, &p); });
  return rc ? rc : create(p, device, out);
}

int mscnn_net_create_from_string(const char* text, int device, mscnn_net** out) {
  caffe::NetParameter p;
  int rc = guarded([&] { p = caffe::NetParameterFromString(text); });
  return rc ? rc : create(p, device, out);
}

int mscnn_net_create_from_string_ex(const char* text, int device, unsigned flags, mscnn_net** out) {
  caffe::NetParameter p;
  int rc = guarded([&] { p = caffe::NetParameterFromString(text); });
  return rc ? rc : create(p, device, out, flags, false);
}

void mscnn_net_destroy(mscnn_net* net) { delete net; }

int mscnn_net_load_caffemodel(mscnn_net* net, const char* path) { return guarded([&] { net->net->CopyTrainedLayersFrom(path); }); }

int mscnn_net_set_stream(void* stream) { Caffe::set_stream(stream); return 0; }

int mscnn_net_num_layers(const mscnn_net* n) { return (int)n->net->layers().size(); }
const char* mscnn_net_layer_name(const mscnn_net* n, int i) { return n->net->layer_names()[i].c_str(); }
const char* mscnn_net_layer_type(const mscnn_net* n, int i) { return n->net->layers()[i]->type(); }
int mscnn_net_layer_index(const mscnn_net* n, const char* name) {
  const auto& v = n->net->layer_names();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == name) return (int)i;
  return -1;
}
int mscnn_net_layer_num_bottoms(const mscnn_net* n, int l) { return (int)n->net->bottom_vecs()[l].size(); }
int mscnn_net_layer_num_tops(const mscnn_net* n, int l) { return (int)n->net->top_vecs()[l].size(); }
static const char* blob_name_of(const mscnn_net* n, const Blob<float>* b) {
  const auto& blobs = n->net->blobs();
  for (size_t i = 0; i < blobs.size(); ++i)
    if (blobs[i].get() == b) return n->net->blob_names()[i].c_str();
  return "";
}
const char* mscnn_net_layer_bottom(const mscnn_net* n, int l, int i) { return blob_name_of(n, n->net->bottom_vecs()[l][i]); }
const char* mscnn_net_layer_top(const mscnn_net* n, int l, int i) { return blob_name_of(n, n->net->top_vecs()[l][i]); }
int mscnn_net_layer_num_params(const mscnn_net* n, int l) { return (int)n->net->layers()[l]->blobs().size(); }
int mscnn_net_layer_param_shape(const mscnn_net* n, int l, int p, int* dims8, int* ndim) {
  return guarded([&] {
    CHECK_LT(p, (int)n->net->layers()[l]->blobs().size());
    fill_shape(*n->net->layers()[l]->blobs()[p], dims8, ndim);
  });
}
const char* mscnn_net_layer_param_text(const mscnn_net* n, int l) {
  static thread_local std::string text;
  text = n->net->layers()[l]->layer_param().raw().DebugString();
  return text.c_str();
}
int mscnn_net_layer_fused_away(const mscnn_net* n, int l) { return n->net->layer_fused_away()[l] ? 1 : 0; }
const char* mscnn_net_layer_kernel(const mscnn_net* n, int l) {
  auto* c = dynamic_cast<caffe::ConvolutionLayer<float>*>(n->net->layers()[l].get());
  if (c) return c->kernel_name();
  auto* ip = dynamic_cast<caffe::InnerProductLayer<float>*>(n->net->layers()[l].get());
  return ip ? ip->kernel_name() : "";
}
int mscnn_net_set_inner_product_algo(mscnn_net* n, int layer, int algo) {
  return guarded([&] {
    CHECK_LT(layer, (int)n->net->layers().size());
    CHECK(algo == 0 || algo == 1) << "inner product algo: 0 auto, 1 gemm.hip's stream-K kernel";
    for (int l = (layer < 0 ? 0 : layer); l < (layer < 0 ? (int)n->net->layers().size() : layer + 1); ++l)
      if (auto* ip = dynamic_cast<caffe::InnerProductLayer<float>*>(n->net->layers()[l].get())) ip->set_algo(algo);
      else CHECK_LT(layer, 0) << "layer " << n->net->layer_names()[l] << " is not an InnerProduct";
  });
}
double mscnn_net_layer_flops(const mscnn_net* n, int l) { return n->net->layers()[l]->ForwardFlops(); }
static caffe::ConvolutionLayer<float>* conv_of(const mscnn_net* n, int l) {
  return dynamic_cast<caffe::ConvolutionLayer<float>*>(n->net->layers()[l].get());
}
double mscnn_net_layer_executed_flops(const mscnn_net* n, int l) {
  auto* c = conv_of(n, l);
  return c ? c->ExecutedFlops() : n->net->layers()[l]->ForwardFlops();
}
int mscnn_net_set_conv_profiling(mscnn_net* n, int on) {
  return guarded([&] {
    for (size_t l = 0; l < n->net->layers().size(); ++l)
      if (auto* c = conv_of(n, (int)l)) c->set_profiling(on != 0);
  });
}
int mscnn_net_layer_stage_ms(const mscnn_net* n, int l, float ms_out[3]) {
  ms_out[0] = ms_out[1] = ms_out[2] = 0.f;
  auto* c = conv_of(n, l);
  return (c && c->StageMs(ms_out)) ? 0 : 1;
}
int mscnn_net_set_conv_algo(mscnn_net* n, int layer, int algo) {
  return guarded([&] {
    CHECK_LT(layer, (int)n->net->layers().size());
    for (int l = (layer < 0 ? 0 : layer); l < (layer < 0 ? (int)n->net->layers().size() : layer + 1); ++l)
      if (auto* c = conv_of(n, l)) { c->set_algo(algo); c->set_calibrated_direct(false); }
      else CHECK_LT(layer, 0) << "layer " << n->net->layer_names()[l] << " is not a Convolution";
  });
}
int mscnn_net_set_conv_tuning(mscnn_net* n, int layer, int variant, int grid, int flags) {
  return guarded([&] {
    CHECK_LT(layer, (int)n->net->layers().size());
    for (int l = (layer < 0 ? 0 : layer); l < (layer < 0 ? (int)n->net->layers().size() : layer + 1); ++l)
      if (auto* c = conv_of(n, l)) c->set_tuning(variant, grid, flags);
  });
}
int mscnn_net_set_precision(mscnn_net* n, const char* dtype) {
  return guarded([&] {
    const std::string d = dtype ? dtype : "";
    CHECK(d == "f32" || d == "f16" || d == "f16x3") << "precision must be f32, f16 or f16x3, not '" << d << "'";
    for (size_t l = 0; l < n->net->layers().size(); ++l) {
      // (a layer the numerical calibration sent to the direct fp32 kernel stays there in the fp32-grade modes: the calibration is about
      // the data, not about the mode that happened to be active; the reduced-precision f16 mode has its own tolerance policy)
      if (auto* c = conv_of(n, (int)l)) c->set_algo(d == "f16" ? 4 : c->calibrated_direct() ? 1 : d == "f16x3" ? 5 : 0);
      if (auto* ip = dynamic_cast<caffe::InnerProductLayer<float>*>(n->net->layers()[l].get())) { ip->set_f16(d == "f16"); ip->set_x3(d == "f16x3"); }
    }
  });
}
const char* mscnn_net_layer_dtype(const mscnn_net* n, int l) {
  if (auto* c = conv_of(n, l)) return c->dtype();
  if (auto* ip = dynamic_cast<caffe::InnerProductLayer<float>*>(n->net->layers()[l].get())) return ip->dtype();
  return "f32";
}
int mscnn_net_calibrate_numerics(mscnn_net* n, double tol, int* num_switched) {
  return guarded([&] {
    const std::vector<int> sw = n->net->CalibrateNumerics(tol);
    if (num_switched) *num_switched = (int)sw.size();
  });
}
double mscnn_net_layer_calibration_err(const mscnn_net* n, int l) { return n->net->calibration_err()[l]; }
int mscnn_net_set_numerics_watch(mscnn_net* n, int period, double tol) {
  return guarded([&] {
    CHECK_GE(period, 0);
    n->net->SetNumericsWatch(period, tol);
  });
}
int mscnn_net_numerics_watch_state(const mscnn_net* n, int* checks, int* switched_layers, int cap) {
  const std::vector<int>& sw = n->net->numerics_watch_switched();
  if (checks) *checks = n->net->numerics_watch_checks();
  for (int i = 0; switched_layers && i < cap && i < (int)sw.size(); ++i) switched_layers[i] = sw[i];
  return (int)sw.size();
}
int mscnn_net_set_auto_calibrate(mscnn_net* n, double tol) {
  return guarded([&] { n->net->SetAutoCalibrate(tol); });
}
int mscnn_net_set_chain_fusion(mscnn_net* n, int on) {
  return guarded([&] { n->net->SetChainFusion(on != 0); });
}
int mscnn_net_chain_pairs(const mscnn_net* n, int* producers, int* consumers, int cap) {
  const auto pairs = n->net->chain_pairs();
  for (int i = 0; i < cap && i < (int)pairs.size(); ++i) {
    if (producers) producers[i] = pairs[i].first;
    if (consumers) consumers[i] = pairs[i].second;
  }
  return (int)pairs.size();
}
int mscnn_net_auto_calibrate_state(const mscnn_net* n, int* checks, int* switched_layers, int cap) {
  const std::vector<int>& sw = n->net->auto_calibrate_switched();
  if (checks) *checks = n->net->auto_calibrate_checks();
  for (int i = 0; switched_layers && i < cap && i < (int)sw.size(); ++i) switched_layers[i] = sw[i];
  return (int)sw.size();
}
int mscnn_net_num_blobs(const mscnn_net* n) { return (int)n->net->blobs().size(); }
const char* mscnn_net_blob_name(const mscnn_net* n, int b) { return n->net->blob_names()[b].c_str(); }
int mscnn_net_blob_shape(const mscnn_net* n, const char* name, int* dims8, int* ndim) {
  return guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    fill_shape(*n->net->blob_by_name(name), dims8, ndim);
  });
}
int mscnn_net_num_inputs(const mscnn_net* n) { return n->net->num_inputs(); }
int mscnn_net_num_outputs(const mscnn_net* n) { return n->net->num_outputs(); }
const char* mscnn_net_output_name(const mscnn_net* n, int i) { return n->net->blob_names()[n->net->output_blob_indices()[i]].c_str(); }

int mscnn_net_set_param(mscnn_net* n, int l, int p, const float* host, size_t count) {
  return guarded([&] {
    auto& blobs = n->net->layers()[l]->blobs();
    CHECK_LT(p, (int)blobs.size());
    CHECK_EQ((size_t)blobs[p]->count(), count) << "param size mismatch for layer " << n->net->layer_names()[l];
    n->net->MaterializeStale();      // (intermediate blobs a chained forward left unwritten: the old weights' outputs, written before they change)
    std::memcpy(blobs[p]->mutable_cpu_data(), host, sizeof(float) * count);
    n->net->layers()[l]->OnWeightsChanged();
  });
}
int mscnn_net_get_param(mscnn_net* n, int l, int p, float* host, size_t count) {
  return guarded([&] {
    auto& blobs = n->net->layers()[l]->blobs();
    CHECK_LT(p, (int)blobs.size());
    CHECK_EQ((size_t)blobs[p]->count(), count);
    std::memcpy(host, blobs[p]->cpu_data(), sizeof(float) * count);
  });
}

int mscnn_net_set_blob(mscnn_net* n, const char* name, const float* host, size_t count) {
  return guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    n->net->MaterializePendingReadersOf(name);      // (a deferred ROI pooling that reads this blob writes its own blob first)
    auto b = n->net->blob_by_name(name);
    CHECK_EQ((size_t)b->count(), count) << "blob " << name << " has shape " << b->shape_string();
    std::memcpy(b->mutable_cpu_data(), host, sizeof(float) * count);
  });
}
int mscnn_net_set_blob_device(mscnn_net* n, const char* name, const float* dev, size_t count) {
  return guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    n->net->MaterializePendingReadersOf(name);      // (a deferred ROI pooling that reads this blob writes its own blob first)
    auto b = n->net->blob_by_name(name);
    CHECK_EQ((size_t)b->count(), count) << "blob " << name << " has shape " << b->shape_string();
    HIP_CHECK(hipMemcpyAsync(b->mutable_gpu_data(), dev, sizeof(float) * count, hipMemcpyDeviceToDevice, (hipStream_t)Caffe::stream()));
  });
}
int mscnn_net_set_image(mscnn_net* n, const char* name, const unsigned char* img_rgb, int on_device, int org_h, int org_w,
                        const float* mean_bgr) {
  return guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    n->net->MaterializePendingReadersOf(name);      // (blobs the last forward left unwritten that hang on this blob are written first, as in the other setters)
    auto b = n->net->blob_by_name(name);
    CHECK(b->num() == 1 && b->channels() == 3) << "set_image: blob " << name << " has shape " << b->shape_string();
    hipStream_t st = (hipStream_t)Caffe::stream();
    const unsigned char* dev_img = img_rgb;
    if (!on_device) {
      const size_t bytes = (size_t)org_h * org_w * 3;
      void* d = n->img_in.Reserve(bytes);
      HIP_CHECK(hipMemcpyAsync(d, img_rgb, bytes, hipMemcpyHostToDevice, st));
      dev_img = static_cast<const unsigned char*>(d);
    }
    const size_t wb = mscnn_preprocess_workspace_bytes(org_h, org_w, b->height(), b->width());
    void* ws = n->img_ws.Reserve(wb);
    static const float kMean[3] = {104.f, 117.f, 123.f};      // run_mscnn_detection.m:38
    const int rc = mscnn_preprocess_u8_f32(dev_img, org_h, org_w, b->mutable_gpu_data(), b->height(), b->width(),
                                           mean_bgr ? mean_bgr : kMean, ws, wb, st);
    CHECK_EQ(rc, 0) << mscnn_last_error();
  });
}
int mscnn_net_get_blob(mscnn_net* n, const char* name, float* host, size_t capacity, size_t* count) {
  return guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    auto b = n->net->blob_by_name(name);
    const float* src = b->cpu_data();      // (synchronises the stream)
    if (n->net->HandoffRecover()) {        // a hand-off timed out in the forward this blob comes from: the Net has run it again
      b = n->net->blob_by_name(name);
      src = b->cpu_data();
    }
    if (count) *count = (size_t)b->count();
    CHECK_LE((size_t)b->count(), capacity) << "buffer too small for blob " << name;
    std::memcpy(host, src, sizeof(float) * b->count());
  });
}
const float* mscnn_net_blob_device_ptr(mscnn_net* n, const char* name) {
  const float* p = nullptr;
  guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    p = n->net->blob_by_name(name)->gpu_data();
  });
  return p;
}

int mscnn_net_forward(mscnn_net* n) { return guarded([&] { n->net->Forward(); }); }
int mscnn_net_forward_from_to(mscnn_net* n, int from, int to) {
  return guarded([&] { n->net->ForwardFromTo(from, to < 0 ? (int)n->net->layers().size() - 1 : to); });
}
int mscnn_net_reshape(mscnn_net* n) { return guarded([&] { n->net->Reshape(); }); }
int mscnn_net_handoff_state(const mscnn_net* n, int* whole_tiles_forced) {
  if (whole_tiles_forced) *whole_tiles_forced = mscnn_wgemm_whole_tiles_forced();
  return n->net->handoff_errors();
}
int mscnn_net_set_layer_timing(mscnn_net* n, int on) { n->net->set_layer_timing(on != 0); return 0; }
float mscnn_net_layer_ms(const mscnn_net* n, int l) { return n->net->layer_ms()[l]; }

size_t mscnn_net_detect_pack_bytes(int cap) {
  const size_t rows = (size_t)(cap > 0 ? cap : 1);
  return (16 + rows * (5 * sizeof(double) + sizeof(int)) + 15) / 16 * 16;
}

// Final stage into the fixed-capacity device pack [count, R, cap, 0 | cap x 5 doubles | cap ints]; no host transfer.
// [row0, row0 + rows) of the ROI blobs (rows < 0: all of them -- the batch-1 form)
// the net's host-side landing place for packs: host-coherent pinned memory, also addressable by the device (det_host_dev)
static void ensure_det_host(mscnn_net* n, size_t total) {
  if (n->det_host_bytes >= total) return;
  if (n->det_host) HIP_CHECK(hipHostFree(n->det_host));
  n->det_host = nullptr; n->det_host_bytes = 0; n->det_host_dev = nullptr;
  HIP_CHECK(hipHostMalloc(&n->det_host, total, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
  HIP_CHECK(hipHostGetDevicePointer(&n->det_host_dev, n->det_host, 0));
  n->det_host_bytes = total;
}

// pack_at: a device-visible address to write the pack to instead of a device buffer (host-coherent pinned memory: the kernels only ever
// WRITE the pack, so the blocking mscnn_net_detect lets them write straight into the host's copy)
static void detect_into_pack(mscnn_net* n, const mscnn_detect_params* p, int cap, int* R_out, bool with_header, int row0 = 0, int nrows = -1,
                             caffe::DeviceBuffer* pack_buf = nullptr, char* pack_at = nullptr) {
  caffe::DeviceBuffer& det_pack = pack_buf ? *pack_buf : n->det_pack;
  CHECK(p != nullptr);
  CHECK(n->net->has_blob("bbox_pred") && n->net->has_blob("cls_pred") && n->net->has_blob("proposals_score"))
      << "net has no bbox_pred / cls_pred / proposals_score outputs";
  auto bbox = n->net->blob_by_name("bbox_pred");
  auto cls = n->net->blob_by_name("cls_pred");
  auto props = n->net->blob_by_name("proposals_score");
  const int R_all = props->num();
  CHECK_EQ(bbox->num(), R_all);
  CHECK_EQ(cls->num(), R_all);
  if (nrows < 0) {
    // the MATLAB stage is written for one image (run_mscnn_detection.m:75-120 runs on the outputs of a batch-1 forward): the NMS
    // must never mix boxes of different images
    if (n->net->num_inputs() > 0 && n->net->input_blobs()[0]->num_axes() == 4)
      CHECK_EQ(n->net->input_blobs()[0]->num(), 1) << "the net's input holds " << n->net->input_blobs()[0]->num()
                                                   << " images: use mscnn_net_detect_image (one final stage per image)";
    row0 = 0; nrows = R_all;
  }
  CHECK(row0 >= 0 && nrows >= 0 && row0 + nrows <= R_all);
  const int R = nrows;
  const int per_row_cls = R_all > 0 ? cls->count() / R_all : 1, per_row_box = R_all > 0 ? bbox->count() / R_all : 4;
  if (with_header && R > cap) {
    // Multi-GPU pack: a rank that fails here alone would leave the others blocked in the all_gather.  Mark the overflow in the header
    // {-1, R, cap, 0} and take part in the exchange: mscnn_net_unpack_detections then fails on EVERY rank, naming the numbers.
    char* pk = static_cast<char*>(det_pack.Reserve(mscnn_net_detect_pack_bytes(cap)));
    hipStream_t s0 = (hipStream_t)Caffe::stream();
    int* h = reinterpret_cast<int*>(pk);
    const int words[4] = {-1, R, cap, 0};
    MSCNN_CHECK(mscnn_store_words_i32(h, words, 4, s0));
    if (R_out) *R_out = R;
    return;
  }
  CHECK_LE(R, cap) << "detection pack capacity " << cap << " < " << R << " ROIs (size it by BoxOutput's max_nms_num)";
  mscnn_detections_desc d;
  d.ncls = per_row_cls;
  if (R_all > 0) CHECK_EQ(per_row_box, 4 * d.ncls);
  d.cls_id = p->cls_id;
  for (int k = 0; k < 4; ++k) { d.bbox_mean[k] = p->bbox_mean[k]; d.bbox_std[k] = p->bbox_std[k]; }
  d.proposal_thr = p->proposal_thr;
  d.ratio_h = p->ratio_h; d.ratio_w = p->ratio_w; d.org_h = p->org_h; d.org_w = p->org_w; d.nms_overlap = p->nms_overlap;
  const size_t wb = mscnn_detections_workspace_bytes(R);
  void* ws = n->det_ws.Reserve(wb);
  const size_t rows = (size_t)(cap > 0 ? cap : 1), total = mscnn_net_detect_pack_bytes(cap);
  char* pack = pack_at ? pack_at : static_cast<char*>(det_pack.Reserve(total));
  int* hdr = reinterpret_cast<int*>(pack);
  double* dets = reinterpret_cast<double*>(pack + 16);
  int* ids = reinterpret_cast<int*>(pack + 16 + sizeof(double) * 5 * rows);
  hipStream_t st = (hipStream_t)Caffe::stream();
  // header {count (written by the kernels), R, cap, 0}: only the multi-GPU pack needs R / cap on the device -- two 4-byte fills
  if (with_header) {
    const int words[3] = {R, cap, 0};
    MSCNN_CHECK(mscnn_store_words_i32(hdr + 1, words, 3, st));      // (one launch; three 4-byte memsets were three)
  }
  MSCNN_CHECK(mscnn_detections_fwd(&d, bbox->gpu_data() + (size_t)row0 * per_row_box, cls->gpu_data() + (size_t)row0 * per_row_cls,
                                   props->gpu_data() + (size_t)row0 * 6, R, dets, ids, hdr, ws, wb, st));
  if (R_out) *R_out = R;
}

// Row range of image `image` in the ROI blobs: BoxOutput emits image after image (box_output_layer.cpp:107), column 0 of every row
// is its image (:156).  One small D2H read of the proposals blob per forward (cached until the next forward).
static void image_rows(mscnn_net* n, int image, int* row0, int* rows) {
  CHECK(n->net->has_blob("proposals_score")) << "net has no proposals_score output";
  auto props = n->net->blob_by_name("proposals_score");
  const int R = props->num(), num = n->net->num_inputs() > 0 ? n->net->input_blobs()[0]->num() : 1;
  CHECK(image >= 0 && image < num) << "image " << image << " of a batch of " << num;
  if (n->rows_forward != n->net->forward_count() || (int)n->row_end.size() != num) {
    const float* h = props->cpu_data();      // (synchronises the stream)
    n->row_end.assign(num, 0);
    int prev = 0;
    for (int r = 0; r < R; ++r) {
      const int img = (int)h[(size_t)r * 6];
      CHECK(img >= prev && img < num) << "proposals are not grouped by image";
      // (the dummy row [0 0 0 0 0 0] the layer emits when nothing survives in the whole batch carries score 0 and image 0)
      prev = img;
      n->row_end[img] = r + 1;
    }
    for (int i = 1; i < num; ++i) if (n->row_end[i] < n->row_end[i - 1]) n->row_end[i] = n->row_end[i - 1];
    n->rows_forward = n->net->forward_count();
  }
  *row0 = image > 0 ? n->row_end[image - 1] : 0;
  *rows = n->row_end[image] - *row0;
}

int mscnn_net_detect_device(mscnn_net* n, const mscnn_detect_params* p, int cap, const void** pack_dev) {
  return guarded([&] {
    CHECK(pack_dev != nullptr);
    detect_into_pack(n, p, cap, nullptr, true);
    *pack_dev = n->det_pack.get();
  });
}

// ---- the final stage of a STREAM of frames: frame i's detections travel to the host under frame i + 1's trunk ----------------------
int mscnn_net_detect_begin(mscnn_net* n, const mscnn_detect_params* p, int cap) {
  return guarded([&] {
    CHECK(n->slots_inflight < 2) << "detect_begin: two frames already in flight (call mscnn_net_detect_end)";
    mscnn_net::Slot& sl = n->slot[n->slot_head];
    detect_into_pack(n, p, cap, nullptr, true, 0, -1, &sl.pack);      // (a device pack per slot: its copy may still run when the next frame's stage starts)
    const size_t total = mscnn_net_detect_pack_bytes(cap);
    if (sl.bytes < total) {
      if (sl.host) HIP_CHECK(hipHostFree(sl.host));
      sl.host = nullptr; sl.bytes = 0;
      HIP_CHECK(hipHostMalloc(&sl.host, total, hipHostMallocDefault));
      sl.bytes = total;
    }
    if (!sl.done) HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    hipStream_t st = (hipStream_t)Caffe::stream();
    // (the copy stays in the compute stream: on a stream of its own, beside the next frame's first kernels, it measured the same --
    // 224.85 against 224.83 images/s over 300 frames, tools/sessions/r06_s8.sh -- and needed a device pack per slot to be safe)
    HIP_CHECK(hipMemcpyAsync(sl.host, sl.pack.get(), total, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(sl.done, st));
    sl.cap = cap;
    sl.handoff_errors = n->net->handoff_errors();      // (an event of THIS frame's trunk was answered inside its forward, before this point)
    n->slot_head ^= 1;
    ++n->slots_inflight;
  });
}

int mscnn_net_detect_end(mscnn_net* n, double* dets_host, int* ids_host, int cap, int* num_dets, int* num_rois) {
  return guarded([&] {
    CHECK(n->slots_inflight > 0) << "detect_end: nothing in flight";
    mscnn_net::Slot& sl = n->slot[n->slot_tail];
    CHECK_EQ(sl.cap, cap) << "detect_end: the oldest frame in flight was begun with another capacity";
    HIP_CHECK(hipEventSynchronize(sl.done));
    n->slot_tail ^= 1;
    --n->slots_inflight;
    // A stream-K hand-off that timed out since the last look may have poisoned a tile of THIS frame or of the one enqueued behind it,
    // and neither can be run again from here (the input blob already holds a later frame): whole-tile scheduling is forced for the
    // process as everywhere else, and the caller is told to submit its last two frames again.
    // (the event may also have been noticed already -- and counted -- by the NEXT frame's forward, behind its BoxOutput read: that forward
    // restarted itself, but this frame's pack had left the device by then)
    const bool seen_now = n->net->HandoffEventSeen();
    CHECK(!seen_now && n->net->handoff_errors() == sl.handoff_errors) << "a stream-K hand-off timed out while frames were in flight (mscnn_net_handoff_state): whole-tile "
                                          "scheduling from now on; the frames begun since the last mscnn_net_detect_end must be submitted again";
    const int rc = mscnn_net_unpack_detections(sl.host, cap, dets_host, ids_host, num_dets, num_rois);
    CHECK_EQ(rc, 0) << g_err;
  });
}

int mscnn_net_unpack_detections(const void* pack_host, int cap, double* dets_host, int* ids_host, int* num_dets, int* num_rois) {
  return guarded([&] {
    CHECK(pack_host && num_dets);
    const char* hp = static_cast<const char*>(pack_host);
    const int* hdr = reinterpret_cast<const int*>(hp);
    const int D = hdr[0], R = hdr[1];
    CHECK_EQ(hdr[2], cap) << "detection pack was written for another capacity";
    CHECK(D != -1) << "detection pack capacity " << cap << " < " << R << " ROIs on the rank that wrote this pack (size it by BoxOutput's max_nms_num)";
    CHECK(D >= 0 && D <= R && R <= cap) << "corrupt detection pack: " << D << " detections, " << R << " ROIs, capacity " << cap;
    const size_t rows = (size_t)(cap > 0 ? cap : 1);
    if (D > 0 && dets_host) std::memcpy(dets_host, hp + 16, sizeof(double) * 5 * D);
    if (D > 0 && ids_host) std::memcpy(ids_host, hp + 16 + sizeof(double) * 5 * rows, sizeof(int) * D);
    *num_dets = D;
    if (num_rois) *num_rois = R;
  });
}

int mscnn_net_detect(mscnn_net* n, const mscnn_detect_params* p, double* dets_host, int* ids_host, int cap, int* num_dets,
                     int* num_rois) {
  return guarded([&] {
    CHECK(p && dets_host && num_dets);
    // single-GPU form: the pack is sized by this image's ROI count (16 + 44 R bytes) and -- round 6 -- lives in host-coherent pinned
    // memory that the final stage's kernels write directly (they only ever store into the pack): no D2H copy behind them, the host
    // waits for the stream and reads.  (An in-stream copy cost ~11 us of copy-engine start-up + its own time per frame:
    // profiles/r06_kernel_gaps.txt.)  Above 4032 ROIs -- the tiled path, whose kernels are not audited for that -- the pack stays on
    // the device and is copied as before.
    int R = 0, rows = 0;
    hipStream_t st = (hipStream_t)Caffe::stream();
    auto run = [&]() {
      rows = n->net->has_blob("proposals_score") ? n->net->blob_by_name("proposals_score")->num() : 0;
      const size_t total = mscnn_net_detect_pack_bytes(rows);
      ensure_det_host(n, total);
      if (rows >= 1 && rows <= 4032) {
        *static_cast<volatile int*>(n->det_host) = -1;
        detect_into_pack(n, p, rows, &R, false, 0, -1, nullptr, static_cast<char*>(n->det_host_dev));
      } else {
        detect_into_pack(n, p, rows, &R, false);
        HIP_CHECK(hipMemcpyAsync(n->det_host, n->det_pack.get(), total, hipMemcpyDeviceToHost, st));
      }
      HIP_CHECK(hipStreamSynchronize(st));
    };
    run();
    // a stream-K hand-off of roi_c1 / fc6 timed out in the forward these detections come from (mscnn_net_handoff_state): the Net has
    // run the frame again on whole tiles -- redo this stage on the new outputs
    if (n->net->HandoffRecover()) run();
    const char* hp = static_cast<const char*>(n->det_host);
    const int Dd = *reinterpret_cast<const int*>(hp);
    CHECK_LE(Dd, cap) << "detections buffer too small";
    CHECK_LE(Dd, rows);
    const size_t prow = (size_t)(rows > 0 ? rows : 1);
    if (Dd > 0) {
      std::memcpy(dets_host, hp + 16, sizeof(double) * 5 * Dd);
      if (ids_host) std::memcpy(ids_host, hp + 16 + sizeof(double) * 5 * prow, sizeof(int) * Dd);
    }
    *num_dets = Dd;
    if (num_rois) *num_rois = R;
  });
}

int mscnn_net_detect_image(mscnn_net* n, const mscnn_detect_params* p, int image, double* dets_host, int* ids_host, int cap,
                           int* num_dets, int* num_rois) {
  return guarded([&] {
    CHECK(p && dets_host && num_dets);
    int row0 = 0, rows = 0, R = 0;
    image_rows(n, image, &row0, &rows);
    if (n->net->HandoffRecover()) image_rows(n, image, &row0, &rows);      // (image_rows synchronised the stream: see mscnn_net_handoff_state)
    hipStream_t st = (hipStream_t)Caffe::stream();
    size_t total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      detect_into_pack(n, p, rows, &R, false, row0, rows);
      total = mscnn_net_detect_pack_bytes(rows);
      ensure_det_host(n, total);
      HIP_CHECK(hipMemcpyAsync(n->det_host, n->det_pack.get(), total, hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      if (attempt == 1 || !n->net->HandoffRecover()) break;
      image_rows(n, image, &row0, &rows);      // the frame has been run again on whole tiles: once more on the new outputs
    }
    const char* hp = static_cast<const char*>(n->det_host);
    const int Dd = *reinterpret_cast<const int*>(hp);
    CHECK_LE(Dd, cap) << "detections buffer too small";
    CHECK_LE(Dd, rows);
    const size_t prow = (size_t)(rows > 0 ? rows : 1);
    if (Dd > 0) {
      std::memcpy(dets_host, hp + 16, sizeof(double) * 5 * Dd);
      if (ids_host) {
        std::memcpy(ids_host, hp + 16 + sizeof(double) * 5 * prow, sizeof(int) * Dd);
        for (int i = 0; i < Dd; ++i) ids_host[i] += row0;      // rows of the net's ROI blobs, not of the image's range
      }
    }
    *num_dets = Dd;
    if (num_rois) *num_rois = R;
  });
}

int mscnn_net_reshape_input(mscnn_net* n, const char* name, const int* dims, int ndim) {
  return guarded([&] {
    CHECK(n->net->has_blob(name)) << "Unknown blob name " << name;
    CHECK(dims && ndim >= 1 && ndim <= 8);
    bool is_input = false;
    for (int i = 0; i < n->net->num_inputs(); ++i)
      is_input = is_input || n->net->blob_names()[n->net->input_blob_indices()[i]] == name;
    CHECK(is_input) << "blob " << name << " is not a net input";
    n->net->MaterializePendingReadersOf(name);
    n->net->MaterializeStale();      // (blobs the last forward left unwritten belong to the OLD shape)
    std::vector<int> shape(dims, dims + ndim);
    n->net->blob_by_name(name)->Reshape(shape);      // blob->Reshape + net->Reshape: matcaffe's net.blobs('data').reshape(..); net.reshape()
    n->net->Reshape();
  });
}

int mscnn_net_detect_cascade(mscnn_net* n, const mscnn_detect_params* p, float det_thr, const char* bbox_blob, const char* prob_blob,
                             const char* proposal_blob, double* dets_host, int* ids_host, int cap, int* num_dets, int* num_rois) {
  return guarded([&] {
    CHECK(p && dets_host && num_dets && bbox_blob && prob_blob && proposal_blob);
    for (const char* b : {bbox_blob, prob_blob, proposal_blob}) CHECK(n->net->has_blob(b)) << "Unknown blob name " << b;
    auto boxes = n->net->blob_by_name(bbox_blob);
    auto prob = n->net->blob_by_name(prob_blob);
    auto props = n->net->blob_by_name(proposal_blob);
    const int R = boxes->num();
    CHECK_EQ(prob->num(), R);
    CHECK_EQ(props->num(), R);
    CHECK_EQ(boxes->count(), 5 * R) << bbox_blob << " is not an [R, 5] box blob";
    mscnn_detections_desc d;
    std::memset(&d, 0, sizeof(d));
    d.ncls = R > 0 ? prob->count() / R : 1;
    d.cls_id = p->cls_id;
    d.ratio_h = p->ratio_h; d.ratio_w = p->ratio_w; d.org_h = p->org_h; d.org_w = p->org_w; d.nms_overlap = p->nms_overlap;
    const size_t wb = mscnn_detections_workspace_bytes(R);
    void* ws = n->det_ws.Reserve(wb);
    const size_t rows = (size_t)(R > 0 ? R : 1), total = mscnn_net_detect_pack_bytes((int)rows);
    char* pack = static_cast<char*>(n->det_pack.Reserve(total));
    hipStream_t st = (hipStream_t)Caffe::stream();
    MSCNN_CHECK(mscnn_detections_cascade_fwd(&d, det_thr, boxes->gpu_data(), prob->gpu_data(), props->gpu_data(), R,
                                             reinterpret_cast<double*>(pack + 16), reinterpret_cast<int*>(pack + 16 + sizeof(double) * 5 * rows),
                                             reinterpret_cast<int*>(pack), ws, wb, st));
    ensure_det_host(n, total);
    HIP_CHECK(hipMemcpyAsync(n->det_host, pack, total, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const char* hp = static_cast<const char*>(n->det_host);
    const int D = *reinterpret_cast<const int*>(hp);
    CHECK_LE(D, cap) << "detections buffer too small";
    if (D > 0) {
      std::memcpy(dets_host, hp + 16, sizeof(double) * 5 * D);
      if (ids_host) std::memcpy(ids_host, hp + 16 + sizeof(double) * 5 * rows, sizeof(int) * D);
    }
    *num_dets = D;
    if (num_rois) *num_rois = R;
  });
}

}  // extern "C"
