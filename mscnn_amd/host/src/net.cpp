// Net<Dtype>: graph construction + sequential forward executor.
// Reference behaviour restated from src/caffe/net.cpp:49-284 (Init), :544-575 (ForwardFromTo / Forward), :743-747
// (Reshape), src/caffe/util/upgrade_proto.cpp:966-1003 (legacy input upgrade) and
// src/caffe/util/insert_splits.cpp:14-126 (automatic Split layers, identical blob/layer naming).
#include "caffe/util/hdf5_lite.hpp"
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <utility>

#include "../../../include/mscnn_hip.h"
#include "caffe/layer_factory.hpp"
#include "caffe/layers/mscnn_layers.hpp"
#include "caffe/net.hpp"
#include <algorithm>

namespace caffe {

using std::make_pair;
using std::map;
using std::pair;
using std::set;

void UpgradeNetInput(const NetParameter& in, vector<LayerParameter>* layers) {
  layers->clear();
  // legacy `input:` + 4 x `input_dim:` (or input_shape) becomes an Input layer named "input" at the front
  const bool has_shape = in.input_shape_size() > 0;
  const bool has_dim = in.input_dim_size() > 0;
  if (in.input_size() > 0) {
    LayerParameter lp;
    lp.set_name("input");
    lp.set_type("Input");
    TextMessage* ip = lp.mutable_raw()->add_message("input_param");
    for (int i = 0; i < in.input_size(); ++i) {
      lp.add_top(in.input(i));
      TextMessage* shape = ip->add_message("shape");
      if (has_shape) {
        const BlobShape s = in.input_shape(i);
        for (int d = 0; d < s.dim_size(); ++d) shape->add_scalar("dim", std::to_string(s.dim(d)));
      } else if (has_dim) {
        CHECK_EQ(in.input_dim_size(), 4 * in.input_size()) << "input_dim must have 4 entries per input";
        for (int d = 0; d < 4; ++d) shape->add_scalar("dim", std::to_string(in.input_dim(4 * i + d)));
      }
    }
    layers->push_back(lp);
  }
  CHECK(!in.has_legacy_layers()) << "V1 'layers' prototxt is not supported; use 'layer'";
  for (int i = 0; i < in.layer_size(); ++i) layers->push_back(in.layer(i).Clone());
}

// Fan-out.  A top that more than one bottom reads gets a Split layer right behind its producer and every reader a copy of its own;
// the names ("<blob>_<producer>_<top index>_split[_<k>]") are the ones insert_splits.cpp:110-124 composes, because users address blobs
// by them.  One table of blob VERSIONS (an in-place layer re-issues its bottom's name: the readers after it read the new version);
// pass 1 counts the readers of each version, pass 2 emits the layers, handing out the copies in reading order.
namespace {
struct BlobVersion {
  int layer, top;      // who wrote it
  int readers;         // bottoms that read this version
  int handed_out;      // copies already given to a reader (pass 2)
};
string SplitStem(const LayerParameter& producer, int top) {
  return producer.top(top) + "_" + producer.name() + "_" + std::to_string(top) + "_split";
}
}  // namespace

void InsertSplits(const vector<LayerParameter>& in, vector<LayerParameter>* out) {
  vector<BlobVersion> versions;
  map<string, int> newest;                           // blob name -> index of its newest version
  vector<vector<int> > reads(in.size());             // reads[i][j]: the version bottom j of layer i reads
  vector<int> first_top(in.size() + 1, 0);           // versions [first_top[i], first_top[i + 1]) are layer i's tops
  for (size_t i = 0; i < in.size(); ++i) {
    for (int j = 0; j < in[i].bottom_size(); ++j) {
      const map<string, int>::const_iterator it = newest.find(in[i].bottom(j));
      CHECK(it != newest.end()) << "Unknown bottom blob '" << in[i].bottom(j) << "' (layer '" << in[i].name() << "', bottom index " << j << ")";
      reads[i].push_back(it->second);
      ++versions[it->second].readers;
    }
    first_top[i] = (int)versions.size();
    for (int j = 0; j < in[i].top_size(); ++j) {
      newest[in[i].top(j)] = (int)versions.size();
      versions.push_back(BlobVersion{(int)i, j, 0, 0});
    }
  }
  first_top[in.size()] = (int)versions.size();
  out->clear();
  for (size_t i = 0; i < in.size(); ++i) {
    LayerParameter lp = in[i].Clone();
    for (int j = 0; j < lp.bottom_size(); ++j) {
      BlobVersion& v = versions[reads[i][j]];
      if (v.readers > 1) lp.set_bottom(j, SplitStem(in[v.layer], v.top) + "_" + std::to_string(v.handed_out++));
    }
    out->push_back(lp);
    for (int vi = first_top[i]; vi < first_top[i + 1]; ++vi) {
      const BlobVersion& v = versions[vi];
      if (v.readers < 2) continue;
      LayerParameter sp;
      sp.set_type("Split");
      sp.set_name(SplitStem(in[i], v.top));
      sp.add_bottom(in[i].top(v.top));
      for (int k = 0; k < v.readers; ++k) sp.add_top(sp.name() + "_" + std::to_string(k));
      out->push_back(sp);
    }
  }
}

static bool DefaultFusion() {
  const char* nofuse = std::getenv("MSCNN_NO_FUSE");      // host-runtime option (not a kernel switch): results are identical
  return !(nofuse && *nofuse && *nofuse != '0');
}

template <typename Dtype>
Net<Dtype>::Net(const NetParameter& param, Phase phase) : phase_(phase), fusion_(DefaultFusion()), timing_(false) { Init(param); }
template <typename Dtype>
Net<Dtype>::Net(const NetParameter& param, Phase phase, bool fusion) : phase_(phase), fusion_(fusion), timing_(false) { Init(param); }

template <typename Dtype>
Net<Dtype>::Net(const string& param_file, Phase phase) : phase_(phase), fusion_(DefaultFusion()), timing_(false) {
  NetParameter param;
  ReadNetParamsFromTextFileOrDie(param_file, &param);
  Init(param);
}

static bool StateMeetsRule(Phase phase, const NetStateRule& rule) { return !rule.has_phase() || rule.phase() == phase; }

template <typename Dtype>
void Net<Dtype>::Init(const NetParameter& in_param) {
  name_ = in_param.name();
  vector<LayerParameter> upgraded, filtered, layers;
  UpgradeNetInput(in_param, &upgraded);
  // FilterNet (net.cpp:286-318): include/exclude rules on phase
  for (const LayerParameter& lp : upgraded) {
    CHECK(lp.include_size() == 0 || lp.exclude_size() == 0) << "Specify either include rules or exclude rules; not both.";
    bool included = (lp.include_size() == 0);
    for (int j = 0; included && j < lp.exclude_size(); ++j)
      if (StateMeetsRule(phase_, lp.exclude(j))) included = false;
    for (int j = 0; !included && j < lp.include_size(); ++j)
      if (StateMeetsRule(phase_, lp.include(j))) included = true;
    if (included) filtered.push_back(lp);
  }
  InsertSplits(filtered, &layers);

  map<string, int> blob_name_to_idx;
  set<string> available_blobs;
  bottom_vecs_.resize(layers.size());
  top_vecs_.resize(layers.size());
  bottom_id_vecs_.resize(layers.size());
  top_id_vecs_.resize(layers.size());
  for (size_t layer_id = 0; layer_id < layers.size(); ++layer_id) {
    LayerParameter lp = layers[layer_id];
    if (!lp.has_phase()) lp.set_phase(phase_);
    layers_.push_back(LayerRegistry<Dtype>::CreateLayer(lp));
    layer_names_.push_back(lp.name());
    LOG(INFO) << "Creating Layer " << lp.name();
    // AppendBottom (net.cpp:426-451)
    for (int b = 0; b < lp.bottom_size(); ++b) {
      const string& blob_name = lp.bottom(b);
      CHECK(available_blobs.count(blob_name)) << "Unknown bottom blob '" << blob_name << "' (layer '" << lp.name() << "', bottom index " << b << ")";
      const int blob_id = blob_name_to_idx[blob_name];
      LOG(INFO) << lp.name() << " <- " << blob_name;
      bottom_vecs_[layer_id].push_back(blobs_[blob_id].get());
      bottom_id_vecs_[layer_id].push_back(blob_id);
      available_blobs.erase(blob_name);
    }
    // AppendTop (net.cpp:376-423)
    for (int t = 0; t < lp.top_size(); ++t) {
      const string& blob_name = lp.top(t);
      if (lp.bottom_size() > t && blob_name == lp.bottom(t)) {
        LOG(INFO) << lp.name() << " -> " << blob_name << " (in-place)";
        top_vecs_[layer_id].push_back(blobs_[blob_name_to_idx[blob_name]].get());
        top_id_vecs_[layer_id].push_back(blob_name_to_idx[blob_name]);
      } else {
        CHECK(!blob_name_to_idx.count(blob_name)) << "Top blob '" << blob_name << "' produced by multiple sources.";
        LOG(INFO) << lp.name() << " -> " << blob_name;
        shared_ptr<Blob<Dtype> > blob_pointer(new Blob<Dtype>());
        const int blob_id = (int)blobs_.size();
        blobs_.push_back(blob_pointer);
        blob_names_.push_back(blob_name);
        blob_name_to_idx[blob_name] = blob_id;
        top_id_vecs_[layer_id].push_back(blob_id);
        top_vecs_[layer_id].push_back(blob_pointer.get());
        if (lp.type() == "Input") {
          net_input_blob_indices_.push_back(blob_id);
          net_input_blobs_.push_back(blob_pointer.get());
        }
      }
      available_blobs.insert(blob_name);
    }
    layers_[layer_id]->SetUp(bottom_vecs_[layer_id], top_vecs_[layer_id]);
    for (size_t t = 0; t < top_vecs_[layer_id].size(); ++t)
      LOG(INFO) << "Top shape: " << top_vecs_[layer_id][t]->shape_string();   // net.cpp:158
  }
  // remaining blobs are outputs, in std::set (alphabetical) order -- net.cpp:267-274
  for (set<string>::iterator it = available_blobs.begin(); it != available_blobs.end(); ++it) {
    LOG(INFO) << "This network produces output " << *it;
    net_output_blobs_.push_back(blobs_[blob_name_to_idx[*it]].get());
    net_output_blob_indices_.push_back(blob_name_to_idx[*it]);
  }
  for (size_t i = 0; i < blob_names_.size(); ++i) blob_names_index_[blob_names_[i]] = (int)i;
  for (size_t i = 0; i < layer_names_.size(); ++i) layer_names_index_[layer_names_[i]] = (int)i;
  fused_away_.assign(layers_.size(), false);
  fused_producers_.assign(layers_.size(), vector<int>());
  calib_err_.assign(layers_.size(), 0.0);
  layer_ms_.assign(layers_.size(), 0.f);
  if (fusion_) ApplyFusion();
  WireAmax();
  handoff_seen_ = mscnn_wgemm_handoff_event();      // (events of launches before this net existed are not its business)
  LOG(INFO) << "Network initialization done.";
}

template <typename Dtype>
Net<Dtype>::~Net() {
  if (amax_slots_) (void)hipFree(amax_slots_);
}

template <typename Dtype>
void Net<Dtype>::WireAmax() {
  // (the slots are allocated by the first forward that has a split-fp16 layer: building a net needs no device)
  amax_src_.assign(layers_.size(), -1);
  vector<int> src(blobs_.size(), -1);           // blob id -> convolution layer whose max |y| bounds the blob
  for (size_t i = 0; i < layers_.size(); ++i) {
    const string t = layers_[i]->type();
    ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
    int s = -1;
    if (c) {
      const int b = bottom_id_vecs_[i].empty() ? -1 : src[bottom_id_vecs_[i][0]];
      amax_src_[i] = b;
      s = (int)i;
    } else if (t == "InnerProduct") {
      amax_src_[i] = bottom_id_vecs_[i].empty() ? -1 : src[bottom_id_vecs_[i][0]];      // (its own output bounds nothing)
    } else if (t == "Split" || t == "ReLU" || t == "Dropout" || t == "ROIPooling" ||
               (t == "Pooling" && layers_[i]->layer_param().pooling_param().pool() == PoolingParameter_PoolMethod_MAX)) {
      // max |.| never grows through these (ReLU with a negative slope in [-1, 1] included)
      s = bottom_id_vecs_[i].empty() ? -1 : src[bottom_id_vecs_[i][0]];
      if (t == "ReLU" && std::fabs(layers_[i]->layer_param().relu_param().negative_slope()) > 1) s = -1;
    } else if (t == "Concat" && !bottom_id_vecs_[i].empty()) {
      s = src[bottom_id_vecs_[i][0]];
      for (size_t b = 1; b < bottom_id_vecs_[i].size(); ++b) if (src[bottom_id_vecs_[i][b]] != s) s = -1;
    }
    for (size_t k = 0; k < top_id_vecs_[i].size(); ++k) src[top_id_vecs_[i][k]] = s;
  }
}

// Conv / InnerProduct followed by an in-place ReLU on its top (every trunk conv of the deploy nets): the ReLU is
// applied in the producer's epilogue and the ReLU layer becomes a no-op.  Bit-identical to the unfused pair.
template <typename Dtype>
void Net<Dtype>::ApplyFusion() {
  for (size_t i = 0; i + 1 < layers_.size(); ++i) {
    const string t = layers_[i]->type();
    if (t != "Convolution" && t != "InnerProduct") continue;
    if (string(layers_[i + 1]->type()) != "ReLU") continue;
    if (top_vecs_[i].size() != 1 || bottom_vecs_[i + 1].size() != 1) continue;
    if (bottom_vecs_[i + 1][0] != top_vecs_[i][0] || top_vecs_[i + 1][0] != top_vecs_[i][0]) continue;   // in-place only
    const Dtype slope = layers_[i + 1]->layer_param().relu_param().negative_slope();
    if (layers_[i]->FuseReLU(slope)) { fused_away_[i + 1] = true; fused_producers_[i + 1].push_back((int)i); }
  }
  // Convolution (+ fused ReLU) -> [Split ->] Pooling(MAX 2x2 / 2): the convolution's epilogue also writes the pooled blob
  // (pool1..pool6 of the VGG trunk); the Pooling layer becomes a no-op.  The convolution's own top is still written.
  for (size_t i = 0; i < layers_.size(); ++i) {
    if (string(layers_[i]->type()) != "Pooling" || !layers_[i]->IsMaxPool2x2()) continue;
    if (bottom_vecs_[i].size() != 1 || top_vecs_[i].size() != 1) continue;
    Blob<Dtype>* src = bottom_vecs_[i][0];
    int prod = -1;
    for (int hop = 0; hop < 2 && prod < 0; ++hop) {          // look through one Split
      int l = -1;
      for (size_t k = 0; k < i; ++k)
        for (Blob<Dtype>* t : top_vecs_[k]) if (t == src) l = (int)k;      // last writer before the pooling layer
      if (l < 0) break;
      while (l > 0 && fused_away_[l]) --l;                                  // an in-place ReLU that was fused away
      if (string(layers_[l]->type()) == "Split") { src = bottom_vecs_[l][0]; continue; }
      if (string(layers_[l]->type()) == "Convolution" && top_vecs_[l].size() == 1 && top_vecs_[l][0] == src) prod = l;
      break;
    }
    if (prod < 0) continue;
    // nothing between the convolution and the pooling layer may rewrite the blob (only its fused in-place ReLU does)
    bool clean = true;
    for (size_t k = prod + 1; k < i && clean; ++k)
      for (Blob<Dtype>* t : top_vecs_[k]) if (t == top_vecs_[prod][0] && !fused_away_[k]) clean = false;
    if (!clean) continue;
    if (layers_[prod]->FusePool2x2(top_vecs_[i][0])) { fused_away_[i] = true; fused_producers_[i].push_back(prod); }
  }
  // Channel Concat whose bottoms all come straight from ROIPooling layers (roi_pool_org + roi_pool_ctx -> roi_pool): the
  // producers write their channel window of the concatenated blob and the copy layer disappears (concat_layer.cu:28-46
  // moves R x 1024 x 49 floats per image otherwise).  The individual ROIPooling tops are then not materialised.
  for (size_t i = 0; i < layers_.size(); ++i) {
    if (string(layers_[i]->type()) != "Concat" || bottom_vecs_[i].size() < 2 || top_vecs_[i].size() != 1) continue;
    const ConcatParameter cp = layers_[i]->layer_param().concat_param();
    if ((cp.has_concat_dim() ? (int)cp.concat_dim() : cp.axis()) != 1) continue;
    vector<int> producers;
    int c_total = 0;
    bool ok = true;
    for (size_t b = 0; b < bottom_vecs_[i].size() && ok; ++b) {
      int prod = -1, consumers = 0;
      for (size_t l = 0; l < layers_.size(); ++l) {
        for (Blob<Dtype>* t : top_vecs_[l]) if (t == bottom_vecs_[i][b]) prod = (int)l;
        for (Blob<Dtype>* bb : bottom_vecs_[l]) if (bb == bottom_vecs_[i][b]) ++consumers;
      }
      ok = prod >= 0 && consumers == 1 && string(layers_[prod]->type()) == "ROIPooling" && top_vecs_[prod].size() == 1;
      producers.push_back(prod);
      if (ok) c_total += bottom_vecs_[i][b]->channels();
    }
    if (!ok) continue;
    int off = 0;
    for (size_t b = 0; b < producers.size(); ++b) {
      CHECK(layers_[producers[b]]->SetOutputWindow(top_vecs_[i][0], c_total, off));
      Redirect rd;
      rd.target_blob = top_id_vecs_[i][0]; rd.c_total = c_total; rd.c_offset = off;
      redirect_[bottom_id_vecs_[i][b]] = rd;
      redirect_dirty_[bottom_id_vecs_[i][b]] = false;
      off += bottom_vecs_[i][b]->channels();
    }
    fused_away_[i] = true;
    fused_producers_[i] = producers;
    // two poolings of the same map and ROIs: one launch (ROIPoolingLayer::PairWith)
    auto source = [&](int blob) {      // through Split layers (their tops share the bottom's data) to the blob that holds the data
      for (bool moved = true; moved;) {
        moved = false;
        for (size_t l = 0; l < layers_.size() && !moved; ++l)
          if (string(layers_[l]->type()) == "Split")
            for (int t : top_id_vecs_[l])
              if (t == blob) { blob = bottom_id_vecs_[l][0]; moved = true; break; }
      }
      return blob;
    };
    if (producers.size() == 2 && bottom_id_vecs_[producers[0]].size() == 2 && bottom_id_vecs_[producers[1]].size() == 2 &&
        source(bottom_id_vecs_[producers[0]][0]) == source(bottom_id_vecs_[producers[1]][0]) &&
        source(bottom_id_vecs_[producers[0]][1]) == source(bottom_id_vecs_[producers[1]][1])) {
      const int first = std::min(producers[0], producers[1]), second = std::max(producers[0], producers[1]);
      ROIPoolingLayer<Dtype>* fl = static_cast<ROIPoolingLayer<Dtype>*>(layers_[first].get());
      if (fl->PairWith(static_cast<ROIPoolingLayer<Dtype>*>(layers_[second].get()))) {
        // the concatenated blob has ONE reader and it is a Convolution (roi_pool -> roi_c1): the pooling is deferred to that layer's
        // input stage (ConvolutionLayer::FuseRoiPoolInput); the blob itself is written on demand (MaterializeBlob)
        const int tb = top_id_vecs_[i][0];
        int reader = -1, readers = 0;
        for (size_t l = 0; l < layers_.size(); ++l)
          for (int bb : bottom_id_vecs_[l]) if (bb == tb && !fused_away_[l]) { reader = (int)l; ++readers; }
        bool is_output = false;
        for (int ob : net_output_blob_indices_) is_output = is_output || ob == tb;
        ConvolutionLayer<Dtype>* conv = readers == 1 ? dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[reader].get()) : nullptr;
        if (conv && !is_output && reader > second) {
          fl->set_deferred(true);
          conv->FuseRoiPoolInput(fl);
          DeferredPool dp;
          dp.first_layer = first; dp.conv_layer = reader; dp.blob = tb;
          deferred_pools_.push_back(dp);
          // The pooling's maps depend on the feature blob only.  When that blob is complete before the BoxOutput layer that selects
          // the ROIs runs (conv4_3 of the plain nets; not the "-2x" nets' Deconvolution, which follows BoxOutput), they are built
          // UNDER BoxOutput's host round trip: enqueued behind BoxOutput's kernels and in front of the host's wait, they keep the device busy while the host
          // takes R, reshapes and launches the sub-net (29 us of idle device per 7s-576 frame otherwise: profiles/r06_kernel_gaps.txt).
          const int feat_blob = source(bottom_id_vecs_[first][0]), rois_blob = source(bottom_id_vecs_[first][1]);
          int feat_layer = -1, rois_layer = -1;
          for (size_t l = 0; l < layers_.size(); ++l)
            for (int t : top_id_vecs_[l]) {
              if (t == feat_blob) feat_layer = (int)l;      // (the LAST writer: in-place layers re-issue the blob)
              if (t == rois_blob) rois_layer = (int)l;
            }
          if (feat_layer >= 0 && rois_layer > feat_layer && string(layers_[rois_layer]->type()) == "BoxOutput") {
            const int conv_layer = reader, pool_layer = first;
            static_cast<BoxOutputLayer<Dtype>*>(layers_[rois_layer].get())->set_before_sync([this, conv_layer, pool_layer]() {
              if (!fusion_ || last_end_ < conv_layer) return;      // (the sub-net is not part of this call)
              static_cast<ConvolutionLayer<Dtype>*>(layers_[conv_layer].get())->PrebuildRoiMaps(bottom_vecs_[pool_layer][0]);
            });
          }
        }
      }
    }
  }
  // Chains of convolutions (ConvolutionLayer::ChainTo): the top of convolution i is read by exactly one layer that still runs -- a
  // 3x3 convolution j > i with nothing running in between (the in-place ReLU is fused away) -- and is not a net output.
  if (const char* e = getenv("MSCNN_NO_CHAIN")) chain_fusion_ = !(e[0] && e[0] != '0');
  for (size_t i = 0; i < layers_.size(); ++i) {
    ConvolutionLayer<Dtype>* ci = fused_away_[i] ? nullptr : dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
    if (!ci || top_id_vecs_[i].size() != 1) continue;
    const int tb = top_id_vecs_[i][0];
    int reader = -1, readers = 0;
    bool only_relu = true;
    for (size_t l = 0; l < layers_.size(); ++l) {
      if (l == i) continue;
      for (int bb : bottom_id_vecs_[l]) {
        if (bb != tb) continue;
        if (fused_away_[l]) only_relu = only_relu && string(layers_[l]->type()) == "ReLU";
        else { reader = (int)l; ++readers; }
      }
    }
    bool is_output = false;
    for (int ob : net_output_blob_indices_) is_output = is_output || ob == tb;
    if (readers == 0 && !is_output && !redirect_.count(tb)) {
      // read by fused-away layers only (its ReLU, the 2x2 pooling in this convolution's epilogue): the top need not be written
      bool pooled_here = false;
      for (size_t l = 0; l < layers_.size(); ++l)
        if (fused_away_[l] && string(layers_[l]->type()) == "Pooling")
          for (int p : fused_producers_[l]) pooled_here = pooled_here || p == (int)i;
      if (pooled_here) {
        ChainPair cp;
        cp.producer = (int)i; cp.consumer = -1; cp.blob = tb;
        chain_pairs_.push_back(cp);
      }
      continue;
    }
    if (readers != 1 || !only_relu || is_output || reader <= (int)i || redirect_.count(tb)) continue;
    ConvolutionLayer<Dtype>* cj = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[reader].get());
    bool adjacent = cj != nullptr;
    for (int l = (int)i + 1; l < reader && adjacent; ++l) adjacent = fused_away_[l];
    if (!adjacent) continue;
    ci->ChainTo(cj);
    ChainPair cp;
    cp.producer = (int)i; cp.consumer = reader; cp.blob = tb;
    chain_pairs_.push_back(cp);
  }
}

template <typename Dtype>
int Net<Dtype>::SplitSource(int blob) const {
  for (bool moved = true; moved;) {
    moved = false;
    for (size_t l = 0; l < layers_.size() && !moved; ++l)
      if (string(layers_[l]->type()) == "Split")
        for (int t : top_id_vecs_[l])
          if (t == blob) { blob = bottom_id_vecs_[l][0]; moved = true; break; }
  }
  return blob;
}

template <typename Dtype>
void Net<Dtype>::MaterializePendingReadersOf(const string& blob_name) const {
  if (!has_blob(blob_name)) return;
  {
    // blobs inside convolution chains that the last Forward did not write and whose value hangs on this blob: written now, from the
    // bottoms their layers were given (the reference's blobs hold exactly that)
    std::vector<int> affected(1, blob_names_index_.find(blob_name)->second);
    for (size_t a = 0; a < affected.size(); ++a)
      for (size_t k = 0; k < chain_pairs_.size(); ++k) {
        const int i = chain_pairs_[k].producer;
        if (!static_cast<ConvolutionLayer<Dtype>*>(layers_[i].get())->top_stale()) continue;
        for (int bb : bottom_id_vecs_[i])
          if ((bb == affected[a] || SplitSource(bb) == SplitSource(affected[a])) &&
              std::find(affected.begin(), affected.end(), chain_pairs_[k].blob) == affected.end())
            affected.push_back(chain_pairs_[k].blob);
      }
    for (size_t a = 1; a < affected.size(); ++a) MaterializeBlob(affected[a]);
  }
  if (deferred_pools_.empty()) return;
  const int src = SplitSource(blob_names_index_.find(blob_name)->second);
  for (size_t k = 0; k < deferred_pools_.size(); ++k) {
    const int fl = deferred_pools_[k].first_layer;
    for (int bb : bottom_id_vecs_[fl])
      if (SplitSource(bb) == src) static_cast<ROIPoolingLayer<Dtype>*>(layers_[fl].get())->Materialize();
  }
}

template <typename Dtype>
void Net<Dtype>::MaterializeStale() const {
  for (size_t k = 0; k < chain_pairs_.size(); ++k) MaterializeBlob(chain_pairs_[k].blob);
}

template <typename Dtype>
void Net<Dtype>::MaterializePending() const {
  for (size_t k = 0; k < deferred_pools_.size(); ++k)
    static_cast<ROIPoolingLayer<Dtype>*>(layers_[deferred_pools_[k].first_layer].get())->Materialize();
}

template <typename Dtype>
void Net<Dtype>::MaterializeBlob(int blob_id) const {
  // the blob between two members of a convolution chain that the last Forward did not write: re-run its producer unchained, from a
  // bottom that is brought up to date the same way first (same kernels as the chain: bit-identical)
  for (size_t k = 0; k < chain_pairs_.size(); ++k)
    if (chain_pairs_[k].blob == blob_id) {
      const int i = chain_pairs_[k].producer;
      ConvolutionLayer<Dtype>* c = static_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
      if (c->top_stale()) {
        for (int bb : bottom_id_vecs_[i]) MaterializeBlob(bb);
        c->ForwardUnchained(bottom_vecs_[i], top_vecs_[i]);
      }
    }
  typename std::map<int, Redirect>::const_iterator it = redirect_.find(blob_id);
  // a deferred ROIPooling pair's blob (asked for directly, or as the home of a redirected roi_pool_org / roi_pool_ctx): write it now
  for (size_t k = 0; k < deferred_pools_.size(); ++k)
    if (deferred_pools_[k].blob == blob_id || (it != redirect_.end() && it->second.target_blob == deferred_pools_[k].blob))
      static_cast<ROIPoolingLayer<Dtype>*>(layers_[deferred_pools_[k].first_layer].get())->Materialize();
  if (it == redirect_.end() || !redirect_dirty_[blob_id]) return;
  const Redirect& rd = it->second;
  const Blob<Dtype>* src = blobs_[rd.target_blob].get();
  Blob<Dtype>* dst = blobs_[blob_id].get();
  const int R = src->num(), inner = src->count(2), c = dst->channels();
  dst->Reshape(R, c, src->height(), src->width());
  if (R > 0)       // one strided D2D copy: R rows of c * inner floats out of rows of c_total * inner floats
    HIP_CHECK(hipMemcpy2DAsync(dst->mutable_gpu_data(), sizeof(Dtype) * c * inner, src->gpu_data() + (size_t)rd.c_offset * inner,
                               sizeof(Dtype) * rd.c_total * inner, sizeof(Dtype) * c * inner, R, hipMemcpyDeviceToDevice,
                               (hipStream_t)Caffe::stream()));
  redirect_dirty_[blob_id] = false;
}

template <typename Dtype>
vector<int> Net<Dtype>::CalibrateNumerics(double tol) {
  vector<int> switched;
  for (size_t i = 0; i < layers_.size(); ++i) {
    calib_err_[i] = 0.0;
    ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
    if (!c || c->algo() == 1 || c->algo() == 4) continue;      // direct already / fp16 mode has its own tolerance policy
    for (int bb : bottom_id_vecs_[i]) MaterializeBlob(bb);      // (blobs inside a convolution chain are written on demand)
    for (int tb : top_id_vecs_[i]) MaterializeBlob(tb);
    calib_err_[i] = c->ErrorAgainstDirect(bottom_vecs_[i], top_vecs_[i]);
    if (!(calib_err_[i] <= tol)) {      // (NaN counts as a failure)
      LOG(WARNING) << "layer " << layer_names_[i] << ": Winograd result off the direct sum by " << calib_err_[i] << " > " << tol
                   << " on the calibration input: using the direct kernel";
      c->set_algo(1);
      c->set_calibrated_direct(true);
      switched.push_back((int)i);
    }
  }
  return switched;
}

template <typename Dtype>
void Net<Dtype>::SetAutoCalibrate(double tol) {
  CHECK_GE(tol, 0.0);
  for (size_t i = 0; i < layers_.size(); ++i)
    if (ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get())) c->set_selfcheck(tol);
  auto_tol_ = tol;
  // both checks off: give the checks' device scratch back (a direct plan's workspace + a copy of the largest checked top)
  if (tol == 0.0 && watch_period_ == 0) ConvolutionLayer<Dtype>::ReleaseCheckScratch();
}

template <typename Dtype>
void Net<Dtype>::SetNumericsWatch(int period, double tol) {
  WatchCollect(true);      // (a verdict still out is judged by the tolerance it was started under)
  watch_period_ = period; watch_tol_ = tol; watch_frame_ = 0;
  if (period == 0 && auto_tol_ == 0.0) ConvolutionLayer<Dtype>::ReleaseCheckScratch();
}

// The numerics watch (round 6 form: no check inside a frame's latency).  On a watch frame ONE Winograd layer -- round robin -- has its
// bottom and top written as blobs although it stays in its convolution chain (the producers write y beside the next layer's planes
// / the pooled map: ConvolutionLayer::set_keep_top), and one band of it (round robin as well) is recomputed with
// the direct kernel BEHIND the frame on the same stream (ConvolutionLayer::BeginBandCheck: a copy of the band, a direct convolution of
// that band -- ~30 us of work --, two reductions, 4 bytes to pinned memory, an event).  Nobody waits for it: the verdict is collected at
// the end of a later whole forward (or when somebody asks for the watch's state) and, when the layer strayed, it runs the direct
// kernel from the frame after that.
template <typename Dtype>
int Net<Dtype>::NextWatchLayer() const {
  const int L = (int)layers_.size();
  for (int k = 0; k < L; ++k) {
    const int i = (watch_next_ + k) % L;
    ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
    if (!c || fused_away_[i] || c->algo() == 1 || c->algo() == 4 || std::strncmp(c->kernel_name(), "winograd", 8) != 0) continue;
    return i;
  }
  return -1;
}

template <typename Dtype>
void Net<Dtype>::WatchCollect(bool wait) {
  if (watch_pending_ < 0) return;
  const int i = watch_pending_;
  ConvolutionLayer<Dtype>* c = static_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
  double e = 0.0;
  const int r = c->PollBandCheck(&e, wait);
  if (r == 1) return;
  watch_pending_ = -1;
  if (r != 2) return;
  calib_err_[i] = e;
  if (!(e <= watch_tol_) && c->algo() != 1) {      // (NaN counts as a failure)
    LOG(WARNING) << "layer " << layer_names_[i] << ": Winograd result off the direct sum by " << e << " > " << watch_tol_
                 << " on a live frame: using the direct kernel from the next frame on";
    c->set_algo(1);
    c->set_calibrated_direct(true);
    watch_switched_.push_back(i);
  }
}

template <typename Dtype>
bool Net<Dtype>::HandoffEventPending() {
  const unsigned long long ev = mscnn_wgemm_handoff_event();
  if (ev == handoff_seen_) return false;
  handoff_seen_ = ev;
  ++handoff_errors_;
  mscnn_wgemm_force_whole_tiles(1);
  LOG(WARNING) << "a stream-K hand-off of the plane-GEMM kernel timed out (a workgroup of the persistent grid was not co-resident): "
                  "whole-tile scheduling from now on, running the frame again";
  return true;
}

// Called where the caller has just synchronised the stream.  No event: everything this net launched so far is known good (the ranges run
// since are forgotten).  An event: EVERY range run since the last known-good point is suspect -- a caller that forwards 0 .. k and then
// k + 1 .. end and only then synchronises may have the poisoned tile in the first range -- so the re-run goes from the smallest start
// seen since that point to the end of the last range.
template <typename Dtype>
bool Net<Dtype>::HandoffRecover() {
  if (!HandoffEventPending()) { suspect_start_ = -1; return false; }
  const int from = suspect_start_ >= 0 ? suspect_start_ : last_start_, to = last_end_;
  suspect_start_ = -1;
  if (to >= from) ForwardFromTo(from, to);
  return true;
}

template <typename Dtype>
Dtype Net<Dtype>::ForwardFromTo(int start, int end) {
  CHECK_GE(start, 0);
  CHECK_LT(end, (int)layers_.size());
  if (suspect_start_ < 0) {
    // nothing of this net is in flight unchecked: hand-off events raised so far belong to other nets on this device (the status word is
    // per device), not to the frame that begins here
    handoff_seen_ = mscnn_wgemm_handoff_event();
    suspect_start_ = start;
  } else if (start < suspect_start_) suspect_start_ = start;
  last_start_ = start; last_end_ = end;
  ++forward_count_;
  bool handoff_restart = false;
  // a watch frame: the verdict still out (if any) is taken first -- its kernels ran in front of everything this call will enqueue --,
  // then the layer to look at in this frame is chosen; the producers of its bottom and top keep their top blobs below
  const bool whole = start == 0 && end == (int)layers_.size() - 1;
  int watch_layer = -1;
  if (watch_period_ > 0 && whole && (watch_frame_ + 1) % watch_period_ == 0) {
    WatchCollect(true);
    watch_layer = NextWatchLayer();
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timing_) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); }
  for (size_t i = 0; i < layers_.size(); ++i)      // (a paired ROIPooling skips only inside the call in which its partner ran)
    if (string(layers_[i]->type()) == "ROIPooling") static_cast<ROIPoolingLayer<Dtype>*>(layers_[i].get())->set_skip(false);
  // a deferred ROI pooling still pending from an earlier call reads blobs this range is about to rewrite without re-running the
  // pooling itself: write its blob first, from the bottoms it was given (the reference's blob would hold exactly that)
  for (size_t k = 0; k < deferred_pools_.size(); ++k) {
    if (start <= deferred_pools_[k].first_layer && end < deferred_pools_[k].first_layer)
      static_cast<ROIPoolingLayer<Dtype>*>(layers_[deferred_pools_[k].first_layer].get())->Materialize();
    // maps built early for an earlier frame's feature blob are never carried into this call (BoxOutput's hook rebuilds them)
    static_cast<ConvolutionLayer<Dtype>*>(layers_[deferred_pools_[k].conv_layer].get())->InvalidateRoiMaps();
  }
  {
    // split-fp16 convolutions take the bound of their input from the producing convolution when it runs in this same call
    // from the top (slots zeroed here, once); a partial range makes them measure it themselves
    bool any = false;
    for (size_t i = 0; i < layers_.size(); ++i)
      if (ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get())) c->set_amax_wanted(false);
    for (size_t i = 0; i < layers_.size(); ++i) {
      if (InnerProductLayer<Dtype>* ip = dynamic_cast<InnerProductLayer<Dtype>*>(layers_[i].get())) {
        ip->set_amax_trusted(start == 0);
        if (ip->x3() && amax_src_[i] >= 0 && start == 0) {
          static_cast<ConvolutionLayer<Dtype>*>(layers_[amax_src_[i]].get())->set_amax_wanted(true);
          any = true;
        }
      }
      ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get());
      if (!c) continue;
      c->set_amax_trusted(start == 0);
      if (c->algo() == MSCNN_CONV_ALGO_WINO_F3_X3 && amax_src_[i] >= 0 && start == 0) {
        static_cast<ConvolutionLayer<Dtype>*>(layers_[amax_src_[i]].get())->set_amax_wanted(true);
        any = true;
      }
    }
    if (any && !amax_slots_) {
      HIP_CHECK(hipMalloc(&amax_slots_, sizeof(unsigned) * MSCNN_AMAX_SLOTS * layers_.size()));
      unsigned* slots = static_cast<unsigned*>(amax_slots_);
      for (size_t i = 0; i < layers_.size(); ++i)
        if (ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get())) {
          const int b = amax_src_[i];
          c->set_amax_io(b >= 0 ? static_cast<const ConvolutionLayer<Dtype>*>(layers_[b].get()) : nullptr,
                         b >= 0 ? slots + (size_t)b * MSCNN_AMAX_SLOTS : nullptr, slots + i * MSCNN_AMAX_SLOTS);
        } else if (InnerProductLayer<Dtype>* ip = dynamic_cast<InnerProductLayer<Dtype>*>(layers_[i].get())) {
          const int b = amax_src_[i];
          if (b >= 0) ip->set_amax_in(static_cast<const ConvolutionLayer<Dtype>*>(layers_[b].get()), slots + (size_t)b * MSCNN_AMAX_SLOTS);
        }
    }
    if (any) HIP_CHECK(hipMemsetAsync(amax_slots_, 0, sizeof(unsigned) * MSCNN_AMAX_SLOTS * layers_.size(), (hipStream_t)Caffe::stream()));
  }
  // convolution chains: a pair is live when both members run in this call; a consumer that runs WITHOUT its producer reads a blob
  // the last whole Forward may not have written
  for (size_t k = 0; k < chain_pairs_.size(); ++k) {
    const ChainPair& cp = chain_pairs_[k];
    ConvolutionLayer<Dtype>* c = static_cast<ConvolutionLayer<Dtype>*>(layers_[cp.producer].get());
    // (the layer under watch needs its bottom and its top as blobs: the producer of either writes y beside the planes / the pooled map)
    c->set_keep_top(watch_layer >= 0 && (cp.producer == watch_layer || cp.consumer == watch_layer));
    if (cp.consumer < 0) {      // (a top read by the fused pooling only)
      c->set_pool_only_live(fusion_ && chain_fusion_ && start <= cp.producer && cp.producer <= end);
      if (!(start <= cp.producer && cp.producer <= end)) MaterializeBlob(cp.blob);
      continue;
    }
    const bool both = start <= cp.producer && cp.consumer <= end;
    c->set_chain_live(fusion_ && chain_fusion_ && both);
    static_cast<ConvolutionLayer<Dtype>*>(layers_[cp.consumer].get())->set_chain_live(false);
    // a blob the last Forward left unwritten whose producer does not run in this call: write it now, from the bottom its layer was
    // given -- this call may rewrite that bottom, or run the consumer from the blob
    if (!(start <= cp.producer && cp.producer <= end)) MaterializeBlob(cp.blob);
  }
  for (size_t k = 0; k < chain_pairs_.size(); ++k) {      // (a consumer that is itself a producer: its own live mark, set above, stands)
    const ChainPair& cp = chain_pairs_[k];
    if (cp.consumer < 0) continue;
    const bool both = start <= cp.producer && cp.consumer <= end;
    static_cast<ConvolutionLayer<Dtype>*>(layers_[cp.producer].get())->set_chain_live(fusion_ && chain_fusion_ && both);
  }
  for (int i = start; i <= end; ++i) {
    bool run = !fused_away_[i];
    if (fused_away_[i]) {
      // A fused-away layer's work is done by its producer(s).  When a partial range starts after one of them (the reference
      // would recompute this layer from its bottoms, net.cpp:544-555), run the layer itself -- the un-fused kernels are
      // bit-identical.  Its bottoms may be tops that were redirected into a Concat: bring them up to date first.
      for (size_t k = 0; k < fused_producers_[i].size(); ++k)
        if (fused_producers_[i][k] < start) run = true;
      if (run)
        for (size_t b = 0; b < bottom_id_vecs_[i].size(); ++b) MaterializeBlob(bottom_id_vecs_[i][b]);
      else
        layers_[i]->Reshape(bottom_vecs_[i], top_vecs_[i]);      // shapes follow the bottoms exactly as Layer::Forward would
    }
    if (!run) { layer_ms_[i] = 0.f; continue; }
    if (timing_) HIP_CHECK(hipEventRecord(e0, (hipStream_t)Caffe::stream()));
    layers_[i]->Forward(bottom_vecs_[i], top_vecs_[i]);
    for (size_t t = 0; t < top_id_vecs_[i].size(); ++t)
      if (redirect_.count(top_id_vecs_[i][t])) redirect_dirty_[top_id_vecs_[i][t]] = true;
    if (timing_) {
      HIP_CHECK(hipEventRecord(e1, (hipStream_t)Caffe::stream()));
      HIP_CHECK(hipEventSynchronize(e1));
      HIP_CHECK(hipEventElapsedTime(&layer_ms_[i], e0, e1));
    }
    // BoxOutput has just synchronised the stream for its row count: everything in front of it has run.  A hand-off that timed out
    // there left NaN tiles whose scores BoxOutput drops without a trace (NaN >= fg_thr is false, box_output_layer.cpp:128): look at
    // the status word now (a plain load of pinned memory) and run the range again on whole tiles (once: nothing is split afterwards;
    // all chains / deferred poolings in front of this layer have completed, so their per-call state is clean).
    if (std::strcmp(layers_[i]->type(), "BoxOutput") == 0) {
      if (HandoffEventPending()) {
        handoff_restart = true;    // leave the loop: the call's own tear-down below, then the whole call again (its set-up included:
        break;                     // the max |x| slots a NaN tile has already raised are zeroed again, chain / skip marks re-made)
      }
      suspect_start_ = i + 1;      // the stream was synchronised and nothing was reported: only what follows is still unchecked
    }
  }
  if (timing_) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  if (watch_layer >= 0 && !handoff_restart) {
    ConvolutionLayer<Dtype>* c = static_cast<ConvolutionLayer<Dtype>*>(layers_[watch_layer].get());
    watch_next_ = (watch_layer + 1) % (int)layers_.size();
    if (watch_next_ <= watch_layer) ++watch_band_;          // one band per trip round the layers
    if (c->BeginBandCheck(bottom_vecs_[watch_layer], top_vecs_[watch_layer], watch_band_)) {
      watch_pending_ = watch_layer;
      ++watch_checks_;
    }
  }
  // a paired ROIPooling's skip mark must not outlive the call in which its partner ran (a range that ends between the two, then a
  // direct Layer::Forward on the second: it has to pool, not skip)
  for (size_t i = 0; i < layers_.size(); ++i)
    if (string(layers_[i]->type()) == "ROIPooling") static_cast<ROIPoolingLayer<Dtype>*>(layers_[i].get())->set_skip(false);
  for (size_t k = 0; k < chain_pairs_.size(); ++k) {
    static_cast<ConvolutionLayer<Dtype>*>(layers_[chain_pairs_[k].producer].get())->set_chain_live(false);
    static_cast<ConvolutionLayer<Dtype>*>(layers_[chain_pairs_[k].producer].get())->set_pool_only_live(false);
    static_cast<ConvolutionLayer<Dtype>*>(layers_[chain_pairs_[k].producer].get())->set_keep_top(false);
    if (chain_pairs_[k].consumer >= 0) static_cast<ConvolutionLayer<Dtype>*>(layers_[chain_pairs_[k].consumer].get())->set_chain_live(false);
  }
  // books of the layers' own first-forward checks (safe-by-default numerics: ConvolutionLayer::set_selfcheck)
  for (int i = start; i <= end; ++i)
    if (ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get())) {
      double e = 0.0;
      bool fell = false;
      if (c->take_selfcheck(&e, &fell)) {
        calib_err_[i] = e;
        ++auto_checks_;
        if (fell) auto_switched_.push_back(i);
      }
    }
  // The max |x| slots a split-fp16 layer takes from its producer are only valid inside this call: a later Layer::Forward called
  // directly on a layer (caffe time, user code, the boundary tests) must measure its bottom itself instead of trusting the slots of
  // the frame that went through here.
  for (size_t i = 0; i < layers_.size(); ++i) {
    if (ConvolutionLayer<Dtype>* c = dynamic_cast<ConvolutionLayer<Dtype>*>(layers_[i].get())) c->set_amax_trusted(false);
    else if (InnerProductLayer<Dtype>* ip = dynamic_cast<InnerProductLayer<Dtype>*>(layers_[i].get())) ip->set_amax_trusted(false);
  }
  if (handoff_restart) {
    // (whole-tile scheduling is forced by now: the second pass cannot raise the event again, so this recursion is one level deep.  The
    // restart covers every range since the last known-good point, like HandoffRecover)
    const int from = suspect_start_ >= 0 && suspect_start_ < start ? suspect_start_ : start;
    --forward_count_;
    return ForwardFromTo(from, end);
  }
  if (watch_period_ > 0 && whole) {
    ++watch_frame_;
    if (watch_layer < 0) WatchCollect(false);      // (never blocks: a verdict that is not in yet is taken by a later frame)
  }
  return 0;
}

template <typename Dtype>
const vector<Blob<Dtype>*>& Net<Dtype>::Forward(Dtype* loss) {
  const Dtype l = ForwardFromTo(0, (int)layers_.size() - 1);
  if (loss != NULL) *loss = l;
  return net_output_blobs_;
}

template <typename Dtype>
void Net<Dtype>::Reshape() {
  for (size_t i = 0; i < layers_.size(); ++i) layers_[i]->Reshape(bottom_vecs_[i], top_vecs_[i]);
}

template <typename Dtype>
void Net<Dtype>::MarkWeightsChanged() {
  // (the caller has ALREADY written the parameter blobs: blobs still unwritten would be re-created with the new weights -- code that
  // writes parameters directly and wants the old frame's intermediate blobs calls MaterializeStale() first, as the C ABI's setter does)
  for (size_t i = 0; i < layers_.size(); ++i) layers_[i]->OnWeightsChanged();
}

template <typename Dtype>
bool Net<Dtype>::has_blob(const string& blob_name) const { return blob_names_index_.find(blob_name) != blob_names_index_.end(); }

template <typename Dtype>
const shared_ptr<Blob<Dtype> > Net<Dtype>::blob_by_name(const string& blob_name) const {
  shared_ptr<Blob<Dtype> > blob_ptr;
  if (has_blob(blob_name)) {
    const int id = blob_names_index_.find(blob_name)->second;
    MaterializeBlob(id);        // no-op unless the blob's producer writes into a fused Concat's top
    blob_ptr = blobs_[id];
  } else LOG(WARNING) << "Unknown blob name " << blob_name;
  return blob_ptr;
}

template <typename Dtype>
bool Net<Dtype>::has_layer(const string& layer_name) const { return layer_names_index_.find(layer_name) != layer_names_index_.end(); }

template <typename Dtype>
const shared_ptr<Layer<Dtype> > Net<Dtype>::layer_by_name(const string& layer_name) const {
  shared_ptr<Layer<Dtype> > layer_ptr;
  if (has_layer(layer_name)) layer_ptr = layers_[layer_names_index_.find(layer_name)->second];
  else LOG(WARNING) << "Unknown layer name " << layer_name;
  return layer_ptr;
}

// ---- .caffemodel (binary NetParameter) reader: varint / length-delimited / packed float only ------------------
// caffe.proto: NetParameter{ layer = 100, layers = 2 (V1) } ; LayerParameter{ name = 1, blobs = 7 } ; V1LayerParameter{ name = 4, blobs = 6 } ;
// BlobProto{ num=1 channels=2 height=3 width=4 data=5 (packed float) shape=7 } ; BlobShape{ dim=1 (packed int64) }
namespace {
// Every advance is bounds-checked: a truncated or corrupt file is a CHECK failure, never a read past the buffer.
struct Reader {
  const unsigned char* p; const unsigned char* end;
  bool ok() const { return p < end; }
  size_t left() const { return (size_t)(end - p); }
  void need(size_t n) const { CHECK_LE(n, left()) << "truncated caffemodel"; }
  unsigned long long varint() {
    unsigned long long v = 0; int shift = 0;
    for (;;) {
      need(1);
      const unsigned char b = *p++;
      CHECK_LT(shift, 64) << "malformed varint in caffemodel";
      v |= (unsigned long long)(b & 0x7f) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
    }
    return v;
  }
  Reader sub() { const size_t n = (size_t)varint(); need(n); Reader r{p, p + n}; p += n; return r; }
  void skip(int wire) {
    if (wire == 0) varint();
    else if (wire == 1) { need(8); p += 8; }
    else if (wire == 2) { const size_t n = (size_t)varint(); need(n); p += n; }
    else if (wire == 5) { need(4); p += 4; }
    else LOG(FATAL) << "unsupported wire type " << wire << " in caffemodel";
  }
};
struct ParsedBlob { vector<int> shape; vector<float> data; int legacy[4] = {0, 0, 0, 0}; bool has_legacy = false; };

// Blob::FromProto's shape rule (blob.cpp:448-470): explicit `shape`, else the legacy 4-D num/channels/height/width;
// Blob::ShapeEquals (blob.cpp:417-437) lets a legacy 4-D source match a lower-rank target whose leading dims are 1.
bool ShapeMatches(const ParsedBlob& pb, const vector<int>& target) {
  if (!pb.shape.empty()) return pb.shape == target;
  if (!pb.has_legacy) return target.empty();    // neither given: an empty BlobShape only equals a 0-axis blob
  if (target.size() > 4) return false;
  vector<int> t4(4, 1);
  for (size_t i = 0; i < target.size(); ++i) t4[4 - target.size() + i] = target[i];
  return t4[0] == pb.legacy[0] && t4[1] == pb.legacy[1] && t4[2] == pb.legacy[2] && t4[3] == pb.legacy[3];
}
}  // namespace

template <typename Dtype>
void Net<Dtype>::CopyTrainedLayersFrom(const string& trained_filename) {
  MaterializeStale();      // blobs a chained Forward left unwritten are the OLD weights' outputs
  if (trained_filename.size() >= 3 && trained_filename.compare(trained_filename.size() - 3, 3, ".h5") == 0) {   // net.cpp:788-795
    CopyTrainedLayersFromHDF5(trained_filename);
    return;
  }
  string bytes;
  CHECK(ReadFileToString(trained_filename, &bytes)) << "cannot read " << trained_filename;
  Reader net{(const unsigned char*)bytes.data(), (const unsigned char*)bytes.data() + bytes.size()};
  int copied = 0;
  while (net.ok()) {
    const unsigned long long key = net.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    // `layer` = 100 (LayerParameter: name = 1, blobs = 7) or the deprecated `layers` = 2 (V1LayerParameter: name = 4, blobs = 6;
    // caffe.proto:95, 1358-1361, 1405) -- the reference upgrades V1 nets on load (upgrade_proto.cpp UpgradeV1Net: name and
    // blobs carried over unchanged), and weights are matched by layer name only, so both feed the same copy below.
    const bool v1 = field == 2 && wire == 2;
    if (!(field == 100 && wire == 2) && !v1) { net.skip(wire); continue; }
    const int f_name = v1 ? 4 : 1, f_blobs = v1 ? 6 : 7;
    Reader lr = net.sub();
    string lname;
    vector<ParsedBlob> pblobs;
    while (lr.ok()) {
      const unsigned long long k = lr.varint();
      const int f = (int)(k >> 3), w = (int)(k & 7);
      CHECK(!(v1 && f == 1 && w == 2)) << "V0 caffemodel (V1LayerParameter.layer = 1, 2013-era format) is not supported: upgrade it "
                                          "with the reference's upgrade_net_proto_binary";
      if (f == f_name && w == 2) { Reader s = lr.sub(); lname.assign((const char*)s.p, s.left()); }
      else if (f == f_blobs && w == 2) {
        Reader br = lr.sub();
        ParsedBlob pb;
        while (br.ok()) {
          const unsigned long long bk = br.varint();
          const int bf = (int)(bk >> 3), bw = (int)(bk & 7);
          if (bf >= 1 && bf <= 4 && bw == 0) { pb.legacy[bf - 1] = (int)br.varint(); pb.has_legacy = true; }
          else if (bf == 5 && bw == 2) {                       // data: packed float
            Reader d = br.sub();
            CHECK_EQ(d.left() % 4, 0u) << "packed float field of odd length in caffemodel";
            const size_t n = d.left() / 4, o = pb.data.size();
            pb.data.resize(o + n);
            if (n) memcpy(pb.data.data() + o, d.p, n * 4);
          } else if (bf == 5 && bw == 5) { br.need(4); float v; memcpy(&v, br.p, 4); br.p += 4; pb.data.push_back(v); }
          else if (bf == 8 && (bw == 2 || bw == 1)) {          // double_data (blob.cpp:472-476): narrowed to the net's float
            Reader d = bw == 2 ? br.sub() : Reader{br.p, br.p};
            if (bw == 1) { br.need(8); d = Reader{br.p, br.p + 8}; br.p += 8; }
            CHECK_EQ(d.left() % 8, 0u) << "packed double field of odd length in caffemodel";
            for (; d.ok(); d.p += 8) { double v; memcpy(&v, d.p, 8); pb.data.push_back((float)v); }
          } else if (bf == 7 && bw == 2) {
            Reader sr = br.sub();
            while (sr.ok()) {
              const unsigned long long sk = sr.varint();
              if ((sk >> 3) == 1 && (sk & 7) == 2) { Reader dr = sr.sub(); while (dr.ok()) pb.shape.push_back((int)dr.varint()); }
              else if ((sk >> 3) == 1 && (sk & 7) == 0) pb.shape.push_back((int)sr.varint());
              else sr.skip((int)(sk & 7));
            }
          } else br.skip(bw);
        }
        pblobs.push_back(pb);
      } else lr.skip(w);
    }
    if (!has_layer(lname)) { LOG(INFO) << "Ignoring source layer " << lname; continue; }   // net.cpp:760-764
    shared_ptr<Layer<Dtype> > layer = layer_by_name(lname);
    vector<shared_ptr<Blob<Dtype> > >& target = layer->blobs();
    if (pblobs.empty()) continue;
    CHECK_EQ(target.size(), pblobs.size()) << "Incompatible number of blobs for layer " << lname;
    for (size_t j = 0; j < target.size(); ++j) {
      // net.cpp:771-780: shape mismatch is fatal, with both shapes in the message
      CHECK(ShapeMatches(pblobs[j], target[j]->shape()) && (size_t)target[j]->count() == pblobs[j].data.size())
          << "Cannot copy param " << j << " weights from layer '" << lname << "'; shape mismatch (target " << target[j]->shape_string()
          << ", source holds " << pblobs[j].data.size() << " values)";
      memcpy(target[j]->mutable_cpu_data(), pblobs[j].data.data(), sizeof(float) * pblobs[j].data.size());   // blob.cpp:448-482
    }
    layer->OnWeightsChanged();
    ++copied;
  }
  LOG(INFO) << "Copied weights of " << copied << " layers from " << trained_filename;
}

// Net::CopyTrainedLayersFromHDF5, net.cpp:806-848: group "data" / <layer name> / "<blob index>" datasets.  The file is parsed by
// caffe/util/hdf5_lite.hpp (no libhdf5 in the deployment image); the calls below map one to one onto the reference's H5Gopen2 /
// hdf5_get_num_links / H5Lexists / hdf5_load_nd_dataset sequence.
template <typename Dtype>
void Net<Dtype>::CopyTrainedLayersFromHDF5(const string& trained_filename) {
  MaterializeStale();      // blobs a chained Forward left unwritten are the OLD weights' outputs
  string bytes;
  CHECK(ReadFileToString(trained_filename, &bytes)) << "Couldn't open " << trained_filename;
  h5lite::File f(bytes);
  CHECK(f.ok()) << f.error() << " (" << trained_filename << ")";
  uint64_t data_group = 0;
  bool found = false;
  CHECK(f.Find(f.root(), "data", &data_group, &found)) << f.error() << " (" << trained_filename << ")";
  CHECK(found) << "Error reading weights from " << trained_filename << ": no group \"data\"";
  vector<std::pair<string, uint64_t> > layers;
  CHECK(f.ListGroup(data_group, &layers)) << f.error() << " (" << trained_filename << ")";
  int copied = 0;
  for (size_t i = 0; i < layers.size(); ++i) {
    const string& source_layer_name = layers[i].first;
    if (!has_layer(source_layer_name)) { LOG(INFO) << "Ignoring source layer " << source_layer_name; continue; }   // :815-818
    shared_ptr<Layer<Dtype> > layer = layer_by_name(source_layer_name);
    vector<shared_ptr<Blob<Dtype> > >& target_blobs = layer->blobs();
    vector<std::pair<string, uint64_t> > params;
    CHECK(f.ListGroup(layers[i].second, &params)) << "Error reading weights from " << trained_filename << ": " << f.error();
    // :827-830 the source must not hold more params than the target layer
    CHECK_LE(params.size(), target_blobs.size()) << "Incompatible number of blobs for layer " << source_layer_name;
    for (size_t j = 0; j < target_blobs.size(); ++j) {
      std::ostringstream oss;
      oss << j;
      size_t k = 0;
      while (k < params.size() && params[k].first != oss.str()) ++k;
      // :836-845 a missing dataset is only tolerated for weight-shared params; this Net has no param sharing
      CHECK_LT(k, params.size()) << "Incompatible number of blobs for layer " << source_layer_name;
      h5lite::Dataset ds;
      CHECK(f.ReadDatasetInfo(params[k].second, &ds))
          << "Failed to read dataset " << oss.str() << " of layer " << source_layer_name << ": " << f.error();
      CHECK_LE((int)ds.dims.size(), kMaxBlobAxes);                                  // hdf5_load_nd_dataset(.., 0, kMaxBlobAxes, ..)
      // hdf5_load_nd_dataset_helper reshapes the target blob to the dataset's dims (util/hdf5.cpp:61-65).  The layers here keep
      // device-side packed copies sized at set-up, so a different element count is refused instead of silently re-shaping.
      CHECK_EQ((long long)target_blobs[j]->count(), ds.count())
          << "Cannot copy param " << j << " weights from layer '" << source_layer_name << "'; shape mismatch (target "
          << target_blobs[j]->shape_string() << ", source holds " << ds.count() << " values)";
      vector<int> dims(ds.dims.begin(), ds.dims.end());
      if (dims != target_blobs[j]->shape()) target_blobs[j]->Reshape(dims);
      CHECK(f.ReadFloats(ds, target_blobs[j]->mutable_cpu_data())) << f.error();   // H5LTread_dataset_float
    }
    layer->OnWeightsChanged();
    ++copied;
  }
  LOG(INFO) << "Copied weights of " << copied << " layers from " << trained_filename;
}

INSTANTIATE_CLASS(Net);

}  // namespace caffe
