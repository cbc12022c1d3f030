// Net<Dtype>: graph build + sequential executor (mirror of include/caffe/net.hpp / src/caffe/net.cpp for
// the TEST-phase forward path): legacy `input:` upgrade, automatic Split insertion with the reference's
// blob/layer naming, in-place tops, by-name lookup, outputs = un-consumed blobs in alphabetical order.
#ifndef MSCNN_CAFFE_NET_HPP_
#define MSCNN_CAFFE_NET_HPP_

#include <map>
#include <set>
#include <string>
#include <vector>

#include "caffe/blob.hpp"
#include "caffe/common.hpp"
#include "caffe/layer.hpp"
#include "caffe/proto/caffe_param.hpp"

namespace caffe {

// Applied by Net::Init before layers are created (mirrors UpgradeNetInput + InsertSplits).
void UpgradeNetInput(const NetParameter& in, vector<LayerParameter>* layers);
void InsertSplits(const vector<LayerParameter>& in, vector<LayerParameter>* out);

template <typename Dtype>
class Net {
 public:
  explicit Net(const NetParameter& param, Phase phase = TEST);
  explicit Net(const string& param_file, Phase phase);
  virtual ~Net() {}

  const vector<Blob<Dtype>*>& Forward(Dtype* loss = NULL);
  const vector<Blob<Dtype>*>& ForwardPrefilled(Dtype* loss = NULL) { return Forward(loss); }
  Dtype ForwardFromTo(int start, int end);
  void Reshape();

  // Loads weights by layer name from a binary NetParameter (.caffemodel), net.cpp:750-803.
  void CopyTrainedLayersFrom(const string& trained_filename);
  // Must be called after parameter blobs were written through mutable_cpu_data()/mutable_gpu_data().
  void MarkWeightsChanged();

  inline const string& name() const { return name_; }
  inline const vector<string>& layer_names() const { return layer_names_; }
  inline const vector<string>& blob_names() const { return blob_names_; }
  inline const vector<shared_ptr<Blob<Dtype> > >& blobs() const { return blobs_; }
  inline const vector<shared_ptr<Layer<Dtype> > >& layers() const { return layers_; }
  inline const vector<vector<Blob<Dtype>*> >& bottom_vecs() const { return bottom_vecs_; }
  inline const vector<vector<Blob<Dtype>*> >& top_vecs() const { return top_vecs_; }
  inline int num_inputs() const { return (int)net_input_blobs_.size(); }
  inline int num_outputs() const { return (int)net_output_blobs_.size(); }
  inline const vector<Blob<Dtype>*>& input_blobs() const { return net_input_blobs_; }
  inline const vector<Blob<Dtype>*>& output_blobs() const { return net_output_blobs_; }
  inline const vector<int>& output_blob_indices() const { return net_output_blob_indices_; }
  bool has_blob(const string& blob_name) const;
  const shared_ptr<Blob<Dtype> > blob_by_name(const string& blob_name) const;
  bool has_layer(const string& layer_name) const;
  const shared_ptr<Layer<Dtype> > layer_by_name(const string& layer_name) const;

  // --- extensions of this build ---
  // Conv/InnerProduct + in-place ReLU fusion is decided at construction: on by default, MSCNN_NO_FUSE=1 disables it.
  const vector<bool>& layer_fused_away() const { return fused_away_; }
  // Per-layer HIP-event timing of the last Forward when enabled (the `caffe time` loop, tools/caffe.cpp:380-400).
  void set_layer_timing(bool on) { timing_ = on; }
  const vector<float>& layer_ms() const { return layer_ms_; }

 protected:
  void Init(const NetParameter& param);
  void ApplyFusion();

  string name_;
  Phase phase_;
  vector<shared_ptr<Layer<Dtype> > > layers_;
  vector<string> layer_names_;
  std::map<string, int> layer_names_index_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;
  vector<string> blob_names_;
  std::map<string, int> blob_names_index_;
  vector<vector<Blob<Dtype>*> > bottom_vecs_;
  vector<vector<int> > bottom_id_vecs_;
  vector<vector<Blob<Dtype>*> > top_vecs_;
  vector<vector<int> > top_id_vecs_;
  vector<int> net_input_blob_indices_, net_output_blob_indices_;
  vector<Blob<Dtype>*> net_input_blobs_, net_output_blobs_;
  vector<bool> fused_away_;
  bool fusion_, timing_;
  vector<float> layer_ms_;
  DISABLE_COPY_AND_ASSIGN(Net);
};

}  // namespace caffe
#endif
