// Net<Dtype>: graph build + sequential executor (mirror of include/caffe/net.hpp / src/caffe/net.cpp for
// the TEST-phase forward path): legacy `input:` upgrade, automatic Split insertion with the reference's
// blob/layer naming, in-place tops, by-name lookup, outputs = un-consumed blobs in alphabetical order.
#ifndef MSCNN_CAFFE_NET_HPP_
#define MSCNN_CAFFE_NET_HPP_

#include <map>
#include <utility>
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
  // fusion: Net-level operator fusion (ReLU / MAX 2x2 pooling into the producing convolution, ROIPooling -> Concat).  On by
  // default; the process-wide default can be turned off with MSCNN_NO_FUSE=1 (results are bit-identical either way).
  explicit Net(const NetParameter& param, Phase phase = TEST);
  explicit Net(const string& param_file, Phase phase);
  Net(const NetParameter& param, Phase phase, bool fusion);
  virtual ~Net();

  const vector<Blob<Dtype>*>& Forward(Dtype* loss = NULL);
  const vector<Blob<Dtype>*>& ForwardPrefilled(Dtype* loss = NULL) { return Forward(loss); }
  Dtype ForwardFromTo(int start, int end);
  void Reshape();

  // Loads weights by layer name from a binary NetParameter (.caffemodel), net.cpp:750-803; a name ending in ".h5" is read as
  // an HDF5 snapshot like the reference does (net.cpp:788-795, 806-848).
  void CopyTrainedLayersFrom(const string& trained_filename);
  void CopyTrainedLayersFromHDF5(const string& trained_filename);
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
  inline const vector<int>& input_blob_indices() const { return net_input_blob_indices_; }
  // number of ForwardFromTo calls so far (readers that cache something derived from the blobs key it on this)
  long forward_count() const { return forward_count_; }
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
  // Tops whose producer was redirected into a fused Concat's top (roi_pool_org / roi_pool_ctx) are materialised lazily:
  // blob_by_name() copies the channel window back into the blob's own storage when the producer has run since the last
  // copy.  Code that reaches into blobs() directly must call this first.
  void MaterializeBlob(int blob_id) const;
  // Deferred ROI pooling (ROIPoolingLayer::set_deferred): the pair's blob is written now if it is still pending.  Call before writing
  // -- from outside the Net -- into a blob the pooling reads (the C ABI's blob setters do).
  void MaterializePending() const;
  // ... only where `blob_name` is (or shares its data, through Split layers, with) a blob the pending pooling reads
  void MaterializePendingReadersOf(const string& blob_name) const;
  // Chains of same-resolution F(4x4,3x3) convolutions (ConvolutionLayer::ChainTo: conv2_1 -> conv2_2, conv3_1 -> 3_2 -> 3_3,
  // conv4_1 -> 4_2 -> 4_3): the blob between two members is not written while both run in one ForwardFromTo call; blob_by_name()
  // / MaterializeBlob() re-run the producer unchained when somebody asks for it (bit-identical).  On by default with `fusion`;
  // MSCNN_NO_CHAIN=1 or SetChainFusion(false) turn it off (every blob is then written by every Forward).
  void SetChainFusion(bool on) { chain_fusion_ = on; }
  // Every blob the last Forward left unwritten is written now (the weights are about to change: what the blobs hold must stay what
  // the layers computed with the OLD weights, as in the reference).  Called by the weight loaders and the C ABI's parameter setter.
  void MaterializeStale() const;
  bool chain_fusion() const { return chain_fusion_; }
  // (producer layer, consumer layer or -1 for a top read by its fused pooling only) of every registered pair
  vector<std::pair<int, int> > chain_pairs() const {
    vector<std::pair<int, int> > out;
    for (size_t k = 0; k < chain_pairs_.size(); ++k) out.push_back(std::make_pair(chain_pairs_[k].producer, chain_pairs_[k].consumer));
    return out;
  }
  // Numerical calibration on representative data: call after a Forward.  Every Convolution layer that runs a Winograd
  // form is re-computed with the direct k-ordered kernel on the same bottom; where max |dy| / max(1, |y|) exceeds `tol`
  // the layer is switched to the direct kernel for good (ConvolutionLayer::set_algo).  Returns the layers switched;
  // errors (per layer index, 0 for layers not checked) are left in calibration_err().
  vector<int> CalibrateNumerics(double tol);
  const vector<double>& calibration_err() const { return calib_err_; }
  // Safe by default (no call needed): every Convolution layer checks its Winograd result against the direct kernel by itself on the
  // first Forward after construction / CopyTrainedLayersFrom / MarkWeightsChanged and falls back before that Forward returns
  // (ConvolutionLayer::set_selfcheck); the Net only keeps the books -- calibration_err(), auto_calibrate_checks / _switched -- and
  // offers the opt-OUT: SetAutoCalibrate(0).  tol > 0 re-arms the check on every layer with that tolerance.
  void SetAutoCalibrate(double tol);
  int auto_calibrate_checks() const { return auto_checks_; }
  const vector<int>& auto_calibrate_switched() const { return auto_switched_; }
  // The same comparison while a stream of frames runs (the numerics watch): every `period`-th whole Forward looks at ONE Winograd
  // layer (round robin) -- its bottom and top are written as blobs in that frame although it stays in its convolution chain, and one band of it (a few
  // rows / images: ~30 us of direct-kernel work; round robin too) is recomputed with the direct kernel BEHIND the frame on the same stream, without a
  // host synchronisation.  The verdict is collected by a later Forward (or by the state accessors below); a layer that strayed by more
  // than tol runs the direct kernel for good from the frame after.  A watch frame costs up to two extra blob writes plus that band
  // (2 - 8 % of a 7s-576 frame); no frame ever waits for a check.  ON by default (every
  // kDefaultWatchPeriod-th frame, tolerance 5e-5: ~0.1 % of a stream); period 0 turns it off.
  static constexpr int kDefaultWatchPeriod = 25;
  void SetNumericsWatch(int period, double tol);
  // (both wait for a verdict that is still out, so that what they return is final for the frames forwarded so far)
  int numerics_watch_checks() { WatchCollect(true); return watch_checks_; }
  const vector<int>& numerics_watch_switched() { WatchCollect(true); return watch_switched_; }
  void WatchCollect(bool wait);
  // Health of the plane-GEMM kernel's stream-K hand-off (include/mscnn_hip.h: mscnn_wgemm_handoff_event).  A launch whose finisher
  // gave up on a contributor leaves a NaN tile behind and its tag in a pinned status word; the Net looks at that word wherever the
  // stream has just been synchronised anyway -- behind BoxOutput's row-count read inside ForwardFromTo (covers the trunk and the
  // heads, whose NaN scores BoxOutput would otherwise silently drop) and, through HandoffRecover(), behind the final stage's / a blob
  // read's synchronisation (covers roi_c1 and fc6).  The answer is always the same: whole-tile scheduling for the rest of the
  // process (mscnn_wgemm_force_whole_tiles) and the affected range once more.  handoff_errors() = how often that happened.
  //   HandoffRecover(): call right after synchronising Caffe::stream() on outputs of the last ForwardFromTo.  false = nothing was
  //   reported; true = the range has been run again (asynchronously, like any Forward): read the outputs again.
  bool HandoffRecover();
  // For callers that pipeline frames and therefore can NOT have a range run again (mscnn_net_detect_end): true when a hand-off event
  // was raised since the last look -- whole tiles are forced and the event is counted as in HandoffRecover, nothing is re-run.
  // (the stream is synchronised only up to the frame being collected: what was enqueued behind it stays "unchecked in flight")
  bool HandoffEventSeen() { return HandoffEventPending(); }
  int handoff_errors() const { return handoff_errors_; }

 protected:
  void Init(const NetParameter& param);
  void ApplyFusion();
  // max |x| hand-over for split-fp16 convolutions (ConvolutionLayer::set_amax_io): one device slot per layer; a blob's bound
  // follows it through Split / in-place ReLU / Dropout / MAX pooling / ROIPooling / Concat of blobs with one common source.
  void WireAmax();
  void* amax_slots_ = nullptr;
  vector<int> amax_src_;          // convolution layer -> the convolution layer whose max |y| bounds its bottom, or -1

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
  // fused layer i -> the layers that now produce its top(s) (ReLU / Pooling: one convolution; Concat: the ROIPooling layers)
  vector<vector<int> > fused_producers_;
  struct Redirect { int target_blob, c_total, c_offset; };
  std::map<int, Redirect> redirect_;               // blob id -> where its data really lives
  mutable std::map<int, bool> redirect_dirty_;     // producer ran since the last MaterializeBlob
  struct DeferredPool { int first_layer, conv_layer, blob; };      // a ROIPooling pair whose only reader pools in its own input stage
  vector<DeferredPool> deferred_pools_;
  struct ChainPair { int producer, consumer, blob; };              // convolution -> its only reader, a same-resolution 3x3 convolution (-1: read by its fused pooling only)
  vector<ChainPair> chain_pairs_;
  bool chain_fusion_ = true;
  int SplitSource(int blob) const;      // through Split layers (their tops share the bottom's data) to the blob that holds the data
  vector<double> calib_err_;
  int NextWatchLayer() const;
  int watch_period_ = kDefaultWatchPeriod, watch_frame_ = 0, watch_next_ = 0, watch_checks_ = 0;
  int watch_pending_ = -1, watch_band_ = 0;      // layer whose band check is in flight (-1: none); band index (advances once per trip round the layers)
  double watch_tol_ = 5e-5;
  int auto_checks_ = 0;
  double auto_tol_ = 5e-5;      // (= ConvolutionLayer::kDefaultSelfcheckTol; 0 once SetAutoCalibrate(0) opted out)
  vector<int> auto_switched_;
  vector<int> watch_switched_;
  bool HandoffEventPending();      // (the stream must have been synchronised) a new tag in the status word: whole tiles forced, counted
  unsigned long long handoff_seen_ = 0;
  int handoff_errors_ = 0;
  int last_start_ = 0, last_end_ = -1;
  int suspect_start_ = -1;      // smallest `start` of the ranges run since the last synchronisation point that saw no hand-off event (-1: none)
  long forward_count_ = 0;
  DISABLE_COPY_AND_ASSIGN(Net);
};

}  // namespace caffe
#endif
