// Blob<Dtype>: N-d array with lazily synchronised host/device storage
// (mirror of include/caffe/blob.hpp:23-277 / src/caffe/blob.cpp for the data half; diff is not
// allocated: inference only).
#ifndef MSCNN_CAFFE_BLOB_HPP_
#define MSCNN_CAFFE_BLOB_HPP_

#include <algorithm>
#include <string>
#include <vector>

#include "caffe/common.hpp"
#include "caffe/syncedmem.hpp"

const int kMaxBlobAxes = 32;

namespace caffe {

template <typename Dtype>
class Blob {
 public:
  Blob() : count_(0), capacity_(0) {}
  explicit Blob(const int num, const int channels, const int height, const int width) : count_(0), capacity_(0) {
    Reshape(num, channels, height, width);
  }
  explicit Blob(const vector<int>& shape) : count_(0), capacity_(0) { Reshape(shape); }

  void Reshape(const int num, const int channels, const int height, const int width);
  // Reallocates only when growing (blob.cpp:40-44): shrinking keeps the buffer, which is what lets the
  // data-dependent ROI count change every image without touching the allocator.
  void Reshape(const vector<int>& shape);
  void ReshapeLike(const Blob& other) { Reshape(other.shape()); }
  string shape_string() const;
  const vector<int>& shape() const { return shape_; }
  int shape(int index) const { return shape_[CanonicalAxisIndex(index)]; }
  int num_axes() const { return (int)shape_.size(); }
  int count() const { return count_; }
  int count(int start_axis, int end_axis) const;
  int count(int start_axis) const { return count(start_axis, num_axes()); }
  int CanonicalAxisIndex(int axis_index) const;
  int num() const { return LegacyShape(0); }
  int channels() const { return LegacyShape(1); }
  int height() const { return LegacyShape(2); }
  int width() const { return LegacyShape(3); }
  int LegacyShape(int index) const;
  int offset(const int n, const int c = 0, const int h = 0, const int w = 0) const {
    return ((n * channels() + c) * height() + h) * width() + w;
  }
  const Dtype* cpu_data() const;
  void set_cpu_data(Dtype* data);
  const Dtype* gpu_data() const;
  Dtype* mutable_cpu_data();
  Dtype* mutable_gpu_data();
  const shared_ptr<SyncedMemory>& data() const { return data_; }
  void ShareData(const Blob& other);
  void CopyFrom(const Blob<Dtype>& source, bool copy_diff = false, bool reshape = false);
  Dtype asum_data() const;

 protected:
  shared_ptr<SyncedMemory> data_;
  vector<int> shape_;
  int count_;
  int capacity_;
  DISABLE_COPY_AND_ASSIGN(Blob);
};

}  // namespace caffe
#endif
