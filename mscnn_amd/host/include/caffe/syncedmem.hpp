// SyncedMemory: lazily mirrored host/device buffer with the reference's 4-state head machine
// (include/caffe/syncedmem.hpp:45-83, src/caffe/syncedmem.cpp:25-139), on hipMalloc / hipHostMalloc.
#ifndef MSCNN_CAFFE_SYNCEDMEM_HPP_
#define MSCNN_CAFFE_SYNCEDMEM_HPP_

#include <cstddef>

#include "caffe/common.hpp"

namespace caffe {

class SyncedMemory {
 public:
  SyncedMemory() : cpu_ptr_(nullptr), gpu_ptr_(nullptr), size_(0), head_(UNINITIALIZED), own_cpu_(false), own_gpu_(false), cpu_pinned_(false) {}
  explicit SyncedMemory(size_t size)
      : cpu_ptr_(nullptr), gpu_ptr_(nullptr), size_(size), head_(UNINITIALIZED), own_cpu_(false), own_gpu_(false), cpu_pinned_(false) {}
  ~SyncedMemory();
  const void* cpu_data();
  void set_cpu_data(void* data);
  const void* gpu_data();
  void set_gpu_data(void* data);
  void* mutable_cpu_data();
  void* mutable_gpu_data();
  enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };
  SyncedHead head() const { return head_; }
  size_t size() const { return size_; }

 private:
  void to_cpu();
  void to_gpu();
  void* cpu_ptr_;
  void* gpu_ptr_;
  size_t size_;
  SyncedHead head_;
  bool own_cpu_, own_gpu_, cpu_pinned_;
  DISABLE_COPY_AND_ASSIGN(SyncedMemory);
};

}  // namespace caffe
#endif
