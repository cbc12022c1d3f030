// Caffe singleton, SyncedMemory, Blob on the HIP runtime.
// Reference behaviour: src/caffe/common.cpp, src/caffe/syncedmem.cpp:25-139, src/caffe/blob.cpp:22-45.
#include <hip/hip_runtime_api.h>

#include <climits>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "../../../include/mscnn_hip.h"
#include "caffe/blob.hpp"
#include "caffe/common.hpp"
#include "caffe/layers/mscnn_layers.hpp"
#include "caffe/syncedmem.hpp"

namespace caffe {

Caffe& Caffe::Get() {
  static thread_local Caffe instance;
  return instance;
}

void Caffe::SetDevice(int device_id) {
  HIP_CHECK(hipSetDevice(device_id));
  Get().device_ = device_id;
}

void Caffe::DeviceQuery() {
  hipDeviceProp_t prop;
  int device;
  HIP_CHECK(hipGetDevice(&device));
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  LOG(WARNING) << "Device id: " << device << "  Name: " << prop.name << " (" << prop.gcnArchName << ")  CUs: "
               << prop.multiProcessorCount << "  Total global memory: " << prop.totalGlobalMem;
}

void CheckMscnnStatus(int status, const char* what) {
  CHECK_EQ(status, 0) << what << ": " << mscnn_last_error();
}

void HipCheck(int hip_error, const char* what) {
  CHECK_EQ(hip_error, (int)hipSuccess) << what << ": " << hipGetErrorString((hipError_t)hip_error);
}

// ---------------------------------------------------------------------------------------------- DeviceBuffer
DeviceBuffer::~DeviceBuffer() {
  if (ptr_) (void)hipFree(ptr_);
}

void DeviceBuffer::Release() {
  if (ptr_) HIP_CHECK(hipFree(ptr_));
  ptr_ = nullptr;
  bytes_ = 0;
}

void* DeviceBuffer::Reserve(size_t bytes) {
  if (bytes > bytes_) {
    if (ptr_) HIP_CHECK(hipFree(ptr_));
    ptr_ = nullptr;
    bytes_ = 0;
    HIP_CHECK(hipMalloc(&ptr_, bytes));
    bytes_ = bytes;
  }
  return ptr_;
}

// ---------------------------------------------------------------------------------------------- SyncedMemory
// Host side is pinned (hipHostMalloc) like the reference in GPU mode (syncedmem.hpp:15-26); when no HIP device is
// usable (graph-construction tests on a CPU-only box) it degrades to malloc -- host memory only, never a compute path.
static void* HostAlloc(size_t size, bool* pinned) {
  void* p = nullptr;
  if (hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault) == hipSuccess && p) { *pinned = true; return p; }
  (void)hipGetLastError();
  *pinned = false;
  p = std::malloc(size ? size : 1);
  CHECK(p != nullptr) << "host allocation of " << size << " bytes failed";
  return p;
}
static void HostFree(void* p, bool pinned) {
  if (pinned) (void)hipHostFree(p); else std::free(p);
}

SyncedMemory::~SyncedMemory() {
  if (cpu_ptr_ && own_cpu_) HostFree(cpu_ptr_, cpu_pinned_);
  if (gpu_ptr_ && own_gpu_) (void)hipFree(gpu_ptr_);
}

inline void SyncedMemory::to_cpu() {
  switch (head_) {
    case UNINITIALIZED:
      cpu_ptr_ = HostAlloc(size_, &cpu_pinned_);
      std::memset(cpu_ptr_, 0, size_);
      head_ = HEAD_AT_CPU;
      own_cpu_ = true;
      break;
    case HEAD_AT_GPU:
      if (cpu_ptr_ == nullptr) {
        cpu_ptr_ = HostAlloc(size_, &cpu_pinned_);
        own_cpu_ = true;
      }
      // ordered after the kernels of this thread's stream, then synchronous like the reference's cudaMemcpy
      HIP_CHECK(hipMemcpyAsync(cpu_ptr_, gpu_ptr_, size_, hipMemcpyDeviceToHost, (hipStream_t)Caffe::stream()));
      HIP_CHECK(hipStreamSynchronize((hipStream_t)Caffe::stream()));
      head_ = SYNCED;
      break;
    case HEAD_AT_CPU:
    case SYNCED:
      break;
  }
}

inline void SyncedMemory::to_gpu() {
  switch (head_) {
    case UNINITIALIZED:
      HIP_CHECK(hipMalloc(&gpu_ptr_, size_ ? size_ : 1));
      HIP_CHECK(hipMemsetAsync(gpu_ptr_, 0, size_, (hipStream_t)Caffe::stream()));
      head_ = HEAD_AT_GPU;
      own_gpu_ = true;
      break;
    case HEAD_AT_CPU:
      if (gpu_ptr_ == nullptr) {
        HIP_CHECK(hipMalloc(&gpu_ptr_, size_ ? size_ : 1));
        own_gpu_ = true;
      }
      HIP_CHECK(hipMemcpyAsync(gpu_ptr_, cpu_ptr_, size_, hipMemcpyHostToDevice, (hipStream_t)Caffe::stream()));
      HIP_CHECK(hipStreamSynchronize((hipStream_t)Caffe::stream()));
      head_ = SYNCED;
      break;
    case HEAD_AT_GPU:
    case SYNCED:
      break;
  }
}

const void* SyncedMemory::cpu_data() { to_cpu(); return cpu_ptr_; }
const void* SyncedMemory::gpu_data() { to_gpu(); return gpu_ptr_; }
void* SyncedMemory::mutable_cpu_data() { to_cpu(); head_ = HEAD_AT_CPU; return cpu_ptr_; }
void* SyncedMemory::mutable_gpu_data() { to_gpu(); head_ = HEAD_AT_GPU; return gpu_ptr_; }

void SyncedMemory::set_cpu_data(void* data) {
  CHECK(data);
  if (own_cpu_) HostFree(cpu_ptr_, cpu_pinned_);
  cpu_ptr_ = data;
  head_ = HEAD_AT_CPU;
  own_cpu_ = false;
}

void SyncedMemory::set_gpu_data(void* data) {
  CHECK(data);
  if (own_gpu_) (void)hipFree(gpu_ptr_);
  gpu_ptr_ = data;
  head_ = HEAD_AT_GPU;
  own_gpu_ = false;
}

// ---------------------------------------------------------------------------------------------- Blob
template <typename Dtype>
void Blob<Dtype>::Reshape(const int num, const int channels, const int height, const int width) {
  vector<int> shape(4);
  shape[0] = num; shape[1] = channels; shape[2] = height; shape[3] = width;
  Reshape(shape);
}

template <typename Dtype>
void Blob<Dtype>::Reshape(const vector<int>& shape) {
  CHECK_LE(shape.size(), (size_t)kMaxBlobAxes);
  count_ = 1;
  shape_.resize(shape.size());
  for (size_t i = 0; i < shape.size(); ++i) {
    CHECK_GE(shape[i], 0);
    if (count_ != 0) CHECK_LE(shape[i], INT_MAX / count_) << "blob size exceeds INT_MAX";
    count_ *= shape[i];
    shape_[i] = shape[i];
  }
  if (count_ > capacity_) {
    capacity_ = count_;
    data_.reset(new SyncedMemory(capacity_ * sizeof(Dtype)));
  }
}

template <typename Dtype>
string Blob<Dtype>::shape_string() const {
  std::ostringstream stream;
  for (size_t i = 0; i < shape_.size(); ++i) stream << shape_[i] << " ";
  stream << "(" << count_ << ")";
  return stream.str();
}

template <typename Dtype>
int Blob<Dtype>::count(int start_axis, int end_axis) const {
  CHECK_LE(start_axis, end_axis);
  CHECK_GE(start_axis, 0);
  CHECK_LE(end_axis, num_axes());
  int count = 1;
  for (int i = start_axis; i < end_axis; ++i) count *= shape_[i];
  return count;
}

template <typename Dtype>
int Blob<Dtype>::CanonicalAxisIndex(int axis_index) const {
  CHECK_GE(axis_index, -num_axes()) << "axis " << axis_index << " out of range for " << num_axes() << "-D Blob with shape " << shape_string();
  CHECK_LT(axis_index, num_axes()) << "axis " << axis_index << " out of range for " << num_axes() << "-D Blob with shape " << shape_string();
  return axis_index < 0 ? axis_index + num_axes() : axis_index;
}

template <typename Dtype>
int Blob<Dtype>::LegacyShape(int index) const {
  CHECK_LE(num_axes(), 4) << "Cannot use legacy accessors on Blobs with > 4 axes.";
  CHECK_LT(index, 4);
  CHECK_GE(index, -4);
  if (index >= num_axes() || index < -num_axes()) return 1;   // blob.hpp:153-164
  return shape(index);
}

template <typename Dtype>
const Dtype* Blob<Dtype>::cpu_data() const { CHECK(data_); return (const Dtype*)data_->cpu_data(); }
template <typename Dtype>
void Blob<Dtype>::set_cpu_data(Dtype* data) { CHECK(data); data_->set_cpu_data(data); }
template <typename Dtype>
const Dtype* Blob<Dtype>::gpu_data() const { CHECK(data_); return (const Dtype*)data_->gpu_data(); }
template <typename Dtype>
Dtype* Blob<Dtype>::mutable_cpu_data() { CHECK(data_); return static_cast<Dtype*>(data_->mutable_cpu_data()); }
template <typename Dtype>
Dtype* Blob<Dtype>::mutable_gpu_data() { CHECK(data_); return static_cast<Dtype*>(data_->mutable_gpu_data()); }

template <typename Dtype>
void Blob<Dtype>::ShareData(const Blob& other) {
  CHECK_EQ(count_, other.count());
  data_ = other.data();
}

template <typename Dtype>
void Blob<Dtype>::CopyFrom(const Blob& source, bool copy_diff, bool reshape) {
  if (source.count() != count_ || source.shape() != shape_) {
    if (reshape) ReshapeLike(source);
    else LOG(FATAL) << "Trying to copy blobs of different sizes.";
  }
  CHECK(!copy_diff) << "diffs are not allocated in this inference build";
  HIP_CHECK(hipMemcpyAsync(mutable_gpu_data(), source.gpu_data(), sizeof(Dtype) * count_, hipMemcpyDeviceToDevice,
                           (hipStream_t)Caffe::stream()));
}

template <typename Dtype>
Dtype Blob<Dtype>::asum_data() const {
  if (!data_) return 0;
  const Dtype* p = cpu_data();
  double s = 0;
  for (int i = 0; i < count_; ++i) s += p[i] < 0 ? -p[i] : p[i];
  return (Dtype)s;
}

template class Blob<float>;
template class Blob<int>;

}  // namespace caffe
