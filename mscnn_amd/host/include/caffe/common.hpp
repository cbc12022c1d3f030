// Caffe-compatible common definitions (mirror of include/caffe/common.hpp for the inference path).
// The `Caffe` singleton is thread-local like the reference's (src/caffe/common.cpp:12-20): one host
// thread per GPU, each with its own mode / device / HIP stream.
#ifndef MSCNN_CAFFE_COMMON_HPP_
#define MSCNN_CAFFE_COMMON_HPP_

#include <memory>
#include <string>
#include <vector>

#include "caffe/util/logging.hpp"

namespace caffe {

using std::shared_ptr;
using std::string;
using std::vector;

#define DISABLE_COPY_AND_ASSIGN(classname) \
 private:                                  \
  classname(const classname&);             \
  classname& operator=(const classname&)

// Only float is instantiated: the MI355X kernels are fp32 (the reference also registers double,
// include/caffe/common.hpp:41-44; nothing in the MS-CNN deploy path uses it).
#define INSTANTIATE_CLASS(classname) template class classname<float>

class Caffe {
 public:
  enum Brew { CPU, GPU };
  static Caffe& Get();
  static Brew mode() { return Get().mode_; }
  // CPU mode is accepted for API compatibility, but every layer's Forward_cpu is fatal: this build
  // has no CPU compute path (the CPU oracle lives in oracle/ and is test infrastructure only).
  static void set_mode(Brew mode) { Get().mode_ = mode; }
  static void SetDevice(int device_id);
  static int device() { return Get().device_; }
  // HIP stream every layer of this thread launches on (default: the null stream, like the reference).
  static void* stream() { return Get().stream_; }
  static void set_stream(void* s) { Get().stream_ = s; }
  static void DeviceQuery();

 private:
  Caffe() : mode_(GPU), device_(0), stream_(nullptr) {}
  Brew mode_;
  int device_;
  void* stream_;
};

// Turns a libmscnn_hip status into the reference's CHECK-style fatal (device_alternate.hpp:48-53).
void CheckMscnnStatus(int status, const char* what);
#define MSCNN_CHECK(expr) ::caffe::CheckMscnnStatus((expr), #expr)

void HipCheck(int hip_error, const char* what);
#define HIP_CHECK(expr) ::caffe::HipCheck((int)(expr), #expr)

}  // namespace caffe
#endif
