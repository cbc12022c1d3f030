// Executed boundary check (tests/test_cabi.py builds it on the CPU box, tests/test_gpu_net.py runs it on the MI355X):
//   A. SyncedMemory's head machine -- the reference's own test cases (src/caffe/test/test_syncedmem.cpp:13-120) restated
//      without gtest, on hipMemset / hipMemcpy instead of caffe_gpu_memset / caffe_gpu_memcpy;
//   B. Blob shape / count / sharing semantics (src/caffe/test/test_blob.cpp:28-60 and blob.cpp:22-50, 84-90);
//   C. user_layers.cpp in a live Net: a prototxt that names the two user-registered layer types beside the stock ROIPooling,
//      with a top consumed twice (automatic Split), forwarded on the GPU: the user layer that calls the C ABI must produce the
//      stock layer's bytes, the pure Layer-interface one must see and produce host data through SyncedMemory.
// `run_boundary <prototxt path> construct-only` stops after building the Net (no device needed).
// Exit code 0 and a last line "BOUNDARY OK" on success; the first failed expectation prints file:line and exits 1.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "user_layers.cpp"

#define EXPECT(cond)                                                              \
  do {                                                                            \
    if (!(cond)) {                                                                \
      std::fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

using namespace caffe;

static void TestSyncedMemory() {
  {  // TestInitialization
    SyncedMemory mem(10);
    EXPECT(mem.head() == SyncedMemory::UNINITIALIZED);
    EXPECT(mem.size() == 10u);
    SyncedMemory* p_mem = new SyncedMemory(10 * sizeof(float));
    EXPECT(p_mem->size() == 10 * sizeof(float));
    delete p_mem;
  }
  {  // TestAllocationCPUGPU / CPU / GPU
    SyncedMemory mem(10);
    EXPECT(mem.cpu_data() != nullptr);
    EXPECT(mem.gpu_data() != nullptr);
    EXPECT(mem.mutable_cpu_data() != nullptr);
    EXPECT(mem.mutable_gpu_data() != nullptr);
    SyncedMemory gpu_first(10);
    EXPECT(gpu_first.gpu_data() != nullptr);
    EXPECT(gpu_first.mutable_gpu_data() != nullptr);
  }
  {  // TestCPUWrite
    SyncedMemory mem(10);
    void* cpu_data = mem.mutable_cpu_data();
    EXPECT(mem.head() == SyncedMemory::HEAD_AT_CPU);
    std::memset(cpu_data, 1, mem.size());
    for (size_t i = 0; i < mem.size(); ++i) EXPECT(static_cast<char*>(cpu_data)[i] == 1);
    cpu_data = mem.mutable_cpu_data();
    EXPECT(mem.head() == SyncedMemory::HEAD_AT_CPU);
    std::memset(cpu_data, 2, mem.size());
    for (size_t i = 0; i < mem.size(); ++i) EXPECT(static_cast<char*>(cpu_data)[i] == 2);
  }
  {  // TestGPURead
    SyncedMemory mem(10);
    void* cpu_data = mem.mutable_cpu_data();
    EXPECT(mem.head() == SyncedMemory::HEAD_AT_CPU);
    std::memset(cpu_data, 1, mem.size());
    const void* gpu_data = mem.gpu_data();
    EXPECT(mem.head() == SyncedMemory::SYNCED);
    char recovered[10];
    EXPECT(hipMemcpy(recovered, gpu_data, 10, hipMemcpyDeviceToHost) == hipSuccess);
    for (int i = 0; i < 10; ++i) EXPECT(recovered[i] == 1);
    cpu_data = mem.mutable_cpu_data();
    EXPECT(mem.head() == SyncedMemory::HEAD_AT_CPU);
    std::memset(cpu_data, 2, mem.size());
    gpu_data = mem.gpu_data();
    EXPECT(mem.head() == SyncedMemory::SYNCED);
    EXPECT(hipMemcpy(recovered, gpu_data, 10, hipMemcpyDeviceToHost) == hipSuccess);
    for (int i = 0; i < 10; ++i) EXPECT(recovered[i] == 2);
  }
  {  // TestGPUWrite
    SyncedMemory mem(10);
    void* gpu_data = mem.mutable_gpu_data();
    EXPECT(mem.head() == SyncedMemory::HEAD_AT_GPU);
    EXPECT(hipMemset(gpu_data, 1, mem.size()) == hipSuccess);
    const void* cpu_data = mem.cpu_data();
    for (size_t i = 0; i < mem.size(); ++i) EXPECT(static_cast<const char*>(cpu_data)[i] == 1);
    EXPECT(mem.head() == SyncedMemory::SYNCED);
    gpu_data = mem.mutable_gpu_data();
    EXPECT(mem.head() == SyncedMemory::HEAD_AT_GPU);
    EXPECT(hipMemset(gpu_data, 2, mem.size()) == hipSuccess);
    cpu_data = mem.cpu_data();
    for (size_t i = 0; i < mem.size(); ++i) EXPECT(static_cast<const char*>(cpu_data)[i] == 2);
    EXPECT(mem.head() == SyncedMemory::SYNCED);
  }
}

static void TestBlob() {
  Blob<float> blob;                                       // test_blob.cpp:28-36
  EXPECT(blob.num_axes() == 0 && blob.count() == 0);
  Blob<float> pre(2, 3, 4, 5);
  EXPECT(pre.num() == 2 && pre.channels() == 3 && pre.height() == 4 && pre.width() == 5 && pre.count() == 120);
  blob.Reshape(2, 3, 4, 5);                               // :43-50
  EXPECT(blob.num() == 2 && blob.channels() == 3 && blob.height() == 4 && blob.width() == 5 && blob.count() == 120);
  EXPECT(blob.count(1) == 60 && blob.count(1, 3) == 12 && blob.shape(-1) == 5 && blob.offset(1, 2, 3, 4) == 119);
  std::vector<int> low(2);                                // legacy accessors on a 2-axis blob (blob.hpp:150-163)
  low[0] = 7; low[1] = 5;
  Blob<float> rois(low);
  EXPECT(rois.num() == 7 && rois.channels() == 5 && rois.height() == 1 && rois.width() == 1);
  // host write -> device read -> device write -> host read, through the Blob accessors
  float* h = blob.mutable_cpu_data();
  for (int i = 0; i < 120; ++i) h[i] = (float)i;
  float back[120];
  EXPECT(hipMemcpy(back, blob.gpu_data(), sizeof(back), hipMemcpyDeviceToHost) == hipSuccess);
  for (int i = 0; i < 120; ++i) EXPECT(back[i] == (float)i);
  EXPECT(hipMemset(blob.mutable_gpu_data(), 0, sizeof(back)) == hipSuccess);
  for (int i = 0; i < 120; ++i) EXPECT(blob.cpu_data()[i] == 0.f);
  // ShareData: the sharer sees the owner's bytes (blob.cpp:84-87); shrinking Reshape keeps the allocation (:40-44)
  Blob<float> view(2, 3, 4, 5);
  view.ShareData(blob);
  blob.mutable_cpu_data()[7] = 42.f;
  EXPECT(view.cpu_data()[7] == 42.f);
  const float* before = blob.gpu_data();
  blob.Reshape(1, 3, 4, 5);
  EXPECT(blob.count() == 60 && blob.gpu_data() == before);
}

static const char kNet[] =
    "name: \"boundary\"\n"
    "input: \"data\" input_dim: 1 input_dim: 8 input_dim: 12 input_dim: 20\n"
    "input: \"rois\" input_dim: 3 input_dim: 5 input_dim: 1 input_dim: 1\n"
    "layer { name: \"twice\" type: \"ScaleByTwo\" bottom: \"data\" top: \"twice\" }\n"
    "layer { name: \"pool_user\" type: \"UserROIPooling\" bottom: \"twice\" bottom: \"rois\" top: \"pool_user\"\n"
    "        roi_pooling_param { pooled_h: 3 pooled_w: 3 spatial_scale: 0.5 pad_ratio: 0.25 } }\n"
    "layer { name: \"pool_stock\" type: \"ROIPooling\" bottom: \"twice\" bottom: \"rois\" top: \"pool_stock\"\n"
    "        roi_pooling_param { pooled_h: 3 pooled_w: 3 spatial_scale: 0.5 pad_ratio: 0.25 } }\n";

static void TestUserLayersInNet(const char* prototxt_path, bool forward) {
  FILE* f = std::fopen(prototxt_path, "w");
  EXPECT(f != nullptr);
  std::fputs(kNet, f);
  std::fclose(f);
  bool user = false, by_two = false;
  const std::vector<std::string> types = LayerRegistry<float>::LayerTypeList();
  for (size_t i = 0; i < types.size(); ++i) { user = user || types[i] == "UserROIPooling"; by_two = by_two || types[i] == "ScaleByTwo"; }
  EXPECT(user && by_two);                                               // REGISTER_LAYER_CLASS ran (layer_factory.hpp:116-137)
  Net<float> net(prototxt_path, TEST);
  EXPECT(net.has_layer("pool_user") && std::string(net.layer_by_name("pool_user")->type()) == "UserROIPooling");
  EXPECT(net.has_layer("twice_twice_0_split"));                         // InsertSplits naming (insert_splits.cpp:112-126)
  EXPECT(net.num_inputs() == 2);
  if (!forward) return;                                                 // CPU box: construction only
  float* x = net.blob_by_name("data")->mutable_cpu_data();
  unsigned s = 12345u;
  for (int i = 0; i < 8 * 12 * 20; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)((int)(s >> 9) % 2001 - 1000) / 250.f; }
  const float rois[15] = {0, 2, 3, 30, 20,   0, 0, 0, 39, 23,   0, 10, 4, 11, 5};
  std::memcpy(net.blob_by_name("rois")->mutable_cpu_data(), rois, sizeof(rois));
  net.Forward();
  const shared_ptr<Blob<float> > twice = net.blob_by_name("twice"), pu = net.blob_by_name("pool_user"), ps = net.blob_by_name("pool_stock");
  EXPECT(twice && pu && ps);
  for (int i = 0; i < twice->count(); ++i) EXPECT(twice->cpu_data()[i] == 2.f * x[i]);
  EXPECT(pu->shape() == ps->shape() && pu->num() == 3 && pu->channels() == 8 && pu->height() == 3 && pu->width() == 3);
  EXPECT(std::memcmp(pu->cpu_data(), ps->cpu_data(), sizeof(float) * pu->count()) == 0);
  float mx = -1e30f;
  for (int i = 0; i < pu->count(); ++i) mx = pu->cpu_data()[i] > mx ? pu->cpu_data()[i] : mx;
  EXPECT(mx > 4.f && mx <= 8.f);                                        // maxima of 2 * x, |x| <= 4
  // a second forward with other ROIs: Reshape follows the bottoms (layer.hpp:451-456)
  std::vector<int> two(4, 1);
  two[0] = 2; two[1] = 5;
  net.blob_by_name("rois")->Reshape(two);
  std::memcpy(net.blob_by_name("rois")->mutable_cpu_data(), rois + 5, sizeof(float) * 10);
  net.Forward();
  EXPECT(net.blob_by_name("pool_user")->num() == 2);
  EXPECT(std::memcmp(net.blob_by_name("pool_user")->cpu_data(), net.blob_by_name("pool_stock")->cpu_data(),
                     sizeof(float) * net.blob_by_name("pool_user")->count()) == 0);
}

int main(int argc, char** argv) {
  const char* prototxt = argc > 1 ? argv[1] : "/tmp/mscnn_boundary.prototxt";
  if (argc > 2 && std::string(argv[2]) == "construct-only") {           // no GPU: registry, parser, Split insertion, set-up
    TestUserLayersInNet(prototxt, false);
    std::printf("CONSTRUCTION OK\n");
    return 0;
  }
  Caffe::set_mode(Caffe::GPU);
  Caffe::SetDevice(0);
  TestSyncedMemory();
  std::printf("SyncedMemory ok\n");
  TestBlob();
  std::printf("Blob ok\n");
  TestUserLayersInNet(prototxt, true);
  std::printf("user layers in a Net ok\nBOUNDARY OK\n");
  return 0;
}
