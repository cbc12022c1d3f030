// detect_multi_gpu: the C++ host path of multi-GPU MS-CNN inference on one node.
//
// One host thread per GPU, each with its own net replica -- the reference's Caffe singleton is thread-local
// (src/caffe/common.cpp:13-20), so a thread IS a device context -- image k goes to GPU k mod G, and the only exchange is one
// RCCL all-gather per step of the device-resident detection packs (include/mscnn_dist.h).  Everything goes through the two C
// ABIs (include/mscnn_net.h, include/mscnn_dist.h); no Python, no torch.
//
//   detect_multi_gpu <deploy.prototxt> [--gpus G] [--images K] [--caffemodel file] [--precision f32|f16|f16x3] [--cls-id c] [--cap n]
//
// Without a caffemodel the replicas get identical seeded He-normal weights (a throughput / plumbing run; every replica must
// then produce the same detections for the same image, which rank 0 checks on the gathered packs of the first step).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/mscnn_dist.h"
#include "../../../include/mscnn_net.h"

namespace {

struct Options {
  std::string prototxt, caffemodel, precision = "f32";
  int gpus = 0, images = 16, cls_id = 2, cap = 2000;
  bool verbose = false;
};

#define NET_CHECK(expr) do { if ((expr) != 0) { fprintf(stderr, "[rank %d] %s: %s\n", rank, #expr, mscnn_net_last_error()); std::exit(2); } } while (0)
#define DIST_CHECK(expr) do { if ((expr) != 0) { fprintf(stderr, "[rank %d] %s: %s\n", rank, #expr, mscnn_dist_last_error()); std::exit(3); } } while (0)

// identical weights on every replica: He-normal per layer from a layer-indexed seed; the proposal heads get a class-0 bias
// so that a few hundred anchors pass fg_thr (as mscnn_amd/synth.py does for the Python benchmark)
void seed_weights(mscnn_net* net, int rank) {
  const int L = mscnn_net_num_layers(net);
  for (int l = 0; l < L; ++l) {
    const std::string type = mscnn_net_layer_type(net, l), name = mscnn_net_layer_name(net, l);
    if ((type != "Convolution" && type != "InnerProduct") || mscnn_net_layer_num_params(net, l) == 0) continue;
    int dims[8], nd = 0;
    NET_CHECK(mscnn_net_layer_param_shape(net, l, 0, dims, &nd));
    size_t count = 1, fan_in = 1;
    for (int i = 0; i < nd; ++i) { count *= dims[i]; if (i > 0) fan_in *= dims[i]; }
    std::mt19937 gen(1701u + (unsigned)l);
    std::normal_distribution<float> dist(0.f, std::sqrt(2.f / (float)fan_in));
    std::vector<float> w(count);
    for (float& v : w) v = dist(gen);
    if (name == "conv1_1") for (float& v : w) v *= 1.f / 57.f;
    const bool head = name.compare(0, 5, "LFCN_") == 0;
    if (head) {
      const size_t per = count / dims[0];
      for (int c = 0; c < dims[0]; ++c)
        for (size_t i = 0; i < per; ++i) w[c * per + i] *= (c < dims[0] - 4 ? 1.14f : 0.13f);
    }
    NET_CHECK(mscnn_net_set_param(net, l, 0, w.data(), count));
    if (mscnn_net_layer_num_params(net, l) > 1 && head) {
      std::vector<float> b(dims[0], 0.f);
      b[0] = 8.2f;
      NET_CHECK(mscnn_net_set_param(net, l, 1, b.data(), b.size()));
    }
  }
}

// procedural frame k: smooth colour gradients + a few rectangles, BGR minus the Caffe mean
void make_frame(int k, int H, int W, std::vector<float>* out) {
  out->resize((size_t)3 * H * W);
  std::mt19937 gen(97u * (unsigned)k + 5u);
  std::uniform_real_distribution<float> u(0.f, 1.f);
  const float mean[3] = {104.f, 117.f, 123.f};
  float fx[3], fy[3], ph[3];
  for (int c = 0; c < 3; ++c) { fx[c] = 2.f + 6.f * u(gen); fy[c] = 1.f + 4.f * u(gen); ph[c] = 6.28f * u(gen); }
  for (int c = 0; c < 3; ++c)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x)
        (*out)[((size_t)c * H + y) * W + x] = 127.f + 100.f * std::sin(fx[c] * x / W * 6.28f + fy[c] * y / H * 6.28f + ph[c]) - mean[c];
  for (int r = 0; r < 10; ++r) {
    const int w = 30 + (int)(u(gen) * W / 6), h = 30 + (int)(u(gen) * H / 4);
    const int x0 = (int)(u(gen) * (W - w)), y0 = (int)(u(gen) * (H - h));
    float col[3] = {255.f * u(gen), 255.f * u(gen), 255.f * u(gen)};
    for (int c = 0; c < 3; ++c)
      for (int y = y0; y < y0 + h; ++y)
        for (int x = x0; x < x0 + w; ++x) (*out)[((size_t)c * H + y) * W + x] = col[c] - mean[c];
  }
}

void worker(int rank, int world, const Options& opt, const unsigned char* id, std::vector<double>* seconds, std::vector<int>* total_dets) {
  // this thread (and the helper threads it will create) onto its GPU's NUMA node, in a slice of its own among the ranks sharing the node
  char placement[512] = "";
  if (mscnn_dist_pin_host_thread(rank, rank, world, 0, nullptr, placement, sizeof(placement)) != 0)
    fprintf(stderr, "[rank %d] not pinned: %s\n", rank, mscnn_dist_last_error());
  else if (opt.verbose) fprintf(stderr, "[rank %d] %s\n", rank, placement);
  mscnn_net* net = nullptr;
  NET_CHECK(mscnn_net_create_from_file(opt.prototxt.c_str(), rank, &net));            // also binds this thread to device `rank`
  if (!opt.caffemodel.empty()) NET_CHECK(mscnn_net_load_caffemodel(net, opt.caffemodel.c_str()));
  else seed_weights(net, rank);
  if (opt.precision != "f32") NET_CHECK(mscnn_net_set_precision(net, opt.precision.c_str()));      // "f16" | "f16x3"
  int dims[8], nd = 0;
  NET_CHECK(mscnn_net_blob_shape(net, "data", dims, &nd));
  const int H = dims[2], W = dims[3];
  mscnn_dist* comm = nullptr;
  const size_t pack_bytes = mscnn_net_detect_pack_bytes(opt.cap);
  DIST_CHECK(mscnn_dist_init(id, rank, world, rank, pack_bytes, &comm));
  mscnn_detect_params p;
  std::memset(&p, 0, sizeof(p));
  p.cls_id = opt.cls_id;
  const float stds[4] = {0.1f, 0.1f, 0.2f, 0.2f};
  for (int k = 0; k < 4; ++k) p.bbox_std[k] = stds[k];
  p.proposal_thr = -10.f; p.ratio_h = H / 375.0; p.ratio_w = W / 1242.0; p.org_h = 375; p.org_w = 1242; p.nms_overlap = 0.5;
  std::vector<float> frame;
  std::vector<double> dets((size_t)opt.cap * 5);
  std::vector<int> ids(opt.cap);
  const int steps = (opt.images + world - 1) / world;
  double t0 = 0;
  int sum = 0;
  // the gathered packs of step s: replicas' consistency (warm-up) or rank 0's report
  auto consume = [&](int s, const void* gathered) {
    const char* g = static_cast<const char*>(gathered);
    for (int r = 0; r < world; ++r) {
      int D = 0, R = 0;
      NET_CHECK(mscnn_net_unpack_detections(g + (size_t)r * pack_bytes, opt.cap, dets.data(), ids.data(), &D, &R));
      if (s < 0) {
        // every replica ran image 0 with identical weights: the gathered packs must be byte-identical
        if (std::memcmp(g, g + (size_t)r * pack_bytes, 16 + sizeof(double) * 5 * D) != 0) {
          fprintf(stderr, "[rank %d] replica %d disagrees with replica 0 on image 0\n", rank, r);
          std::exit(4);
        }
      } else if (rank == 0 && s * world + r < opt.images) {
        sum += D;
        printf("image %4d (gpu %d): %4d ROIs -> %4d detections%s", s * world + r, r, R, D, D ? "" : "\n");
        if (D) printf("   best [%.1f %.1f %.1f %.1f] p=%.4f\n", dets[0], dets[1], dets[2], dets[3], dets[4]);
      }
    }
  };
  for (int s = -1; s < steps; ++s) {                       // s = -1: warm-up step on image `rank` (also the consistency check)
    const int k = s < 0 ? 0 : s * world + rank;            // warm-up: every replica runs image 0
    if (s == 0) { DIST_CHECK(mscnn_dist_barrier(comm, nullptr)); t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    make_frame(k, H, W, &frame);
    NET_CHECK(mscnn_net_set_blob(net, "data", frame.data(), frame.size()));
    NET_CHECK(mscnn_net_forward(net));
    const void* pack = nullptr;
    NET_CHECK(mscnn_net_detect_device(net, &p, opt.cap, &pack));
    const void* gathered = nullptr;
    if (s < 0) {                                           // warm-up: one blocking exchange
      DIST_CHECK(mscnn_dist_all_gather(comm, pack, nullptr, &gathered));
      consume(s, gathered);
      continue;
    }
    // pipelined: this step's collective + D2H copy run on the communicator's own stream under the next image's trunk; the host
    // takes step s - 1's packs here (two exchanges in flight at most)
    DIST_CHECK(mscnn_dist_all_gather_begin(comm, pack, nullptr));
    if (s >= 1) {
      DIST_CHECK(mscnn_dist_all_gather_end(comm, &gathered));
      consume(s - 1, gathered);
    }
  }
  if (steps >= 1) {
    const void* gathered = nullptr;
    DIST_CHECK(mscnn_dist_all_gather_end(comm, &gathered));
    consume(steps - 1, gathered);
  }
  DIST_CHECK(mscnn_dist_barrier(comm, nullptr));
  (*seconds)[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
  (*total_dets)[rank] = sum;
  mscnn_dist_destroy(comm);
  mscnn_net_destroy(net);
}

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); } return argv[++i]; };
    if (a == "--gpus") opt.gpus = std::atoi(next());
    else if (a == "--images") opt.images = std::atoi(next());
    else if (a == "--caffemodel") opt.caffemodel = next();
    else if (a == "--precision") opt.precision = next();
    else if (a == "--cls-id") opt.cls_id = std::atoi(next());
    else if (a == "--cap") opt.cap = std::atoi(next());
    else if (a == "--verbose") opt.verbose = true;
    else if (opt.prototxt.empty()) opt.prototxt = a;
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
  }
  if (opt.prototxt.empty()) { fprintf(stderr, "usage: %s deploy.prototxt [--gpus G] [--images K] [--caffemodel f] [--precision p] [--cls-id c] [--cap n]\n", argv[0]); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device\n"); return 1; }
  const int world = opt.gpus > 0 ? (opt.gpus < ndev ? opt.gpus : ndev) : ndev;
  unsigned char id[MSCNN_DIST_ID_BYTES];
  if (mscnn_dist_unique_id(id) != 0) { fprintf(stderr, "%s\n", mscnn_dist_last_error()); return 3; }
  std::vector<double> seconds(world, 0.0);
  std::vector<int> total(world, 0);
  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) threads.emplace_back(worker, r, world, std::cref(opt), id, &seconds, &total);
  for (std::thread& t : threads) t.join();
  double worst = 0;
  for (double s : seconds) worst = s > worst ? s : worst;
  printf("%d images on %d GPU(s): %.3f s, %.1f images/s, %d detections gathered on rank 0\n", opt.images, world, worst,
         opt.images / worst, total[0]);
  return 0;
}
