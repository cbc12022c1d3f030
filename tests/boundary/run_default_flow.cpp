// The flow INTEGRATION.md section 2 shows, verbatim, as a program: construct the Net from a prototxt, CopyTrainedLayersFrom a
// .caffemodel, fill the input, Forward -- and NOTHING else (no calibration call, no algorithm knob).  The drop-in must be safe
// exactly like this (reference flow: src/caffe/net.cpp:750-785 + :544-555): every Winograd layer checks itself against the direct
// kernel on this first frame and falls back before Forward returns.
//   run_default_flow <prototxt> <caffemodel> <input.f32> <outdir> [auto_calibrate_tol]
// Writes <outdir>/<blob>.f32 for the blobs listed below and prints one "LAYER <name> <kernel>" line per convolution, then
// "AUTOCAL checks <n> switched <k>".  The optional 5th argument is the documented knob (Net::SetAutoCalibrate) -- used by the test to
// force every layer over the tolerance and see the fall-back happen inside the first Forward.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "caffe/caffe.hpp"
#include "caffe/layers/mscnn_layers.hpp"

using namespace caffe;

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s prototxt caffemodel input.f32 outdir [tol]\n", argv[0]); return 2; }
  Caffe::set_mode(Caffe::GPU);
  Caffe::SetDevice(0);
  Net<float> net(argv[1], TEST);
  if (argc > 5) net.SetAutoCalibrate(std::atof(argv[5]));
  net.CopyTrainedLayersFrom(argv[2]);
  Blob<float>* in = net.input_blobs()[0];
  FILE* f = std::fopen(argv[3], "rb");
  if (!f || std::fread(in->mutable_cpu_data(), sizeof(float), in->count(), f) != (size_t)in->count()) { std::fprintf(stderr, "bad input file\n"); return 2; }
  std::fclose(f);
  const std::vector<Blob<float>*>& out = net.Forward();
  std::printf("OUTPUTS %d\n", (int)out.size());
  for (size_t i = 0; i < net.layers().size(); ++i)
    if (ConvolutionLayer<float>* c = dynamic_cast<ConvolutionLayer<float>*>(net.layers()[i].get()))
      std::printf("LAYER %s %s %g\n", net.layer_names()[i].c_str(), c->kernel_name(), net.calibration_err()[i]);
  std::printf("AUTOCAL checks %d switched %d\n", net.auto_calibrate_checks(), (int)net.auto_calibrate_switched().size());
  const std::vector<std::string>& names = net.blob_names();
  for (size_t b = 0; b < names.size(); ++b) {
    const std::string& nm = names[b];
    if (nm.find("split") != std::string::npos || nm == "data") continue;
    const shared_ptr<Blob<float> > blob = net.blob_by_name(nm);
    const std::string path = std::string(argv[4]) + "/" + nm + ".f32";
    FILE* o = std::fopen(path.c_str(), "wb");
    if (!o) return 2;
    const int nd = blob->num_axes();
    std::fwrite(&nd, sizeof(int), 1, o);
    for (int d = 0; d < nd; ++d) { const int v = blob->shape(d); std::fwrite(&v, sizeof(int), 1, o); }
    if (blob->count()) std::fwrite(blob->cpu_data(), sizeof(float), blob->count(), o);
    std::fclose(o);
  }
  std::printf("DEFAULT FLOW OK\n");
  return 0;
}
