/*
 * mscnn_net.h -- C ABI of libmscnn_caffe.so: the Caffe-compatible host runtime (caffe::Net<float> and the
 * caffe::Layer<float> subclasses in mscnn_amd/host) for foreign-language callers.  This is what a MATLAB MEX /
 * Python wrapper binds instead of the reference's matcaffe (matlab/+caffe/private/caffe_.cpp:253 get_net,
 * :300-306 net_forward, :57-107 blob get/set) or pycaffe (python/caffe/_caffe.cpp).
 * C++ callers use the caffe:: classes directly (mscnn_amd/host/include/caffe/caffe.hpp).
 *
 * All functions return 0 on success; on failure the text is available from mscnn_net_last_error().
 * A net and everything it owns lives on ONE device and must be used from ONE host thread (the reference's
 * Caffe singleton is thread-local, src/caffe/common.cpp:12-20).
 */
#ifndef MSCNN_NET_H_
#define MSCNN_NET_H_

#include <stddef.h>

#if defined(__GNUC__)
#define MSCNN_NET_API __attribute__((visibility("default")))
#else
#define MSCNN_NET_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mscnn_net mscnn_net;

MSCNN_NET_API const char* mscnn_net_last_error(void);

/* caffe.Net(prototxt, 'test') -- Net::Net(file, TEST), src/caffe/net.cpp:31-46.  device: HIP device ordinal;
 * device < 0 builds the graph without touching a device (shape / naming inspection only; forward then fails). */
MSCNN_NET_API int mscnn_net_create_from_file(const char* prototxt_path, int device, mscnn_net** out);
MSCNN_NET_API int mscnn_net_create_from_string(const char* prototxt_text, int device, mscnn_net** out);
/* flags: MSCNN_NET_NO_FUSION builds the net without the Net-level operator fusion (every layer runs its own kernel, as in the
 * reference; results are bit-identical, roi_pool_org / roi_pool_ctx style intermediate blobs are written eagerly). */
#define MSCNN_NET_NO_FUSION 1
MSCNN_NET_API int mscnn_net_create_from_string_ex(const char* prototxt_text, int device, unsigned flags, mscnn_net** out);
MSCNN_NET_API void mscnn_net_destroy(mscnn_net* net);
/* net.copy_from(weights) -- Net::CopyTrainedLayersFrom, net.cpp:750-803: a binary NetParameter (.caffemodel), or -- for a path
 * ending in ".h5", like the reference (net.cpp:788-795) -- an HDF5 snapshot (Net::CopyTrainedLayersFromHDF5, net.cpp:806-848). */
MSCNN_NET_API int mscnn_net_load_caffemodel(mscnn_net* net, const char* path);
/* HIP stream (hipStream_t as void*) for every subsequent call on this thread; NULL = default stream. */
MSCNN_NET_API int mscnn_net_set_stream(void* stream);

/* graph introspection (Net::layer_names / blob_names / layers()[i]->blobs()) */
MSCNN_NET_API int mscnn_net_num_layers(const mscnn_net* net);
MSCNN_NET_API const char* mscnn_net_layer_name(const mscnn_net* net, int layer);
MSCNN_NET_API const char* mscnn_net_layer_type(const mscnn_net* net, int layer);
MSCNN_NET_API int mscnn_net_layer_index(const mscnn_net* net, const char* name);           /* -1 if absent */
MSCNN_NET_API int mscnn_net_layer_num_bottoms(const mscnn_net* net, int layer);
MSCNN_NET_API int mscnn_net_layer_num_tops(const mscnn_net* net, int layer);
MSCNN_NET_API const char* mscnn_net_layer_bottom(const mscnn_net* net, int layer, int i);  /* blob name */
MSCNN_NET_API const char* mscnn_net_layer_top(const mscnn_net* net, int layer, int i);
MSCNN_NET_API int mscnn_net_layer_num_params(const mscnn_net* net, int layer);
MSCNN_NET_API int mscnn_net_layer_param_shape(const mscnn_net* net, int layer, int param, int* dims8, int* ndim);
/* The layer's LayerParameter in prototxt text form (valid until the next call on this thread). */
MSCNN_NET_API const char* mscnn_net_layer_param_text(const mscnn_net* net, int layer);
MSCNN_NET_API int mscnn_net_layer_fused_away(const mscnn_net* net, int layer);              /* 1: ReLU folded into its producer */
MSCNN_NET_API const char* mscnn_net_layer_kernel(const mscnn_net* net, int layer);          /* conv / InnerProduct kernel family, "" otherwise */
MSCNN_NET_API double mscnn_net_layer_flops(const mscnn_net* net, int layer);                /* of the last forward */
/* Roofline accounting of Convolution layers: FLOPs the MFMA pipe executes (Winograd forms: fewer than the algorithmic
 * count above) and, with conv profiling on, the HIP-event time of the last forward split into {input transform, MFMA GEMM
 * kernels, output transform} (direct kernels: {0, total, 0}); non-convolution layers report zeros. */
MSCNN_NET_API double mscnn_net_layer_executed_flops(const mscnn_net* net, int layer);
MSCNN_NET_API int mscnn_net_set_conv_profiling(mscnn_net* net, int on);
MSCNN_NET_API int mscnn_net_layer_stage_ms(const mscnn_net* net, int layer, float ms_out[3]);
/* Convolution algorithm of one layer (layer < 0: all): mscnn_conv_algo of include/mscnn_hip.h (0 auto, 1 direct,
 * 2 / 3 Winograd F(2x2,3x3) / F(3x3,3x3) wherever legal); tuning knobs = mscnn_conv_desc::tune_* (A/B runs). */
MSCNN_NET_API int mscnn_net_set_conv_algo(mscnn_net* net, int layer, int algo);
MSCNN_NET_API int mscnn_net_set_conv_tuning(mscnn_net* net, int layer, int variant, int grid, int flags);
/* fp32 kernel of one InnerProduct layer (layer < 0: all): 0 = auto (fc6-class shapes on the plane-GEMM kernel, mscnn_inner_product_wg_*),
 * 1 = always the stream-K kernel of gemm.hip (A/B runs, second witness).  mscnn_net_layer_kernel reports what the last forward ran. */
MSCNN_NET_API int mscnn_net_set_inner_product_algo(mscnn_net* net, int layer, int algo);
/* Arithmetic of the MFMA layers of the whole net: "f32" (default: the path that matches the reference within 1e-4) or "f16"
 * (fp16 operands, fp32 accumulate, no reference counterpart -- BASELINE config 5): 3x3 stride-1 convolutions and InnerProduct
 * layers with N >= 64 switch, everything else keeps its fp32 kernel.  "f16x3": the convolutions that run Winograd F(3x3,3x3)
 * multiply their planes on the fp16 MFMA pipe with every fp32 operand split exactly into fp16 hi + lo (three products, fp32
 * accumulate; MSCNN_CONV_ALGO_WINO_F3_X3) -- fp32-grade results held to the SAME parity gates as "f32", max |x| handed from
 * layer to layer on the device.  mscnn_net_layer_dtype: what layer i really runs. */
MSCNN_NET_API int mscnn_net_set_precision(mscnn_net* net, const char* dtype);
MSCNN_NET_API const char* mscnn_net_layer_dtype(const mscnn_net* net, int layer);
/* Numerics are SAFE BY DEFAULT (no call needed; the reference's flow -- Net(prototxt, TEST); CopyTrainedLayersFrom; Forward,
 * net.cpp:750-785, 544-555 -- stays as it is): the Winograd forms AUTO picks carry ~10x the rounding error of the direct sum, so
 * every Convolution layer compares its Winograd result with the direct k-ordered kernel on the FIRST bottom it sees after
 * construction or a weight change (mscnn_net_set_param, mscnn_net_copy_trained_from, Layer::OnWeightsChanged), metric
 * max |dy| / max(1, |y|, rms(y)), tolerance 5e-5, and when it is off switches to the direct kernel and recomputes its tops before
 * that forward returns.  Cost: the first forward after a weight change runs each Winograd layer twice.
 *   mscnn_net_set_auto_calibrate(net, tol): tol = 0 is the opt-OUT; tol > 0 re-arms every layer's check with that tolerance.
 *   mscnn_net_auto_calibrate_state: *checks = first-forward checks done so far; returns how many layers fell back and writes up to
 *   cap of their indices; mscnn_net_layer_calibration_err gives each layer's last measured value.
 * mscnn_net_calibrate_numerics is the same check on demand (call after a forward on representative input): every convolution that
 * runs a Winograd form is re-computed with the direct kernel on the same bottom; layers over tol run the direct kernel from then on.
 * *num_switched = how many. */
MSCNN_NET_API int mscnn_net_set_auto_calibrate(mscnn_net* net, double tol);
/* Chains of same-resolution F(4x4,3x3) convolutions (conv2_1 -> conv2_2, conv3_1 -> 3_2 -> 3_3, conv4_1 -> 4_2 -> 4_3; mscnn_hip.h:
 * mscnn_conv2d_fwd_chain_f32): inside a forward the blob between two members is not written -- the producer's output stage writes the
 * consumer's transform planes.  mscnn_net_get_blob / _blob_device on such a blob re-run its producer on demand (bit-identical to the
 * unchained forward; the reference's contract that every blob holds its layer's output after Forward, net.cpp:544-555, is kept for
 * every reader that goes through this ABI).  ON by default (with the other Net-level fusions); on = 0 writes every blob in every
 * forward (MSCNN_NO_CHAIN=1 does the same process-wide). */
MSCNN_NET_API int mscnn_net_set_chain_fusion(mscnn_net* net, int on);
/* The pairs the net registered at construction: producers[i] -> consumers[i] (layer indices; consumers[i] = -1: a top that only its
 * fused 2x2 pooling reads).  Returns their number and writes up to cap of them; whether a pair actually runs chained is decided per
 * forward (both planned kernels on the fp32 F(4x4,3x3) path, no numerical check pending, both layers inside the range). */
MSCNN_NET_API int mscnn_net_chain_pairs(const mscnn_net* net, int* producers, int* consumers, int cap);
MSCNN_NET_API int mscnn_net_auto_calibrate_state(const mscnn_net* net, int* checks, int* switched_layers, int cap);
MSCNN_NET_API int mscnn_net_calibrate_numerics(mscnn_net* net, double tol, int* num_switched);
MSCNN_NET_API double mscnn_net_layer_calibration_err(const mscnn_net* net, int layer);
/* The same comparison on live frames (the numerics watch): every period-th whole forward looks at ONE Winograd layer (round robin) --
 * its bottom and top blobs are written in that frame (it stays in its convolution chain), and one band of it (a few rows / images: ~30 us of direct-kernel work; round robin too) is
 * recomputed with the direct kernel BEHIND the frame on the same stream, with no host synchronisation; the verdict is collected by a
 * later forward, and a layer off by more than tol runs the direct kernel from the frame after.  No frame waits for a check: a watch
 * frame is 2 - 8 % longer (7s-576), the others not at all.  ON by default with period 25, tol 5e-5 (~0.1 % of a stream); period 0
 * switches the watch off.  _state: waits for a verdict that is still out, then *checks = band checks done so far; returns how many
 * layers were switched and writes up to cap of their indices. */
MSCNN_NET_API int mscnn_net_set_numerics_watch(mscnn_net* net, int period, double tol);
MSCNN_NET_API int mscnn_net_numerics_watch_state(const mscnn_net* net, int* checks, int* switched_layers, int cap);
MSCNN_NET_API int mscnn_net_num_blobs(const mscnn_net* net);
MSCNN_NET_API const char* mscnn_net_blob_name(const mscnn_net* net, int blob);
MSCNN_NET_API int mscnn_net_blob_shape(const mscnn_net* net, const char* name, int* dims8, int* ndim);
MSCNN_NET_API int mscnn_net_num_inputs(const mscnn_net* net);
MSCNN_NET_API int mscnn_net_num_outputs(const mscnn_net* net);
MSCNN_NET_API const char* mscnn_net_output_name(const mscnn_net* net, int i);               /* alphabetical, net.cpp:267-274 */

/* weights: layer->blobs()[param]  (host fp32, Caffe layout).  The setter marks the layer's packed copy stale. */
MSCNN_NET_API int mscnn_net_set_param(mscnn_net* net, int layer, int param, const float* host, size_t count);
MSCNN_NET_API int mscnn_net_get_param(mscnn_net* net, int layer, int param, float* host, size_t count);

/* blob.set_data / get_data.  *_device copy device->device on the net's stream (no PCIe). */
MSCNN_NET_API int mscnn_net_set_blob(mscnn_net* net, const char* name, const float* host, size_t count);
MSCNN_NET_API int mscnn_net_set_blob_device(mscnn_net* net, const char* name, const float* dev, size_t count);
/* The MATLAB pre-processing in front of net.forward (run_mscnn_detection.m:64-69), on the device: a decoded uint8 RGB frame
 * [org_h][org_w][3] (host memory, or device memory when on_device != 0) is resized to the input blob's H x W (imresize
 * semantics), swapped to BGR, mean-subtracted (mean_bgr == NULL: {104,117,123}) and written into blob `name`. */
MSCNN_NET_API int mscnn_net_set_image(mscnn_net* net, const char* name, const unsigned char* img_rgb, int on_device,
                                      int org_h, int org_w, const float* mean_bgr);
MSCNN_NET_API int mscnn_net_get_blob(mscnn_net* net, const char* name, float* host, size_t capacity, size_t* count);
MSCNN_NET_API const float* mscnn_net_blob_device_ptr(mscnn_net* net, const char* name);

/* net.forward_prefilled(): Net::ForwardFromTo(0, L-1), net.cpp:544-575.  from/to: layer range, (0, -1) = all. */
MSCNN_NET_API int mscnn_net_forward(mscnn_net* net);
MSCNN_NET_API int mscnn_net_forward_from_to(mscnn_net* net, int from, int to);
MSCNN_NET_API int mscnn_net_reshape(mscnn_net* net);
/* Input reshape: blob->Reshape(dims) on a net input followed by Net::Reshape() (net.cpp:743-747; matcaffe's
 * net.blobs('data').reshape([w h c n]); net.reshape()) -- another batch size (dims[0]) or frame size without re-creating the net.
 * dims in Caffe's order (N, C, H, W).  The whole path is batch-generic like the reference's (conv_layer.cpp:25-40 loops over num_,
 * box_output_layer.cpp:107 over images, roi_pooling_layer.cpp:62-66 takes the image from column 0 of every ROI): a forward of
 * N images returns the ROI blobs grouped by image; the final stage then runs once per image (mscnn_net_detect_image). */
MSCNN_NET_API int mscnn_net_reshape_input(mscnn_net* net, const char* name, const int* dims, int ndim);
/* Health of the plane-GEMM kernel's stream-K hand-off (mscnn_hip.h: mscnn_wgemm_handoff_event).  A hand-off that times out (a
 * workgroup of the persistent grid was not co-resident: CU masking, a partition mode, another stream's kernels) can never hand out a
 * wrong frame through this ABI: the Net reads the kernel's status word wherever it synchronises anyway -- behind BoxOutput's
 * row-count read inside a forward, behind the copy of mscnn_net_detect and of mscnn_net_get_blob -- and on an event forces
 * whole-tile scheduling for the rest of the process and runs the frame again before the call returns (cost: one frame twice, once).
 * Returns how many events this net has answered so far; *whole_tiles_forced (may be NULL) = 1 once the process runs on whole tiles.
 * (mscnn_net_detect_device + the RCCL exchange, and C++ callers that read Blob::cpu_data() themselves, synchronise outside the Net:
 * they call caffe::Net::HandoffRecover() / look at mscnn_wgemm_handoff_event() behind their own synchronisation.) */
MSCNN_NET_API int mscnn_net_handoff_state(const mscnn_net* net, int* whole_tiles_forced);
/* `caffe time`-style per-layer HIP-event timing (tools/caffe.cpp:380-400); serialises the layers. */
MSCNN_NET_API int mscnn_net_set_layer_timing(mscnn_net* net, int on);
MSCNN_NET_API float mscnn_net_layer_ms(const mscnn_net* net, int layer);

/*
 * Final detection stage on the net's own output blobs (bbox_pred, cls_pred, proposals_score), on the device:
 * the MATLAB post-processing of run_mscnn_detection.m:75-120 + bbNms.m:112-126.  Only the detections
 * (dets_host[cap][5] doubles [x y w h prob], ids_host[cap] ROI row per detection) cross PCIe.
 */
typedef struct {
  int cls_id;                       /* 1-based class column (car = 2 for the KITTI car nets) */
  float bbox_mean[4], bbox_std[4];  /* [0 0 0 0], [.1 .1 .2 .2] in the scripts */
  float proposal_thr;               /* -10 */
  double ratio_h, ratio_w;          /* net input size / original image size */
  double org_h, org_w;              /* original image size (clip bounds) */
  double nms_overlap;               /* 0.5 */
} mscnn_detect_params;
MSCNN_NET_API int mscnn_net_detect(mscnn_net* net, const mscnn_detect_params* p, double* dets_host, int* ids_host,
                                   int cap, int* num_dets, int* num_rois);
/* The same stage for a STREAM of frames (batch-1 nets), pipelined: _begin runs the final stage of the forward just done into the
 * device pack, enqueues ONE copy of it into one of two pinned host slots behind an event on the net's stream, and returns at once --
 * the caller goes on with the next frame (set_blob / forward); _end waits for the OLDEST frame in flight and unpacks it: frame i's
 * detections reach the host under frame i + 1's trunk instead of idling the device for a host round trip per frame.  At most two
 * frames in flight; cap as in mscnn_net_detect_device (BoxOutput's max_nms_num bounds R).  Results are bit-identical to
 * mscnn_net_detect.  A stream-K hand-off time-out while frames are in flight can not be answered by a re-run here (the input blob
 * already holds a later frame): _end then FAILS, naming it, after forcing whole-tile scheduling -- submit the frames begun since the
 * last _end again. */
MSCNN_NET_API int mscnn_net_detect_begin(mscnn_net* net, const mscnn_detect_params* p, int cap);
MSCNN_NET_API int mscnn_net_detect_end(mscnn_net* net, double* dets_host, int* ids_host, int cap, int* num_dets, int* num_rois);
/* The same stage for image `image` of a batched forward (the MATLAB stage is per image: its NMS never mixes images; mscnn_net_detect
 * itself refuses a net whose input holds more than one image).  ids_host = rows of the net's ROI blobs (bbox_pred, cls_pred,
 * proposals_score), *num_rois = the image's own ROI count. */
MSCNN_NET_API int mscnn_net_detect_image(mscnn_net* net, const mscnn_detect_params* p, int image, double* dets_host, int* ids_host,
                                         int cap, int* num_dets, int* num_rois);


/* Final stage of the cascade drivers (examples/kitti_car/run_cascademscnn.m:84-127) for ONE cascade output nn: the decoded boxes
 * of stage nn (`bbox_blob`, e.g. "output_bbox_3rd"), its in-net probabilities (`prob_blob`, "cls_prob_3rd" / "cls_prob_3rd_avg")
 * and the proposals it refined (`proposal_blob`, "proposals_3rd"), all read from the net on the device: rescale, clip, drop
 * zero-size proposals, optional det_thr (> 0), bbNms.  Same outputs as mscnn_net_detect; p->bbox_mean/std, proposal_thr unused. */
MSCNN_NET_API int mscnn_net_detect_cascade(mscnn_net* net, const mscnn_detect_params* p, float det_thr, const char* bbox_blob,
                                           const char* prob_blob, const char* proposal_blob, double* dets_host, int* ids_host,
                                           int cap, int* num_dets, int* num_rois);

/* Multi-GPU form of the same stage (include/mscnn_dist.h gathers its result over RCCL): the detections stay in HBM, in a
 * fixed-size pack  [int32 count, int32 num_rois, int32 cap, int32 0][cap x 5 doubles x y w h prob][cap x int32 roi row]
 * of mscnn_net_detect_pack_bytes(cap) bytes (a multiple of 16).  cap must be >= the net's ROI count (BoxOutput's
 * max_nms_num bounds it); more ROIs than cap is an error, never a truncation.  Asynchronous on the net's stream;
 * *pack_dev stays valid until the next detect call on this net.  mscnn_net_unpack_detections reads one pack on the host. */
MSCNN_NET_API size_t mscnn_net_detect_pack_bytes(int cap);
MSCNN_NET_API int mscnn_net_detect_device(mscnn_net* net, const mscnn_detect_params* p, int cap, const void** pack_dev);
MSCNN_NET_API int mscnn_net_unpack_detections(const void* pack_host, int cap, double* dets_host, int* ids_host, int* num_dets,
                                              int* num_rois);

#ifdef __cplusplus
}
#endif
#endif
