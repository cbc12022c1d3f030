// DecodeBBox and the final detection stage for gfx950.
//
// DecodeBBox: DecodeBBoxLayer<Dtype>::Forward_cpu (decode_bbox_layer.cpp:54-123) +
//   DecodeBBoxesWithPrior (math_functions.cpp:46-75), TEST phase (no filtering).  CPU-only in the
//   reference (decode_bbox_layer.hpp:41-42).
// Final stage: the MATLAB post-processing of the net outputs, examples/kitti_car/run_mscnn_detection.m:75-120
//   + utils/bbNms.m:112-126 (nmsMax, greedy, union).  MATLAB semantics restated: single-precision
//   arithmetic until `double([...])`, stable descending sort on prob (ties: lower row first), IoU and
//   threshold test in double, pairs with iw <= 0 or ih <= 0 skipped.
// Same kernel structure as BoxOutput's NMS: parallel bit-matrix + one-wavefront greedy scan.
#include "common.h"
#include "box_device.h"
#include "nms_large.h"

namespace {
using namespace mscnn_dev;

__global__ __launch_bounds__(256) void decode_bbox_kernel(const float* __restrict__ bbox, const float* __restrict__ prior,
                                                          float* __restrict__ out, int R, int bbox_dim, float m0, float m1,
                                                          float m2, float m3, float s0, float s1, float s2, float s3) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const float xmin = prior[i * 5 + 1], ymin = prior[i * 5 + 2], xmax = prior[i * 5 + 3], ymax = prior[i * 5 + 4];
  const float pw = xmax - xmin + 1, ph = ymax - ymin + 1;
  const float cx = (float)(0.5 * (double)(xmax + xmin)), cy = (float)(0.5 * (double)(ymax + ymin));
  const float* b = bbox + (size_t)i * bbox_dim + 4;          // class 1 columns (decode_bbox_layer.cpp:116)
  const float bx = b[0] * s0 + m0, by = b[1] * s1 + m1, bw = b[2] * s2 + m2, bh = b[3] * s3 + m3;
  float tx = bx * pw + cx, ty = by * ph + cy;
  const float tw = pw * expf_libm(bw), th = ph * expf_libm(bh);
  tx -= (tw - 1) / 2; ty -= (th - 1) / 2;
  out[i * 5 + 0] = prior[i * 5];
  out[i * 5 + 1] = tx; out[i * 5 + 2] = ty; out[i * 5 + 3] = tx + tw - 1; out[i * 5 + 4] = ty + th - 1;
}

struct DetBox { double x, y, w, h; };

struct DetArgs {
  const float* bbox_pred; const float* cls_pred; const float* props;
  int R, ncls, cls_id;
  float mean[4], stdv[4];
  float proposal_thr, ratio_h, ratio_w, org_h, org_w;
  int cascade;        // 1: run_cascademscnn.m:84-117 -- bbox_pred = decoded boxes [R][5], cls_pred = in-net probabilities,
                      //    props = proposal rows [R][5]; proposal_thr = det_thr
};

enum { DC_N = 0, DC_WORDS = 4, DC_BIG = 4 /* + BIG_STATE_WORDS (nms_large.h) */ };

// One input row -> box and prob in MATLAB's types.  false: the row is filtered out.
__device__ __forceinline__ bool det_row(const DetArgs& a, int r, DetBox* box, float* prob_out) {
  if (a.cascade) {
    const float* q = a.props + 5 * (size_t)r;
    const float cw = q[3] - q[1] + 1.f, ch = q[4] - q[2] + 1.f;                      // run_cascademscnn.m:104
    if (!(cw != 0 && ch != 0)) return false;                                         // :107
    const float* t = a.bbox_pred + 5 * (size_t)r;
    float x1 = t[1] / a.ratio_w, x2 = t[3] / a.ratio_w;                              // :88-89
    float y1 = t[2] / a.ratio_h, y2 = t[4] / a.ratio_h;
    x1 = fmaxf(0.f, x1); y1 = fmaxf(0.f, y1);                                        // :91
    x2 = fminf(x2, a.org_w); y2 = fminf(y2, a.org_h);                                // :92
    const float w = x2 - x1 + 1.f, h = y2 - y1 + 1.f;                                // :93
    const float prob = a.cls_pred[(size_t)r * a.ncls + (a.cls_id - 1)];              // :113
    if (a.proposal_thr > 0 && !(prob >= a.proposal_thr)) return false;               // :115-117 (det_thr)
    if (!(prob > -INFINITY)) return false;
    *box = DetBox{(double)x1, (double)y1, (double)w, (double)h};
    *prob_out = prob;
    return true;
  }
  const float* q = a.props + 6 * (size_t)r;
  const float px = q[1], py = q[2], pw = q[3] - q[1], ph = q[4] - q[2], sc = q[5];
  if (!(sc >= a.proposal_thr && pw != 0 && ph != 0)) return false;                   // :82
  const float* bp = a.bbox_pred + (size_t)r * 4 * a.ncls + 4 * (a.cls_id - 1);      // :95
  float b0 = bp[0] * a.stdv[0], b1 = bp[1] * a.stdv[1], b2 = bp[2] * a.stdv[2], b3 = bp[3] * a.stdv[3];
  b0 += a.mean[0]; b1 += a.mean[1]; b2 += a.mean[2]; b3 += a.mean[3];
  const float* cp = a.cls_pred + (size_t)r * a.ncls;
  float se = 0.f;
  for (int k = 0; k < a.ncls; ++k) se += expf_libm(cp[k]);                           // :101-102
  const float prob = expf_libm(cp[a.cls_id - 1]) / se;
  const float ctr_x = px + 0.5f * pw, ctr_y = py + 0.5f * ph;
  float tx = b0 * pw + ctr_x, ty = b1 * ph + ctr_y;
  float tw = pw * expf_libm(b2), th = ph * expf_libm(b3);
  tx = tx - tw / 2.f; ty = ty - th / 2.f;
  tx = tx / a.ratio_w; tw = tw / a.ratio_w;
  ty = ty / a.ratio_h; th = th / a.ratio_h;
  tx = fmaxf(0.f, tx); ty = fmaxf(0.f, ty);
  tw = fminf(tw, a.org_w - tx); th = fminf(th, a.org_h - ty);
  if (!(prob > -INFINITY)) return false;                                             // bbNms.m:76 (NaN drops out)
  *box = DetBox{(double)tx, (double)ty, (double)tw, (double)th};
  *prob_out = prob;
  return true;
}

// stable descending: larger prob first, then LOWER row first (never 0: orderable(x) > 0 for every x > -inf)
__device__ __forceinline__ u64 det_key(float prob, int r) {
  return ((u64)orderable(prob) << 32) | (u64)(0xffffffffu - (unsigned)r);
}

// One workgroup: per-row transform + filter, key sort, write sorted boxes.
__global__ __launch_bounds__(kSortThreads) void det_transform_sort_kernel(DetArgs a, DetBox* __restrict__ sbox,
                                                                          double* __restrict__ sprob, int* __restrict__ ssrc,
                                                                          DetBox* __restrict__ tmp_box,
                                                                          float* __restrict__ tmp_prob, int* __restrict__ cnt) {
  __shared__ u64 sk[kSortCap];
  __shared__ int s_fill;
  const int tid = threadIdx.x;
  if (tid == 0) s_fill = 0;
  __syncthreads();
  for (int r = tid; r < a.R; r += kSortThreads) {
    DetBox b;
    float prob;
    if (!det_row(a, r, &b, &prob)) continue;
    tmp_box[r] = b;
    tmp_prob[r] = prob;
    const int pos = atomicAdd(&s_fill, 1);
    if (pos < kMaxK) sk[pos] = det_key(prob, r);
  }
  __syncthreads();
  const int n = min(s_fill, kMaxK);
  if (tid == 0) cnt[DC_N] = n;
  if (n == 0) return;
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = n + tid; i < P; i += kSortThreads) sk[i] = 0ull;
  __syncthreads();
  bitonic_desc(sk, P, tid, kSortThreads);
  for (int i = tid; i < n; i += kSortThreads) {
    const int r = (int)(0xffffffffu - (unsigned)(sk[i] & 0xffffffffull));
    sbox[i] = tmp_box[r];
    sprob[i] = (double)tmp_prob[r];
    ssrc[i] = r;
  }
}

// ---- more than kMaxK rows (nms_large.h): the same three steps over HBM-resident lists -------------------------------------------
// keys[r] = the row's key, or 0 (padding, sorts last) when it is filtered out; cnt[DC_N] counts the survivors
__global__ __launch_bounds__(256) void det_transform_big_kernel(DetArgs a, u64* __restrict__ keys, DetBox* __restrict__ tmp_box,
                                                                float* __restrict__ tmp_prob, int* __restrict__ cnt) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.R) return;
  DetBox b;
  float prob;
  if (!det_row(a, r, &b, &prob)) return;          // keys[] was cleared
  tmp_box[r] = b;
  tmp_prob[r] = prob;
  keys[r] = det_key(prob, r);
  atomicAdd(&cnt[DC_N], 1);
}

__global__ __launch_bounds__(256) void det_gather_big_kernel(const u64* __restrict__ keys, const DetBox* __restrict__ tmp_box,
                                                             const float* __restrict__ tmp_prob, DetBox* __restrict__ sbox,
                                                             double* __restrict__ sprob, int* __restrict__ ssrc,
                                                             const int* __restrict__ cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt[DC_N]) return;
  const int r = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
  sbox[i] = tmp_box[r];
  sprob[i] = (double)tmp_prob[r];
  ssrc[i] = r;
}

struct DetTr {
  typedef DetBox Box;
  struct Params { double overlap; };
  // == det_mask_kernel's test (A the earlier box): bbNms.m:117-124, all in double
  static __device__ __forceinline__ bool over(const DetBox& A, const DetBox& B, const Params& p) {
    const double iw = fmin(A.x + A.w, B.x + B.w) - fmax(A.x, B.x);
    if (iw <= 0) return false;
    const double ih = fmin(A.y + A.h, B.y + B.h) - fmax(A.y, B.y);
    if (ih <= 0) return false;
    double o = iw * ih;
    const double u = A.w * A.h + B.w * B.h - o;
    o = o / u;
    return o > p.overlap;
  }
};

__global__ __launch_bounds__(256) void det_emit_big_kernel(const int* __restrict__ kept_idx, const int* __restrict__ state,
                                                           const DetBox* __restrict__ sbox, const double* __restrict__ sprob,
                                                           const int* __restrict__ ssrc, double* __restrict__ dets,
                                                           int* __restrict__ ids, int* __restrict__ count_out) {
  const int nk = state[BIG_NKEPT];
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row == 0) count_out[0] = nk;
  if (row >= nk) return;
  const int k = kept_idx[row];
  const DetBox b = sbox[k];
  double* d = dets + 5 * (size_t)row;
  d[0] = b.x; d[1] = b.y; d[2] = b.w; d[3] = b.h; d[4] = sprob[k];
  if (ids) ids[row] = ssrc[k];
}

// (256 threads per 64 x 64 block, the four waves split the columns -- as nms_mask_kernel of boxoutput.hip)
__global__ __launch_bounds__(256) void det_mask_kernel(const DetBox* __restrict__ boxes, const int* __restrict__ cnt,
                                                       double overlap, u64* __restrict__ mask, int wpr) {
  const int n = cnt[DC_N];
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  __shared__ DetBox cbox[64];
  __shared__ unsigned part[4][64];
  const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = cb * 64;
  if (wv == 0 && j0 + t < n) cbox[t] = boxes[j0 + t];
  __syncthreads();
  const int i = rb * 64 + t;
  unsigned bits = 0;
  if (i < n) {
    const DetBox A = boxes[i];
    const double as_a = A.w * A.h, xe_a = A.x + A.w, ye_a = A.y + A.h;
    const int jn = min(64, n - j0);
    for (int q = wv * 16; q < wv * 16 + 16; ++q) {
      if (q >= jn || j0 + q <= i) continue;
      const DetBox B = cbox[q];
      const double iw = fmin(xe_a, B.x + B.w) - fmax(A.x, B.x);
      if (iw <= 0) continue;
      const double ih = fmin(ye_a, B.y + B.h) - fmax(A.y, B.y);
      if (ih <= 0) continue;
      double o = iw * ih;
      const double u = as_a + B.w * B.h - o;
      o = o / u;
      if (o > overlap) bits |= 1u << (q & 15);
    }
  }
  part[wv][t] = bits;
  __syncthreads();
  if (wv == 0 && i < n)
    mask[(size_t)i * wpr + cb] = (u64)part[0][t] | ((u64)part[1][t] << 16) | ((u64)part[2][t] << 32) | ((u64)part[3][t] << 48);
}

__global__ __launch_bounds__(256) void det_scan_emit_kernel(const u64* __restrict__ mask, int wpr,
                                                            const DetBox* __restrict__ sbox, const double* __restrict__ sprob,
                                                            const int* __restrict__ ssrc, double* __restrict__ dets,
                                                            int* __restrict__ ids, const int* __restrict__ cnt,
                                                            int* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) u64 dyn_lds[];
  __shared__ u64 keepw[64];
  __shared__ int pre[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = cnt[DC_N];
  if (n <= 0) { if (tid == 0) count_out[0] = 0; return; }
  const u64 mykeep = greedy_scan(mask, n, wpr, wpr, dyn_lds);
  if (wave == 0) {
    keepw[lane] = mykeep;
    const int mine = __popcll(mykeep);
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    pre[lane] = incl - mine;
    if (lane == 63) count_out[0] = incl;
  }
  __syncthreads();
  for (int k = tid; k < n; k += 256) {
    const int c = k >> 6, l = k & 63;
    const u64 kw = keepw[c];
    if (!((kw >> l) & 1ull)) continue;
    const int row = pre[c] + __popcll(kw & ((1ull << l) - 1ull));
    const DetBox b = sbox[k];
    double* d = dets + 5 * (size_t)row;
    d[0] = b.x; d[1] = b.y; d[2] = b.w; d[3] = b.h; d[4] = sprob[k];
    if (ids) ids[row] = ssrc[k];
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct DetLayout { size_t cnt, sbox, sprob, ssrc, tbox, tprob, mask, keys, rinit, kidx, kbox, total; int wpr, sortP; bool big; };
DetLayout det_layout(int R) {
  DetLayout L;
  const int n = R < 1 ? 1 : R;
  L.big = n > kMaxK;                         // more rows than the LDS-resident path holds: sort in HBM, tiled NMS
  L.wpr = L.big ? kTileWords : (n + 63) / 64;
  L.sortP = L.big ? big_sort_pow2(n) : 0;
  size_t o = 0;
  L.cnt = o; o += 256;                       // [DC_N ...] and, at DC_BIG, the tiled path's state words
  L.sbox = o; o += align_up((size_t)n * sizeof(DetBox), 256);
  L.sprob = o; o += align_up((size_t)n * sizeof(double), 256);
  L.ssrc = o; o += align_up((size_t)n * sizeof(int), 256);
  L.tbox = o; o += align_up((size_t)n * sizeof(DetBox), 256);
  L.tprob = o; o += align_up((size_t)n * sizeof(float), 256);
  L.mask = o; o += align_up((size_t)(L.big ? kMaxK : n) * L.wpr * sizeof(u64), 256);
  L.keys = o; o += L.big ? align_up((size_t)L.sortP * sizeof(u64), 256) : 0;
  L.rinit = o; o += L.big ? align_up(64 * sizeof(u64), 256) : 0;
  L.kidx = o; o += L.big ? align_up((size_t)n * sizeof(int), 256) : 0;
  L.kbox = o; o += L.big ? align_up((size_t)n * sizeof(DetBox), 256) : 0;
  L.total = o;
  return L;
}
}  // namespace

using namespace mscnn;

extern "C" int mscnn_decodebbox_fwd_f32(const float* bbox, const float* prior, float* out, int R, int bbox_dim,
                                        const float* mean_host, const float* std_host, void* stream) {
  MSCNN_REQUIRE(R >= 0, "decodebbox: R < 0");
  MSCNN_REQUIRE(bbox_dim == 8, "decodebbox: bbox channels must be 8 (decode_bbox_layer.cpp:47), got %d", bbox_dim);
  if (R == 0) return MSCNN_OK;
  MSCNN_REQUIRE(bbox && prior && out && mean_host && std_host, "decodebbox: null pointer");
  decode_bbox_kernel<<<cdiv(R, 256), 256, 0, as_stream(stream)>>>(bbox, prior, out, R, bbox_dim, mean_host[0], mean_host[1],
                                                                  mean_host[2], mean_host[3], std_host[0], std_host[1],
                                                                  std_host[2], std_host[3]);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" size_t mscnn_detections_workspace_bytes(int R) { return det_layout(R).total; }

static int detections_launch(const mscnn_detections_desc* desc, int cascade, float det_thr, const float* bbox_pred,
                             const float* cls_pred, const float* props, int R, double* dets_out, int* ids_out,
                             int* count_out_dev, void* workspace, size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(desc && count_out_dev && workspace, "detections: null pointer");
  MSCNN_REQUIRE(R >= 0, "detections: R < 0");
  MSCNN_REQUIRE(desc->ncls >= 2 && desc->cls_id >= 1 && desc->cls_id <= desc->ncls, "detections: cls_id %d of %d",
                desc->cls_id, desc->ncls);
  const DetLayout L = det_layout(R);
  if (workspace_bytes < L.total) {
    set_error("detections: workspace %zu < %zu", workspace_bytes, L.total);
    return MSCNN_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int* cnt = reinterpret_cast<int*>(ws + L.cnt);
  // (cnt needs no clearing: det_transform_sort_kernel writes cnt[DC_N] unconditionally before anything reads it)
  if (R == 0) {
    MSCNN_HIP_TRY(hipMemsetAsync(count_out_dev, 0, sizeof(int), st));
    return MSCNN_OK;
  }
  MSCNN_REQUIRE(bbox_pred && cls_pred && props && dets_out, "detections: null pointer");
  DetBox* sbox = reinterpret_cast<DetBox*>(ws + L.sbox);
  double* sprob = reinterpret_cast<double*>(ws + L.sprob);
  int* ssrc = reinterpret_cast<int*>(ws + L.ssrc);
  DetBox* tbox = reinterpret_cast<DetBox*>(ws + L.tbox);
  float* tprob = reinterpret_cast<float*>(ws + L.tprob);
  u64* mask = reinterpret_cast<u64*>(ws + L.mask);
  DetArgs a;
  a.bbox_pred = bbox_pred; a.cls_pred = cls_pred; a.props = props;
  a.R = R; a.ncls = desc->ncls; a.cls_id = desc->cls_id;
  for (int k = 0; k < 4; ++k) { a.mean[k] = desc->bbox_mean[k]; a.stdv[k] = desc->bbox_std[k]; }
  a.proposal_thr = cascade ? det_thr : desc->proposal_thr;
  a.cascade = cascade;
  // MATLAB: single op double -> single (the double operand is converted to single first)
  a.ratio_h = (float)desc->ratio_h; a.ratio_w = (float)desc->ratio_w;
  a.org_h = (float)desc->org_h; a.org_w = (float)desc->org_w;
  if (L.big) {
    u64* keys = reinterpret_cast<u64*>(ws + L.keys);
    int* kidx = reinterpret_cast<int*>(ws + L.kidx);
    MSCNN_HIP_TRY(hipMemsetAsync(cnt, 0, 256, st));
    MSCNN_HIP_TRY(hipMemsetAsync(keys, 0, (size_t)L.sortP * sizeof(u64), st));
    det_transform_big_kernel<<<cdiv(R, 256), 256, 0, st>>>(a, keys, tbox, tprob, cnt);
    MSCNN_POST_LAUNCH();
    MSCNN_HIP_TRY(big_sort_desc(keys, L.sortP, st));
    det_gather_big_kernel<<<cdiv(R, 256), 256, 0, st>>>(keys, tbox, tprob, sbox, sprob, ssrc, cnt);
    MSCNN_POST_LAUNCH();
    const double overlap = desc->nms_overlap;
    auto launch_mask = [&](const DetBox* tile, const int* tile_n) {
      det_mask_kernel<<<dim3(kTileWords, kTileWords), 256, 0, st>>>(tile, tile_n, overlap, mask, kTileWords);
    };
    MSCNN_HIP_TRY((big_nms_tiles<DetTr>(sbox, cnt + DC_N, 0, R, DetTr::Params{overlap}, mask, reinterpret_cast<u64*>(ws + L.rinit), kidx,
                                        reinterpret_cast<DetBox*>(ws + L.kbox), cnt + DC_BIG, launch_mask, st)));
    det_emit_big_kernel<<<cdiv(R, 256), 256, 0, st>>>(kidx, cnt + DC_BIG, sbox, sprob, ssrc, dets_out, ids_out, count_out_dev);
    MSCNN_POST_LAUNCH();
    return MSCNN_OK;
  }
  det_transform_sort_kernel<<<1, kSortThreads, 0, st>>>(a, sbox, sprob, ssrc, tbox, tprob, cnt);
  MSCNN_POST_LAUNCH();
  det_mask_kernel<<<dim3(L.wpr, L.wpr), 256, 0, st>>>(sbox, cnt, desc->nms_overlap, mask, L.wpr);
  MSCNN_POST_LAUNCH();
  det_scan_emit_kernel<<<1, 256, (size_t)2 * 64 * L.wpr * sizeof(u64), st>>>(mask, L.wpr, sbox, sprob, ssrc, dets_out, ids_out, cnt,
                                                                                count_out_dev);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

extern "C" int mscnn_detections_fwd(const mscnn_detections_desc* desc, const float* bbox_pred, const float* cls_pred,
                                    const float* props, int R, double* dets_out, int* ids_out, int* count_out_dev,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return detections_launch(desc, 0, 0.f, bbox_pred, cls_pred, props, R, dets_out, ids_out, count_out_dev, workspace,
                           workspace_bytes, stream);
}

extern "C" int mscnn_detections_cascade_fwd(const mscnn_detections_desc* desc, float det_thr, const float* boxes,
                                            const float* cls_prob, const float* props, int R, double* dets_out, int* ids_out,
                                            int* count_out_dev, void* workspace, size_t workspace_bytes, void* stream) {
  return detections_launch(desc, 1, det_thr, boxes, cls_prob, props, R, dets_out, ids_out, count_out_dev, workspace,
                           workspace_bytes, stream);
}
