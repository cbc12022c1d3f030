// BoxOutput (proposal decode + top-K + greedy NMS) and the box utilities for gfx950.
//
// Reference: BoxOutputLayer<Dtype>::Forward_cpu, src/caffe/layers/box_output_layer.cpp:66-234 (CPU only
// in the reference: 7 D2H copies, std::sort, O(N^2) nmsMax with a heap-allocated vector per box).
// Here the whole layer stays on the device; only the row count R crosses PCIe.
//
// Pipeline per image (all on one stream, no host sync):
//   1. decode_filter_kernel   every anchor of every head in parallel: fg score, threshold, decode with
//                             the reference's exact fp32 operation order, min_size test; survivors
//                             append a 64-bit key (orderable(score) << 32 | anchor id) -- the append
//                             order is irrelevant because keys are unique and get sorted.
//   2. select_sort_kernel     one 1024-thread workgroup: 8-pass radix select of the K-th largest key
//                             (K = max_nms_num) over the L2-resident key list, compaction of the top K
//                             into LDS, bitonic sort (descending) in LDS, gather of the boxes.
//                             Descending (score, anchor id) == the reference's
//                             std::greater<pair<score, candidate idx>> (:168) because the candidate
//                             index is monotone in the anchor id.
//   3. nms_mask_kernel        64x64 blocks of the upper triangle of the K x K "IoU > thr" bit matrix.
//   4. nms_scan_emit_kernel   one wavefront: lane w owns word w of the removed-bitmap (K <= 4096 ->
//                             <= 64 words); per 64-box chunk a 64-step register-only greedy pass over
//                             the diagonal block, then OR of the kept rows (prefetched one chunk ahead);
//                             finally popcount-prefix compaction and emission of the output rows.
// No MFMA anywhere here: integer / compare work, wave-level scans and ballots.
//
// More than kMaxK = 4032 candidates into NMS (max_nms_num 0 -- the caffe.proto default, "no cap" -- or > 4032 while the heads
// have that many anchors): steps 2-4 are replaced by a global-memory bitonic sort of all keys and the tiled greedy NMS of
// nms_large.h (same predicate, same order: the same rows).  The deploy files of the reference set 2000-3000 and never get here.
#include "common.h"
#include "box_device.h"
#include "nms_large.h"
#include <cfloat>

#ifdef MSCNN_BO_TRACE
// debug build only (make -C mscnn_amd/csrc trace; tools/bo_trace.py): thread 0 of the one-workgroup kernels stamps s_memrealtime
// (100 MHz) at its phase boundaries into a buffer handed over by mscnn_debug_set_bo_trace
__device__ unsigned long long* g_bo_trace = nullptr;
#define BO_STAMP(slot) do { if (threadIdx.x == 0 && g_bo_trace) g_bo_trace[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int mscnn_debug_set_bo_trace(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bo_trace), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#else
#define BO_STAMP(slot) do { } while (0)
#endif

namespace {

using namespace mscnn_dev;

// BoxIOU, src/caffe/util/math_functions.cpp:12-35, same operation order (fp contraction is off).
__device__ __forceinline__ float box_iou(float x1, float y1, float w1, float h1, float x2, float y2, float w2, float h2,
                                         int mode) {
  if (w1 <= 0 || h1 <= 0 || w2 <= 0 || h2 <= 0) return 0.f;
  const float tlx = fmaxf(x1, x2), tly = fmaxf(y1, y2);
  const float brx = fminf(x1 + w1, x2 + w2), bry = fminf(y1 + h1, y2 + h2);
  float over;
  if ((tlx >= brx) || (tly >= bry)) over = 0.f;
  else over = (brx - tlx) * (bry - tly);
  float u;
  if (mode == 1) u = fminf(w1 * h1, w2 * h2);
  else if (mode == 2) u = w1 * h1;
  else u = w1 * h1 + w2 * h2 - over;
  return over / u;
}

struct DecodeArgs {
  const float* head[MSCNN_BOXOUT_MAX_HEADS];
  int head_h[MSCNN_BOXOUT_MAX_HEADS], head_w[MSCNN_BOXOUT_MAX_HEADS], head_off[MSCNN_BOXOUT_MAX_HEADS + 1];
  float field_w[MSCNN_BOXOUT_MAX_HEADS], field_h[MSCNN_BOXOUT_MAX_HEADS], ds[MSCNN_BOXOUT_MAX_HEADS];
  int num_heads, channels, image;
  float fg_thr, min_whr, max_whr, min_xyr, max_xyr, min_size;
  int do_norm;
  float mean[4], stdv[4];
};

// workspace counters
enum { CNT_CAND = 0, CNT_ROWS = 1, CNT_REAL = 2, CNT_K = 3, CNT_BIG = 4 /* + BIG_STATE_WORDS */, CNT_WORDS = 8 };

__global__ __launch_bounds__(256) void decode_filter_kernel(DecodeArgs a, u64* __restrict__ keys,
                                                            float4* __restrict__ box_by_anchor,
                                                            float* __restrict__ score_by_anchor, int* __restrict__ cnt) {
  const int total = a.head_off[a.num_heads];
  const int aid = blockIdx.x * 256 + threadIdx.x;
  if (aid >= total) return;
  int j = 0;
  while (aid >= a.head_off[j + 1]) ++j;
  const int id = aid - a.head_off[j];
  const int width = a.head_w[j], height = a.head_h[j];
  const int spatial = width * height;
  const int cls_num = a.channels - 4;
  const float* d = a.head[j] + (size_t)a.image * a.channels * spatial + id;
  const int h = id / width, w = id % width;
  // box_output_layer.cpp:120-127
  float fg = -FLT_MAX;
  for (int k = 1; k < cls_num; ++k) fg = fmaxf(fg, d[(size_t)k * spatial]);
  fg -= d[0];
  if (!(fg >= a.fg_thr)) return;
  const float* cd = d + (size_t)cls_num * spatial;
  float bbx = cd[0], bby = cd[spatial], bbw = cd[2 * (size_t)spatial], bbh = cd[3 * (size_t)spatial];
  if (a.do_norm) {   // :138-143
    bbx *= a.stdv[0]; bby *= a.stdv[1]; bbw *= a.stdv[2]; bbh *= a.stdv[3];
    bbx += a.mean[0]; bby += a.mean[1]; bbw += a.mean[2]; bbh += a.mean[3];
  }
  const float fw = a.field_w[j], fh = a.field_h[j], s = a.ds[j];
  const int img_width = (int)(width * s), img_height = (int)(height * s);     // :115
  bbx = fmaxf(a.min_xyr, bbx); bbx = fminf(a.max_xyr, bbx);                       // :145-146
  bby = fmaxf(a.min_xyr, bby); bby = fminf(a.max_xyr, bby);
  bbx = bbx * fw + (w + 0.5f) * s;
  bby = bby * fh + (h + 0.5f) * s;
  bbw = fmaxf(a.min_whr, bbw); bbw = fminf(a.max_whr, bbw);
  bbh = fmaxf(a.min_whr, bbh); bbh = fminf(a.max_whr, bbh);
  bbw = fw * expf_libm(bbw); bbh = fh * expf_libm(bbh);
  bbx = bbx - bbw / 2.f; bby = bby - bbh / 2.f;
  bbx = fmaxf(bbx, 0.f); bby = fmaxf(bby, 0.f);
  bbw = fminf(bbw, img_width - bbx); bbh = fminf(bbh, img_height - bby);
  if (bbw >= a.min_size && bbh >= a.min_size) {
    const int pos = atomicAdd(&cnt[CNT_CAND], 1);
    keys[pos] = ((u64)orderable(fg) << 32) | (unsigned)aid;
    box_by_anchor[aid] = make_float4(bbx, bby, bbw, bbh);
    score_by_anchor[aid] = fg;
  }
}

// One workgroup.  keys[0..n) unordered, unique.  Output: sorted_box[k], sorted_score[k], sorted_aid[k],
// cnt[CNT_K] = K.
__global__ __launch_bounds__(kSortThreads) void select_sort_kernel(const u64* __restrict__ keys,
                                                                   const float4* __restrict__ box_by_anchor,
                                                                   const float* __restrict__ score_by_anchor,
                                                                   float4* __restrict__ sorted_box,
                                                                   float* __restrict__ sorted_score,
                                                                   int* __restrict__ sorted_aid, int* __restrict__ cnt,
                                                                   int max_nms_num) {
  __shared__ u64 sk[kSortCap];
  __shared__ unsigned hist[256];
  __shared__ u64 s_prefix;
  __shared__ int s_need, s_fill, s_done;
  constexpr int kListCap = 1024;
  __shared__ u64 lst[kListCap];
  __shared__ int s_lfill;
  __shared__ unsigned hist12[4096];
  __shared__ int s_wtot[kSortThreads / 64];
  const int tid = threadIdx.x;
  BO_STAMP(0);
  const int n = cnt[CNT_CAND];
  int K = n;
  if (max_nms_num > 0 && K > max_nms_num) K = max_nms_num;
  if (K > kMaxK) K = kMaxK;   // host guarantees this cannot bind (checked against the anchor count)
  if (tid == 0) { cnt[CNT_K] = K; s_fill = 0; s_lfill = 0; }
  if (K == 0) return;

  // The candidate keys of a frame (27 k of the 45.6 k anchors on the 7s-576 bench frame; up to 32 k here) are read ONCE, kRegKeys per thread with all loads in flight together, and
  // every pass below works on the registers: the passes used to re-read the list from L2 with one dependent round trip per 1024
  // keys each (3-4 select passes + the compaction: ~50 round trips, most of the kernel's 70 us).  Longer lists take the loops.
  constexpr int kRegKeys = 32;
  const bool inreg = n <= kRegKeys * kSortThreads;
  u64 rk[kRegKeys];
  if (inreg) {
#pragma unroll
    for (int j = 0; j < kRegKeys; ++j) {
      const int i = tid + j * kSortThreads;
      rk[j] = i < n ? keys[i] : 0ull;
    }
  }
  BO_STAMP(1);
#ifdef MSCNN_BO_TRACE
  if (tid == 0 && g_bo_trace) { g_bo_trace[30] = (unsigned long long)n; g_bo_trace[31] = (unsigned long long)K; }
#endif
  u64 thresh = 0;             // keep keys >= thresh
  bool filled = false;        // the sort buffer already holds the K keys (sweep + list path)
  if (n > K) {
    // radix select, MSB first, 8 bits per pass: find the K-th largest key.  The bucket walk is a 64-lane suffix scan
    // (4 buckets per lane); the passes stop as soon as a bucket holds exactly the keys still needed (with the score in the
    // upper 32 bits that is normally after 3-4 passes: the anchor-id passes only separate equal scores).
    if (tid == 0) { s_prefix = 0; s_need = K; s_done = 0; }
    __syncthreads();
    // One digit of the select (LDS histogram of the keys that share the prefix found so far + the bucket walk).  A pass costs
    // 3 - 6 us whatever the histogram holds (barriers, a 32-key loop per thread, same-address LDS atomics on the first digits), so
    // only the first two are run; see below.  (Tried and measured slower: counting those two digits chip-wide in
    // decode_filter_kernel -- 27 k device-scope atomics, + 12 us there; and ballot-aggregating the LDS adds -- 50 us.)
    auto digit_pass = [&](int pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const u64 prefix = s_prefix;
      const u64 himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
      if (inreg) {
#pragma unroll
        for (int j = 0; j < kRegKeys; ++j) {
          const u64 k = rk[j];
          const bool act = tid + j * kSortThreads < n && (k & himask) == prefix;
          const unsigned bucket = (unsigned)(k >> shift) & 255u;
          if (pass < 2) {      // the skewed digits: a wave whose active lanes all share one bucket adds once (one ballot round)
            const u64 am = __ballot(act);
            if (am) {
              const int first = __ffsll((long long)am) - 1;
              const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bucket, first);
              const u64 same = __ballot(act && bucket == b0);
              if (same == am) { if ((tid & 63) == first) atomicAdd(&hist[b0], (unsigned)__popcll(am)); }
              else if (act) atomicAdd(&hist[bucket], 1u);
            }
          } else if (act) {
            atomicAdd(&hist[bucket], 1u);
          }
        }
      } else {
        for (int i = tid; i < n; i += kSortThreads) {
          const u64 k = keys[i];
          if ((k & himask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      if (tid < 64) {
        // lane l owns buckets 255-4l .. 252-4l (descending key order)
        unsigned c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = hist[255 - (4 * tid + j)];
        const int mine = (int)(c[0] + c[1] + c[2] + c[3]);
        int incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
          const int v = __shfl_up(incl, d, 64);
          if (tid >= d) incl += v;
        }
        const int need = s_need;
        const u64 ball = __ballot(incl >= need);
        const int first = __ffsll((long long)ball) - 1;      // ball != 0: the buckets hold at least `need` keys
        if (tid == first) {
          int rem = need - (incl - mine);
          int j = 0;
          for (; j < 3; ++j) {
            if ((int)c[j] >= rem) break;
            rem -= (int)c[j];
          }
          const int b = 255 - (4 * tid + j);
          s_need = rem;
          s_prefix = prefix | ((u64)b << shift);
          if ((int)c[j] == rem) s_done = 1;                  // the whole bucket is taken: lower bits of the threshold are 0
        }
      }
      __syncthreads();
    };
    // The same with 12-bit digits (4096 buckets, thread t owns buckets 4095 - 4 t .. 4092 - 4 t, workgroup-wide running total):
    // the first 8-bit digit of these keys -- sign + 7 exponent bits -- has 4 - 6 populated buckets, i.e. ~20 lanes of every wave
    // on one LDS address (20 us for 27 k keys, tools/bo_trace.py); 12 bits spread the same keys over ~40.
    auto digit_pass12 = [&](int p12) {
      const int shift = 52 - 12 * p12;
#pragma unroll
      for (int q = 0; q < 4; ++q) hist12[tid + q * kSortThreads] = 0;
      __syncthreads();
      const u64 prefix = s_prefix;
      const u64 himask = p12 == 0 ? 0ull : (~0ull << (shift + 12));
#pragma unroll
      for (int j = 0; j < kRegKeys; ++j) {
        const u64 k = rk[j];
        if (tid + j * kSortThreads < n && (k & himask) == prefix) atomicAdd(&hist12[(unsigned)(k >> shift) & 4095u], 1u);
      }
      __syncthreads();
      unsigned c[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = hist12[4095 - (4 * tid + j)];
      const int mine = (int)(c[0] + c[1] + c[2] + c[3]);
      int incl = mine;
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += v;
      }
      if ((tid & 63) == 63) s_wtot[tid >> 6] = incl;
      __syncthreads();
      for (int w = 0; w < (tid >> 6); ++w) incl += s_wtot[w];
      const int need = s_need;
      __syncthreads();                                       // (everybody has read s_need before the one writer below)
      if (incl >= need && incl - mine < need) {              // exactly one thread: the buckets hold at least `need` keys
        int rem = need - (incl - mine);
        int j = 0;
        for (; j < 3; ++j) {
          if ((int)c[j] >= rem) break;
          rem -= (int)c[j];
        }
        const int b = 4095 - (4 * tid + j);
        s_need = rem;
        s_prefix = prefix | ((u64)b << shift);
        if ((int)c[j] == rem) s_done = 1;
      }
      __syncthreads();
    };
    int pass = 0;
    if (inreg) {
      for (int p12 = 0; p12 < 2 && !s_done; ++p12) { digit_pass12(p12); BO_STAMP(11 + p12); }
      pass = 3;                                              // 24 bits are fixed: an 8-bit pass would continue at shift 32
    } else {
      for (; pass < 8 && !s_done; ++pass) digit_pass(pass);
    }
    if (inreg && !s_done) {
      // After two 12-bit digits: one sweep over the register keys -- keys whose top 24 bits lie above the threshold's go straight to the
      // sort buffer, the threshold bin's own keys (a few dozen) to a list, where the `need` largest are found by rank (keys are
      // unique) -- instead of more digit passes, each ~3 us of barriers and a 32-key loop per thread whatever the histogram holds.
      const unsigned bstar = (unsigned)(s_prefix >> 40);
#pragma unroll
      for (int j = 0; j < kRegKeys; ++j) {
        const u64 k = rk[j];
        const bool valid = tid + j * kSortThreads < n;
        const unsigned t16 = (unsigned)(k >> 40);
        if (valid && t16 > bstar) {                     // (one LDS atomic per wave and key slot, not per key)
          const u64 takers = __ballot(1);
          const int leader = __ffsll((long long)takers) - 1, lane = tid & 63;
          int base = 0;
          if (lane == leader) base = atomicAdd(&s_fill, __popcll(takers));
          base = __shfl(base, leader, 64);
          const int pos = base + __popcll(takers & ((1ull << lane) - 1ull));
          if (pos < kMaxK) sk[pos] = k;
        } else if (valid && t16 == bstar) {
          const int pos = atomicAdd(&s_lfill, 1);
          if (pos < kListCap) lst[pos] = k;
        }
      }
      __syncthreads();
      BO_STAMP(20);
      const int Ln = s_lfill, need = s_need;
      if (Ln <= kListCap) {
        for (int t = tid; t < Ln; t += kSortThreads) {
          const u64 my = lst[t];
          int rank = 0;
          for (int u = 0; u < Ln; ++u) rank += lst[u] > my ? 1 : 0;
          if (rank < need) {
            const int pos = atomicAdd(&s_fill, 1);
            if (pos < kMaxK) sk[pos] = my;
          }
        }
        BO_STAMP(21);
#ifdef MSCNN_BO_TRACE
        if (tid == 0 && g_bo_trace) g_bo_trace[29] = (unsigned long long)Ln;
#endif
        filled = true;
      } else {
        __syncthreads();
        if (tid == 0) s_fill = 0;      // (a bin with more than kListCap keys: the remaining digit passes, then the general compaction)
        for (; pass < 8 && !s_done; ++pass) digit_pass(pass);
      }
    }
    thresh = s_prefix;        // exactly K keys are >= thresh (keys are unique)
  }
  __syncthreads();
  BO_STAMP(2);
  int P = 1;
  while (P < K) P <<= 1;
  if (filled) {
  } else if (inreg) {
#pragma unroll
    for (int j = 0; j < kRegKeys; ++j) {
      const u64 k = rk[j];
      if (tid + j * kSortThreads < n && k >= thresh) {
        const int pos = atomicAdd(&s_fill, 1);
        if (pos < kMaxK) sk[pos] = k;
      }
    }
  } else {
    for (int i = tid; i < n; i += kSortThreads) {
      const u64 k = keys[i];
      if (k >= thresh) {
        const int pos = atomicAdd(&s_fill, 1);
        if (pos < kMaxK) sk[pos] = k;
      }
    }
  }
  // pad sorts last (real keys are never 0); the network runs on registers (element i in thread i % 1024, register i / 1024)
  const int P2 = P < 1024 ? 1024 : P;
  for (int i = K + tid; i < P2; i += kSortThreads) sk[i] = 0ull;
  __syncthreads();
  BO_STAMP(3);
  if (P2 == 1024) { u64 r[1] = {sk[tid]}; bitonic_desc_regs<1>(r, sk, tid); }
  else if (P2 == 2048) { u64 r[2] = {sk[tid], sk[tid + 1024]}; bitonic_desc_regs<2>(r, sk, tid); }
  else { u64 r[4] = {sk[tid], sk[tid + 1024], sk[tid + 2048], sk[tid + 3072]}; bitonic_desc_regs<4>(r, sk, tid); }
  BO_STAMP(4);
  for (int i = tid; i < K; i += kSortThreads) {
    const int aid = (int)(unsigned)(sk[i] & 0xffffffffull);
    sorted_box[i] = box_by_anchor[aid];
    sorted_score[i] = score_by_anchor[aid];
    sorted_aid[i] = aid;
  }
  BO_STAMP(5);
}

// mask[i][cb] bit t set <=> j = cb*64+t > i and IoU(box_i, box_j) > thr.  Upper-triangle blocks only.
// 256 threads per 64 x 64 block: wave v tests its 64 rows against columns 16v .. 16v+15 and the four 16-bit pieces meet in LDS
// (one wave per block did 64 dependent IoUs -- with their divisions -- per lane, 2 waves per CU: 20 us for K = 2000; round 3).
__global__ __launch_bounds__(256) void nms_mask_kernel(const float4* __restrict__ boxes, const int* __restrict__ cnt_k,
                                                       int n_fixed, float thr, int mode, u64* __restrict__ mask, int wpr) {
  const int n = cnt_k ? *cnt_k : n_fixed;
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  __shared__ float4 cbox[64];
  __shared__ unsigned part[4][64];
  const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j0 = cb * 64;
  if (wv == 0 && j0 + t < n) cbox[t] = boxes[j0 + t];
  __syncthreads();
  const int i = rb * 64 + t;
  unsigned bits = 0;
  if (i < n) {
    const float4 a = boxes[i];
    const int jn = min(64, n - j0);
#pragma unroll 4
    for (int q = wv * 16; q < wv * 16 + 16; ++q) {
      const int j = j0 + q;
      if (q >= jn || j <= i) continue;
      const float4 b = cbox[q];
      if (box_iou(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, mode) > thr) bits |= 1u << (q & 15);
    }
  }
  part[wv][t] = bits;
  __syncthreads();
  if (wv == 0 && i < n)
    mask[(size_t)i * wpr + cb] = (u64)part[0][t] | ((u64)part[1][t] << 16) | ((u64)part[2][t] << 32) | ((u64)part[3][t] << 48);
}

struct EmitArgs {
  const float4* sorted_box;
  const float* sorted_score;
  const int* sorted_aid;
  float* rois;
  float* props;
  int* aids;
  int cap, image, max_post;
  int* count_out;      // the call's last image on the one-workgroup path: the scan kernel also does boxoutput_finish_kernel's work (r5)
};

// rows == 0: the reference's dummy row (box_output_layer.cpp:195-199, :214-218); count_out = {rows of the tops, real rows}
__device__ __forceinline__ void boxoutput_finish(int rows, float* rois, float* props, int* aids, int* count_out) {
  if (rows <= 0) {
    rois[0] = 0; rois[1] = 1; rois[2] = 1; rois[3] = 10; rois[4] = 10;
    if (props) for (int k = 0; k < 6; ++k) props[k] = 0.f;
    if (aids) aids[0] = -1;
    count_out[0] = 1; count_out[1] = 0;
  } else {
    count_out[0] = rows; count_out[1] = rows;
  }
}

__global__ __launch_bounds__(256) void nms_scan_emit_kernel(const u64* __restrict__ mask, int wpr, int W, EmitArgs e,
                                                            int* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) u64 dyn_lds[];
  __shared__ u64 keepw[64];
  __shared__ int pre[64];
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  BO_STAMP(8);
  const int n = cnt[CNT_K];
  if (n <= 0) {
    if (e.count_out && tid == 0) boxoutput_finish(cnt[CNT_ROWS], e.rois, e.props, e.aids, e.count_out);
    return;
  }
  const u64 mykeep = greedy_scan(mask, n, wpr, W, dyn_lds);
  BO_STAMP(9);
  if (wave == 0) {
    keepw[lane] = mykeep;
    // exclusive prefix of popcounts over chunks
    const int mine = __popcll(mykeep);
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    pre[lane] = incl - mine;
    if (lane == 63) s_total = incl;
  }
  __syncthreads();
  int kept_total = s_total;
  if (e.max_post > 0 && kept_total > e.max_post) kept_total = e.max_post;     // :184-186
  const int row0 = cnt[CNT_ROWS];
  for (int k = tid; k < n; k += 256) {
    const int c = k >> 6, l = k & 63;
    const u64 kw = keepw[c];
    if (!((kw >> l) & 1ull)) continue;
    const int local = pre[c] + __popcll(kw & ((1ull << l) - 1ull));
    if (local >= kept_total) continue;
    const int row = row0 + local;
    if (row >= e.cap) continue;
    const float4 b = e.sorted_box[k];
    float* r = e.rois + 5 * (size_t)row;
    r[0] = (float)e.image; r[1] = b.x; r[2] = b.y; r[3] = b.x + b.z; r[4] = b.y + b.w;   // :201-210
    if (e.props) {
      float* q = e.props + 6 * (size_t)row;
      q[0] = (float)e.image; q[1] = b.x; q[2] = b.y; q[3] = b.x + b.z; q[4] = b.y + b.w; q[5] = e.sorted_score[k];
    }
    if (e.aids) e.aids[row] = e.sorted_aid[k];
  }
  __syncthreads();
  BO_STAMP(10);
  if (tid == 0) {
    cnt[CNT_ROWS] = row0 + kept_total; cnt[CNT_CAND] = 0;
    if (e.count_out) boxoutput_finish(row0 + kept_total, e.rois, e.props, e.aids, e.count_out);
  }
}

// ---- large path (nms_large.h) ----------------------------------------------------------------------------------------------
struct RoiTr {
  typedef float4 Box;
  struct Params { float thr; int mode; };
  static __device__ __forceinline__ bool over(const float4& a, const float4& b, const Params& p) {
    return box_iou(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, p.mode) > p.thr;      // == nms_mask_kernel's test (a earlier)
  }
};

// keys sorted descending (zero padding last): the first K = min(n, max_nms_num) rows, boxes gathered by anchor id
__global__ __launch_bounds__(256) void big_gather_kernel(const u64* __restrict__ keys, const float4* __restrict__ box_by_anchor,
                                                         const float* __restrict__ score_by_anchor,
                                                         float4* __restrict__ sorted_box, float* __restrict__ sorted_score,
                                                         int* __restrict__ sorted_aid, int* __restrict__ cnt, int max_nms_num) {
  int K = cnt[CNT_CAND];
  if (max_nms_num > 0 && K > max_nms_num) K = max_nms_num;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) cnt[CNT_K] = K;
  if (i >= K) return;
  const int aid = (int)(unsigned)(keys[i] & 0xffffffffull);
  sorted_box[i] = box_by_anchor[aid];
  sorted_score[i] = score_by_anchor[aid];
  sorted_aid[i] = aid;
}

// rows of the kept boxes (kept_idx ascending = descending score), one workgroup
__global__ __launch_bounds__(256) void big_emit_kernel(EmitArgs e, const int* __restrict__ kept_idx, int* __restrict__ cnt) {
  int kept_total = cnt[CNT_BIG + BIG_NKEPT];
  if (e.max_post > 0 && kept_total > e.max_post) kept_total = e.max_post;       // :184-186
  const int row0 = cnt[CNT_ROWS];
  for (int local = threadIdx.x; local < kept_total; local += 256) {
    const int row = row0 + local;
    if (row >= e.cap) continue;
    const int k = kept_idx[local];
    const float4 b = e.sorted_box[k];
    float* r = e.rois + 5 * (size_t)row;
    r[0] = (float)e.image; r[1] = b.x; r[2] = b.y; r[3] = b.x + b.z; r[4] = b.y + b.w;   // :201-210
    if (e.props) {
      float* q = e.props + 6 * (size_t)row;
      q[0] = (float)e.image; q[1] = b.x; q[2] = b.y; q[3] = b.x + b.z; q[4] = b.y + b.w; q[5] = e.sorted_score[k];
    }
    if (e.aids) e.aids[row] = e.sorted_aid[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) { cnt[CNT_ROWS] = row0 + kept_total; cnt[CNT_CAND] = 0; cnt[CNT_BIG + BIG_NKEPT] = 0; }
}

// stand-alone NMS, large n: keep bytes from the kept list
__global__ __launch_bounds__(256) void big_keep_bytes_kernel(const int* __restrict__ kept_idx, const int* __restrict__ state,
                                                             unsigned char* __restrict__ keep_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < state[BIG_NKEPT]) keep_out[kept_idx[i]] = 1;
}

__global__ void boxoutput_finish_kernel(int* __restrict__ cnt, float* rois, float* props, int* aids, int* count_out) {
  boxoutput_finish(cnt[CNT_ROWS], rois, props, aids, count_out);
}

// stand-alone NMS (parity tests): keep_out bytes
__global__ __launch_bounds__(256) void nms_scan_bytes_kernel(const u64* __restrict__ mask, int n, int wpr, int W,
                                                             unsigned char* __restrict__ keep_out) {
  extern __shared__ __attribute__((aligned(16))) u64 dyn_lds[];
  __shared__ u64 keepw[64];
  const int tid = threadIdx.x;
  const u64 mykeep = greedy_scan(mask, n, wpr, W, dyn_lds);
  if (tid < 64) keepw[tid] = mykeep;
  __syncthreads();
  for (int k = tid; k < n; k += 256) keep_out[k] = (unsigned char)((keepw[k >> 6] >> (k & 63)) & 1ull);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
  size_t cnt, keys, box, score, sbox, sscore, said, mask, rinit, kidx, kbox, total;
  int anchors, wpr, kcap, sortP;
  bool big;
};

// kcap = host-side bound on the boxes that enter NMS; above kMaxK the large path's buffers are sized for it
WsLayout layout_for(int anchors, int max_nms_num) {
  WsLayout L;
  L.anchors = anchors;
  L.wpr = kMaxK / 64;
  L.kcap = (max_nms_num > 0 && max_nms_num < anchors) ? max_nms_num : anchors;
  L.big = L.kcap > kMaxK;
  L.sortP = L.big ? big_sort_pow2(anchors) : 0;
  const size_t rows = L.big ? (size_t)L.kcap : (size_t)kMaxK;
  size_t o = 0;
  L.cnt = o; o += align_up(CNT_WORDS * sizeof(int), 256);
  L.keys = o; o += align_up((L.big ? (size_t)L.sortP : (size_t)anchors) * sizeof(u64), 256);
  L.box = o; o += align_up((size_t)anchors * sizeof(float4), 256);
  L.score = o; o += align_up((size_t)anchors * sizeof(float), 256);
  L.sbox = o; o += align_up(rows * sizeof(float4), 256);
  L.sscore = o; o += align_up(rows * sizeof(float), 256);
  L.said = o; o += align_up(rows * sizeof(int), 256);
  L.mask = o; o += align_up((size_t)kMaxK * L.wpr * sizeof(u64), 256);
  L.rinit = o; o += L.big ? align_up(64 * sizeof(u64), 256) : 0;
  L.kidx = o; o += L.big ? align_up(rows * sizeof(int), 256) : 0;
  L.kbox = o; o += L.big ? align_up(rows * sizeof(float4), 256) : 0;
  L.total = o;
  return L;
}

int total_anchors(const mscnn_boxoutput_desc* d) {
  long t = 0;
  for (int j = 0; j < d->num_heads; ++j) t += (long)d->head_h[j] * d->head_w[j];
  return (int)t;
}

}  // namespace

using namespace mscnn;

static int check_desc(const mscnn_boxoutput_desc* d) {
  MSCNN_REQUIRE(d, "boxoutput: null desc");
  MSCNN_REQUIRE(d->num_heads >= 1 && d->num_heads <= MSCNN_BOXOUT_MAX_HEADS, "boxoutput: %d heads (max %d)", d->num_heads,
                MSCNN_BOXOUT_MAX_HEADS);
  MSCNN_REQUIRE(d->num >= 1 && d->channels >= 6, "boxoutput: need num >= 1 and channels >= 6 (2 classes + 4)");
  MSCNN_REQUIRE(d->nms_mode >= 0 && d->nms_mode <= 2, "boxoutput: nms_mode");
  MSCNN_REQUIRE(d->field_whr > 0 && d->field_xyr > 0, "boxoutput: field_whr / field_xyr must be > 0");
  const int anchors = total_anchors(d);
  MSCNN_REQUIRE(anchors > 0, "boxoutput: no anchors");
  MSCNN_REQUIRE(d->max_nms_num >= 0 && d->max_post_nms_num >= 0, "boxoutput: negative max_nms_num / max_post_nms_num");
  MSCNN_REQUIRE(anchors <= (1 << 26), "boxoutput: %d anchors", anchors);
  return MSCNN_OK;
}

extern "C" size_t mscnn_boxoutput_workspace_bytes(const mscnn_boxoutput_desc* desc) {
  if (check_desc(desc) != MSCNN_OK) return 0;
  return layout_for(total_anchors(desc), desc->max_nms_num).total;
}

extern "C" int mscnn_boxoutput_max_rows(const mscnn_boxoutput_desc* desc) {
  if (check_desc(desc) != MSCNN_OK) return 0;
  const int anchors = total_anchors(desc);
  int per_img = (desc->max_nms_num > 0 && desc->max_nms_num < anchors) ? desc->max_nms_num : anchors;
  if (desc->max_post_nms_num > 0 && desc->max_post_nms_num < per_img) per_img = desc->max_post_nms_num;
  const long rows = (long)per_img * desc->num;
  return (int)(rows < 1 ? 1 : rows);
}

extern "C" int mscnn_boxoutput_fwd_f32(const mscnn_boxoutput_desc* d, const float* const* heads_host, float* rois_out,
                                       float* props_out, int* anchor_ids_out, int cap, int* count_out_dev, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != MSCNN_OK) return rc;
  MSCNN_REQUIRE(heads_host && rois_out && count_out_dev && workspace, "boxoutput: null pointer");
  MSCNN_REQUIRE(cap >= 1, "boxoutput: cap must be >= 1");
  const int anchors = total_anchors(d);
  const WsLayout L = layout_for(anchors, d->max_nms_num);
  if (workspace_bytes < L.total) {
    set_error("boxoutput: workspace %zu < %zu", workspace_bytes, L.total);
    return MSCNN_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  char* ws = static_cast<char*>(workspace);
  int* cnt = reinterpret_cast<int*>(ws + L.cnt);
  u64* keys = reinterpret_cast<u64*>(ws + L.keys);
  float4* box = reinterpret_cast<float4*>(ws + L.box);
  float* score = reinterpret_cast<float*>(ws + L.score);
  float4* sbox = reinterpret_cast<float4*>(ws + L.sbox);
  float* sscore = reinterpret_cast<float*>(ws + L.sscore);
  int* said = reinterpret_cast<int*>(ws + L.said);
  u64* mask = reinterpret_cast<u64*>(ws + L.mask);

  DecodeArgs a;
  a.num_heads = d->num_heads;
  a.channels = d->channels;
  a.head_off[0] = 0;
  for (int j = 0; j < d->num_heads; ++j) {
    MSCNN_REQUIRE(heads_host[j], "boxoutput: head %d is null", j);
    a.head[j] = heads_host[j];
    a.head_h[j] = d->head_h[j]; a.head_w[j] = d->head_w[j];
    a.head_off[j + 1] = a.head_off[j] + d->head_h[j] * d->head_w[j];
    a.field_w[j] = d->field_w[j]; a.field_h[j] = d->field_h[j]; a.ds[j] = d->downsample_rate[j];
  }
  // box_output_layer.cpp:76-77 (host libm, like the reference layer)
  a.min_whr = logf(1.f / d->field_whr); a.max_whr = logf(d->field_whr);
  a.min_xyr = -1.f / d->field_xyr; a.max_xyr = 1.f / d->field_xyr;
  a.fg_thr = d->fg_thr; a.min_size = d->min_size;
  a.do_norm = d->do_bbox_norm;
  for (int k = 0; k < 4; ++k) { a.mean[k] = d->bbox_mean[k]; a.stdv[k] = d->bbox_std[k]; }

  MSCNN_HIP_TRY(hipMemsetAsync(cnt, 0, CNT_WORDS * sizeof(int), st));
  const int kcap = L.kcap;
  const int kblocks = cdiv(kcap < kMaxK ? kcap : kMaxK, 64);
  for (int img = 0; img < d->num; ++img) {
    a.image = img;
    if (L.big) MSCNN_HIP_TRY(hipMemsetAsync(keys, 0, (size_t)L.sortP * sizeof(u64), st));     // zero keys = padding, sorts last
    decode_filter_kernel<<<cdiv(anchors, 256), 256, 0, st>>>(a, keys, box, score, cnt);
    MSCNN_POST_LAUNCH();
    if (L.big) {
      // every candidate sorted in HBM, the first K = min(n, max_nms_num) gathered, tiled greedy NMS, rows from the kept list
      MSCNN_HIP_TRY(big_sort_desc(keys, L.sortP, st));
      big_gather_kernel<<<cdiv(kcap, 256), 256, 0, st>>>(keys, box, score, sbox, sscore, said, cnt, d->max_nms_num);
      MSCNN_POST_LAUNCH();
      const float thr = d->iou_thr;
      const int mode = d->nms_mode;
      auto launch_mask = [&](const float4* tile, const int* tile_n) {
        nms_mask_kernel<<<dim3(kTileWords, kTileWords), 256, 0, st>>>(tile, tile_n, 0, thr, mode, mask, kTileWords);
      };
      MSCNN_HIP_TRY((big_nms_tiles<RoiTr>(sbox, cnt + CNT_K, 0, kcap, RoiTr::Params{thr, mode}, mask,
                                          reinterpret_cast<u64*>(ws + L.rinit), reinterpret_cast<int*>(ws + L.kidx),
                                          reinterpret_cast<float4*>(ws + L.kbox), cnt + CNT_BIG, launch_mask, st)));
      EmitArgs eb{sbox, sscore, said, rois_out, props_out, anchor_ids_out, cap, img, d->max_post_nms_num, nullptr};
      big_emit_kernel<<<1, 256, 0, st>>>(eb, reinterpret_cast<const int*>(ws + L.kidx), cnt);
      MSCNN_POST_LAUNCH();
      continue;
    }
    select_sort_kernel<<<1, kSortThreads, 0, st>>>(keys, box, score, sbox, sscore, said, cnt, d->max_nms_num);
    MSCNN_POST_LAUNCH();
    nms_mask_kernel<<<dim3(kblocks, kblocks), 256, 0, st>>>(sbox, cnt + CNT_K, 0, d->iou_thr, d->nms_mode, mask, L.wpr);
    MSCNN_POST_LAUNCH();
    EmitArgs e{sbox, sscore, said, rois_out, props_out, anchor_ids_out, cap, img, d->max_post_nms_num,
               img == d->num - 1 ? count_out_dev : nullptr};
    nms_scan_emit_kernel<<<1, 256, (size_t)2 * 64 * kblocks * sizeof(u64), st>>>(mask, L.wpr, kblocks, e, cnt);
    MSCNN_POST_LAUNCH();
  }
  if (L.big || d->num <= 0) {          // (the one-workgroup path finished in its last scan kernel)
    boxoutput_finish_kernel<<<1, 1, 0, st>>>(cnt, rois_out, props_out, anchor_ids_out, count_out_dev);
    MSCNN_POST_LAUNCH();
  }
  return MSCNN_OK;
}

namespace {
struct NmsBigLayout { size_t mask, rinit, state, kidx, kbox, total; };
NmsBigLayout nms_big_layout(int n) {
  NmsBigLayout L;
  size_t o = 0;
  L.mask = o; o += align_up((size_t)kMaxK * kTileWords * sizeof(u64), 256);
  L.rinit = o; o += align_up(64 * sizeof(u64), 256);
  L.state = o; o += 256;
  L.kidx = o; o += align_up((size_t)n * sizeof(int), 256);
  L.kbox = o; o += align_up((size_t)n * sizeof(float4), 256);
  L.total = o;
  return L;
}
}  // namespace

extern "C" size_t mscnn_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  if (n > kMaxK) return nms_big_layout(n).total;
  const size_t wpr = (size_t)(n + 63) / 64;
  return align_up((size_t)n * wpr * sizeof(u64), 256);
}

extern "C" int mscnn_nms_greedy_f32(const float* boxes_xywh, int n, float iou_thr, int nms_mode, unsigned char* keep_out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  MSCNN_REQUIRE(n >= 0 && nms_mode >= 0 && nms_mode <= 2, "nms: bad argument");
  if (n == 0) return MSCNN_OK;
  MSCNN_REQUIRE(boxes_xywh && keep_out && workspace, "nms: null pointer");
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(boxes_xywh) % 16 == 0, "nms: boxes must be 16-byte aligned");
  if (workspace_bytes < mscnn_nms_workspace_bytes(n)) {
    set_error("nms: workspace too small");
    return MSCNN_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  if (n > kMaxK) {     // tiled greedy NMS (nms_large.h): the boxes are taken in the order given
    const NmsBigLayout B = nms_big_layout(n);
    char* ws = static_cast<char*>(workspace);
    u64* bmask = reinterpret_cast<u64*>(ws + B.mask);
    int* state = reinterpret_cast<int*>(ws + B.state);
    int* kidx = reinterpret_cast<int*>(ws + B.kidx);
    MSCNN_HIP_TRY(hipMemsetAsync(state, 0, BIG_STATE_WORDS * sizeof(int), st));
    MSCNN_HIP_TRY(hipMemsetAsync(keep_out, 0, (size_t)n, st));
    auto launch_mask = [&](const float4* tile, const int* tile_n) {
      nms_mask_kernel<<<dim3(kTileWords, kTileWords), 256, 0, st>>>(tile, tile_n, 0, iou_thr, nms_mode, bmask, kTileWords);
    };
    MSCNN_HIP_TRY((big_nms_tiles<RoiTr>(reinterpret_cast<const float4*>(boxes_xywh), nullptr, n, n, RoiTr::Params{iou_thr, nms_mode},
                                        bmask, reinterpret_cast<u64*>(ws + B.rinit), kidx,
                                        reinterpret_cast<float4*>(ws + B.kbox), state, launch_mask, st)));
    big_keep_bytes_kernel<<<cdiv(n, 256), 256, 0, st>>>(kidx, state, keep_out);
    MSCNN_POST_LAUNCH();
    return MSCNN_OK;
  }
  const int wpr = (n + 63) / 64;
  u64* mask = static_cast<u64*>(workspace);
  nms_mask_kernel<<<dim3(wpr, wpr), 256, 0, st>>>(reinterpret_cast<const float4*>(boxes_xywh), nullptr, n, iou_thr, nms_mode,
                                                  mask, wpr);
  MSCNN_POST_LAUNCH();
  nms_scan_bytes_kernel<<<1, 256, (size_t)2 * 64 * wpr * sizeof(u64), st>>>(mask, n, wpr, wpr, keep_out);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}
