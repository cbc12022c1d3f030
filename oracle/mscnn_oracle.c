/*
 * oracle/mscnn_oracle.c -- CPU restatement of the MS-CNN inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mscnn_amd/ (the product) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * Every function restates one reference function in plain C (fp32, same
 * operation order, no FMA contraction: build with -ffp-contract=off) and cites
 * the reference file:line (paths relative to the reference checkout) it follows.
 *
 * Pinning status:
 *   - pooling / conv / relu / inner-product / deconv / softmax: pinned against the
 *     reference's own known-answer tests (tests/test_oracle_kat.py restates
 *     src/caffe/test/test_pooling_layer.cpp:57-117,487-521,
 *     test_convolution_layer.cpp:19-139 comparator + Sobel :498-589,
 *     test_deconvolution_layer.cpp:117-134) and against oracle/_ref (the
 *     reference's own .cpp files compiled against oracle/shim) when built.
 *   - BoxOutput / ROIPooling / DecodeBBox: the reference holds no tests or
 *     fixtures for them; they are pinned BIT-EXACTLY against
 *     oracle/_ref -- the reference's own .cpp files compiled by oracle/ref.mk --
 *     in tests/test_oracle_vs_ref.py, and against the outputs of that library
 *     committed as tests/golden/reference_layers.npz (tests/test_golden.py).
 *   - final detection stage (MATLAB, run_mscnn_detection.m:75-120 + bbNms.m):
 *     PARITY UNPINNED -- MATLAB is not available; restated from source only.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------ */
/* Convolution                                                          */
/* ------------------------------------------------------------------ */

/* Output size: src/caffe/layers/conv_layer.cpp:8-22 (dilation 1). */
ORC_API int orc_conv_out_dim(int in, int k, int pad, int stride) {
  return (in + 2 * pad - k) / stride + 1;
}

/*
 * Definitional cross-correlation, the comparator the reference's own tests use:
 * src/caffe/test/test_convolution_layer.cpp:19-139 (caffe_conv), restricted to 2-D,
 * dilation 1.  Weights [Cout][Cin/g][Kh][Kw] (base_conv_layer.cpp:135-140), no
 * kernel flip (im2col.cpp:30-52).  Summation order: (c, kh, kw) ascending, bias
 * added after the sum -- the same order im2col+GEMM(k ascending) produces.
 */
ORC_API int orc_conv2d_naive(const float* x, const float* w, const float* b, float* y,
                             int N, int Cin, int H, int W, int Cout, int Kh, int Kw,
                             int ph, int pw, int sh, int sw, int group) {
  const int Ho = orc_conv_out_dim(H, Kh, ph, sh), Wo = orc_conv_out_dim(W, Kw, pw, sw);
  if (Cin % group || Cout % group) return -1;
  const int cig = Cin / group, cog = Cout / group;
  for (int n = 0; n < N; ++n)
    for (int g = 0; g < group; ++g)
      for (int o = 0; o < cog; ++o) {
        const int oc = g * cog + o;
        for (int oy = 0; oy < Ho; ++oy)
          for (int ox = 0; ox < Wo; ++ox) {
            float acc = 0.f;
            for (int c = 0; c < cig; ++c)
              for (int ky = 0; ky < Kh; ++ky)
                for (int kx = 0; kx < Kw; ++kx) {
                  const int iy = oy * sh - ph + ky, ix = ox * sw - pw + kx;
                  if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                  acc += x[(((size_t)n * Cin + g * cig + c) * H + iy) * W + ix] *
                         w[(((size_t)oc * cig + c) * Kh + ky) * Kw + kx];
                }
            if (b) acc += b[oc];
            y[(((size_t)n * Cout + oc) * Ho + oy) * Wo + ox] = acc;
          }
      }
  return 0;
}

/* im2col: src/caffe/util/im2col.cpp:19-55 (dilation 1). */
static void im2col(const float* im, int C, int H, int W, int Kh, int Kw, int ph, int pw,
                   int sh, int sw, float* col) {
  const int Ho = orc_conv_out_dim(H, Kh, ph, sh), Wo = orc_conv_out_dim(W, Kw, pw, sw);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    const float* src = im + (size_t)c * H * W;
    float* dst = col + (size_t)c * Kh * Kw * Ho * Wo;
    for (int ky = 0; ky < Kh; ++ky)
      for (int kx = 0; kx < Kw; ++kx) {
        int iy = -ph + ky;
        for (int oy = 0; oy < Ho; ++oy, iy += sh) {
          if (iy < 0 || iy >= H) {
            for (int ox = 0; ox < Wo; ++ox) *dst++ = 0.f;
          } else {
            int ix = -pw + kx;
            for (int ox = 0; ox < Wo; ++ox, ix += sw)
              *dst++ = (ix >= 0 && ix < W) ? src[iy * W + ix] : 0.f;
          }
        }
      }
  }
}

/*
 * C[M,N] (+)= A[M,K] * B[K,N], row-major, k ascending per output element with one
 * rounding per multiply and per add -- the reference calls cblas_sgemm here
 * (src/caffe/util/math_functions.cpp:88-96), whose internal order is unspecified;
 * this fixed order equals the definitional loop above bit for bit.
 */
__attribute__((target_clones("avx2", "default")))
static void sgemm_nn(int M, int N, int K, const float* A, const float* B, float* C, int accumulate) {
  enum { JB = 2048, KB = 128 };
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int j0 = 0; j0 < N; j0 += JB)
    for (int i = 0; i < M; ++i) {
      const int jn = imin(JB, N - j0);
      float* c = C + (size_t)i * N + j0;
      if (!accumulate)
        for (int j = 0; j < jn; ++j) c[j] = 0.f;
      for (int k0 = 0; k0 < K; k0 += KB) {
        const int kn = imin(KB, K - k0);
        for (int k = 0; k < kn; ++k) {
          const float a = A[(size_t)i * K + k0 + k];
          const float* bp = B + (size_t)(k0 + k) * N + j0;
          for (int j = 0; j < jn; ++j) c[j] += a * bp[j];
        }
      }
    }
}

/*
 * im2col + GEMM convolution, the reference's CPU path:
 * src/caffe/layers/conv_layer.cpp:25-40 -> base_conv_layer.cpp:257-280 (forward_cpu_gemm,
 * forward_cpu_bias).  One image at a time (col buffer holds one image, :229-241).
 * `col` is caller-provided scratch of Cin*Kh*Kw*Ho*Wo floats (NULL -> malloc).
 */
ORC_API int orc_conv2d(const float* x, const float* w, const float* b, float* y,
                       int N, int Cin, int H, int W, int Cout, int Kh, int Kw,
                       int ph, int pw, int sh, int sw, int group, float* col) {
  const int Ho = orc_conv_out_dim(H, Kh, ph, sh), Wo = orc_conv_out_dim(W, Kw, pw, sw);
  if (Cin % group || Cout % group) return -1;
  const int cig = Cin / group, cog = Cout / group, kdim = cig * Kh * Kw;
  const size_t osp = (size_t)Ho * Wo;
  const int is1x1 = (Kh == 1 && Kw == 1 && ph == 0 && pw == 0 && sh == 1 && sw == 1);
  float* own = NULL;
  if (!is1x1 && !col) { own = col = (float*)malloc(sizeof(float) * (size_t)Cin * Kh * Kw * osp); if (!col) return -2; }
  for (int n = 0; n < N; ++n) {
    const float* xin = x + (size_t)n * Cin * H * W;
    const float* cb = xin;
    if (!is1x1) { im2col(xin, Cin, H, W, Kh, Kw, ph, pw, sh, sw, col); cb = col; }
    float* yo = y + (size_t)n * Cout * osp;
    for (int g = 0; g < group; ++g)
      sgemm_nn(cog, (int)osp, kdim, w + (size_t)g * cog * kdim, cb + (size_t)g * kdim * osp,
               yo + (size_t)g * cog * osp, 0);
    if (b)
      for (int o = 0; o < Cout; ++o) {
        float* yr = yo + (size_t)o * osp;
        const float bv = b[o];
        for (size_t i = 0; i < osp; ++i) yr[i] += bv * 1.f;
      }
  }
  free(own);
  return 0;
}

/* ------------------------------------------------------------------ */
/* ReLU: src/caffe/layers/relu_layer.cpp:9-19                           */
/* ------------------------------------------------------------------ */
ORC_API int orc_relu(const float* x, float* y, long n, float slope) {
  for (long i = 0; i < n; ++i) {
    const float v = x[i];
    y[i] = (v > 0.f ? v : 0.f) + slope * (v < 0.f ? v : 0.f);
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Pooling: src/caffe/layers/pooling_layer.cpp:78-123 (shape), :140-222  */
/* ------------------------------------------------------------------ */
ORC_API int orc_pool_out_dim(int in, int k, int pad, int stride) {
  int o = (int)ceilf((float)(in + 2 * pad - k) / stride) + 1;
  if (pad && (o - 1) * stride >= in + pad) --o;
  return o;
}

/* method: 0 = MAX (first max wins, init -FLT_MAX; mask optional), 1 = AVE */
ORC_API int orc_pool2d(const float* x, float* y, int* mask, int N, int C, int H, int W,
                       int kh, int kw, int ph, int pw, int sh, int sw, int method) {
  const int Ho = orc_pool_out_dim(H, kh, ph, sh), Wo = orc_pool_out_dim(W, kw, pw, sw);
  for (long nc = 0; nc < (long)N * C; ++nc) {
    const float* src = x + nc * H * W;
    float* dst = y + nc * Ho * Wo;
    int* m = mask ? mask + nc * Ho * Wo : NULL;
    for (int py = 0; py < Ho; ++py)
      for (int px = 0; px < Wo; ++px) {
        int hs = py * sh - ph, ws = px * sw - pw;
        if (method == 0) {
          const int he = imin(hs + kh, H), we = imin(ws + kw, W);
          hs = imax(hs, 0); ws = imax(ws, 0);
          float best = -FLT_MAX; int bi = -1;
          for (int h = hs; h < he; ++h)
            for (int w_ = ws; w_ < we; ++w_)
              if (src[h * W + w_] > best) { best = src[h * W + w_]; bi = h * W + w_; }
          dst[py * Wo + px] = best;
          if (m) m[py * Wo + px] = bi;
        } else {
          int he = imin(hs + kh, H + ph), we = imin(ws + kw, W + pw);
          const int pool_size = (he - hs) * (we - ws);
          hs = imax(hs, 0); ws = imax(ws, 0); he = imin(he, H); we = imin(we, W);
          float acc = 0.f;
          for (int h = hs; h < he; ++h)
            for (int w_ = ws; w_ < we; ++w_) acc += src[h * W + w_];
          dst[py * Wo + px] = acc / pool_size;
        }
      }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* InnerProduct: src/caffe/layers/inner_product_layer.cpp:83-97          */
/*   top[M,N] = bottom[M,K] * W[N,K]^T + 1 * bias                         */
/* ------------------------------------------------------------------ */
ORC_API int orc_inner_product(const float* x, const float* w, const float* b, float* y,
                              int M, int N, int K) {
  /* cblas_sgemm(NoTrans, Trans) in the reference; here W is transposed once and the k-ascending GEMM above is
   * used, which is bit-identical to the plain dot-product loop  acc += x[m][k] * w[n][k]. */
  float* wt = (float*)malloc(sizeof(float) * (size_t)N * K);
  if (!wt) return -2;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) wt[(size_t)k * N + n] = w[(size_t)n * K + k];
  sgemm_nn(M, N, K, x, wt, y, 0);
  free(wt);
  if (b)
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) y[(size_t)m * N + n] += 1.f * b[n];
  return 0;
}

/* ------------------------------------------------------------------ */
/* Deconvolution (transposed conv): src/caffe/layers/deconv_layer.cpp:8-40, */
/* base_conv_layer.cpp:283-298 (backward_cpu_gemm) + col2im (im2col.cpp).  */
/* Weights [Cin][Cout/g][Kh][Kw]; scatter-add in (c, ky, kx, iy, ix) order   */
/* which is col2im's accumulation order for K=1 GEMMs (depthwise) and the   */
/* definitional order otherwise.                                            */
/* ------------------------------------------------------------------ */
ORC_API int orc_deconv_out_dim(int in, int k, int pad, int stride) {
  return stride * (in - 1) + k - 2 * pad;
}

ORC_API int orc_deconv2d(const float* x, const float* w, const float* b, float* y,
                         int N, int Cin, int H, int W, int Cout, int Kh, int Kw,
                         int ph, int pw, int sh, int sw, int group) {
  const int Ho = orc_deconv_out_dim(H, Kh, ph, sh), Wo = orc_deconv_out_dim(W, Kw, pw, sw);
  if (Cin % group || Cout % group) return -1;
  const int cig = Cin / group, cog = Cout / group;
  memset(y, 0, sizeof(float) * (size_t)N * Cout * Ho * Wo);
  for (int n = 0; n < N; ++n)
    for (int g = 0; g < group; ++g)
      for (int o = 0; o < cog; ++o) {
        float* yo = y + (((size_t)n * Cout + g * cog + o) * Ho) * Wo;
        for (int ky = 0; ky < Kh; ++ky)
          for (int kx = 0; kx < Kw; ++kx)
            for (int iy = 0; iy < H; ++iy) {
              const int oy = iy * sh - ph + ky;
              if (oy < 0 || oy >= Ho) continue;
              for (int ix = 0; ix < W; ++ix) {
                const int ox = ix * sw - pw + kx;
                if (ox < 0 || ox >= Wo) continue;
                float v = 0.f;
                for (int c = 0; c < cig; ++c)
                  v += w[(((size_t)(g * cig + c) * cog + o) * Kh + ky) * Kw + kx] *
                       x[(((size_t)n * Cin + g * cig + c) * H + iy) * W + ix];
                yo[oy * Wo + ox] += v;
              }
            }
        if (b)
          for (int i = 0; i < Ho * Wo; ++i) yo[i] += b[g * cog + o];
      }
  return 0;
}

/* BilinearFiller: include/caffe/filler.hpp:244-262 */
ORC_API int orc_bilinear_filler(float* data, int count, int kh, int kw) {
  if (kh != kw) return -1;
  const int f = (int)ceil(kw / 2.);
  const float c = (2 * f - 1 - f % 2) / (2. * f);
  for (int i = 0; i < count; ++i) {
    const float xx = i % kw, yy = (i / kw) % kh;
    data[i] = (1 - fabs(xx / f - c)) * (1 - fabs(yy / f - c));
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Softmax over channels: src/caffe/layers/softmax_layer.cpp:27-60       */
/* x[outer][C][inner]                                                     */
/* ------------------------------------------------------------------ */
ORC_API int orc_softmax(const float* x, float* y, int outer, int C, int inner) {
  for (int i = 0; i < outer; ++i)
    for (int k = 0; k < inner; ++k) {
      const float* xp = x + (size_t)i * C * inner + k;
      float* yp = y + (size_t)i * C * inner + k;
      float mx = xp[0];
      for (int j = 0; j < C; ++j) mx = fmaxf(mx, xp[(size_t)j * inner]);
      float sum = 0.f;
      for (int j = 0; j < C; ++j) {
        const float e = expf(xp[(size_t)j * inner] - mx);
        yp[(size_t)j * inner] = e;
        sum += e;
      }
      for (int j = 0; j < C; ++j) yp[(size_t)j * inner] /= sum;
    }
  return 0;
}

/* ------------------------------------------------------------------ */
/* BoxIOU: src/caffe/util/math_functions.cpp:12-35                       */
/* mode: 0 = IOU, 1 = IOMU, 2 = IOFU                                      */
/* ------------------------------------------------------------------ */
ORC_API float orc_box_iou(float x1, float y1, float w1, float h1,
                          float x2, float y2, float w2, float h2, int mode) {
  if (w1 <= 0 || h1 <= 0 || w2 <= 0 || h2 <= 0) return 0.f;
  const float tlx = fmaxf(x1, x2), tly = fmaxf(y1, y2);
  const float brx = fminf(x1 + w1, x2 + w2), bry = fminf(y1 + h1, y2 + h2);
  float over;
  if (tlx >= brx || tly >= bry) over = 0.f; else over = (brx - tlx) * (bry - tly);
  float u;
  if (mode == 1) u = fminf(w1 * h1, w2 * h2);
  else if (mode == 2) u = w1 * h1;
  else u = w1 * h1 + w2 * h2 - over;
  return over / u;
}

/*
 * Greedy NMS over score-sorted boxes [x y w h]: box_output_layer.cpp:38-63 (nmsMax,
 * greedy=true).  keep[] receives 0/1; returns number kept.
 */
ORC_API int orc_nms_greedy(const float* boxes, int n, float thr, int mode, unsigned char* keep) {
  for (int i = 0; i < n; ++i) keep[i] = 1;
  for (int i = 0; i < n; ++i) {
    if (!keep[i]) continue;
    const float* a = boxes + 4 * (size_t)i;
    for (int j = i + 1; j < n; ++j) {
      if (!keep[j]) continue;
      const float* c = boxes + 4 * (size_t)j;
      const float o = orc_box_iou(a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3], mode);
      if (o > thr) keep[j] = 0;
    }
  }
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += keep[i];
  return cnt;
}

/* ------------------------------------------------------------------ */
/* BoxOutput: src/caffe/layers/box_output_layer.cpp:66-234               */
/* ------------------------------------------------------------------ */
typedef struct {
  float fg_thr, iou_thr;
  int nms_mode;               /* 0 IOU, 1 IOMU, 2 IOFU  (nms_type string) */
  float field_whr, field_xyr;
  int max_nms_num, max_post_nms_num;
  float min_size;
  int do_bbox_norm;
  float bbox_mean[4], bbox_std[4];
} orc_boxoutput_params;

typedef struct { float score; int idx; } score_idx;

/* std::greater<std::pair<Dtype,int>> : descending on score, then on index (:168) */
static int cmp_score_idx_desc(const void* a, const void* b) {
  const score_idx* p = (const score_idx*)a; const score_idx* q = (const score_idx*)b;
  if (p->score > q->score) return -1;
  if (p->score < q->score) return 1;
  if (p->idx > q->idx) return -1;
  if (p->idx < q->idx) return 1;
  return 0;
}

/*
 * heads[j]: (num, channels, hs[j], ws[j]) fp32 NCHW; channels = cls_num + 4.
 * rois_out: capacity cap rows x 5, props_out: cap rows x 6 (may be NULL).
 * cand_idx_out (optional, cap ints): for every emitted row, the candidate insertion
 *   index (per image) of the kept box -- the "NMS index selection" the parity bar
 *   compares bit-exactly.  anchor_id_out (optional): the same selection expressed as the
 *   global anchor id (sum of h*w of the preceding heads + id), which is what the device
 *   path reports (bookkeeping added by this restatement; it does not alter any result).
 * Returns R (number of rows incl. the dummy row when nothing survives) or <0.
 * *num_real receives the number of real boxes (0 when the dummy row was emitted).
 */
ORC_API int orc_boxoutput(const float* const* heads, const int* hs, const int* ws, int nheads,
                          int num, int channels, const float* field_w, const float* field_h,
                          const float* downsample, const orc_boxoutput_params* p,
                          float* rois_out, float* props_out, int* cand_idx_out, int* anchor_id_out,
                          int cap, int* num_real) {
  const int cls_num = channels - 4;
  const float field_whr = p->field_whr, field_xyr = p->field_xyr;
  const float min_whr = logf(1.f / field_whr), max_whr = logf(field_whr);
  const float min_xyr = -1.f / field_xyr, max_xyr = 1.f / field_xyr;
  int total = 0;
  long ncand_max = 0;
  for (int j = 0; j < nheads; ++j) ncand_max += (long)hs[j] * ws[j];
  float* boxes = (float*)malloc(sizeof(float) * 6 * (size_t)ncand_max);
  score_idx* si = (score_idx*)malloc(sizeof(score_idx) * (size_t)ncand_max);
  float* sorted = (float*)malloc(sizeof(float) * 4 * (size_t)ncand_max);
  unsigned char* keep = (unsigned char*)malloc((size_t)ncand_max);
  int* cand_aid = (int*)malloc(sizeof(int) * (size_t)ncand_max);
  if (!boxes || !si || !sorted || !keep || !cand_aid) return -2;

  for (int i = 0; i < num; ++i) {
    int bb_count = 0;
    int head_off = 0;
    for (int j = 0; j < nheads; ++j) {
      const float* bottom_data = heads[j];
      const int width = ws[j], height = hs[j];
      const int spatial_dim = width * height;
      const int bottom_dim = channels * spatial_dim;
      const int img_width = (int)(width * downsample[j]), img_height = (int)(height * downsample[j]);
      for (int id = 0; id < spatial_dim; ++id) {
        const int base_idx = i * bottom_dim + id;
        const int coord_idx = base_idx + cls_num * spatial_dim;
        const int h = id / width, w = id % width;
        float fg_score = -FLT_MAX;
        for (int k = 1; k < cls_num; ++k) fg_score = fmaxf(fg_score, bottom_data[base_idx + k * spatial_dim]);
        fg_score -= bottom_data[base_idx];
        if (fg_score >= p->fg_thr) {
          float bbx = bottom_data[coord_idx], bby = bottom_data[coord_idx + spatial_dim];
          float bbw = bottom_data[coord_idx + 2 * spatial_dim], bbh = bottom_data[coord_idx + 3 * spatial_dim];
          if (p->do_bbox_norm) {
            bbx *= p->bbox_std[0]; bby *= p->bbox_std[1]; bbw *= p->bbox_std[2]; bbh *= p->bbox_std[3];
            bbx += p->bbox_mean[0]; bby += p->bbox_mean[1]; bbw += p->bbox_mean[2]; bbh += p->bbox_mean[3];
          }
          bbx = fmaxf(min_xyr, bbx); bbx = fminf(max_xyr, bbx);
          bby = fmaxf(min_xyr, bby); bby = fminf(max_xyr, bby);
          bbx = bbx * field_w[j] + (w + 0.5f) * downsample[j];
          bby = bby * field_h[j] + (h + 0.5f) * downsample[j];
          bbw = fmaxf(min_whr, bbw); bbw = fminf(max_whr, bbw);
          bbh = fmaxf(min_whr, bbh); bbh = fminf(max_whr, bbh);
          bbw = field_w[j] * expf(bbw); bbh = field_h[j] * expf(bbh);
          bbx = bbx - bbw / 2.f; bby = bby - bbh / 2.f;
          bbx = fmaxf(bbx, 0.f); bby = fmaxf(bby, 0.f);
          bbw = fminf(bbw, img_width - bbx); bbh = fminf(bbh, img_height - bby);
          if (bbw >= p->min_size && bbh >= p->min_size) {
            float* bb = boxes + 6 * (size_t)bb_count;
            bb[0] = (float)i; bb[1] = bbx; bb[2] = bby; bb[3] = bbw; bb[4] = bbh; bb[5] = fg_score;
            si[bb_count].score = fg_score; si[bb_count].idx = bb_count;
            cand_aid[bb_count] = head_off + id;
            ++bb_count;
          }
        }
      }
      head_off += spatial_dim;
    }
    if (bb_count <= 0) continue;
    qsort(si, (size_t)bb_count, sizeof(score_idx), cmp_score_idx_desc);
    int n = bb_count;
    if (p->max_nms_num > 0 && bb_count > p->max_nms_num) n = p->max_nms_num;
    for (int k = 0; k < n; ++k) memcpy(sorted + 4 * (size_t)k, boxes + 6 * (size_t)si[k].idx + 1, 4 * sizeof(float));
    orc_nms_greedy(sorted, n, p->iou_thr, p->nms_mode, keep);
    int emitted = 0;
    for (int k = 0; k < n; ++k) {
      if (!keep[k]) continue;
      if (p->max_post_nms_num > 0 && emitted >= p->max_post_nms_num) break;
      if (total >= cap) { free(boxes); free(si); free(sorted); free(keep); free(cand_aid); return -3; }
      const float* bb = boxes + 6 * (size_t)si[k].idx;
      float* r = rois_out + 5 * (size_t)total;
      r[0] = bb[0]; r[1] = bb[1]; r[2] = bb[2]; r[3] = bb[1] + bb[3]; r[4] = bb[2] + bb[4];
      if (props_out) {
        float* q = props_out + 6 * (size_t)total;
        q[0] = bb[0]; q[1] = bb[1]; q[2] = bb[2]; q[3] = bb[1] + bb[3]; q[4] = bb[2] + bb[4]; q[5] = bb[5];
      }
      if (cand_idx_out) cand_idx_out[total] = si[k].idx;
      if (anchor_id_out) anchor_id_out[total] = cand_aid[si[k].idx];
      ++total; ++emitted;
    }
  }
  free(boxes); free(si); free(sorted); free(keep); free(cand_aid);
  if (num_real) *num_real = total;
  if (total <= 0) {               /* special case, :195-199, :214-218 */
    if (cap < 1) return -3;
    rois_out[0] = 0; rois_out[1] = 1; rois_out[2] = 1; rois_out[3] = 10; rois_out[4] = 10;
    if (props_out) for (int k = 0; k < 6; ++k) props_out[k] = 0.f;
    if (cand_idx_out) cand_idx_out[0] = -1;
    if (anchor_id_out) anchor_id_out[0] = -1;
    return 1;
  }
  return total;
}

/* ------------------------------------------------------------------ */
/* ROIPooling: src/caffe/layers/roi_pooling_layer.cpp:48-139             */
/* ------------------------------------------------------------------ */
ORC_API int orc_roipool(const float* feat, const float* rois, float* out, int* argmax,
                        int R, int batch, int C, int H, int W, int PH, int PW,
                        float spatial_scale, float pad_ratio) {
  for (int n = 0; n < R; ++n) {
    const float* roi = rois + 5 * (size_t)n;
    const int roi_batch_ind = (int)roi[0];
    if (roi_batch_ind < 0 || roi_batch_ind >= batch) return -1;
    const float pad_w = (roi[3] - roi[1] + 1) * pad_ratio;
    const float pad_h = (roi[4] - roi[2] + 1) * pad_ratio;
    /* `round` on a float expression: <cmath> float overload, half away from zero */
    const int roi_start_w = (int)roundf((roi[1] - pad_w) * spatial_scale);
    const int roi_start_h = (int)roundf((roi[2] - pad_h) * spatial_scale);
    const int roi_end_w = (int)roundf((roi[3] + pad_w) * spatial_scale);
    const int roi_end_h = (int)roundf((roi[4] + pad_h) * spatial_scale);
    const int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
    const int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
    const float bin_size_h = (float)roi_height / (float)PH;
    const float bin_size_w = (float)roi_width / (float)PW;
    for (int c = 0; c < C; ++c) {
      const float* batch_data = feat + ((size_t)roi_batch_ind * C + c) * H * W;
      float* top = out + ((size_t)n * C + c) * PH * PW;
      int* am = argmax ? argmax + ((size_t)n * C + c) * PH * PW : NULL;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          int hstart = (int)floorf((float)ph * bin_size_h);
          int wstart = (int)floorf((float)pw * bin_size_w);
          int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
          int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
          hstart = imin(imax(hstart + roi_start_h, 0), H);
          hend = imin(imax(hend + roi_start_h, 0), H);
          wstart = imin(imax(wstart + roi_start_w, 0), W);
          wend = imin(imax(wend + roi_start_w, 0), W);
          const int is_empty = (hend <= hstart) || (wend <= wstart);
          float best = is_empty ? 0.f : -FLT_MAX;
          int bi = -1;
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w)
              if (batch_data[h * W + w] > best) { best = batch_data[h * W + w]; bi = h * W + w; }
          top[ph * PW + pw] = best;
          if (am) am[ph * PW + pw] = bi;
        }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* DecodeBBox (TEST phase): src/caffe/layers/decode_bbox_layer.cpp:54-123 */
/*   + DecodeBBoxesWithPrior, src/caffe/util/math_functions.cpp:46-75     */
/* bbox (R,8) uses columns 4..7; prior (R,5); out (R,5)                     */
/* ------------------------------------------------------------------ */
ORC_API int orc_decode_bbox(const float* bbox, const float* prior, float* out, int R, int bbox_dim,
                            const float* mean, const float* stdv) {
  if (bbox_dim != 8) return -1;
  for (int i = 0; i < R; ++i) {
    const float xmin = prior[i * 5 + 1], ymin = prior[i * 5 + 2];
    const float xmax = prior[i * 5 + 3], ymax = prior[i * 5 + 4];
    const float pw = xmax - xmin + 1, ph = ymax - ymin + 1;
    /* `0.5*(float+float)`: double multiply, narrowed on assignment to Dtype */
    const float cx = (float)(0.5 * (double)(xmax + xmin)), cy = (float)(0.5 * (double)(ymax + ymin));
    const int c = 1;
    const float bx = bbox[i * bbox_dim + 4 * c] * stdv[0] + mean[0];
    const float by = bbox[i * bbox_dim + 4 * c + 1] * stdv[1] + mean[1];
    const float bw = bbox[i * bbox_dim + 4 * c + 2] * stdv[2] + mean[2];
    const float bh = bbox[i * bbox_dim + 4 * c + 3] * stdv[3] + mean[3];
    float tx = bx * pw + cx, ty = by * ph + cy;
    const float tw = pw * expf(bw), th = ph * expf(bh);
    tx -= (tw - 1) / 2; ty -= (th - 1) / 2;
    out[i * 5 + 0] = prior[i * 5];
    out[i * 5 + 1] = tx; out[i * 5 + 2] = ty;
    out[i * 5 + 3] = tx + tw - 1; out[i * 5 + 4] = ty + th - 1;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* Final detection stage (MATLAB): examples/kitti_car/run_mscnn_detection.m:75-120 */
/* + utils/bbNms.m:112-126 (nmsMax, greedy, union), PARITY UNPINNED.               */
/* Single-precision arithmetic up to `double([...])` (MATLAB: single op double ->   */
/* single), NMS in double after a STABLE descending sort (ties: lower index first). */
/* cls_id is 1-based as in the script.  dets_out: rows [x y w h prob], ids_out:     */
/* index (0-based) into the *input* rows.  Returns D.                               */
typedef struct { double s; int i; } ds_idx;
static int cmp_ds_desc_stable(const void* a, const void* b) {
  const ds_idx* p = (const ds_idx*)a; const ds_idx* q = (const ds_idx*)b;
  if (p->s > q->s) return -1;
  if (p->s < q->s) return 1;
  return (p->i > q->i) - (p->i < q->i);
}

/* bbNms.m:112-126 (nmsMax, greedy, 'union') on n rows bb[n][5] = [x y w h score] (double); src[n] = input row of each.
 * Writes the kept rows in score order, returns their number. */
static int bbnms_max_union(const double* bb, const int* src, int n, double nms_overlap, double* dets_out, int* ids_out) {
  ds_idx* ord = (ds_idx*)malloc(sizeof(ds_idx) * (size_t)(n > 0 ? n : 1));
  unsigned char* kp = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
  /* bbNms.m:76 keeps bbs(:,5) > thr with thr = -inf: all rows (NaN excluded). */
  int m = 0;
  for (int i = 0; i < n; ++i) if (bb[5 * i + 4] > -INFINITY) { ord[m].s = bb[5 * i + 4]; ord[m].i = i; ++m; }
  qsort(ord, (size_t)m, sizeof(ds_idx), cmp_ds_desc_stable);             /* bbNms.m:114 */
  for (int i = 0; i < m; ++i) kp[i] = 1;
  for (int a = 0; a < m; ++a) {
    if (!kp[a]) continue;
    const double* A = bb + 5 * (size_t)ord[a].i;
    const double as_a = A[2] * A[3], xe_a = A[0] + A[2], ye_a = A[1] + A[3];
    for (int c = a + 1; c < m; ++c) {
      if (!kp[c]) continue;
      const double* B = bb + 5 * (size_t)ord[c].i;
      const double iw = fmin(xe_a, B[0] + B[2]) - fmax(A[0], B[0]); if (iw <= 0) continue;
      const double ih = fmin(ye_a, B[1] + B[3]) - fmax(A[1], B[1]); if (ih <= 0) continue;
      double o = iw * ih; const double u = as_a + B[2] * B[3] - o;
      o = o / u; if (o > nms_overlap) kp[c] = 0;
    }
  }
  int D = 0;
  for (int a = 0; a < m; ++a) {
    if (!kp[a]) continue;
    memcpy(dets_out + 5 * (size_t)D, bb + 5 * (size_t)ord[a].i, 5 * sizeof(double));
    if (ids_out) ids_out[D] = src[ord[a].i];
    ++D;
  }
  free(ord); free(kp);
  return D;
}

/* ------------------------------------------------------------------ */
ORC_API int orc_detections(const float* bbox_pred, const float* cls_pred, const float* props,
                           int R, int ncls, int cls_id, const float* bbox_mean, const float* bbox_std,
                           float proposal_thr, double ratio_h, double ratio_w, double org_h, double org_w,
                           double nms_overlap, double* dets_out, int* ids_out) {
  double* bb = (double*)malloc(sizeof(double) * 5 * (size_t)(R > 0 ? R : 1));
  int* src = (int*)malloc(sizeof(int) * (size_t)(R > 0 ? R : 1));
  int n = 0;
  for (int r = 0; r < R; ++r) {
    const float* q = props + 6 * (size_t)r;
    const float px = q[1], py = q[2];
    const float pw = q[3] - q[1], ph = q[4] - q[2];
    const float sc = q[5];
    if (!(sc >= proposal_thr && pw != 0 && ph != 0)) continue;           /* :82 */
    const float* bp = bbox_pred + (size_t)r * 4 * ncls + 4 * (cls_id - 1);  /* :95 */
    float b0 = bp[0] * bbox_std[0], b1 = bp[1] * bbox_std[1], b2 = bp[2] * bbox_std[2], b3 = bp[3] * bbox_std[3];
    b0 += bbox_mean[0]; b1 += bbox_mean[1]; b2 += bbox_mean[2]; b3 += bbox_mean[3];
    const float* cp = cls_pred + (size_t)r * ncls;
    float se = 0.f;
    for (int k = 0; k < ncls; ++k) se += expf(cp[k]);                        /* :101-102, no max-sub */
    const float prob = expf(cp[cls_id - 1]) / se;
    const float ctr_x = px + 0.5f * pw, ctr_y = py + 0.5f * ph;
    float tx = b0 * pw + ctr_x, ty = b1 * ph + ctr_y;
    float tw = pw * expf(b2), th = ph * expf(b3);
    tx = tx - tw / 2.f; ty = ty - th / 2.f;
    tx = tx / (float)ratio_w; tw = tw / (float)ratio_w;
    ty = ty / (float)ratio_h; th = th / (float)ratio_h;
    tx = fmaxf(0.f, tx); ty = fmaxf(0.f, ty);
    tw = fminf(tw, (float)org_w - tx); th = fminf(th, (float)org_h - ty);
    double* d = bb + 5 * (size_t)n;
    d[0] = tx; d[1] = ty; d[2] = tw; d[3] = th; d[4] = prob;
    src[n] = r;
    ++n;
  }
  const int D = bbnms_max_union(bb, src, n, nms_overlap, dets_out, ids_out);
  free(bb); free(src);
  return D;
}

/* Final stage of the cascade drivers (examples/kitti_car/run_cascademscnn.m:84-112), PARITY UNPINNED like orc_detections:
 * boxes come decoded out of the net (DecodeBBox blob rows [img x1 y1 x2 y2]), the class score is the in-net probability
 * (Softmax / Eltwise blob).  Rescale to the original image, clip, width = x2 - x1 + 1, drop rows whose PROPOSAL has zero
 * width or height, optional det_thr, then bbNms (maxg, union).  Single-precision arithmetic until double([...]). */
ORC_API int orc_detections_cascade(const float* boxes, const float* cls_prob, const float* props, int R, int ncls, int cls_id,
                                   float det_thr, double ratio_h, double ratio_w, double org_h, double org_w,
                                   double nms_overlap, double* dets_out, int* ids_out) {
  double* bb = (double*)malloc(sizeof(double) * 5 * (size_t)(R > 0 ? R : 1));
  int* src = (int*)malloc(sizeof(int) * (size_t)(R > 0 ? R : 1));
  int n = 0;
  for (int r = 0; r < R; ++r) {
    const float* q = props + 5 * (size_t)r;
    const float pw = q[3] - q[1] + 1.f, ph = q[4] - q[2] + 1.f;             /* :104 */
    if (!(pw != 0 && ph != 0)) continue;                                    /* :107 */
    const float* t = boxes + 5 * (size_t)r;
    float x1 = t[1] / (float)ratio_w, x2 = t[3] / (float)ratio_w;           /* :88-89 */
    float y1 = t[2] / (float)ratio_h, y2 = t[4] / (float)ratio_h;
    x1 = fmaxf(0.f, x1); y1 = fmaxf(0.f, y1);                               /* :91 */
    x2 = fminf(x2, (float)org_w); y2 = fminf(y2, (float)org_h);             /* :92 */
    const float w = x2 - x1 + 1.f, h = y2 - y1 + 1.f;                       /* :93 */
    const float prob = cls_prob[(size_t)r * ncls + (cls_id - 1)];           /* :113 */
    if (det_thr > 0 && !(prob >= det_thr)) continue;                        /* :115-117 */
    double* d = bb + 5 * (size_t)n;
    d[0] = x1; d[1] = y1; d[2] = w; d[3] = h; d[4] = prob;
    src[n] = r;
    ++n;
  }
  const int D = bbnms_max_union(bb, src, n, nms_overlap, dets_out, ids_out);
  free(bb); free(src);
  return D;
}

/* ------------------------------------------------------------------ */
/* ROIAlign: src/caffe/layers/roi_align_layer.cpp:48-147 (CPU) /         */
/* roi_align_layer.cu:21-98.  out (R, C, PH+1, PW+1): bilinear samples on  */
/* the (PH+1)x(PW+1) grid of the padded ROI; the deploy nets follow it with */
/* a 2x2 stride-1 AVE Pooling.                                              */
/* ------------------------------------------------------------------ */
ORC_API int orc_roialign(const float* feat, const float* rois, float* out, int R, int batch, int C, int H, int W,
                         int PH, int PW, float spatial_scale, float pad_ratio) {
  const int GH = PH + 1, GW = PW + 1;
  for (int n = 0; n < R; ++n) {
    const float* roi = rois + 5 * (size_t)n;
    const int b = (int)roi[0];
    if (b < 0 || b >= batch) return -1;
    const float pad_w = (roi[3] - roi[1] + 1) * pad_ratio, pad_h = (roi[4] - roi[2] + 1) * pad_ratio;
    float roi_start_w = (roi[1] - pad_w) * spatial_scale, roi_start_h = (roi[2] - pad_h) * spatial_scale;
    float roi_end_w = (roi[3] + pad_w) * spatial_scale, roi_end_h = (roi[4] + pad_h) * spatial_scale;
    roi_start_w -= 0.5; roi_start_h -= 0.5; roi_end_w -= 0.5; roi_end_h -= 0.5;
    const float roi_height = roi_end_h - roi_start_h, roi_width = roi_end_w - roi_start_w;
    const float bin_size_h = roi_height / (float)PH, bin_size_w = roi_width / (float)PW;
    for (int c = 0; c < C; ++c) {
      const float* bd = feat + ((size_t)b * C + c) * H * W;
      float* top = out + ((size_t)n * C + c) * GH * GW;
      for (int ph = 0; ph <= PH; ++ph)
        for (int pw = 0; pw <= PW; ++pw) {
          float* o = top + ph * GW + pw;
          if (roi_height <= 0 || roi_width <= 0) { *o = 0.f; continue; }
          float hfloat = roi_start_h + (float)ph * bin_size_h, wfloat = roi_start_w + (float)pw * bin_size_w;
          if (hfloat < -0.5 || hfloat > (H - 0.5) || wfloat < -0.5 || wfloat > (W - 0.5)) { *o = 0.f; continue; }
          int hfloor = (int)floorf(hfloat), wfloor = (int)floorf(wfloat);
          int hceil = hfloor + 1, wceil = wfloor + 1;
          hfloat = fminf(fmaxf(hfloat, 0.f), (float)(H - 1)); wfloat = fminf(fmaxf(wfloat, 0.f), (float)(W - 1));
          hfloor = imin(imax(hfloor, 0), H - 1); wfloor = imin(imax(wfloor, 0), W - 1);
          hceil = imin(imax(hceil, 0), H - 1); wceil = imin(imax(wceil, 0), W - 1);
          const float lh = hfloat - hfloor, lw = wfloat - wfloor, hh = 1 - lh, hw = 1 - lw;
          const float w00 = hw * hh, w10 = lw * hh, w01 = hw * lh, w11 = lw * lh;
          const float v00 = bd[hfloor * W + wfloor], v10 = bd[hfloor * W + wceil], v01 = bd[hceil * W + wfloor], v11 = bd[hceil * W + wceil];
          *o = w00 * v00 + w10 * v10 + w01 * v01 + w11 * v11;
        }
    }
  }
  return 0;
}

/* Eltwise: src/caffe/layers/eltwise_layer.cpp:44-100.  op: 0 PROD, 1 SUM (coeffs), 2 MAX */
ORC_API int orc_eltwise(const float* const* xs, int nb, const float* coeffs, float* y, long count, int op) {
  if (nb < 2) return -1;
  for (long i = 0; i < count; ++i) {
    float v;
    if (op == 0) { v = xs[0][i] * xs[1][i]; for (int b = 2; b < nb; ++b) v = v * xs[b][i]; }
    else if (op == 1) { v = 0.f; for (int b = 0; b < nb; ++b) v = coeffs[b] * xs[b][i] + v; }   /* axpy: y = a*x + y */
    else {                                      /* MAX: a > b ? a : b, then strict '>' for bottoms 2.. (:72-92) */
      v = xs[0][i] > xs[1][i] ? xs[0][i] : xs[1][i];
      for (int b = 2; b < nb; ++b) if (xs[b][i] > v) v = xs[b][i];
    }
    y[i] = v;
  }
  return 0;
}

/* Concat along channels: src/caffe/layers/concat_layer.cpp:57-74 */
ORC_API int orc_concat_channels(const float* const* xs, const int* cs, int nb, float* y, int N, int inner) {
  int ctot = 0;
  for (int b = 0; b < nb; ++b) ctot += cs[b];
  int off = 0;
  for (int b = 0; b < nb; ++b) {
    for (int n = 0; n < N; ++n)
      memcpy(y + ((size_t)n * ctot + off) * inner, xs[b] + (size_t)n * cs[b] * inner,
             sizeof(float) * (size_t)cs[b] * inner);
    off += cs[b];
  }
  return 0;
}
