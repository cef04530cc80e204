/*
 * sniper_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the integer / box-geometry part of the SNIPER hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (sniper_amd/) never does.
 *
 * Every function cites the reference lines (relative to the upstream checkout) whose
 * algorithm it restates.  The restatement is pinned against the reference's own native
 * code compiled into oracle/_ref/ (see oracle/build.py, tests/test_oracle_vs_ref.py) and
 * against committed golden vectors in tests/golden/.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (no fast-math: float32 rounding is part
 * of the contract).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline float fminf_(float a, float b) { return a < b ? a : b; }

/* ------------------------------------------------------------------------------------
 * Candidate chip enumeration.  lib/chips/cchips.cpp:62-108 (python twin
 * lib/chips/chip_generator.py:33-56).  Order: 3 corner chips, grid (x outer, y inner),
 * right-edge column, bottom-edge row.  Note the reference's quirks, preserved:
 *   corner 0: y2 = min(chipsize, height-1)  (not chipsize-1)
 *   corner 1: x2 = min(chipsize, width-1)
 *   edge column x1 = max(width-chipsize-1, 0), edge row y1 = max(height-chipsize-1, 0)
 * Returns the number of candidates; out may be NULL to query the count.
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_candidate_chips(int width, int height, int chipsize, int stride, float *out) {
  int c = 0;
#define PUT(a, b, cc, d)          \
  do {                            \
    if (out) {                    \
      out[4 * c + 0] = (float)(a); \
      out[4 * c + 1] = (float)(b); \
      out[4 * c + 2] = (float)(cc); \
      out[4 * c + 3] = (float)(d); \
    }                             \
    ++c;                          \
  } while (0)
  PUT(imax(width - chipsize, 0), 0, width - 1, imin(chipsize, height - 1));
  PUT(0, imax(height - chipsize, 0), imin(chipsize, width - 1), height - 1);
  PUT(imax(width - chipsize, 0), imax(height - chipsize, 0), width - 1, height - 1);
  for (int i = 0; i < width - chipsize; i += stride)
    for (int j = 0; j < height - chipsize; j += stride) PUT(i, j, i + chipsize - 1, j + chipsize - 1);
  for (int i = 0; i < height - chipsize; i += stride)
    PUT(imax(width - chipsize - 1, 0), i, width - 1, i + chipsize - 1);
  for (int i = 0; i < width - chipsize; i += stride)
    PUT(i, imax(height - chipsize - 1, 0), i + chipsize - 1, height - 1);
#undef PUT
  return c;
}

/* ------------------------------------------------------------------------------------
 * The shuffle of lib/chips/cchips.cpp:117 is std::random_shuffle(ids.begin(), ids.end())
 * which in libstdc++ (bits/stl_algo.h, the two-iterator overload) is
 *     for (i = first+1; i != last; ++i) iter_swap(i, first + rand() % ((i-first)+1));
 * driven by libc rand().  perm[k] = index of the candidate that lands in slot k.
 * With seed >= 0 we srand(seed) first (that is how the reference is made deterministic
 * in the golden-vector generator); seed < 0 continues the current libc stream.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_shuffle_perm(int n, long seed, int *perm) {
  if (seed >= 0) srand((unsigned)seed);
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int i = 1; i < n; ++i) {
    int j = rand() % (i + 1);
    if (i != j) {
      int t = perm[i];
      perm[i] = perm[j];
      perm[j] = t;
    }
  }
}

/* ------------------------------------------------------------------------------------
 * chips::cgenerate.  lib/chips/cchips.cpp:54-177.
 *   boxes  : (n,4) float32, already scaled to the pyramid level and clipped by the caller
 *            (lib/chips/chip_generator.py:24).
 *   perm   : candidate order after the shuffle (length = candidate count), or NULL for
 *            the identity order.
 *   out    : (max_out,4) float32 selected chips, in greedy order.
 *   out_ids: optional, index into the *shuffled* candidate list of each selected chip.
 * Match predicate (cchips.cpp:43-44,135): float32  iw*ih/area2 == 1  with iw>0, ih>0.
 * Greedy cover (cchips.cpp:146-167): repeatedly take the first chip with the strictly
 * largest number of still-unmatched boxes, remove its boxes from every chip.
 * Returns number of chips written (0 for empty boxes, cchips.cpp:56-57).
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_chips_generate(const float *boxes, int n, int width, int height, int chipsize, int stride,
                               const int *perm, float *out, int *out_ids, int max_out) {
  if (n <= 0) return 0;
  int C = orc_candidate_chips(width, height, chipsize, stride, NULL);
  float *cand = (float *)malloc(sizeof(float) * 4 * C);
  float *v = (float *)malloc(sizeof(float) * 4 * C);
  orc_candidate_chips(width, height, chipsize, stride, cand);
  for (int i = 0; i < C; ++i) {
    int s = perm ? perm[i] : i;
    memcpy(v + 4 * i, cand + 4 * s, 4 * sizeof(float));
  }
  int W = (n + 63) / 64;
  uint64_t *m = (uint64_t *)calloc((size_t)C * W, sizeof(uint64_t));
  for (int i = 0; i < C; ++i) {
    float x1 = v[4 * i], y1 = v[4 * i + 1], x2 = v[4 * i + 2], y2 = v[4 * i + 3];
    for (int j = 0; j < n; ++j) {
      float xx1 = boxes[4 * j], yy1 = boxes[4 * j + 1], xx2 = boxes[4 * j + 2], yy2 = boxes[4 * j + 3];
      float area2 = (xx2 - xx1 + 1) * (yy2 - yy1 + 1);
      float iw = fminf_(x2, xx2) - fmaxf_(x1, xx1) + 1;
      float ov = 0.f;
      if (iw > 0) {
        float ih = fminf_(y2, yy2) - fmaxf_(y1, yy1) + 1;
        if (ih > 0) ov = iw * ih / area2;
      }
      if (ov == 1.0f) m[(size_t)i * W + j / 64] |= 1ull << (j % 64);
    }
  }
  int nout = 0;
  for (;;) {
    int best = 0, mid = 0;
    for (int i = 0; i < C; ++i) {
      int cnt = 0;
      for (int w = 0; w < W; ++w) cnt += __builtin_popcountll(m[(size_t)i * W + w]);
      if (cnt > best) {
        best = cnt;
        mid = i;
      }
    }
    if (best == 0) break;
    if (nout < max_out) {
      memcpy(out + 4 * nout, v + 4 * mid, 4 * sizeof(float));
      if (out_ids) out_ids[nout] = mid;
    }
    ++nout;
    uint64_t *sel = (uint64_t *)malloc(sizeof(uint64_t) * W);
    memcpy(sel, m + (size_t)mid * W, sizeof(uint64_t) * W);
    for (int i = 0; i < C; ++i)
      for (int w = 0; w < W; ++w) m[(size_t)i * W + w] &= ~sel[w];
    free(sel);
  }
  free(m);
  free(v);
  free(cand);
  return nout;
}

/* ------------------------------------------------------------------------------------
 * bbox_overlaps_cython / ignore_overlaps_cython.  lib/bbox/bbox.pyx:17-57, 59-95.
 * float64, +1 pixel convention, zero unless iw>0 and ih>0.  out is (N,K) row-major.
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_bbox_overlaps_f64(const double *boxes, int N, const double *query, int K, double *out) {
  memset(out, 0, sizeof(double) * (size_t)N * K);
  for (int k = 0; k < K; ++k) {
    const double *q = query + 4 * k;
    double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    for (int n = 0; n < N; ++n) {
      const double *b = boxes + 4 * n;
      double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
      if (iw > 0) {
        double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
        if (ih > 0) {
          double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
          out[(size_t)n * K + k] = iw * ih / ua;
        }
      }
    }
  }
}

ORC_API void orc_ignore_overlaps_f64(const double *boxes, int N, const double *query, int K, double *out) {
  memset(out, 0, sizeof(double) * (size_t)N * K);
  for (int k = 0; k < K; ++k) {
    const double *q = query + 4 * k;
    double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    for (int n = 0; n < N; ++n) {
      const double *b = boxes + 4 * n;
      double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
      if (iw > 0) {
        double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
        if (ih > 0) out[(size_t)n * K + k] = iw * ih / box_area;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------
 * Bitmask hard NMS on score-sorted boxes.  lib/nms/nms_kernel.cu:24-32 (devIoU, float32),
 * :61-77 (suppress iff IoU > thresh), :118-140 (sequential keep scan).  Equivalent to the
 * numpy nms() of lib/nms/nms.py:90-127 (keeps ovr <= thresh) on sorted input.
 * boxes: (n, dim) float32, dim >= 4, sorted by descending score.  keep: int32[n].
 * max_keep <= 0 means unlimited (the proposal ops stop at rpn_post_nms_top_n).
 * ---------------------------------------------------------------------------------- */
static inline float dev_iou_f32(const float *a, const float *b) {
  float left = fmaxf_(a[0], b[0]), right = fminf_(a[2], b[2]);
  float top = fmaxf_(a[1], b[1]), bottom = fminf_(a[3], b[3]);
  float width = fmaxf_(right - left + 1, 0.f), height = fmaxf_(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

ORC_API int orc_nms_sorted_f32(const float *boxes, int n, int dim, float thresh, int max_keep, int *keep) {
  unsigned char *rem = (unsigned char *)calloc(n > 0 ? n : 1, 1);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (rem[i]) continue;
    keep[nk++] = i;
    if (max_keep > 0 && nk >= max_keep) break;
    for (int j = i + 1; j < n; ++j)
      if (!rem[j] && dev_iou_f32(boxes + (size_t)i * dim, boxes + (size_t)j * dim) > thresh) rem[j] = 1;
  }
  free(rem);
  return nk;
}

/* ------------------------------------------------------------------------------------
 * cpu_nms.  lib/nms/cpu_nms.pyx:112-163.  Differs from the bitmask/numpy version: float32
 * areas, suppresses when ovr >= thresh (ties suppressed), order supplied by the caller
 * (scores.argsort()[::-1] in the reference).
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_cpu_nms_f32(const float *dets, int n, const int *order, float thresh, int *keep) {
  unsigned char *sup = (unsigned char *)calloc(n > 0 ? n : 1, 1);
  float *areas = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i)
    areas[i] = (dets[5 * i + 2] - dets[5 * i + 0] + 1) * (dets[5 * i + 3] - dets[5 * i + 1] + 1);
  int nk = 0;
  for (int _i = 0; _i < n; ++_i) {
    int i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    float ix1 = dets[5 * i], iy1 = dets[5 * i + 1], ix2 = dets[5 * i + 2], iy2 = dets[5 * i + 3], iarea = areas[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      int j = order[_j];
      if (sup[j]) continue;
      float xx1 = fmaxf_(ix1, dets[5 * j]), yy1 = fmaxf_(iy1, dets[5 * j + 1]);
      float xx2 = fminf_(ix2, dets[5 * j + 2]), yy2 = fminf_(iy2, dets[5 * j + 3]);
      float w = fmaxf_(0.0f, xx2 - xx1 + 1), h = fmaxf_(0.0f, yy2 - yy1 + 1);
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr >= thresh) sup[j] = 1;
    }
  }
  free(sup);
  free(areas);
  return nk;
}

/* ------------------------------------------------------------------------------------
 * cpu_soft_nms.  lib/nms/cpu_nms.pyx:17-110.  In-place selection sort by score with
 * score decay; boxes is (N,5) float32 and is mutated exactly like the reference; the
 * return value is the surviving N (rows [0,N) are the result).
 * Temporaries are C float, with the promotions Cython emits: an integer literal next to a C float becomes the
 * DOUBLE literal 1.0 (the reference's own pre-generated lib/nms/cpu_nms.c:2946-3036 shows the same code Cython 3
 * produces), so `(x2 - x1 + 1)` is float-subtract, then double-add, and products / sums of such terms are double
 * until they are stored in a float variable.  For area, iw, ih and the linear weight the double detour rounds to the
 * same float as float arithmetic would (exact sums / products of 24-bit values); for
 *     ua = float((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih)
 * it does not: the three-term sum is formed in double and rounded once.  Pinned bit for bit against the
 * reference's compiled cpu_nms.pyx by tests/test_oracle_vs_ref.py and tests/golden/nms_v1.npz.
 * The gaussian weight: np.exp(-(ov*ov)/sigma) is evaluated in double on the float quotient, then narrowed to float
 * when stored in `weight` (cpu_nms.pyx:25,86).
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_soft_nms_f32(float *boxes, int N, float sigma, float Nt, float threshold, unsigned method) {
  for (int i = 0; i < N; ++i) {
    float maxscore = boxes[5 * i + 4];
    int maxpos = i;
    float tx1 = boxes[5 * i], ty1 = boxes[5 * i + 1], tx2 = boxes[5 * i + 2], ty2 = boxes[5 * i + 3],
          ts = boxes[5 * i + 4];
    for (int pos = i + 1; pos < N; ++pos)
      if (maxscore < boxes[5 * pos + 4]) {
        maxscore = boxes[5 * pos + 4];
        maxpos = pos;
      }
    for (int c = 0; c < 5; ++c) boxes[5 * i + c] = boxes[5 * maxpos + c];
    boxes[5 * maxpos] = tx1;
    boxes[5 * maxpos + 1] = ty1;
    boxes[5 * maxpos + 2] = tx2;
    boxes[5 * maxpos + 3] = ty2;
    boxes[5 * maxpos + 4] = ts;
    tx1 = boxes[5 * i];
    ty1 = boxes[5 * i + 1];
    tx2 = boxes[5 * i + 2];
    ty2 = boxes[5 * i + 3];
    int pos = i + 1;
    while (pos < N) {
      float x1 = boxes[5 * pos], y1 = boxes[5 * pos + 1], x2 = boxes[5 * pos + 2], y2 = boxes[5 * pos + 3];
      float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
      float iw = (float)((double)(fminf_(tx2, x2) - fmaxf_(tx1, x1)) + 1.0);
      if (iw > 0) {
        float ih = (float)((double)(fminf_(ty2, y2) - fmaxf_(ty1, y1)) + 1.0);
        if (ih > 0) {
          float ua = (float)(((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0) + (double)area - (double)(iw * ih));
          float ov = iw * ih / ua;
          float weight;
          if (method == 1)
            weight = ov > Nt ? (float)(1.0 - (double)ov) : 1;
          else if (method == 2)
            weight = (float)exp((double)(-(ov * ov) / sigma));
          else
            weight = ov > Nt ? 0 : 1;
          boxes[5 * pos + 4] = weight * boxes[5 * pos + 4];
          if (boxes[5 * pos + 4] < threshold) {
            for (int c = 0; c < 5; ++c) boxes[5 * pos + c] = boxes[5 * (N - 1) + c];
            N = N - 1;
            pos = pos - 1;
          }
        }
      }
      pos = pos + 1;
    }
  }
  return N;
}

/* ------------------------------------------------------------------------------------------------------------------
 * cv::resize(8UC3, INTER_LINEAR, fx, fy, dsize empty) -- scalar restatement in the loop structure of OpenCV's
 * modules/imgproc/src/resize.cpp (cv::resize -> cv::hal::resize -> resizeGeneric_ / resizeAreaFast_), see
 * oracle/cv_resize.py for the function-by-function citation.  What im_worker (lib/data_utils/data_workers.py:65,107)
 * calls on the decoded uint8 image.  dst must hold orc_cv_dsize() pixels.  Returns 0, or -1 for an empty dsize.
 * Pinned to the published algorithm (no OpenCV in this image to mint vectors).                                        */
static int orc_cv_round(double v) { return (int)lrint(v); }            /* cvRound: nearest, ties to even */
static int orc_cv_roundf(float v) { return (int)lrintf(v); }
static int orc_cv_floorf(float v) { int i = (int)v; return i - (i > v); }
static short orc_sat_short_f(float v) { int i = orc_cv_roundf(v); return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i); }
static unsigned char orc_sat_u8_f(float v) { int i = orc_cv_roundf(v); return (unsigned char)(i < 0 ? 0 : i > 255 ? 255 : i); }
static int orc_clip(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

void orc_cv_dsize(int h, int w, double fx, double fy, int *dh, int *dw) {
  *dw = orc_cv_round(w * fx);
  *dh = orc_cv_round(h * fy);
}

int orc_cv_resize_linear_u8c3(const unsigned char *src, int H, int W, double inv_scale_x, double inv_scale_y, unsigned char *dst) {
  const int cn = 3;
  int dw, dh;
  orc_cv_dsize(H, W, inv_scale_x, inv_scale_y, &dh, &dw);
  if (dw <= 0 || dh <= 0) return -1;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int iscale_x = orc_cv_round(scale_x), iscale_y = orc_cv_round(scale_y);
  const int is_area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {          /* INTER_LINEAR -> INTER_AREA (fast), 2 x 2 */
    const int dwidth1 = (W / 2) * cn, dwidth = dw * cn, swidth = W * cn;
    for (int dy = 0; dy < dh; ++dy) {
      unsigned char *D = dst + (size_t)dy * dwidth;
      const int sy0 = dy * 2;
      const int w = sy0 + 2 <= H ? dwidth1 : 0;
      int dx = 0;
      if (sy0 >= H) { for (; dx < dwidth; ++dx) D[dx] = 0; continue; }
      const unsigned char *S = src + (size_t)sy0 * swidth, *nextS = S + swidth;
      for (; dx < w; ++dx) {                                     /* ResizeAreaFastVec, fast_mode, cn == 3 */
        const int index = (dx / cn) * 2 * cn + dx % cn;
        D[dx] = (unsigned char)((S[index] + S[index + cn] + nextS[index] + nextS[index + cn] + 2) >> 2);
      }
      for (; dx < dwidth; ++dx) {
        int sum = 0, count = 0;
        const int sx0 = (dx / cn) * 2 * cn + dx % cn;
        if (sx0 >= swidth) D[dx] = 0;
        for (int sy = 0; sy < 2; ++sy) {
          if (sy0 + sy >= H) break;
          const unsigned char *R = src + (size_t)(sy0 + sy) * swidth + sx0;
          for (int sx = 0; sx < 2 * cn; sx += cn) {
            if (sx0 + sx >= swidth) break;
            sum += R[sx];
            ++count;
          }
        }
        D[dx] = orc_sat_u8_f((float)sum / count);
      }
    }
    return 0;
  }
  int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
  short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw), *ibeta = (short *)malloc(sizeof(short) * 2 * (size_t)dh);
  int *row0 = (int *)malloc(sizeof(int) * (size_t)dw * cn), *row1 = (int *)malloc(sizeof(int) * (size_t)dw * cn);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = orc_cv_floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0, sx = 0; }
    if (sx + 1 >= W) {
      xmax = xmax < dx ? xmax : dx;
      if (sx >= W - 1) { fx = 0, sx = W - 1; }
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = orc_sat_short_f((1.f - fx) * 2048);
    ialpha[dx * 2 + 1] = orc_sat_short_f(fx * 2048);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = orc_cv_floorf(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = orc_sat_short_f((1.f - fy) * 2048);
    ibeta[dy * 2 + 1] = orc_sat_short_f(fy * 2048);
  }
  for (int dy = 0; dy < dh; ++dy) {
    int *rows[2] = {row0, row1};
    for (int k = 0; k < 2; ++k) {                                /* HResizeLinear over the two source rows of dy */
      const unsigned char *S = src + (size_t)orc_clip(yofs[dy] + k, 0, H) * W * cn;
      int *Dr = rows[k];
      for (int dx = 0; dx < dw; ++dx)
        for (int c = 0; c < cn; ++c)
          Dr[dx * cn + c] = dx < xmax ? S[xofs[dx] * cn + c] * ialpha[dx * 2] + S[(xofs[dx] + 1) * cn + c] * ialpha[dx * 2 + 1]
                                      : S[xofs[dx] * cn + c] * 2048;
    }
    const short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    unsigned char *D = dst + (size_t)dy * dw * cn;
    for (int x = 0; x < dw * cn; ++x) D[x] = (unsigned char)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs); free(yofs); free(ialpha); free(ibeta); free(row0); free(row1);
  return 0;
}
