// infer.hip -- the two host loops either side of the test-time graph, on the GPU.
//
//  * im_prepare: im_worker.worker / worker_autofocus (lib/data_utils/data_workers.py:49-121) after the JPEG decode:
//    optional horizontal flip, integer crop, cv2.resize(fx = fy = scale, INTER_LINEAR) on uint8, zero padding to
//    (3, Hm, Wm) float32 with channel j = BGR[2 - j] - PIXEL_MEANS[2 - j].  The resize is OpenCV's published 8-bit
//    algorithm in its own integer arithmetic (11-bit fixed-point coefficients, integer horizontal and vertical passes,
//    the 2 x 2 INTER_AREA substitution), bit-exact against oracle/cv_resize.py and its scalar C twin: pinned to the
//    published algorithm (OpenCV itself is not in this image).
//  * bbox_decode: bbox_pred (= nonlinear_pred, lib/bbox/bbox_transform.py:93-130) + clip_boxes (:35-50) + division by
//    the image scale, as lib/inference.py:127-131 applies them per chip; float64 like the numpy original
//    (boxes.astype(np.float)), compiled with -ffp-contract=off, exp() in double.
#include "common.h"
#include <string.h>

// cv::resize's 8-bit INTER_LINEAR in OpenCV's own integer arithmetic (modules/imgproc/src/resize.cpp; restated with the
// function-by-function citation in oracle/cv_resize.py): coefficient of a destination column / row from
// f = (float)((d + 0.5) * scale - 0.5) in double-then-float as written there, 11-bit fixed-point coefficients
// (saturate_cast<short>(c * 2048), ties to even), integer horizontal pass, vertical pass
// (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  An exact 2 x 2 decimation is OpenCV's INTER_AREA fast path
// instead (AREA = true): whole blocks (a + b + c + d + 2) >> 2, the ragged last column / row of an odd source size the float mean
// of the pixels that exist, rounded to even.  One thread per destination pixel, three channels; this file is compiled with
// -ffp-contract=off (no FMA where OpenCV's scalar code has none).
__device__ __forceinline__ int cv_sat_short(float v) {
  const int i = (int)rintf(v);                          // cvRound: nearest, ties to even
  return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}

template <bool AREA>
__global__ __launch_bounds__(256) void im_prepare_kernel(const unsigned char *__restrict__ src, int SH, int SW, int x1, int y1,
                                                         int cw, int ch, double scale, int flip, double m0, double m1, double m2,
                                                         float *__restrict__ out, int Hm, int Wm, int oh, int ow) {
  const long total = (long)Hm * Wm;
  const double mean[3] = {m0, m1, m2};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wm), oy = (int)(i / Wm);
    float v[3] = {0.f, 0.f, 0.f};
    if (oy < oh && ox < ow) {
      auto px = [&](int yy, int xx, int c) -> int {
        int gx = x1 + xx;
        if (flip) gx = SW - 1 - gx;                      // im[:, ::-1, :] before the crop (worker:88-89)
        return (int)src[((size_t)(y1 + yy) * SW + gx) * 3 + c];
      };
      int res[3];
      if (AREA) {
        const int sx0 = 2 * ox, sy0 = 2 * oy;
        const bool whole = sx0 + 2 <= cw && sy0 + 2 <= ch;
        const int nx = sx0 + 2 <= cw ? 2 : 1, ny = sy0 + 2 <= ch ? 2 : 1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int sum = 0;
          for (int yy = 0; yy < ny; ++yy)
            for (int xx = 0; xx < nx; ++xx) sum += px(sy0 + yy, sx0 + xx, c);
          if (whole) {
            res[c] = (sum + 2) >> 2;
          } else {
            const int r = (int)rintf((float)sum / (float)(nx * ny));          // saturate_cast<uchar>((float)sum / count)
            res[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
          }
        }
      } else {
        float fx = (float)(((double)ox + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= cw - 1) { fx = 0.f; sx = cw - 1; }
        const int a0 = cv_sat_short((1.f - fx) * 2048.f), a1 = cv_sat_short(fx * 2048.f);
        const int sx1 = sx + 1 < cw ? sx + 1 : cw - 1;   // (a1 = 0 wherever sx + 1 leaves the row: OpenCV's S[sx] * 2048 branch)
        float fy = (float)(((double)oy + 0.5) * scale - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;                                 // rows are clipped, the coefficients are not
        const int b0 = cv_sat_short((1.f - fy) * 2048.f), b1 = cv_sat_short(fy * 2048.f);
        const int r0 = sy < 0 ? 0 : (sy < ch ? sy : ch - 1), r1 = sy + 1 < 0 ? 0 : (sy + 1 < ch ? sy + 1 : ch - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int t0 = px(r0, sx, c) * a0 + px(r0, sx1, c) * a1;
          const int t1 = px(r1, sx, c) * a0 + px(r1, sx1, c) * a1;
          res[c] = ((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2) & 0xFF;
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) v[j] = (float)((double)res[2 - j] - mean[2 - j]);   // uint8 - float64 -> float64 -> float32 (:71,117)
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) out[(size_t)j * Hm * Wm + i] = v[j];
  }
}

SN_EXPORT int sn_im_prepare(const uint8_t *d_src_bgr, int src_h, int src_w, int crop_x1, int crop_y1, int crop_x2, int crop_y2,
                            double scale, int flip, const double *pixel_means_bgr3, float *d_out, int out_h, int out_w,
                            int32_t *resized_hw2, sn_stream_t stream) {
  SN_REQUIRE(d_src_bgr && d_out && pixel_means_bgr3 && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0 && scale > 0.,
             "sn_im_prepare: bad arguments");
  const int x1 = crop_x1 < 0 ? 0 : crop_x1, y1 = crop_y1 < 0 ? 0 : crop_y1;
  const int x2 = crop_x2 > src_w ? src_w : crop_x2, y2 = crop_y2 > src_h ? src_h : crop_y2;
  SN_REQUIRE(x2 > x1 && y2 > y1, "sn_im_prepare: empty crop");
  const int cw = x2 - x1, ch = y2 - y1;
  const int rw = (int)lrint((double)cw * scale), rh = (int)lrint((double)ch * scale);   // cv::resize: saturate_cast<int>(n * f)
  SN_REQUIRE(rw > 0 && rh > 0, "sn_im_prepare: the resized image is empty (cv::resize asserts !dsize.empty())");
  if (resized_hw2) { resized_hw2[0] = rh; resized_hw2[1] = rw; }
  const int oh = rh < out_h ? rh : out_h, ow = rw < out_w ? rw : out_w;
  // cv::hal::resize: scale = 1 / inv_scale; INTER_LINEAR with an exact 2 x 2 decimation runs as INTER_AREA (fast)
  const double inv = 1. / scale;
  const int iscale = (int)lrint(inv);
  const bool area = fabs(inv - (double)iscale) < 2.220446049250313e-16 && iscale == 2;
  long blocks = ((long)out_h * out_w + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (area)
    hipLaunchKernelGGL(im_prepare_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), d_src_bgr, src_h, src_w, x1, y1,
                       cw, ch, /* scale_x = scale_y = */ inv, flip, pixel_means_bgr3[0], pixel_means_bgr3[1], pixel_means_bgr3[2], d_out, out_h, out_w, oh, ow);
  else
    hipLaunchKernelGGL(im_prepare_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), d_src_bgr, src_h, src_w, x1, y1,
                       cw, ch, /* scale_x = scale_y = */ inv, flip, pixel_means_bgr3[0], pixel_means_bgr3[1], pixel_means_bgr3[2], d_out, out_h, out_w, oh, ow);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// rois (B*R,5) f32 [batch, x1,y1,x2,y2], deltas (B,R,4) f32, im_info (B,3) f32 [h,w,scale] -> boxes (B,R,4) f64
__global__ __launch_bounds__(256) void bbox_decode_kernel(const float *__restrict__ rois, const float *__restrict__ deltas,
                                                          const float *__restrict__ im_info, double *__restrict__ out, int B, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * R) return;
  const int b = i / R;
  const double x1 = rois[i * 5 + 1], y1 = rois[i * 5 + 2], x2 = rois[i * 5 + 3], y2 = rois[i * 5 + 4];
  const double w = x2 - x1 + 1.0, h = y2 - y1 + 1.0;
  const double cx = x1 + 0.5 * (w - 1.0), cy = y1 + 0.5 * (h - 1.0);
  const float dx = deltas[i * 4 + 0], dy = deltas[i * 4 + 1], dw = deltas[i * 4 + 2], dh = deltas[i * 4 + 3];
  const double pcx = (double)dx * w + cx, pcy = (double)dy * h + cy;
  // np.exp on the float32 delta column returns float32: exp evaluated once and narrowed before the multiply
  const double pw = (double)(float)exp((double)dw) * w, ph = (double)(float)exp((double)dh) * h;
  double o[4] = {pcx - 0.5 * (pw - 1.0), pcy - 0.5 * (ph - 1.0), pcx + 0.5 * (pw - 1.0), pcy + 0.5 * (ph - 1.0)};
  // clip_boxes(boxes, im_shape = im_info[:2]): np.minimum(..., float32 h-1) promotes to float64
  const double wm = (double)(im_info[b * 3 + 1] - 1.f), hm = (double)(im_info[b * 3 + 0] - 1.f);
  const double sc = (double)im_info[b * 3 + 2];
  o[0] = fmax(fmin(o[0], wm), 0.0); o[1] = fmax(fmin(o[1], hm), 0.0);
  o[2] = fmax(fmin(o[2], wm), 0.0); o[3] = fmax(fmin(o[3], hm), 0.0);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[(size_t)i * 4 + k] = o[k] / sc;
}

SN_EXPORT int sn_bbox_decode(const float *d_rois, const float *d_deltas, const float *d_im_info, double *d_boxes, int B, int R,
                             sn_stream_t stream) {
  SN_REQUIRE(d_rois && d_deltas && d_im_info && d_boxes && B > 0 && R > 0, "sn_bbox_decode: bad arguments");
  hipLaunchKernelGGL(bbox_decode_kernel, dim3(sn_div_up(B * R, 256)), dim3(256), 0, sn_stream(stream), d_rois, d_deltas, d_im_info,
                     d_boxes, B, R);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// det_compact: the per-class score threshold of Tester.get_detections (lib/inference.py:289-295) and, for AutoFocus chips,
// the border pruning that follows it (:336-353 with check_valid :236-259), for every chip of a batch on the GPU.  The host
// loop does, per chip and class j, `inds = where(scores[:, j] > thresh)` -> rows [boxes[inds, 0:4], scores[inds, j]] (float64),
// then shifts the rows by the chip's origin and drops those within `delta` px of a chip border that is not an image border.
// Here one wave owns one (chip, class): a counting pass, then a writing pass that places the class's surviving rows (RoIs
// ascending, as np.where walks them) behind those of the classes before it -- the host receives rows already grouped by class
// plus the per-class counts and only slices.  All arithmetic in double, adds and compares only: bit-exact with numpy.
struct DetChips {
  double crop[64][4];     // chip [x1, y1, x2, y2] in image coordinates
  double wh[64][2];       // image width, height
};

__device__ __forceinline__ bool det_keep(const float *__restrict__ scores, const double *__restrict__ boxes, const DetChips &ch,
                                         int prune, float thresh, double delta, int b, int r, int j, int R, int NC, double *row) {
  const float s = scores[((size_t)b * R + r) * NC + j];
  if (!(s > thresh)) return false;
  const double *bx = boxes + ((size_t)b * R + r) * 4;
  double x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  if (prune) {
    const double cx1 = ch.crop[b][0], cy1 = ch.crop[b][1], cx2 = ch.crop[b][2], cy2 = ch.crop[b][3];
    x1 += cx1; x2 += cx1; y1 += cy1; y2 += cy1;
    // ~(abs(d - c) < delta): a NaN coordinate keeps the row, as the numpy mask does
    if (cx1 >= 0.5 && fabs(x1 - cx1) < delta) return false;
    if (cy1 >= 0.5 && fabs(y1 - cy1) < delta) return false;
    if (cx2 < ch.wh[b][0] - 0.5 && fabs(x2 - cx2) < delta) return false;
    if (cy2 < ch.wh[b][1] - 0.5 && fabs(y2 - cy2) < delta) return false;
  }
  row[0] = x1; row[1] = y1; row[2] = x2; row[3] = y2; row[4] = (double)s;
  return true;
}

// grid (NC - 1, B), one wave per block.  WRITE = false: counts[b][j - 1]; WRITE = true: rows, after the counts are complete.
template <bool WRITE>
__global__ __launch_bounds__(64) void det_compact_kernel(const float *__restrict__ scores, const double *__restrict__ boxes,
                                                         const DetChips ch, int prune, float thresh, double delta, int R, int NC,
                                                         double *__restrict__ rows, int *__restrict__ counts) {
  const int j = blockIdx.x + 1, b = blockIdx.y, lane = threadIdx.x;
  int base = 0;
  if (WRITE) {                                  // rows of the classes before this one
    for (int k = lane; k < j - 1; k += 64) base += counts[b * (NC - 1) + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
  }
  int n = 0;
  for (int r0 = 0; r0 < R; r0 += 64) {
    const int r = r0 + lane;
    double row[5];
    const bool keep = r < R && det_keep(scores, boxes, ch, prune, thresh, delta, b, r, j, R, NC, row);
    const unsigned long long m = __ballot(keep);
    if (WRITE && keep) {
      const int at = base + n + __popcll(m & ((1ull << lane) - 1ull));
      double *dst = rows + ((size_t)b * (NC - 1) * R + at) * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) dst[k] = row[k];
    }
    n += __popcll(m);
  }
  if (!WRITE && lane == 0) counts[b * (NC - 1) + j - 1] = n;
}

SN_EXPORT int sn_det_compact(const float *d_scores, const double *d_boxes, const double *h_crops, const double *h_im_wh,
                             float thresh, double delta, int B, int R, int NC, double *d_rows, int32_t *d_counts,
                             sn_stream_t stream) {
  SN_REQUIRE(d_scores && d_boxes && d_rows && d_counts && B > 0 && B <= 64 && R > 0 && NC > 1,
             "sn_det_compact: bad arguments (at most 64 chips per call)");
  SN_REQUIRE((h_crops == nullptr) == (h_im_wh == nullptr), "sn_det_compact: crops and image sizes come together");
  DetChips ch;
  memset(&ch, 0, sizeof(ch));
  if (h_crops)
    for (int b = 0; b < B; ++b) {
      for (int k = 0; k < 4; ++k) ch.crop[b][k] = h_crops[b * 4 + k];
      ch.wh[b][0] = h_im_wh[b * 2];
      ch.wh[b][1] = h_im_wh[b * 2 + 1];
    }
  const dim3 grid(NC - 1, B);
  hipLaunchKernelGGL(det_compact_kernel<false>, grid, dim3(64), 0, sn_stream(stream), d_scores, d_boxes, ch, h_crops ? 1 : 0, thresh,
                     delta, R, NC, d_rows, d_counts);
  hipLaunchKernelGGL(det_compact_kernel<true>, grid, dim3(64), 0, sn_stream(stream), d_scores, d_boxes, ch, h_crops ? 1 : 0, thresh,
                     delta, R, NC, d_rows, d_counts);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-scale aggregation on the device: the regrouping half of Tester.aggregate (lib/inference.py:166-190) and its
// MAX_PER_IMAGE rule (:203-211), over the rows sn_det_compact left in HBM -- nothing of a chip's detections goes to the host
// between the forward pass and the final boxes.  A PART is one chip of one scale: rows (n_p, 5) float64 grouped by class +
// rows per class.  The reference walks classes x images x scales x chips, filters every chip's rows by the scale's valid range
// (_valid_range_filter :176-186: areas = (y2 - y1) * (x2 - x1) on the float32 rows, compared in float32) and stacks what is left
// per (image, class) for the NMS pool; here
//   sn_aggregate_count    one wave per (part, class): rows inside the part's valid range            -> kept (P, nc)
//   (host: the (image, class, part) exclusive scan of `kept`: a few thousand integers)
//   sn_aggregate_scatter  one wave per (part, class): the rows that pass, narrowed to float32, written in order behind those of
//                         the parts before it -> the stacked rows of all (image, class) problems, (image, class, scale, chip, row)
//                         order = the array sn_soft_nms_batch takes
//   sn_det_cap_per_image  after the NMS: one workgroup per image; with more than max_per_image rows over all classes every class
//                         keeps its rows whose score reaches the max_per_image-th best of the image (ties stay): radix select on
//                         the score bits, then a stable in-place compaction per class.
// Order-preserving throughout (soft-NMS breaks score ties by position).  Tested equal to the host statement.
struct AggPart {
  const double *rows;      // (n_p, 5) float64, grouped by class
  const int32_t *counts;   // (nc) rows per class
  float lo2, hi2;          // the scale's valid range squared as float32; <= 0: no bound
};

template <bool WRITE>
__global__ __launch_bounds__(64) void aggregate_kernel(const AggPart *__restrict__ parts, int nc, int32_t *__restrict__ kept,
                                                       const int32_t *__restrict__ dst_off, float *__restrict__ out) {
  const int j = blockIdx.x, p = blockIdx.y, lane = threadIdx.x;
  const AggPart part = parts[p];
  int base = 0;
  for (int k = lane; k < j; k += 64) base += part.counts[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) base += __shfl_xor(base, o);
  const int cnt = part.counts[j];
  const double *src = part.rows + (size_t)base * 5;
  float *dst = WRITE ? out + (size_t)dst_off[(size_t)p * nc + j] * 5 : nullptr;
  int n = 0;
  for (int r0 = 0; r0 < cnt; r0 += 64) {
    const int r = r0 + lane;
    bool keep = false;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < cnt) {
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = (float)src[(size_t)r * 5 + k];
      const float area = (v[3] - v[1]) * (v[2] - v[0]);           // float32 products, as _valid_range_filter
      keep = !(part.lo2 > 0.f && !(area > part.lo2)) && !(part.hi2 > 0.f && !(area <= part.hi2));
    }
    const unsigned long long m = __ballot(keep);
    if (WRITE && keep) {
      float *d = dst + (size_t)(n + __popcll(m & ((1ull << lane) - 1ull))) * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) d[k] = v[k];
    }
    n += __popcll(m);
  }
  if (!WRITE && lane == 0) kept[(size_t)p * nc + j] = n;
}

SN_EXPORT int sn_aggregate_count(const void *d_parts, int P, int nc, int32_t *d_kept, sn_stream_t stream) {
  SN_REQUIRE(d_parts && d_kept && P > 0 && nc > 0 && P <= 65535, "sn_aggregate_count: bad arguments");
  hipLaunchKernelGGL(aggregate_kernel<false>, dim3(nc, P), dim3(64), 0, sn_stream(stream), (const AggPart *)d_parts, nc, d_kept,
                     (const int32_t *)nullptr, (float *)nullptr);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_aggregate_scatter(const void *d_parts, int P, int nc, const int32_t *d_dst_off, float *d_out_rows, sn_stream_t stream) {
  SN_REQUIRE(d_parts && d_dst_off && d_out_rows && P > 0 && nc > 0 && P <= 65535, "sn_aggregate_scatter: bad arguments");
  hipLaunchKernelGGL(aggregate_kernel<true>, dim3(nc, P), dim3(64), 0, sn_stream(stream), (const AggPart *)d_parts, nc,
                     (int32_t *)nullptr, d_dst_off, d_out_rows);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// rows (total, 5) f32, problems q = image * nc + class at off[q] .. off[q] + count[q] (count: survivors of the NMS, a prefix of the
// problem's segment).  In place: count[q] shrinks, the kept rows move to the front of the segment in order.
__device__ __forceinline__ unsigned score_key(float s) {            // order-preserving map float32 -> uint32 (any sign, no NaN)
  const unsigned u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void det_cap_kernel(float *__restrict__ rows, const int32_t *__restrict__ off,
                                                      int32_t *__restrict__ count, int nc, int max_per_image) {
  __shared__ int hist[256];
  __shared__ int s_total, s_bucket, s_above;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int q0 = img * nc;
  if (tid == 0) {
    int t = 0;
    for (int j = 0; j < nc; ++j) t += count[q0 + j];
    s_total = t;
  }
  __syncthreads();
  const int total = s_total;
  if (total <= max_per_image) return;
  // the max_per_image-th largest key: 4 radix passes from the top byte; `want` = rank (1-based, from the top) still to find
  unsigned prefix = 0, mask = 0;
  int want = max_per_image;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    for (int j = 0; j < nc; ++j) {
      const float *seg = rows + (size_t)off[q0 + j] * 5;
      const int n = count[q0 + j];
      for (int r = tid; r < n; r += 256) {
        const unsigned k = score_key(seg[(size_t)r * 5 + 4]);
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int above = 0, b = 255;
      for (; b > 0; --b) {
        if (above + hist[b] >= want) break;
        above += hist[b];
      }
      s_bucket = b;
      s_above = above;
    }
    __syncthreads();
    prefix |= (unsigned)s_bucket << shift;
    mask |= 255u << shift;
    want -= s_above;
    __syncthreads();
  }
  const unsigned thresh = prefix;               // key of the max_per_image-th best score: keep key >= thresh (ties stay)
  // stable in-place compaction, one wave per class at a time (a chunk is read whole before any of it is written, and writes
  // land at or before the positions read)
  const int wave = tid >> 6, lane = tid & 63;
  for (int j = wave; j < nc; j += 4) {
    float *seg = rows + (size_t)off[q0 + j] * 5;
    const int n = count[q0 + j];
    int kept = 0;
    for (int r0 = 0; r0 < n; r0 += 64) {
      const int r = r0 + lane;
      float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      bool keep = false;
      if (r < n) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = seg[(size_t)r * 5 + k];
        keep = score_key(v[4]) >= thresh;
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        float *d = seg + (size_t)(kept + __popcll(m & ((1ull << lane) - 1ull))) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] = v[k];
      }
      kept += __popcll(m);
    }
    if (lane == 0) count[q0 + j] = kept;
  }
}

SN_EXPORT int sn_det_cap_per_image(float *d_rows, const int32_t *d_off, int32_t *d_count, int num_images, int nc, int max_per_image,
                                   sn_stream_t stream) {
  SN_REQUIRE(d_rows && d_off && d_count && num_images > 0 && nc > 0 && max_per_image > 0, "sn_det_cap_per_image: bad arguments");
  hipLaunchKernelGGL(det_cap_kernel, dim3(num_images), dim3(256), 0, sn_stream(stream), d_rows, d_off, d_count, nc, max_per_image);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
