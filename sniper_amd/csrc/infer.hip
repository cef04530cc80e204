// infer.hip -- the two host loops either side of the test-time graph, on the GPU.
//
//  * im_prepare: im_worker.worker / worker_autofocus (lib/data_utils/data_workers.py:49-121) after the JPEG decode:
//    optional horizontal flip, integer crop, bilinear resize by `scale`, zero padding to (3, Hm, Wm) float32 with
//    channel j = BGR[2 - j] - PIXEL_MEANS[2 - j].  The reference resizes with cv2.resize(INTER_LINEAR) on uint8
//    (fixed-point coefficients, rounded uint8 output); OpenCV is not in this image, so the arithmetic here --
//    half-pixel centres, source size round(n * scale), float bilinear, result rounded to the nearest integer like the
//    uint8 image cv2 returns -- is a documented restatement: PARITY UNPINNED.
//  * bbox_decode: bbox_pred (= nonlinear_pred, lib/bbox/bbox_transform.py:93-130) + clip_boxes (:35-50) + division by
//    the image scale, as lib/inference.py:127-131 applies them per chip; float64 like the numpy original
//    (boxes.astype(np.float)), compiled with -ffp-contract=off, exp() in double.
#include "common.h"

__global__ __launch_bounds__(256) void im_prepare_kernel(const unsigned char *__restrict__ src, int SH, int SW, int x1, int y1,
                                                         int cw, int ch, float scale, int flip, float m0, float m1, float m2,
                                                         float *__restrict__ out, int Hm, int Wm, int oh, int ow) {
  const long total = (long)Hm * Wm;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wm), oy = (int)(i / Wm);
    float v[3] = {0.f, 0.f, 0.f};
    if (oy < oh && ox < ow) {
      // cv2.resize with fx = fy = scale: source coordinate of the centre of destination pixel (ox, oy)
      const float inv = 1.f / scale;
      float fx = ((float)ox + 0.5f) * inv - 0.5f, fy = ((float)oy + 0.5f) * inv - 0.5f;
      int sx = (int)floorf(fx), sy = (int)floorf(fy);
      float ax = fx - (float)sx, ay = fy - (float)sy;
      if (sx < 0) { sx = 0; ax = 0.f; }
      if (sy < 0) { sy = 0; ay = 0.f; }
      if (sx >= cw - 1) { sx = cw - 1 > 0 ? cw - 2 : 0; ax = cw > 1 ? 1.f : 0.f; }
      if (sy >= ch - 1) { sy = ch - 1 > 0 ? ch - 2 : 0; ay = ch > 1 ? 1.f : 0.f; }
      const int sx1 = sx + (cw > 1 ? 1 : 0), sy1 = sy + (ch > 1 ? 1 : 0);
      auto px = [&](int yy, int xx, int c) -> float {
        int gx = x1 + xx;
        if (flip) gx = SW - 1 - gx;                      // im[:, ::-1, :] before the crop (worker:88-89)
        return (float)src[((size_t)(y1 + yy) * SW + gx) * 3 + c];
      };
      const float mean[3] = {m0, m1, m2};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int c = 2 - j;                              // output channel j = BGR[2 - j]
        const float top = px(sy, sx, c) * (1.f - ax) + px(sy, sx1, c) * ax;
        const float bot = px(sy1, sx, c) * (1.f - ax) + px(sy1, sx1, c) * ax;
        v[j] = rintf(top * (1.f - ay) + bot * ay) - mean[c];
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) out[(size_t)j * Hm * Wm + i] = v[j];
  }
}

SN_EXPORT int sn_im_prepare(const uint8_t *d_src_bgr, int src_h, int src_w, int crop_x1, int crop_y1, int crop_x2, int crop_y2,
                            float scale, int flip, const float *pixel_means_bgr3, float *d_out, int out_h, int out_w,
                            int32_t *resized_hw2, sn_stream_t stream) {
  SN_REQUIRE(d_src_bgr && d_out && pixel_means_bgr3 && src_h > 0 && src_w > 0 && out_h > 0 && out_w > 0 && scale > 0.f,
             "sn_im_prepare: bad arguments");
  const int x1 = crop_x1 < 0 ? 0 : crop_x1, y1 = crop_y1 < 0 ? 0 : crop_y1;
  const int x2 = crop_x2 > src_w ? src_w : crop_x2, y2 = crop_y2 > src_h ? src_h : crop_y2;
  SN_REQUIRE(x2 > x1 && y2 > y1, "sn_im_prepare: empty crop");
  const int cw = x2 - x1, ch = y2 - y1;
  int rw = (int)lrintf((float)cw * scale), rh = (int)lrintf((float)ch * scale);   // cv2: saturate_cast<int>(n * f)
  if (rw < 1) rw = 1;
  if (rh < 1) rh = 1;
  if (resized_hw2) { resized_hw2[0] = rh; resized_hw2[1] = rw; }
  const int oh = rh < out_h ? rh : out_h, ow = rw < out_w ? rw : out_w;
  long blocks = ((long)out_h * out_w + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(im_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), d_src_bgr, src_h, src_w, x1, y1, cw,
                     ch, scale, flip, pixel_means_bgr3[0], pixel_means_bgr3[1], pixel_means_bgr3[2], d_out, out_h, out_w, oh, ow);
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
