/*
 * sniper_hip.h -- C ABI of libsniper_hip.so, the MI355X (gfx950) kernel library behind the SNIPER
 * hot path.  Plain C: pointers, sizes, no C++/torch types.
 *
 * Conventions ("Inner boundary"):
 *   - every pointer named d_* / documented "device" is a HIP device pointer owned by the caller;
 *   - no allocation and no synchronisation inside any call: kernels are enqueued on `stream` (a hipStream_t passed
 *     as void*; NULL = default stream) and the call returns; the calls are re-entrant per stream and per device;
 *   - process-wide state is limited to the explicit test / tuning hooks sn_debug_option, sn_conv_tune,
 *     sn_conv_wgrad_impl, sn_conv_trace and sn_conv_wgrad_trace (atomics, defaults = production behaviour; their
 *     environment-variable spellings are read ONCE at load time) -- no hot call reads the environment or mutates a
 *     global; the per-device "large LDS" opt-in of two kernels is applied once per (kernel, device) under a mutex;
 *   - scratch memory comes from the caller: sn_*_workspace_bytes() tells how much;
 *   - return value: 0 = ok, SN_ERR_* otherwise (argument errors are detected before any launch).
 * The only native ABI the reference itself defines is
 *     void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
 *               float nms_overlap_thresh, int device_id);            (lib/nms/gpu_nms.hpp:1-2)
 * which sn_nms_host() replaces one-to-one; everything else replaces a Cython/C++ extension entry
 * point or an operator of the un-vendored SNIPER-mxnet fork and cites it below.
 */
#ifndef SNIPER_HIP_H
#define SNIPER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *sn_stream_t; /* hipStream_t */

enum { SN_OK = 0, SN_ERR_ARG = 1, SN_ERR_HIP = 2, SN_ERR_WORKSPACE = 3, SN_ERR_UNSUPPORTED = 4 };

/* Library / device sanity. */
int sn_version(void);
const char *sn_last_error(void); /* thread-local text of the last failure */
/* Test / A-B switch (process-wide, atomic; no reference counterpart): "proposal_full_sort" = 1 orders the proposals with the
 * general global-memory bitonic sort instead of radix select + LDS sort, "nms_full_mask" = 1 runs the full bitmask + scan
 * instead of the lazy kernel.  Results are identical either way (that is what the tests use it for; environment spellings, read
 * once at load: SNIPER_FULL_SORT, SNIPER_NMS_FULL). */
int sn_debug_option(const char *name, int value);

/* ------------------------------------------------------------------ box geometry ------------- */
/* bbox_overlaps_cython / ignore_overlaps_cython (lib/bbox/bbox.pyx:17-57, 59-95).
 * boxes (N,4) f64, query (K,4) f64 -> out (N,K) f64 row-major.  mode 0 = IoU, 1 = intersection/query-area. */
int sn_iou_f64(const double *d_boxes, int N, const double *d_query, int K, double *d_out, int mode, sn_stream_t stream);

/* ------------------------------------------------------------------ chip generation ---------- */
/* chips::cgenerate (lib/chips/cchips.cpp:54-177) for a ragged batch of U (image, scale) units.
 *   d_boxes       (total_boxes,4) f32, already scaled + clipped (chip_generator.py:24);
 *   d_box_off     (U+1) i32 prefix offsets into d_boxes;
 *   d_meta        (U,4) i32 = width, height, chipsize, stride;
 *   d_perm        (total_cand) i32 candidate order after the shuffle of cchips.cpp:117, or NULL for identity;
 *   d_cand_off    (U+1) i32 prefix offsets into d_perm (= candidate counts, sn_chips_num_candidates);
 *   d_mask_ws     workspace, sn_chips_workspace_bytes(total_cand, max_boxes_per_unit);
 *   d_out_chips   (total_boxes,4) f32: unit u writes its chips at row d_box_off[u] (a unit never
 *                 selects more chips than it has boxes);  d_out_ids same layout, slot index in the
 *                 shuffled candidate list;  d_out_count (U) i32.
 */
int sn_chips_num_candidates(int width, int height, int chipsize, int stride);
size_t sn_chips_workspace_bytes(int total_cand, int max_boxes_per_unit);
int sn_chips_generate_batch(const float *d_boxes, const int32_t *d_box_off, const int32_t *d_meta, const int32_t *d_perm,
                            const int32_t *d_cand_off, int U, int max_boxes_per_unit, void *d_mask_ws, float *d_out_chips,
                            int32_t *d_out_ids, int32_t *d_out_count, sn_stream_t stream);

/* Box -> chip assignment of chip_worker.box_assigner (lib/data_utils/data_workers.py:516-535, 557-572) for a
 * ragged batch of U (image, scale) units: d_chips (total_chips,4) f64 + d_chip_off (U+1); d_boxes (total_boxes,4)
 * f64 + d_box_off (U+1) + d_unit_of_box (total_boxes); d_range (U,2) f64 valid range; d_mode (U) 0 = coarsest
 * (area >= lo), 1 = area <= hi, 2 = area < hi.  d_out_chip (total_boxes): chip index inside the unit or -1. */
int sn_assign_boxes_batch(const double *d_chips, const int32_t *d_chip_off, const double *d_boxes, const int32_t *d_box_off,
                          const double *d_range, const int32_t *d_mode, const int32_t *d_unit_of_box, int U, int total_boxes,
                          int32_t *d_out_chip, sn_stream_t stream);

/* ------------------------------------------------------------------ RPN anchor labelling ----- */
/* anchor_worker.worker (lib/data_utils/data_workers.py:164-371) + the dense scatter of
 * MNIteratorE2E._get_batch (lib/iterators/MNIteratorE2E.py:175-194), batched over B chips.
 * Geometry is fixed per call: F x F feature cells, A anchors per cell, d_base_anchors (A,4) f64 =
 * generate_anchors() output, feat_stride, chip height/width (im_info).
 * Per chip b:
 *   d_gt         (B,G,4) f32 GT boxes in image coordinates (rows >= d_ngt[b] ignored), G <= 128
 *   d_gt_cls     (B,G)   f32 class id
 *   d_gt_inchip  (B,G)   u8  1 if that GT row is in props_in_chips of this chip (gtids ∩ nids)
 *   d_ngt        (B)     i32
 *   d_crop       (B,2)   f64 chip origin x,y in image coordinates;  d_scale (B) f32 im_scale
 * Sub-sampling (data_workers.py:327-338): the kept fg / bg anchors are those with the smallest
 * 47-bit key (d_keys[b][anchor] << 15 | anchor); d_keys (B, A*F*F) u32 in reference anchor order
 * (cell-major, anchor-minor) is either supplied by the host (bit-exact replay of numpy's draws) or
 * NULL, in which case keys come from a counter-based hash of (seed, b, anchor).
 * Outputs (all device, layouts of the reference batch tensors):
 *   d_label (B, A*F*F) f32 in (a,y,x) order, values {-1,0,1};
 *   d_bbox_target, d_bbox_weight (B, 4A, F, F) f32;  d_gt_out (B,100,5) f32 (-1 padded);
 *   d_counts (B,4) i32 = n_inside, n_fg_before, n_bg_before, n_valid_gt  (diagnostics / host RNG replay);
 *   d_label_pre (B, A*F*F) i8 optional (may be NULL): labels before sub-sampling in reference
 *   anchor order, -2 for anchors outside the chip.
 */
size_t sn_anchor_workspace_bytes(int B, int A, int F, int G);
int sn_anchor_assign(const float *d_gt, const float *d_gt_cls, const uint8_t *d_gt_inchip, const int32_t *d_ngt,
                     const double *d_crop, const float *d_scale, int B, int G, const double *d_base_anchors, int A, int F,
                     int feat_stride, int im_h, int im_w, double pos_thresh, double neg_thresh, int rpn_batch, int num_fg,
                     const uint32_t *d_keys, uint64_t seed, void *d_ws, float *d_label, float *d_bbox_target,
                     float *d_bbox_weight, float *d_gt_out, int32_t *d_counts, int8_t *d_label_pre, sn_stream_t stream);

/* AutoFocus FocusPixel labels (gen_mask, lib/data_utils/data_workers.py:165-192) for the same chip batch: d_mask (B, F*F) f32
 * in {1, -1, 0} = small object / don't care / background; thresholds TRAIN.AUTO_FOCUS_DC_LOW, _SMALL_THRESH, _DC_HIGH. */
int sn_focus_mask(const float *d_gt, const int32_t *d_ngt, const double *d_crop, const float *d_scale, int B, int G, int F,
                  int feat_stride, int im_h, int im_w, float dc_low, float small_thresh, float dc_high, float *d_mask,
                  sn_stream_t stream);

/* ------------------------------------------------------------------ NMS ----------------------- */
/* Bitmask hard NMS (lib/nms/nms_kernel.cu:34-78 mask, :118-140 scan; IoU > thresh suppresses),
 * batched: d_boxes (B, N, dim) f32 sorted by descending score, rows >= d_n[b] ignored (d_n may be
 * NULL = all N).  Keeps at most max_keep survivors (<=0: N).  d_keep (B, max_keep) i32 row indices,
 * d_nkeep (B) i32.  The mask never leaves the device. */
size_t sn_nms_workspace_bytes(int B, int N);
int sn_nms_batch(const float *d_boxes, const int32_t *d_n, int B, int N, int dim, float thresh, int max_keep, void *d_ws,
                 int32_t *d_keep, int32_t *d_nkeep, sn_stream_t stream);
/* Drop-in for the reference's _nms() (host buffers, synchronous, allocates like the original).
 * Returns 0/err; keep_out[num_out] are row indices into the (sorted) input. */
int sn_nms_host(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                float nms_overlap_thresh, int device_id);

/* Soft-NMS (cpu_soft_nms, lib/nms/cpu_nms.pyx:17-110; method 1 linear, 2 gaussian, else hard) for P independent
 * problems: d_boxes (total_rows,5) f32 [x1,y1,x2,y2,score], problem p owns rows [d_off[p], d_off[p+1]), max_n = the
 * largest problem.  No size cap, like the reference: problems of at most sn_soft_nms_max_boxes() boxes run out of one
 * workgroup's LDS; larger ones run the same phases in global memory and need d_ws =
 * sn_soft_nms_workspace_bytes(total_rows) of scratch (NULL is accepted when max_n fits).  In place, like the reference: on
 * return rows [d_off[p], d_off[p] + d_count[p]) are the surviving boxes in the reference's order with their decayed scores. */
size_t sn_soft_nms_max_boxes(void);
size_t sn_soft_nms_workspace_bytes(size_t total_rows);
int sn_soft_nms_batch(float *d_boxes, const int32_t *d_off, int P, int max_n, size_t total_rows, float sigma, float Nt,
                      float threshold, int method, void *d_ws, int32_t *d_count, sn_stream_t stream);

/* ------------------------------------------------------------------ test-time host loops ----- */
/* im_worker.worker / worker_autofocus after the decode (lib/data_utils/data_workers.py:49-121): d_src_bgr (H,W,3) u8,
 * flip, crop [x1,x2) x [y1,y2) (clamped to the image), cv2.resize(fx = fy = scale, INTER_LINEAR) in OpenCV's 8-bit
 * fixed-point arithmetic (scale is the double the reference passes as fx / fy), zero padded (3,out_h,out_w) f32 with
 * channel j = BGR[2-j] - pixel_means_bgr3[2-j] (host array of 3 doubles: numpy subtracts in float64).  resized_hw2
 * (host, may be NULL) receives the resized height, width (im_info).  Bit-exact against oracle/cv_resize.py, the
 * restatement of OpenCV's published algorithm. */
int sn_im_prepare(const uint8_t *d_src_bgr, int src_h, int src_w, int crop_x1, int crop_y1, int crop_x2, int crop_y2, double scale,
                  int flip, const double *pixel_means_bgr3, float *d_out, int out_h, int out_w, int32_t *resized_hw2,
                  sn_stream_t stream);
/* bbox_pred + clip_boxes + /scale as applied per chip by lib/inference.py:127-131 (lib/bbox/bbox_transform.py:93-130,35-50):
 * rois (B*R,5) f32 with rows of chip b contiguous, deltas (B,R,4) f32, im_info (B,3) f32 -> boxes (B,R,4) f64. */
int sn_bbox_decode(const float *d_rois, const float *d_deltas, const float *d_im_info, double *d_boxes, int B, int R,
                   sn_stream_t stream);
/* Bounding rectangles (x, y, w, h) of the contours cv2.findContours(mask, RETR_LIST) reports for a 0 / non-zero HOST mask, with
 * cv2.boundingRect: mode 0 = border following (Suzuki & Abe 1985, OpenCV contours.cpp), in cv2's order (newest contour first) --
 * what sn_focus_chips_host walks; mode 1 = 8-connected components + enclosed 4-connected background components grown by one cell,
 * raster order (the independent cross-check: same rectangles).  rects_xywh (cap,4) i32. */
int sn_focus_rects_host(const uint8_t *mask_hw, int H, int W, int mode, int32_t *rects_xywh, int cap, int32_t *n_rects);
/* `gmask` of lib/chips/chips_inference.py:12-89 on the HOST (no device work, no stream): FocusPixel map (H,W) f32 -> FocusChips.
 * threshold (>= thresh), cv2.dilate d x d, bounding rectangles of the RETR_LIST contours (8-connected components + holes), minimum
 * side ms cells, paint-and-repeat until the chip count is stable, x16, clamp to (im_width, im_height), / cscale.
 * chips_xyxy (max_chips,4) f64 host, n_chips host.  Restatement of the cv2 calls: parity unpinned (no OpenCV here). */
int sn_focus_chips_host(const float *map_hw, int H, int W, int d, float thresh, int ms, double im_width, double im_height,
                        double cscale, double *chips_xyxy, int max_chips, int32_t *n_chips);
/* The regrouping half of Tester.aggregate (lib/inference.py:166-190) on the HOST in one pass: parts = the chips of all scales as
 * sn_det_compact returned them (part_rows[p] = address of float64 (n_p,5) rows grouped by class, lens (P,nc) rows per class), the
 * parts of image i = part_of_image[i] .. part_of_image[i+1]-1 in (scale, chip) order, range2 (P,2) f32 = the scale's valid range
 * squared (<= 0: no bound; areas and comparisons in float32, _valid_range_filter :176-186).  -> out_rows (total,5) f32: the rows of
 * all (image, class) problems back to back in (image, class, scale, chip, row) order; out_sizes (num_images*nc) rows per problem. */
int sn_aggregate_problems_host(const uint64_t *part_rows, const int64_t *lens, const int32_t *part_of_image, const float *range2,
                               int P, int nc, int num_images, float *out_rows, long capacity_rows, int64_t *out_sizes,
                               int64_t *total_rows);
/* The same regrouping ON THE DEVICE, over the rows sn_det_compact left in HBM (Tester.aggregate, lib/inference.py:166-190): d_parts
 * = P records {const double *rows; const int32_t *counts; float lo2, hi2;} (device addresses of one chip's float64 rows grouped
 * by class and its nc rows-per-class; the scale's valid range squared as float32, <= 0: no bound), in (image, scale, chip) order.
 * sn_aggregate_count -> d_kept (P,nc) i32: rows of (part, class) inside the range; the caller scans them into d_dst_off (P,nc):
 * first row of (part, class) in the stacked output, (image, class, scale, chip) order; sn_aggregate_scatter writes the rows that
 * pass, narrowed to float32, there in order -- the array sn_soft_nms_batch takes.  Equal to sn_aggregate_problems_host. */
int sn_aggregate_count(const void *d_parts, int P, int nc, int32_t *d_kept, sn_stream_t stream);
int sn_aggregate_scatter(const void *d_parts, int P, int nc, const int32_t *d_dst_off, float *d_out_rows, sn_stream_t stream);
/* TEST.MAX_PER_IMAGE (lib/inference.py:203-211) on the device, after the NMS: problems q = image*nc + class hold d_count[q] rows
 * at d_off[q]; an image with more than max_per_image rows over its classes keeps, per class, the rows whose score reaches the
 * max_per_image-th best score of the image (ties stay) -- in place, in order; d_count is updated. */
int sn_det_cap_per_image(float *d_rows, const int32_t *d_off, int32_t *d_count, int num_images, int nc, int max_per_image,
                         sn_stream_t stream);
/* The per-class score threshold of Tester.get_detections (lib/inference.py:289-295: inds = where(scores[:, j] > thresh), rows
 * hstack(boxes[inds, 0:4], scores[inds, j])) and, when h_crops != NULL, the AutoFocus border pruning that follows it (:336-353,
 * check_valid :236-259: rows shifted by the chip origin, dropped within `delta` px of a chip border that is not an image
 * border), for the B <= 64 chips of a batch.  scores (B,R,NC) f32, boxes (B,R,4) f64 (sn_bbox_decode); h_crops (B,4) f64 and
 * h_im_wh (B,2) f64 are HOST arrays (chip [x1,y1,x2,y2] in image coordinates; image width, height), passed to the kernel by
 * value.  -> rows (B,(NC-1)*R,5) f64: the surviving rows of chip b, grouped by class 1..NC-1, RoIs ascending inside a class;
 * counts (B,NC-1) i32 rows per class.  float64 adds and compares only: bit-exact with the numpy loops. */
int sn_det_compact(const float *d_scores, const double *d_boxes, const double *h_crops, const double *h_im_wh, float thresh,
                   double delta, int B, int R, int NC, double *d_rows, int32_t *d_counts, sn_stream_t stream);

/* ================================================================ network operators ===========
 * The graph operators of the un-vendored SNIPER-mxnet fork, at the call sites of
 * symbols/faster/resnet_mx_101_e2e.py / mobilenetv2_e2e.py.  Activations are channels-last fp16
 * ("NHWC", explicit pixel stride in elements so a tensor may be a channel slice of a wider buffer);
 * loss-side tensors are fp32 in the reference's NCHW order.  Weights are [Cout][KH*KW][Cin].
 * dtype codes: 0 = fp16, 1 = fp32. */

/* Convolution / FullyConnected forward (resnet_mx_101_e2e.py:43-66,147-155,256,288-303).
 * y[n,oy,ox,co] = bias[co] + residual + sum x[n, oy*s-p+kh*d, ox*s-p+kw*d, ci] * w[co][kh*KW+kw][ci];
 * optional ReLU; output fp16 or fp32.  FC = 1x1 convolution over H=W=1. */
int sn_conv_fwd(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W, int Cin,
                int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW, int stride, int pad, int dil,
                int relu, int out_f32, sn_stream_t stream);
/* sn_conv_fwd (fp16 output) with the contraction split over several copies of the tile grid, for launches with far fewer output
 * tiles than CUs (test-time batches of two FocusChips: 82 tiles on 256 CUs, each a latency-bound chain of K-steps): fp32 partial
 * tiles in `ws`, added in split order together with bias / residual / ReLU (deterministic).  The workspace query returns 0 when the
 * layer would not be split (enough tiles, a short contraction, a layer the pipelined kernel does not take): call sn_conv_fwd then --
 * sn_conv_fwd_splitk itself also falls back to it.  Same operator as sn_conv_fwd (symbols/faster/resnet_mx_101_e2e.py:43-66 at
 * test time, BatchNorm folded); the sum is formed in fp32 across the splits and rounded once. */
size_t sn_conv_fwd_splitk_workspace_bytes(int N, int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride,
                                          int res_pix_stride, int KH, int KW, int stride, int pad, int dil);
int sn_conv_fwd_splitk(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W, int Cin,
                       int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW, int stride, int pad,
                       int dil, int relu, void *ws, size_t ws_bytes, sn_stream_t stream);
/* ... with an fp32 result and no residual (sn_conv_fwd's out_f32 = 1): the offset FullyConnected of the deformable RoI pooling
 * (resnet_mx_101_e2e.py:288-291: 2 x 7 x 7 = 98 outputs over the RoIs of a test batch -- 10 tiles, 196 K-steps each -- read as the
 * pooling's fp32 `trans`).  Any Cout; same workspace query. */
int sn_conv_fwd_splitk_f32(const void *x, const void *w, const float *bias, float *y, int N, int H, int W, int Cin, int in_pix_stride,
                           int Cout, int out_pix_stride, int KH, int KW, int stride, int pad, int dil, int relu, void *ws,
                           size_t ws_bytes, sn_stream_t stream);
/* Forward convolution with a SECOND output (test-time residual units): y as sn_conv_fwd / sn_conv_fwd_splitk write it (fp16) and
 * y2 = act(y2_scale * y + y2_shift) of the stored y, fp16, pixel stride y2_pix_stride -- the moving-statistics BatchNorm + ReLU that
 * opens the next pre-activation unit (resnet_mx_101_e2e.py:38-40) and reads the residual sum this epilogue writes.  ws / ws_bytes:
 * the split-K scratch of sn_conv_fwd_splitk_workspace_bytes (0 / NULL: never split).  sn_conv_fwd_dual_ok -> 1 when the layer
 * qualifies (pipelined kernel, 16-byte rows), else use sn_conv_fwd + sn_bn_apply. */
int sn_conv_fwd_dual_ok(int N, int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH,
                        int KW, int stride, int pad, int dil, int y2_pix_stride);
int sn_conv_fwd_dual(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W, int Cin,
                     int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW, int stride, int pad, int dil,
                     int relu, void *y2, int y2_pix_stride, const float *y2_scale, const float *y2_shift, int y2_relu, void *ws,
                     size_t ws_bytes, sn_stream_t stream);
/* Kernel-selection override for sn_conv_fwd / sn_conv_dgrad (tuning hook, tools/conv_tune.py; no reference counterpart):
 * -1 = built-in per-layer table (default), 0 = the register-staged kernel only, 4 / 5 / 6 / 7 / 14 / 16 / 18 = that LDS-DMA tile
 * configuration for every layer that qualifies (see conv_dma.hip; other numbers are rejected).  Results are identical up to fp32 summation order inside a tile's K loop
 * (the K order itself does not change).  Process-wide; set it while no launch is in flight. */
int sn_conv_tune(int cfg);

/* sn_conv_fwd that also emits the BatchNorm statistics of its fp16 output (sum and sum of squares of the STORED values per
 * row tile): stats (blocks, 2, Cout) fp32, blocks = sn_conv_fwd_stats_blocks(...) (0: the layer does not qualify -- narrow or
 * unaligned layers -- use sn_conv_fwd + sn_bn_stats).  Consumed by sn_bn_finalize_blocks: the batch-statistics pass of
 * BatchNorm(train) after a convolution (resnet_mx_101_e2e.py:38-66) costs no extra read of the tensor. */
int sn_conv_fwd_stats_blocks(int N, int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride,
                             int KH, int KW, int stride, int pad, int dil);
int sn_conv_fwd_stats(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W, int Cin,
                      int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW, int stride, int pad,
                      int dil, int relu, float *stats, sn_stream_t stream);
/* conv0 on the packed stem input (xp (N,Hp,Wp,4) fp16 from sn_pack_stem_input; w [Cout][KH][KWP*4]). */
int sn_conv_stem_fwd(const void *xp, const void *w, const float *bias, void *y, int N, int Hp, int Wp, int Ho, int Wo, int Cout,
                     int out_pix_stride, int KH, int KWP, int stride, int relu, int out_f32, sn_stream_t stream);
/* Weight gradient of the stem convolution, accumulated into the packed fp32 weight [Cout][KH][KWP*4].  The pixel range is
 * split over workgroups whose partials go to `ws` (sn_conv_stem_wgrad_workspace_bytes) and are summed in split order --
 * deterministic, no atomics; without scratch the layer runs unsplit. */
size_t sn_conv_stem_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cout, int KH, int KWP);
int sn_conv_stem_wgrad(const void *dy, const void *xp, float *dw, int N, int Hp, int Wp, int Ho, int Wo, int Cout, int dy_pix_stride,
                       int KH, int KWP, int stride, void *ws, size_t ws_bytes, sn_stream_t stream);
/* Data gradient; wt = weights as [Cin][KH*KW][Cout] fp16; `accumulate` (fp16, may alias dx) is added. */
int sn_conv_dgrad(const void *dy, const void *wt, const void *accumulate, void *dx, int N, int H, int W, int Cin,
                  int dx_pix_stride, int Cout, int dy_pix_stride, int acc_pix_stride, int KH, int KW, int stride, int pad, int dil,
                  int out_f32, sn_stream_t stream);
/* Test / A-B switch (process-wide, atomic): 1 (default) = the data gradient of a stride-2, dilation-1 convolution with even
 * destination dims walks its destination pixels parity class by parity class, each class only the taps that reach it (9 tap visits
 * instead of 36 for a 3x3 kernel, 1 instead of 4 for a 1x1 shortcut; same sums in the same order per pixel); 0 = every tap for
 * every pixel.  Affects the row-tile count sn_conv_dgrad_bn_blocks reports: set it before that query. */
int sn_conv_dgrad_by_class(int on);
/* sn_conv_dgrad that also emits the reduction of the BatchNorm(+activation) backward below it: dx is dL/dy of
 * y = act(BN(bn_x)) (bn_act: 0 none, 1 ReLU, 2 ReLU6), partials (blocks, 2, Cin) fp32 = per row tile sum g and sum g*(bn_x - mean)
 * with g = the stored dx masked by the activation; blocks = sn_conv_dgrad_bn_blocks(...) (0: the layer does not qualify).
 * Only valid when this convolution is y's only consumer (dx complete).  Consumed by sn_bn_backward_blocks. */
int sn_conv_dgrad_bn_blocks(int N, int H, int W, int Cin, int dx_pix_stride, int Cout, int dy_pix_stride, int acc_pix_stride, int KH,
                            int KW, int stride, int pad, int dil);
int sn_conv_dgrad_bn(const void *dy, const void *wt, const void *accumulate, void *dx, int N, int H, int W, int Cin, int dx_pix_stride,
                     int Cout, int dy_pix_stride, int acc_pix_stride, int KH, int KW, int stride, int pad, int dil, const void *bn_x,
                     int bn_x_pix_stride, const float *bn_scale, const float *bn_shift, const float *bn_mean, int bn_act,
                     float *partials, sn_stream_t stream);
/* Weight gradient, accumulated (+=) into dw fp32 [Cout][KH*KW][Cin].  Layers whose weight tensor is small relative to
 * the pixel count are split over K; with ws = sn_conv_wgrad_workspace_bytes(...) bytes of scratch the partials are
 * reduced without atomics (deterministic); ws may be NULL (atomic accumulation). */
size_t sn_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int x_pix_stride, int Cout, int dy_pix_stride, int KH, int KW,
                                     int stride, int pad, int dil);
int sn_conv_wgrad(const void *dy, const void *x, float *dw, int N, int H, int W, int Cin, int x_pix_stride, int Cout,
                  int dy_pix_stride, int KH, int KW, int stride, int pad, int dil, void *ws, size_t ws_bytes, sn_stream_t stream);
/* Several layers' weight gradients in ONE launch (same result as n calls of sn_conv_wgrad): a layer alone has too few output tiles
 * for 256 CUs and needs split-K partial slabs; a table of layers fills the chip with whole-K jobs.  descs: n entries of sn_wgrad_desc in host
 * memory, read during the call); scratch from sn_conv_wgrad_batch_workspace_bytes on the same table.  Replaces the weight-gradient
 * half of the fork's Convolution / FullyConnected backward (symbols/faster/resnet_mx_101_e2e.py:43-66,147-155,288-303). */
typedef struct sn_wgrad_desc {
  const void *dy, *x;
  float *dw;
  int N, H, W, Cin, x_pix_stride, Cout, dy_pix_stride, KH, KW, stride, pad, dil;
} sn_wgrad_desc;
size_t sn_conv_wgrad_batch_workspace_bytes(const sn_wgrad_desc *descs, int n);
int sn_conv_wgrad_batch(const sn_wgrad_desc *descs, int n, void *ws, size_t ws_bytes, sn_stream_t stream);
/* A/B and tuning hook: impl 1 = wave-specialised batched kernel (default), 0 = the gather kernel every layer can run on (the
 * fallback for operands that are not 16-byte addressable); job_steps = longest job in
 * 64-pixel K-steps (0 = built-in).  Set before the workspace query of the launches it affects. */
int sn_conv_wgrad_impl(int impl, int job_steps);
/* diagnostics (tools/wgrad_batch_bench.py --trace): per-job phase cycles [jobs][8] x uint64 into buf; NULL = off */
int sn_conv_wgrad_trace(void *buf);
/* diagnostics (tools/conv_trace.py): per-workgroup phase cycles [grid][8] x uint64 of the LDS-DMA forward / data-gradient
 * launches that follow into buf; NULL = off (default) */
int sn_conv_trace(void *buf);
/* bias gradient: db[c] += sum_rows dy[r][c].  Row blocks leave partial sums in `ws` (sn_bias_grad_workspace_bytes) and are added in
 * block order -- deterministic, no atomics; without scratch one block per 64 channels walks all rows. */
size_t sn_bias_grad_workspace_bytes(long rows, int C);
int sn_bias_grad(const void *dy, float *db, long rows, int C, int ld, int dtype, void *ws, size_t ws_bytes, sn_stream_t stream);

/* Stem: NCHW fp32 images -> zero padded NHWC4 fp16 with the bn_data affine folded in (:402-404). */
int sn_pack_stem_input(const float *x_nchw, void *out, int N, int C, int H, int W, int Hp, int Wp, int pad_t, int pad_l,
                       const float *scale, const float *shift, sn_stream_t stream);

/* BatchNorm (eps, momentum; :38-58).  Training statistics: sn_bn_stats writes fp32 per-row-block partials into ws
 * (sn_bn_workspace_bytes(M, C), no atomics, fixed summation order), sn_bn_finalize reduces them in double to
 * scale = gamma*invstd, shift = beta - mean*scale, the saved batch statistics and the moving averages.
 * sn_bn_backward: gradients through relu?(BN_train(x)); dgamma/dbeta are accumulated (+=); same ws size. */
size_t sn_bn_workspace_bytes(int M, int C);
int sn_bn_stats(const void *x, int M, int C, int ps, void *ws, sn_stream_t stream);
int sn_bn_finalize(const void *ws, int M, int C, float eps, float momentum, const float *gamma, const float *beta,
                   float *run_mean, float *run_var, float *scale, float *shift, float *save_mean, float *save_invstd,
                   sn_stream_t stream);
int sn_bn_finalize_blocks(const float *partials, int nblk, int M, int C, float eps, float momentum, const float *gamma,
                          const float *beta, float *run_mean, float *run_var, float *scale, float *shift, float *save_mean,
                          float *save_invstd, sn_stream_t stream);
int sn_bn_global_scale_shift(const float *gamma, const float *beta, const float *mean, const float *var, int C, float eps,
                             float *scale, float *shift, sn_stream_t stream);
int sn_bn_apply(const void *x, void *y, int M, int C, int ps_in, int ps_out, const float *scale, const float *shift, int relu,
                sn_stream_t stream); /* relu: 0 none, 1 ReLU, 2 ReLU6 = clip(0,6); same code in sn_bn_backward */
/* sn_bn_finalize_blocks + sn_bn_apply, optionally in ONE launch: with few partial rows (nblk <= 160, C % 64 == 0 -- the convolution row tiles
 * of a 20-chip stage-3 / stage-4 map) every applying workgroup owns a 64-channel slab and reduces the slab's partials itself, in
 * sn_bn_finalize_blocks' summation order (same scale / shift / saved statistics / moving averages, bit for bit); otherwise the two
 * launches.  sn_bn_backward_blocks folds its finalize into the dx pass under the same condition.
 * OPT-IN: sn_debug_option("bn_fused_finalize", 1) (SNIPER_BN_FUSED_FINALIZE); by default both entries issue the separate launches --
 * in the training step the fused form measured 1.1 ms per step SLOWER (profiles/r06_ab_bn_fused.txt). */
int sn_bn_apply_blocks(const float *partials, int nblk, const void *x, void *y, int M, int C, int ps_in, int ps_out, float eps,
                       float momentum, const float *gamma, const float *beta, float *run_mean, float *run_var, float *scale,
                       float *shift, float *save_mean, float *save_invstd, int relu, sn_stream_t stream);
int sn_bn_backward(const void *dy, const void *x, const void *accumulate, void *dx, int M, int C, int ps_dy, int ps_x, int ps_acc,
                   int ps_dx, const float *scale, const float *shift, const float *mean, const float *invstd, int relu, void *ws,
                   float *dgamma, float *dbeta, sn_stream_t stream);
int sn_bn_backward_blocks(const float *partials, int nblk, const void *dy, const void *x, const void *accumulate, void *dx, int M,
                          int C, int ps_dy, int ps_x, int ps_acc, int ps_dx, const float *scale, const float *shift, const float *mean,
                          const float *invstd, int relu, void *ws, float *dgamma, float *dbeta, sn_stream_t stream);

/* Depthwise 3x3 convolution (Convolution with num_group == channels; mobilenetv2_e2e.py:27-43,57-66), channels-last
 * fp16, weights [C][KH*KW] fp16.  dgrad adds `accumulate` (may be NULL / alias dx); wgrad accumulates (+=) into fp32
 * dw [C][KH*KW] through per-block partials in `ws` summed in block order (deterministic, no atomics). */
int sn_dwconv_fwd(const void *x, const void *w, void *y, int N, int H, int W, int C, int in_pix_stride, int out_pix_stride, int KH,
                  int KW, int stride, int pad, int dil, sn_stream_t stream);
int sn_dwconv_dgrad(const void *dy, const void *w, const void *accumulate, void *dx, int N, int H, int W, int C, int dy_pix_stride,
                    int acc_pix_stride, int dx_pix_stride, int KH, int KW, int stride, int pad, int dil, sn_stream_t stream);
size_t sn_dwconv_wgrad_workspace_bytes(int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil);
int sn_dwconv_wgrad(const void *dy, const void *x, float *dw, int N, int H, int W, int C, int dy_pix_stride, int x_pix_stride,
                    int KH, int KW, int stride, int pad, int dil, void *ws, size_t ws_bytes, sn_stream_t stream);
/* clip(lo, hi) (mx.sym.clip; relu6) forward, or backward = (lo <= ref <= hi ? a : 0) [+ accumulate]. */
int sn_clip_f16(const void *a, const void *ref, const void *accumulate, void *y, long rows, int C, int ps_a, int ps_ref, int ps_acc,
                int ps_y, float lo, float hi, int backward, sn_stream_t stream);

/* fp16 channels-last element-wise: mode 0 relu(a), 1 a+b, 2 relu-backward (ref>0 ? a : 0) [+ b]. */
int sn_ew_f16(const void *a, const void *b, const void *ref, void *y, long rows, int C, int ps_a, int ps_b, int ps_ref, int ps_y,
              int mode, sn_stream_t stream);
/* fp32 element-wise: op 0 a-b, 1 a+b, 2 a*b, 3 a*scalar, 4 fill(scalar). */
int sn_ew_f32(const float *a, const float *b, float *out, long n, int op, float scalar, sn_stream_t stream);
int sn_maxpool_fwd(const void *x, void *y, int N, int H, int W, int C, int k, int stride, int pad, sn_stream_t stream);
/* Pooling(pool_type='avg', global_pool=True) over a channels-last fp16 tensor x (N, HW, C) -> y (N, C) fp32 (the R-FCN
 * vote over the position-sensitive bins, BASELINE config C4), and its gradient dx = dy / HW (fp16, overwritten). */
int sn_avgpool_global_fwd(const void *x, float *y, int N, int HW, int C, sn_stream_t stream);
int sn_avgpool_global_bwd(const float *dy, void *dx, int N, int HW, int C, sn_stream_t stream);
/* out[b][c][r] = in[b][r][c] with dtype conversion (NHWC <-> NCHW). */
int sn_transpose_batched(const void *in, void *out, int batch, int rows, int cols, long in_batch_stride, long out_batch_stride,
                         int in_ld, int out_ld, int in_dtype, int out_dtype, sn_stream_t stream);
int sn_copy2d(const void *in, void *out, long rows, int cols, int in_ld, int out_ld, int in_dtype, int out_dtype,
              sn_stream_t stream);

/* SoftmaxOutput (:279-281,310-311): data viewed as (outer, K, inner) fp32. */
int sn_softmax_fwd(const float *x, float *p, long outer, int K, long inner, sn_stream_t stream);
int sn_softmax_output_bwd(const float *p, const float *label, float *grad, long outer, int K, long inner, float ignore_label,
                          int use_ignore, float grad_scale, int normalize_valid, int *ws, sn_stream_t stream);
/* weight * smooth_l1(pred - target) and the MakeLoss gradient (:317-319,330-334). */
int sn_smooth_l1_loss(const float *pred, const float *target, const float *weight, float *loss, float *dpred, long n, float sigma,
                      float grad_scale, sn_stream_t stream);

/* MultiProposal (:347-355) / MultiProposalTarget (:283-284).  cls_prob (B,2,A*Fh,Fw), bbox_pred (B,4A,Fh,Fw),
 * im_info (B,3), gt_boxes (B,G,5), valid_ranges (B,2), base_anchors (A,4): all fp32 device. */
size_t sn_proposal_workspace_bytes(int B, int A, int Fh, int Fw, int pre_nms_top_n, int post_nms_top_n);
int sn_multi_proposal(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *base_anchors, int B, int A,
                      int Fh, int Fw, int feat_stride, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, void *ws,
                      float *rois, float *scores, sn_stream_t stream);
int sn_multi_proposal_target(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *gt_boxes,
                             const float *valid_ranges, const float *base_anchors, int B, int A, int Fh, int Fw, int feat_stride, int G,
                             int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float fg_thresh,
                             const float *bbox_stds4, void *ws, float *rois, float *label, float *bbox_target,
                             float *bbox_weight, sn_stream_t stream);

/* Mask branch (symbols/faster/resnet_mx_101_e2e_mask.py:317-318,374-405; fork operators, semantics = DESIGN.md / oracle/nn.py).
 * sn_multi_proposal_target_mask: sn_multi_proposal_target plus, per chip, the first num_mask_rois foreground RoIs
 *   (mask_rois (B*num_mask_rois, 5), padded with [b,0,0,0,0]) and the gt_boxes row each matched (mask_ids, -1 = padding);
 *   match_ws = B * post_nms_top_n floats of scratch.
 * sn_mask_rcnn_target: the matched object's polygons (mask_polys (B, max_gts, max_len), rows as lib/data_utils/mask_utils.py
 *   :22-46 encodes them) rasterised into each RoI's mask_size^2 grid -> targets (N, ms, ms) in {1, 0, -1 = ignore}, cls (N).
 * sn_depth_to_space2 / sn_space_to_depth2: (N,H,W,4C) <-> (N,2H,2W,C) fp16, channel (a*2+b)*C + c <-> pixel (2h+a, 2w+b): the
 *   2x2 / stride-2 Deconvolution is a 1x1 convolution to 4C channels followed by this shuffle (relu fused on the way out).
 * sn_pick_fwd / sn_pick_bwd: pick(axis=1, keepdims) on a channels-last fp16 tensor (N, HW, C) with one index per n. */
int sn_multi_proposal_target_mask(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *gt_boxes,
                                  const float *valid_ranges, const float *base_anchors, int B, int A, int Fh, int Fw, int feat_stride,
                                  int G, int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float fg_thresh,
                                  const float *bbox_stds4, void *ws, float *match_ws, int num_mask_rois, float *rois, float *label,
                                  float *bbox_target, float *bbox_weight, float *mask_rois, float *mask_ids, sn_stream_t stream);
int sn_mask_rcnn_target(const float *rois, const float *mask_polys, const float *mask_ids, int N, int rois_per_image, int max_gts,
                        int max_len, int mask_size, float *targets, float *cls, sn_stream_t stream);
int sn_depth_to_space2(const void *in, void *out, int N, int H, int W, int C, int relu, sn_stream_t stream);
int sn_space_to_depth2(const void *in, void *out, int N, int H, int W, int C, sn_stream_t stream);
int sn_pick_fwd(const void *x, const float *index, void *y, int N, int HW, int C, sn_stream_t stream);
int sn_pick_bwd(const void *dy, const float *index, void *dx, int N, int HW, int C, int accumulate, sn_stream_t stream);

/* DeformablePSROIPooling, group_size 1 (:286-293).  data (B,H,W,C) fp16, rois (R,5), trans (R,2,P,P) or NULL,
 * out (R,P,P,C) fp16.  Backward: d_data (B,H,W,C) fp16 (d_data_f32 = 0) or fp32 and d_trans (R,2,P,P) fp32 are
 * OVERWRITTEN (every element written exactly once, no atomics); ws = sn_dpsroi_bwd_workspace_bytes(R). */
int sn_dpsroi_pool_fwd(const void *data, const float *rois, const float *trans, void *out, int R, int H, int W, int C, int pooled,
                       int sample_per_part, float spatial_scale, float trans_std, sn_stream_t stream);
/* ... with the number of images B of `data` (every RoI's image index in [0, B)).  OPT-IN sn_debug_option("dpsroi_slab", 1) /
 * SNIPER_DPSROI_SLAB: the (image, 64-channel slab)-stationary kernel (maps of <= 1024 cells, C % 64 == 0, <= 49 bins) -- same outputs
 * bit for bit, measured 1.6x SLOWER at R = 6000 (the bin geometry is recomputed per slab: profiles/r06_kab_dpsroi.txt). */
int sn_dpsroi_pool_fwd_images(const void *data, const float *rois, const float *trans, void *out, int R, int B, int H, int W, int C,
                              int pooled, int sample_per_part, float spatial_scale, float trans_std, sn_stream_t stream);
size_t sn_dpsroi_bwd_workspace_bytes(int R);
int sn_dpsroi_pool_bwd(const void *dout, const void *data, const float *rois, const float *trans, void *d_data, int d_data_f32,
                       float *d_trans, int R, int B, int H, int W, int C, int pooled, int sample_per_part, float spatial_scale,
                       float trans_std, void *ws, sn_stream_t stream);

/* Position-sensitive variant (group_size G > 1: the R-FCN head of BASELINE config C4; msracver Deformable R-FCN's
 * DeformablePSROIPooling(group_size=7)).  data (B,H,W,C = output_dim*G*G) fp16 in the operator's channel order
 * c = (d*G + gh)*G + gw, out (R,P,P,output_dim) fp16; bin (ph,pw) of output channel d reads channel
 * (d*G + floor(ph*G/P))*G + floor(pw*G/P).  trans (R,2,P,P) class-agnostic or NULL.  Backward as above (overwrites).
 * group_major = 1: the map (and d_data) is laid out (gh*G + gw)*output_dim + d instead -- a bin's output_dim channels are one
 * contiguous run, so a wave's gathers coalesce (the operator order pulls the whole 7.9 KB pixel of a 7*7*81 map through L2 for every
 * bin: 7.5 ms per forward at 16 x 300 RoIs; group-major 50x less).  The caller owns the permutation: the executor applies it to the
 * OUTPUT CHANNELS of the convolution that produces the map (engine/ops.py DPSROIPoolStep), which costs nothing at run time. */
int sn_psroi_pool_fwd(const void *data, const float *rois, const float *trans, void *out, int R, int H, int W, int output_dim,
                      int group_size, int pooled, int sample_per_part, float spatial_scale, float trans_std, int group_major,
                      sn_stream_t stream);
int sn_psroi_pool_bwd(const void *dout, const void *data, const float *rois, const float *trans, void *d_data, int d_data_f32,
                      float *d_trans, int R, int B, int H, int W, int output_dim, int group_size, int pooled, int sample_per_part,
                      float spatial_scale, float trans_std, int group_major, void *ws, sn_stream_t stream);

/* DeformableConvolution sampling (:124-128): column buffer (M, KH*KW, C) fp16 for the 1x1 GEMM, and its backward:
 * d_data (N,H,W,C) fp16/fp32 and d_offset (same layout/dtype as offset) are OVERWRITTEN; either may be NULL.  ws: 16 bytes of
 * device scratch (the launch's max |offset| prunes the data gradient's candidate scan) or NULL (full scan). */
int sn_deform_im2col(const void *data, const void *offset, void *col, int N, int H, int W, int C, int KH, int KW, int stride,
                     int pad, int dil, int deformable_groups, int offset_pix_stride, int offset_dtype, sn_stream_t stream);
int sn_deform_col2im(const void *dcol, const void *data, const void *offset, void *d_data, int d_data_f32, void *d_offset, int N,
                     int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                     int offset_pix_stride, int offset_dtype, void *ws, sn_stream_t stream);

/* Multi-precision SGD with momentum (lib/train_utils/utils.py:26-33). */
int sn_sgd_mom_update(float *w32, const float *grad, float *mom, void *w16, long n, float lr, float wd, float momentum,
                      float rescale, sn_stream_t stream);
/* Same, with lr/wd/momentum/rescale read from device memory d_hyper[4] (for launches captured in a hipGraph):
 * lr = d_hyper[0]*lr_mult, wd = d_hyper[1]*wd_mult. */
int sn_sgd_mom_update_dev(float *w32, const float *grad, float *mom, void *w16, long n, const float *d_hyper, float lr_mult,
                          float wd_mult, sn_stream_t stream);
/* fp32 [O][T][I] -> fp16 [I][T][O_pad] (weights for sn_conv_dgrad; O_pad = O rounded up to 8, zero filled). */
int sn_weight_transpose(const float *w_oti, void *wt_ito_f16, int O, int T, int I, int O_pad, sn_stream_t stream);
/* The same transposition for MANY weights in one launch (the optimizer step re-emits every data-gradient copy).  desc is a
 * device array of n_desc 48-byte records {const float *src; void *dst; int O, T, I, O_pad, tile0, tiles_o, tiles_i, pad}
 * with tiles_o = ceil(O_pad / 64), tiles_i = ceil(I / 64), tile0 = running sum of T * tiles_o * tiles_i (ascending);
 * total_tiles = the final sum. */
int sn_weight_transpose_batched(const void *desc, int n_desc, int total_tiles, sn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SNIPER_HIP_H */
