// mask.hip -- device side of the auxiliary mask branch (symbols/faster/resnet_mx_101_e2e_mask.py:374-405): the operators
// live in the un-vendored SNIPER-mxnet fork, so their semantics are this repository's documented choice (DESIGN.md,
// oracle/nn.py) -- parity unpinned.
//   MaskRcnnTarget   polygon of the matched GT rasterised into the RoI's mask_size x mask_size grid
//   Deconvolution    2x2 stride-2 up-sampling = a 1x1 convolution to 4*C channels (conv.hip) + depth-to-space here
//   pick             one channel per RoI (its class's negative / positive mask map) and the scatter back
#include "common.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// MaskRcnnTarget.  rois (N,5) [b,x1,y1,x2,y2] in chip pixels, N = B * rois_per_image; mask_polys (B, max_gts, max_len):
// row = [category, n_seg, len_1..len_n, seg_1 coords.., seg_n coords.., -1 pad] (lib/data_utils/mask_utils.py:22-46);
// mask_ids (N) = gt row or -1.  targets (N, ms, ms): 1 / 0 = the centre of RoI cell (i, j),
//   (x1 + (j + 0.5) * (x2 - x1 + 1) / ms,  y1 + (i + 0.5) * (y2 - y1 + 1) / ms),
// lies inside / outside the union of the object's polygons (even-odd rule per polygon, crossing test in float32);
// -1 (ignored by the loss) for padding RoIs and for objects none of whose polygons fitted the row.  cls (N) = category.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_rcnn_target_kernel(const float *__restrict__ rois, const float *__restrict__ polys,
                                                               const float *__restrict__ ids, int rois_per_image, int max_gts,
                                                               int max_len, int ms, float *__restrict__ targets,
                                                               float *__restrict__ cls) {
  extern __shared__ __attribute__((aligned(16))) float row[];   // [max_len]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int b = n / rois_per_image;
  const int gid = (int)ids[n];
  float *out = targets + (size_t)n * ms * ms;
  if (gid < 0 || gid >= max_gts) {
    for (int p = tid; p < ms * ms; p += blockDim.x) out[p] = -1.f;
    if (tid == 0) cls[n] = 0.f;
    return;
  }
  const float *src = polys + ((size_t)b * max_gts + gid) * max_len;
  for (int k = tid; k < max_len; k += blockDim.x) row[k] = src[k];
  __syncthreads();
  const int nseg = (int)row[1];
  if (tid == 0) cls[n] = row[0] < 0.f ? 0.f : row[0];
  if (row[0] < 0.f || nseg <= 0) {
    for (int p = tid; p < ms * ms; p += blockDim.x) out[p] = -1.f;
    return;
  }
  const float *r = rois + (size_t)n * 5;
  const float x1 = r[1], y1 = r[2];
  const float cw = (r[3] - r[1] + 1.f) / (float)ms, ch = (r[4] - r[2] + 1.f) / (float)ms;
  for (int p = tid; p < ms * ms; p += blockDim.x) {
    const int i = p / ms, j = p - i * ms;
    const float px = x1 + ((float)j + 0.5f) * cw, py = y1 + ((float)i + 0.5f) * ch;
    bool inside = false;
    int off = 2 + nseg;
    for (int sgm = 0; sgm < nseg; ++sgm) {
      const int len = (int)row[2 + sgm], nv = len >> 1;
      bool in = false;
      for (int a = 0, c = nv - 1; a < nv; c = a++) {
        const float xa = row[off + 2 * a], ya = row[off + 2 * a + 1], xc = row[off + 2 * c], yc = row[off + 2 * c + 1];
        if (((ya > py) != (yc > py)) && (px < (xc - xa) * (py - ya) / (yc - ya) + xa)) in = !in;
      }
      inside = inside || in;
      off += len;
    }
    out[p] = inside ? 1.f : 0.f;
  }
}

SN_EXPORT int sn_mask_rcnn_target(const float *rois, const float *mask_polys, const float *mask_ids, int N, int rois_per_image,
                                  int max_gts, int max_len, int mask_size, float *targets, float *cls, sn_stream_t stream) {
  SN_REQUIRE(rois && mask_polys && mask_ids && targets && cls && N > 0 && rois_per_image > 0 && max_gts > 0 && max_len > 2 &&
                 max_len <= 8192 && mask_size > 0, "sn_mask_rcnn_target: bad arguments");
  hipLaunchKernelGGL(mask_rcnn_target_kernel, dim3(N), dim3(256), (size_t)max_len * sizeof(float), sn_stream(stream), rois, mask_polys,
                     mask_ids, rois_per_image, max_gts, max_len, mask_size, targets, cls);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// depth <-> space for the 2x2 / stride-2 deconvolution: in (N,H,W,4*C) with channel (a*2+b)*C + c  <->  out (N,2H,2W,C),
// out[n, 2h+a, 2w+b, c] = in[n, h, w, (a*2+b)*C + c].  C % 8 == 0: 16-byte moves.  relu on the way out (forward only).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_space2_kernel(const half_t *__restrict__ in, half_t *__restrict__ out, long total, int H,
                                                           int W, int C, int to_space, int relu) {
  const int cpr = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i enumerates the (N, 2H, 2W, C/8) chunks of the full-resolution tensor
    const int ck = (int)(i % cpr);
    long t = i / cpr;
    const int X = (int)(t % (2 * W)); t /= 2 * W;
    const int Y = (int)(t % (2 * H));
    const long n = t / (2 * H);
    const int h = Y >> 1, a = Y & 1, w = X >> 1, b = X & 1;
    const size_t deep = ((((size_t)n * H + h) * W + w) * 4 + (a * 2 + b)) * C + ck * 8;
    const size_t flat = (size_t)i * 8;
    if (to_space) {
      half8 v = *reinterpret_cast<const half8 *>(in + deep);
      if (relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] > (half_t)0 ? v[j] : (half_t)0;
      }
      *reinterpret_cast<half8 *>(out + flat) = v;
    } else {
      *reinterpret_cast<half8 *>(out + deep) = *reinterpret_cast<const half8 *>(in + flat);
    }
  }
}

SN_EXPORT int sn_depth_to_space2(const void *in, void *out, int N, int H, int W, int C, int relu, sn_stream_t stream) {
  SN_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "sn_depth_to_space2: bad arguments (C %% 8 == 0)");
  const long total = (long)N * 2 * H * 2 * W * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(depth_space2_kernel, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), (const half_t *)in, (half_t *)out,
                     total, H, W, C, 1, relu);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_space_to_depth2(const void *in, void *out, int N, int H, int W, int C, sn_stream_t stream) {
  SN_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "sn_space_to_depth2: bad arguments (C %% 8 == 0)");
  const long total = (long)N * 2 * H * 2 * W * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(depth_space2_kernel, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), (const half_t *)in, (half_t *)out,
                     total, H, W, C, 0, 0);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// pick(data, index, axis=1, keepdims=True) on a channels-last tensor: y[n, p] = x[n, p, index[n]]  (x (N, HW, C) fp16), and
// its gradient: dx[n, p, c] = (c == index[n]) ? dy[n, p] : 0, overwritten (accumulate = 0) or added at the picked channel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pick_fwd_kernel(const half_t *__restrict__ x, const float *__restrict__ index,
                                                       half_t *__restrict__ y, long total, int HW, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long n = i / HW;
  int c = (int)index[n];
  c = c < 0 ? 0 : (c >= C ? C - 1 : c);           // mx.nd.pick's default mode 'clip'
  y[i] = x[(size_t)i * C + c];
}

__global__ __launch_bounds__(256) void pick_bwd_kernel(const half_t *__restrict__ dy, const float *__restrict__ index,
                                                       half_t *__restrict__ dx, long total, int HW, int C, int accumulate) {
  const int cpr = C >> 3;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one 8-channel chunk of one (n, p)
  if (i >= total) return;
  const long np_ = i / cpr;
  const int ck = (int)(i - np_ * cpr);
  int c = (int)index[np_ / HW];
  c = c < 0 ? 0 : (c >= C ? C - 1 : c);
  half_t *dst = dx + (size_t)np_ * C + ck * 8;
  if (accumulate) {
    if ((c >> 3) == ck) dst[c & 7] = (half_t)((float)dst[c & 7] + (float)dy[np_]);
  } else {
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((c >> 3) == ck) v[c & 7] = dy[np_];
    *reinterpret_cast<half8 *>(dst) = v;
  }
}

SN_EXPORT int sn_pick_fwd(const void *x, const float *index, void *y, int N, int HW, int C, sn_stream_t stream) {
  SN_REQUIRE(x && index && y && N > 0 && HW > 0 && C > 0, "sn_pick_fwd: bad arguments");
  const long total = (long)N * HW;
  hipLaunchKernelGGL(pick_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sn_stream(stream), (const half_t *)x, index,
                     (half_t *)y, total, HW, C);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_pick_bwd(const void *dy, const float *index, void *dx, int N, int HW, int C, int accumulate, sn_stream_t stream) {
  SN_REQUIRE(dy && index && dx && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "sn_pick_bwd: bad arguments (C %% 8 == 0)");
  const long total = (long)N * HW * (C / 8);
  hipLaunchKernelGGL(pick_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sn_stream(stream), (const half_t *)dy,
                     index, (half_t *)dx, total, HW, C, accumulate);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
