// roi_deform.hip -- DeformablePSROIPooling and DeformableConvolution sampling kernels for gfx950,
// channels-last.  Operators of the un-vendored SNIPER-mxnet fork (call sites
// symbols/faster/resnet_mx_101_e2e.py:286-293 and :121-128); the algorithms are the published
// Deformable ConvNets v1 operators (Dai et al. 2017, msracver/Deformable-ConvNets, not pinned by the
// reference tree), restated in oracle/nn.py -- parity unpinned, spec ours where the paper is silent.
//
// Channels-last makes both operators gather-friendly: every bilinear corner is a contiguous run of
// channels, so each lane moves 16 bytes per corner and a wave covers 512 channels of one sample.
// HBM-bound: DPSROIPool forward writes R*49*C*2 bytes and reads the (L2-resident) feature map.
#include "common.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// DeformablePSROIPooling, group_size == 1 (the only setting the reference's symbols use).
//   data (B,H,W,C) fp16, rois (R,5) fp32 [batch, x1,y1,x2,y2], trans (R,2,P,P) fp32 or null
//   out (R,P,P,C) fp16          P = pooled_size = part_size, S = sample_per_part
// Bin geometry (identical for forward and backward):
//   roi_start = round(x1)*scale - 0.5, roi_end = (round(x2)+1)*scale - 0.5, size = max(end-start, 0.1)
//   bin = size / P, sub = bin / S, start = p*bin + roi_start + trans*trans_std*size
//   samples at start + i*sub, skipped if outside [-0.5, dim-0.5], clamped to [0, dim-1], bilinear;
//   output = mean over the samples taken (0 if none).
// ---------------------------------------------------------------------------------------------
struct RoiGeom {
  int b;
  float wstart, hstart, sub_w, sub_h, roi_w, roi_h;
};

__device__ __forceinline__ RoiGeom roi_geom(const float *__restrict__ rois, const float *__restrict__ trans, int r, int ph,
                                            int pw, int P, int S, float scale, float trans_std) {
  const float *q = rois + (size_t)r * 5;
  RoiGeom g;
  g.b = (int)q[0];
  const float sw = roundf(q[1]) * scale - 0.5f, sh = roundf(q[2]) * scale - 0.5f;
  const float ew = (roundf(q[3]) + 1.f) * scale - 0.5f, eh = (roundf(q[4]) + 1.f) * scale - 0.5f;
  g.roi_w = fmaxf(ew - sw, 0.1f);
  g.roi_h = fmaxf(eh - sh, 0.1f);
  const float bin_w = g.roi_w / (float)P, bin_h = g.roi_h / (float)P;
  g.sub_w = bin_w / (float)S;
  g.sub_h = bin_h / (float)S;
  float tx = 0.f, ty = 0.f;
  if (trans) {
    tx = trans[(((size_t)r * 2 + 0) * P + ph) * P + pw] * trans_std;
    ty = trans[(((size_t)r * 2 + 1) * P + ph) * P + pw] * trans_std;
  }
  g.wstart = (float)pw * bin_w + sw + tx * g.roi_w;
  g.hstart = (float)ph * bin_h + sh + ty * g.roi_h;
  return g;
}

__global__ __launch_bounds__(256) void dpsroi_fwd_kernel(const half_t *__restrict__ data, const float *__restrict__ rois,
                                                         const float *__restrict__ trans, half_t *__restrict__ out, int R, int H,
                                                         int W, int C, int P, int S, float scale, float trans_std) {
  const int cpr = C >> 3;
  const long total = (long)R * P * P * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    long t = i / cpr;
    const int pw = (int)(t % P); t /= P;
    const int ph = (int)(t % P);
    const int r = (int)(t / P);
    const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
    const half_t *img = data + (size_t)g.b * H * W * C + ch;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
    int count = 0;
    for (int ih = 0; ih < S; ++ih) {
      for (int iw = 0; iw < S; ++iw) {
        float w = g.wstart + (float)iw * g.sub_w, h = g.hstart + (float)ih * g.sub_h;
        if (w < -0.5f || w > (float)W - 0.5f || h < -0.5f || h > (float)H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)H - 1.f);
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        const half8 v00 = *reinterpret_cast<const half8 *>(img + ((size_t)y0 * W + x0) * C);
        const half8 v01 = *reinterpret_cast<const half8 *>(img + ((size_t)y0 * W + x1) * C);
        const half8 v10 = *reinterpret_cast<const half8 *>(img + ((size_t)y1 * W + x0) * C);
        const half8 v11 = *reinterpret_cast<const half8 *>(img + ((size_t)y1 * W + x1) * C);
        const float w00 = (1.f - dx) * (1.f - dy), w01 = dx * (1.f - dy), w10 = (1.f - dx) * dy, w11 = dx * dy;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          sum[j] += w00 * (float)v00[j] + w01 * (float)v01[j] + w10 * (float)v10[j] + w11 * (float)v11[j];
        ++count;
      }
    }
    const float inv = count ? 1.f / (float)count : 0.f;
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(sum[j] * inv);
    *reinterpret_cast<half8 *>(out + i * 8) = o;
  }
}

// Backward: d_data (B,H,W,C) fp32 (atomic scatter, caller zeroes), d_trans (R,2,P,P) fp32 (caller
// zeroes).  One thread per (r, ph, pw, 8-channel chunk); the trans gradient is reduced over the
// channels of a wave segment with shuffles before a single atomic per (r,ph,pw,xy).
__global__ __launch_bounds__(256) void dpsroi_bwd_kernel(const half_t *__restrict__ dout, const half_t *__restrict__ data,
                                                         const float *__restrict__ rois, const float *__restrict__ trans,
                                                         float *__restrict__ d_data, float *__restrict__ d_trans, int R, int H, int W,
                                                         int C, int P, int S, float scale, float trans_std) {
  const int cpr = C >> 3;  // chunks per (r,ph,pw); host guarantees cpr is a power of two <= 64
  const long total = (long)R * P * P * cpr;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < total;
  const long ii = active ? i : total - 1;
  const int ch = (int)(ii % cpr) * 8;
  long t = ii / cpr;
  const int pw = (int)(t % P); t /= P;
  const int ph = (int)(t % P);
  const int r = (int)(t / P);
  const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
  const size_t img_off = (size_t)g.b * H * W * C + ch;
  const half8 go = *reinterpret_cast<const half8 *>(dout + ii * 8);
  // first pass: sample count
  int count = 0;
  for (int ih = 0; ih < S; ++ih)
    for (int iw = 0; iw < S; ++iw) {
      const float w = g.wstart + (float)iw * g.sub_w, h = g.hstart + (float)ih * g.sub_h;
      if (!(w < -0.5f || w > (float)W - 0.5f || h < -0.5f || h > (float)H - 0.5f)) ++count;
    }
  float gtx = 0.f, gty = 0.f;
  if (active && count > 0) {
    const float inv = 1.f / (float)count;
    for (int ih = 0; ih < S; ++ih) {
      for (int iw = 0; iw < S; ++iw) {
        float w = g.wstart + (float)iw * g.sub_w, h = g.hstart + (float)ih * g.sub_h;
        if (w < -0.5f || w > (float)W - 0.5f || h < -0.5f || h > (float)H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)H - 1.f);
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        const float w00 = (1.f - dx) * (1.f - dy), w01 = dx * (1.f - dy), w10 = (1.f - dx) * dy, w11 = dx * dy;
        float *p00 = d_data + img_off + ((size_t)y0 * W + x0) * C, *p01 = d_data + img_off + ((size_t)y0 * W + x1) * C;
        float *p10 = d_data + img_off + ((size_t)y1 * W + x0) * C, *p11 = d_data + img_off + ((size_t)y1 * W + x1) * C;
        half8 u00, u01, u10, u11;
        if (trans) {
          u00 = *reinterpret_cast<const half8 *>(data + img_off + ((size_t)y0 * W + x0) * C);
          u01 = *reinterpret_cast<const half8 *>(data + img_off + ((size_t)y0 * W + x1) * C);
          u10 = *reinterpret_cast<const half8 *>(data + img_off + ((size_t)y1 * W + x0) * C);
          u11 = *reinterpret_cast<const half8 *>(data + img_off + ((size_t)y1 * W + x1) * C);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dv = (float)go[j] * inv;
          atomicAdd(p00 + j, w00 * dv);
          atomicAdd(p01 + j, w01 * dv);
          atomicAdd(p10 + j, w10 * dv);
          atomicAdd(p11 + j, w11 * dv);
          if (trans) {
            const float U00 = (float)u00[j], U01 = (float)u01[j], U10 = (float)u10[j], U11 = (float)u11[j];
            gtx += (U11 * dy + U01 * (1.f - dy) - U10 * dy - U00 * (1.f - dy)) * trans_std * dv * g.roi_w;
            gty += (U11 * dx + U10 * (1.f - dx) - U01 * dx - U00 * (1.f - dx)) * trans_std * dv * g.roi_h;
          }
        }
      }
    }
  }
  if (trans && d_trans) {
    // lanes [k*cpr, (k+1)*cpr) of a wave share (r,ph,pw): segmented butterfly reduce
    for (int off = cpr >> 1; off > 0; off >>= 1) {
      gtx += __shfl_xor(gtx, off, 64);
      gty += __shfl_xor(gty, off, 64);
    }
    if (active && (threadIdx.x & (cpr - 1)) == 0) {
      atomicAdd(d_trans + (((size_t)r * 2 + 0) * P + ph) * P + pw, gtx);
      atomicAdd(d_trans + (((size_t)r * 2 + 1) * P + ph) * P + pw, gty);
    }
  }
}

static long blocks_for(long total) {
  long b = (total + 255) / 256;
  return b < 1 ? 1 : (b > 16384 ? 16384 : b);
}

SN_EXPORT int sn_dpsroi_pool_fwd(const void *data, const float *rois, const float *trans, void *out, int R, int H, int W, int C,
                                 int pooled, int sample_per_part, float spatial_scale, float trans_std, sn_stream_t stream) {
  SN_REQUIRE(data && rois && out && R > 0 && C % 8 == 0 && pooled > 0 && sample_per_part > 0, "sn_dpsroi_pool_fwd: bad arguments");
  hipLaunchKernelGGL(dpsroi_fwd_kernel, dim3((unsigned)blocks_for((long)R * pooled * pooled * (C / 8))), dim3(256), 0,
                     sn_stream(stream), (const half_t *)data, rois, trans, (half_t *)out, R, H, W, C, pooled, sample_per_part,
                     spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_dpsroi_pool_bwd(const void *dout, const void *data, const float *rois, const float *trans, float *d_data,
                                 float *d_trans, int R, int H, int W, int C, int pooled, int sample_per_part,
                                 float spatial_scale, float trans_std, sn_stream_t stream) {
  SN_REQUIRE(dout && data && rois && d_data && R > 0 && C % 8 == 0, "sn_dpsroi_pool_bwd: bad arguments");
  const int cpr = C / 8;
  SN_REQUIRE(cpr <= 64 && (cpr & (cpr - 1)) == 0, "sn_dpsroi_pool_bwd: C/8 must be a power of two <= 64 (C=%d)", C);
  SN_REQUIRE(!trans || d_trans, "sn_dpsroi_pool_bwd: d_trans required with trans");
  const long total = (long)R * pooled * pooled * cpr;
  hipLaunchKernelGGL(dpsroi_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sn_stream(stream),
                     (const half_t *)dout, (const half_t *)data, rois, trans, d_data, d_trans, R, H, W, C, pooled, sample_per_part,
                     spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// DeformableConvolution sampling (DCN v1 bilinear): column buffer col (M, T, C) fp16 with
// M = N*Ho*Wo output pixels, T = KH*KW taps, from data (N,H,W,C) fp16 and offset (N,Ho,Wo,2*T*DG)
// fp32 (channel g*2T + 2*tap = dy, +1 = dx).  The contraction itself runs on the implicit-GEMM
// kernel as a 1x1 convolution over T*C "channels".
//   p = (oy*s - pad + kh*dil + dy,  ox*s - pad + kw*dil + dx);  zero unless 0 <= p < (H, W)
//   low = floor(p); if low >= dim-1: high = low = dim-1, frac = 0; else high = low+1
// ---------------------------------------------------------------------------------------------
struct DeformSample {
  bool ok;
  int y0, y1, x0, x1;
  float ly, lx;
};

__device__ __forceinline__ DeformSample deform_sample(float py, float px, int H, int W) {
  DeformSample s;
  s.ok = py >= 0.f && px >= 0.f && py < (float)H && px < (float)W;
  s.y0 = (int)floorf(py);
  s.x0 = (int)floorf(px);
  if (s.y0 >= H - 1) { s.y0 = s.y1 = H - 1; s.ly = 0.f; } else { s.y1 = s.y0 + 1; s.ly = py - (float)s.y0; }
  if (s.x0 >= W - 1) { s.x0 = s.x1 = W - 1; s.lx = 0.f; } else { s.x1 = s.x0 + 1; s.lx = px - (float)s.x0; }
  return s;
}

template <typename TO>
__global__ __launch_bounds__(256) void deform_im2col_kernel(const half_t *__restrict__ data, const TO *__restrict__ offset,
                                                            half_t *__restrict__ col, int N, int H, int W, int C, int Ho, int Wo,
                                                            int KH, int KW, int stride, int pad, int dil, int DG, int off_ps) {
  const int cpr = C >> 3, T = KH * KW, cg = C / DG;
  const long total = (long)N * Ho * Wo * T * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    long t = i / cpr;
    const int tap = (int)(t % T);
    const long m = t / T;
    const int ox = (int)(m % Wo);
    const long t2 = m / Wo;
    const int oy = (int)(t2 % Ho), n = (int)(t2 / Ho);
    const int g = ch / cg, kh = tap / KW, kw = tap - kh * KW;
    const TO *op = offset + m * off_ps + g * 2 * T + 2 * tap;
    const float py = (float)(oy * stride - pad + kh * dil) + (float)op[0], px = (float)(ox * stride - pad + kw * dil) + (float)op[1];
    const DeformSample s = deform_sample(py, px, H, W);
    half8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s.ok) {
      const half_t *img = data + (size_t)n * H * W * C + ch;
      const half8 v1 = *reinterpret_cast<const half8 *>(img + ((size_t)s.y0 * W + s.x0) * C);
      const half8 v2 = *reinterpret_cast<const half8 *>(img + ((size_t)s.y0 * W + s.x1) * C);
      const half8 v3 = *reinterpret_cast<const half8 *>(img + ((size_t)s.y1 * W + s.x0) * C);
      const half8 v4 = *reinterpret_cast<const half8 *>(img + ((size_t)s.y1 * W + s.x1) * C);
      const float w1 = (1.f - s.ly) * (1.f - s.lx), w2 = (1.f - s.ly) * s.lx, w3 = s.ly * (1.f - s.lx), w4 = s.ly * s.lx;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        o[j] = (half_t)(w1 * (float)v1[j] + w2 * (float)v2[j] + w3 * (float)v3[j] + w4 * (float)v4[j]);
    }
    *reinterpret_cast<half8 *>(col + i * 8) = o;
  }
}

// Backward of the sampling: from dcol (M,T,C) fp16 produce
//   d_data   (N,H,W,C) fp32, atomic scatter (caller zeroes)
//   d_offset (N,Ho,Wo,2*T*DG) fp32: sum over the group's channels of dcol * d(sample)/d(offset);
//            each (m, tap, group) is owned by `cg/8` consecutive lanes, reduced with shuffles.
template <typename TO>
__global__ __launch_bounds__(256) void deform_col2im_kernel(const half_t *__restrict__ dcol, const half_t *__restrict__ data,
                                                            const TO *__restrict__ offset, float *__restrict__ d_data,
                                                            TO *__restrict__ d_offset, int N, int H, int W, int C, int Ho, int Wo,
                                                            int KH, int KW, int stride, int pad, int dil, int DG, int off_ps) {
  const int cpr = C >> 3, T = KH * KW, cg = C / DG, lpg = cg >> 3;  // lanes per group (power of two <= 64)
  const long total = (long)N * Ho * Wo * T * cpr;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < total;
  const long ii = active ? i : total - 1;
  const int ch = (int)(ii % cpr) * 8;
  long t = ii / cpr;
  const int tap = (int)(t % T);
  const long m = t / T;
  const int ox = (int)(m % Wo);
  const long t2 = m / Wo;
  const int oy = (int)(t2 % Ho), n = (int)(t2 / Ho);
  const int g = ch / cg, kh = tap / KW, kw = tap - kh * KW;
  const TO *op = offset + m * off_ps + g * 2 * T + 2 * tap;
  const float py = (float)(oy * stride - pad + kh * dil) + (float)op[0], px = (float)(ox * stride - pad + kw * dil) + (float)op[1];
  const DeformSample s = deform_sample(py, px, H, W);
  float gy = 0.f, gx = 0.f;
  if (active && s.ok) {
    const half8 go = *reinterpret_cast<const half8 *>(dcol + ii * 8);
    const size_t base = (size_t)n * H * W * C + ch;
    const size_t o1 = base + ((size_t)s.y0 * W + s.x0) * C, o2 = base + ((size_t)s.y0 * W + s.x1) * C;
    const size_t o3 = base + ((size_t)s.y1 * W + s.x0) * C, o4 = base + ((size_t)s.y1 * W + s.x1) * C;
    const half8 v1 = *reinterpret_cast<const half8 *>(data + o1), v2 = *reinterpret_cast<const half8 *>(data + o2);
    const half8 v3 = *reinterpret_cast<const half8 *>(data + o3), v4 = *reinterpret_cast<const half8 *>(data + o4);
    const float w1 = (1.f - s.ly) * (1.f - s.lx), w2 = (1.f - s.ly) * s.lx, w3 = s.ly * (1.f - s.lx), w4 = s.ly * s.lx;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (float)go[j];
      atomicAdd(d_data + o1 + j, w1 * d);
      atomicAdd(d_data + o2 + j, w2 * d);
      atomicAdd(d_data + o3 + j, w3 * d);
      atomicAdd(d_data + o4 + j, w4 * d);
      const float a = (float)v1[j], b = (float)v2[j], c = (float)v3[j], e = (float)v4[j];
      gy += d * ((1.f - s.lx) * (c - a) + s.lx * (e - b));
      gx += d * ((1.f - s.ly) * (b - a) + s.ly * (e - c));
    }
  }
  for (int off = lpg >> 1; off > 0; off >>= 1) {
    gy += __shfl_xor(gy, off, 64);
    gx += __shfl_xor(gx, off, 64);
  }
  if (active && ((ch >> 3) & (lpg - 1)) == 0) {
    TO *dp = d_offset + m * off_ps + g * 2 * T + 2 * tap;
    dp[0] = (TO)gy;
    dp[1] = (TO)gx;
  }
}

SN_EXPORT int sn_deform_im2col(const void *data, const void *offset, void *col, int N, int H, int W, int C, int KH, int KW,
                               int stride, int pad, int dil, int deformable_groups, int offset_pix_stride, int offset_dtype,
                               sn_stream_t stream) {
  SN_REQUIRE(data && offset && col && C % 8 == 0 && deformable_groups > 0 && (C / deformable_groups) % 8 == 0,
             "sn_deform_im2col: bad arguments");
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long total = (long)N * Ho * Wo * KH * KW * (C / 8);
  if (offset_dtype == 0)
    hipLaunchKernelGGL((deform_im2col_kernel<half_t>), dim3((unsigned)blocks_for(total)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)data, (const half_t *)offset, (half_t *)col, N, H, W, C, Ho, Wo, KH, KW, stride, pad, dil,
                       deformable_groups, offset_pix_stride);
  else
    hipLaunchKernelGGL((deform_im2col_kernel<float>), dim3((unsigned)blocks_for(total)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)data, (const float *)offset, (half_t *)col, N, H, W, C, Ho, Wo, KH, KW, stride, pad, dil,
                       deformable_groups, offset_pix_stride);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_deform_col2im(const void *dcol, const void *data, const void *offset, float *d_data, void *d_offset, int N,
                               int H, int W, int C, int KH, int KW, int stride, int pad, int dil, int deformable_groups,
                               int offset_pix_stride, int offset_dtype, sn_stream_t stream) {
  SN_REQUIRE(dcol && data && offset && d_data && d_offset && C % 8 == 0 && deformable_groups > 0, "sn_deform_col2im: bad arguments");
  const int lpg = C / deformable_groups / 8;
  SN_REQUIRE(lpg >= 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0 && (C / deformable_groups) % 8 == 0,
             "sn_deform_col2im: channels per deformable group / 8 must be a power of two <= 64");
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long total = (long)N * Ho * Wo * KH * KW * (C / 8);
  if (offset_dtype == 0)
    hipLaunchKernelGGL((deform_col2im_kernel<half_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)dcol, (const half_t *)data, (const half_t *)offset, d_data, (half_t *)d_offset, N, H, W, C, Ho,
                       Wo, KH, KW, stride, pad, dil, deformable_groups, offset_pix_stride);
  else
    hipLaunchKernelGGL((deform_col2im_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)dcol, (const half_t *)data, (const float *)offset, d_data, (float *)d_offset, N, H, W, C, Ho,
                       Wo, KH, KW, stride, pad, dil, deformable_groups, offset_pix_stride);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
