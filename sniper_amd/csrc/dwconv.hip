// dwconv.hip -- depthwise 3x3 convolution (Convolution with num_group == channels) and clip, channels-last fp16.
//
// MobileNetV2's inverted-residual bottlenecks (symbols/faster/mobilenetv2_e2e.py:27-86: `mobilenet_unit` with
// num_group = num_filter, and `relu6` = mx.sym.clip(0, 6)).  One multiply per loaded element: these are HBM-bound
// (forward 4 B/element: read x once -- the 9 taps hit L1/L2 -- and write y), so nothing is reshaped into a GEMM;
// a lane owns 8 consecutive channels of one pixel (16-byte accesses), a wave covers 512 channels.
// Weights are [C][KH*KW] fp16 (= the reference's (C,1,KH,KW) order); a lane's 8 channels x 9 taps are one
// contiguous 144-byte run.
#include "common.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kT = 9;  // 3x3

__device__ __forceinline__ void load_w(const half_t *__restrict__ w, int ch, float wf[8][kT]) {
  half8 run[kT];
#pragma unroll
  for (int v = 0; v < kT; ++v) run[v] = *reinterpret_cast<const half8 *>(w + (size_t)ch * kT + v * 8);
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < kT; ++t) wf[j][t] = (float)run[(j * kT + t) >> 3][(j * kT + t) & 7];
}

// y[n,oy,ox,c] = sum_tap x[n, oy*s-p+kh*d, ox*s-p+kw*d, c] * w[c][tap]
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const half_t *__restrict__ x, const half_t *__restrict__ w,
                                                         half_t *__restrict__ y, int N, int H, int W, int C, int in_ps, int out_ps,
                                                         int Ho, int Wo, int stride, int pad, int dil) {
  const int cpr = C >> 3;
  const long total = (long)N * Ho * Wo * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    long m = i / cpr;
    const int ox = (int)(m % Wo); m /= Wo;
    const int oy = (int)(m % Ho);
    const int n = (int)(m / Ho);
    float wf[8][kT];
    load_w(w, ch, wf);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int sy = oy * stride - pad + kh * dil;
      if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int sx = ox * stride - pad + kw * dil;
        if ((unsigned)sx >= (unsigned)W) continue;
        const half8 v = *reinterpret_cast<const half8 *>(x + (((size_t)n * H + sy) * W + sx) * in_ps + ch);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j] * wf[j][kh * 3 + kw];
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *reinterpret_cast<half8 *>(y + ((((size_t)n * Ho + oy) * Wo + ox)) * out_ps + ch) = o;
  }
}

// dx[n,y,x,c] = sum_tap dy[n,(y+p-kh*d)/s,(x+p-kw*d)/s,c] * w[c][tap]  (taps on the stride lattice) [+ acc]
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const half_t *__restrict__ dy, const half_t *__restrict__ w,
                                                           const half_t *__restrict__ accp, half_t *__restrict__ dx, int N, int H,
                                                           int W, int C, int dy_ps, int acc_ps, int dx_ps, int Ho, int Wo,
                                                           int stride, int pad, int dil) {
  const int cpr = C >> 3;
  const long total = (long)N * H * W * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    long m = i / cpr;
    const int xx = (int)(m % W); m /= W;
    const int yy = (int)(m % H);
    const int n = (int)(m / H);
    float wf[8][kT];
    load_w(w, ch, wf);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (accp) {
      const half8 a = *reinterpret_cast<const half8 *>(accp + (((size_t)n * H + yy) * W + xx) * acc_ps + ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = (float)a[j];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ty = yy + pad - kh * dil;
      if (ty < 0 || ty % stride != 0 || ty / stride >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tx = xx + pad - kw * dil;
        if (tx < 0 || tx % stride != 0 || tx / stride >= Wo) continue;
        const half8 v = *reinterpret_cast<const half8 *>(dy + (((size_t)n * Ho + ty / stride) * Wo + tx / stride) * dy_ps + ch);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j] * wf[j][kh * 3 + kw];
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
    *reinterpret_cast<half8 *>(dx + (((size_t)n * H + yy) * W + xx) * dx_ps + ch) = o;
  }
}

// dw[c][tap] += sum_pix dy[pix][c] * x[src(pix, tap)][c].  A thread owns 8 channels for its whole life (72 fp32
// accumulators) and walks output pixels; the block folds its row lanes through LDS in lane order and writes ONE partial
// per (channel, tap) to its slab [pixel block][C][9]; dwconv_wgrad_finish_kernel adds the slabs to dw in block order.
// No atomics anywhere: the same bits every run (the previous version folded with ds_add_f32 + global atomics).
constexpr int kDwChunks = 32;  // 8-channel chunks per block (256 channels): LDS rpp * cb * 72 * 4 B <= 72 KB
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const half_t *__restrict__ dy, const half_t *__restrict__ x,
                                                           float *__restrict__ part, int N, int H, int W, int C, int dy_ps, int x_ps,
                                                           int Ho, int Wo, int stride, int pad, int dil, int pix_per_block) {
  __shared__ float red[256 * 8 * kT];
  const int cpr = C >> 3;
  const int cb = cpr < kDwChunks ? cpr : kDwChunks, rpp = 256 / cb;
  const int rl = threadIdx.x / cb, lc = (int)(threadIdx.x - rl * cb), chunk = blockIdx.y * kDwChunks + lc;
  const bool on = rl < rpp && chunk < cpr;
  float acc[8][kT];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[j][t] = 0.f;
  if (on) {
    const int ch = chunk * 8;
    const long M = (long)N * Ho * Wo;
    const long m0 = (long)blockIdx.x * pix_per_block, m1 = m0 + pix_per_block < M ? m0 + pix_per_block : M;
    for (long m = m0 + rl; m < m1; m += rpp) {
      const int ox = (int)(m % Wo);
      const long t2 = m / Wo;
      const int oy = (int)(t2 % Ho), n = (int)(t2 / Ho);
      const half8 g = *reinterpret_cast<const half8 *>(dy + (size_t)m * dy_ps + ch);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int sy = oy * stride - pad + kh * dil;
        if ((unsigned)sy >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int sx = ox * stride - pad + kw * dil;
          if ((unsigned)sx >= (unsigned)W) continue;
          const half8 v = *reinterpret_cast<const half8 *>(x + (((size_t)n * H + sy) * W + sx) * x_ps + ch);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j][kh * 3 + kw] += (float)g[j] * (float)v[j];
        }
      }
    }
  }
  // [row lane][local chunk][8 channels][9 taps]
  if (rl < rpp) {
    float *r = red + ((size_t)rl * cb + lc) * 8 * kT;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < kT; ++t) r[j * kT + t] = acc[j][t];
  }
  __syncthreads();
  const int nloc = cpr - blockIdx.y * kDwChunks < cb ? cpr - blockIdx.y * kDwChunks : cb;   // chunks of this block that exist
  float *po = part + ((size_t)blockIdx.x * C + (size_t)blockIdx.y * kDwChunks * 8) * kT;
  for (int k = threadIdx.x; k < nloc * 8 * kT; k += 256) {
    float sum = 0.f;
    for (int q = 0; q < rpp; ++q) sum += red[(size_t)q * cb * 8 * kT + k];
    po[k] = sum;
  }
}

__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float *__restrict__ part, int nblk, long n, float *__restrict__ dw) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += part[(size_t)k * n + e];
  dw[e] += s;
}

static int dw_check(const void *a, const void *b, const void *c, int N, int H, int W, int C, int KH, int KW, int stride, int pad,
                    int dil, const char *who) {
  SN_REQUIRE(a && b && c && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "%s: bad arguments (C %% 8 == 0 required)", who);
  SN_REQUIRE(KH == 3 && KW == 3, "%s: only the 3x3 depthwise kernel is built (MobileNetV2)", who);
  SN_REQUIRE(stride > 0 && dil > 0 && pad >= 0, "%s: bad geometry", who);
  return SN_OK;
}
static int dw_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

SN_EXPORT int sn_dwconv_fwd(const void *x, const void *w, void *y, int N, int H, int W, int C, int in_pix_stride,
                            int out_pix_stride, int KH, int KW, int stride, int pad, int dil, sn_stream_t stream) {
  if (int rc = dw_check(x, w, y, N, H, W, C, KH, KW, stride, pad, dil, "sn_dwconv_fwd")) return rc;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  SN_REQUIRE(Ho > 0 && Wo > 0 && in_pix_stride % 8 == 0 && out_pix_stride % 8 == 0, "sn_dwconv_fwd: bad strides");
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(dw_blocks((long)N * Ho * Wo * (C / 8))), dim3(256), 0, sn_stream(stream),
                     (const half_t *)x, (const half_t *)w, (half_t *)y, N, H, W, C, in_pix_stride, out_pix_stride, Ho, Wo, stride,
                     pad, dil);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_dwconv_dgrad(const void *dy, const void *w, const void *accumulate, void *dx, int N, int H, int W, int C,
                              int dy_pix_stride, int acc_pix_stride, int dx_pix_stride, int KH, int KW, int stride, int pad,
                              int dil, sn_stream_t stream) {
  if (int rc = dw_check(dy, w, dx, N, H, W, C, KH, KW, stride, pad, dil, "sn_dwconv_dgrad")) return rc;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  SN_REQUIRE(Ho > 0 && Wo > 0 && dy_pix_stride % 8 == 0 && dx_pix_stride % 8 == 0, "sn_dwconv_dgrad: bad strides");
  hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3(dw_blocks((long)N * H * W * (C / 8))), dim3(256), 0, sn_stream(stream),
                     (const half_t *)dy, (const half_t *)w, (const half_t *)accumulate, (half_t *)dx, N, H, W, C, dy_pix_stride,
                     acc_pix_stride, dx_pix_stride, Ho, Wo, stride, pad, dil);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

static int dw_wgrad_blocks(long M, int C, int *pix_per_block) {
  const int cpr = C / 8, cb = cpr < kDwChunks ? cpr : kDwChunks, rpp = 256 / cb;
  long blocks = (M + (long)rpp * 16 - 1) / ((long)rpp * 16);   // >= 16 pixels per thread
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  *pix_per_block = (int)((M + blocks - 1) / blocks);
  return (int)((M + *pix_per_block - 1) / *pix_per_block);
}

SN_EXPORT size_t sn_dwconv_wgrad_workspace_bytes(int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return 0;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  int ppb;
  return sn_align(sizeof(float) * (size_t)dw_wgrad_blocks((long)N * Ho * Wo, C, &ppb) * C * kT);
}

SN_EXPORT int sn_dwconv_wgrad(const void *dy, const void *x, float *dw, int N, int H, int W, int C, int dy_pix_stride,
                              int x_pix_stride, int KH, int KW, int stride, int pad, int dil, void *ws, size_t ws_bytes,
                              sn_stream_t stream) {
  if (int rc = dw_check(dy, x, dw, N, H, W, C, KH, KW, stride, pad, dil, "sn_dwconv_wgrad")) return rc;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  SN_REQUIRE(Ho > 0 && Wo > 0 && dy_pix_stride % 8 == 0 && x_pix_stride % 8 == 0, "sn_dwconv_wgrad: bad strides");
  const long M = (long)N * Ho * Wo;
  int ppb;
  const int blocks = dw_wgrad_blocks(M, C, &ppb);
  SN_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)blocks * C * kT,
             "sn_dwconv_wgrad: needs sn_dwconv_wgrad_workspace_bytes(...) of scratch (per-block partials, summed in order)");
  hipStream_t s = sn_stream(stream);
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((unsigned)blocks, sn_div_up(C / 8, kDwChunks)), dim3(256), 0, s, (const half_t *)dy,
                     (const half_t *)x, (float *)ws, N, H, W, C, dy_pix_stride, x_pix_stride, Ho, Wo, stride, pad, dil, ppb);
  SN_CHECK_LAUNCH();
  const long n = (long)C * kT;
  hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)ws, blocks, n, dw);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// clip(lo, hi) on channels-last fp16 and its gradient (pass-through where lo <= x <= hi, the rule of
// mx.sym.clip; MobileNetV2's relu6 when it is not fused into the preceding BatchNorm).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clip_f16_kernel(const half_t *__restrict__ a, const half_t *__restrict__ ref,
                                                       const half_t *__restrict__ accp, half_t *__restrict__ y, long rows, int C,
                                                       int ps_a, int ps_ref, int ps_acc, int ps_y, float lo, float hi, int bwd) {
  const int cpr = C >> 3;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    const long r = i / cpr;
    const half8 va = *reinterpret_cast<const half8 *>(a + r * ps_a + ch);
    half8 o;
    if (!bwd) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)fminf(fmaxf((float)va[j], lo), hi);
    } else {
      const half8 vr = *reinterpret_cast<const half8 *>(ref + r * ps_ref + ch);
      half8 vb = {0, 0, 0, 0, 0, 0, 0, 0};
      if (accp) vb = *reinterpret_cast<const half8 *>(accp + r * ps_acc + ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xr = (float)vr[j];
        o[j] = (half_t)((xr >= lo && xr <= hi ? (float)va[j] : 0.f) + (float)vb[j]);
      }
    }
    *reinterpret_cast<half8 *>(y + r * ps_y + ch) = o;
  }
}

SN_EXPORT int sn_clip_f16(const void *a, const void *ref, const void *accumulate, void *y, long rows, int C, int ps_a, int ps_ref,
                          int ps_acc, int ps_y, float lo, float hi, int backward, sn_stream_t stream) {
  SN_REQUIRE(a && y && rows > 0 && C % 8 == 0 && (!backward || ref), "sn_clip_f16: bad arguments");
  hipLaunchKernelGGL(clip_f16_kernel, dim3(dw_blocks(rows * (C / 8))), dim3(256), 0, sn_stream(stream), (const half_t *)a,
                     (const half_t *)ref, (const half_t *)accumulate, (half_t *)y, rows, C, ps_a, ps_ref, ps_acc, ps_y, lo, hi,
                     backward);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
