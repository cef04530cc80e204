// nn_ops.hip -- the HBM-bound graph operators of the SNIPER training step on gfx950:
// stem input packing, BatchNorm (batch-stat and global-stat) fused with ReLU, max pooling,
// SoftmaxOutput, smooth-L1, element-wise glue, layout conversion, multi-precision SGD.
//
// They replace operators of the un-vendored SNIPER-mxnet fork; call sites are cited per kernel
// (symbols/faster/resnet_mx_101_e2e.py).  All activations are channels-last fp16 with an explicit
// pixel stride so that producers can write into slices of a wider buffer (free Concat).
// Roofline: every kernel here is bound by HBM bytes; all global accesses are 16 bytes per lane.
#include "common.h"
#include <algorithm>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------
// Stem input: NCHW fp32 image batch -> zero-padded NHWC4 fp16, with the bn_data affine folded in
// (resnet_mx_101_e2e.py:402, use_global_stats, fix_gamma).  The 7x7/2 conv0 then runs on the
// generic implicit-GEMM kernel as a (7 taps) x (8 pixels x 4 channels) contraction: one kernel row
// of conv0 is one contiguous 64-byte run of this buffer.
// out[n][y][x][c], y in [0,Hp), x in [0,Wp); source pixel (y-pad_t, x-pad_l); channel 3 = 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_stem_kernel(const float *__restrict__ x, half_t *__restrict__ out, int N, int C,
                                                        int H, int W, int Hp, int Wp, int pad_t, int pad_l,
                                                        const float *__restrict__ scale, const float *__restrict__ shift) {
  const long total = (long)N * Hp * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    const long t = i / Wp;
    const int yp = (int)(t % Hp), n = (int)(t / Hp);
    const int sy = yp - pad_t, sx = xp - pad_l;
    half_t v[4] = {0, 0, 0, 0};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
      for (int c = 0; c < C && c < 4; ++c) {
        float f = x[(((long)n * C + c) * H + sy) * W + sx];
        if (scale) f = f * scale[c] + shift[c];
        v[c] = (half_t)f;
      }
    }
    *reinterpret_cast<uint2 *>(out + i * 4) = *reinterpret_cast<uint2 *>(v);
  }
}

SN_EXPORT int sn_pack_stem_input(const float *x_nchw, void *out, int N, int C, int H, int W, int Hp, int Wp, int pad_t,
                                 int pad_l, const float *scale, const float *shift, sn_stream_t stream) {
  SN_REQUIRE(x_nchw && out && N > 0 && C >= 1 && C <= 4 && Hp >= H + pad_t && Wp >= W + pad_l,
             "sn_pack_stem_input: bad arguments");
  const long total = (long)N * Hp * Wp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_stem_kernel, dim3(blocks), dim3(256), 0, sn_stream(stream), x_nchw, (half_t *)out, N, C, H, W,
                     Hp, Wp, pad_t, pad_l, scale, shift);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm.  x is (M pixels, C channels) fp16 with pixel stride ps, C % 8 == 0.
// Row-walking kernels (stats, backward reduce, backward dx) share one mapping: a block covers up to 256 8-channel
// chunks (blockIdx.y selects the slab when C > 2048); thread t owns chunk t % CB for its whole life -- per-channel
// coefficients live in registers -- and walks rows t / CB, + rpp, ... of the block's row range with 4 independent
// 16-byte loads in flight (HBM-bound: bytes in flight per CU decide the rate, guide section "memory").
//   bn_stats    : per-channel sum and sum of squares -> fp64 accumulators (one atomic per block and channel)
//   bn_finalize : mean/var -> scale = gamma*invstd, shift = beta - mean*scale; running stats
//   bn_apply    : y = act(x*scale + shift), act = none / relu / relu6 (`relu` argument 0 / 1 / 2)
//   bn_bwd_reduce / bn_bwd_dx : gradients through (ReLU o BN) in training mode
// BatchNorm(fix_gamma=False, eps=2e-5, momentum) call sites: resnet_mx_101_e2e.py:38-58.
// ---------------------------------------------------------------------------------------------
constexpr int kBnThreads = 256;
constexpr int kBnUnroll = 4;

// gradient mask of the fused activation: act 0 none, 1 relu (y > 0), 2 relu6 = clip(0,6) (0 <= y <= 6, mx.sym.clip's rule)
__device__ __forceinline__ bool bn_act_pass(float y, int act) {
  return act == 0 || (act == 1 ? y > 0.f : (y >= 0.f && y <= 6.f));
}

// The per-channel coefficients and the per-element forms, spelled with explicit roundings: the same values come out of the separate
// finalize kernels and out of the consumers that fold the finalize into themselves (bn_apply_fin_kernel / bn_bwd_dx_fin_kernel),
// whatever the compiler would contract in either context.
__device__ __forceinline__ void bn_fwd_coefs(double mean, double var, float eps, float g, float beta, float &sc, float &sh, float &invstd) {
  invstd = (float)(1.0 / sqrt(var + (double)eps));
  sc = __fmul_rn(g, invstd);
  sh = __fsub_rn(beta, __fmul_rn(__fmul_rn((float)mean, g), invstd));
}
__device__ __forceinline__ float bn_running(float run, float stat, float momentum) {
  return __fmaf_rn(run, momentum, __fmul_rn(stat, 1.f - momentum));
}
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return __fmaf_rn(x, sc, sh); }
// dx = scale * (g - dbeta/M - xhat*dgamma/M) [+ acc] = sc*g + kb*x + kd
__device__ __forceinline__ void bn_dx_coefs(float sc, float invstd, float mean, float dgamma, float dbeta, float invM, float &kb, float &kd) {
  const float t = __fmul_rn(__fmul_rn(invstd, dgamma), invM);   // xhat coefficient / scale
  kb = -__fmul_rn(sc, t);
  kd = __fmul_rn(sc, __fmaf_rn(mean, t, -__fmul_rn(dbeta, invM)));
}
__device__ __forceinline__ float bn_dx_value(float sc, float gf, float kb, float xf, float kd, float a) {
  return __fadd_rn(__fmaf_rn(sc, gf, __fmaf_rn(kb, xf, kd)), a);
}

struct BnMap {
  int cb, rpp, chunk, rl;
  bool on;
};
__device__ __forceinline__ BnMap bn_map(int C) {
  const int cpr = C >> 3;
  BnMap m;
  m.cb = cpr < kBnThreads ? cpr : kBnThreads;
  m.rpp = kBnThreads / m.cb;
  m.rl = threadIdx.x / m.cb;
  m.chunk = blockIdx.y * kBnThreads + (int)(threadIdx.x - m.rl * m.cb);
  m.on = m.rl < m.rpp && m.chunk < cpr;
  return m;
}

// block-level reduction of 16 per-thread partials over the row lanes, then `emit(channel, j, value)`
template <typename F>
__device__ __forceinline__ void bn_block_reduce(const BnMap &m, int C, const float s[8], const float q[8], F emit) {
  __shared__ float red[kBnThreads][17];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[threadIdx.x][j] = m.on ? s[j] : 0.f;
    red[threadIdx.x][8 + j] = m.on ? q[j] : 0.f;
  }
  __syncthreads();
  const int cpr = C >> 3;
  for (int w = threadIdx.x; w < m.cb * 16; w += kBnThreads) {
    const int ch = w >> 4, j = w & 15;
    if (blockIdx.y * kBnThreads + ch >= cpr) continue;
    float a = 0.f;
    for (int k = 0; k < m.rpp; ++k) a += red[k * m.cb + ch][j];
    emit((blockIdx.y * kBnThreads + ch) * 8 + (j & 7), j, a);
  }
}

__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const half_t *__restrict__ x, int M, int C, int ps,
                                                              int rows_per_block, float *__restrict__ part) {
  const BnMap m = bn_map(C);
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (m.on) {
    const half_t *px = x + m.chunk * 8;
    int r = r0 + m.rl;
    for (; r + (kBnUnroll - 1) * m.rpp < r1; r += kBnUnroll * m.rpp) {
      half8 v[kBnUnroll];
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) v[u] = *reinterpret_cast<const half8 *>(px + (size_t)(r + u * m.rpp) * ps);
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[u][j];
          s[j] += f;
          q[j] += f * f;
        }
    }
    for (; r < r1; r += m.rpp) {
      const half8 v = *reinterpret_cast<const half8 *>(px + (size_t)r * ps);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        s[j] += f;
        q[j] += f * f;
      }
    }
  }
  float *po = part + (size_t)blockIdx.x * 2 * C;  // [row block][sum | sumsq][C]
  bn_block_reduce(m, C, s, q, [&](int c, int j, float a) { po[(j < 8 ? 0 : C) + c] = a; });
}

// sums the per-row-block partials [nblk][2][C] in double: 32 channels x 32 partial lanes per 1024-thread block
// (nblk <= 512: at most 16 independent loads per thread and statistic -- the chain is latency, not bandwidth)
constexpr int kBnFinThreads = 1024;
__device__ __forceinline__ void bn_sum_partials(const float *__restrict__ part, int nblk, int C, double &a, double &b, int &c) {
  __shared__ double red[2][32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  c = blockIdx.x * 32 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int k = rl; k < nblk; k += 32) {
      s0 += (double)part[(size_t)k * 2 * C + c];
      s1 += (double)part[(size_t)k * 2 * C + C + c];
    }
  }
  red[0][rl][cl] = s0;
  red[1][rl][cl] = s1;
  __syncthreads();
  a = b = 0.0;
  if (rl == 0)
    for (int k = 0; k < 32; ++k) {
      a += red[0][k][cl];
      b += red[1][k][cl];
    }
}

__global__ __launch_bounds__(kBnFinThreads) void bn_finalize_kernel(const float *__restrict__ part, int nblk, int M, int C, float eps,
                                                          float momentum, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ run_mean,
                                                          float *__restrict__ run_var, float *__restrict__ scale,
                                                          float *__restrict__ shift, float *__restrict__ save_mean,
                                                          float *__restrict__ save_invstd) {
  // the per-channel inputs are requested BEFORE the partial sums: behind the reduction's barrier they would be a second dependent
  // round trip to memory in a launch that is nothing but latency (180 of these per R101 step)
  const int c0 = blockIdx.x * 32 + (threadIdx.x & 31);
  const bool head = threadIdx.x < 32 && c0 < C;
  float g = 1.f, bt = 0.f, rm = 0.f, rv = 0.f;
  if (head) {
    if (gamma) g = gamma[c0];
    bt = beta[c0];
    if (run_mean) { rm = run_mean[c0]; rv = run_var[c0]; }
  }
  double sum, sumsq;
  int c;
  bn_sum_partials(part, nblk, C, sum, sumsq, c);
  if (!head) return;
  const double mean = sum / M;
  double var = sumsq / M - mean * mean;  // biased
  if (var < 0) var = 0;
  float sc, sh, invstd;
  bn_fwd_coefs(mean, var, eps, g, bt, sc, sh, invstd);
  scale[c] = sc;
  shift[c] = sh;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (run_mean) {
    run_mean[c] = bn_running(rm, (float)mean, momentum);
    run_var[c] = bn_running(rv, (float)var, momentum);
  }
}

// scale/shift from running statistics (use_global_stats=True)
__global__ void bn_global_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                                 const float *__restrict__ mean, const float *__restrict__ var, int C, float eps,
                                 float *__restrict__ scale, float *__restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f;
  scale[c] = g * invstd;
  shift[c] = beta[c] - mean[c] * g * invstd;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const half_t *__restrict__ x, half_t *__restrict__ y, int M, int C,
                                                       int ps_in, int ps_out, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, int relu) {
  const int cpr = C >> 3;
  const long total = (long)M * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int chunk = (int)(i % cpr);
    const long r = i / cpr;
    const half8 v = *reinterpret_cast<const half8 *>(x + r * ps_in + chunk * 8);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = bn_affine((float)v[j], scale[chunk * 8 + j], shift[chunk * 8 + j]);
      if (relu) f = f > 0.f ? f : 0.f;
      if (relu == 2) f = f < 6.f ? f : 6.f;   // relu6 = clip(0, 6) (mobilenetv2_e2e.py:18-19)
      o[j] = (half_t)f;
    }
    *reinterpret_cast<half8 *>(y + r * ps_out + chunk * 8) = o;
  }
}

// dbeta[c] = sum g, dgamma[c] = sum g * xhat, g = dy * (relu ? (x*scale+shift > 0) : 1)
__global__ __launch_bounds__(kBnThreads) void bn_bwd_reduce_kernel(const half_t *__restrict__ dy, const half_t *__restrict__ x,
                                                                   int M, int C, int ps_dy, int ps_x, int rows_per_block,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift,
                                                                   const float *__restrict__ mean,
                                                                   int relu, float *__restrict__ part) {
  const BnMap m = bn_map(C);
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  if (m.on) {
    float sc[8], sh[8], mu[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = m.chunk * 8 + j;
      sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c];
    }
    const half_t *pg = dy + m.chunk * 8, *px = x + m.chunk * 8;
    auto body = [&](const half8 &g, const half8 &v) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xf = (float)v[j];
        float gf = (float)g[j];
        if (!bn_act_pass(bn_affine(xf, sc[j], sh[j]), relu)) gf = 0.f;
        s[j] += gf;
        q[j] += gf * (xf - mu[j]);
      }
    };
    int r = r0 + m.rl;
    for (; r + (kBnUnroll - 1) * m.rpp < r1; r += kBnUnroll * m.rpp) {
      half8 g[kBnUnroll], v[kBnUnroll];
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) {
        g[u] = *reinterpret_cast<const half8 *>(pg + (size_t)(r + u * m.rpp) * ps_dy);
        v[u] = *reinterpret_cast<const half8 *>(px + (size_t)(r + u * m.rpp) * ps_x);
      }
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) body(g[u], v[u]);
    }
    for (; r < r1; r += m.rpp)
      body(*reinterpret_cast<const half8 *>(pg + (size_t)r * ps_dy), *reinterpret_cast<const half8 *>(px + (size_t)r * ps_x));
  }
  float *po = part + (size_t)blockIdx.x * 2 * C;  // [row block][sum g | sum g*(x-mean)][C]
  bn_block_reduce(m, C, s, q, [&](int c, int j, float a) { po[(j < 8 ? 0 : C) + c] = a; });
}

// partials -> fin[0..C) = dbeta, fin[C..2C) = dgamma (fp32, read by the dx kernel) and += into the gradient arena
__global__ __launch_bounds__(kBnFinThreads) void bn_bwd_finalize_kernel(const float *__restrict__ part, int nblk, int C,
                                                              const float *__restrict__ invstd, float *__restrict__ fin,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta) {
  const int c0 = blockIdx.x * 32 + (threadIdx.x & 31);      // (inputs first: see bn_finalize_kernel)
  const bool head = threadIdx.x < 32 && c0 < C;
  float is = 0.f, db0 = 0.f, dg0 = 0.f;
  if (head) {
    is = invstd[c0];
    if (dbeta) db0 = dbeta[c0];
    if (dgamma) dg0 = dgamma[c0];
  }
  double sg, sgx;
  int c;
  bn_sum_partials(part, nblk, C, sg, sgx, c);
  if (!head) return;
  const float db = (float)sg, dg = (float)(sgx * (double)is);
  fin[c] = db;
  fin[C + c] = dg;
  if (dbeta) dbeta[c] = db0 + db;
  if (dgamma) dgamma[c] = dg0 + dg;
}

// dx = scale * (g - dbeta/M - xhat*dgamma/M) [+ acc] = ka*g + kb*x + kd with three per-channel registers
__global__ __launch_bounds__(kBnThreads) void bn_bwd_dx_kernel(const half_t *__restrict__ dy, const half_t *__restrict__ x,
                                                               const half_t *__restrict__ acc, half_t *__restrict__ dx, int M,
                                                               int C, int ps_dy, int ps_x, int ps_acc, int ps_dx,
                                                               int rows_per_block, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               const float *__restrict__ dgamma,
                                                               const float *__restrict__ dbeta, int relu) {
  const BnMap m = bn_map(C);
  if (!m.on) return;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  const float invM = 1.f / (float)M;
  float sc[8], sh[8], kb[8], kd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = m.chunk * 8 + j;
    sc[j] = scale[c]; sh[j] = shift[c];
    bn_dx_coefs(sc[j], invstd[c], mean[c], dgamma[c], dbeta[c], invM, kb[j], kd[j]);
  }
  const half_t *pg = dy + m.chunk * 8, *px = x + m.chunk * 8, *pa = acc ? acc + m.chunk * 8 : nullptr;
  half_t *po = dx + m.chunk * 8;
  auto body = [&](const half8 &g, const half8 &v, const half8 &a, size_t r) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = (float)v[j];
      float gf = (float)g[j];
      if (!bn_act_pass(bn_affine(xf, sc[j], sh[j]), relu)) gf = 0.f;
      o[j] = (half_t)bn_dx_value(sc[j], gf, kb[j], xf, kd[j], (float)a[j]);
    }
    *reinterpret_cast<half8 *>(po + r * ps_dx) = o;
  };
  const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  int r = r0 + m.rl;
  for (; r + (kBnUnroll - 1) * m.rpp < r1; r += kBnUnroll * m.rpp) {
    half8 g[kBnUnroll], v[kBnUnroll], a[kBnUnroll];
#pragma unroll
    for (int u = 0; u < kBnUnroll; ++u) {
      const size_t rr = (size_t)(r + u * m.rpp);
      g[u] = *reinterpret_cast<const half8 *>(pg + rr * ps_dy);
      v[u] = *reinterpret_cast<const half8 *>(px + rr * ps_x);
      a[u] = pa ? *reinterpret_cast<const half8 *>(pa + rr * ps_acc) : zero;
    }
#pragma unroll
    for (int u = 0; u < kBnUnroll; ++u) body(g[u], v[u], a[u], (size_t)(r + u * m.rpp));
  }
  for (; r < r1; r += m.rpp)
    body(*reinterpret_cast<const half8 *>(pg + (size_t)r * ps_dy), *reinterpret_cast<const half8 *>(px + (size_t)r * ps_x),
         pa ? *reinterpret_cast<const half8 *>(pa + (size_t)r * ps_acc) : zero, (size_t)r);
}

// ---- finalize fused into its consumer (round 6) ------------------------------------------------------------------------------
// bn_finalize_kernel / bn_bwd_finalize_kernel are launches that stream nothing: 8 - 32 blocks turning [nblk][2][C] partials into C
// coefficients between the convolution that produced the partials and the pass that applies them -- 180 launches of 5 us per R101
// step.  Here the CONSUMER does that reduction itself: a workgroup owns a 64-channel slab (its 128-byte share of every pixel row)
// for its life, sums the slab's partials (nblk x 2 x 64 floats, L2-resident, every load independent) in double IN THE ORDER OF
// bn_sum_partials -- lane rl takes k = rl, rl + 32, ..., the 32 lanes are added 0 .. 31 -- so the coefficients are bit-identical to
// the separate kernel's, and then streams its rows.  The workgroups of row block 0 also write the per-channel outputs (scale /
// shift / saved statistics / moving averages; dgamma / dbeta) that later launches read.  Taken when nblk <= kBnFusedMaxBlocks and
// C % 64 == 0 (every train-mode BatchNorm of stages 3 - 4: 128 row tiles) AND the option is on; otherwise the separate finalize launch stays.
constexpr int kBnSlab = 64;              // channels per workgroup
constexpr int kBnFusedMaxBlocks = 160;   // partial rows a workgroup re-reduces (64 independent loads per thread at 128)
__device__ __forceinline__ void bn_slab_partials(const float *__restrict__ part, int nblk, int C, int c0, double (*red)[32][kBnSlab + 1],
                                                 double &a, double &b) {
  const int cl = threadIdx.x & 63, lg = threadIdx.x >> 6;     // channel of the slab, lane group 0 .. 3 (lanes rl = 8 lg .. 8 lg + 7)
  double s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.0;
  const float *p = part + c0 + cl;
  for (int k0 = 0; k0 < nblk; k0 += 32) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + lg * 8 + j;
      if (k < nblk) {
        s0[j] += (double)p[(size_t)k * 2 * C];
        s1[j] += (double)p[(size_t)k * 2 * C + C];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][lg * 8 + j][cl] = s0[j];
    red[1][lg * 8 + j][cl] = s1[j];
  }
  __syncthreads();
  a = b = 0.0;
  if (threadIdx.x < kBnSlab)
    for (int k = 0; k < 32; ++k) {
      a += red[0][k][cl];
      b += red[1][k][cl];
    }
}

// y = act(x * scale + shift) with scale / shift from the partials: finalize + apply in one launch.  grid = (row blocks, C / 64)
__global__ __launch_bounds__(256) void bn_apply_fin_kernel(const float *__restrict__ part, int nblk, const half_t *__restrict__ x,
                                                           half_t *__restrict__ y, int M, int C, int ps_in, int ps_out, int rows_per_block,
                                                           float eps, float momentum, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float *__restrict__ run_mean,
                                                           float *__restrict__ run_var, float *__restrict__ scale,
                                                           float *__restrict__ shift, float *__restrict__ save_mean,
                                                           float *__restrict__ save_invstd, int relu) {
  __shared__ double red[2][32][kBnSlab + 1];
  __shared__ float coef[2][kBnSlab];
  const int c0 = blockIdx.y * kBnSlab;
  double sum, sumsq;
  bn_slab_partials(part, nblk, C, c0, red, sum, sumsq);
  if (threadIdx.x < kBnSlab) {
    const int c = c0 + threadIdx.x;
    const double mean = sum / M;
    double var = sumsq / M - mean * mean;  // biased
    if (var < 0) var = 0;
    const float g = gamma ? gamma[c] : 1.f;
    float sc, sh, invstd;
    bn_fwd_coefs(mean, var, eps, g, beta[c], sc, sh, invstd);
    coef[0][threadIdx.x] = sc;
    coef[1][threadIdx.x] = sh;
    if (blockIdx.x == 0) {
      scale[c] = sc;
      shift[c] = sh;
      save_mean[c] = (float)mean;
      save_invstd[c] = invstd;
      if (run_mean) {
        run_mean[c] = bn_running(run_mean[c], (float)mean, momentum);
        run_var[c] = bn_running(run_var[c], (float)var, momentum);
      }
    }
  }
  __syncthreads();
  const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;       // 8-channel chunk of the slab, row lane 0 .. 31
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = coef[0][ch * 8 + j]; sh[j] = coef[1][ch * 8 + j]; }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  const half_t *px = x + c0 + ch * 8;
  half_t *py = y + c0 + ch * 8;
  auto body = [&](const half8 &v, size_t r) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = bn_affine((float)v[j], sc[j], sh[j]);
      if (relu) f = f > 0.f ? f : 0.f;
      if (relu == 2) f = f < 6.f ? f : 6.f;
      o[j] = (half_t)f;
    }
    *reinterpret_cast<half8 *>(py + r * ps_out) = o;
  };
  int r = r0 + rl;
  for (; r + (kBnUnroll - 1) * 32 < r1; r += kBnUnroll * 32) {
    half8 v[kBnUnroll];
#pragma unroll
    for (int u = 0; u < kBnUnroll; ++u) v[u] = *reinterpret_cast<const half8 *>(px + (size_t)(r + u * 32) * ps_in);
#pragma unroll
    for (int u = 0; u < kBnUnroll; ++u) body(v[u], (size_t)(r + u * 32));
  }
  for (; r < r1; r += 32) body(*reinterpret_cast<const half8 *>(px + (size_t)r * ps_in), (size_t)r);
}

// dx of (act o BN_train) with dgamma / dbeta from the partials: backward finalize + dx in one launch, same decomposition
__global__ __launch_bounds__(256) void bn_bwd_dx_fin_kernel(const float *__restrict__ part, int nblk, const half_t *__restrict__ dy,
                                                            const half_t *__restrict__ x, const half_t *__restrict__ acc,
                                                            half_t *__restrict__ dx, int M, int C, int ps_dy, int ps_x, int ps_acc,
                                                            int ps_dx, int rows_per_block, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, float *__restrict__ fin, int relu) {
  __shared__ double red[2][32][kBnSlab + 1];
  __shared__ float coef[4][kBnSlab];     // scale, shift, kb, kd
  const int c0 = blockIdx.y * kBnSlab;
  double sg, sgx;
  bn_slab_partials(part, nblk, C, c0, red, sg, sgx);
  if (threadIdx.x < kBnSlab) {
    const int c = c0 + threadIdx.x;
    const float db = (float)sg, dg = (float)(sgx * (double)invstd[c]);
    const float invM = 1.f / (float)M;
    const float scv = scale[c];
    float kb, kd;
    bn_dx_coefs(scv, invstd[c], mean[c], dg, db, invM, kb, kd);
    coef[0][threadIdx.x] = scv;
    coef[1][threadIdx.x] = shift[c];
    coef[2][threadIdx.x] = kb;
    coef[3][threadIdx.x] = kd;
    if (blockIdx.x == 0) {
      if (fin) { fin[c] = db; fin[C + c] = dg; }
      if (dbeta) dbeta[c] += db;
      if (dgamma) dgamma[c] += dg;
    }
  }
  __syncthreads();
  if (!dx) return;
  const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
  float sc[8], sh[8], kb[8], kd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = coef[0][ch * 8 + j]; sh[j] = coef[1][ch * 8 + j]; kb[j] = coef[2][ch * 8 + j]; kd[j] = coef[3][ch * 8 + j];
  }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  const half_t *pg = dy + c0 + ch * 8, *px = x + c0 + ch * 8, *pa = acc ? acc + c0 + ch * 8 : nullptr;
  half_t *po = dx + c0 + ch * 8;
  auto body = [&](const half8 &g, const half8 &v, const half8 &a, size_t r) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = (float)v[j];
      float gf = (float)g[j];
      if (!bn_act_pass(bn_affine(xf, sc[j], sh[j]), relu)) gf = 0.f;
      o[j] = (half_t)bn_dx_value(sc[j], gf, kb[j], xf, kd[j], (float)a[j]);
    }
    *reinterpret_cast<half8 *>(po + r * ps_dx) = o;
  };
  const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  int r = r0 + rl;
  for (; r + (kBnUnroll - 1) * 32 < r1; r += kBnUnroll * 32) {
    half8 g[kBnUnroll], v[kBnUnroll], a[kBnUnroll];
#pragma unroll
    for (int u = 0; u < kBnUnroll; ++u) {
      const size_t rr = (size_t)(r + u * 32);
      g[u] = *reinterpret_cast<const half8 *>(pg + rr * ps_dy);
      v[u] = *reinterpret_cast<const half8 *>(px + rr * ps_x);
      a[u] = pa ? *reinterpret_cast<const half8 *>(pa + rr * ps_acc) : zero;
    }
#pragma unroll
    for (int u = 0; u < kBnUnroll; ++u) body(g[u], v[u], a[u], (size_t)(r + u * 32));
  }
  for (; r < r1; r += 32)
    body(*reinterpret_cast<const half8 *>(pg + (size_t)r * ps_dy), *reinterpret_cast<const half8 *>(px + (size_t)r * ps_x),
         pa ? *reinterpret_cast<const half8 *>(pa + (size_t)r * ps_acc) : zero, (size_t)r);
}

// OPT-IN (sn_debug_option("bn_fused_finalize", 1) / SNIPER_BN_FUSED_FINALIZE): measured in the step it LOSES -- 22.09 vs 20.97 ms per step
// (profiles/r06_ab_bn_fused.txt, same card, interleaved): a slab workgroup re-reads 64 KB of partials for the 40 - 80 KB it streams
// and starts streaming only behind that dependent chain, so each of the 180 fused launches is ~11 us slower than the streaming
// kernel it replaces and saves a 5 us launch.  The partials would have to be ~8x fewer for this to pay.
static bool bn_fused_ok(int nblk, int M, int C) {
  return sn_debug_get(SN_OPT_BN_FUSED_FINALIZE) != 0 && nblk <= kBnFusedMaxBlocks && C % kBnSlab == 0 && M >= 1024;
}
// grid of the slab-owning kernels: about 1024 workgroups, >= 128 rows each (one unrolled pass of the 32 row lanes)
static dim3 bn_slab_grid(int M, int C, int *rows_per_block) {
  const int slabs = C / kBnSlab;
  int rb = 1024 / slabs;
  if (rb < 1) rb = 1;
  const int max_rb = sn_div_up(M, 128);
  if (rb > max_rb) rb = max_rb;
  *rows_per_block = sn_div_up(sn_div_up(M, rb), 32) * 32;
  return dim3(sn_div_up(M, *rows_per_block), slabs);
}

static int bn_shape_ok(int C) { return C >= 8 && C % 8 == 0; }
static int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
// grid of the row-walking BN kernels: x = row blocks (>= kBnUnroll*2 rows per thread, at most `cap`), y = 256-chunk slabs
static dim3 bn_grid(int M, int C, int cap, int *rows_per_block) {
  const int cpr = C / 8, cb = cpr < kBnThreads ? cpr : kBnThreads, rpp = kBnThreads / cb;
  int blocks = sn_div_up(M, rpp * kBnUnroll * 2);
  if (blocks > cap) blocks = cap;
  *rows_per_block = sn_div_up(sn_div_up(M, blocks), rpp) * rpp;
  return dim3(sn_div_up(M, *rows_per_block), sn_div_up(cpr, kBnThreads));
}
// row blocks of the two reduction kernels: their fp32 partials [blocks][2][C] stay <= 4 MB and the finalize
// kernels' per-thread chains short
static int bn_reduce_cap(int C) {
  int cap = (1 << 22) / (8 * C);
  return cap < 256 ? 256 : (cap > 512 ? 512 : cap);
}

SN_EXPORT size_t sn_bn_workspace_bytes(int M, int C) {
  if (M <= 0 || !bn_shape_ok(C)) return 0;
  int rpb;
  const dim3 g = bn_grid(M, C, bn_reduce_cap(C), &rpb);
  return sn_align(sizeof(float) * 2 * (size_t)C * (g.x + 1));
}

SN_EXPORT int sn_bn_stats(const void *x, int M, int C, int ps, void *ws, sn_stream_t stream) {
  SN_REQUIRE(x && ws && M > 0 && bn_shape_ok(C), "sn_bn_stats: bad arguments (C=%d must be a multiple of 8)", C);
  int rows_per_block;
  const dim3 grid = bn_grid(M, C, bn_reduce_cap(C), &rows_per_block);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(kBnThreads), 0, sn_stream(stream), (const half_t *)x, M, C, ps, rows_per_block,
                     (float *)ws + 2 * C);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_bn_finalize(const void *ws, int M, int C, float eps, float momentum, const float *gamma, const float *beta,
                             float *run_mean, float *run_var, float *scale, float *shift, float *save_mean, float *save_invstd,
                             sn_stream_t stream) {
  SN_REQUIRE(ws && beta && scale && shift && save_mean && save_invstd && M > 0 && bn_shape_ok(C), "sn_bn_finalize: bad arguments");
  int rows_per_block;
  const dim3 grid = bn_grid(M, C, bn_reduce_cap(C), &rows_per_block);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(sn_div_up(C, 32)), dim3(kBnFinThreads), 0, sn_stream(stream), (const float *)ws + 2 * C,
                     (int)grid.x, M, C, eps, momentum, gamma, beta, run_mean, run_var, scale, shift, save_mean, save_invstd);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// bn_finalize over partials produced elsewhere (sn_conv_fwd_stats: one row block per convolution row tile)
SN_EXPORT int sn_bn_finalize_blocks(const float *partials, int nblk, int M, int C, float eps, float momentum, const float *gamma,
                                    const float *beta, float *run_mean, float *run_var, float *scale, float *shift,
                                    float *save_mean, float *save_invstd, sn_stream_t stream) {
  SN_REQUIRE(partials && nblk > 0 && beta && scale && shift && save_mean && save_invstd && M > 0 && C > 0,
             "sn_bn_finalize_blocks: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(sn_div_up(C, 32)), dim3(kBnFinThreads), 0, sn_stream(stream), partials, nblk, M, C, eps,
                     momentum, gamma, beta, run_mean, run_var, scale, shift, save_mean, save_invstd);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_bn_global_scale_shift(const float *gamma, const float *beta, const float *mean, const float *var, int C,
                                       float eps, float *scale, float *shift, sn_stream_t stream) {
  SN_REQUIRE(beta && mean && var && scale && shift, "sn_bn_global_scale_shift: null pointer");
  hipLaunchKernelGGL(bn_global_kernel, dim3(sn_div_up(C, 256)), dim3(256), 0, sn_stream(stream), gamma, beta, mean, var, C,
                     eps, scale, shift);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_bn_apply(const void *x, void *y, int M, int C, int ps_in, int ps_out, const float *scale,
                          const float *shift, int relu, sn_stream_t stream) {
  SN_REQUIRE(x && y && scale && shift && M > 0 && C % 8 == 0, "sn_bn_apply: bad arguments");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks((long)M * (C / 8))), dim3(256), 0, sn_stream(stream),
                     (const half_t *)x, (half_t *)y, M, C, ps_in, ps_out, scale, shift, relu);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// sn_bn_finalize_blocks + sn_bn_apply as ONE launch where the partials are few enough for the applying workgroups to reduce
// themselves (bn_apply_fin_kernel), two launches otherwise: same outputs either way, bit for bit.
SN_EXPORT int sn_bn_apply_blocks(const float *partials, int nblk, const void *x, void *y, int M, int C, int ps_in, int ps_out, float eps,
                                 float momentum, const float *gamma, const float *beta, float *run_mean, float *run_var, float *scale,
                                 float *shift, float *save_mean, float *save_invstd, int relu, sn_stream_t stream) {
  SN_REQUIRE(partials && nblk > 0 && x && y && beta && scale && shift && save_mean && save_invstd && M > 0 && bn_shape_ok(C),
             "sn_bn_apply_blocks: bad arguments (C=%d)", C);
  if (!bn_fused_ok(nblk, M, C)) {
    if (int rc = sn_bn_finalize_blocks(partials, nblk, M, C, eps, momentum, gamma, beta, run_mean, run_var, scale, shift, save_mean,
                                       save_invstd, stream))
      return rc;
    return sn_bn_apply(x, y, M, C, ps_in, ps_out, scale, shift, relu, stream);
  }
  int rows_per_block;
  const dim3 grid = bn_slab_grid(M, C, &rows_per_block);
  hipLaunchKernelGGL(bn_apply_fin_kernel, grid, dim3(256), 0, sn_stream(stream), partials, nblk, (const half_t *)x, (half_t *)y, M, C,
                     ps_in, ps_out, rows_per_block, eps, momentum, gamma, beta, run_mean, run_var, scale, shift, save_mean, save_invstd,
                     relu);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// Backward of y = relu?(BN_train(x)).  dgamma/dbeta (fp32, length C) are ACCUMULATED into (+=, the optimizer's
// gradient arena is zeroed once per step); ws = sn_bn_workspace_bytes(M, C).  Three launches, no atomics, no memset:
// partial reduce -> finalize -> dx.
SN_EXPORT int sn_bn_backward(const void *dy, const void *x, const void *accumulate, void *dx, int M, int C, int ps_dy,
                             int ps_x, int ps_acc, int ps_dx, const float *scale, const float *shift, const float *mean,
                             const float *invstd, int relu, void *ws, float *dgamma, float *dbeta, sn_stream_t stream) {
  SN_REQUIRE(dy && x && scale && shift && mean && invstd && ws && bn_shape_ok(C) && M > 0,
             "sn_bn_backward: bad arguments (C=%d)", C);
  hipStream_t s = sn_stream(stream);
  float *fin = (float *)ws, *part = fin + 2 * C;
  int rows_per_block;
  dim3 grid = bn_grid(M, C, bn_reduce_cap(C), &rows_per_block);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(kBnThreads), 0, s, (const half_t *)dy, (const half_t *)x, M, C, ps_dy,
                     ps_x, rows_per_block, scale, shift, mean, relu, part);
  SN_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sn_div_up(C, 32)), dim3(kBnFinThreads), 0, s, (const float *)part, (int)grid.x, C,
                     invstd, fin, dgamma, dbeta);
  SN_CHECK_LAUNCH();
  if (dx) {
    grid = bn_grid(M, C, 8192, &rows_per_block);
    hipLaunchKernelGGL(bn_bwd_dx_kernel, grid, dim3(kBnThreads), 0, s, (const half_t *)dy, (const half_t *)x,
                       (const half_t *)accumulate, (half_t *)dx, M, C, ps_dy, ps_x, ps_acc, ps_dx, rows_per_block, scale, shift,
                       mean, invstd, (const float *)fin + C, (const float *)fin, relu);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}

// sn_bn_backward with the reduction already done elsewhere (sn_conv_dgrad_bn: partials (nblk, 2, C) = sum g, sum g*(x-mean)
// per row tile of the data-gradient convolution): finalize + dx only.
SN_EXPORT int sn_bn_backward_blocks(const float *partials, int nblk, const void *dy, const void *x, const void *accumulate, void *dx,
                                    int M, int C, int ps_dy, int ps_x, int ps_acc, int ps_dx, const float *scale, const float *shift,
                                    const float *mean, const float *invstd, int relu, void *ws, float *dgamma, float *dbeta,
                                    sn_stream_t stream) {
  SN_REQUIRE(partials && nblk > 0 && dy && x && scale && shift && mean && invstd && ws && bn_shape_ok(C) && M > 0,
             "sn_bn_backward_blocks: bad arguments (C=%d)", C);
  hipStream_t s = sn_stream(stream);
  float *fin = (float *)ws;
  if (bn_fused_ok(nblk, M, C)) {      // finalize inside the dx pass (bn_bwd_dx_fin_kernel): one launch
    int rows_per_block;
    dim3 grid = bn_slab_grid(M, C, &rows_per_block);
    if (!dx) grid.x = 1;
    hipLaunchKernelGGL(bn_bwd_dx_fin_kernel, grid, dim3(256), 0, s, partials, nblk, (const half_t *)dy, (const half_t *)x,
                       (const half_t *)accumulate, (half_t *)dx, M, C, ps_dy, ps_x, ps_acc, ps_dx, rows_per_block, scale, shift, mean,
                       invstd, dgamma, dbeta, fin, relu);
    SN_CHECK_LAUNCH();
    return SN_OK;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sn_div_up(C, 32)), dim3(kBnFinThreads), 0, s, partials, nblk, C, invstd, fin, dgamma,
                     dbeta);
  SN_CHECK_LAUNCH();
  if (dx) {
    int rows_per_block;
    const dim3 grid = bn_grid(M, C, 8192, &rows_per_block);
    hipLaunchKernelGGL(bn_bwd_dx_kernel, grid, dim3(kBnThreads), 0, s, (const half_t *)dy, (const half_t *)x,
                       (const half_t *)accumulate, (half_t *)dx, M, C, ps_dy, ps_x, ps_acc, ps_dx, rows_per_block, scale, shift,
                       mean, invstd, (const float *)fin + C, (const float *)fin, relu);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// ReLU / add / relu-gradient on channels-last fp16 (used where BN fusion does not apply)
// ---------------------------------------------------------------------------------------------
// mode 0: y = relu(a);  1: y = a + b;  2: y = (ref > 0) ? a : 0 [+ b]  (relu backward, optional accumulate)
__global__ __launch_bounds__(256) void ew_f16_kernel(const half_t *__restrict__ a, const half_t *__restrict__ b,
                                                     const half_t *__restrict__ ref, half_t *__restrict__ y, long rows, int C,
                                                     int ps_a, int ps_b, int ps_ref, int ps_y, int mode) {
  const int cpr = C >> 3;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    const long r = i / cpr;
    const half8 va = *reinterpret_cast<const half8 *>(a + r * ps_a + ch);
    half8 o;
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = va[j] > (half_t)0 ? va[j] : (half_t)0;
    } else if (mode == 1) {
      const half8 vb = *reinterpret_cast<const half8 *>(b + r * ps_b + ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)va[j] + (float)vb[j]);
    } else {
      const half8 vr = *reinterpret_cast<const half8 *>(ref + r * ps_ref + ch);
      half8 vb = {0, 0, 0, 0, 0, 0, 0, 0};
      if (b) vb = *reinterpret_cast<const half8 *>(b + r * ps_b + ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)((vr[j] > (half_t)0 ? (float)va[j] : 0.f) + (float)vb[j]);
    }
    *reinterpret_cast<half8 *>(y + r * ps_y + ch) = o;
  }
}

SN_EXPORT int sn_ew_f16(const void *a, const void *b, const void *ref, void *y, long rows, int C, int ps_a, int ps_b,
                        int ps_ref, int ps_y, int mode, sn_stream_t stream) {
  SN_REQUIRE(a && y && rows > 0 && C % 8 == 0 && mode >= 0 && mode <= 2, "sn_ew_f16: bad arguments");
  SN_REQUIRE(mode != 1 || b, "sn_ew_f16: add needs b");
  SN_REQUIRE(mode != 2 || ref, "sn_ew_f16: relu backward needs ref");
  hipLaunchKernelGGL(ew_f16_kernel, dim3(ew_blocks(rows * (C / 8))), dim3(256), 0, sn_stream(stream), (const half_t *)a,
                     (const half_t *)b, (const half_t *)ref, (half_t *)y, rows, C, ps_a, ps_b, ps_ref, ps_y, mode);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// Max pooling, channels-last fp16 (resnet_mx_101_e2e.py:409: 3x3 stride 2 pad 1).  Forward only:
// the stem is frozen (network.FIXED_PARAMS), nothing upstream needs a gradient.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const half_t *__restrict__ x, half_t *__restrict__ y, int N, int H,
                                                      int W, int C, int Ho, int Wo, int k, int stride, int pad) {
  const int cpr = C >> 3;
  const long total = (long)N * Ho * Wo * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr) * 8;
    long t = i / cpr;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -65504.f;
    for (int ky = 0; ky < k; ++ky) {
      const int sy = oy * stride - pad + ky;
      if ((unsigned)sy >= (unsigned)H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int sx = ox * stride - pad + kx;
        if ((unsigned)sx >= (unsigned)W) continue;
        const half8 v = *reinterpret_cast<const half8 *>(x + (((long)n * H + sy) * W + sx) * C + ch);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)m[j];
    *reinterpret_cast<half8 *>(y + i * 8) = o;
  }
}

SN_EXPORT int sn_maxpool_fwd(const void *x, void *y, int N, int H, int W, int C, int k, int stride, int pad,
                             sn_stream_t stream) {
  SN_REQUIRE(x && y && C % 8 == 0 && k > 0 && stride > 0, "sn_maxpool_fwd: bad arguments");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  hipLaunchKernelGGL(maxpool_kernel, dim3(ew_blocks((long)N * Ho * Wo * (C / 8))), dim3(256), 0, sn_stream(stream),
                     (const half_t *)x, (half_t *)y, N, H, W, C, Ho, Wo, k, stride, pad);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// Global average pooling (the R-FCN vote over the P x P position-sensitive bins, BASELINE config C4):
// x (N, HW, C) fp16 channels-last -> y (N, C) fp32 = mean over the HW positions, and its gradient dx = dy / HW.
__global__ __launch_bounds__(256) void avgpool_global_fwd_kernel(const half_t *__restrict__ x, float *__restrict__ y, long NC,
                                                                 int HW, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NC) return;
  const long n = i / C;
  const int c = (int)(i - n * C);
  const half_t *src = x + (size_t)n * HW * C + c;
  float s = 0.f;
  for (int k = 0; k < HW; ++k) s += (float)src[(size_t)k * C];
  y[i] = s / (float)HW;
}

__global__ __launch_bounds__(256) void avgpool_global_bwd_kernel(const float *__restrict__ dy, half_t *__restrict__ dx, long total,
                                                                 int HW, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long n = i / ((long)HW * C);
  dx[i] = (half_t)(dy[n * C + c] / (float)HW);
}

SN_EXPORT int sn_avgpool_global_fwd(const void *x, float *y, int N, int HW, int C, sn_stream_t stream) {
  SN_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "sn_avgpool_global_fwd: bad arguments");
  const long NC = (long)N * C;
  hipLaunchKernelGGL(avgpool_global_fwd_kernel, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, sn_stream(stream),
                     (const half_t *)x, y, NC, HW, C);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_avgpool_global_bwd(const float *dy, void *dx, int N, int HW, int C, sn_stream_t stream) {
  SN_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0, "sn_avgpool_global_bwd: bad arguments");
  const long total = (long)N * HW * C;
  hipLaunchKernelGGL(avgpool_global_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sn_stream(stream), dy,
                     (half_t *)dx, total, HW, C);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// Layout / dtype conversion between the reference's NCHW tensors and internal NHWC.
//   nhwc(src, pixel stride ps, dtype f16|f32) -> nchw (dst contiguous, dtype f16|f32) and back.
// Tiled through LDS so that both sides are coalesced.
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI *__restrict__ in, TO *__restrict__ out, int rows, int cols,
                                                        long in_batch, long out_batch, int in_ld, int out_ld) {
  // out[b][c][r] = in[b][r][c]  (rows x cols per batch)
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) tile[k][tx] = (float)in[(long)b * in_batch + (long)r * in_ld + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (r < rows && c < cols) out[(long)b * out_batch + (long)c * out_ld + r] = (TO)tile[tx][k];
  }
}

// dtype codes: 0 = f16, 1 = f32
SN_EXPORT int sn_transpose_batched(const void *in, void *out, int batch, int rows, int cols, long in_batch_stride,
                                   long out_batch_stride, int in_ld, int out_ld, int in_dtype, int out_dtype,
                                   sn_stream_t stream) {
  SN_REQUIRE(in && out && batch > 0 && rows > 0 && cols > 0, "sn_transpose_batched: bad arguments");
  SN_REQUIRE(batch <= 65535, "sn_transpose_batched: batch too large");
  dim3 grid(sn_div_up(cols, 32), sn_div_up(rows, 32), batch);
  hipStream_t s = sn_stream(stream);
  if (in_dtype == 0 && out_dtype == 0)
    hipLaunchKernelGGL((transpose_kernel<half_t, half_t>), grid, dim3(256), 0, s, (const half_t *)in, (half_t *)out, rows, cols,
                       in_batch_stride, out_batch_stride, in_ld, out_ld);
  else if (in_dtype == 0 && out_dtype == 1)
    hipLaunchKernelGGL((transpose_kernel<half_t, float>), grid, dim3(256), 0, s, (const half_t *)in, (float *)out, rows, cols,
                       in_batch_stride, out_batch_stride, in_ld, out_ld);
  else if (in_dtype == 1 && out_dtype == 0)
    hipLaunchKernelGGL((transpose_kernel<float, half_t>), grid, dim3(256), 0, s, (const float *)in, (half_t *)out, rows, cols,
                       in_batch_stride, out_batch_stride, in_ld, out_ld);
  else
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float *)in, (float *)out, rows, cols,
                       in_batch_stride, out_batch_stride, in_ld, out_ld);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// strided 2-D copy with dtype conversion: out[r][c] = in[r][c], c < cols (Concat slices, Cast)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy2d_kernel(const TI *__restrict__ in, TO *__restrict__ out, long rows, int cols,
                                                     int in_ld, int out_ld) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i - r * cols);
    out[r * out_ld + c] = (TO)in[r * in_ld + c];
  }
}

// the same copy, eight elements per thread (16-byte accesses of the fp16 side, 2 x 16 of an fp32 side): rows and pitches that
// are multiples of 8 elements on 16-byte aligned pointers -- Concat slices (63 MB per step), the step's input copies (80 MB).
// cpr8 = cols / 8; row index by multiplication (fd = conv_fastdiv_make(cpr8))
template <typename T>
__device__ __forceinline__ void load8(const T *p, float v[8]);
template <>
__device__ __forceinline__ void load8<half_t>(const half_t *p, float v[8]) {
  const half8 h = *reinterpret_cast<const half8 *>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
}
template <>
__device__ __forceinline__ void load8<float>(const float *p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T>
__device__ __forceinline__ void store8(T *p, const float v[8]);
template <>
__device__ __forceinline__ void store8<half_t>(half_t *p, const float v[8]) {
  half8 h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
  *reinterpret_cast<half8 *>(p) = h;
}
template <>
__device__ __forceinline__ void store8<float>(float *p, const float v[8]) {
  *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy2d_vec8_kernel(const TI *__restrict__ in, TO *__restrict__ out, long total8, int cpr8,
                                                          int in_ld, int out_ld, unsigned fd_mul, unsigned fd_sh) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    long r;
    int c;
    if (in_ld == cpr8 * 8 && out_ld == in_ld) { r = 0; c = 0; }      // contiguous on both sides: a flat copy
    else if (i < (1l << 31)) { r = (long)(((unsigned long long)__umulhi((unsigned)i, fd_mul) + (unsigned)i) >> fd_sh); c = (int)(i - r * cpr8); }
    else { r = i / cpr8; c = (int)(i - r * cpr8); }
    const size_t si = (in_ld == cpr8 * 8 && out_ld == in_ld) ? (size_t)i * 8 : (size_t)r * in_ld + (size_t)c * 8;
    const size_t di = (in_ld == cpr8 * 8 && out_ld == in_ld) ? (size_t)i * 8 : (size_t)r * out_ld + (size_t)c * 8;
    float v[8];
    load8<TI>(in + si, v);
    store8<TO>(out + di, v);
  }
}
static void fastdiv_make_u32(unsigned d, unsigned &mul, unsigned &sh) {
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << s) - d)) / d + 1);
  sh = s;
}

SN_EXPORT int sn_copy2d(const void *in, void *out, long rows, int cols, int in_ld, int out_ld, int in_dtype, int out_dtype,
                        sn_stream_t stream) {
  SN_REQUIRE(in && out && rows > 0 && cols > 0, "sn_copy2d: bad arguments");
  hipStream_t s = sn_stream(stream);
  if (cols % 8 == 0 && (in_ld % 8 == 0 || rows == 1) && (out_ld % 8 == 0 || rows == 1) && ((uintptr_t)in % 16) == 0 &&
      ((uintptr_t)out % 16) == 0) {
    const int cpr8 = cols / 8;
    const long total8 = rows * cpr8;
    if (rows == 1) in_ld = out_ld = cols;
    unsigned mul, sh;
    fastdiv_make_u32((unsigned)cpr8, mul, sh);
    const dim3 g(ew_blocks(total8));
    if (in_dtype == 0 && out_dtype == 0)
      hipLaunchKernelGGL((copy2d_vec8_kernel<half_t, half_t>), g, dim3(256), 0, s, (const half_t *)in, (half_t *)out, total8, cpr8, in_ld, out_ld, mul, sh);
    else if (in_dtype == 0 && out_dtype == 1)
      hipLaunchKernelGGL((copy2d_vec8_kernel<half_t, float>), g, dim3(256), 0, s, (const half_t *)in, (float *)out, total8, cpr8, in_ld, out_ld, mul, sh);
    else if (in_dtype == 1 && out_dtype == 0)
      hipLaunchKernelGGL((copy2d_vec8_kernel<float, half_t>), g, dim3(256), 0, s, (const float *)in, (half_t *)out, total8, cpr8, in_ld, out_ld, mul, sh);
    else
      hipLaunchKernelGGL((copy2d_vec8_kernel<float, float>), g, dim3(256), 0, s, (const float *)in, (float *)out, total8, cpr8, in_ld, out_ld, mul, sh);
    SN_CHECK_LAUNCH();
    return SN_OK;
  }
  const dim3 grid(ew_blocks(rows * cols));
  if (in_dtype == 0 && out_dtype == 0)
    hipLaunchKernelGGL((copy2d_kernel<half_t, half_t>), grid, dim3(256), 0, s, (const half_t *)in, (half_t *)out, rows, cols, in_ld, out_ld);
  else if (in_dtype == 0 && out_dtype == 1)
    hipLaunchKernelGGL((copy2d_kernel<half_t, float>), grid, dim3(256), 0, s, (const half_t *)in, (float *)out, rows, cols, in_ld, out_ld);
  else if (in_dtype == 1 && out_dtype == 0)
    hipLaunchKernelGGL((copy2d_kernel<float, half_t>), grid, dim3(256), 0, s, (const float *)in, (half_t *)out, rows, cols, in_ld, out_ld);
  else
    hipLaunchKernelGGL((copy2d_kernel<float, float>), grid, dim3(256), 0, s, (const float *)in, (float *)out, rows, cols, in_ld, out_ld);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// SoftmaxOutput(multi_output, use_ignore, ignore_label, normalization='valid', grad_scale)
// (resnet_mx_101_e2e.py:279-281, 310-311, 314-315).  Logical view: data (outer, K, inner) fp32
// contiguous, label (outer, inner).  forward: softmax over K.  backward:
//     grad = grad_scale / max(1, #labels != ignore) * (p - onehot(label)),  0 where label == ignore
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float *__restrict__ x, float *__restrict__ p, long outer, int K,
                                                          long inner) {
  const long total = outer * inner;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long o = i / inner, s = i - o * inner;
    const float *xi = x + o * K * inner + s;
    float m = xi[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, xi[(long)k * inner]);
    float z = 0.f;
    for (int k = 0; k < K; ++k) z += expf(xi[(long)k * inner] - m);
    const float iz = 1.f / z;
    float *pi = p + o * K * inner + s;
    for (int k = 0; k < K; ++k) pi[(long)k * inner] = expf(xi[(long)k * inner] - m) * iz;
  }
}

__global__ __launch_bounds__(256) void count_valid_kernel(const float *__restrict__ label, long n, float ignore,
                                                          int *__restrict__ count) {
  int c = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    c += (label[i] != ignore);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ p, const float *__restrict__ label,
                                                          float *__restrict__ g, long outer, int K, long inner,
                                                          float ignore, int use_ignore, float grad_scale, int normalize_valid,
                                                          const int *__restrict__ valid_count) {
  const long total = outer * inner;
  float mul = grad_scale;
  if (normalize_valid) {
    const int vc = *valid_count;
    mul = grad_scale / (float)(vc > 1 ? vc : 1);
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long o = i / inner, s = i - o * inner;
    const float l = label[i];
    const bool ign = use_ignore && l == ignore;
    const int li = (int)l;
    for (int k = 0; k < K; ++k) {
      const long idx = o * K * inner + (long)k * inner + s;
      g[idx] = ign ? 0.f : mul * (p[idx] - (k == li ? 1.f : 0.f));
    }
  }
}

SN_EXPORT int sn_softmax_fwd(const float *x, float *p, long outer, int K, long inner, sn_stream_t stream) {
  SN_REQUIRE(x && p && outer > 0 && K > 0 && inner > 0, "sn_softmax_fwd: bad arguments");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(ew_blocks(outer * inner)), dim3(256), 0, sn_stream(stream), x, p, outer, K, inner);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ws: one int (valid-label counter)
SN_EXPORT int sn_softmax_output_bwd(const float *p, const float *label, float *grad, long outer, int K, long inner,
                                    float ignore_label, int use_ignore, float grad_scale, int normalize_valid, int *ws,
                                    sn_stream_t stream) {
  SN_REQUIRE(p && label && grad && ws, "sn_softmax_output_bwd: null pointer");
  hipStream_t s = sn_stream(stream);
  if (normalize_valid) {
    SN_HIP(hipMemsetAsync(ws, 0, sizeof(int), s));
    hipLaunchKernelGGL(count_valid_kernel, dim3(ew_blocks(outer * inner)), dim3(256), 0, s, label, outer * inner,
                       use_ignore ? ignore_label : -1e30f, ws);
    SN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ew_blocks(outer * inner)), dim3(256), 0, s, p, label, grad, outer, K, inner,
                     ignore_label, use_ignore, grad_scale, normalize_valid, ws);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// weight * smooth_l1(pred - target, sigma) and its MakeLoss gradient (resnet_mx_101_e2e.py:317-319,
// 330-334).  fwd: loss = w * f(d), d = pred - target;  bwd: dpred = grad_scale * w * f'(d)
//   f(d) = 0.5 (sigma d)^2 if |d| < 1/sigma^2 else |d| - 0.5/sigma^2
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                        const float *__restrict__ weight, float *__restrict__ loss,
                                                        float *__restrict__ dpred, long n, float sigma, float grad_scale) {
  const float s2 = sigma * sigma, th = 1.f / s2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = pred[i] - target[i], w = weight[i];
    const float ad = fabsf(d);
    if (loss) loss[i] = w * (ad < th ? 0.5f * s2 * d * d : ad - 0.5f * th);
    if (dpred) dpred[i] = grad_scale * w * (ad < th ? s2 * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
  }
}

SN_EXPORT int sn_smooth_l1_loss(const float *pred, const float *target, const float *weight, float *loss, float *dpred,
                                long n, float sigma, float grad_scale, sn_stream_t stream) {
  SN_REQUIRE(pred && target && weight && n > 0, "sn_smooth_l1_loss: bad arguments");
  hipLaunchKernelGGL(smooth_l1_kernel, dim3(ew_blocks(n)), dim3(256), 0, sn_stream(stream), pred, target, weight, loss, dpred,
                     n, sigma, grad_scale);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-precision SGD with momentum (lib/train_utils/utils.py:26-33 -> mx 'sgd', multi_precision):
//   g   = rescale * grad + wd * w32
//   mom = momentum * mom - lr * g ;  w32 += mom ;  w16 = half(w32)
// One launch per parameter group (same lr/wd); grad is the fp32 arena the wgrad kernels and the
// RCCL all-reduce work on.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ w32, const float *__restrict__ grad, float *__restrict__ mom,
                                                  half_t *__restrict__ w16, long n, float lr, float wd, float momentum,
                                                  float rescale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float w = w32[i];
    const float g = rescale * grad[i] + wd * w;
    const float m = momentum * mom[i] - lr * g;
    mom[i] = m;
    const float nw = w + m;
    w32[i] = nw;
    if (w16) w16[i] = (half_t)nw;
  }
}

SN_EXPORT int sn_sgd_mom_update(float *w32, const float *grad, float *mom, void *w16, long n, float lr, float wd,
                                float momentum, float rescale, sn_stream_t stream) {
  SN_REQUIRE(w32 && grad && mom && n >= 0, "sn_sgd_mom_update: bad arguments");
  if (n == 0) return SN_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_blocks(n)), dim3(256), 0, sn_stream(stream), w32, grad, mom, (half_t *)w16, n, lr, wd,
                     momentum, rescale);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// Same update with the per-step hyper-parameters read from device memory (d_hyper = [lr, wd, momentum, rescale]):
// a launch captured in a hipGraph must not bake the scheduled learning rate into its arguments.
__global__ __launch_bounds__(256) void sgd_dev_kernel(float *__restrict__ w32, const float *__restrict__ grad,
                                                      float *__restrict__ mom, half_t *__restrict__ w16, long n,
                                                      const float *__restrict__ hyper, float lr_mult, float wd_mult) {
  const float lr = hyper[0] * lr_mult, wd = hyper[1] * wd_mult, momentum = hyper[2], rescale = hyper[3];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float w = w32[i];
    const float g = rescale * grad[i] + wd * w;
    const float m = momentum * mom[i] - lr * g;
    mom[i] = m;
    const float nw = w + m;
    w32[i] = nw;
    if (w16) w16[i] = (half_t)nw;
  }
}

// four parameters per thread (16-B loads of master / gradient / momentum, 8-B store of the fp16 copy): the update is a pure
// 22 B-per-parameter stream, and scalar 4-B accesses left it at a third of the HBM rate
__global__ __launch_bounds__(256) void sgd_dev_vec4_kernel(float4 *__restrict__ w32, const float4 *__restrict__ grad,
                                                           float4 *__restrict__ mom, half_t *__restrict__ w16, long n4,
                                                           const float *__restrict__ hyper, float lr_mult, float wd_mult) {
  const float lr = hyper[0] * lr_mult, wd = hyper[1] * wd_mult, momentum = hyper[2], rescale = hyper[3];
  typedef half_t half4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 w = w32[i], g4 = grad[i], m4 = mom[i];
    float4 nm, nw;
#define SN_SGD_LANE(f)                                                  \
    {                                                                   \
      const float g = rescale * g4.f + wd * w.f;                        \
      nm.f = momentum * m4.f - lr * g;                                  \
      nw.f = w.f + nm.f;                                                \
    }
    SN_SGD_LANE(x) SN_SGD_LANE(y) SN_SGD_LANE(z) SN_SGD_LANE(w)
#undef SN_SGD_LANE
    mom[i] = nm;
    w32[i] = nw;
    if (w16) {
      half4 h;
      h[0] = (half_t)nw.x; h[1] = (half_t)nw.y; h[2] = (half_t)nw.z; h[3] = (half_t)nw.w;
      *reinterpret_cast<half4 *>(w16 + 4 * i) = h;
    }
  }
}

SN_EXPORT int sn_sgd_mom_update_dev(float *w32, const float *grad, float *mom, void *w16, long n, const float *d_hyper,
                                    float lr_mult, float wd_mult, sn_stream_t stream) {
  SN_REQUIRE(w32 && grad && mom && d_hyper && n >= 0, "sn_sgd_mom_update_dev: bad arguments");
  if (n == 0) return SN_OK;
  // the parameter groups are slices [a, b) of one arena: peel the 0-3 leading elements up to a 16-B boundary (the same count for
  // all four arrays when the arenas themselves are aligned), run the body four wide, finish the tail one at a time
  auto scalar = [&](long from, long count) {
    hipLaunchKernelGGL(sgd_dev_kernel, dim3(ew_blocks(count)), dim3(256), 0, sn_stream(stream), w32 + from, grad + from,
                       mom + from, w16 ? (half_t *)w16 + from : (half_t *)nullptr, count, d_hyper, lr_mult, wd_mult);
  };
  long head = (long)(((16 - ((uintptr_t)w32 & 15)) & 15) / 4);
  if (head > n) head = n;
  const bool same = (((uintptr_t)(w32 + head) | (uintptr_t)(grad + head) | (uintptr_t)(mom + head)) & 15) == 0 &&
                    (w16 == nullptr || ((uintptr_t)((half_t *)w16 + head) & 7) == 0);
  if (!same) {
    scalar(0, n);
    SN_CHECK_LAUNCH();
    return SN_OK;
  }
  if (head > 0) scalar(0, head);
  const long n4 = (n - head) / 4;
  if (n4 > 0)
    hipLaunchKernelGGL(sgd_dev_vec4_kernel, dim3(ew_blocks(n4)), dim3(256), 0, sn_stream(stream), (float4 *)(w32 + head),
                       (const float4 *)(grad + head), (float4 *)(mom + head), w16 ? (half_t *)w16 + head : (half_t *)nullptr, n4,
                       d_hyper, lr_mult, wd_mult);
  const long done = head + 4 * n4;
  if (done < n) scalar(done, n - done);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// fp32 [O][T][I] -> fp16 [I][T][Opad]  (weights for the data-gradient GEMM; Opad >= O is the
// 8-aligned contraction length, columns O..Opad-1 are zero so a zero-padded dY is harmless)
__global__ __launch_bounds__(256) void weight_oti_to_ito_kernel(const float *__restrict__ w, half_t *__restrict__ wt, int O, int T,
                                                                int I, int Opad) {
  const long total = (long)Opad * T * I;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i indexes the destination so that writes are coalesced
    const int o = (int)(i % Opad);
    const long r = i / Opad;
    const int t = (int)(r % T), ci = (int)(r / T);
    wt[i] = o < O ? (half_t)w[((long)o * T + t) * I + ci] : (half_t)0;
  }
}

SN_EXPORT int sn_weight_transpose(const float *w_oti, void *wt_ito_f16, int O, int T, int I, int O_pad, sn_stream_t stream) {
  SN_REQUIRE(w_oti && wt_ito_f16 && O > 0 && T > 0 && I > 0 && O_pad >= O, "sn_weight_transpose: bad arguments");
  hipLaunchKernelGGL(weight_oti_to_ito_kernel, dim3(ew_blocks((long)O_pad * T * I)), dim3(256), 0, sn_stream(stream), w_oti,
                     (half_t *)wt_ito_f16, O, T, I, O_pad);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// All transposed copies of a step in ONE launch: 64 x 64 tiles through LDS (coalesced 256-B reads of the fp32 masters,
// 128-B writes of the fp16 copies) instead of one strided-read launch per weight (120 launches, 0.87 ms per R101 step).
// desc[k] = {src, dst, O, T, I, Opad, tile0, tiles_o, tiles_i}; weight k owns tiles [tile0, tile0 + T*tiles_o*tiles_i).
struct WtDesc {
  const float *src;
  half_t *dst;
  int O, T, I, Opad, tile0, tiles_o, tiles_i, pad_;
};
static_assert(sizeof(WtDesc) == 48, "WtDesc layout is part of the C ABI (sn_weight_transpose_batched)");

__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const WtDesc *__restrict__ desc, int nd) {
  __shared__ float sm[64][65];
  const int tile = blockIdx.x;
  int lo = 0, hi = nd - 1;                 // last descriptor with tile0 <= tile
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
  }
  const WtDesc d = desc[lo];
  const int local = tile - d.tile0, per_t = d.tiles_o * d.tiles_i;
  const int t = local / per_t, rem = local - t * per_t;
  const int o0 = (rem / d.tiles_i) * 64, i0 = (rem % d.tiles_i) * 64;
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int o = o0 + r + 4 * k, ci = i0 + c;
    sm[r + 4 * k][c] = (o < d.O && ci < d.I) ? d.src[((size_t)o * d.T + t) * d.I + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int ci = i0 + r + 4 * k, o = o0 + c;
    if (ci < d.I && o < d.Opad) d.dst[((size_t)ci * d.T + t) * d.Opad + o] = (half_t)sm[c][r + 4 * k];
  }
}

SN_EXPORT int sn_weight_transpose_batched(const void *desc, int n_desc, int total_tiles, sn_stream_t stream) {
  SN_REQUIRE(desc && n_desc > 0 && total_tiles > 0, "sn_weight_transpose_batched: bad arguments");
  hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, sn_stream(stream), (const WtDesc *)desc,
                     n_desc);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// generic fp32 element-wise helpers for the small loss-side tensors: out = a (op) b
// op 0: a - b, 1: a + b, 2: a * b, 3: a * scalar, 4: fill(scalar)
__global__ __launch_bounds__(256) void ew_f32_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out,
                                                     long n, int op, float scalar) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v;
    switch (op) {
      case 0: v = a[i] - b[i]; break;
      case 1: v = a[i] + b[i]; break;
      case 2: v = a[i] * b[i]; break;
      case 3: v = a[i] * scalar; break;
      default: v = scalar; break;
    }
    out[i] = v;
  }
}

SN_EXPORT int sn_ew_f32(const float *a, const float *b, float *out, long n, int op, float scalar, sn_stream_t stream) {
  SN_REQUIRE(out && n > 0 && op >= 0 && op <= 4, "sn_ew_f32: bad arguments");
  SN_REQUIRE(op == 4 || a, "sn_ew_f32: null a");
  SN_REQUIRE(op > 2 || b, "sn_ew_f32: null b");
  hipLaunchKernelGGL(ew_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, sn_stream(stream), a, b, out, n, op, scalar);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// bias gradient: db[c] += sum over rows of dy[r][c]   (dy fp16 or fp32, row stride ld)
// Row blocks write partial sums [row block][C]; bias_grad_finish_kernel adds them to db in block order: no atomics, the
// same bits every run.  Without scratch one block per 64 channels walks all rows and adds its sum to db itself.
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T *__restrict__ dy, float *__restrict__ db, float *__restrict__ part,
                                                        long rows, int C, int ld, long rows_per_block) {
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rl; r < r1; r += 4) s += (float)dy[r * ld + c];
  __shared__ float red[4][64];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (part) part[(size_t)blockIdx.y * C + c] = t;
    else db[c] += t;          // single row block: this thread is the only writer of db[c]
  }
}

// the same reduction with 16-byte loads: a thread owns 8 consecutive channels (fp16 dy, a pitch that is a multiple of 8 and covers the
// last chunk -- the padded gradients of the narrow heads qualify): a wave reads 1 KB
// of a row per instruction instead of 128 B, four rows in flight per thread (the 2-byte form ran at 1.2 TB/s: 17 us for the RPN's
// 21 MB gradient)
__global__ __launch_bounds__(256) void bias_grad_vec8_kernel(const half_t *__restrict__ dy, float *__restrict__ db, float *__restrict__ part,
                                                             long rows, int C, int ld, long rows_per_block) {
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    long r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      half8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8 *>(dy + (r + 4 * u) * ld + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v[u][j];
    }
    for (; r < r1; r += 4) {
      const half8 v = *reinterpret_cast<const half8 *>(dy + r * ld + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
  }
  __shared__ float red[4][64][9];
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][lane][j] = s[j];
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (c + j >= C) break;          // (a last chunk that reaches into the row's padding: C = 42 in a pitch of 48)
      const float t = (red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j]);
      if (part) part[(size_t)blockIdx.y * C + c + j] = t;
      else db[c + j] += t;
    }
  }
}

__global__ __launch_bounds__(256) void bias_grad_finish_kernel(const float *__restrict__ part, int nblk, int C, float *__restrict__ db) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  // the row blocks are added in block order (fixed bits), but their loads are independent: eight in flight instead of a chain
  // of nblk dependent L2 round trips (13.5 us for 96 blocks)
  float s = 0.f;
  int k = 0;
  for (; k + 8 <= nblk; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * C + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < nblk; ++k) s += part[(size_t)k * C + c];
  db[c] += s;
}

static int bias_grad_blocks(long rows, long *rows_per_block) {
  int by = (int)((rows + 255) / 256);
  if (by > 96) by = 96;      // the finish kernel walks the row blocks serially per channel
  if (by < 1) by = 1;
  *rows_per_block = (rows + by - 1) / by;
  return (int)((rows + *rows_per_block - 1) / *rows_per_block);
}

SN_EXPORT size_t sn_bias_grad_workspace_bytes(long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  long rpb;
  int by = bias_grad_blocks(rows, &rpb);
  if (by > 1) by = (int)std::max<long>(by, std::min<long>(128, (rows + 63) / 64));      // (the 16-byte form uses up to 128 row blocks)
  return by > 1 ? sn_align(sizeof(float) * (size_t)by * C) : 0;
}

SN_EXPORT int sn_bias_grad(const void *dy, float *db, long rows, int C, int ld, int dtype, void *ws, size_t ws_bytes,
                           sn_stream_t stream) {
  SN_REQUIRE(dy && db && rows > 0 && C > 0, "sn_bias_grad: bad arguments");
  long rpb;
  int by = bias_grad_blocks(rows, &rpb);
  const bool vec8 = dtype == 0 && ld % 8 == 0 && sn_div_up(C, 8) * 8 <= ld && ((uintptr_t)dy % 16) == 0;
  if (vec8 && by > 1 && sn_div_up(C, 512) * by < 256) {        // one block spans 512 channels: more row blocks to fill the CUs
    by = (int)std::min<long>(std::min<long>(128, (rows + 63) / 64), by * 4L);      // (256 blocks: the ordered finish 5 -> 8.7 us)
    if (ws && ws_bytes < sizeof(float) * (size_t)by * C) by = bias_grad_blocks(rows, &rpb);
    rpb = (rows + by - 1) / by;
  }
  float *part = nullptr;
  if (by > 1) {
    if (ws && ws_bytes >= sizeof(float) * (size_t)by * C) part = (float *)ws;
    else { by = 1; rpb = rows; }       // no scratch: one (slow) owner per channel, still deterministic
  }
  dim3 grid(sn_div_up(C, 64), by);
  hipStream_t s = sn_stream(stream);
  if (vec8)
    hipLaunchKernelGGL(bias_grad_vec8_kernel, dim3(sn_div_up(C, 512), by), dim3(256), 0, s, (const half_t *)dy, db, part, rows, C, ld, rpb);
  else if (dtype == 0)
    hipLaunchKernelGGL((bias_grad_kernel<half_t>), grid, dim3(256), 0, s, (const half_t *)dy, db, part, rows, C, ld, rpb);
  else
    hipLaunchKernelGGL((bias_grad_kernel<float>), grid, dim3(256), 0, s, (const float *)dy, db, part, rows, C, ld, rpb);
  SN_CHECK_LAUNCH();
  if (part) {
    hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(sn_div_up(C, 256)), dim3(256), 0, s, (const float *)part, by, C, db);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}
