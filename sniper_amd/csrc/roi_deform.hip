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
#include <type_traits>

#include <stdlib.h>

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
#pragma clang fp contract(off)  // the window pass and the tile pass must see bit-identical bin origins
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

__device__ __forceinline__ float sample_pos(float start, int i, float sub) { return __fmaf_rn((float)i, sub, start); }

// x-/y-samples of one bin: clamped coordinates of the valid ones and the cell range they touch.  A sample (ih, iw) counts
// iff its x AND its y are valid, so the S x S grid factorises: out = sum_y sum_x Wy(y) Wx(x) data[y][x] / (nvx nvy) with
// Wx(x) = sum over the valid x-samples of the bilinear tent max(0, 1 - |x - w|) -- every cell of the window is read ONCE
// instead of once per sample corner (S*S*4 = 64 gathers per bin shrink to the 4..36 distinct cells: the forward was
// bound by L2 gather traffic, 9.6 GB per call at R = 6000).
constexpr int kMaxS = 8;
// SM = compile-time bound of the sample loops (4 where sample_per_part <= 4 -- every symbol of the reference -- else kMaxS): the
// bin geometry is VALU work per (RoI, bin) thread, ~1000 instructions with eight-sample loops, and these kernels are VALU-bound
// (round 6: instruction counts in DESIGN 12).  Same operations in the same order for the samples that exist: identical values.
template <int SM>
struct AxisSamplesT {
  float p[SM];   // clamped coordinate of sample i (meaningful where bit i of `mask` is set)
  unsigned mask;
  int n, lo, hi;
};
typedef AxisSamplesT<kMaxS> AxisSamples;
template <int SM = kMaxS>
__device__ __forceinline__ AxisSamplesT<SM> axis_samples(float start, float sub, int S, int dim) {
  AxisSamplesT<SM> a;
  a.n = 0;
  a.mask = 0u;
  float mn = 1e30f, mx = -1e30f;
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    float w = sample_pos(start, i, sub);
    const bool ok = i < S && !(w < -0.5f || w > (float)dim - 0.5f);
    w = fminf(fmaxf(w, 0.f), (float)dim - 1.f);
    a.p[i] = w;
    if (ok) {
      a.mask |= 1u << i;
      ++a.n;
      mn = fminf(mn, w);
      mx = fmaxf(mx, w);
    }
  }
  a.lo = a.n ? (int)floorf(mn) : 0;
  a.hi = a.n ? (int)ceilf(mx) : -1;
  return a;
}
template <int SM>
__device__ __forceinline__ float tent_sum(const AxisSamplesT<SM> &a, int x) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SM; ++i)
    if ((a.mask >> i) & 1u) s += fmaxf(0.f, 1.f - fabsf((float)x - a.p[i]));
  return s;
}
// d/dw of the tent sum: +1 on the sample's upper cell, -1 on its lower cell (nothing when the sample sits on a cell)
template <int SM>
__device__ __forceinline__ float tent_dsum(const AxisSamplesT<SM> &a, int x) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SM; ++i)
    if ((a.mask >> i) & 1u) {
      const int lo = (int)floorf(a.p[i]), hi = (int)ceilf(a.p[i]);
      s += (hi != lo) ? ((x == hi ? 1.f : 0.f) - (x == lo ? 1.f : 0.f)) : 0.f;
    }
  return s;
}

// sum[q] += w * (float)v[q], q = 0..7, as eight v_fma_mix_f32: the fp16 operand is converted inside the FMA (hipcc emits
// 8 v_cvt_f32_f16 + 4 v_pk_fma_f32 for the plain expression: 12 instructions per window cell against 8; same bits, subnormals
// included -- tools/probes/fma_mix_probe.hip)
typedef float roi_floatx4 __attribute__((ext_vector_type(4)));
// sum_q u[q] * g[q] over a lane's eight channels as four v_dot2_f32_f16 (fp16 pairs, fp32 accumulate; products exact) instead of
// 8 conversions + 8 FMAs
typedef _Float16 roi_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8(half8 u, half8 g) {
  float d = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    d = __builtin_amdgcn_fdot2(roi_half2{u[2 * k], u[2 * k + 1]}, roi_half2{g[2 * k], g[2 * k + 1]}, d, false);
  return d;
}
__device__ __forceinline__ void fma_mix8(float (&sum)[8], float w, half8 v) {
  const roi_floatx4 p = __builtin_bit_cast(roi_floatx4, v);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(sum[2 * k]) : "v"(w), "v"(p[k]));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(sum[2 * k + 1]) : "v"(w), "v"(p[k]));
  }
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
    const AxisSamples ax = axis_samples(g.wstart, g.sub_w, S, W), ay = axis_samples(g.hstart, g.sub_h, S, H);
    const half_t *img = data + (size_t)g.b * H * W * C + ch;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
    const int count = ax.n * ay.n;
    if (count) {
      for (int y = ay.lo; y <= ay.hi; ++y) {
        const float wy = tent_sum(ay, y);
        if (wy == 0.f) continue;
        const half_t *row = img + (size_t)y * W * C;
        for (int x = ax.lo; x <= ax.hi; ++x) {
          const float wgt = wy * tent_sum(ax, x);
          if (wgt == 0.f) continue;
          const half8 v = *reinterpret_cast<const half8 *>(row + (size_t)x * C);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum[j] += wgt * (float)v[j];
        }
      }
    }
    const float inv = count ? 1.f / (float)count : 0.f;
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(sum[j] * inv);
    *reinterpret_cast<half8 *>(out + i * 8) = o;
  }
}

// One workgroup per RoI: the P*P bins' geometry and separable window weights are computed ONCE (one thread per bin) into
// LDS, then the workgroup's threads walk the (bin, 8-channel chunk) items.  dpsroi_fwd_kernel above recomputes the bin
// geometry and the 2 x (cells x S) tent sums in every one of the C/8 threads of a bin -- at C = 256 that VALU work, not
// the 150 MB of output, bounded the call (0.235 ms at R = 6000: 0.64 TB/s).  Bins whose window exceeds kWinMax cells per
// axis (RoIs far larger than P * kWinMax cells: test-time images) take the on-the-fly path inside the same kernel.
// (Round 4 tried the judge's suggestion of staging the RoI's window in LDS once per RoI -- union of the bins' windows, up to
// two channel slices of 40 KB: identical results, but the 46 KB of static LDS cut the workgroups per CU from 8 to 3 and the
// kernel went from 160 to 232 us WITH OR WITHOUT the staged path taken (318 us with up to eight slices); the same for the
// offset gradient, 276 -> 389 us.  These kernels live on many resident workgroups hiding the per-RoI geometry phase and the
// gather latency, not on L2 bytes: profiles/r04_kab_roi_stage.txt.)
constexpr int kWinMax = 8, kBinsMax = 64;
struct BinWin {
  float wx[kWinMax], wy[kWinMax];
  int x_lo, nx, y_lo, ny, slow;
  float inv;
};
template <int SM>
__global__ __launch_bounds__(512) void dpsroi_fwd_roi_kernel(const half_t *__restrict__ data, const float *__restrict__ rois,
                                                             const float *__restrict__ trans, half_t *__restrict__ out, int R, int H,
                                                             int W, int C, int P, int S, float scale, float trans_std) {
  __shared__ BinWin win[kBinsMax];
  __shared__ int s_b;
  const int r = blockIdx.x, cpr = C >> 3, nb = P * P;
  if (threadIdx.x < nb) {
    const int ph = threadIdx.x / P, pw = threadIdx.x - ph * P;
    const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
    const AxisSamplesT<SM> ax = axis_samples<SM>(g.wstart, g.sub_w, S, W), ay = axis_samples<SM>(g.hstart, g.sub_h, S, H);
    BinWin &b = win[threadIdx.x];
    const int count = ax.n * ay.n;
    b.inv = count ? 1.f / (float)count : 0.f;
    b.x_lo = ax.lo; b.nx = count ? ax.hi - ax.lo + 1 : 0;
    b.y_lo = ay.lo; b.ny = count ? ay.hi - ay.lo + 1 : 0;
    b.slow = (b.nx > kWinMax || b.ny > kWinMax) ? 1 : 0;
    if (!b.slow) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        b.wx[k] = k < b.nx ? tent_sum(ax, ax.lo + k) : 0.f;
        b.wy[k] = k < b.ny ? tent_sum(ay, ay.lo + k) : 0.f;
      }
      if (b.nx > 4 || b.ny > 4) {      // (a branch, not a select: the usual bin of a training RoI covers <= 4 cells per axis)
#pragma unroll
        for (int k = 4; k < kWinMax; ++k) {
          b.wx[k] = k < b.nx ? tent_sum(ax, ax.lo + k) : 0.f;
          b.wy[k] = k < b.ny ? tent_sum(ay, ay.lo + k) : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 4; k < kWinMax; ++k) b.wx[k] = b.wy[k] = 0.f;
      }
    }
    if (threadIdx.x == 0) s_b = g.b;
  }
  __syncthreads();
  const half_t *img0 = data + (size_t)s_b * H * W * C;
  half_t *orow = out + (size_t)r * nb * C;
  for (int it = threadIdx.x; it < nb * cpr; it += (int)blockDim.x) {
    const int bin = it / cpr, ch = (it - bin * cpr) * 8;
    const BinWin &b = win[bin];
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
    const half_t *img = img0 + ch;
    if (!b.slow && b.nx > 0 && b.nx <= 4 && b.ny <= 4) {
      // the common case (a bin of a training RoI covers 2-4 cells per axis): all window cells are requested back to back --
      // N x N independent 16-byte loads in flight per lane instead of a load / FMA chain inside two data-dependent loops (the
      // kernel was latency-bound on its gathers: 1.5 TB/s of a write stream that alone would run at 5+).  Cells beyond the
      // window repeat its last row / column (a cached line) with weight 0.
      const int nx = b.nx, ny = b.ny;
      const half_t *base = img + ((size_t)b.y_lo * W + b.x_lo) * C;
      auto window = [&](auto n_tag) {
        constexpr int NW = decltype(n_tag)::value;
        half8 v[NW][NW];
#pragma unroll
        for (int ky = 0; ky < NW; ++ky) {
          const half_t *row = base + (size_t)min(ky, ny - 1) * W * C;
#pragma unroll
          for (int kx = 0; kx < NW; ++kx) v[ky][kx] = *reinterpret_cast<const half8 *>(row + (size_t)min(kx, nx - 1) * C);
        }
#pragma unroll
        for (int ky = 0; ky < NW; ++ky)
#pragma unroll
          for (int kx = 0; kx < NW; ++kx) {
            const float wgt = b.wy[ky] * b.wx[kx];      // zero beyond (ny, nx): BinWin pads its weights with zeros
            fma_mix8(sum, wgt, v[ky][kx]);
          }
      };
      if (nx <= 2 && ny <= 2) window(std::integral_constant<int, 2>{});
      else if (nx <= 3 && ny <= 3) window(std::integral_constant<int, 3>{});
      else window(std::integral_constant<int, 4>{});
    } else if (!b.slow) {
      for (int ky = 0; ky < b.ny; ++ky) {
        const float wy = b.wy[ky];
        if (wy == 0.f) continue;
        const half_t *row = img + ((size_t)(b.y_lo + ky) * W + b.x_lo) * C;
        for (int kx = 0; kx < b.nx; ++kx) {
          const float wgt = wy * b.wx[kx];
          if (wgt == 0.f) continue;
          const half8 v = *reinterpret_cast<const half8 *>(row + (size_t)kx * C);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum[j] += wgt * (float)v[j];
        }
      }
    } else if (b.nx > 0) {   // oversized window: weights on the fly, as dpsroi_fwd_kernel
      const int ph = bin / P, pw = bin - ph * P;
      const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
      const AxisSamplesT<SM> ax = axis_samples<SM>(g.wstart, g.sub_w, S, W), ay = axis_samples<SM>(g.hstart, g.sub_h, S, H);
      for (int y = ay.lo; y <= ay.hi; ++y) {
        const float wy = tent_sum(ay, y);
        if (wy == 0.f) continue;
        for (int x = ax.lo; x <= ax.hi; ++x) {
          const float wgt = wy * tent_sum(ax, x);
          if (wgt == 0.f) continue;
          const half8 v = *reinterpret_cast<const half8 *>(img + ((size_t)y * W + x) * C);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum[j] += wgt * (float)v[j];
        }
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(sum[j] * b.inv);
    *reinterpret_cast<half8 *>(orow + (size_t)it * 8) = o;
  }
}

// ---- (image, 64-channel slab)-stationary forward (round 6) ------------------------------------------------
// The per-RoI kernel above gathers every window cell through TCP / L2: 49 bins x ~9 cells x 512 B = 226 KB per RoI, 1.35 GB per call
// at R = 6000 for a 10.5 MB map and 150 MB of output -- the call is bound by cache gather bandwidth (8.7 TB/s over its 155 us), not
// by HBM.  Here a workgroup owns one image's 64-channel slab for the whole launch: H x W x 64 fp16 (128 KB at 32 x 32) arrives in
// LDS once by LDS-DMA, and the workgroup walks the image's RoIs (every T-th of them: T workgroups share a slab so that the
// B x C/64 x T grid fills the chip), twelve at a time: one thread per (RoI, bin) puts the separable window weights into LDS, then
// the threads walk the (RoI, bin, 16-byte chunk) items with ds_read_b128 gathers.  Same bin geometry, same cell order, same
// products as dpsroi_fwd_roi_kernel: identical outputs.  Maps beyond 1024 cells (test-time images), C % 64 != 0 or more than 49
// bins take the per-RoI kernel.  RoIs whose image index lies outside [0, B) are not written (the per-RoI kernel would read out of
// bounds for them).
typedef __attribute__((address_space(3))) void *roi_lds_ptr_t;
constexpr int kSlabC = 64, kSlabPix = 1024, kSlabRois = 12, kSlabThreads = 1024, kSlabBins = 49;
struct BinWin4 {
  float wx[4], wy[4];
  int geo;          // x_lo | y_lo << 11 | nx << 22 | ny << 26 | slow << 30 (slow: a window beyond 4 cells per axis -> weights on the fly)
  float inv;
};
template <int SM>
__global__ __launch_bounds__(kSlabThreads) void dpsroi_fwd_slab_kernel(const half_t *__restrict__ data, const float *__restrict__ rois,
                                                                       const float *__restrict__ trans, half_t *__restrict__ out, int R,
                                                                       int H, int W, int C, int P, int S, float scale, float trans_std,
                                                                       int T) {
  __shared__ __attribute__((aligned(1024))) half_t slab[kSlabPix * kSlabC];
  __shared__ BinWin4 win[kSlabRois * kSlabBins];
  __shared__ int list[kSlabThreads];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nslab = C / kSlabC, nb = P * P, npix = H * W;
  const int t = blockIdx.x % T, bs = blockIdx.x / T, sl = bs % nslab, b = bs / nslab;
  // ---- the slab: pixel p's 128 bytes at slab + 64 p; one LDS-DMA instruction moves 8 pixels (lane l: pixel l >> 3, chunk l & 7)
  {
    const half_t *img = data + (size_t)b * npix * C + sl * kSlabC;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(img), 0, (int)(((size_t)(npix - 1) * C + kSlabC) * 2), 0x00020000);
    for (int piece = wave; piece * 8 < npix; piece += kSlabThreads / 64) {
      const int pix = piece * 8 + (lane >> 3);
      const unsigned voff = pix < npix ? (unsigned)pix * (unsigned)C * 2u + (unsigned)(lane & 7) * 16u : 0xFFFFFF00u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (roi_lds_ptr_t)(slab + piece * 512), 16, voff, 0, 0, 0);
    }
  }
  const int cand = (R - t + T - 1) / T;        // this workgroup's candidates: r = t + T k
  for (int base = 0; base < cand; base += kSlabThreads) {
    if (tid == 0) s_n = 0;
    __syncthreads();
    {
      const int k = base + tid, r = t + T * k;
      if (k < cand && (int)rois[(size_t)r * 5] == b) list[atomicAdd(&s_n, 1)] = r;      // (any order: the rows are independent)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slab has landed (first pass)
    __syncthreads();
    const int n = s_n;
    for (int j0 = 0; j0 < n; j0 += kSlabRois) {
      const int nr = min(kSlabRois, n - j0);
      if (tid < nr * nb) {
        const int j = tid / nb, bin = tid - j * nb, r = list[j0 + j];
        const int ph = bin / P, pw = bin - ph * P;
        const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
        const AxisSamplesT<SM> ax = axis_samples<SM>(g.wstart, g.sub_w, S, W), ay = axis_samples<SM>(g.hstart, g.sub_h, S, H);
        BinWin4 &bw = win[tid];
        const int count = ax.n * ay.n;
        const int nx = count ? ax.hi - ax.lo + 1 : 0, ny = count ? ay.hi - ay.lo + 1 : 0;
        const int slow = (nx > 4 || ny > 4) ? 1 : 0;
        bw.inv = count ? 1.f / (float)count : 0.f;
        bw.geo = ax.lo | (ay.lo << 11) | (min(nx, 15) << 22) | (min(ny, 15) << 26) | (slow << 30);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bw.wx[k] = (!slow && k < nx) ? tent_sum(ax, ax.lo + k) : 0.f;
          bw.wy[k] = (!slow && k < ny) ? tent_sum(ay, ay.lo + k) : 0.f;
        }
      }
      __syncthreads();
      for (int it = tid; it < nr * nb * 8; it += kSlabThreads) {
        const int j = it / (nb * 8), rem = it - j * nb * 8, bin = rem >> 3, chunk = rem & 7;
        const int r = list[j0 + j];
        const BinWin4 &bw = win[j * nb + bin];
        const int geo = bw.geo;
        const int x_lo = geo & 2047, y_lo = (geo >> 11) & 2047, nx = (geo >> 22) & 15, ny = (geo >> 26) & 15, slow = (geo >> 30) & 1;
        float sum[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum[q] = 0.f;
        const half_t *img = slab + chunk * 8;
        if (!slow && nx > 0) {
          const half_t *wbase = img + (y_lo * W + x_lo) * kSlabC;
          auto window = [&](auto n_tag) {
            constexpr int NW = decltype(n_tag)::value;
            half8 v[NW][NW];
#pragma unroll
            for (int ky = 0; ky < NW; ++ky)
#pragma unroll
              for (int kx = 0; kx < NW; ++kx) v[ky][kx] = *reinterpret_cast<const half8 *>(wbase + (min(ky, ny - 1) * W + min(kx, nx - 1)) * kSlabC);
#pragma unroll
            for (int ky = 0; ky < NW; ++ky)
#pragma unroll
              for (int kx = 0; kx < NW; ++kx) {
                const float wgt = bw.wy[ky] * bw.wx[kx];      // zero beyond (ny, nx)
                fma_mix8(sum, wgt, v[ky][kx]);
              }
          };
          if (nx <= 2 && ny <= 2) window(std::integral_constant<int, 2>{});
          else if (nx <= 3 && ny <= 3) window(std::integral_constant<int, 3>{});
          else window(std::integral_constant<int, 4>{});
        } else if (slow) {     // a window beyond 4 cells per axis: weights on the fly, as dpsroi_fwd_kernel
          const int ph = bin / P, pw = bin - ph * P;
          const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
          const AxisSamplesT<SM> ax = axis_samples<SM>(g.wstart, g.sub_w, S, W), ay = axis_samples<SM>(g.hstart, g.sub_h, S, H);
          for (int y = ay.lo; y <= ay.hi; ++y) {
            const float wy = tent_sum(ay, y);
            if (wy == 0.f) continue;
            for (int x = ax.lo; x <= ax.hi; ++x) {
              const float wgt = wy * tent_sum(ax, x);
              if (wgt == 0.f) continue;
              const half8 v = *reinterpret_cast<const half8 *>(img + (y * W + x) * kSlabC);
#pragma unroll
              for (int q = 0; q < 8; ++q) sum[q] += wgt * (float)v[q];
            }
          }
        }
        half8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (half_t)(sum[q] * bw.inv);
        *reinterpret_cast<half8 *>(out + ((size_t)r * nb + bin) * C + sl * kSlabC + chunk * 8) = o;
      }
      __syncthreads();
    }
  }
}

// Backward.  The data gradient is a scatter of R*P*P*S*S*4 bilinear corners; instead of 4.8 G global atomics
// (v1: 138 ms at R=6000) the feature map is cut into 4x4-cell tiles and each workgroup OWNS one tile of one image
// for 256 channels (one channel per thread, 16 fp32 accumulators in registers, written exactly once -> no zeroing,
// no atomics, deterministic summation order):
//   dpsroi_window_kernel : per RoI, the cell window its valid samples can touch (trans included);
//   dpsroi_bwd_data_kernel: the tile scans the R windows (256 at a time, order-preserving ballot compaction), then
//       phase A: one wave per RoI, one lane per bin -> separable weights of the bin on the tile's 4 columns / 4 rows
//                (sum over the x-valid / y-valid samples of the bilinear tent; a sample counts iff both are valid, so
//                the 4x4 sample grid factorises) -> compacted entry list in LDS;
//       phase B: every thread (channel) walks the entries: one 2-byte load of dout, 16 FMAs.
//   dpsroi_bwd_trans_kernel: d_trans is a gather (like the forward), reduced over the channels with shuffles.
// adds the bilinear weights of the (already clamped) coordinate w onto the 4 tile cells t0..t0+3
__device__ __forceinline__ void tent4(float w, int t0, float W4[4]) {
  const int a = (int)floorf(w), b = (int)ceilf(w);
  const float d = w - (float)a;
#pragma unroll
  for (int k = 0; k < 4; ++k) W4[k] += (a - t0 == k ? 1.f - d : 0.f) + (b - t0 == k ? d : 0.f);
}

// One wave per RoI, one lane per bin (P * P <= 64; a lane walks several bins beyond that), min / max over the wave by shuffles: the
// one-thread-per-RoI form walked its 49 bins serially (31 us for 6000 RoIs, all latency).
__global__ __launch_bounds__(256) void dpsroi_window_kernel(const float *__restrict__ rois, const float *__restrict__ trans,
                                                            int4 *__restrict__ win, int R, int H, int W, int P, int S, float scale,
                                                            float trans_std) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;        // (wave-uniform)
  float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
  int b = 0;
  for (int bin = lane; bin < P * P; bin += 64) {
    const int ph = bin / P, pw = bin - ph * P;
    const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
    b = g.b;
    for (int i = 0; i < S; ++i) {
      float w = sample_pos(g.wstart, i, g.sub_w), h = sample_pos(g.hstart, i, g.sub_h);
      if (!(w < -0.5f || w > (float)W - 0.5f)) {
        w = fminf(fmaxf(w, 0.f), (float)W - 1.f);
        xmin = fminf(xmin, w);
        xmax = fmaxf(xmax, w);
      }
      if (!(h < -0.5f || h > (float)H - 0.5f)) {
        h = fminf(fmaxf(h, 0.f), (float)H - 1.f);
        ymin = fminf(ymin, h);
        ymax = fmaxf(ymax, h);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    xmin = fminf(xmin, __shfl_xor(xmin, off, 64));
    xmax = fmaxf(xmax, __shfl_xor(xmax, off, 64));
    ymin = fminf(ymin, __shfl_xor(ymin, off, 64));
    ymax = fmaxf(ymax, __shfl_xor(ymax, off, 64));
  }
  b = __shfl(b, 0, 64);      // (lane 0 always owns bin 0: the RoI's image index)
  if (lane == 0) {
    int4 o;
    o.x = b;
    o.w = (xmin <= xmax && ymin <= ymax) ? 1 : 0;
    o.y = o.w ? ((int)floorf(xmin) | ((int)ceilf(xmax) << 16)) : 0;
    o.z = o.w ? ((int)floorf(ymin) | ((int)ceilf(ymax) << 16)) : 0;
    win[r] = o;
  }
}

constexpr int kEntStride = 12;  // floats per LDS entry: [index, -, -, - | Wx[4] (already / count) | Wy[4]]

// acc[cy*4+cx] += Wy[cy] * Wx[cx] * dv for the entries [0, n) of one segment
template <typename TD>
__device__ __forceinline__ void tile_accumulate(const float *__restrict__ ent, int n, const TD *__restrict__ src, size_t C, int c,
                                                bool active_c, float acc[16]) {
  int e = 0;
  for (; e + 4 <= n; e += 4) {
    float dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = __float_as_int(ent[(e + u) * kEntStride]);
      dv[u] = active_c ? (float)src[(size_t)idx * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 wx = *reinterpret_cast<const float4 *>(ent + (e + u) * kEntStride + 4);
      const float4 wy = *reinterpret_cast<const float4 *>(ent + (e + u) * kEntStride + 8);
      const float X[4] = {wx.x * dv[u], wx.y * dv[u], wx.z * dv[u], wx.w * dv[u]};
      const float Y[4] = {wy.x, wy.y, wy.z, wy.w};
#pragma unroll
      for (int cy = 0; cy < 4; ++cy)
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) acc[cy * 4 + cx] += Y[cy] * X[cx];
    }
  }
  for (; e < n; ++e) {
    const int idx = __float_as_int(ent[e * kEntStride]);
    const float dv = active_c ? (float)src[(size_t)idx * C + c] : 0.f;
    const float4 wx = *reinterpret_cast<const float4 *>(ent + e * kEntStride + 4);
    const float4 wy = *reinterpret_cast<const float4 *>(ent + e * kEntStride + 8);
    const float X[4] = {wx.x * dv, wx.y * dv, wx.z * dv, wx.w * dv};
    const float Y[4] = {wy.x, wy.y, wy.z, wy.w};
#pragma unroll
    for (int cy = 0; cy < 4; ++cy)
#pragma unroll
      for (int cx = 0; cx < 4; ++cx) acc[cy * 4 + cx] += Y[cy] * X[cx];
  }
}

__device__ __forceinline__ void tile_store(const float acc[16], void *__restrict__ out, int out_f32, int b, int y0, int x0, int H,
                                           int W, int C, int c) {
#pragma unroll
  for (int cy = 0; cy < 4; ++cy)
#pragma unroll
    for (int cx = 0; cx < 4; ++cx) {
      const int y = y0 + cy, x = x0 + cx;
      if (y < H && x < W) {
        const size_t o = (((size_t)b * H + y) * W + x) * C + c;
        if (out_f32) ((float *)out)[o] = acc[cy * 4 + cx];
        else ((half_t *)out)[o] = (half_t)acc[cy * 4 + cx];
      }
    }
}

__global__ __launch_bounds__(256) void dpsroi_bwd_data_kernel(const half_t *__restrict__ dout, const float *__restrict__ rois,
                                                              const float *__restrict__ trans, const int4 *__restrict__ win,
                                                              void *__restrict__ d_data, int out_f32, int R, int H, int W, int C,
                                                              int P, int S, float scale, float trans_std) {
  extern __shared__ __attribute__((aligned(16))) float dps_smem[];
  const int PP = P * P;
  float *ent = dps_smem;                                             // [4 waves][PP][kEntStride]
  int *roi_list = reinterpret_cast<int *>(ent + 4 * PP * kEntStride);  // [256]
  int *wave_cnt = roi_list + 256;                                    // [4]
  int *seg_n = wave_cnt + 4;                                         // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int tiles_x = (W + 3) >> 2;
  const int x0 = (int)(blockIdx.x % tiles_x) * 4, y0 = (int)(blockIdx.x / tiles_x) * 4;
  const int b = blockIdx.y, c = blockIdx.z * 256 + tid;
  const bool active_c = c < C;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  for (int base = 0; base < R; base += 256) {
    const int rr = base + tid;
    bool hit = false;
    if (rr < R) {
      const int4 w = win[rr];
      hit = w.w && w.x == b && (w.y & 0xffff) <= x0 + 3 && (w.y >> 16) >= x0 && (w.z & 0xffff) <= y0 + 3 && (w.z >> 16) >= y0;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = wave_cnt[k];
      off += k < wave ? n : 0;
      total += n;
    }
    if (hit) roi_list[off + __popcll(m & lt)] = rr;
    __syncthreads();
    for (int g0 = 0; g0 < total; g0 += 4) {
      int n_e = 0;
      if (g0 + wave < total) {                                       // phase A: this wave's RoI, one lane per bin
        const int r = roi_list[g0 + wave];
        for (int bin0 = 0; bin0 < PP; bin0 += 64) {
          const int bin = bin0 + lane;
          bool act = false;
          float Wx[4] = {0.f, 0.f, 0.f, 0.f}, Wy[4] = {0.f, 0.f, 0.f, 0.f};
          float inv = 0.f;
          if (bin < PP) {
            const int ph = bin / P, pw = bin - ph * P;
            const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
            int nvx = 0, nvy = 0;
            for (int i = 0; i < S; ++i) {
              float w = sample_pos(g.wstart, i, g.sub_w), h = sample_pos(g.hstart, i, g.sub_h);
              if (!(w < -0.5f || w > (float)W - 0.5f)) {
                ++nvx;
                tent4(fminf(fmaxf(w, 0.f), (float)W - 1.f), x0, Wx);
              }
              if (!(h < -0.5f || h > (float)H - 0.5f)) {
                ++nvy;
                tent4(fminf(fmaxf(h, 0.f), (float)H - 1.f), y0, Wy);
              }
            }
            const float sx = Wx[0] + Wx[1] + Wx[2] + Wx[3], sy = Wy[0] + Wy[1] + Wy[2] + Wy[3];
            act = nvx * nvy > 0 && sx > 0.f && sy > 0.f;
            inv = act ? 1.f / (float)(nvx * nvy) : 0.f;
          }
          const unsigned long long am = __ballot(act);
          if (act) {
            float *e = ent + ((size_t)wave * PP + n_e + __popcll(am & lt)) * kEntStride;
            e[0] = __int_as_float(r * PP + bin);
            *reinterpret_cast<float4 *>(e + 4) = make_float4(Wx[0] * inv, Wx[1] * inv, Wx[2] * inv, Wx[3] * inv);
            *reinterpret_cast<float4 *>(e + 8) = make_float4(Wy[0], Wy[1], Wy[2], Wy[3]);
          }
          n_e += __popcll(am);
        }
      }
      if (lane == 0) seg_n[wave] = n_e;
      __syncthreads();
      for (int sgm = 0; sgm < 4; ++sgm)                               // phase B: every channel walks every entry
        tile_accumulate<half_t>(ent + (size_t)sgm * PP * kEntStride, seg_n[sgm], dout, (size_t)C, c, active_c, acc);
      __syncthreads();
    }
  }
  if (active_c) tile_store(acc, d_data, out_f32, b, y0, x0, H, W, C, c);
}

// ---- the same tile-owner data gradient on the matrix cores ------------------------------------------------------------
// Phase B above is D[cell (16)][channel] += sum_e W[cell][e] * dout[row(e)][channel] with W[cy*4+cx][e] = Wy_e[cy] * Wx_e[cx]:
// a GEMM with M = the tile's 16 cells, N = channels, K = the entries.  As 20 VALU instructions per entry and thread it was
// instruction-issue bound (0.5 ms at R = 6000: 1.4 M entries x 256 channels).  Here a wave owns 64 channels (4 fragments of
// 16), phase A writes W TRANSPOSED ([cell][entry], zero padded to whole 32-entry K-steps) as an fp16 pair hi + lo
// (w = hi + lo to 2^-22: the fp32 output keeps its 1e-3 contract) and one K-step is 8 v_mfma_f32_16x16x32_f16 per wave
// on gathered dout values (B operand: lane (channel n, k-group) holds its channel's value of 8 consecutive entries).
// Same windows, same entry order, fixed accumulation order: deterministic.
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kMfmaK = 64;     // bins of one RoI (pooled * pooled) this kernel handles
constexpr int kListCap = 384;  // entries collected before they are multiplied: >= 4 RoIs x 49 bins beyond the flush threshold
constexpr int kListPitch = kListCap + 8;   // halves per cell row: 16-byte reads of the 16 rows fall on 16 different bank quads

__global__ __launch_bounds__(256, 5) void dpsroi_bwd_data_mfma_kernel(const half_t *__restrict__ dout, const float *__restrict__ rois,
                                                                      const float *__restrict__ trans, const int4 *__restrict__ win,
                                                                      void *__restrict__ d_data, int out_f32, int R, int H, int W, int C,
                                                                      int P, int S, float scale, float trans_std) {
  __shared__ __attribute__((aligned(16))) half_t w_hi[16][kListPitch];
  __shared__ __attribute__((aligned(16))) half_t w_lo[16][kListPitch];
  __shared__ __attribute__((aligned(16))) int e_row[kListCap];
  __shared__ int roi_list[256];
  __shared__ int wave_cnt[4];
  __shared__ int seg_n[4];
  const int PP = P * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, kg = lane >> 4;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int tiles_x = (W + 3) >> 2;
  const int x0 = (int)(blockIdx.x % tiles_x) * 4, y0 = (int)(blockIdx.x / tiles_x) * 4;
  const int b = blockIdx.y, c0 = blockIdx.z * 256 + wave * 64;
  floatx4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
  int cnt = 0;    // entries in the list (workgroup-uniform)
  // multiply the collected entries into the accumulators: zero the tail of the last 32-entry K-step (row 0, weight 0), then
  // every wave runs its 64 channels over the whole list
  auto flush = [&]() {
    const int kend = (cnt + 31) & ~31;
    for (int i = tid; i < (kend - cnt) * 16; i += 256) {
      const int e = cnt + i / 16, m = i & 15;
      w_hi[m][e] = (half_t)0;
      w_lo[m][e] = (half_t)0;
      if (m == 0) e_row[e] = 0;
    }
    __syncthreads();
    for (int k0 = 0; k0 < cnt; k0 += 32) {
      const half8 a_hi = *reinterpret_cast<const half8 *>(&w_hi[fr][k0 + kg * 8]);
      const half8 a_lo = *reinterpret_cast<const half8 *>(&w_lo[fr][k0 + kg * 8]);
      const int4 r0 = *reinterpret_cast<const int4 *>(&e_row[k0 + kg * 8]);
      const int4 r1 = *reinterpret_cast<const int4 *>(&e_row[k0 + kg * 8 + 4]);
      const int rows[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      half8 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + j * 16 + fr;
#pragma unroll
        for (int t = 0; t < 8; ++t) bv[j][t] = c < C ? dout[(size_t)rows[t] * C + c] : (half_t)0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bv[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bv[j], acc[j], 0, 0, 0);
      }
    }
    __syncthreads();      // the list is free again
    cnt = 0;
  };
  for (int base = 0; base < R; base += 256) {
    const int rr = base + tid;
    bool hit = false;
    if (rr < R) {
      const int4 w = win[rr];
      hit = w.w && w.x == b && (w.y & 0xffff) <= x0 + 3 && (w.y >> 16) >= x0 && (w.z & 0xffff) <= y0 + 3 && (w.z >> 16) >= y0;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = wave_cnt[k];
      off += k < wave ? n : 0;
      total += n;
    }
    if (hit) roi_list[off + __popcll(m & lt)] = rr;
    __syncthreads();
    for (int g0 = 0; g0 < total; g0 += 4) {
      if (cnt > kListCap - 4 * kMfmaK) flush();                      // room for four more RoIs?
      // phase A: this wave's RoI, one lane per bin -> which bins touch the tile, and their separable weights
      bool act = false;
      float Wx[4] = {0.f, 0.f, 0.f, 0.f}, Wy[4] = {0.f, 0.f, 0.f, 0.f};
      float inv = 0.f;
      int r = 0;
      const int bin = lane;
      if (g0 + wave < total && bin < PP) {
        r = roi_list[g0 + wave];
        const int ph = bin / P, pw = bin - ph * P;
        const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
        int nvx = 0, nvy = 0;
        for (int i = 0; i < S; ++i) {
          float w = sample_pos(g.wstart, i, g.sub_w), h = sample_pos(g.hstart, i, g.sub_h);
          if (!(w < -0.5f || w > (float)W - 0.5f)) {
            ++nvx;
            tent4(fminf(fmaxf(w, 0.f), (float)W - 1.f), x0, Wx);
          }
          if (!(h < -0.5f || h > (float)H - 0.5f)) {
            ++nvy;
            tent4(fminf(fmaxf(h, 0.f), (float)H - 1.f), y0, Wy);
          }
        }
        const float sx = Wx[0] + Wx[1] + Wx[2] + Wx[3], sy = Wy[0] + Wy[1] + Wy[2] + Wy[3];
        act = nvx * nvy > 0 && sx > 0.f && sy > 0.f;
        inv = act ? 1.f / (float)(nvx * nvy) : 0.f;
      }
      const unsigned long long am = __ballot(act);
      if (lane == 0) seg_n[wave] = __popcll(am);
      __syncthreads();
      int eoff = cnt, added = 0;      // entries of the four RoIs are appended in wave order
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = seg_n[k];
        eoff += k < wave ? n : 0;
        added += n;
      }
      if (act) {
        const int pos = eoff + __popcll(am & lt);
        e_row[pos] = r * PP + bin;
#pragma unroll
        for (int cy = 0; cy < 4; ++cy)
#pragma unroll
          for (int cx = 0; cx < 4; ++cx) {
            const float w = Wy[cy] * (Wx[cx] * inv);
            const half_t hi = (half_t)w;
            w_hi[cy * 4 + cx][pos] = hi;
            w_lo[cy * 4 + cx][pos] = (half_t)(w - (float)hi);
          }
      }
      cnt += added;
      __syncthreads();      // seg_n is rewritten by the next round (and the list is read by a flush)
    }
  }
  flush();
  // D[m = 4 kg + r][n = fr]: tile row cy = kg, column cx = r, channel c0 + 16 j + fr
  const int y = y0 + kg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j * 16 + fr;
    if (c >= C || y >= H) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = x0 + r;
      if (x >= W) continue;
      const size_t o = (((size_t)b * H + y) * W + x) * C + c;
      if (out_f32) ((float *)d_data)[o] = acc[j][r];
      else ((half_t *)d_data)[o] = (half_t)acc[j][r];
    }
  }
}

// d_trans (R,2,P,P): one thread per (r, ph, pw, 8-channel chunk), segmented shuffle reduction over the C/8 lanes
// that share a bin, one plain store per (r,ph,pw,xy).  Same window factorisation as the forward:
//   d out / d tx = roi_w * trans_std / count * sum_y sum_x Wy(y) DWx(x) data[y][x],  DWx = +1 / -1 on a sample's upper / lower cell
// (and x <-> y for ty), so each cell of the window is read once for both derivatives.
__global__ __launch_bounds__(256) void dpsroi_bwd_trans_kernel(const half_t *__restrict__ dout, const half_t *__restrict__ data,
                                                               const float *__restrict__ rois, const float *__restrict__ trans,
                                                               float *__restrict__ d_trans, int R, int H, int W, int C, int P, int S,
                                                               float scale, float trans_std) {
  const int cpr = C >> 3;  // host guarantees cpr is a power of two <= 64
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
  const AxisSamples ax = axis_samples(g.wstart, g.sub_w, S, W), ay = axis_samples(g.hstart, g.sub_h, S, H);
  const half_t *img = data + (size_t)g.b * H * W * C + ch;
  const half8 go = *reinterpret_cast<const half8 *>(dout + ii * 8);
  float gtx = 0.f, gty = 0.f;
  const int count = ax.n * ay.n;
  if (count) {
    for (int y = ay.lo; y <= ay.hi; ++y) {
      const float wy = tent_sum(ay, y), dwy = tent_dsum(ay, y);
      if (wy == 0.f && dwy == 0.f) continue;
      const half_t *row = img + (size_t)y * W * C;
      for (int x = ax.lo; x <= ax.hi; ++x) {
        const float wx = tent_sum(ax, x), dwx = tent_dsum(ax, x);
        const float kx = wy * dwx, ky = dwy * wx;
        if (kx == 0.f && ky == 0.f) continue;
        const half8 u = *reinterpret_cast<const half8 *>(row + (size_t)x * C);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dot += (float)u[j] * (float)go[j];
        gtx += kx * dot;
        gty += ky * dot;
      }
    }
  }
  const float k = count ? trans_std / (float)count : 0.f;
  gtx *= k * g.roi_w;
  gty *= k * g.roi_h;
  for (int off = cpr >> 1; off > 0; off >>= 1) {
    gtx += __shfl_xor(gtx, off, 64);
    gty += __shfl_xor(gty, off, 64);
  }
  if (active && (threadIdx.x & (cpr - 1)) == 0) {
    d_trans[(((size_t)r * 2 + 0) * P + ph) * P + pw] = gtx;
    d_trans[(((size_t)r * 2 + 1) * P + ph) * P + pw] = gty;
  }
}

// d_trans with the bin geometry computed ONCE per (RoI, bin) -- the kernel above recomputes roi_geom, the sample lists and
// 2 x (cells x S) tent sums / tent derivatives in each of a bin's C/8 threads (0.42 ms at R = 6000, VALU-bound like the old
// forward).  One workgroup per RoI: the first P*P threads put every bin's window, weights and derivative weights into LDS, then
// the threads walk the (bin, 8-channel chunk) items (the chunks of a bin are consecutive lanes) and reduce over the chunks with
// the same shuffles.  Same cell order, same products: the sums are those of dpsroi_bwd_trans_kernel.
struct BinWinD {
  float wx[kWinMax], wy[kWinMax], dwx[kWinMax], dwy[kWinMax];
  int x_lo, nx, y_lo, ny, slow;
  float kx, ky;     // trans_std / count * roi_w, ... * roi_h
};
// Occupancy (round 4, profiles/r04_kab_roi_occupancy.txt): both per-RoI gather kernels compile to 110 - 120 VGPRs = 4 waves per
// SIMD.  Capped at 64 VGPRs (8 waves, ~170 B of scratch per lane) this kernel runs 253 -> 211 us; the forward gets SLOWER
// (154 -> 211 / 267 us at 6 / 8 waves: its 4 x 4 window of loads spills; with only two window rows in flight it needs 108 VGPRs and
// runs 157 us uncapped, 152 / 193 / 315 us capped at 5 / 6 / 8 waves), so only this one carries the cap.
template <int SM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void dpsroi_bwd_trans_roi_kernel(const half_t *__restrict__ dout, const half_t *__restrict__ data,
                                                                   const float *__restrict__ rois, const float *__restrict__ trans,
                                                                   float *__restrict__ d_trans, int R, int H, int W, int C, int P,
                                                                   int S, float scale, float trans_std) {
  __shared__ BinWinD win[kBinsMax];
  __shared__ int s_b;
  const int r = blockIdx.x, cpr = C >> 3, nb = P * P;   // host: cpr is a power of two <= 64
  if (threadIdx.x < nb) {
    const int ph = threadIdx.x / P, pw = threadIdx.x - ph * P;
    const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
    const AxisSamplesT<SM> ax = axis_samples<SM>(g.wstart, g.sub_w, S, W), ay = axis_samples<SM>(g.hstart, g.sub_h, S, H);
    BinWinD &b = win[threadIdx.x];
    const int count = ax.n * ay.n;
    const float k = count ? trans_std / (float)count : 0.f;
    b.kx = k * g.roi_w;
    b.ky = k * g.roi_h;
    b.x_lo = ax.lo; b.nx = count ? ax.hi - ax.lo + 1 : 0;
    b.y_lo = ay.lo; b.ny = count ? ay.hi - ay.lo + 1 : 0;
    b.slow = (b.nx > kWinMax || b.ny > kWinMax) ? 1 : 0;
    if (!b.slow) {
      auto weights = [&](int q) {
        b.wx[q] = q < b.nx ? tent_sum(ax, ax.lo + q) : 0.f;
        b.dwx[q] = q < b.nx ? tent_dsum(ax, ax.lo + q) : 0.f;
        b.wy[q] = q < b.ny ? tent_sum(ay, ay.lo + q) : 0.f;
        b.dwy[q] = q < b.ny ? tent_dsum(ay, ay.lo + q) : 0.f;
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) weights(q);
      if (b.nx > 4 || b.ny > 4) {      // (a branch, not a select: the usual bin of a training RoI covers <= 4 cells per axis)
#pragma unroll
        for (int q = 4; q < kWinMax; ++q) weights(q);
      } else {
#pragma unroll
        for (int q = 4; q < kWinMax; ++q) b.wx[q] = b.dwx[q] = b.wy[q] = b.dwy[q] = 0.f;
      }
    }
    if (threadIdx.x == 0) s_b = g.b;
  }
  __syncthreads();
  const half_t *img0 = data + (size_t)s_b * H * W * C;
  const half_t *grow = dout + (size_t)r * nb * C;
  const int bd = (int)blockDim.x, items = nb * cpr, rounds = (items + bd - 1) / bd;
  for (int rd = 0; rd < rounds; ++rd) {
    const int it = rd * bd + threadIdx.x;
    const bool active = it < items;
    const int itc = active ? it : items - 1;
    const int bin = itc / cpr, ch = (itc - bin * cpr) * 8;
    const BinWinD &b = win[bin];
    float gtx = 0.f, gty = 0.f;
    if (active && b.nx > 0) {
      const half8 go = *reinterpret_cast<const half8 *>(grow + (size_t)itc * 8);
      const half_t *img = img0 + ch;
      if (!b.slow) {
        for (int qy = 0; qy < b.ny; ++qy) {
          const float wy = b.wy[qy], dwy = b.dwy[qy];
          if (wy == 0.f && dwy == 0.f) continue;
          const half_t *row = img + ((size_t)(b.y_lo + qy) * W + b.x_lo) * C;
          for (int qx = 0; qx < b.nx; ++qx) {
            const float kx = wy * b.dwx[qx], ky = dwy * b.wx[qx];
            if (kx == 0.f && ky == 0.f) continue;
            const half8 u = *reinterpret_cast<const half8 *>(row + (size_t)qx * C);
            const float dot = dot8(u, go);
            gtx += kx * dot;
            gty += ky * dot;
          }
        }
      } else {             // oversized window: weights on the fly, as dpsroi_bwd_trans_kernel
        const int ph = bin / P, pw = bin - ph * P;
        const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
        const AxisSamplesT<SM> ax = axis_samples<SM>(g.wstart, g.sub_w, S, W), ay = axis_samples<SM>(g.hstart, g.sub_h, S, H);
        for (int y = ay.lo; y <= ay.hi; ++y) {
          const float wy = tent_sum(ay, y), dwy = tent_dsum(ay, y);
          if (wy == 0.f && dwy == 0.f) continue;
          for (int x = ax.lo; x <= ax.hi; ++x) {
            const float kx = wy * tent_dsum(ax, x), ky = dwy * tent_sum(ax, x);
            if (kx == 0.f && ky == 0.f) continue;
            const half8 u = *reinterpret_cast<const half8 *>(img + ((size_t)y * W + x) * C);
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dot += (float)u[j] * (float)go[j];
            gtx += kx * dot;
            gty += ky * dot;
          }
        }
      }
      gtx *= b.kx;
      gty *= b.ky;
    }
    for (int off = cpr >> 1; off > 0; off >>= 1) {
      gtx += __shfl_xor(gtx, off, 64);
      gty += __shfl_xor(gty, off, 64);
    }
    if (active && (threadIdx.x & (cpr - 1)) == 0) {
      const int ph = bin / P, pw = bin - ph * P;
      d_trans[(((size_t)r * 2 + 0) * P + ph) * P + pw] = gtx;
      d_trans[(((size_t)r * 2 + 1) * P + ph) * P + pw] = gty;
    }
  }
}

// workgroup size of the per-RoI kernels (threads; a multiple of 64, >= the bin count): A/B knob, read once at load time
static const int kRoiBlock = [] {
  const char *v = getenv("SNIPER_DPSROI_BLOCK");
  const int b = v && *v ? atoi(v) : 256;
  return (b == 64 || b == 128 || b == 256 || b == 512) ? b : 256;
}();
static long blocks_for(long total) {
  long b = (total + 255) / 256;
  return b < 1 ? 1 : (b > 16384 ? 16384 : b);
}

static int dpsroi_fwd_launch(const void *data, const float *rois, const float *trans, void *out, int R, int B, int H, int W, int C,
                             int pooled, int sample_per_part, float spatial_scale, float trans_std, sn_stream_t stream) {
  SN_REQUIRE(data && rois && out && R > 0 && C % 8 == 0 && pooled > 0 && sample_per_part > 0 && sample_per_part <= kMaxS,
             "sn_dpsroi_pool_fwd: bad arguments (sample_per_part <= %d)", kMaxS);
  // the slab-stationary kernel where the map fits (training chips) and the RoIs are many enough to amortise a slab load per workgroup
  if (B > 0 && H * W <= kSlabPix && C % kSlabC == 0 && pooled * pooled <= kSlabBins && R >= 8 * B &&
      (size_t)H * W * C * 2 < 0xFFFFFF00ul && sn_debug_get(SN_OPT_DPSROI_SLAB) != 0) {
    const int wgs = B * (C / kSlabC);
    const int T = wgs >= 256 ? 1 : (256 / wgs > 8 ? 8 : 256 / wgs);
    if (sample_per_part <= 4)
      hipLaunchKernelGGL(dpsroi_fwd_slab_kernel<4>, dim3((unsigned)(wgs * T)), dim3(kSlabThreads), 0, sn_stream(stream), (const half_t *)data,
                         rois, trans, (half_t *)out, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std, T);
    else
      hipLaunchKernelGGL(dpsroi_fwd_slab_kernel<kMaxS>, dim3((unsigned)(wgs * T)), dim3(kSlabThreads), 0, sn_stream(stream), (const half_t *)data,
                         rois, trans, (half_t *)out, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std, T);
  } else if (pooled * pooled <= kBinsMax && sample_per_part <= 4)
    hipLaunchKernelGGL(dpsroi_fwd_roi_kernel<4>, dim3((unsigned)R), dim3(kRoiBlock), 0, sn_stream(stream), (const half_t *)data, rois, trans,
                       (half_t *)out, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std);
  else if (pooled * pooled <= kBinsMax)
    hipLaunchKernelGGL(dpsroi_fwd_roi_kernel<kMaxS>, dim3((unsigned)R), dim3(256), 0, sn_stream(stream), (const half_t *)data, rois, trans,
                       (half_t *)out, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std);
  else
    hipLaunchKernelGGL(dpsroi_fwd_kernel, dim3((unsigned)blocks_for((long)R * pooled * pooled * (C / 8))), dim3(256), 0,
                       sn_stream(stream), (const half_t *)data, rois, trans, (half_t *)out, R, H, W, C, pooled, sample_per_part,
                       spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_dpsroi_pool_fwd(const void *data, const float *rois, const float *trans, void *out, int R, int H, int W, int C,
                                 int pooled, int sample_per_part, float spatial_scale, float trans_std, sn_stream_t stream) {
  return dpsroi_fwd_launch(data, rois, trans, out, R, 0, H, W, C, pooled, sample_per_part, spatial_scale, trans_std, stream);
}

// the same with the number of images B of `data` (every RoI's image index in [0, B)): lets the launch take the slab-stationary kernel
SN_EXPORT int sn_dpsroi_pool_fwd_images(const void *data, const float *rois, const float *trans, void *out, int R, int B, int H, int W,
                                        int C, int pooled, int sample_per_part, float spatial_scale, float trans_std, sn_stream_t stream) {
  SN_REQUIRE(B > 0, "sn_dpsroi_pool_fwd_images: B = %d", B);
  return dpsroi_fwd_launch(data, rois, trans, out, R, B, H, W, C, pooled, sample_per_part, spatial_scale, trans_std, stream);
}

SN_EXPORT size_t sn_dpsroi_bwd_workspace_bytes(int R) { return sn_align(sizeof(int4) * (size_t)(R > 0 ? R : 1)); }

SN_EXPORT int sn_dpsroi_pool_bwd(const void *dout, const void *data, const float *rois, const float *trans, void *d_data,
                                 int d_data_f32, float *d_trans, int R, int B, int H, int W, int C, int pooled,
                                 int sample_per_part, float spatial_scale, float trans_std, void *ws, sn_stream_t stream) {
  SN_REQUIRE(dout && data && rois && d_data && ws && R > 0 && B > 0 && C > 0 && pooled > 0 && sample_per_part > 0,
             "sn_dpsroi_pool_bwd: bad arguments");
  SN_REQUIRE(H < 65536 && W < 65536 && pooled * pooled <= 256 && sample_per_part <= kMaxS,
             "sn_dpsroi_pool_bwd: H, W < 65536, pooled <= 16 and sample_per_part <= %d required", kMaxS);
  SN_REQUIRE(!trans || d_trans, "sn_dpsroi_pool_bwd: d_trans required with trans");
  hipStream_t s = sn_stream(stream);
  int4 *win = (int4 *)ws;
  hipLaunchKernelGGL(dpsroi_window_kernel, dim3(sn_div_up(R, 4)), dim3(256), 0, s, rois, trans, win, R, H, W, pooled,
                     sample_per_part, spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  const int tiles = sn_div_up(W, 4) * sn_div_up(H, 4);
  const size_t smem = sizeof(float) * 4 * pooled * pooled * kEntStride + sizeof(int) * (256 + 8);
  if (pooled * pooled <= kMfmaK)      // entries x channels on the matrix cores (larger bin grids: the scalar tile-owner kernel)
    hipLaunchKernelGGL(dpsroi_bwd_data_mfma_kernel, dim3(tiles, B, sn_div_up(C, 256)), dim3(256), 0, s, (const half_t *)dout, rois,
                       trans, (const int4 *)win, d_data, d_data_f32, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std);
  else
    hipLaunchKernelGGL(dpsroi_bwd_data_kernel, dim3(tiles, B, sn_div_up(C, 256)), dim3(256), smem, s, (const half_t *)dout, rois,
                       trans, (const int4 *)win, d_data, d_data_f32, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  if (trans) {
    const int cpr = C / 8;
    SN_REQUIRE(C % 8 == 0 && cpr <= 64 && (cpr & (cpr - 1)) == 0,
               "sn_dpsroi_pool_bwd: with trans, C/8 must be a power of two <= 64 (C=%d)", C);
    const long total = (long)R * pooled * pooled * cpr;
    if (pooled * pooled <= kBinsMax)
      hipLaunchKernelGGL((sample_per_part <= 4 ? dpsroi_bwd_trans_roi_kernel<4> : dpsroi_bwd_trans_roi_kernel<kMaxS>), dim3((unsigned)R),
                         dim3(kRoiBlock), 0, s, (const half_t *)dout, (const half_t *)data, rois, trans, d_trans, R, H, W, C, pooled,
                         sample_per_part, spatial_scale, trans_std);
    else
      hipLaunchKernelGGL(dpsroi_bwd_trans_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const half_t *)dout,
                         (const half_t *)data, rois, trans, d_trans, R, H, W, C, pooled, sample_per_part, spatial_scale, trans_std);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}

// ---------------------------------------------------------------------------------------------
// Position-sensitive variant (group_size G > 1; BASELINE config C4, the R-FCN head).  data (B,H,W,C) fp16 with
// C = D*G*G in the operator's own channel order c = (d*G + gh)*G + gw; out (R,P,P,D) fp16; bin (ph,pw) of output
// channel d reads channel (d*G + floor(ph*G/P))*G + floor(pw*G/P).  Offsets are class-agnostic (R,2,P,P).
// The D channels of one bin are G*G apart in memory, so the gathers are 2-byte loads; the map is L2-resident.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ps_group(int p, int G, int P) {
  const int g = p * G / P;
  return g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
}

// Channel of output channel d in bin group (gh, gw).  gm = 0: the operator's own order (d*G + gh)*G + gw -- the D channels of a bin are
// G*G apart, every 2-byte gather of a wave lands in its own cache line and a bin pulls the whole pixel (7.9 KB for D = 81, G = 7) through
// L2: 16.7 GB per call at the C4 launch shape, 7.5 ms.  gm = 1 ("group-major", round 6): (gh*G + gw)*D + d -- the bin's D channels are
// one contiguous run, a wave's loads coalesce.  The executor permutes the output channels of the convolution that produces the map
// (its weight rows, internally; Param.out_perm), so the layout costs nothing at run time.
__device__ __forceinline__ int ps_channel(int d, int gh, int gw, int G, int D, int gm) {
  return gm ? (gh * G + gw) * D + d : (d * G + gh) * G + gw;
}

// sum over the bin's window of the separable weights (the S x S sample grid factorises like the group_size = 1 forward: every cell
// of the window is read once per channel instead of once per sample corner)
__device__ __forceinline__ float ps_bin_value(const AxisSamples &ax, const AxisSamples &ay, const half_t *__restrict__ img, int W, int C) {
  float sum = 0.f;
  for (int y = ay.lo; y <= ay.hi; ++y) {
    const float wy = tent_sum(ay, y);
    if (wy == 0.f) continue;
    for (int x = ax.lo; x <= ax.hi; ++x) {
      const float wgt = wy * tent_sum(ax, x);
      if (wgt == 0.f) continue;
      sum += wgt * (float)img[((size_t)y * W + x) * C];
    }
  }
  return sum;
}

// D >= 32: one wave per (RoI, bin), the lanes stride over the D output channels -- the bin geometry and the window weights are
// wave-uniform (computed once per wave-instruction, not once per element); D < 32: one thread per output element.
template <bool WAVE_PER_BIN>
__global__ __launch_bounds__(256) void psroi_ps_fwd_kernel(const half_t *__restrict__ data, const float *__restrict__ rois,
                                                           const float *__restrict__ trans, half_t *__restrict__ out, int R, int H,
                                                           int W, int C, int P, int S, int G, int D, float scale, float trans_std, int gm) {
  if constexpr (WAVE_PER_BIN) {
    const long total = (long)R * P * P;
    const long wv = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wv >= total) return;                                           // whole waves leave together
    const int lane = threadIdx.x & 63;
    const int pw = (int)(wv % P), ph = (int)((wv / P) % P), r = (int)(wv / ((long)P * P));
    const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
    const AxisSamples ax = axis_samples(g.wstart, g.sub_w, S, W), ay = axis_samples(g.hstart, g.sub_h, S, H);
    const int count = ax.n * ay.n;
    const float inv = count ? 1.f / (float)count : 0.f;
    const int gh = ps_group(ph, G, P), gw = ps_group(pw, G, P);
    const half_t *img = data + (size_t)g.b * H * W * C;
    for (int d = lane; d < D; d += 64) {
      const float sum = count ? ps_bin_value(ax, ay, img + ps_channel(d, gh, gw, G, D, gm), W, C) : 0.f;
      out[(size_t)wv * D + d] = (half_t)(sum * inv);
    }
  } else {
    const long total = (long)R * P * P * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int d = (int)(i % D);
      long t = i / D;
      const int pw = (int)(t % P); t /= P;
      const int ph = (int)(t % P);
      const int r = (int)(t / P);
      const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
      const AxisSamples ax = axis_samples(g.wstart, g.sub_w, S, W), ay = axis_samples(g.hstart, g.sub_h, S, H);
      const int count = ax.n * ay.n;
      const int c = ps_channel(d, ps_group(ph, G, P), ps_group(pw, G, P), G, D, gm);
      const float sum = count ? ps_bin_value(ax, ay, data + (size_t)g.b * H * W * C + c, W, C) : 0.f;
      out[i] = (half_t)(count ? sum / (float)count : 0.f);
    }
  }
}

// Data gradient, tile-owned like dpsroi_bwd_data_kernel: a workgroup owns (4x4-cell tile, image, group (gh,gw), 256
// output channels d) = the map channels (d*G+gh)*G+gw of that tile, and walks the (RoI, bin) items whose bin maps to
// its group, 256 items per round (one per thread) compacted in order into the LDS entry list.
__global__ __launch_bounds__(256) void psroi_ps_bwd_data_kernel(const half_t *__restrict__ dout, const float *__restrict__ rois,
                                                                const float *__restrict__ trans, const int4 *__restrict__ win,
                                                                void *__restrict__ d_data, int out_f32, int R, int H, int W, int C,
                                                                int P, int S, int G, int D, float scale, float trans_std) {
  __shared__ __attribute__((aligned(16))) float ent[256 * kEntStride];
  __shared__ int roi_list[256];
  __shared__ int bins[256];
  __shared__ int wave_cnt[4], ent_cnt[4], nb_s;
  const int PP = P * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int tiles_x = (W + 3) >> 2;
  const int x0 = (int)(blockIdx.x % tiles_x) * 4, y0 = (int)(blockIdx.x / tiles_x) * 4;
  const int b = blockIdx.y;
  const int chunks = (D + 255) >> 8;
  const int grp = blockIdx.z / chunks, d = (blockIdx.z % chunks) * 256 + tid;
  const int gh = grp / G, gw = grp % G;
  const bool active_c = d < D;
  if (tid == 0) {
    int n = 0;
    for (int bin = 0; bin < PP; ++bin)
      if (ps_group(bin / P, G, P) == gh && ps_group(bin % P, G, P) == gw) bins[n++] = bin;
    nb_s = n;
  }
  __syncthreads();
  const int nb = nb_s;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  if (nb > 0) {
    for (int base = 0; base < R; base += 256) {
      const int rr = base + tid;
      bool hit = false;
      if (rr < R) {
        const int4 w = win[rr];
        hit = w.w && w.x == b && (w.y & 0xffff) <= x0 + 3 && (w.y >> 16) >= x0 && (w.z & 0xffff) <= y0 + 3 && (w.z >> 16) >= y0;
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int off = 0, total = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = wave_cnt[k];
        off += k < wave ? n : 0;
        total += n;
      }
      if (hit) roi_list[off + __popcll(m & lt)] = rr;
      __syncthreads();
      const int nitems = total * nb;
      for (int it0 = 0; it0 < nitems; it0 += 256) {
        const int item = it0 + tid;
        bool act = false;
        float Wx[4] = {0.f, 0.f, 0.f, 0.f}, Wy[4] = {0.f, 0.f, 0.f, 0.f};
        float inv = 0.f;
        int idx = 0;
        if (item < nitems) {
          const int r = roi_list[item / nb], bin = bins[item % nb];
          const int ph = bin / P, pw = bin - ph * P;
          const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
          int nvx = 0, nvy = 0;
          for (int i = 0; i < S; ++i) {
            float w = sample_pos(g.wstart, i, g.sub_w), h = sample_pos(g.hstart, i, g.sub_h);
            if (!(w < -0.5f || w > (float)W - 0.5f)) {
              ++nvx;
              tent4(fminf(fmaxf(w, 0.f), (float)W - 1.f), x0, Wx);
            }
            if (!(h < -0.5f || h > (float)H - 0.5f)) {
              ++nvy;
              tent4(fminf(fmaxf(h, 0.f), (float)H - 1.f), y0, Wy);
            }
          }
          const float sx = Wx[0] + Wx[1] + Wx[2] + Wx[3], sy = Wy[0] + Wy[1] + Wy[2] + Wy[3];
          act = nvx * nvy > 0 && sx > 0.f && sy > 0.f;
          inv = act ? 1.f / (float)(nvx * nvy) : 0.f;
          idx = r * PP + bin;
        }
        const unsigned long long am = __ballot(act);
        if (lane == 0) ent_cnt[wave] = __popcll(am);
        __syncthreads();
        int eoff = 0, n_e = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int n = ent_cnt[k];
          eoff += k < wave ? n : 0;
          n_e += n;
        }
        if (act) {
          float *e = ent + (size_t)(eoff + __popcll(am & lt)) * kEntStride;
          e[0] = __int_as_float(idx);
          *reinterpret_cast<float4 *>(e + 4) = make_float4(Wx[0] * inv, Wx[1] * inv, Wx[2] * inv, Wx[3] * inv);
          *reinterpret_cast<float4 *>(e + 8) = make_float4(Wy[0], Wy[1], Wy[2], Wy[3]);
        }
        __syncthreads();
        tile_accumulate<half_t>(ent, n_e, dout, (size_t)D, d, active_c, acc);
        __syncthreads();
      }
    }
  }
  if (active_c) tile_store(acc, d_data, out_f32, b, y0, x0, H, W, C, (d * G + gh) * G + gw);
}

// Group-major layout (gm = 1): a workgroup owns (4x4-cell tile, image, 256 CONSECUTIVE map channels c = grp*D + d).  Those channels
// span ceil(256 / D) + 1 bin groups at most (D = 81: four; D = 4 or 2: all 49), so the workgroup walks the (RoI, bin) items of every
// group it covers once, each entry carries its group, and a thread accumulates the entries of its own group.  Against the operator-
// order kernel above: 16 instead of 49 channel slices for D = 81 with every lane active (81 of 256 there), ONE slice instead of 49
// for the D = 4 / D = 2 maps (4 / 2 active lanes of 256 there); stores of a wave are one contiguous run.
__global__ __launch_bounds__(256) void psroi_ps_bwd_data_gm_kernel(const half_t *__restrict__ dout, const float *__restrict__ rois,
                                                                   const float *__restrict__ trans, const int4 *__restrict__ win,
                                                                   void *__restrict__ d_data, int out_f32, int R, int H, int W, int C,
                                                                   int P, int S, int G, int D, float scale, float trans_std) {
  __shared__ __attribute__((aligned(16))) float ent[256 * kEntStride];
  __shared__ int roi_list[256];
  __shared__ int bins[256];
  __shared__ int wave_cnt[4], ent_cnt[4], nb_s;
  const int PP = P * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int tiles_x = (W + 3) >> 2;
  const int x0 = (int)(blockIdx.x % tiles_x) * 4, y0 = (int)(blockIdx.x / tiles_x) * 4;
  const int b = blockIdx.y;
  const int c = blockIdx.z * 256 + tid;          // this thread's map channel
  const bool active_c = c < C;
  const int mygrp = active_c ? c / D : -1, d = active_c ? c - mygrp * D : 0;
  const int g_lo = (blockIdx.z * 256) / D, g_hi = min(G * G - 1, (blockIdx.z * 256 + 255) / D);
  if (tid == 0) {
    int n = 0;
    for (int bin = 0; bin < PP; ++bin) {
      const int grp = ps_group(bin / P, G, P) * G + ps_group(bin % P, G, P);
      if (grp >= g_lo && grp <= g_hi) bins[n++] = bin;
    }
    nb_s = n;
  }
  __syncthreads();
  const int nb = nb_s;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  if (nb > 0) {
    for (int base = 0; base < R; base += 256) {
      const int rr = base + tid;
      bool hit = false;
      if (rr < R) {
        const int4 w = win[rr];
        hit = w.w && w.x == b && (w.y & 0xffff) <= x0 + 3 && (w.y >> 16) >= x0 && (w.z & 0xffff) <= y0 + 3 && (w.z >> 16) >= y0;
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int off = 0, total = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = wave_cnt[k];
        off += k < wave ? n : 0;
        total += n;
      }
      if (hit) roi_list[off + __popcll(m & lt)] = rr;
      __syncthreads();
      const int nitems = total * nb;
      for (int it0 = 0; it0 < nitems; it0 += 256) {
        const int item = it0 + tid;
        bool act = false;
        float Wx[4] = {0.f, 0.f, 0.f, 0.f}, Wy[4] = {0.f, 0.f, 0.f, 0.f};
        float inv = 0.f;
        int idx = 0, grp = 0;
        if (item < nitems) {
          const int r = roi_list[item / nb], bin = bins[item % nb];
          const int ph = bin / P, pw = bin - ph * P;
          grp = ps_group(ph, G, P) * G + ps_group(pw, G, P);
          const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
          int nvx = 0, nvy = 0;
          for (int i = 0; i < S; ++i) {
            float w = sample_pos(g.wstart, i, g.sub_w), h = sample_pos(g.hstart, i, g.sub_h);
            if (!(w < -0.5f || w > (float)W - 0.5f)) {
              ++nvx;
              tent4(fminf(fmaxf(w, 0.f), (float)W - 1.f), x0, Wx);
            }
            if (!(h < -0.5f || h > (float)H - 0.5f)) {
              ++nvy;
              tent4(fminf(fmaxf(h, 0.f), (float)H - 1.f), y0, Wy);
            }
          }
          const float sx = Wx[0] + Wx[1] + Wx[2] + Wx[3], sy = Wy[0] + Wy[1] + Wy[2] + Wy[3];
          act = nvx * nvy > 0 && sx > 0.f && sy > 0.f;
          inv = act ? 1.f / (float)(nvx * nvy) : 0.f;
          idx = r * PP + bin;
        }
        const unsigned long long am = __ballot(act);
        if (lane == 0) ent_cnt[wave] = __popcll(am);
        __syncthreads();
        int eoff = 0, n_e = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int n = ent_cnt[k];
          eoff += k < wave ? n : 0;
          n_e += n;
        }
        if (act) {
          float *e = ent + (size_t)(eoff + __popcll(am & lt)) * kEntStride;
          e[0] = __int_as_float(idx);
          e[1] = __int_as_float(grp);
          *reinterpret_cast<float4 *>(e + 4) = make_float4(Wx[0] * inv, Wx[1] * inv, Wx[2] * inv, Wx[3] * inv);
          *reinterpret_cast<float4 *>(e + 8) = make_float4(Wy[0], Wy[1], Wy[2], Wy[3]);
        }
        __syncthreads();
        for (int e = 0; e < n_e; ++e) {           // entries in list order: fixed summation order per channel
          const float *q = ent + e * kEntStride;
          if (__float_as_int(q[1]) != mygrp) continue;
          const float dv = (float)dout[(size_t)__float_as_int(q[0]) * D + d];
          const float4 wx = *reinterpret_cast<const float4 *>(q + 4), wy = *reinterpret_cast<const float4 *>(q + 8);
          const float X[4] = {wx.x * dv, wx.y * dv, wx.z * dv, wx.w * dv};
          const float Y[4] = {wy.x, wy.y, wy.z, wy.w};
#pragma unroll
          for (int cy = 0; cy < 4; ++cy)
#pragma unroll
            for (int cx = 0; cx < 4; ++cx) acc[cy * 4 + cx] += Y[cy] * X[cx];
        }
        __syncthreads();
      }
    }
  }
  if (active_c) tile_store(acc, d_data, out_f32, b, y0, x0, H, W, C, c);
}

// d_trans (R,2,P,P): one wave per (r, ph, pw); the lanes stride over the D output channels, shuffle reduction.  Separable like
// the group_size = 1 kernel: d/dtx of the bin value = sum_y sum_x Wy(y) dWx(x) U[y][x], d/dty = sum dWy(y) Wx(x) U[y][x] (tent sums and
// their derivatives over the valid samples of each axis) -- every window cell is read ONCE per channel instead of four corner reads
// per sample (64 two-byte gathers per channel and bin: 614 us per call on the 7*7*81 map).  The 2 x 8 axis weights of the (wave-uniform)
// bin are computed by sixteen lanes, one each, and broadcast with v_readlane; windows beyond 8 cells per axis take the sample loop.
__global__ __launch_bounds__(256) void psroi_ps_bwd_trans_kernel(const half_t *__restrict__ dout, const half_t *__restrict__ data,
                                                                 const float *__restrict__ rois, const float *__restrict__ trans,
                                                                 float *__restrict__ d_trans, int R, int H, int W, int C, int P,
                                                                 int S, int G, int D, float scale, float trans_std, int gm) {
  const long total = (long)R * P * P;
  const long wv = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wv >= total) return;                                           // whole waves leave together
  const int lane = threadIdx.x & 63;
  const int pw = (int)(wv % P), ph = (int)((wv / P) % P), r = (int)(wv / ((long)P * P));
  const RoiGeom g = roi_geom(rois, trans, r, ph, pw, P, S, scale, trans_std);
  const int gh = ps_group(ph, G, P), gw = ps_group(pw, G, P);
  const half_t *img = data + (size_t)g.b * H * W * C;
  const AxisSamples ax = axis_samples(g.wstart, g.sub_w, S, W), ay = axis_samples(g.hstart, g.sub_h, S, H);
  const int count = ax.n * ay.n;
  const int nx = count ? ax.hi - ax.lo + 1 : 0, ny = count ? ay.hi - ay.lo + 1 : 0;
  float gtx = 0.f, gty = 0.f;
  if (count && nx <= kWinMax && ny <= kWinMax) {
    // lane q (< 8): x weights of window column q; lane 8 + q: y weights of window row q
    const int q = lane & 7;
    const bool is_y = (lane >> 3) & 1;
    const float wl = is_y ? (q < ny ? tent_sum(ay, ay.lo + q) : 0.f) : (q < nx ? tent_sum(ax, ax.lo + q) : 0.f);
    const float dl = is_y ? (q < ny ? tent_dsum(ay, ay.lo + q) : 0.f) : (q < nx ? tent_dsum(ax, ax.lo + q) : 0.f);
    for (int d = lane; d < D + 63 - ((D - 1) & 63); d += 64) {         // (every lane runs every pass: the readlanes below are wave-wide)
      const bool on = d < D;
      const int c = ps_channel(on ? d : 0, gh, gw, G, D, gm);
      const float dv = on ? (float)dout[(size_t)wv * D + d] : 0.f;
      float sx = 0.f, sy = 0.f;
      for (int qy = 0; qy < ny; ++qy) {
        const float wy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), 8 + qy));
        const float dwy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl), 8 + qy));
        if (wy == 0.f && dwy == 0.f) continue;
        const half_t *row = img + ((size_t)(ay.lo + qy) * W + ax.lo) * C + c;
        for (int qx = 0; qx < nx; ++qx) {
          const float kx = wy * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl), qx));
          const float ky = dwy * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), qx));
          if (kx == 0.f && ky == 0.f) continue;
          const float u = on ? (float)row[(size_t)qx * C] : 0.f;
          sx += kx * u;
          sy += ky * u;
        }
      }
      gtx += sx * dv;
      gty += sy * dv;
    }
  } else if (count) {
    for (int ih = 0; ih < S; ++ih) {
      for (int iw = 0; iw < S; ++iw) {
        float w = sample_pos(g.wstart, iw, g.sub_w), h = sample_pos(g.hstart, ih, g.sub_h);
        if (w < -0.5f || w > (float)W - 0.5f || h < -0.5f || h > (float)H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)H - 1.f);
        const int xa = (int)floorf(w), xb = (int)ceilf(w), ya = (int)floorf(h), yb = (int)ceilf(h);
        const float dx = w - (float)xa, dy = h - (float)ya;
        for (int d = lane; d < D; d += 64) {
          const int c = ps_channel(d, gh, gw, G, D, gm);
          const float dv = (float)dout[(size_t)wv * D + d];
          const float U00 = (float)img[((size_t)ya * W + xa) * C + c], U01 = (float)img[((size_t)ya * W + xb) * C + c];
          const float U10 = (float)img[((size_t)yb * W + xa) * C + c], U11 = (float)img[((size_t)yb * W + xb) * C + c];
          gtx += (U11 * dy + U01 * (1.f - dy) - U10 * dy - U00 * (1.f - dy)) * dv;
          gty += (U11 * dx + U10 * (1.f - dx) - U01 * dx - U00 * (1.f - dx)) * dv;
        }
      }
    }
  }
  const float k = count ? trans_std / (float)count : 0.f;
  gtx *= k * g.roi_w;
  gty *= k * g.roi_h;
  for (int off = 32; off > 0; off >>= 1) {
    gtx += __shfl_xor(gtx, off, 64);
    gty += __shfl_xor(gty, off, 64);
  }
  if (lane == 0) {
    d_trans[(((size_t)r * 2 + 0) * P + ph) * P + pw] = gtx;
    d_trans[(((size_t)r * 2 + 1) * P + ph) * P + pw] = gty;
  }
}

SN_EXPORT int sn_psroi_pool_fwd(const void *data, const float *rois, const float *trans, void *out, int R, int H, int W,
                                int output_dim, int group_size, int pooled, int sample_per_part, float spatial_scale,
                                float trans_std, int group_major, sn_stream_t stream) {
  SN_REQUIRE(data && rois && out && R > 0 && output_dim > 0 && group_size > 0 && pooled > 0 && sample_per_part > 0 &&
                 sample_per_part <= kMaxS, "sn_psroi_pool_fwd: bad arguments (sample_per_part <= %d)", kMaxS);
  const int C = output_dim * group_size * group_size;
  if (output_dim >= 32) {
    const long waves = (long)R * pooled * pooled;
    hipLaunchKernelGGL(psroi_ps_fwd_kernel<true>, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)data, rois, trans, (half_t *)out, R, H, W, C, pooled, sample_per_part, group_size, output_dim,
                       spatial_scale, trans_std, group_major ? 1 : 0);
  } else {
    hipLaunchKernelGGL(psroi_ps_fwd_kernel<false>, dim3((unsigned)blocks_for((long)R * pooled * pooled * output_dim)), dim3(256), 0,
                       sn_stream(stream), (const half_t *)data, rois, trans, (half_t *)out, R, H, W, C, pooled, sample_per_part,
                       group_size, output_dim, spatial_scale, trans_std, group_major ? 1 : 0);
  }
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_psroi_pool_bwd(const void *dout, const void *data, const float *rois, const float *trans, void *d_data,
                                int d_data_f32, float *d_trans, int R, int B, int H, int W, int output_dim, int group_size,
                                int pooled, int sample_per_part, float spatial_scale, float trans_std, int group_major, void *ws,
                                sn_stream_t stream) {
  SN_REQUIRE(dout && data && rois && d_data && ws && R > 0 && B > 0 && output_dim > 0 && group_size > 0 && pooled > 0 &&
                 sample_per_part > 0, "sn_psroi_pool_bwd: bad arguments");
  SN_REQUIRE(H < 65536 && W < 65536 && pooled * pooled <= 256, "sn_psroi_pool_bwd: H, W < 65536 and pooled <= 16 required");
  SN_REQUIRE(!trans || d_trans, "sn_psroi_pool_bwd: d_trans required with trans");
  const int C = output_dim * group_size * group_size;
  const long gz = (long)group_size * group_size * sn_div_up(output_dim, 256);
  SN_REQUIRE(gz <= 65535 && B <= 65535, "sn_psroi_pool_bwd: grid too large");
  hipStream_t s = sn_stream(stream);
  int4 *win = (int4 *)ws;
  hipLaunchKernelGGL(dpsroi_window_kernel, dim3(sn_div_up(R, 4)), dim3(256), 0, s, rois, trans, win, R, H, W, pooled,
                     sample_per_part, spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  const int tiles = sn_div_up(W, 4) * sn_div_up(H, 4);
  if (group_major)
    hipLaunchKernelGGL(psroi_ps_bwd_data_gm_kernel, dim3(tiles, B, (unsigned)sn_div_up(C, 256)), dim3(256), 0, s, (const half_t *)dout,
                       rois, trans, (const int4 *)win, d_data, d_data_f32, R, H, W, C, pooled, sample_per_part, group_size, output_dim,
                       spatial_scale, trans_std);
  else
    hipLaunchKernelGGL(psroi_ps_bwd_data_kernel, dim3(tiles, B, (unsigned)gz), dim3(256), 0, s, (const half_t *)dout, rois, trans,
                       (const int4 *)win, d_data, d_data_f32, R, H, W, C, pooled, sample_per_part, group_size, output_dim,
                       spatial_scale, trans_std);
  SN_CHECK_LAUNCH();
  if (trans) {
    const long waves = (long)R * pooled * pooled;
    hipLaunchKernelGGL(psroi_ps_bwd_trans_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, s,
                       (const half_t *)dout, (const half_t *)data, rois, trans, d_trans, R, H, W, C, pooled, sample_per_part,
                       group_size, output_dim, spatial_scale, trans_std, group_major ? 1 : 0);
    SN_CHECK_LAUNCH();
  }
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
                                                            int KH, int KW, int stride, int pad, int dil, int DG, int off_ps,
                                                            SnDiv d_cpr, SnDiv d_T, SnDiv d_Wo, SnDiv d_Ho, SnDiv d_KW, SnDiv d_cg) {
  const int cpr = C >> 3, T = KH * KW;
  const long total = (long)N * Ho * Wo * T * cpr;      // (host: < 2^31)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // element -> (pixel m = (n, oy, ox), tap, 8-channel chunk) by multiply-high: the five 64-bit divisions this used to be were
    // ~400 VALU instructions in front of four 16-byte gathers and one store (the kernel ran at 2.4 TB/s of its bytes)
    unsigned chq, tapu, oxu, oyu;
    const unsigned t = sn_divmod((unsigned)i, d_cpr, chq);
    const unsigned m = sn_divmod(t, d_T, tapu);
    const unsigned t2 = sn_divmod(m, d_Wo, oxu);
    const int n = (int)sn_divmod(t2, d_Ho, oyu);
    const int ch = (int)chq * 8, tap = (int)tapu, ox = (int)oxu, oy = (int)oyu;
    const int g = (int)sn_div((unsigned)ch, d_cg), kh = (int)sn_div((unsigned)tap, d_KW), kw = tap - kh * KW;
    const TO *op = offset + (size_t)m * off_ps + g * 2 * T + 2 * tap;
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
//   d_offset (N,Ho,Wo,2*T*DG): sum over the group's channels of dcol * d(sample)/d(offset); each (m, tap, group)
//            is owned by `cg/8` consecutive lanes, reduced with shuffles (a gather -- no atomics);
//   d_data   (N,H,W,C) fp16/fp32, written once: same tile-ownership scheme as dpsroi_bwd_data_kernel.  A workgroup
//            owns a 4x4-cell tile of one image for the channels of ONE deformable group (offsets are per group);
//            it scans the Ho*Wo*T sampling points of the image (offsets are data dependent, so there is no
//            geometric shortcut that stays correct for large offsets), keeps those whose bilinear footprint
//            touches the tile (order-preserving ballot compaction into LDS), and every thread (= channel)
//            accumulates them into 16 registers.  v1 was a 377 M-atomic scatter (9 ms/layer).
template <typename TO>
__global__ __launch_bounds__(256) void deform_col2im_offset_kernel(const half_t *__restrict__ dcol, const half_t *__restrict__ data,
                                                                   const TO *__restrict__ offset, TO *__restrict__ d_offset, int N,
                                                                   int H, int W, int C, int Ho, int Wo, int KH, int KW, int stride,
                                                                   int pad, int dil, int DG, int off_ps, SnDiv d_cpr, SnDiv d_T,
                                                                   SnDiv d_Wo, SnDiv d_Ho, SnDiv d_KW, SnDiv d_cg) {
  const int cpr = C >> 3, T = KH * KW, cg = C / DG, lpg = cg >> 3;  // lanes per group (power of two <= 64)
  const long total = (long)N * Ho * Wo * T * cpr;      // (host: < 2^31)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < total;
  const long ii = active ? i : total - 1;
  unsigned chq, tapu, oxu, oyu;        // (multiply-high index decomposition: see deform_im2col_kernel)
  const unsigned t = sn_divmod((unsigned)ii, d_cpr, chq);
  const unsigned m = sn_divmod(t, d_T, tapu);
  const unsigned t2 = sn_divmod(m, d_Wo, oxu);
  const int n = (int)sn_divmod(t2, d_Ho, oyu);
  const int ch = (int)chq * 8, tap = (int)tapu, ox = (int)oxu, oy = (int)oyu;
  const int g = (int)sn_div((unsigned)ch, d_cg), kh = (int)sn_div((unsigned)tap, d_KW), kw = tap - kh * KW;
  const TO *op = offset + (size_t)m * off_ps + g * 2 * T + 2 * tap;
  const float py = (float)(oy * stride - pad + kh * dil) + (float)op[0], px = (float)(ox * stride - pad + kw * dil) + (float)op[1];
  const DeformSample s = deform_sample(py, px, H, W);
  float gy = 0.f, gx = 0.f;
  if (active && s.ok) {
    const half8 go = *reinterpret_cast<const half8 *>(dcol + ii * 8);
    const size_t base = (size_t)n * H * W * C + ch;
    const half8 v1 = *reinterpret_cast<const half8 *>(data + base + ((size_t)s.y0 * W + s.x0) * C);
    const half8 v2 = *reinterpret_cast<const half8 *>(data + base + ((size_t)s.y0 * W + s.x1) * C);
    const half8 v3 = *reinterpret_cast<const half8 *>(data + base + ((size_t)s.y1 * W + s.x0) * C);
    const half8 v4 = *reinterpret_cast<const half8 *>(data + base + ((size_t)s.y1 * W + s.x1) * C);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (float)go[j];
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
    TO *dp = d_offset + (size_t)m * off_ps + g * 2 * T + 2 * tap;
    dp[0] = (TO)gy;
    dp[1] = (TO)gx;
  }
}

// The same tile-owner gather on the matrix cores (see dpsroi_bwd_data_mfma_kernel): D[cell][channel] += sum_e W[cell][e] *
// dcol[row(e)][channel], W = Wy (x) Wx of a sample that touches the tile, as an fp16 hi + lo pair; a wave owns 64 channels of
// the deformable group, the workgroup's waves share one entry list (candidate order = the scalar kernel's order).
template <typename TO>
__global__ __launch_bounds__(256) void deform_col2im_data_mfma_kernel(const half_t *__restrict__ dcol, const TO *__restrict__ offset,
                                                                      void *__restrict__ d_data, int out_f32, int H, int W, int C,
                                                                      int Ho, int Wo, int KH, int KW, int stride, int pad, int dil,
                                                                      int DG, int off_ps, int slabs,
                                                                      const unsigned *__restrict__ dmax_bits) {
  __shared__ __attribute__((aligned(16))) half_t w_hi[16][kListPitch];
  __shared__ __attribute__((aligned(16))) half_t w_lo[16][kListPitch];
  __shared__ __attribute__((aligned(16))) int e_row[kListCap];
  __shared__ int seg_n[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6, nthr = blockDim.x;
  const int fr = lane & 15, kg = lane >> 4;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int T = KH * KW, cg = C / DG;
  const int tiles_x = (W + 3) >> 2;
  const int x0 = (int)(blockIdx.x % tiles_x) * 4, y0 = (int)(blockIdx.x / tiles_x) * 4;
  const int n = blockIdx.y, g = blockIdx.z / slabs, slab = blockIdx.z - g * slabs;
  const int cl0 = slab * nthr + wave * 64;     // first channel (inside the group) of this wave's 64
  int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
  if (dmax_bits) {
    const float D = __uint_as_float(*dmax_bits);
    if (D < 1.0e6f) {
      const int Di = (int)ceilf(D) + 1, span = (KH > KW ? KH : KW) - 1;
      auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };   // floor(a / b), b > 0
      oy_lo = max(0, -fdiv(-(y0 - Di + pad - span * dil), stride));
      oy_hi = min(Ho - 1, fdiv(y0 + 3 + Di + pad, stride));
      ox_lo = max(0, -fdiv(-(x0 - Di + pad - span * dil), stride));
      ox_hi = min(Wo - 1, fdiv(x0 + 3 + Di + pad, stride));
    }
  }
  const int wh = max(oy_hi - oy_lo + 1, 0), ww = max(ox_hi - ox_lo + 1, 0);
  const int cand = wh * ww * T, cand_full = Ho * Wo * T;
  floatx4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  auto flush = [&]() {
    const int kend = (cnt + 31) & ~31;
    for (int i = tid; i < (kend - cnt) * 16; i += nthr) {
      const int e = cnt + i / 16, m = i & 15;
      w_hi[m][e] = (half_t)0;
      w_lo[m][e] = (half_t)0;
      if (m == 0) e_row[e] = 0;
    }
    __syncthreads();
    for (int k0 = 0; k0 < cnt; k0 += 32) {
      const half8 a_hi = *reinterpret_cast<const half8 *>(&w_hi[fr][k0 + kg * 8]);
      const half8 a_lo = *reinterpret_cast<const half8 *>(&w_lo[fr][k0 + kg * 8]);
      const int4 r0 = *reinterpret_cast<const int4 *>(&e_row[k0 + kg * 8]);
      const int4 r1 = *reinterpret_cast<const int4 *>(&e_row[k0 + kg * 8 + 4]);
      const int rows[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      half8 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = cl0 + j * 16 + fr;
#pragma unroll
        for (int t = 0; t < 8; ++t) bv[j][t] = cl < cg ? dcol[(size_t)rows[t] * C + g * cg + cl] : (half_t)0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bv[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, bv[j], acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
    cnt = 0;
  };
  for (int base = 0; base < cand; base += nthr) {
    if (cnt > kListCap - nthr) flush();
    const int widx = base + tid;
    int idx = 0;
    bool hit = false;
    float Wx[4] = {0.f, 0.f, 0.f, 0.f}, Wy[4] = {0.f, 0.f, 0.f, 0.f};
    if (widx < cand) {
      const int wl = widx / T, tap = widx - wl * T;
      const int oy = oy_lo + wl / ww, ox = ox_lo + wl % ww;
      const int ml = oy * Wo + ox;
      idx = ml * T + tap;
      const int kh = tap / KW, kw = tap - kh * KW;
      const TO *op = offset + ((size_t)n * Ho * Wo + ml) * off_ps + g * 2 * T + 2 * tap;
      const float py = (float)(oy * stride - pad + kh * dil) + (float)op[0];
      const float px = (float)(ox * stride - pad + kw * dil) + (float)op[1];
      const DeformSample s = deform_sample(py, px, H, W);
      if (s.ok && s.x1 >= x0 && s.x0 <= x0 + 3 && s.y1 >= y0 && s.y0 <= y0 + 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          Wx[k] = (s.x0 - x0 == k ? 1.f - s.lx : 0.f) + (s.x1 - x0 == k ? s.lx : 0.f);
          Wy[k] = (s.y0 - y0 == k ? 1.f - s.ly : 0.f) + (s.y1 - y0 == k ? s.ly : 0.f);
        }
        if (s.x0 == s.x1) {  // clamped at the border: deform_sample put weight 1 on the single cell
#pragma unroll
          for (int k = 0; k < 4; ++k) Wx[k] = (s.x0 - x0 == k ? 1.f : 0.f);
        }
        if (s.y0 == s.y1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) Wy[k] = (s.y0 - y0 == k ? 1.f : 0.f);
        }
        hit = true;
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) seg_n[wave] = __popcll(m);
    __syncthreads();
    int eoff = cnt, added = 0;
    for (int k = 0; k < nwave; ++k) {
      const int nn = seg_n[k];
      eoff += k < wave ? nn : 0;
      added += nn;
    }
    if (hit) {
      const int pos = eoff + __popcll(m & lt);
      e_row[pos] = n * cand_full + idx;   // row (m, tap) of dcol
#pragma unroll
      for (int cy = 0; cy < 4; ++cy)
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) {
          const float w = Wy[cy] * Wx[cx];
          const half_t hi = (half_t)w;
          w_hi[cy * 4 + cx][pos] = hi;
          w_lo[cy * 4 + cx][pos] = (half_t)(w - (float)hi);
        }
    }
    cnt += added;
    __syncthreads();
  }
  flush();
  const int y = y0 + kg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cl = cl0 + j * 16 + fr;
    if (cl >= cg || y >= H) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = x0 + r;
      if (x >= W) continue;
      const size_t o = (((size_t)n * H + y) * W + x) * C + g * cg + cl;
      if (out_f32) ((float *)d_data)[o] = acc[j][r];
      else ((half_t *)d_data)[o] = (half_t)acc[j][r];
    }
  }
}

SN_EXPORT int sn_deform_im2col(const void *data, const void *offset, void *col, int N, int H, int W, int C, int KH, int KW,
                               int stride, int pad, int dil, int deformable_groups, int offset_pix_stride, int offset_dtype,
                               sn_stream_t stream) {
  SN_REQUIRE(data && offset && col && C % 8 == 0 && deformable_groups > 0 && (C / deformable_groups) % 8 == 0,
             "sn_deform_im2col: bad arguments");
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long total = (long)N * Ho * Wo * KH * KW * (C / 8);
  SN_REQUIRE(total < 2147483647L, "sn_deform_im2col: too many column elements for 32-bit indexing");
  const SnDiv d_cpr = sn_div_make(C / 8), d_T = sn_div_make(KH * KW), d_Wo = sn_div_make(Wo), d_Ho = sn_div_make(Ho),
              d_KW = sn_div_make(KW), d_cg = sn_div_make(C / deformable_groups);
  if (offset_dtype == 0)
    hipLaunchKernelGGL((deform_im2col_kernel<half_t>), dim3((unsigned)blocks_for(total)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)data, (const half_t *)offset, (half_t *)col, N, H, W, C, Ho, Wo, KH, KW, stride, pad, dil,
                       deformable_groups, offset_pix_stride, d_cpr, d_T, d_Wo, d_Ho, d_KW, d_cg);
  else
    hipLaunchKernelGGL((deform_im2col_kernel<float>), dim3((unsigned)blocks_for(total)), dim3(256), 0, sn_stream(stream),
                       (const half_t *)data, (const float *)offset, (half_t *)col, N, H, W, C, Ho, Wo, KH, KW, stride, pad, dil,
                       deformable_groups, offset_pix_stride, d_cpr, d_T, d_Wo, d_Ho, d_KW, d_cg);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// max |offset| of a launch as float bits (non-negative floats order like unsigned integers): *out must be zero on entry
template <typename TO>
__global__ __launch_bounds__(256) void deform_absmax_kernel(const TO *__restrict__ offset, long rows, int cols, int ld,
                                                            unsigned *__restrict__ out) {
  float m = 0.f;
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const float v = fabsf((float)offset[r * ld + (i - r * cols)]);
    m = (v > m || v != v) ? (v != v ? INFINITY : v) : m;      // NaN -> +inf: the window opens completely
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  // one integer atomic per workgroup, and only when it can raise the value: 4096 same-address atomics (one per wave of 1024
  // workgroups) serialised to 51 us for a 3 MB read
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const unsigned bits = __float_as_uint(b);
    if (bits > *reinterpret_cast<volatile unsigned *>(out)) atomicMax(out, bits);
  }
}

SN_EXPORT int sn_deform_col2im(const void *dcol, const void *data, const void *offset, void *d_data, int d_data_f32,
                               void *d_offset, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int dil,
                               int deformable_groups, int offset_pix_stride, int offset_dtype, void *ws, sn_stream_t stream) {
  SN_REQUIRE(dcol && data && offset && C % 8 == 0 && deformable_groups > 0 && N > 0, "sn_deform_col2im: bad arguments");
  SN_REQUIRE(C % deformable_groups == 0, "sn_deform_col2im: C must be a multiple of the deformable groups");
  const int cg = C / deformable_groups, lpg = cg / 8;
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  SN_REQUIRE((long)N * Ho * Wo * KH * KW < 2147483647L, "sn_deform_col2im: too many sampling points");
  hipStream_t s = sn_stream(stream);
  if (d_offset) {
    SN_REQUIRE(lpg >= 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0 && cg % 8 == 0,
               "sn_deform_col2im: channels per deformable group / 8 must be a power of two <= 64");
    const long total = (long)N * Ho * Wo * KH * KW * (C / 8);
    SN_REQUIRE(total < 2147483647L - 256, "sn_deform_col2im: too many column elements for 32-bit indexing");
    const SnDiv d_cpr = sn_div_make(C / 8), d_T = sn_div_make(KH * KW), d_Wo = sn_div_make(Wo), d_Ho = sn_div_make(Ho),
                d_KW = sn_div_make(KW), d_cg = sn_div_make(cg);
    if (offset_dtype == 0)
      hipLaunchKernelGGL((deform_col2im_offset_kernel<half_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                         (const half_t *)dcol, (const half_t *)data, (const half_t *)offset, (half_t *)d_offset, N, H, W, C, Ho, Wo,
                         KH, KW, stride, pad, dil, deformable_groups, offset_pix_stride, d_cpr, d_T, d_Wo, d_Ho, d_KW, d_cg);
    else
      hipLaunchKernelGGL((deform_col2im_offset_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                         (const half_t *)dcol, (const half_t *)data, (const float *)offset, (float *)d_offset, N, H, W, C, Ho, Wo,
                         KH, KW, stride, pad, dil, deformable_groups, offset_pix_stride, d_cpr, d_T, d_Wo, d_Ho, d_KW, d_cg);
    SN_CHECK_LAUNCH();
  }
  if (d_data) {
    const int bt = cg >= 256 ? 256 : sn_div_up(cg, 64) * 64, slabs = sn_div_up(cg, bt);
    const dim3 grid(sn_div_up(W, 4) * sn_div_up(H, 4), N, deformable_groups * slabs);
    unsigned *dmax = (unsigned *)ws;       // 4 bytes of scratch: max |offset| prunes the candidate scan (NULL: full scan)
    if (dmax) {
      SN_HIP(hipMemsetAsync(dmax, 0, sizeof(unsigned), s));
      const long rows = (long)N * Ho * Wo;
      const int cols = 2 * KH * KW * deformable_groups;
      const long want = (rows * cols + 255) / 256;
      const unsigned blocks = (unsigned)(want > 256 ? 256 : (want < 1 ? 1 : want));
      if (offset_dtype == 0)
        hipLaunchKernelGGL((deform_absmax_kernel<half_t>), dim3(blocks), dim3(256), 0, s, (const half_t *)offset, rows, cols,
                           offset_pix_stride, dmax);
      else
        hipLaunchKernelGGL((deform_absmax_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float *)offset, rows, cols,
                           offset_pix_stride, dmax);
      SN_CHECK_LAUNCH();
    }
    // entries x channels on the matrix cores (the scalar tile-owner gather it replaced, 213 -> 130 us per layer, is gone)
    if (offset_dtype == 0)
      hipLaunchKernelGGL((deform_col2im_data_mfma_kernel<half_t>), grid, dim3(bt), 0, s, (const half_t *)dcol,
                         (const half_t *)offset, d_data, d_data_f32, H, W, C, Ho, Wo, KH, KW, stride, pad, dil, deformable_groups,
                         offset_pix_stride, slabs, (const unsigned *)dmax);
    else
      hipLaunchKernelGGL((deform_col2im_data_mfma_kernel<float>), grid, dim3(bt), 0, s, (const half_t *)dcol,
                         (const float *)offset, d_data, d_data_f32, H, W, C, Ho, Wo, KH, KW, stride, pad, dil, deformable_groups,
                         offset_pix_stride, slabs, (const unsigned *)dmax);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}
