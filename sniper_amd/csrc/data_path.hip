// data_path.hip -- box geometry, chip generation and RPN anchor labelling for gfx950.
//
// Everything here is integer / box-coordinate work that must reproduce the reference bit for bit,
// so this file is compiled with -ffp-contract=off (no FMA contraction: the reference's x86 code
// rounds after every multiply) and uses IEEE division (hipcc default).
//
// Reference algorithms: lib/bbox/bbox.pyx:17-95, lib/chips/cchips.cpp:14-177,
// lib/data_utils/data_workers.py:133-371, lib/data_utils/generate_anchor.py.
#include "common.h"

// ============================================================================================
// IoU / ignore-overlap, float64.  One thread per (n,k) pair; consecutive lanes walk k so the
// (N,K) row-major store is coalesced and the 32-byte box of row n is a wave-broadcast load.
// HBM-bound: 32*(N+K) bytes in, 8*N*K bytes out.
// ============================================================================================
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }

// bbox.pyx:17-57 (mode 0) / :59-95 (mode 1); b = box, q = query box.
__device__ __forceinline__ double overlap_f64(const double b0, const double b1, const double b2, const double b3,
                                              const double q0, const double q1, const double q2, const double q3,
                                              const int mode) {
  const double box_area = (q2 - q0 + 1) * (q3 - q1 + 1);
  const double iw = dmin(b2, q2) - dmax(b0, q0) + 1;
  if (iw > 0) {
    const double ih = dmin(b3, q3) - dmax(b1, q1) + 1;
    if (ih > 0) {
      if (mode == 0) {
        const double ua = (b2 - b0 + 1) * (b3 - b1 + 1) + box_area - iw * ih;
        return iw * ih / ua;
      }
      return iw * ih / box_area;
    }
  }
  return 0.0;
}

__global__ __launch_bounds__(256) void iou_f64_kernel(const double *__restrict__ boxes, int N,
                                                      const double *__restrict__ query, int K,
                                                      double *__restrict__ out, int mode) {
  const long total = (long)N * K;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / K), k = (int)(idx % K);
    const double4 b = reinterpret_cast<const double4 *>(boxes)[n];
    const double4 q = reinterpret_cast<const double4 *>(query)[k];
    out[idx] = overlap_f64(b.x, b.y, b.z, b.w, q.x, q.y, q.z, q.w, mode);
  }
}

SN_EXPORT int sn_iou_f64(const double *d_boxes, int N, const double *d_query, int K, double *d_out, int mode,
                         sn_stream_t stream) {
  SN_REQUIRE(N >= 0 && K >= 0 && (mode == 0 || mode == 1), "sn_iou_f64: bad sizes N=%d K=%d mode=%d", N, K, mode);
  if (N == 0 || K == 0) return SN_OK;
  SN_REQUIRE(d_boxes && d_query && d_out, "sn_iou_f64: null pointer");
  const long total = (long)N * K;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;  // 16 blocks/CU, grid-stride the rest
  hipLaunchKernelGGL(iou_f64_kernel, dim3(blocks), dim3(256), 0, sn_stream(stream), d_boxes, N, d_query, K, d_out,
                     mode);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ============================================================================================
// Chip generation: one workgroup per (image, scale) unit.
//   phase 1  every candidate chip (in shuffled order) gets a bitmask of the boxes it fully
//            contains: float32 iw*ih/area2 == 1, cchips.cpp:43-44,135;
//   phase 2  greedy set cover, cchips.cpp:146-167: pick the first chip with the strictly largest
//            popcount, clear its boxes from every row, repeat.
// The mask lives in a caller-supplied workspace (C x ceil(n/64) u64 per unit, L2 resident).
// ============================================================================================
__host__ __device__ inline int chips_steps(int extent, int chipsize, int stride) {
  const int span = extent - chipsize;  // loop `for (i = 0; i < extent - chipsize; i += stride)`
  return span > 0 ? (span + stride - 1) / stride : 0;
}

// Coordinates of candidate `s` in the enumeration order of cchips.cpp:62-108.
__device__ inline void chips_candidate(int s, int W, int H, int cs, int stride, int nx, int ny, float c[4]) {
  const int mw = max(W - cs, 0), mh = max(H - cs, 0);
  if (s == 0) {
    c[0] = (float)mw; c[1] = 0.f; c[2] = (float)(W - 1); c[3] = (float)min(cs, H - 1);
  } else if (s == 1) {
    c[0] = 0.f; c[1] = (float)mh; c[2] = (float)min(cs, W - 1); c[3] = (float)(H - 1);
  } else if (s == 2) {
    c[0] = (float)mw; c[1] = (float)mh; c[2] = (float)(W - 1); c[3] = (float)(H - 1);
  } else {
    s -= 3;
    if (s < nx * ny) {
      const int i = (s / ny) * stride, j = (s % ny) * stride;  // x outer, y inner
      c[0] = (float)i; c[1] = (float)j; c[2] = (float)(i + cs - 1); c[3] = (float)(j + cs - 1);
    } else if (s < nx * ny + ny) {
      const int i = (s - nx * ny) * stride;  // right-edge column, cchips.cpp:94-100
      c[0] = (float)max(W - cs - 1, 0); c[1] = (float)i; c[2] = (float)(W - 1); c[3] = (float)(i + cs - 1);
    } else {
      const int i = (s - nx * ny - ny) * stride;  // bottom-edge row, cchips.cpp:102-108
      c[0] = (float)i; c[1] = (float)max(H - cs - 1, 0); c[2] = (float)(i + cs - 1); c[3] = (float)(H - 1);
    }
  }
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), off, 64);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o > v ? o : v;
  }
  return v;
}

constexpr int kChipsThreads = 256;
constexpr int kChipsMaxWords = 64;  // <= 4096 boxes per unit

__global__ __launch_bounds__(kChipsThreads) void chips_generate_kernel(
    const float *__restrict__ boxes, const int32_t *__restrict__ box_off, const int32_t *__restrict__ meta,
    const int32_t *__restrict__ perm, const int32_t *__restrict__ cand_off, int words_max,
    unsigned long long *__restrict__ mask_ws, float *__restrict__ out_chips, int32_t *__restrict__ out_ids,
    int32_t *__restrict__ out_count) {
  const int u = blockIdx.x, tid = threadIdx.x;
  const int b0 = box_off[u], n = box_off[u + 1] - b0;
  const int c0 = cand_off[u], C = cand_off[u + 1] - c0;
  if (n <= 0) {  // cchips.cpp:56-57
    if (tid == 0) out_count[u] = 0;
    return;
  }
  const int W = meta[4 * u], H = meta[4 * u + 1], cs = meta[4 * u + 2], stride = meta[4 * u + 3];
  const int nx = chips_steps(W, cs, stride), ny = chips_steps(H, cs, stride);
  const int Wd = (n + 63) >> 6;
  unsigned long long *mask = mask_ws + (size_t)c0 * words_max;
  const float *bx = boxes + (size_t)4 * b0;

  // ---- phase 1: containment bitmasks
  for (int c = tid; c < C; c += kChipsThreads) {
    float ch[4];
    chips_candidate(perm ? perm[c0 + c] : c, W, H, cs, stride, nx, ny, ch);
    for (int w = 0; w < Wd; ++w) {
      unsigned long long word = 0;
      const int jend = min(64, n - w * 64);
      for (int j = 0; j < jend; ++j) {
        const float4 q = reinterpret_cast<const float4 *>(bx)[w * 64 + j];
        const float area2 = (q.z - q.x + 1) * (q.w - q.y + 1);
        const float iw = fminf(ch[2], q.z) - fmaxf(ch[0], q.x) + 1;
        float ov = 0.f;
        if (iw > 0) {
          const float ih = fminf(ch[3], q.w) - fmaxf(ch[1], q.y) + 1;
          if (ih > 0) ov = iw * ih / area2;
        }
        if (ov == 1.0f) word |= 1ull << j;
      }
      mask[(size_t)c * Wd + w] = word;
    }
  }
  __syncthreads();

  // ---- phase 2: greedy cover
  __shared__ unsigned long long red[kChipsThreads / kWave];
  __shared__ unsigned long long sel[kChipsMaxWords];
  __shared__ unsigned long long best_s;
  int nout = 0;
  for (;;) {
    unsigned long long key = 0;
    for (int c = tid; c < C; c += kChipsThreads) {
      int cnt = 0;
      for (int w = 0; w < Wd; ++w) cnt += __popcll(mask[(size_t)c * Wd + w]);
      if (cnt > 0) {
        // max count first, then lowest slot index ("first max", strict > in cchips.cpp:151)
        const unsigned long long k = ((unsigned long long)cnt << 32) | (unsigned long long)(0xffffffffu - (unsigned)c);
        key = k > key ? k : key;
      }
    }
    key = wave_max_u64(key);
    if ((tid & (kWave - 1)) == 0) red[tid / kWave] = key;
    __syncthreads();
    if (tid == 0) {
      unsigned long long b = red[0];
      for (int i = 1; i < kChipsThreads / kWave; ++i) b = red[i] > b ? red[i] : b;
      best_s = b;
    }
    __syncthreads();
    const unsigned long long best = best_s;
    if (best == 0) break;  // block-uniform
    const int mid = (int)(0xffffffffu - (unsigned)(best & 0xffffffffu));
    if (tid == 0) {
      float ch[4];
      chips_candidate(perm ? perm[c0 + mid] : mid, W, H, cs, stride, nx, ny, ch);
      float *o = out_chips + (size_t)4 * (b0 + nout);
      o[0] = ch[0]; o[1] = ch[1]; o[2] = ch[2]; o[3] = ch[3];
      out_ids[b0 + nout] = mid;
    }
    for (int w = tid; w < Wd; w += kChipsThreads) sel[w] = mask[(size_t)mid * Wd + w];
    __syncthreads();
    for (int c = tid; c < C; c += kChipsThreads)
      for (int w = 0; w < Wd; ++w) mask[(size_t)c * Wd + w] &= ~sel[w];
    ++nout;
    __syncthreads();
  }
  if (tid == 0) out_count[u] = nout;
}

SN_EXPORT int sn_chips_num_candidates(int width, int height, int chipsize, int stride) {
  if (stride <= 0) return -1;
  const int nx = chips_steps(width, chipsize, stride), ny = chips_steps(height, chipsize, stride);
  return 3 + nx * ny + ny + nx;
}

SN_EXPORT size_t sn_chips_workspace_bytes(int total_cand, int max_boxes_per_unit) {
  const size_t words = (size_t)((max_boxes_per_unit + 63) / 64);
  return sn_align((size_t)total_cand * (words ? words : 1) * sizeof(unsigned long long));
}

SN_EXPORT int sn_chips_generate_batch(const float *d_boxes, const int32_t *d_box_off, const int32_t *d_meta,
                                      const int32_t *d_perm, const int32_t *d_cand_off, int U, int max_boxes_per_unit,
                                      void *d_mask_ws, float *d_out_chips, int32_t *d_out_ids, int32_t *d_out_count,
                                      sn_stream_t stream) {
  SN_REQUIRE(U >= 0, "sn_chips_generate_batch: U=%d", U);
  if (U == 0) return SN_OK;
  SN_REQUIRE(d_box_off && d_meta && d_cand_off && d_mask_ws && d_out_chips && d_out_ids && d_out_count,
             "sn_chips_generate_batch: null pointer");
  SN_REQUIRE(max_boxes_per_unit >= 0 && max_boxes_per_unit <= 64 * kChipsMaxWords,
             "sn_chips_generate_batch: at most %d boxes per unit (got %d)", 64 * kChipsMaxWords, max_boxes_per_unit);
  const int words_max = max(1, (max_boxes_per_unit + 63) / 64);
  hipLaunchKernelGGL(chips_generate_kernel, dim3(U), dim3(kChipsThreads), 0, sn_stream(stream), d_boxes, d_box_off,
                     d_meta, d_perm, d_cand_off, words_max, (unsigned long long *)d_mask_ws, d_out_chips, d_out_ids,
                     d_out_count);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ============================================================================================
// Box -> chip assignment, chip_worker.box_assigner's inner loops (data_workers.py:516-535 for the
// positive chips, :557-572 for the negative chips), for a ragged batch of (image, scale) units.
// One thread per box: arg-max over the unit's chips of intersection / box-area (first maximum, as
// ignore_overlaps(...).argmax(axis=0)), then the box is accepted iff the clipped intersection is at
// least one pixel each way and sqrt(|inter area|) is inside the scale's valid range:
//   mode 0 (coarsest scale): area >= lo;  mode 1: area <= hi (positive pass);  mode 2: area < hi.
// out_chip[box] = chip index inside the unit, or -1.  All float64 like the reference.
// ============================================================================================
__global__ __launch_bounds__(256) void assign_boxes_kernel(const double *__restrict__ chips, const int32_t *__restrict__ chip_off,
                                                           const double *__restrict__ boxes, const int32_t *__restrict__ box_off,
                                                           const double *__restrict__ range, const int32_t *__restrict__ mode,
                                                           const int32_t *__restrict__ unit_of_box, int total_boxes,
                                                           int32_t *__restrict__ out_chip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total_boxes) return;
  const int u = unit_of_box[i];
  const int c0 = chip_off[u], nc = chip_off[u + 1] - c0;
  const double4 b = reinterpret_cast<const double4 *>(boxes)[i];
  int best = -1;
  double bestv = -1.0;
  for (int c = 0; c < nc; ++c) {
    const double4 ch = reinterpret_cast<const double4 *>(chips)[c0 + c];
    const double ov = overlap_f64(ch.x, ch.y, ch.z, ch.w, b.x, b.y, b.z, b.w, 1);
    if (ov > bestv) { bestv = ov; best = c; }
  }
  int res = -1;
  if (best >= 0) {
    const double4 ch = reinterpret_cast<const double4 *>(chips)[c0 + best];
    const double x1 = dmax(ch.x, b.x), x2 = dmin(ch.z, b.z), y1 = dmax(ch.y, b.y), y2 = dmin(ch.w, b.w);
    const double area = sqrt(fabs((x2 - x1) * (y2 - y1)));
    const double lo = range[2 * u], hi = range[2 * u + 1];
    const int m = mode[u];
    const bool in_range = m == 0 ? (area >= lo) : (m == 1 ? (area <= hi) : (area < hi));
    if (x2 - x1 >= 1 && y2 - y1 >= 1 && in_range) res = best;
  }
  out_chip[i] = res;
}

SN_EXPORT int sn_assign_boxes_batch(const double *d_chips, const int32_t *d_chip_off, const double *d_boxes,
                                    const int32_t *d_box_off, const double *d_range, const int32_t *d_mode,
                                    const int32_t *d_unit_of_box, int U, int total_boxes, int32_t *d_out_chip,
                                    sn_stream_t stream) {
  SN_REQUIRE(U >= 0 && total_boxes >= 0, "sn_assign_boxes_batch: bad sizes");
  if (U == 0 || total_boxes == 0) return SN_OK;
  SN_REQUIRE(d_chips && d_chip_off && d_boxes && d_box_off && d_range && d_mode && d_unit_of_box && d_out_chip,
             "sn_assign_boxes_batch: null pointer");
  hipLaunchKernelGGL(assign_boxes_kernel, dim3(sn_div_up(total_boxes, 256)), dim3(256), 0, sn_stream(stream), d_chips,
                     d_chip_off, d_boxes, d_box_off, d_range, d_mode, d_unit_of_box, total_boxes, d_out_chip);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// ============================================================================================
// RPN anchor labelling, batched over chips.  Four kernels on one stream:
//   K1 anchor_prep_kernel     per chip: GT shift/scale/round/clip/filter, valid/invalid split
//   K2 anchor_gtmax_kernel    per (chip, anchor): IoU vs valid GT, atomic per-GT max (column max)
//   K3 anchor_label_kernel    per (chip, anchor): labels before sub-sampling, argmax GT
//   K4 anchor_finish_kernel   per chip: fg/bg sub-sampling by smallest key, targets, dense outputs
// Anchor "reference order": idx = cell*A + a with cell = y*F + x (data_workers.py:151-158).
// ============================================================================================
struct AnchorChipWs {  // per-chip scratch header
  int nvalid, ninvalid, nkept, pad;
};

struct AnchorWsLayout {
  size_t hdr, vbox, ibox, gtmax, argmax, label_pre, per_chip_total;
  int G, total;
};

static AnchorWsLayout anchor_ws_layout(int A, int F, int G) {
  AnchorWsLayout L;
  L.G = G;
  L.total = A * F * F;
  size_t off = 0;
  L.hdr = off; off += sn_align(sizeof(AnchorChipWs), 64);
  L.vbox = off; off += sn_align((size_t)G * 4 * sizeof(double), 64);
  L.ibox = off; off += sn_align((size_t)G * 4 * sizeof(double), 64);
  L.gtmax = off; off += sn_align((size_t)G * sizeof(unsigned long long), 64);
  L.argmax = off; off += sn_align((size_t)L.total * sizeof(int16_t), 64);
  L.label_pre = off; off += sn_align((size_t)L.total * sizeof(int8_t), 64);
  L.per_chip_total = sn_align(off, 256);
  return L;
}

SN_EXPORT size_t sn_anchor_workspace_bytes(int B, int A, int F, int G) {
  return anchor_ws_layout(A, F, G).per_chip_total * (size_t)(B > 0 ? B : 1);
}

constexpr int kAnchorMaxG = 128;

// K1 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAnchorMaxG) void anchor_prep_kernel(
    const float *__restrict__ gt, const float *__restrict__ gt_cls, const uint8_t *__restrict__ inchip,
    const int32_t *__restrict__ ngt, const double *__restrict__ crop, const float *__restrict__ scale, int G, int im_h,
    int im_w, char *__restrict__ ws, AnchorWsLayout L, float *__restrict__ gt_out) {
  const int b = blockIdx.x, g = threadIdx.x;
  char *w = ws + (size_t)b * L.per_chip_total;
  AnchorChipWs *hdr = reinterpret_cast<AnchorChipWs *>(w + L.hdr);
  double *vbox = reinterpret_cast<double *>(w + L.vbox);
  double *ibox = reinterpret_cast<double *>(w + L.ibox);
  unsigned long long *gtmax = reinterpret_cast<unsigned long long *>(w + L.gtmax);
  __shared__ float sb[kAnchorMaxG][4];
  __shared__ unsigned char skept[kAnchorMaxG], svalid[kAnchorMaxG], sinchip[kAnchorMaxG];
  const int n = min(ngt[b], G);
  float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
  bool kept = false;
  if (g < n) {
    const float *p = gt + ((size_t)b * G + g) * 4;
    const double cx = crop[2 * b], cy = crop[2 * b + 1];
    const float s = scale[b];
    // data_workers.py:203-206: float32 array -= float64 scalar (evaluated in f64, stored as f32)
    x1 = (float)((double)p[0] - cx);
    y1 = (float)((double)p[1] - cy);
    x2 = (float)((double)p[2] - cx);
    y2 = (float)((double)p[3] - cy);
    // :217 np.round(gt * im_scale) in float32 (round half to even), then clip_boxes
    x1 = rintf(x1 * s); y1 = rintf(y1 * s); x2 = rintf(x2 * s); y2 = rintf(y2 * s);
    const float wm = (float)(im_w - 1), hm = (float)(im_h - 1);
    x1 = fmaxf(fminf(x1, wm), 0.f); y1 = fmaxf(fminf(y1, hm), 0.f);
    x2 = fmaxf(fminf(x2, wm), 0.f); y2 = fmaxf(fminf(y2, hm), 0.f);
    kept = (x2 - x1 + 1 >= 10.f) && (y2 - y1 + 1 >= 10.f);  // filter_boxes(.., 10), :226
  }
  if (g < kAnchorMaxG) {
    sb[g][0] = x1; sb[g][1] = y1; sb[g][2] = x2; sb[g][3] = y2;
    skept[g] = kept ? 1 : 0;
    sinchip[g] = (g < n && kept && inchip[(size_t)b * G + g]) ? 1 : 0;
  }
  __syncthreads();
  // valid <=> IoU == 1 with some box assigned to this chip (:262-280); with integer coordinates
  // IoU == 1 exactly when the two boxes are identical.
  bool valid = false;
  if (kept) {
    for (int j = 0; j < n; ++j)
      if (sinchip[j] && sb[j][0] == x1 && sb[j][1] == y1 && sb[j][2] == x2 && sb[j][3] == y2) { valid = true; break; }
  }
  svalid[g] = valid ? 1 : 0;
  __syncthreads();
  if (g < n && kept) {
    int rk = 0, rv = 0, ri = 0;  // order-preserving compaction ranks
    for (int j = 0; j < g; ++j) {
      rk += skept[j];
      rv += (skept[j] && svalid[j]);
      ri += (skept[j] && !svalid[j]);
    }
    double *dst = valid ? (vbox + 4 * rv) : (ibox + 4 * ri);
    dst[0] = x1; dst[1] = y1; dst[2] = x2; dst[3] = y2;
    if (rk < 100) {  // fgt_boxes, :359-361
      float *o = gt_out + ((size_t)b * 100 + rk) * 5;
      o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = gt_cls[(size_t)b * G + g];
    }
  }
  if (g < G) gtmax[g] = 0ull;
  if (g == 0) {
    int nk = 0, nv = 0;
    for (int j = 0; j < n; ++j) { nk += skept[j]; nv += (skept[j] && svalid[j]); }
    hdr->nkept = nk; hdr->nvalid = nv; hdr->ninvalid = nk - nv; hdr->pad = 0;
  }
  // -1 padding of the rows not written above
  __syncthreads();
  {
    int nk = 0;
    for (int j = 0; j < n; ++j) nk += skept[j];
    for (int r = nk + g; r < 100; r += blockDim.x) {
      float *o = gt_out + ((size_t)b * 100 + r) * 5;
      o[0] = o[1] = o[2] = o[3] = o[4] = -1.f;
    }
  }
}

__device__ __forceinline__ void anchor_box(const double *__restrict__ base, int A, int F, int stride, int idx,
                                           double a[4]) {
  const int cell = idx / A, an = idx - cell * A;
  const double sx = (double)((cell % F) * stride), sy = (double)((cell / F) * stride);
  a[0] = base[4 * an + 0] + sx; a[1] = base[4 * an + 1] + sy;
  a[2] = base[4 * an + 2] + sx; a[3] = base[4 * an + 3] + sy;
}

__device__ __forceinline__ bool anchor_inside(const double a[4], int im_h, int im_w) {
  // data_workers.py:198-201 (x2 is compared with im_info[0] = height, y2 with im_info[1]: preserved)
  return a[0] >= -32 && a[1] >= -32 && a[2] < (double)(im_h + 32) && a[3] < (double)(im_w + 32);
}

// K2 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void anchor_gtmax_kernel(const double *__restrict__ base, int A, int F, int stride,
                                                           int im_h, int im_w, char *__restrict__ ws,
                                                           AnchorWsLayout L) {
  const int b = blockIdx.y;
  char *w = ws + (size_t)b * L.per_chip_total;
  const AnchorChipWs *hdr = reinterpret_cast<const AnchorChipWs *>(w + L.hdr);
  const int nv = hdr->nvalid;
  if (nv == 0) return;
  const double *vbox = reinterpret_cast<const double *>(w + L.vbox);
  unsigned long long *gtmax = reinterpret_cast<unsigned long long *>(w + L.gtmax);
  __shared__ unsigned long long smax[kAnchorMaxG];
  for (int g = threadIdx.x; g < nv; g += blockDim.x) smax[g] = 0ull;
  __syncthreads();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < L.total) {
    double a[4];
    anchor_box(base, A, F, stride, idx, a);
    if (anchor_inside(a, im_h, im_w)) {
      for (int g = 0; g < nv; ++g) {
        const double ov = overlap_f64(a[0], a[1], a[2], a[3], vbox[4 * g], vbox[4 * g + 1], vbox[4 * g + 2],
                                      vbox[4 * g + 3], 0);
        // IoU >= 0, so the IEEE bit pattern orders like the value
        if (ov > 0.0) atomicMax(&smax[g], (unsigned long long)__double_as_longlong(ov));
      }
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < nv; g += blockDim.x)
    if (smax[g]) atomicMax(&gtmax[g], smax[g]);
}

// K3 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void anchor_label_kernel(const double *__restrict__ base, int A, int F, int stride,
                                                           int im_h, int im_w, double pos_thresh, double neg_thresh,
                                                           char *__restrict__ ws, AnchorWsLayout L) {
  const int b = blockIdx.y;
  char *w = ws + (size_t)b * L.per_chip_total;
  const AnchorChipWs *hdr = reinterpret_cast<const AnchorChipWs *>(w + L.hdr);
  const int nv = hdr->nvalid, ni = hdr->ninvalid;
  const double *vbox = reinterpret_cast<const double *>(w + L.vbox);
  const double *ibox = reinterpret_cast<const double *>(w + L.ibox);
  const unsigned long long *gtmax = reinterpret_cast<const unsigned long long *>(w + L.gtmax);
  int16_t *argmax_o = reinterpret_cast<int16_t *>(w + L.argmax);
  int8_t *label_o = reinterpret_cast<int8_t *>(w + L.label_pre);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L.total) return;
  double a[4];
  anchor_box(base, A, F, stride, idx, a);
  if (!anchor_inside(a, im_h, im_w)) {
    label_o[idx] = -2;
    argmax_o[idx] = 0;
    return;
  }
  int label = -1, am = 0;
  if (nv > 0) {
    double mx = -1.0;
    bool hit = false;
    for (int g = 0; g < nv; ++g) {
      const double ov = overlap_f64(a[0], a[1], a[2], a[3], vbox[4 * g], vbox[4 * g + 1], vbox[4 * g + 2],
                                    vbox[4 * g + 3], 0);
      if (ov > mx) { mx = ov; am = g; }  // argmax: first maximum
      hit |= ((unsigned long long)__double_as_longlong(ov) == gtmax[g]);  // overlaps == gt_max_overlaps, :303
    }
    if (mx < neg_thresh) label = 0;   // :305
    if (hit) label = 1;               // :306
    if (mx >= pos_thresh) label = 1;  // :309
  } else {
    label = 0;  // :319
  }
  if (ni > 0) {
    double mn = 0.0;
    for (int g = 0; g < ni; ++g) {
      const double ov = overlap_f64(a[0], a[1], a[2], a[3], ibox[4 * g], ibox[4 * g + 1], ibox[4 * g + 2],
                                    ibox[4 * g + 3], 0);
      mn = ov > mn ? ov : mn;
    }
    if (mn > 0.3) label = -1;  // :311-317, :320-325
  }
  label_o[idx] = (int8_t)label;
  argmax_o[idx] = (int16_t)am;
}

// K4 ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned hash_key(unsigned long long seed, unsigned b, unsigned idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)b * 0x100000001ull + idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (unsigned)(z >> 32);
}

constexpr int kFinishThreads = 1024;

// Block-wide: among a population of 47-bit composites (key32 << 15 | idx), find the `keep`-th smallest.  Returns the threshold;
// entries with composite <= threshold are kept.  Requires keep >= 1 and keep < population.  MSB-first radix select, 8 bits per
// pass.  each(fn) calls fn(composite) for every member of the population this thread holds.
template <typename Each>
__device__ __forceinline__ unsigned long long radix_select_kth(Each each, int keep, unsigned *hist /*256, shared*/,
                                                               unsigned long long *sh_prefix, int *sh_keep) {
  const int tid = threadIdx.x;
  unsigned long long prefix = 0;  // bits decided so far (left-aligned in the 48-bit field)
  int k = keep;                   // rank (1-based) inside the current bucket
  for (int shift = 40; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += kFinishThreads) hist[i] = 0;
    __syncthreads();
    const unsigned long long hi_mask = shift == 40 ? 0ull : (~0ull << (shift + 8));
    each([&](unsigned long long comp) {
      if ((comp & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(comp >> shift) & 0xffu], 1u);
    });
    __syncthreads();
    // the bucket that holds rank k: first bin whose inclusive count reaches k.  One wave, four bins per lane, shuffle scan
    // (a single thread walking the 256 LDS bins took ~7 us per pass, 12 passes per chip: most of this kernel's 157 us)
    if (tid < 64) {
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned sum4 = h0 + h1 + h2 + h3;
      unsigned incl = sum4;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, 64);
        if (tid >= d) incl += up;
      }
      const unsigned excl = incl - sum4;
      if (excl < (unsigned)k && (unsigned)k <= incl) {       // exactly one lane: 1 <= k <= population of the current bucket
        unsigned acc = excl;
        int bin = 4 * tid;
        if (acc + h0 < (unsigned)k) { acc += h0; ++bin;
          if (acc + h1 < (unsigned)k) { acc += h1; ++bin;
            if (acc + h2 < (unsigned)k) { acc += h2; ++bin; } } }
        const unsigned in_bin = bin == 4 * tid ? h0 : (bin == 4 * tid + 1 ? h1 : (bin == 4 * tid + 2 ? h2 : h3));
        unsigned long long np = prefix | ((unsigned long long)bin << shift);
        int nk = k - (int)acc;
        if (in_bin == (unsigned)nk) {       // the whole bucket is kept: every composite with this prefix is <= prefix | ones -- done
          np |= shift ? ((1ull << shift) - 1ull) : 0ull;          // (the keys are 32 random bits: two or three passes, not six)
          nk = -1;
        }
        *sh_prefix = np;
        *sh_keep = nk;
      }
    }
    __syncthreads();
    prefix = *sh_prefix;
    k = *sh_keep;
    __syncthreads();
    if (k < 0) break;
  }
  return prefix;
}

// Round 4: the chip's (label, composite key) pairs are read / hashed ONCE into registers (21 per thread for the training map's
// 21 504 anchors); the counting pass, the up to twelve radix passes of the two selections and the output pass run on them.
// Each pass used to re-read the 1-byte labels with a dependent L2 load per iteration and re-hash every key (124 us per 20 chips).
constexpr int kFinishReg = 24;       // anchors per thread held in registers (x 1024 threads); larger maps stream

__global__ __launch_bounds__(kFinishThreads) void anchor_finish_kernel(
    const double *__restrict__ base, int A, int F, int stride, int rpn_batch, int num_fg,
    const uint32_t *__restrict__ keys, unsigned long long seed, char *__restrict__ ws, AnchorWsLayout L,
    float *__restrict__ label, float *__restrict__ bbox_target, float *__restrict__ bbox_weight,
    int32_t *__restrict__ counts, int8_t *__restrict__ label_pre_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  char *w = ws + (size_t)b * L.per_chip_total;
  const AnchorChipWs *hdr = reinterpret_cast<const AnchorChipWs *>(w + L.hdr);
  const double *vbox = reinterpret_cast<const double *>(w + L.vbox);
  const int16_t *argmax_i = reinterpret_cast<const int16_t *>(w + L.argmax);
  const int8_t *lp = reinterpret_cast<const int8_t *>(w + L.label_pre);
  const int total = L.total;
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh_prefix;
  __shared__ int sh_keep;
  __shared__ int cnt[3];
  if (tid < 3) cnt[tid] = 0;
  __syncthreads();
  const bool in_regs = total <= kFinishReg * kFinishThreads;
  // creg[u]: anchor idx = u * 1024 + tid: bits 0-46 the composite (key32 << 15 | idx), bits 60-61 the class (1 fg, 2 bg, 0 neither)
  unsigned long long creg[kFinishReg];
  auto composite = [&](int idx) {
    const unsigned kk = keys ? keys[(size_t)b * total + idx] : hash_key(seed, b, idx);
    return ((unsigned long long)kk << 15) | (unsigned)idx;
  };
  int c_in = 0, c_fg = 0, c_bg = 0;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < kFinishReg; ++u) {
      const int idx = u * kFinishThreads + tid;
      creg[u] = 0ull;
      if (idx < total) {
        const int l = lp[idx];
        c_in += (l != -2);
        c_fg += (l == 1);
        c_bg += (l == 0);
        if (label_pre_out) label_pre_out[(size_t)b * total + idx] = (int8_t)l;
        if (l == 1 || l == 0) creg[u] = composite(idx) | ((unsigned long long)(l == 1 ? 1 : 2) << 60);
      }
    }
  } else {
    for (int idx = tid; idx < total; idx += kFinishThreads) {
      const int l = lp[idx];
      c_in += (l != -2);
      c_fg += (l == 1);
      c_bg += (l == 0);
      if (label_pre_out) label_pre_out[(size_t)b * total + idx] = (int8_t)l;
    }
  }
  atomicAdd(&cnt[0], c_in); atomicAdd(&cnt[1], c_fg); atomicAdd(&cnt[2], c_bg);
  __syncthreads();
  // the population of class `which` (1 = fg, 0 = bg) held by this thread
  auto select = [&](int which, int keep) {
    const unsigned long long tag = (unsigned long long)(which == 1 ? 1 : 2) << 60;
    return radix_select_kth(
        [&](auto fn) {
          if (in_regs) {
#pragma unroll
            for (int u = 0; u < kFinishReg; ++u)
              if ((creg[u] >> 60) == (tag >> 60)) fn(creg[u] & ((1ull << 47) - 1ull));
          } else {
            for (int idx = tid; idx < total; idx += kFinishThreads)
              if (lp[idx] == which) fn(composite(idx));
          }
        },
        keep, hist, &sh_prefix, &sh_keep);
  };
  const int n_fg = cnt[1], n_bg = cnt[2];
  if (tid == 0) {
    counts[4 * b + 0] = cnt[0]; counts[4 * b + 1] = n_fg; counts[4 * b + 2] = n_bg; counts[4 * b + 3] = hdr->nvalid;
  }
  // data_workers.py:327-331 -- keep at most num_fg foreground anchors
  bool sub_fg = n_fg > num_fg;
  unsigned long long thr_fg = ~0ull, thr_bg = ~0ull;
  if (sub_fg) thr_fg = select(1, num_fg);
  // :333-338 -- background fills the rest of the batch
  const int fg_after = sub_fg ? num_fg : n_fg;
  const int num_bg = rpn_batch - fg_after;
  const bool sub_bg = n_bg > num_bg;
  bool drop_all_bg = false;
  if (sub_bg) {
    if (num_bg <= 0) drop_all_bg = true;
    else thr_bg = select(0, num_bg);
  }
  // dense outputs in the reference's batch layouts; o walks the (a, y, x) label order
  const int FF = F * F;
  for (int o = tid; o < total; o += kFinishThreads) {
    const int an = o / FF, cell = o - an * FF;
    const int idx = cell * A + an;
    int l = lp[idx];
    if (l == -2) l = -1;  // _unmap fill, :347
    if (l == 1 || l == 0) {
      const unsigned kk = keys ? keys[(size_t)b * total + idx] : hash_key(seed, b, idx);
      const unsigned long long comp = ((unsigned long long)kk << 15) | (unsigned)idx;
      if (l == 1 && sub_fg && comp > thr_fg) l = -1;
      if (l == 0 && sub_bg && (drop_all_bg || comp > thr_bg)) l = -1;
    }
    label[(size_t)b * total + o] = (float)l;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    const float wv = l == 1 ? 1.f : 0.f;
    if (l == 1 && hdr->nvalid > 0) {
      double a[4];
      anchor_box(base, A, F, stride, idx, a);
      const double *g = vbox + 4 * argmax_i[idx];
      // nonlinear_transform, lib/bbox/bbox_transform.py:64-90, float64 then stored as float32
      const double ew = a[2] - a[0] + 1.0, eh = a[3] - a[1] + 1.0;
      const double ecx = a[0] + 0.5 * (ew - 1.0), ecy = a[1] + 0.5 * (eh - 1.0);
      const double gw = g[2] - g[0] + 1.0, gh = g[3] - g[1] + 1.0;
      const double gcx = g[0] + 0.5 * (gw - 1.0), gcy = g[1] + 0.5 * (gh - 1.0);
      t[0] = (float)((gcx - ecx) / (ew + 1e-7));
      t[1] = (float)((gcy - ecy) / (eh + 1e-7));
      t[2] = (float)log(gw / (ew + 1e-7));
      t[3] = (float)log(gh / (eh + 1e-7));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t off = ((size_t)b * 4 * A + 4 * an + c) * FF + cell;
      bbox_target[off] = t[c];
      bbox_weight[off] = wv;
    }
  }
}

SN_EXPORT int sn_anchor_assign(const float *d_gt, const float *d_gt_cls, const uint8_t *d_gt_inchip,
                               const int32_t *d_ngt, const double *d_crop, const float *d_scale, int B, int G,
                               const double *d_base_anchors, int A, int F, int feat_stride, int im_h, int im_w,
                               double pos_thresh, double neg_thresh, int rpn_batch, int num_fg,
                               const uint32_t *d_keys, uint64_t seed, void *d_ws, float *d_label,
                               float *d_bbox_target, float *d_bbox_weight, float *d_gt_out, int32_t *d_counts,
                               int8_t *d_label_pre, sn_stream_t stream) {
  SN_REQUIRE(B >= 0 && G >= 1 && G <= kAnchorMaxG, "sn_anchor_assign: G must be in [1,%d] (got %d)", kAnchorMaxG, G);
  SN_REQUIRE(A >= 1 && F >= 1 && A * F * F < 32768, "sn_anchor_assign: A*F*F=%d must be < 32768", A * F * F);
  SN_REQUIRE(rpn_batch > 0 && num_fg > 0 && num_fg <= rpn_batch, "sn_anchor_assign: bad batch sizes");
  if (B == 0) return SN_OK;
  SN_REQUIRE(d_gt && d_gt_cls && d_gt_inchip && d_ngt && d_crop && d_scale && d_base_anchors && d_ws && d_label &&
                 d_bbox_target && d_bbox_weight && d_gt_out && d_counts,
             "sn_anchor_assign: null pointer");
  const AnchorWsLayout L = anchor_ws_layout(A, F, G);
  hipStream_t s = sn_stream(stream);
  hipLaunchKernelGGL(anchor_prep_kernel, dim3(B), dim3(kAnchorMaxG), 0, s, d_gt, d_gt_cls, d_gt_inchip, d_ngt, d_crop,
                     d_scale, G, im_h, im_w, (char *)d_ws, L, d_gt_out);
  SN_CHECK_LAUNCH();
  const dim3 grid(sn_div_up(L.total, 256), B);
  hipLaunchKernelGGL(anchor_gtmax_kernel, grid, dim3(256), 0, s, d_base_anchors, A, F, feat_stride, im_h, im_w,
                     (char *)d_ws, L);
  SN_CHECK_LAUNCH();
  hipLaunchKernelGGL(anchor_label_kernel, grid, dim3(256), 0, s, d_base_anchors, A, F, feat_stride, im_h, im_w,
                     pos_thresh, neg_thresh, (char *)d_ws, L);
  SN_CHECK_LAUNCH();
  hipLaunchKernelGGL(anchor_finish_kernel, dim3(B), dim3(kFinishThreads), 0, s, d_base_anchors, A, F, feat_stride,
                     rpn_batch, num_fg, d_keys, (unsigned long long)seed, (char *)d_ws, L, d_label, d_bbox_target,
                     d_bbox_weight, d_counts, d_label_pre);
  SN_CHECK_LAUNCH();
  return SN_OK;
}


// ============================================================================================
// AutoFocus FocusPixel labels: gen_mask (lib/data_utils/data_workers.py:165-192) for B chips.
// Per chip the GT boxes get the same preparation as anchor_prep_kernel (:203-217: shift by the crop origin in
// double, round(gt * scale) in float32, clip); every box -- the < 10 px ones included, gen_mask runs before
// filter_boxes -- then paints the feature cells [int(x1/s), min(int(ceil(x2/s)) + 1, F)) x (same in y) with
//   +1  dc_low < sqrt(w*h) < small        -1  small <= sqrt(w*h) < dc_high  or  sqrt(w*h) <= dc_low        (else nothing)
// in box order, later boxes overwriting earlier ones.  One thread per cell walks the boxes in order and keeps the
// last one that paints it: same result, no write races.
// ============================================================================================
__global__ __launch_bounds__(256) void focus_mask_kernel(const float *__restrict__ gt, const int32_t *__restrict__ ngt,
                                                         const double *__restrict__ crop, const float *__restrict__ scale, int G,
                                                         int F, int feat_stride, int im_h, int im_w, float dc_low, float small,
                                                         float dc_high, float *__restrict__ out) {
  __shared__ int sx1[kAnchorMaxG], sy1[kAnchorMaxG], sx2[kAnchorMaxG], sy2[kAnchorMaxG];
  __shared__ signed char sflag[kAnchorMaxG];
  const int b = blockIdx.x;
  const int n = min(ngt[b], G);
  for (int g = threadIdx.x; g < n; g += blockDim.x) {
    const float *p = gt + ((size_t)b * G + g) * 4;
    const double cx = crop[2 * b], cy = crop[2 * b + 1];
    const float s = scale[b];
    float x1 = (float)((double)p[0] - cx), y1 = (float)((double)p[1] - cy);
    float x2 = (float)((double)p[2] - cx), y2 = (float)((double)p[3] - cy);
    x1 = rintf(x1 * s); y1 = rintf(y1 * s); x2 = rintf(x2 * s); y2 = rintf(y2 * s);
    const float wm = (float)(im_w - 1), hm = (float)(im_h - 1);
    x1 = fmaxf(fminf(x1, wm), 0.f); y1 = fmaxf(fminf(y1, hm), 0.f);
    x2 = fmaxf(fminf(x2, wm), 0.f); y2 = fmaxf(fminf(y2, hm), 0.f);
    const float area = sqrtf((x2 - x1) * (y2 - y1));
    int flag = 0;
    if (area > dc_low && area < small) flag = 1;
    else if (area >= small && area < dc_high) flag = -1;
    else if (area <= dc_low) flag = -1;
    const float fs = (float)feat_stride;
    sx1[g] = (int)(x1 / fs);
    sy1[g] = (int)(y1 / fs);
    sx2[g] = min((int)ceilf(x2 / fs) + 1, F);
    sy2[g] = min((int)ceilf(y2 / fs) + 1, F);
    sflag[g] = (signed char)flag;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < F * F; c += blockDim.x) {
    const int y = c / F, x = c - y * F;
    float v = 0.f;
    for (int g = 0; g < n; ++g)
      if (sflag[g] != 0 && x >= sx1[g] && x < sx2[g] && y >= sy1[g] && y < sy2[g]) v = (float)sflag[g];
    out[(size_t)b * F * F + c] = v;
  }
}

SN_EXPORT int sn_focus_mask(const float *d_gt, const int32_t *d_ngt, const double *d_crop, const float *d_scale, int B, int G, int F,
                            int feat_stride, int im_h, int im_w, float dc_low, float small_thresh, float dc_high, float *d_mask,
                            sn_stream_t stream) {
  SN_REQUIRE(d_gt && d_ngt && d_crop && d_scale && d_mask && B > 0 && G > 0 && G <= kAnchorMaxG && F > 0 && feat_stride > 0,
             "sn_focus_mask: bad arguments (G <= %d)", kAnchorMaxG);
  hipLaunchKernelGGL(focus_mask_kernel, dim3(B), dim3(256), 0, sn_stream(stream), d_gt, d_ngt, d_crop, d_scale, G, F, feat_stride,
                     im_h, im_w, dc_low, small_thresh, dc_high, d_mask);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
