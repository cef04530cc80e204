// nms.hip -- wave64 bitmask hard NMS for gfx950, batched, mask + scan both on device.
//
// Algorithm of the reference's only CUDA kernel (lib/nms/nms_kernel.cu:34-78) and its host scan
// (:118-140): boxes sorted by descending score; mask[i][w] bit j set iff IoU(box i, box 64w+j) >
// thresh and 64w+j > i; a box survives iff no earlier survivor has its bit set.
// What is different here (MI355X-first, not a translation):
//   * one 64-lane block IS one wavefront: a row box per lane, 64 column boxes staged in LDS;
//     only the upper triangle of 64x64 tiles is launched (the reference computes all N^2 pairs,
//     nms_kernel.cu:39 commented out) -> 18M instead of 36M IoUs at N=6000;
//   * the keep/remove scan runs on the GPU (one workgroup per image): no 4.5 MB mask D2H, no
//     per-call hipMalloc (nms_kernel.cu:100-108 is the anti-pattern); 64 rows are resolved per step
//     with ballot/readlane on the diagonal word, and the scan stops after max_keep survivors;
//   * batched over images (the proposal op runs it for every chip of the minibatch in one launch).
// float32 arithmetic identical to devIoU (nms_kernel.cu:24-32): compiled with -ffp-contract=off.
#include "common.h"

#include <stdlib.h>

__device__ __forceinline__ float dev_iou(const float *a, const float *b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  const float interS = width * height;
  const float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  const float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

// grid = (col_blocks, row_blocks, B), block = 64 (one wave).
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, const int32_t *__restrict__ n_per,
                                                      int N, int dim, float thresh, int col_blocks,
                                                      unsigned long long *__restrict__ mask) {
  const int col_start = blockIdx.x, row_start = blockIdx.y, b = blockIdx.z;
  if (row_start > col_start) return;  // strictly-lower tiles are never read by the scan
  const int n = n_per ? min(n_per[b], N) : N;
  if (row_start * 64 >= n) return;
  const float *bb = boxes + (size_t)b * N * dim;
  const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
  __shared__ float cb[64 * 4];
  const int t = threadIdx.x;
  if (t < col_size) {
    const float *p = bb + (size_t)(64 * col_start + t) * dim;
    cb[t * 4 + 0] = p[0]; cb[t * 4 + 1] = p[1]; cb[t * 4 + 2] = p[2]; cb[t * 4 + 3] = p[3];
  }
  __syncthreads();
  if (t < row_size) {
    const int cur = 64 * row_start + t;
    const float *p = bb + (size_t)cur * dim;
    const float me[4] = {p[0], p[1], p[2], p[3]};
    unsigned long long bits = 0;
    const int start = (row_start == col_start) ? t + 1 : 0;
    for (int i = start; i < col_size; ++i)
      if (dev_iou(me, cb + i * 4) > thresh) bits |= 1ull << i;
    mask[((size_t)b * N + cur) * col_blocks + col_start] = bits;
  }
}

// One workgroup per image.  remv (suppressed bitmap) in LDS; rows are consumed 64 at a time:
// wave 0 resolves the chunk sequentially on its diagonal words (scalar ballot/readlane work),
// then all waves OR the surviving rows' mask words into remv.
constexpr int kScanThreads = 256;

__global__ __launch_bounds__(kScanThreads) void nms_scan_kernel(const unsigned long long *__restrict__ mask,
                                                                const int32_t *__restrict__ n_per, int N,
                                                                int col_blocks, int max_keep,
                                                                int32_t *__restrict__ keep, int32_t *__restrict__ nkeep) {
  // all LDS in the dynamic region (16-byte aligned base, cdna guide G17): remv[col_blocks], kept, nk
  extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];
  unsigned long long &kept_s = remv[col_blocks];
  int &nk_s = *reinterpret_cast<int *>(&remv[col_blocks + 1]);
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_per ? min(n_per[b], N) : N;
  const unsigned long long *m = mask + (size_t)b * N * col_blocks;
  int32_t *kp = keep + (size_t)b * max_keep;
  for (int w = tid; w < col_blocks; w += kScanThreads) remv[w] = 0ull;
  if (tid == 0) nk_s = 0;
  __syncthreads();
  const int nchunks = (n + 63) / 64;
  for (int c = 0; c < nchunks; ++c) {
    if (tid < 64) {  // wave 0
      const int row = c * 64 + tid;
      const bool in = row < n;
      const unsigned long long diag = in ? m[(size_t)row * col_blocks + c] : 0ull;
      const unsigned long long alive = __ballot(in) & ~remv[c];
      unsigned long long sup = 0ull, kept = 0ull;
      int nk = nk_s;
      for (int r = 0; r < 64; ++r) {  // wave-uniform loop
        const unsigned long long d =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(diag >> 32), r) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)(diag & 0xffffffffu), r);
        const bool take = ((alive >> r) & 1ull) && !((sup >> r) & 1ull) && nk < max_keep;
        if (take) {
          kept |= 1ull << r;
          sup |= d;
          if (tid == 0) kp[nk] = c * 64 + r;
          ++nk;
        }
      }
      if (tid == 0) {
        kept_s = kept;
        nk_s = nk;
      }
    }
    __syncthreads();
    const unsigned long long kept = kept_s;
    const bool done = nk_s >= max_keep;
    if (kept != 0ull && !done) {
      for (int w = c + tid; w < col_blocks; w += kScanThreads) {
        unsigned long long acc = 0ull, k = kept;
        while (k) {
          const int r = __ffsll((long long)k) - 1;
          k &= k - 1;
          acc |= m[(size_t)(c * 64 + r) * col_blocks + w];
        }
        remv[w] |= acc;
      }
    }
    __syncthreads();
    if (done) break;
  }
  if (tid == 0) nkeep[b] = nk_s;
}

// ---- lazy variant for max_keep << N (MultiProposal / MultiProposalTarget: the first 300 survivors of 6000 sorted boxes).
// The full bitmask is 18 M IoUs and 4.5 MB per image of which the scan then reads the rows of 300 survivors; here one
// workgroup per image walks the candidates 64 at a time and stops at max_keep survivors:
//   1. every candidate of the chunk against the survivors so far (kept boxes live in LDS; wave w takes survivors w, w+4, ...,
//      lane = candidate, ballot -> 64-bit "dead" word per wave);
//   2. the 64 x 64 upper triangle inside the chunk (wave w computes columns 16 w .. 16 w + 15 of every row);
//   3. wave 0 resolves the chunk sequentially on those diagonal words exactly as nms_scan_kernel does and appends the new
//      survivors (boxes to LDS, indices to `keep`).
// Same predicate (dev_iou(a, b) > thresh, symmetric bit for bit), same greedy order -> the same survivor list as the
// mask + scan pair; work ~ (candidates examined) x (survivors) instead of N^2 / 2.
constexpr int kLazyMaxKeep = 1024;
// Round 4: 16 waves per image instead of 4 (one workgroup per image leaves 236 of 256 CUs idle whatever it does, so the only lever
// is the latency of ONE workgroup: the survivor sweep of step 1 and the triangle of step 2 are split over four times the waves),
// and step 3 visits only the candidates that are still alive (lowest set bit of `alive & ~suppressed`, one iteration per new
// survivor) instead of all 64 rows: 259 us -> see profiles/r04_*.
constexpr int kLazyThreads = 1024, kLazyWaves = kLazyThreads / 64, kLazyCols = 64 / kLazyWaves;

__global__ __launch_bounds__(kLazyThreads) void nms_lazy_kernel(const float *__restrict__ boxes, const int32_t *__restrict__ n_per,
                                                                int N, int dim, float thresh, int max_keep,
                                                                int32_t *__restrict__ keep, int32_t *__restrict__ nkeep) {
  __shared__ float kb[kLazyMaxKeep * 4];
  __shared__ float cand[64 * 4];
  __shared__ unsigned long long dead_w[kLazyWaves];
  __shared__ unsigned dpart[kLazyWaves][64];
  __shared__ int nk_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = n_per ? min(n_per[b], N) : N;
  const float *bb = boxes + (size_t)b * N * dim;
  int32_t *kp = keep + (size_t)b * max_keep;
  if (tid == 0) nk_s = 0;
  __syncthreads();
  const int nchunks = (n + 63) / 64;
  for (int c = 0; c < nchunks; ++c) {
    const int nk = nk_s;
    if (nk >= max_keep) break;
    const int row = c * 64 + lane, csize = min(n - c * 64, 64);
    if (wave == 0 && lane < csize) {
      const float *p = bb + (size_t)row * dim;
      cand[lane * 4 + 0] = p[0]; cand[lane * 4 + 1] = p[1]; cand[lane * 4 + 2] = p[2]; cand[lane * 4 + 3] = p[3];
    }
    __syncthreads();
    const bool in = lane < csize;
    float me[4] = {0.f, 0.f, 0.f, 0.f};
    if (in) { me[0] = cand[lane * 4]; me[1] = cand[lane * 4 + 1]; me[2] = cand[lane * 4 + 2]; me[3] = cand[lane * 4 + 3]; }
    // 1. suppressed by an earlier survivor?  (wave w takes survivors w, w + 16, ...)
    bool dead = false;
    for (int k = wave; k < nk; k += kLazyWaves) dead = dead || (dev_iou(kb + k * 4, me) > thresh);
    const unsigned long long dw = __ballot(dead && in);
    if (lane == 0) dead_w[wave] = dw;
    // 2. this wave's columns of the chunk's upper triangle
    unsigned bits = 0u;
    if (in) {
#pragma unroll
      for (int jj = 0; jj < kLazyCols; ++jj) {
        const int j = wave * kLazyCols + jj;
        if (j > lane && j < csize && dev_iou(me, cand + j * 4) > thresh) bits |= 1u << jj;
      }
    }
    dpart[wave][lane] = bits;
    __syncthreads();
    // 3. sequential resolution of the chunk (wave 0), as in nms_scan_kernel: walk the candidates still alive in order
    if (wave == 0) {
      unsigned long long diag = 0ull, dead_all = 0ull;
#pragma unroll
      for (int w = 0; w < kLazyWaves; ++w) {
        diag |= (unsigned long long)dpart[w][lane] << (w * kLazyCols);
        dead_all |= dead_w[w];
      }
      unsigned long long avail = __ballot(in) & ~dead_all, kept = 0ull;
      int nk2 = nk;
      while (avail != 0ull && nk2 < max_keep) {      // wave-uniform loop: one iteration per new survivor
        const int r = __ffsll((long long)avail) - 1;
        const unsigned long long d =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(diag >> 32), r) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)(diag & 0xffffffffu), r);
        kept |= 1ull << r;
        avail &= ~(d | (1ull << r));       // row r's bits are columns j > r: everything it suppresses leaves the list
        ++nk2;
      }
      if ((kept >> lane) & 1ull) {
        const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
        kb[pos * 4 + 0] = me[0]; kb[pos * 4 + 1] = me[1]; kb[pos * 4 + 2] = me[2]; kb[pos * 4 + 3] = me[3];
        kp[pos] = row;
      }
      if (lane == 0) nk_s = nk2;
    }
    __syncthreads();
  }
  if (tid == 0) nkeep[b] = nk_s;
}

SN_EXPORT size_t sn_nms_workspace_bytes(int B, int N) {
  const size_t cb = (size_t)sn_div_up(N > 0 ? N : 1, 64);
  return sn_align((size_t)(B > 0 ? B : 1) * (size_t)(N > 0 ? N : 1) * cb * sizeof(unsigned long long));
}

SN_EXPORT int sn_nms_batch(const float *d_boxes, const int32_t *d_n, int B, int N, int dim, float thresh, int max_keep,
                           void *d_ws, int32_t *d_keep, int32_t *d_nkeep, sn_stream_t stream) {
  SN_REQUIRE(B >= 0 && N >= 0 && dim >= 4, "sn_nms_batch: bad sizes B=%d N=%d dim=%d", B, N, dim);
  if (B == 0) return SN_OK;
  SN_REQUIRE(d_nkeep, "sn_nms_batch: null d_nkeep");
  if (max_keep <= 0 || max_keep > N) max_keep = N;
  hipStream_t s = sn_stream(stream);
  if (N == 0) {
    SN_HIP(hipMemsetAsync(d_nkeep, 0, sizeof(int32_t) * B, s));
    return SN_OK;
  }
  SN_REQUIRE(d_boxes && d_keep, "sn_nms_batch: null pointer");
  if (max_keep <= kLazyMaxKeep && max_keep * 4 <= N && !sn_debug_get(SN_OPT_NMS_FULL_MASK)) {
    hipLaunchKernelGGL(nms_lazy_kernel, dim3(B), dim3(kLazyThreads), 0, s, d_boxes, d_n, N, dim, thresh, max_keep, d_keep, d_nkeep);
    SN_CHECK_LAUNCH();
    return SN_OK;
  }
  SN_REQUIRE(d_ws, "sn_nms_batch: the full-mask path needs sn_nms_workspace_bytes(B, N) of scratch");
  const int cb = sn_div_up(N, 64);
  SN_REQUIRE((size_t)cb * sizeof(unsigned long long) <= 64 * 1024, "sn_nms_batch: N=%d too large for the LDS scan", N);
  // The scan reads mask words of rows it keeps for columns >= the row's own block; every such
  // word is written by nms_mask_kernel (upper triangle incl. diagonal), so no memset is needed.
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, B), dim3(64), 0, s, d_boxes, d_n, N, dim, thresh, cb,
                     (unsigned long long *)d_ws);
  SN_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(kScanThreads), (size_t)(cb + 2) * sizeof(unsigned long long), s,
                     (const unsigned long long *)d_ws, d_n, N, cb, max_keep, d_keep, d_nkeep);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// Drop-in for lib/nms/gpu_nms.hpp:1-2 (host buffers in, host keep list out).  Synchronous and
// self-allocating like the original; the training/inference graph uses sn_nms_batch instead.
SN_EXPORT int sn_nms_host(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                          float nms_overlap_thresh, int device_id) {
  SN_REQUIRE(keep_out && num_out && boxes_num >= 0 && boxes_dim >= 4, "sn_nms_host: bad arguments");
  *num_out = 0;
  if (boxes_num == 0) return SN_OK;
  SN_REQUIRE(boxes_host, "sn_nms_host: null boxes");
  int prev = 0;
  SN_HIP(hipGetDevice(&prev));
  if (prev != device_id) SN_HIP(hipSetDevice(device_id));
  float *d_boxes = nullptr;
  void *d_ws = nullptr;
  int32_t *d_keep = nullptr, *d_nk = nullptr;
  int rc = SN_OK;
  hipError_t e;
  const size_t bytes = (size_t)boxes_num * boxes_dim * sizeof(float);
  if ((e = hipMalloc(&d_boxes, bytes)) != hipSuccess || (e = hipMalloc(&d_ws, sn_nms_workspace_bytes(1, boxes_num))) != hipSuccess ||
      (e = hipMalloc(&d_keep, sizeof(int32_t) * boxes_num)) != hipSuccess || (e = hipMalloc(&d_nk, sizeof(int32_t))) != hipSuccess) {
    sn_set_error("sn_nms_host: hipMalloc failed: %s", hipGetErrorString(e));
    rc = SN_ERR_HIP;
  }
  if (rc == SN_OK && (e = hipMemcpy(d_boxes, boxes_host, bytes, hipMemcpyHostToDevice)) != hipSuccess) rc = SN_ERR_HIP;
  if (rc == SN_OK) rc = sn_nms_batch(d_boxes, nullptr, 1, boxes_num, boxes_dim, nms_overlap_thresh, boxes_num, d_ws, d_keep, d_nk, nullptr);
  if (rc == SN_OK && (e = hipMemcpy(num_out, d_nk, sizeof(int32_t), hipMemcpyDeviceToHost)) != hipSuccess) rc = SN_ERR_HIP;
  if (rc == SN_OK && *num_out > 0 &&
      (e = hipMemcpy(keep_out, d_keep, sizeof(int32_t) * (size_t)*num_out, hipMemcpyDeviceToHost)) != hipSuccess)
    rc = SN_ERR_HIP;
  if (rc == SN_ERR_HIP && e != hipSuccess) sn_set_error("sn_nms_host: %s", hipGetErrorString(e));
  (void)hipFree(d_boxes); (void)hipFree(d_ws); (void)hipFree(d_keep); (void)hipFree(d_nk);
  if (prev != device_id) (void)hipSetDevice(prev);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// Soft-NMS (lib/nms/cpu_nms.pyx:17-110): the test-time per-class suppression (`TEST.NMS_SIGMA`, gaussian method 2),
// 80 classes x images of independent problems on a Pool(32) in the reference (lib/inference.py:152-230).
//
// The reference algorithm is sequential and ORDER DEPENDENT (in-place selection sort; a box whose decayed score
// drops below `threshold` is overwritten by the last box and N shrinks), so a GPU version that wants the same rows in
// the same order has to reproduce the array permutation, not just the arithmetic.  One workgroup per problem, the
// problem's (n,5) array in LDS, and per outer iteration i three parallel phases that are provably the sequential
// pass:
//   1. arg-max of the scores in [i, N), first position on ties (the reference's strict `<` scan) -> swap into i;
//   2. every box in (i, N) is decayed exactly once by the reference pass, whatever the removal order -> all
//      weights in parallel, same float expressions (compiled -ffp-contract=off; the gaussian weight is
//      exp() in double of the float quotient, narrowed once, as `np.exp` on a C float does);
//   3. the removals: "overwrite the dead box with the last one, shrink, re-examine" fills the dead slots below the
//      new N in INCREASING position order with the surviving tail boxes in DECREASING position order -> ranks from
//      a block-wide prefix sum of the keep flags, one parallel move.
// LDS: 25 bytes per box (array + flags + tail list): problems of n <= 4096 boxes run out of LDS; larger ones (the reference has
// no cap) run the same phases on their rows in global memory with the flags / tail list in a caller-provided workspace.
// ---------------------------------------------------------------------------------------------
constexpr int kSoftThreads = 256;
constexpr int kSoftLdsBoxes = 4096;

// One problem.  IN_LDS: `bx` / `tail` / `keepf` are the workgroup's LDS image of the problem (copied in and out by the caller);
// otherwise they are global memory -- the problem's own rows, updated in place, and its slice of the caller's workspace -- for
// problems beyond the LDS capacity.  Same phases, same arithmetic: a workgroup barrier orders the workgroup's global accesses too.
template <bool IN_LDS>
__device__ __forceinline__ int soft_nms_problem(float *__restrict__ bx, int *__restrict__ tail, unsigned char *__restrict__ keepf,
                                                int N, float sigma, float Nt, float threshold, int method, float *r_score,
                                                int *r_pos, int *cnt) {
  const int t = threadIdx.x;
  for (int i = 0; i < N; ++i) {
    // ---- 1. arg-max over [i, N), smallest position among equals
    const int len = N - i, per = (len + kSoftThreads - 1) / kSoftThreads;
    const int lo = i + t * per, hi = min(N, lo + per);
    float best = -INFINITY;
    int bpos = 0x7fffffff;
    for (int q = lo; q < hi; ++q) {
      const float s = bx[q * 5 + 4];
      if (bpos == 0x7fffffff || best < s) { best = s; bpos = q; }
    }
    r_score[t] = best;
    r_pos[t] = bpos;
    __syncthreads();
    for (int w = kSoftThreads / 2; w > 0; w >>= 1) {
      if (t < w) {
        const float s2 = r_score[t + w];
        const int p2 = r_pos[t + w];
        // first position among equal scores (an interleaved tree does not keep the halves in position order)
        if (p2 != 0x7fffffff && (r_pos[t] == 0x7fffffff || r_score[t] < s2 || (r_score[t] == s2 && p2 < r_pos[t]))) {
          r_score[t] = s2;
          r_pos[t] = p2;
        }
      }
      __syncthreads();
    }
    const int maxpos = r_pos[0];
    __syncthreads();
    if (t < 5 && maxpos != i) {
      const float a = bx[i * 5 + t], b = bx[maxpos * 5 + t];
      bx[i * 5 + t] = b;
      bx[maxpos * 5 + t] = a;
    }
    __syncthreads();
    const float tx1 = bx[i * 5], ty1 = bx[i * 5 + 1], tx2 = bx[i * 5 + 2], ty2 = bx[i * 5 + 3];
    // ---- 2. decay every box in (i, N)
    const int len2 = N - i - 1, per2 = (len2 + kSoftThreads - 1) / kSoftThreads;
    const int lo2 = i + 1 + t * per2, hi2 = min(N, lo2 + per2);
    int kept = 0;
    for (int q = lo2; q < hi2; ++q) {
      const float x1 = bx[q * 5], y1 = bx[q * 5 + 1], x2 = bx[q * 5 + 2], y2 = bx[q * 5 + 3];
      bool keep = true;
      // Cython turns the `+ 1` beside C floats into the double literal 1.0 (the reference's own generated
      // lib/nms/cpu_nms.c:2946-3036): differences are float, the `+ 1.0`, the products and the three-term sum of `ua` are
      // double, rounded once when stored in a float variable.  iw / ih / the linear weight round like float arithmetic.
      const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
      const float iw = fminf(tx2, x2) - fmaxf(tx1, x1) + 1;
      if (iw > 0) {
        const float ih = fminf(ty2, y2) - fmaxf(ty1, y1) + 1;
        if (ih > 0) {
          const float ua = (float)(((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0) + (double)area - (double)(iw * ih));
          const float ov = iw * ih / ua;
          float weight;
          if (method == 1) weight = ov > Nt ? 1 - ov : 1;
          else if (method == 2) weight = (float)exp((double)(-(ov * ov) / sigma));
          else weight = ov > Nt ? 0 : 1;
          const float ns = weight * bx[q * 5 + 4];
          bx[q * 5 + 4] = ns;
          keep = !(ns < threshold);
        }
      }
      keepf[q] = keep ? 1 : 0;
      kept += keep ? 1 : 0;
    }
    cnt[t + 1] = kept;
    if (t == 0) cnt[0] = 0;
    __syncthreads();
    // ---- 3. inclusive scan of the per-thread keep counts (Hillis-Steele over 256 entries)
    for (int d = 1; d < kSoftThreads; d <<= 1) {
      int v = 0;
      if (t + 1 > d) v = cnt[t + 1 - d];
      __syncthreads();
      if (t + 1 > d) cnt[t + 1] += v;
      __syncthreads();
    }
    const int K = cnt[kSoftThreads];
    const int Nn = i + 1 + K;
    if (Nn < N) {
      // surviving boxes at positions >= Nn, ranked from the end -> tail[rank]
      int kb = cnt[t];   // kept boxes before this thread's chunk
      for (int q = lo2; q < hi2; ++q) {
        if (keepf[q]) {
          if (q >= Nn) tail[K - kb - 1] = q;
          ++kb;
        }
      }
      __syncthreads();
      kb = cnt[t];
      for (int q = lo2; q < hi2; ++q) {
        if (keepf[q]) { ++kb; continue; }
        if (q < Nn) {
          const int hole_rank = (q - (i + 1)) - kb;   // dead boxes before q
          const int src = tail[hole_rank];
#pragma unroll
          for (int c = 0; c < 5; ++c) bx[q * 5 + c] = bx[src * 5 + c];
        }
      }
      __syncthreads();
      N = Nn;
    }
  }
  return N;
}

__global__ __launch_bounds__(kSoftThreads) void soft_nms_kernel(float *__restrict__ boxes, const int32_t *__restrict__ off,
                                                                float sigma, float Nt, float threshold, int method,
                                                                int32_t *__restrict__ count, int lds_n, unsigned char *__restrict__ ws,
                                                                long long total_rows) {
  extern __shared__ __attribute__((aligned(16))) float soft_smem[];
  __shared__ float r_score[kSoftThreads];
  __shared__ int r_pos[kSoftThreads];
  __shared__ int cnt[kSoftThreads + 1];
  const int p = blockIdx.x, t = threadIdx.x;
  const int base = off[p];
  int N = off[p + 1] - base;
  float *g = boxes + (size_t)base * 5;
  if (N <= lds_n) {
    float *bx = soft_smem;                                           // [lds_n][5]
    int *tail = reinterpret_cast<int *>(bx + (size_t)lds_n * 5);     // [lds_n]
    unsigned char *keepf = reinterpret_cast<unsigned char *>(tail + lds_n);   // [lds_n]
    for (int k = t; k < N * 5; k += kSoftThreads) bx[k] = g[k];
    __syncthreads();
    N = soft_nms_problem<true>(bx, tail, keepf, N, sigma, Nt, threshold, method, r_score, r_pos, cnt);
    for (int k = t; k < N * 5; k += kSoftThreads) g[k] = bx[k];
  } else {
    // (the host wrapper refuses such a problem without a workspace)
    int *tail = reinterpret_cast<int *>(ws) + base;
    unsigned char *keepf = ws + (size_t)total_rows * sizeof(int) + base;
    N = soft_nms_problem<false>(g, tail, keepf, N, sigma, Nt, threshold, method, r_score, r_pos, cnt);
  }
  if (t == 0) count[p] = N;
}

SN_EXPORT size_t sn_soft_nms_max_boxes(void) { return kSoftLdsBoxes; }

SN_EXPORT size_t sn_soft_nms_workspace_bytes(size_t total_rows) { return (total_rows * 5 + 15) / 16 * 16; }

SN_EXPORT int sn_soft_nms_batch(float *d_boxes, const int32_t *d_off, int P, int max_n, size_t total_rows, float sigma, float Nt,
                                float threshold, int method, void *d_ws, int32_t *d_count, sn_stream_t stream) {
  SN_REQUIRE(d_boxes && d_off && d_count && P > 0 && max_n > 0, "sn_soft_nms_batch: bad arguments");
  SN_REQUIRE(max_n <= kSoftLdsBoxes || d_ws,
             "sn_soft_nms_batch: a problem of %d boxes (more than the %d one workgroup holds in LDS) needs "
             "sn_soft_nms_workspace_bytes(total_rows) of scratch", max_n, kSoftLdsBoxes);
  const int mn = (std::min(max_n, kSoftLdsBoxes) + 3) / 4 * 4;
  const size_t smem = (size_t)mn * 5 * sizeof(float) + (size_t)mn * sizeof(int) + (size_t)mn;
  SN_HIP(sn_once_per_device_max_lds(reinterpret_cast<const void *>(soft_nms_kernel), 110 * 1024));
  hipLaunchKernelGGL(soft_nms_kernel, dim3(P), dim3(kSoftThreads), smem, sn_stream(stream), d_boxes, d_off, sigma, Nt, threshold,
                     method, d_count, mn, static_cast<unsigned char *>(d_ws), (long long)total_rows);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
