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
  SN_REQUIRE(d_boxes && d_ws && d_keep, "sn_nms_batch: null pointer");
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
