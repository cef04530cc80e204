// proposal.hip -- MultiProposal / MultiProposalTarget for gfx950 (SNIPER-mxnet fork operators,
// call sites symbols/faster/resnet_mx_101_e2e.py:283-284 and :347-355; source not vendored, the
// semantics below are the documented choice of this repository -- see DESIGN.md "MultiProposalTarget").
//
// Per image: decode all A*F*F anchors with the RPN deltas (nonlinear_pred, lib/bbox/bbox_transform.py
// :93-130, in float32), clip to the chip, order by foreground score (descending, ties by anchor
// index), keep the top pre_nms, bitmask NMS (sn_nms_batch), first post_nms survivors (cyclically
// repeated if fewer), then label every RoI against the chip's GT boxes under the chip's valid range.
// Index work, compiled with -ffp-contract=off; everything stays on the device.
#include "common.h"

extern "C" int sn_nms_batch(const float *, const int32_t *, int, int, int, float, int, void *, int32_t *, int32_t *, sn_stream_t);
extern "C" size_t sn_nms_workspace_bytes(int, int);

struct PropLayout {
  int total, sort_n, pre, post;
  size_t boxes_all, keys, sorted, nvalid, keep, nkeep, nms, bytes;
};

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

static PropLayout prop_layout(int B, int A, int Fh, int Fw, int pre, int post) {
  PropLayout L;
  L.total = A * Fh * Fw;
  L.sort_n = next_pow2(L.total);
  L.pre = pre < L.total ? pre : L.total;
  L.post = post < L.pre ? post : L.pre;
  size_t off = 0;
  L.boxes_all = off; off += sn_align((size_t)B * L.total * 4 * sizeof(float));
  L.keys = off; off += sn_align((size_t)B * L.sort_n * sizeof(unsigned long long));
  L.sorted = off; off += sn_align((size_t)B * L.pre * 5 * sizeof(float));
  L.nvalid = off; off += sn_align((size_t)B * sizeof(int32_t));
  L.keep = off; off += sn_align((size_t)B * L.post * sizeof(int32_t));
  L.nkeep = off; off += sn_align((size_t)B * sizeof(int32_t));
  L.nms = off; off += sn_nms_workspace_bytes(B, L.pre);
  L.bytes = off;
  return L;
}

SN_EXPORT size_t sn_proposal_workspace_bytes(int B, int A, int Fh, int Fw, int pre_nms_top_n, int post_nms_top_n) {
  return prop_layout(B, A, Fh, Fw, pre_nms_top_n, post_nms_top_n).bytes;
}

__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-order preserving
}

// K1: decode + sort keys.  cls_prob (B,2,A*Fh,Fw), bbox_pred (B,4A,Fh,Fw), both fp32 reference layout.
__global__ __launch_bounds__(256) void proposal_decode_kernel(const float *__restrict__ cls_prob, const float *__restrict__ bbox_pred,
                                                              const float *__restrict__ im_info, const float *__restrict__ base,
                                                              int A, int Fh, int Fw, int stride, float min_size, int total, int sort_n,
                                                              float *__restrict__ boxes_all, unsigned long long *__restrict__ keys) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sort_n) return;
  unsigned long long *kb = keys + (size_t)b * sort_n;
  if (i >= total) { kb[i] = ~0ull; return; }
  const int cell = i / A, a = i - cell * A;
  const int h = cell / Fw, w = cell - h * Fw;
  const int FF = Fh * Fw;
  const float score0 = cls_prob[((size_t)b * 2 + 1) * A * FF + (size_t)a * FF + cell];
  const float *bp = bbox_pred + ((size_t)b * 4 * A + 4 * a) * FF + cell;
  const float dx = bp[0], dy = bp[FF], dw = bp[2 * FF], dh = bp[3 * FF];
  const float ax1 = base[4 * a] + (float)(w * stride), ay1 = base[4 * a + 1] + (float)(h * stride);
  const float ax2 = base[4 * a + 2] + (float)(w * stride), ay2 = base[4 * a + 3] + (float)(h * stride);
  const float aw = ax2 - ax1 + 1.0f, ah = ay2 - ay1 + 1.0f;
  const float cx = ax1 + 0.5f * (aw - 1.0f), cy = ay1 + 0.5f * (ah - 1.0f);
  const float pcx = dx * aw + cx, pcy = dy * ah + cy;
  // exp evaluated in double and rounded once: reproducible across libms (bit-exact RoIs vs the oracle)
  const float pw = (float)exp((double)dw) * aw, ph = (float)exp((double)dh) * ah;
  float x1 = pcx - 0.5f * (pw - 1.0f), y1 = pcy - 0.5f * (ph - 1.0f);
  float x2 = pcx + 0.5f * (pw - 1.0f), y2 = pcy + 0.5f * (ph - 1.0f);
  const float im_h = im_info[3 * b], im_w = im_info[3 * b + 1], im_s = im_info[3 * b + 2];
  x1 = fmaxf(fminf(x1, im_w - 1.0f), 0.f); y1 = fmaxf(fminf(y1, im_h - 1.0f), 0.f);
  x2 = fmaxf(fminf(x2, im_w - 1.0f), 0.f); y2 = fmaxf(fminf(y2, im_h - 1.0f), 0.f);
  float score = score0;
  const float ms = min_size * im_s;
  if ((x2 - x1 + 1.0f) < ms || (y2 - y1 + 1.0f) < ms) score = -1.0f;
  float *o = boxes_all + ((size_t)b * total + i) * 4;
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
  // descending score, ascending index on ties
  kb[i] = ((unsigned long long)(~orderable(score)) << 32) | (unsigned)i;
}

// K2: in-place ascending bitonic sort of sort_n (power of two) 64-bit keys, one workgroup per image.
__global__ __launch_bounds__(1024) void bitonic_sort_kernel(unsigned long long *__restrict__ keys, int n) {
  unsigned long long *k = keys + (size_t)blockIdx.x * n;
  for (int size = 2; size <= n; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (j - 1));  // index with bit j clear
        const int hi = lo + j;
        const bool up = (lo & size) == 0;
        const unsigned long long a = k[lo], c = k[hi];
        if ((a > c) == up) { k[lo] = c; k[hi] = a; }
      }
      __syncthreads();
    }
  }
}

// K2': only the `pre` best keys are ever used (pre_nms_top_n = 6000 of 21 504 anchors), and keys are unique (the anchor
// index sits in the low word), so "the first K of the full sort" == "the K smallest keys, sorted":
//   1. radix select, 8 bits per pass from the top: LDS histogram of the digit over the keys that match the decided prefix
//      -> the K-th smallest key exactly (passes stop as soon as the boundary bucket is taken whole);
//   2. compaction of the keys <= that key into LDS (any order), padded with ~0 to a power of two;
//   3. bitonic sort of those <= 16 384 keys in LDS; the first K go back to keys[0..K).
// One workgroup per image, every pass reads the image's keys (L2-resident, coalesced); 907 us -> see profiles/.
// Round 4: REG = the image's keys are read from global memory ONCE into registers (n <= 32 keys per thread: the training map's
// 21 504 anchors are 21) and every radix pass and the compaction run on them -- each pass used to re-read the 172 KB with one
// dependent L2 load per iteration (the passes, not the sort, were most of the 147 us).  Larger maps (test images at the finest
// scale: 231 k anchors) keep the streaming form.
constexpr int kSelThreads = 1024, kSelRegKeys = 32;
template <bool REG>
__global__ __launch_bounds__(kSelThreads) void topk_select_sort_kernel(unsigned long long *__restrict__ keys, int n, int K, int P2) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sel_buf[];   // [P2]
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix, s_mask;
  __shared__ int s_need, s_cnt, s_done;
  unsigned long long *k = keys + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  unsigned long long kreg[REG ? kSelRegKeys : 1];
  if constexpr (REG) {
#pragma unroll
    for (int u = 0; u < kSelRegKeys; ++u) {
      const int i = u * kSelThreads + tid;
      kreg[u] = i < n ? k[i] : 0ull;
    }
  }
  const int iters = REG ? (n + kSelThreads - 1) / kSelThreads : 0;
  if (tid == 0) { s_prefix = 0ull; s_mask = 0ull; s_need = K; s_cnt = 0; s_done = 0; }
  __syncthreads();
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    if (s_done) break;
    const unsigned long long prefix = s_prefix, mask = s_mask;
    auto count = [&](unsigned long long key, bool valid) {
      const bool in = valid && (key & mask) == prefix;
      const unsigned d = (unsigned)(key >> shift) & 255u;
      // scores cluster in a few exponent buckets: when the whole wave agrees, one atomic instead of 64 serialised ones
      const unsigned long long act = __ballot(in);
      if (act) {
        const int leader = __ffsll((long long)act) - 1;
        const unsigned d0 = (unsigned)__shfl((int)d, leader, 64);
        const bool same = __ballot(in && d == d0) == act;
        if (same) {
          if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[d0], (unsigned)__popcll(act));
        } else if (in) {
          atomicAdd(&hist[d], 1u);
        }
      }
    };
    if constexpr (REG) {
#pragma unroll
      for (int u = 0; u < kSelRegKeys; ++u)
        if (u < iters) count(kreg[u], u * kSelThreads + tid < n);       // (u < iters is wave-uniform: the ballots see whole waves)
    }
    // streaming form (maps beyond the register budget: 74 k - 231 k anchors at the finer test scales): eight independent loads
    // in flight per thread and pass -- one dependent L2 load per iteration left a pass latency-bound (424 us for 74 k keys)
    for (int base = 0; !REG && base < n; base += kSelThreads * 8) {
      unsigned long long kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * kSelThreads + tid;
        kk[u] = i < n ? k[i] : 0ull;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) count(kk[u], base + u * kSelThreads + tid < n);
    }
    __syncthreads();
    if (tid == 0) {
      int need = s_need, d = 0;
      unsigned cum = 0;
      for (; d < 256; ++d) {
        if (cum + hist[d] >= (unsigned)need) break;
        cum += hist[d];
      }
      s_need = need - (int)cum;
      s_prefix = prefix | ((unsigned long long)d << shift);
      s_mask = mask | (0xFFull << shift);
      if (hist[d] == (unsigned)(need - (int)cum)) {     // the whole boundary bucket is selected: largest key of it decides
        s_done = 1;
        s_prefix |= (shift ? ((1ull << shift) - 1ull) : 0ull);   // every key with this prefix is <= prefix | low ones
        s_mask = ~0ull;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  const unsigned long long kth = s_prefix;     // keys <= kth are exactly the K smallest
  for (int i = tid; i < P2; i += kSelThreads) sel_buf[i] = ~0ull;
  __syncthreads();
  if constexpr (REG) {
#pragma unroll
    for (int u = 0; u < kSelRegKeys; ++u)
      if (u * kSelThreads + tid < n && kreg[u] <= kth) {
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < P2) sel_buf[pos] = kreg[u];
      }
  }
  for (int base = 0; !REG && base < n; base += kSelThreads * 8) {
    unsigned long long kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * kSelThreads + tid;
      kk[u] = i < n ? k[i] : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (base + u * kSelThreads + tid < n && kk[u] <= kth) {
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < P2) sel_buf[pos] = kk[u];
      }
  }
  __syncthreads();
  for (int size = 2; size <= P2; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P2 >> 1); t += kSelThreads) {
        const int lo = 2 * t - (t & (j - 1));
        const int hi = lo + j;
        const bool up = (lo & size) == 0;
        const unsigned long long a = sel_buf[lo], c = sel_buf[hi];
        if ((a > c) == up) { sel_buf[lo] = c; sel_buf[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < K; i += kSelThreads) k[i] = sel_buf[i];
}

// K3: gather the top `pre` boxes in score order -> sorted (B, pre, 5) = x1,y1,x2,y2,score; nvalid[b]
__global__ __launch_bounds__(256) void proposal_gather_kernel(const float *__restrict__ boxes_all, const unsigned long long *__restrict__ keys,
                                                              int total, int sort_n, int pre, float *__restrict__ sorted,
                                                              int32_t *__restrict__ nvalid) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= pre) return;
  const unsigned long long key = keys[(size_t)b * sort_n + r];
  const unsigned idx = (unsigned)(key & 0xffffffffu);
  const unsigned ob = ~(unsigned)(key >> 32);
  const unsigned u = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;  // inverse of orderable()
  const float *s = boxes_all + ((size_t)b * total + idx) * 4;
  float *o = sorted + ((size_t)b * pre + r) * 5;
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; o[4] = __uint_as_float(u);
  if (r == 0) nvalid[b] = pre;
}

// K5a: MultiProposal output: rois (B*post, 5) = batch index + box, scores (B*post, 1)
__global__ __launch_bounds__(256) void proposal_output_kernel(const float *__restrict__ sorted, const int32_t *__restrict__ keep,
                                                              const int32_t *__restrict__ nkeep, int pre, int post,
                                                              float *__restrict__ rois, float *__restrict__ scores) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= post) return;
  const int nk = max(nkeep[b], 1);
  const int src = keep[(size_t)b * post + (k % nk)];
  const float *s = sorted + ((size_t)b * pre + src) * 5;
  float *o = rois + ((size_t)b * post + k) * 5;
  o[0] = (float)b; o[1] = s[0]; o[2] = s[1]; o[3] = s[2]; o[4] = s[3];
  if (scores) scores[(size_t)b * post + k] = s[4];
}

// K5b: RoI labelling.  gt_boxes (B,G,5) = x1,y1,x2,y2,class (-1 padded), valid_ranges (B,2).
//   size(gt) = sqrt((x2-x1+1)*(y2-y1+1)); gt is "valid" for the chip iff lo <= size <= hi
//   fg    : max IoU with a valid GT >= fg_thresh      -> label = class of the arg-max GT (first max)
//   ignore: else max IoU with an invalid GT >= fg_thresh -> label = -1
//   bg    : otherwise                                   -> label = 0
//   bbox_target = nonlinear_transform(roi, gt) / stds (means are zero), bbox_weight = 1, fg only.
__global__ __launch_bounds__(256) void proposal_target_kernel(const float *__restrict__ rois, const float *__restrict__ gt_boxes,
                                                              const float *__restrict__ valid_ranges, int G, int post,
                                                              float fg_thresh, float4 stds, float *__restrict__ label,
                                                              float *__restrict__ bbox_target, float *__restrict__ bbox_weight,
                                                              float *__restrict__ match) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) float sgt[];  // G x 6: box, class, valid flag
  const float lo = valid_ranges[2 * b], hi = valid_ranges[2 * b + 1];
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    const float *p = gt_boxes + ((size_t)b * G + g) * 5;
    const float size = sqrtf((p[2] - p[0] + 1.0f) * (p[3] - p[1] + 1.0f));
    sgt[6 * g + 0] = p[0]; sgt[6 * g + 1] = p[1]; sgt[6 * g + 2] = p[2]; sgt[6 * g + 3] = p[3]; sgt[6 * g + 4] = p[4];
    sgt[6 * g + 5] = (p[4] < 0.f) ? -1.f : ((size >= lo && size <= hi) ? 1.f : 0.f);
  }
  __syncthreads();
  if (k >= post) return;
  const float *r = rois + ((size_t)b * post + k) * 5;
  const float x1 = r[1], y1 = r[2], x2 = r[3], y2 = r[4];
  const float area = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);
  float best_v = -1.f, best_i = -1.f;
  int arg = -1;
  for (int g = 0; g < G; ++g) {
    const float flag = sgt[6 * g + 5];
    if (flag < 0.f) continue;
    const float gx1 = sgt[6 * g], gy1 = sgt[6 * g + 1], gx2 = sgt[6 * g + 2], gy2 = sgt[6 * g + 3];
    const float iw = fminf(x2, gx2) - fmaxf(x1, gx1) + 1.0f;
    float ov = 0.f;
    if (iw > 0.f) {
      const float ih = fminf(y2, gy2) - fmaxf(y1, gy1) + 1.0f;
      if (ih > 0.f) ov = iw * ih / (area + (gx2 - gx1 + 1.0f) * (gy2 - gy1 + 1.0f) - iw * ih);
    }
    if (flag > 0.f) { if (ov > best_v) { best_v = ov; arg = g; } }
    else if (ov > best_i) best_i = ov;
  }
  float lab = 0.f, t[4] = {0.f, 0.f, 0.f, 0.f}, wv = 0.f;
  if (arg >= 0 && best_v >= fg_thresh) {
    lab = sgt[6 * arg + 4];
    wv = 1.f;
    const float gx1 = sgt[6 * arg], gy1 = sgt[6 * arg + 1], gx2 = sgt[6 * arg + 2], gy2 = sgt[6 * arg + 3];
    const float ew = x2 - x1 + 1.0f, eh = y2 - y1 + 1.0f;
    const float ecx = x1 + 0.5f * (ew - 1.0f), ecy = y1 + 0.5f * (eh - 1.0f);
    const float gw = gx2 - gx1 + 1.0f, gh = gy2 - gy1 + 1.0f;
    const float gcx = gx1 + 0.5f * (gw - 1.0f), gcy = gy1 + 0.5f * (gh - 1.0f);
    t[0] = (gcx - ecx) / (ew + 1e-7f) / stds.x;
    t[1] = (gcy - ecy) / (eh + 1e-7f) / stds.y;
    t[2] = logf(gw / (ew + 1e-7f)) / stds.z;
    t[3] = logf(gh / (eh + 1e-7f)) / stds.w;
  } else if (best_i >= fg_thresh) {
    lab = -1.f;
  }
  const size_t o = (size_t)b * post + k;
  label[o] = lab;
  if (match) match[o] = wv > 0.f ? (float)arg : -1.f;   // gt_boxes row of a foreground RoI (mask branch)
#pragma unroll
  for (int c = 0; c < 4; ++c) { bbox_target[4 * o + c] = t[c]; bbox_weight[4 * o + c] = wv; }
}

static int proposal_common(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *base_anchors, int B,
                           int A, int Fh, int Fw, int stride, int pre, int post, float nms_thresh, float min_size, char *ws,
                           const PropLayout &L, hipStream_t s) {
  float *boxes_all = (float *)(ws + L.boxes_all);
  unsigned long long *keys = (unsigned long long *)(ws + L.keys);
  float *sorted = (float *)(ws + L.sorted);
  int32_t *nvalid = (int32_t *)(ws + L.nvalid);
  int32_t *keep = (int32_t *)(ws + L.keep);
  int32_t *nkeep = (int32_t *)(ws + L.nkeep);
  hipLaunchKernelGGL(proposal_decode_kernel, dim3(sn_div_up(L.sort_n, 256), B), dim3(256), 0, s, cls_prob, bbox_pred, im_info,
                     base_anchors, A, Fh, Fw, stride, min_size, L.total, L.sort_n, boxes_all, keys);
  SN_CHECK_LAUNCH();
  const int P2 = next_pow2(L.pre);
  if (P2 <= 16384 && L.pre < L.sort_n && !sn_debug_get(SN_OPT_PROPOSAL_FULL_SORT)) {
    // > 64 KB of dynamic LDS needs the opt-in once per (kernel, device)
    if (L.sort_n <= kSelRegKeys * kSelThreads) {
      SN_HIP(sn_once_per_device_max_lds(reinterpret_cast<const void *>(topk_select_sort_kernel<true>), 16384 * 8));
      hipLaunchKernelGGL(topk_select_sort_kernel<true>, dim3(B), dim3(kSelThreads), (size_t)P2 * 8, s, keys, L.sort_n, L.pre, P2);
    } else {
      SN_HIP(sn_once_per_device_max_lds(reinterpret_cast<const void *>(topk_select_sort_kernel<false>), 16384 * 8));
      hipLaunchKernelGGL(topk_select_sort_kernel<false>, dim3(B), dim3(kSelThreads), (size_t)P2 * 8, s, keys, L.sort_n, L.pre, P2);
    }
  } else {
    hipLaunchKernelGGL(bitonic_sort_kernel, dim3(B), dim3(1024), 0, s, keys, L.sort_n);
  }
  SN_CHECK_LAUNCH();
  hipLaunchKernelGGL(proposal_gather_kernel, dim3(sn_div_up(L.pre, 256), B), dim3(256), 0, s, boxes_all, keys, L.total, L.sort_n,
                     L.pre, sorted, nvalid);
  SN_CHECK_LAUNCH();
  return sn_nms_batch(sorted, nvalid, B, L.pre, 5, nms_thresh, L.post, ws + L.nms, keep, nkeep, (sn_stream_t)s);
}

// base_anchors: (A,4) fp32 device.  rois (B*post,5), scores (B*post) optional.
SN_EXPORT int sn_multi_proposal(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *base_anchors,
                                int B, int A, int Fh, int Fw, int feat_stride, int pre_nms_top_n, int post_nms_top_n, float nms_thresh,
                                float min_size, void *ws, float *rois, float *scores, sn_stream_t stream) {
  SN_REQUIRE(cls_prob && bbox_pred && im_info && base_anchors && ws && rois && B > 0 && A > 0 && Fh > 0 && Fw > 0,
             "sn_multi_proposal: bad arguments");
  const PropLayout L = prop_layout(B, A, Fh, Fw, pre_nms_top_n, post_nms_top_n);
  hipStream_t s = sn_stream(stream);
  if (int rc = proposal_common(cls_prob, bbox_pred, im_info, base_anchors, B, A, Fh, Fw, feat_stride, L.pre, L.post, nms_thresh,
                               min_size, (char *)ws, L, s))
    return rc;
  hipLaunchKernelGGL(proposal_output_kernel, dim3(sn_div_up(L.post, 256), B), dim3(256), 0, s,
                     (const float *)((char *)ws + L.sorted), (const int32_t *)((char *)ws + L.keep),
                     (const int32_t *)((char *)ws + L.nkeep), L.pre, L.post, rois, scores);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_multi_proposal_target(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *gt_boxes,
                                       const float *valid_ranges, const float *base_anchors, int B, int A, int Fh, int Fw,
                                       int feat_stride, int G, int pre_nms_top_n, int post_nms_top_n, float nms_thresh,
                                       float min_size, float fg_thresh, const float *bbox_stds4, void *ws, float *rois,
                                       float *label, float *bbox_target, float *bbox_weight, sn_stream_t stream) {
  SN_REQUIRE(gt_boxes && valid_ranges && label && bbox_target && bbox_weight && bbox_stds4 && G > 0 && G <= 1024,
             "sn_multi_proposal_target: bad arguments");
  if (int rc = sn_multi_proposal(cls_prob, bbox_pred, im_info, base_anchors, B, A, Fh, Fw, feat_stride, pre_nms_top_n, post_nms_top_n,
                                 nms_thresh, min_size, ws, rois, nullptr, stream))
    return rc;
  const PropLayout L = prop_layout(B, A, Fh, Fw, pre_nms_top_n, post_nms_top_n);
  const float4 stds = make_float4(bbox_stds4[0], bbox_stds4[1], bbox_stds4[2], bbox_stds4[3]);
  hipLaunchKernelGGL(proposal_target_kernel, dim3(sn_div_up(L.post, 256), B), dim3(256), (size_t)G * 6 * sizeof(float),
                     sn_stream(stream), rois, gt_boxes, valid_ranges, G, L.post, fg_thresh, stds, label, bbox_target, bbox_weight,
                     (float *)nullptr);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// MultiProposalTargetMask (symbols/faster/resnet_mx_101_e2e_mask.py:317-318; fork operator, spec ours -- DESIGN.md):
// MultiProposalTarget plus, per chip, the first `num_mask_rois` FOREGROUND RoIs in RoI order (mask_rois (B*nm, 5)) and the
// gt_boxes row each one matched (mask_ids (B*nm)); chips with fewer foreground RoIs are padded with [b, 0, 0, 0, 0] / -1,
// which MaskRcnnTarget turns into all-ignore targets.  match_ws: B*post floats of scratch.
__global__ __launch_bounds__(256) void mask_rois_select_kernel(const float *__restrict__ rois, const float *__restrict__ label,
                                                               const float *__restrict__ match, int post, int nm,
                                                               float *__restrict__ mask_rois, float *__restrict__ mask_ids) {
  __shared__ int wave_cnt[4];
  __shared__ int taken_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  if (tid == 0) taken_s = 0;
  __syncthreads();
  for (int base = 0; base < post; base += 256) {
    const int k = base + tid;
    const bool fg = k < post && label[(size_t)b * post + k] > 0.f;
    const unsigned long long m = __ballot(fg);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = taken_s, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      off += w < wave ? wave_cnt[w] : 0;
      total += wave_cnt[w];
    }
    const int slot = off + __popcll(m & lt);
    if (fg && slot < nm) {
      const float *r = rois + ((size_t)b * post + k) * 5;
      float *o = mask_rois + ((size_t)b * nm + slot) * 5;
      o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[4];
      mask_ids[(size_t)b * nm + slot] = match[(size_t)b * post + k];
    }
    __syncthreads();
    if (tid == 0) taken_s += total;
    __syncthreads();
    if (taken_s >= nm) break;
  }
  for (int slot = min(taken_s, nm) + tid; slot < nm; slot += 256) {
    float *o = mask_rois + ((size_t)b * nm + slot) * 5;
    o[0] = (float)b; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f;
    mask_ids[(size_t)b * nm + slot] = -1.f;
  }
}

SN_EXPORT int sn_multi_proposal_target_mask(const float *cls_prob, const float *bbox_pred, const float *im_info, const float *gt_boxes,
                                            const float *valid_ranges, const float *base_anchors, int B, int A, int Fh, int Fw,
                                            int feat_stride, int G, int pre_nms_top_n, int post_nms_top_n, float nms_thresh,
                                            float min_size, float fg_thresh, const float *bbox_stds4, void *ws, float *match_ws,
                                            int num_mask_rois, float *rois, float *label, float *bbox_target, float *bbox_weight,
                                            float *mask_rois, float *mask_ids, sn_stream_t stream) {
  SN_REQUIRE(gt_boxes && valid_ranges && label && bbox_target && bbox_weight && bbox_stds4 && G > 0 && G <= 1024 && match_ws &&
                 mask_rois && mask_ids && num_mask_rois > 0, "sn_multi_proposal_target_mask: bad arguments");
  if (int rc = sn_multi_proposal(cls_prob, bbox_pred, im_info, base_anchors, B, A, Fh, Fw, feat_stride, pre_nms_top_n, post_nms_top_n,
                                 nms_thresh, min_size, ws, rois, nullptr, stream))
    return rc;
  const PropLayout L = prop_layout(B, A, Fh, Fw, pre_nms_top_n, post_nms_top_n);
  const float4 stds = make_float4(bbox_stds4[0], bbox_stds4[1], bbox_stds4[2], bbox_stds4[3]);
  hipLaunchKernelGGL(proposal_target_kernel, dim3(sn_div_up(L.post, 256), B), dim3(256), (size_t)G * 6 * sizeof(float),
                     sn_stream(stream), rois, gt_boxes, valid_ranges, G, L.post, fg_thresh, stds, label, bbox_target, bbox_weight,
                     match_ws);
  SN_CHECK_LAUNCH();
  hipLaunchKernelGGL(mask_rois_select_kernel, dim3(B), dim3(256), 0, sn_stream(stream), rois, label, match_ws, L.post, num_mask_rois,
                     mask_rois, mask_ids);
  SN_CHECK_LAUNCH();
  return SN_OK;
}
