// conv_wgrad_ps.hip -- weight gradient dW[co][tap][ci] += sum_pixels dY[pix][co] * X[src(pix, tap)][ci], wave-specialised and batched.
//
// Replaces the weight-gradient half of the Convolution / FullyConnected / DeformableConvolution operators of the un-vendored fork
// (symbols/faster/resnet_mx_101_e2e.py:43-66,121-155,256,288-303) for every layer whose operands are 16-byte addressable.
//
// Why another kernel (profiles/r03_wgrad_xcd_ab.txt, profiles/r02_dma_rate_probe.txt): the round-2 kernels spend ~2000 cycles per
// 64-pixel K-step of a 128 x 128 tile although its 32 MFMAs per wave take 512 -- with the fabric traffic already at the algorithmic
// minimum (XCD-aware block order: FETCH 126 -> 53 MB for a stage-3 1x1 layer, duration unchanged).  A wave's instruction stream is in
// order: staging a K-step (32 LDS-DMA pieces, ~70 cycles of issue each, or loads + ds_write) and multiplying it cannot overlap inside
// one wave, and the barrier puts all waves of a workgroup into the same phase.  Here the two jobs belong to different waves:
//
//   * waves 0-3 (one per SIMD) are CONSUMERS: barrier, 32 transposing fragment reads, 32 MFMAs, nothing else in the loop;
//   * waves 4-7 (the second wave of each SIMD) are PRODUCERS: each stages 16 pixels of every K-step (4 dY + 4 X pieces of 1 KB,
//     buffer_load ... lds) into an S-deep ring, S - 1 stages ahead, and publishes a stage with a counted s_waitcnt vmcnt + the
//     step's one barrier.  (Two producers issued 16 pieces each in ~700 cycles per step, profiles/r03_wgrad_ps_trace.txt: the
//     producers, not the consumers, set the pace; four issue 8 each.)
// The consumers keep the fragments of a K-step in two register sets (pixels 0-31 / 32-63): the step's barrier sits between the two
// MFMA blocks and each set is re-read for the next stage right behind the block that used it, so the LDS latency of a stage's first
// fragments (~300 cycles per step when every step began with barrier -> reads -> wait) runs under the other set's MFMAs.
//
// Both operands are K(pixel)-major in HBM, so tiles are staged as they lie -- [pixel][channel], 256-byte rows, the XOR swizzle applied
// to the per-lane SOURCE address -- and transposed by the LDS read ds_read_b64_tr_b16 (as in the round-1/2 kernels).  A KxK convolution
// puts the tap into the grid: tap (kh, kw) of output pixel (oy, ox) reads X[oy s - p + kh d][ox s - p + kw d]; a K-step half is one
// 32-pixel run of one output row, so the source row is a scalar and row / column padding is the buffer descriptor's range check.
//
// Batching: the launch takes a TABLE of problems (layers).  A single stage-3 layer has 16-36 tiles; filling 256 CUs with it needs a
// 7-16-way K split whose fp32 partial slabs (workgroups x 64 KB, written and read back) cost as much HBM time as the operands.  Several
// layers per launch fill the chip with whole-K jobs instead: no slabs, no reduce launch, one fill / drain per 320 K-steps instead of
// per 20.  Jobs are numbered problem-major and every XCD gets a contiguous range of them, so the tiles that share a problem's dY / X
// panels fetch them through ONE L2.  Problems whose jobs would be much longer than the rest are still split (slab + reduce).
#include "conv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef short short4v __attribute__((vector_size(8)));

namespace {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t *dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}

// one 8-deep MFMA fragment = two transposing reads, 16 tile rows apart
__device__ __forceinline__ half8 tr_frag(const half_t *lds_tile, int off) {
  typedef __attribute__((address_space(3))) short4v *lds_v4;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_tile + off));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_tile + off + 16 * 128));
  union { short4v s[2]; half8 h; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.h;
}

// a lane whose channel chunk lies beyond Cout / Cin carries this voffset: with every tensor < 1 GB (checked on the host) no step
// offset can bring it back under a descriptor's num_records, so it reads zeros for the whole job without a select per piece
constexpr unsigned kPoison = 0x80000000u;

constexpr int kTile = 64 * 128;       // half_t per 128-channel operand sub-tile and stage: 64 pixels x 128 channels

// TRACE: wave 0 (consumer) and wave 4 (producer) of every workgroup sum the shader-clock cycles of their phases into tab.trace
// [job][8] = {life, K loop, consumer barrier wait, epilogue, producer vmcnt wait, producer barrier wait, producer issue, K-steps}
// NA = 128-channel sub-tiles of dY per job: the job's output tile is (128 NA) co x 128 ci.  NA = 1: 4-stage ring of 32 KB (narrow
// layers, Cout <= 128).  NA = 2: 3-stage ring of 48 KB: a K-step of two tiles' worth of MFMAs is staged with 48 pieces instead of 64.
// MI = 16-row co blocks per consumer wave (wave tile 16 MI x 64): 4.  NA = 2 runs EIGHT consumer waves of 64 x 64 (two per SIMD).
// Measured (profiles/r03_wgrad_ps_trace.txt): a K-step costs 893-933 cycles for NA = 1 and 1375-1450 for NA = 2 with twice the
// work -- both about (LDS-DMA bytes + fragment bytes) / 128 B per clock: the LDS, not the matrix pipe (512 / 1024 cycles per SIMD)
// and not HBM, is what a step waits for.  Tried and removed: NA = 2 with FOUR consumers of 128 x 64 (MI = 8; 96 KB instead of
// 128 KB of fragment reads per step, quarter-step register schedule to stay under 256 VGPRs): 1444-1817 cycles per step -- one
// wave per SIMD exposes every MFMA issue stall that a second wave on the SIMD hides -- 23.4 ms per training step against 22.1.
template <int S, int NA, int MI, bool TRACE>
__global__ __launch_bounds__(64 * (16 * NA / MI + 4)) void wgrad_ps_kernel(const WgradBatch tab) {
  constexpr int NI = 4, NC = 16 * NA / MI, L = 4 * (NA + 1);   // NC consumer waves (8 NA / MI along co x 2 along ci); L: LDS-DMA pieces a producer issues per stage
  constexpr int kStage = (NA + 1) * kTile;                       // NA dY sub-tiles, then the X tile
  static_assert((S - 1) * L < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) half_t lds[S * kStage];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long t_entry = TRACE ? __builtin_amdgcn_s_memtime() : 0;
  // ---- which job ----
  const int b = blockIdx.x;
  const int item = tab.xcd_start[b & 7] + (b >> 3);
  if (item >= tab.xcd_start[(b & 7) + 1]) return;
  int pi = 0;
  while (pi + 1 < tab.n && item >= tab.p[pi + 1].item0) ++pi;
  const WgradProblem &P = tab.p[pi];
  const int gx = P.gx, taps = P.taps, tiles = gx * P.gy;
  int r = item - P.item0;
  const int bz = r / tiles;
  r -= bz * tiles;
  const int by = r / gx, bx = r - by * gx;
  const int split = bz / taps, tap = bz - split * taps;
  const int kh = tap / P.KW, kw = tap - kh * P.KW;
  const int co0 = bx * (128 * NA), ci0 = by * 128;
  const int Cout = P.Cout, Cin = P.Cin;
  const int u_begin = split * P.units_per_split, u_end = min(P.nunits, u_begin + P.units_per_split);
  const int nk = u_end > u_begin ? (u_end - u_begin + 1) >> 1 : 0;

  if (wave >= NC) {
    // ================================ producer ================================
    const int h = (wave - NC) >> 1, sub = (wave - NC) & 1;   // 32-pixel unit of every K-step, 16-pixel half of that unit
    const int r4 = lane >> 4, slot = lane & 15;
    const unsigned dy_ps_b = (unsigned)P.dy_ps * 2u, x_ps_b = (unsigned)P.x_ps * 2u;
    const int stride = P.stride, Wo = P.Wo, Ho = P.Ho, H = P.H, W = P.W, cpr = P.cpr;
    const int row_shift = kh * P.dil - P.pad;      // source row of output row oy: oy * stride + row_shift
    // piece j = 0..3: unit rows 16 sub + 4j + r4, tile rows 32h + 16 sub + 4j + r4; (tile row & 7) = ((j & 1) << 2) | r4 -> the source
    // chunk that belongs in this lane's 16-byte slot differs between even and odd pieces
    unsigned a_base[NA][2], b_base[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int gc = ((((slot >> 1) ^ ((o << 2) | r4)) << 1) | (slot & 1));
      const int ur = 16 * sub + r4;
#pragma unroll
      for (int ta = 0; ta < NA; ++ta) {
        const int co = co0 + ta * 128 + gc * 8;
        a_base[ta][o] = co < Cout ? (unsigned)ur * dy_ps_b + (unsigned)co * 2u : kPoison;
      }
      b_base[o] = ci0 + gc * 8 < Cin ? (unsigned)((ur * stride - P.pad + kw * P.dil) * (int)x_ps_b) + (unsigned)(ci0 + gc * 8) * 2u : kPoison;
    }
    const unsigned a_step = 4u * dy_ps_b, b_step = 4u * (unsigned)stride * x_ps_b;   // piece j -> j + 1
    const char *dyb = reinterpret_cast<const char *>(P.dy), *xbp = reinterpret_cast<const char *>(P.x);
    // unit iterator of this producer: units u_begin + h, + 2, + 2, ...  (workgroup-uniform scalars)
    int u = u_begin + h;
    int it_r = u / cpr, it_xc = u - it_r * cpr;
    int it_img = it_r / Ho, it_oy = it_r - it_img * Ho;
    half_t *const dst0 = lds + (h * 8 + sub * 4) * 512;
    auto issue = [&](int buf) {
      const bool have = u < u_end;
      const int ox0 = it_xc * 32;
      const int sy = it_oy * stride + row_shift;
      const bool rok = have && (unsigned)sy < (unsigned)H;
      const size_t dy_off = have ? ((size_t)it_r * Wo + ox0) * dy_ps_b : 0;
      const size_t x_off = rok ? ((size_t)(it_img * H + sy) * W) * x_ps_b : 0;
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(dyb) + dy_off, 0, have ? (int)((unsigned)(Wo - ox0) * dy_ps_b) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xbp) + x_off, 0, rok ? (int)((unsigned)W * x_ps_b) : 0, 0x00020000);
      const unsigned b_ox = (unsigned)(ox0 * stride) * x_ps_b;
      half_t *const sa = dst0 + buf * kStage, *const sb = sa + NA * kTile;
#pragma unroll
      for (int ta = 0; ta < NA; ++ta)
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16(rdy, sa + ta * kTile + j * 512, a_base[ta][j & 1] + (unsigned)j * a_step);
#pragma unroll
      for (int j = 0; j < 4; ++j) dma16(rxx, sb + j * 512, b_base[j & 1] + b_ox + (unsigned)j * b_step);
      // advance by two units
      u += 2;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (++it_xc == cpr) {
          it_xc = 0;
          ++it_r;
          if (++it_oy == Ho) { it_oy = 0; ++it_img; }
        }
    };
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
      if (s < nk) issue(s);
    int nxt = S - 1;
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, t0 = 0, t1 = 0, t2 = 0;
    for (int t = 0; t < nk; ++t) {
      if (TRACE) t0 = __builtin_amdgcn_s_memtime();
      // stage t must have landed; the (up to S - 2) younger stages stay in flight across the barrier
      const int young = min(S - 2, nk - 1 - t);
      if (S > 3 && young >= 2) wait_vmcnt<(S > 3 ? 2 : 0) * L>();
      else if (S > 2 && young == 1) wait_vmcnt<(S > 2 ? 1 : 0) * L>();
      else wait_vmcnt<0>();
      if (TRACE) t1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      if (TRACE) t2 = __builtin_amdgcn_s_memtime();
      // the consumers have passed barrier t, i.e. finished reading stage t - 1: its buffer is free
      if (t + S - 1 < nk) issue(nxt);
      nxt = nxt + 1 == S ? 0 : nxt + 1;
      if (TRACE) { c_wait += t1 - t0; c_bar += t2 - t1; c_issue += __builtin_amdgcn_s_memtime() - t2; }
    }
    if (TRACE && wave == NC && lane == 0 && tab.trace) {
      unsigned long long *o = tab.trace + (size_t)item * 8;
      o[4] = c_wait; o[5] = c_bar; o[6] = c_issue;
    }
    return;
  }

  // ================================ consumer ================================
  const int wm = wave >> 1, wn = wave & 1;          // wm: block of 16 MI co rows of the tile
  const int fr = lane & 15, fq = lane >> 4;
  // fragment reads: lane (fr, fq) points at row fq*4 + fr/4, 8-byte piece fr%4 of the fragment's 32-byte segment
  const int row0 = fq * 4 + (fr >> 2), q7 = row0 & 7;
  int a_off[MI], b_off[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int rb = wm * MI + i;                      // 16-row co block of the tile: sub-tile rb >> 3, 32-byte segment rb & 7 of its rows
    a_off[i] = (rb >> 3) * kTile + row0 * 128 + ((((rb & 7) ^ q7) << 4) | ((fr & 3) << 2));
  }
#pragma unroll
  for (int jn = 0; jn < NI; ++jn) b_off[jn] = row0 * 128 + ((((wn * 4 + jn) ^ q7) << 4) | ((fr & 3) << 2));
  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) acc[i][jn] = floatx4{0.f, 0.f, 0.f, 0.f};
  unsigned long long c_bar = 0, tb = 0;
  const unsigned long long t_loop = TRACE ? __builtin_amdgcn_s_memtime() : 0;
  half8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
  auto read = [&](int buf, int ks, half8 (&fa)[MI], half8 (&fb)[NI]) {
    const half_t *const sa = lds + buf * kStage + ks * 32 * 128, *const sb = sa + NA * kTile;
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = tr_frag(sa, a_off[i]);
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) fb[jn] = tr_frag(sb, b_off[jn]);
  };
  auto mma = [&](const half8 (&fa)[MI], const half8 (&fb)[NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
  };
  // barrier t = "stage t has landed" (producers) and "stage t - 1 has been read" (consumers: every fragment read of a stage is
  // retired -- lgkmcnt(0) -- before the wave arrives at the next barrier, because the producers overwrite that buffer behind it)
  auto step_barrier = [&]() {
    if (TRACE) tb = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (TRACE) c_bar += __builtin_amdgcn_s_memtime() - tb;
  };
  // Schedule of a K-step t (two register sets: pixels 0-31 in fa0 / fb0, pixels 32-63 in fa1 / fb1):
  //   region A: 16 MFMAs on set 0 (stage t) interleaved 1:1 with the 16 fragment reads of set 1 <- stage t
  //   barrier t + 1
  //   region B: 16 MFMAs on set 1 (stage t) interleaved 1:1 with the 16 fragment reads of set 0 <- stage t + 1
  // A read is issued in the shadow of an MFMA (the matrix pipe takes 16 cycles per instruction, the wave's issue slot is free in
  // between) and has a whole region to land; sched_group_barrier pins the interleave, sched_barrier keeps hipcc from sinking all
  // reads behind the last MFMA that uses their destination registers (what it does on its own: 32 reads back to back, then a
  // wait at the next step's first MFMA).
  auto interleave = [&]() {
    constexpr int NR = 2 * (MI + NI), NM = MI * NI;     // transposing reads / MFMAs of a region
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
    }
    if (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  if (nk > 0) {
    step_barrier();
    read(0, 0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
  }
  int cur = 0, nxt = 1;
  for (int t = 0; t + 1 < nk; ++t) {
    mma(fa0, fb0);
    read(cur, 1, fa1, fb1);
    interleave();
    step_barrier();               // barrier t + 1: every read of stage t has been issued above and is retired inside
    __builtin_amdgcn_sched_barrier(0);
    mma(fa1, fb1);
    read(nxt, 0, fa0, fb0);
    interleave();
    cur = nxt;
    nxt = nxt + 1 == S ? 0 : nxt + 1;
  }
  if (nk > 0) {
    mma(fa0, fb0);
    read(cur, 1, fa1, fb1);
    interleave();
    mma(fa1, fb1);
  }
  const unsigned long long t_epi = TRACE ? __builtin_amdgcn_s_memtime() : 0;
  float *dst = P.slab ? P.slab + (size_t)split * P.slab_stride : P.dw;
  const bool to_slab = P.slab != nullptr;
  if ((Cin & 3) == 0) {
    // every lane's 16 accesses are 16 bytes; an unsplit job first brings all 16 of its dw vectors in (loads back to back, one wait
    // chain), then adds and stores -- a load / wait / add / store per vector is 16 dependent memory round trips per lane
#pragma unroll
    for (int i0 = 0; i0 < MI; i0 += 4) {          // four co blocks (16 vectors per lane) at a time: 64 registers of dw in flight
      float *q[4][NI];
      bool ok[4][NI];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
          const int co = co0 + (wm * MI + i0 + i) * 16 + fr, ci = ci0 + wn * 64 + jn * 16 + fq * 4;
          ok[i][jn] = co < Cout && ci < Cin;
          q[i][jn] = dst + ((size_t)co * taps + tap) * Cin + ci;
        }
      if (!to_slab) {
        float4 o[4][NI];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < NI; ++jn) o[i][jn] = ok[i][jn] ? *reinterpret_cast<const float4 *>(q[i][jn]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < NI; ++jn) {
            floatx4 &a = acc[i0 + i][jn];
            a[0] += o[i][jn].x; a[1] += o[i][jn].y; a[2] += o[i][jn].z; a[3] += o[i][jn].w;
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
          if (ok[i][jn]) *reinterpret_cast<float4 *>(q[i][jn]) = make_float4(acc[i0 + i][jn][0], acc[i0 + i][jn][1], acc[i0 + i][jn][2], acc[i0 + i][jn][3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int co = co0 + (wm * MI + i) * 16 + fr;
      if (co >= Cout) continue;
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) {
        const int ci = ci0 + wn * 64 + jn * 16 + fq * 4;
        float *qq = dst + ((size_t)co * taps + tap) * Cin + ci;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (ci + rr < Cin) qq[rr] = to_slab ? acc[i][jn][rr] : qq[rr] + acc[i][jn][rr];
      }
    }
  }
  if (TRACE && wave == 0 && lane == 0 && tab.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned long long *o = tab.trace + (size_t)item * 8;
    o[0] = t_end - t_entry; o[1] = t_epi - t_loop; o[2] = c_bar; o[3] = t_end - t_epi; o[7] = (unsigned long long)nk;
  }
}

// The slab reductions of every split problem of a table in ONE launch (round 4; they were one launch of ~7 us per split layer,
// 32 per step): workgroup b finds its problem from the per-problem block counts (<= 24 problems: a scalar walk) and reduces its
// share: dw[e] += sum over the splits of slab[s][e], eight slabs' loads in flight per thread, fixed per-element summation order.
constexpr int kReduceBlocksMax = 1024;
__device__ __forceinline__ int reduce_blocks_of(const WgradProblem &q) {
  if (!q.slab) return 0;
  const long b = (long)((q.slab_stride / 4 + 255) / 256);
  return (int)(b < 1 ? 1 : (b > kReduceBlocksMax ? kReduceBlocksMax : b));
}
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgradBatch tab) {
  int b = blockIdx.x, pi = 0, nb = 0;
  for (; pi < tab.n; ++pi) {
    nb = reduce_blocks_of(tab.p[pi]);
    if (b < nb) break;
    b -= nb;
  }
  if (pi >= tab.n) return;
  const WgradProblem &q = tab.p[pi];
  const float *__restrict__ slab = q.slab;
  float *__restrict__ dw = q.dw;
  const size_t n = q.slab_stride, n4 = n >> 2;
  const int splits = q.splits;
  for (size_t i = (size_t)b * 256 + threadIdx.x; i < n4; i += (size_t)nb * 256) {
    float4 a = reinterpret_cast<const float4 *>(dw)[i];
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const float4 *>(slab + (size_t)(s + k) * n)[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
    }
    for (; s < splits; ++s) {
      const float4 v = reinterpret_cast<const float4 *>(slab + (size_t)s * n)[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4 *>(dw)[i] = a;
  }
  if (b == 0 && threadIdx.x < (n & 3)) {
    const size_t e = (n4 << 2) + threadIdx.x;
    float a = dw[e];
    for (int s = 0; s < splits; ++s) a += slab[(size_t)s * n + e];
    dw[e] = a;
  }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool wgrad_ps_ok(const WgradParams &p) {
  const size_t xb = (size_t)p.N * p.H * p.W * p.x_ps * 2, dyb = (size_t)p.N * p.Ho * p.Wo * p.dy_ps * 2;
  return p.dy_ps % 8 == 0 && p.x_ps % 8 == 0 && p.dy_ps >= sn_div_up(p.Cout, 8) * 8 && p.x_ps >= sn_div_up(p.Cin, 8) * 8 &&
         ((uintptr_t)p.dy % 16) == 0 && ((uintptr_t)p.x % 16) == 0 && xb < (1ull << 30) && dyb < (1ull << 30) &&
         (size_t)p.Cout * p.KH * p.KW * p.Cin < (1ull << 31);
}

static std::atomic<unsigned long long *> g_wgrad_trace{nullptr};   // phase timeline buffer [jobs][8] (tools/wgrad_batch_bench.py --trace), normally null
void wgrad_ps_set_trace(unsigned long long *buf) { g_wgrad_trace.store(buf, std::memory_order_relaxed); }
static std::atomic<int> g_wgrad_job_steps{0};   // tuning override of the longest job (K-steps); 0 = built-in
void wgrad_ps_set_job_steps(int steps) { g_wgrad_job_steps.store(steps, std::memory_order_relaxed); }

// Fill tab (geometry, splits, item ranges) from n problems; slab pointers are offsets into `ws` (nullptr = size query).
// Returns the scratch bytes the split problems need.
size_t wgrad_ps_plan(const WgradParams *ps, int n, WgradBatch &tab, void *ws, bool allow_split, int na) {
  tab.n = n;
  long total_steps = 0;
  for (int i = 0; i < n; ++i) {
    const WgradParams &p = ps[i];
    WgradProblem &q = tab.p[i];
    q.dy = p.dy; q.x = p.x; q.dw = p.dw; q.slab = nullptr;
    const bool flat = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;   // one long row of pixels
    q.N = flat ? 1 : p.N; q.H = flat ? 1 : p.H; q.W = flat ? p.N * p.H * p.W : p.W;
    q.Ho = flat ? 1 : p.Ho; q.Wo = flat ? p.N * p.H * p.W : p.Wo;
    q.Cin = p.Cin; q.Cout = p.Cout; q.dy_ps = p.dy_ps; q.x_ps = p.x_ps;
    q.KH = p.KH; q.KW = p.KW; q.stride = p.stride; q.pad = p.pad; q.dil = p.dil;
    q.gx = sn_div_up(p.Cout, 128 * na); q.gy = sn_div_up(p.Cin, 128); q.taps = p.KH * p.KW;
    q.cpr = sn_div_up(q.Wo, 32);
    q.nunits = q.N * q.Ho * q.cpr;
    q.slab_stride = (size_t)p.Cout * q.taps * p.Cin;
    total_steps += (long)q.gx * q.gy * q.taps * sn_div_up(q.nunits, 2);
  }
  // Longest job (`cap`, in K-steps).  All jobs of a launch run in rounds of 256 (one workgroup per CU), so the launch takes about
  // rounds(cap) * cap with rounds = ceil(jobs(cap) / 256): dividing the longest problem's K by s = 1, 2, ... the cheapest s wins
  // (126 whole-K jobs of 320 steps: s = 2 -> 252 jobs of 160, one round; s = 3 -> 378 jobs of 107, TWO rounds = 214).  A split
  // costs its slabs and a reduce launch (weighted 5 %), jobs under 16 steps are all fill and drain.
  long cap = g_wgrad_job_steps.load(std::memory_order_relaxed);
  if (cap <= 0) {
    long longest = 1;
    for (int i = 0; i < n; ++i) { const long st = sn_div_up(tab.p[i].nunits, 2); longest = st > longest ? st : longest; }
    double best = 1e300;
    cap = longest;
    for (int sdiv = 1; sdiv <= 32; ++sdiv) {
      const long c = (longest + sdiv - 1) / sdiv;
      if (c < 16 && sdiv > 1) break;
      long jobs = 0;
      bool any_split = false;
      for (int i = 0; i < n; ++i) {
        const long st = sn_div_up(tab.p[i].nunits, 2), sp = (st + c - 1) / c;
        jobs += (long)tab.p[i].gx * tab.p[i].gy * tab.p[i].taps * sp;
        any_split |= sp > 1;
      }
      const double cost = (double)((jobs + 255) / 256) * (double)(c + 12) * (any_split ? 1.05 : 1.0);   // + 12: a job's fill and drain
      if (cost < best - 1e-9) { best = cost; cap = c; }
    }
    (void)total_steps;
  }
  if (!allow_split) cap = 1l << 30;     // no scratch: every job runs its whole K range (one owner per element, same result)
  size_t off = 0;
  int item = 0;
  for (int i = 0; i < n; ++i) {
    WgradProblem &q = tab.p[i];
    const int steps = sn_div_up(q.nunits, 2);
    int splits = sn_div_up(steps, (int)cap);
    if (splits < 1) splits = 1;
    int sps = sn_div_up(steps, splits);            // K-steps per split
    splits = sn_div_up(steps, sps);
    q.units_per_split = 2 * sps;
    q.splits = splits;
    if (splits > 1) {
      if (ws) q.slab = reinterpret_cast<float *>(static_cast<char *>(ws) + off);
      off += sn_align(sizeof(float) * (size_t)splits * q.slab_stride);
    }
    q.item0 = item;
    item += q.gx * q.gy * q.taps * splits;
  }
  tab.total_items = item;
  // hardware block b runs on XCD b % 8: give every XCD a contiguous range of the problem-major job list holding about 1/8 of the
  // K-steps (jobs of different problems differ in length), so a problem's tiles share one L2 and the XCDs finish together
  long work = 0;
  for (int i = 0; i < n; ++i) work += (long)(tab.p[i].units_per_split / 2) * tab.p[i].gx * tab.p[i].gy * tab.p[i].taps * tab.p[i].splits;
  {
    int x = 1, pi = 0, longest = 0;
    long done = 0;
    tab.xcd_start[0] = 0;
    for (int it = 0; it < item && x < 8; ++it) {
      while (pi + 1 < n && it >= tab.p[pi + 1].item0) ++pi;
      done += tab.p[pi].units_per_split / 2;
      if (done * 8 >= work * x) tab.xcd_start[x++] = it + 1;
    }
    for (; x <= 8; ++x) tab.xcd_start[x] = item;
    tab.xcd_start[8] = item;
    for (x = 0; x < 8; ++x) longest = tab.xcd_start[x + 1] - tab.xcd_start[x] > longest ? tab.xcd_start[x + 1] - tab.xcd_start[x] : longest;
    tab.per_xcd = longest;
  }
  tab.trace = g_wgrad_trace.load(std::memory_order_relaxed);
  return off;
}

int wgrad_ps_launch(const WgradBatch &tab, hipStream_t s, int na) {
  if (tab.total_items <= 0) return SN_OK;
  const dim3 grid(8u * (unsigned)tab.per_xcd);
  if (na == 2) {
    if (tab.trace) hipLaunchKernelGGL((wgrad_ps_kernel<3, 2, 4, true>), grid, dim3(768), 0, s, tab);
    else hipLaunchKernelGGL((wgrad_ps_kernel<3, 2, 4, false>), grid, dim3(768), 0, s, tab);
  } else {
    if (tab.trace) hipLaunchKernelGGL((wgrad_ps_kernel<4, 1, 4, true>), grid, dim3(512), 0, s, tab);
    else hipLaunchKernelGGL((wgrad_ps_kernel<4, 1, 4, false>), grid, dim3(512), 0, s, tab);
  }
  SN_CHECK_LAUNCH();
  long blocks = 0;
  for (int i = 0; i < tab.n; ++i) {
    const WgradProblem &q = tab.p[i];
    if (!q.slab) continue;
    const long b = (long)((q.slab_stride / 4 + 255) / 256);
    blocks += b < 1 ? 1 : (b > kReduceBlocksMax ? kReduceBlocksMax : b);
  }
  if (blocks > 0) {
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}
