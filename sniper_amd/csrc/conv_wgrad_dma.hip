// conv_wgrad_dma.hip -- weight gradient dW[co][tap][ci] += sum_pixels dY[pix][co] * X[src(pix, tap)][ci] with LDS-DMA staging.
//
// Replaces the weight-gradient half of the Convolution / FullyConnected operators of the un-vendored fork
// (symbols/faster/resnet_mx_101_e2e.py:43-66,121-155,256,288-303).  Both operands are K(pixel)-major in HBM, so tiles are
// staged as they lie -- [pixel][channel], 16-byte channel runs -- and transposed by the LDS read ds_read_b64_tr_b16 (as in
// conv_wgrad_tr_kernel, conv.hip).  Two things change against that kernel:
//
//  * staging is LDS-DMA (buffer_load_dwordx4 ... lds) into an S-deep ring, the swizzle applied to the per-lane SOURCE
//    address, one barrier per K-step and counted vmcnt waits -- no ds_write, no staging registers (see conv_dma.hip);
//  * a KxK convolution keeps ALL taps in one workgroup (wgrad_taps_dma_kernel): a K-step is one 32-pixel run of an output
//    row; its dY tile is staged ONCE and the three source rows it touches are staged once with a halo, and the nine taps
//    are nine row-shifted views of those rows in LDS.  The tap-per-workgroup grid of round 1 fetched every dY tile 9 times
//    and every X tile 9 times (196 MB per launch for a 21 MB problem, profiles/pmc_traffic.json).
#include "conv_common.h"

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef short short4v __attribute__((vector_size(8)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t *dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}

// one 8-deep MFMA fragment = two transposing reads, `second` half_t apart (16 tile rows)
template <int SECOND>
__device__ __forceinline__ half8 tr_frag2(const half_t *lds_tile, int off) {
  typedef __attribute__((address_space(3))) short4v *lds_v4;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_tile + off));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_tile + off + SECOND));
  union { short4v s[2]; half8 h; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.h;
}

constexpr unsigned kOob = 0xFFFFFF00u;

// ---------------------------------------------------------------------------------------------------------------------
// flat: one long row of p.W pixels (host passes N = H = Ho = 1, W = Wo = pixels).  Tile 128 (co) x 128 (ci), waves 2 x 2,
// K-step 64 pixels.  LDS image per operand and stage: 64 rows (pixels) x 256 B (128 channels), the 32-byte segment index
// XORed with (row & 7) exactly as conv_wgrad_tr_kernel writes it; a DMA instruction covers 4 rows (lane l: row l >> 4,
// 16-byte slot l & 15, source chunk = slot with its segment bits XORed).
template <int S, int MINW>
__global__ __launch_bounds__(256, MINW) void wgrad_flat_dma_kernel(const WgradParams p) {
  constexpr int MI = 4, NI = 4, TILE = 64 * 128, STAGE = 2 * TILE, L = 8;
  __shared__ __attribute__((aligned(1024))) half_t lds[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  int bx, by, split;
  if (!wgrad_block(p, bx, by, split)) return;
  const int co0 = bx * 128, ci0 = by * 128;
  const long px_begin = (long)split * p.units_per_split * 32;
  long px_end = px_begin + (long)p.units_per_split * 32;
  if (px_end > p.W) px_end = p.W;
  const int nk = px_end > px_begin ? (int)((px_end - px_begin + 63) / 64) : 0;

  const int r4 = lane >> 4, slot = lane & 15;
  const int r7 = ((wave & 1) << 2) | r4;                     // (tile row & 7) of this lane's rows: groups wave + 4 i
  const int gc = ((((slot >> 1) ^ r7) << 1) | (slot & 1));   // source 16-byte chunk that belongs in this slot
  const unsigned dy_ps_b = (unsigned)p.dy_ps * 2u, x_ps_b = (unsigned)p.x_ps * 2u;
  const bool a_cok = co0 + gc * 8 < p.Cout, b_cok = ci0 + gc * 8 < p.Cin;
  unsigned a_voff[4], b_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned row = (unsigned)(4 * (wave + 4 * i) + r4);
    a_voff[i] = a_cok ? row * dy_ps_b + (unsigned)(co0 + gc * 8) * 2u : kOob;
    b_voff[i] = b_cok ? row * x_ps_b + (unsigned)(ci0 + gc * 8) * 2u : kOob;
  }
  const char *dyb = reinterpret_cast<const char *>(p.dy), *xbp = reinterpret_cast<const char *>(p.x);
  half_t *const dst0 = lds + wave * 512;
  int g_t = 0;
  auto issue = [&](int buf) {
    const long base = px_begin + (long)g_t * 64;
    const long rem = px_end - base;   // > 0: issue is only called for steps < nk; rows beyond it read as zeros
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(dyb) + (size_t)base * dy_ps_b, 0, (int)(rem * dy_ps_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xbp) + (size_t)base * x_ps_b, 0, (int)(rem * x_ps_b), 0x00020000);
    half_t *const sa = dst0 + buf * STAGE, *const sb = sa + TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rdy, sa + i * 4 * 512, a_voff[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rxx, sb + i * 4 * 512, b_voff[i]);
    ++g_t;
  };

  // fragment reads: lane (fr, fq) points at row fq*4 + fr/4, 8-byte piece fr%4 of the fragment's 32-byte segment
  const int row0 = fq * 4 + (fr >> 2), q7 = row0 & 7;
  int a_off[MI], b_off[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_off[i] = row0 * 128 + ((((wm * 4 + i) ^ q7) << 4) | ((fr & 3) << 2));
#pragma unroll
  for (int jn = 0; jn < NI; ++jn) b_off[jn] = row0 * 128 + ((((wn * 4 + jn) ^ q7) << 4) | ((fr & 3) << 2));
  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) acc[i][jn] = floatx4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const half_t *const sa = lds + buf * STAGE, *const sb = sa + TILE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = tr_frag2<16 * 128>(sa + ks * 32 * 128, a_off[i]);
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) fb[jn] = tr_frag2<16 * 128>(sb + ks * 32 * 128, b_off[jn]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
    }
  };

#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s);
  int cur = 0, nxt = S - 1, t = 0;
  for (; t + S - 1 < nk; ++t) {
    wait_vmcnt<(S - 2) * L>();
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    compute(cur);
    cur = cur + 1 == S ? 0 : cur + 1;
    nxt = nxt + 1 == S ? 0 : nxt + 1;
  }
  for (; t < nk; ++t) {
    const int young = nk - 1 - t;
    if (S > 3 && young >= 2) wait_vmcnt<(S > 3 ? 2 : 0) * L>();
    else if (S > 2 && young == 1) wait_vmcnt<(S > 2 ? 1 : 0) * L>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute(cur);
    cur = cur + 1 == S ? 0 : cur + 1;
  }

  // lane (fr, fq): output channel co = ..+fr, 4 consecutive input channels ci = ..+fq*4 .. +3 -> one 16-byte access per (i, jn).
  // With K-splits the partial tile goes to this split's slab (plain stores; wgrad_reduce_kernel sums the slabs in split
  // order); a single split adds into dw directly -- each element has exactly one owner, so no atomics either way.
  float *dst = p.slab ? p.slab + (size_t)split * p.slab_stride : p.dw;
  const bool vec4 = (p.Cin % 4) == 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int co = co0 + wm * 64 + i * 16 + fr;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      const int ci = ci0 + wn * 64 + jn * 16 + fq * 4;
      if (ci >= p.Cin) continue;
      float *q = dst + (size_t)co * p.Cin + ci;
      if (vec4) {
        float4 v = make_float4(acc[i][jn][0], acc[i][jn][1], acc[i][jn][2], acc[i][jn][3]);
        if (!p.slab) {
          const float4 o = *reinterpret_cast<const float4 *>(q);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4 *>(q) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ci + r < p.Cin) q[r] = p.slab ? acc[i][jn][r] : q[r] + acc[i][jn][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 taps, stride 1, dilation d <= 4, any padding.  Tile 64 (co) x 64 (ci) x 9 taps; wave w owns input channels
// ci0 + 16 w .. +15 for all 64 co and all taps (36 accumulator fragments).  K-step = unit (img, oy, 32-pixel run ox0..):
//   A: dY[img, oy, ox0 .. ox0+31][co0 .. +63]                          32 rows x 128 B
//   B: X[img, oy - pad + kh d, ox0 - pad + s][ci0 .. +63], s = 0..39    3 x 40 rows x 128 B   (34 + 2(d-1) of the 40 used)
// tap (kh, kw) of output pixel ox0 + j reads B row kh*40 + j + kw d.  128-byte LDS rows, 32-byte segment index XORed with
// ((row >> 1) & 3): the 8 consecutive rows a ds_read_b64_tr_b16 service group touches (two 4-row blocks) then fall on 8
// distinct bank octets, for every tap shift.  A DMA instruction covers 8 rows (lane l: row l >> 3, slot l & 7).
// Per stage 4 (A) + 15 (B) + 1 (padding, all out of range) = 20 DMA instructions, 5 per wave.
template <int S>
__global__ __launch_bounds__(256, 2) void wgrad_taps_dma_kernel(const WgradParams p) {
  constexpr int B_OFF = 32 * 64, STAGE = B_OFF + 3 * 40 * 64 + 512, L = 5, ROW = 64;   // half_t units
  __shared__ __attribute__((aligned(1024))) half_t lds[S * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  int bx, by, split;
  if (!wgrad_block(p, bx, by, split)) return;
  const int co0 = bx * 64, ci0 = by * 64;
  const int cpr = (p.Wo + 31) / 32;
  const int nunits = p.N * p.Ho * cpr;
  const int u_begin = split * p.units_per_split, u_end = min(nunits, u_begin + p.units_per_split);
  const int nk = u_end > u_begin ? u_end - u_begin : 0;

  const int lr = lane >> 3, slot = lane & 7;
  const int gc = ((((slot >> 1) ^ ((lane >> 4) & 3)) << 1) | (slot & 1));   // (row >> 1) & 3 = (lr >> 1) & 3: groups are 8-row aligned
  const unsigned dy_ps_b = (unsigned)p.dy_ps * 2u, x_ps_b = (unsigned)p.x_ps * 2u;
  const bool a_cok = co0 + gc * 8 < p.Cout, b_cok = ci0 + gc * 8 < p.Cin;
  const unsigned a_voff = a_cok ? (unsigned)(8 * wave + lr) * dy_ps_b + (unsigned)(co0 + gc * 8) * 2u : kOob;
  const unsigned b_coff = (unsigned)(ci0 + gc * 8) * 2u;
  // B slots i = 0..3 of this wave: group b = wave + 4 i in 0..15 -> source row kh = b / 5, pixel group pg = b % 5 (b = 15: padding)
  int b_kh[4], b_pg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = wave + 4 * i;
    b_kh[i] = b / 5;
    b_pg[i] = b - b_kh[i] * 5;
  }
  const char *dyb = reinterpret_cast<const char *>(p.dy), *xbp = reinterpret_cast<const char *>(p.x);
  // unit iterator of the NEXT stage to issue (workgroup-uniform)
  int it_r = u_begin / cpr, it_xc = u_begin - it_r * cpr;
  int it_img = it_r / p.Ho, it_oy = it_r - it_img * p.Ho;
  auto issue = [&](int buf) {
    const int ox0 = it_xc * 32;
    half_t *const st = lds + buf * STAGE;
    {
      const size_t base = ((size_t)it_r * p.Wo + ox0) * dy_ps_b;
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(dyb) + base, 0, (int)((unsigned)(p.Wo - ox0) * dy_ps_b), 0x00020000);
      dma16(rdy, st + wave * 512, a_voff);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sy = it_oy - p.pad + b_kh[i] * p.dil;
      const bool rok = b_kh[i] < 3 && (unsigned)sy < (unsigned)p.H;
      const size_t base = rok ? ((size_t)(it_img * p.H + sy) * p.W) * x_ps_b : 0;
      const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xbp) + base, 0, rok ? (int)((unsigned)p.W * x_ps_b) : 0, 0x00020000);
      const int sx = ox0 - p.pad + 8 * b_pg[i] + lr;
      const bool ok = b_cok & ((unsigned)sx < (unsigned)p.W);
      dma16(rxx, st + B_OFF + (wave + 4 * i) * 512, ok ? (unsigned)sx * x_ps_b + b_coff : kOob);
    }
    if (++it_xc == cpr) {
      it_xc = 0;
      ++it_r;
      if (++it_oy == p.Ho) { it_oy = 0; ++it_img; }
    }
  };

  const int row0 = fq * 4 + (fr >> 2);
  int a_off[4], b_off[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = row0 * ROW + (((i ^ ((row0 >> 1) & 3)) << 4) | ((fr & 3) << 2));
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int row = row0 + kw * p.dil;
    b_off[kw] = row * ROW + (((wave ^ ((row >> 1) & 3)) << 4) | ((fr & 3) << 2));
  }
  floatx4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) acc[i][tp] = floatx4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const half_t *const st = lds + buf * STAGE;
    half8 fa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = tr_frag2<16 * ROW>(st, a_off[i]);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const half8 fb = tr_frag2<16 * ROW>(st + B_OFF + kh * 40 * ROW, b_off[kw]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, fa[i], acc[i][kh * 3 + kw], 0, 0, 0);
      }
  };

#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s);
  int cur = 0, nxt = S - 1, t = 0;
  for (; t + S - 1 < nk; ++t) {
    wait_vmcnt<(S - 2) * L>();
    __builtin_amdgcn_s_barrier();
    issue(nxt);
    compute(cur);
    cur = cur + 1 == S ? 0 : cur + 1;
    nxt = nxt + 1 == S ? 0 : nxt + 1;
  }
  for (; t < nk; ++t) {
    const int young = nk - 1 - t;
    if (S > 3 && young >= 2) wait_vmcnt<(S > 3 ? 2 : 0) * L>();
    else if (S > 2 && young == 1) wait_vmcnt<(S > 2 ? 1 : 0) * L>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute(cur);
    cur = cur + 1 == S ? 0 : cur + 1;
  }

  float *dst = p.slab ? p.slab + (size_t)split * p.slab_stride : p.dw;
  const bool vec4 = (p.Cin % 4) == 0;
  const int ci = ci0 + wave * 16 + fq * 4;
  if (ci < p.Cin) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + i * 16 + fr;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        float *q = dst + ((size_t)co * 9 + tp) * p.Cin + ci;
        if (vec4) {
          float4 v = make_float4(acc[i][tp][0], acc[i][tp][1], acc[i][tp][2], acc[i][tp][3]);
          if (!p.slab) {
            const float4 o = *reinterpret_cast<const float4 *>(q);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *reinterpret_cast<float4 *>(q) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ci + r < p.Cin) q[r] = p.slab ? acc[i][tp][r] : q[r] + acc[i][tp][r];
        }
      }
    }
  }
}

// kind 1 / 2 as in conv_common.h; stages: 2..4 (flat: 2 or 3; taps: 3 or 4)
extern int g_wgrad_xcd;
int wgrad_dma_launch(const WgradParams &p_in, int kind, int stages, int splits, hipStream_t s) {
  WgradParams p = p_in;
  if (kind == 1) {
    const dim3 grid(wgrad_grid(p, sn_div_up(p.Cout, 128), sn_div_up(p.Cin, 128), splits, g_wgrad_xcd != 0));
    if (stages == 3) hipLaunchKernelGGL((wgrad_flat_dma_kernel<3, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_flat_dma_kernel<2, 2>), grid, dim3(256), 0, s, p);
  } else {
    const dim3 grid(wgrad_grid(p, sn_div_up(p.Cout, 64), sn_div_up(p.Cin, 64), splits, g_wgrad_xcd != 0));
    if (stages == 4) hipLaunchKernelGGL((wgrad_taps_dma_kernel<4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wgrad_taps_dma_kernel<3>), grid, dim3(256), 0, s, p);
  }
  SN_CHECK_LAUNCH();
  return SN_OK;
}
