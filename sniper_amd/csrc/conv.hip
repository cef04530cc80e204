// conv.hip -- implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16),
// fp16 storage / fp32 accumulate, channels-last (NHWC) activations.
//
// Replaces the Convolution / FullyConnected operators of the un-vendored SNIPER-mxnet fork at the
// call sites symbols/faster/resnet_mx_101_e2e.py:43-66,121-155,256,288-303 (forward, data gradient,
// weight gradient).  Definitions follow the standard cross-correlation the reference's symbols ask
// for (kernel, stride, pad, dilate, no_bias); parity is checked against oracle/nn.py.
//
// One GEMM view serves forward and data-gradient:
//     Y[m][n] = sum_{tap, c} A(m, tap, c) * Wt[n][tap][c]        m = (img, y, x) output pixel
//   forward : A = X[img][y*s - p + kh*d][x*s - p + kw*d][c]      Wt = W as [Cout][KH*KW][Cin]
//   dgrad   : A = dY[img][(y + p - kh*d)/s][(x + p - kw*d)/s][c] Wt = W as [Cin][KH*KW][Cout]
// (taps whose source pixel is out of range, or not on the stride lattice for dgrad, contribute
// zero).  No im2col buffer exists: the (tap, channel-chunk) loop gathers 16-byte channel runs of
// the source pixel straight into an LDS tile.  Both MFMA operands are therefore K-contiguous
// ("A row-major, B^T row-major"), which is what makes NHWC the natural layout on CDNA: every
// fragment is one ds_read_b128.
//
// Tile: BM x BN outputs per 256-thread workgroup (4 waves as 2x2), BK = 32 (one MFMA K-step),
// register-staged double-buffered LDS, one barrier per K-step.  LDS rows are 64 B; the 16-byte
// chunk index is XOR-swizzled with g((row>>2)&3), g = {0,2,3,1}, which makes every ds_read_b128
// lane group (MI355X_MICROARCH.md, LDS table) hit 16 distinct 16-B slots.
//
// The weight gradient needs both operands transposed (the contraction index is the pixel, which is
// the slow dimension of both dY and X); see conv_wgrad_kernel below.
#include "conv_common.h"

#include <stdlib.h>

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3); }

template <int BM, int BN, bool DGRAD>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int BK = 32;
  constexpr int WM = BM / 2, WN = BN / 2;   // per-wave tile
  constexpr int MI = WM / 16, NI = WN / 16;  // 16x16 MFMA tiles per wave
  constexpr int AR = BM / 64, BR = BN / 64;  // 16-byte chunks per thread per K-step
  __shared__ __attribute__((aligned(16))) half_t sA[2][BM * BK];
  __shared__ __attribute__((aligned(16))) half_t sB[2][BN * BK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int chunk = tid & 3, lrow = tid >> 2;

  // ---- per-thread gather state for its A rows
  int a_base[AR], a_h[AR], a_w[AR];
  bool a_ok[AR];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lrow + 64 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int img = mm / HoWo, rem = mm - img * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    a_base[i] = img * p.H * p.W;
    if (DGRAD) { a_h[i] = oy + p.pad; a_w[i] = ox + p.pad; }
    else { a_h[i] = oy * p.stride - p.pad; a_w[i] = ox * p.stride - p.pad; }
  }
  const int taps = p.KH * p.KW;
  const int kpt = (p.Cin + BK - 1) / BK;  // K-steps per tap
  const int nk = taps * kpt;
  const int wrow_stride = taps * p.Cin;   // elements per weight row

  half8 ra[AR], rb[BR];
  auto gload = [&](int kt) {
    const int tap = kt / kpt, c0 = (kt - tap * kpt) * BK + chunk * 8;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const bool c_ok = c0 < p.Cin;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int sy, sx;
      bool ok = a_ok[i] && c_ok;
      if (DGRAD) {
        const int ty = a_h[i] - kh * p.dil, tx = a_w[i] - kw * p.dil;
        sy = ty / p.stride; sx = tx / p.stride;
        ok = ok && ty >= 0 && tx >= 0 && (sy * p.stride == ty) && (sx * p.stride == tx) && sy < p.H && sx < p.W;
      } else {
        sy = a_h[i] + kh * p.dil; sx = a_w[i] + kw * p.dil;
        ok = ok && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
      }
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) v = *reinterpret_cast<const half8 *>(p.x + (size_t)(a_base[i] + sy * p.W + sx) * p.in_ps + c0);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int n = n0 + lrow + 64 * i;
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (n < p.Nout && c_ok) v = *reinterpret_cast<const half8 *>(p.w + (size_t)n * wrow_stride + tap * p.Cin + c0);
      rb[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int r = lrow + 64 * i;
      *reinterpret_cast<half8 *>(&sA[buf][(r * 4 + swz(r, chunk)) * 8]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int r = lrow + 64 * i;
      *reinterpret_cast<half8 *>(&sB[buf][(r * 4 + swz(r, chunk)) * 8]) = rb[i];
    }
  };

  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;  // fragment row/col and k-chunk of this lane
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);  // next tile's HBM/L2 latency hides under this tile's MFMAs
    half8 fa[MI], fb[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * WM + i * 16 + fr;
      fa[i] = *reinterpret_cast<const half8 *>(&sA[cur][(r * 4 + swz(r, fq)) * 8]);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * WN + j * 16 + fr;
      fb[j] = *reinterpret_cast<const half8 *>(&sB[cur][(r * 4 + swz(r, fq)) * 8]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: D layout col = lane&15 (channel), row = (lane>>4)*4 + reg (pixel)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * WM + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 16 + fr;
        if (n >= p.Nout) continue;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[n];
        if (p.res) v += (float)p.res[(size_t)m * p.res_ps + n];
        if (p.relu) v = v > 0.f ? v : 0.f;
        if (p.out_f32) reinterpret_cast<float *>(p.y)[(size_t)m * p.out_ps + n] = v;
        else reinterpret_cast<half_t *>(p.y)[(size_t)m * p.out_ps + n] = (half_t)v;
      }
    }
  }
}

// ---- pipelined variant: BK = 64, two register sets in flight ------------------------------------
// The BK = 32 kernel above hides one K-step (~0.1 us of MFMA) of global-load latency; measured, a K-step costs
// ~1 us even with 2.5 workgroups per CU (rpn 3x3: 612 TFLOP/s).  A 128x128 fp16 tile needs ~150 GB/s per CU at the
// MFMA rate, i.e. ~100 KB in flight per CU at ~1 us of loaded latency.  This variant keeps TWO tiles (2 x 32 KB per
// workgroup) in flight in registers ahead of the one being multiplied and halves the barriers per FLOP:
//   step t:  ds_write tile t+1 (set (t+1)&1, issued two steps ago) -> LDS[(t+1)&1]
//            issue global loads of tile t+3 into that set
//            16 ds_read_b128 + 32 MFMA on LDS[t&1]
//            barrier
// LDS rows are 128 B (64 channels); 16-byte chunk c of row r sits at c ^ (r & 7): conflict-free for the 8-lane
// ds_write_b128 groups (8 chunks of one row) and the 16-lane ds_read_b128 groups (16 rows, chunk q / q+1).
// Workgroup -> tile mapping is XCD-aware: the Nout/128 column tiles that share one 128-row A panel run back to back
// on the SAME XCD (linear id % 8 = XCD), so the panel is fetched into one L2 once instead of once per column tile.
// BM = 128 for launches that fill the chip; BM = 64 (wave tile 32 x 64, 48 KB LDS, 3 workgroups per CU) for the many
// R101 layers whose 128-row tiling yields only ~1 workgroup per CU: those launches are latency-bound (each workgroup
// waits ~1 us per K-step on its own two tiles in flight), so more, smaller workgroups per CU finish sooner even though
// each reads its B panel for half the rows.
template <bool DGRAD, int BM>
__global__ __launch_bounds__(256, 2) void conv_igemm_p2_kernel(const ConvParams p, int mtiles, int ntiles) {
  constexpr int BN = 128, BK = 64, WM = BM / 2;
  constexpr int MI = BM / 32, NI = 4, AR = BM / 32, BR = 4;   // 16-byte chunks per thread per tile: rows x 8 chunks / 256
  __shared__ __attribute__((aligned(16))) half_t sA[2][BM * BK];
  __shared__ __attribute__((aligned(16))) half_t sB[2][BN * BK];

  const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
  const int nt = j % ntiles, mt = (j / ntiles) * 8 + xcd;
  if (mt >= mtiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = mt * BM, n0 = nt * BN;
  const int chunk = tid & 7, lrow = tid >> 3;   // rows lrow + 32*i

  int a_base[AR], a_h[AR], a_w[AR];
  bool a_ok[AR];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lrow + 32 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int img = mm / HoWo, rem = mm - img * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    a_base[i] = img * p.H * p.W;
    if (DGRAD) { a_h[i] = oy + p.pad; a_w[i] = ox + p.pad; }
    else { a_h[i] = oy * p.stride - p.pad; a_w[i] = ox * p.stride - p.pad; }
  }
  const int taps = p.KH * p.KW;
  const int kpt = p.Cin / BK;          // host guarantees Cin % 64 == 0
  const int nk = taps * kpt;
  const unsigned wrow_bytes = (unsigned)(taps * p.Cin) * 2u;
  // rows lrow + 32*i share (row & 7): one swizzled LDS store address, the others are immediates (+32 rows = 4 KB)
  const int st_off = (lrow * 8 + (chunk ^ (lrow & 7))) * 8;
  // weight rows n0 + lrow + 32*i: byte offset of row i = w_off0 + i * 32 * wrow_bytes (zero-filled beyond Nout)
  const unsigned w_off0 = (unsigned)(n0 + lrow) * wrow_bytes + (unsigned)chunk * 16u;
  const char *xb = reinterpret_cast<const char *>(p.x), *wb = reinterpret_cast<const char *>(p.w);
  const unsigned in_ps_bytes = (unsigned)p.in_ps * 2u;

  // Loads go through buffer descriptors (buffer_load_dwordx4 ... offen): a row that must read as zero -- padded tap,
  // pixel beyond M, weight row beyond Nout -- simply carries an out-of-range voffset and the hardware returns 0, so
  // there is no branch and no select anywhere.  K-steps are visited in order: the per-row voffsets change only when
  // the tap changes (the integer divisions of the tap decomposition cost ~40 VALU instructions each on CDNA and
  // dominated the step before); inside a tap a step only moves the descriptors' base (scalar ALU).  The loop body is
  // 32 MFMA + 16 ds_read + 8 ds_write + 8 buffer_load and a handful of VALU.
  constexpr unsigned kOob = 0xFFFFFF00u;
  unsigned w_voff[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) w_voff[i] = (n0 + lrow + 32 * i < p.Nout) ? w_off0 + (unsigned)i * 32u * wrow_bytes : kOob;
  int g_kh = 0, g_kw = 0, g_kc = 0, g_kt = 0;   // next tile to load: tap (g_kh, g_kw), channel block g_kc, K-step g_kt
  unsigned a_voff[AR];
  auto tap_setup = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int sy, sx;
      bool ok = a_ok[i];
      if (DGRAD) {
        const int ty = a_h[i] - g_kh * p.dil, tx = a_w[i] - g_kw * p.dil;
        if (p.stride == 1) { sy = ty; sx = tx; }
        else { sy = ty / p.stride; sx = tx / p.stride; ok = ok && (sy * p.stride == ty) && (sx * p.stride == tx); }
        ok = ok && ty >= 0 && tx >= 0 && sy < p.H && sx < p.W;
      } else {
        sy = a_h[i] + g_kh * p.dil; sx = a_w[i] + g_kw * p.dil;
        ok = ok && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
      }
      a_voff[i] = ok ? (unsigned)(a_base[i] + sy * p.W + sx) * in_ps_bytes + (unsigned)chunk * 16u : kOob;
    }
  };
  tap_setup();
  auto gload = [&](half8 (&ra)[AR], half8 (&rb)[BR]) {
    const unsigned cbo = (unsigned)g_kc * (BK * 2), wbo = (unsigned)g_kt * (BK * 2);   // uniform
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb) + cbo, 0, (int)(p.x_bytes - cbo), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb) + wbo, 0, (int)(p.w_bytes - wbo), 0x00020000);
#pragma unroll
    for (int i = 0; i < AR; ++i) ra[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rx, a_voff[i], 0, 0));
#pragma unroll
    for (int i = 0; i < BR; ++i) rb[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff[i], 0, 0));
    ++g_kt;
    if (++g_kc == kpt) {
      g_kc = 0;
      if (++g_kw == p.KW) { g_kw = 0; ++g_kh; }
      tap_setup();
    }
  };
  auto lstore = [&](int buf, const half8 (&ra)[AR], const half8 (&rb)[BR]) {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<half8 *>(&sA[buf][st_off + i * 32 * BK]) = ra[i];
#pragma unroll
    for (int i = 0; i < BR; ++i) *reinterpret_cast<half8 *>(&sB[buf][st_off + i * 32 * BK]) = rb[i];
  };

  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) acc[i][jn] = floatx4{0.f, 0.f, 0.f, 0.f};

  // fragment rows wm*64 + i*16 + fr all share (row & 7) = fr & 7: one base per k-substep, rows by immediates
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = fq ^ (fr & 7);
  const int a_rd = (wm * WM + fr) * BK, b_rd = (wn * 64 + fr) * BK;
  // The product is formed TRANSPOSED (weights as the MFMA A operand): D^T[n][m] puts 4 consecutive output channels
  // n = fq*4 + r of one pixel m = fr into each lane, so the epilogue stores 8 bytes per (i, jn) instead of 4 x 2.
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = (sw ^ (ks * 4)) * 8;
      half8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const half8 *>(&sA[buf][a_rd + i * 16 * BK + co]);
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) fb[jn] = *reinterpret_cast<const half8 *>(&sB[buf][b_rd + jn * 16 * BK + co]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
    }
  };

  half8 ra0[AR], rb0[BR], ra1[AR], rb1[BR];
  int t = 0;
  gload(ra0, rb0);
  if (nk > 4) {
    // Steady state WITHOUT a branch around any load or store.  The compiler places s_waitcnt vmcnt(N) from a dataflow
    // model of the outstanding loads; a conditional gload / lstore inside the loop (or in the code leading to it) makes
    // the two joined paths disagree, the merge is conservative, and every ds_write of tile t+1 ended up behind
    // `s_waitcnt vmcnt(0)` -- i.e. behind the loads of tile t+2 issued one step earlier: ONE tile in flight, not two (seen
    // in the ISA: vmcnt(7)..vmcnt(0) before the eight ds_write_b128).  Straight-line code from the first load on lets it
    // count exactly: the stores of one register set wait with the other set's eight loads still in flight.
    gload(ra1, rb1);
    lstore(0, ra0, rb0);
    gload(ra0, rb0);
    __syncthreads();
    for (; t + 4 < nk; t += 2) {
      lstore(1, ra1, rb1);      // even step t: tile t+1 lives in set 1, tile t+2 in set 0
      gload(ra1, rb1);          // tile t+3
      compute(0);
      __syncthreads();
      lstore(0, ra0, rb0);      // odd step t+1: tile t+2 lives in set 0, tile t+3 in set 1
      gload(ra0, rb0);          // tile t+4
      compute(1);
      __syncthreads();
    }
  } else {
    if (nk > 1) gload(ra1, rb1);
    lstore(0, ra0, rb0);
    if (nk > 2) gload(ra0, rb0);
    __syncthreads();
  }
  // tail (and the whole loop of a short contraction): the guarded form; tiles t+1 / t+2 are in flight as above
  for (; t < nk; t += 2) {
    if (t + 1 < nk) lstore(1, ra1, rb1);
    if (t + 3 < nk) gload(ra1, rb1);
    compute(0);
    __syncthreads();
    if (t + 1 >= nk) break;
    if (t + 2 < nk) lstore(0, ra0, rb0);
    if (t + 4 < nk) gload(ra0, rb0);
    compute(1);
    __syncthreads();
  }

  // ---- epilogue: lane (fr, fq) holds, for each (i, jn), pixel m = ..+fr and channels n = ..+fq*4 .. +3
  const bool vec = (p.out_ps % 4 == 0) && (p.Nout % 4 == 0) && (!p.res || p.res_ps % 4 == 0);
  float st_s[NI][4], st_q[NI][4];        // BatchNorm statistics of this lane's 16 output channels (host: only with `vec`)
#pragma unroll
  for (int jn = 0; jn < NI; ++jn)
#pragma unroll
    for (int r = 0; r < 4; ++r) st_s[jn][r] = st_q[jn][r] = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WM + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      const int n = n0 + wn * 64 + jn * 16 + fq * 4;
      if (n >= p.Nout) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][jn][r];
      if (vec) {
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (p.res) {
          const half4 rv = *reinterpret_cast<const half4 *>(p.res + (size_t)m * p.res_ps + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
        if (p.out_f32) {
          *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.y) + (size_t)m * p.out_ps + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          half4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
          *reinterpret_cast<half4 *>(reinterpret_cast<half_t *>(p.y) + (size_t)m * p.out_ps + n) = o;
          if (p.stats) {
            if (p.bn_x) {
              const half4 xv = *reinterpret_cast<const half4 *>(p.bn_x + (size_t)m * p.bn_x_ps + n);
              const float4 sc = *reinterpret_cast<const float4 *>(p.bn_scale + n), sh = *reinterpret_cast<const float4 *>(p.bn_shift + n);
              const float4 mu = *reinterpret_cast<const float4 *>(p.bn_mean + n);
              const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float xf = (float)xv[r], yv = xf * scv[r] + shv[r];
                // same mask as bn_act_pass (nn_ops.hip): 0 none, 1 relu (y > 0), 2 relu6 (0 <= y <= 6)
                const bool pass = p.bn_act == 0 || (p.bn_act == 1 ? yv > 0.f : (yv >= 0.f && yv <= 6.f));
                const float gf = pass ? (float)o[r] : 0.f;
                st_s[jn][r] += gf;
                st_q[jn][r] += gf * (xf - muv[r]);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float f = (float)o[r];
                st_s[jn][r] += f;
                st_q[jn][r] += f * f;
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.Nout) continue;
          float x = v[r];
          if (p.bias) x += p.bias[n + r];
          if (p.res) x += (float)p.res[(size_t)m * p.res_ps + n + r];
          if (p.relu) x = x > 0.f ? x : 0.f;
          if (p.out_f32) reinterpret_cast<float *>(p.y)[(size_t)m * p.out_ps + n + r] = x;
          else reinterpret_cast<half_t *>(p.y)[(size_t)m * p.out_ps + n + r] = (half_t)x;
        }
      }
    }
  }
  if (p.stats) {
    // rows: the 16 lanes that share fq hold different pixels of the same 4 channels -> xor-shuffle over fr, then the two
    // waves of a column pair (wm = 0, 1) through LDS (the K loop is over: sA is free); fixed order -> deterministic
#pragma unroll
    for (int jn = 0; jn < NI; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          st_s[jn][r] += __shfl_xor(st_s[jn][r], off, 64);
          st_q[jn][r] += __shfl_xor(st_q[jn][r], off, 64);
        }
      }
    float *red = reinterpret_cast<float *>(&sA[0][0]);   // [wm][wn][2][64]
    __syncthreads();
    if (fr == 0) {
#pragma unroll
      for (int jn = 0; jn < NI; ++jn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = jn * 16 + fq * 4 + r;
          red[((wm * 2 + wn) * 2 + 0) * 64 + c] = st_s[jn][r];
          red[((wm * 2 + wn) * 2 + 1) * 64 + c] = st_q[jn][r];
        }
    }
    __syncthreads();
    // 256 threads: (which, column of the 128-wide tile)
    const int which = tid >> 7, col = tid & 127, cw = col >> 6, cc = col & 63;
    const int n = n0 + col;
    if (n < p.Nout) {
      const float a = red[((0 * 2 + cw) * 2 + which) * 64 + cc] + red[((1 * 2 + cw) * 2 + which) * 64 + cc];
      p.stats[((size_t)mt * 2 + which) * p.Nout + n] = a;
    }
  }
}

static int conv_check(const ConvParams &p, const char *who) {
  SN_REQUIRE(p.x && p.w && p.y, "%s: null pointer", who);
  SN_REQUIRE(p.N > 0 && p.H > 0 && p.W > 0 && p.Ho > 0 && p.Wo > 0 && p.Nout > 0, "%s: bad dims", who);
  SN_REQUIRE(p.Cin > 0 && p.Cin % 8 == 0, "%s: channels per tap must be a multiple of 8 (got %d)", who, p.Cin);
  // 16-byte gathers: pixel stride * 2 B must keep every (pixel, chunk) address 16-byte aligned.  The
  // packed 4-channel stem input (in_ps == 4) is the one exception: its gathers start on even pixels.
  SN_REQUIRE(p.in_ps % 8 == 0 || (p.in_ps == 4 && p.stride % 2 == 0 && p.KW == 1 && p.pad == 0 && p.W % 2 == 0),
             "%s: source pixel stride must be a multiple of 8 elements (got %d)", who, p.in_ps);
  SN_REQUIRE(p.KH > 0 && p.KW > 0 && p.stride > 0 && p.dil > 0 && p.pad >= 0, "%s: bad kernel geometry", who);
  SN_REQUIRE((long)p.N * p.H * p.W * p.in_ps < (1l << 31) && (long)p.M * p.out_ps < (1l << 31),
             "%s: tensor too large for 32-bit offsets", who);
  return SN_OK;
}

// which kernel a layer takes: dma = 1..kConvDmaConfigs -> conv_dma_kernel (conv_dma.hip) with that tile configuration;
// else BM = 64 / 128 -> conv_igemm_p2_kernel, 0 -> conv_igemm_kernel
struct ConvPlan {
  int bm, mtiles, ntiles, dma;
  unsigned x_bytes, w_bytes;
};
// Tuning override (tools/conv_tune.py, tests): -1 = the built-in table, 0 = register-staged kernels only, 1..kConvDmaConfigs = that
// LDS-DMA configuration for every layer that qualifies.  Process-wide; not meant to change while launches are in flight.
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static int g_conv_cfg = env_int("SNIPER_CONV_CFG", -1);   // A/B runs of whole programs (bench.py) without code changes
SN_EXPORT int sn_conv_tune(int cfg) {
  SN_REQUIRE(cfg >= -1 && cfg <= kConvDmaConfigs, "sn_conv_tune: configuration %d out of range", cfg);
  g_conv_cfg = cfg;
  return SN_OK;
}

// Built-in choice for a layer that qualifies for the pipelined kernels: M output pixels, Nout channels, nk 64-deep K-steps.
// Table measured on MI355X with tools/conv_tune.py (profiles/r02_conv_tune.txt).
static int conv_dma_choice_warm(int M, int Nout, int nk) {
  // table 1: tools/conv_tune.py on L2-warm operands (profiles/r02_conv_tune.txt)
  const long t128 = (long)sn_div_up(M, 128) * sn_div_up(Nout, 128), t64 = (long)sn_div_up(M, 64) * sn_div_up(Nout, 128);
  if (Nout <= 128) return nk >= 64 ? 5 : 6;          // one column tile: 64-row tiles, deeper ring for long contractions
  if (t128 >= 3840 && nk >= 8) return 7;              // >= 3.75 tiles of 256 x 256 per CU: the tile with the least LDS / L2 bytes per FLOP
  if (nk >= 128) return (Nout >= 1024 && M < 8192) ? 4 : 1;
  if (t64 <= 2560) return 6;                          // <= 10 resident 64 x 128 workgroups per CU over the launch: 3 per CU co-resident
  return 1;
}

static int conv_dma_choice_cold(int M, int Nout, int nk) {
  // table 2: tools/conv_tune.py --cold 600 --insitu (profiles/r02_conv_tune_insitu.txt): operands no XCD has cached and the
  // epilogues the training step uses (BatchNorm statistics, residual / accumulate)
  const long t128 = (long)sn_div_up(M, 128) * sn_div_up(Nout, 128);
  if (Nout <= 128) return nk >= 64 ? 5 : 6;
  if (t128 >= 3840 && nk >= 8) return 7;
  if (Nout <= 256) return M <= 32768 ? 4 : 1;           // 128 x 256: the whole channel range in one tile, A panel fetched once
  if (Nout <= 512 && M <= 32768 && nk >= 16) return 7;  // stage 4 / RPN / deformable GEMM at 20 480 pixels: 160 tiles of 256 x 256
  if (nk >= 128 || M < 8192) return (Nout >= 1024 && M < 8192) ? 4 : 1;   // FC over 6000 RoIs
  return 1;
}

static int conv_dma_choice_balanced(int M, int Nout, int nk, bool dgrad) {
  // table 3: tools/conv_tune.py --insitu with the 160-row tiles in the candidate set (profiles/r02_conv_tune_insitu_v3.txt).
  // tools/conv_trace.py shows why they win at 20 chips: a K-step's operand delivery is an LDS-DMA issue cost per wave, so a
  // CU wants >= 8 waves in K loops (two 4-wave workgroups) and every CU the same number of tiles -- 20 480 pixels / 160 = 128
  // row tiles = 256 / 512 / 1024 / 2048 workgroups for 256 / 512 / 1024 / 2048 output channels, 81 920 / 160 = 512.
  const long t128 = (long)sn_div_up(M, 128) * sn_div_up(Nout, 128);
  if (t128 >= 3840 && nk >= 8) return 7;              // the long-K data gradients (RPN, deformable GEMM, fc_new_1): 256 x 256
  if (Nout < 128) return nk >= 64 ? 5 : 6;           // narrow heads: one partial column tile
  if (M < 8192) return (Nout >= 1024 && nk >= 128) ? 4 : 6;   // FullyConnected over 6000 RoIs
  static const int bal = env_int("SNIPER_CONV_BAL", 14), bal_d = env_int("SNIPER_CONV_BAL_D", 16);
  if (nk <= 1) return 6;                              // stage 1: a single K-step, all epilogue
  // round 3: long contractions with about one 160 x 128 tile per CU take the producer / consumer specialised kernel (cfg 18):
  // its K loop is ~1.3x faster (profiles/r03_conv_tune_ps.txt: stage-3 3x3 38 -> 30 us, 1x1 1024 -> 256 24 -> 21 us), but it is one
  // 8-wave workgroup per CU, so short contractions with 4 tiles per CU -- all fill and epilogue -- lose (32 -> 47 us)
  static const int ps = env_int("SNIPER_CONV_PS", 18), ps_nk = env_int("SNIPER_CONV_PS_NK", 16);
  if (ps > 0 && nk >= ps_nk && (long)sn_div_up(M, 160) * sn_div_up(Nout, 128) <= 320) return ps;
  return dgrad ? bal_d : bal;
}

// SNIPER_CONV_TABLE = 1 | 2 | 3 picks the table; SNIPER_CONV_N128 / _N256 / _N512 / _NBIG override the configuration of a whole
// output-width class (A/B runs of bench.py: the step itself is the only measurement that includes what precedes each launch).
static int conv_dma_choice(int M, int Nout, int nk, bool dgrad) {
  // default: the balanced table for both directions, 160 x 128 tiles with 4 waves (cfg 14) forward and 8 waves (cfg 16) for the
  // data gradients -- whole-step A/B on one box each (tools/conv_ab.sh, profiles/r02_ab_balanced_tiles.txt):
  //   tables (fwd, dgrad) = (1, 1) 29.44 ms, (3, 3) 29.16, (3, 1) 28.60 with cfg 14 everywhere and 8-byte epilogue stores;
  //   with 16-byte stores: (3, 1) 26.93 ms, (3, 3 / dgrad cfg 16) 26.60, (3, 3 / dgrad cfg 14) 27.59, (3, 3 / dgrad cfg 17) 26.76
  static const int table_f = env_int("SNIPER_CONV_TABLE", 3), table_d = env_int("SNIPER_CONV_TABLE_DGRAD", table_f);
  const int table = dgrad ? table_d : table_f;
  static const int o128 = env_int("SNIPER_CONV_N128", -1), o256 = env_int("SNIPER_CONV_N256", -1),
                   o512 = env_int("SNIPER_CONV_N512", -1), obig = env_int("SNIPER_CONV_NBIG", -1);
  const int o = Nout <= 128 ? o128 : Nout <= 256 ? o256 : Nout <= 512 ? o512 : obig;
  if (o >= 0) return o;
  return table == 3 ? conv_dma_choice_balanced(M, Nout, nk, dgrad) : table == 2 ? conv_dma_choice_cold(M, Nout, nk) : conv_dma_choice_warm(M, Nout, nk);
}

static ConvPlan conv_plan(const ConvParams &p, bool dgrad) {
  // Layers whose taps are whole 64-channel K-steps and 16-byte addressable take a pipelined kernel; narrow outputs
  // (stage1 / RPN heads) and the packed stem stay on conv_igemm_kernel.
  ConvPlan q = {0, 0, 0, 0, 0u, 0u};
  const unsigned long x_bytes = ((unsigned long)p.N * p.H * p.W - 1) * p.in_ps * 2 + (unsigned long)p.Cin * 2;
  const unsigned long w_bytes = (unsigned long)p.Nout * p.KH * p.KW * p.Cin * 2;
  if (p.Nout > 64 && p.Cin % 64 == 0 && p.in_ps % 8 == 0 && x_bytes <= 0xFFFFFF00ul && w_bytes <= 0xFFFFFF00ul &&
      !getenv("SNIPER_CONV_V1")) {
    q.x_bytes = (unsigned)x_bytes;
    q.w_bytes = (unsigned)w_bytes;
    const int cfg = g_conv_cfg >= 0 ? g_conv_cfg : conv_dma_choice(p.M, p.Nout, p.KH * p.KW * (p.Cin / 64), dgrad);
    if (cfg > 0) {
      const ConvDmaConfig c = conv_dma_config(cfg);
      q.dma = cfg;
      q.bm = c.bm;
      q.ntiles = sn_div_up(p.Nout, c.bn);
      q.mtiles = sn_div_up(p.M, c.bm);
      return q;
    }
    q.ntiles = sn_div_up(p.Nout, 128);
    const char *force = getenv("SNIPER_CONV_BM");
    const bool small = force ? atoi(force) == 64 : sn_div_up(p.M, 128) * q.ntiles < 448;   // < ~1.75 workgroups per CU
    q.bm = small ? 64 : 128;
    q.mtiles = sn_div_up(p.M, q.bm);
  }
  return q;
}

template <bool DGRAD>
static int conv_launch(const ConvParams &p, hipStream_t s) {
  const ConvPlan pl = conv_plan(p, DGRAD);
  if (pl.dma) {
    ConvParams q = p;
    q.x_bytes = pl.x_bytes;
    q.w_bytes = pl.w_bytes;
    return conv_dma_launch(q, DGRAD, pl.dma, s);
  }
  if (pl.bm) {
    ConvParams q = p;
    q.x_bytes = pl.x_bytes;
    q.w_bytes = pl.w_bytes;
    const dim3 grid(sn_div_up(pl.mtiles, 8) * 8 * pl.ntiles);
    if (pl.bm == 64) hipLaunchKernelGGL((conv_igemm_p2_kernel<DGRAD, 64>), grid, dim3(256), 0, s, q, pl.mtiles, pl.ntiles);
    else hipLaunchKernelGGL((conv_igemm_p2_kernel<DGRAD, 128>), grid, dim3(256), 0, s, q, pl.mtiles, pl.ntiles);
  } else {
    SN_REQUIRE(!p.stats, "convolution statistics are emitted by the pipelined kernel only (query sn_conv_fwd_stats_blocks first)");
    if (p.Nout <= 64) {
      dim3 grid(sn_div_up(p.M, 128), sn_div_up(p.Nout, 64));
      hipLaunchKernelGGL((conv_igemm_kernel<128, 64, DGRAD>), grid, dim3(256), 0, s, p);
    } else {
      dim3 grid(sn_div_up(p.M, 128), sn_div_up(p.Nout, 128));
      hipLaunchKernelGGL((conv_igemm_kernel<128, 128, DGRAD>), grid, dim3(256), 0, s, p);
    }
  }
  SN_CHECK_LAUNCH();
  return SN_OK;
}

static void conv_fwd_params(ConvParams &p, const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H,
                            int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW,
                            int stride, int pad, int dil, int relu, int out_f32) {
  p.x = (const half_t *)x; p.w = (const half_t *)w; p.y = y; p.bias = bias; p.res = (const half_t *)residual;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.in_ps = in_pix_stride;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  p.Nout = Cout; p.out_ps = out_pix_stride; p.res_ps = res_pix_stride;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = N * p.Ho * p.Wo; p.relu = relu; p.out_f32 = out_f32;
}

SN_EXPORT int sn_conv_fwd(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H,
                          int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH,
                          int KW, int stride, int pad, int dil, int relu, int out_f32, sn_stream_t stream) {
  ConvParams p;
  conv_fwd_params(p, x, w, bias, residual, y, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH, KW, stride, pad,
                  dil, relu, out_f32);
  if (int rc = conv_check(p, "sn_conv_fwd")) return rc;
  return conv_launch<false>(p, sn_stream(stream));
}

// Forward convolution that also emits the BatchNorm statistics of its (fp16) output: partials (blocks, 2, Cout) fp32 with
// blocks = sn_conv_fwd_stats_blocks(...) row tiles, consumed by sn_bn_finalize_blocks -- the separate read pass of
// sn_bn_stats over the tensor disappears.  sn_conv_fwd_stats_blocks returns 0 when the layer does not qualify (narrow or
// unaligned layers, fp32 output): use sn_conv_fwd + sn_bn_stats then.
SN_EXPORT int sn_conv_fwd_stats_blocks(int N, int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride,
                                       int res_pix_stride, int KH, int KW, int stride, int pad, int dil) {
  ConvParams p;
  conv_fwd_params(p, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH,
                  KW, stride, pad, dil, 0, 0);
  if (p.Ho <= 0 || p.Wo <= 0 || Cout % 4 != 0 || out_pix_stride % 4 != 0 || (res_pix_stride % 4) != 0) return 0;
  const ConvPlan pl = conv_plan(p, false);
  return pl.bm ? pl.mtiles : 0;
}

SN_EXPORT int sn_conv_fwd_stats(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W,
                                int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW,
                                int stride, int pad, int dil, int relu, float *stats, sn_stream_t stream) {
  ConvParams p;
  conv_fwd_params(p, x, w, bias, residual, y, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH, KW, stride, pad,
                  dil, relu, 0);
  if (int rc = conv_check(p, "sn_conv_fwd_stats")) return rc;
  SN_REQUIRE(stats && Cout % 4 == 0 && out_pix_stride % 4 == 0 && (!residual || res_pix_stride % 4 == 0),
             "sn_conv_fwd_stats: statistics need 8-byte aligned output rows (query sn_conv_fwd_stats_blocks)");
  p.stats = stats;
  return conv_launch<false>(p, sn_stream(stream));
}

// Stem convolution (conv0, resnet_mx_101_e2e.py:403) on the packed input of sn_pack_stem_input:
// xp is (N, Hp, Wp, 4) fp16, already zero padded; every kernel row kh is one contiguous run of
// KWP pixels x 4 channels, so the 7x7/2 conv is a (KH taps) x (4*KWP) contraction on the generic
// kernel with explicit output geometry: y[n,oy,ox,:] = sum_kh  xp[n, oy*s+kh, ox*s .. ox*s+KWP) . w[:,kh,:]
// w is [Cout][KH][KWP*4] fp16 (columns beyond the real kernel width / channel 3 are zero).
SN_EXPORT int sn_conv_stem_fwd(const void *xp, const void *w, const float *bias, void *y, int N, int Hp, int Wp, int Ho, int Wo,
                               int Cout, int out_pix_stride, int KH, int KWP, int stride, int relu, int out_f32,
                               sn_stream_t stream) {
  ConvParams p;
  p.x = (const half_t *)xp; p.w = (const half_t *)w; p.y = y; p.bias = bias; p.res = nullptr;
  p.N = N; p.H = Hp; p.W = Wp; p.Cin = 4 * KWP; p.in_ps = 4;
  p.Ho = Ho; p.Wo = Wo; p.Nout = Cout; p.out_ps = out_pix_stride; p.res_ps = 0;
  p.KH = KH; p.KW = 1; p.stride = stride; p.pad = 0; p.dil = 1;
  p.M = N * Ho * Wo; p.relu = relu; p.out_f32 = out_f32;
  SN_REQUIRE((Ho - 1) * stride + KH <= Hp && (Wo - 1) * stride + KWP <= Wp, "sn_conv_stem_fwd: padded input too small");
  SN_REQUIRE(stride % 2 == 0 && Wp % 2 == 0 && KWP % 2 == 0, "sn_conv_stem_fwd: 16-byte alignment needs even stride/pitch");
  if (int rc = conv_check(p, "sn_conv_stem_fwd")) return rc;
  return conv_launch<false>(p, sn_stream(stream));
}

// Data gradient: dX (N,H,W,Cin) from dY (N,Ho,Wo,Cout) and Wt = weights as [Cin][KH*KW][Cout].
// `accumulate` (fp16 tensor with dX's geometry, may alias dx) is added in the epilogue: that is how a
// tensor with several consumers sums its gradients without a separate pass.
static void conv_dgrad_params(ConvParams &p, const void *dy, const void *wt, const void *accumulate, void *dx, int N, int H, int W,
                              int Cin, int dx_pix_stride, int Cout, int dy_pix_stride, int acc_pix_stride, int KH, int KW, int stride,
                              int pad, int dil, int out_f32) {
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  p.x = (const half_t *)dy; p.w = (const half_t *)wt; p.y = dx; p.bias = nullptr; p.res = (const half_t *)accumulate;
  p.N = N; p.H = Ho; p.W = Wo; p.Cin = Cout; p.in_ps = dy_pix_stride;
  p.Ho = H; p.Wo = W; p.Nout = Cin; p.out_ps = dx_pix_stride; p.res_ps = acc_pix_stride;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = N * H * W; p.relu = 0; p.out_f32 = out_f32;
}

SN_EXPORT int sn_conv_dgrad(const void *dy, const void *wt, const void *accumulate, void *dx, int N, int H, int W,
                            int Cin, int dx_pix_stride, int Cout, int dy_pix_stride, int acc_pix_stride, int KH, int KW,
                            int stride, int pad, int dil, int out_f32, sn_stream_t stream) {
  ConvParams p;
  conv_dgrad_params(p, dy, wt, accumulate, dx, N, H, W, Cin, dx_pix_stride, Cout, dy_pix_stride, acc_pix_stride, KH, KW, stride, pad,
                    dil, out_f32);
  if (int rc = conv_check(p, "sn_conv_dgrad")) return rc;
  return conv_launch<true>(p, sn_stream(stream));
}

// Data gradient that also emits the reduction of the BatchNorm(+activation) backward below it (dx is dL/dy of
// y = act(BN(bn_x))): partials (blocks, 2, Cin) = sum g, sum g * (bn_x - mean) per row tile, consumed by
// sn_bn_backward_blocks -- the bn_bwd_reduce pass over (dy, x) disappears.  Valid only when dx is the COMPLETE gradient
// of y (this convolution is y's only consumer).  sn_conv_dgrad_bn_blocks = 0: the layer does not qualify.
SN_EXPORT int sn_conv_dgrad_bn_blocks(int N, int H, int W, int Cin, int dx_pix_stride, int Cout, int dy_pix_stride,
                                      int acc_pix_stride, int KH, int KW, int stride, int pad, int dil) {
  ConvParams p;
  conv_dgrad_params(p, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, dx_pix_stride, Cout, dy_pix_stride, acc_pix_stride, KH, KW,
                    stride, pad, dil, 0);
  if (p.H <= 0 || p.W <= 0 || Cin % 4 != 0 || dx_pix_stride % 4 != 0 || acc_pix_stride % 4 != 0) return 0;
  const ConvPlan pl = conv_plan(p, true);
  return pl.bm ? pl.mtiles : 0;
}

SN_EXPORT int sn_conv_dgrad_bn(const void *dy, const void *wt, const void *accumulate, void *dx, int N, int H, int W, int Cin,
                               int dx_pix_stride, int Cout, int dy_pix_stride, int acc_pix_stride, int KH, int KW, int stride, int pad,
                               int dil, const void *bn_x, int bn_x_pix_stride, const float *bn_scale, const float *bn_shift,
                               const float *bn_mean, int bn_act, float *partials, sn_stream_t stream) {
  ConvParams p;
  conv_dgrad_params(p, dy, wt, accumulate, dx, N, H, W, Cin, dx_pix_stride, Cout, dy_pix_stride, acc_pix_stride, KH, KW, stride, pad,
                    dil, 0);
  if (int rc = conv_check(p, "sn_conv_dgrad_bn")) return rc;
  SN_REQUIRE(bn_x && bn_scale && bn_shift && bn_mean && partials && Cin % 4 == 0 && dx_pix_stride % 4 == 0 &&
                 bn_x_pix_stride % 4 == 0 && (!accumulate || acc_pix_stride % 4 == 0),
             "sn_conv_dgrad_bn: bad arguments (query sn_conv_dgrad_bn_blocks)");
  p.bn_x = (const half_t *)bn_x; p.bn_x_ps = bn_x_pix_stride; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.bn_mean = bn_mean;
  p.bn_act = bn_act; p.stats = partials;
  return conv_launch<true>(p, sn_stream(stream));
}

// ============================================================================================
// Weight gradient:  dW[co][tap][ci] += sum_{pixels} dY[pix][co] * X[src(pix, tap)][ci]   (fp32)
//
// GEMM with M' = Cout, N' = Cin, K' = pixels: the contraction index is the slow dimension of both
// operands, so each MFMA fragment (8 consecutive k for one row/column) is a strided gather.
// Version 1 (conv_wgrad_kernel, kept for operands that are not 16-byte addressable) gathers the fragments straight
// from global memory with 2-byte loads (every 16-lane group reads 32 contiguous bytes, the lines stay in L1/L2
// for the next 7 k), no LDS, no barriers: each wave owns a 64x64 (co x ci) tile.  Version 2
// (conv_wgrad_tr_kernel, below) stages natural-layout tiles in LDS and transposes with ds_read_b64_tr_b16.
// K' is split over (tap, row-range) blocks that accumulate with fp32 atomics into the zeroed dW.
// Pixels are walked row by row (img, oy) in chunks of 32 consecutive ox so that the source row and
// validity are scalar per step; 1x1/stride-1 layers and FCs pass the whole tensor as one long row.
// ============================================================================================

__device__ __forceinline__ half8 gather8(const half_t *base, int stride_elems, unsigned valid_mask) {
  half8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = ((valid_mask >> j) & 1u) ? base[(size_t)j * stride_elems] : (half_t)0;
  return v;
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int MI = 4, NI = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  int bx, by, bz;
  if (!wgrad_block(p, bx, by, bz)) return;
  const int co0 = bx * 128 + (wave >> 1) * 64, ci0 = by * 128 + (wave & 1) * 64;
  const int taps = p.KH * p.KW;
  const int tap = bz % taps, split = bz / taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  if (co0 >= p.Cout || ci0 >= p.Cin) return;  // wave-uniform
  const int cpr = (p.Wo + 31) / 32;  // 32-pixel chunks per (img, oy) row
  const int nunits = p.N * p.Ho * cpr;
  const int u_begin = split * p.units_per_split, u_end = min(nunits, u_begin + p.units_per_split);

  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  for (int u = u_begin; u < u_end; ++u) {
    const int r = u / cpr, ox0 = (u - r * cpr) * 32;
    const int img = r / p.Ho, oy = r - img * p.Ho;
    const int sy = oy * p.stride - p.pad + kh * p.dil;
    if ((unsigned)sy >= (unsigned)p.H) continue;  // whole source row is padding
    // this lane's 8 consecutive output pixels ox = ox0 + fq*8 + j
    const int oxb = ox0 + fq * 8;
    unsigned vdy = 0, vx = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ox = oxb + j, sx = ox * p.stride - p.pad + kw * p.dil;
      if (ox < p.Wo) {
        vdy |= 1u << j;
        if ((unsigned)sx < (unsigned)p.W) vx |= 1u << j;
      }
    }
    const half_t *dyp = p.dy + ((size_t)r * p.Wo + oxb) * p.dy_ps;
    const int sxb = oxb * p.stride - p.pad + kw * p.dil;
    const half_t *xp = p.x + (((size_t)img * p.H + sy) * p.W + sxb) * (size_t)p.x_ps;  // may point before the row; masked
    half8 fa[MI], fb[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int co = co0 + i * 16 + fr;
      fa[i] = gather8(dyp + co, p.dy_ps, co < p.Cout ? vdy : 0u);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ci = ci0 + j * 16 + fr;
      fb[j] = gather8(xp + ci, p.x_ps * p.stride, ci < p.Cin ? vx : 0u);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  }
  // D: row = co (A operand), col = ci (B operand)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int co = co0 + i * 16 + fq * 4 + rr;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ci = ci0 + j * 16 + fr;
        if (ci < p.Cin) {
          // one owner per element: plain store into this split's slab, or read-modify-write of dw when unsplit
          const size_t e = ((size_t)co * taps + tap) * p.Cin + ci;
          if (p.slab) p.slab[(size_t)split * p.slab_stride + e] = acc[i][j][rr];
          else p.dw[e] += acc[i][j][rr];
        }
      }
    }
}

// ---- version 2: natural-layout LDS tiles + ds_read_b64_tr_b16 ---------------------------------
// Both operands are staged exactly as they lie in HBM -- [k = pixel][m = channel], 16-byte channel runs, fully
// coalesced 256-byte rows -- and the transpose happens in the LDS read: ds_read_b64_tr_b16 hands lane (c, g) of a
// 16-lane group the column c of the 4x16 block whose rows the group's lanes point at (4 lanes x 8 B per row;
// semantics pinned on the hardware by tools/probes/tr16_probe.hip).  Two reads (rows g*4.., 16+g*4..) make one
// 8-deep MFMA fragment; the k order inside a fragment is therefore permuted (k = {4g..4g+3, 16+4g..16+4g+3}) but
// identically for A and B, and a contraction does not care.
// LDS image: 32 rows (pixels) x 256 B (128 channels); the 32-byte segment index is XORed with (row & 7) so that the
// 8 rows a 32-lane service group touches fall on 8 distinct bank octets (reads) and the 8 lanes of a
// ds_write_b128 group on 8 distinct 16-byte slots (writes).
// Tile 128 (co) x 128 (ci) per 256-thread workgroup, waves 2x2.
typedef short short4v __attribute__((vector_size(8)));

__device__ __forceinline__ half8 tr_frag(const half_t *lds_tile, int off) {
  typedef __attribute__((address_space(3))) short4v *lds_v4;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_tile + off));
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds_tile + off + 16 * 128));
  union { short4v s[2]; half8 h; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.h;
}

// K-step = 64 pixels = two 32-pixel units (rows 0-31 / 32-63 of the LDS tile); two register sets in flight ahead of
// the tile being multiplied, one barrier per 32 MFMAs (same pipeline as conv_igemm_p2_kernel).  Units are walked
// with an incremental (img, oy, x-chunk) iterator -- no integer division in the loop -- that skips units whose
// source row is padding; all of its state is workgroup-uniform.
__global__ __launch_bounds__(256, 2) void conv_wgrad_tr_kernel(const WgradParams p) {
  constexpr int MI = 4, NI = 4, TILE = 64 * 128;
  __shared__ __attribute__((aligned(16))) half_t sA[2][TILE];
  __shared__ __attribute__((aligned(16))) half_t sB[2][TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  int bx, by, bz;
  if (!wgrad_block(p, bx, by, bz)) return;
  const int co0 = bx * 128, ci0 = by * 128;
  const int taps = p.KH * p.KW;
  const int tap = bz % taps, split = bz / taps;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int cpr = (p.Wo + 31) / 32;  // 32-pixel units per (img, oy) row
  const int nunits = p.N * p.Ho * cpr;
  const int u_begin = split * p.units_per_split, u_end = min(nunits, u_begin + p.units_per_split);
  // phase stamps of thread 0 (shader clock): [0] entry, [1] first tile in LDS, [2] contraction done, [3] stores drained
  auto stamp = [&](int k) {
    if (p.trace && tid == 0)
      p.trace[((size_t)(bz * p.gy + by) * p.gx + bx) * 8 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

  // loader: rows lr, lr+16 of each unit; 16-byte chunk `chunk` of the 128 channels
  const int lr = tid >> 4, chunk = tid & 15;
  const int a_c = co0 + chunk * 8, b_c = ci0 + chunk * 8;
  const bool a_cok = a_c < p.Cout, b_cok = b_c < p.Cin;
  // rows lr + 16*i (i = 0..3): (row & 7) alternates between lr & 7 and (lr + 16) & 7 = lr & 7 -> one swizzle for all
  const int st_off = lr * 128 + ((((chunk >> 1) ^ (lr & 7)) << 4) | ((chunk & 1) << 3));
  // fragment reads: lane (fr, fq) points at row fq*4 + fr/4, 8-byte piece fr%4 of the fragment's 32-byte segment
  const int row0 = fq * 4 + (fr >> 2), r7 = row0 & 7;
  int a_off[MI], b_off[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_off[i] = row0 * 128 + ((((wm * 4 + i) ^ r7) << 4) | ((fr & 3) << 2));
#pragma unroll
  for (int j = 0; j < NI; ++j) b_off[j] = row0 * 128 + ((((wn * 4 + j) ^ r7) << 4) | ((fr & 3) << 2));

  // unit iterator (uniform)
  int it_left = u_end - u_begin;
  int it_r = u_begin / cpr, it_xc = u_begin - it_r * cpr;
  int it_img = it_r / p.Ho, it_oy = it_r - it_img * p.Ho;
  auto it_advance = [&]() {
    --it_left;
    if (++it_xc == cpr) {
      it_xc = 0;
      ++it_r;
      if (++it_oy == p.Ho) { it_oy = 0; ++it_img; }
    }
  };
  const char *dyb = reinterpret_cast<const char *>(p.dy), *xbp = reinterpret_cast<const char *>(p.x);
  const unsigned dy_ps_b = (unsigned)p.dy_ps * 2u, x_ps_b = (unsigned)p.x_ps * 2u;
  // Buffer-descriptor loads (out-of-range voffset / exhausted descriptor -> zeros, no branch, no select): the
  // descriptor of a unit covers exactly the rest of its dY row and its source X row, so pixels beyond Wo and source
  // columns outside [0, W) read as zero by construction.
  constexpr unsigned kOob = 0xFFFFFF00u;
  unsigned dy_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) dy_voff[i] = a_cok ? (unsigned)(lr + 16 * i) * dy_ps_b + (unsigned)a_c * 2u : kOob;
  // one register set = one 64-pixel tile: rows {lr, lr+16} of unit 0 and of unit 1, A (dY) and B (X)
  // (the two units of a tile are located first -- scalar loops that skip padding rows -- and the eight loads are then
  // issued back to back, with no control flow between them)
  auto gload = [&](half8 (&ra)[4], half8 (&rb)[4]) -> bool {
    bool have[2];
    int ox0[2];
    size_t dy_base[2], x_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int sy = 0;
      while (it_left > 0) {   // skip units whose whole source row is padding
        sy = it_oy * p.stride - p.pad + kh * p.dil;
        if ((unsigned)sy < (unsigned)p.H) break;
        it_advance();
      }
      // (readfirstlane: these are block-uniform, but the compiler must KNOW it -- a descriptor it believes divergent is
      // loaded inside an exec-masked waterfall, i.e. control flow around every buffer_load)
      have[h] = __builtin_amdgcn_readfirstlane(it_left > 0 ? 1 : 0) != 0;
      ox0[h] = __builtin_amdgcn_readfirstlane(it_xc * 32);
      const unsigned dyrow = (unsigned)__builtin_amdgcn_readfirstlane(it_r * p.Wo + ox0[h]);
      const unsigned xrow = (unsigned)__builtin_amdgcn_readfirstlane((it_img * p.H + sy) * p.W);
      dy_base[h] = have[h] ? (size_t)dyrow * dy_ps_b : 0;
      x_base[h] = have[h] ? (size_t)xrow * x_ps_b : 0;
      if (have[h]) it_advance();
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(dyb) + dy_base[h], 0, have[h] ? (int)((unsigned)(p.Wo - ox0[h]) * dy_ps_b) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char *>(xbp) + x_base[h], 0, have[h] ? (int)((unsigned)p.W * x_ps_b) : 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ox = ox0[h] + lr + 16 * i, sx = ox * p.stride - p.pad + kw * p.dil;
        // bitwise, not short-circuit: `&&` here became an exec-masked if/else with the dY load duplicated into both arms
        const bool okb = (ox < p.Wo) & b_cok & ((unsigned)sx < (unsigned)p.W);
        ra[h * 2 + i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rdy, dy_voff[i], 0, 0));
        rb[h * 2 + i] = __builtin_bit_cast(
            half8, __builtin_amdgcn_raw_buffer_load_b128(rxx, okb ? (unsigned)sx * x_ps_b + (unsigned)b_c * 2u : kOob, 0, 0));
      }
    }
    return have[0] || have[1];
  };
  auto lstore = [&](int buf, const half8 (&ra)[4], const half8 (&rb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // q = h*2 + i -> tile row h*32 + lr + 16*i
      *reinterpret_cast<half8 *>(&sA[buf][st_off + q * 16 * 128]) = ra[q];
      *reinterpret_cast<half8 *>(&sB[buf][st_off + q * 16 * 128]) = rb[q];
    }
  };

  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = tr_frag(sA[buf] + ks * 32 * 128, a_off[i]);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = tr_frag(sB[buf] + ks * 32 * 128, b_off[j]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  };

  // Same pipeline discipline as conv_igemm_p2_kernel: the steady-state loop has NO branch around a load or a store (an
  // exhausted iterator loads through empty descriptors = zeros, which is harmless), so the compiler's s_waitcnt vmcnt
  // model is exact and the stores of one register set wait with the other set's loads still in flight.  Which of the
  // three tiles in flight are real is tracked in block-uniform flags that only the drain looks at.
  half8 ra0[4], rb0[4], ra1[4], rb1[4];
  bool fa = gload(ra0, rb0);               // tile 0 -> LDS[0]
  bool fb = gload(ra1, rb1);               // tile 1 in set 1
  lstore(0, ra0, rb0);
  bool fc = gload(ra0, rb0);               // tile 2 in set 0
  __syncthreads();
  stamp(1);
  while (it_left > 0) {                    // tiles t (LDS[0]), t+1 (set 1), t+2 (set 0) are real, and so is tile t+3
    lstore(1, ra1, rb1);
    const bool fb_next = gload(ra1, rb1);  // tile t+3
    compute(0);
    __syncthreads();
    lstore(0, ra0, rb0);
    const bool fc_next = gload(ra0, rb0);  // tile t+4 (may be empty)
    compute(1);
    __syncthreads();
    fa = fc;
    fb = fb_next;
    fc = fc_next;
  }
  // drain: LDS[0] = tile A (fa), set 1 = tile B (fb), set 0 = tile C (fc)
  if (fb) lstore(1, ra1, rb1);
  if (fa) compute(0);
  __syncthreads();
  if (fc) lstore(0, ra0, rb0);
  if (fb) compute(1);
  __syncthreads();
  if (fc) compute(0);
  stamp(2);
  // The product was formed transposed (X as the MFMA A operand): lane (fr, fq) holds, for each (i, j), output channel
  // co = ..+fr and 4 consecutive input channels ci = ..+fq*4 .. +3 -> one 16-byte store per (i, j).
  // With K-splits the partial tile goes to this split's slab with plain stores (a split without any valid unit
  // stores zeros) and wgrad_reduce_kernel sums the slabs into dw: no atomics, fixed summation order.
  float *dst = p.slab ? p.slab + (size_t)split * p.slab_stride : p.dw;
  const bool vec4 = (p.Cin % 4) == 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int co = co0 + wm * 64 + i * 16 + fr;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ci = ci0 + wn * 64 + j * 16 + fq * 4;
      if (ci >= p.Cin) continue;
      float *q = dst + ((size_t)co * taps + tap) * p.Cin + ci;
      if (p.slab) {
        if (vec4) *reinterpret_cast<float4 *>(q) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ci + r < p.Cin) q[r] = acc[i][j][r];
      } else {   // unsplit: this workgroup is the only writer of its tile
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ci + r < p.Cin) q[r] += acc[i][j][r];
      }
    }
  }
  if (p.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
  }
}

// dw[e] += sum over splits of slab[s][e]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slab, int splits, size_t n, float *__restrict__ dw) {
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<const float4 *>(dw)[i];
    for (int sidx = 0; sidx < splits; ++sidx) {
      const float4 v = reinterpret_cast<const float4 *>(slab + (size_t)sidx * n)[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4 *>(dw)[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t e = (n4 << 2) + threadIdx.x;
    float a = dw[e];
    for (int sidx = 0; sidx < splits; ++sidx) a += slab[(size_t)sidx * n + e];
    dw[e] = a;
  }
}

// K-split plan shared by the workspace query and the launch.  kind: 0 = conv_wgrad_tr_kernel (taps in the grid) or, for
// operands that are not 16-byte addressable, conv_wgrad_kernel; 1 / 2 = the LDS-DMA kernels of conv_wgrad_dma.hip.
struct WgradPlan {
  int gx, gy, taps, nunits, splits, units_per_split, Ho, Wo, flat, kind, stages;
  bool vec_ok;
};
// Tuning override (tools/wgrad_tune.py, tests): stages of the flat / all-taps LDS-DMA kernels; -1 = built-in choice,
// 0 = that kernel off (register-staged kernel instead).  Process-wide.
static int g_wgrad_flat = env_int("SNIPER_WGRAD_FLAT", -1), g_wgrad_taps = env_int("SNIPER_WGRAD_TAPS", -1);
static int g_wgrad_wgs = env_int("SNIPER_WGRAD_WGS", 0);   // K-split target (workgroups per launch); 0 = built-in
int g_wgrad_xcd = env_int("SNIPER_WGRAD_XCD", 1);          // 0 = dispatch-order block mapping (A/B against the XCD-aware one)
SN_EXPORT int sn_conv_wgrad_tune(int flat_stages, int taps_stages, int target_workgroups) {
  SN_REQUIRE(target_workgroups >= 0 && target_workgroups <= 4096, "sn_conv_wgrad_tune: bad workgroup target");
  g_wgrad_wgs = target_workgroups;
  SN_REQUIRE((flat_stages == -1 || flat_stages == 0 || flat_stages == 2 || flat_stages == 3) &&
                 (taps_stages == -1 || taps_stages == 0 || taps_stages == 3 || taps_stages == 4),
             "sn_conv_wgrad_tune: flat stages in {-1, 0, 2, 3}, taps stages in {-1, 0, 3, 4}");
  g_wgrad_flat = flat_stages;
  g_wgrad_taps = taps_stages;
  return SN_OK;
}

static WgradPlan wgrad_plan(const void *dy, const void *x, int N, int H, int W, int Cin, int x_ps, int Cout, int dy_ps, int KH,
                            int KW, int stride, int pad, int dil) {
  WgradPlan q;
  q.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  q.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  q.flat = KH == 1 && KW == 1 && stride == 1 && pad == 0;   // one long row of pixels
  q.taps = KH * KW;
  q.gx = sn_div_up(Cout, 128);
  q.gy = sn_div_up(Cin, 128);
  q.nunits = q.flat ? sn_div_up(N * H * W, 32) : N * q.Ho * sn_div_up(q.Wo, 32);
  // 16-byte channel runs: pixel strides multiples of 8 that cover the last (possibly partial) chunk, aligned bases
  q.vec_ok = dy_ps % 8 == 0 && x_ps % 8 == 0 && dy_ps >= sn_div_up(Cout, 8) * 8 && x_ps >= sn_div_up(Cin, 8) * 8 &&
             ((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0;
  q.kind = 0;
  q.stages = 0;
  int tiles = q.gx * q.gy * q.taps, min_units = 2;   // workgroups per split; least K per split (one 64-pixel step)
  const int flat_tiles = q.gx * q.gy, taps_tiles = sn_div_up(Cout, 64) * sn_div_up(Cin, 64);
  const int fs = g_wgrad_flat >= 0 ? g_wgrad_flat : (flat_tiles >= 64 ? 2 : 0);      // built-in choice: profiles/r02_wgrad_tune.txt
  const int ts = g_wgrad_taps >= 0 ? g_wgrad_taps : (taps_tiles >= 256 ? 4 : 0);
  if (q.vec_ok && q.flat && fs) {
    q.kind = 1;
    q.stages = fs;
    tiles = flat_tiles;
  } else if (q.vec_ok && KH == 3 && KW == 3 && stride == 1 && dil <= 4 && ts) {
    q.kind = 2;
    q.stages = ts;
    q.gx = sn_div_up(Cout, 64);
    q.gy = sn_div_up(Cin, 64);
    tiles = taps_tiles;
    min_units = 1;
  }
  // K-splits: at most `target` workgroups (2 per CU resident: one more than fits starts a second, nearly empty round --
  // measured +30 % for 540 instead of 504 workgroups); few tiles -> fewer, longer splits (the slabs are the cost there)
  const int target = g_wgrad_wgs > 0 ? g_wgrad_wgs : (tiles <= 16 ? 256 : 512);
  int splits = target / tiles;
  if (splits > q.nunits / min_units) splits = q.nunits / min_units;
  if (splits < 1) splits = 1;
  q.units_per_split = sn_div_up(q.nunits, splits);
  if (q.kind == 1 && (q.units_per_split & 1)) ++q.units_per_split;   // whole 64-pixel K-steps
  q.splits = sn_div_up(q.nunits, q.units_per_split);
  return q;
}

// Scratch for the split-K partials of sn_conv_wgrad (0 when the layer needs no split).  Without it (or with too little)
// the layer runs unsplit -- slower, same result modulo summation order; there is no atomic accumulation anywhere.
static size_t wgrad_legacy_workspace_bytes(int N, int H, int W, int Cin, int x_pix_stride, int Cout, int dy_pix_stride, int KH,
                                           int KW, int stride, int pad, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const WgradPlan q = wgrad_plan(nullptr, nullptr, N, H, W, Cin, x_pix_stride, Cout, dy_pix_stride, KH, KW, stride, pad, dil);
  if (q.splits <= 1) return 0;
  return sn_align(sizeof(float) * (size_t)q.splits * Cout * q.taps * Cin);
}

static int wgrad_legacy(const void *dy, const void *x, float *dw, int N, int H, int W, int Cin, int x_pix_stride,
                        int Cout, int dy_pix_stride, int KH, int KW, int stride, int pad, int dil, void *ws,
                        size_t ws_bytes, hipStream_t s) {
  SN_REQUIRE(dy && x && dw, "sn_conv_wgrad: null pointer");
  WgradParams p;
  p.dy = (const half_t *)dy; p.x = (const half_t *)x; p.dw = dw;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.dy_ps = dy_pix_stride; p.x_ps = x_pix_stride;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  SN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "sn_conv_wgrad: bad dims");
  WgradPlan q = wgrad_plan(dy, x, N, H, W, Cin, x_pix_stride, Cout, dy_pix_stride, KH, KW, stride, pad, dil);
  p.Ho = q.Ho; p.Wo = q.Wo;
  SN_REQUIRE(p.Ho > 0 && p.Wo > 0, "sn_conv_wgrad: bad dims");
  SN_REQUIRE((long)N * H * W * x_pix_stride < (1l << 31) && (long)N * p.Ho * p.Wo * dy_pix_stride < (1l << 31),
             "sn_conv_wgrad: tensor too large for 32-bit offsets");
  if (q.flat) { p.W = p.Wo = N * H * W; p.H = p.Ho = 1; p.N = 1; }
  const size_t n = (size_t)Cout * q.taps * Cin;
  p.slab = nullptr;
  p.slab_stride = n;
  if (q.splits > 1) {
    if (ws && ws_bytes >= sizeof(float) * (size_t)q.splits * n && ((uintptr_t)ws % 16) == 0) p.slab = (float *)ws;
    else { q.splits = 1; q.units_per_split = q.nunits + (q.nunits & 1); }   // no scratch: one owner per element, no split
  }
  p.units_per_split = q.units_per_split;
  static const bool trace_armed = getenv("SNIPER_CONV_TRACE") != nullptr;
  if (trace_armed) {
    const char *e = getenv("SNIPER_CONV_TRACE_PTR");
    if (e && *e) p.trace = reinterpret_cast<unsigned long long *>(strtoull(e, nullptr, 16));
  }
  if (q.kind) {
    if (int rc = wgrad_dma_launch(p, q.kind, q.stages, q.splits, s)) return rc;
  } else {
    const dim3 grid(wgrad_grid(p, q.gx, q.gy, q.taps * q.splits, g_wgrad_xcd != 0));
    if (q.vec_ok) hipLaunchKernelGGL(conv_wgrad_tr_kernel, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, s, p);
    SN_CHECK_LAUNCH();
  }
  if (p.slab) {
    long blocks = (long)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float *)p.slab, q.splits, n, dw);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}

// ---- entry points: one layer, or a table of layers in one launch (conv_wgrad_ps.hip) ----
static int g_wgrad_impl = env_int("SNIPER_WGRAD_IMPL", 1);   // 1 = wave-specialised batched kernel, 0 = the round-1/2 kernels (A/B)
SN_EXPORT int sn_conv_wgrad_impl(int impl, int job_steps) {
  SN_REQUIRE((impl == 0 || impl == 1) && job_steps >= 0, "sn_conv_wgrad_impl: impl in {0, 1}, job_steps >= 0");
  g_wgrad_impl = impl;
  wgrad_ps_set_job_steps(job_steps);
  return SN_OK;
}

// diagnostics: phase timeline of the batched kernel's jobs into buf ([jobs][8] x uint64, see conv_wgrad_ps.hip); NULL = off
SN_EXPORT int sn_conv_wgrad_trace(void *buf) {
  wgrad_ps_set_trace(static_cast<unsigned long long *>(buf));
  return SN_OK;
}

static bool wgrad_desc_params(const sn_wgrad_desc &d, WgradParams &p) {
  p.dy = (const half_t *)d.dy; p.x = (const half_t *)d.x; p.dw = d.dw;
  p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.Cout = d.Cout; p.dy_ps = d.dy_pix_stride; p.x_ps = d.x_pix_stride;
  p.KH = d.KH; p.KW = d.KW; p.stride = d.stride; p.pad = d.pad; p.dil = d.dil;
  if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.KH <= 0 || d.KW <= 0 || d.stride <= 0 || d.dil <= 0) return false;
  p.Ho = (d.H + 2 * d.pad - d.dil * (d.KH - 1) - 1) / d.stride + 1;
  p.Wo = (d.W + 2 * d.pad - d.dil * (d.KW - 1) - 1) / d.stride + 1;
  p.slab = nullptr; p.slab_stride = 0; p.units_per_split = 0;
  return p.Ho > 0 && p.Wo > 0;
}

// launch == false: only the scratch size.  Scratch layout: the split-K slabs of every chunk of <= 24 batched problems, then the
// scratch of each problem that runs on the round-1/2 kernels (operands not 16-byte addressable, or the A/B switch).
static int wgrad_batch_run(const sn_wgrad_desc *descs, int n, void *ws, size_t ws_bytes, hipStream_t s, bool launch, size_t *need) {
  size_t off = 0;
  WgradParams chunk[kWgradMaxProblems];
  int nc = 0;
  auto flush = [&]() -> int {
    if (!nc) return SN_OK;
    WgradBatch tab;
    char *base = ws ? static_cast<char *>(ws) + off : nullptr;
    size_t bytes = wgrad_ps_plan(chunk, nc, tab, base);
    if (launch) {
      // too little scratch (or none): the chunk runs unsplit -- slower, same result modulo summation order, never atomics
      if (bytes && !(ws && off + bytes <= ws_bytes && ((uintptr_t)base % 16) == 0)) bytes = wgrad_ps_plan(chunk, nc, tab, nullptr, false);
      if (int rc = wgrad_ps_launch(tab, s)) return rc;
    }
    off += bytes;
    nc = 0;
    return SN_OK;
  };
  // chunks: <= 24 problems (the table is a kernel argument), cut where the job count fills whole rounds of the 256 CUs -- a chunk
  // of 528 equal jobs runs as long as one of 768; one of 492 runs like 512
  long jobs = 0;
  auto fill = [](long j) { return j <= 0 ? 0.0 : (double)j / (256.0 * (double)((j + 255) / 256)); };
  for (int i = 0; i < n; ++i) {
    const sn_wgrad_desc &d = descs[i];
    WgradParams p;
    SN_REQUIRE(wgrad_desc_params(d, p), "sn_conv_wgrad_batch: bad dims in problem %d", i);
    if (launch) SN_REQUIRE(d.dy && d.x && d.dw, "sn_conv_wgrad_batch: null pointer in problem %d", i);
    if (g_wgrad_impl == 1 && wgrad_ps_ok(p)) {
      const bool flat = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
      const long steps = flat ? sn_div_up(sn_div_up(p.N * p.H * p.W, 32), 2) : sn_div_up(p.N * p.Ho * sn_div_up(p.Wo, 32), 2);
      const long j = (long)sn_div_up(p.Cout, 128) * sn_div_up(p.Cin, 128) * p.KH * p.KW * ((steps + 399) / 400);
      if (nc && jobs >= 200 && fill(jobs) >= 0.93 && fill(jobs + j) < fill(jobs) - 0.02) {
        if (int rc = flush()) return rc;
        jobs = 0;
      }
      chunk[nc++] = p;
      jobs += j;
      if (nc == kWgradMaxProblems) {
        if (int rc = flush()) return rc;
        jobs = 0;
      }
    } else {
      const size_t bytes = wgrad_legacy_workspace_bytes(d.N, d.H, d.W, d.Cin, d.x_pix_stride, d.Cout, d.dy_pix_stride, d.KH, d.KW, d.stride, d.pad, d.dil);
      if (launch) {
        char *base = ws ? static_cast<char *>(ws) + off : nullptr;
        const size_t have = ws && off + bytes <= ws_bytes ? bytes : 0;
        if (int rc = wgrad_legacy(d.dy, d.x, d.dw, d.N, d.H, d.W, d.Cin, d.x_pix_stride, d.Cout, d.dy_pix_stride, d.KH, d.KW, d.stride, d.pad,
                                  d.dil, have ? base : nullptr, have, s))
          return rc;
      }
      off += bytes;
    }
  }
  if (int rc = flush()) return rc;
  if (need) *need = off;
  return SN_OK;
}

SN_EXPORT size_t sn_conv_wgrad_batch_workspace_bytes(const sn_wgrad_desc *descs, int n) {
  size_t need = 0;
  if (!descs || n <= 0 || wgrad_batch_run(descs, n, nullptr, 0, nullptr, false, &need)) return 0;
  return need;
}

SN_EXPORT int sn_conv_wgrad_batch(const sn_wgrad_desc *descs, int n, void *ws, size_t ws_bytes, sn_stream_t stream) {
  SN_REQUIRE(descs && n > 0, "sn_conv_wgrad_batch: empty table");
  return wgrad_batch_run(descs, n, ws, ws_bytes, sn_stream(stream), true, nullptr);
}

SN_EXPORT size_t sn_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int x_pix_stride, int Cout, int dy_pix_stride, int KH,
                                               int KW, int stride, int pad, int dil) {
  const sn_wgrad_desc d = {nullptr, nullptr, nullptr, N, H, W, Cin, x_pix_stride, Cout, dy_pix_stride, KH, KW, stride, pad, dil};
  return sn_conv_wgrad_batch_workspace_bytes(&d, 1);
}

SN_EXPORT int sn_conv_wgrad(const void *dy, const void *x, float *dw, int N, int H, int W, int Cin, int x_pix_stride,
                            int Cout, int dy_pix_stride, int KH, int KW, int stride, int pad, int dil, void *ws,
                            size_t ws_bytes, sn_stream_t stream) {
  const sn_wgrad_desc d = {dy, x, dw, N, H, W, Cin, x_pix_stride, Cout, dy_pix_stride, KH, KW, stride, pad, dil};
  return sn_conv_wgrad_batch(&d, 1, ws, ws_bytes, stream);
}

// Weight gradient of the stem convolution on the packed input of sn_pack_stem_input (MobileNetV2's first 3x3/2 conv
// is trainable: `conv1` in its FIXED_PARAMS matches no parameter name of that network).  dw is the packed weight
// [Cout][KH][KWP*4] fp32 (accumulated into); geometry as sn_conv_stem_fwd.  The packed rows are only 8-byte aligned,
// so this runs on the gather kernel (version 1).
static int stem_wgrad_splits(int Ho, int Wo, int N, int Cout, int KH, int KWP, int *units_per_split) {
  const int gx = sn_div_up(Cout, 128), gy = sn_div_up(4 * KWP, 128);
  const int nunits = N * Ho * sn_div_up(Wo, 32);
  int splits = sn_div_up(1024, gx * gy * KH);
  if (splits > nunits) splits = nunits;
  if (splits < 1) splits = 1;
  *units_per_split = sn_div_up(nunits, splits);
  return sn_div_up(nunits, *units_per_split);
}

SN_EXPORT size_t sn_conv_stem_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cout, int KH, int KWP) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || Cout <= 0 || KH <= 0 || KWP <= 0) return 0;
  int ups;
  const int splits = stem_wgrad_splits(Ho, Wo, N, Cout, KH, KWP, &ups);
  return splits > 1 ? sn_align(sizeof(float) * (size_t)splits * Cout * KH * KWP * 4) : 0;
}

SN_EXPORT int sn_conv_stem_wgrad(const void *dy, const void *xp, float *dw, int N, int Hp, int Wp, int Ho, int Wo, int Cout,
                                 int dy_pix_stride, int KH, int KWP, int stride, void *ws, size_t ws_bytes, sn_stream_t stream) {
  SN_REQUIRE(dy && xp && dw && N > 0 && Ho > 0 && Wo > 0 && Cout > 0, "sn_conv_stem_wgrad: bad arguments");
  SN_REQUIRE((Ho - 1) * stride + KH <= Hp && (Wo - 1) * stride + KWP <= Wp, "sn_conv_stem_wgrad: padded input too small");
  WgradParams p;
  p.dy = (const half_t *)dy; p.x = (const half_t *)xp; p.dw = dw;
  p.N = N; p.H = Hp; p.W = Wp; p.Ho = Ho; p.Wo = Wo; p.Cin = 4 * KWP; p.Cout = Cout; p.dy_ps = dy_pix_stride; p.x_ps = 4;
  p.KH = KH; p.KW = 1; p.stride = stride; p.pad = 0; p.dil = 1;
  const size_t n = (size_t)Cout * KH * p.Cin;
  p.slab = nullptr; p.slab_stride = n;
  const int gx = sn_div_up(Cout, 128), gy = sn_div_up(p.Cin, 128);
  int splits = stem_wgrad_splits(Ho, Wo, N, Cout, KH, KWP, &p.units_per_split);
  if (splits > 1) {
    if (ws && ws_bytes >= sizeof(float) * (size_t)splits * n && ((uintptr_t)ws % 16) == 0) p.slab = (float *)ws;
    else { splits = 1; p.units_per_split = N * Ho * sn_div_up(Wo, 32); }   // no scratch: unsplit, one owner per element
  }
  hipStream_t s = sn_stream(stream);
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(wgrad_grid(p, gx, gy, KH * splits)), dim3(256), 0, s, p);
  SN_CHECK_LAUNCH();
  if (p.slab) {
    long blocks = (long)((n / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float *)p.slab, splits, n, dw);
    SN_CHECK_LAUNCH();
  }
  return SN_OK;
}
