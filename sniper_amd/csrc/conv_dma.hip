// conv_dma.hip -- implicit-GEMM convolution (forward and data gradient) whose operand tiles travel HBM/L2 -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds) instead of through registers.
//
// Same GEMM view, data layout and epilogue as conv_igemm_p2_kernel (conv.hip; call sites
// symbols/faster/resnet_mx_101_e2e.py:43-66,121-155,256,288-303): Y[m][n] = sum_{tap,c} A(m,tap,c) * Wt[n][tap][c],
// channels-last fp16, BK = 64 channels per K-step, 128-byte LDS rows whose 16-byte slots are XOR-swizzled with (row & 7).
// What changes is the staging pipeline, which is what bounded the register-staged kernel (8
// ds_write_b128 per thread per K-step on the LDS store path ~ the time of the step's MFMAs, 64 staging VGPRs):
//
//   * one LDS-DMA instruction moves 8 tile rows x 128 B: lane l supplies the global address of row (l >> 3), 16-byte chunk
//     (l & 7) ^ (l >> 3), and the hardware writes lane l's 16 bytes at (wave-uniform base) + 16 l -- i.e. the swizzle
//     is applied on the SOURCE side and the LDS image is the one the fragment reads expect.  Out-of-range voffsets
//     (padding taps, rows beyond M / Nout) deliver zeros, as with the register loads.
//   * no ds_write, no staging registers, no VALU on the load path except the per-tap voffset update;
//   * an S-deep ring of stages: in iteration t the stage t+S-1 is issued right after the barrier that retires stage t, so
//     S-1 stages are in flight under every compute phase, one barrier per K-step, counted s_waitcnt vmcnt (never 0 in
//     the steady state for S > 2).  The whole LDS footprint is ONE __shared__ array: with two, hipcc orders every
//     LDS-DMA against every later ds_read with vmcnt(0) (cdna_hip_programming.md section 5, trap (a)), which is what made
//     round 1's attempt a no-gain.
//
// The tile shape is a template parameter set (BM x BN outputs, WMW x WNW waves, S stages); conv_plan() in conv.hip picks
// one per layer from the measured table (tools/conv_tune.py).  Nine configurations are instantiated (kCfg below).
#include "conv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void *lds_ptr_t;

// 64 lanes x 16 B: lane l's bytes land at dst + 16 l (dst wave-uniform).  Kept out of the kernel template: the host pass
// rejects the address-space cast, and an error inside a __global__ template silently drops its host stub.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t *dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}

// sum over the 16 lanes of a DPP row (the lanes that share lane >> 4), result in every lane: four VALU adds with DPP operands
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) instead of four ds_bpermute round trips per value
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of its half row (the other quad)
  return dpp_add<0x140>(v);   // row_mirror: lane i <-> 15 - i (the other half row)
}

// 16-byte global load the compiler's wait-count pass does not see (PERSIST: the next tile's residual rows are requested behind one
// tile's epilogue and consumed in the next one's; tracked loads in flight across the tile loop's back edge make hipcc drain the
// whole queue -- vmcnt(0) -- wherever it is unsure, and loads return in order).  The consumer waits by hand: wait_vmcnt + tie().
__device__ __forceinline__ half8 load16_untracked(const half_t *ptr) {
  floatx4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return __builtin_bit_cast(half8, v);
}
// 16-byte LDS read the wait-count pass does not see either: a tracked ds_read of an LDS range an LDS-DMA wrote earlier gets a vmcnt
// wait in front of it (the pass cannot know that DMA was waited for by hand), which in the epilogue means waiting for the stores
__device__ __forceinline__ floatx4 lds_read16_untracked(const void *ptr) {
  floatx4 v;
  const unsigned a = (unsigned)(unsigned long)(lds_ptr_t)const_cast<void *>(ptr);
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void tie(floatx4 &v) { asm volatile("" : "+v"(v)); }
// orders later uses of x behind the asm statements in front of this one (an s_waitcnt): x is "redefined" here
__device__ __forceinline__ void tie(half8 &x) {
  floatx4 v = __builtin_bit_cast(floatx4, x);
  asm volatile("" : "+v"(v));
  x = __builtin_bit_cast(half8, v);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PS ("producer / consumer specialised", round 3): the workgroup has WMW x WNW CONSUMER waves (one per SIMD for 2 x 2) that only read
// fragments and multiply, plus four PRODUCER waves (the second wave of each SIMD) that only compute gather addresses and issue the
// LDS-DMA pieces, S - 1 stages ahead.  A wave's instruction stream is in order, so in the unspecialised kernel a K-step costs
// DMA issue (~70 cycles per piece) PLUS the MFMAs (tools/probes/dma_rate_probe.hip); with the two jobs in different waves a step
// costs the longer of the two (the same split made the weight gradient's K loop 2x faster: conv_wgrad_ps.hip).  The consumers keep
// the two 32-channel halves of a K-step in two register sets; the step's barrier sits between the two MFMA blocks and every set is
// re-read for the next half right behind the block that used it, so no step begins with barrier -> ds_read -> wait.
//
// PERSIST (round 6): the workgroup walks `tpw` output tiles (tile b, b + G, b + 2 G, ... of the XCD-ordered list, G = the grid) as ONE
// pipeline: the stage ring does not drain at a tile boundary -- the last K-step of tile k issues the first stage of tile k + 1, which
// lands under that step's MFMAs, and the residual / BatchNorm-input tile of k + 1 is requested from inside the epilogue of k (each
// register group right after the epilogue consumed it), so the fill that every workgroup of the one-tile-per-workgroup launch pays
// in front of its first MFMA (profiles/r05_conv_trace_s3.txt: 8 of a workgroup's 28 thousand cycles on the 4-K-step layers, more
// with a cold residual) is paid once per workgroup instead of once per tile.  Taken for launches of >= 4 tiles per CU whose tile
// count divides over the 512 resident workgroups (conv_dma_choice: the stage-3 expansions forward, the reductions' data gradients,
// stages 2 and 4).  The statistics scratch sits behind the ring (the ring is live while a tile's statistics are reduced).
template <bool DGRAD, int BM, int BN, int WMW, int WNW, int S, int MINW, bool PS = false, bool PERSIST = false>
__global__ __launch_bounds__(64 * (WMW * WNW + (PS ? 4 : 0)), MINW) void conv_dma_kernel(const ConvParams p, int mtiles, int ntiles) {
  constexpr int NW = WMW * WNW, T = 64 * (NW + (PS ? 4 : 0)), BK = 64;
  static_assert(!PERSIST || (!PS && S == 2), "the persistent tile loop is written for the two-stage unspecialised pipeline");
  constexpr int NL = PS ? 4 : NW;                       // waves that stage the tiles
  constexpr int WTM = BM / WMW, WTN = BN / WNW;   // wave tile
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int PA = BM / 8, PB = BN / 8;               // 8-row DMA groups (1 KB pieces) of the two operand tiles
  constexpr int AGW = (PA + NL - 1) / NL, BGW = (PB + NL - 1) / NL;   // ... per staging wave
  constexpr int L = AGW + BGW;                          // DMA instructions per staging wave per stage
  constexpr int STAGE = (BM + BN) * BK;                 // half_t elements per stage
  constexpr bool kPingPong = NW == 8 && S >= 3;         // see the K loop
  // Every wave issues exactly L pieces per stage (the counted s_waitcnt needs one number): where the groups do not divide
  // over the waves, a wave without a group of its own in the last round fetches its previous group once more (same
  // bytes to the same LDS address).
  static_assert(BM % 8 == 0 && BN % 8 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile / wave shape");
  static_assert(PA >= NL * (AGW - 1) + 1 && PB >= NL * (BGW - 1) + 1 && (AGW == 1 ? PA >= NL : true) && (BGW == 1 ? PB >= NL : true),
                "a wave's repeated piece must exist");
  static_assert((S - 1) * L < 64, "vmcnt is a 6-bit counter");
  constexpr int RED = PERSIST ? WMW * 2 * BN * 2 : 0;   // half_t elements of the statistics scratch behind the ring ([wm][2][BN] floats)
  // (PERSIST data gradient) the BatchNorm coefficients of the tile's BN output channels, [scale | shift | mean] x 256 floats (the upper
  // half of each zero: one 1 KB LDS-DMA piece per vector), fetched by LDS-DMA while the tile's K loop runs: the persistent
  // kernels contain NO load the compiler's wait-count pass tracks -- a tracked load that may be pending at the tile loop's back edge
  // (every conditional one is, statically) makes hipcc drain the queue in front of unrelated register writes, and with it the next
  // tile's residual requests
  constexpr bool kBnLds = PERSIST && DGRAD;
  constexpr int COEF = kBnLds ? 3 * 512 : 0;
  static_assert(!kBnLds || BN <= 128, "one LDS-DMA piece carries 128 coefficients");
  __shared__ __attribute__((aligned(1024))) half_t lds[S * STAGE + RED + COEF];

  // split-K forward: grid copy z of the tile grid walks its own K range into its own fp32 slab (ConvParams::ksplit)
  constexpr bool kSplitOk = !PERSIST && !DGRAD && BM * BN <= 160 * 128;      // (few-tile launches never take the 8-fragment-wide tiles)
  int kz = 0;
  int lin = blockIdx.x;
  if (kSplitOk && p.ksplit > 1) {
    kz = lin / p.ksplit_grid;
    lin -= kz * p.ksplit_grid;
  }
  // tile `lin` of the XCD-ordered list -> (row tile, column tile): hardware block b runs on XCD b & 7, and the column tiles that
  // share a row tile's A panel are neighbours on ONE XCD's L2
  auto tile_of = [&](int l, int &mt_, int &nt_) {
    const int xcd = l & 7, j = l >> 3;
    nt_ = j % ntiles;
    mt_ = (j / ntiles) * 8 + xcd;
  };
  int nt, mt;
  tile_of(lin, mt, nt);
  if (!PERSIST && mt >= mtiles) return;     // (a persistent launch has no surplus tiles: launch_one)
  const int tpw = PERSIST ? p.tiles_per_wg : 1;
  void *const ybase = (kSplitOk && p.ksplit > 1) ? (void *)(reinterpret_cast<float *>(p.y) + (size_t)kz * p.ksplit_stride) : p.y;
  // stride-2 data gradient by parity class (ConvParams::cls): row tile mt = (class, tile of the class's rows)
  // (not instantiated for the 8-fragment-wide tiles: their epilogue has no register to spare, and no stride-2 layer takes them)
  // ... nor for the specialised kernel (round 6): with the class arithmetic in it hipcc's register assignment for the CONSUMER loop
  // changes -- accumulators rotate through the fragment registers and the pinned read / MFMA interleave comes out as nine reads in a
  // burst behind twenty MFMAs (tools/isa_loop_pattern.py; the data gradient's K loop ran 47k cycles against the forward's 32k on the
  // same 3 x 3 layer, profiles/r06_conv_probe_skip_a.txt).  conv_plan sends a by-class launch to configuration 16 instead.
  constexpr bool kClassOk = !PERSIST && !PS && DGRAD && BM * BN <= 160 * 128;
  const bool by_class = kClassOk && p.cls != 0;
  int mt_l = mt, cls_ph = 0, cls_pw = 0;
  if (by_class) {
    const int mtc = mtiles >> 2, c = mt / mtc;
    mt_l = mt - c * mtc;
    cls_ph = c >> 1;
    cls_pw = c & 1;
  }
  const int Mrows = by_class ? p.cls_mc : p.M;
  // GEMM row -> destination (image, y, x) and flat pixel index, by multiplication (ConvParams::fda / fdb)
  auto row_decompose = [&](int m, int &img, int &oy, int &ox) {
    img = conv_fastdiv(m, p.fda_mul, p.fda_sh);
    const int rem = m - img * p.rows_img;
    oy = conv_fastdiv(rem, p.fdb_mul, p.fdb_sh);
    ox = rem - oy * p.row_len;
    if (by_class) { oy = 2 * oy + cls_ph; ox = 2 * ox + cls_pw; }
  };
  auto row_pixel = [&](int m) {
    if (!by_class) return m;
    int img, oy, ox;
    row_decompose(m, img, oy, ox);
    return (img * p.Ho + oy) * p.Wo + ox;
  };
  // phase stamps (shader clock) of wave 0: [0] entry, [1] first stage landed, [2] K loop done, [3] stores drained, [4] exit
  auto stamp = [&](int k) {
    if (p.trace && threadIdx.x == 0) {
      p.trace[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
      // [5] / [6]: the 100 MHz constant clock at entry / exit -- (t[4] - t[0]) / (t[6] - t[5]) x 100 = the shader clock the workgroup ran at
      if (k == 0 || k == 4) p.trace[(size_t)blockIdx.x * 8 + (k ? 6 : 5)] = __builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = PS && wave >= NW;
  const int lw = PS ? (producer ? wave - NW : 0) : wave;   // index among the staging waves
  const int cw = producer ? 0 : wave;                      // index among the multiplying waves
  const int wm = cw / WNW, wn = cw % WNW;
  int m0 = mt_l * BM, n0 = nt * BN;       // the tile being MULTIPLIED / stored (PERSIST: advanced per tile; the gather state below runs ahead)
  int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;   // row inside an 8-row group, global 16-byte chunk

  // ---- per-lane gather state of the tile being FETCHED: A rows m0 + 8 (wave + NW i) + lrow
  int a_base[AGW], a_h[AGW], a_w[AGW];
  bool a_ok[AGW];
  int a_grp[AGW], b_grp[BGW];          // wave-uniform group index of this wave's i-th piece
#pragma unroll
  for (int i = 0; i < AGW; ++i) a_grp[i] = (lw + NL * i < PA) ? lw + NL * i : lw + NL * (i - 1);
#pragma unroll
  for (int i = 0; i < BGW; ++i) b_grp[i] = (lw + NL * i < PB) ? lw + NL * i : lw + NL * (i - 1);
  // (PERSIST: 1 x 1, stride 1, no padding, whole tiles -- conv_plan -- so a row's source pixel is the row itself and the per-tap state
  //  (a_base / a_h / a_w / a_ok: 20 VGPRs that would stay live across the K loop) does not exist: a_voff comes straight from the row)
  unsigned a_voff[AGW];
  auto gather_rows = [&](int gm0) {
    if constexpr (PERSIST) {
#pragma unroll
      for (int i = 0; i < AGW; ++i) a_voff[i] = (unsigned)(gm0 + 8 * a_grp[i] + lrow) * ((unsigned)p.in_ps * 2u) + (unsigned)gchunk * 16u;
      return;
    }
#pragma unroll
    for (int i = 0; i < AGW; ++i) {
      const int m = gm0 + 8 * a_grp[i] + lrow;
      a_ok[i] = m < Mrows;
      int img, oy, ox;
      row_decompose(a_ok[i] ? m : 0, img, oy, ox);
      a_base[i] = img * p.H * p.W;
      if (DGRAD) { a_h[i] = oy + p.pad; a_w[i] = ox + p.pad; }
      else { a_h[i] = oy * p.stride - p.pad; a_w[i] = ox * p.stride - p.pad; }
    }
  };
  gather_rows(m0);
  const int taps = p.KH * p.KW;
  const int kpt = p.Cin / BK;          // host guarantees Cin % 64 == 0
  // taps this workgroup walks: all of them, or (by class, stride 2, dilation 1) those of its parity: kh = kh0, kh0 + 2, ...
  const int kh0 = by_class ? ((cls_ph + p.pad) & 1) : 0, kw0 = by_class ? ((cls_pw + p.pad) & 1) : 0, kstep = by_class ? 2 : 1;
  const int nkh = by_class ? (p.KH > kh0 ? (p.KH - kh0 + 1) >> 1 : 0) : p.KH, nkw = by_class ? (p.KW > kw0 ? (p.KW - kw0 + 1) >> 1 : 0) : p.KW;
  const int nk_all = nkh * nkw * kpt;
  // this workgroup's K-steps [t_begin, t_begin + nk): all of them, or its share of a split-K launch
  int t_begin = 0, nk = nk_all;
  if (kSplitOk && p.ksplit > 1) {
    t_begin = (int)((long)kz * nk_all / p.ksplit);
    nk = (int)((long)(kz + 1) * nk_all / p.ksplit) - t_begin;
  }
  const unsigned wrow_bytes = (unsigned)(taps * p.Cin) * 2u;
  const char *xb = reinterpret_cast<const char *>(p.x), *wb = reinterpret_cast<const char *>(p.w);
  const unsigned in_ps_bytes = (unsigned)p.in_ps * 2u;
  constexpr unsigned kOob = 0xFFFFFF00u;
  unsigned w_voff[BGW];
  auto gather_cols = [&](int gn0) {
#pragma unroll
    for (int i = 0; i < BGW; ++i) {
      // LDS row r of the weight tile holds output channel n0 + perm(r): fragment pair (2j, 2j+1), MFMA row ii = 4 fq + rr
      // -> channel 32 j + 8 (ii >> 2) + 4 (jn & 1) + (ii & 3), so that a lane's two accumulators of a pair are EIGHT consecutive
      // channels of its pixel (16-byte epilogue loads / stores).  The permutation lives in the DMA source address only.
      const int r = 8 * b_grp[i] + lrow;
      const int n = gn0 + (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
      w_voff[i] = n < p.Nout ? (unsigned)n * wrow_bytes + (unsigned)gchunk * 16u : kOob;
    }
  };
  gather_cols(n0);
  int g_kh = kh0, g_kw = kw0, g_kc = 0;   // next stage to issue: tap (g_kh, g_kw), channel block g_kc
  if (kSplitOk && t_begin > 0) {            // (split-K: start in the middle of the walk; forward launches are never by class)
    const int tap = t_begin / kpt;
    g_kc = t_begin - tap * kpt;
    g_kh = tap / p.KW;
    g_kw = tap - g_kh * p.KW;
  }
  auto tap_setup = [&]() {
    if constexpr (PERSIST) return;
#pragma unroll
    for (int i = 0; i < AGW; ++i) {
      int sy, sx;
      bool ok = a_ok[i];
      if (DGRAD) {
        const int ty = a_h[i] - g_kh * p.dil, tx = a_w[i] - g_kw * p.dil;
        if (p.stride == 1) { sy = ty; sx = tx; }
        else if (p.stride == 2) { sy = ty >> 1; sx = tx >> 1; ok = ok && !((ty | tx) & 1); }     // (negative ty / tx fail the range test below)
        else { sy = ty / p.stride; sx = tx / p.stride; ok = ok && (sy * p.stride == ty) && (sx * p.stride == tx); }
        ok = ok && ty >= 0 && tx >= 0 && sy < p.H && sx < p.W;
      } else {
        sy = a_h[i] + g_kh * p.dil; sx = a_w[i] + g_kw * p.dil;
        ok = ok && (unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W;
      }
      a_voff[i] = ok ? (unsigned)(a_base[i] + sy * p.W + sx) * in_ps_bytes + (unsigned)gchunk * 16u : kOob;
    }
  };
  tap_setup();
  // wave-uniform LDS destinations: stage base + group * 1 KB (the DMA adds 16 B per lane)
  half_t *const b_lds = lds + BM * BK;
  auto issue = [&](int buf) {
    const unsigned cbo = (unsigned)g_kc * (BK * 2), wbo = (unsigned)((g_kh * p.KW + g_kw) * p.Cin + g_kc * BK) * 2u;   // uniform
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb) + cbo, 0, (int)(p.x_bytes - cbo), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb) + wbo, 0, (int)(p.w_bytes - wbo), 0x00020000);
    half_t *const sa = lds + buf * STAGE, *const sb = b_lds + buf * STAGE;
    if (!(PS && (p.probe & 1) && (g_kh != kh0 || g_kw != kw0))) {      // (ConvParams::probe: timing probe, normally 0)
#pragma unroll
      for (int i = 0; i < AGW; ++i)
        dma16(rx, sa + a_grp[i] * 512, a_voff[i]);
    }
    if (!(PS && (p.probe & 2) && (g_kh != kh0 || g_kw != kw0 || g_kc != 0))) {      // (bit 1: the weight pieces of every K-step but the first)
#pragma unroll
      for (int i = 0; i < BGW; ++i)
        dma16(rw, sb + b_grp[i] * 512, w_voff[i]);
    }
    if constexpr (PERSIST) {
      ++g_kc;      // (one tap: fetch_seek re-positions at the tile boundary)
    } else if (++g_kc == kpt) {
      g_kc = 0;
      g_kw += kstep;
      if (g_kw >= p.KW) { g_kw = kw0; g_kh += kstep; }
      tap_setup();
    }
  };

  // (PERSIST) position the fetch side at K-step `step` of tile l.  The gather registers are re-derived at every tile boundary and
  // again behind the epilogue, so that they are dead while the epilogue runs (its own pressure is the kernel's peak)
  auto fetch_seek = [&](int l, int step) {
    int fmt, fnt;
    tile_of(l, fmt, fnt);
    gather_rows(fmt * BM);
    gather_cols(fnt * BN);
    g_kc = step;       // (one tap; step < kpt)
  };

  floatx4 acc[MI][NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) acc[i][jn] = floatx4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  int fr = lane & 15, fq = lane >> 4;
  int sw = fq ^ (fr & 7);
  int a_rd = (wm * WTM + fr) * BK, b_rd = BM * BK + (wn * WTN + fr) * BK;
  // (PERSIST) the lane-derived constants above are re-derived per tile from an opaque copy of the lane id: as loop invariants they
  // would all stay live across the epilogue, whose own pressure is the kernel's peak
  auto rederive_lane_constants = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    lrow = l >> 3; gchunk = (l & 7) ^ lrow;
    fr = l & 15; fq = l >> 4;
    sw = fq ^ (fr & 7);
    a_rd = (wm * WTM + fr) * BK; b_rd = BM * BK + (wn * WTN + fr) * BK;
  };
  // The product is formed TRANSPOSED (weights as the MFMA A operand): D^T[n][m] puts 4 consecutive output channels
  // n = fq*4 + r of one pixel m = fr into each lane -> 8-byte epilogue stores.
  auto compute = [&](int buf) {
    const half_t *const base = lds + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = (sw ^ (ks * 4)) * 8;
      half8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const half8 *>(base + a_rd + i * 16 * BK + co);
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) fb[jn] = *reinterpret_cast<const half8 *>(base + b_rd + jn * 16 * BK + co);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
    }
  };

  // ---- the epilogue's residual tile is requested NOW: its (cold) latency then hides under the K loop instead of standing
  // between the last MFMA and the stores (measured on the stage-3 expansion, K = 256: 54.6 -> us with the loads in the
  // epilogue, 32 us for the same layer without a residual).  Ordinary loads retire in order with the LDS-DMA requests, so
  // the first counted vmcnt of the loop also covers them.
  static_assert(NI % 2 == 0 && WTN % 32 == 0, "the epilogue works on fragment pairs (32 channels)");
  constexpr int NP = NI / 2;
  const bool vec4 = (p.out_ps % 4 == 0) && (p.Nout % 4 == 0) && (!p.res || p.res_ps % 4 == 0);
  const bool vec8 = (p.out_ps % 8 == 0) && (p.Nout % 8 == 0) && (!p.res || p.res_ps % 8 == 0);
  constexpr bool kPre = MI * NI <= 20;      // 2 VGPRs per fragment; the 8-fragment-wide tiles have none to spare
  half8 rpre[kPre ? MI : 1][kPre ? NP : 1];
  const bool pre_res = kPre && p.res != nullptr && vec8;
  // the BatchNorm input a fused backward reduction reads (sn_conv_dgrad_bn) takes the same slot when there is no residual
  const bool pre_bnx = DGRAD && kPre && p.res == nullptr && p.bn_x != nullptr && p.stats != nullptr && vec8 && p.bn_x_ps % 8 == 0;   // (only sn_conv_dgrad_bn sets bn_x)
  const half_t *const pre_src = pre_res ? p.res : p.bn_x;
  const int pre_ps = pre_res ? p.res_ps : p.bn_x_ps;
  // group (i, jp) of the tile at (pm0, pn0)
  auto pre_load = [&](int pm0, int pn0, int i, int jp) {
    const int mr = pm0 + wm * WTM + i * 16 + (lane & 15);
    const int n = pn0 + wn * WTN + jp * 32 + (lane >> 4) * 8;
    if constexpr (PERSIST) {       // whole tiles only (launch_one): no bounds, and a load the compiler does not track
      return load16_untracked(pre_src + (size_t)mr * pre_ps + n);
    } else {
      const int m = row_pixel(mr < Mrows ? mr : 0);
      return (mr < Mrows && n < p.Nout) ? *reinterpret_cast<const half8 *>(pre_src + (size_t)m * pre_ps + n) : half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  auto pre_load_tile = [&](int pm0, int pn0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) rpre[i][jp] = pre_load(pm0, pn0, i, jp);
  };
  if constexpr (kPre) if ((pre_res || pre_bnx) && !producer) pre_load_tile(m0, n0);

  if constexpr (PS) {
    if (producer) {
      // ---- producer: stage t must have landed before barrier t; behind it the consumers are done with stage t - 1, whose buffer
      // takes stage t + S - 1
#pragma unroll
      for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(s);
      int nxt = S - 1;
      for (int t = 0; t < nk; ++t) {
        const int young = min(S - 2, nk - 1 - t);
        if (S > 3 && young >= 2) wait_vmcnt<(S > 3 ? 2 : 0) * L>();
        else if (S > 2 && young == 1) wait_vmcnt<(S > 2 ? 1 : 0) * L>();
        else wait_vmcnt<0>();
        if (!(p.probe & 4)) __builtin_amdgcn_s_barrier();      // (probe bit 2: the K loop without its step barrier -- timing only)
        if (t + S - 1 < nk) issue(nxt);
        nxt = nxt + 1 == S ? 0 : nxt + 1;
      }
    } else {
      // ---- consumer: register set 0 = channels 0-31 of the K-step, set 1 = channels 32-63
      half8 fa0[MI], fb0[NI], fa1[MI], fb1[NI];
      auto read = [&](int buf, int ks, half8 (&fa)[MI], half8 (&fb)[NI]) {
        const half_t *const base = lds + buf * STAGE;
        const int co = (sw ^ (ks * 4)) * 8;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const half8 *>(base + a_rd + i * 16 * BK + co);
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) fb[jn] = *reinterpret_cast<const half8 *>(base + b_rd + jn * 16 * BK + co);
      };
      auto mma = [&](const half8 (&fa)[MI], const half8 (&fb)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jn], fa[i], acc[i][jn], 0, 0, 0);
      };
      // MI * NI MFMAs and MI + NI fragment reads per half step: one read behind every second MFMA (a ds_read_b128 occupies the
      // LDS for 4 cycles per wave; the matrix pipe takes 16 per MFMA), pinned -- left alone hipcc sinks every read behind the
      // last MFMA that uses its destination
      auto interleave = [&]() {
#pragma unroll
        for (int k = 0; k < MI + NI; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // two MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - 2 * (MI + NI) > 0 ? MI * NI - 2 * (MI + NI) : 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto step_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of the stage behind this barrier has returned
        if (!(p.probe & 4)) __builtin_amdgcn_s_barrier();
      };
      if (nk > 0) {
        step_barrier();
        stamp(1);
        read(0, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
      }
      int cur = 0, nxt = 1;
      for (int t = 0; t + 1 < nk; ++t) {
        mma(fa0, fb0);
        read(cur, 1, fa1, fb1);
        interleave();
        step_barrier();
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
    }
  } else if constexpr (!PERSIST) {
    // ---- pipeline: stages t+1 .. t+S-1 in flight under compute(t); one barrier per K-step
  #pragma unroll
    for (int s = 0; s < S - 1; ++s)
      if (s < nk) issue(s);
    int cur = 0, nxt = S - 1;   // buffer of stage t / of stage t+S-1
    int t = 0;
    for (; t + S - 1 < nk; ++t) {
      wait_vmcnt<(S - 2) * L>();          // stage t has landed (this wave's part); S-2 younger stages stay in flight
      __builtin_amdgcn_s_barrier();       // ... everybody's part has, and everybody is done reading buffer `nxt` (stage t-1)
      if (t == 0) stamp(1);
      // A wave's instruction stream is in order: while it issues its DMA pieces (tools/probes/dma_rate_probe.hip: ~70 cycles each
      // under load) it issues no MFMA, and the barrier puts every wave of the workgroup in the same phase.  With two waves
      // per SIMD (8-wave workgroups: waves w and w + 4 share a SIMD) the upper half therefore multiplies FIRST and fetches
      // afterwards: one half's MFMAs run under the other half's DMA issue.  Buffer `nxt` is free for the whole K-step.
      if (kPingPong && wave >= NW / 2) {
        compute(cur);
        issue(nxt);
      } else {
        issue(nxt);
        compute(cur);
      }
      cur = cur + 1 == S ? 0 : cur + 1;
      nxt = nxt + 1 == S ? 0 : nxt + 1;
    }
    for (; t < nk; ++t) {                 // drain: nothing left to issue; nk-1-t younger stages are still in flight
      const int young = nk - 1 - t;
      if (S > 3 && young >= 2) wait_vmcnt<(S > 3 ? 2 : 0) * L>();
      else if (S > 2 && young == 1) wait_vmcnt<(S > 2 ? 1 : 0) * L>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      compute(cur);
      cur = cur + 1 == S ? 0 : cur + 1;
    }
  }

  // ---- everything behind a tile's last MFMA: epilogue stores and the statistics partials.  PERSIST: `has_next` -- the workgroup
  // has another tile, at (nm0, nn0): its residual / BatchNorm-input groups are requested as the epilogue releases their registers
  auto finish_tile = [&](bool has_next, int nm0, int nn0) {
  stamp(2);
  // ---- epilogue: lane (fr, fq) holds, for each (i, jp), pixel m = ..+fr and the 8 channels n = ..+fq*8 .. +7 (first four in
  // the accumulator of fragment 2 jp, last four in that of 2 jp + 1: the weight-row permutation above)
  float st_s[NP][8], st_q[NP][8];        // BatchNorm statistics of this lane's output channels (host: only with `vec4`)
#pragma unroll
  for (int jp = 0; jp < NP; ++jp)
#pragma unroll
    for (int r = 0; r < 8; ++r) st_s[jp][r] = st_q[jp][r] = 0.f;
  // statistics of four stored values o[0..3] at channels n .. n+3 of pixel m (forward: sum, sum of squares; data gradient with
  // bn_x: sum g, sum g (x - mean) of the BatchNorm the gradient is about to pass)
  auto stats4 = [&](int m, int n, const half4 o, int jp, int h) {
    if (DGRAD && !PERSIST && p.bn_x) {
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
        st_s[jp][4 * h + r] += gf;
        st_q[jp][4 * h + r] += gf * (xf - muv[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float f = (float)o[r];
        st_s[jp][4 * h + r] += f;
        st_q[jp][4 * h + r] += f * f;
      }
    }
  };
  // one instantiation per store width (the three bodies in ONE unrolled loop nest exceed the full-unroll budget for the 8-fragment
  // tiles, and a rolled loop indexes the accumulators dynamically -> scratch)
  auto epilogue = [&](auto path_tag) {
    constexpr int PATH = decltype(path_tag)::value;
    // fused BatchNorm-backward reduction, 16-byte path: the per-channel constants of this lane's channels are loaded once per
    // fragment pair, not once per pixel row (narrow tiles only: 24 VGPRs per pair)
    constexpr bool kBnHoist = DGRAD && !PERSIST && PATH == 0 && NP <= 2;
    float bsc[kBnHoist ? NP : 1][8], bsh[kBnHoist ? NP : 1][8], bmu[kBnHoist ? NP : 1][8];
    if constexpr (kBnHoist) {
      if (p.bn_x && p.stats) {
  #pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          const int n = n0 + wn * WTN + jp * 32 + fq * 8;
          if (n < p.Nout) {
  #pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 sc = *reinterpret_cast<const float4 *>(p.bn_scale + n + 4 * h), sh = *reinterpret_cast<const float4 *>(p.bn_shift + n + 4 * h);
              const float4 mu = *reinterpret_cast<const float4 *>(p.bn_mean + n + 4 * h);
              bsc[jp][4 * h] = sc.x; bsc[jp][4 * h + 1] = sc.y; bsc[jp][4 * h + 2] = sc.z; bsc[jp][4 * h + 3] = sc.w;
              bsh[jp][4 * h] = sh.x; bsh[jp][4 * h + 1] = sh.y; bsh[jp][4 * h + 2] = sh.z; bsh[jp][4 * h + 3] = sh.w;
              bmu[jp][4 * h] = mu.x; bmu[jp][4 * h + 1] = mu.y; bmu[jp][4 * h + 2] = mu.z; bmu[jp][4 * h + 3] = mu.w;
            }
          }
        }
      }
    }
  #pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int mr = m0 + wm * WTM + i * 16 + fr;
      if (mr >= Mrows) continue;
      const int m = row_pixel(mr);        // flat destination pixel (the GEMM row itself unless the rows run class by class)
  #pragma unroll
      for (int jp = 0; jp < NP; ++jp) {
        const int n = n0 + wn * WTN + jp * 32 + fq * 8;
        if (n >= p.Nout) continue;
        float v[8];
  #pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[i][2 * jp][r]; v[4 + r] = acc[i][2 * jp + 1][r]; }
        if constexpr (PATH == 0) {          // whole 16-byte groups: n + 8 <= Nout
          if (!PERSIST && p.bias) {      // (no persistent launch has a bias: conv_plan)
            const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
          if constexpr (kPre) {
            if (pre_res) {
  #pragma unroll
              for (int r = 0; r < 8; ++r) v[r] += (float)rpre[i][jp][r];
            }
          } else if (p.res) {
            const half8 rv = *reinterpret_cast<const half8 *>(p.res + (size_t)m * p.res_ps + n);
  #pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
          }
          if (p.relu) {
  #pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
          }
          if (!PERSIST && p.out_f32) {
            float *yo = reinterpret_cast<float *>(ybase) + (size_t)m * p.out_ps + n;
            *reinterpret_cast<float4 *>(yo) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            half8 o;
  #pragma unroll
            for (int r = 0; r < 8; ++r) o[r] = (half_t)v[r];
            // (non-temporal stores measured: no difference in the step, profiles/r04_ab_class_nt_fold.txt)
            *reinterpret_cast<half8 *>(reinterpret_cast<half_t *>(ybase) + (size_t)m * p.out_ps + n) = o;
            if constexpr (!DGRAD && !PERSIST) if (p.out2) {   // the next unit's moving-statistics BatchNorm (+ ReLU) of the value just stored
              const float4 s0 = *reinterpret_cast<const float4 *>(p.o2_scale + n), s1 = *reinterpret_cast<const float4 *>(p.o2_scale + n + 4);
              const float4 h0 = *reinterpret_cast<const float4 *>(p.o2_shift + n), h1 = *reinterpret_cast<const float4 *>(p.o2_shift + n + 4);
              const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
              half8 o2;
  #pragma unroll
              for (int r = 0; r < 8; ++r) {
                float f = (float)o[r] * sc[r] + sh[r];              // as bn_apply_kernel (nn_ops.hip) forms it
                if (p.o2_relu) f = f > 0.f ? f : 0.f;
                o2[r] = (half_t)f;
              }
              *reinterpret_cast<half8 *>(p.out2 + (size_t)m * p.out2_ps + n) = o2;
            }
            if (p.stats) {
              bool done = false;
              if constexpr (kBnLds) {
                if (p.bn_x) {      // (conv_plan: a persistent launch with bn_x has the BatchNorm input in `rpre`)
                  // re-read per group (asm volatile: never merged over the groups, where the 24 values would be live across all of them)
                  const float *cf = reinterpret_cast<const float *>(lds + S * STAGE + RED) + (wn * WTN + jp * 32 + fq * 8);
                  floatx4 cv[6];
#pragma unroll
                  for (int w3 = 0; w3 < 3; ++w3)
#pragma unroll
                    for (int h = 0; h < 2; ++h) cv[2 * w3 + h] = lds_read16_untracked(cf + 256 * w3 + 4 * h);
                  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                  for (int k6 = 0; k6 < 6; ++k6) tie(cv[k6]);
                  float sc8[8], sh8[8], mu8[8];
#pragma unroll
                  for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sc8[4 * h + r] = cv[h][r]; sh8[4 * h + r] = cv[2 + h][r]; mu8[4 * h + r] = cv[4 + h][r]; }
                  const half8 xv = rpre[i][jp];
  #pragma unroll
                  for (int r = 0; r < 8; ++r) {
                    const float xf = (float)xv[r], yv = xf * sc8[r] + sh8[r];
                    const bool pass = p.bn_act == 0 || (p.bn_act == 1 ? yv > 0.f : (yv >= 0.f && yv <= 6.f));
                    const float gf = pass ? (float)o[r] : 0.f;
                    st_s[jp][r] += gf;
                    st_q[jp][r] += gf * (xf - mu8[r]);
                  }
                  done = true;
                }
              }
              if constexpr (kBnHoist) {
                if (p.bn_x) {
                  half8 xv;
                  if constexpr (kPre) xv = pre_bnx ? rpre[i][jp] : *reinterpret_cast<const half8 *>(p.bn_x + (size_t)m * p.bn_x_ps + n);
                  else xv = *reinterpret_cast<const half8 *>(p.bn_x + (size_t)m * p.bn_x_ps + n);
  #pragma unroll
                  for (int r = 0; r < 8; ++r) {
                    const float xf = (float)xv[r], yv = xf * bsc[jp][r] + bsh[jp][r];
                    const bool pass = p.bn_act == 0 || (p.bn_act == 1 ? yv > 0.f : (yv >= 0.f && yv <= 6.f));
                    const float gf = pass ? (float)o[r] : 0.f;
                    st_s[jp][r] += gf;
                    st_q[jp][r] += gf * (xf - bmu[jp][r]);
                  }
                  done = true;
                }
              }
              if (!done) {
                stats4(m, n, half4{o[0], o[1], o[2], o[3]}, jp, 0);
                stats4(m, n + 4, half4{o[4], o[5], o[6], o[7]}, jp, 1);
              }
            }
          }
        } else if constexpr (PATH == 1) {   // 8-byte groups, each with its own bound (Nout = 84, ...)
  #pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int nh = n + 4 * h;
            if (nh >= p.Nout) continue;
            if (p.bias) {
              const float4 bv = *reinterpret_cast<const float4 *>(p.bias + nh);
              v[4 * h + 0] += bv.x; v[4 * h + 1] += bv.y; v[4 * h + 2] += bv.z; v[4 * h + 3] += bv.w;
            }
            if (p.res) {
              const half4 rv = *reinterpret_cast<const half4 *>(p.res + (size_t)m * p.res_ps + nh);
  #pragma unroll
              for (int r = 0; r < 4; ++r) v[4 * h + r] += (float)rv[r];
            }
            if (p.relu) {
  #pragma unroll
              for (int r = 0; r < 4; ++r) v[4 * h + r] = v[4 * h + r] > 0.f ? v[4 * h + r] : 0.f;
            }
            if (p.out_f32) {
              *reinterpret_cast<float4 *>(reinterpret_cast<float *>(ybase) + (size_t)m * p.out_ps + nh) =
                  make_float4(v[4 * h + 0], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            } else {
              half4 o;
  #pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (half_t)v[4 * h + r];
              *reinterpret_cast<half4 *>(reinterpret_cast<half_t *>(ybase) + (size_t)m * p.out_ps + nh) = o;
              if (p.stats) stats4(m, nh, o, jp, h);
            }
          }
        } else {
  #pragma unroll
          for (int r = 0; r < 8; ++r) {
            if (n + r >= p.Nout) continue;
            float x = v[r];
            if (p.bias) x += p.bias[n + r];
            if (p.res) x += (float)p.res[(size_t)m * p.res_ps + n + r];
            if (p.relu) x = x > 0.f ? x : 0.f;
            if (p.out_f32) reinterpret_cast<float *>(ybase)[(size_t)m * p.out_ps + n + r] = x;
            else reinterpret_cast<half_t *>(ybase)[(size_t)m * p.out_ps + n + r] = (half_t)x;
          }
        }
      }
    }
  };
  if constexpr (PERSIST) {
    epilogue(std::integral_constant<int, 0>{});      // (the launcher takes the persistent twin only for 16-byte fp16 rows)
    // the residual / BatchNorm-input tile of the workgroup's NEXT tile, requested behind the last store: it lands under the
    // statistics and the first K-step of that tile (not group by group inside the epilogue: whatever wait the compiler places there
    // -- a spill reload is enough -- would drain these cold requests one by one, loads return in order)
    if constexpr (kPre) if (has_next && (pre_res || pre_bnx)) pre_load_tile(nm0, nn0);
  } else if (!producer) {
    if (vec8) epilogue(std::integral_constant<int, 0>{});
    else if (vec4) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 2>{});
  }
  if (p.trace) {
    wait_vmcnt<0>();
    stamp(3);
  }
  if (p.stats) {
    // the 16 lanes that share fq hold different pixels of the same 4 channels -> xor-shuffle over fr, then the WMW waves of a
    // column block through LDS (the K loop is over), summed in wave order: fixed order -> deterministic
#pragma unroll
    for (int jp = 0; jp < NP; ++jp)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        st_s[jp][r] = row16_sum(st_s[jp][r]);
        st_q[jp][r] = row16_sum(st_q[jp][r]);
      }
    // [wm][2][BN]; PERSIST: behind the ring (the next tile's first stage already sits in it) and between raw barriers (a
    // __syncthreads would also wait for the epilogue's stores and the next tile's residual requests)
    float *red = reinterpret_cast<float *>(lds + (PERSIST ? S * STAGE : 0));
    auto sync = [&]() {
      if constexpr (PERSIST) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else {
        __syncthreads();
      }
    };
    sync();
    if (fr == 0 && !producer) {
#pragma unroll
      for (int jp = 0; jp < NP; ++jp)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int c = wn * WTN + jp * 32 + fq * 8 + r;
          red[(wm * 2 + 0) * BN + c] = st_s[jp][r];
          red[(wm * 2 + 1) * BN + c] = st_q[jp][r];
        }
    }
    sync();
    int tid_s = tid;
    if constexpr (PERSIST) asm volatile("" : "+v"(tid_s));      // (derived per tile: hoisted out of the tile loop these addresses are spilled)
    for (int idx = tid_s; idx < 2 * BN; idx += T) {
      const int which = idx / BN, col = idx - which * BN;
      const int n = n0 + col;
      if (n < p.Nout) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WMW; ++w) a += red[(w * 2 + which) * BN + col];
        p.stats[((size_t)mt * 2 + which) * p.Nout + n] = a;
      }
    }
  }
  stamp(4);
  };   // finish_tile

  if constexpr (PERSIST) {
    // ---- the workgroup's tiles as ONE two-stage pipeline: g counts K-steps across tiles; stage g + 1 (possibly the next tile's first)
    // is issued behind barrier g and lands under compute(g)
    const int total = tpw * nk;
    issue(0);
    int cur = 0, g = 0;
    for (int seq = 0;; ++seq) {
      for (int t = 0; t < nk; ++t, ++g) {
        if (t > 0 || seq == 0) wait_vmcnt<0>();   // stage g has landed (this wave's part; a tile's first stage was waited for in front of the epilogue before it)
        __builtin_amdgcn_s_barrier();     // ... everybody's part has, and everybody is done with stage g - 1 (and its tile's statistics scratch)
        if (g == 0) stamp(1);
        if constexpr (kBnLds) {
          // this tile's BatchNorm coefficients -> LDS (wave 0, three 1 KB pieces: lanes 0 - 31 carry 128 floats, the others read
          // out of range = zeros).  Behind barrier t = 0 nobody reads the previous tile's any more; this wave's vmcnt(0) of step 1 and
          // that step's barrier (nk >= 2: conv_plan) put them in front of every wave's epilogue.
          if (t == 0 && wave == 0 && p.bn_x) {
            half_t *const cdst = lds + S * STAGE + RED;
            const float *const src[3] = {p.bn_scale, p.bn_shift, p.bn_mean};
#pragma unroll
            for (int w3 = 0; w3 < 3; ++w3) {
              const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src[w3] + n0), 0, BN * 4, 0x00020000);
              dma16(rc, cdst + w3 * 512, (unsigned)lane * 16u);
            }
          }
        }
        if (g + 1 < total) {
          if (t == nk - 1) fetch_seek(lin + (seq + 1) * (int)gridDim.x, 0);    // the next tile's first stage
          issue(cur ^ 1);
        }
        compute(cur);
        cur ^= 1;
      }
      // the next tile's first stage was issued one compute phase ago: wait for it HERE, so that nothing the epilogue issues (stores,
      // the next residual tile) stands between that stage and the first barrier of the next tile's loop
      wait_vmcnt<0>();
      if constexpr (kPre) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jp = 0; jp < NP; ++jp) tie(rpre[i][jp]);
      }
      const bool has_next = seq + 1 < tpw;
      int nmt = mt, nnt = nt;
      if (has_next) tile_of(lin + (seq + 1) * (int)gridDim.x, nmt, nnt);
      finish_tile(has_next, nmt * BM, nnt * BN);
      if (!has_next) break;
      mt = nmt; nt = nnt;
      m0 = mt * BM; n0 = nt * BN;
      zero_acc();
      __builtin_amdgcn_sched_barrier(0);      // (the re-derivation below must not be scheduled up into the epilogue: its registers are the point)
      rederive_lane_constants();
      fetch_seek(lin + (seq + 1) * (int)gridDim.x, nk > 1 ? 1 : 0);   // (nk == 1: re-positioned again before the next issue)
    }
  } else {
    finish_tile(false, 0, 0);
  }
}

// cfg -> tile shape.  LDS = stages * (bm + bn) * 128 B.  The numbers are those of round 2's seventeen-entry table (profiles/r02_conv_tune*.txt
// name them); the entries no selection rule and no whole-step A/B ever chose were removed in round 3 (bm = 0: no such configuration).
static const ConvDmaConfig kCfg[kConvDmaConfigs + 1] = {
    {0, 0, 0, 0, 0},
    {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
    {128, 256, 512, 3, 3 * 384 * 128},   // 4: 144 KB, 8 waves: FullyConnected over 6000 RoIs with >= 1024 outputs
    {64, 128, 256, 3, 3 * 192 * 128},    // 5: 72 KB, 2 / CU, wave tile 32 x 64: narrow heads, long contractions
    {64, 128, 256, 2, 2 * 192 * 128},    // 6: 48 KB, 3 / CU: narrow heads, FC, single-K-step layers
    {256, 256, 512, 2, 2 * 512 * 128},   // 7: 128 KB, wave tile 64 x 128: >= 3.75 tiles of 256 x 256 per CU
    {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
    // 160-row tiles: 20 480 pixels (20 chips x 32 x 32) = 128 row tiles, i.e. 256 / 512 / 1024 workgroups for 256 / 512 / 1024
    // output channels -- whole multiples of the 256 CUs
    {160, 128, 256, 2, 2 * 288 * 128},   // 14: 72 KB, 4 waves, 2 workgroups / CU (forward default)
    {0, 0, 0, 0, 0},
    {160, 128, 512, 2, 2 * 288 * 128},   // 16: 8 waves (2 x 4), wave tile 80 x 32, 2 workgroups / CU (data-gradient default)
    {0, 0, 0, 0, 0},
    // producer / consumer specialised (round 3): 4 multiplying waves (2 x 2) + 4 staging waves, one workgroup per CU
    {160, 128, 512, 4, 4 * 288 * 128},   // 18: 144 KB, wave tile 80 x 64: long contractions with about one tile per CU
    {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0},
    // persistent tile loop (round 6; PERSIST in conv_dma_kernel): 512 workgroups walk tiles / 512 tiles each as one pipeline
    {160, 128, 256, 2, 2 * 288 * 128 + 2048},   // 24: 14's shape (forward)
    {0, 0, 0, 0, 0},
    {160, 128, 512, 2, 2 * 288 * 128 + 2048 + 3072},   // 26: 16's shape (data gradient; + the BatchNorm coefficients' 3 KB)
};

ConvDmaConfig conv_dma_config(int cfg) { return (cfg >= 1 && cfg <= kConvDmaConfigs) ? kCfg[cfg] : kCfg[0]; }

template <bool DGRAD, int BM, int BN, int WMW, int WNW, int S, int MINW, bool PS = false, bool PERSIST = false>
static void launch_one(const ConvParams &p, hipStream_t s) {
  const int mtiles = (DGRAD && !PS && p.cls && BM * BN <= 160 * 128) ? 4 * sn_div_up(p.cls_mc, BM) : sn_div_up(p.M, BM), ntiles = sn_div_up(p.Nout, BN);
  const int base = sn_div_up(mtiles, 8) * 8 * ntiles;
  ConvParams q = p;
  q.ksplit_grid = base;
  if constexpr (PERSIST) {
    q.tiles_per_wg = conv_persist_tiles_per_wg(p.M, p.Nout, BM, BN);      // (conv_plan chose this configuration only where it is > 0)
    q.ksplit = 1;
    hipLaunchKernelGGL((conv_dma_kernel<DGRAD, BM, BN, WMW, WNW, S, MINW, PS, true>), dim3((unsigned)(base / q.tiles_per_wg)),
                       dim3(64 * WMW * WNW), 0, s, q, mtiles, ntiles);
  } else {
    const dim3 grid((unsigned)base * (unsigned)((!DGRAD && p.ksplit > 1 && BM * BN <= 160 * 128) ? p.ksplit : 1));
    hipLaunchKernelGGL((conv_dma_kernel<DGRAD, BM, BN, WMW, WNW, S, MINW, PS>), grid, dim3(64 * (WMW * WNW + (PS ? 4 : 0))), 0, s, q, mtiles, ntiles);
  }
}

template <bool DGRAD>
static int launch_cfg(const ConvParams &p, int cfg, hipStream_t s) {
  switch (cfg) {
    case 4: launch_one<DGRAD, 128, 256, 2, 4, 3, 2>(p, s); break;
    case 5: launch_one<DGRAD, 64, 128, 2, 2, 3, 2>(p, s); break;
    case 6: launch_one<DGRAD, 64, 128, 2, 2, 2, 3>(p, s); break;
    case 7: launch_one<DGRAD, 256, 256, 4, 2, 2, 1>(p, s); break;
    case 14: launch_one<DGRAD, 160, 128, 2, 2, 2, 2>(p, s); break;
    case 16: launch_one<DGRAD, 160, 128, 2, 4, 2, 2>(p, s); break;
    case 18: launch_one<DGRAD, 160, 128, 2, 2, 4, 1, true>(p, s); break;
    case 24: launch_one<DGRAD, 160, 128, 2, 2, 2, 2, false, true>(p, s); break;
    case 26: launch_one<DGRAD, 160, 128, 2, 4, 2, 4, false, true>(p, s); break;
    default: SN_REQUIRE(false, "conv_dma_launch: unknown configuration %d", cfg);
  }
  SN_CHECK_LAUNCH();
  return SN_OK;
}

// diagnostics: phase timeline of the next launches' workgroups into buf (>= 8 x grid 64-bit words; tools/conv_trace.py), NULL = off
static std::atomic<unsigned long long *> g_conv_trace{nullptr};
void conv_dma_set_trace(unsigned long long *buf) { g_conv_trace.store(buf, std::memory_order_relaxed); }

int conv_dma_launch(const ConvParams &p, bool dgrad, int cfg, hipStream_t s) {
  unsigned long long *trace = g_conv_trace.load(std::memory_order_relaxed);
  if (trace) {
    ConvParams q = p;
    q.trace = trace;
    static const int probe = getenv("SNIPER_CONV_PROBE_SKIP_A") ? atoi(getenv("SNIPER_CONV_PROBE_SKIP_A")) : 0;
    q.probe = probe;
    return dgrad ? launch_cfg<true>(q, cfg, s) : launch_cfg<false>(q, cfg, s);
  }
  return dgrad ? launch_cfg<true>(p, cfg, s) : launch_cfg<false>(p, cfg, s);
}
