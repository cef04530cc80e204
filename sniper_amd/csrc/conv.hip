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
#include <algorithm>
#include <atomic>

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

// which kernel a layer takes: dma = a configuration of conv_dma_kernel (conv_dma.hip), 0 -> conv_igemm_kernel
struct ConvPlan {
  int bm, mtiles, ntiles, dma;
  unsigned x_bytes, w_bytes;
  int cls, cls_mc;       // stride-2 data gradient enumerated by parity class (ConvParams::cls): mtiles = 4 x tiles of cls_mc rows
};
// Tuning override (tools/conv_tune.py, tests): -1 = the built-in table, 0 = the register-staged kernel (conv_igemm_kernel) only,
// else that LDS-DMA configuration (conv_dma_config) for every layer that qualifies.  Process-wide; not meant to change while
// launches are in flight.
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static std::atomic<int> g_conv_cfg{env_int("SNIPER_CONV_CFG", -1)};   // whole-program A/B (tools/ab.sh) without code changes
SN_EXPORT int sn_conv_tune(int cfg) {
  SN_REQUIRE(cfg == -1 || cfg == 0 || conv_dma_config(cfg).bm > 0, "sn_conv_tune: no configuration %d", cfg);
  g_conv_cfg.store(cfg, std::memory_order_relaxed);
  return SN_OK;
}

// Built-in choice for a layer that qualifies for the pipelined kernels: M output pixels, Nout channels, nk 64-deep K-steps.
static int conv_dma_choice_balanced(int M, int Nout, int nk, bool dgrad) {
  // tools/conv_tune.py --insitu with the 160-row tiles in the candidate set (profiles/r02_conv_tune_insitu_v3.txt); round 2 also
  // carried an L2-warm and a cold table (profiles/r02_conv_tune*.txt) and eleven more tile configurations -- every whole-step A/B
  // (tools/ab.sh, profiles/r02_ab_balanced_tiles.txt) chose this one, so round 3 removed the others.
  // tools/conv_trace.py shows why they win at 20 chips: a K-step's operand delivery is an LDS-DMA issue cost per wave, so a
  // CU wants >= 8 waves in K loops (two 4-wave workgroups) and every CU the same number of tiles -- 20 480 pixels / 160 = 128
  // row tiles = 256 / 512 / 1024 / 2048 workgroups for 256 / 512 / 1024 / 2048 output channels, 81 920 / 160 = 512.
  const long t128 = (long)sn_div_up(M, 128) * sn_div_up(Nout, 128);
  if (t128 >= 3840 && nk >= 8) return 7;              // the long-K data gradients (RPN, deformable GEMM, fc_new_1): 256 x 256
  if (Nout < 128) return nk >= 64 ? 5 : 6;           // narrow heads: one partial column tile
  if (M < 8192) return (Nout >= 1024 && nk >= 128) ? 4 : 6;   // FullyConnected over 6000 RoIs
  const int bal = 14, bal_d = 16;                     // 160 x 128: 4 waves x 2 workgroups per CU forward, 8 waves for the data gradients
  if (nk <= 1) return 6;                              // stage 1: a single K-step, all epilogue
  // round 3: long contractions with about one 160 x 128 tile per CU take the producer / consumer specialised kernel (cfg 18):
  // its K loop is ~1.3x faster (profiles/r03_conv_tune_ps.txt: stage-3 3x3 38 -> 30 us, 1x1 1024 -> 256 24 -> 21 us), but it is one
  // 8-wave workgroup per CU, so short contractions with 4 tiles per CU -- all fill and epilogue -- lose (32 -> 47 us)
  if (nk >= 16 && (long)sn_div_up(M, 160) * sn_div_up(Nout, 128) <= 320) return 18;
  return dgrad ? bal_d : bal;
}

static int conv_dma_choice(int M, int Nout, int nk, bool dgrad) { return conv_dma_choice_balanced(M, Nout, nk, dgrad); }

// A/B and test switch: 0 = a stride-2 data gradient visits every tap for every destination pixel (three quarters of them
// zero-filled), 1 (default) = by parity class (ConvParams::cls)
static std::atomic<int> g_dgrad_by_class{env_int("SNIPER_DGRAD_BY_CLASS", 1)};
SN_EXPORT int sn_conv_dgrad_by_class(int on) {
  g_dgrad_by_class.store(on ? 1 : 0, std::memory_order_relaxed);
  return SN_OK;
}

static ConvPlan conv_plan(const ConvParams &p, bool dgrad, int use_cfg = -1) {
  // Layers whose taps are whole 64-channel K-steps and 16-byte addressable take a pipelined kernel; narrow outputs
  // (stage1 / RPN heads) and the packed stem stay on conv_igemm_kernel.
  ConvPlan q = {0, 0, 0, 0, 0u, 0u, 0, 0};
  const unsigned long x_bytes = ((unsigned long)p.N * p.H * p.W - 1) * p.in_ps * 2 + (unsigned long)p.Cin * 2;
  const unsigned long w_bytes = (unsigned long)p.Nout * p.KH * p.KW * p.Cin * 2;
  // (Nout == 64 -- the stage-1 reductions and 3 x 3 layers, 327 680 pixels each -- takes the pipelined 64 x 128 tile with half of its
  //  columns zero-filled when SN_OPT_CONV_DMA_NOUT64 is set: HBM-bound layers, the wasted MFMA half is free; A/B profiles/r06_ab_nout64.txt)
  const int min_nout = sn_debug_get(SN_OPT_CONV_DMA_NOUT64) ? 64 : 65;
  if (p.Nout >= min_nout && p.Cin % 64 == 0 && p.in_ps % 8 == 0 && x_bytes <= 0xFFFFFF00ul && w_bytes <= 0xFFFFFF00ul) {
    q.x_bytes = (unsigned)x_bytes;
    q.w_bytes = (unsigned)w_bytes;
    const int forced = g_conv_cfg.load(std::memory_order_relaxed);
    // a stride-2 data gradient walks its destination pixels parity class by parity class (each class only its own taps)
    const bool by_class = dgrad && p.stride == 2 && p.dil == 1 && p.Ho % 2 == 0 && p.Wo % 2 == 0 &&
                          g_dgrad_by_class.load(std::memory_order_relaxed) != 0;
    const int nk_full = p.KH * p.KW * (p.Cin / 64);
    const int nk = by_class ? std::max(1, ((p.KH + 1) / 2) * ((p.KW + 1) / 2) * (p.Cin / 64)) : nk_full;    // the busiest class
    int cfg = use_cfg > 0 ? use_cfg : (forced >= 0 ? forced : conv_dma_choice(p.M, p.Nout, nk, dgrad));
    if (by_class && cfg == 18 && use_cfg <= 0 && forced < 0) cfg = 16;      // (the specialised kernel carries no class arithmetic: conv_dma.hip kClassOk)
    // the persistent twins (24 / 26) of the 160 x 128 configurations: launches of >= 4 whole tiles per CU that divide over the 512
    // resident workgroups, short contractions (what a tile pays outside its K loop is what the persistent loop overlaps)
    // its epilogue: 16-byte fp16 rows, no bias, the residual / BatchNorm-input tile through the register prefetch
    const bool rows16 = !p.out_f32 && !p.out2 && !p.bias && p.out_ps % 8 == 0 && p.Nout % 8 == 0 && (!p.res || p.res_ps % 8 == 0) &&
                        (!p.bn_x || (p.stats && !p.res && p.bn_x_ps % 8 == 0));
    // (and 1 x 1, stride 1, unpadded: the persistent kernel addresses a row's source pixel as the row itself)
    const bool may_persist = !by_class && p.ksplit <= 1 && rows16 && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 &&
                             conv_persist_tiles_per_wg(p.M, p.Nout, 160, 128) > 0;
    if (use_cfg <= 0 && forced < 0 && (cfg == 14 || cfg == 16) && may_persist && nk >= 2 && nk <= 16 && sn_debug_get(SN_OPT_CONV_NO_PERSIST) == 0)
      cfg += 10;
    if ((cfg == 24 || cfg == 26) && (!may_persist || nk < 2)) cfg -= 10;      // forced on a launch that does not qualify: the plain twin
    if (cfg > 0) {
      const ConvDmaConfig c = conv_dma_config(cfg);
      q.dma = cfg;
      q.bm = c.bm;
      q.ntiles = sn_div_up(p.Nout, c.bn);
      const bool cls = by_class && c.bm * c.bn <= 160 * 128 && cfg != 18;      // (neither the 8-fragment-wide tiles nor the specialised kernel are instantiated for it)
      q.cls = cls ? 1 : 0;
      q.cls_mc = cls ? p.N * (p.Ho / 2) * (p.Wo / 2) : 0;
      q.mtiles = cls ? 4 * sn_div_up(q.cls_mc, c.bm) : sn_div_up(p.M, c.bm);
      return q;
    }
  }
  return q;
}

template <bool DGRAD>
static int conv_launch(const ConvParams &p, hipStream_t s, int use_cfg = -1) {
  const ConvPlan pl = conv_plan(p, DGRAD, use_cfg);
  if (pl.dma) {
    ConvParams q = p;
    q.x_bytes = pl.x_bytes;
    q.w_bytes = pl.w_bytes;
    q.cls = pl.cls;
    q.cls_mc = pl.cls_mc;
    conv_fastdiv_fill(q);
    return conv_dma_launch(q, DGRAD, pl.dma, s);
  }
  SN_REQUIRE(!p.stats, "convolution statistics are emitted by the pipelined kernel only (query sn_conv_fwd_stats_blocks first)");
  if (p.Nout <= 64) {
    dim3 grid(sn_div_up(p.M, 128), sn_div_up(p.Nout, 64));
    hipLaunchKernelGGL((conv_igemm_kernel<128, 64, DGRAD>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(sn_div_up(p.M, 128), sn_div_up(p.Nout, 128));
    hipLaunchKernelGGL((conv_igemm_kernel<128, 128, DGRAD>), grid, dim3(256), 0, s, p);
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

// ---- split-K forward for launches with few output tiles ------------------------------------------------------------------
// A test-time batch of two FocusChips at the finest scale is M = 2 x 36 x 36 pixels: 82 tiles of 64 x 128 for a 256-channel layer,
// one workgroup each on 256 CUs, and a lone workgroup needs ~0.5 us per K-step whatever the tile: 28 us for
// the 36-step 3x3 at 108 TF/s.  With the contraction split over `ksplit` copies of the grid every CU has work and a workgroup's
// serial chain is ksplit times shorter; the fp32 partial tiles are added by splitk_reduce_kernel in split order (deterministic)
// together with the bias / residual / ReLU epilogue.
// A launch whose plan is one of the 8-fragment-wide tiles (FullyConnected over a few hundred RoIs: 600 x 12544 -> 1024 is 20 tiles of
// 128 x 256 with 196 K-steps each -- 163 us at 94 TF/s) takes the 64 x 128 three-stage configuration instead when it splits.
struct SplitPlan { int ksplit; size_t slab_elems; int cfg; };
static SplitPlan conv_split_plan(const ConvParams &p) {
  SplitPlan sp = {1, 0, -1};
  if (p.out_f32 || p.stats) return sp;
  ConvPlan pl = conv_plan(p, false);
  if (!pl.dma) return sp;
  ConvDmaConfig c = conv_dma_config(pl.dma);
  const int nk = p.KH * p.KW * (p.Cin / 64);
  if (c.bm * c.bn > 160 * 128) {               // (the 8-fragment-wide tiles are not instantiated for the split)
    if (g_conv_cfg.load(std::memory_order_relaxed) >= 0 || p.M >= 8192 || nk < 32) return sp;
    sp.cfg = 5;
    pl = conv_plan(p, false, sp.cfg);
    c = conv_dma_config(pl.dma);
    if (!pl.dma || c.bm * c.bn > 160 * 128) return SplitPlan{1, 0, -1};
  }
  const long tiles = (long)pl.mtiles * pl.ntiles;
  // fewer tiles than CUs and a contraction worth splitting; about one tile per CU only with a long one (the RPN's 3 x 3 over 3072
  // channels on a 2-chip batch: 164 tiles x 432 K-steps)
  if (nk < 8 || tiles > 256 || (tiles >= 160 && nk < 64)) return SplitPlan{1, 0, -1};
  // (a dozen tiles -- the 98-channel offset FullyConnected over 600 RoIs: 10 tiles x 196 K-steps -- split 16 ways)
  int ks = (int)std::min<long>(std::min<long>(nk / 4, (448 + tiles - 1) / tiles), tiles <= 16 ? 16 : 8);
  if (ks < 2) return SplitPlan{1, 0, -1};
  sp.ksplit = ks;
  sp.slab_elems = (size_t)p.M * p.Nout;
  return sp;
}

// out[m][n] = act(sum_z slab[z][m][n] + bias[n] + res[m][n]) as fp16; 8 channels per thread, 16-byte accesses when every pitch allows
// (Nout % 8 == 0 ...), element by element otherwise (a 98-channel FullyConnected; no second output there)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slab, int ksplit, size_t stride, long M, int Nout,
                                                            const float *__restrict__ bias, const half_t *__restrict__ res, int res_ps,
                                                            half_t *__restrict__ out, int out_ps, int relu,
                                                            half_t *__restrict__ out2, int out2_ps, const float *__restrict__ o2_scale,
                                                            const float *__restrict__ o2_shift, int o2_relu, int out_f32) {
  const int cpr = (Nout + 7) >> 3;
  const long total = M * cpr;
  const bool vec = !out_f32 && (Nout & 7) == 0 && (out_ps & 7) == 0 && (!res || (res_ps & 7) == 0);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / cpr;
    const int n = (int)(i - m * cpr) * 8;
    if (!vec) {
      for (int r = 0; r < 8 && n + r < Nout; ++r) {
        float a = 0.f;
        for (int z = 0; z < ksplit; ++z) a += slab[(size_t)z * stride + (size_t)m * Nout + n + r];
        if (bias) a += bias[n + r];
        if (res) a += (float)res[(size_t)m * res_ps + n + r];
        if (relu && a < 0.f) a = 0.f;
        if (out_f32) reinterpret_cast<float *>(out)[(size_t)m * out_ps + n + r] = a;
        else out[(size_t)m * out_ps + n + r] = (half_t)a;
      }
      continue;
    }
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ksplit; ++z) {
      const float *sp = slab + (size_t)z * stride + (size_t)m * Nout + n;
      const float4 a = *reinterpret_cast<const float4 *>(sp), b = *reinterpret_cast<const float4 *>(sp + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (bias) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bias[n + r];
    }
    if (res) {
      const half8 rv = *reinterpret_cast<const half8 *>(res + (size_t)m * res_ps + n);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
    }
    half8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (half_t)((relu && v[r] < 0.f) ? 0.f : v[r]);
    *reinterpret_cast<half8 *>(out + (size_t)m * out_ps + n) = o;
    if (out2) {                              // second output: act(scale * stored value + shift), as the dual epilogue of conv_dma.hip
      half8 o2;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float f = (float)o[r] * o2_scale[n + r] + o2_shift[n + r];
        if (o2_relu) f = f > 0.f ? f : 0.f;
        o2[r] = (half_t)f;
      }
      *reinterpret_cast<half8 *>(out2 + (size_t)m * out2_ps + n) = o2;
    }
  }
}

SN_EXPORT size_t sn_conv_fwd_splitk_workspace_bytes(int N, int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride,
                                                    int res_pix_stride, int KH, int KW, int stride, int pad, int dil) {
  ConvParams p;
  conv_fwd_params(p, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH,
                  KW, stride, pad, dil, 0, 0);
  if (p.Ho <= 0 || p.Wo <= 0) return 0;
  const SplitPlan sp = conv_split_plan(p);
  return sp.ksplit > 1 ? sn_align((size_t)sp.ksplit * sp.slab_elems * sizeof(float)) : 0;
}

static int conv_fwd_splitk_impl(const char *who, int out_f32, const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W,
                                 int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW,
                                 int stride, int pad, int dil, int relu, void *ws, size_t ws_bytes, sn_stream_t stream) {
  ConvParams p;
  conv_fwd_params(p, x, w, bias, residual, y, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH, KW, stride, pad,
                  dil, relu, 0);
  if (int rc = conv_check(p, who)) return rc;
  const SplitPlan sp = conv_split_plan(p);
  if (sp.ksplit <= 1) {
    p.out_f32 = out_f32;
    return conv_launch<false>(p, sn_stream(stream));
  }          // nothing to split: the plain forward
  SN_REQUIRE(ws && ws_bytes >= (size_t)sp.ksplit * sp.slab_elems * sizeof(float),
             "%s: %zu bytes of scratch needed (sn_conv_fwd_splitk_workspace_bytes), got %zu", who,
             (size_t)sp.ksplit * sp.slab_elems * sizeof(float), ws_bytes);
  ConvParams q = p;
  q.y = ws; q.out_f32 = 1; q.out_ps = p.Nout; q.bias = nullptr; q.res = nullptr; q.res_ps = 0; q.relu = 0;
  q.ksplit = sp.ksplit;
  q.ksplit_stride = (long)sp.slab_elems;
  if (int rc = conv_launch<false>(q, sn_stream(stream), sp.cfg)) return rc;
  const long total = (long)p.M * ((p.Nout + 7) / 8);
  long blocks = (total + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), (const float *)ws, sp.ksplit,
                     sp.slab_elems, (long)p.M, p.Nout, bias, (const half_t *)residual, res_pix_stride, (half_t *)y, out_pix_stride, relu,
                     (half_t *)nullptr, 0, (const float *)nullptr, (const float *)nullptr, 0, out_f32);
  SN_CHECK_LAUNCH();
  return SN_OK;
}

SN_EXPORT int sn_conv_fwd_splitk(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W,
                                 int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW,
                                 int stride, int pad, int dil, int relu, void *ws, size_t ws_bytes, sn_stream_t stream) {
  return conv_fwd_splitk_impl("sn_conv_fwd_splitk", 0, x, w, bias, residual, y, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride,
                              res_pix_stride, KH, KW, stride, pad, dil, relu, ws, ws_bytes, stream);
}

// ... with an fp32 result (the 2 x 7 x 7 offset FullyConnected over the RoIs of a test batch feeds the pooling's fp32 `trans`)
SN_EXPORT int sn_conv_fwd_splitk_f32(const void *x, const void *w, const float *bias, float *y, int N, int H, int W, int Cin,
                                     int in_pix_stride, int Cout, int out_pix_stride, int KH, int KW, int stride, int pad, int dil,
                                     int relu, void *ws, size_t ws_bytes, sn_stream_t stream) {
  return conv_fwd_splitk_impl("sn_conv_fwd_splitk_f32", 1, x, w, bias, nullptr, y, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, 0, KH,
                              KW, stride, pad, dil, relu, ws, ws_bytes, stream);
}

// ---- forward with a second output (test-time residual units) ---------------------------------------------------------------
// y = conv(x) (+ bias, residual, ReLU) as sn_conv_fwd / sn_conv_fwd_splitk write it, and y2 = act(y2_scale * y + y2_shift) of the
// STORED fp16 y: the moving-statistics BatchNorm + ReLU that opens the next pre-activation unit (resnet_mx_101_e2e.py:38-40).  That
// BatchNorm reads the residual SUM (two readers: itself and the next add), so it cannot fold into the convolution's weights; as a
// second output of the epilogue that has the sum in registers it costs one more 16-byte store per lane instead of a launch that
// reads and writes the whole tensor (2 880 sn_bn_apply launches per 64-image AutoFocus pass).  sn_conv_fwd_dual_ok: 1 when the
// layer takes the pipelined kernel's 16-byte epilogue (else the caller keeps sn_conv_fwd + sn_bn_apply).
static bool conv_dual_ok(const ConvParams &p, int y2_pix_stride) {
  if (p.Ho <= 0 || p.Wo <= 0 || p.out_f32 || p.Nout % 8 != 0 || p.out_ps % 8 != 0 || (p.res && p.res_ps % 8 != 0) || y2_pix_stride % 8 != 0)
    return false;
  return conv_plan(p, false).dma != 0;
}

SN_EXPORT int sn_conv_fwd_dual_ok(int N, int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride,
                                  int KH, int KW, int stride, int pad, int dil, int y2_pix_stride) {
  ConvParams p;
  conv_fwd_params(p, nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH,
                  KW, stride, pad, dil, 0, 0);
  if (res_pix_stride % 8 != 0) return 0;
  return conv_dual_ok(p, y2_pix_stride) ? 1 : 0;
}

SN_EXPORT int sn_conv_fwd_dual(const void *x, const void *w, const float *bias, const void *residual, void *y, int N, int H, int W,
                               int Cin, int in_pix_stride, int Cout, int out_pix_stride, int res_pix_stride, int KH, int KW, int stride,
                               int pad, int dil, int relu, void *y2, int y2_pix_stride, const float *y2_scale, const float *y2_shift,
                               int y2_relu, void *ws, size_t ws_bytes, sn_stream_t stream) {
  ConvParams p;
  conv_fwd_params(p, x, w, bias, residual, y, N, H, W, Cin, in_pix_stride, Cout, out_pix_stride, res_pix_stride, KH, KW, stride, pad,
                  dil, relu, 0);
  if (int rc = conv_check(p, "sn_conv_fwd_dual")) return rc;
  SN_REQUIRE(y2 && y2_scale && y2_shift && conv_dual_ok(p, y2_pix_stride),
             "sn_conv_fwd_dual: the layer does not take the 16-byte pipelined epilogue (query sn_conv_fwd_dual_ok)");
  const SplitPlan sp = conv_split_plan(p);
  if (sp.ksplit > 1 && ws && ws_bytes >= (size_t)sp.ksplit * sp.slab_elems * sizeof(float)) {
    ConvParams q = p;
    q.y = ws; q.out_f32 = 1; q.out_ps = p.Nout; q.bias = nullptr; q.res = nullptr; q.res_ps = 0; q.relu = 0;
    q.ksplit = sp.ksplit;
    q.ksplit_stride = (long)sp.slab_elems;
    if (int rc = conv_launch<false>(q, sn_stream(stream), sp.cfg)) return rc;
    const long total = (long)p.M * (p.Nout / 8);
    long blocks = (total + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, sn_stream(stream), (const float *)ws, sp.ksplit,
                       sp.slab_elems, (long)p.M, p.Nout, bias, (const half_t *)residual, res_pix_stride, (half_t *)y, out_pix_stride, relu,
                       (half_t *)y2, y2_pix_stride, y2_scale, y2_shift, y2_relu, 0);
    SN_CHECK_LAUNCH();
    return SN_OK;
  }
  p.out2 = (half_t *)y2; p.out2_ps = y2_pix_stride; p.o2_scale = y2_scale; p.o2_shift = y2_shift; p.o2_relu = y2_relu;
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
// conv_wgrad_kernel (the fallback for operands that are not 16-byte addressable, and the packed stem) gathers the fragments
// straight from global memory with 2-byte loads (every 16-lane group reads 32 contiguous bytes, the lines stay in L1/L2
// for the next 7 k), no LDS, no barriers: each wave owns a 64x64 (co x ci) tile.  Every other layer runs on the
// wave-specialised batched kernel of conv_wgrad_ps.hip (natural-layout LDS tiles + ds_read_b64_tr_b16).
// K' is split over (tap, row-range) blocks whose partial slabs wgrad_reduce_kernel sums in split order (no atomics).
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

// ---- fallback path: the gather kernel for operands the batched kernel cannot take (pixel strides / bases that are not 16-byte
// addressable, tensors of 1 GB and more), and the A/B arm of sn_conv_wgrad_impl(0).  K-splits over (tap, row-range) blocks, partial
// slabs + wgrad_reduce_kernel: no atomics, fixed order.
struct WgradPlan {
  int gx, gy, taps, nunits, splits, units_per_split, Ho, Wo;
};

static WgradPlan wgrad_plan(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
  WgradPlan q;
  q.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  q.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  q.taps = KH * KW;
  q.gx = sn_div_up(Cout, 128);
  q.gy = sn_div_up(Cin, 128);
  q.nunits = N * q.Ho * sn_div_up(q.Wo, 32);
  const int tiles = q.gx * q.gy * q.taps;
  int splits = (tiles <= 16 ? 256 : 512) / tiles;      // at most two resident workgroups per CU
  if (splits > q.nunits / 2) splits = q.nunits / 2;
  if (splits < 1) splits = 1;
  q.units_per_split = sn_div_up(q.nunits, splits);
  q.splits = sn_div_up(q.nunits, q.units_per_split);
  return q;
}

static size_t wgrad_legacy_workspace_bytes(int N, int H, int W, int Cin, int x_pix_stride, int Cout, int dy_pix_stride, int KH,
                                           int KW, int stride, int pad, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const WgradPlan q = wgrad_plan(N, H, W, Cin, Cout, KH, KW, stride, pad, dil);
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
  WgradPlan q = wgrad_plan(N, H, W, Cin, Cout, KH, KW, stride, pad, dil);
  p.Ho = q.Ho; p.Wo = q.Wo;
  SN_REQUIRE(p.Ho > 0 && p.Wo > 0, "sn_conv_wgrad: bad dims");
  const size_t n = (size_t)Cout * q.taps * Cin;
  p.slab = nullptr;
  p.slab_stride = n;
  if (q.splits > 1) {
    if (ws && ws_bytes >= sizeof(float) * (size_t)q.splits * n && ((uintptr_t)ws % 16) == 0) p.slab = (float *)ws;
    else { q.splits = 1; q.units_per_split = q.nunits; }   // no scratch: one owner per element, no split
  }
  p.units_per_split = q.units_per_split;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(wgrad_grid(p, q.gx, q.gy, q.taps * q.splits)), dim3(256), 0, s, p);
  SN_CHECK_LAUNCH();
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
static std::atomic<int> g_wgrad_na{0};     // 0 = by layer width (Cout > 128: 256 x 128 tiles), 1 / 2 = forced (sn_conv_wgrad_impl's job_steps >= 100000)
static std::atomic<int> g_wgrad_impl{1};   // 1 = wave-specialised batched kernel, 0 = the gather kernel (sn_conv_wgrad_impl: tests, A/B)
SN_EXPORT int sn_conv_wgrad_impl(int impl, int job_steps) {
  SN_REQUIRE((impl == 0 || impl == 1) && job_steps >= 0, "sn_conv_wgrad_impl: impl in {0, 1}, job_steps >= 0");
  SN_REQUIRE(job_steps / 100000 <= 2, "sn_conv_wgrad_impl: bad tile selector");
  g_wgrad_impl.store(impl, std::memory_order_relaxed);
  g_wgrad_na.store(job_steps / 100000, std::memory_order_relaxed);   // tuning only: 100000 / 200000 + steps forces 128- / 256-row tiles
  wgrad_ps_set_job_steps(job_steps % 100000);
  return SN_OK;
}

// diagnostics: phase timeline of the LDS-DMA forward / data-gradient workgroups into buf (tools/conv_trace.py); NULL = off
SN_EXPORT int sn_conv_trace(void *buf) {
  conv_dma_set_trace(static_cast<unsigned long long *>(buf));
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
  int nc = 0, na = 1;
  auto flush = [&]() -> int {
    if (!nc) return SN_OK;
    WgradBatch tab;
    char *base = ws ? static_cast<char *>(ws) + off : nullptr;
    size_t bytes = wgrad_ps_plan(chunk, nc, tab, base, true, na);
    if (launch) {
      // too little scratch (or none): the chunk runs unsplit -- slower, same result modulo summation order, never atomics
      if (bytes && !(ws && off + bytes <= ws_bytes && ((uintptr_t)base % 16) == 0)) bytes = wgrad_ps_plan(chunk, nc, tab, nullptr, false, na);
      if (int rc = wgrad_ps_launch(tab, s, na)) return rc;
    }
    off += bytes;
    nc = 0;
    return SN_OK;
  };
  for (int i = 0; i < n; ++i) {
    WgradParams p;
    SN_REQUIRE(wgrad_desc_params(descs[i], p), "sn_conv_wgrad_batch: bad dims in problem %d", i);
    if (launch) SN_REQUIRE(descs[i].dy && descs[i].x && descs[i].dw, "sn_conv_wgrad_batch: null pointer in problem %d", i);
  }
  // Two passes over the table: the wide layers (Cout > 128) on 256 x 128 tiles with eight consumer waves, the narrow ones on
  // 128 x 128 tiles with four (a 256-row tile of a 128-channel layer would multiply zeros in half of its waves).
  // chunks: <= 24 problems (the table is a kernel argument), cut where the job count fills whole rounds of the 256 CUs -- a chunk
  // of 528 equal jobs runs as long as one of 768; one of 492 runs like 512
  auto fill = [](long j) { return j <= 0 ? 0.0 : (double)j / (256.0 * (double)((j + 255) / 256)); };
  for (int pass = 2; pass >= 1; --pass) {
    na = pass;
    long jobs = 0;
    for (int i = 0; i < n; ++i) {
      const sn_wgrad_desc &d = descs[i];
      WgradParams p;
      wgrad_desc_params(d, p);
      if (!(g_wgrad_impl.load(std::memory_order_relaxed) == 1 && wgrad_ps_ok(p))) continue;
      const int na_forced = g_wgrad_na.load(std::memory_order_relaxed);
      const int want = na_forced ? na_forced : (p.Cout > 128 ? 2 : 1);
      if (want != pass) continue;
      const bool flat = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
      const long steps = flat ? sn_div_up(sn_div_up(p.N * p.H * p.W, 32), 2) : sn_div_up(p.N * p.Ho * sn_div_up(p.Wo, 32), 2);
      const long j = (long)sn_div_up(p.Cout, 128 * na) * sn_div_up(p.Cin, 128) * p.KH * p.KW * ((steps + 399) / 400);
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
    }
    if (int rc = flush()) return rc;
  }
  for (int i = 0; i < n; ++i) {      // what the batched kernel cannot take: the gather kernel, one launch per layer
    const sn_wgrad_desc &d = descs[i];
    WgradParams p;
    wgrad_desc_params(d, p);
    if (g_wgrad_impl.load(std::memory_order_relaxed) == 1 && wgrad_ps_ok(p)) continue;
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
