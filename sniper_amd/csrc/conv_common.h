// conv_common.h -- types shared by the convolution translation units (conv.hip: register-staged kernels, weight gradient,
// C-ABI entry points; conv_dma.hip: LDS-DMA pipelined implicit GEMM).  Internal, not part of the C ABI.
#pragma once
#include <atomic>
#include "common.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  const half_t *x;  // source activations
  const half_t *w;  // [Nout][taps][Cin] fp16
  void *y;          // destination (fp16 or fp32)
  const float *bias;   // [Nout] or null
  const half_t *res;   // residual added in the epilogue (fp16, pixel stride res_ps) or null
  int N, H, W;         // source dims
  int Cin, in_ps;      // K per tap, source pixel stride (elements)
  int Ho, Wo;          // destination spatial dims
  int Nout, out_ps, res_ps;
  int KH, KW, stride, pad, dil;
  int M;               // N*Ho*Wo
  int relu, out_f32;
  unsigned x_bytes, w_bytes;  // addressable extent of x / w (buffer-descriptor bounds of the pipelined kernel)
  // optional BatchNorm BACKWARD statistics of a data gradient (dgrad output = dL/d(act(BN(bn_x)))): with bn_x set, `stats`
  // receives per row tile [mt][2][Nout] = sum g, sum g * (bn_x - mean), g = stored dx masked by the fused activation
  const half_t *bn_x = nullptr;
  const float *bn_scale = nullptr, *bn_shift = nullptr, *bn_mean = nullptr;
  int bn_x_ps = 0, bn_act = 0;
  // Row index -> (image, y, x) without integer division in the kernel (conv_dma.hip): q = (umulhi(n, mul) + n) >> sh for n < 2^31,
  // by rows per image (fda) and by row length (fdb); filled by conv_launch (conv_fastdiv_fill).
  unsigned fda_mul = 1, fda_sh = 0, fdb_mul = 1, fdb_sh = 0;
  int rows_img = 1, row_len = 1;
  // Data gradient of a stride-2 convolution by PARITY CLASS (cls = 1, conv_dma.hip): a destination pixel (y, x) only receives the
  // taps with (y + pad - kh) and (x + pad - kw) even, so the rows of the GEMM are enumerated class by class ((y & 1, x & 1) =
  // 00, 01, 10, 11; cls_mc = N * Ho/2 * Wo/2 rows each, row tiles never straddle classes) and every class walks only ITS taps:
  // 9 tap visits over the four classes of a 3 x 3 kernel instead of 36, 1 instead of 4 for a 1 x 1 shortcut.
  int cls = 0, cls_mc = 0;
  // Split-K forward (sn_conv_fwd_splitk; launches with far fewer output tiles than CUs -- the 2-chip batches of the finest test
  // scale are 82 tiles of a 36-step contraction, one latency-bound workgroup each): ksplit > 1 -> the grid is ksplit copies of the
  // tile grid (ksplit_grid blocks each), copy z walks K-steps [z nk / ksplit, (z + 1) nk / ksplit) and writes its fp32 partial tile
  // into slab z of `y` (ksplit_stride elements apart); splitk_reduce_kernel adds the slabs in order (+ bias, residual, ReLU).
  int ksplit = 1, ksplit_grid = 0;
  long ksplit_stride = 0;
  unsigned long long *trace = nullptr;   // phase timeline of every workgroup (tools/conv_trace.py; SNIPER_CONV_TRACE), normally null
  int probe = 0;          // TIMING PROBE (only with `trace`; SNIPER_CONV_PROBE_SKIP_A): bit 0 -- the specialised producers skip the A pieces of every tap but the first (what a halo-resident A operand would leave of the K loop); bit 1 -- the weight pieces of every K-step but the first (3: a K loop without operand traffic); bit 2 -- no step barrier (the loop without its rendezvous).  Wrong results
  // optional SECOND output of a forward convolution (sn_conv_fwd_dual, test-time graphs): y2 = act(scale * y + shift) of the stored
  // fp16 output -- the moving-statistics BatchNorm (+ ReLU) of the NEXT residual unit, which reads the residual sum this
  // convolution's epilogue writes and cannot fold into it (the sum has a second reader, the next add)
  half_t *out2 = nullptr;
  const float *o2_scale = nullptr, *o2_shift = nullptr;
  int out2_ps = 0, o2_relu = 0;
  int tiles_per_wg = 1;    // persistent configurations (24, 26): output tiles a workgroup walks (conv_dma.hip PERSIST; set by the launcher)
  float *stats = nullptr;  // optional BatchNorm statistics of the output: per row tile [mt][2][Nout] = sum, sum of squares of the
                       // STORED fp16 values (what bn_stats_kernel would read back), or null
};


static inline void conv_fastdiv_make(unsigned d, unsigned &mul, unsigned &sh) {
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << s) - d)) / d + 1);
  sh = s;
}
static inline void conv_fastdiv_fill(ConvParams &p) {
  p.rows_img = p.cls ? (p.Ho / 2) * (p.Wo / 2) : p.Ho * p.Wo;
  p.row_len = p.cls ? p.Wo / 2 : p.Wo;
  conv_fastdiv_make((unsigned)p.rows_img, p.fda_mul, p.fda_sh);
  conv_fastdiv_make((unsigned)p.row_len, p.fdb_mul, p.fdb_sh);
}
__device__ __forceinline__ int conv_fastdiv(int n, unsigned mul, unsigned sh) { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> sh); }

// LDS-DMA pipelined implicit-GEMM kernels (conv_dma.hip).  cfg: see conv_dma_config().
struct ConvDmaConfig { int bm, bn, threads, stages, lds_bytes; };
// persistent twins of the 160 x 128 configurations (24 of 14, 26 of 16): a launch qualifies when its tiles divide over the resident
// workgroups -- whole tiles only (M % bm == 0, Nout % bn == 0, row tiles a multiple of 8), no parity classes, no split
constexpr int kConvPersistWgs = 512;      // 2 workgroups per CU x 256 CUs
static inline int conv_persist_tiles_per_wg(int M, int Nout, int bm, int bn) {
  if (M % bm || Nout % bn) return 0;
  const long mt = M / bm, nt = Nout / bn, tiles = mt * nt;
  if (mt % 8 || tiles < 2 * kConvPersistWgs || tiles % kConvPersistWgs) return 0;
  return (int)(tiles / kConvPersistWgs);
}
constexpr int kConvDmaConfigs = 26;    // highest configuration number (the table has holes: conv_dma_config(c).bm == 0)
ConvDmaConfig conv_dma_config(int cfg);
int conv_dma_launch(const ConvParams &p, bool dgrad, int cfg, hipStream_t s);
void conv_dma_set_trace(unsigned long long *buf);

// ---- weight gradient (conv.hip: gather kernel for unaligned operands and the packed stem; conv_wgrad_ps.hip: everything else)
struct WgradParams {
  const half_t *dy;  // (N, Ho, Wo, Cout) pixel stride dy_ps
  const half_t *x;   // (N, H, W, Cin)    pixel stride x_ps
  float *dw;         // [Cout][taps][Cin] fp32, accumulated into
  int N, H, W, Ho, Wo, Cin, Cout, dy_ps, x_ps;
  int KH, KW, stride, pad, dil;
  int units_per_split;  // 32-pixel K chunks handled per blockIdx.z split
  float *slab;          // split-K partials [split][Cout][taps][Cin] (plain stores, reduced by wgrad_reduce_kernel) or null
  size_t slab_stride;   // elements per split
  // logical grid (tile columns, tile rows, taps x splits).  The launch is one-dimensional and XCD-aware: hardware block b runs
  // on XCD b % 8, and every XCD is handed a CONTIGUOUS range of the split-major item list, so the tiles that share one
  // K-split's dY / X panels fetch them through ONE L2 (wgrad_block() below)
  int gx, gy, gz, per_xcd;
};

// logical block index of a weight-gradient workgroup; false = surplus block of the rounded-up launch
__device__ __forceinline__ bool wgrad_block(const WgradParams &p, int &bx, int &by, int &bz) {
  const int b = blockIdx.x;
  const int item = (b & 7) * p.per_xcd + (b >> 3);
  const int tiles = p.gx * p.gy;
  if (item >= tiles * p.gz) return false;
  bz = item / tiles;
  const int r = item - bz * tiles;
  by = r / p.gx;
  bx = r - by * p.gx;
  return true;
}
static inline unsigned wgrad_grid(WgradParams &p, int gx, int gy, int gz) {
  p.gx = gx; p.gy = gy; p.gz = gz;
  p.per_xcd = (gx * gy * gz + 7) / 8;
  return 8u * (unsigned)p.per_xcd;
}

// ---- wave-specialised, batched weight gradient (conv_wgrad_ps.hip) ----
struct WgradProblem {
  const half_t *dy, *x;
  float *dw, *slab;          // slab: split-K partials [split][Cout][taps][Cin] of this problem, or null (unsplit: dw += tile)
  size_t slab_stride;        // Cout * taps * Cin
  int N, H, W, Ho, Wo, Cin, Cout, dy_ps, x_ps, KH, KW, stride, pad, dil;   // a flat problem (1x1 / s1 / p0, FC) is one row of N*H*W pixels
  int gx, gy, taps, splits, units_per_split, cpr, nunits, item0;            // jobs [split][tap][ci tile][co tile] start at item0
};
constexpr int kWgradMaxProblems = 24;     // 24 x 128 B + header < the 4 KB kernel-argument segment
struct WgradBatch {
  int n, total_items, per_xcd, reserved;
  int xcd_start[9];            // XCD x runs jobs [xcd_start[x], xcd_start[x + 1]): contiguous, about equal K-steps each
  int reserved2;
  unsigned long long *trace;   // phase timeline of every job (diagnostics), normally null
  WgradProblem p[kWgradMaxProblems];
};
bool wgrad_ps_ok(const WgradParams &p);                                           // operands 16-byte addressable, < 1 GB each
// na = 128-channel dY sub-tiles per job (1: 128 x 128 output tiles, 2: 256 x 128); every problem of a table uses the same
size_t wgrad_ps_plan(const WgradParams *ps, int n, WgradBatch &tab, void *ws, bool allow_split, int na);   // -> scratch bytes of the split problems
int wgrad_ps_launch(const WgradBatch &tab, hipStream_t s, int na);
void wgrad_ps_set_job_steps(int steps);
void wgrad_ps_set_trace(unsigned long long *buf);


