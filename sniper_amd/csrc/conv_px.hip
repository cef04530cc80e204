// conv_px.hip -- "pixel-stationary" 1 x 1 convolution for SHORT contractions (Cin = 128 .. 256): the bottleneck expansions
// 256 -> 1024 / 128 -> 512 of resnetc4 (symbols/faster/resnet_mx_101_e2e.py:43-66, conv3 of every residual unit) forward, and the
// data gradients of the reductions 1024 -> 256 / 512 -> 128 (the same GEMM on the transposed weights).
//
// Why another kernel.  conv_dma_kernel gives such a layer to 160 x 128 tiles of a 2 .. 4 step contraction: a workgroup fills its
// pipeline (one HBM latency), runs four K-steps each of which waits for the next stage, reduces its statistics and stores -- 16 us
// of residency for 1.7 us of MFMA work, two co-resident workgroups per CU, 4 rounds: 36 us for a layer whose HBM floor is 9 us
// and whose matrix floor is 5 us.  Here a workgroup lives for the whole row tile:
//
//   * the PIXEL operand is stationary in registers.  A wave owns 80 pixels x the whole contraction: 5 row fragments x Cin / 32
//     K-steps x 4 VGPRs = 160 VGPRs at Cin = 256, loaded from HBM once, straight into registers (no LDS round trip).
//   * the WEIGHTS stream through LDS by LDS-DMA in 128-channel chunks (128 rows x Cin, 64 KB at Cin = 256), double buffered; they
//     are L2 resident (512 KB per layer).  Eight waves = 2 pixel halves x 4 channel groups; per chunk a wave reads 2 weight
//     fragments per K-step for 10 MFMAs: 64 KB of fragment reads + 64 KB of DMA per 2560 MFMA cycles -- the LDS is at 40 % where
//     the 160 x 128 K loop needs 115 % (the LDS-bandwidth bound of DESIGN 11.6 does not apply).
//   * one barrier per 128-channel chunk instead of one per 64-deep K-step; the stores of chunk c drain under chunk c + 1.
//
// Arithmetic is that of conv_dma_kernel to the bit: the same v_mfma_f32_16x16x32_f16 sequence in the same K order (weights as the A
// operand, the same weight-row permutation, so a lane holds 8 consecutive channels of one pixel), the same epilogue expressions,
// and the BatchNorm partials in the same association (5 row fragments per lane, 16 lanes by DPP, the two pixel halves in wave
// order) per 160-row tile -- the layer's `blocks` (sn_conv_fwd_stats_blocks / sn_conv_dgrad_bn_blocks) do not change.
#include "conv_common.h"
#include <algorithm>

namespace {

typedef __attribute__((address_space(3))) void *lds_ptr_t;

__device__ __forceinline__ void px_dma16(__amdgpu_buffer_rsrc_t rsrc, half_t *dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}
template <int CTRL>
__device__ __forceinline__ float px_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes that share lane >> 4, in the order conv_dma.hip's row16_sum adds them
__device__ __forceinline__ float px_row16_sum(float v) {
  v = px_dpp_add<0xB1>(v);
  v = px_dpp_add<0x4E>(v);
  v = px_dpp_add<0x141>(v);
  return px_dpp_add<0x140>(v);
}

constexpr int kPxBM = 160;        // pixels per workgroup (two wave halves of 80)
constexpr int kPxChunk = 128;     // output channels per chunk (4 channel groups of 32)
constexpr int kPxMaxChunks = 4;   // chunks per workgroup: up to 512 output channels

// KC = Cin / 64.  LDS: two weight stages of KC x [128 rows][64 channels] fp16 (KC x 16 KB each) + the statistics exchange.
template <int KC, bool BNX>
__global__ __launch_bounds__(512, 1) void conv_px_kernel(const ConvParams p, int mtiles, int ntiles, int nchunks) {
  constexpr int BK = 64, MI = kPxBM / 2 / 16, KS = 2 * KC;
  constexpr int BLK = kPxChunk * BK;              // half_t elements of one 64-channel block of a stage
  constexpr int STAGE = KC * BLK;
  constexpr int PIECES = KC * 16 / 8;             // 1 KB DMA pieces per wave per stage
  constexpr int RED = kPxMaxChunks * 2 * 2 * kPxChunk;   // floats: [chunk][pixel half][sum | second moment][channel]
  __shared__ __attribute__((aligned(1024))) half_t lds[2 * STAGE + RED * 2];
  float *const red = reinterpret_cast<float *>(lds + 2 * STAGE);

  const int lin = blockIdx.x;
  const int xcd = lin & 7, j = lin >> 3;
  const int nt = j % ntiles, mt = (j / ntiles) * 8 + xcd;      // the column tiles of one row tile run on ONE XCD (they share the pixels)
  if (mt >= mtiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = mt * kPxBM, n0 = nt * nchunks * kPxChunk;

  // ---- weight stage `c` -> buffer: wave w moves the 8-row groups w, w + 8, ... of the KC x 16 groups (block kb = g / 16, rows 8 (g % 16) ..)
  const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
  const unsigned wrow_bytes = (unsigned)p.Cin * 2u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(p.w), 0, (int)p.w_bytes, 0x00020000);
  auto issue = [&](int c, int buf) {
    half_t *const sb = lds + buf * STAGE;
    const int nc = n0 + c * kPxChunk;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int g = wave + 8 * i, kb = g >> 4, rg = g & 15;
      // LDS row r holds channel nc + perm(r) (conv_dma.hip: a lane's two accumulators of a fragment pair are 8 consecutive channels)
      const int r = 8 * rg + lrow;
      const int n = nc + (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3);
      px_dma16(rw, sb + kb * BLK + rg * 512, (unsigned)n * wrow_bytes + (unsigned)kb * 128u + (unsigned)gchunk * 16u);
    }
  };
  issue(0, 0);

  // ---- this wave's pixels, the whole contraction: fa[i][ks] = pixel m0 + 80 wm + 16 i + fr, channels 32 ks + 8 fq .. + 7
  half8 fa[MI][KS];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    // rows beyond M read row 0: their accumulators are never stored and never enter the statistics
    const int mr = m0 + wm * (kPxBM / 2) + i * 16 + fr;
    const half_t *src = p.x + (size_t)(mr < p.M ? mr : 0) * p.in_ps + fq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fa[i][ks] = *reinterpret_cast<const half8 *>(src + ks * 32);
  }
  // The compiler must see these loads RETIRE here: its scoreboard does not read an inline-asm s_waitcnt, and a load it still
  // believes pending makes it drain the LDS-DMA queue (vmcnt(0)) in front of the first MFMA of EVERY chunk
  // (cdna_hip_programming.md, trap (b)).  An empty asm that "uses" each fragment puts the compiler's own wait here.
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa[i][ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (and this wave's pieces of weight stage 0)
  __builtin_amdgcn_s_barrier();

  const int sw = fq ^ (fr & 7);
  const int b_rd = (wn * 32 + fr) * BK;
  const bool has_stats = p.stats != nullptr;

  for (int c = 0; c < nchunks; ++c) {
    // buffer (c + 1) & 1 was read by chunk c - 1, which every wave finished before the barrier that closed it
    if (c + 1 < nchunks) issue(c + 1, (c + 1) & 1);
    floatx4 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i][0] = acc[i][1] = floatx4{0.f, 0.f, 0.f, 0.f};
    const half_t *const sb = lds + (c & 1) * STAGE;
    // weight fragments one K-step ahead, and no further: left alone hipcc hoists all 2 KS reads (64 VGPRs) above the first MFMA and
    // spills the stationary pixels to make room
    auto rd = [&](int ks, int half) {
      return *reinterpret_cast<const half8 *>(sb + (ks >> 1) * BLK + b_rd + half * 16 * BK + (sw ^ ((ks & 1) * 4)) * 8);
    };
    half8 nb0 = rd(0, 0), nb1 = rd(0, 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8 fb0 = nb0, fb1 = nb1;
      if (ks + 1 < KS) { nb0 = rd(ks + 1, 0); nb1 = rd(ks + 1, 1); }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb0, fa[i][ks], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb1, fa[i][ks], acc[i][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next stage (issued a chunk's MFMAs ago) and the stores of the previous chunk: landed / drained before this chunk's stores go out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue of the chunk: lane (fr, fq) holds pixel .. + fr, channels n .. n + 7 (conv_dma.hip, the 16-byte path).  Addresses are
    // a wave-uniform 64-bit base + one 32-bit lane offset per tensor (conv_px_ok bounds the tensors to 2 GB): the pixel operand
    // leaves no room for a 64-bit pointer per row fragment
    const int n = n0 + c * kPxChunk + wn * 32 + fq * 8;
    const int mrow = m0 + wm * (kPxBM / 2) + fr;
    const unsigned y_off = ((unsigned)mrow * (unsigned)p.out_ps + (unsigned)n) * 2u;
    const unsigned r_off = ((unsigned)mrow * (unsigned)p.res_ps + (unsigned)n) * 2u;
    const unsigned x_off = ((unsigned)mrow * (unsigned)p.bn_x_ps + (unsigned)n) * 2u;
    float st_s[8], st_q[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) st_s[r] = st_q[r] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (mrow + i * 16 >= p.M) continue;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[i][0][r]; v[4 + r] = acc[i][1][r]; }
      if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (p.res) {
        const half8 rv = *reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(p.res) + (size_t)(i * 16) * p.res_ps * 2 + r_off);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      half8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (half_t)v[r];
      *reinterpret_cast<half8 *>(reinterpret_cast<char *>(p.y) + (size_t)(i * 16) * p.out_ps * 2 + y_off) = o;
      if (has_stats) {
        if constexpr (BNX) {
          const half8 xv = *reinterpret_cast<const half8 *>(reinterpret_cast<const char *>(p.bn_x) + (size_t)(i * 16) * p.bn_x_ps * 2 + x_off);
          // the per-channel constants four channels at a time (L1 hits): twelve registers instead of twenty-four held across the chunk
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 sc = *reinterpret_cast<const float4 *>(p.bn_scale + n + 4 * h), sh = *reinterpret_cast<const float4 *>(p.bn_shift + n + 4 * h);
            const float4 mu = *reinterpret_cast<const float4 *>(p.bn_mean + n + 4 * h);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float xf = (float)xv[4 * h + r], yv = xf * scv[r] + shv[r];
              const bool pass = p.bn_act == 0 || (p.bn_act == 1 ? yv > 0.f : (yv >= 0.f && yv <= 6.f));
              const float gf = pass ? (float)o[4 * h + r] : 0.f;
              st_s[4 * h + r] += gf;
              st_q[4 * h + r] += gf * (xf - muv[r]);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float f = (float)o[r];
            st_s[r] += f;
            st_q[r] += f * f;
          }
        }
      }
    }
    if (has_stats) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        st_s[r] = px_row16_sum(st_s[r]);
        st_q[r] = px_row16_sum(st_q[r]);
      }
      if (fr == 0) {
        float *const rr = red + ((c * 2 + wm) * 2) * kPxChunk + wn * 32 + fq * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          rr[r] = st_s[r];
          rr[kPxChunk + r] = st_q[r];
        }
      }
    }
    // every wave's pieces of the next stage have landed (its vmcnt(0) above), and every wave is done reading this chunk's buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  if (has_stats) {
    // the two pixel halves in wave order (conv_dma.hip sums its WMW = 2 wave rows the same way): deterministic
    for (int idx = tid; idx < nchunks * 2 * kPxChunk; idx += 512) {
      const int c = idx / (2 * kPxChunk), rem = idx - c * 2 * kPxChunk;
      const int which = rem / kPxChunk, col = rem - which * kPxChunk;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 2; ++w) a += red[((c * 2 + w) * 2 + which) * kPxChunk + col];
      p.stats[((size_t)mt * 2 + which) * p.Nout + n0 + c * kPxChunk + col] = a;
    }
  }
}

}  // namespace

// Does the layer qualify?  (1 x 1, unit stride, no padding; Cin = 128 / 192 / 256; whole 128-channel chunks; 16-byte rows; fp16 output.)
bool conv_px_ok(const ConvParams &p) {
  if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.H != p.Ho || p.W != p.Wo) return false;
  if (p.Cin % 64 != 0 || p.Cin < 128 || p.Cin > 256 || p.in_ps % 8 != 0) return false;
  if (p.Nout % kPxChunk != 0 || (p.Nout > kPxMaxChunks * kPxChunk && p.Nout % (kPxMaxChunks * kPxChunk) != 0)) return false;
  if (p.out_f32 || p.out_ps % 8 != 0 || (p.res && p.res_ps % 8 != 0) || p.out2 || p.ksplit > 1 || p.cls) return false;
  if (p.bn_x && (p.bn_x_ps % 8 != 0 || !p.stats)) return false;
  const size_t widest = (size_t)std::max(std::max(p.out_ps, p.res ? p.res_ps : 0), p.bn_x ? p.bn_x_ps : 0);
  if ((size_t)(p.M + kPxBM) * widest * 2 >= ((size_t)1 << 31)) return false;      // 32-bit lane offsets in the epilogue
  if (p.M < 4096) return false;            // a handful of row tiles: the tile kernels' more numerous workgroups fill the chip better
  return true;
}

int conv_px_launch(const ConvParams &p, hipStream_t s) {
  const int mtiles = sn_div_up(p.M, kPxBM);
  const int nchunks = p.Nout > kPxMaxChunks * kPxChunk ? kPxMaxChunks : p.Nout / kPxChunk;
  const int ntiles = p.Nout / (nchunks * kPxChunk);
  const dim3 grid((unsigned)(sn_div_up(mtiles, 8) * 8 * ntiles));
  const bool bnx = p.bn_x != nullptr;
#define SN_PX_LAUNCH(KC)                                                                                       \
  do {                                                                                                         \
    if (bnx) hipLaunchKernelGGL((conv_px_kernel<KC, true>), grid, dim3(512), 0, s, p, mtiles, ntiles, nchunks);  \
    else hipLaunchKernelGGL((conv_px_kernel<KC, false>), grid, dim3(512), 0, s, p, mtiles, ntiles, nchunks);     \
  } while (0)
  switch (p.Cin / 64) {
    case 2: SN_PX_LAUNCH(2); break;
    case 3: SN_PX_LAUNCH(3); break;
    case 4: SN_PX_LAUNCH(4); break;
    default: SN_REQUIRE(false, "conv_px_launch: Cin = %d", p.Cin);
  }
#undef SN_PX_LAUNCH
  SN_CHECK_LAUNCH();
  return SN_OK;
}
