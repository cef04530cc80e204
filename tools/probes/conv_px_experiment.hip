// EXPERIMENT, NOT PART OF THE SHIPPED LIBRARY (round 5; the library is sniper_amd/csrc/*.hip only).  A pixel-stationary 1 x 1 convolution
// for short contractions; tools/probes/conv_px_build.sh builds sniper_amd/lib/libsniper_hip_px.so from a scratch copy of csrc + this
// file + the hook of conv_px_hook.patch, tools/probes/conv_px_trace.py measures it.  Bit-identical to the 160 x 128 tile kernel (outputs
// and BatchNorm partials) on every case tried.  profiles/r05_conv_px_experiment.txt:
//   first version (every wave loads its 80 pixels itself; commits 8661762 - 190963f): 28.3 us against the tile kernel's 28.5 on the
//     stage-3 expansion, no gain in the timed step -- pixel loads 8.7 us (the four channel groups of a workgroup load the same pixels:
//     320 KB through one CU's load path for 80 KB of data), 8.7 us of launch / first weight stage / barriers, statistics 4.1,
//     stores 3.1, MFMAs 2.8, all serialised;
//   this version (the pixel tile ONCE per workgroup by LDS-DMA, fragments copied LDS -> registers): 256 -> 1024 forward 28.6 -> 20.9 us
//     (1.37x), 128 -> 512 forward 50.5 -> 36.1 (1.40x), the 1024 <- 256 data gradient with the fused BatchNorm-backward reduction
//     42.7 -> 40.1 (its epilogue, not its prologue, is the cost).  In the timed step (tools/probes/conv_px_step_ab.sh): 945.6 -> 949.3
//     chips/s with the forward layers on it, 0.4 % -- the tile kernel's operands come warm from the producing kernel there.  It stays
//     an experiment.
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
typedef unsigned int px_u4 __attribute__((ext_vector_type(4)));

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
template <int KC, bool BNX, bool STAG>
__global__ __launch_bounds__(512, 1) void conv_px_kernel(const ConvParams p, int mtiles, int ntiles, int nchunks, int dbg) {
  constexpr int BK = 64, MI = kPxBM / 2 / 16, KS = 2 * KC;
  constexpr int BLK = kPxChunk * BK;              // half_t elements of one 64-channel block of a stage
  constexpr int STAGE = KC * BLK;
  constexpr int PIECES = KC * 16 / 8;             // 1 KB DMA pieces per wave per stage
  constexpr int RED = kPxMaxChunks * 2 * 2 * kPxChunk;   // floats: [chunk][pixel half][sum | second moment][channel]
  constexpr int CST = BNX ? 3 * kPxMaxChunks * kPxChunk : 0;   // floats: BatchNorm scale | shift | mean of this workgroup's channels
  // the pixel tile (160 rows x Cin: KC blocks of [160 rows][64 channels], 20 KB each) passes through LDS ONCE per workgroup -- it lands
  // in weight buffer 1 and the 4 KC KB behind it while weight stage 0 lands in buffer 0; every wave then copies ITS fragments to
  // registers, and buffer 1 takes weight stage 1.  (First version: the four channel groups each loaded the same 80 pixels from
  // global memory, 320 KB through the CU's load path for 80 KB of data: 8.7 us of a 28 us launch.)
  constexpr int ABLK = kPxBM * BK;                 // half_t elements of one 64-channel block of the pixel tile
  constexpr int AEXTRA = KC * ABLK - STAGE;        // what the tile needs beyond weight buffer 1
  __shared__ __attribute__((aligned(1024))) half_t lds[2 * STAGE + AEXTRA + (RED + CST) * 2];
  float *const red = reinterpret_cast<float *>(lds + 2 * STAGE + AEXTRA);
  float *const cst = red + RED;

  const int lin = blockIdx.x;
  const int xcd = lin & 7, j = lin >> 3;
  const int nt = j % ntiles, mt = (j / ntiles) * 8 + xcd;      // the column tiles of one row tile run on ONE XCD (they share the pixels)
  if (mt >= mtiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  // diagnostics (tools/conv_px_trace.py): shader-clock stamps of wave 0 -- [0] entry, [1] pixels + first weight stage landed,
  // [2 + 2c] chunk c multiplied, [3 + 2c] chunk c stored, [10] exit; dbg bits switch PARTS of the kernel off (wrong results,
  // timing only): 1 output stores, 2 weight DMA after the first stage, 4 pixel loads, 8 MFMAs, 16 statistics
  auto stamp = [&](int k) {
    if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 16 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = mt * kPxBM, n0 = nt * nchunks * kPxChunk;

  // ---- weight stage `c` -> buffer: wave w moves the 8-row groups w, w + 8, ... of the KC x 16 groups (block kb = g / 16, rows 8 (g % 16) ..)
  // LDS row r of a block holds channel nc + perm(r), perm(r) = (r & ~31) + ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3) (conv_dma.hip: a
  // lane's two accumulators of a fragment pair are then 8 consecutive channels).  With r = 8 rg + lrow the lane's share of the source
  // address, 8 (lrow >> 2) + (lrow & 3) rows and its 16-byte chunk, is ONE register for the whole kernel; the group's share goes
  // in the instruction's scalar offset.
  const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;
  const unsigned wrow_bytes = (unsigned)p.Cin * 2u;
  const unsigned w_lane = (unsigned)(8 * (lrow >> 2) + (lrow & 3)) * wrow_bytes + (unsigned)gchunk * 16u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(p.w), 0, (int)p.w_bytes, 0x00020000);
  auto issue = [&](int c, int buf) {
    half_t *const sb = lds + buf * STAGE;
    const int nc = n0 + c * kPxChunk;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int g = wave + 8 * i, kb = g >> 4, rg = g & 15;
      const int nu = nc + 32 * (rg >> 2) + 16 * (rg & 1) + 4 * ((rg >> 1) & 1);       // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(sb + kb * BLK + rg * 512), 16, w_lane,
                                               (int)((unsigned)nu * wrow_bytes + (unsigned)kb * 128u), 0, 0);
    }
  };
  issue(0, 0);
  if constexpr (BNX) {
    // the per-channel constants of the fused BatchNorm-backward reduction, once per workgroup: read back from LDS in every chunk's
    // epilogue, where a global load would have to wait behind the chunk's stores (no alias information) -- ten exposed round trips
    for (int idx = tid; idx < nchunks * kPxChunk; idx += 512) {
      cst[idx] = p.bn_scale[n0 + idx];
      cst[kPxMaxChunks * kPxChunk + idx] = p.bn_shift[n0 + idx];
      cst[2 * kPxMaxChunks * kPxChunk + idx] = p.bn_mean[n0 + idx];
    }
  }

  // ---- this wave's pixels, the whole contraction: fa[i][ks] = pixel m0 + 80 wm + 16 i + fr, channels 32 ks + 8 fq .. + 7
  // Tensors are addressed through buffer descriptors: ONE 32-bit lane offset per tensor + a wave-uniform scalar offset per row
  // fragment (the stationary pixels leave no registers for 64-bit addresses per fragment), and rows beyond M read zeros / are not
  // written by the descriptor's bound (conv_px_ok keeps every tensor under 2 GB).
  const unsigned kFlags = 0x00020000;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(p.x), 0, (int)p.x_bytes, kFlags);
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(((unsigned)(p.M - 1) * (unsigned)p.out_ps + (unsigned)p.Nout) * 2u), kFlags);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t *>(p.res), 0, p.res ? (int)(((unsigned)(p.M - 1) * (unsigned)p.res_ps + (unsigned)p.Nout) * 2u) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t *>(p.bn_x), 0, BNX ? (int)(((unsigned)(p.M - 1) * (unsigned)p.bn_x_ps + (unsigned)p.Nout) * 2u) : 0, kFlags);
  const int mrow = m0 + wm * (kPxBM / 2) + fr;        // this lane's pixel of row fragment 0 (fragment i: + 16 i)

  half8 fa[MI][KS];
  half_t *const atile = lds + STAGE;
  if (!(dbg & 4)) {
    // 8-row pieces of the tile: piece g = block kb = g / 20, rows 8 (g % 20) .. + 7; lane l supplies row (l >> 3), 16-byte chunk (l & 7) ^ (l >> 3)
    // (the row goes in the LANE offset: a raw buffer's bound check does not see the scalar offset, and rows beyond M must read zeros)
#pragma unroll
    for (int i = 0; i < (KC * 20 + 7) / 8; ++i) {
      const int g = wave + 8 * i;
      if (g < KC * 20) {
        const int kb = g / 20, rg = g - kb * 20;
        const unsigned a_lane = ((unsigned)(m0 + 8 * rg + lrow) * (unsigned)p.in_ps) * 2u + (unsigned)gchunk * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(atile + kb * ABLK + rg * 512), 16, a_lane, kb * 128, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of the tile and of weight stage 0, its constants
  __builtin_amdgcn_s_barrier();
  {
    const int a_rd = (wm * (kPxBM / 2) + fr) * BK, asw = fq ^ (fr & 7);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        fa[i][ks] = (dbg & 4) ? half8{1, 1, 1, 1, 1, 1, 1, 1}
                              : *reinterpret_cast<const half8 *>(atile + (ks >> 1) * ABLK + a_rd + i * 16 * BK + (asw ^ ((ks & 1) * 4)) * 8);
  }
  // every wave holds its fragments before weight stage 1 may overwrite the tile (the empty asm makes the compiler retire the reads here)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(fa[i][ks]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp(1);

  const int sw = fq ^ (fr & 7);
  const int b_rd = (wn * 32 + fr) * BK;
  const bool has_stats = p.stats != nullptr;
  const int a_none = p.bn_act == 0, a_relu = p.bn_act == 1, a_relu6 = p.bn_act == 2;

  floatx4 acc[MI][2];
  // The BatchNorm input the fused backward reduction reads (bn_x: HBM, 16 bytes per lane, row fragment and chunk) is needed AFTER the
  // chunk's stores, where a load cannot be hoisted (no alias information) and five exposed HBM round trips per chunk would be the
  // whole kernel.  Holding it in registers across the MFMAs (20 VGPRs) does not fit beside the stationary pixels -- hipcc spills
  // pixel fragments and reloads them inside the MFMA stream, each reload a vmcnt(0) that also drains the weight DMA.  So the lines
  // are only TOUCHED before the MFMAs (one dword per lane and row fragment: the whole 64-byte segment arrives in this XCD's L2) and
  // read for real in the epilogue, from L2.
  unsigned touch[BNX ? MI : 1];
  auto prefetch = [&](int c) {
    if constexpr (BNX) {
      const int n = n0 + c * kPxChunk + wn * 32 + fq * 8;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = mrow + i * 16 < p.M ? mrow + i * 16 : 0;      // (in the lane offset: the bound check does not see a scalar offset)
        touch[i] = __builtin_amdgcn_raw_buffer_load_b32(rbx, ((unsigned)r * (unsigned)p.bn_x_ps + (unsigned)n) * 2u, 0, 0);
      }
    }
  };
  // (the touches retire where the hand-written vmcnt(0) already stands; the empty asm keeps them from being dropped as dead)
  auto retire_prefetch = [&]() {
    if constexpr (BNX) {
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(touch[i]));
    }
  };
  auto compute = [&](int c) {
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i][0] = acc[i][1] = floatx4{0.f, 0.f, 0.f, 0.f};
    const half_t *const sb = lds + (c & 1) * STAGE;
    // The fragment reads run ONE K-step ahead where the registers allow it (not beside the 160 pixel registers of Cin = 256 in the
    // BatchNorm-backward variant: there the SIMD's other wave covers the LDS latency), and never further: left alone hipcc hoists all
    // 2 KS reads (64 VGPRs) above the first MFMA and spills the stationary pixels to make room.
    constexpr bool kAhead = !(BNX && KC == 4);
    auto rd = [&](int ks, int half) {
      return *reinterpret_cast<const half8 *>(sb + (ks >> 1) * BLK + b_rd + half * 16 * BK + (sw ^ ((ks & 1) * 4)) * 8);
    };
    half8 nb0, nb1;
    if constexpr (kAhead) { nb0 = rd(0, 0); nb1 = rd(0, 1); }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      half8 fb0, fb1;
      if constexpr (kAhead) {
        fb0 = nb0; fb1 = nb1;
        if (ks + 1 < KS) { nb0 = rd(ks + 1, 0); nb1 = rd(ks + 1, 1); }
      } else {
        fb0 = rd(ks, 0); fb1 = rd(ks, 1);
      }
      if (!(dbg & 8)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb0, fa[i][ks], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb1, fa[i][ks], acc[i][1], 0, 0, 0);
        }
      } else {
        acc[0][0][0] += (float)fb0[0] + (float)fb1[0];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto epilogue = [&](int c) {
    // ---- epilogue of the chunk: lane (fr, fq) holds pixel mrow + 16 i, channels n .. n + 7 (conv_dma.hip, the 16-byte path)
    const int n = n0 + c * kPxChunk + wn * 32 + fq * 8;
    const unsigned y_off = ((unsigned)mrow * (unsigned)p.out_ps + (unsigned)n) * 2u;
    const unsigned r_off = ((unsigned)mrow * (unsigned)p.res_ps + (unsigned)n) * 2u;
    // pass 1: convert and store (the 40 accumulator registers become 20 of fp16 output); pass 2: statistics of the STORED values.
    // In one pass the accumulators, the prefetched BatchNorm input, the constants and the sums are all live at once: ~270 VGPRs,
    // and what hipcc then spills is the stationary pixel operand, reloaded in the middle of the MFMA stream.
    half8 o[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[i][0][r]; v[4 + r] = acc[i][1][r]; }
      if (mrow + i * 16 < p.M) {
        if (p.bias) {
          const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.res) {
          const half8 rv = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rres, r_off, i * 16 * p.res_ps * 2, 0));
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
        }
        if (p.relu) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) o[i][r] = (half_t)v[r];
      // (guarded: a raw buffer's bound check does not include the scalar offset)
      if (!(dbg & 1) && mrow + i * 16 < p.M) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(px_u4, o[i]), ry, y_off, i * 16 * p.out_ps * 2, 0);
    }
    // pass 2, four channels at a time (twelve constant registers and eight sums live, not twenty-four and sixteen)
    half8 xv[BNX ? MI : 1];
    if constexpr (BNX) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = mrow + i * 16 < p.M ? mrow + i * 16 : 0;
        xv[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rbx, ((unsigned)r * (unsigned)p.bn_x_ps + (unsigned)n) * 2u, 0, 0));
      }
    }
    if (has_stats && !(dbg & 16)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        __builtin_amdgcn_sched_barrier(0);      // (one half's constants and conversions must not be hoisted into the other's)
        float st_s[4], st_q[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) st_s[r] = st_q[r] = 0.f;
        float scv[BNX ? 4 : 1], shv[BNX ? 4 : 1], muv[BNX ? 4 : 1];
        if constexpr (BNX) {
          const float *const cc = cst + (n - n0) + 4 * h;      // the per-channel constants from LDS (filled once per workgroup)
          const float4 sc = *reinterpret_cast<const float4 *>(cc), sh = *reinterpret_cast<const float4 *>(cc + kPxMaxChunks * kPxChunk);
          const float4 mu = *reinterpret_cast<const float4 *>(cc + 2 * kPxMaxChunks * kPxChunk);
          scv[0] = sc.x; scv[1] = sc.y; scv[2] = sc.z; scv[3] = sc.w;
          shv[0] = sh.x; shv[1] = sh.y; shv[2] = sh.z; shv[3] = sh.w;
          muv[0] = mu.x; muv[1] = mu.y; muv[2] = mu.z; muv[3] = mu.w;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if (mrow + i * 16 >= p.M) continue;
          if constexpr (BNX) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float xf = (float)xv[i][4 * h + r], yv = xf * scv[r] + shv[r];
              // bn_act_pass (nn_ops.hip: 0 none, 1 ReLU y > 0, 2 ReLU6 0 <= y <= 6) as mask arithmetic: with || and && hipcc builds a
              // divergent exec-mask ladder per element (25 instructions)
              const int pass = a_none | (a_relu & (int)(yv > 0.f)) | (a_relu6 & (int)(yv >= 0.f) & (int)(yv <= 6.f));
              const float gf = pass ? (float)o[i][4 * h + r] : 0.f;
              st_s[r] += gf;
              st_q[r] += gf * (xf - muv[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float f = (float)o[i][4 * h + r];
              st_s[r] += f;
              st_q[r] += f * f;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st_s[r] = px_row16_sum(st_s[r]);
          st_q[r] = px_row16_sum(st_q[r]);
        }
        if (fr == 0) {
          float *const rr = red + ((c * 2 + wm) * 2) * kPxChunk + wn * 32 + fq * 8 + 4 * h;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            rr[r] = st_s[r];
            rr[kPxChunk + r] = st_q[r];
          }
        }
      }
    }
  };

  // One barrier-delimited interval per chunk.  Plain: every wave multiplies chunk t, then stores it -- the two waves of a SIMD are in the
  // same phase, the matrix pipe idles through both epilogues.  Staggered (STAG): the second pixel half (waves 4 .. 7, the SIMDs' second
  // waves) stores chunk t - 1 FIRST and multiplies chunk t afterwards, so one wave's conversions, statistics and stores run under the
  // other wave's MFMAs; one more interval at the end for its last epilogue.  Same buffers, same barriers, same sums.
  auto landed = [&]() {
    // this wave's pieces of the next stage (issued a chunk's MFMAs ago) have landed, and its earlier stores have drained
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    retire_prefetch();
  };
  auto close_interval = [&]() {
    // every wave's pieces of the next stage have landed (its vmcnt(0)), and every wave is done reading this interval's buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // (buffer (t + 1) & 1 was read by chunk t - 1, which every wave finished before the barrier that closed interval t - 1)
  if (!STAG || wm == 0) {
    for (int t = 0; t < nchunks; ++t) {
      if (t + 1 < nchunks && !(dbg & 2)) issue(t + 1, (t + 1) & 1);
      prefetch(t);
      compute(t);
      stamp(2 + 2 * t);
      landed();
      epilogue(t);
      stamp(3 + 2 * t);
      close_interval();
    }
    if constexpr (STAG) __builtin_amdgcn_s_barrier();      // the other half's last interval
  } else {
    if (1 < nchunks && !(dbg & 2)) issue(1, 1);
    prefetch(0);
    compute(0);
    landed();
    close_interval();
    for (int t = 1; t < nchunks; ++t) {
      if (t + 1 < nchunks && !(dbg & 2)) issue(t + 1, (t + 1) & 1);
      epilogue(t - 1);
      __builtin_amdgcn_sched_barrier(0);      // (the next chunk's prefetch must not start while the previous one's registers are live)
      prefetch(t);
      compute(t);
      landed();
      close_interval();
    }
    epilogue(nchunks - 1);
    close_interval();
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
  stamp(10);
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

int conv_px_launch(const ConvParams &p, bool stag, int dbg, hipStream_t s) {
  const int mtiles = sn_div_up(p.M, kPxBM);
  const int nchunks = p.Nout > kPxMaxChunks * kPxChunk ? kPxMaxChunks : p.Nout / kPxChunk;
  const int ntiles = p.Nout / (nchunks * kPxChunk);
  const dim3 grid((unsigned)(sn_div_up(mtiles, 8) * 8 * ntiles));
  const bool bnx = p.bn_x != nullptr;
#define SN_PX_LAUNCH(KC)                                                                                       \
  do {                                                                                                         \
    if (bnx && stag) hipLaunchKernelGGL((conv_px_kernel<KC, true, true>), grid, dim3(512), 0, s, p, mtiles, ntiles, nchunks, dbg);        \
    else if (bnx) hipLaunchKernelGGL((conv_px_kernel<KC, true, false>), grid, dim3(512), 0, s, p, mtiles, ntiles, nchunks, dbg);          \
    else if (stag) hipLaunchKernelGGL((conv_px_kernel<KC, false, true>), grid, dim3(512), 0, s, p, mtiles, ntiles, nchunks, dbg);         \
    else hipLaunchKernelGGL((conv_px_kernel<KC, false, false>), grid, dim3(512), 0, s, p, mtiles, ntiles, nchunks, dbg);                  \
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
