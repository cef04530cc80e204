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
// one per layer from the measured table (tools/conv_tune.py).
#include "conv_common.h"

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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool DGRAD, int BM, int BN, int WMW, int WNW, int S, int MINW>
__global__ __launch_bounds__(64 * WMW *WNW, MINW) void conv_dma_kernel(const ConvParams p, int mtiles, int ntiles) {
  constexpr int NW = WMW * WNW, T = 64 * NW, BK = 64;
  constexpr int WTM = BM / WMW, WTN = BN / WNW;   // wave tile
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int AGW = BM / 8 / NW, BGW = BN / 8 / NW;   // 8-row DMA groups per wave
  constexpr int L = AGW + BGW;                          // DMA instructions per wave per stage
  constexpr int STAGE = (BM + BN) * BK;                 // half_t elements per stage
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile / wave shape");
  static_assert((S - 1) * L < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) half_t lds[S * STAGE];

  const int lin = blockIdx.x, xcd = lin & 7, j = lin >> 3;
  const int nt = j % ntiles, mt = (j / ntiles) * 8 + xcd;
  if (mt >= mtiles) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = lane >> 3, gchunk = (lane & 7) ^ lrow;   // row inside an 8-row group, global 16-byte chunk

  // ---- per-lane gather state: A rows m0 + 8 (wave + NW i) + lrow
  int a_base[AGW], a_h[AGW], a_w[AGW];
  bool a_ok[AGW];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AGW; ++i) {
    const int m = m0 + 8 * (wave + NW * i) + lrow;
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
  const char *xb = reinterpret_cast<const char *>(p.x), *wb = reinterpret_cast<const char *>(p.w);
  const unsigned in_ps_bytes = (unsigned)p.in_ps * 2u;
  constexpr unsigned kOob = 0xFFFFFF00u;
  unsigned w_voff[BGW];
#pragma unroll
  for (int i = 0; i < BGW; ++i) {
    const int n = n0 + 8 * (wave + NW * i) + lrow;
    w_voff[i] = n < p.Nout ? (unsigned)n * wrow_bytes + (unsigned)gchunk * 16u : kOob;
  }
  int g_kh = 0, g_kw = 0, g_kc = 0, g_kt = 0;   // next stage to issue: tap (g_kh, g_kw), channel block g_kc, K-step g_kt
  unsigned a_voff[AGW];
  auto tap_setup = [&]() {
#pragma unroll
    for (int i = 0; i < AGW; ++i) {
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
      a_voff[i] = ok ? (unsigned)(a_base[i] + sy * p.W + sx) * in_ps_bytes + (unsigned)gchunk * 16u : kOob;
    }
  };
  tap_setup();
  // wave-uniform LDS destinations: stage base + group * 1 KB (the DMA adds 16 B per lane)
  half_t *const a_dst0 = lds + wave * 512, *const b_dst0 = lds + BM * BK + wave * 512;
  auto issue = [&](int buf) {
    const unsigned cbo = (unsigned)g_kc * (BK * 2), wbo = (unsigned)g_kt * (BK * 2);   // uniform
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb) + cbo, 0, (int)(p.x_bytes - cbo), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb) + wbo, 0, (int)(p.w_bytes - wbo), 0x00020000);
    half_t *const sa = a_dst0 + buf * STAGE, *const sb = b_dst0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AGW; ++i)
      dma16(rx, sa + i * NW * 512, a_voff[i]);
#pragma unroll
    for (int i = 0; i < BGW; ++i)
      dma16(rw, sb + i * NW * 512, w_voff[i]);
    ++g_kt;
    if (++g_kc == kpt) {
      g_kc = 0;
      if (++g_kw == p.KW) { g_kw = 0; ++g_kh; }
      tap_setup();
    }
  };

  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) acc[i][jn] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int sw = fq ^ (fr & 7);
  const int a_rd = (wm * WTM + fr) * BK, b_rd = BM * BK + (wn * WTN + fr) * BK;
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
  const bool vec = (p.out_ps % 4 == 0) && (p.Nout % 4 == 0) && (!p.res || p.res_ps % 4 == 0);
  constexpr bool kPre = MI * NI <= 16;      // 2 VGPRs per fragment; the 256 x 256 tile has none to spare
  half4 rpre[kPre ? MI : 1][kPre ? NI : 1];
  const bool pre_res = kPre && p.res != nullptr && vec;
  if constexpr (kPre) if (pre_res) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WTM + i * 16 + (lane & 15);
#pragma unroll
      for (int jn = 0; jn < NI; ++jn) {
        const int n = n0 + wn * WTN + jn * 16 + (lane >> 4) * 4;
        rpre[i][jn] = (m < p.M && n < p.Nout) ? *reinterpret_cast<const half4 *>(p.res + (size_t)m * p.res_ps + n) : half4{0, 0, 0, 0};
      }
    }
  }

  // ---- pipeline: stages t+1 .. t+S-1 in flight under compute(t); one barrier per K-step
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s);
  int cur = 0, nxt = S - 1;   // buffer of stage t / of stage t+S-1
  int t = 0;
  for (; t + S - 1 < nk; ++t) {
    wait_vmcnt<(S - 2) * L>();          // stage t has landed (this wave's part); S-2 younger stages stay in flight
    __builtin_amdgcn_s_barrier();       // ... everybody's part has, and everybody is done reading buffer `nxt` (stage t-1)
    issue(nxt);
    compute(cur);
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

  // ---- epilogue: lane (fr, fq) holds, for each (i, jn), pixel m = ..+fr and channels n = ..+fq*4 .. +3
  float st_s[NI][4], st_q[NI][4];        // BatchNorm statistics of this lane's output channels (host: only with `vec`)
#pragma unroll
  for (int jn = 0; jn < NI; ++jn)
#pragma unroll
    for (int r = 0; r < 4; ++r) st_s[jn][r] = st_q[jn][r] = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WTM + i * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
      const int n = n0 + wn * WTN + jn * 16 + fq * 4;
      if (n >= p.Nout) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][jn][r];
      if (vec) {
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if constexpr (kPre) {
          if (pre_res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rpre[i][jn][r];
          }
        } else if (p.res) {
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
    // the 16 lanes that share fq hold different pixels of the same 4 channels -> xor-shuffle over fr, then the WMW waves of a
    // column block through LDS (the K loop is over), summed in wave order: fixed order -> deterministic
#pragma unroll
    for (int jn = 0; jn < NI; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st_s[jn][r] = row16_sum(st_s[jn][r]);
        st_q[jn][r] = row16_sum(st_q[jn][r]);
      }
    float *red = reinterpret_cast<float *>(lds);   // [wm][2][BN]
    __syncthreads();
    if (fr == 0) {
#pragma unroll
      for (int jn = 0; jn < NI; ++jn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = wn * WTN + jn * 16 + fq * 4 + r;
          red[(wm * 2 + 0) * BN + c] = st_s[jn][r];
          red[(wm * 2 + 1) * BN + c] = st_q[jn][r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * BN; idx += T) {
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
}

// cfg -> tile shape.  LDS = stages * (bm + bn) * 128 B.
static const ConvDmaConfig kCfg[kConvDmaConfigs + 1] = {
    {0, 0, 0, 0, 0},
    {128, 128, 256, 2, 2 * 256 * 128},   // 1: 64 KB, 2 workgroups / CU
    {128, 128, 256, 3, 3 * 256 * 128},   // 2: 96 KB
    {256, 128, 512, 3, 3 * 384 * 128},   // 3: 144 KB, 8 waves, wave tile 64 x 64
    {128, 256, 512, 3, 3 * 384 * 128},   // 4
    {64, 128, 256, 3, 3 * 192 * 128},    // 5: 72 KB, 2 / CU, wave tile 32 x 64
    {64, 128, 256, 2, 2 * 192 * 128},    // 6: 48 KB, 3 / CU
    {256, 256, 512, 2, 2 * 512 * 128},   // 7: 128 KB, wave tile 64 x 128
    {128, 128, 512, 3, 3 * 256 * 128},   // 8: 96 KB, 8 waves, wave tile 32 x 64
    {128, 128, 256, 4, 4 * 256 * 128},   // 9: 128 KB
};

ConvDmaConfig conv_dma_config(int cfg) { return (cfg >= 1 && cfg <= kConvDmaConfigs) ? kCfg[cfg] : kCfg[0]; }

template <bool DGRAD, int BM, int BN, int WMW, int WNW, int S, int MINW>
static void launch_one(const ConvParams &p, hipStream_t s) {
  const int mtiles = sn_div_up(p.M, BM), ntiles = sn_div_up(p.Nout, BN);
  const dim3 grid(sn_div_up(mtiles, 8) * 8 * ntiles);
  hipLaunchKernelGGL((conv_dma_kernel<DGRAD, BM, BN, WMW, WNW, S, MINW>), grid, dim3(64 * WMW * WNW), 0, s, p, mtiles, ntiles);
}

template <bool DGRAD>
static int launch_cfg(const ConvParams &p, int cfg, hipStream_t s) {
  switch (cfg) {
    case 1: launch_one<DGRAD, 128, 128, 2, 2, 2, 2>(p, s); break;
    case 2: launch_one<DGRAD, 128, 128, 2, 2, 3, 1>(p, s); break;
    case 3: launch_one<DGRAD, 256, 128, 4, 2, 3, 2>(p, s); break;
    case 4: launch_one<DGRAD, 128, 256, 2, 4, 3, 2>(p, s); break;
    case 5: launch_one<DGRAD, 64, 128, 2, 2, 3, 2>(p, s); break;
    case 6: launch_one<DGRAD, 64, 128, 2, 2, 2, 3>(p, s); break;
    case 7: launch_one<DGRAD, 256, 256, 4, 2, 2, 2>(p, s); break;
    case 8: launch_one<DGRAD, 128, 128, 4, 2, 3, 2>(p, s); break;
    case 9: launch_one<DGRAD, 128, 128, 2, 2, 4, 1>(p, s); break;
    default: SN_REQUIRE(false, "conv_dma_launch: unknown configuration %d", cfg);
  }
  SN_CHECK_LAUNCH();
  return SN_OK;
}

int conv_dma_launch(const ConvParams &p, bool dgrad, int cfg, hipStream_t s) {
  return dgrad ? launch_cfg<true>(p, cfg, s) : launch_cfg<false>(p, cfg, s);
}
