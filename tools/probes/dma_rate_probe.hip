// dma_rate_probe.hip -- how many bytes per clock an MI355X CU moves L2 -> LDS with LDS-DMA (buffer_load_dwordx4 ... lds), as a
// function of waves per workgroup, workgroups per CU, pieces per wave per step, barrier per step and MFMA work beside it.
// Answers: is the K loop of conv_dma_kernel (tools/conv_trace.py: ~25 B/clk per workgroup) at a hardware limit of the load
// path, of one workgroup's issue, or of its barrier / wait structure?
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_rate_probe.hip -o /tmp/dma_rate_probe && /tmp/dma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t *dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NW waves, each issues L pieces (1 KB) per step into a ring of S stages; MF MFMAs per wave per step; BAR: barrier per step
template <int NW, int L, int S, int MF, bool BAR, int LDSPAD>
__global__ __launch_bounds__(64 * NW) void probe(const char *src, unsigned src_bytes, int steps, unsigned long long *out, float *sink) {
  constexpr int STAGE = NW * L * 512;   // halfs
  __shared__ __attribute__((aligned(1024))) half_t lds[S * STAGE + LDSPAD / 2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src), 0, (int)src_bytes, 0x00020000);
  // every workgroup walks its own window of the (L2-resident) source: rows of 128 B, 8 rows per piece, like the conv tiles
  unsigned base = (unsigned)((blockIdx.x * 7919u) % (src_bytes / (STAGE * 2))) * (STAGE * 2);
  const unsigned lane_off = (unsigned)(lane >> 3) * 128u + (unsigned)((lane & 7) ^ (lane >> 3)) * 16u;
  floatx4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  half8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  int issued = 0;
  auto issue = [&](int buf) {
    const unsigned o = (base + (unsigned)issued * (STAGE * 2)) % (src_bytes - STAGE * 2);
#pragma unroll
    for (int i = 0; i < L; ++i) dma16(r, lds + buf * STAGE + (wave * L + i) * 512, o + (unsigned)(wave * L + i) * 1024u + lane_off);
    ++issued;
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue(s);
  int cur = 0, nxt = S - 1;
  for (int t = 0; t < steps; ++t) {
    wait_vmcnt<(S - 2) * L>();
    if (BAR) __builtin_amdgcn_s_barrier();
    issue(nxt);
    if (MF > 0) {
      const half8 v = *reinterpret_cast<const half8 *>(lds + cur * STAGE + wave * L * 512 + lane * 8);
#pragma unroll
      for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(m & 1 ? fa : v, fb, acc[m & 3], 0, 0, 0);
    }
    cur = cur + 1 == S ? 0 : cur + 1;
    nxt = nxt + 1 == S ? 0 : nxt + 1;
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.f) sink[0] = 1.f;
}

template <int NW, int L, int S, int MF, bool BAR, int LDSPAD>
static void run(const char *name, int per_cu, const char *src, unsigned src_bytes, unsigned long long *out, float *sink) {
  const int steps = 200, grid = 256 * per_cu;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NW, L, S, MF, BAR, LDSPAD>), dim3(grid), dim3(64 * NW), 0, 0, src, src_bytes, steps, out, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  const double cyc = s / grid / steps;   // cycles per step of one workgroup
  const double bytes = (double)NW * L * 1024;
  printf("%-44s waves %d pieces/wave %d stages %d mfma/wave/step %2d barrier %d WG/CU %d | %6.0f cyc/step | %5.1f B/clk/WG  %5.1f B/clk/CU\n", name, NW,
         L, S, MF, (int)BAR, per_cu, cyc, bytes / cyc, bytes / cyc * per_cu);
}

int main() {
  const unsigned src_bytes = 3u << 20;   // 3 MB: inside one XCD's 4 MB L2
  char *src;
  unsigned long long *out;
  float *sink;
  hipMalloc(&src, src_bytes);
  hipMemset(src, 1, src_bytes);
  hipMalloc(&out, 8 * 4096);
  hipMalloc(&sink, 4);
  // LDSPAD forces the number of co-resident workgroups: 160 KB per CU
  run<4, 8, 2, 0, true, 96 * 1024 - 65536>("4 waves x 8, S2, DMA only", 1, src, src_bytes, out, sink);
  run<4, 8, 3, 0, true, 0>("4 waves x 8, S3, DMA only", 1, src, src_bytes, out, sink);
  run<4, 8, 4, 0, true, 0>("4 waves x 8, S4, DMA only", 1, src, src_bytes, out, sink);
  run<4, 8, 3, 0, false, 0>("4 waves x 8, S3, DMA only, no barrier", 1, src, src_bytes, out, sink);
  run<4, 8, 2, 0, true, 0>("4 waves x 8, S2, DMA only", 2, src, src_bytes, out, sink);
  run<4, 4, 2, 0, true, 16384>("4 waves x 4, S2, DMA only", 3, src, src_bytes, out, sink);
  run<8, 4, 3, 0, true, 0>("8 waves x 4, S3, DMA only", 1, src, src_bytes, out, sink);
  run<8, 8, 2, 0, true, 0>("8 waves x 8, S2, DMA only", 1, src, src_bytes, out, sink);
  run<4, 8, 3, 32, true, 0>("4 waves x 8, S3, + 32 MFMA", 1, src, src_bytes, out, sink);
  run<4, 8, 2, 32, true, 0>("4 waves x 8, S2, + 32 MFMA", 2, src, src_bytes, out, sink);
  run<8, 4, 3, 16, true, 0>("8 waves x 4, S3, + 16 MFMA", 1, src, src_bytes, out, sink);
  run<4, 4, 2, 16, true, 16384>("4 waves x 4, S2, + 16 MFMA", 3, src, src_bytes, out, sink);
  run<4, 8, 3, 64, true, 0>("4 waves x 8, S3, + 64 MFMA", 1, src, src_bytes, out, sink);
  run<4, 2, 3, 0, true, 80 * 1024>("4 waves x 2, S3, DMA only", 1, src, src_bytes, out, sink);
  run<1, 8, 3, 0, false, 100 * 1024>("1 wave x 8, S3, DMA only", 1, src, src_bytes, out, sink);
  return 0;
}
