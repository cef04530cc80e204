// clock_probe.hip -- what shader clock does the chip SUSTAIN under a matrix-core load?  The roofline's peak (2.5 PFLOP/s dense fp16)
// is the 2.4 GHz figure; the phase stamps of the convolution kernels count shader cycles (s_memtime) and put a workgroup's life at
// about half of the kernel's wall-clock duration.  Here every workgroup reads the shader-cycle counter (s_memtime) and the constant
// 100 MHz counter (s_memrealtime) before and after (a) a sleep loop, (b) a dense MFMA loop on all four SIMDs of all 256 CUs, launched
// back to back for ~2 s so that power management has settled; cycles per 10 ns tick x 100 = effective MHz.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

typedef float float16v __attribute__((ext_vector_type(16)));

// the same loop with v_mfma_f32_32x32x16_f16 (4 independent 16-register accumulators)
__global__ __launch_bounds__(256) void load32_kernel(unsigned long long *rec, float *sink, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float16v acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { rec[blockIdx.x * 2] = c1 - c0; rec[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <bool MFMA>
__global__ __launch_bounds__(256) void load_kernel(unsigned long long *rec, float *sink, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  if (MFMA) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)          // in place (hipcc's own allocation of the builtin shuffled accumulators through ~50 moves per trip)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  } else {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(64);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { rec[blockIdx.x * 2] = c1 - c0; rec[blockIdx.x * 2 + 1] = r1 - r0; }
}

static void report(const char *what, unsigned long long *d, int W, double flop_per_wg, float ms_per_launch) {
  std::vector<unsigned long long> h(W * 2);
  CK(hipMemcpy(h.data(), d, (size_t)W * 16, hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  for (int b = 0; b < W; ++b) mhz.push_back(h[b * 2 + 1] ? (double)h[b * 2] / (double)h[b * 2 + 1] * 100.0 : 0.0);
  std::sort(mhz.begin(), mhz.end());
  printf("%-28s %4d workgroups: shader clock median %6.0f MHz (p10 %6.0f, p90 %6.0f), %7.1f us per launch", what, W, mhz[W / 2], mhz[W / 10],
         mhz[W * 9 / 10], ms_per_launch * 1e3);
  if (flop_per_wg > 0) printf(", %7.1f TFLOP/s = %.3f of 2500", flop_per_wg * W / (ms_per_launch * 1e-3) / 1e12, flop_per_wg * W / (ms_per_launch * 1e-3) / 2.5e15);
  printf("\n");
}

int main() {
  unsigned long long *d;
  float *sink;
  const int W = 1024;                       // 4 workgroups of 4 waves per CU: every SIMD holds 4 waves issuing MFMAs back to back
  CK(hipMalloc(&d, W * 16));
  CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  hipLaunchKernelGGL(load_kernel<false>, dim3(W), dim3(256), 0, 0, d, sink, 200);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(load_kernel<false>, dim3(W), dim3(256), 0, 0, d, sink, 200);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  report("idle (sleep loop)", d, W, 0, ms);
  const int iters = 4000;                   // 8 x 4000 MFMAs per wave
  const double flop_per_wg = 4.0 * 8.0 * iters * 16 * 16 * 32 * 2;
  for (int phase = 0; phase < 4; ++phase) {
    const int reps = phase == 0 ? 1 : 1500;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(load_kernel<true>, dim3(W), dim3(256), 0, 0, d, sink, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    char what[64];
    snprintf(what, sizeof(what), "MFMA load, %s", phase == 0 ? "first launch" : (phase == 1 ? "after 1500 launches" : "after 1500 more"));
    report(what, d, W, flop_per_wg, ms / reps);
  }
  // waves per SIMD and instruction shape: 256 / 512 / 1024 workgroups of four waves = 1 / 2 / 4 waves per SIMD
  for (int Wv : {256, 512, 1024}) {
    const int it16 = 4000 * (1024 / Wv), it32 = 2000 * (1024 / Wv);
    for (int shape = 0; shape < 2; ++shape) {
      const int reps = 300;
      for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
          if (shape == 0) hipLaunchKernelGGL(load_kernel<true>, dim3(Wv), dim3(256), 0, 0, d, sink, it16);
          else hipLaunchKernelGGL(load32_kernel, dim3(Wv), dim3(256), 0, 0, d, sink, it32);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      char what[64];
      snprintf(what, sizeof(what), "%s, %d waves/SIMD", shape == 0 ? "16x16x32 x8 acc" : "32x32x16 x4 acc", Wv / 256);
      const double fl = shape == 0 ? 4.0 * 8.0 * it16 * 16 * 16 * 32 * 2 : 4.0 * 4.0 * it32 * 32 * 32 * 16 * 2;
      report(what, d, Wv, fl, ms / reps);
    }
  }
  return 0;
}
