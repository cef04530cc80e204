// occupancy_probe.hip -- how many convolution-shaped workgroups does the chip actually run at once?  The phase stamps of
// conv_dma_kernel (profiles/r04_conv_trace.txt) show a workgroup's life at about HALF of the kernel's duration for launches that
// should be one or two waves of workgroups (256 x 147 KB LDS: life 20.6 us, kernel 38.5 us; 1024 x 72 KB: 4 x 8.1 us in 33.9 us).
// Here: W workgroups of 512 threads with L bytes of dynamic LDS each spin for ~SPIN us and record start / end on the constant
// 100 MHz clock (s_memrealtime: the same time base on every XCD) plus the XCC / SE / CU they ran on.  Printed: kernel duration,
// median workgroup life, the number of workgroups alive at the kernel's mid-point, distinct (XCC, CU) pairs used.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/occupancy_probe.hip -o /tmp/occupancy_probe && /tmp/occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void spin_kernel(unsigned long long *rec, int spin_ticks) {
  extern __shared__ char lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) lds[0] = 1;
  unsigned hw = 0, xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(4);
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    rec[blockIdx.x * 4 + 0] = t0;
    rec[blockIdx.x * 4 + 1] = t1;
    rec[blockIdx.x * 4 + 2] = hw;
    rec[blockIdx.x * 4 + 3] = xcc;
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s: multiProcessorCount %d, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlock %zu, clockRate %d kHz\n", prop.name,
         prop.multiProcessorCount, prop.maxSharedMemoryPerMultiProcessor, prop.sharedMemPerBlock, prop.clockRate);
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  unsigned long long *d;
  CK(hipMalloc(&d, 4096 * 4 * 8));
  std::vector<unsigned long long> h(4096 * 4);
  const int spin_us = 10;
  for (int lds_kb : {16, 72, 147}) {
    for (int W : {256, 512, 1024}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(W), dim3(512), lds_kb * 1024, 0, d, spin_us * 100);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;
        CK(hipMemcpy(h.data(), d, (size_t)W * 32, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        std::vector<double> life;
        std::set<unsigned long long> places;
        for (int b = 0; b < W; ++b) {
          tmin = std::min(tmin, h[b * 4]); tmax = std::max(tmax, h[b * 4 + 1]);
          life.push_back((h[b * 4 + 1] - h[b * 4]) / 100.0);
          const unsigned hw = (unsigned)h[b * 4 + 2];
          places.insert(((h[b * 4 + 3] & 0xF) << 16) | (((hw >> 8) & 0xF) << 0) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 0x1) << 8));
        }
        std::sort(life.begin(), life.end());
        const unsigned long long mid = tmin + (unsigned long long)(spin_us * 100 / 2);        // middle of the FIRST wave of workgroups
        int alive = 0, started_first = 0;
        for (int b = 0; b < W; ++b) {
          alive += h[b * 4] <= mid && h[b * 4 + 1] > mid;
          started_first += h[b * 4] < tmin + 300;       // within 3 us of the first
        }
        printf("LDS %3d KB  W %4d : event %7.1f us, first start -> last end %7.1f us, life median %5.1f us, alive at the first wave's "
               "mid-point %4d, started within 3 us %4d, distinct (xcc, se, cu) %3zu\n", lds_kb, W, ms * 1e3, (tmax - tmin) / 100.0,
               life[life.size() / 2], alive, started_first, places.size());
      }
    }
  }
  return 0;
}
