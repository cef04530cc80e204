// launch_probe.hip -- what does one MORE kernel in a dependent chain cost on MI355X?  A training step is ~1400 dependent launches of
// 5 - 60 us; rocprofv3 shows no gaps between them (busy 0.999) because a kernel's duration runs from its first wave to its last --
// dispatch ramp, end-of-kernel cache write-back and invalidate of the 8 XCD L2s are inside somebody's duration.  Measured here:
// time per launch of back-to-back dependent kernels on one stream, eager and as a hipGraph, for
//   empty1      1 workgroup, no memory access
//   empty1k     1024 workgroups of 256 threads, no memory access
//   touch       1024 workgroups each writing 40 KB (a convolution tile's output: 42 MB per launch)
//   touch+empty the pair (does a tiny kernel behind a writer wait for the writer's L2 write-back?)
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/launch_probe.hip -o /tmp/launch_probe && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ __launch_bounds__(256) void touch_kernel(uint4 *out, int per_block16) {
  uint4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
  uint4 *o = out + (size_t)blockIdx.x * per_block16;
  for (int i = threadIdx.x; i < per_block16; i += 256) o[i] = v;
}

template <typename F>
static float time_us(F launch, int reps, hipStream_t s) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  uint4 *buf; CK(hipMalloc(&buf, (size_t)1024 * 40 * 1024));
  const int reps = 400;
  auto l_empty1 = [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int *)nullptr); };
  auto l_empty1k = [&] { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, s, (int *)nullptr); };
  auto l_touch = [&] { hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, s, buf, 40 * 1024 / 16); };
  auto l_pair = [&] { l_touch(); l_empty1(); };
  printf("eager  : empty1 %.2f us  empty1k %.2f us  touch(42MB) %.2f us  touch+empty1 %.2f us\n", time_us(l_empty1, reps, s),
         time_us(l_empty1k, reps, s), time_us(l_touch, reps, s), time_us(l_pair, reps, s));
  // the same chains as hipGraphs of 200 kernel nodes (stream capture)
  auto graph_time = [&](auto launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 200; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    const float us = time_us([&] { CK(hipGraphLaunch(ge, s)); }, 20, s) / 200.f;
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return us;
  };
  printf("graph  : empty1 %.2f us  empty1k %.2f us  touch(42MB) %.2f us  touch+empty1 %.2f us (per pair)\n", graph_time(l_empty1),
         graph_time(l_empty1k), graph_time(l_touch), graph_time(l_pair));
  return 0;
}
