// common.h -- shared helpers for the gfx950 kernel library (internal, not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sniper_hip.h"

#define SN_EXPORT extern "C" __attribute__((visibility("default")))

void sn_set_error(const char *fmt, ...);

#define SN_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      sn_set_error(__VA_ARGS__);   \
      return SN_ERR_ARG;           \
    }                              \
  } while (0)

// Launch-time errors only (no sync): a failed launch is reported, asynchronous faults surface at
// the caller's next synchronisation like any HIP error.
#define SN_CHECK_LAUNCH()                                                      \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      sn_set_error("%s:%d HIP launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return SN_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

#define SN_HIP(call)                                                           \
  do {                                                                         \
    hipError_t e__ = (call);                                                   \
    if (e__ != hipSuccess) {                                                   \
      sn_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return SN_ERR_HIP;                                                       \
    }                                                                          \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: applied once per (kernel, device),
// thread-safe (a mutex-guarded registry, no unsynchronised flag).
hipError_t sn_once_per_device_max_lds(const void *kernel, int bytes);
// Process-wide test / A-B switches (sn_debug_option in the header): relaxed atomics, initialised from the environment once at
// load time (SNIPER_FULL_SORT, SNIPER_NMS_FULL, SNIPER_BN_FUSED_FINALIZE, SNIPER_CONV_NO_PERSIST, SNIPER_CONV_DMA_NOUT64, SNIPER_DPSROI_SLAB), never a getenv inside a launch path.
enum SnDebugOption { SN_OPT_PROPOSAL_FULL_SORT = 0, SN_OPT_NMS_FULL_MASK = 1, SN_OPT_BN_FUSED_FINALIZE = 2, SN_OPT_CONV_NO_PERSIST = 3, SN_OPT_CONV_DMA_NOUT64 = 4, SN_OPT_DPSROI_SLAB = 5, SN_OPT_COUNT };
int sn_debug_get(SnDebugOption which);

static inline hipStream_t sn_stream(sn_stream_t s) { return (hipStream_t)s; }
static inline int sn_div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t sn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

constexpr int kWave = 64;  // CDNA wavefront width

// Division by a launch-invariant divisor as multiply-high + shift (n < 2^31, d >= 1): q = (umulhi(n, mul) + n) >> sh.  An integer
// division costs ~40 VALU instructions on gfx950 (64-bit: ~80); index decompositions of element-wise kernels use these instead.
struct SnDiv { unsigned mul, sh, d; };
static inline SnDiv sn_div_make(unsigned d) {
  SnDiv f;
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << s) - d)) / d + 1);
  f.sh = s;
  f.d = d;
  return f;
}
__device__ __forceinline__ unsigned sn_div(unsigned n, const SnDiv f) { return (__umulhi(n, f.mul) + n) >> f.sh; }
// n -> (n / d, n % d)
__device__ __forceinline__ unsigned sn_divmod(unsigned n, const SnDiv f, unsigned &rem) {
  const unsigned q = sn_div(n, f);
  rem = n - q * f.d;
  return q;
}
