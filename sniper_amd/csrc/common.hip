// common.hip -- error reporting + version for libsniper_hip.so
#include "common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

static thread_local char g_err[512] = "";

void sn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

SN_EXPORT const char *sn_last_error(void) { return g_err; }
SN_EXPORT int sn_version(void) { return 100; }

hipError_t sn_once_per_device_max_lds(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::vector<std::pair<const void *, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto &d : done)
    if (d.first == kernel && d.second == dev) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.emplace_back(kernel, dev);
  return e;
}

static std::atomic<int> g_debug[SN_OPT_COUNT];
static const char *const kDebugNames[SN_OPT_COUNT] = {"proposal_full_sort", "nms_full_mask", "bn_fused_finalize", "conv_no_persist", "conv_dma_nout64", "dpsroi_slab"};
static const char *const kDebugEnv[SN_OPT_COUNT] = {"SNIPER_FULL_SORT", "SNIPER_NMS_FULL", "SNIPER_BN_FUSED_FINALIZE", "SNIPER_CONV_NO_PERSIST", "SNIPER_CONV_DMA_NOUT64", "SNIPER_DPSROI_SLAB"};
namespace {
struct DebugInit {
  DebugInit() {
    for (int i = 0; i < SN_OPT_COUNT; ++i) g_debug[i].store(getenv(kDebugEnv[i]) != nullptr ? 1 : 0, std::memory_order_relaxed);
  }
} g_debug_init;
}  // namespace

int sn_debug_get(SnDebugOption which) { return g_debug[which].load(std::memory_order_relaxed); }

SN_EXPORT int sn_debug_option(const char *name, int value) {
  SN_REQUIRE(name, "sn_debug_option: null name");
  for (int i = 0; i < SN_OPT_COUNT; ++i)
    if (strcmp(name, kDebugNames[i]) == 0) {
      g_debug[i].store(value, std::memory_order_relaxed);
      return SN_OK;
    }
  SN_REQUIRE(false, "sn_debug_option: unknown option '%s'", name);
}
