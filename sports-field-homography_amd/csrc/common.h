// Shared helpers for the gfx950 kernels of libsfh_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sfh_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Error plumbing (thread-local message, returned by sfh_last_error()).
void sfh_set_error(const char* fmt, ...);

#define SFH_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      sfh_set_error(__VA_ARGS__);   \
      return SFH_E_ARG;             \
    }                               \
  } while (0)

static inline int sfh_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sfh_set_error("%s: %s", what, hipGetErrorString(e));
    return SFH_E_LAUNCH;
  }
  return SFH_OK;
}

static inline int sfh_cdiv(int a, int b) { return (a + b - 1) / b; }

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once
// per (kernel instance, device).  The template parameter gives every call site its own flag word; bit d of it
// records device d.  The only process-wide state of the library besides the last-error string: idempotent,
// lock-free, and correct when several host threads or devices race (setting the attribute twice is harmless).
#include <atomic>
template <int = 0>
static inline void sfh_allow_big_lds_impl(const void* fn, std::atomic<unsigned long long>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.fetch_or(bit, std::memory_order_release);
}
#define sfh_allow_big_lds(fn)                                  \
  do {                                                         \
    static std::atomic<unsigned long long> sfh_done_{0};       \
    sfh_allow_big_lds_impl(fn, sfh_done_);                     \
  } while (0)
