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

// ReLU and max with torch's NaN behaviour (relu(NaN) = NaN, max_pool propagates NaN): v_max_f32 would return the other
// operand.  A non-finite input frame then produces non-finite outputs, as it does in the reference.
__device__ __forceinline__ float sfh_relu(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float sfh_max_nan(float a, float b) { return (a > b || a != a) ? a : b; }

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

// Diagnostic build only (-DSFH_DIAG_STAMPS -fgpu-rdc, libsfh_amd_diag.so; profiles/diag_stamps.py): per-segment
// s_memtime sums of a kernel's phases, accumulated per wave and added to g_stamps (conv_mfma.hip) by lane 0.
// SFH_STAMP(i) closes segment i; BASE offsets a kernel's segments inside the 16 counters.
#ifdef SFH_DIAG_STAMPS
extern __device__ unsigned long long g_stamps[16];
#ifdef SFH_DIAG_CLOCK_ONLY   // only the in-kernel clock of SFH_CLOCK_BEGIN / END: no phase stamp fences the schedule
#define SFH_STAMP(i) do {} while (0)
#else
#define SFH_STAMP(i)                                                                         \
  do {                                                                                       \
    unsigned long long t_;                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    seg_[i] += t_ - tprev_;                                                                  \
    tprev_ = t_;                                                                             \
  } while (0)
#endif
#define SFH_STAMP_INIT()                                                                     \
  unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_;                             \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_)::"memory")
#define SFH_STAMP_FLUSH_AT(BASE)                                                             \
  do {                                                                                       \
    /* one workgroup in 32 reports: 8 same-address atomics per wave of EVERY workgroup serialise and */ \
    /* dominated the run time of short kernels (14,480 workgroups: 5.6 ms instead of 0.76 ms)        */ \
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 0)                                   \
      for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamps[(BASE) + i_], seg_[i_]);            \
  } while (0)
#define SFH_STAMP_FLUSH() SFH_STAMP_FLUSH_AT(0)
// in-kernel clock (MI355X_MICROARCH.md, DVFS give-back (6)): shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime)
// over the life of the wave go to segments 6 and 7: clock = 100 MHz * sum[6] / sum[7]
#define SFH_CLOCK_BEGIN()                                                                    \
  unsigned long long ck0_, rt0_;                                                             \
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck0_), "=s"(rt0_)::"memory")
#define SFH_CLOCK_END()                                                                      \
  do {                                                                                       \
    unsigned long long ck1_, rt1_;                                                           \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck1_), "=s"(rt1_)::"memory"); \
    seg_[6] += ck1_ - ck0_;                                                                  \
    seg_[7] += rt1_ - rt0_;                                                                  \
  } while (0)
#else
#define SFH_CLOCK_BEGIN() do {} while (0)
#define SFH_CLOCK_END() do {} while (0)
#define SFH_STAMP(i) do {} while (0)
#define SFH_STAMP_INIT() do {} while (0)
#define SFH_STAMP_FLUSH() do {} while (0)
#define SFH_STAMP_FLUSH_AT(BASE) do {} while (0)
#endif
