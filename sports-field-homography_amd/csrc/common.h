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
