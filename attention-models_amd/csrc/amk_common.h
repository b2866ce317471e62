// Shared device/host helpers for the libamk.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/amk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AMK_WAVE 64
#define AMK_LOG2E 1.4426950408889634f
#define AMK_LN2 0.6931471805599453f

// ---- host side -----------------------------------------------------------
void amk_set_error(const char* fmt, ...);

#define AMK_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      amk_set_error(__VA_ARGS__);       \
      return AMK_EINVAL;                \
    }                                   \
  } while (0)

#define AMK_CHECK_SUPPORTED(cond, ...)  \
  do {                                  \
    if (!(cond)) {                      \
      amk_set_error(__VA_ARGS__);       \
      return AMK_EUNSUPPORTED;          \
    }                                   \
  } while (0)

#define AMK_CHECK_LAUNCH(what)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      amk_set_error("%s: %s", what, hipGetErrorString(e_));                 \
      return AMK_ELAUNCH;                                                   \
    }                                                                       \
  } while (0)

// ---- device side ---------------------------------------------------------
// v_mfma_f32_32x32x2_f32: exact-f32 matrix FMA, D(32x32) += A(32x2) * B(2x32).
//   lane l supplies A[l & 31][l >> 5] and B[l >> 5][l & 31];
//   accumulator register r of lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Row of accumulator register r for lane half hf (0/1) in the 32x32 C/D tile.
__device__ __forceinline__ constexpr int acc_row(int r, int hf) {
  return (r & 3) + 8 * (r >> 2) + 4 * hf;
}

// The 8 XCDs are dealt workgroups round-robin; remap so that logically
// consecutive ids (which share K/V panels or codebook tiles) land on one XCD's L2.
// Bijective for any grid size (cdna guide, section 5 "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float wave_xor32_max(float x) {
  return fmaxf(x, __shfl_xor(x, 32, 64));
}
