// Shared device/host helpers for libmmdyn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mmdyn_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MMDYN_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
    return MMDYN_OK;                              \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 64-lane wavefront sum (CDNA wave = 64; never 32)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// d/dx [x*sigmoid(x)] = s * (1 + x*(1-s))
__device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == MMDYN_ACT_SWISH) return swishf_(x);
  if (act == MMDYN_ACT_RELU) return x > 0.f ? x : 0.f;
  return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == MMDYN_ACT_SWISH) return swish_gradf_(x);
  if (act == MMDYN_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  return 1.f;
}

// grid size for a grid-stride element-wise launch: enough blocks to fill 256 CUs x 8, no more
static inline int ew_grid(int64_t work_items, int block = 256) {
  int64_t b = ceil_div64(work_items, block);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}
