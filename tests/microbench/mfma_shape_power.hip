// Diagnostic: the bf16 matrix pipe alone, on random operands, in its two shapes -- v_mfma_f32_16x16x32_bf16 against
// v_mfma_f32_32x32x16_bf16 -- with 8 waves per CU (two per SIMD) on every CU, register operands only (no LDS, no memory in the loop).
// Under the package power cap the chip lowers its clock; which shape turns a joule into more flops?
// build: hipcc -O3 --offload-arch=gfx950 mfma_shape_power.hip -o mfma_shape_power ; run: ./mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bf16x8 rnd_frag(unsigned& s) {
  u32x4 v;
  for (int i = 0; i < 4; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned a = 0x3F00u | ((s >> 9) & 0x7Fu) | ((s >> 3) & 0x8000u);      // bf16 in [0.5, 1) with a random sign
    s = s * 1664525u + 1013904223u;
    const unsigned b = 0x3F00u | ((s >> 9) & 0x7Fu) | ((s >> 3) & 0x8000u);
    v[i] = a | (b << 16);
  }
  return __builtin_bit_cast(bf16x8, v);
}

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1u;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd_frag(s); b[i] = rnd_frag(s); }
  float sum = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)          // two K = 16 halves: the same flops per iteration as the 16x16x32 loop (4x4 tiles of 16x16 x 32)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * kk + i], b[2 * kk + j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
  }
  if (sum == 12345.678f) out[0] = sum;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 4));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount, iters = 20000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int round = 0; round < 3; ++round)
    for (int shape : {16, 32}) {
      for (int w = 0; w < 2; ++w) {
        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 0, 0, out, iters);
        else hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 0, 0, out, iters);
      }
      CK(hipEventRecord(e0));
      const int reps = 20;
      for (int r = 0; r < reps; ++r) {
        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 0, 0, out, iters);
        else hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 0, 0, out, iters);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      // per wave and iteration: 16 MFMAs of 16x16x32 (= 8 of 32x32x16) = 16 * 16384 flops
      const double fl = (double)blocks * 8 * iters * 16.0 * 16384.0 * reps;
      printf("v_mfma_f32_%s_bf16: %8.2f ms per %d launches  %7.1f TFLOP/s bf16 dense = %6.1f fp32-equivalent (/6)\n",
             shape == 16 ? "16x16x32" : "32x32x16", ms, reps, fl / ms / 1e9, fl / ms / 1e9 / 6);
      fflush(stdout);
    }
  return 0;
}
