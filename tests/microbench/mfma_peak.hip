// Diagnostic only (not part of the product library): sustained fp32 MFMA rate and in-kernel clock on this device.
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// same loop fed with pseudo-random full-mantissa operands (8 rotating register pairs per lane): switching activity
// like real data, which is what sets the sustained clock under load
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_rand(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float xs[8], ys[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int k = 0; k < 8; ++k) {
    h = h * 1664525u + 1013904223u;
    xs[k] = ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.01f;
    h = h * 1664525u + 1013904223u;
    ys[k] = ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.01f;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[k], ys[(k + a) & 7], acc[a], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = (float)(threadIdx.x % 7) * 0.125f, y = (float)(threadIdx.x % 5) * 0.25f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int NACC>
void run(int blocks_per_cu, int iters, bool rnd = false) {
  int blocks = 256 * blocks_per_cu;
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    if (rnd)
      hipLaunchKernelGGL(mfma_loop_rand<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    else
      hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long h[4];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double flops = (double)blocks * 4 * iters * NACC * 4096.0;
  printf("%s NACC=%d blocks/CU=%d iters=%d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", rnd ? "random  " : "constant", NACC, blocks_per_cu, iters,
         ms, flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
  hipFree(out);
  hipFree(clk);
}

int main() {
  run<4>(1, 20000);
  run<4>(2, 10000);
  run<2>(1, 40000);
  run<1>(1, 40000);
  run<1>(2, 40000);
  run<4>(1, 200000);
  run<4>(1, 200000, true);
  run<4>(2, 200000, true);
  run<2>(2, 400000, true);
  return 0;
}
