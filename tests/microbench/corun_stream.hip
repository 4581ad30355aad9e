// Diagnostic only (not part of the product library): how much does a memory-bound streaming kernel on a second stream
// slow the product's implicit-GEMM launches down, and does the way the streaming kernel touches memory matter?
// Stream A: 20 x mmdyn_igemm_nt (CONV 16x16x64 -> 8x8x128 on 4 x 256 samples).  Stream B: a read-modify-write pass over
// 2 x 64 MB, repeated to ~half of A's solo time.  Variants of the streaming kernel: plain loads/stores, non-temporal
// loads/stores (streaming data should not displace the GEMM's operands in L2), and both with the grid capped (fewer waves
// in flight = shallower memory queues).
//   hipcc -O3 --offload-arch=gfx950 -Iinclude tests/microbench/corun_stream.hip -o tests/microbench/corun_stream.bin \
//         -Lmultimodal-dynamics_amd/mmdyn_hip -lmmdyn_hip -Wl,-rpath,'$ORIGIN/../../multimodal-dynamics_amd/mmdyn_hip'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mmdyn_hip.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    v = v * 1.0001f + 0.5f;
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}

int main() {
  const int G = 4, Bg = 256, Hi = 16, Cin = 64, Ho = 8, N = 128;
  const long rowsA = (long)G * Bg * Hi * Hi, rowsC = (long)G * Bg * Ho * Ho;
  float *A, *Bp, *C, *S0, *S1;
  const long n4 = 16l << 20;                    // 64 MB in, 64 MB out per pass
  CK(hipMalloc(&A, rowsA * Cin * 4)); CK(hipMalloc(&Bp, 16l * N * Cin * 4)); CK(hipMalloc(&C, rowsC * N * 4));
  CK(hipMalloc(&S0, n4 * 16)); CK(hipMalloc(&S1, n4 * 16));
  CK(hipMemset(A, 0, rowsA * Cin * 4)); CK(hipMemset(Bp, 0, 16l * N * Cin * 4)); CK(hipMemset(S0, 0, n4 * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, f0, f1, go;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&go));
  const int NG = 20;
  auto gemms = [&]() {
    for (int i = 0; i < NG; ++i)
      if (mmdyn_igemm_nt(A, Bp, nullptr, C, nullptr, nullptr, nullptr, 1, G, Bg, Hi, Hi, Cin, Ho, Ho, N, N, 2, -1, 0, 1, sa)) { printf("igemm failed\n"); exit(1); }
  };
  auto streams = [&](bool nt, int grid, int reps) {
    for (int i = 0; i < reps; ++i) {
      if (nt) hipLaunchKernelGGL(stream_kernel<true>, dim3(grid), dim3(256), 0, sb, (const f32x4*)S0, (f32x4*)S1, n4);
      else hipLaunchKernelGGL(stream_kernel<false>, dim3(grid), dim3(256), 0, sb, (const f32x4*)S0, (f32x4*)S1, n4);
    }
  };
  // solo GEMMs
  float tg = 0;
  for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0, sa)); gemms(); CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa)); CK(hipEventElapsedTime(&tg, e0, e1)); }
  printf("GEMMs alone: %d launches %.3f ms (%.1f us each)\n", NG, tg, tg * 1e3 / NG);
  const int grids[] = {16384, 4096, 1024, 512, 256};
  for (int nt = 0; nt < 2; ++nt)
    for (int grid : grids) {
      // solo streaming: one pass
      float t1 = 0;
      for (int r = 0; r < 3; ++r) { CK(hipEventRecord(f0, sb)); streams(nt, grid, 4); CK(hipEventRecord(f1, sb)); CK(hipStreamSynchronize(sb)); CK(hipEventElapsedTime(&t1, f0, f1)); }
      t1 /= 4;
      const int reps = (int)(0.5f * tg / t1 + 0.5f) > 0 ? (int)(0.5f * tg / t1 + 0.5f) : 1;
      float ts = t1 * reps, ta = 0, tb = 0, tall = 0;
      for (int r = 0; r < 3; ++r) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(go, sa)); CK(hipStreamWaitEvent(sb, go, 0));
        CK(hipEventRecord(e0, sa)); CK(hipEventRecord(f0, sb));
        gemms(); streams(nt, grid, reps);
        CK(hipEventRecord(e1, sa)); CK(hipEventRecord(f1, sb));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tb, f0, f1));
        float x = 0, y = 0; CK(hipEventElapsedTime(&x, e0, f1)); CK(hipEventElapsedTime(&y, e0, e1)); tall = x > y ? x : y;
      }
      printf("%s grid %5d: pass alone %6.1f us (%.2f TB/s); %2d passes alone %.3f ms | together: GEMMs %.3f ms, streaming %.3f ms, "
             "both done after %.3f ms  (serial %.3f; hidden %.0f %% of the streaming time)\n", nt ? "non-temporal" : "plain       ", grid,
             t1 * 1e3, 2.0 * n4 * 16 / (t1 * 1e-3) / 1e12, reps, ts, ta, tb, tall, tg + ts, 100.0 * (tg + ts - tall) / ts);
    }
  return 0;
}
