// Diagnostic: does an out-of-range buffer_load ... lds (LDS-DMA) write ZEROS to LDS, or leave the bytes untouched?
// (the wave-specialised implicit GEMM wants the hardware range check to zero-fill out-of-image rows)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ A, float* C, int nbytes, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // poison LDS
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = -7.f;
  __syncthreads();
  if (wave == 0) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, nbytes, 0x00020000);
    // odd lanes out of range (voffset = 0xFFFFFF00), even lanes in range
    const unsigned voff = (lane & 1) ? 0xFFFFFF00u : (unsigned)(lane * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem), 16, voff, 0, 0, 0);
    // second piece: in-range voffset + scalar offset (is soffset range-checked?  lanes >= 32 go past the end with it)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 1024), 16, lane * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) C[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
  float *A, *C;
  const int n = 1024;            // floats in the buffer: 4096 bytes
  CK(hipMalloc(&A, n * 4 * 2));  // (twice the declared size, so a missing range check reads valid memory)
  CK(hipMalloc(&C, 512 * 4));
  float h[2048];
  for (int i = 0; i < 2048; ++i) h[i] = (float)(i + 1);
  CK(hipMemcpy(A, h, sizeof(h), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096, 0, A, C, n * 4, 3584);
  CK(hipDeviceSynchronize());
  float o[512];
  CK(hipMemcpy(o, C, sizeof(o), hipMemcpyDeviceToHost));
  printf("piece 0 (odd lanes out of range by voffset): lane0 %.0f %.0f | lane1 %.0f %.0f | lane2 %.0f | lane3 %.0f\n", o[0], o[1], o[4], o[5], o[8], o[12]);
  int zeros = 0, poison = 0, other = 0;
  for (int l = 1; l < 64; l += 2) for (int j = 0; j < 4; ++j) { float v = o[l * 4 + j]; if (v == 0.f) ++zeros; else if (v == -7.f) ++poison; else ++other; }
  printf("  out-of-range lanes: %d zeros, %d untouched (poison), %d other\n", zeros, poison, other);
  printf("piece 1 (soffset 3584 B, buffer 4096 B: lanes >= 32 past the end): lane0 %.0f lane31 %.0f lane32 %.0f lane63 %.0f\n",
         o[256], o[256 + 31 * 4], o[256 + 32 * 4], o[256 + 63 * 4]);
  return 0;
}
