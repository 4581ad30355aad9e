// Diagnostic only (not part of the product library): is an fp32-MFMA GEMM faster when every wave is independent --
// operand fragments loaded straight from global memory into registers, no LDS, no block barrier?
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/direct_mfma.hip -o /tmp/direct_mfma && /tmp/direct_mfma
// C[m][n] = sum_k A[m][k] * B[n][k]   (NT form, both operands K-contiguous: the implicit GEMM's DENSE mode)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- 32x32x2: lane (r = l&31, h = l>>5) holds A[row r][k = h]; a 16-byte load at k0 + 8q + 4h feeds MFMAs j = 0..3
//      with k = {k0+8q+j, k0+8q+4+j} (A and B share the permutation)
template <int MT, int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void direct32(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ntn = N / (32 * NT);
  const int wt = blockIdx.x * WAVES + wave;
  const int tm = wt / ntn, tn = wt - tm * ntn;
  if (tm * 32 * MT >= M) return;
  const float* ap[MT];
  const float* bp[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ap[mt] = A + (size_t)(tm * 32 * MT + mt * 32 + r) * K + 4 * h;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = B + (size_t)(tn * 32 * NT + nt * 32 + r) * K + 4 * h;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
  f32x4 a0[4][MT], b0[4][NT], a1[4][MT], b1[4][NT];
#define LOAD(AA, BB, k0)                                                                   \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int q = 0; q < 4; ++q)   \
      AA[q][mt] = *reinterpret_cast<const f32x4*>(ap[mt] + (k0) + 8 * q);                  \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int q = 0; q < 4; ++q)   \
      BB[q][nt] = *reinterpret_cast<const f32x4*>(bp[nt] + (k0) + 8 * q);
#define COMPUTE(AA, BB)                                                                    \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int j = 0; j < 4; ++j)       \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AA[q][mt][j], BB[q][nt][j], acc[mt][nt], 0, 0, 0);
  LOAD(a0, b0, 0);
  for (int k = 0; k < K; k += 64) {
    const int k1 = k + 32;                       // K % 64 == 0
    LOAD(a1, b1, k1);
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE(a0, b0);
    const int k2 = (k + 64 < K) ? k + 64 : 0;    // last prefetch: harmless repeat of a valid tile
    LOAD(a0, b0, k2);
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE(a1, b1);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = tm * 32 * MT + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * 32 * NT + nt * 32 + r] = acc[mt][nt][e];
    }
#undef LOAD
#undef COMPUTE
}

// ---- 16x16x4: lane (r = l&15, g = l>>4) holds A[row r][k = g]; a 16-byte load at k0 + 4g feeds MFMAs j = 0..3 with
//      k = {k0 + 4g' + j, g' = 0..3}: 16 k per load.  Wave tile (16*MT) x (16*NT), 4 accumulator registers per 16x16 tile.
template <int MT, int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void direct16(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int ntn = N / (16 * NT);
  const int wt = blockIdx.x * WAVES + wave;
  const int tm = wt / ntn, tn = wt - tm * ntn;
  if (tm * 16 * MT >= M) return;
  const float* ap[MT];
  const float* bp[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ap[mt] = A + (size_t)(tm * 16 * MT + mt * 16 + r) * K + 4 * g;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bp[nt] = B + (size_t)(tn * 16 * NT + nt * 16 + r) * K + 4 * g;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a0[2][MT], b0[2][NT], a1[2][MT], b1[2][NT];
#define LOAD(AA, BB, k0)                                                                   \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int q = 0; q < 2; ++q)   \
      AA[q][mt] = *reinterpret_cast<const f32x4*>(ap[mt] + (k0) + 16 * q);                 \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int q = 0; q < 2; ++q)   \
      BB[q][nt] = *reinterpret_cast<const f32x4*>(bp[nt] + (k0) + 16 * q);
#define COMPUTE(AA, BB)                                                                    \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < 4; ++j)       \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[q][mt][j], BB[q][nt][j], acc[mt][nt], 0, 0, 0);
  LOAD(a0, b0, 0);
  for (int k = 0; k < K; k += 64) {
    LOAD(a1, b1, k + 32);
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE(a0, b0);
    const int k2 = (k + 64 < K) ? k + 64 : 0;
    LOAD(a0, b0, k2);
    __builtin_amdgcn_sched_barrier(0);
    COMPUTE(a1, b1);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = tm * 16 * MT + mt * 16 + g * 4 + e;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * 16 * NT + nt * 16 + r] = acc[mt][nt][e];
    }
#undef LOAD
#undef COMPUTE
}

// ---- 16x16x4 with buffer loads: per-lane 32-bit row offsets (MT + NT VGPRs), K position in an SGPR offset, optional
//      persistence (grid-stride over wave tiles), KS = K per register stage (16: one load per row tile, 32: two loads =
//      one full 128-byte line per row), MINW = waves per SIMD the register budget must allow
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int MT, int NT, int KS, int MINW, bool PERSIST>
__global__ __launch_bounds__(256, MINW) void dbuf16(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, int M, int N, int K, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int ntn = N / (16 * NT);
  constexpr int Q = KS / 16;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)((size_t)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (int)((size_t)N * K * 4), 0x00020000);
  for (int wt = blockIdx.x * 4 + wave; wt < ntiles; wt += PERSIST ? gridDim.x * 4 : ntiles) {
    const int tm = wt / ntn, tn = wt - tm * ntn;
    int ao[MT], bo[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ao[mt] = ((tm * 16 * MT + mt * 16 + r) * K + 4 * g) * 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bo[nt] = ((tn * 16 * NT + nt * 16 + r) * K + 4 * g) * 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a0[Q][MT], b0[Q][NT], a1[Q][MT], b1[Q][NT];
#define LOADB(AA, BB, k0)                                                                                            \
  _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int q = 0; q < Q; ++q)                   \
      AA[q][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ao[mt], (k0) * 4 + 64 * q, 0)); \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int q = 0; q < Q; ++q)                   \
      BB[q][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, bo[nt], (k0) * 4 + 64 * q, 0));
#define COMPB(AA, BB)                                                                                                \
  _Pragma("unroll") for (int q = 0; q < Q; ++q) _Pragma("unroll") for (int j = 0; j < 4; ++j)                       \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)           \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[q][mt][j], BB[q][nt][j], acc[mt][nt], 0, 0, 0);
    LOADB(a0, b0, 0);
    for (int k = 0; k < K; k += 2 * KS) {
      LOADB(a1, b1, k + KS);
      __builtin_amdgcn_sched_barrier(0);
      COMPB(a0, b0);
      const int k2 = (k + 2 * KS < K) ? k + 2 * KS : 0;
      LOADB(a0, b0, k2);
      __builtin_amdgcn_sched_barrier(0);
      COMPB(a1, b1);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = tm * 16 * MT + mt * 16 + g * 4 + e;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * 16 * NT + nt * 16 + r] = acc[mt][nt][e];
      }
#undef LOADB
#undef COMPB
  }
}

struct Shape { int M, N, K; };

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t s, e;
  CK(hipEventCreate(&s));
  CK(hipEventCreate(&e));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(s));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e));
  CK(hipEventSynchronize(e));
  float ms;
  CK(hipEventElapsedTime(&ms, s, e));
  return ms / reps;
}

static double check(const std::vector<float>& A, const std::vector<float>& B, const float* C, Shape s) {
  double worst = 0;
  for (int t = 0; t < 256; ++t) {
    int m = (int)((1103515245u * (unsigned)t + 12345u) % (unsigned)s.M), n = (int)((69069u * (unsigned)t + 1u) % (unsigned)s.N);
    double ref = 0;
    for (int k = 0; k < s.K; ++k) ref += (double)A[(size_t)m * s.K + k] * B[(size_t)n * s.K + k];
    worst = fmax(worst, fabs(ref - C[(size_t)m * s.N + n]) / (fabs(ref) + 1e-3));
  }
  return worst;
}

int main() {
  Shape shapes[] = {{65536, 128, 1024}, {262144, 64, 512}, {65536, 256, 2048}, {6400, 2048, 256}, {1048576, 32, 256}};
  for (Shape s : shapes) {
    std::vector<float> hA((size_t)s.M * s.K), hB((size_t)s.N * s.K), hC((size_t)s.M * s.N);
    unsigned x = 12345u;
    for (auto& v : hA) { x = x * 1664525u + 1013904223u; v = ((float)(x >> 8) / 16777216.0f - 0.5f); }
    for (auto& v : hB) { x = x * 1664525u + 1013904223u; v = ((float)(x >> 8) / 16777216.0f - 0.5f) * 0.2f; }
    float *A, *B, *C;
    CK(hipMalloc(&A, hA.size() * 4));
    CK(hipMalloc(&B, hB.size() * 4));
    CK(hipMalloc(&C, hC.size() * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const double fl = 2.0 * s.M * s.N * s.K;
    auto report = [&](const char* name, float ms) {
      CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
      printf("M=%7d N=%4d K=%4d  %-28s %8.1f us  %6.1f TF/s  err %.1e\n", s.M, s.N, s.K, name, ms * 1e3, fl / ms / 1e9,
             check(hA, hB, hC.data(), s));
      CK(hipMemset(C, 0, hC.size() * 4));
      fflush(stdout);
    };
#define RUN32(MT, NT, W)                                                                                   \
  if (s.N % (32 * NT) == 0 && s.M % (32 * MT) == 0) {                                                      \
    const int tiles = (s.M / (32 * MT)) * (s.N / (32 * NT));                                               \
    report("direct32 " #MT "x" #NT " w" #W, time_ms([&] {                                                  \
      hipLaunchKernelGGL((direct32<MT, NT, W>), dim3((tiles + W - 1) / W), dim3(64 * W), 0, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
#define RUN16(MT, NT, W)                                                                                   \
  if (s.N % (16 * NT) == 0 && s.M % (16 * MT) == 0) {                                                      \
    const int tiles = (s.M / (16 * MT)) * (s.N / (16 * NT));                                               \
    report("direct16 " #MT "x" #NT " w" #W, time_ms([&] {                                                  \
      hipLaunchKernelGGL((direct16<MT, NT, W>), dim3((tiles + W - 1) / W), dim3(64 * W), 0, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
    RUN32(2, 2, 4)
    RUN16(4, 4, 4)
    RUN16(8, 2, 4)
#define RUNB(MT, NT, KS, MINW, PERSIST)                                                                    \
  if (s.N % (16 * NT) == 0 && s.M % (16 * MT) == 0 && (size_t)s.M * s.K * 4 < (1ull << 31)) {              \
    const int tiles = (s.M / (16 * MT)) * (s.N / (16 * NT));                                               \
    const int grid = PERSIST ? (256 * MINW < (tiles + 3) / 4 ? 256 * MINW : (tiles + 3) / 4) : (tiles + 3) / 4; \
    report("dbuf16 " #MT "x" #NT " ks" #KS " w" #MINW " p" #PERSIST, time_ms([&] {                         \
      hipLaunchKernelGGL((dbuf16<MT, NT, KS, MINW, PERSIST>), dim3(grid), dim3(256), 0, 0, A, B, C, s.M, s.N, s.K, tiles); }, 20)); \
  }
    RUNB(4, 4, 32, 1, false)
    RUNB(4, 4, 32, 2, false)
    RUNB(4, 4, 32, 2, true)
    RUNB(4, 4, 16, 2, false)
    RUNB(4, 4, 16, 3, false)
    RUNB(4, 4, 16, 3, true)
    RUNB(8, 2, 16, 2, false)
    RUNB(8, 2, 32, 1, false)
    RUNB(4, 2, 32, 2, false)
    RUNB(4, 2, 32, 3, false)
    RUNB(4, 2, 16, 4, false)
    RUNB(2, 4, 32, 3, false)
    CK(hipFree(A));
    CK(hipFree(B));
    CK(hipFree(C));
  }
  return 0;
}
