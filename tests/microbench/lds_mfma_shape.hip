// Diagnostic only (not part of the product library): the LDS-tiled fp32 GEMM of igemm_nt.hip (dense mode, register-staged
// prefetch, single LDS stage, two barriers per K-step) with v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 at the
// same block / wave tile.  VERDICT r1 item 2: does the 16x16 shape hold a higher clock / rate in an LDS-fed loop?
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/lds_mfma_shape.hip -o tests/microbench/lds_mfma_shape.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BK = 32, GRANS = 8, RPP = 32;   // 16-byte granules per tile row, rows per load pass of 256 threads

// SHAPE 32: wave tile (WM x WN) of 32x32 MFMA tiles, LDS row stride 36 floats (conflict-free ds_read_b128 for that map)
// SHAPE 16: the same wave tile as 16x16 MFMA tiles, LDS row stride 40 floats
// DIAG (timing diagnostics, results are wrong for DIAG != 0): 1 = no block barriers in the K loop, 2 = no global loads /
// LDS stores in the K loop (the MFMAs re-read the first tile), 3 = both
template <int SHAPE, int BM, int BN, int WM, int WN, int DIAG = 0>
__global__ __launch_bounds__(256) void lds_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int LD = SHAPE == 32 ? 36 : 40;
  constexpr int WAVES_N = BN / WN;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);
  float* Bs = As + BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int ntn = N / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int lrow = tid / GRANS, gran = tid % GRANS;
  f32x4 ra[A_LOADS], rb[B_LOADS];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(tm * BM + lrow + RPP * i) * K + k0 + gran * 4);
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j)
      rb[j] = *reinterpret_cast<const f32x4*>(B + (size_t)(tn * BN + lrow + RPP * j) * K + k0 + gran * 4);
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + RPP * i) * LD + gran * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * j) * LD + gran * 4]) = rb[j];
  };
  if constexpr (SHAPE == 32) {
    constexpr int MT = WM / 32, NT = WN / 32;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
    const int frag = (lane & 31) * LD + (lane >> 5) * 4;
    gload(0);
    lds_store();
    __syncthreads();
    for (int k = 0; k < K; k += BK) {
      gload(k + BK < K ? k + BK : 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(&As[(wm * WM + mt * 32) * LD + frag + q * 8]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WN + nt * 32) * LD + frag + q * 8]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
      }
      __syncthreads();
      lds_store();
      __syncthreads();
    }
    const int h = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tm * BM + wm * WM + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * BN + wn * WN + nt * 32 + cl] = acc[mt][nt][e];
      }
  } else {
    constexpr int MT = WM / 16, NT = WN / 16;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frag = (lane & 15) * LD + (lane >> 4) * 4;
    gload(0);
    lds_store();
    __syncthreads();
    for (int k = 0; k < K; k += BK) {
      if (!(DIAG & 2) || (DIAG & 4)) gload(k + BK < K ? k + BK : 0);       // DIAG 6: loads issued, data never stored
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(&As[(wm * WM + mt * 16) * LD + frag + q * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WN + nt * 16) * LD + frag + q * 16]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
      }
      __syncthreads();
      lds_store();
      __syncthreads();
    }
    const int g = lane >> 4, r = lane & 15;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = tm * BM + wm * WM + mt * 16 + g * 4 + e;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * BN + wn * WN + nt * 16 + r] = acc[mt][nt][e];
      }
  }
}

// 16x16x4 variant with the global loads issued PF K-steps ahead (PF register sets, rotated by full unrolling): is the
// single-stage kernel exposed to memory latency?  (diag2 above says the loads cost ~20 % and the barriers nothing)
template <int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__(256) void lds_gemm_pf(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, int M, int N, int K) {
  constexpr int LD = 40;
  constexpr int WAVES_N = BN / WN;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  constexpr int MT = WM / 16, NT = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);
  float* Bs = As + BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int ntn = N / BN;
  // XCD-aware: blocks that share A rows (the N tiles of one M tile) get ids equal modulo 8
  const int L = blockIdx.x, m_lo = L & 7, r8 = L >> 3;
  const int tn = r8 % ntn, tm = (r8 / ntn) * 8 + m_lo;
  const int lrow = tid / GRANS, gran = tid % GRANS;
  f32x4 ra[PF][A_LOADS], rb[PF][B_LOADS];
  const float* ap = A + (size_t)(tm * BM + lrow) * K + gran * 4;
  const float* bp = B + (size_t)(tn * BN + lrow) * K + gran * 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag = (lane & 15) * LD + (lane >> 4) * 4;
#define GLOAD(S, k0)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                                   \
      ra[S][i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(RPP * i) * K + ((k0) < K ? (k0) : 0));   \
  _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j)                                                   \
      rb[S][j] = *reinterpret_cast<const f32x4*>(bp + (size_t)(RPP * j) * K + ((k0) < K ? (k0) : 0));
#define LSTORE(S)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                                   \
      *reinterpret_cast<f32x4*>(&As[(lrow + RPP * i) * LD + gran * 4]) = ra[S][i];                      \
  _Pragma("unroll") for (int j = 0; j < B_LOADS; ++j)                                                   \
      *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * j) * LD + gran * 4]) = rb[S][j];
  // prologue: tile 0 into LDS, tiles 1..PF-1 in flight in sets 1..PF-1 (set 0 is free again)
  GLOAD(0, 0);
  LSTORE(0);
#pragma unroll
  for (int s = 1; s < PF; ++s) { GLOAD(s, s * BK); }
  __syncthreads();
  for (int k = 0; k < K; k += PF * BK) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {            // K-step k + u*BK is in LDS; set u is free: fetch step k + (u + PF)*BK into it
      GLOAD(u, k + (u + PF) * BK);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(&As[(wm * WM + mt * 16) * LD + frag + q * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(&Bs[(wn * WN + nt * 16) * LD + frag + q * 16]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
      }
      __syncthreads();
      LSTORE((u + 1) % PF);                   // step k + (u+1)*BK, fetched PF-1 iterations ago
      __syncthreads();
    }
  }
#undef GLOAD
#undef LSTORE
  const int g = lane >> 4, r = lane & 15;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = tm * BM + wm * WM + mt * 16 + g * 4 + e;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * BN + wn * WN + nt * 16 + r] = acc[mt][nt][e];
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
  Shape shapes[] = {{65536, 128, 1024}, {262144, 64, 512}, {65536, 256, 2048}, {6400, 2048, 256}};
  for (int round = 0; round < 2; ++round)
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
        printf("M=%7d N=%4d K=%4d  %-24s %8.1f us  %6.1f TF/s  err %.1e\n", s.M, s.N, s.K, name, ms * 1e3, fl / ms / 1e9,
               check(hA, hB, hC.data(), s));
        CK(hipMemset(C, 0, hC.size() * 4));
        fflush(stdout);
      };
#define RUN(SH, BM, BN, WM, WN)                                                                                       \
  if (s.M % BM == 0 && s.N % BN == 0) {                                                                               \
    const size_t smem = (size_t)(BM + BN) * (SH == 32 ? 36 : 40) * 4;                                                 \
    report("mfma" #SH " " #BM "x" #BN, time_ms([&] {                                                                  \
      hipLaunchKernelGGL((lds_gemm<SH, BM, BN, WM, WN>), dim3((s.M / BM) * (s.N / BN)), dim3(256), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
#define RUNPF(BM, BN, WM, WN, PF_)                                                                                    \
  if (s.M % (BM * 8) == 0 && s.N % BN == 0 && s.K % (PF_ * 32) == 0) {                                               \
    const size_t smem = (size_t)(BM + BN) * 40 * 4;                                                                   \
    report("mfma16 " #BM "x" #BN " xcd pf" #PF_, time_ms([&] {                                                        \
      hipLaunchKernelGGL((lds_gemm_pf<BM, BN, WM, WN, PF_>), dim3((s.M / BM) * (s.N / BN)), dim3(256), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
#define RUND(SH, BM, BN, WM, WN, DG)                                                                                  \
  if (s.M % BM == 0 && s.N % BN == 0) {                                                                               \
    const size_t smem = (size_t)(BM + BN) * (SH == 32 ? 36 : 40) * 4;                                                 \
    report("mfma" #SH " " #BM "x" #BN " diag" #DG, time_ms([&] {                                                      \
      hipLaunchKernelGGL((lds_gemm<SH, BM, BN, WM, WN, DG>), dim3((s.M / BM) * (s.N / BN)), dim3(256), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
      RUN(32, 64, 64, 32, 32)
      RUN(16, 64, 64, 32, 32)
      RUND(16, 64, 64, 32, 32, 1)
      RUND(16, 64, 64, 32, 32, 2)
      RUND(16, 64, 64, 32, 32, 3)
      RUND(16, 64, 64, 32, 32, 6)
      RUND(16, 64, 64, 32, 32, 10)
      RUNPF(64, 64, 32, 32, 1)
      RUNPF(64, 64, 32, 32, 2)
      RUNPF(64, 64, 32, 32, 3)
      RUNPF(64, 64, 32, 32, 4)
      RUNPF(128, 64, 64, 32, 2)
      RUNPF(128, 128, 64, 64, 2)
      RUN(32, 128, 128, 64, 64)
      RUN(16, 128, 128, 64, 64)
      RUN(32, 128, 64, 64, 32)
      RUN(16, 128, 64, 64, 32)
      CK(hipFree(A));
      CK(hipFree(B));
      CK(hipFree(C));
    }
  return 0;
}
