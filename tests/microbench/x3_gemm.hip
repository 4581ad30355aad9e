// Diagnostic only (not part of the product library): fp32 GEMM C = A x B^T on the bf16 matrix cores through a three-term
// split of every fp32 operand, x = hi + mid + lo with hi, mid, lo bf16 (8 + 8 + 8 significant bits: exact), and six of the
// nine cross products (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the three dropped ones are <= 2^-23 of |a||b| together),
// accumulated in fp32.  v_mfma_f32_32x32x16_bf16 runs 16x the fp32 rate of v_mfma_f32_32x32x2_f32, so six products cost 6/16 of
// the native matrix time -- if the split (VALU, on the store side: once per element per block) and the 3x LDS traffic fit.
// Register-staged single-stage pipeline as in igemm_nt.hip.  Prints TFLOP/s and the error against fp64 next to the native
// fp32 kernel of the same structure.
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/x3_gemm.hip -o tests/microbench/x3_gemm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BK = 32, GRANS = 8, RPP = 32;
constexpr int LDH = 40;       // bf16 row stride of one plane (32 + 8 pad): 80 bytes, conflict-free ds_read_b128 for the map below

// TRUNC split: hi = x with the low 16 bits cleared, mid likewise of x - hi, lo = x - hi - mid (8 bits left: exact in bf16)
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  hi = __builtin_amdgcn_perm(u1, u0, 0x07060302);                 // (u0 >> 16) | (u1 & 0xffff0000)
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  mid = __builtin_amdgcn_perm(v1, v0, 0x07060302);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  lo = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}

// NPROD: 6 (the fp32-accurate form), 3 (hi*hi + hi*mid + mid*hi: ~2^-16), 1 (plain bf16), 9 (all products)
template <int BM, int BN, int WM, int WN, int NPROD>
__global__ __launch_bounds__(256) void x3_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K) {
  constexpr int WAVES_N = BN / WN;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int NP = NPROD == 1 ? 1 : 3;              // planes kept
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* As = reinterpret_cast<uint16_t*>(smem);            // [NP][BM][LDH]
  uint16_t* Bs = As + NP * BM * LDH;                           // [NP][BN][LDH]
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
  auto store_split = [&](uint16_t* T, int rows, int row, f32x4 v) {
    uint2 h, m, l;
    split3(v[0], v[1], h.x, m.x, l.x);
    split3(v[2], v[3], h.y, m.y, l.y);
    *reinterpret_cast<uint2*>(&T[row * LDH + gran * 4]) = h;
    if constexpr (NP == 3) {
      *reinterpret_cast<uint2*>(&T[(rows + row) * LDH + gran * 4]) = m;
      *reinterpret_cast<uint2*>(&T[(2 * rows + row) * LDH + gran * 4]) = l;
    }
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) store_split(As, BM, lrow + RPP * i, ra[i]);
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) store_split(Bs, BN, lrow + RPP * j, rb[j]);
  };
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
  const int frag = (lane & 31) * LDH + (lane >> 5) * 8;      // lane (row i, half h): k = 16m + 8h .. +7
  gload(0);
  lds_store();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    gload(k0 + BK < K ? k0 + BK : k0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < BK / 16; ++m) {
      bf16x8 pa[NP][MT], pb[NP][NT];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          pa[p][mt] = *reinterpret_cast<const bf16x8*>(&As[(p * BM + wm * WM + mt * 32) * LDH + frag + m * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          pb[p][nt] = *reinterpret_cast<const bf16x8*>(&Bs[(p * BN + wn * WN + nt * 32) * LDH + frag + m * 16]);
      }
      // smallest terms first
      constexpr int order[9][2] = {{2, 2}, {1, 2}, {2, 1}, {0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int pi = order[t][0], pj = order[t][1];
        const bool use = NPROD == 9 || (NPROD == 6 && t >= 3) || (NPROD == 3 && t >= 6) || (NPROD == 1 && t == 8);
        if (use) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[pi < NP ? pi : 0][mt], pb[pj < NP ? pj : 0][nt], acc[mt][nt], 0, 0, 0);
        }
      }
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
      const int r = tm * BM + wm * WM + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) C[(size_t)r * N + tn * BN + wn * WN + nt * 32 + cl] = acc[mt][nt][e];
    }
}

// native fp32 matrix cores, same pipeline (the reference point)
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void f32_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                float* __restrict__ C, int M, int N, int K) {
  constexpr int LD = 36;
  constexpr int WAVES_N = BN / WN;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  constexpr int MT = WM / 32, NT = WN / 32;
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
  for (int k0 = 0; k0 < K; k0 += BK) {
    gload(k0 + BK < K ? k0 + BK : k0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
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
      const int r = tm * BM + wm * WM + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) C[(size_t)r * N + tn * BN + wn * WN + nt * 32 + cl] = acc[mt][nt][e];
    }
}

struct Err { double rel_l2, max_rel; };
static Err error_vs_fp64(const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& C, int M, int N, int K) {
  double num = 0, den = 0, mx = 0;
  for (int r = 0; r < M; r += M / 48) {            // 48 sampled rows
    for (int c = 0; c < N; ++c) {
      double s = 0, sa = 0;
      for (int k = 0; k < K; ++k) {
        const double p = (double)A[(size_t)r * K + k] * (double)B[(size_t)c * K + k];
        s += p;
        sa += fabs(p);
      }
      const double d = (double)C[(size_t)r * N + c] - s;
      num += d * d;
      den += s * s;
      mx = fmax(mx, fabs(d) / sa);                 // error relative to sum |a_k b_k| (the bound fp32 summation is stated in)
    }
  }
  return {sqrt(num / den), mx};
}

template <typename F>
static void run(const char* name, F launch, const float* dA, const float* dB, float* dC, const std::vector<float>& A,
                const std::vector<float>& B, int M, int N, int K) {
  CK(hipMemset(dC, 0, (size_t)M * N * 4));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<float> C((size_t)M * N);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  const Err e = error_vs_fp64(A, B, C, M, N, K);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("M=%7d N=%4d K=%5d  %-34s %8.1f us  %7.1f TF/s (fp32-equivalent)  rel-L2 %.2e  max |err|/sum|ab| %.2e\n", M, N, K, name, us,
         2.0 * M * N * K / us * 1e-6, e.rel_l2, e.max_rel);
  fflush(stdout);
}

#define X3(BM, BN, WM, WN, NPROD)                                                                                    \
  do {                                                                                                               \
    const size_t smem = (size_t)(NPROD == 1 ? 1 : 3) * (BM + BN) * LDH * 2;                                           \
    CK(hipFuncSetAttribute((const void*)x3_gemm<BM, BN, WM, WN, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    run("x3 " #BM "x" #BN " w" #WM "x" #WN " products " #NPROD,                                                       \
        [&] { hipLaunchKernelGGL((x3_gemm<BM, BN, WM, WN, NPROD>), dim3((M / BM) * (N / BN)), dim3(256), smem, 0, dA, dB, dC, M, N, K); }, \
        dA, dB, dC, A, B, M, N, K);                                                                                  \
  } while (0)
#define F32(BM, BN, WM, WN)                                                                                          \
  do {                                                                                                               \
    const size_t smem = (size_t)(BM + BN) * 36 * 4;                                                                  \
    CK(hipFuncSetAttribute((const void*)f32_gemm<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    run("native fp32 " #BM "x" #BN " w" #WM "x" #WN,                                                                  \
        [&] { hipLaunchKernelGGL((f32_gemm<BM, BN, WM, WN>), dim3((M / BM) * (N / BN)), dim3(256), smem, 0, dA, dB, dC, M, N, K); }, \
        dA, dB, dC, A, B, M, N, K);                                                                                  \
  } while (0)

int main() {
  const int shapes[][3] = {{65536, 128, 1024}, {262144, 64, 512}, {65536, 256, 2048}, {16384, 128, 1024}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    uint32_t st = 12345u;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) * (1.0f / 8388608.0f) - 1.0f); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd() * 0.25f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    F32(64, 64, 32, 32);
    if (N % 128 == 0) F32(128, 128, 64, 64);
    X3(64, 64, 32, 32, 6);
    X3(128, 64, 64, 32, 6);
    X3(128, 64, 32, 64, 6);
    if (N % 128 == 0) X3(128, 128, 64, 64, 6);
    if (N % 128 == 0) X3(128, 128, 64, 64, 9);
    if (N % 128 == 0) X3(128, 128, 64, 64, 3);
    X3(64, 64, 32, 32, 1);
    CK(hipFree(dA));
    CK(hipFree(dB));
    CK(hipFree(dC));
  }
  return 0;
}
