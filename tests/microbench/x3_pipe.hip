// Diagnostic only (not part of the product library): pipeline variants of the three-term-split fp32 GEMM C = A x B^T on the bf16
// matrix cores (see x3_gemm.hip for the arithmetic).  Variants: DB = 0 single LDS buffer, two barriers per K-step (the product's
// igemm_nt X3 structure); DB = 1 two LDS buffers, ONE barrier per K-step -- the split + store of step s+1 sits in the same basic
// block as the MFMAs of step s.  Block = (BM/WM)*(BN/WN) waves (4 or 8).
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/x3_pipe.hip -o tests/microbench/x3_pipe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BK = 32, GRANS = 8;
constexpr int LDH = 40;

__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 r;
  r[0] = (__bf16)lo;
  r[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = pack2_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = pack2_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = pack2_bf16(s0, s1);
}

template <int BM, int BN, int WM, int WN, int DB>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void x3_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                                      float* __restrict__ C, int M, int N, int K) {
  constexpr int WAVES_N = BN / WN, NTHR = 64 * (BM / WM) * WAVES_N;
  constexpr int RPP = NTHR / GRANS;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int STAGE = 3 * (BM + BN) * LDH;          // halves per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* S = reinterpret_cast<uint16_t*>(smem);
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
    *reinterpret_cast<uint2*>(&T[(rows + row) * LDH + gran * 4]) = m;
    *reinterpret_cast<uint2*>(&T[(2 * rows + row) * LDH + gran * 4]) = l;
  };
  auto lds_store = [&](uint16_t* As, uint16_t* Bs) {
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
  const int frag = (lane & 31) * LDH + (lane >> 5) * 8;
  auto mfma_step = [&](const uint16_t* As, const uint16_t* Bs) {
#pragma unroll
    for (int m = 0; m < BK / 16; ++m) {
      bf16x8 pa[3][MT], pb[3][NT];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          pa[p][mt] = *reinterpret_cast<const bf16x8*>(&As[(p * BM + wm * WM + mt * 32) * LDH + frag + m * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          pb[p][nt] = *reinterpret_cast<const bf16x8*>(&Bs[(p * BN + wn * WN + nt * 32) * LDH + frag + m * 16]);
      }
      constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[order[t][0]][mt], pb[order[t][1]][nt], acc[mt][nt], 0, 0, 0);
    }
  };
  gload(0);
  lds_store(S, S + 3 * BM * LDH);
  __syncthreads();
  if constexpr (DB == 0) {
    for (int k0 = 0; k0 < K; k0 += BK) {
      gload(k0 + BK < K ? k0 + BK : k0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(S, S + 3 * BM * LDH);
      __syncthreads();
      lds_store(S, S + 3 * BM * LDH);
      __syncthreads();
    }
  } else {
    gload(BK < K ? BK : 0);
    int cur = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
      uint16_t* Sc = S + cur * STAGE;
      uint16_t* Sn = S + (cur ^ 1) * STAGE;
      lds_store(Sn, Sn + 3 * BM * LDH);            // step k0 + BK (its loads were issued one iteration ago)
      gload(k0 + 2 * BK < K ? k0 + 2 * BK : k0);
      mfma_step(Sc, Sc + 3 * BM * LDH);
      __syncthreads();
      cur ^= 1;
    }
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

// WS: producer / consumer waves.  NPW producer waves load the next K-step to registers, split it and write the three planes of the
// other LDS buffer; the MFMA waves only read fragments and issue.  One raw s_barrier per K-step (no vmcnt wait in it: the producers'
// global loads stay in flight across it).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int BM, int BN, int WM, int WN, int NPW>
__global__ __launch_bounds__(64 * (NPW + (BM / WM) * (BN / WN))) void x3_ws(const float* __restrict__ A, const float* __restrict__ B,
                                                                            float* __restrict__ C, int M, int N, int K) {
  constexpr int WAVES_N = BN / WN;
  constexpr int RPP = NPW * 64 / GRANS;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int STAGE = 3 * (BM + BN) * LDH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* S = reinterpret_cast<uint16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntn = N / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int nsteps = K / BK;
  if (wave < NPW) {
    const int lrow = tid / GRANS, gran = tid % GRANS;
    f32x4 ra[2][A_LOADS], rb[2][B_LOADS];
    auto gload = [&](int set, int k0) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        ra[set][i] = *reinterpret_cast<const f32x4*>(A + (size_t)(tm * BM + lrow + RPP * i) * K + k0 + gran * 4);
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j)
        rb[set][j] = *reinterpret_cast<const f32x4*>(B + (size_t)(tn * BN + lrow + RPP * j) * K + k0 + gran * 4);
    };
    auto store_split = [&](uint16_t* T, int rows, int row, f32x4 v) {
      uint2 h, m, l;
      split3(v[0], v[1], h.x, m.x, l.x);
      split3(v[2], v[3], h.y, m.y, l.y);
      *reinterpret_cast<uint2*>(&T[row * LDH + gran * 4]) = h;
      *reinterpret_cast<uint2*>(&T[(rows + row) * LDH + gran * 4]) = m;
      *reinterpret_cast<uint2*>(&T[(2 * rows + row) * LDH + gran * 4]) = l;
    };
    auto lds_store = [&](int set, uint16_t* As) {
      uint16_t* Bs = As + 3 * BM * LDH;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) store_split(As, BM, lrow + RPP * i, ra[set][i]);
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) store_split(Bs, BN, lrow + RPP * j, rb[set][j]);
    };
    auto kk = [&](int st) { return st < nsteps ? st * BK : 0; };
    gload(0, 0);
    gload(1, kk(1));
    lds_store(0, S);                      // step 0
    gload(0, kk(2));
    lds_barrier();
    for (int s = 0; s < nsteps; s += 2) {      // (nsteps is even here)
      lds_store(1, S + STAGE);            // step s + 1 (set 1), loads of step s + 2 (set 0) in flight
      gload(1, kk(s + 3));
      lds_barrier();
      lds_store(0, S);                    // step s + 2
      gload(0, kk(s + 4));
      lds_barrier();
    }
    return;
  }
  const int cw = wave - NPW;
  const int wm = cw / WAVES_N, wn = cw % WAVES_N;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
  const int frag = (lane & 31) * LDH + (lane >> 5) * 8;
  lds_barrier();
  for (int s = 0; s < nsteps; ++s) {
    const uint16_t* As = S + (s & 1) * STAGE;
    const uint16_t* Bs = As + 3 * BM * LDH;
#pragma unroll
    for (int m = 0; m < BK / 16; ++m) {
      bf16x8 pa[3][MT], pb[3][NT];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          pa[p][mt] = *reinterpret_cast<const bf16x8*>(&As[(p * BM + wm * WM + mt * 32) * LDH + frag + m * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          pb[p][nt] = *reinterpret_cast<const bf16x8*>(&Bs[(p * BN + wn * WN + nt * 32) * LDH + frag + m * 16]);
      }
      constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[order[t][0]][mt], pb[order[t][1]][nt], acc[mt][nt], 0, 0, 0);
    }
    lds_barrier();
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

static double rel_err(const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& C, int M, int N, int K) {
  double num = 0, den = 0;
  for (int r = 0; r < M; r += M / 32) {
    for (int c = 0; c < N; ++c) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A[(size_t)r * K + k] * (double)B[(size_t)c * K + k];
      const double d = (double)C[(size_t)r * N + c] - s;
      num += d * d;
      den += s * s;
    }
  }
  return sqrt(num / den);
}

template <typename F>
static void run(const char* name, F launch, float* dC, const std::vector<float>& A, const std::vector<float>& B, int M, int N, int K) {
  CK(hipMemset(dC, 0, (size_t)M * N * 4));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<float> C((size_t)M * N);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  const double e = rel_err(A, B, C, M, N, K);
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
  printf("M=%7d N=%4d K=%5d  %-44s %8.1f us  %7.1f TF/s (fp32-equivalent)  rel-L2 %.2e\n", M, N, K, name, us, 2.0 * M * N * K / us * 1e-6, e);
  fflush(stdout);
}

#define X3(BM, BN, WM, WN, DB)                                                                                        \
  do {                                                                                                                \
    const size_t smem = (size_t)(DB + 1) * 3 * (BM + BN) * LDH * 2;                                                    \
    constexpr int NTHR = 64 * (BM / WM) * (BN / WN);                                                                  \
    CK(hipFuncSetAttribute((const void*)x3_gemm<BM, BN, WM, WN, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    run(DB ? "x3 " #BM "x" #BN " waves " #WM "x" #WN " double-buffered" : "x3 " #BM "x" #BN " waves " #WM "x" #WN " single buffer", \
        [&] { hipLaunchKernelGGL((x3_gemm<BM, BN, WM, WN, DB>), dim3((M / BM) * (N / BN)), dim3(NTHR), smem, 0, dA, dB, dC, M, N, K); }, \
        dC, A, B, M, N, K);                                                                                           \
  } while (0)

#define WS(BM, BN, WM, WN, NPW)                                                                                       \
  do {                                                                                                                \
    const size_t smem = (size_t)2 * 3 * (BM + BN) * LDH * 2;                                                           \
    constexpr int NTHR = 64 * (NPW + (BM / WM) * (BN / WN));                                                          \
    CK(hipFuncSetAttribute((const void*)x3_ws<BM, BN, WM, WN, NPW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    run("x3 " #BM "x" #BN " consumers " #WM "x" #WN " + " #NPW " producer waves",                                      \
        [&] { hipLaunchKernelGGL((x3_ws<BM, BN, WM, WN, NPW>), dim3((M / BM) * (N / BN)), dim3(NTHR), smem, 0, dA, dB, dC, M, N, K); }, \
        dC, A, B, M, N, K);                                                                                           \
  } while (0)

int main() {
  const int shapes[][3] = {{65536, 128, 1024}, {262144, 64, 512}, {16384, 128, 1024}};
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
    X3(64, 64, 32, 32, 0);
    X3(64, 64, 32, 32, 1);
    X3(128, 64, 64, 32, 0);
    X3(128, 64, 64, 32, 1);
    X3(128, 64, 32, 32, 0);
    X3(128, 64, 32, 32, 1);
    WS(64, 64, 32, 32, 2);
    WS(64, 64, 32, 32, 4);
    WS(128, 64, 64, 32, 4);
    WS(128, 64, 32, 32, 4);
    if (N % 128 == 0) {
      WS(128, 128, 64, 64, 4);
      WS(128, 128, 64, 32, 4);
      WS(128, 128, 64, 64, 2);
      WS(128, 128, 64, 32, 2);
      X3(128, 128, 64, 64, 0);
      X3(128, 128, 64, 64, 1);
      X3(128, 128, 64, 32, 0);
      X3(128, 128, 64, 32, 1);
    }
    CK(hipFree(dA));
    CK(hipFree(dB));
    CK(hipFree(dC));
  }
  return 0;
}
