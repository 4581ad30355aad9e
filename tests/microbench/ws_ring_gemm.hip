// Diagnostic only (not part of the product library).  VERDICT r2 item 1: does WAVE SPECIALISATION hide the operand fetch of
// the LDS-tiled fp32 GEMM?  NL loader waves per block own all address arithmetic and issue global_load_lds_dwordx4 into an
// S-slot LDS ring S-1 K-steps ahead; the MFMA waves do ds_read + MFMA only (no VMEM instruction, no vmcnt wait in their
// stream).  One raw s_barrier per K-step; the loader drains with a counted s_waitcnt vmcnt(N).  NL = 0: the same ring with
// every MFMA wave issuing its share of the DMA pieces (the non-specialised LDS-DMA ring of round 1).
// LDS image: rows of BK floats, no padding (a DMA piece is 1 KiB, lane-linear); bank conflicts are removed by an XOR swizzle
// of the 16-byte slot index with f(row) = (row / rows-per-256-B) mod slots-per-row, applied to the per-lane SOURCE address
// of the DMA and to the fragment ds_read_b128 (cdna_hip_programming.md rule 21).
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/ws_ring_gemm.hip -o tests/microbench/ws_ring_gemm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define GLDS16(gptr, lptr)                                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),               \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
// the same with a cache-policy immediate (gfx940+: 1 = sc0, 2 = nt, 16 = sc1) -- fill-rate experiments
#define GLDS16_AUX(gptr, lptr, AUX_)                                                                    \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),               \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, AUX_)

__device__ unsigned long long g_stall[8];   // ... [4] loader in s_barrier, [5] loader issuing, [6] MFMA wave NM-1 in s_barrier   // DIAG 4: [0] cycles MFMA waves spent in s_barrier, [1] their K-loop cycles, [2] loader vmcnt-wait cycles, [3] loader loop cycles
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DIAG (timing diagnostics, results are wrong for DIAG != 0): 1 = no s_barrier in the K loops (loader and MFMA waves run free),
// 2 = no s_barrier and no DMA (LDS read + MFMA only: the ceiling of the consumer side), 3 = barriers kept, no DMA
template <int BM, int BN, int WM, int WN, int BK, int S, int NL, bool M16, int DIAG = 0, int AUXA = 0, int AUXB = 0>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NL)) void ws_gemm(const float* __restrict__ A,
                                                                             const float* __restrict__ B,
                                                                             float* __restrict__ C, int M, int N, int K) {
  constexpr int NM = (BM / WM) * (BN / WN);       // MFMA waves
  constexpr int RB = BK * 4;                      // bytes per tile row
  constexpr int SPR = BK / 4;                     // 16-byte slots per row
  constexpr int RPP = 1024 / RB;                  // rows per DMA piece (one wave instruction = 1 KiB)
  constexpr int R256 = 256 / RB;                  // rows per 256-byte bank row
  constexpr int PA = BM / RPP, PB = BN / RPP, PT = PA + PB;
  constexpr int SLOT = (BM + BN) * RB;
  constexpr int NLW = NL > 0 ? NL : NM;           // waves that issue DMA pieces
  constexpr int PPL = (PT + NLW - 1) / NLW;       // pieces per issuing wave and K-step
  static_assert(PT % NLW == 0, "pieces must divide over the issuing waves");
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int TS = M16 ? 16 : 32;
  constexpr int MT = WM / TS, NT = WN / TS;
  constexpr int NQ = M16 ? BK / 16 : BK / 8;
  constexpr int WAVES_N = BN / WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / BN;
  const int L = blockIdx.x, m_lo = L & 7, r8 = L >> 3;
  const int tn = r8 % ntn, tm = (r8 / ntn) * 8 + m_lo;
  const int nk = K / BK;

  // ---- DMA issue (loader waves, or every wave when NL == 0) ----
  const int lw = NL > 0 ? wave : wave;            // index among the issuing waves
  const bool issues = NL == 0 || wave < NL;
  unsigned voff[PPL];                             // per-lane byte offset of each piece's source row / slot (K-step 0)
  const char* gbase[PPL];
  int loff[PPL];
  if (issues) {
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
      const int p = lw + i * NLW;                 // piece index: [0, PA) = A, [PA, PT) = B
      const bool isA = p < PA;
      const int pr = (isA ? p : p - PA) * RPP + lane / SPR;        // row inside the tile
      const int c = (lane % SPR) ^ ((pr / R256) % SPR);            // logical slot stored at this lane's physical slot
      const size_t grow = isA ? (size_t)tm * BM + pr : (size_t)tn * BN + pr;
      gbase[i] = reinterpret_cast<const char*>(isA ? A : B);
      voff[i] = (unsigned)((grow * K + c * 4) * 4);
      loff[i] = (isA ? 0 : BM * RB) + (isA ? p : p - PA) * 1024;
    }
  }
  auto issue = [&](int ks) {
    const int t = ks % S;
#pragma unroll
    for (int i = 0; i < PPL; ++i)
      if (DIAG < 2 || DIAG == 4) {
        // (piece index lw + i * NLW < PA: an A piece -- compile-time for NLW | PA)
        if ((i * NLW) < PA) GLDS16_AUX(gbase[i] + (size_t)ks * RB + voff[i], smem + t * SLOT + loff[i], AUXA);
        else GLDS16_AUX(gbase[i] + (size_t)ks * RB + voff[i], smem + t * SLOT + loff[i], AUXB);
      }
  };

  if (NL > 0 && wave < NL) {
    // ================= loader wave =================
    for (int ks = 0; ks < S - 1 && ks < nk; ++ks) issue(ks);
    long long t_wait = 0, t_loop = 0, t_lbar = 0, t_iss = 0;
    if (DIAG == 4) t_loop = -(long long)__builtin_readcyclecounter();
    for (int k = 0; k < nk; ++k) {
      if (DIAG == 4) t_wait -= (long long)__builtin_readcyclecounter();
      if (k + S - 1 <= nk) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>();
      if (DIAG == 4) { const long long c = (long long)__builtin_readcyclecounter(); t_wait += c; t_lbar -= c; }
      if (DIAG == 0 || DIAG >= 3) __builtin_amdgcn_s_barrier();
      if (DIAG == 4) { const long long c = (long long)__builtin_readcyclecounter(); t_lbar += c; t_iss -= c; }
      if (k + S - 1 < nk) issue(k + S - 1);
      if (DIAG == 4) t_iss += (long long)__builtin_readcyclecounter();
    }
    if (DIAG == 4 && wave == 0 && lane == 0) {
      t_loop += (long long)__builtin_readcyclecounter();
      atomicAdd(&g_stall[2], (unsigned long long)t_wait);
      atomicAdd(&g_stall[3], (unsigned long long)t_loop);
      atomicAdd(&g_stall[4], (unsigned long long)t_lbar);
      atomicAdd(&g_stall[5], (unsigned long long)t_iss);
    }
    return;
  }
  // ================= MFMA waves =================
  const int mw = wave - NL;
  const int wm = mw / WAVES_N, wn = mw % WAVES_N;
  typedef float accv_t __attribute__((ext_vector_type(M16 ? 4 : 16)));
  accv_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < (M16 ? 4 : 16); ++e) acc[mt][nt][e] = 0.f;
  const int r = M16 ? (lane & 15) : (lane & 31), hq = M16 ? (lane >> 4) : (lane >> 5);
  const int fr = (r / R256) % SPR;
  int foff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) foff[q] = r * RB + 16 * (((M16 ? 4 * q : 2 * q) + hq) ^ fr);
  const int abase = (wm * WM) * RB, bbase = BM * RB + (wn * WN) * RB;

  if (NL == 0) for (int ks = 0; ks < S - 1 && ks < nk; ++ks) issue(ks);
  long long t_bar = 0, t_all = 0;
  if (DIAG == 4) t_all = -(long long)__builtin_readcyclecounter();
  for (int k = 0; k < nk; ++k) {
    if (NL == 0) { if (k + S - 1 <= nk) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>(); }
    if (DIAG == 4) t_bar -= (long long)__builtin_readcyclecounter();
    if (DIAG == 0 || DIAG >= 3) __builtin_amdgcn_s_barrier();
    if (DIAG == 4) t_bar += (long long)__builtin_readcyclecounter();
    if (NL == 0) { if (k + S - 1 < nk) issue(k + S - 1); }
    const char* sl = smem + (k % S) * SLOT;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(sl + abase + mt * TS * RB + foff[q]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(sl + bbase + nt * TS * RB + foff[q]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (M16) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
            else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
          }
    }
  }
  if (DIAG == 4 && mw == 0 && lane == 0) {
    t_all += (long long)__builtin_readcyclecounter();
    atomicAdd(&g_stall[0], (unsigned long long)t_bar);
    atomicAdd(&g_stall[1], (unsigned long long)t_all);
  }
  if (DIAG == 4 && mw == NM - 1 && lane == 0) atomicAdd(&g_stall[6], (unsigned long long)t_bar);
  if constexpr (M16) {
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = tm * BM + wm * WM + mt * 16 + g * 4 + e;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * BN + wn * WN + nt * 16 + r] = acc[mt][nt][e];
      }
  } else {
    const int h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tm * BM + wm * WM + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + tn * BN + wn * WN + nt * 32 + r] = acc[mt][nt][e];
      }
  }
}

// ---- reference structure: the product's register-staged single-stage loop (16x16x4, XCD-aware order) ----
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void lds_gemm_ref(const float* __restrict__ A, const float* __restrict__ B,
                                                    float* __restrict__ C, int M, int N, int K) {
  constexpr int BK = 32, GRANS = 8, RPP = 32, LD = 40;
  constexpr int WAVES_N = BN / WN;
  constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
  constexpr int MT = WM / 16, NT = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);
  float* Bs = As + BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int ntn = N / BN;
  const int L = blockIdx.x, m_lo = L & 7, r8 = L >> 3;
  const int tn = r8 % ntn, tm = (r8 / ntn) * 8 + m_lo;
  const int lrow = tid / GRANS, gran = tid % GRANS;
  f32x4 ra[A_LOADS], rb[B_LOADS];
  const float* ap = A + (size_t)(tm * BM + lrow) * K + gran * 4;
  const float* bp = B + (size_t)(tn * BN + lrow) * K + gran * 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag = (lane & 15) * LD + (lane >> 4) * 4;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap + (size_t)(RPP * i) * K + (k0 < K ? k0 : 0));
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) rb[j] = *reinterpret_cast<const f32x4*>(bp + (size_t)(RPP * j) * K + (k0 < K ? k0 : 0));
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) *reinterpret_cast<f32x4*>(&As[(lrow + RPP * i) * LD + gran * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) *reinterpret_cast<f32x4*>(&Bs[(lrow + RPP * j) * LD + gran * 4]) = rb[j];
  };
  gload(0);
  lstore();
  __syncthreads();
  for (int k = 0; k < K; k += BK) {
    gload(k + BK);
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
    lstore();
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
  for (int t = 0; t < 512; ++t) {
    int m = (int)((1103515245u * (unsigned)t + 12345u) % (unsigned)s.M), n = (int)((69069u * (unsigned)t + 1u) % (unsigned)s.N);
    double ref = 0;
    for (int k = 0; k < s.K; ++k) ref += (double)A[(size_t)m * s.K + k] * B[(size_t)n * s.K + k];
    worst = fmax(worst, fabs(ref - C[(size_t)m * s.N + n]) / (fabs(ref) + 1e-3));
  }
  return worst;
}

int main(int argc, char** argv) {
  Shape shapes[] = {{65536, 128, 1024}, {262144, 64, 512}, {65536, 256, 2048}, {16384, 128, 1024}, {6400, 2048, 256},
                    {262144, 32, 512}, {25600, 256, 2048}, {1024, 512, 512}};
  const int rounds = argc > 1 ? atoi(argv[1]) : 2;
  const bool quick = argc > 2;
  for (int round = 0; round < rounds; ++round)
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
        printf("M=%7d N=%4d K=%4d  %-34s %8.1f us  %6.1f TF/s  err %.1e\n", s.M, s.N, s.K, name, ms * 1e3, fl / ms / 1e9,
               check(hA, hB, hC.data(), s));
        CK(hipMemset(C, 0, hC.size() * 4));
        fflush(stdout);
      };
#define RUNREF(BM, BN, WM, WN)                                                                                        \
  if (s.M % (BM * 8) == 0 && s.N % BN == 0) {                                                                         \
    const size_t smem = (size_t)(BM + BN) * 40 * 4;                                                                   \
    report("ref regstage " #BM "x" #BN, time_ms([&] {                                                                 \
      hipLaunchKernelGGL((lds_gemm_ref<BM, BN, WM, WN>), dim3((s.M / BM) * (s.N / BN)), dim3(256), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
#define RUNWS(BM, BN, WM, WN, BK_, S_, NL_, M16_)                                                                     \
  if (s.M % (BM * 8) == 0 && s.N % BN == 0 && s.K % BK_ == 0) {                                                       \
    const size_t smem = (size_t)(BM + BN) * BK_ * 4 * S_;                                                             \
    constexpr int thr = 64 * ((BM / WM) * (BN / WN) + NL_);                                                           \
    CK(hipFuncSetAttribute((const void*)ws_gemm<BM, BN, WM, WN, BK_, S_, NL_, M16_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    report("ws " #BM "x" #BN " w" #WM "x" #WN " bk" #BK_ " s" #S_ " nl" #NL_ " m16=" #M16_, time_ms([&] {            \
      hipLaunchKernelGGL((ws_gemm<BM, BN, WM, WN, BK_, S_, NL_, M16_>), dim3((s.M / BM) * (s.N / BN)), dim3(thr), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
#define RUNDIAG(BM, BN, WM, WN, S_, D_)                                                                               \
  if (s.M % (BM * 8) == 0 && s.N % BN == 0) {                                                                         \
    const size_t smem = (size_t)(BM + BN) * 32 * 4 * S_;                                                              \
    constexpr int thr = 64 * ((BM / WM) * (BN / WN) + 2);                                                             \
    CK(hipFuncSetAttribute((const void*)ws_gemm<BM, BN, WM, WN, 32, S_, 2, true, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    report("ws " #BM "x" #BN " w" #WM "x" #WN " s" #S_ " nl2 DIAG " #D_, time_ms([&] {                                 \
      hipLaunchKernelGGL((ws_gemm<BM, BN, WM, WN, 32, S_, 2, true, D_>), dim3((s.M / BM) * (s.N / BN)), dim3(thr), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
#define RUNAUX(AA, AB)                                                                                                \
  if (s.M % 512 == 0 && s.N % 64 == 0) {                                                                              \
    const size_t smem = (size_t)(64 + 64) * 32 * 4 * 3;                                                               \
    CK(hipFuncSetAttribute((const void*)ws_gemm<64, 64, 32, 32, 32, 3, 2, true, 0, AA, AB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    report("ws 64x64 s3 nl2 cache policy A=" #AA " B=" #AB, time_ms([&] {                                              \
      hipLaunchKernelGGL((ws_gemm<64, 64, 32, 32, 32, 3, 2, true, 0, AA, AB>), dim3((s.M / 64) * (s.N / 64)), dim3(384), smem, 0, A, B, C, s.M, s.N, s.K); }, 20)); \
  }
      if (argc > 2 && argv[2][0] == 't') {          // tile shapes at warm clocks: three interleaved rounds after a long warm-up
        // (the kernels' block order deals M-tiles in groups of eight: same guard as RUNREF / RUNWS)
        for (int w = 0; w < 150 && s.M % 512 == 0 && s.N % 64 == 0; ++w)
          hipLaunchKernelGGL((lds_gemm_ref<64, 64, 32, 32>), dim3((s.M / 64) * (s.N / 64)), dim3(256), (size_t)128 * 40 * 4, 0, A, B, C, s.M, s.N, s.K);
        CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 3 && s.M % 512 == 0; ++rep) {
          RUNWS(64, 64, 32, 32, 32, 3, 2, true)
          RUNWS(128, 64, 64, 32, 32, 3, 2, true)
          RUNWS(128, 64, 32, 64, 32, 3, 2, true)
          RUNWS(64, 128, 32, 64, 32, 3, 2, true)
          RUNWS(64, 128, 64, 32, 32, 3, 2, true)
          RUNWS(128, 128, 64, 32, 32, 3, 2, true)
        }
      } else if (argc > 2 && argv[2][0] == 'c') {
        RUNAUX(0, 0)
        RUNAUX(0, 0)
        RUNAUX(2, 0)
        RUNAUX(2, 2)
        RUNAUX(1, 0)
        RUNAUX(1, 1)
        RUNAUX(16, 0)
        RUNAUX(16, 16)
        RUNAUX(17, 0)
        RUNAUX(3, 0)
        RUNAUX(0, 0)
      } else if (argc > 2 && argv[2][0] == 'd') {
        RUNDIAG(64, 64, 32, 32, 3, 0)
        RUNDIAG(64, 64, 32, 32, 3, 1)
        RUNDIAG(64, 64, 32, 32, 3, 2)
        RUNDIAG(64, 64, 32, 32, 3, 3)
        {
          unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hst[8];
          CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stall), z, sizeof(z)));
          RUNDIAG(64, 64, 32, 32, 3, 4)
          CK(hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_stall), sizeof(hst)));
          printf("    64x64 s3: MFMA wave 0 / 3 in s_barrier %.1f / %.1f %% of the K loop; loader 0: vmcnt wait %.1f %%, s_barrier %.1f %%, issuing %.1f %% (mean loop %.0f cycles per K-step)\n",
                 100.0 * hst[0] / hst[1], 100.0 * hst[6] / hst[1], 100.0 * hst[2] / hst[3], 100.0 * hst[4] / hst[3], 100.0 * hst[5] / hst[3],
                 (double)hst[3] / ((s.M / 64) * (s.N / 64)) / (s.K / 32) / 23.0);
          CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stall), z, sizeof(z)));
          RUNDIAG(64, 64, 32, 32, 5, 4)
          CK(hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_stall), sizeof(hst)));
          printf("    64x64 s5: MFMA wave in s_barrier %.1f %% of its K loop; loader in vmcnt wait %.1f %% of its loop\n",
                 100.0 * hst[0] / hst[1], 100.0 * hst[2] / hst[3]);
        }
        RUNDIAG(64, 64, 32, 32, 4, 0)
        RUNDIAG(64, 64, 32, 32, 5, 0)
        RUNDIAG(64, 64, 32, 32, 3, 0)
        RUNDIAG(64, 64, 32, 32, 4, 0)
        RUNDIAG(64, 64, 32, 32, 5, 0)
        RUNDIAG(64, 64, 32, 32, 3, 0)
        RUNDIAG(128, 128, 64, 32, 3, 0)
        RUNDIAG(128, 128, 64, 32, 3, 1)
        RUNDIAG(128, 128, 64, 32, 3, 2)
        RUNDIAG(128, 128, 64, 32, 3, 3)
      } else if (quick) {
        RUNREF(64, 64, 32, 32)
        RUNWS(64, 64, 32, 32, 32, 3, 2, true)
        RUNWS(64, 64, 32, 32, 32, 4, 2, true)
        RUNWS(128, 64, 64, 32, 32, 3, 2, true)
        RUNWS(128, 64, 64, 32, 32, 3, 2, false)
        RUNWS(128, 64, 64, 32, 32, 4, 2, true)
        RUNWS(128, 64, 32, 64, 32, 3, 2, true)
        RUNWS(128, 32, 32, 32, 32, 3, 2, true)
        RUNWS(128, 32, 32, 32, 32, 4, 2, true)
        RUNWS(256, 32, 64, 32, 32, 3, 2, true)
        RUNWS(256, 64, 64, 64, 32, 3, 2, true)
        RUNWS(128, 128, 64, 32, 32, 3, 2, true)
      } else {
      RUNREF(64, 64, 32, 32)
      RUNREF(128, 128, 64, 64)
      // 64x64 block tile, 4 MFMA waves of 32x32
      RUNWS(64, 64, 32, 32, 32, 3, 1, true)
      RUNWS(64, 64, 32, 32, 32, 3, 1, false)
      RUNWS(64, 64, 32, 32, 32, 3, 2, true)
      RUNWS(64, 64, 32, 32, 32, 3, 0, true)
      RUNWS(64, 64, 32, 32, 32, 4, 1, true)
      RUNWS(64, 64, 32, 32, 16, 4, 1, true)
      RUNWS(64, 64, 32, 32, 16, 4, 0, true)
      RUNWS(64, 64, 32, 32, 16, 6, 1, true)
      // 128x64, 4 MFMA waves of 64x32 / 8 MFMA waves of 32x32
      RUNWS(128, 64, 64, 32, 32, 3, 1, true)
      RUNWS(128, 64, 64, 32, 32, 3, 2, true)
      RUNWS(128, 64, 32, 32, 32, 3, 2, true)
      RUNWS(128, 64, 64, 32, 16, 4, 1, true)
      // 128x128, 4 MFMA waves of 64x64 / 8 of 64x32
      RUNWS(128, 128, 64, 64, 32, 3, 2, true)
      RUNWS(128, 128, 64, 64, 32, 3, 2, false)
      RUNWS(128, 128, 64, 64, 16, 4, 2, true)
      RUNWS(128, 128, 64, 32, 32, 3, 2, true)
      RUNWS(128, 128, 64, 64, 32, 3, 0, true)
      }
      CK(hipFree(A));
      CK(hipFree(B));
      CK(hipFree(C));
    }
  return 0;
}
