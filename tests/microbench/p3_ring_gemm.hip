// Diagnostic only (not part of the product library).  VERDICT r4 item 1: "split once, not once per tile".  Stand-alone GEMM
// C = A x B^T in the fp32x3 arithmetic (exact three-term bf16 split, six of nine plane products on v_mfma_f32_16x16x32_bf16,
// fp32 accumulate) with the operands ALREADY SPLIT in HBM -- rows of [plane][K] bf16, 6 bytes per element -- and an LDS-DMA
// ring in front of MFMA waves that execute ds_read_b128 + v_mfma only (no VALU in the K loop):
//   p3_gemm   (design ii)  A and B both arrive as planes: loader waves, S-slot ring of 32-channel K-steps, one raw s_barrier per
//                          K-step, counted vmcnt -- the structure of csrc/igemm_wsp.hip with three planes per tile row;
//   p3a_gemm  (design i')  B arrives as planes (weights: split once per optimiser step by the pack kernel), A arrives as fp32:
//                          producer waves DMA it into a staging ring, split their OWN rows (once per block and K-step) and
//                          write the planes into the two-slot plane ring.
// LDS image of a K-step: per operand, 16-row blocks x 3 planes x 1 KiB pieces ([16 rows][64 B]); a piece is one DMA wave
// instruction (lane l -> row l>>2, 16-byte position l&3 holding source granule (l&3) ^ f(row), f(r) = (r>>2)&2), and one
// conflict-free ds_read_b128 per 16x16x32 fragment (lane (r, g) reads position g ^ f(r) of row r).
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/p3_ring_gemm.hip -o tests/microbench/p3_ring_gemm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 r;
  r[0] = (__bf16)lo;
  r[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void split3_bf16(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = pack2_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = pack2_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = pack2_bf16(s0, s1);
}
__device__ __forceinline__ int swz(int r) { return (r >> 2) & 2; }

// fp32 [rows][K] -> planes [rows][3][K] bf16
__global__ void split_planes(const float* __restrict__ X, uint16_t* __restrict__ P, size_t rows, int K) {
  const size_t pairs = rows * (size_t)(K / 2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (K / 2);
    const int kp = (int)(i - row * (K / 2)) * 2;
    uint32_t h, m, l;
    split3_bf16(X[row * K + kp], X[row * K + kp + 1], h, m, l);
    uint32_t* base = reinterpret_cast<uint32_t*>(P + row * 3 * (size_t)K + kp);
    base[0] = h;
    base[K / 2] = m;
    base[K] = l;
  }
}

constexpr int PIECE = 1024;
#define ORDER6 constexpr int order6[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}}   /* (plane of A, plane of B), smallest first */

// one K-step of one MFMA wave on a complete plane slot
template <int MT, int NT>
__device__ __forceinline__ void p3_kstep(f32x4 (&acc)[MT][NT], const char* a_rb0, const char* b_rb0, int foff) {
  ORDER6;
  bf16x8 bp[3][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int p = 0; p < 3; ++p) bp[p][nt] = *reinterpret_cast<const bf16x8*>(b_rb0 + (nt * 3 + p) * PIECE + foff);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    bf16x8 ap[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) ap[p] = *reinterpret_cast<const bf16x8*>(a_rb0 + (mt * 3 + p) * PIECE + foff);
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[order6[t][0]], bp[order6[t][1]][nt], acc[mt][nt], 0, 0, 0);
  }
}

template <int MT, int NT>
__device__ __forceinline__ void p3_store(const f32x4 (&acc)[MT][NT], float* C, int N, int row0, int col0, int lane) {
  const int g = lane >> 4, r = lane & 15;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row0 + mt * 16 + g * 4 + e;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) C[(size_t)row * N + col0 + nt * 16 + r] = acc[mt][nt][e];
    }
}

// ---- design (ii): both operands as planes ----
// DIAG: 0 = the kernel; 2 = no DMA and no barrier (LDS reads + MFMA only); 3 = barriers kept, no DMA; 5 = MFMA only (fragments read once)
template <int BM, int BN, int WM, int WN, int S, int NL, int DIAG>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NL)) void p3_gemm(const uint16_t* __restrict__ A,
                                                                             const uint16_t* __restrict__ B, float* __restrict__ C,
                                                                             int M, int N, int K, unsigned a_bytes, unsigned b_bytes) {
  constexpr int RBA = BM / 16, RBB = BN / 16;
  static_assert(RBA % NL == 0 && RBB % NL == 0, "row blocks split evenly over the loader waves");
  constexpr int RAL = RBA / NL, RBL = RBB / NL, PPL = 3 * (RAL + RBL);
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int SLOT = (BM + BN) * 192;
  constexpr int MT = WM / 16, NT = WN / 16, WAVES_N = BN / WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / BN;
  const int L = blockIdx.x, m_lo = L & 7, r8 = L >> 3;
  const int tn = r8 % ntn, tm = (r8 / ntn) * 8 + m_lo;
  const int nk = K / 32;

  if (wave < NL) {
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
    const int pr = lane >> 2, c = (lane & 3) ^ swz(pr);
    unsigned voffA[RAL], voffB[RBL];
#pragma unroll
    for (int i = 0; i < RAL; ++i) voffA[i] = (unsigned)(((size_t)(tm * BM + (wave + NL * i) * 16 + pr) * 3 * K + c * 8) * 2);
#pragma unroll
    for (int j = 0; j < RBL; ++j) voffB[j] = (unsigned)(((size_t)(tn * BN + (wave + NL * j) * 16 + pr) * 3 * K + c * 8) * 2);
    auto issue = [&](int ks) {
      if (DIAG == 2 || DIAG == 3 || DIAG == 5) return;
      char* slot = smem + (ks % S) * SLOT;
      const unsigned so = (unsigned)ks * 64u;
#pragma unroll
      for (int i = 0; i < RAL; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(rsA, slot + ((wave + NL * i) * 3 + p) * PIECE, voffA[i], so + (unsigned)(p * K * 2));
#pragma unroll
      for (int j = 0; j < RBL; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          dma16(rsB, slot + BM * 192 + ((wave + NL * j) * 3 + p) * PIECE, voffB[j], so + (unsigned)(p * K * 2));
    };
    for (int ks = 0; ks < S - 1 && ks < nk; ++ks) issue(ks);
    for (int k = 0; k < nk; ++k) {
      if (k + S - 1 <= nk) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>();
      if (DIAG != 2 && DIAG != 5) ring_barrier();
      if (k + S - 1 < nk) issue(k + S - 1);
    }
    return;
  }
  const int mw = wave - NL;
  const int wm = mw / WAVES_N, wn = mw % WAVES_N;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r = lane & 15, g = lane >> 4;
  const int foff = r * 64 + ((g ^ swz(r)) * 16);
  const int abase = (wm * WM / 16) * 3 * PIECE, bbase = BM * 192 + (wn * WN / 16) * 3 * PIECE;
  if (DIAG == 5) {
    ORDER6;
    bf16x8 bp[3][NT], ap[MT][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[p][nt] = *reinterpret_cast<const bf16x8*>(smem + bbase + (nt * 3 + p) * PIECE + foff);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p) ap[mt][p] = *reinterpret_cast<const bf16x8*>(smem + abase + (mt * 3 + p) * PIECE + foff);
    for (int k = 0; k < nk; ++k) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[mt][order6[t][0]], bp[order6[t][1]][nt], acc[mt][nt], 0, 0, 0);
    }
  } else {
    for (int k = 0; k < nk; ++k) {
      if (DIAG != 2) ring_barrier();
      const char* sl = smem + (k % S) * SLOT;
      p3_kstep<MT, NT>(acc, sl + abase, sl + bbase, foff);
    }
  }
  p3_store<MT, NT>(acc, C, N, tm * BM + wm * WM, tn * BN + wn * WN, lane);
}

// ---- design (i'): B as planes, A as fp32 split by the producer waves ----
template <int BM, int BN, int WM, int WN, int NP>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NP)) void p3a_gemm(const float* __restrict__ A,
                                                                              const uint16_t* __restrict__ B, float* __restrict__ C,
                                                                              int M, int N, int K, unsigned a_bytes, unsigned b_bytes) {
  constexpr int RBA = BM / 16, RBB = BN / 16;
  static_assert(RBA % NP == 0 && RBB % NP == 0, "row blocks split evenly over the producer waves");
  constexpr int RAL = RBA / NP, RBL = RBB / NP;
  constexpr int NA = 2 * RAL, NB = 3 * RBL;          // DMA pieces per producer wave and K-step: fp32 staging / B planes
  constexpr int SLOT = (BM + BN) * 192, STAGE = BM * 128, SA = 3;
  constexpr int MT = WM / 16, NT = WN / 16, WAVES_N = BN / WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stage = smem + 2 * SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / BN;
  const int L = blockIdx.x, m_lo = L & 7, r8 = L >> 3;
  const int tn = r8 % ntn, tm = (r8 / ntn) * 8 + m_lo;
  const int nk = K / 32;

  if (wave < NP) {
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
    unsigned voffA[RAL][2], voffB[RBL];
    const int sr = lane >> 3, sc = lane & 7;          // staging piece: 8 rows x 128 B, lane -> (row, 16-byte granule of 4 channels)
#pragma unroll
    for (int i = 0; i < RAL; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        voffA[i][h] = (unsigned)(((size_t)(tm * BM + (wave + NP * i) * 16 + 8 * h + sr) * K + sc * 4) * 4);
    const int pr = lane >> 2, c = (lane & 3) ^ swz(pr);
#pragma unroll
    for (int j = 0; j < RBL; ++j) voffB[j] = (unsigned)(((size_t)(tn * BN + (wave + NP * j) * 16 + pr) * 3 * K + c * 8) * 2);
    auto issueB = [&](int ks) {
      char* slot = smem + (ks & 1) * SLOT + BM * 192;
#pragma unroll
      for (int j = 0; j < RBL; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(rsB, slot + ((wave + NP * j) * 3 + p) * PIECE, voffB[j], (unsigned)ks * 64u + (unsigned)(p * K * 2));
    };
    auto issueA = [&](int ks) {
      char* st = stage + (ks % SA) * STAGE;
#pragma unroll
      for (int i = 0; i < RAL; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) dma16(rsA, st + (((wave + NP * i) * 2 + h) * PIECE), voffA[i][h], (unsigned)ks * 128u);
    };
    // split this wave's staged rows of K-step ks into the plane slot
    auto splitA = [&](int ks) {
      const char* st = stage + (ks % SA) * STAGE;
      char* slot = smem + (ks & 1) * SLOT;
#pragma unroll
      for (int i = 0; i < RAL; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(st + ((wave + NP * i) * 2 + h) * PIECE + lane * 16);
          uint2 hi, mid, lo;
          split3_bf16(x[0], x[1], hi.x, mid.x, lo.x);
          split3_bf16(x[2], x[3], hi.y, mid.y, lo.y);
          const int row16 = 8 * h + sr;
          char* d = slot + ((wave + NP * i) * 3) * PIECE + row16 * 64 + (((sc >> 1) ^ swz(row16)) * 16) + (sc & 1) * 8;
          *reinterpret_cast<uint2*>(d) = hi;
          *reinterpret_cast<uint2*>(d + PIECE) = mid;
          *reinterpret_cast<uint2*>(d + 2 * PIECE) = lo;
        }
    };
    issueB(0);
    issueA(0);
    if (nk > 1) { issueA(1); wait_vmcnt<NA>(); } else wait_vmcnt<0>();
    splitA(0);
    for (int k = 0; k < nk; ++k) {
      ring_barrier();                       // plane slot k is published; slot k + 1 is free
      if (k + 1 < nk) {
        issueB(k + 1);
        if (k + 2 < nk) {
          issueA(k + 2);
          wait_vmcnt<NB + NA>();            // A(k + 1) has landed (younger: B(k + 1), A(k + 2))
        } else {
          wait_vmcnt<NB>();
        }
        splitA(k + 1);
        if (k + 2 < nk) wait_vmcnt<NA>(); else wait_vmcnt<0>();      // B(k + 1) has landed
      }
    }
    return;
  }
  const int mw = wave - NP;
  const int wm = mw / WAVES_N, wn = mw % WAVES_N;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r = lane & 15, g = lane >> 4;
  const int foff = r * 64 + ((g ^ swz(r)) * 16);
  const int abase = (wm * WM / 16) * 3 * PIECE, bbase = BM * 192 + (wn * WN / 16) * 3 * PIECE;
  for (int k = 0; k < nk; ++k) {
    ring_barrier();
    const char* sl = smem + (k & 1) * SLOT;
    p3_kstep<MT, NT>(acc, sl + abase, sl + bbase, foff);
  }
  p3_store<MT, NT>(acc, C, N, tm * BM + wm * WM, tn * BN + wn * WN, lane);
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
  CK(hipEventDestroy(s));
  CK(hipEventDestroy(e));
  return ms / reps;
}
// relative L2 error over 512 sampled outputs against fp64
static double check(const std::vector<float>& A, const std::vector<float>& B, const float* C, Shape s) {
  double num = 0, den = 0;
  for (int t = 0; t < 512; ++t) {
    int m = (int)((1103515245u * (unsigned)t + 12345u) % (unsigned)s.M), n = (int)((69069u * (unsigned)t + 1u) % (unsigned)s.N);
    double ref = 0;
    for (int k = 0; k < s.K; ++k) ref += (double)A[(size_t)m * s.K + k] * B[(size_t)n * s.K + k];
    const double d = ref - C[(size_t)m * s.N + n];
    num += d * d;
    den += ref * ref;
  }
  return sqrt(num / den);
}

int main(int argc, char** argv) {
  Shape shapes[] = {{65536, 128, 1024}, {262144, 64, 512}, {65536, 256, 2048}, {16384, 128, 1024}, {262144, 128, 512}};
  const int rounds = argc > 1 ? atoi(argv[1]) : 1;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  for (int round = 0; round < rounds; ++round)
    for (Shape s : shapes) {
      std::vector<float> hA((size_t)s.M * s.K), hB((size_t)s.N * s.K), hC((size_t)s.M * s.N);
      unsigned x = 12345u;
      for (auto& v : hA) { x = x * 1664525u + 1013904223u; v = ((float)(x >> 8) / 16777216.0f - 0.5f); }
      for (auto& v : hB) { x = x * 1664525u + 1013904223u; v = ((float)(x >> 8) / 16777216.0f - 0.5f) * 0.2f; }
      float *A, *B, *C;
      uint16_t *Ap, *Bpl;
      CK(hipMalloc(&A, hA.size() * 4));
      CK(hipMalloc(&B, hB.size() * 4));
      CK(hipMalloc(&C, hC.size() * 4));
      CK(hipMalloc(&Ap, hA.size() * 6));
      CK(hipMalloc(&Bpl, hB.size() * 6));
      CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(split_planes, dim3(4096), dim3(256), 0, 0, A, Ap, (size_t)s.M, s.K);
      hipLaunchKernelGGL(split_planes, dim3(1024), dim3(256), 0, 0, B, Bpl, (size_t)s.N, s.K);
      CK(hipDeviceSynchronize());
      {
        const float ms = time_ms([&] { hipLaunchKernelGGL(split_planes, dim3(4096), dim3(256), 0, 0, A, Ap, (size_t)s.M, s.K); }, 5);
        printf("M=%7d K=%4d  split_planes of A (4 B read + 6 B written per element)  %8.1f us  %6.2f TB/s\n", s.M, s.K, ms * 1e3,
               (double)s.M * s.K * 10 / ms / 1e9);
      }
      const unsigned a_pl = (unsigned)(hA.size() * 6), b_pl = (unsigned)(hB.size() * 6), a_f = (unsigned)(hA.size() * 4);
      const double fl = 2.0 * s.M * s.N * s.K;
      auto report = [&](const char* name, float ms, bool checked) {
        double err = -1;
        if (checked) {
          CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
          err = check(hA, hB, hC.data(), s);
        }
        printf("M=%7d N=%4d K=%4d  %-46s %8.1f us  %6.1f TF/s (fp32-equivalent)  rel-L2 %.2e\n", s.M, s.N, s.K, name, ms * 1e3,
               fl / ms / 1e9, err);
        CK(hipMemset(C, 0, hC.size() * 4));
        fflush(stdout);
      };
#define RUNP3(BM, BN, WM, WN, S_, NL_, D_)                                                                            \
  if (s.M % (BM * 8) == 0 && s.N % BN == 0) {                                                                         \
    const size_t smem = (size_t)(BM + BN) * 192 * S_;                                                                 \
    constexpr int thr = 64 * ((BM / WM) * (BN / WN) + NL_);                                                           \
    CK(hipFuncSetAttribute((const void*)p3_gemm<BM, BN, WM, WN, S_, NL_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    report("planes " #BM "x" #BN " w" #WM "x" #WN " s" #S_ " nl" #NL_ " diag" #D_, time_ms([&] {                       \
      hipLaunchKernelGGL((p3_gemm<BM, BN, WM, WN, S_, NL_, D_>), dim3((s.M / BM) * (s.N / BN)), dim3(thr), smem, 0, Ap, Bpl, C, s.M, s.N, s.K, a_pl, b_pl); }, reps), D_ == 0); \
  }
#define RUNP3A(BM, BN, WM, WN, NP_)                                                                                   \
  if (s.M % (BM * 8) == 0 && s.N % BN == 0) {                                                                         \
    const size_t smem = (size_t)(BM + BN) * 192 * 2 + (size_t)BM * 128 * 3;                                           \
    constexpr int thr = 64 * ((BM / WM) * (BN / WN) + NP_);                                                           \
    CK(hipFuncSetAttribute((const void*)p3a_gemm<BM, BN, WM, WN, NP_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    report("A fp32 + producers, B planes " #BM "x" #BN " w" #WM "x" #WN " np" #NP_, time_ms([&] {                      \
      hipLaunchKernelGGL((p3a_gemm<BM, BN, WM, WN, NP_>), dim3((s.M / BM) * (s.N / BN)), dim3(thr), smem, 0, A, Bpl, C, s.M, s.N, s.K, a_f, b_pl); }, reps), true); \
  }
      // warm the clocks
      for (int w = 0; w < 30; ++w) RUNP3(128, 64, 64, 32, 3, 2, 5)
      if (s.N % 128 == 0) {
        RUNP3(128, 128, 64, 64, 3, 4, 0)
        RUNP3(128, 128, 64, 64, 3, 2, 0)
        RUNP3(128, 128, 64, 64, 2, 4, 0)
        RUNP3(128, 128, 64, 32, 3, 4, 0)
        RUNP3(128, 128, 64, 32, 3, 2, 0)
        RUNP3(256, 128, 64, 64, 2, 4, 0)
        RUNP3(128, 128, 64, 64, 3, 4, 3)
        RUNP3(128, 128, 64, 64, 3, 4, 2)
        RUNP3(128, 128, 64, 64, 3, 4, 5)
        RUNP3(128, 128, 64, 32, 3, 4, 2)
        RUNP3(128, 128, 64, 32, 3, 4, 5)
        RUNP3A(128, 128, 64, 64, 4)
        RUNP3A(128, 128, 64, 32, 4)
      }
      RUNP3(128, 64, 64, 32, 3, 4, 0)
      RUNP3(128, 64, 64, 32, 3, 2, 0)
      RUNP3(128, 64, 32, 32, 3, 4, 0)
      RUNP3(128, 64, 64, 32, 2, 4, 0)
      RUNP3(256, 64, 64, 64, 2, 4, 0)
      RUNP3(256, 64, 64, 32, 2, 4, 0)
      RUNP3(64, 64, 32, 32, 3, 4, 0)
      RUNP3(64, 64, 32, 32, 3, 2, 0)
      RUNP3(64, 64, 32, 32, 2, 2, 0)
      RUNP3(128, 64, 64, 32, 3, 4, 2)
      RUNP3(128, 64, 64, 32, 3, 4, 5)
      RUNP3A(128, 64, 64, 32, 4)
      RUNP3A(128, 64, 32, 32, 4)
      RUNP3A(64, 64, 32, 32, 4)
      CK(hipFree(A));
      CK(hipFree(B));
      CK(hipFree(C));
      CK(hipFree(Ap));
      CK(hipFree(Bpl));
    }
  return 0;
}
