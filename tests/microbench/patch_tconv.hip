// Diagnostic prototype (not part of the product library): a PATCH-RESIDENT k4 s2 p1 transposed convolution in fp32.
// One 512-thread block per image: the whole H x W x CIN input (+ a zero halo) is staged in LDS once and serves all four
// output-parity classes and their four taps each; the weights of one (class, tap) -- [N][CIN] -- stream through a two-slot
// LDS ring.  v_mfma_f32_16x16x4_f32 with the shared K permutation of igemm_d16.hip (one 16-byte LDS read feeds four MFMAs).
// Question: does removing the 16-fold re-read of the input through L2 (4 classes x 4 taps) beat the implicit GEMM of
// igemm_nt.hip on the 64 -> 32 channel layer (82 TFLOP/s there)?
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/patch_tconv.hip -o tests/microbench/patch_tconv.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// X [B][H][W][CIN], Wp [16][N][CIN] (tap kh*4+kw), Y [B][2H][2W][N]
template <int H, int W, int CIN, int N>
__global__ __launch_bounds__(512) void patch_tconv(const float* __restrict__ X, const float* __restrict__ Wp,
                                                   float* __restrict__ Y, int B) {
  constexpr int CP = CIN + 4, PW = W + 2, PH = H + 2;          // padded channel stride: conflict-free ds_read_b128 rows
  constexpr int WAVES = 8, ROWS_PER_WAVE = H * W / 16 / WAVES;  // m-tiles (16 pixels = one image row for W = 16) per wave
  static_assert(W == 16 && ROWS_PER_WAVE >= 1, "one m-tile per image row");
  constexpr int NT = N / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                       // [PH][PW][CP]
  float* Bs = smem + PH * PW * CP;           // [2][N][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    // zero halo + interior load (16-byte accesses)
    for (int i = tid; i < PH * PW * (CIN / 4); i += 512) {
      const int c4 = i % (CIN / 4), p = i / (CIN / 4);
      const int py = p / PW, px = p - py * PW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (py >= 1 && py <= H && px >= 1 && px <= W)
        v = *reinterpret_cast<const f32x4*>(X + (((size_t)b * H + (py - 1)) * W + (px - 1)) * CIN + c4 * 4);
      *reinterpret_cast<f32x4*>(patch + (size_t)p * CP + c4 * 4) = v;
    }
    // first weight slice (class 0, tap 0)
    auto widx = [](int cls, int tap) {
      const int ph = cls >> 1, pw = cls & 1, th = tap >> 1, tw = tap & 1;
      return (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);
    };
    constexpr int BV = N * CIN / 4 / 512;      // 16-byte loads per thread per slice
    static_assert(BV >= 1, "slice smaller than one pass");
    f32x4 rb[2][BV];                            // two register sets: a slice is loaded TWO taps before it is needed
    auto bload = [&](int s, f32x4 (&dst)[BV]) {
      const float* src = Wp + (size_t)widx(s >> 2, s & 3) * N * CIN;
#pragma unroll
      for (int i = 0; i < BV; ++i) dst[i] = *reinterpret_cast<const f32x4*>(src + (size_t)(tid + 512 * i) * 4);
    };
    auto bstore = [&](int slot, const f32x4 (&src)[BV]) {
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        const int e = (tid + 512 * i) * 4, n = e / CIN, c = e - n * CIN;
        *reinterpret_cast<f32x4*>(Bs + ((size_t)slot * N + n) * CP + c) = src[i];
      }
    };
    bload(0, rb[0]);
    bload(1, rb[1]);
    bstore(0, rb[0]);
    __syncthreads();
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      f32x4 acc[ROWS_PER_WAVE][NT];
#pragma unroll
      for (int m = 0; m < ROWS_PER_WAVE; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int s = cls * 4 + tap, slot = s & 1;        // (s & 1 == tap & 1: the register-set index is compile-time)
        if (s + 2 < 16) bload(s + 2, rb[tap & 1]);      // rb[tap&1] held slice s, already in LDS
        const int dh = ph - (tap >> 1), dw = pw - (tap & 1);
        const float* bbase = Bs + (size_t)slot * N * CP + r * CP + q * 4;
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += 16) {
          f32x4 af[ROWS_PER_WAVE], bf[NT];
#pragma unroll
          for (int m = 0; m < ROWS_PER_WAVE; ++m) {
            const int y = wave * ROWS_PER_WAVE + m;    // image row of this m-tile; lane r = pixel x
            af[m] = *reinterpret_cast<const f32x4*>(patch + ((size_t)(y + dh + 1) * PW + (r + dw + 1)) * CP + c0 + q * 4);
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const f32x4*>(bbase + (size_t)n * 16 * CP + c0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < ROWS_PER_WAVE; ++m)
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
        }
        if (s + 1 < 16) {
          bstore(slot ^ 1, rb[(tap + 1) & 1]);         // slice s+1, loaded during tap s-1 (slot^1 was last read at tap s-1)
          __syncthreads();
        }
      }
      // epilogue: acc element e of tile (m, n): pixel x = 4*(lane>>4)+e of image row y, channel n*16 + (lane&15)
#pragma unroll
      for (int m = 0; m < ROWS_PER_WAVE; ++m) {
        const int y = wave * ROWS_PER_WAVE + m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int x = 4 * q + e;
          float* dst = Y + ((((size_t)b * 2 * H + (2 * y + ph)) * 2 * W) + (2 * x + pw)) * N + r;
#pragma unroll
          for (int n = 0; n < NT; ++n) dst[n * 16] = acc[m][n][e];
        }
      }
    }
  }
}

int main() {
  constexpr int H = 16, W = 16, CIN = 64, N = 32;
  const int B = 1024;
  const size_t nx = (size_t)B * H * W * CIN, nw = (size_t)16 * N * CIN, ny = (size_t)B * 4 * H * W * N;
  std::vector<float> hx(nx), hw(nw);
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
  float *X, *Wp, *Y;
  CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&Wp, nw * 4)); CK(hipMalloc(&Y, ny * 4));
  CK(hipMemcpy(X, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wp, hw.data(), nw * 4, hipMemcpyHostToDevice));
  const size_t smem = ((size_t)(H + 2) * (W + 2) * (CIN + 4) + 2 * N * (CIN + 4)) * 4;
  CK(hipFuncSetAttribute((const void*)patch_tconv<H, W, CIN, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256, 512, 1024}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((patch_tconv<H, W, CIN, N>), dim3(grid), dim3(512), smem, 0, X, Wp, Y, B);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((patch_tconv<H, W, CIN, N>), dim3(grid), dim3(512), smem, 0, X, Wp, Y, B);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = 2.0 * B * 4 * H * W * N * 4 * CIN;
    printf("patch-resident tconv %dx%dx%d -> %dx%dx%d, B=%d, grid %4d (LDS %zu KB): %7.1f us  %6.1f TFLOP/s\n", H, W, CIN, 2 * H, 2 * W,
           N, B, grid, smem / 1024, ms * 1e3, fl / ms / 1e9);
  }
  // spot check against a host reference on a few outputs
  std::vector<float> hy(ny);
  CK(hipMemcpy(hy.data(), Y, ny * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int t = 0; t < 2000; ++t) {
    const int b = rand() % B, oy = rand() % (2 * H), ox = rand() % (2 * W), n = rand() % N;
    const int ph = oy & 1, pw = ox & 1, y = oy >> 1, x = ox >> 1;
    double ref = 0;
    for (int th = 0; th < 2; ++th)
      for (int tw = 0; tw < 2; ++tw) {
        const int iy = y + ph - th, ix = x + pw - tw, kh = 1 - ph + 2 * th, kw = 1 - pw + 2 * tw;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int c = 0; c < CIN; ++c)
          ref += (double)hx[(((size_t)b * H + iy) * W + ix) * CIN + c] * hw[((size_t)(kh * 4 + kw) * N + n) * CIN + c];
      }
    const double got = hy[(((size_t)b * 2 * H + oy) * 2 * W + ox) * N + n];
    maxerr = fmax(maxerr, fabs(got - ref));
  }
  printf("max abs error vs host reference on 2000 outputs: %.3e\n", maxerr);
  return 0;
}
