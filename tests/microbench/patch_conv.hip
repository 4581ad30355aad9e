// Diagnostic prototype (not part of the product library): a PATCH-RESIDENT k4 s2 p1 convolution in fp32, the form of the
// encoder layers and of the decoder's input-gradient launches.  32x32x32 -> 16x16x64 (K = 512): a 512-thread block owns 8
// output rows of one image; their 18 x 34 input pixels (halo included) are staged in LDS once and serve all 16 taps; the
// weights of one tap -- [64][32] -- stream through a two-slot LDS ring, loaded two taps ahead.  v_mfma_f32_16x16x4_f32 with
// the shared K permutation (tests/microbench/patch_tconv.hip, csrc/tconv_patch.hip).  The implicit GEMM of igemm_nt.hip
// runs this launch (4 x 256 samples) at 105 TFLOP/s and re-reads every input pixel 4 times through L2.
//   hipcc -O3 --offload-arch=gfx950 tests/microbench/patch_conv.hip -o tests/microbench/patch_conv.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// X [B][H][W][CIN], Wp [16][N][CIN] (tap kh*4+kw), Y [B][H/2][W/2][N]; TO output rows per block
template <int H, int W, int CIN, int N, int TO, int TS>
__global__ __launch_bounds__(512) void patch_conv(const float* __restrict__ X, const float* __restrict__ Wp, float* __restrict__ Y,
                                                  int B) {
  constexpr int HO = H / 2, WO = W / 2, CP = CIN + 4, PW = W + 2, PH = 2 * TO + 2, NT = N / 16;
  constexpr int MT = TO * WO / 16 / 8;        // m-tiles (16 output pixels) per wave
  constexpr int TILES = HO / TO, SV = TS * N * CIN / 4, BV = (SV + 511) / 512;   // 16-byte pieces per slice (TS taps), per thread
  static_assert(WO == 16 && MT >= 1 && SV % 512 == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                        // [PH][PW][CP]
  float* Bs = smem + PH * PW * CP;            // [2][TS*N][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  f32x4 rb[2][BV];
  auto bload = [&](int s, f32x4 (&dst)[BV]) {
#pragma unroll
    for (int i = 0; i < BV; ++i) dst[i] = *reinterpret_cast<const f32x4*>(Wp + (size_t)s * TS * N * CIN + (size_t)(tid + 512 * i) * 4);
  };
  auto bstore = [&](int slot, const f32x4 (&src)[BV]) {
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int e = (tid + 512 * i) * 4, n = e / CIN, c = e - n * CIN;       // n runs over TS*N rows (tap-major)
      *reinterpret_cast<f32x4*>(Bs + ((size_t)slot * TS * N + n) * CP + c) = src[i];
    }
  };
  for (int u = blockIdx.x; u < B * TILES; u += gridDim.x) {
    const int b = u / TILES, oy0 = (u - b * TILES) * TO;
    __syncthreads();
    for (int i = tid; i < PH * PW * (CIN / 4); i += 512) {
      const int c4 = i % (CIN / 4), p = i / (CIN / 4);
      const int py = p / PW, px = p - py * PW;
      const int iy = 2 * oy0 - 1 + py, ix = px - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(X + (((size_t)b * H + iy) * W + ix) * CIN + c4 * 4);
      *reinterpret_cast<f32x4*>(patch + (size_t)p * CP + c4 * 4) = v;
    }
    bload(0, rb[0]);
    bload(1, rb[1]);
    bstore(0, rb[0]);
    __syncthreads();
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16 / TS; ++s) {
      const int slot = s & 1;
      if (s + 2 < 16 / TS) bload(s + 2, rb[s & 1]);
#pragma unroll
      for (int ts = 0; ts < TS; ++ts) {
      const int tap = s * TS + ts, kh = tap >> 2, kw = tap & 3;
      const float* bbase = Bs + ((size_t)slot * TS + ts) * N * CP + r * CP + q * 4;
#pragma unroll
      for (int c0 = 0; c0 < CIN; c0 += 16) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int oy = wave * MT + m;           // output row of this m-tile inside the tile; lane r = output column
          af[m] = *reinterpret_cast<const f32x4*>(patch + ((size_t)(2 * oy + kh) * PW + (2 * r + kw)) * CP + c0 + q * 4);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const f32x4*>(bbase + (size_t)n * 16 * CP + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
      }
      }
      if (s + 1 < 16 / TS) {
        bstore(slot ^ 1, rb[(s + 1) & 1]);
        __syncthreads();
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int oy = oy0 + wave * MT + m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* dst = Y + (((size_t)b * HO + oy) * WO + 4 * q + e) * N + r;
#pragma unroll
        for (int n = 0; n < NT; ++n) dst[n * 16] = acc[m][n][e];
      }
    }
  }
}

int main() {
  constexpr int H = 32, W = 32, CIN = 32, N = 64, TO = 8, TS = 2;
  const int B = 1024;
  const size_t nx = (size_t)B * H * W * CIN, nw = (size_t)16 * N * CIN, ny = (size_t)B * (H / 2) * (W / 2) * N;
  std::vector<float> hx(nx), hw(nw);
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
  float *X, *Wp, *Y;
  CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&Wp, nw * 4)); CK(hipMalloc(&Y, ny * 4));
  CK(hipMemcpy(X, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wp, hw.data(), nw * 4, hipMemcpyHostToDevice));
  const size_t smem = ((size_t)(2 * TO + 2) * (W + 2) * (CIN + 4) + 2 * TS * N * (CIN + 4)) * 4;
  CK(hipFuncSetAttribute((const void*)patch_conv<H, W, CIN, N, TO, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256, 2048}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((patch_conv<H, W, CIN, N, TO, TS>), dim3(grid), dim3(512), smem, 0, X, Wp, Y, B);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((patch_conv<H, W, CIN, N, TO, TS>), dim3(grid), dim3(512), smem, 0, X, Wp, Y, B);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = 2.0 * B * (H / 2) * (W / 2) * N * 16 * CIN;
    printf("patch-resident conv %dx%dx%d -> %dx%dx%d, B=%d, grid %4d (LDS %zu KB): %7.1f us  %6.1f TFLOP/s\n", H, W, CIN, H / 2, W / 2, N,
           B, grid, smem / 1024, ms * 1e3, fl / ms / 1e9);
  }
  std::vector<float> hy(ny);
  CK(hipMemcpy(hy.data(), Y, ny * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int t = 0; t < 2000; ++t) {
    const int b = rand() % B, oy = rand() % (H / 2), ox = rand() % (W / 2), n = rand() % N;
    double ref = 0;
    for (int kh = 0; kh < 4; ++kh)
      for (int kw = 0; kw < 4; ++kw) {
        const int iy = 2 * oy - 1 + kh, ix = 2 * ox - 1 + kw;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        for (int c = 0; c < CIN; ++c)
          ref += (double)hx[(((size_t)b * H + iy) * W + ix) * CIN + c] * hw[((size_t)(kh * 4 + kw) * N + n) * CIN + c];
      }
    maxerr = fmax(maxerr, fabs(hy[(((size_t)b * (H / 2) + oy) * (W / 2) + ox) * N + n] - ref));
  }
  printf("max abs error vs host reference on 2000 outputs: %.3e\n", maxerr);
  return 0;
}
