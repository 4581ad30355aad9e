// Patch-resident k4 s2 p1 transposed convolutions, fp32: the up-sampling layers with 32 output channels
// (ConvTranspose2d(64, 32, 4, 2, 1) on 16x16 inputs -- the decoder's third up-sampling layer, vae.py:274 -- and the
// 32 -> 32 channel stages of the 128 / 256 pixel stacks on 32x32 / 64x64 inputs).  Their implicit-GEMM launches have few
// output channels per input row and re-read the input 16 times -- 4 output-parity classes x 4 taps -- through L2
// (82 TFLOP/s in igemm_nt.hip).  Here one 512-thread block owns a tile of TH input rows of one image: the tile plus a halo
// is staged in LDS ONCE ((TH+2) x (W+2) pixels of CIN+4 floats, 88 KB) and serves all classes and taps; the weights of one
// (class, tap) -- [N][CIN] -- stream through a two-slot LDS ring, loaded two taps ahead of their use.
// v_mfma_f32_16x16x4_f32: an m-tile is 16 consecutive pixels of the tile, lane (r = l & 15, q = l >> 4) reads 16 bytes at
// [pixel r][c0 + 4q ..] and MFMA j of the four it feeds multiplies k = c0 + 4q + j (the K permutation shared by both
// operands, as in igemm_d16.hip).  The K loop has no gather arithmetic and no global operand load.  64 -> 32 channels on
// 4 x 256 samples: 169 us against 208 us (tests/microbench/patch_tconv.hip is the prototype), and the gain survives next to
// the other lane's kernels (step 6.93 -> 6.86 ms).  BatchNorm partial sums are written per tile (T = Bg * H / TH per group).
#include "igemm_geom.h"

namespace {

struct PatchEpi {            // the epilogue set of igemm_nt_kernel (all optional)
  const float* bias;         // [N]
  float* C_act;              // second output act(C + bias)
  int act;
  const float* bn_y;         // BatchNorm+Swish backward epilogue (see IgemmGeom): pre-BN output at the C positions, row stride N
  const float* bn_mean;
  const float* bn_rstd;
  const float* bn_gamma;
  const float* bn_beta;
  int bwd_act;               // bn_mean == nullptr: activation-only backward, C = acc * act'(bn_y)
};

template <int H, int W, int CIN, int N, int TH>
struct PatchCfg {
  static constexpr int CP = CIN + 4, PW2 = W + 2, PH2 = TH + 2;       // padded channel stride: conflict-free 16-byte LDS reads
  static constexpr int NT = N / 16, MT = TH * W / 16 / 8;              // n-tiles; m-tiles (16 pixels) per wave, 8 waves
  static constexpr int TILES = H / TH;                                  // row tiles per image
  static constexpr int SLICE_V = N * CIN / 4;                           // 16-byte pieces of one weight slice
  static constexpr size_t SMEM = ((size_t)PH2 * PW2 * CP + 2 * N * CP) * sizeof(float);
  static_assert(H % TH == 0 && (TH * W) % 128 == 0 && (W & (W - 1)) == 0 && CIN % 16 == 0 && N % 16 == 0, "tile geometry");
  static_assert(SLICE_V <= 512, "one 16-byte piece of a weight slice per thread");
  static_assert(SMEM <= 160 * 1024, "LDS of one CU");
};

template <int H, int W, int CIN, int N, int TH>
__global__ __launch_bounds__(512) void tconv_patch_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                          float* __restrict__ Y, float* __restrict__ stats, int Bt, int Bg,
                                                          int ldc, const PatchEpi ep) {
  using K = PatchCfg<H, W, CIN, N, TH>;
  constexpr int CP = K::CP, PW2 = K::PW2, PH2 = K::PH2, NT = K::NT, MT = K::MT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                         // [TH+2][W+2][CIN+4]
  float* Bs = smem + PH2 * PW2 * CP;           // [2][N][CIN+4]; reused as the statistics scratch at the end of a tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  auto widx = [](int s) {                      // slice s = class*4 + tap -> kernel tap kh*4 + kw (igemm_nt.hip, TCONV_S2P1)
    const int ph = s >> 3, pw = (s >> 2) & 1, th = (s >> 1) & 1, tw = s & 1;
    return (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);
  };
  f32x4 rb[2];                                 // one 16-byte piece of a weight slice per thread, two slices in flight
  auto bload = [&](int s, f32x4& dst) {
    if (tid < K::SLICE_V) dst = *reinterpret_cast<const f32x4*>(Wp + (size_t)widx(s) * N * CIN + (size_t)tid * 4);
  };
  auto bstore = [&](int slot, const f32x4& src) {
    const int e = tid * 4, n = e / CIN, c = e - n * CIN;
    if (tid < K::SLICE_V) *reinterpret_cast<f32x4*>(Bs + ((size_t)slot * N + n) * CP + c) = src;
  };
  // pixel of (m-tile m of this wave, lane r) inside the tile
  int py_[MT], px_[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = (wave * MT + m) * 16 + r;
    py_[m] = p / W;
    px_[m] = p % W;
  }
  for (int u = blockIdx.x; u < Bt * K::TILES; u += gridDim.x) {
    const int b = u / K::TILES, y0 = (u - b * K::TILES) * TH;          // image, first input row of the tile
    __syncthreads();                           // the previous tile's LDS reads are done
    // (all of a thread's patch loads are issued before its first LDS store: as a load -> store loop the ten round trips
    //  were serial -- ~10 us of a block's ~70 at one block per CU)
    constexpr int NPIECE = PH2 * PW2 * (CIN / 4), NLD = (NPIECE + 511) / 512;
    f32x4 pv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 512 * k;
      const int c4 = i % (CIN / 4), p = i / (CIN / 4);
      const int py = p / PW2, px = p - py * PW2;
      const int iy = y0 + py - 1, ix = px - 1;
      const bool ok = i < NPIECE && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      // (masked lanes read a valid dummy address: a predicated load would be sunk into its own branch)
      const f32x4 r = __builtin_nontemporal_load(
          reinterpret_cast<const f32x4*>(X + (ok ? (((size_t)b * H + iy) * W + ix) * CIN + c4 * 4 : (size_t)0)));
      pv[k] = ok ? r : z;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + 512 * k;
      if (i < NPIECE) *reinterpret_cast<f32x4*>(patch + (size_t)(i / (CIN / 4)) * CP + (i % (CIN / 4)) * 4) = pv[k];
    }
    bload(0, rb[0]);
    bload(1, rb[1]);
    bstore(0, rb[0]);
    __syncthreads();
    float colsum[NT], colsq[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) colsum[n] = colsq[n] = 0.f;
    const int grp_b = b / Bg;
    const bool bnbwd = ep.bn_y != nullptr;
    float bn_m[NT], bn_r[NT], bn_g[NT], bn_b[NT], bias_v[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n * 16 + r;
      const bool bn = bnbwd && ep.bn_mean != nullptr;
      bn_m[n] = bn ? ep.bn_mean[(size_t)grp_b * N + col] : 0.f;
      bn_r[n] = bn ? ep.bn_rstd[(size_t)grp_b * N + col] : 1.f;
      bn_g[n] = bn ? ep.bn_gamma[col] : 1.f;
      bn_b[n] = bn ? ep.bn_beta[col] : 0.f;
      bias_v[n] = ep.bias ? ep.bias[col] : 0.f;
    }
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      f32x4 acc[MT][NT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // backward epilogues: the saved pre-activation values of this class's outputs are requested NOW and land under the
      // MFMAs of the four taps (issued inside the epilogue they were 16 dependent round trips: 77 vs 50 us on 256 samples)
      float yv[MT][4][NT];
      if (bnbwd) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = (wave * MT + m) * 16 + 4 * q + e;
            const int y = y0 + p / W, x = p % W;
            const size_t ooff = ((((size_t)b * 2 * H + (2 * y + ph)) * 2 * W) + (2 * x + pw)) * ldc + r;
#pragma unroll
            for (int n = 0; n < NT; ++n) yv[m][e][n] = ep.bn_y[ooff + n * 16];
          }
      }
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int s = cls * 4 + tap, slot = tap & 1;          // (s & 1 == tap & 1: register set and slot are compile-time)
        if (s + 2 < 16) bload(s + 2, rb[tap & 1]);            // rb[tap & 1] held slice s, which is in LDS already
        const int dh = ph - (tap >> 1), dw = pw - (tap & 1);
        const float* bbase = Bs + (size_t)slot * N * CP + r * CP + q * 4;
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += 16) {
          f32x4 af[MT], bf[NT];
#pragma unroll
          for (int m = 0; m < MT; ++m)
            af[m] = *reinterpret_cast<const f32x4*>(patch + ((size_t)(py_[m] + dh + 1) * PW2 + (px_[m] + dw + 1)) * CP + c0 + q * 4);
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
        if (s + 1 < 16) {
          bstore(slot ^ 1, rb[(tap + 1) & 1]);                 // slice s+1, loaded during tap s-1; slot^1 was last read then
          __syncthreads();
        }
      }
      // accumulator element e of tile (m, n): pixel 4q + e of the m-tile, channel n*16 + r
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = (wave * MT + m) * 16 + 4 * q + e;
          const int y = y0 + p / W, x = p % W;
          const size_t ooff = ((((size_t)b * 2 * H + (2 * y + ph)) * 2 * W) + (2 * x + pw)) * ldc + r;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            float v = acc[m][n][e];
            if (bnbwd) {               // du = da * swish'(gamma * xhat + beta); the sums are those of the BatchNorm backward
              const float xh = (yv[m][e][n] - bn_m[n]) * bn_r[n];
              v *= act_grad(bn_g[n] * xh + bn_b[n], ep.bwd_act);
              colsum[n] += v;
              colsq[n] += v * xh;
            } else {
              colsum[n] += v;
              colsq[n] += v * v;
            }
            v += bias_v[n];
            Y[ooff + n * 16] = v;
            if (ep.C_act) ep.C_act[ooff + n * 16] = apply_act(v, ep.act);
          }
        }
      }
    }
    if (stats) {
      // per-tile column sums: over the four lane groups q (shuffles), then over the eight waves (LDS)
      __syncthreads();                         // every wave is past its last read of the weight ring
      float* red = Bs;                         // [8][2][N]
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float s0 = colsum[n], s1 = colsq[n];
        s0 += __shfl_xor(s0, 16, 64);
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (q == 0) {
          red[(wave * 2 + 0) * N + n * 16 + r] = s0;
          red[(wave * 2 + 1) * N + n * 16 + r] = s1;
        }
      }
      __syncthreads();
      if (tid < 2 * N) {
        const int which = tid / N, col = tid - which * N;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[(w * 2 + which) * N + col];
        const int img = b - grp_b * Bg, slot = img * K::TILES + (u - b * K::TILES);
        stats[(((size_t)grp_b * Bg * K::TILES + slot) * 2 + which) * N + col] = t;
      }
    }
  }
}

bool patch_enabled() {
  static const bool off = [] {
    const char* e = lab_env("MMDYN_TCONV_PATCH");
    return e && atoi(e) == 0;
  }();
  return !off;
}

// the served shapes: (input size, Cin, N) -> row tiles per image; 0 = not served
int patch_tiles(int mode, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  if (!patch_enabled() || mode != MMDYN_TCONV_S2P1 || Hi != Wi || Ho != 2 * Hi || Wo != 2 * Wi || N != 32) return 0;
  if (Hi == 16 && Cin == 64) return 1;
  if (Hi == 32 && Cin == 32) return 2;
  if (Hi == 64 && Cin == 32) return 8;
  return 0;
}

template <int H, int W, int CIN, int N, int TH>
int patch_launch(const float* A, const float* Bp, float* C, float* stats, const IgemmGeom& g, const PatchEpi& ep, hipStream_t st) {
  using K = PatchCfg<H, W, CIN, N, TH>;
  static LdsOptIn lds_opt_in;                               // (~105 KB of the CU's 160 KB)
  if (int e = lds_opt_in.ensure((const void*)tconv_patch_kernel<H, W, CIN, N, TH>, (int)K::SMEM)) return e;
  const int Bt = g.G * g.Bg;
  hipLaunchKernelGGL((tconv_patch_kernel<H, W, CIN, N, TH>), dim3(Bt * K::TILES), dim3(512), K::SMEM, st, A, Bp, C, stats, Bt, g.Bg,
                     g.ldc, ep);
  MMDYN_LAUNCH_CHECK();
}

}  // namespace

// Number of BatchNorm partial-sum tiles per group this kernel writes (one per row tile of an image), 0 when the shape is not
// served.
int mmdyn_tconv_patch_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  return Bg * patch_tiles(mode, Hi, Wi, Cin, Ho, Wo, N);
}

// Returns MMDYN_OK / an error code, or 1 when the launch is not served (fp32 only: the caller has checked that).
int mmdyn_tconv_patch_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                          const IgemmGeom& g, hipStream_t st) {
  if (!patch_tiles(g.mode, g.Hi, g.Wi, g.Cin, g.Ho, g.Wo, g.N)) return 1;
  if (ws || g.splitk != 1) return MMDYN_ERR_SHAPE;          // (split-K is a DENSE-mode feature: the entry point has refused it)
  if (g.bn_y && g.ldc != g.N) return MMDYN_ERR_SHAPE;
  const PatchEpi ep{bias, C_act, g.act, g.bn_y, g.bn_mean, g.bn_rstd, g.bn_gamma, g.bn_beta, g.bwd_act};
  if (g.Hi == 16) return patch_launch<16, 16, 64, 32, 16>(A, Bp, C, stats, g, ep, st);
  if (g.Hi == 32) return patch_launch<32, 32, 32, 32, 16>(A, Bp, C, stats, g, ep, st);
  return patch_launch<64, 64, 32, 32, 8>(A, Bp, C, stats, g, ep, st);
}
