// Patch-resident k4 s2 p1 convolutions with 32 input and 32 output channels, fp32: the stride-2 32 -> 32 channel stages of
// the 128 / 256 pixel stacks (models/shapes.py: the extra Conv+BN+Swish stages behind the first encoder convolution; and, as
// the input gradient of the extra ConvT stages of the decoders, the same GEMM with four groups and the BatchNorm+Swish
// backward in its epilogue).  Their implicit-GEMM launches fill one byte of LDS per 12.8 flop (128 x 32 tiles: the gathered
// operand is everything) and re-read every input pixel for four taps through L2: 59-68 TFLOP/s in igemm_ws.hip, against a
// fill roofline of ~96.  Here one 256-thread block (two per CU) owns TO output rows of one image: the (2 TO + 2) x (W + 2)
// input pixels under them are staged in LDS ONCE -- as two column-parity planes, so that the stride-2 pixel walk of a tap
// becomes a unit-stride walk of one plane: conflict-free 16-byte reads with the CIN + 4 channel stride of tconv_patch.hip --
// and serve all 16 taps; the weights of one tap -- [N][CIN] -- stream through a two-slot LDS ring, loaded two taps ahead.
// v_mfma_f32_16x16x4_f32, fragment map and K permutation as in tconv_patch.hip.  44 flop per filled byte.
#include "igemm_geom.h"

namespace {

struct ConvPatchEpi {        // the epilogue set of igemm_nt_kernel (all optional)
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

template <int H, int W, int CIN, int N, int TO, int NW>
struct ConvPatchCfg {
  static constexpr int HO = H / 2, WO = W / 2;
  static constexpr int CP = CIN + 4;                                   // padded channel stride: conflict-free 16-byte LDS reads
  static constexpr int PH = 2 * TO + 2, PWH = W / 2 + 1;               // patch rows; columns per parity plane
  static constexpr int NT = N / 16, MT = TO * WO / 16 / NW;            // n-tiles; m-tiles (16 output pixels) per wave, NW waves
  static constexpr int TILES = HO / TO;                                // row tiles per image
  static constexpr int SLICE_V = N * CIN / 4;                          // 16-byte pieces of one tap's weights
  static constexpr size_t SMEM = ((size_t)2 * PH * PWH * CP + 2 * N * CP) * sizeof(float);
  static_assert(HO % TO == 0 && (TO * WO) % (16 * NW) == 0 && (WO & (WO - 1)) == 0 && WO >= 16 && CIN % 16 == 0 && N % 16 == 0, "tile geometry");
  static_assert(SLICE_V <= 64 * NW, "one 16-byte piece of a weight slice per thread");
  static_assert(SMEM <= 160 * 1024, "LDS of one CU");
};

template <int H, int W, int CIN, int N, int TO, int NW>
__global__ __launch_bounds__(64 * NW) void conv_patch_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                         float* __restrict__ Y, float* __restrict__ stats, int Bt, int Bg,
                                                         int ldc, const ConvPatchEpi ep) {
  using K = ConvPatchCfg<H, W, CIN, N, TO, NW>;
  constexpr int NTHR = 64 * NW;
  constexpr int CP = K::CP, PH = K::PH, PWH = K::PWH, NT = K::NT, MT = K::MT, HO = K::HO, WO = K::WO;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                         // [2 parities][PH][PWH][CIN+4]: patch column pc = 2 * idx + parity <-> input column pc - 1
  float* Bs = smem + 2 * PH * PWH * CP;        // [2][N][CIN+4]; reused as the statistics scratch at the end of a tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  f32x4 rb[2];                                 // one 16-byte piece of a tap's weights per thread, two taps in flight
  auto bload = [&](int tap, f32x4& dst) {      // Wp: [16 taps = kh*4 + kw][N][CIN]
    if (tid < K::SLICE_V) dst = *reinterpret_cast<const f32x4*>(Wp + (size_t)tap * N * CIN + (size_t)tid * 4);
  };
  auto bstore = [&](int slot, const f32x4& src) {
    const int e = tid * 4, n = e / CIN, c = e - n * CIN;
    if (tid < K::SLICE_V) *reinterpret_cast<f32x4*>(Bs + ((size_t)slot * N + n) * CP + c) = src;
  };
  // output pixel of (m-tile m of this wave, lane r) inside the tile
  int py_[MT], px_[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = (wave * MT + m) * 16 + r;
    py_[m] = p / WO;
    px_[m] = p % WO;
  }
  for (int u = blockIdx.x; u < Bt * K::TILES; u += gridDim.x) {
    const int b = u / K::TILES, oy0 = (u - b * K::TILES) * TO;         // image, first output row of the tile
    __syncthreads();                           // the previous tile's LDS reads are done
    for (int i = tid; i < PH * (W + 2) * (CIN / 4); i += NTHR) {
      const int c4 = i % (CIN / 4), p = i / (CIN / 4);
      const int pr = p / (W + 2), pc = p - pr * (W + 2);
      const int iy = 2 * oy0 - 1 + pr, ix = pc - 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + (((size_t)b * H + iy) * W + ix) * CIN + c4 * 4));
      *reinterpret_cast<f32x4*>(patch + ((size_t)((pc & 1) * PH + pr) * PWH + (pc >> 1)) * CP + c4 * 4) = v;
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
    // two accumulator sets, even and odd taps: with one m-tile per wave the two n-tiles alone would leave the MFMA pipe waiting
    // on its own results
    f32x4 acc2[2][MT][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc2[h][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // backward epilogues: the saved pre-activation values of the tile's outputs are requested NOW and land under the MFMAs
    float yv[MT][4][NT];
    if (bnbwd) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int p = (wave * MT + m) * 16 + 4 * q + e;
          const size_t ooff = (((size_t)b * HO + (oy0 + p / WO)) * WO + p % WO) * ldc + r;
#pragma unroll
          for (int n = 0; n < NT; ++n) yv[m][e][n] = ep.bn_y[ooff + n * 16];
        }
    }
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const int slot = tap & 1, kh = tap >> 2, kw = tap & 3;
      if (tap + 2 < 16) bload(tap + 2, rb[tap & 1]);            // rb[tap & 1] held tap's slice, which is in LDS already
      const float* plane = patch + (size_t)(kw & 1) * PH * PWH * CP;
      const float* bbase = Bs + (size_t)slot * N * CP + r * CP + q * 4;
#pragma unroll
      for (int c0 = 0; c0 < CIN; c0 += 16) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          af[m] = *reinterpret_cast<const f32x4*>(plane + ((size_t)(2 * py_[m] + kh) * PWH + (px_[m] + (kw >> 1))) * CP + c0 + q * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const f32x4*>(bbase + (size_t)n * 16 * CP + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc2[tap & 1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc2[tap & 1][m][n], 0, 0, 0);
      }
      if (tap + 1 < 16) {
        bstore(slot ^ 1, rb[(tap + 1) & 1]);                     // tap+1's slice, loaded during tap-1; slot^1 was last read then
        __syncthreads();
      }
    }
    // accumulator element e of tile (m, n): output pixel 4q + e of the m-tile, channel n*16 + r
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = (wave * MT + m) * 16 + 4 * q + e;
        const size_t ooff = (((size_t)b * HO + (oy0 + p / WO)) * WO + p % WO) * ldc + r;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float v = acc2[0][m][n][e] + acc2[1][m][n][e];
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
    if (stats) {
      // per-tile column sums: over the four lane groups q (shuffles), then over the waves (LDS)
      __syncthreads();                         // every wave is past its last read of the weight ring
      float* red = Bs;                         // [NW][2][N]
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
        for (int w = 0; w < NW; ++w) t += red[(w * 2 + which) * N + col];
        const int img = b - grp_b * Bg, slot = img * K::TILES + (u - b * K::TILES);
        stats[(((size_t)grp_b * Bg * K::TILES + slot) * 2 + which) * N + col] = t;
      }
    }
  }
}

bool conv_patch_enabled() {
  static const bool off = [] {
    const char* e = lab_env("MMDYN_CONV_PATCH");
    return e && atoi(e) == 0;
  }();
  return !off;
}

// the served shapes: (input size, Cin, N, stride, offset) -> row tiles per image; 0 = not served
int conv_patch_tiles(int mode, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int stride, int offset) {
  if (!conv_patch_enabled() || mode != MMDYN_CONV || stride != 2 || offset != -1 || Hi != Wi || Hi != 2 * Ho || Wi != 2 * Wo ||
      N != 32 || Cin != 32)
    return 0;
  if (Hi == 64) return 32 / 2;                 // 32 output rows, 2 per tile
  if (Hi == 128) return 64 / 1;                // 64 output rows, 1 per tile
  return 0;
}

template <int H, int W, int CIN, int N, int TO, int NW>
int conv_patch_launch(const float* A, const float* Bp, float* C, float* stats, const IgemmGeom& g, const ConvPatchEpi& ep,
                      hipStream_t st) {
  using K = ConvPatchCfg<H, W, CIN, N, TO, NW>;
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)conv_patch_kernel<H, W, CIN, N, TO, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)K::SMEM) == hipSuccess;
  }();
  if (!attr_ok) return MMDYN_ERR_SHAPE;
  const int Bt = g.G * g.Bg;
  hipLaunchKernelGGL((conv_patch_kernel<H, W, CIN, N, TO, NW>), dim3(Bt * K::TILES), dim3(64 * NW), K::SMEM, st, A, Bp, C, stats, Bt, g.Bg,
                     g.ldc, ep);
  MMDYN_LAUNCH_CHECK();
}

}  // namespace

// Number of BatchNorm partial-sum tiles per group this kernel writes (one per row tile of an image), 0 when the shape is not
// served.
int mmdyn_conv_patch_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int stride, int offset) {
  return Bg * conv_patch_tiles(mode, Hi, Wi, Cin, Ho, Wo, N, stride, offset);
}

// Returns MMDYN_OK / an error code, or 1 when the launch is not served (fp32 only: the caller has checked that).
int mmdyn_conv_patch_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                         const IgemmGeom& g, int stride, int offset, hipStream_t st) {
  if (!conv_patch_tiles(g.mode, g.Hi, g.Wi, g.Cin, g.Ho, g.Wo, g.N, stride, offset)) return 1;
  if (ws || g.splitk != 1) return MMDYN_ERR_SHAPE;
  if (g.bn_y && g.ldc != g.N) return MMDYN_ERR_SHAPE;
  const ConvPatchEpi ep{bias, C_act, g.act, g.bn_y, g.bn_mean, g.bn_rstd, g.bn_gamma, g.bn_beta, g.bwd_act};
  if (g.Hi == 64) return conv_patch_launch<64, 64, 32, 32, 2, 4>(A, Bp, C, stats, g, ep, st);
  return conv_patch_launch<128, 128, 32, 32, 1, 4>(A, Bp, C, stats, g, ep, st);
}
