// Patch-resident k4 s2 p1 transposed convolution, fp32 (ConvTranspose2d(64, 32, 4, 2, 1) on 16x16 inputs: the decoder's
// third up-sampling layer, vae.py:274, whose implicit-GEMM launch has only 32 output channels per 64-channel input row
// and re-reads the input 16 times -- 4 output-parity classes x 4 taps -- through L2: 82 TFLOP/s in igemm_nt.hip).
// One 512-thread block per image: the whole 16 x 16 x 64 input plus a zero halo is staged in LDS ONCE (18 x 18 pixels of
// 68 floats, 88 KB) and serves all classes and taps; the weights of one (class, tap) -- [32][64] -- stream through a two-slot
// LDS ring, loaded two taps ahead of their use.  v_mfma_f32_16x16x4_f32: an m-tile is one image row (16 pixels), lane
// (r = l & 15, q = l >> 4) reads 16 bytes at [pixel r][c0 + 4q ..] and MFMA j of the four it feeds multiplies k = c0 + 4q + j
// (the K permutation shared by both operands, as in igemm_d16.hip).  The K loop has no gather arithmetic and no global
// operand load.  Stand-alone 161 us against 208 us (tests/microbench/patch_tconv.hip is the prototype); BatchNorm partial
// sums are written per image (T = Bg tiles per group).
#include "igemm_geom.h"

namespace {

constexpr int PH = 16, PWD = 16, PCIN = 64, PN = 32;
constexpr int CP = PCIN + 4, PW2 = PWD + 2, PH2 = PH + 2;   // padded channel stride: conflict-free 16-byte LDS reads
constexpr int NT = PN / 16, MROWS = 2;                      // n-tiles, image rows (m-tiles) per wave: 8 waves x 2 = 16 rows
constexpr size_t PATCH_SMEM = ((size_t)PH2 * PW2 * CP + 2 * PN * CP) * sizeof(float);

struct PatchEpi {            // the epilogue set of igemm_nt_kernel (all optional)
  const float* bias;         // [N]
  float* C_act;              // second output act(C + bias)
  int act;
  const float* bn_y;         // BatchNorm+Swish backward epilogue (see IgemmGeom): pre-BN output at the C positions, row stride N
  const float* bn_mean;
  const float* bn_rstd;
  const float* bn_gamma;
  const float* bn_beta;
};

__global__ __launch_bounds__(512) void tconv_patch_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                          float* __restrict__ Y, float* __restrict__ stats, int Bt, int Bg,
                                                          int ldc, const PatchEpi ep) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                         // [18][18][68]
  float* Bs = smem + PH2 * PW2 * CP;           // [2][32][68]; reused as the statistics scratch at the end of an image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  auto widx = [](int s) {                      // slice s = class*4 + tap -> kernel tap kh*4 + kw (igemm_nt.hip, TCONV_S2P1)
    const int ph = s >> 3, pw = (s >> 2) & 1, th = (s >> 1) & 1, tw = s & 1;
    return (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);
  };
  f32x4 rb[2];                                 // one 16-byte piece of a weight slice per thread, two slices in flight
  auto bload = [&](int s, f32x4& dst) { dst = *reinterpret_cast<const f32x4*>(Wp + (size_t)widx(s) * PN * PCIN + (size_t)tid * 4); };
  auto bstore = [&](int slot, const f32x4& src) {
    const int e = tid * 4, n = e / PCIN, c = e - n * PCIN;
    *reinterpret_cast<f32x4*>(Bs + ((size_t)slot * PN + n) * CP + c) = src;
  };
  for (int b = blockIdx.x; b < Bt; b += gridDim.x) {
    __syncthreads();                           // the previous image's LDS reads are done
    for (int i = tid; i < PH2 * PW2 * (PCIN / 4); i += 512) {
      const int c4 = i % (PCIN / 4), p = i / (PCIN / 4);
      const int py = p / PW2, px = p - py * PW2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (py >= 1 && py <= PH && px >= 1 && px <= PWD)
        v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(X + (((size_t)b * PH + (py - 1)) * PWD + (px - 1)) * PCIN + c4 * 4));
      *reinterpret_cast<f32x4*>(patch + (size_t)p * CP + c4 * 4) = v;
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
      bn_m[n] = bnbwd ? ep.bn_mean[(size_t)grp_b * PN + col] : 0.f;
      bn_r[n] = bnbwd ? ep.bn_rstd[(size_t)grp_b * PN + col] : 0.f;
      bn_g[n] = bnbwd ? ep.bn_gamma[col] : 0.f;
      bn_b[n] = bnbwd ? ep.bn_beta[col] : 0.f;
      bias_v[n] = ep.bias ? ep.bias[col] : 0.f;
    }
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      f32x4 acc[MROWS][NT];
#pragma unroll
      for (int m = 0; m < MROWS; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int s = cls * 4 + tap, slot = tap & 1;          // (s & 1 == tap & 1: register set and slot are compile-time)
        if (s + 2 < 16) bload(s + 2, rb[tap & 1]);            // rb[tap & 1] held slice s, which is in LDS already
        const int dh = ph - (tap >> 1), dw = pw - (tap & 1);
        const float* bbase = Bs + (size_t)slot * PN * CP + r * CP + q * 4;
#pragma unroll
        for (int c0 = 0; c0 < PCIN; c0 += 16) {
          f32x4 af[MROWS], bf[NT];
#pragma unroll
          for (int m = 0; m < MROWS; ++m) {
            const int y = wave * MROWS + m;                    // image row of this m-tile; lane r = pixel x
            af[m] = *reinterpret_cast<const f32x4*>(patch + ((size_t)(y + dh + 1) * PW2 + (r + dw + 1)) * CP + c0 + q * 4);
          }
#pragma unroll
          for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const f32x4*>(bbase + (size_t)n * 16 * CP + c0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MROWS; ++m)
#pragma unroll
              for (int n = 0; n < NT; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
        }
        if (s + 1 < 16) {
          bstore(slot ^ 1, rb[(tap + 1) & 1]);                 // slice s+1, loaded during tap s-1; slot^1 was last read then
          __syncthreads();
        }
      }
      // accumulator element e of tile (m, n): pixel x = 4q + e of image row y, channel n*16 + r
#pragma unroll
      for (int m = 0; m < MROWS; ++m) {
        const int y = wave * MROWS + m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int x = 4 * q + e;
          const size_t ooff = ((((size_t)b * 2 * PH + (2 * y + ph)) * 2 * PWD) + (2 * x + pw)) * ldc + r;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            float v = acc[m][n][e];
            if (bnbwd) {               // du = da * swish'(gamma * xhat + beta); the sums are those of the BatchNorm backward
              const float xh = (ep.bn_y[ooff + n * 16] - bn_m[n]) * bn_r[n];
              v *= swish_gradf_(bn_g[n] * xh + bn_b[n]);
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
      // per-image column sums: over the four lane groups q (shuffles), then over the eight waves (LDS)
      __syncthreads();                         // every wave is past its last read of the weight ring
      float* red = Bs;                         // [8][2][32]
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float s0 = colsum[n], s1 = colsq[n];
        s0 += __shfl_xor(s0, 16, 64);
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (q == 0) {
          red[(wave * 2 + 0) * PN + n * 16 + r] = s0;
          red[(wave * 2 + 1) * PN + n * 16 + r] = s1;
        }
      }
      __syncthreads();
      if (tid < 2 * PN) {
        const int which = tid / PN, col = tid - which * PN;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[(w * 2 + which) * PN + col];
        const int grp = b / Bg, img = b - grp * Bg;
        stats[(((size_t)grp * Bg + img) * 2 + which) * PN + col] = t;
      }
    }
  }
}

bool patch_serves(int mode, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  static const bool off = [] {
    const char* e = getenv("MMDYN_TCONV_PATCH");
    return e && atoi(e) == 0;
  }();
  return !off && mode == MMDYN_TCONV_S2P1 && Hi == PH && Wi == PWD && Cin == PCIN && Ho == 2 * PH && Wo == 2 * PWD && N == PN;
}

}  // namespace

// Number of BatchNorm partial-sum tiles per group this kernel writes (one per image), 0 when the shape is not served.
int mmdyn_tconv_patch_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  return patch_serves(mode, Hi, Wi, Cin, Ho, Wo, N) ? Bg : 0;
}

// Returns MMDYN_OK / an error code, or 1 when the launch is not served (fp32 only: the caller has checked that).
int mmdyn_tconv_patch_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                          const IgemmGeom& g, hipStream_t st) {
  if (!patch_serves(g.mode, g.Hi, g.Wi, g.Cin, g.Ho, g.Wo, g.N)) return 1;
  if (ws || g.splitk != 1) return MMDYN_ERR_SHAPE;          // (split-K is a DENSE-mode feature: the entry point has refused it)
  if (g.bn_y && g.ldc != g.N) return MMDYN_ERR_SHAPE;
  static const bool attr_ok = [] {
    return hipFuncSetAttribute((const void*)tconv_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)PATCH_SMEM) == hipSuccess;
  }();
  if (!attr_ok) return MMDYN_ERR_SHAPE;                     // (never observed: 105 KB of the CU's 160 KB)
  const int Bt = g.G * g.Bg;
  const PatchEpi ep{bias, C_act, g.act, g.bn_y, g.bn_mean, g.bn_rstd, g.bn_gamma, g.bn_beta};
  hipLaunchKernelGGL(tconv_patch_kernel, dim3(Bt), dim3(512), PATCH_SMEM, st, A, Bp, C, stats, Bt, g.Bg, g.ldc, ep);
  MMDYN_LAUNCH_CHECK();
}
