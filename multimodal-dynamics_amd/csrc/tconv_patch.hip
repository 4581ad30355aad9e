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
//
// P3 = the same kernel in the fp32x3 arithmetic on operands that ARRIVE SPLIT (rows of [plane][CIN] bf16, include/mmdyn_hip.h flag
// bits 7 + 8): the patch and the weight ring hold plane rows (6 * CIN bytes + 32 of padding: a row stride of 32 * odd bytes makes
// the 16-byte fragment reads of 16 consecutive pixels conflict-free in ds_read_b128's lane groups), the K loop is ds_read_b128 +
// v_mfma_f32_16x16x32_bf16 only -- six of the nine plane products per 32 channels, smallest first, as in igemm_wsp3_kernel.
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

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

template <int H, int W, int CIN, int N, int TH, bool P3>
struct PatchCfg {
  static constexpr int CP = CIN + 4, PW2 = W + 2, PH2 = TH + 2;       // fp32: padded channel stride, conflict-free 16-byte LDS reads
  static constexpr int ROWB = P3 ? 6 * CIN : 4 * CIN;                  // bytes of one pixel / one weight row in HBM
  static constexpr int PSB = P3 ? 6 * CIN + 32 : 4 * CP;               // ... and in LDS
  static constexpr int PPP = ROWB / 16;                                 // 16-byte pieces of a row
  static constexpr int NT = N / 16, MT = TH * W / 16 / 8;              // n-tiles; m-tiles (16 pixels) per wave, 8 waves
  static constexpr int TILES = H / TH;                                  // row tiles per image
  static constexpr int SLICE_V = N * PPP;                               // 16-byte pieces of one weight slice
  static constexpr int NWL = (SLICE_V + 511) / 512;                     // ... per thread
  static constexpr size_t SMEM = (size_t)(PH2 * PW2 + 2 * N) * PSB;
  static_assert(H % TH == 0 && (TH * W) % 128 == 0 && (W & (W - 1)) == 0 && W >= 16 && CIN % 32 == 0 && N % 16 == 0, "tile geometry");
  static_assert(!P3 || (PSB / 32) % 2 == 1, "plane rows: a stride of 32 * odd bytes");
  static_assert(SMEM <= 160 * 1024, "LDS of one CU");
};

template <int H, int W, int CIN, int N, int TH, bool P3, bool BWD>
__global__ __launch_bounds__(512) void tconv_patch_kernel(const void* __restrict__ X, const void* __restrict__ Wp,
                                                          float* __restrict__ Y, float* __restrict__ stats, int Bt, int Bg,
                                                          int ldc, const PatchEpi ep) {
  using K = PatchCfg<H, W, CIN, N, TH, P3>;
  constexpr int PW2 = K::PW2, PH2 = K::PH2, NT = K::NT, MT = K::MT, PSB = K::PSB, PPP = K::PPP, ROWB = K::ROWB, NWL = K::NWL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;                          // [TH+2][W+2] pixel rows of PSB bytes
  char* Bs = smem + PH2 * PW2 * PSB;           // [2][N] weight rows of PSB bytes; reused as the statistics scratch at the end of a tile
  const char* Xb = static_cast<const char*>(X);
  const char* Wb = static_cast<const char*>(Wp);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  auto widx = [](int s) {                      // slice s = class*4 + tap -> kernel tap kh*4 + kw (igemm_nt.hip, TCONV_S2P1)
    const int ph = s >> 3, pw = (s >> 2) & 1, th = (s >> 1) & 1, tw = s & 1;
    return (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);
  };
  constexpr int RD = 4;                        // weight slices in flight
  u32x4_t rb[RD][NWL];                         // the 16-byte pieces of a weight slice this thread moves
  auto bload = [&](int s, u32x4_t (&dst)[NWL]) {
#pragma unroll
    for (int k = 0; k < NWL; ++k) {
      const int i = tid + 512 * k;
      if (i < K::SLICE_V) dst[k] = *reinterpret_cast<const u32x4_t*>(Wb + (size_t)widx(s) * N * ROWB + (size_t)i * 16);
    }
  };
  auto bstore = [&](int slot, const u32x4_t (&src)[NWL]) {
#pragma unroll
    for (int k = 0; k < NWL; ++k) {
      const int i = tid + 512 * k, n = i / PPP, c = i - n * PPP;
      if (i < K::SLICE_V) *reinterpret_cast<u32x4_t*>(Bs + ((size_t)slot * N + n) * PSB + c * 16) = src[k];
    }
  };
  // pixel of (m-tile m of this wave, lane r) inside the tile
  int py_[MT], px_[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int p = (wave * MT + m) * 16 + r;
    py_[m] = p / W;
    px_[m] = p % W;
  }
  // The block is persistent: the patch of its NEXT tile is requested while the current one is multiplied (16-byte pieces parked in
  // registers, zero-filled and stored into LDS once the current tile's last fragment read is behind a barrier), and the weight
  // slices -- the same sixteen for every tile -- run four ahead of their use through a register ring, across tile boundaries.
  constexpr int NPIECE = PH2 * PW2 * PPP, NLD = (NPIECE + 511) / 512;
  const int total = Bt * K::TILES;
  u32x4_t pv[NLD];
  unsigned pok = 0;                            // bit k: piece k of this thread lies inside the image
  auto pload = [&](int u, int k0, int k1) {    // pieces k0 .. k1-1 of this thread (compile-time bounds at every call)
    const int b = u / K::TILES, y0 = (u - b * K::TILES) * TH;
    int t = tid;                               // (laundered: the per-piece index arithmetic is redone for every tile instead of being
    asm volatile("" : "+v"(t));                //  hoisted out of the tile loop, where ~60 loop-invariant registers made the allocator spill)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (k < k0 || k >= k1) continue;
      const int i = t + 512 * k;
      const int c = i % PPP, p = i / PPP;
      const int py = p / PW2, px = p - py * PW2;
      const int iy = y0 + py - 1, ix = px - 1;
      const bool ok = i < NPIECE && iy >= 0 && iy < H && ix >= 0 && ix < W;
      pok = (pok & ~(1u << k)) | (ok ? 1u << k : 0u);
      // (masked lanes read a valid dummy address: a predicated load would be sunk into its own branch)
      pv[k] = __builtin_nontemporal_load(
          reinterpret_cast<const u32x4_t*>(Xb + (ok ? (((size_t)b * H + iy) * W + ix) * ROWB + c * 16 : (size_t)0)));
    }
  };
  auto pstore = [&](int k0, int k1) {
    const u32x4_t z = {0u, 0u, 0u, 0u};
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      if (k < k0 || k >= k1) continue;
      const int i = t + 512 * k;
      if (i < NPIECE) *reinterpret_cast<u32x4_t*>(patch + (size_t)(i / PPP) * PSB + (i % PPP) * 16) = ((pok >> k) & 1u) ? pv[k] : z;
    }
  };
  // (the backward-epilogue instances hold the saved pre-activations of a class in registers as well: they park half of a
  //  thread's pieces and request the rest at the tile's start; no instance of this kernel uses scratch memory)
  constexpr int NPRE = BWD ? NLD / 2 : NLD, NH = NLD;
  if ((int)blockIdx.x < total) {
    pload(blockIdx.x, 0, NPRE);
#pragma unroll
    for (int k = 0; k < RD; ++k) bload(k, rb[k]);
  }
  for (int u = blockIdx.x; u < total; u += gridDim.x) {
    const int b = u / K::TILES, y0 = (u - b * K::TILES) * TH;          // image, first input row of the tile
    pload(u, NPRE, NH);
    __syncthreads();                           // the previous tile's LDS reads are done
    pstore(0, NH);
    bstore(0, rb[0]);
    __syncthreads();
    if (u + (int)gridDim.x < total) pload(u + gridDim.x, 0, NPRE);
    float colsum[NT], colsq[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) colsum[n] = colsq[n] = 0.f;
    const int grp_b = b / Bg;
    constexpr bool bnbwd = BWD;                // (the launch picks the instance: ep.bn_y != nullptr)
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
      float yv[BWD ? MT : 1][4][NT];
      if constexpr (bnbwd) {
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
        const int s = cls * 4 + tap, slot = tap & 1;          // (s & 3 == tap: register set and slot are compile-time)
        // slice s+1 (loaded RD-1 taps ago) goes into the other slot now -- last read in tap s-1, behind that tap's barrier -- so
        // that its stores run under this tap's MFMAs; the barrier at the end of the tap publishes it.  Slice s is in LDS: its
        // register set takes slice s + RD (of the next tile past the sixteenth)
        if (s + 1 < 16) bstore(slot ^ 1, rb[(tap + 1) % RD]);
        bload((s + RD) & 15, rb[tap % RD]);
        const int dh = ph - (tap >> 1), dw = pw - (tap & 1);
        if constexpr (P3) {
          const char* bbase = Bs + ((size_t)slot * N + r) * PSB + q * 16;
          const char* abase[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) abase[m] = patch + ((size_t)(py_[m] + dh + 1) * PW2 + (px_[m] + dw + 1)) * PSB + q * 16;
          constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (plane of A, plane of B), smallest first
#pragma unroll
          for (int kk = 0; kk < CIN / 32; ++kk) {
            // (one m-tile's fragments at a time: 9 fragments live instead of 18; every accumulator still takes its six products
            //  in the same order)
            bf16x8v bf[NT][3];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
              for (int p = 0; p < 3; ++p) bf[n][p] = *reinterpret_cast<const bf16x8v*>(bbase + (size_t)n * 16 * PSB + p * CIN * 2 + kk * 64);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              bf16x8v af[3];
#pragma unroll
              for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8v*>(abase[m] + p * CIN * 2 + kk * 64);
#pragma unroll
              for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[order[t][0]], bf[n][order[t][1]], acc[m][n], 0, 0, 0);
            }
          }
        } else {
          const float* bbase = reinterpret_cast<const float*>(Bs + ((size_t)slot * N + r) * PSB) + q * 4;
#pragma unroll
          for (int c0 = 0; c0 < CIN; c0 += 16) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
              af[m] = *reinterpret_cast<const f32x4*>(patch + ((size_t)(py_[m] + dh + 1) * PW2 + (px_[m] + dw + 1)) * PSB + (c0 + q * 4) * 4);
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const f32x4*>(bbase + (size_t)n * 16 * (PSB / 4) + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                  acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
          }
        }
        if (s + 1 < 16) __syncthreads();
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
            if constexpr (bnbwd) {     // du = da * swish'(gamma * xhat + beta); the sums are those of the BatchNorm backward
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
      float* red = reinterpret_cast<float*>(Bs);     // [8][2][N]
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

int patch_cus() {
  static int cus[LdsOptIn::MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LdsOptIn::MAX_DEVICES) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

template <int H, int W, int CIN, int N, int TH, bool P3>
int patch_launch(const void* A, const void* Bp, float* C, float* stats, const IgemmGeom& g, const PatchEpi& ep, hipStream_t st) {
  using K = PatchCfg<H, W, CIN, N, TH, P3>;
  static LdsOptIn lds_opt_in;                               // (~105 KB of the CU's 160 KB; plane rows: 148-158 KB)
  static LdsOptIn lds_opt_in_bwd;
  const int Bt = g.G * g.Bg;
  const int grid = Bt * K::TILES < patch_cus() ? Bt * K::TILES : patch_cus();      // one resident block per CU, persistent
  if (ep.bn_y) {
    if (int e = lds_opt_in_bwd.ensure((const void*)tconv_patch_kernel<H, W, CIN, N, TH, P3, true>, (int)K::SMEM)) return e;
    hipLaunchKernelGGL((tconv_patch_kernel<H, W, CIN, N, TH, P3, true>), dim3(grid), dim3(512), K::SMEM, st, A, Bp, C, stats, Bt, g.Bg,
                       g.ldc, ep);
  } else {
    if (int e = lds_opt_in.ensure((const void*)tconv_patch_kernel<H, W, CIN, N, TH, P3, false>, (int)K::SMEM)) return e;
    hipLaunchKernelGGL((tconv_patch_kernel<H, W, CIN, N, TH, P3, false>), dim3(grid), dim3(512), K::SMEM, st, A, Bp, C, stats, Bt, g.Bg,
                       g.ldc, ep);
  }
  MMDYN_LAUNCH_CHECK();
}

template <bool P3>
int patch_try(const void* A, const void* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws, const IgemmGeom& g,
              hipStream_t st) {
  if (!patch_tiles(g.mode, g.Hi, g.Wi, g.Cin, g.Ho, g.Wo, g.N)) return 1;
  if (ws || g.splitk != 1) return MMDYN_ERR_SHAPE;          // (split-K is a DENSE-mode feature: the entry point has refused it)
  if (g.bn_y && g.ldc != g.N) return MMDYN_ERR_SHAPE;
  const PatchEpi ep{bias, C_act, g.act, g.bn_y, g.bn_mean, g.bn_rstd, g.bn_gamma, g.bn_beta, g.bwd_act};
  if (g.Hi == 16) return patch_launch<16, 16, 64, 32, 16, P3>(A, Bp, C, stats, g, ep, st);
  if (g.Hi == 32) return patch_launch<32, 32, 32, 32, 16, P3>(A, Bp, C, stats, g, ep, st);
  return patch_launch<64, 64, 32, 32, 8, P3>(A, Bp, C, stats, g, ep, st);
}

bool patch_p3_enabled() {      // (LAB, MMDYN_TCONV_PATCH_P3=0: the 32-channel up-sampling layers keep fp32 operands in the fp32x3 mode)
  static const bool off = [] {
    const char* e = lab_env("MMDYN_TCONV_PATCH_P3");
    return e && atoi(e) == 0;
  }();
  return !off;
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
  return patch_try<false>(A, Bp, bias, C, C_act, stats, ws, g, st);
}

// The same shapes on operands that arrive split (fp32x3, rows of [plane][Cin] bf16): same tile count, no workspace.
bool mmdyn_tconv_patch_p3_serves(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  return patch_p3_enabled() && G > 0 && Bg > 0 && patch_tiles(mode, Hi, Wi, Cin, Ho, Wo, N) > 0;
}
int mmdyn_tconv_patch_p3_try(const void* A, const void* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                             const IgemmGeom& g, hipStream_t st) {
  if (!patch_p3_enabled()) return 1;
  return patch_try<true>(A, Bp, bias, C, C_act, stats, ws, g, st);
}
