// fp32 implicit GEMM, NT form, on v_mfma_f32_16x16x4_f32 with WAVE-INDEPENDENT tiles (gfx950).
//
//   C[row][n] = sum_{tap, ci} A_tap[row][ci] * Bp[widx(tap)][n][ci]          (same contract as igemm_nt.hip)
//
// Replaces the ATen kernels behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear forward and input-gradient on the
// reference path (/root/reference/mmdyn/pytorch/models/vae.py:198-216, 264-277) in the fp32 mode.
//
// Why this shape of kernel: an fp32 MFMA occupies its SIMD's matrix pipe for 32 cycles per 16x16x4 instruction, i.e.
// the matrix cores consume operands 16x more slowly than in bf16.  At that rate a wave can feed itself straight from
// global memory: no LDS staging, no block barrier, every wave owns its output tile and never waits for another wave.
// Measured on MI355X (tests/microbench/direct_mfma.hip, dense 65536 x 128..256 x 1024..2048): 121-137 TFLOP/s against
// 84-114 for the LDS-tiled block kernel of igemm_nt.hip, whose four waves meet at two barriers per K-step.
//
//   * wave tile (16*MT) x (16*NT) outputs, MT*NT accumulators of 4 VGPRs; 64 x 64 (MT = NT = 4) wherever the problem
//     still gives every SIMD a wave, narrower tiles for small problems and N = 32;
//   * operand fragments: lane (r = l & 15, q = l >> 4) loads 16 bytes at [row r][k0 + 4q .. 4q+3]; MFMA j of the four fed
//     by one load multiplies k = {k0 + 4q' + j}: A and B share the K permutation, so no data movement between lanes;
//     one wave-load covers 16 rows x 64 contiguous bytes (the 16x16 shape is what makes the direct loads affordable: the
//     32x32x2 shape spreads a load over 32 rows x 32 bytes and measured 75 TFLOP/s);
//   * buffer loads with per-lane 32-bit row offsets and the K position in the scalar offset; rows outside the image or
//     the group get an out-of-range offset and the hardware returns zeros (no select, no branch in the K loop);
//   * K-stage = 32 channels of one tap (two loads per row tile = one full 128-byte line per row), register
//     double-buffered: the loads of stage s+1 are issued before the MFMAs of stage s (4096 matrix-pipe cycles);
//   * 2 waves per SIMD (<= 256 VGPRs): one wave's prologue / epilogue hides behind the other's MFMAs;
//   * the four waves of a workgroup take tiles that read the same rows of A (N-tiles / output-parity classes of one
//     M-tile) where there are such, so the shared lines hit in the CU's L1; workgroups are dealt to XCDs in contiguous
//     ranges of M (speed only).
// Epilogue as in igemm_nt.hip: bias, optional activated second output, per-tile BatchNorm partial sums, the
// BatchNorm+Swish backward of the layer below, split-K slabs.
#include "igemm_geom.h"
#include <cstdlib>
#include <cstdio>

namespace {

constexpr unsigned OOB = 0x80000000u;     // voffset beyond num_records: the buffer load returns 0

struct D16Args {
  IgemmGeom g;
  int tiles;         // wave tiles in the launch
  int NY, S;         // N / BN ; NY * nclasses (tiles that share one M-tile's rows)
  int MX;            // G * tiles_per_group
  float inv_hwr, inv_wr;
  unsigned a_bytes, b_bytes;
  int prio;
};

// n / d for 0 <= n < 2^23 with inv = 1.0f / d (one correction step makes the float estimate exact)
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
  int q = (int)((float)n * inv);
  int rem = n - q * d;
  q += (rem >= d) ? 1 : 0;
  q -= (rem < 0) ? 1 : 0;
  return q;
}

template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256, 2) void igemm_d16_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                           const float* __restrict__ bias, float* __restrict__ C,
                                                           float* __restrict__ C_act, float* __restrict__ stats,
                                                           float* __restrict__ ws, const D16Args p) {
  constexpr int BM = 16 * MT, BN = 16 * NT;
  const IgemmGeom& g = p.g;
  // the wave id is uniform but the compiler cannot see that: pin it to an SGPR, or every quantity derived from the tile
  // index (K position, scalar buffer offsets) lives in VGPRs and each buffer load is wrapped in a waterfall loop
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q4 = lane >> 4;
  if (p.prio) __builtin_amdgcn_s_setprio(3);      // (experiment: win the matrix-pipe arbitration against co-resident thin waves)

  // workgroup -> contiguous range of tiles per XCD (workgroups are dealt round-robin over the 8 XCDs)
  const int nb = gridDim.x, b = blockIdx.x;
  const int xq = nb >> 3, xr = nb & 7, xcd = b & 7;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
  // persistent form (grid smaller than the tile count): a wave walks tiles first, first + slots, ...
  for (int wt = logical * 4 + wave; wt < p.tiles; wt += nb * 4) {      // (no barrier anywhere in this kernel)
  const int per_split = p.MX * p.S;
  const int split = wt / per_split;
  const int rest = wt - split * per_split;
  const int mx = rest / p.S, inner = rest - mx * p.S;
  const int grp = mx / g.tiles_per_group, tile = mx - grp * g.tiles_per_group;
  const int cls = inner / p.NY;
  const int n0 = (inner - cls * p.NY) * BN;
  const int ph = cls >> 1, pw = cls & 1;
  const int HWr = g.Hr * g.Wr, Mg = g.Bg * HWr;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bp), 0, (int)p.b_bytes, 0x00020000);

  unsigned bo[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bo[nt] = (unsigned)(((n0 + nt * 16 + r) * g.Cin + 4 * q4) * 4);

  // TCONV_S1P0: a wave walks the four output pixels {(h,w),(h+4,w),(h,w+4),(h+4,w+4)} of its quad (valid-tap counts
  // always sum to 25: every wave does the same work), one accumulator set and epilogue per pixel
  const int nsub = (MODE == MMDYN_TCONV_S1P0) ? 4 : 1;
  for (int sub = 0; sub < nsub; ++sub) {
    int px_y = 0, px_x = 0, kh0 = 0, kw0 = 0, nkh = 4, nkw = 4;
    if (MODE == MMDYN_TCONV_S1P0) {
      const int quad = tile / g.tiles_per_pixel;
      px_y = (quad >> 2) + 4 * (sub >> 1);
      px_x = (quad & 3) + 4 * (sub & 1);
      kh0 = max(0, px_y - (g.Hi - 1));
      kw0 = max(0, px_x - (g.Wi - 1));
      nkh = min(3, px_y) - kh0 + 1;
      nkw = min(3, px_x) - kw0 + 1;
    }
    // rows this lane LOADS: mt*16 + r
    int rpix[MT], ry[MT], rx[MT];          // pixel index of the sample's (0,0) (< 0: row outside the group), y0, x0
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int rloc = mt * 16 + r;
      rpix[mt] = -1;
      ry[mt] = rx[mt] = 0;
      if (MODE == MMDYN_TCONV_S1P0) {
        const int sidx = (tile % g.tiles_per_pixel) * BM + rloc;
        if (sidx < g.Bg) {
          rpix[mt] = (grp * g.Bg + sidx) * g.Hi * g.Wi;
          ry[mt] = px_y;
          rx[mt] = px_x;
        }
      } else {
        const int ml = tile * BM + rloc;
        if (ml < Mg) {
          if (MODE == MMDYN_DENSE) {
            rpix[mt] = grp * Mg + ml;
          } else {
            const int s = fdiv(ml, HWr, p.inv_hwr);
            const int pp = ml - s * HWr;
            const int rr = fdiv(pp, g.Wr, p.inv_wr);
            const int cc = pp - rr * g.Wr;
            rpix[mt] = (grp * g.Bg + s) * g.Hi * g.Wi;
            ry[mt] = rr * g.rs + g.ro;
            rx[mt] = cc * g.rs + g.ro;
          }
        }
      }
    }

    const int cin_steps = g.Cin >> 5;
    const int total_steps = (MODE == MMDYN_TCONV_S1P0 ? nkh * nkw : g.ntaps) * cin_steps;
    const int per = (total_steps + g.splitk - 1) / g.splitk;
    const int s_begin = split * per;
    const int s_end = min(total_steps, s_begin + per);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int tap = s_begin / cin_steps;               // position of the NEXT fetch
    int cstep = s_begin - tap * cin_steps;
    f32x4 a0[2][MT], b0[2][NT], a1[2][MT], b1[2][NT];

#define D16_LOAD(AA, BB)                                                                                             \
  {                                                                                                                  \
    int dh = 0, dw = 0, wi = 0;                                                                                      \
    if (MODE == MMDYN_CONV) {                                                                                        \
      dh = tap >> 2;                                                                                                 \
      dw = tap & 3;                                                                                                  \
      wi = tap;                                                                                                      \
    } else if (MODE == MMDYN_TCONV_S2P1) {                                                                           \
      const int th = tap >> 1, tw = tap & 1;                                                                         \
      dh = ph - th;                                                                                                  \
      dw = pw - tw;                                                                                                  \
      wi = (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);                                                                \
    } else if (MODE == MMDYN_TCONV_S1P0) {                                                                           \
      const int ta = tap / nkw;                                                                                      \
      const int kh = kh0 + ta, kw = kw0 + (tap - ta * nkw);                                                          \
      dh = -kh;                                                                                                      \
      dw = -kw;                                                                                                      \
      wi = kh * 4 + kw;                                                                                              \
    }                                                                                                                \
    const int soa = cstep * 128;                                                                                     \
    const int sob = (wi * g.N * g.Cin + cstep * 32) * 4;                                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                              \
      unsigned off;                                                                                                  \
      if (MODE == MMDYN_DENSE) {                                                                                     \
        off = rpix[mt] >= 0 ? (unsigned)((rpix[mt] * g.Cin + 4 * q4) * 4) : OOB;                                     \
      } else {                                                                                                       \
        const int y = ry[mt] + dh, x = rx[mt] + dw;                                                                  \
        const bool ok = (rpix[mt] >= 0) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);           \
        off = ok ? (unsigned)(((rpix[mt] + y * g.Wi + x) * g.Cin + 4 * q4) * 4) : OOB;                               \
      }                                                                                                              \
      _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                  \
          AA[q][mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)off, soa + 64 * q, 0)); \
    }                                                                                                                \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int q = 0; q < 2; ++q)                  \
        BB[q][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (int)bo[nt], sob + 64 * q, 0)); \
    /* advance; after the last stage the position stays put: the extra prefetch repeats a valid stage */             \
    const bool last = (tap * cin_steps + cstep + 1 >= s_end);                                                        \
    const bool wrap = (cstep + 1 == cin_steps);                                                                      \
    const int ncstep = wrap ? 0 : cstep + 1;                                                                         \
    const int ntap = tap + (wrap ? 1 : 0);                                                                           \
    cstep = last ? cstep : ncstep;                                                                                   \
    tap = last ? tap : ntap;                                                                                         \
  }
#define D16_COMPUTE(AA, BB)                                                                                          \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int j = 0; j < 4; ++j)                        \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)            \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(AA[q][mt][j], BB[q][nt][j], acc[mt][nt], 0, 0, 0);

    if (s_begin < s_end) {
      D16_LOAD(a0, b0);
      for (int s = s_begin; s < s_end; s += 2) {
        D16_LOAD(a1, b1);
        __builtin_amdgcn_sched_barrier(0);       // keep the fetch of stage s+1 in front of the MFMAs of stage s
        D16_COMPUTE(a0, b0);
        D16_LOAD(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < s_end) {
          D16_COMPUTE(a1, b1);
        }
      }
    }
#undef D16_LOAD
#undef D16_COMPUTE

    // ---- epilogue: accumulator element e of tile (mt, nt) is C[mt*16 + q4*4 + e][nt*16 + r] ----
    float colsum[NT], colsq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) colsum[nt] = colsq[nt] = 0.f;
    const bool bnbwd = g.bn_y != nullptr;
    float bn_m[NT], bn_r[NT], bn_g[NT], bn_b[NT], bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + nt * 16 + r;
      bn_m[nt] = bnbwd ? g.bn_mean[(size_t)grp * g.N + col] : 0.f;
      bn_r[nt] = bnbwd ? g.bn_rstd[(size_t)grp * g.N + col] : 0.f;
      bn_g[nt] = bnbwd ? g.bn_gamma[col] : 0.f;
      bn_b[nt] = bnbwd ? g.bn_beta[col] : 0.f;
      bv[nt] = g.has_bias ? bias[col] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int rloc = mt * 16 + q4 * 4 + e;
        int ooff = -1, grow = 0;
        if (MODE == MMDYN_TCONV_S1P0) {
          const int sidx = (tile % g.tiles_per_pixel) * BM + rloc;
          if (sidx < g.Bg) ooff = (((grp * g.Bg + sidx) * g.Ho + px_y) * g.Wo + px_x) * g.ldc;
        } else {
          const int ml = tile * BM + rloc;
          if (ml < Mg) {
            grow = grp * Mg + ml;
            if (MODE == MMDYN_TCONV_S2P1) {
              const int s = fdiv(ml, HWr, p.inv_hwr);
              const int pp = ml - s * HWr;
              const int rr = fdiv(pp, g.Wr, p.inv_wr);
              const int cc = pp - rr * g.Wr;
              ooff = (((grp * g.Bg + s) * g.Ho + 2 * rr + ph) * g.Wo + 2 * cc + pw) * g.ldc;
            } else {
              ooff = grow * g.ldc;               // DENSE / CONV: rows are the output pixels in order
            }
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = n0 + nt * 16 + r;
          float v = acc[mt][nt][e];
          if (bnbwd) {
            float xh = 0.f;
            if (ooff >= 0) {
              xh = (g.bn_y[(size_t)ooff + col] - bn_m[nt]) * bn_r[nt];
              v *= swish_gradf_(bn_g[nt] * xh + bn_b[nt]);
            }
            colsum[nt] += v;
            colsq[nt] += v * xh;
          } else {
            colsum[nt] += v;
            colsq[nt] += v * v;
          }
          if (ooff >= 0) {
            if (g.splitk > 1) {
              ws[((size_t)split * g.rows_total + grow) * g.N + col] = v;
            } else {
              v += bv[nt];
              C[(size_t)ooff + col] = v;
              if (g.want_act_out) C_act[(size_t)ooff + col] = apply_act(v, g.act);
            }
          }
        }
      }
    }
    if (g.want_stats) {
      // rows beyond the group are zero operands -> contribute exactly 0.  Column (nt, r): sum over the lane's 4*MT
      // elements, then over the four lanes q4 = 0..3 that hold the other rows; deterministic, no atomics
      int T = g.nclasses * g.tiles_per_group, slot = cls * g.tiles_per_group + tile;
      if (MODE == MMDYN_TCONV_S1P0) {
        T = g.Ho * g.Wo * g.tiles_per_pixel;
        slot = (px_y * g.Wo + px_x) * g.tiles_per_pixel + tile % g.tiles_per_pixel;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float s = colsum[nt], q = colsq[nt];
        s += __shfl_xor(s, 16, 64);
        q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (q4 == 0) {
          const size_t base = ((size_t)(grp * T + slot) * 2) * g.N + n0 + nt * 16 + r;
          stats[base] = s;
          stats[base + g.N] = q;
        }
      }
    }
  }  // sub-pixel walk
  }  // tile walk
}

// ---- tile choice ------------------------------------------------------------------------------------------------
// 64 x 64 wave tiles (one load per 8 MFMAs) wherever they still give every SIMD of the chip a wave; smaller problems
// take narrower tiles so that all 1024 SIMDs work (a 64 x 32 tile issues 1.5x the loads per MFMA and measured ~0.78 of
// the 64 x 64 rate, still far better than idle SIMDs); N = 32 (mod 64): 128 x 32.
struct D16Tile { int mt, nt; };

static int d16_tiles(const IgemmGeom& g, int mt, int nt) {
  const int bm = 16 * mt, bn = 16 * nt;
  long per_group = (g.mode == MMDYN_TCONV_S1P0) ? 16L * ceil_div(g.Bg, bm) : ceil_div(g.Bg * g.Hr * g.Wr, bm);
  return (int)(g.G * per_group * (g.N / bn) * g.nclasses * g.splitk);
}

static bool d16_pick(const IgemmGeom& g, D16Tile* t) {
  static const D16Tile wide[] = {{4, 4}, {4, 2}, {2, 2}};
  static const D16Tile narrow[] = {{8, 2}, {4, 2}, {2, 2}};
  if (const char* ov = getenv("MMDYN_D16_TILE")) {          // kernel experiments and tests: force one tile shape
    int mt = 0, nt = 0;
    if (sscanf(ov, "%d,%d", &mt, &nt) == 2) {
      const bool known = (mt == 4 && nt == 4) || (mt == 8 && nt == 2) || (mt == 4 && nt == 2) || (mt == 2 && nt == 2);
      if (!known || g.N % (16 * nt)) return false;
      t->mt = mt;
      t->nt = nt;
      return true;
    }
  }
  const D16Tile* cand = (g.N % 64 == 0) ? wide : narrow;
  static int min_tiles = -1, min_steps = -1, only_big = -1;
  if (min_tiles < 0) {
    const char* a = getenv("MMDYN_D16_MIN_TILES");
    const char* b = getenv("MMDYN_D16_MIN_STEPS");
    const char* c = getenv("MMDYN_D16_ONLY_BIG");
    min_tiles = a ? atoi(a) : 512;
    min_steps = b ? atoi(b) : 0;
    only_big = c ? atoi(c) : 0;
  }
  const int steps = (g.mode == MMDYN_TCONV_S1P0 ? 6 : g.ntaps) * (g.Cin / 32) / g.splitk;
  if (steps < min_steps) return false;
  if (only_big) {                   // experiments: the first candidate tile or nothing
    *t = cand[0];
    return d16_tiles(g, cand[0].mt, cand[0].nt) >= min_tiles;
  }
  D16Tile best = cand[0];
  for (int i = 0; i < 3; ++i) {
    best = cand[i];
    if (d16_tiles(g, cand[i].mt, cand[i].nt) >= 1024) break;
  }
  *t = best;
  // too little work even for the smallest tile: the block-tiled kernel (4 waves per 64 x 64 tile) spreads it further
  return d16_tiles(g, best.mt, best.nt) >= min_tiles;
}

// Opt-in (MMDYN_D16=1, or a forced tile shape): alone on the chip these kernels are the fastest fp32 GEMMs of the library
// on long-K shapes (119-137 TFLOP/s), but a launch keeps two 212-register waves resident on every SIMD for its whole
// duration, and inside the two-lane training step that starves the other lane's kernels: measured 7.26-7.9 ms per step
// against 7.0 ms with the LDS-tiled kernels, whatever subset of shapes was routed here (profiles/r2/d16_step_sweep.txt).
static bool d16_serves(const IgemmGeom& g) {
  const char* en = getenv("MMDYN_D16");
  if (!(en && atoi(en) != 0) && !getenv("MMDYN_D16_TILE")) return false;
  if (g.mode != MMDYN_DENSE && g.mode != MMDYN_CONV && g.mode != MMDYN_TCONV_S2P1 && g.mode != MMDYN_TCONV_S1P0) return false;
  if (g.a_b16 || g.c_b16 || g.bny_b16 || g.b_b16) return false;
  if (g.Cin % 32 || g.N % 32) return false;
  if ((long)g.Bg * g.Hr * g.Wr >= (1L << 23)) return false;                                      // fdiv range
  if ((long)g.G * g.Bg * g.Hi * g.Wi * g.Cin >= (1L << 29)) return false;                        // 32-bit byte offsets
  if ((long)g.G * g.Bg * g.Ho * g.Wo * g.ldc >= (1L << 31)) return false;
  if ((long)16 * g.N * g.Cin >= (1L << 29)) return false;
  return true;
}

template <int MODE, int MT, int NT>
static int launch_mt(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                     IgemmGeom g, hipStream_t st) {
  constexpr int BM = 16 * MT, BN = 16 * NT;
  D16Args p{};
  g.tiles_per_group = ceil_div(g.Bg * g.Hr * g.Wr, BM);
  if (MODE == MMDYN_TCONV_S1P0) {
    g.tiles_per_pixel = ceil_div(g.Bg, BM);
    g.s1p0_split = 1;
    g.tiles_per_group = 16 * g.tiles_per_pixel;       // 16 pixel quads per group, 4 pixels walked per wave
  }
  p.g = g;
  p.NY = g.N / BN;
  p.S = p.NY * g.nclasses;
  p.MX = g.G * g.tiles_per_group;
  p.tiles = p.MX * p.S * g.splitk;
  p.inv_hwr = 1.0f / (float)(g.Hr * g.Wr);
  p.inv_wr = 1.0f / (float)g.Wr;
  p.a_bytes = (unsigned)((long)g.G * g.Bg * g.Hi * g.Wi * g.Cin * 4);
  const int ntaps_w = (g.mode == MMDYN_DENSE) ? 1 : 16;
  p.b_bytes = (unsigned)((long)ntaps_w * g.N * g.Cin * 4);
  // MMDYN_D16_BLOCKS = n: persistent launch of at most n workgroups (256 = one wave per SIMD: the two lanes of the
  // step then hold one wave slot per SIMD each instead of one lane's launch filling both)
  static int max_blocks = -1, prio = 0;
  if (max_blocks < 0) {
    const char* e = getenv("MMDYN_D16_BLOCKS");
    max_blocks = e ? atoi(e) : 0;
    prio = getenv("MMDYN_D16_PRIO") != nullptr;
  }
  p.prio = prio;
  int blocks = ceil_div(p.tiles, 4);
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL((igemm_d16_kernel<MODE, MT, NT>), dim3(blocks), dim3(256), 0, st, A, Bp, bias, C, C_act, stats, ws, p);
  MMDYN_LAUNCH_CHECK();
}

template <int MT, int NT>
static int launch_t(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                    const IgemmGeom& g, hipStream_t st) {
  switch (g.mode) {
    case MMDYN_DENSE: return launch_mt<MMDYN_DENSE, MT, NT>(A, Bp, bias, C, C_act, stats, ws, g, st);
    case MMDYN_CONV: return launch_mt<MMDYN_CONV, MT, NT>(A, Bp, bias, C, C_act, stats, ws, g, st);
    case MMDYN_TCONV_S2P1: return launch_mt<MMDYN_TCONV_S2P1, MT, NT>(A, Bp, bias, C, C_act, stats, ws, g, st);
    default: return launch_mt<MMDYN_TCONV_S1P0, MT, NT>(A, Bp, bias, C, C_act, stats, ws, g, st);
  }
}

}  // namespace

int mmdyn_igemm_d16_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats,
                        float* ws, IgemmGeom g, int stride, int offset, hipStream_t st) {
  if (g.bn_y && !g.bn_mean) return 1;          // activation-only backward epilogue: not built here

  (void)stride;
  (void)offset;
  D16Tile t;
  if (!d16_serves(g) || !d16_pick(g, &t)) return 1;
  if (t.mt == 4 && t.nt == 4) return launch_t<4, 4>(A, Bp, bias, C, C_act, stats, ws, g, st);
  if (t.mt == 8 && t.nt == 2) return launch_t<8, 2>(A, Bp, bias, C, C_act, stats, ws, g, st);
  if (t.mt == 4 && t.nt == 2) return launch_t<4, 2>(A, Bp, bias, C, C_act, stats, ws, g, st);
  return launch_t<2, 2>(A, Bp, bias, C, C_act, stats, ws, g, st);
}

// number of BatchNorm partial-sum tiles per group this kernel family writes for the shape (0: shape not served)
int mmdyn_igemm_d16_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  IgemmGeom g{};
  g.mode = mode;
  g.G = G;
  g.Bg = Bg;
  g.Hi = Hi;
  g.Wi = Wi;
  g.Cin = Cin;
  g.Ho = Ho;
  g.Wo = Wo;
  g.N = N;
  g.ldc = N;
  g.splitk = 1;
  g.nclasses = (mode == MMDYN_TCONV_S2P1) ? 4 : 1;
  g.ntaps = (mode == MMDYN_CONV || mode == MMDYN_TCONV_S1P0) ? 16 : (mode == MMDYN_TCONV_S2P1 ? 4 : 1);   // as igemm_entry
  g.Hr = (mode == MMDYN_TCONV_S2P1) ? Hi : Ho;
  g.Wr = (mode == MMDYN_TCONV_S2P1) ? Wi : Wo;
  D16Tile t;
  if (!d16_serves(g) || !d16_pick(g, &t)) return 0;
  if (mode == MMDYN_TCONV_S1P0) return Ho * Wo * ceil_div(Bg, 16 * t.mt);
  return g.nclasses * ceil_div(Bg * g.Hr * g.Wr, 16 * t.mt);
}
