// MFMA implicit GEMM, NT form, fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32, gfx950).
//
//   C[row][n] = sum_{tap, ci} A_tap[row][ci] * Bp[widx(tap)][n][ci]
//
// Replaces the ATen kernels behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear forward and
// input-gradient on the reference path (/root/reference/mmdyn/pytorch/models/vae.py:198-216, 264-277).
//
// Layout and tiling (designed for CDNA4, not a port of a warp-32 tiling):
//   * activations are channels-last, so one GEMM row of one tap is a contiguous run of Cin floats:
//     every global load is a full 16-byte lane access inside a 128-byte segment;
//   * block tile BM x BN x 32, 256 threads = 4 wavefronts of 64, each wave owns (WM/32) x (WN/32)
//     32x32 MFMA tiles (16 accumulator VGPRs each);
//   * both operand tiles live in LDS as [row][36] (row padded by one 16-byte slot): the fragment
//     read is one conflict-free ds_read_b128 per 4 MFMAs.  The K order inside a group of 8 is
//     permuted (lane half h consumes k = 8q + 4h + j for MFMA j) -- legal because A and B use the
//     same permutation;
//   * register-staged prefetch: the global loads of K-step s+1 are issued before the MFMAs of step s;
//   * epilogue: optional bias, optional second (activated) output, optional per-tile BatchNorm
//     partial sums (column sums of the tile and of its squares), deterministic (no atomics).
#include "common.h"
#include "igemm_geom.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {


constexpr int BK = 32;              // K-step (channels of one tap per stage)
constexpr int LDS_LD32 = BK + 4;    // row stride of the 32x32x2 variants: one 16-byte pad slot keeps ds_read_b128 conflict-free
constexpr int GRANS = BK / 4;       // 16-byte granules per tile row
constexpr int ROWS_PER_PASS = 256 / GRANS;

// BF16 = true: the fp32 tiles in LDS are rounded to bf16 (RNE) as they are read into fragments and multiplied
// with v_mfma_f32_32x32x16_bf16 (fp32 accumulate): 16x fewer matrix-core cycles, the staging is unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pack_bf16(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo[0]; r[1] = (__bf16)lo[1]; r[2] = (__bf16)lo[2]; r[3] = (__bf16)lo[3];
  r[4] = (__bf16)hi[0]; r[5] = (__bf16)hi[1]; r[6] = (__bf16)hi[2]; r[7] = (__bf16)hi[3];
  return r;
}

// WIDE (bf16 activations AND bf16 packed weights, Cin % 64 == 0): K-step of 64 channels moved as 16-byte granules of
// eight bf16; thread -> (row, granule) mapping and LDS bytes per row (144) are the same as in the 32-channel variants
// M16 (fp32 only): the wave's WM x WN outputs are 16x16 tiles of v_mfma_f32_16x16x4_f32 instead of 32x32 tiles of
// v_mfma_f32_32x32x2_f32 -- same rate per clock, but a 32x32-output wave then owns FOUR independent accumulators instead of
// one dependent chain.  Measured (tests/microbench/lds_mfma_shape.hip, MI355X): +7..17 % on the 64x64 block tile, no
// difference on 128x128 (which keeps the 32x32 shape).  LDS row stride 40 floats keeps its ds_read_b128 conflict-free.
// F16 (with BF16, fp32 storage on both sides): the 16-bit operands are IEEE half (RNE) and the product runs on
// v_mfma_f32_32x32x16_f16 -- BASELINE configs[4] "MFMA fp16 conv with fp32 accumulate"; everything else as in the bf16 mode.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// X3 (fp32 operands, fp32 results): the product runs on the bf16 matrix cores through the EXACT three-term split of every fp32
// operand, x = hi + mid + lo with hi, mid, lo bf16 (split3_bf16, common.h), and six of the nine cross products -- hi.hi, hi.mid,
// mid.hi, mid.mid, hi.lo, lo.hi; the dropped three are together below 2^-23 |a||b|, less than one fp32 rounding of the product --
// accumulated in fp32, smallest terms first.  v_mfma_f32_32x32x16_bf16 runs 16x the rate of the fp32 shapes, so the six products
// cost 6/16 of the native matrix time; the split is VALU work on the store side (once per element per block) and the tiles take
// three bf16 planes in LDS.  Error against fp64 on the same data: relative L2 4.7e-7, native fp32 MFMA 5.4e-7
// (tests/microbench/ab_x3.py, profiles/r4/ab_x3_*.txt; docs/LAB_NOTES.md D.g and F).
template <int MODE, int BM, int BN, int WM, int WN, bool BF16, bool A16, bool B16, bool WIDE = false, bool M16 = false,
          bool F16 = false, bool ONEPX = false, bool X3 = false>
__global__ __launch_bounds__(256) void igemm_nt_kernel(const float* __restrict__ A,
                                                       const float* __restrict__ Bp,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ C, float* __restrict__ C_act,
                                                       float* __restrict__ stats, float* __restrict__ ws,
                                                       const IgemmGeom g) {
  static_assert(!M16 || !BF16, "the 16x16x4 shape is the fp32 variant");
  static_assert(!ONEPX || MODE == MMDYN_TCONV_S1P0, "one output pixel per block is a variant of the k4 s1 p0 walk");
  static_assert(!F16 || BF16, "fp16 operands are a 16-bit matrix-core variant");
  static_assert(!X3 || (!BF16 && !M16 && MODE != MMDYN_IM2COL3), "the three-term split is a variant of the fp32 kernel");
  // (F16 with A16 / B16 / WIDE: the 16-bit STORAGE is IEEE half too -- precision "fp16s"; the raw granules go to LDS as they
  // are, exactly like the bf16 ones, and the epilogue reads / writes half)
  typedef typename std::conditional<F16, half_t, bf16_t>::type st16_t;
  constexpr int TS = M16 ? 16 : 32;                   // side of one MFMA output tile
  constexpr int NE = M16 ? 4 : 16;                    // accumulator registers per tile
  constexpr int LDS_LD = M16 ? BK + 8 : BK + 4;       // fp32 tile row stride (floats): conflict-free fragment reads
  constexpr int MT = WM / TS, NT = WN / TS;
  constexpr int WAVES_N = BN / WN;
  constexpr int WAVES_M = BM / WM;
  static_assert(WAVES_N * WAVES_M == 4, "4 waves per block");
  static_assert(!WIDE || (BF16 && A16 && B16), "the 64-channel K-step is a bf16-operand variant");
  constexpr int KB = WIDE ? 64 : BK;                  // channels per K-step
  constexpr int A_LOADS = BM / ROWS_PER_PASS, B_LOADS = BN / ROWS_PER_PASS;
  static_assert(A_LOADS >= 1 && B_LOADS >= 1, "tile smaller than one load pass");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);            // [BM][LDS_LD] then [BN][LDS_LD]
  float* Bs = As + BM * LDS_LD;
  // [BM][4]: b, y0, x0, out offset (-1: none); behind the tiles (X3: three bf16 planes of 80-byte rows)
  int* rowinfo = X3 ? reinterpret_cast<int*>(smem + (size_t)3 * (BM + BN) * (BK + 8) * 2) : reinterpret_cast<int*>(As + (BM + BN) * LDS_LD);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int HWr = g.Hr * g.Wr;
  const int Mg = g.Bg * HWr;
  // XCD-aware block order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
  // All blocks that read the same rows of A -- the N-tiles and the output-parity classes of one M-tile -- get
  // linear ids that are equal modulo 8, i.e. the same XCD, so the tile's activations are fetched into one L2.
  // Fewer than 5 M-tiles (a 256-row Linear layer has 4): dealing M-tiles to XCDs would leave XCDs idle, so each M-tile's
  // N-tiles are spread over 8 / MXp XCDs instead (MXp = M-tiles rounded up to a power of two).
  const int NY = g.N / BN, S = NY * g.nclasses;
  const int MX = g.G * g.tiles_per_group, MX8 = (MX + 7) >> 3;
  const int L = blockIdx.x;
  const int m_lo = L & 7, r8 = L >> 3;
  int inner, mx, split;
  if (MX > 4) {
    const int rest = r8 / S;
    inner = r8 % S;
    mx = (rest % MX8) * 8 + m_lo;
    split = rest / MX8;
  } else {
    const int MXp = MX > 2 ? 4 : MX, nparts = 8 / MXp, Sp = (S + nparts - 1) / nparts;
    mx = m_lo % MXp;
    inner = (r8 % Sp) * nparts + m_lo / MXp;
    split = r8 / Sp;
    if (inner >= S) return;
  }
  if (mx >= MX) return;
  const int grp = mx / g.tiles_per_group, tile = mx - grp * g.tiles_per_group;
  const int cls = inner / NY;
  const int n0 = (inner - cls * NY) * BN;
  const int ph = cls >> 1, pw = cls & 1;
  // grouped launch: this group's own weights and bias (strides are 0 when the groups share them)
  Bp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Bp) + (size_t)grp * g.b_group_stride * (B16 ? 2 : 4));
  if (bias) bias += (size_t)grp * g.bias_group_stride;

  // TCONV_S1P0: rows are ordered (output pixel, sample) so that a tile sees ONE output pixel and multiplies only
  // the kernel taps that reach the input for it (1..16 of them for k4 s1 p0).  To balance the blocks, each block
  // walks the FOUR pixels {(h,w),(h+4,w),(h,w+4),(h+4,w+4)} of the 8x8 output: their valid-tap counts always sum
  // to (1+4)*(1+4) = 25, so every block does identical work.
  // (g.s1p0_split = 2: the quad walk is shared by TWO blocks, {(h,w),(h+4,w+4)} and {(h+4,w),(h,w+4)} -- 8..17 valid taps
  // each instead of a uniform 25, but twice the blocks in flight: used by the latency-bound bf16 variants)
  // (ONEPX: ONE pixel per block, 1..16 taps of work, pixels dealt heaviest first -- along each axis 3,4,2,5,1,6,0,7 with
  // 4,4,3,3,2,2,1,1 valid taps -- so the blocks that finish a launch are the light ones.  Without the four-pixel loop
  // around the K loop and the epilogue the kernel needs about half the registers.)
  const int nsub = (MODE == MMDYN_TCONV_S1P0 && !ONEPX) ? 4 / g.s1p0_split : 1;
  const int qtile = (MODE == MMDYN_TCONV_S1P0 && !ONEPX) ? tile / g.s1p0_split : tile;          // tile index inside the quad
  const int qhalf = (MODE == MMDYN_TCONV_S1P0 && !ONEPX) ? tile - qtile * g.s1p0_split : 0;
  for (int ksub = 0; ksub < nsub; ++ksub) {
  const int sub = (MODE == MMDYN_TCONV_S1P0 && g.s1p0_split == 2) ? (qhalf ? (ksub ? 2 : 1) : (ksub ? 3 : 0)) : ksub;
  int px_y = 0, px_x = 0, kh0 = 0, kw0 = 0, nkh = 4, nkw = 4;
  if (MODE == MMDYN_TCONV_S1P0) {
    const int quad = qtile / g.tiles_per_pixel;
    if constexpr (ONEPX) {
      const int iy = quad >> 3, ix = quad & 7;
      px_y = (iy & 1) ? 4 + (iy >> 1) : 3 - (iy >> 1);
      px_x = (ix & 1) ? 4 + (ix >> 1) : 3 - (ix >> 1);
    } else {
      px_y = (quad >> 2) + 4 * (sub >> 1);
      px_x = (quad & 3) + 4 * (sub & 1);
    }
    kh0 = max(0, px_y - (g.Hi - 1));
    kw0 = max(0, px_x - (g.Wi - 1));
    nkh = min(3, px_y) - kh0 + 1;
    nkw = min(3, px_x) - kw0 + 1;
    if (ksub > 0) __syncthreads();     // previous sub-tile's epilogue still reads rowinfo / stats scratch
  }
  for (int r = tid; r < BM; r += 256) {
    int ml = tile * BM + r;
    int ib = -1, y0 = 0, x0 = 0, ooff = -1;
    if (MODE == MMDYN_TCONV_S1P0) {
      const int sidx = (qtile % g.tiles_per_pixel) * BM + r;
      if (sidx < g.Bg) {
        ib = grp * g.Bg + sidx;
        y0 = px_y;
        x0 = px_x;
        ooff = ((ib * g.Ho + px_y) * g.Wo + px_x) * g.ldc;
      }
    } else if (ml < Mg) {
      int s = ml / HWr;
      int p = ml - s * HWr;
      int rr = p / g.Wr;
      int cc = p - rr * g.Wr;
      ib = grp * g.Bg + s;
      y0 = rr * g.rs + g.ro;
      x0 = cc * g.rs + g.ro;
      int oy = rr * g.os + ph, ox = cc * g.os + pw;
      ooff = ((ib * g.Ho + oy) * g.Wo + ox) * g.ldc;
    }
    rowinfo[r * 4 + 0] = ib;
    rowinfo[r * 4 + 1] = y0;
    rowinfo[r * 4 + 2] = x0;
    rowinfo[r * 4 + 3] = ooff;
  }
  __syncthreads();

  const int lrow = tid / GRANS, gran = tid % GRANS;
  int rb[A_LOADS], ry[A_LOADS], rx[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    int r = lrow + ROWS_PER_PASS * i;
    rb[i] = rowinfo[r * 4 + 0];
    ry[i] = rowinfo[r * 4 + 1];
    rx[i] = rowinfo[r * 4 + 2];
  }

  const int cin_steps = g.Cin / KB;
  const int total_steps = (MODE == MMDYN_TCONV_S1P0 ? nkh * nkw : g.ntaps) * cin_steps;
  const int per_split = (total_steps + g.splitk - 1) / g.splitk;
  const int s_begin = split * per_split;
  const int s_end = min(total_steps, s_begin + per_split);

  // Branch-free operand fetch: an out-of-image (or out-of-range) row reads a valid dummy address and is zeroed by
  // a select, so one K-step is a single basic block and the scheduler can slot the address arithmetic and the
  // global loads between the 64-cycle MFMAs instead of in front of them.
  // (two register sets: the M16 variant fetches TWO K-steps ahead -- see PF2 at the K loop; the others use set 0 only)
  f32x4 raS[2][A_LOADS], rbvS[2][B_LOADS];
  unsigned okS[2] = {0u, 0u};
  uint2 ra16[A_LOADS];          // A16: the raw bf16 granule (widening it here would wait for the load before the MFMAs)
  uint2 rbv16[B_LOADS];         // B16: likewise for bf16 packed weights
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (native vector: arrays of HIP's uint4 struct spill to scratch)
  u32x4 raw[A_LOADS], rbw[B_LOADS];   // WIDE: eight bf16 per granule
  // bf16 matrix-core variants keep the tiles in LDS as bf16 ([row][40] halves: 32 + one 16-byte pad slot): half the
  // LDS bytes, one ds_read_b128 per operand and 16-deep MFMA, conversion once per element on the store side
  constexpr int LDH = KB + 8;
  bf16_t* As16 = reinterpret_cast<bf16_t*>(smem);            // (X3: [plane][BM][LDH], then [plane][BN][LDH])
  bf16_t* Bs16 = As16 + (X3 ? 3 : 1) * BM * LDH;
  int tap = s_begin / cin_steps;            // running (tap, channel-step) position of the NEXT fetch
  int cstep = s_begin - tap * cin_steps;
  auto gload = [&](f32x4 (&ra)[A_LOADS], f32x4 (&rbv)[B_LOADS], unsigned& okmask) {
    const int c0 = cstep * KB + gran * (WIDE ? 8 : 4);
    int dh = 0, dw = 0, wi = 0;
    if (MODE == MMDYN_CONV) {
      dh = tap >> 2;
      dw = tap & 3;
      wi = tap;
    } else if (MODE == MMDYN_TCONV_S2P1) {
      const int th = tap >> 1, tw = tap & 1;
      dh = ph - th;
      dw = pw - tw;
      wi = (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);
    } else if (MODE == MMDYN_TCONV_S1P0) {
      const int a = tap / nkw;
      const int kh = kh0 + a, kw = kw0 + (tap - a * nkw);
      dh = -kh;
      dw = -kw;
      wi = kh * 4 + kw;
    }
    if (MODE == MMDYN_IM2COL3) {
      // A is the reference's NCHW 3-channel image; virtual K index k = ci*16 + kh*4 + kw (48 real + 16 zero):
      // this granule holds the 4 kw taps of (ci, kh) for the row's output pixel (k4 s2 p1 window)
      const int ci = c0 >> 4, kh = (c0 >> 2) & 3;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const int y = 2 * ry[i] - 1 + kh, x0 = 2 * rx[i] - 1;
        const bool okr = (rb[i] >= 0) & (ci < 3) & ((unsigned)y < (unsigned)g.Hi);
        const int base = okr ? ((rb[i] * 3 + ci) * g.Hi + y) * g.Wi : 0;
        unsigned m = 0;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const int xx = x0 + kw;
          const bool ok = okr & ((unsigned)xx < (unsigned)g.Wi);
          ra[i][kw] = A[(size_t)base + (ok ? xx : 0)];
          m |= ok ? (1u << kw) : 0u;
        }
        okmask = (okmask & ~(0xFu << (4 * i))) | (m << (4 * i));
      }
    } else
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int y = ry[i] + dh, x = rx[i] + dw;
      const bool ok = (rb[i] >= 0) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
      const int pix = ok ? (rb[i] * g.Hi + y) * g.Wi + x : 0;
      if constexpr (WIDE)
        raw[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(A) + (size_t)pix * g.Cin + c0);
      else if constexpr (A16)     // compile-time: the fetch stays one branch-free basic block
        ra16[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(A) + (size_t)pix * g.Cin + c0);
      else
        ra[i] = *reinterpret_cast<const f32x4*>(A + (size_t)pix * g.Cin + c0);
      okmask = ok ? (okmask | (0xFu << (4 * i))) : (okmask & ~(0xFu << (4 * i)));   // consumed at lds_store
    }
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
      const int n = n0 + lrow + ROWS_PER_PASS * j;
      if constexpr (WIDE)
        rbw[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(Bp) + ((size_t)wi * g.N + n) * g.Cin + c0);
      else if constexpr (B16)
        rbv16[j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(Bp) + ((size_t)wi * g.N + n) * g.Cin + c0);
      else
        rbv[j] = *reinterpret_cast<const f32x4*>(Bp + ((size_t)wi * g.N + n) * g.Cin + c0);
    }
    // advance; the fetch issued during the last K-step is a harmless repeat of a valid tile (keeps the loop
    // body free of branches)
    const bool last = (tap * cin_steps + cstep + 1 >= s_end);
    const bool wrap = (cstep + 1 == cin_steps);
    const int ncstep = wrap ? 0 : cstep + 1;
    const int ntap = tap + (wrap ? 1 : 0);
    cstep = last ? cstep : ncstep;
    tap = last ? tap : ntap;
  };
  auto lds_store = [&](const f32x4 (&ra)[A_LOADS], const f32x4 (&rbv)[B_LOADS], const unsigned okmask) {
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const unsigned m = okmask >> (4 * i);
        uint2 h, md, l;
        split3_bf16((m & 1u) ? ra[i][0] : 0.f, (m & 2u) ? ra[i][1] : 0.f, h.x, md.x, l.x);
        split3_bf16((m & 4u) ? ra[i][2] : 0.f, (m & 8u) ? ra[i][3] : 0.f, h.y, md.y, l.y);
        const int o = (lrow + ROWS_PER_PASS * i) * LDH + gran * 4;
        *reinterpret_cast<uint2*>(&As16[o]) = h;
        *reinterpret_cast<uint2*>(&As16[BM * LDH + o]) = md;
        *reinterpret_cast<uint2*>(&As16[2 * BM * LDH + o]) = l;
      }
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) {
        uint2 h, md, l;
        split3_bf16(rbv[j][0], rbv[j][1], h.x, md.x, l.x);
        split3_bf16(rbv[j][2], rbv[j][3], h.y, md.y, l.y);
        const int o = (lrow + ROWS_PER_PASS * j) * LDH + gran * 4;
        *reinterpret_cast<uint2*>(&Bs16[o]) = h;
        *reinterpret_cast<uint2*>(&Bs16[BN * LDH + o]) = md;
        *reinterpret_cast<uint2*>(&Bs16[2 * BN * LDH + o]) = l;
      }
      return;
    }
    if constexpr (WIDE) {
      const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        *reinterpret_cast<u32x4*>(&As16[(lrow + ROWS_PER_PASS * i) * LDH + gran * 8]) = ((okmask >> (4 * i)) & 1u) ? raw[i] : zero;
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j)
        *reinterpret_cast<u32x4*>(&Bs16[(lrow + ROWS_PER_PASS * j) * LDH + gran * 8]) = rbw[j];
      return;
    }
    if constexpr (BF16) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        const unsigned m = okmask >> (4 * i);
        uint2 v;
        if constexpr (A16) {
          v.x = (m & 1u) ? ra16[i].x : 0u;           // (all four mask bits are equal outside IM2COL3)
          v.y = (m & 1u) ? ra16[i].y : 0u;
        } else {
          if constexpr (F16) {
            v.x = pack2_f16((m & 1u) ? ra[i][0] : 0.f, (m & 2u) ? ra[i][1] : 0.f);
            v.y = pack2_f16((m & 4u) ? ra[i][2] : 0.f, (m & 8u) ? ra[i][3] : 0.f);
          } else {
            v.x = pack2_bf16((m & 1u) ? ra[i][0] : 0.f, (m & 2u) ? ra[i][1] : 0.f);
            v.y = pack2_bf16((m & 4u) ? ra[i][2] : 0.f, (m & 8u) ? ra[i][3] : 0.f);
          }
        }
        *reinterpret_cast<uint2*>(&As16[(lrow + ROWS_PER_PASS * i) * LDH + gran * 4]) = v;
      }
#pragma unroll
      for (int j = 0; j < B_LOADS; ++j) {
        uint2 v;
        if constexpr (B16) {
          v = rbv16[j];
        } else {
          if constexpr (F16) {
            v.x = pack2_f16(rbv[j][0], rbv[j][1]);
            v.y = pack2_f16(rbv[j][2], rbv[j][3]);
          } else {
            v.x = pack2_bf16(rbv[j][0], rbv[j][1]);
            v.y = pack2_bf16(rbv[j][2], rbv[j][3]);
          }
        }
        *reinterpret_cast<uint2*>(&Bs16[(lrow + ROWS_PER_PASS * j) * LDH + gran * 4]) = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const unsigned m = okmask >> (4 * i);
      f32x4 v;
      v[0] = (m & 1u) ? ra[i][0] : 0.f;
      v[1] = (m & 2u) ? ra[i][1] : 0.f;
      v[2] = (m & 4u) ? ra[i][2] : 0.f;
      v[3] = (m & 8u) ? ra[i][3] : 0.f;
      *reinterpret_cast<f32x4*>(&As[(lrow + ROWS_PER_PASS * i) * LDS_LD + gran * 4]) = v;
    }
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j)
      *reinterpret_cast<f32x4*>(&Bs[(lrow + ROWS_PER_PASS * j) * LDS_LD + gran * 4]) = rbv[j];
  };

  typedef float accv_t __attribute__((ext_vector_type(NE)));
  accv_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[mt][nt][e] = 0.f;

  // 32x32x2: lane (row l&31, half l>>5) reads k = 8q + 4h .. +3;  16x16x4: lane (row l&15, quarter l>>4) reads
  // k = 16q + 4*(l>>4) .. +3 (MFMA j multiplies the j-th element of the four lanes' granules)
  const int frag_off = M16 ? (lane & 15) * LDS_LD + (lane >> 4) * 4 : (lane & 31) * LDS_LD + (lane >> 5) * 4;
  // PF2: global loads issued TWO K-steps ahead of their use (two register sets).  In the stand-alone dense GEMM of
  // tests/microbench/lds_mfma_shape.hip the second step of lead is worth 3-5 % (removing the loads from the loop
  // altogether, a timing diagnostic, +20 %; removing the barriers nothing: the single-stage kernel is exposed to the
  // fetch, not to its barriers).  HERE it loses: with the gather arithmetic of two fetches live the compiler needs 92
  // VGPRs + 40 AGPRs (3 waves per SIMD instead of 6) and the launches run 8-10 % SLOWER (step 7.26 vs 6.98 ms,
  // profiles/r2/igemm_tile_x_mfma_shape_sweep.txt); capping the registers spills.  Kept switched off.
  constexpr bool PF2 = false;
  auto mfma_step = [&]() {
      const float* Ac = As;
    const float* Bc = Bs;
    if constexpr (X3) {
      const int frag16 = (lane & 31) * LDH + (lane >> 5) * 8;      // lane (row i, half h): k = 16m + 8h .. +7
#pragma unroll
      for (int m = 0; m < BK / 16; ++m) {
        bf16x8 pa[3][MT], pb[3][NT];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            pa[p][mt] = *reinterpret_cast<const bf16x8*>(&As16[(p * BM + wm * WM + mt * 32) * LDH + frag16 + m * 16]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            pb[p][nt] = *reinterpret_cast<const bf16x8*>(&Bs16[(p * BN + wn * WN + nt * 32) * LDH + frag16 + m * 16]);
        }
        constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (plane of A, plane of B)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[order[t][0]][mt], pb[order[t][1]][nt], acc[mt][nt], 0, 0, 0);
      }
    } else if constexpr (BF16) {
      const int frag16 = (lane & 31) * LDH + (lane >> 5) * 8;      // lane (row i, half h): k = 16m + 8h .. +7
#pragma unroll
      for (int m = 0; m < KB / 16; ++m) {
        bf16x8 pa[MT], pb[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          pa[mt] = *reinterpret_cast<const bf16x8*>(&As16[(wm * WM + mt * 32) * LDH + frag16 + m * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          pb[nt] = *reinterpret_cast<const bf16x8*>(&Bs16[(wn * WN + nt * 32) * LDH + frag16 + m * 16]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            if constexpr (F16)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pa[mt]),
                                                                   __builtin_bit_cast(f16x8, pb[nt]), acc[mt][nt], 0, 0, 0);
            else
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mt], pb[nt], acc[mt][nt], 0, 0, 0);
      }
    } else if constexpr (M16) {
#pragma unroll
      for (int q = 0; q < BK / 16; ++q) {
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          af[mt] = *reinterpret_cast<const f32x4*>(&Ac[(wm * WM + mt * 16) * LDS_LD + frag_off + q * 16]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bf[nt] = *reinterpret_cast<const f32x4*>(&Bc[(wn * WN + nt * 16) * LDS_LD + frag_off + q * 16]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        af[mt] = *reinterpret_cast<const f32x4*>(&Ac[(wm * WM + mt * 32) * LDS_LD + frag_off + q * 8]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[nt] = *reinterpret_cast<const f32x4*>(&Bc[(wn * WN + nt * 32) * LDS_LD + frag_off + q * 8]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
    }
  };
  if (s_begin < s_end) {
    gload(raS[0], rbvS[0], okS[0]);
    lds_store(raS[0], rbvS[0], okS[0]);
    if constexpr (PF2) gload(raS[1], rbvS[1], okS[1]);
    __syncthreads();
    if constexpr (PF2) {
      for (int s = s_begin; s < s_end; s += 2) {
        gload(raS[0], rbvS[0], okS[0]);        // step s+2 (set 1 holds step s+1, still in flight)
        __builtin_amdgcn_sched_barrier(0);
        mfma_step();
        __syncthreads();
        lds_store(raS[1], rbvS[1], okS[1]);
        __syncthreads();
        if (s + 1 >= s_end) break;
        gload(raS[1], rbvS[1], okS[1]);        // step s+3
        __builtin_amdgcn_sched_barrier(0);
        mfma_step();
        __syncthreads();
        lds_store(raS[0], rbvS[0], okS[0]);
        __syncthreads();
      }
    } else {
      for (int s = s_begin; s < s_end; ++s) {
        gload(raS[0], rbvS[0], okS[0]);
        __builtin_amdgcn_sched_barrier(0);   // keep the fetch of step s+1 in front of the MFMAs of step s
        mfma_step();
        __syncthreads();
        lds_store(raS[0], rbvS[0], okS[0]);
        __syncthreads();
      }
    }
  }

  // ---- epilogue ----
  // accumulator element e of tile (mt, nt): 32x32 -> row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31;
  //                                        16x16 -> row 4*(l>>4) + e, column l&15
  const int h = M16 ? (lane >> 4) : (lane >> 5), cl = M16 ? (lane & 15) : (lane & 31);
  float colsum[NT], colsq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) colsum[nt] = colsq[nt] = 0.f;
  const bool bnbwd = g.bn_y != nullptr;
  float bn_m[NT], bn_r[NT], bn_g[NT], bn_b[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + wn * WN + nt * TS + cl;
    const bool bn = bnbwd && g.bn_mean != nullptr;        // (activation-only backward: xhat = u, gamma = 1, beta = 0)
    bn_m[nt] = bn ? g.bn_mean[(size_t)grp * g.N + col] : 0.f;
    bn_r[nt] = bn ? g.bn_rstd[(size_t)grp * g.N + col] : 1.f;
    bn_g[nt] = bn ? g.bn_gamma[col] : 1.f;
    bn_b[nt] = bn ? g.bn_beta[col] : 0.f;
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int r = wm * WM + mt * TS + (M16 ? 4 * h + e : (e & 3) + 8 * (e >> 2) + 4 * h);
      const int ooff = rowinfo[r * 4 + 3];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = n0 + wn * WN + nt * TS + cl;
        float v = acc[mt][nt][e];
        if (bnbwd) {
          float xh = 0.f;
          if (ooff >= 0) {
            const float yv = (BF16 && g.bny_b16) ? ld1<st16_t>(reinterpret_cast<const st16_t*>(g.bn_y) + (size_t)ooff + col)
                                                 : g.bn_y[(size_t)ooff + col];
            xh = (yv - bn_m[nt]) * bn_r[nt];
            v *= act_grad(bn_g[nt] * xh + bn_b[nt], g.bwd_act);
          }
          colsum[nt] += v;
          colsq[nt] += v * xh;
        } else {
          colsum[nt] += v;
          colsq[nt] += v * v;
        }
        if (ooff >= 0) {
          if (g.splitk > 1) {
            const int grow = grp * Mg + tile * BM + r;  // dense row id (DENSE mode only)
            ws[((size_t)split * g.rows_total + grow) * g.N + col] = v;
          } else {
            if (g.has_bias) v += bias[col];
            if (BF16 && g.c_b16) {
              // two adjacent columns live in adjacent lanes: the even lane stores both as one dword (no sub-dword
              // stores; ooff + col is even there, so the address is 4-byte aligned)
              const float vn = __shfl_down(v, 1, 64);
              if (!(cl & 1)) {
                *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C) + (size_t)ooff + col) = pack2<st16_t>(v, vn);
                if (g.want_act_out)
                  *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C_act) + (size_t)ooff + col) =
                      pack2<st16_t>(apply_act(v, g.act), apply_act(vn, g.act));
              }
            } else {
              C[(size_t)ooff + col] = v;
              if (g.want_act_out) {
                if (BF16 && g.cact_b16) {       // fp32 pre-activation, bf16 activated output (FC level -> conv level)
                  const float av = apply_act(v, g.act);
                  const float an = __shfl_down(av, 1, 64);
                  if (!(cl & 1))
                    *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C_act) + (size_t)ooff + col) = pack2<st16_t>(av, an);
                } else {
                  C_act[(size_t)ooff + col] = apply_act(v, g.act);
                }
              }
            }
          }
        }
      }
    }
  }

  if (g.want_stats) {
    // rows beyond the group are zero-filled operands -> contribute exactly 0
    __syncthreads();
    float* red = As;  // [WAVES_M][2][BN]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float s = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
      float q = colsq[nt] + __shfl_xor(colsq[nt], 32, 64);
      if constexpr (M16) {               // a column's rows sit in the four lanes l, l+16, l+32, l+48
        s += __shfl_xor(s, 16, 64);
        q += __shfl_xor(q, 16, 64);
      }
      if (h == 0) {
        red[(wm * 2 + 0) * BN + wn * WN + nt * TS + cl] = s;
        red[(wm * 2 + 1) * BN + wn * WN + nt * TS + cl] = q;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        s += red[(w * 2 + 0) * BN + tid];
        q += red[(w * 2 + 1) * BN + tid];
      }
      int T = g.nclasses * g.tiles_per_group, slot = cls * g.tiles_per_group + tile;
      if (MODE == MMDYN_TCONV_S1P0) {     // one slot per (output pixel, sample chunk)
        T = g.Ho * g.Wo * g.tiles_per_pixel;
        slot = (px_y * g.Wo + px_x) * g.tiles_per_pixel + qtile % g.tiles_per_pixel;
      }
      const size_t base = ((size_t)(grp * T + slot) * 2) * g.N + n0 + tid;
      stats[base] = s;
      stats[base + g.N] = q;
    }
  }
  }  // sub-tile loop
}

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                     float* __restrict__ C, float* __restrict__ C_act, int splitk,
                                     int64_t total, int N, int act) {
  const int64_t nvec = total / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
    for (int k = 1; k < splitk; ++k) {
      f32x4 t = reinterpret_cast<const f32x4*>(ws + (size_t)k * total)[i];
      s += t;
    }
    if (bias) {
      int c = (int)((i * 4) % N);
      s += *reinterpret_cast<const f32x4*>(bias + c);
    }
    reinterpret_cast<f32x4*>(C)[i] = s;
    if (C_act) {
      f32x4 a;
      a[0] = apply_act(s[0], act);
      a[1] = apply_act(s[1], act);
      a[2] = apply_act(s[2], act);
      a[3] = apply_act(s[3], act);
      reinterpret_cast<f32x4*>(C_act)[i] = a;
    }
  }
}

template <int MODE, int BM, int BN, int WM, int WN>
static int launch_m(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats,
                    float* ws, IgemmGeom g, hipStream_t st, bool bf16) {
  g.tiles_per_group = ceil_div(g.Bg * g.Hr * g.Wr, BM);
  if (MODE == MMDYN_TCONV_S1P0) {
    g.tiles_per_pixel = ceil_div(g.Bg, BM);
    // fp32: the balanced quad walk while it gives every CU two blocks (4 x 256 samples: 512 blocks, 295 us against 346 for the
    // pair walk); below that the pair walk's extra blocks win (4 x 128 samples: 199 vs 214 us; 256 samples: 130 vs 188 us) --
    // tests/microbench/ab_s1p0.py, profiles/r3/ab_s1p0_pair_walk_fp32.txt
    const long quad_blocks = (long)g.G * 16 * g.tiles_per_pixel * (g.N / BN);
    g.s1p0_split = (bf16 || quad_blocks < 512) ? 2 : 1;
    if (const char* e = lab_env("MMDYN_S1P0_SPLIT_F32")) {      // LAB: force one walk for fp32
      if (!bf16 && (e[0] == '1' || e[0] == '2')) g.s1p0_split = e[0] - '0';
    }
    g.tiles_per_group = 16 * g.tiles_per_pixel * g.s1p0_split;   // 16 pixel quads per group, 4 pixels walked per block (pair)
  }
  // all-bf16 operands (the 64-channel K-step variant) with >= 2048 or <= 512 one-pixel blocks: ONE output pixel per block (ONEPX at
  // the kernel) -- round 3, tests/microbench/ab_s1p0_b16.py: 1 x 128 samples 40 vs 46 us, 4 x 64 47 vs 57, 1 x 256 40 vs 46; only
  // the 1024-block case (4 x 128) prefers the pair walk (63 vs 67 us).
  // Measured on 4 x 256 samples, 256 -> 128 channels: 79 us against 96 us for the pair walk (4 x 128 samples, 1024 blocks:
  // 65 vs 59 us, stays on the pair walk).  fp32: the balanced quad walk wins (325 vs 376 us).  MMDYN_S1P0_SPLIT=2 / 4 force
  // the pair walk / one pixel per block for the bf16 variants (kernel experiments).
  bool onepx = false;
  if (MODE == MMDYN_TCONV_S1P0 && bf16 && !g.f16 && g.a_b16 && g.b_b16 && g.Cin % 64 == 0) {
    const char* e = lab_env("MMDYN_S1P0_SPLIT");
    const int forced = e ? atoi(e) : 0;
    const long blocks1 = (long)g.G * 64 * g.tiles_per_pixel * (g.N / BN);
    if (forced == 4 || (forced == 0 && (blocks1 >= 2048 || blocks1 <= 512))) {
      onepx = true;
      g.s1p0_split = 4;
      g.tiles_per_group = 64 * g.tiles_per_pixel;
    }
  }
  const int mx_total = g.G * g.tiles_per_group, mx8 = (mx_total + 7) / 8 * 8;
  const int s_inner = (g.N / BN) * g.nclasses;
  dim3 grid((unsigned)mx8 * s_inner * g.splitk);
  if (mx_total <= 4) {          // few M-tiles: the kernel spreads each M-tile's N-tiles over several XCDs (see its block order)
    const int mxp = mx_total > 2 ? 4 : mx_total, nparts = 8 / mxp;
    grid = dim3((unsigned)8 * ((s_inner + nparts - 1) / nparts) * g.splitk);
  }
  // fp32: 16x16x4 MFMA tiles (see the M16 note at the kernel) where the per-shape sweep of the step's launches shows a win
  // (profiles/r2/igemm_tile_x_mfma_shape_sweep.txt): large launches (>= 2048 blocks, >= 8 K-steps) of every block tile but
  // 128x128: +3..6 % there, -2..4 % on the small encoder-side launches, which keep the 32x32x2 shape.
  // MMDYN_IGEMM_M32=1 / MMDYN_IGEMM_M16=1 force one shape everywhere (kernel experiments)
  constexpr bool M16_TILE = true;
  const bool big_tile_m16 = !(BM == 128 && BN == 128) || lab_env("MMDYN_IGEMM_M16_128") != nullptr;
  const bool force_m32 = lab_env("MMDYN_IGEMM_M32") != nullptr, force_m16 = lab_env("MMDYN_IGEMM_M16") != nullptr;
  const long nblocks = (long)g.G * g.tiles_per_group * (g.N / BN) * g.nclasses * g.splitk;
  const int ksteps = (MODE == MMDYN_TCONV_S1P0 ? 6 : g.ntaps) * (g.Cin / BK) / g.splitk;
  const bool m16 = !bf16 && M16_TILE && !force_m32 && (force_m16 || (big_tile_m16 && nblocks >= 2048 && ksteps >= 8) ||
                                                       (BM == 128 && BN == 128 && lab_env("MMDYN_IGEMM_M16_128") != nullptr));
  size_t smem = (size_t)(BM + BN) * (m16 ? BK + 8 : LDS_LD32) * sizeof(float) + (size_t)BM * 4 * sizeof(int);
  if constexpr (MODE != MMDYN_IM2COL3 && BM * BN >= 64 * 64) {
    if (g.x3 && !bf16) {        // fp32 through the bf16 matrix cores (three-term split, see X3 at the kernel)
      smem = (size_t)3 * (BM + BN) * (BK + 8) * 2 + (size_t)BM * 4 * sizeof(int);
      static LdsOptIn optin;
      auto k = igemm_nt_kernel<MODE, BM, BN, WM, WN, false, false, false, false, false, false, false, true>;
      if (int rc = optin.ensure(reinterpret_cast<const void*>(k), smem)) return rc;
      hipLaunchKernelGGL(k, grid, dim3(256), smem, st, A, Bp, bias, C, C_act, stats, ws, g);
      MMDYN_LAUNCH_CHECK();
    }
  }
#define IGEMM_LAUNCH(BF, A16_, B16_)                                                                                     \
  hipLaunchKernelGGL((igemm_nt_kernel<MODE, BM, BN, WM, WN, BF, A16_, B16_>), grid, dim3(256), smem, st, A, Bp, bias, C, \
                     C_act, stats, ws, g)
  constexpr bool CAN_A16 = MODE != MMDYN_IM2COL3;
  constexpr bool S1P0 = MODE == MMDYN_TCONV_S1P0;
  if (S1P0 && onepx)
    hipLaunchKernelGGL((igemm_nt_kernel<MODE, BM, BN, WM, WN, true, CAN_A16, true, CAN_A16, false, false, S1P0>), grid, dim3(256),
                       smem, st, A, Bp, bias, C, C_act, stats, ws, g);
  else if (m16)
    hipLaunchKernelGGL((igemm_nt_kernel<MODE, BM, BN, WM, WN, false, false, false, false, M16_TILE>), grid, dim3(256), smem, st, A,
                       Bp, bias, C, C_act, stats, ws, g);
  else if (!bf16) IGEMM_LAUNCH(false, false, false);
  else if (g.f16) {
#define IGEMM_LAUNCH_F16(A16_, B16_, WIDE_)                                                                             \
  hipLaunchKernelGGL((igemm_nt_kernel<MODE, BM, BN, WM, WN, true, A16_, B16_, WIDE_, false, true>), grid, dim3(256), smem, st, \
                     A, Bp, bias, C, C_act, stats, ws, g)
    if (g.a_b16 && g.b_b16 && CAN_A16 && g.Cin % 64 == 0) IGEMM_LAUNCH_F16(CAN_A16, true, CAN_A16);
    else if (g.a_b16 && g.b_b16) IGEMM_LAUNCH_F16(CAN_A16, true, false);
    else if (g.a_b16) IGEMM_LAUNCH_F16(CAN_A16, false, false);
    else if (g.b_b16) IGEMM_LAUNCH_F16(false, true, false);
    else IGEMM_LAUNCH_F16(false, false, false);
#undef IGEMM_LAUNCH_F16
  }
  else if (g.a_b16 && g.b_b16 && CAN_A16 && g.Cin % 64 == 0)
    hipLaunchKernelGGL((igemm_nt_kernel<MODE, BM, BN, WM, WN, true, CAN_A16, true, CAN_A16>), grid, dim3(256), smem, st, A, Bp,
                       bias, C, C_act, stats, ws, g);
  else if (g.a_b16 && g.b_b16) IGEMM_LAUNCH(true, CAN_A16, true);
  else if (g.a_b16) IGEMM_LAUNCH(true, CAN_A16, false);
  else if (g.b_b16) IGEMM_LAUNCH(true, false, true);
  else IGEMM_LAUNCH(true, false, false);
#undef IGEMM_LAUNCH
  MMDYN_LAUNCH_CHECK();
}

template <int BM, int BN, int WM, int WN>
static int launch(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats,
                  float* ws, IgemmGeom g, hipStream_t st, bool bf16) {
  if (g.mode == MMDYN_DENSE) return launch_m<MMDYN_DENSE, BM, BN, WM, WN>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (g.mode == MMDYN_CONV) return launch_m<MMDYN_CONV, BM, BN, WM, WN>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (g.mode == MMDYN_IM2COL3)
    return launch_m<MMDYN_IM2COL3, BM, BN, WM, WN>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (g.mode == MMDYN_TCONV_S1P0)
    return launch_m<MMDYN_TCONV_S1P0, BM, BN, WM, WN>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  return launch_m<MMDYN_TCONV_S2P1, BM, BN, WM, WN>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
}

}  // namespace

// The template instances of one block tile are compiled as their own translation unit (Makefile: -DIGEMM_PART=k, k = 0..5;
// -DIGEMM_PART=99: the entry points only): six hipcc jobs of ~30 s instead of one of 3 minutes.  Without IGEMM_PART
// everything is built in one unit.
#if defined(IGEMM_PART)
#define IGEMM_HAS(k) (IGEMM_PART == (k))
#else
#define IGEMM_HAS(k) 1
#endif
#define IGEMM_TILE_ARGS const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws, \
                        IgemmGeom g, hipStream_t st, bool bf16
int mmdyn_igemm_tile0(IGEMM_TILE_ARGS);
int mmdyn_igemm_tile1(IGEMM_TILE_ARGS);
int mmdyn_igemm_tile2(IGEMM_TILE_ARGS);
int mmdyn_igemm_tile3(IGEMM_TILE_ARGS);
int mmdyn_igemm_tile4(IGEMM_TILE_ARGS);
int mmdyn_igemm_tile5(IGEMM_TILE_ARGS);
#if IGEMM_HAS(0)
int mmdyn_igemm_tile0(IGEMM_TILE_ARGS) { return launch<128, 128, 64, 64>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16); }
#endif
#if IGEMM_HAS(1)
int mmdyn_igemm_tile1(IGEMM_TILE_ARGS) { return launch<64, 128, 32, 64>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16); }
#endif
#if IGEMM_HAS(2)
int mmdyn_igemm_tile2(IGEMM_TILE_ARGS) { return launch<128, 64, 64, 32>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16); }
#endif
#if IGEMM_HAS(3)
int mmdyn_igemm_tile3(IGEMM_TILE_ARGS) { return launch<64, 64, 32, 32>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16); }
#endif
#if IGEMM_HAS(4)
int mmdyn_igemm_tile4(IGEMM_TILE_ARGS) { return launch<256, 32, 64, 32>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16); }
#endif
#if IGEMM_HAS(5)
int mmdyn_igemm_tile5(IGEMM_TILE_ARGS) { return launch<128, 32, 32, 32>(A, Bp, bias, C, C_act, stats, ws, g, st, bf16); }
#endif

#if !defined(IGEMM_PART) || IGEMM_PART == 99


// LAB build: MMDYN_IGEMM_WS=0 sends every launch back to the register-staged kernels (A/B measurements, kernel tests)

// fp32 launches that take the three-term split (X3 at the kernel) and their block tile.  `allowed`: the caller asked for it (flag
// bit 7 of the entry points: the "fp32x3" mode of the host side).  LAB: MMDYN_X3=1 / 0 overrides the flag, MMDYN_X3_TILE=BM,BN
// forces one tile, MMDYN_X3_MIN_BLOCKS moves the size threshold.
static bool ws_enabled() {
  const char* e = lab_env("MMDYN_IGEMM_WS");
  return !(e && e[0] == '0');
}
static bool x3_on(bool allowed) {
  if (const char* on = lab_env("MMDYN_X3")) allowed = on[0] == '1';
  return allowed;
}
// the launches the persistent ring kernel serves keep it in the split arithmetic (its MFMA waves split their fragments in registers;
// LAB: MMDYN_X3_WSP=0 sends them to the register-staged split kernels below instead)
static bool x3_wsp(bool allowed) {
  if (!x3_on(allowed) || !ws_enabled()) return false;
  const char* e = lab_env("MMDYN_X3_WSP");
  return !(e && e[0] == '0');
}
// (LAB experiment, MMDYN_X3_WS=1: every launch the one-tile ring kernels serve runs them in the split arithmetic -- ahead of the
//  register-staged split kernels)
static bool x3_ws(bool allowed) {
  if (!x3_on(allowed) || !ws_enabled()) return false;
  const char* e = lab_env("MMDYN_X3_WS");
  return e && e[0] == '1';
}
static bool x3_pick(bool allowed, int mode, int G, int rows_per_group, int N, int splitk, int* bm, int* bn) {
  if (!x3_on(allowed)) return false;
  if (mode == MMDYN_IM2COL3 || splitk != 1) return false;
  if (N == 32) {           // (LAB experiment, MMDYN_X3_N32=256|128: the 32-channel layers on 256x32 / 128x32 tiles)
    const char* e = lab_env("MMDYN_X3_N32");
    if (!e || (atoi(e) != 256 && atoi(e) != 128)) return false;
    *bm = atoi(e);
    *bn = 32;
    return true;
  }
  if (N % 64) return false;
  if (const char* e = lab_env("MMDYN_X3_MODES")) {        // (LAB: bit m set = launches of mode m may take the split)
    if (!((atoi(e) >> mode) & 1)) return false;
  }
  if (const char* e = lab_env("MMDYN_X3_ONLY_G")) {       // (LAB bisecting knobs: only launches with this G / this N)
    if (atoi(e) != G) return false;
  }
  if (const char* e = lab_env("MMDYN_X3_ONLY_N")) {
    if (atoi(e) != N) return false;
  }
  const int ncls = mode == MMDYN_TCONV_S2P1 ? 4 : 1;
  // measured per shape alone on the chip (tests/microbench/ab_x3.py, profiles/r4/ab_x3_*.txt): 128x128 tiles (four waves of
  // 64x64) where they give >= 384 blocks, the k4 s1 p0 layer and the other launches on 64x64 tiles from 512 blocks on
  const long b128 = (long)G * ceil_div(rows_per_group, 128) * (N / 128) * ncls;
  const long b64 = (long)G * ceil_div(rows_per_group, 64) * (N / 64) * ncls;
  long min_blocks = 512;
  if (const char* e = lab_env("MMDYN_X3_MIN_BLOCKS")) min_blocks = atol(e);
  if (mode != MMDYN_TCONV_S1P0 && N % 128 == 0 && b128 >= 384) {
    *bm = 128;
    *bn = 128;
  } else if (mode == MMDYN_TCONV_S2P1 && N == 64 && b64 >= 2 * min_blocks) {
    *bm = 128;
    *bn = 64;
  } else if (b64 >= min_blocks) {
    *bm = 64;
    *bn = 64;
  } else {
    return false;
  }
  if (const char* e = lab_env("MMDYN_X3_TILE")) {
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 64 || a == 128) && (b == 64 || b == 128) && N % b == 0) {
      *bm = a;
      *bn = b;
    }
  }
  return true;
}
static int x3_stat_tiles(bool allowed, int mode, int G, int Bg, int Hi, int Wi, int Ho, int Wo, int N) {
  int bm, bn;
  if (mode == MMDYN_TCONV_S1P0) return x3_pick(allowed, mode, G, Bg * Ho * Wo, N, 1, &bm, &bn) ? Ho * Wo * ceil_div(Bg, bm) : 0;
  const int Hr = mode == MMDYN_TCONV_S2P1 ? Hi : Ho, Wr = mode == MMDYN_TCONV_S2P1 ? Wi : Wo, ncls = mode == MMDYN_TCONV_S2P1 ? 4 : 1;
  return x3_pick(allowed, mode, G, Bg * Hr * Wr, N, 1, &bm, &bn) ? ncls * ceil_div(Bg * Hr * Wr, bm) : 0;
}

static int lds_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  if (mode == MMDYN_TCONV_S1P0) {
    int bm, bn;
    pick_tile(N, Bg, G * 16, 1, 1, 0, &bm, &bn);
    return Ho * Wo * ceil_div(Bg, bm);
  }
  int Hr = Ho, Wr = Wo, ncls = 1, ntaps = (mode == MMDYN_CONV) ? 16 : 1;
  if (mode == MMDYN_TCONV_S2P1) {
    Hr = Hi;
    Wr = Wi;
    ncls = 4;
    ntaps = 4;
  }
  const int ksteps = ntaps * (Cin / BK);
  int bm, bn;
  pick_tile(N, Bg * Hr * Wr, G, ncls, 1, ksteps, &bm, &bn);
  return ncls * ceil_div(Bg * Hr * Wr, bm);
}

// fp32 launches go to the wave-independent kernels of igemm_d16.hip where those serve the shape; the bf16 matrix-core
// modes always take the LDS-tiled kernels of this file.  The number of partial-sum tiles follows the kernel.
static int f32_stat_tiles(bool x3, int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  int t = x3_wsp(x3) ? mmdyn_igemm_wsp_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, false) : 0;   // (same order as igemm_entry)
  if (t > 0) return t;
  t = (x3_ws(x3) && mode != MMDYN_TCONV_S1P0) ? mmdyn_igemm_ws_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) : 0;
  if (t > 0 && mmdyn_tconv_patch_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) == 0) return t;
  t = x3_stat_tiles(x3, mode, G, Bg, Hi, Wi, Ho, Wo, N);
  if (t > 0) return t;
  const int tp = mmdyn_tconv_patch_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  if (tp > 0) return tp;
  t = mmdyn_igemm_d16_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  if (t > 0) return t;
  t = ws_enabled() ? mmdyn_igemm_wsp_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, false) : 0;
  if (t > 0) return t;
  t = ws_enabled() ? mmdyn_igemm_ws_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) : 0;
  return t > 0 ? t : lds_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
}
extern "C" int mmdyn_igemm_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  return f32_stat_tiles(false, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
}
/* Floats of workspace the fp32 launch of this shape wants in `ws` (with splitk == 1): the slabs of the persistent kernel's
 * split tiles (igemm_wsp.hip).  0: none. */
static int f32_slab_floats(bool x3, int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  if (!ws_enabled()) return 0;
  if (x3_wsp(x3) && mmdyn_igemm_wsp_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, false) > 0)
    return (int)(mmdyn_igemm_wsp_slab_bytes(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, false) / 4);
  if (x3_ws(x3) && mode != MMDYN_TCONV_S1P0 && mmdyn_tconv_patch_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) == 0 &&
      mmdyn_igemm_ws_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) > 0)
    return 0;
  if (x3_stat_tiles(x3, mode, G, Bg, Hi, Wi, Ho, Wo, N) > 0) return 0;
  if (mmdyn_tconv_patch_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) > 0) return 0;
  if (mmdyn_igemm_d16_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) > 0) return 0;
  return (int)(mmdyn_igemm_wsp_slab_bytes(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, false) / 4);
}
extern "C" int mmdyn_igemm_slab_floats(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  return f32_slab_floats(false, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
}
/* 1: the fp32x3 launch of this shape can take its operands ALREADY SPLIT (flag bits 7 + 8 of mmdyn_igemm_nt_mx); its
 * partial-sum tile count and slab workspace are those of the flags == 128 queries.  0: keep fp32 operands. */
extern "C" int mmdyn_igemm_planes_served(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  if (G <= 0 || Bg <= 0 || Cin <= 0 || N <= 0 || Cin % BK || N % 32) return 0;
  if (mmdyn_tconv_patch_p3_serves(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)) return 1;      // the patch-resident 32-channel up-sampling layers
  if (!ws_enabled()) return 0;
  return mmdyn_igemm_wsp3_serves(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) ? 1 : 0;
}
extern "C" int mmdyn_igemm_stat_tiles_bf16(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  return lds_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
}
/* ... of the mixed-storage entry points, flags as mmdyn_igemm_nt_mx: launches whose operands are BOTH 16-bit in HBM (bits 1 and
 * 4) may run the persistent ring kernel (igemm_wsp.hip), which writes one partial tile per wave row */
extern "C" int mmdyn_igemm_stat_tiles_mx(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int flags) {
  if ((flags & 384) == 384) {     // plane launch
    if (mmdyn_tconv_patch_p3_serves(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)) return mmdyn_tconv_patch_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
    return ws_enabled() ? mmdyn_igemm_wsp3_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) : 0;
  }
  if (!(flags & 1) && !(flags & 32)) return f32_stat_tiles((flags & 128) != 0, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  if ((flags & 2) && (flags & 16) && ws_enabled()) {
    const int t = mmdyn_igemm_wsp_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, true);
    if (t > 0) return t;
  }
  return lds_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
}
extern "C" int mmdyn_igemm_slab_floats_mx(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int flags) {
  if ((flags & 384) == 384) {
    if (mmdyn_tconv_patch_p3_serves(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N)) return 0;
    return ws_enabled() ? (int)(mmdyn_igemm_wsp3_slab_bytes(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) / 4) : 0;
  }
  if (!(flags & 1) && !(flags & 32)) return f32_slab_floats((flags & 128) != 0, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  if (!((flags & 2) && (flags & 16)) || !ws_enabled()) return 0;
  return (int)(mmdyn_igemm_wsp_slab_bytes(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, true) / 4);
}

static int igemm_entry(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                       float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                       int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                       void* stream, bool bf16, const float* bn_y = nullptr, const float* bn_mean = nullptr,
                       const float* bn_rstd = nullptr, const float* bn_gamma = nullptr,
                       const float* bn_beta = nullptr, int storage_flags = 0, int b_group_stride = 0,
                       int bias_group_stride = 0, uint32_t* arrival_flags = nullptr) {
  if (!A || !Bp || !C) return MMDYN_ERR_NULL;
  if (Cin <= 0 || N <= 0 || Cin % BK || N % 32 || G <= 0 || Bg <= 0 || ldc < N) return MMDYN_ERR_SHAPE;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && (mode != MMDYN_DENSE || !ws || stats)) return MMDYN_ERR_SHAPE;
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
  g.ldc = ldc;
  g.act = act;
  g.has_bias = bias != nullptr;
  g.want_stats = stats != nullptr;
  g.bn_y = bn_y;
  g.bn_mean = bn_mean;
  g.bn_rstd = bn_rstd;
  g.bn_gamma = bn_gamma;
  g.bn_beta = bn_beta;
  g.bwd_act = bn_mean ? MMDYN_ACT_SWISH : act;       // activation-only backward epilogue: `act` names the activation
  if (bn_y && !bn_mean) {
    if (stats || C_act || bias || splitk > 1 || mode == MMDYN_IM2COL3 || ldc != N) return MMDYN_ERR_SHAPE;
    g.act = MMDYN_ACT_NONE;
  }
  const bool x3_allowed = (storage_flags & 128) != 0;      // fp32 launch that may take the three-term split
  // bit 8 (with bit 7): A and Bp ARRIVE split -- rows of [plane][Cin] bf16, written by mmdyn_split_planes / the plane kinds of the
  // pack plan / a producing kernel -- and the launch runs igemm_wsp3_kernel: no split anywhere in the GEMM
  const bool planes = (storage_flags & 256) != 0;
  // bit 10 (with bits 7 + 8): C_act is written as a plane tensor (the operand of the next plane launch), C stays fp32
  // (bits 16-27: the plane rows' channel count / 8 when it is not N -- it must divide N: an output row is then N / count plane rows)
  const bool cact_planes = (storage_flags & 1024) != 0;
  const int cact_c = ((storage_flags >> 16) & 0xfff) * 8;
  if (cact_planes && (!planes || !C_act || ldc != N || splitk > 1 || mode != MMDYN_DENSE || (cact_c && (N % cact_c || cact_c % 8))))
    return MMDYN_ERR_SHAPE;
  g.cact_planes = cact_planes ? (cact_c ? cact_c : N) : 0;
  storage_flags &= ~(128 | 256 | 1024 | (0xfff << 16));
  if (x3_allowed && (bf16 || storage_flags)) return MMDYN_ERR_SHAPE;
  if (planes && !x3_allowed) return MMDYN_ERR_SHAPE;
  g.a_b16 = (storage_flags & 2) != 0;
  g.c_b16 = (storage_flags & 4) != 0;
  g.bny_b16 = (storage_flags & 8) != 0;
  g.b_b16 = (storage_flags & 16) != 0;
  g.f16 = (storage_flags & 32) != 0;
  g.cact_b16 = (storage_flags & 64) != 0;
  // (f16 alone: fp16 operands on fp32 storage; f16 with storage bits: those 16-bit tensors are IEEE half -- "fp16s")
  if (g.cact_b16 && (g.c_b16 || !C_act || splitk > 1 || (N & 1))) return MMDYN_ERR_SHAPE;
  if (storage_flags && (!bf16 || (g.c_b16 && splitk > 1) || (g.a_b16 && mode == MMDYN_IM2COL3))) return MMDYN_ERR_SHAPE;
  g.want_act_out = C_act != nullptr;
  g.b_group_stride = b_group_stride;
  g.bias_group_stride = bias_group_stride;
  g.flags = arrival_flags;
  g.splitk = splitk;
  g.nclasses = 1;
  g.os = 1;
  if (mode == MMDYN_DENSE) {
    if (Hi != Ho || Wi != Wo) return MMDYN_ERR_SHAPE;
    g.Hr = Ho;
    g.Wr = Wo;
    g.rs = 1;
    g.ro = 0;
    g.ntaps = 1;
  } else if (mode == MMDYN_CONV) {
    g.Hr = Ho;
    g.Wr = Wo;
    g.rs = stride;
    g.ro = offset;
    g.ntaps = 16;
  } else if (mode == MMDYN_IM2COL3) {
    if (Hi != 2 * Ho || Wi != 2 * Wo || Cin != 64 || splitk != 1) return MMDYN_ERR_SHAPE;
    g.Hr = Ho;
    g.Wr = Wo;
    g.rs = 1;
    g.ro = 0;
    g.ntaps = 1;
  } else if (mode == MMDYN_TCONV_S1P0) {
    if (Ho != Hi + 3 || Wo != Wi + 3 || Ho != 8 || Wo != 8 || splitk != 1) return MMDYN_ERR_SHAPE;
    g.Hr = Ho;
    g.Wr = Wo;
    g.rs = 1;
    g.ro = 0;
    g.ntaps = 16;
  } else if (mode == MMDYN_TCONV_S2P1) {
    if (Ho != 2 * Hi || Wo != 2 * Wi) return MMDYN_ERR_SHAPE;
    g.Hr = Hi;
    g.Wr = Wi;
    g.rs = 1;
    g.ro = 0;
    g.os = 2;
    g.ntaps = 4;
    g.nclasses = 4;
  } else {
    return MMDYN_ERR_SHAPE;
  }
  const int64_t rows = (int64_t)G * Bg * g.Hr * g.Wr;
  if ((int64_t)G * Bg * Hi * Wi * (mode == MMDYN_IM2COL3 ? 3 : Cin) >= (1LL << 31) ||
      (int64_t)G * Bg * Ho * Wo * ldc >= (1LL << 31))
    return MMDYN_ERR_RANGE;
  g.rows_total = (int)rows;
  hipStream_t st = (hipStream_t)stream;
  if (planes && mode == MMDYN_IM2COL3) return MMDYN_ERR_SHAPE;   // (the 3-channel layers read the NCHW image: no plane form)
  if (mode == MMDYN_IM2COL3) {     // the 3-channel layers have their own direct kernel (conv3.hip)
    const int rc = mmdyn_conv3_nt_try(A, Bp, bias, C, C_act, stats, G, Bg, Hi, Wi, Ho, Wo, N, ldc, act, splitk, bn_y,
                                      bn_mean, bn_rstd, bn_gamma, bn_beta, g.c_b16 << g.f16, g.bny_b16 << g.f16, g.b_b16 << g.f16,
                                      st);
    if (rc != 1) return rc;
  }
  if (planes) {       // operands already split: served by a plane kernel or not at all (mmdyn_igemm_planes_served)
    if (mode == MMDYN_TCONV_S2P1 && !b_group_stride) {
      const int rp = mmdyn_tconv_patch_p3_try(A, Bp, bias, C, C_act, stats, ws, g, st);
      if (rp != 1) return rp;
    }
    const int rc = ws_enabled() ? mmdyn_igemm_wsp3_try(A, Bp, bias, C, C_act, stats, ws, g, st) : 1;
    return rc == 1 ? MMDYN_ERR_SHAPE : rc;
  }
  if (!bf16 && mode != MMDYN_IM2COL3 && x3_wsp(x3_allowed)) {     // the persistent ring kernel in the split arithmetic
    IgemmGeom gx = g;
    gx.x3 = 1;
    const int rc = mmdyn_igemm_wsp_try(A, Bp, bias, C, C_act, stats, ws, gx, false, st);
    if (rc != 1) return rc;
  }
  if (!bf16 && mode != MMDYN_IM2COL3 && mode != MMDYN_TCONV_S1P0 && x3_ws(x3_allowed) &&
      mmdyn_tconv_patch_stat_tiles(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N) == 0) {      // (LAB experiment: see x3_ws)
    IgemmGeom gx = g;
    gx.x3 = 1;
    const int rc = mmdyn_igemm_ws_try(A, Bp, bias, C, C_act, stats, ws, gx, false, st);
    if (rc != 1) return rc;
  }
  if (!bf16) {        // fp32 through the bf16 matrix cores (three-term operand split)
    int bm, bn;
    if (x3_pick(x3_allowed, mode, G, mode == MMDYN_TCONV_S1P0 ? Bg * Ho * Wo : Bg * g.Hr * g.Wr, N, splitk, &bm, &bn)) {
      g.x3 = 1;
      if (bn == 128 && bm == 128) return mmdyn_igemm_tile0(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
      if (bn == 128 && bm == 64) return mmdyn_igemm_tile1(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
      if (bn == 64 && bm == 128) return mmdyn_igemm_tile2(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
      if (bn == 32 && bm == 256) return mmdyn_igemm_tile4(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
      if (bn == 32) return mmdyn_igemm_tile5(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
      return mmdyn_igemm_tile3(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
    }
  }
  if (!bf16 && mode == MMDYN_TCONV_S2P1) {     // the 64 -> 32 channel up-sampling layer has a patch-resident kernel (tconv_patch.hip)
    const int rc = mmdyn_tconv_patch_try(A, Bp, bias, C, C_act, stats, ws, g, st);
    if (rc != 1) return rc;
  }
  if (!bf16 && mode != MMDYN_IM2COL3 && !b_group_stride) {
    const int rc = mmdyn_igemm_d16_try(A, Bp, bias, C, C_act, stats, ws, g, stride, offset, st);
    if (rc != 1) return rc;
  }
  if (ws_enabled() && mode != MMDYN_IM2COL3 && (!bf16 || (g.a_b16 && g.b_b16))) {   // persistent ring kernel (igemm_wsp.hip)
    const int rc = mmdyn_igemm_wsp_try(A, Bp, bias, C, C_act, stats, ws, g, bf16, st);
    if (rc != 1) return rc;
  }
  if (ws_enabled() && mode != MMDYN_IM2COL3 && mode != MMDYN_TCONV_S1P0) {   // wave-specialised LDS-DMA ring kernels (igemm_ws.hip)
    const int rc = mmdyn_igemm_ws_try(A, Bp, bias, C, C_act, stats, ws, g, bf16, st);
    if (rc != 1) return rc;
  }
  int bm, bn;
  if (mode == MMDYN_TCONV_S1P0)
    pick_tile(N, Bg, G * 16, 1, 1, 0, &bm, &bn);
  else
    pick_tile(N, Bg * g.Hr * g.Wr, G, g.nclasses, splitk, g.ntaps * (Cin / BK), &bm, &bn);
  if (bn == 128 && bm == 128) return mmdyn_igemm_tile0(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (bn == 128 && bm == 64) return mmdyn_igemm_tile1(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (bn == 64 && bm == 128) return mmdyn_igemm_tile2(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (bn == 64 && bm == 64) return mmdyn_igemm_tile3(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  if (bn == 32 && bm == 256) return mmdyn_igemm_tile4(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
  return mmdyn_igemm_tile5(A, Bp, bias, C, C_act, stats, ws, g, st, bf16);
}

extern "C" int mmdyn_igemm_nt(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                              float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                              int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                              void* stream) {
  return igemm_entry(A, Bp, bias, C, C_act, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act,
                     splitk, stream, false);
}

extern "C" int mmdyn_igemm_nt_dgrad_bn(const float* A, const float* Bp, float* C, float* stats, const float* y,
                                       const float* mean, const float* rstd, const float* gamma, const float* beta,
                                       int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N,
                                       int stride, int offset, int bf16, float* ws, uint32_t* flags, void* stream) {
  if (!stats || !y || !mean || !rstd || !gamma || !beta) return MMDYN_ERR_NULL;
  if (bf16 < 0 || bf16 > 4) return MMDYN_ERR_SHAPE;        // (3: fp32 arithmetic, the three-term split allowed; 4: A and Bp arrive split)
  return igemm_entry(A, Bp, nullptr, C, nullptr, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, N, stride, offset,
                     MMDYN_ACT_NONE, 1, stream, bf16 == 1 || bf16 == 2, y, mean, rstd, gamma, beta,
                     bf16 == 2 ? 32 : bf16 == 3 ? 128 : bf16 == 4 ? 384 : 0, 0, 0, flags);
}

/* Input-gradient GEMM with the backward of a plain ACTIVATION in its epilogue: C = (A x Bp) * act'(u), u the layer's saved
 * pre-activation, same rows / columns as C.  flags as mmdyn_igemm_nt_mx (bit 3: u is bf16); 0 = fp32 everywhere. */
extern "C" int mmdyn_igemm_nt_dgrad_act(const void* A, const void* Bp, void* C, const void* u, int act, int mode, int G, int Bg,
                                        int Hi, int Wi, int Cin, int Ho, int Wo, int N, int stride, int offset, int flags,
                                        float* ws, uint32_t* arrival_flags, void* stream) {
  if (!u) return MMDYN_ERR_NULL;
  if (act != MMDYN_ACT_SWISH && act != MMDYN_ACT_RELU) return MMDYN_ERR_SHAPE;
  return igemm_entry((const float*)A, (const float*)Bp, nullptr, (float*)C, nullptr, nullptr, ws, mode, G, Bg, Hi, Wi, Cin,
                     Ho, Wo, N, N, stride, offset, act, 1, stream, (flags & 1) != 0 || (flags & 32) != 0, (const float*)u, nullptr,
                     nullptr, nullptr, nullptr, flags & ~1, 0, 0, arrival_flags);
}

/* fp16 matrix cores (v_mfma_f32_32x32x16_f16): operands rounded to IEEE half (RNE) on their way into the MFMA, fp32
 * accumulate, everything in HBM fp32 -- the arithmetic BASELINE configs[4] names */
extern "C" int mmdyn_igemm_nt_f16(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                                  float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                                  int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                                  void* stream) {
  return igemm_entry(A, Bp, bias, C, C_act, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act,
                     splitk, stream, true, nullptr, nullptr, nullptr, nullptr, nullptr, 32);
}

/* One entry point for the mixed-storage variants: flags bit 0 = bf16 matrix cores (required for the others),
 * bit 1 = A is bf16 in HBM, bit 2 = C / C_act are bf16, bit 3 = the BatchNorm-backward operand y is bf16. */
extern "C" int mmdyn_igemm_nt_mx(const void* A, const void* Bp, const float* bias, void* C, void* C_act, float* stats,
                                 float* ws, const void* bn_y, const float* bn_mean, const float* bn_rstd,
                                 const float* bn_gamma, const float* bn_beta, int mode, int G, int Bg, int Hi, int Wi,
                                 int Cin, int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                                 int flags, uint32_t* arrival_flags, void* stream) {
  if (bn_y && (!stats || !bn_mean || !bn_rstd || !bn_gamma || !bn_beta)) return MMDYN_ERR_NULL;
  return igemm_entry((const float*)A, (const float*)Bp, bias, (float*)C, (float*)C_act, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho,
                     Wo, N, ldc, stride, offset, act, splitk, stream, (flags & 1) != 0, (const float*)bn_y, bn_mean,
                     bn_rstd, bn_gamma, bn_beta, flags & ~1, 0, 0, arrival_flags);
}

/* Grouped dense GEMM: G independent problems of ONE shape in one launch -- C_g[rows][N] = A_g[rows][K] . Bp_g[N][K]^T (+ bias_g),
 * group g at A + g*rows*K, Bp + g*N*K, bias + g*N, C (and C_act, u) + g*rows*N.  The heads / pose GEMMs at the product-of-experts
 * join of the fused step (vae.py:211-216, 239-240: linear_means | linear_log_var of the visual, tactile and pose encoders) are
 * three such problems of 1024 x 512 x 512 that fill half the chip each when launched one by one.
 * u != NULL: the activation-backward epilogue C = (A . Bp^T) * act'(u) (then bias / C_act must be NULL).
 * flags as mmdyn_igemm_nt_mx (0 = fp32 everywhere). */
extern "C" int mmdyn_igemm_nt_grouped(const void* A, const void* Bp, const float* bias, void* C, void* C_act, const void* u,
                                      int G, int rows, int K, int N, int act, int flags, void* stream) {
  if (G < 1 || rows < 1) return MMDYN_ERR_SHAPE;
  if (u && (bias || C_act || (act != MMDYN_ACT_SWISH && act != MMDYN_ACT_RELU))) return MMDYN_ERR_SHAPE;
  if ((int64_t)G * N * K >= (1LL << 31)) return MMDYN_ERR_RANGE;
  return igemm_entry((const float*)A, (const float*)Bp, bias, (float*)C, (float*)C_act, nullptr, nullptr, MMDYN_DENSE, G, rows, 1, 1,
                     K, 1, 1, N, N, 1, 0, act, 1, stream, (flags & 1) != 0 || (flags & 32) != 0, (const float*)u, nullptr, nullptr,
                     nullptr, nullptr, flags & ~1, N * K, N);
}

extern "C" int mmdyn_igemm_nt_bf16(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                                   float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                                   int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                                   void* stream) {
  return igemm_entry(A, Bp, bias, C, C_act, stats, ws, mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N, ldc, stride, offset, act,
                     splitk, stream, true);
}

extern "C" int mmdyn_splitk_reduce(const float* ws, const float* bias, float* C, float* C_act, int splitk,
                                   int rows, int N, int act, void* stream) {
  if (!ws || !C) return MMDYN_ERR_NULL;
  if (N % 4) return MMDYN_ERR_SHAPE;
  int64_t total = (int64_t)rows * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, ws,
                     bias, C, C_act, splitk, total, N, act);
  MMDYN_LAUNCH_CHECK();
}
#endif   // entry points
