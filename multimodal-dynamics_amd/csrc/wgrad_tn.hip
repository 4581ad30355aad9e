// MFMA weight-gradient GEMM, TN form (reduction over rows), fp32 (v_mfma_f32_32x32x2_f32, gfx950).
//
//   partial[chunk][tap][cd][cg] = sum_{row in chunk} D[row][cd] * G_tap[row][cg]
//
// Replaces the ATen weight-gradient kernels of nn.Conv2d / nn.ConvTranspose2d / nn.Linear used by
// loss.backward() on the reference path (/root/reference/mmdyn/pytorch/problems/problems.py:153).
//
// D is the "dense" operand (one contiguous row per GEMM row), G the "gathered" one (for the 16 taps of
// a k=4 convolution the row's pixel is shifted by (kh, kw); out-of-image rows read as zero).  For a
// Conv2d, D = dY and G = X; for a ConvTranspose2d, D = X and G = dY -- in both cases the reference's
// canonical weight layout is [cd][cg][kh][kw], which mmdyn_wgrad_reduce produces.
//
// Both tiles sit in LDS row-major ([32 rows][channels]); the MFMA A operand A[i][k] is read as
// Ds[k][cd0+i] and the B operand B[k][j] as Gs[k][cg0+j]: consecutive lanes read consecutive floats,
// conflict-free ds_read_b32.  The row reduction is split over `chunks` blocks (deterministic partial
// slabs, no atomics); a second kernel sums the slabs and scatters into the canonical layout.
#include <stdlib.h>
#include "common.h"
#include "wgrad_geom.h"

namespace {


constexpr int RK = 32;  // rows per K-step

// WK = number of waves that split the rows of one K-step between them (small channel tiles cannot
// occupy four waves with distinct 32x32 outputs); each of them owns its own partial slab.
// BF16 = true: fragments are rounded to bf16 (RNE) as they leave LDS and multiplied 8 rows at a time with
// v_mfma_f32_32x32x8_bf16 (fp32 accumulate); the staging and the partial-slab scheme are unchanged.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 pack4_f16(float a, float b, float c, float d) {
  f16x4 r;
  r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d;
  return r;
}
__device__ __forceinline__ s16x4 pack4_bf16(float a, float b, float c, float d) {
  bf16x4 r;
  r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;
  return __builtin_bit_cast(s16x4, r);
}

// ST: operand storage, always a compile-time constant (a run-time test inside the fetch puts every load in its own
// branch and the loads then wait for one another): 0 = D and Gt fp32, 1 = both bf16, 2 = D bf16 / Gt fp32,
// 3 = D fp32 / Gt bf16, 4 = both fp32 in HBM, fragments rounded to fp16 (v_mfma_f32_32x32x8_f16: BASELINE configs[4])
// 5 / 6 / 7 = as 1 / 2 / 3 with IEEE half in place of bf16, in HBM and on the matrix cores (precision "fp16s")
// MODE (DENSE / CONV / IM2COL3) is a template parameter: with a run-time mode test inside the fetch each gathered load
// sat in its own branch, and because the two sides of the branch write the same registers the compiler put
// s_waitcnt vmcnt(0) in front of every one of them -- five serial memory round trips per K-step instead of one
// (found in the ISA; the comment "branch-free fetch" below was only true of the source).
// X3 (fp32 operands in HBM, ST = 0): the product runs on the bf16 matrix cores through the exact three-term split of every fp32
// operand (x = hi + mid + lo, all bf16: split3_bf16 in common.h; six of the nine cross products, fp32 accumulate -- see X3 at
// igemm_nt_kernel).  The split
// is done once per element as the tile is written to LDS (three bf16 planes per operand); the k-strided fragments (the reduction
// index is the tile ROW) come from the transposing read ds_read_b64_tr_b16, as in wgrad_b16_kernel below.
template <int BX> struct x3_ld { static constexpr int v = (BX == 32) ? 32 : BX + 32; };   // plane row stride in elements
// PRE (X3 only; bit 0: D, bit 1: Gt): that operand ARRIVES split -- rows of [plane][C] bf16 written by its producer (the exact
// three-term split, mmdyn_split_planes) -- and goes from HBM to the LDS planes as it is: three 8-byte loads per four channels, no
// VALU work ("split once, not once per tile that reads the operand").
template <int MODE, int BD, int BG, int WD, int WG, int WK, bool BF16, int ST, bool X3 = false, int PRE = 0>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(const float* __restrict__ D,
                                                       const float* __restrict__ Gt,
                                                       float* __restrict__ partial, const WgradGeom g) {
  static_assert(!X3 || (!BF16 && ST == 0 && WK == 1 && MODE != MMDYN_IM2COL3), "the three-term split is a variant of the fp32 kernel");
  static_assert(PRE == 0 || X3, "operands that arrive split belong to the three-term split arithmetic");
  constexpr bool PD = (PRE & 1) != 0, PG = (PRE & 2) != 0;
  constexpr int DT = WD / 32, GT = WG / 32;
  constexpr int WAVES_G = BG / WG;
  constexpr int WAVES_DG = (BD / WD) * WAVES_G;
  static_assert(WAVES_DG * WK == 4, "4 waves per block");
  constexpr int KPW = (RK / 2) / WK;
  constexpr int DV = BD / 4, GV = BG / 4;            // float4 per tile row
  constexpr int D_LOADS = (RK * DV) / 256, G_LOADS = (RK * GV) / 256;
  static_assert(D_LOADS >= 1 && G_LOADS >= 1, "tile too small for 256 threads");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ds = reinterpret_cast<float*>(smem);  // [RK][BD]
  float* Gs = Ds + RK * BD;                    // [RK][BG]
  constexpr int LDD = x3_ld<BD>::v, LDG = x3_ld<BG>::v;
  bf16_t* Ds16 = reinterpret_cast<bf16_t*>(smem);     // X3: [plane][RK][LDD], then [plane][RK][LDG]
  bf16_t* Gs16 = Ds16 + 3 * RK * LDD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WAVES_DG, wdg = wave % WAVES_DG;
  const int wd = wdg / WAVES_G, wg = wdg % WAVES_G;
  const int tiles_g = g.Cg / BG;
  const int td = blockIdx.x / tiles_g, tg = blockIdx.x - td * tiles_g;
  const int cd0 = td * BD, cg0 = tg * BG;
  const int tap = blockIdx.y;
  // grouped launch: blockIdx.z = group * zblocks + chunk (plain launch: one group)
  const int ngrp = g.groups > 1 ? g.groups : 1;
  const int zb = gridDim.z / ngrp;
  const int grp = blockIdx.z / zb, chunk = blockIdx.z - grp * zb;
  const int dh = (MODE == MMDYN_CONV) ? (tap >> 2) : 0;
  const int dw = (MODE == MMDYN_CONV) ? (tap & 3) : 0;
  const int HWr = g.Hr * g.Wr;

  const int row_begin = grp * g.rows + chunk * g.rows_per_chunk;
  const int row_end = min((grp + 1) * g.rows, row_begin + g.rows_per_chunk);

  // Branch-free fetch (same idea as igemm_nt): rows past the chunk or outside the image read a valid dummy
  // address and are zeroed when the tile is written to LDS; pixel decode uses a float reciprocal (rows < 2^23).
  const float inv_hw = 1.0f / (float)HWr, inv_w = 1.0f / (float)g.Wr;
  auto fdiv = [](int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
  };
  f32x4 rd[D_LOADS], rg[G_LOADS];
  // PRE: a thread moves 8 channels of a row per slot -- one 16-byte granule of each of the three planes
  constexpr int DV8 = BD / 8, GV8 = BG / 8;
  constexpr int D_LOADS8 = (RK * DV8) / 256, G_LOADS8 = (RK * GV8) / 256;
  static_assert(PRE == 0 || (D_LOADS8 >= 1 && G_LOADS8 >= 1), "plane granules: tile too small for 256 threads");
  u32x4 pd[PD ? D_LOADS8 : 1][3], pg[PG ? G_LOADS8 : 1][3];      // (ext_vector_type: HIP uint4 structs selected as a whole go to scratch)
  unsigned okd = 0, okg = 0;   // okg: 4 bits per load (per-element validity in the im2col mode)
  auto gload = [&](int r0) {
    if constexpr (PD) {
#pragma unroll
      for (int i = 0; i < D_LOADS8; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / DV8, v = idx - r * DV8;
        const int row = r0 + r;
        const bool ok = row < row_end;
        const bf16_t* b = reinterpret_cast<const bf16_t*>(D) + (size_t)(ok ? row : 0) * 3 * g.Cd + cd0 + v * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) pd[i][p] = *reinterpret_cast<const u32x4*>(b + p * g.Cd);
        okd = ok ? (okd | (1u << i)) : (okd & ~(1u << i));
      }
    } else
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / DV, v = idx - r * DV;
      const int row = r0 + r;
      const bool ok = row < row_end;
      if constexpr (BF16 && (ST == 1 || ST == 2))
        rd[i] = ld4<bf16_t>(reinterpret_cast<const bf16_t*>(D) + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 4);
      else if constexpr (BF16 && (ST == 5 || ST == 6))
        rd[i] = ld4<half_t>(reinterpret_cast<const half_t*>(D) + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 4);
      else
        rd[i] = *reinterpret_cast<const f32x4*>(D + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 4);
      okd = ok ? (okd | (1u << i)) : (okd & ~(1u << i));
    }
    if constexpr (PG) {
#pragma unroll
      for (int i = 0; i < G_LOADS8; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / GV8, v = idx - r * GV8;
        const int row = r0 + r;
        bool ok = row < row_end;
        int pix = row;
        if constexpr (MODE == MMDYN_CONV) {
          int bb, p, rr, cc;
          fdiv(row, HWr, inv_hw, bb, p);
          fdiv(p, g.Wr, inv_w, rr, cc);
          const int y = rr * g.rs + g.ro + dh, x = cc * g.rs + g.ro + dw;
          ok = ok & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
          pix = (bb * g.Hi + y) * g.Wi + x;
        }
        const bf16_t* b = reinterpret_cast<const bf16_t*>(Gt) + (size_t)(ok ? pix : 0) * 3 * g.Cg + cg0 + v * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) pg[i][p] = *reinterpret_cast<const u32x4*>(b + p * g.Cg);
        okg = ok ? (okg | (0xFu << (4 * i))) : (okg & ~(0xFu << (4 * i)));
      }
    } else
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / GV, v = idx - r * GV;
      const int row = r0 + r;
      bool ok = row < row_end;
      int pix = row;
      if constexpr (MODE == MMDYN_IM2COL3) {
        // G row = im2col of the NCHW 3-channel tensor: columns ci*16 + kh*4 + kw (48 real + 16 zero)
        int bb, p, rr, cc;
        fdiv(row, HWr, inv_hw, bb, p);
        fdiv(p, g.Wr, inv_w, rr, cc);
        const int k0 = cg0 + v * 4, ci = k0 >> 4, kh = (k0 >> 2) & 3;
        const int y = 2 * rr - 1 + kh, x0 = 2 * cc - 1;
        const bool okr = ok & (ci < 3) & ((unsigned)y < (unsigned)g.Hi);
        const int base = okr ? ((bb * 3 + ci) * g.Hi + y) * g.Wi : 0;
        unsigned m = 0;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          const int xx = x0 + kw;
          const bool okk = okr & ((unsigned)xx < (unsigned)g.Wi);
          rg[i][kw] = Gt[(size_t)base + (okk ? xx : 0)];
          m |= okk ? (1u << kw) : 0u;
        }
        okg = (okg & ~(0xFu << (4 * i))) | (m << (4 * i));
        continue;
      }
      if constexpr (MODE == MMDYN_CONV) {
        int bb, p, rr, cc;
        fdiv(row, HWr, inv_hw, bb, p);
        fdiv(p, g.Wr, inv_w, rr, cc);
        const int y = rr * g.rs + g.ro + dh, x = cc * g.rs + g.ro + dw;
        ok = ok & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
        pix = (bb * g.Hi + y) * g.Wi + x;
      }
      if constexpr (BF16 && (ST == 1 || ST == 3))
        rg[i] = ld4<bf16_t>(reinterpret_cast<const bf16_t*>(Gt) + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 4);
      else if constexpr (BF16 && (ST == 5 || ST == 7))
        rg[i] = ld4<half_t>(reinterpret_cast<const half_t*>(Gt) + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 4);
      else
        rg[i] = *reinterpret_cast<const f32x4*>(Gt + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 4);
      okg = ok ? (okg | (0xFu << (4 * i))) : (okg & ~(0xFu << (4 * i)));
    }
  };
  auto lds_store = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      if constexpr (PD) {
#pragma unroll
        for (int i = 0; i < D_LOADS8; ++i) {
          const int idx = tid + 256 * i;
          const int r = idx / DV8, v = idx - r * DV8;
          const int o = r * LDD + v * 8;
          const bool ok = (okd >> i) & 1u;
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&Ds16[p * RK * LDD + o]) = ok ? pd[i][p] : z4;
        }
      } else {
#pragma unroll
        for (int i = 0; i < D_LOADS; ++i) {
          const int idx = tid + 256 * i;
          const int r = idx / DV, v = idx - r * DV;
          const int o = r * LDD + v * 4;
          const f32x4 x = ((okd >> i) & 1u) ? rd[i] : zero;
          uint2 hh, mm, ll;
          split3_bf16(x[0], x[1], hh.x, mm.x, ll.x);
          split3_bf16(x[2], x[3], hh.y, mm.y, ll.y);
          *reinterpret_cast<uint2*>(&Ds16[o]) = hh;
          *reinterpret_cast<uint2*>(&Ds16[RK * LDD + o]) = mm;
          *reinterpret_cast<uint2*>(&Ds16[2 * RK * LDD + o]) = ll;
        }
      }
      if constexpr (PG) {
#pragma unroll
        for (int i = 0; i < G_LOADS8; ++i) {
          const int idx = tid + 256 * i;
          const int r = idx / GV8, v = idx - r * GV8;
          const int o = r * LDG + v * 8;
          const bool ok = (okg >> (4 * i)) & 1u;
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&Gs16[p * RK * LDG + o]) = ok ? pg[i][p] : z4;
        }
      } else {
#pragma unroll
        for (int i = 0; i < G_LOADS; ++i) {
          const int idx = tid + 256 * i;
          const int r = idx / GV, v = idx - r * GV;
          const int o = r * LDG + v * 4;
          const f32x4 x = ((okg >> (4 * i)) & 1u) ? rg[i] : zero;        // (all four mask bits are equal outside IM2COL3)
          uint2 hh, mm, ll;
          split3_bf16(x[0], x[1], hh.x, mm.x, ll.x);
          split3_bf16(x[2], x[3], hh.y, mm.y, ll.y);
          *reinterpret_cast<uint2*>(&Gs16[o]) = hh;
          *reinterpret_cast<uint2*>(&Gs16[RK * LDG + o]) = mm;
          *reinterpret_cast<uint2*>(&Gs16[2 * RK * LDG + o]) = ll;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i)
      reinterpret_cast<f32x4*>(Ds)[tid + 256 * i] = ((okd >> i) & 1u) ? rd[i] : zero;
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const unsigned m = okg >> (4 * i);
      f32x4 v;
      v[0] = (m & 1u) ? rg[i][0] : 0.f;
      v[1] = (m & 2u) ? rg[i][1] : 0.f;
      v[2] = (m & 4u) ? rg[i][2] : 0.f;
      v[3] = (m & 8u) ? rg[i][3] : 0.f;
      reinterpret_cast<f32x4*>(Gs)[tid + 256 * i] = v;
    }
  };

  f32x16 acc[DT][GT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < GT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int h = lane >> 5, cl = lane & 31;
  if (row_begin < row_end) {
    gload(row_begin);
    lds_store();
    __syncthreads();
    for (int r0 = row_begin; r0 < row_end; r0 += RK) {
      gload(r0 + RK);                      // past the chunk end every row is masked: a harmless dummy fetch
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (X3) {
        // transposed-read lane roles as in wgrad_b16_kernel: group gq = lane>>4 (columns 16*(gq&1).., k half gq>>1), lane 4q+p of
        // the group supplies the address of row q, columns 4p..4p+3 of its block
        const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
        const int trow = 8 * (gq >> 1) + tq, tcol = 16 * (gq & 1) + 4 * tp;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
        typedef __bf16 xb16x8 __attribute__((ext_vector_type(8)));
        auto frag = [&](const bf16_t* tile, int ld, int col0, int k0) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow) * ld + col0 + tcol]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow + 4) * ld + col0 + tcol]));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          return __builtin_bit_cast(xb16x8, v);
        };
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {          // 16 rows per MFMA
          xb16x8 pa[3][DT], pb[3][GT];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < DT; ++a) pa[p][a] = frag(Ds16 + p * RK * LDD, LDD, wd * WD + a * 32, kc * 16);
#pragma unroll
            for (int b = 0; b < GT; ++b) pb[p][b] = frag(Gs16 + p * RK * LDG, LDG, wg * WG + b * 32, kc * 16);
          }
          constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // smallest terms first
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < DT; ++a)
#pragma unroll
              for (int b = 0; b < GT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[order[t][0]][a], pb[order[t][1]][b], acc[a][b], 0, 0, 0);
        }
      } else if constexpr (BF16) {
#pragma unroll
        for (int r8 = wk * 2 * KPW; r8 < (wk + 1) * 2 * KPW; r8 += 8) {     // 8 rows per bf16 MFMA: lane half h -> 4 rows
          if constexpr (ST >= 4) {
            f16x4 pa[DT], pb[GT];
#pragma unroll
            for (int a = 0; a < DT; ++a) {
              const float* p = &Ds[(r8 + 4 * h) * BD + wd * WD + a * 32 + cl];
              pa[a] = pack4_f16(p[0], p[BD], p[2 * BD], p[3 * BD]);
            }
#pragma unroll
            for (int b = 0; b < GT; ++b) {
              const float* p = &Gs[(r8 + 4 * h) * BG + wg * WG + b * 32 + cl];
              pb[b] = pack4_f16(p[0], p[BG], p[2 * BG], p[3 * BG]);
            }
#pragma unroll
            for (int a = 0; a < DT; ++a)
#pragma unroll
              for (int b = 0; b < GT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8f16(pa[a], pb[b], acc[a][b], 0, 0, 0);
          } else {
          s16x4 pa[DT], pb[GT];
#pragma unroll
          for (int a = 0; a < DT; ++a) {
            const float* p = &Ds[(r8 + 4 * h) * BD + wd * WD + a * 32 + cl];
            pa[a] = pack4_bf16(p[0], p[BD], p[2 * BD], p[3 * BD]);
          }
#pragma unroll
          for (int b = 0; b < GT; ++b) {
            const float* p = &Gs[(r8 + 4 * h) * BG + wg * WG + b * 32 + cl];
            pb[b] = pack4_bf16(p[0], p[BG], p[2 * BG], p[3 * BG]);
          }
#pragma unroll
          for (int a = 0; a < DT; ++a)
#pragma unroll
            for (int b = 0; b < GT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(pa[a], pb[b], acc[a][b], 0, 0, 0);
          }
        }
      } else
#pragma unroll
      for (int kk = wk * KPW; kk < (wk + 1) * KPW; ++kk) {
        float af[DT], bf[GT];
#pragma unroll
        for (int a = 0; a < DT; ++a) af[a] = Ds[(2 * kk + h) * BD + wd * WD + a * 32 + cl];
#pragma unroll
        for (int b = 0; b < GT; ++b) bf[b] = Gs[(2 * kk + h) * BG + wg * WG + b * 32 + cl];
#pragma unroll
        for (int a = 0; a < DT; ++a)
#pragma unroll
          for (int b = 0; b < GT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // the masking selects of lds_store (which wait for the loads) stay behind the MFMAs
      __syncthreads();
      lds_store();
      __syncthreads();
    }
  }

  float* out = partial + ((size_t)(((chunk * WK + wk) * ngrp + grp) * g.ntaps + tap) * g.Cd) * g.Cg;
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cd = cd0 + wd * WD + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int b = 0; b < GT; ++b) {
        const int cg = cg0 + wg * WG + b * 32 + cl;
        out[(size_t)cd * g.Cg + cg] = acc[a][b][e];
      }
    }
}

// ---- all-bf16 variant (bf16-storage mode: D and Gt are bf16 in HBM) ------------------------------------------------
// Same problem split (tile, tap, chunk) and partial-slab output as wgrad_tn_kernel, but the operand path is built for
// 16-bit data on CDNA4: 16-byte global loads (eight bf16), bf16 tiles in LDS, and the k-strided MFMA fragments (the
// reduction index is the tile ROW) fetched with the hardware transposing read ds_read_b64_tr_b16: per 16-lane group it
// returns, for each of 16 consecutive columns, the four elements of four consecutive rows -- two of them are one
// operand of v_mfma_f32_32x32x16_bf16.  Against the fp32-LDS path above (four strided ds_read_b32 + a pack per 8-deep
// MFMA) this is 4x fewer LDS cycles per k, no conversion VALU work and half the MFMA issues.
// Row stride of a tile: bytes = 64 (mod 256), so the four rows of a transposed read (64 B each) fill all 64 banks.
typedef __bf16 wb16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef float wacc16_t __attribute__((ext_vector_type(16)));
// H16: the 16-bit operands are IEEE half instead of bf16 (precision "fp16s") -- same bytes, same data path, other MFMA
template <bool H16> __device__ __forceinline__ wacc16_t mfma16(wb16x8 a, wb16x8 b, wacc16_t c) {
  if constexpr (H16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, a), __builtin_bit_cast(wf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int BX> struct b16_ld { static constexpr int v = (BX == 32) ? 32 : BX + 32; };   // in elements

// (waves_per_eu: without the hint the register allocator aims at eight waves per SIMD and spills the prefetched
//  granules to scratch right after loading them)
template <int MODE, int BD, int BG, int WD, int WG, int WK, bool H16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void wgrad_b16_kernel(const bf16_t* __restrict__ D, const bf16_t* __restrict__ Gt,
                                                        float* __restrict__ partial, const WgradGeom g) {
  constexpr int DT = WD / 32, GT = WG / 32;
  constexpr int WAVES_G = BG / WG;
  constexpr int WAVES_DG = (BD / WD) * WAVES_G;
  static_assert(WAVES_DG * WK == 4 && (WK == 1 || WK == 2), "4 waves per block, at most two of them share a tile");
  constexpr int LDD = b16_ld<BD>::v, LDG = b16_ld<BG>::v;
  constexpr int DV = BD / 8, GV = BG / 8;                    // 16-byte granules per tile row
  constexpr int D_LOADS = (RK * DV + 255) / 256, G_LOADS = (RK * GV + 255) / 256;
  constexpr bool D_ALL = (RK * DV) % 256 == 0, G_ALL = (RK * GV) % 256 == 0;

  // (tiles narrower than 64 channels need fewer than 256 granules per K-step: the surplus threads load a valid dummy
  //  address and store into rows RK.. that nobody reads -- predicating them instead makes the compiler sink the load
  //  into the store branch, behind the barrier, where its latency is exposed)
  constexpr int D_ROWS = D_LOADS * 256 / DV, G_ROWS = G_LOADS * 256 / GV;
  __shared__ __attribute__((aligned(16))) bf16_t Ds[D_ROWS * LDD];
  __shared__ __attribute__((aligned(16))) bf16_t Gs[G_ROWS * LDG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WAVES_DG, wdg = wave % WAVES_DG;
  const int wd = wdg / WAVES_G, wg = wdg % WAVES_G;
  // XCD-aware block order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  All
  // blocks of one row chunk -- its channel tiles x taps, which re-read the same rows of D and Gt -- get linear ids that
  // are equal modulo 8, so a chunk's rows are fetched into ONE L2 instead of all eight (the operands of the large
  // layers do not fit a 4 MB L2 eight times over: 157 -> 300 TFLOP/s class difference between bs 1024 and bs 256).
  const int tiles_g = g.Cg / BG;
  const int per_chunk = (g.Cd / BD) * tiles_g * g.ntaps;
  const int L = blockIdx.x, xcd = L & 7, jj = L >> 3;
  const int chunk = xcd + 8 * (jj / per_chunk), inner = jj % per_chunk;
  if (chunk >= g.chunks / WK) return;                 // (uniform per block: before any barrier)
  const int tap = inner % g.ntaps, tile = inner / g.ntaps;
  const int td = tile / tiles_g, tg = tile - td * tiles_g;
  const int cd0 = td * BD, cg0 = tg * BG;
  const int dh = (MODE == MMDYN_CONV) ? (tap >> 2) : 0;      // (the gather mode is a template parameter: the fetch below
  const int dw = (MODE == MMDYN_CONV) ? (tap & 3) : 0;       //  must stay one basic block, see wgrad_tn_kernel)
  const int HWr = g.Hr * g.Wr;
  const int row_begin = chunk * g.rows_per_chunk;
  const int row_end = min(g.rows, row_begin + g.rows_per_chunk);
  const float inv_hw = 1.0f / (float)HWr, inv_w = 1.0f / (float)g.Wr;
  auto fdiv = [](int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
  };

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));     // (a native vector: arrays of HIP's uint4 struct with a
  u32x4 rd[D_LOADS], rg[G_LOADS];                                  //  whole-struct select end up in scratch memory)
  unsigned okd = 0, okg = 0;
  auto gload = [&](int r0) {          // branch-free: masked rows read a valid dummy address, zeroed at the LDS store
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / DV, v = idx - r * DV;
      const int row = r0 + r;
      const bool ok = (D_ALL || r < RK) & (row < row_end);
      rd[i] = *reinterpret_cast<const u32x4*>(D + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 8);
      okd = ok ? (okd | (1u << i)) : (okd & ~(1u << i));
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / GV, v = idx - r * GV;
      const int row = r0 + r;
      bool ok = (G_ALL || r < RK) & (row < row_end);
      int pix = row;
      if constexpr (MODE == MMDYN_CONV) {
        int bb, p, rr, cc;
        fdiv(row, HWr, inv_hw, bb, p);
        fdiv(p, g.Wr, inv_w, rr, cc);
        const int y = rr * g.rs + g.ro + dh, xx = cc * g.rs + g.ro + dw;
        ok = ok & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)xx < (unsigned)g.Wi);
        pix = (bb * g.Hi + y) * g.Wi + xx;
      }
      rg[i] = *reinterpret_cast<const u32x4*>(Gt + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 8);
      okg = ok ? (okg | (1u << i)) : (okg & ~(1u << i));
    }
  };
  auto lds_store = [&]() {
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / DV, v = idx - r * DV;
      *reinterpret_cast<u32x4*>(&Ds[r * LDD + v * 8]) = ((okd >> i) & 1u) ? rd[i] : zero;
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / GV, v = idx - r * GV;
      *reinterpret_cast<u32x4*>(&Gs[r * LDG + v * 8]) = ((okg >> i) & 1u) ? rg[i] : zero;
    }
  };

  f32x16 acc[DT][GT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < GT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // transposed-read lane roles: group gq = lane>>4 (columns 16*(gq&1).., k half gq>>1), lane 4q+p of the group supplies
  // the address of row q, columns 4p..4p+3 of its block
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (gq >> 1) + tq, tcol = 16 * (gq & 1) + 4 * tp;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
  auto frag = [&](const bf16_t* tile, int ld, int col0, int k0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow) * ld + col0 + tcol]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow + 4) * ld + col0 + tcol]));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(wb16x8, v);
  };

  const int h = lane >> 5, cl = lane & 31;
  if (row_begin < row_end) {
    gload(row_begin);
    lds_store();
    __syncthreads();
    for (int r0 = row_begin; r0 < row_end; r0 += RK) {
      gload(r0 + RK);                      // past the chunk end every row is masked: a harmless dummy fetch
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = wk * (2 / WK); kc < (wk + 1) * (2 / WK); ++kc) {          // 16 rows per MFMA; WK waves split the 32
        wb16x8 pa[DT], pb[GT];
#pragma unroll
        for (int a = 0; a < DT; ++a) pa[a] = frag(Ds, LDD, wd * WD + a * 32, kc * 16);
#pragma unroll
        for (int b = 0; b < GT; ++b) pb[b] = frag(Gs, LDG, wg * WG + b * 32, kc * 16);
#pragma unroll
        for (int a = 0; a < DT; ++a)
#pragma unroll
          for (int b = 0; b < GT; ++b)
            acc[a][b] = mfma16<H16>(pa[a], pb[b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);   // the masking selects of lds_store (which wait for the loads) stay behind the MFMAs
      __syncthreads();
      lds_store();
      __syncthreads();
    }
  }

  float* out = partial + ((size_t)((chunk * WK + wk) * g.ntaps + tap) * g.Cd) * g.Cg;
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cd = cd0 + wd * WD + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int b = 0; b < GT; ++b) {
        const int cg = cg0 + wg * WG + b * 32 + cl;
        out[(size_t)cd * g.Cg + cg] = acc[a][b][e];
      }
    }
}

// All-bf16 four-taps-per-block variant (narrow channel tiles of the k4 convolutions, see wgrad_tn4_kernel below for the
// idea): the four waves own the four kw taps of kernel row kh = blockIdx.y and share one D tile; each wave stages its own
// tap's gathered G tile.  Operand path as in wgrad_b16_kernel (16-byte granules, bf16 LDS tiles, transposing reads).
template <int BD, int BG, bool H16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void wgrad_b16_tn4_kernel(
    const bf16_t* __restrict__ D, const bf16_t* __restrict__ Gt, float* __restrict__ partial, const WgradGeom g) {
  constexpr int DT = BD / 32, GT = BG / 32;
  constexpr int LDD = b16_ld<BD>::v, LDG = b16_ld<BG>::v;
  constexpr int DV = BD / 8, GV = BG / 8;                  // 16-byte granules per tile row
  constexpr int D_LOADS = (RK * DV + 255) / 256;           // granules per thread of the shared D tile
  constexpr int D_ROWS = D_LOADS * 256 / DV;               // (surplus threads fill rows RK.. that nobody reads)
  constexpr int G_LOADS = (RK * GV + 63) / 64;             // granules per lane of the wave's own G tile
  constexpr int G_ROWS = G_LOADS * 64 / GV;
  static_assert(DT <= 2 && GT <= 2, "tile too large for four accumulator sets");

  __shared__ __attribute__((aligned(16))) bf16_t Ds[D_ROWS * LDD];
  __shared__ __attribute__((aligned(16))) bf16_t Gs[4 * G_ROWS * LDG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_g = g.Cg / BG;
  const int td = blockIdx.x / tiles_g, tg = blockIdx.x - td * tiles_g;
  const int cd0 = td * BD, cg0 = tg * BG;
  const int kh = blockIdx.y, kw = wave, chunk = blockIdx.z;
  const int HWr = g.Hr * g.Wr;
  const int row_begin = chunk * g.rows_per_chunk;
  const int row_end = min(g.rows, row_begin + g.rows_per_chunk);
  const float inv_hw = 1.0f / (float)HWr, inv_w = 1.0f / (float)g.Wr;
  auto fdiv = [](int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
  };
  bf16_t* Gw = Gs + wave * G_ROWS * LDG;

  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rd[D_LOADS], rg[G_LOADS];
  unsigned okd = 0, okg = 0;
  auto gload = [&](int r0) {
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / DV, v = idx - r * DV;
      const int row = r0 + r;
      const bool ok = (r < RK) & (row < row_end);
      rd[i] = *reinterpret_cast<const u32x4*>(D + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 8);
      okd = ok ? (okd | (1u << i)) : (okd & ~(1u << i));
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int idx = lane + 64 * i;
      const int r = idx / GV, v = idx - r * GV;
      const int row = r0 + r;
      int bb, p, rr, cc;
      fdiv(row, HWr, inv_hw, bb, p);
      fdiv(p, g.Wr, inv_w, rr, cc);
      const int y = rr * g.rs + g.ro + kh, xx = cc * g.rs + g.ro + kw;
      const bool ok = (r < RK) & (row < row_end) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)xx < (unsigned)g.Wi);
      const int pix = (bb * g.Hi + y) * g.Wi + xx;
      rg[i] = *reinterpret_cast<const u32x4*>(Gt + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 8);
      okg = ok ? (okg | (1u << i)) : (okg & ~(1u << i));
    }
  };
  auto lds_store = [&]() {
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / DV, v = idx - r * DV;
      *reinterpret_cast<u32x4*>(&Ds[r * LDD + v * 8]) = ((okd >> i) & 1u) ? rd[i] : zero;
    }
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int idx = lane + 64 * i;
      const int r = idx / GV, v = idx - r * GV;
      *reinterpret_cast<u32x4*>(&Gw[r * LDG + v * 8]) = ((okg >> i) & 1u) ? rg[i] : zero;
    }
  };

  f32x16 acc[DT][GT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < GT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (gq >> 1) + tq, tcol = 16 * (gq & 1) + 4 * tp;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
  auto frag = [&](const bf16_t* tile, int ld, int col0, int k0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow) * ld + col0 + tcol]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow + 4) * ld + col0 + tcol]));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(wb16x8, v);
  };

  const int h = lane >> 5, cl = lane & 31;
  if (row_begin < row_end) {
    gload(row_begin);
    lds_store();
    __syncthreads();
    for (int r0 = row_begin; r0 < row_end; r0 += RK) {
      gload(r0 + RK);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        wb16x8 pa[DT], pb[GT];
#pragma unroll
        for (int a = 0; a < DT; ++a) pa[a] = frag(Ds, LDD, a * 32, kc * 16);
#pragma unroll
        for (int b = 0; b < GT; ++b) pb[b] = frag(Gw, LDG, b * 32, kc * 16);
#pragma unroll
        for (int a = 0; a < DT; ++a)
#pragma unroll
          for (int b = 0; b < GT; ++b)
            acc[a][b] = mfma16<H16>(pa[a], pb[b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      lds_store();
      __syncthreads();
    }
  }
  float* out = partial + ((size_t)(chunk * g.ntaps + kh * 4 + kw) * g.Cd) * g.Cg;
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cd = cd0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int b = 0; b < GT; ++b) out[(size_t)cd * g.Cg + cg0 + b * 32 + cl] = acc[a][b][e];
    }
}

// Four-taps-per-block variant for the small-channel convolutions (tile <= 64x64): the four waves of a block own
// the four kw taps of one kernel row kh and share ONE dense-operand tile, so D is fetched once per 4 taps instead
// of once per tap (half the L2 traffic of the one-tap kernel on these shapes) and no wave has to split the row
// reduction (WK = 1: a quarter / half of the partial slabs).  grid.y = kh.
// X3: the three-term split of the fp32 operands on the bf16 matrix cores, as in wgrad_tn_kernel (three bf16 planes per tile in
// LDS, transposing fragment reads, six products).
// PRE: as in wgrad_tn_kernel (bit 0: D arrives split, bit 1: Gt).
template <int BD, int BG, bool BF16, int ST, bool X3 = false, int PRE = 0>
__global__ __launch_bounds__(256) void wgrad_tn4_kernel(const float* __restrict__ D, const float* __restrict__ Gt,
                                                        float* __restrict__ partial, const WgradGeom g) {
  static_assert(!X3 || (!BF16 && ST == 0), "the three-term split is a variant of the fp32 kernel");
  static_assert(PRE == 0 || X3, "operands that arrive split belong to the three-term split arithmetic");
  constexpr bool PD = (PRE & 1) != 0, PG = (PRE & 2) != 0;
  constexpr int DT = BD / 32, GT = BG / 32;
  constexpr int DV = BD / 4, GV = BG / 4;
  constexpr int D_LOADS = (RK * DV + 255) / 256;          // float4 per thread for the shared D tile
  // (compile-time: with a run-time `idx < RK * DV` test the compiler sinks that load into the LDS-store branch, behind
  //  the barrier, and waits for it there -- a full memory round trip exposed in every K-step)
  constexpr bool D_ALL = (RK * DV) % 256 == 0;
  constexpr int G_LOADS = (RK * GV) / 64;                 // float4 per lane: each wave fetches its own tap's tile
  static_assert(DT <= 2 && GT <= 2, "tile too large for four accumulator sets");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ds = reinterpret_cast<float*>(smem);            // [RK][BD]
  float* Gs = Ds + RK * BD;                              // [4 taps][RK][BG]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_g = g.Cg / BG;
  const int td = blockIdx.x / tiles_g, tg = blockIdx.x - td * tiles_g;
  const int cd0 = td * BD, cg0 = tg * BG;
  const int kh = blockIdx.y, kw = wave, chunk = blockIdx.z;
  const int HWr = g.Hr * g.Wr;
  const int row_begin = chunk * g.rows_per_chunk;
  const int row_end = min(g.rows, row_begin + g.rows_per_chunk);
  const float inv_hw = 1.0f / (float)HWr, inv_w = 1.0f / (float)g.Wr;
  auto fdiv = [](int n, int d, float inv, int& q, int& r) {
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
  };
  float* Gw = Gs + wave * RK * BG;
  constexpr int LDD = x3_ld<BD>::v, LDG = x3_ld<BG>::v;
  bf16_t* Ds16 = reinterpret_cast<bf16_t*>(smem);                 // X3: [plane][RK][LDD], then per wave [plane][RK][LDG]
  bf16_t* Gw16 = Ds16 + 3 * RK * LDD + wave * 3 * RK * LDG;

  f32x4 rd[D_LOADS], rg[G_LOADS];
  // PRE: 16-byte plane granules (8 channels of a row): the shared D tile by all 256 threads, each wave's tap tile by its 64 lanes
  constexpr int DV8 = BD / 8, GV8 = BG / 8;
  constexpr bool D_ALL8 = (RK * DV8) % 256 == 0;          // (BD = 32: the first 128 threads load)
  constexpr int G_LOADS8 = (RK * GV8) / 64;
  u32x4 pd8[3], pg[PG ? G_LOADS8 : 1][3];
  unsigned okd = 0, okg = 0;
  auto gload = [&](int r0) {
    if constexpr (PD) {
      const int r = tid / DV8, v = tid - r * DV8;
      const int row = r0 + r;
      const bool ok = (D_ALL8 || tid < RK * DV8) & (row < row_end);
      const bf16_t* b = reinterpret_cast<const bf16_t*>(D) + (size_t)(ok ? row : 0) * 3 * g.Cd + cd0 + (D_ALL8 || tid < RK * DV8 ? v : 0) * 8;
#pragma unroll
      for (int p = 0; p < 3; ++p) pd8[p] = *reinterpret_cast<const u32x4*>(b + p * g.Cd);
      okd = ok ? 1u : 0u;
    } else
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i) {
      const int idx = tid + 256 * i;
      const int r = idx / DV, v = idx - r * DV;
      const int row = r0 + r;
      const bool ok = (D_ALL || idx < RK * DV) & (row < row_end);
      if constexpr (BF16 && (ST == 1 || ST == 2))
        rd[i] = ld4<bf16_t>(reinterpret_cast<const bf16_t*>(D) + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 4);
      else if constexpr (BF16 && (ST == 5 || ST == 6))
        rd[i] = ld4<half_t>(reinterpret_cast<const half_t*>(D) + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 4);
      else
        rd[i] = *reinterpret_cast<const f32x4*>(D + (size_t)(ok ? row : 0) * g.Cd + cd0 + v * 4);
      okd = ok ? (okd | (1u << i)) : (okd & ~(1u << i));
    }
    if constexpr (PG) {
#pragma unroll
      for (int i = 0; i < G_LOADS8; ++i) {
        const int idx = lane + 64 * i;
        const int r = idx / GV8, v = idx - r * GV8;
        const int row = r0 + r;
        int bb, p, rr, cc;
        fdiv(row, HWr, inv_hw, bb, p);
        fdiv(p, g.Wr, inv_w, rr, cc);
        const int y = rr * g.rs + g.ro + kh, x = cc * g.rs + g.ro + kw;
        const bool ok = (row < row_end) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
        const int pix = (bb * g.Hi + y) * g.Wi + x;
        const bf16_t* b = reinterpret_cast<const bf16_t*>(Gt) + (size_t)(ok ? pix : 0) * 3 * g.Cg + cg0 + v * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) pg[i][q] = *reinterpret_cast<const u32x4*>(b + q * g.Cg);
        okg = ok ? (okg | (1u << i)) : (okg & ~(1u << i));
      }
    } else
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i) {
      const int idx = lane + 64 * i;
      const int r = idx / GV, v = idx - r * GV;
      const int row = r0 + r;
      int bb, p, rr, cc;
      fdiv(row, HWr, inv_hw, bb, p);
      fdiv(p, g.Wr, inv_w, rr, cc);
      const int y = rr * g.rs + g.ro + kh, x = cc * g.rs + g.ro + kw;
      const bool ok = (row < row_end) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
      const int pix = (bb * g.Hi + y) * g.Wi + x;
      if constexpr (BF16 && (ST == 1 || ST == 3))
        rg[i] = ld4<bf16_t>(reinterpret_cast<const bf16_t*>(Gt) + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 4);
      else if constexpr (BF16 && (ST == 5 || ST == 7))
        rg[i] = ld4<half_t>(reinterpret_cast<const half_t*>(Gt) + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 4);
      else
        rg[i] = *reinterpret_cast<const f32x4*>(Gt + (size_t)(ok ? pix : 0) * g.Cg + cg0 + v * 4);
      okg = ok ? (okg | (1u << i)) : (okg & ~(1u << i));
    }
  };
  auto lds_store = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (X3) {
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      if constexpr (PD) {
        if (D_ALL8 || tid < RK * DV8) {
          const int r = tid / DV8, v = tid - r * DV8;
          const int o = r * LDD + v * 8;
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&Ds16[p * RK * LDD + o]) = okd ? pd8[p] : z4;
        }
      } else {
#pragma unroll
        for (int i = 0; i < D_LOADS; ++i) {
          const int idx = tid + 256 * i;
          if (D_ALL || idx < RK * DV) {
            const int r = idx / DV, v = idx - r * DV;
            const f32x4 x = ((okd >> i) & 1u) ? rd[i] : zero;
            uint2 hh, mm, ll;
            split3_bf16(x[0], x[1], hh.x, mm.x, ll.x);
            split3_bf16(x[2], x[3], hh.y, mm.y, ll.y);
            const int o = r * LDD + v * 4;
            *reinterpret_cast<uint2*>(&Ds16[o]) = hh;
            *reinterpret_cast<uint2*>(&Ds16[RK * LDD + o]) = mm;
            *reinterpret_cast<uint2*>(&Ds16[2 * RK * LDD + o]) = ll;
          }
        }
      }
      if constexpr (PG) {
#pragma unroll
        for (int i = 0; i < G_LOADS8; ++i) {
          const int idx = lane + 64 * i;
          const int r = idx / GV8, v = idx - r * GV8;
          const int o = r * LDG + v * 8;
          const bool ok = (okg >> i) & 1u;
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(&Gw16[p * RK * LDG + o]) = ok ? pg[i][p] : z4;
        }
      } else {
#pragma unroll
        for (int i = 0; i < G_LOADS; ++i) {
          const int idx = lane + 64 * i;
          const int r = idx / GV, v = idx - r * GV;
          const f32x4 x = ((okg >> i) & 1u) ? rg[i] : zero;
          uint2 hh, mm, ll;
          split3_bf16(x[0], x[1], hh.x, mm.x, ll.x);
          split3_bf16(x[2], x[3], hh.y, mm.y, ll.y);
          const int o = r * LDG + v * 4;
          *reinterpret_cast<uint2*>(&Gw16[o]) = hh;
          *reinterpret_cast<uint2*>(&Gw16[RK * LDG + o]) = mm;
          *reinterpret_cast<uint2*>(&Gw16[2 * RK * LDG + o]) = ll;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < D_LOADS; ++i)
      if (D_ALL || tid + 256 * i < RK * DV) reinterpret_cast<f32x4*>(Ds)[tid + 256 * i] = ((okd >> i) & 1u) ? rd[i] : zero;
#pragma unroll
    for (int i = 0; i < G_LOADS; ++i)
      reinterpret_cast<f32x4*>(Gw)[lane + 64 * i] = ((okg >> i) & 1u) ? rg[i] : zero;
  };

  f32x16 acc[DT][GT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < GT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int h = lane >> 5, cl = lane & 31;
  if (row_begin < row_end) {
    gload(row_begin);
    lds_store();
    __syncthreads();
    for (int r0 = row_begin; r0 < row_end; r0 += RK) {
      gload(r0 + RK);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (X3) {
        const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;        // transposed-read lane roles (wgrad_b16_kernel)
        const int trow = 8 * (gq >> 1) + tq, tcol = 16 * (gq & 1) + 4 * tp;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
        typedef __bf16 xb16x8 __attribute__((ext_vector_type(8)));
        auto frag = [&](const bf16_t* tile, int ld, int col0, int k0) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow) * ld + col0 + tcol]));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(&tile[(k0 + trow + 4) * ld + col0 + tcol]));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          return __builtin_bit_cast(xb16x8, v);
        };
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          xb16x8 pa[3][DT], pb[3][GT];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < DT; ++a) pa[p][a] = frag(Ds16 + p * RK * LDD, LDD, a * 32, kc * 16);
#pragma unroll
            for (int b = 0; b < GT; ++b) pb[p][b] = frag(Gw16 + p * RK * LDG, LDG, b * 32, kc * 16);
          }
          constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // smallest terms first
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < DT; ++a)
#pragma unroll
              for (int b = 0; b < GT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[order[t][0]][a], pb[order[t][1]][b], acc[a][b], 0, 0, 0);
        }
      } else if constexpr (BF16) {
#pragma unroll
        for (int r8 = 0; r8 < RK; r8 += 8) {
          if constexpr (ST >= 4) {
            f16x4 pa[DT], pb[GT];
#pragma unroll
            for (int a = 0; a < DT; ++a) {
              const float* p = &Ds[(r8 + 4 * h) * BD + a * 32 + cl];
              pa[a] = pack4_f16(p[0], p[BD], p[2 * BD], p[3 * BD]);
            }
#pragma unroll
            for (int b = 0; b < GT; ++b) {
              const float* p = &Gw[(r8 + 4 * h) * BG + b * 32 + cl];
              pb[b] = pack4_f16(p[0], p[BG], p[2 * BG], p[3 * BG]);
            }
#pragma unroll
            for (int a = 0; a < DT; ++a)
#pragma unroll
              for (int b = 0; b < GT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8f16(pa[a], pb[b], acc[a][b], 0, 0, 0);
          } else {
          s16x4 pa[DT], pb[GT];
#pragma unroll
          for (int a = 0; a < DT; ++a) {
            const float* p = &Ds[(r8 + 4 * h) * BD + a * 32 + cl];
            pa[a] = pack4_bf16(p[0], p[BD], p[2 * BD], p[3 * BD]);
          }
#pragma unroll
          for (int b = 0; b < GT; ++b) {
            const float* p = &Gw[(r8 + 4 * h) * BG + b * 32 + cl];
            pb[b] = pack4_bf16(p[0], p[BG], p[2 * BG], p[3 * BG]);
          }
#pragma unroll
          for (int a = 0; a < DT; ++a)
#pragma unroll
            for (int b = 0; b < GT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(pa[a], pb[b], acc[a][b], 0, 0, 0);
          }
        }
      } else
#pragma unroll
      for (int kk = 0; kk < RK / 2; ++kk) {
        float af[DT], bf[GT];
#pragma unroll
        for (int a = 0; a < DT; ++a) af[a] = Ds[(2 * kk + h) * BD + a * 32 + cl];
#pragma unroll
        for (int b = 0; b < GT; ++b) bf[b] = Gw[(2 * kk + h) * BG + b * 32 + cl];
#pragma unroll
        for (int a = 0; a < DT; ++a)
#pragma unroll
          for (int b = 0; b < GT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // the masking selects of lds_store (which wait for the loads) stay behind the MFMAs
      __syncthreads();
      lds_store();
      __syncthreads();
    }
  }
  float* out = partial + ((size_t)(chunk * g.ntaps + kh * 4 + kw) * g.Cd) * g.Cg;
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cd = cd0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
      for (int b = 0; b < GT; ++b) out[(size_t)cd * g.Cg + cg0 + b * 32 + cl] = acc[a][b][e];
    }
}

template <int BD, int BG>
static int launch4(const float* D, const float* Gt, float* partial, WgradGeom g, hipStream_t st, bool bf16) {
  int rpc = ceil_div(g.rows, g.chunks);
  g.rows_per_chunk = ceil_div(rpc, RK) * RK;
  dim3 grid((g.Cd / BD) * (g.Cg / BG), 4, g.chunks);
  size_t smem = (size_t)RK * (BD + 4 * BG) * sizeof(float);
  if (g.x3 && !bf16) {       // fp32 through the bf16 matrix cores (three-term split)
    smem = (size_t)3 * RK * (x3_ld<BD>::v + 4 * x3_ld<BG>::v) * 2;
    // three planes of four tap tiles: the <32,64> instance needs 78 KB: opt in like every other > 64 KB kernel
#define WGRAD4_X3(PRE_)                                                                                               \
  do {                                                                                                                \
    static LdsOptIn opt_in_;                                                                                          \
    if (smem > 65536)                                                                                                 \
      if (int e = opt_in_.ensure((const void*)wgrad_tn4_kernel<BD, BG, false, 0, true, PRE_>, (int)smem)) return e;   \
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, false, 0, true, PRE_>), grid, dim3(256), smem, st, D, Gt, partial, g); \
  } while (0)
    if (g.pre == 3) WGRAD4_X3(3);
    else if (g.pre == 2) WGRAD4_X3(2);
    else if (g.pre == 1) WGRAD4_X3(1);
    else WGRAD4_X3(0);
#undef WGRAD4_X3
    MMDYN_LAUNCH_CHECK();
  }
  if (bf16 && g.d_b16 && g.g_b16 && g.f16)      // both operands IEEE half
    hipLaunchKernelGGL((wgrad_b16_tn4_kernel<BD, BG, true>), grid, dim3(256), 0, st, reinterpret_cast<const bf16_t*>(D),
                       reinterpret_cast<const bf16_t*>(Gt), partial, g);
  else if (bf16 && g.d_b16 && g.g_b16)      // both operands bf16: transposing-LDS-read variant
    hipLaunchKernelGGL((wgrad_b16_tn4_kernel<BD, BG, false>), grid, dim3(256), 0, st, reinterpret_cast<const bf16_t*>(D),
                       reinterpret_cast<const bf16_t*>(Gt), partial, g);
  else if (bf16 && g.d_b16 && g.f16)
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, true, 6>), grid, dim3(256), smem, st, D, Gt, partial, g);
  else if (bf16 && g.g_b16 && g.f16)
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, true, 7>), grid, dim3(256), smem, st, D, Gt, partial, g);
  else if (bf16 && g.d_b16)
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, true, 2>), grid, dim3(256), smem, st, D, Gt, partial, g);
  else if (bf16 && g.g_b16)
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, true, 3>), grid, dim3(256), smem, st, D, Gt, partial, g);
  else if (bf16 && g.f16)
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, true, 4>), grid, dim3(256), smem, st, D, Gt, partial, g);
  else if (bf16)
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, true, 0>), grid, dim3(256), smem, st, D, Gt, partial, g);
  else
    hipLaunchKernelGGL((wgrad_tn4_kernel<BD, BG, false, 0>), grid, dim3(256), smem, st, D, Gt, partial, g);
  MMDYN_LAUNCH_CHECK();
}

// canon (+)= sum_chunks partial, scattered into the reference layout.
// 256 threads = 32 element lanes x 8 chunk lanes: the chunk sum is split 8 ways (short dependent chains even
// for 128+ slabs) and finished through LDS; slab reads stay 128-byte coalesced.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ canon, int chunks, int taps,
                                                           int Cd, int Cg, int cg_canon, int perm, float beta) {
  __shared__ float red[8][33];
  const int64_t slab = (int64_t)taps * Cd * Cg;
  const int il = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int64_t ntiles = (slab + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t i = tile * 32 + il;
    // four independent loads per round (two made the 24 rounds of a 192-slab layer a chain of memory latencies: 12 us)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < slab) {
      int c = cl;
      for (; c + 24 < chunks; c += 32) {
        s0 += partial[(size_t)c * slab + i];
        s1 += partial[(size_t)(c + 8) * slab + i];
        s2 += partial[(size_t)(c + 16) * slab + i];
        s3 += partial[(size_t)(c + 24) * slab + i];
      }
      for (; c < chunks; c += 8) s0 += partial[(size_t)c * slab + i];
    }
    red[cl][il] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (cl == 0 && i < slab) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += red[k][il];
      int cg = (int)(i % Cg);
      int64_t t = i / Cg;
      int cd = (int)(t % Cd);
      int tap = (int)(t / Cd);
      if (cg < cg_canon) {
        int64_t o;
        if (perm == 0) {
          o = ((int64_t)cd * cg_canon + cg) * taps + tap;
        } else if (perm == 1) {  // cg = hw*256 + ch  -> column ch*25 + hw
          int hw = cg / 256, ch = cg - hw * 256;
          o = (int64_t)cd * cg_canon + ch * 25 + hw;
        } else {                 // cd = hw*256 + ch  -> row ch*25 + hw
          int hw = cd / 256, ch = cd - hw * 256;
          o = (int64_t)(ch * 25 + hw) * cg_canon + cg;
        }
        canon[o] = (beta != 0.f) ? (beta * canon[o] + s) : s;
      }
    }
    __syncthreads();
  }
}

// Few slabs (<= 64: every layer with many weights): one thread sums ALL slabs of four consecutive elements --
// `chunks` independent 16-byte loads in flight, no LDS, no barrier -- in slab order (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_vec_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ canon, int chunks, int taps,
                                                               int Cd, int Cg, int cg_canon, int perm, float beta) {
  const int64_t slab = (int64_t)taps * Cd * Cg, nvec = slab >> 2;     // Cg % 32 == 0: a float4 never straddles a row
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = v << 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int c = 0; c < chunks; ++c) s += *reinterpret_cast<const f32x4*>(partial + (size_t)c * slab + i);
    const int cg0 = (int)(i % Cg);
    const int64_t t = i / Cg;
    const int cd = (int)(t % Cd), tap = (int)(t / Cd);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cg = cg0 + k;
      if (cg >= cg_canon) continue;
      int64_t o;
      if (perm == 0) {
        o = ((int64_t)cd * cg_canon + cg) * taps + tap;
      } else if (perm == 1) {
        int hw = cg / 256, ch = cg - hw * 256;
        o = (int64_t)cd * cg_canon + ch * 25 + hw;
      } else {
        int hw = cd / 256, ch = cd - hw * 256;
        o = (int64_t)(ch * 25 + hw) * cg_canon + cg;
      }
      canon[o] = (beta != 0.f) ? (beta * canon[o] + s[k]) : s[k];
    }
  }
}

template <int BD, int BG, int WD, int WG, int WK>
static int launch_b16(const float* D, const float* Gt, float* partial, WgradGeom g, hipStream_t st) {
  const int zblocks = g.chunks / WK;
  int rpc = ceil_div(g.rows, zblocks);
  g.rows_per_chunk = ceil_div(rpc, RK) * RK;
  dim3 grid((unsigned)((g.Cd / BD) * (g.Cg / BG) * g.ntaps) * (unsigned)((zblocks + 7) / 8 * 8));
#define WGRAD_B16(M, H)                                                                                    \
  hipLaunchKernelGGL((wgrad_b16_kernel<M, BD, BG, WD, WG, WK, H>), grid, dim3(256), 0, st,                 \
                     reinterpret_cast<const bf16_t*>(D), reinterpret_cast<const bf16_t*>(Gt), partial, g)
  if (g.mode == MMDYN_CONV) {
    if (g.f16) WGRAD_B16(MMDYN_CONV, true);
    else WGRAD_B16(MMDYN_CONV, false);
  } else {
    if (g.f16) WGRAD_B16(MMDYN_DENSE, true);
    else WGRAD_B16(MMDYN_DENSE, false);
  }
#undef WGRAD_B16
  MMDYN_LAUNCH_CHECK();
}

template <int BD, int BG, int WD, int WG, int WK>
static int launch(const float* D, const float* Gt, float* partial, WgradGeom g, hipStream_t st, bool bf16) {
  const int zblocks = g.chunks / WK;  // chunks % 4 == 0 is checked by the caller
  int rpc = ceil_div(g.rows, zblocks);
  g.rows_per_chunk = ceil_div(rpc, RK) * RK;
  dim3 grid((g.Cd / BD) * (g.Cg / BG), g.ntaps, zblocks * (g.groups > 1 ? g.groups : 1));
  size_t smem = (size_t)RK * (BD + BG) * sizeof(float);
  if constexpr (WK == 1) {
    if (g.x3 && !bf16 && g.mode != MMDYN_IM2COL3) {       // fp32 through the bf16 matrix cores (three-term split)
      smem = (size_t)3 * RK * (x3_ld<BD>::v + x3_ld<BG>::v) * 2;
      static LdsOptIn x3_dense_opt_in;
      if (g.mode == MMDYN_CONV) {
#define WGRAD_X3(PRE_)                                                                                                          \
  do {                                                                                                                          \
    static LdsOptIn opt_in_;                                                                                                    \
    if (smem > 65536)                                                                                                           \
      if (int e = opt_in_.ensure((const void*)wgrad_tn_kernel<MMDYN_CONV, BD, BG, WD, WG, WK, false, 0, true, PRE_>, (int)smem)) return e; \
    hipLaunchKernelGGL((wgrad_tn_kernel<MMDYN_CONV, BD, BG, WD, WG, WK, false, 0, true, PRE_>), grid, dim3(256), smem, st, D, Gt, partial, g); \
  } while (0)
        if (g.pre == 3) WGRAD_X3(3);
        else if (g.pre == 2) WGRAD_X3(2);
        else if (g.pre == 1) WGRAD_X3(1);
        else WGRAD_X3(0);
#undef WGRAD_X3
      } else {
        if (g.pre) return MMDYN_ERR_SHAPE;      // (operands that arrive split: convolution-level weight gradients only)
        if (smem > 65536)
          if (int e = x3_dense_opt_in.ensure((const void*)wgrad_tn_kernel<MMDYN_DENSE, BD, BG, WD, WG, WK, false, 0, true>, (int)smem)) return e;
        hipLaunchKernelGGL((wgrad_tn_kernel<MMDYN_DENSE, BD, BG, WD, WG, WK, false, 0, true>), grid, dim3(256), smem, st, D, Gt, partial, g);
      }
      MMDYN_LAUNCH_CHECK();
    }
  }
#define WGRAD_LAUNCH(M, BF, ST_)                                                                                      \
  hipLaunchKernelGGL((wgrad_tn_kernel<M, BD, BG, WD, WG, WK, BF, ST_>), grid, dim3(256), smem, st, D, Gt, partial, g)
#define WGRAD_MODE(M)                                                  \
  do {                                                                 \
    if (bf16 && g.f16 && g.d_b16 && g.g_b16) WGRAD_LAUNCH(M, true, 5); \
    else if (bf16 && g.f16 && g.d_b16) WGRAD_LAUNCH(M, true, 6);       \
    else if (bf16 && g.f16 && g.g_b16) WGRAD_LAUNCH(M, true, 7);       \
    else if (bf16 && g.d_b16 && g.g_b16) WGRAD_LAUNCH(M, true, 1);     \
    else if (bf16 && g.d_b16) WGRAD_LAUNCH(M, true, 2);                \
    else if (bf16 && g.g_b16) WGRAD_LAUNCH(M, true, 3);                \
    else if (bf16 && g.f16) WGRAD_LAUNCH(M, true, 4);                  \
    else if (bf16) WGRAD_LAUNCH(M, true, 0);                           \
    else WGRAD_LAUNCH(M, false, 0);                                    \
  } while (0)
  if (g.mode == MMDYN_CONV) WGRAD_MODE(MMDYN_CONV);
  else if (g.mode == MMDYN_DENSE) WGRAD_MODE(MMDYN_DENSE);
  else if constexpr (BG == 64) WGRAD_MODE(MMDYN_IM2COL3);     // (Cg = 64: only the 64-wide gathered tiles can occur)
  else return MMDYN_ERR_SHAPE;
#undef WGRAD_MODE
#undef WGRAD_LAUNCH
  MMDYN_LAUNCH_CHECK();
}

}  // namespace

// (LAB build: MMDYN_WGRAD_128x64=0 puts the 128 x 64 channel layers back on 64x64 tiles -- step A/B)
static bool tile_128x64() {
  const char* e = lab_env("MMDYN_WGRAD_128x64");
  return !(e && e[0] == '0');
}

// (LAB build: MMDYN_WGRAD_P3=0 keeps the register-staged kernels on plane operands -- per-shape A/B)
static bool p3_enabled() {
  const char* e = lab_env("MMDYN_WGRAD_P3");
  return !(e && e[0] == '0');
}

static int wgrad_entry(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr,
                       int Wr, int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks,
                       void* stream, bool bf16, int storage_flags = 0, int groups = 1) {
  if (!D || !Gt || !partial) return MMDYN_ERR_NULL;
  if (groups < 1 || (groups > 1 && mode != MMDYN_DENSE)) return MMDYN_ERR_SHAPE;
  if (Cd % 32 || Cg % 32 || Cd <= 0 || Cg <= 0 || chunks < 4 || chunks % 4) return MMDYN_ERR_SHAPE;
  if (mode != MMDYN_DENSE && mode != MMDYN_CONV && mode != MMDYN_IM2COL3) return MMDYN_ERR_SHAPE;
  if (mode == MMDYN_IM2COL3 && (Cg != 64 || Hi != 2 * Hr || Wi != 2 * Wr)) return MMDYN_ERR_SHAPE;
  WgradGeom g{};
  bool x3 = (storage_flags & 128) != 0;        // fp32 launch that may take the three-term split ("fp32x3" on the host side)
  const int pre = (storage_flags >> 8) & 3;    // bits 8 / 9 (with bit 7): D / Gt ARRIVE split (rows of [plane][C] bf16)
  storage_flags &= ~(128 | 256 | 512);
  if (x3 && (bf16 || storage_flags)) return MMDYN_ERR_SHAPE;
  if (pre && (!x3 || mode != MMDYN_CONV || groups > 1)) return MMDYN_ERR_SHAPE;
  if (const char* e = lab_env("MMDYN_X3_WGRAD")) x3 = e[0] == '1';      // (LAB: override)
  g.x3 = x3 && !bf16;
  g.pre = pre;
  if (pre && !g.x3) return MMDYN_ERR_SHAPE;   // (a LAB override switched the split off: plane operands cannot be served)
  g.d_b16 = (storage_flags & 2) != 0;
  g.g_b16 = (storage_flags & 4) != 0;
  g.f16 = (storage_flags & 32) != 0;
  if (storage_flags && (!bf16 || (g.g_b16 && mode == MMDYN_IM2COL3))) return MMDYN_ERR_SHAPE;
  g.mode = mode;
  g.groups = groups;
  const int64_t rows = (int64_t)Bt * Hr * Wr;
  if (groups * rows * Cd >= (1LL << 31) || groups * rows * Cg >= (1LL << 31) || (int64_t)Bt * Hi * Wi * (mode == MMDYN_IM2COL3 ? 3 : Cg) >= (1LL << 31))
    return MMDYN_ERR_RANGE;
  g.rows = (int)rows;
  g.Hr = Hr;
  g.Wr = Wr;
  g.Cd = Cd;
  g.Hi = Hi;
  g.Wi = Wi;
  g.Cg = Cg;
  g.rs = stride;
  g.ro = offset;
  g.ntaps = (mode == MMDYN_CONV) ? 16 : 1;
  g.chunks = chunks;
  hipStream_t st = (hipStream_t)stream;
  if (mode == MMDYN_IM2COL3) {     // the 3-channel layers have their own direct kernel (conv3.hip)
    const int rc = mmdyn_conv3_wgrad_try(D, Gt, partial, Bt, Hr, Wr, Cd, Hi, Wi, Cg, chunks, g.d_b16 << g.f16, st);
    if (rc != 1) return rc;
  }
  const bool d64 = (Cd % 64 == 0), g64 = (Cg % 64 == 0);
  if (g.x3 && g.pre == 3 && mode == MMDYN_CONV && p3_enabled()) {      // both operands arrive split: the plane-ring kernel (wgrad_p3.hip)
    const int rc = mmdyn_wgrad_p3_try(D, Gt, partial, g, Bt, st);
    if (rc != 1) return rc;
  }
  if (!bf16 && mode != MMDYN_IM2COL3) {
    // LAB build, MMDYN_WGRAD_WS=1: the wave-specialised LDS-DMA ring form of this GEMM (wgrad_ws.hip).  Measured per shape
    // against the kernels below (profiles/r3/ab_ws_wgrad.txt): x0.97-1.10, one launch x0.69, sum +3 % -- unlike the implicit
    // GEMM this kernel gains nothing from the ring, so the product keeps one code path and does not build it.
    const char* e = lab_env("MMDYN_WGRAD_WS");
    if (e && e[0] == '1' && groups == 1) {
      const int rc = mmdyn_wgrad_ws_try(D, Gt, partial, g, st);
      if (rc != 1) return rc;
    }
  }
  if (groups > 1 && g.d_b16 && g.g_b16) return MMDYN_ERR_SHAPE;     // (grouped: the register-staged kernel below only)
  if (bf16 && g.d_b16 && g.g_b16 && mode != MMDYN_IM2COL3) {
    // both operands bf16 in HBM: the transposing-LDS-read kernel (two waves share a 64x32 / 32x64 tile's rows)
    if (Cd % 128 == 0 && Cg % 128 == 0) return launch_b16<128, 128, 64, 64, 1>(D, Gt, partial, g, st);
    if (d64 && g64) return launch_b16<64, 64, 32, 32, 1>(D, Gt, partial, g, st);
    if (mode == MMDYN_DENSE) {           // (narrow convolution tiles stay on the four-tap kernel below: measured faster)
      if (d64) return launch_b16<64, 32, 32, 32, 2>(D, Gt, partial, g, st);
      if (g64) return launch_b16<32, 64, 32, 32, 2>(D, Gt, partial, g, st);
    }
  }
  if (Cd % 128 == 0 && Cg % 128 == 0) return launch<128, 128, 64, 64, 1>(D, Gt, partial, g, st, bf16);
  // 128 x 64 channel layers: one 128x64 tile instead of two 64x64 ones -- the gathered operand is filled once for all 128
  // output channels (21 instead of 16 flop per filled byte): x1.10-1.16 with twice the partial slabs (mmdyn_wgrad_chunks),
  // profiles/r3/ab_wgrad_tile_128x64.txt
  if (Cd % 128 == 0 && g64 && tile_128x64()) return launch<128, 64, 64, 32, 1>(D, Gt, partial, g, st, bf16);
  if (mode == MMDYN_CONV && !(d64 && g64)) {   // narrow channel tiles: four kw taps per block share the dense
    if (d64) return launch4<64, 32>(D, Gt, partial, g, st, bf16);   // operand (measured +11 %; 64x64 tiles are faster
                                                              // on the one-tap kernel, so they stay there)
    if (g64) return launch4<32, 64>(D, Gt, partial, g, st, bf16);
    return launch4<32, 32>(D, Gt, partial, g, st, bf16);
  }
  if (d64 && g64) return launch<64, 64, 32, 32, 1>(D, Gt, partial, g, st, bf16);
  if (d64) return launch<64, 32, 32, 32, 2>(D, Gt, partial, g, st, bf16);
  if (g64) return launch<32, 64, 32, 32, 2>(D, Gt, partial, g, st, bf16);
  return launch<32, 32, 32, 32, 4>(D, Gt, partial, g, st, bf16);
}

extern "C" int mmdyn_wgrad_tn(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr,
                              int Wr, int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks,
                              void* stream) {
  return wgrad_entry(D, Gt, partial, mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks, stream, false);
}

/* mixed storage: flags bit 0 = 16-bit matrix cores (required), bit 1 = D is 16-bit in HBM, bit 2 = Gt is 16-bit,
 * bit 5 = the 16-bit format is IEEE half instead of bf16 (storage and matrix cores) */
extern "C" int mmdyn_wgrad_tn_mx(const void* D, const void* Gt, float* partial, int mode, int Bt, int Hr, int Wr,
                                 int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks, int flags,
                                 void* stream) {
  return wgrad_entry((const float*)D, (const float*)Gt, partial, mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset,
                     chunks, stream, (flags & 1) != 0, flags & ~1);
}

extern "C" int mmdyn_wgrad_tn_bf16(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr,
                                   int Wr, int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks,
                                   void* stream) {
  return wgrad_entry(D, Gt, partial, mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks, stream, true);
}

/* Grouped weight gradient (DENSE): G independent problems of one shape in one launch -- group g owns rows [g*rows, (g+1)*rows)
 * of D [G*rows][Cd] and Gt [G*rows][Cg]; partial is laid out [chunks][G][Cd][Cg], so ONE mmdyn_wgrad_reduce(partial, canon,
 * chunks, 1, G*Cd, Cg, ...) produces the G gradients [G][Cd][Cg] (the heads of the visual / tactile / pose encoders,
 * vae.py:211-216, whose weights the fused engine keeps adjacent).  flags as mmdyn_wgrad_tn_mx (0 = fp32). */
extern "C" int mmdyn_wgrad_tn_grouped(const void* D, const void* Gt, float* partial, int G, int rows, int Cd, int Cg, int chunks,
                                      int flags, void* stream) {
  return wgrad_entry((const float*)D, (const float*)Gt, partial, MMDYN_DENSE, rows, 1, 1, Cd, 1, 1, Cg, 1, 0, chunks, stream,
                     (flags & 1) != 0 || (flags & 32) != 0, flags & ~1, G);
}

/* Weight gradient of the decoder's LAST layer (nn.ConvTranspose2d(32, 3, 4, 2, 1), vae.py:277) with the BatchNorm2d + Swish in
 * front of it recomputed on the operand fetch: y [G*Bg][Hr][Hr][32] is the pre-BatchNorm tensor (fp32, or 16-bit with y_b16 =
 * 1 / 2), the dense operand swish(gamma * ((y - mean[g]) * rstd[g]) + beta); Gt the NCHW logit gradient [G*Bg][3][2Hr][2Hr]
 * (im2col mode).  partial: [chunks][1][32][64] as mmdyn_wgrad_tn(MMDYN_IM2COL3).  Hr = 32, 64 or 128 (the direct kernel). */
extern "C" int mmdyn_wgrad_out3_bn(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                   const float* Gt, float* partial, int G, int Bg, int Hr, int chunks, int y_b16, void* stream) {
  if (!y || !mean || !rstd || !gamma || !beta || !Gt || !partial) return MMDYN_ERR_NULL;
  if (G <= 0 || Bg <= 0 || chunks < 1 || y_b16 < 0 || y_b16 > 2) return MMDYN_ERR_SHAPE;
  const int rc = mmdyn_conv3_wgrad_try(y, Gt, partial, G * Bg, Hr, Hr, 32, 2 * Hr, 2 * Hr, 64, chunks, y_b16, (hipStream_t)stream,
                                       mean, rstd, gamma, beta, Bg);
  return rc == 1 ? MMDYN_ERR_SHAPE : rc;
}

/* fp16 matrix cores (v_mfma_f32_32x32x8_f16), fp32 accumulate, fp32 storage: BASELINE configs[4] */
extern "C" int mmdyn_wgrad_tn_f16(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr,
                                  int Wr, int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks,
                                  void* stream) {
  return wgrad_entry(D, Gt, partial, mode, Bt, Hr, Wr, Cd, Hi, Wi, Cg, stride, offset, chunks, stream, true, 32);
}

// recommended number of partial slabs: ~768 blocks in flight, at least 128 rows per block
// b16_storage: both operands are 16-bit in HBM (the wgrad_b16 kernels: their own tiles)
// x3: the launch may take the three-term split (flag bit 7): its tiles hold three bf16 planes in LDS, so fewer blocks fit a CU
static int wgrad_chunks_impl(int mode, int rows, int Cd, int Cg, bool b16_storage, bool x3 = false) {
  if (Cd % 32 || Cg % 32 || rows <= 0) return MMDYN_ERR_SHAPE;
  int bd, bg, wk;
  if (Cd % 128 == 0 && Cg % 128 == 0) { bd = 128; bg = 128; wk = 1; }
  else if (Cd % 128 == 0 && Cg % 64 == 0 && !b16_storage && tile_128x64()) { bd = 128; bg = 64; wk = 1; }
  else if (Cd % 64 == 0 && Cg % 64 == 0) { bd = 64; bg = 64; wk = 1; }
  else if (Cd % 64 == 0) { bd = 64; bg = 32; wk = 2; }
  else if (Cg % 64 == 0) { bd = 32; bg = 64; wk = 2; }
  else { bd = 32; bg = 32; wk = 4; }
  const int taps = (mode == MMDYN_CONV) ? 16 : 1;
  long tiles = (long)(Cd / bd) * (Cg / bg) * taps;
  if (mode == MMDYN_CONV && !(Cd % 64 == 0 && Cg % 64 == 0)) {     // four-tap kernel: one block per kernel row
    tiles = (long)(Cd / bd) * (Cg / bg) * 4;
    wk = 1;
  }
  long target = 768;      // blocks in flight (whole-step sweep after the fetch fixes: 768 beats 512 / 1024 in all three precisions)
  if (x3 && mode != MMDYN_IM2COL3) {
    // the split kernels: 61 KB of LDS per 128x128 tile = two blocks per CU, so 768 blocks would run as one and a half rounds;
    // step sweep in that arithmetic (profiles/r4/step_ab_x3_wgrad_blocks.txt): 512 beats 384 / 768 / 1024 / 1536 (5.67 against
    // 5.78 ms at 768; 256 x the blocks per CU of each tile: 5.67), and the slab reduction has a third less to read
    target = 512;
  }
  if (const char* ov = lab_env("MMDYN_WGRAD_BLOCKS")) target = atol(ov);   // kernel experiments only
  if (mode == MMDYN_IM2COL3) {            // conv3_wgrad: one block (four waves, one slab) per chunk
    tiles = 1;
    wk = 1;
  }
  long z = target / tiles;
  const long zmax = rows / 128;
  if (z > zmax) z = zmax;
  if (z < 1) z = 1;
  long chunks = z * wk;
  chunks = (chunks + 3) / 4 * 4;
  return (int)chunks;
}

extern "C" int mmdyn_wgrad_chunks(int mode, int rows, int Cd, int Cg) { return wgrad_chunks_impl(mode, rows, Cd, Cg, false); }
/* flags as mmdyn_wgrad_tn_mx: the count for the kernel those storage flags select */
extern "C" int mmdyn_wgrad_chunks_mx(int mode, int rows, int Cd, int Cg, int flags) {
  if ((flags & 896) == 896 && !(flags & 1) && mode == MMDYN_CONV && p3_enabled()) {     // both operands arrive split: the plane-ring kernel's cut
    const int c = mmdyn_wgrad_p3_chunks(rows, Cd, Cg);
    if (c > 0) return c;
  }
  return wgrad_chunks_impl(mode, rows, Cd, Cg, (flags & 6) == 6, (flags & 128) != 0 && !(flags & 1));
}

extern "C" int mmdyn_wgrad_reduce(const float* partial, float* canon, int chunks, int taps, int Cd, int Cg,
                                  int cg_canon, int perm, float beta, void* stream) {
  if (!partial || !canon) return MMDYN_ERR_NULL;
  if (cg_canon > Cg || cg_canon <= 0) return MMDYN_ERR_SHAPE;
  if (perm == 1 && (Cg % 256 || Cg / 256 != 25)) return MMDYN_ERR_SHAPE;
  if (perm == 2 && (Cd % 256 || Cd / 256 != 25)) return MMDYN_ERR_SHAPE;
  int64_t slab = (int64_t)taps * Cd * Cg;
  if (chunks <= 64 && slab >= 65536) {
    hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3(ew_grid(slab >> 2)), dim3(256), 0, (hipStream_t)stream, partial,
                       canon, chunks, taps, Cd, Cg, cg_canon, perm, beta);
    MMDYN_LAUNCH_CHECK();
  }
  int64_t ntiles = (slab + 31) / 32;
  int grid = (int)(ntiles < 8192 ? ntiles : 8192);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partial, canon, chunks,
                     taps, Cd, Cg, cg_canon, perm, beta);
  MMDYN_LAUNCH_CHECK();
}
