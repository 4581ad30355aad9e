// Geometry of one implicit-GEMM launch, shared by the LDS-tiled kernels (igemm_nt.hip) and the wave-independent
// direct-fragment kernels (igemm_d16.hip).
#pragma once
#include "common.h"
#include <cstdio>

struct IgemmGeom {
  int mode;            // MMDYN_DENSE / MMDYN_CONV / MMDYN_TCONV_S2P1
  int G, Bg;           // groups, samples per group
  int Hr, Wr;          // row grid per sample (rows per sample = Hr*Wr)
  int Hi, Wi, Cin;     // gathered operand
  int Ho, Wo, N, ldc;  // output pixel grid, channels, row stride
  int rs, ro;          // input base of a row: y0 = r*rs + ro
  int os;              // output pixel of a row: (r*os + ph, c*os + pw)
  int ntaps, nclasses, splitk;
  int act, has_bias, want_stats, want_act_out;
  int rows_total;      // G*Bg*Hr*Wr (split-K workspace stride)
  int tiles_per_group; // ceil(Bg*Hr*Wr / BM); TCONV_S1P0: Ho*Wo*ceil(Bg/BM) (tiles never straddle an output pixel)
  int tiles_per_pixel; // TCONV_S1P0 only: ceil(Bg/BM)
  int s1p0_split;      // TCONV_S1P0 only: 1 = a block walks the four pixels of its quad, 2 = two blocks share the walk
  // optional BatchNorm+Swish backward epilogue (input-gradient launches): the tile of dL/d(activation) is turned
  // into du = da * swish'(gamma*xhat+beta) with xhat from the layer's saved pre-BN output, written to C, and the
  // per-tile column sums (du, du*xhat) go to `stats` -- the reduction pass of the BatchNorm backward disappears
  const float* bn_y;
  const float* bn_mean;   // [G][N]
  const float* bn_rstd;   // [G][N]
  const float* bn_gamma;  // [N]
  const float* bn_beta;   // [N]
  // bn_mean == nullptr with bn_y set: the plain ACTIVATION backward epilogue -- C = acc * act'(u), u = bn_y (the layer's
  // saved pre-activation at the C positions), act' chosen by bwd_act (MMDYN_ACT_SWISH / MMDYN_ACT_RELU); no statistics.
  // (with bn_mean set bwd_act is MMDYN_ACT_SWISH: the BatchNorm + Swish backward above)
  int bwd_act;
  // bf16 activation storage (bf16 matrix-core variants only): which of the activation tensors are bf16 in HBM
  int a_b16, c_b16, bny_b16;
  int cact_b16;   // the second (activated) output is bf16 while C itself is fp32
  int cact_planes;  // fp32x3 plane launches: the second (activated) output is a PLANE tensor, rows of [plane][cact_planes] bf16
                    // (cact_planes divides N, ldc == N); 0: a plain tensor
  int f16;     // the 16-bit format is IEEE half instead of bf16 (v_mfma_*_f16): operands, and every tensor the *_b16 fields mark
  int b_b16;   // packed weights are bf16 (written so by the pack kernels in the bf16 modes: half the L2 -> LDS traffic)
  // Grouped launch (mmdyn_igemm_nt_grouped): every group multiplies its OWN weights -- group grp reads Bp + grp * b_group_stride
  // and bias + grp * bias_group_stride (elements; 0 = one weight matrix shared by all groups, the BatchNorm-group form)
  int b_group_stride, bias_group_stride;
  int x3;          // fp32 launch on the bf16 matrix cores through the exact three-term operand split (igemm_nt.hip, X3)
  int tap_order;   // ring kernels, stride-2 CONV: the four taps of one input-pixel class back to back (0 = raster order)
  // persistent stream-K kernels (igemm_wsp.hip): arrival words of the split tiles, MMDYN_IGEMM_FLAG_WORDS zeroed uint32 the launch
  // leaves zero -- the tiles are then finished inside the launch; nullptr: slabs + the fix-up launch
  unsigned* flags;
};

// igemm_d16.hip: fp32 implicit GEMM on v_mfma_f32_16x16x4_f32 with operand fragments loaded straight from global
// memory (no LDS, no block barrier).  Returns MMDYN_OK / an error code, or 1 when the shape is not served (the caller
// then takes the LDS-tiled kernel).  d16_stat_tiles: number of BatchNorm partial-sum tiles per group it writes, 0 when
// the shape is not served.
// LAB build only (igemm_d16.hip is not part of the product library: measured slower inside the two-lane step, LAB_NOTES A.a).
#ifdef MMDYN_LAB
int mmdyn_igemm_d16_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats,
                        float* ws, IgemmGeom g, int stride, int offset, hipStream_t st);
int mmdyn_igemm_d16_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
#else
static inline int mmdyn_igemm_d16_try(const float*, const float*, const float*, float*, float*, float*, float*, const IgemmGeom&,
                                      int, int, hipStream_t) { return 1; }
static inline int mmdyn_igemm_d16_stat_tiles(int, int, int, int, int, int, int, int, int) { return 0; }
#endif

// tconv_patch.hip: patch-resident k4 s2 p1 transposed convolution (fp32, 16x16x64 -> 32x32x32).  Same protocol as the d16 hooks.
int mmdyn_tconv_patch_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                          const IgemmGeom& g, hipStream_t st);
int mmdyn_tconv_patch_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
// ... on operands that arrive split (fp32x3 plane rows): the shapes served, and the launch (1: not served)
bool mmdyn_tconv_patch_p3_serves(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
int mmdyn_tconv_patch_p3_try(const void* A, const void* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                             const IgemmGeom& g, hipStream_t st);

// igemm_ws.hip: wave-specialised fp32 implicit GEMM (loader waves + LDS-DMA ring).  Same protocol as the hooks above.
int mmdyn_igemm_ws_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                       const IgemmGeom& g, bool bf16_ops, hipStream_t st);
int mmdyn_igemm_ws_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);

// igemm_wsp.hip: the persistent, stream-K-scheduled form of the ring kernel for the large launches.  Same protocol as the
// hooks above; `slabs` is the workspace for the pieces of split tiles (mmdyn_igemm_wsp_slab_bytes; may be null when that is 0).
int mmdyn_igemm_wsp_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                        const IgemmGeom& g, bool bf16_ops, hipStream_t st);
int mmdyn_igemm_wsp_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, bool b16);
// ... with BOTH operands arriving as the three-plane bf16 split of fp32 tensors (rows of [plane][Cin]); same protocol
int mmdyn_igemm_wsp3_try(const void* A, const void* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                         const IgemmGeom& g, hipStream_t st);
bool mmdyn_igemm_wsp3_serves(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
int mmdyn_igemm_wsp3_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
int64_t mmdyn_igemm_wsp3_slab_bytes(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
int64_t mmdyn_igemm_wsp_slab_bytes(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, bool b16);

// tile choice shared by the launcher and mmdyn_igemm_stat_tiles.  Measured on MI355X over every shape of the
// bs=256 step (tests/microbench/sweep_tiles.py): the 64x64 tile (more resident blocks per CU to hide the
// single-stage fetch latency) wins or ties everywhere except long-K problems that still fill the chip with
// 128x128 tiles; N == 32 (mod 64) takes 128x32.
// (rows_per_group, G) describe the row segments a tile may not straddle: groups, or (group, output pixel) pairs
static inline void pick_tile(int N, int rows_per_group, int G, int ncls, int splitk, int ksteps, int* bm, int* bn) {
  if (N % 64) {
    *bm = 128;
    *bn = 32;
  } else {
    *bm = 64;
    *bn = 64;
    if (N % 128 == 0 && splitk == 1 && ksteps >= 32 &&
        (long)G * ceil_div(rows_per_group, 128) * (N / 128) * ncls >= 512) {
      *bm = 128;
      *bn = 128;
    }
  }
  if (const char* ov = lab_env("MMDYN_IGEMM_TILE")) {   // kernel experiments only
    int a = 0, b = 0;
    if (sscanf(ov, "%d,%d", &a, &b) == 2 && N % b == 0) {
      *bm = a;
      *bn = b;
    }
  }
}

