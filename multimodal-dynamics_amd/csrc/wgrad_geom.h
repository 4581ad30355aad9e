// Geometry of one weight-gradient launch, shared by wgrad_tn.hip (register-staged kernels) and wgrad_ws.hip
// (wave-specialised LDS-DMA ring kernels).
#pragma once
#include "common.h"

struct WgradGeom {
  int d_b16, g_b16;  // bf16 activation storage (bf16 matrix-core variants only): D / Gt are bf16 in HBM
  int f16;           // the 16-bit format is IEEE half instead of bf16: operands, and D / Gt where d_b16 / g_b16 say 16-bit
  int mode;  // MMDYN_DENSE or MMDYN_CONV
  int rows;  // Bt*Hr*Wr
  int Hr, Wr, Cd;
  int Hi, Wi, Cg;
  int rs, ro;
  int ntaps, chunks, rows_per_chunk;
  // Grouped launch (mmdyn_wgrad_tn_grouped, DENSE only): `groups` independent problems whose rows follow one another in D
  // and Gt (group g = rows [g*rows, (g+1)*rows)); each is cut into `chunks` slabs and the slabs are laid out
  // [slab][group][Cd][Cg], so that ONE mmdyn_wgrad_reduce over Cd' = groups*Cd sums them all.  0 / 1 = a plain launch.
  int groups;
  int x3;      // fp32 launch on the bf16 matrix cores through the exact three-term operand split (wgrad_tn_kernel X3)
  int pre;     // x3: bit 0 = D, bit 1 = Gt arrives already split (rows of [plane][C] bf16) -- wgrad_tn_kernel / wgrad_tn4_kernel PRE
};

// wgrad_ws.hip (LAB build only: measured no faster than wgrad_tn.hip, see wgrad_entry): fp32 weight-gradient GEMM with loader
// waves + LDS-DMA ring.  Returns MMDYN_OK / an error code, or 1 when the launch is not served.  Same partial-slab layout.
#ifdef MMDYN_LAB
int mmdyn_wgrad_ws_try(const float* D, const float* Gt, float* partial, const WgradGeom& g, hipStream_t st);
#else
static inline int mmdyn_wgrad_ws_try(const float*, const float*, float*, const WgradGeom&, hipStream_t) { return 1; }
#endif

// wgrad_p3.hip: the plane-ring form for convolution-level weight gradients whose two operands arrive split (LDS-DMA ring + MFMA
// waves, no register staging).  try: MMDYN_OK / an error code, or 1 when the shape is not served; chunks: its slab count (0: not
// served).
int mmdyn_wgrad_p3_try(const void* D, const void* Gt, float* partial, const WgradGeom& g, int Bt, hipStream_t st);
int mmdyn_wgrad_p3_chunks(int rows, int Cd, int Cg);
