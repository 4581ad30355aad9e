// Weight-gradient GEMM of the k4 convolutions in the fp32x3 arithmetic on operands that ARRIVE split (plane tensors: rows of
// [plane][C] bf16, the exact three-term split -- mmdyn_split_planes / the producers' plane outputs), gfx950.
//
//   partial[chunk][tap][cd][cg] = sum_{row in chunk} D[row][cd] * G[pix(row, tap)][cg]        (wgrad_tn.hip's contract and slabs)
//
// Replaces the ATen weight-gradient kernels of nn.Conv2d / nn.ConvTranspose2d on the reference path
// (/root/reference/mmdyn/pytorch/problems/problems.py:153 loss.backward() through vae.py:198-216, 264-277) for the launches it
// serves; everything else stays on wgrad_tn.hip.
//
// Structure = the plane ring of igemm_wsp3_kernel (csrc/igemm_wsp.hip) turned to this GEMM: four loader waves move 1-KiB pieces of
// both operands' planes by LDS-DMA into a three-slot ring, one K-step (32 rows) per slot; eight MFMA waves execute transposing
// fragment reads (ds_read_b64_tr_b16: the reduction index is the tile ROW) + v_mfma_f32_32x32x16_bf16 only -- six plane products
// per fragment pair, smallest first, fp32 accumulate: the terms and the order of wgrad_tn_kernel<X3>.  No register staging, no
// VALU in the K loop, one raw s_barrier per K-step, counted vmcnt.  Out-of-image / out-of-chunk rows are zero-filled by the buffer
// range check (offset 0x80000000; the dense operand's descriptor ends at the chunk's last row).
// Tile = BD channels of D x 128 "columns" of G, where the 128 columns are TAPS kw-taps x CGB channels: 1 x 128, 2 x 64 or 4 x 32 --
// narrow layers share one D tile between the taps of a kernel row (what wgrad_tn4_kernel does with four waves).
// LDS image of a plane tile: row-major [32 rows][BD or 128 channels], the 64-byte chunk index of a row XOR-ed with f(row) so that
// the four rows of a transposing read fall into four different bank groups (DMA source address and fragment read use the same f).
#include "common.h"
#include "wgrad_geom.h"

namespace {

constexpr int RK = 32;            // rows per K-step
constexpr int NLD = 4;            // loader waves
constexpr int NMW = 8;            // MFMA waves
constexpr int S = 3;              // ring slots
constexpr unsigned OOB = 0x80000000u;
constexpr int64_t MAX_BUFFER_BYTES = 0x7FFFFF00LL;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xb16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

// chunk-index swizzle of row k of a tile whose rows are ROWB bytes: 256-byte rows (four 64-byte chunks) k & 3, 128-byte rows
// (two chunks, two rows per 256-byte bank row) (k >> 1) & 1
template <int ROWB> __device__ __forceinline__ int chunk_swz(int k) { return ROWB == 256 ? (k & 3) : ((k >> 1) & 1); }

// BD: channels of the dense operand per tile (128 or 64); TAPS x CGB = 128 columns of the gathered operand
template <int BD, int TAPS, int CGB>
__global__ __launch_bounds__(64 * (NLD + NMW)) void wgrad_p3_kernel(const bf16_t* __restrict__ D, const bf16_t* __restrict__ Gt,
                                                                    float* __restrict__ partial, const WgradGeom g,
                                                                    const unsigned g_bytes) {
  static_assert(TAPS * CGB == 128 && (BD == 128 || BD == 64), "tile shapes");
  constexpr int ROWA = BD * 2, ROWBG = 256;                // bytes per plane row of the two tiles
  constexpr int PLA = RK * ROWA, PLB = RK * ROWBG;         // bytes per plane
  constexpr int SLOT = 3 * (PLA + PLB);
  constexpr int RPA = 1024 / ROWA, RPB = 4;                // rows per DMA piece
  constexpr int NBA = RK / RPA, NBB = RK / RPB;            // row blocks per K-step
  static_assert(NBA % NLD == 0 && NBB % NLD == 0, "row blocks split evenly over the loader waves");
  constexpr int RAL = NBA / NLD, RBL = NBB / NLD, PPL = 3 * (RAL + RBL);
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int WD = BD / 2, DT = WD / 32;                 // MFMA wave tile: WD x 32 (two wave rows x four wave columns)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block order (speed only, as wgrad_tn_kernel): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
  // All blocks of one row chunk -- its channel tiles x tap groups, which re-read the same rows of D and Gt -- get linear ids that
  // are equal modulo 8, so a chunk's rows pass through ONE L2 instead of up to eight.
  const int tiles_g = TAPS == 1 ? g.Cg / 128 : 1;
  constexpr int NTG = 16 / TAPS;
  const int per_chunk = (g.Cd / BD) * tiles_g * NTG;
  // (chunk counts that are no multiple of 8 -- small launches -- keep the plain order: the fold would leave XCDs idle)
  const int L = blockIdx.x, xcd = L & 7, jj = L >> 3;
  const bool fold = (g.chunks & 7) == 0;
  const int chunk = fold ? xcd + 8 * (jj / per_chunk) : L / per_chunk, inner = fold ? jj % per_chunk : L % per_chunk;
  const int tgrp = inner % NTG, tile = inner / NTG;
  const int td = tile / tiles_g, tg = tile - td * tiles_g;
  const int cd0 = td * BD, cg0 = TAPS == 1 ? tg * 128 : 0;
  const int kh = TAPS == 1 ? (tgrp >> 2) : (TAPS == 2 ? (tgrp >> 1) : tgrp);
  const int kw0 = TAPS == 1 ? (tgrp & 3) : (TAPS == 2 ? 2 * (tgrp & 1) : 0);
  const int row_begin = chunk * g.rows_per_chunk;
  const int row_end = min(g.rows, row_begin + g.rows_per_chunk);
  if (row_begin >= row_end) {                              // (an empty trailing chunk still owns a slab: zeros)
    if (wave >= NLD) {
      const int mw = wave - NLD, wm = mw >> 2, wn = mw & 3, h = lane >> 5, cl = lane & 31;
      const int n = wn * 32 + cl, tap = kh * 4 + kw0 + n / CGB, cg = cg0 + n % CGB;
      for (int a = 0; a < DT; ++a)
        for (int e = 0; e < 16; ++e) {
          const int cd = cd0 + wm * WD + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          partial[((size_t)(chunk * g.ntaps + tap) * g.Cd + cd) * g.Cg + cg] = 0.f;
        }
    }
    return;
  }
  const int nk = (row_end - row_begin + RK - 1) / RK;

  if (wave < NLD) {
    // ===================================== loader wave =====================================
    // the dense operand's descriptor ends at the chunk's last row: rows past it read as zeros
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)D, 0, (int)((unsigned)row_end * (unsigned)(g.Cd * 6)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)Gt, 0, (int)g_bytes, 0x00020000);
    constexpr int LPRA = 64 / RPA, LPRB = 64 / RPB;        // lanes per row of a piece
    const int HWr = g.Hr * g.Wr;
    const float inv_hw = 1.0f / (float)HWr, inv_w = 1.0f / (float)g.Wr;
    auto fdiv = [](int n, int d, float inv, int& q, int& r) {
      q = (int)((float)n * inv);
      r = n - q * d;
      if (r < 0) { q -= 1; r += d; }
      if (r >= d) { q += 1; r -= d; }
    };
    // dense operand: per-lane offset of (row block i, this lane's row and granule) relative to row_begin; the K-step advances a
    // scalar offset
    unsigned voffA[RAL];
#pragma unroll
    for (int i = 0; i < RAL; ++i) {
      const int k = (wave + NLD * i) * RPA + lane / LPRA;                    // row inside the K-step
      const int gs = (lane % LPRA) ^ (chunk_swz<ROWA>(k) << 2);              // source granule (8 channels) stored at this lane's position
      voffA[i] = (unsigned)((row_begin + k) * (g.Cd * 6) + (cd0 + gs * 8) * 2);
    }
    const unsigned plA = (unsigned)(g.Cd * 2), plB = (unsigned)(g.Cg * 2);
    auto issue = [&](int ks) {
      char* slot = smem + (ks % S) * SLOT;
      const unsigned soA = (unsigned)ks * (unsigned)(RK * g.Cd * 6);
#pragma unroll
      for (int i = 0; i < RAL; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(rsD, slot + p * PLA + (wave + NLD * i) * 1024, voffA[i], soA + (unsigned)p * plA);
#pragma unroll
      for (int j = 0; j < RBL; ++j) {
        const int k = (wave + NLD * j) * RPB + lane / LPRB;
        const int gs = (lane % LPRB) ^ (chunk_swz<ROWBG>(k) << 2);
        const int tl = (gs * 8) / CGB, co = (gs * 8) % CGB;                  // tap of the kernel row and channel offset of this granule
        const int row = row_begin + ks * RK + k;
        int bb, pp, rr, cc;
        fdiv(row, HWr, inv_hw, bb, pp);
        fdiv(pp, g.Wr, inv_w, rr, cc);
        const int y = rr * g.rs + g.ro + kh, x = cc * g.rs + g.ro + kw0 + tl;
        const bool ok = (row < row_end) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
        const unsigned voff = ok ? (unsigned)(((bb * g.Hi + y) * g.Wi + x) * (g.Cg * 6) + (cg0 + co) * 2) : OOB;
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(rsG, slot + 3 * PLA + p * PLB + (wave + NLD * j) * 1024, voff, (unsigned)p * plB);
      }
    };
    for (int ks = 0; ks < S - 1 && ks < nk; ++ks) issue(ks);
    for (int k = 0; k < nk; ++k) {
      if (k + S - 1 <= nk) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>();
      ring_barrier();                                      // slot k is complete; slot k-1 has been read by every MFMA wave
      if (k + S - 1 < nk) issue(k + S - 1);
    }
    return;
  }

  // ===================================== MFMA waves =====================================
  const int mw = wave - NLD;
  const int wm = mw >> 2, wn = mw & 3;
  f32x16 acc[DT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  // transposed-read lane roles (wgrad_tn_kernel<X3>): group gq = lane >> 4 (columns 16 * (gq & 1) .., k half gq >> 1), lane 4q + p of
  // the group supplies the address of row q, columns 4p .. 4p + 3 of its block
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int trow = 8 * (gq >> 1) + tq, tcol = 16 * (gq & 1) + 4 * tp;
  // byte offset of (row trow [+4] [+16 kc], column col0 + tcol) inside a plane tile; the swizzle term only depends on tq
  auto off = [&](int rowb, int col0, int swz) { return trow * rowb + (((col0 >> 5) ^ swz) * 64) + tcol * 2; };
  int offA[DT];
#pragma unroll
  for (int a = 0; a < DT; ++a) offA[a] = off(ROWA, wm * WD + a * 32, chunk_swz<ROWA>(tq));
  const int offB = off(ROWBG, wn * 32, chunk_swz<ROWBG>(tq));
  auto frag = [&](const char* tile, int rowb, int o, int k0) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(tile + k0 * rowb + o));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(tile + (k0 + 4) * rowb + o));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(xb16x8, v);
  };
  for (int k = 0; k < nk; ++k) {
    ring_barrier();
    const char* sl = smem + (k % S) * SLOT;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {                       // 16 rows per MFMA
      xb16x8 pa[3][DT], pb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int a = 0; a < DT; ++a) pa[p][a] = frag(sl + p * PLA, ROWA, offA[a], kc * 16);
        pb[p] = frag(sl + 3 * PLA + p * PLB, ROWBG, offB, kc * 16);
      }
      constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (plane of D, plane of G), smallest first
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int a = 0; a < DT; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[order[t][0]][a], pb[order[t][1]], acc[a], 0, 0, 0);
    }
  }
  const int h = lane >> 5, cl = lane & 31;
  const int n = wn * 32 + cl, tap = kh * 4 + kw0 + n / CGB, cg = cg0 + n % CGB;
  float* out = partial + ((size_t)(chunk * g.ntaps + tap) * g.Cd) * g.Cg + cg;
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int cd = cd0 + wm * WD + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      out[(size_t)cd * g.Cg] = acc[a][e];
    }
}

// tile configuration of a shape: 0 = not served
static int p3_cfg(int Cd, int Cg) {
  if (Cd % 128 == 0 && Cg % 128 == 0) return 1;          // 128 x (1 tap x 128)
  if (Cd % 128 == 0 && Cg == 64) return 2;               // 128 x (2 taps x 64)
  if (Cd % 64 == 0 && Cg == 32) return 3;                // 64 x (4 taps x 32)
  return 0;
}

template <int BD, int TAPS, int CGB>
static int p3_launch(const bf16_t* D, const bf16_t* Gt, float* partial, WgradGeom g, unsigned g_bytes, hipStream_t st) {
  g.rows_per_chunk = ceil_div(ceil_div(g.rows, g.chunks), RK) * RK;
  const int tiles = (g.Cd / BD) * (TAPS == 1 ? g.Cg / 128 : 1);
  dim3 grid(g.chunks * tiles * (16 / TAPS));             // (chunk, tile, tap group) folded XCD-aware: see the kernel
  const size_t smem = (size_t)S * 3 * (RK * BD * 2 + RK * 256);
  static LdsOptIn opt_in;
  if (int e = opt_in.ensure((const void*)wgrad_p3_kernel<BD, TAPS, CGB>, (int)smem)) return e;
  hipLaunchKernelGGL((wgrad_p3_kernel<BD, TAPS, CGB>), grid, dim3(64 * (NLD + NMW)), smem, st, D, Gt, partial, g, g_bytes);
  MMDYN_LAUNCH_CHECK();
}

}  // namespace

// recommended number of partial slabs of the plane-ring weight gradient (0: the shape is not served): one block per CU -- a block
// holds 108-144 KB of LDS -- so about 256 blocks, at least 1024 rows (32 K-steps) each; a multiple of four like wgrad_tn's
int mmdyn_wgrad_p3_chunks(int rows, int Cd, int Cg) {
  const int cfg = p3_cfg(Cd, Cg);
  if (!cfg || rows <= 0) return 0;
  const int bd = cfg == 3 ? 64 : 128, taps = cfg == 1 ? 1 : (cfg == 2 ? 2 : 4);
  const long blocks = (long)(Cd / bd) * (cfg == 1 ? Cg / 128 : 1) * (16 / taps);
  long z = 256 / blocks, zmax = rows / 1024;
  if (z > zmax) z = zmax;
  if (z < 4) z = 4;
  return (int)((z + 3) / 4 * 4);
}

// both operands arrive split; returns 1 when the shape is not served (the caller then takes wgrad_tn_kernel<X3, PRE = 3>)
int mmdyn_wgrad_p3_try(const void* D, const void* Gt, float* partial, const WgradGeom& g, int Bt, hipStream_t st) {
  const int cfg = p3_cfg(g.Cd, g.Cg);
  if (!cfg || g.mode != MMDYN_CONV || g.groups > 1 || g.rows >= (1 << 23)) return 1;
  const int64_t d_bytes = (int64_t)g.rows * g.Cd * 6, g_bytes = (int64_t)Bt * g.Hi * g.Wi * g.Cg * 6;
  if (d_bytes >= MAX_BUFFER_BYTES || g_bytes >= MAX_BUFFER_BYTES) return 1;
  const bf16_t* Dp = reinterpret_cast<const bf16_t*>(D);
  const bf16_t* Gp = reinterpret_cast<const bf16_t*>(Gt);
  if (cfg == 1) return p3_launch<128, 1, 128>(Dp, Gp, partial, g, (unsigned)g_bytes, st);
  if (cfg == 2) return p3_launch<128, 2, 64>(Dp, Gp, partial, g, (unsigned)g_bytes, st);
  return p3_launch<64, 4, 32>(Dp, Gp, partial, g, (unsigned)g_bytes, st);
}
