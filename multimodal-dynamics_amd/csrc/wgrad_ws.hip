// Wave-specialised fp32 weight-gradient GEMM, TN form (v_mfma_f32_32x32x2_f32, gfx950): LDS-DMA ring + loader waves.
//
//   partial[chunk][tap][cd][cg] = sum_{row in chunk} D[row][cd] * G_tap[row][cg]      (same contract as wgrad_tn.hip)
//
// Replaces the ATen weight-gradient kernels of nn.Conv2d / nn.ConvTranspose2d / nn.Linear used by loss.backward() on the
// reference path (/root/reference/mmdyn/pytorch/problems/problems.py:153) for the fp32 launches it serves (channel tiles
// of 128x128 or 64x64); every other launch keeps wgrad_tn.hip.  Same structure as igemm_ws.hip (see there and
// docs/LAB_NOTES.md D): two loader waves issue buffer_load_dwordx4 ... lds into a ring of S K-step slots S-1 steps ahead and
// retire a slot with a counted s_waitcnt vmcnt(N); four MFMA waves execute ds_read_b32 + v_mfma only; one raw s_barrier per
// K-step; rows past the chunk end or outside the image carry an out-of-range buffer offset and arrive as zeros.
// Here the reduction index is the tile ROW: a slot is [RK rows][BD floats] + [RK rows][BG floats], row-major, unpadded (one
// DMA piece = 1 KiB = 2 or 4 whole rows).  The fragment reads are ds_read_b32 along the channel axis -- lane (i = l&31,
// k = l>>5) reads row 2kk+k, channel c0+i: 32 consecutive floats per half-wave, conflict-free without a swizzle.
#include "common.h"
#include "wgrad_geom.h"

namespace {

constexpr int NL = 2;                      // loader waves per block
constexpr unsigned OOB = 0x80000000u;      // see igemm_ws.hip
constexpr int64_t MAX_BUFFER_BYTES = 0x7FFFFF00LL;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ring_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}

// RK rows per K-step; 4 MFMA waves as (BD / WD) x (BG / WG)
template <int MODE, int BD, int BG, int WD, int WG, int RK, int S>
__global__ __launch_bounds__(64 * (4 + NL)) void wgrad_ws_kernel(const float* __restrict__ D, const float* __restrict__ Gt,
                                                                 float* __restrict__ partial, const WgradGeom g,
                                                                 const unsigned d_bytes, const unsigned g_bytes) {
  constexpr int DT = WD / 32, GT = WG / 32;
  constexpr int WAVES_G = BG / WG;
  static_assert((BD / WD) * WAVES_G == 4, "4 MFMA waves per block");
  constexpr int RBD = BD * 4, RBG = BG * 4;                    // bytes per tile row
  constexpr int RPD = 1024 / RBD, RPG = 1024 / RBG;            // rows per DMA piece
  constexpr int PD = RK / RPD, PG = RK / RPG;                  // pieces per K-step
  static_assert(PD % NL == 0 && PG % NL == 0, "pieces split evenly over the loader waves");
  constexpr int PDL = PD / NL, PGL = PG / NL, PPL = PDL + PGL;
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int SLOT = RK * (RBD + RBG);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block order (speed only; as wgrad_b16_kernel): all blocks of one row chunk -- its channel tiles x taps, which
  // re-read the same rows of D and Gt -- get linear ids that are equal modulo 8
  const int tiles_g = g.Cg / BG;
  const int per_chunk = (g.Cd / BD) * tiles_g * g.ntaps;
  const int L = blockIdx.x, xcd = L & 7, jj = L >> 3;
  const int chunk = xcd + 8 * (jj / per_chunk), inner = jj % per_chunk;
  if (chunk >= g.chunks) return;
  const int tap = inner % g.ntaps, tile = inner / g.ntaps;
  const int td = tile / tiles_g, tg = tile - td * tiles_g;
  const int cd0 = td * BD, cg0 = tg * BG;
  const int dh = (MODE == MMDYN_CONV) ? (tap >> 2) : 0;
  const int dw = (MODE == MMDYN_CONV) ? (tap & 3) : 0;
  const int HWr = g.Hr * g.Wr;
  const int row_begin = chunk * g.rows_per_chunk;
  const int row_end = min(g.rows, row_begin + g.rows_per_chunk);
  const int nsteps = row_begin < row_end ? (row_end - row_begin + RK - 1) / RK : 0;

  if (wave < NL) {
    // ===================================== loader wave =====================================
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)D, 0, (int)d_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)Gt, 0, (int)g_bytes, 0x00020000);
    const float inv_hw = 1.0f / (float)HWr, inv_w = 1.0f / (float)g.Wr;
    auto fdiv = [](int n, int d, float inv, int& q, int& r) {      // pixel decode by float reciprocal (rows < 2^23)
      q = (int)((float)n * inv);
      r = n - q * d;
      if (r < 0) { q -= 1; r += d; }
      if (r >= d) { q += 1; r -= d; }
    };
    // this lane's row inside a piece and its 16-byte column slot
    const int drow = lane / (RBD / 16), dcol = (lane % (RBD / 16)) * 16;
    const int grow = lane / (RBG / 16), gcol = (lane % (RBG / 16)) * 16;
    int issued = 0;
    auto issue = [&]() {
      char* slot = smem + (issued % S) * SLOT;
      const int r0 = row_begin + issued * RK;
#pragma unroll
      for (int i = 0; i < PDL; ++i) {
        const int p = wave + NL * i;
        const int row = r0 + p * RPD + drow;
        const unsigned voff = row < row_end ? (unsigned)row * (unsigned)(g.Cd * 4) + (unsigned)(cd0 * 4 + dcol) : OOB;
        dma16(rsD, slot + p * 1024, voff);
      }
#pragma unroll
      for (int i = 0; i < PGL; ++i) {
        const int p = wave + NL * i;
        const int row = r0 + p * RPG + grow;
        bool ok = row < row_end;
        int pix = row;
        if constexpr (MODE == MMDYN_CONV) {
          int bb, pp, rr, cc;
          fdiv(row, HWr, inv_hw, bb, pp);
          fdiv(pp, g.Wr, inv_w, rr, cc);
          const int y = rr * g.rs + g.ro + dh, x = cc * g.rs + g.ro + dw;
          ok = ok & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
          pix = (bb * g.Hi + y) * g.Wi + x;
        }
        const unsigned voff = ok ? (unsigned)pix * (unsigned)(g.Cg * 4) + (unsigned)(cg0 * 4 + gcol) : OOB;
        dma16(rsG, slot + RK * RBD + p * 1024, voff);
      }
      ++issued;
    };
    for (int k = 0; k < S - 1 && k < nsteps; ++k) issue();
    for (int k = 0; k < nsteps; ++k) {
      if (k + S - 1 <= nsteps) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>();
      ring_barrier();                                    // slot k is complete; slot k-1 has been read by every MFMA wave
      if (k + S - 1 < nsteps) issue();
    }
    return;
  }

  // ===================================== MFMA waves =====================================
  const int mw = wave - NL;
  const int wd = mw / WAVES_G, wg = mw - wd * WAVES_G;
  const int h = lane >> 5, cl = lane & 31;
  f32x16 acc[DT][GT];
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < GT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int dbase = h * RBD + (wd * WD + cl) * 4, gbase = RK * RBD + h * RBG + (wg * WG + cl) * 4;
  for (int k = 0; k < nsteps; ++k) {
    ring_barrier();
    const char* sl = smem + (k % S) * SLOT;
#pragma unroll
    for (int kk = 0; kk < RK / 2; ++kk) {
      float af[DT], bf[GT];
#pragma unroll
      for (int a = 0; a < DT; ++a) af[a] = *reinterpret_cast<const float*>(sl + dbase + 2 * kk * RBD + a * 128);
#pragma unroll
      for (int b = 0; b < GT; ++b) bf[b] = *reinterpret_cast<const float*>(sl + gbase + 2 * kk * RBG + b * 128);
#pragma unroll
      for (int a = 0; a < DT; ++a)
#pragma unroll
        for (int b = 0; b < GT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }

  float* out = partial + ((size_t)(chunk * g.ntaps + tap) * g.Cd) * g.Cg;
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

template <int MODE, int BD, int BG, int WD, int WG, int RK, int S>
static int ws_launch(const float* D, const float* Gt, float* partial, WgradGeom g, unsigned d_bytes, unsigned g_bytes,
                     hipStream_t st) {
  const int rpc = ceil_div(g.rows, g.chunks);
  g.rows_per_chunk = ceil_div(rpc, RK) * RK;
  dim3 grid((unsigned)((g.Cd / BD) * (g.Cg / BG) * g.ntaps) * (unsigned)((g.chunks + 7) / 8 * 8));
  const size_t smem = (size_t)S * RK * (BD + BG) * 4;
  static LdsOptIn lds_opt_in;
  if (int e = lds_opt_in.ensure((const void*)wgrad_ws_kernel<MODE, BD, BG, WD, WG, RK, S>, (int)smem)) return e;
  hipLaunchKernelGGL((wgrad_ws_kernel<MODE, BD, BG, WD, WG, RK, S>), grid, dim3(64 * (4 + NL)), smem, st, D, Gt, partial, g,
                     d_bytes, g_bytes);
  MMDYN_LAUNCH_CHECK();
}

}  // namespace

int mmdyn_wgrad_ws_try(const float* D, const float* Gt, float* partial, const WgradGeom& g, hipStream_t st) {
  if (g.mode != MMDYN_DENSE && g.mode != MMDYN_CONV) return 1;
  const bool t128 = g.Cd % 128 == 0 && g.Cg % 128 == 0, t64 = g.Cd % 64 == 0 && g.Cg % 64 == 0;
  if (!t128 && !t64) return 1;
  const int64_t d_bytes = (int64_t)g.rows * g.Cd * 4;
  const int64_t gpix = g.mode == MMDYN_DENSE ? (int64_t)g.rows : (int64_t)(g.rows / (g.Hr * g.Wr)) * g.Hi * g.Wi;
  const int64_t g_bytes = gpix * g.Cg * 4;
  if (d_bytes >= MAX_BUFFER_BYTES || g_bytes >= MAX_BUFFER_BYTES) return 1;
  // 16 KB slots in both shapes (16 DMA pieces per K-step, 48 KB of LDS per block: three blocks per CU)
  if (t128) {
    if (g.mode == MMDYN_CONV)
      return ws_launch<MMDYN_CONV, 128, 128, 64, 64, 16, 3>(D, Gt, partial, g, (unsigned)d_bytes, (unsigned)g_bytes, st);
    return ws_launch<MMDYN_DENSE, 128, 128, 64, 64, 16, 3>(D, Gt, partial, g, (unsigned)d_bytes, (unsigned)g_bytes, st);
  }
  if (g.mode == MMDYN_CONV)
    return ws_launch<MMDYN_CONV, 64, 64, 32, 32, 32, 3>(D, Gt, partial, g, (unsigned)d_bytes, (unsigned)g_bytes, st);
  return ws_launch<MMDYN_DENSE, 64, 64, 32, 32, 32, 3>(D, Gt, partial, g, (unsigned)d_bytes, (unsigned)g_bytes, st);
}
