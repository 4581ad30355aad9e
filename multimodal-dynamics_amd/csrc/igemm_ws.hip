// Wave-specialised fp32 implicit GEMM (v_mfma_f32_16x16x4_f32, gfx950): LDS-DMA operand ring + dedicated loader waves.
//
//   C[row][n] = sum_{tap, ci} A_tap[row][ci] * Bp[widx(tap)][n][ci]        (same contract as igemm_nt.hip)
//
// Replaces the ATen kernels behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear forward and input-gradient on the
// reference path (/root/reference/mmdyn/pytorch/models/vae.py:198-216, 264-277) for the fp32 launches it serves; every
// other launch keeps the register-staged kernel of igemm_nt.hip.
//
// Why a second structure (profiles/r3/ws_ring_microbench_*.txt; docs/LAB_NOTES.md D): in the register-staged kernel every
// wave gathers, loads, waits, writes LDS and multiplies; its one exposed resource is the operand fetch (+20 % without it).
// Here a block is TWO loader waves + 4 (or 8) MFMA waves:
//   * loader waves own the gather arithmetic.  They issue buffer_load_dwordx4 ... lds (LDS-DMA: no VGPR round trip, no
//     ds_write) into a ring of S K-step slots, S-1 K-steps ahead of the matrix waves, and retire a slot with a COUNTED
//     s_waitcnt vmcnt(N) -- the queue is never drained inside the loop.  Out-of-image rows carry an out-of-range
//     buffer offset: the hardware range check writes zeros (measured: tests/microbench/oob_lds_dma.hip), so the K loop
//     has no select and no zero page;
//   * MFMA waves execute ds_read_b128 + v_mfma only: no VMEM instruction and no vmcnt wait in their stream, ~60 VGPRs;
//   * ONE raw s_barrier per K-step hands slot k to the matrix waves and slot k-1 back to the loaders.
// LDS image of a slot: [BM + BN rows][32 floats], unpadded (a DMA piece is 1 KiB, lane-linear: 8 rows x 128 B).  Bank
// conflicts of the fragment reads are removed by XOR-ing the 16-byte slot index with f(row) = (row >> 1) & 7 on BOTH
// sides: in the per-lane SOURCE address of the DMA and in the ds_read_b128 address (a swizzled destination is impossible).
// Stand-alone dense GEMM, MI355X: 128x64 tile 106-123 TFLOP/s where the register-staged 64x64 kernel does 82-101.
#include "common.h"
#include "igemm_geom.h"
#include <cstdio>
#include <type_traits>

namespace {

constexpr int BK = 32;          // K-step: 32 channels = one 128-byte row segment
constexpr int RB = BK * 4;      // bytes per tile row
constexpr int RPP = 8;          // rows per DMA piece (1 KiB per wave instruction)
constexpr int NL = 2;           // loader waves per block
// voffset of a row that must read as zeros: beyond any num_records we accept (< 2 GiB), and far enough from 2^32 that adding
// the scalar offset of a K-step (which the hardware includes in its range check) cannot wrap around into the buffer
constexpr unsigned OOB = 0x80000000u;
constexpr int64_t MAX_BUFFER_BYTES = 0x7FFFFF00LL;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Raw block barriers.  __syncthreads() would drain the DMA queue (its fence waits for vmcnt(0): an LDS-DMA is a pending LDS
// write on the VM counter); the "memory" clobber keeps the compiler from moving LDS accesses across the barrier.
__device__ __forceinline__ void ring_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// B16: both operands are bf16 in HBM (the bf16-storage mode's convolution-level GEMMs with bf16 packed weights): a tile row
// is 64 channels = the same 128 bytes, the loaders and the LDS image are unchanged, and a 16-byte fragment read feeds ONE
// v_mfma_f32_16x16x32_bf16 (lane (row l&15, quarter l>>4) holds k = 8*(l>>4) .. +7 of each 32-deep half of the K-step).
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));     // B16 == 2: the 16-bit storage is IEEE half ("fp16s")

// X3 (B16 = 0): the MFMA waves split their fp32 fragments in registers into three bf16 terms and multiply six plane pairs on
// v_mfma_f32_16x16x32_bf16 (see X3 at igemm_wsp_kernel / igemm_nt_kernel); loaders, ring and epilogue are the fp32 kernel's.
template <int MODE, int BM, int BN, int WM, int WN, int S, int B16, bool X3 = false>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NL)) void igemm_ws_kernel(
    const float* __restrict__ A, const float* __restrict__ Bp, const float* __restrict__ bias, float* __restrict__ C,
    float* __restrict__ C_act, float* __restrict__ stats, float* __restrict__ ws, const IgemmGeom g, const unsigned a_bytes,
    const unsigned b_bytes) {
  constexpr int NM = (BM / WM) * (BN / WN);        // MFMA waves
  constexpr int NTHREADS = 64 * (NM + NL);
  constexpr int PA = BM / RPP, PB = BN / RPP;      // DMA pieces per K-step
  static_assert(PA % NL == 0 && PB % NL == 0, "pieces split evenly over the loader waves");
  constexpr int PAL = PA / NL, PBL = PB / NL, PPL = PAL + PBL;
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  constexpr int SLOT = (BM + BN) * RB;
  constexpr int TS = 16, MT = WM / TS, NT = WN / TS;
  constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
  typedef typename std::conditional<B16 == 2, half_t, bf16_t>::type st16_t;
  constexpr int ESZ = B16 ? 2 : 4;                 // operand element size
  constexpr int KB = RB / ESZ;                     // channels per K-step (32 fp32 / 64 bf16)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* rowinfo = reinterpret_cast<int*>(smem + S * SLOT);      // [BM][4]: b, y0, x0, out offset (-1: none)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HWr = g.Hr * g.Wr;
  const int Mg = g.Bg * HWr;
  // XCD-aware block order, identical to igemm_nt.hip (speed only): the N-tiles and parity classes of one M-tile get ids
  // that are equal modulo 8; launches with at most four M-tiles spread each M-tile's N-tiles over 8 / MXp XCDs.
  const int NY = g.N / BN, SI = NY * g.nclasses;
  const int MX = g.G * g.tiles_per_group, MX8 = (MX + 7) >> 3;
  const int L = blockIdx.x;
  const int m_lo = L & 7, r8 = L >> 3;
  int inner, mx, split;
  if (MX > 4) {
    const int rest = r8 / SI;
    inner = r8 % SI;
    mx = (rest % MX8) * 8 + m_lo;
    split = rest / MX8;
  } else {
    const int MXp = MX > 2 ? 4 : MX, nparts = 8 / MXp, Sp = (SI + nparts - 1) / nparts;
    mx = m_lo % MXp;
    inner = (r8 % Sp) * nparts + m_lo / MXp;
    split = r8 / Sp;
    if (inner >= SI) return;
  }
  if (mx >= MX) return;
  const int grp = mx / g.tiles_per_group, tile = mx - grp * g.tiles_per_group;
  const int cls = inner / NY;
  const int n0 = (inner - cls * NY) * BN;
  const int ph = cls >> 1, pw = cls & 1;

  for (int r = tid; r < BM; r += NTHREADS) {
    const int ml = tile * BM + r;
    int ib = -1, y0 = 0, x0 = 0, ooff = -1;
    if (ml < Mg) {
      const int s = ml / HWr;
      const int p = ml - s * HWr;
      const int rr = p / g.Wr;
      const int cc = p - rr * g.Wr;
      ib = grp * g.Bg + s;
      y0 = rr * g.rs + g.ro;
      x0 = cc * g.rs + g.ro;
      const int oy = rr * g.os + ph, ox = cc * g.os + pw;
      ooff = ((ib * g.Ho + oy) * g.Wo + ox) * g.ldc;
    }
    rowinfo[r * 4 + 0] = ib;
    rowinfo[r * 4 + 1] = y0;
    rowinfo[r * 4 + 2] = x0;
    rowinfo[r * 4 + 3] = ooff;
  }
  __syncthreads();

  const int cin_steps = g.Cin / KB;
  const int total_steps = g.ntaps * cin_steps;
  const int per_split = (total_steps + g.splitk - 1) / g.splitk;
  const int s_begin = split * per_split;
  const int s_end = min(total_steps, s_begin + per_split);
  const int nsteps = max(0, s_end - s_begin);

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int mw = wave - NL;                              // index among the MFMA waves (negative: loader)
  const int wm = mw / WAVES_N, wn = mw - wm * WAVES_N;
  const int h = lane >> 4, cl = lane & 15;
  const bool bnbwd = g.bn_y != nullptr;
  int ooff[MT][4];
  float yv[MT][4][NT];                                   // BatchNorm-backward epilogue: the tile's pre-BN values

  if (wave < NL) {
    // ===================================== loader wave =====================================
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)b_bytes, 0x00020000);
    const int prow = lane >> 3;                          // row of this lane inside a piece
    int rb[PAL], ry[PAL], rx[PAL];
    unsigned cslotA[PAL], voffA[PAL], voffB[PBL];
#pragma unroll
    for (int i = 0; i < PAL; ++i) {
      const int r = (wave + NL * i) * RPP + prow;        // tile row
      rb[i] = rowinfo[r * 4 + 0];
      ry[i] = rowinfo[r * 4 + 1];
      rx[i] = rowinfo[r * 4 + 2];
      cslotA[i] = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);   // bytes: the logical slot this lane's position holds
      voffA[i] = OOB;
    }
#pragma unroll
    for (int j = 0; j < PBL; ++j) {
      const int r = (wave + NL * j) * RPP + prow;
      voffB[j] = (unsigned)((grp * g.b_group_stride + (n0 + r) * g.Cin) * ESZ + ((lane & 7) ^ ((r >> 1) & 7)) * 16);
    }
    int tap = s_begin / cin_steps;
    int cstep = s_begin - tap * cin_steps;
    unsigned sB = 0;
    auto settap = [&]() {
      int dh = 0, dw = 0, wi = 0;
      if (MODE == MMDYN_CONV) {
        // Stride 2: the taps that touch the SAME input pixels are walked back to back -- (kh, kw), (kh, kw ^ 2), (kh ^ 2, kw),
        // (kh ^ 2, kw ^ 2) read the same columns of the same input rows one output pixel apart -- so that a block's re-reads
        // of a line follow its first touch by at most three K-steps instead of up to ten.  Order: class (kh & 1, kw & 1) major;
        // the weight slice follows the tap.  Measured (profiles/r4/ab_taporder*.txt): L2 misses of the launch -18 %, hit rate
        // 0.872 -> 0.894, time x1.01 -- the kernel is not bound by where its fills are served from (docs/LAB_NOTES.md E).
        const int cls = tap >> 2, j = tap & 3;           // class (kh & 1, kw & 1); member (kh >> 1, kw >> 1)
        const int t2 = g.tap_order ? ((cls >> 1) + 2 * (j >> 1)) * 4 + (cls & 1) + 2 * (j & 1) : tap;
        dh = t2 >> 2;
        dw = t2 & 3;
        wi = t2;
      } else if (MODE == MMDYN_TCONV_S2P1) {
        const int th = tap >> 1, tw = tap & 1;
        dh = ph - th;
        dw = pw - tw;
        wi = (1 - ph + 2 * th) * 4 + (1 - pw + 2 * tw);
      }
      sB = (unsigned)wi * (unsigned)(g.N * g.Cin) * (unsigned)ESZ;
#pragma unroll
      for (int i = 0; i < PAL; ++i) {
        const int y = ry[i] + dh, x = rx[i] + dw;
        const bool ok = (rb[i] >= 0) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
        const unsigned pix = (unsigned)((rb[i] * g.Hi + y) * g.Wi + x);
        voffA[i] = ok ? pix * (unsigned)(g.Cin * ESZ) + cslotA[i] : OOB;
      }
    };
    int issued = 0;
    auto issue = [&]() {
      char* slot = smem + (issued % S) * SLOT;
      const unsigned so = (unsigned)cstep * RB;
#pragma unroll
      for (int i = 0; i < PAL; ++i) dma16(rsA, slot + (wave + NL * i) * 1024, voffA[i], so);
#pragma unroll
      for (int j = 0; j < PBL; ++j) dma16(rsB, slot + BM * RB + (wave + NL * j) * 1024, voffB[j], sB + so);
      ++issued;
      if (++cstep == cin_steps) {
        cstep = 0;
        ++tap;
        if (issued < nsteps) settap();
      }
    };
    if (nsteps > 0) settap();
    for (int k = 0; k < S - 1 && k < nsteps; ++k) issue();
    for (int k = 0; k < nsteps; ++k) {
      if (k + S - 1 <= nsteps) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>();
      ring_barrier();                                    // slot k is complete; slot k-1 has been read by every MFMA wave
      if (k + S - 1 < nsteps) issue();
    }
  } else {
    // ===================================== MFMA waves =====================================
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) ooff[mt][e] = rowinfo[(wm * WM + mt * TS + 4 * h + e) * 4 + 3];
    if (bnbwd) {
      // the epilogue's operand: requested NOW, in flight under the whole K loop (at two blocks per CU nothing else would
      // hide these 32 dependent round trips)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
          {
            const size_t yo = (size_t)max(ooff[mt][e], 0) + n0 + wn * WN + nt * TS + cl;
            yv[mt][e][nt] = (B16 && g.bny_b16) ? ld1<st16_t>(reinterpret_cast<const st16_t*>(g.bn_y) + yo) : g.bn_y[yo];
          }
    }
    const int fr = (cl >> 1) & 7;
    const int foff0 = cl * RB + 16 * ((0 + h) ^ fr), foff1 = cl * RB + 16 * ((4 + h) ^ fr);
    const int abase = wm * WM * RB, bbase = BM * RB + wn * WN * RB;
    for (int k = 0; k < nsteps; ++k) {
      ring_barrier();
      const char* sl = smem + (k % S) * SLOT;
      if constexpr (X3) {
        static_assert(B16 == 0, "the three-term split is a variant of the fp32 kernel");
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        bf16x8v ap[3][MT], bp[3][NT];
        auto split_frag = [&](const char* base, bf16x8v& hi, bf16x8v& mid, bf16x8v& lo) {
          const f32x4v x0 = *reinterpret_cast<const f32x4v*>(base + foff0);
          const f32x4v x1 = *reinterpret_cast<const f32x4v*>(base + foff1);
          uint32_t h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
          split3_bf16(x0[0], x0[1], h0, m0, l0);
          split3_bf16(x0[2], x0[3], h1, m1, l1);
          split3_bf16(x1[0], x1[1], h2, m2, l2);
          split3_bf16(x1[2], x1[3], h3, m3, l3);
          hi = __builtin_bit_cast(bf16x8v, (u32x4v){h0, h1, h2, h3});
          mid = __builtin_bit_cast(bf16x8v, (u32x4v){m0, m1, m2, m3});
          lo = __builtin_bit_cast(bf16x8v, (u32x4v){l0, l1, l2, l3});
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) split_frag(sl + bbase + nt * TS * RB, bp[0][nt], bp[1][nt], bp[2][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) split_frag(sl + abase + mt * TS * RB, ap[0][mt], ap[1][mt], ap[2][mt]);
        constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (plane of A, plane of B), smallest first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[order[t][0]][mt], bp[order[t][1]][nt], acc[mt][nt], 0, 0, 0);
      } else
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int fo = q ? foff1 : foff0;
        if constexpr (B16) {
          bf16x8v af[MT], bf[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const bf16x8v*>(sl + abase + mt * TS * RB + fo);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const bf16x8v*>(sl + bbase + nt * TS * RB + fo);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if constexpr (B16 == 2)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8v, af[mt]),
                                                                     __builtin_bit_cast(f16x8v, bf[nt]), acc[mt][nt], 0, 0, 0);
              else
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        } else {
          f32x4v af[MT], bf[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const f32x4v*>(sl + abase + mt * TS * RB + fo);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const f32x4v*>(sl + bbase + nt * TS * RB + fo);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue (MFMA waves; the loader waves only keep the block barriers company) ----
  // accumulator element e of tile (mt, nt): row 4*(l>>4) + e, column l&15
  const bool mfma_wave = wave >= NL;
  float colsum[NT], colsq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) colsum[nt] = colsq[nt] = 0.f;
  if (mfma_wave) {
    float bn_m[NT], bn_r[NT], bn_g[NT], bn_b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wn * WN + nt * TS + cl;
      const bool bn = bnbwd && g.bn_mean != nullptr;      // (activation-only backward: xhat = u, gamma = 1, beta = 0)
      bn_m[nt] = bn ? g.bn_mean[(size_t)grp * g.N + col] : 0.f;
      bn_r[nt] = bn ? g.bn_rstd[(size_t)grp * g.N + col] : 1.f;
      bn_g[nt] = bn ? g.bn_gamma[col] : 1.f;
      bn_b[nt] = bn ? g.bn_beta[col] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int oo = ooff[mt][e];
        const bool live = oo >= 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = n0 + wn * WN + nt * TS + cl;
          float v = acc[mt][nt][e];
          if (bnbwd) {
            const float xh = live ? (yv[mt][e][nt] - bn_m[nt]) * bn_r[nt] : 0.f;
            v = live ? v * act_grad(bn_g[nt] * xh + bn_b[nt], g.bwd_act) : v;
            colsum[nt] += v;
            colsq[nt] += v * xh;
          } else {
            colsum[nt] += v;
            colsq[nt] += v * v;
          }
          if (live) {
            if (g.splitk > 1) {
              const int grow = grp * Mg + tile * BM + wm * WM + mt * TS + 4 * h + e;   // dense row id (DENSE mode only)
              ws[((size_t)split * g.rows_total + grow) * g.N + col] = v;
            } else {
              if (g.has_bias) v += bias[grp * g.bias_group_stride + col];
              if (B16 && g.c_b16) {
                // two adjacent columns live in adjacent lanes: the even lane stores both as one dword
                const float vn = __shfl_down(v, 1, 64);
                if (!(cl & 1)) {
                  *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C) + (size_t)oo + col) = pack2<st16_t>(v, vn);
                  if (g.want_act_out)
                    *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C_act) + (size_t)oo + col) =
                        pack2<st16_t>(apply_act(v, g.act), apply_act(vn, g.act));
                }
              } else {
                C[(size_t)oo + col] = v;
                if (g.want_act_out) {
                  if (B16 && g.cact_b16) {      // fp32 pre-activation, bf16 activated output (FC level -> conv level)
                    const float av = apply_act(v, g.act);
                    const float an = __shfl_down(av, 1, 64);
                    if (!(cl & 1))
                      *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C_act) + (size_t)oo + col) = pack2<st16_t>(av, an);
                  } else {
                    C_act[(size_t)oo + col] = apply_act(v, g.act);
                  }
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
    lds_barrier();                                       // every MFMA wave is done with the ring: reuse it as scratch
    float* red = reinterpret_cast<float*>(smem);         // [WAVES_M][2][BN]
    if (mfma_wave) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float s = colsum[nt] + __shfl_xor(colsum[nt], 32, 64);
        float q = colsq[nt] + __shfl_xor(colsq[nt], 32, 64);
        s += __shfl_xor(s, 16, 64);                      // a column's rows sit in the four lanes l, l+16, l+32, l+48
        q += __shfl_xor(q, 16, 64);
        if (h == 0) {
          red[(wm * 2 + 0) * BN + wn * WN + nt * TS + cl] = s;
          red[(wm * 2 + 1) * BN + wn * WN + nt * TS + cl] = q;
        }
      }
    }
    lds_barrier();
    const int t = tid - 64 * NL;
    if (t >= 0 && t < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        s += red[(w * 2 + 0) * BN + t];
        q += red[(w * 2 + 1) * BN + t];
      }
      const int T = g.nclasses * g.tiles_per_group, slot = cls * g.tiles_per_group + tile;
      const size_t base = ((size_t)(grp * T + slot) * 2) * g.N + n0 + t;
      stats[base] = s;
      stats[base + g.N] = q;
    }
  }
}

// ---- which launches this file serves, and with which tile (ONE function for the launcher and the stat-tile query) ----
struct WsPick {
  int bm, bn;      // 0: not served
};

static WsPick ws_pick(int mode, int G, int Bg, int Hi, int Wi, int Hr, int Wr, int Cin, int N, int ncls, int splitk,
                      bool b16 = false) {
  WsPick p{0, 0};
  if (mode != MMDYN_DENSE && mode != MMDYN_CONV && mode != MMDYN_TCONV_S2P1) return p;
  if (Cin % (b16 ? 64 : BK) || N % 32) return p;
  if (N % 64) {                 // 32 (mod 64) output channels (the 32 -> 32 stages of the 128 / 256 pixel stacks): fp32, 128 x 32
    if (b16 || splitk > 1) return p;
    if ((int64_t)G * Bg * Hi * Wi * Cin * 4 >= MAX_BUFFER_BYTES) return p;
    p.bm = 128;
    p.bn = 32;
    return p;
  }
  // buffer descriptors carry 32-bit byte counts, and the out-of-range marker must stay beyond them
  const int esz = b16 ? 2 : 4;
  if ((int64_t)G * Bg * Hi * Wi * Cin * esz >= MAX_BUFFER_BYTES || (int64_t)16 * N * Cin * esz >= MAX_BUFFER_BYTES) return p;
  if (b16) {
    // bf16 operands: the block tile follows the register-staged kernels' rule (igemm_geom.h pick_tile), so the number of
    // BatchNorm partial-sum tiles a launch writes does not depend on which of the two kernels serves it
    // (mmdyn_igemm_stat_tiles_bf16 knows the shape, not the storage types)
    int bm, bn;
    pick_tile(N, Bg * Hr * Wr, G, ncls, splitk, (mode == MMDYN_CONV ? 16 : (mode == MMDYN_TCONV_S2P1 ? 4 : 1)) * (Cin / BK), &bm,
              &bn);
    if (bm == bn && (bm == 64 || bm == 128)) p.bm = p.bn = bm;
    return p;
  }
  // Measured per shape against the register-staged kernels, each launch alone on the chip (tests/microbench/ab_ws.py,
  // profiles/r3/ab_ws_per_shape.txt):
  //   * 64x64 tiles (4 MFMA waves of 32x32, 49 KB of LDS: three blocks per CU) win or tie on every launch of the step but
  //     one -- x1.13..1.25 on the single-group encoder-side convolutions and the FC-level GEMMs, x1.19 on the decoder's
  //     first input gradient, x1.0..1.04 on the four-group decoder launches;
  //   * 128x128 tiles (8 MFMA waves of 64x32, 98 KB: one block per CU) win where they still give every CU two tiles and
  //     the K loop is >= 32 steps (x1.12..1.13), and lose everywhere else (pipeline fill and epilogue of a lone block);
  //   * very many rows with a short K loop (the 32 -> 64 channel layer's input gradient: 262144 rows, 16 K-steps): the
  //     register-staged kernel at six blocks per CU hides its epilogue better (x0.93 here): not served.
  const long rows_g = (long)Bg * Hr * Wr;
  const int ksteps = (mode == MMDYN_CONV ? 16 : (mode == MMDYN_TCONV_S2P1 ? 4 : 1)) * (Cin / BK) / splitk;
  if (mode == MMDYN_CONV && ksteps <= 16 && (long)G * rows_g >= 200000) return p;
  const long tiles128 = (long)G * ((rows_g + 127) / 128) * (N / 128) * ncls;
  p.bm = p.bn = 64;
  if (N % 128 == 0 && ksteps >= 32 && tiles128 >= 512) p.bm = p.bn = 128;
  if (const char* e = lab_env("MMDYN_WS_TILE")) {          // LAB build: force one tile (kernel experiments)
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && (a == 64 || a == 128) && (b == 64 || b == 128) && N % b == 0) {
      p.bm = a;
      p.bn = b;
    }
  }
  return p;
}

template <int MODE, int BM, int BN, int WM, int WN, int S, int B16 = 0>
static int ws_launch(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                     IgemmGeom g, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  constexpr int NM = (BM / WM) * (BN / WN);
  g.tiles_per_group = ceil_div(g.Bg * g.Hr * g.Wr, BM);
  const int mx_total = g.G * g.tiles_per_group, mx8 = (mx_total + 7) / 8 * 8;
  const int s_inner = (g.N / BN) * g.nclasses;
  dim3 grid((unsigned)mx8 * s_inner * g.splitk);
  if (mx_total <= 4) {
    const int mxp = mx_total > 2 ? 4 : mx_total, nparts = 8 / mxp;
    grid = dim3((unsigned)8 * ((s_inner + nparts - 1) / nparts) * g.splitk);
  }
  const size_t smem = (size_t)S * (BM + BN) * RB + (size_t)BM * 16;
#ifdef MMDYN_LAB
  // (LAB experiment, MMDYN_X3_WS=1 at igemm_entry: measured no better than the register-staged split kernels on the step -- the
  //  64x64 ring is balanced against its loaders, which the split does not speed up; docs/LAB_NOTES.md F.g)
  if constexpr (B16 == 0) {
    if (g.x3) {            // fp32 on the bf16 matrix cores (three-term split in the MFMA waves)
      static LdsOptIn x3_opt_in;
      if (int e = x3_opt_in.ensure((const void*)igemm_ws_kernel<MODE, BM, BN, WM, WN, S, 0, true>, (int)smem)) return e;
      hipLaunchKernelGGL((igemm_ws_kernel<MODE, BM, BN, WM, WN, S, 0, true>), grid, dim3(64 * (NM + NL)), smem, st, A, Bp, bias, C,
                         C_act, stats, ws, g, a_bytes, b_bytes);
      MMDYN_LAUNCH_CHECK();
    }
  }
#endif
  static LdsOptIn lds_opt_in;        // (per kernel instance; per device inside)
  if (int e = lds_opt_in.ensure((const void*)igemm_ws_kernel<MODE, BM, BN, WM, WN, S, B16>, (int)smem)) return e;
  hipLaunchKernelGGL((igemm_ws_kernel<MODE, BM, BN, WM, WN, S, B16>), grid, dim3(64 * (NM + NL)), smem, st, A, Bp, bias, C, C_act,
                     stats, ws, g, a_bytes, b_bytes);
  MMDYN_LAUNCH_CHECK();
}

template <int MODE, int B16>
static int ws_launch_mode(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                          const IgemmGeom& g, WsPick p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  if constexpr (!B16)
    if (p.bm == 128 && p.bn == 32)
      return ws_launch<MODE, 128, 32, 32, 32, 3, false>(A, Bp, bias, C, C_act, stats, ws, g, a_bytes, b_bytes, st);
  if (p.bm == 128 && p.bn == 128)
    return ws_launch<MODE, 128, 128, 64, 32, 3, B16>(A, Bp, bias, C, C_act, stats, ws, g, a_bytes, b_bytes, st);
  if (p.bm == 128) return ws_launch<MODE, 128, 64, 32, 64, 3, B16>(A, Bp, bias, C, C_act, stats, ws, g, a_bytes, b_bytes, st);
#ifdef MMDYN_LAB
  if constexpr (!B16)         // LAB: 64 rows x 128 channels (the gathered operand filled once per 128 output channels)
    if (p.bn == 128) return ws_launch<MODE, 64, 128, 32, 64, 3, B16>(A, Bp, bias, C, C_act, stats, ws, g, a_bytes, b_bytes, st);
#endif
  if constexpr (MODE == MMDYN_DENSE) {
    // Launches of at most two 64x64 blocks per CU (the heads / pose / FC-level GEMMs: K loops of 8-16 steps at one or two
    // blocks per CU) are chains of DMA round trips, not matrix work: a four-slot ring keeps three K-steps in flight instead of
    // two (65 KB of LDS: still two blocks per CU).  LAB build: MMDYN_WS_DENSE_S=3 switches it off (A/B).
    const long blocks = (long)g.G * ceil_div(g.Bg * g.Hr * g.Wr, 64) * (g.N / 64) * g.splitk;
    const char* e = lab_env("MMDYN_WS_DENSE_S");
    if (blocks <= 2L * 256 && !(e && e[0] == '3'))
      return ws_launch<MODE, 64, 64, 32, 32, 4, B16>(A, Bp, bias, C, C_act, stats, ws, g, a_bytes, b_bytes, st);
  }
  return ws_launch<MODE, 64, 64, 32, 32, 3, B16>(A, Bp, bias, C, C_act, stats, ws, g, a_bytes, b_bytes, st);
}

template <int B16>
static int ws_dispatch(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                       const IgemmGeom& g, WsPick p, hipStream_t st) {
  const int esz = B16 ? 2 : 4;
  const unsigned a_bytes = (unsigned)((int64_t)g.G * g.Bg * g.Hi * g.Wi * g.Cin * esz);
  const unsigned b_bytes = (unsigned)(((int64_t)(g.mode == MMDYN_DENSE ? 1 : 16) * g.N * g.Cin +
                                       (int64_t)(g.G - 1) * g.b_group_stride) * esz);
  if (g.mode == MMDYN_DENSE)
    return ws_launch_mode<MMDYN_DENSE, B16>(A, Bp, bias, C, C_act, stats, ws, g, p, a_bytes, b_bytes, st);
  if (g.mode == MMDYN_CONV)
    return ws_launch_mode<MMDYN_CONV, B16>(A, Bp, bias, C, C_act, stats, ws, g, p, a_bytes, b_bytes, st);
  return ws_launch_mode<MMDYN_TCONV_S2P1, B16>(A, Bp, bias, C, C_act, stats, ws, g, p, a_bytes, b_bytes, st);
}

}  // namespace

int mmdyn_igemm_ws_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  int Hr = Ho, Wr = Wo, ncls = 1;
  if (mode == MMDYN_TCONV_S2P1) {
    Hr = Hi;
    Wr = Wi;
    ncls = 4;
  }
  const WsPick p = ws_pick(mode, G, Bg, Hi, Wi, Hr, Wr, Cin, N, ncls, 1);
  return p.bm ? ncls * ceil_div(Bg * Hr * Wr, p.bm) : 0;
}

// bf16_ops: the launch runs on the 16-bit matrix cores; served here only when BOTH operands are 16-bit in HBM (bf16, or IEEE
// half when g.f16 is set too)
int mmdyn_igemm_ws_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* ws,
                       const IgemmGeom& g_in, bool bf16_ops, hipStream_t st) {
  IgemmGeom g = g_in;
  g.tap_order = g.mode == MMDYN_CONV && g.rs == 2;
  if (const char* e = lab_env("MMDYN_WS_TAPORDER")) g.tap_order = g.tap_order && e[0] != '0';      // LAB: raster order (A/B)
  if (bf16_ops && (!g.a_b16 || !g.b_b16)) return 1;
  if ((int64_t)g.G * g.b_group_stride * 4 >= MAX_BUFFER_BYTES) return 1;      // (grouped weights: one buffer descriptor)
  const WsPick p = ws_pick(g.mode, g.G, g.Bg, g.Hi, g.Wi, g.Hr, g.Wr, g.Cin, g.N, g.nclasses, g.splitk, bf16_ops);
  if (!p.bm) return 1;
  if (bf16_ops && g.f16) return ws_dispatch<2>(A, Bp, bias, C, C_act, stats, ws, g, p, st);
  if (bf16_ops) return ws_dispatch<1>(A, Bp, bias, C, C_act, stats, ws, g, p, st);
  return ws_dispatch<0>(A, Bp, bias, C, C_act, stats, ws, g, p, st);
}
