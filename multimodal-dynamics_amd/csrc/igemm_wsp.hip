// Persistent, stream-K-scheduled form of the wave-specialised implicit GEMM (igemm_ws.hip) for the LARGE launches of the
// step (v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_{bf16,f16}, gfx950).
//
//   C[row][n] = sum_{tap, ci} A_tap[row][ci] * Bp[widx(tap)][n][ci]        (same contract as igemm_nt.hip / igemm_ws.hip)
//
// Replaces the ATen kernels behind nn.Conv2d / nn.ConvTranspose2d forward and input-gradient on the reference path
// (/root/reference/mmdyn/pytorch/models/vae.py:198-216, 264-277) for the launches it serves (wsp_pick below).
//
// Why a third structure (docs/LAB_NOTES.md D.c, E): 128-row tiles do 21-32 flop per filled LDS byte
// against 16 for the 64x64 tile and measure 122-134 TFLOP/s as a dense GEMM, but one block per tile hands that back:
// a lone 98 KB block per CU exposes its row decode, its first DMA round trip and its epilogue, and 200-400 tiles over 256
// CUs leave a wave of CUs idle at the end.  Here a launch is ONE resident block per CU slot that walks a contiguous range
// of (tile, K-step) units:
//   * the loader waves run the DMA ring STRAIGHT THROUGH the tile boundary: the first S-1 K-steps of tile t+1 land
//     while the MFMA waves are still in the epilogue of tile t; the per-tile row decode (sample, y, x of every gathered
//     row) is done by the LOADER lanes for their own 8 rows each, in the shadow of the DMA queue, with a float-reciprocal
//     division -- no integer division on any MFMA lane, no block barrier at the tile boundary.  The MFMA waves get the
//     output offsets of their rows through a four-deep LDS table the loaders fill one segment ahead;
//   * the unit ranges are equal for all blocks (stream-K): a tile whose K range straddles two blocks is accumulated in
//     pieces; each piece goes to a private slab (the accumulator fragments as they sit in registers, 16 bytes per lane)
//     and a second, tiny launch (igemm_wsp_fixup_kernel) sums the pieces of each split tile in piece order and runs the
//     SAME epilogue -- deterministic, no atomics, no in-kernel inter-block hand-off (the kernel boundary is the fence);
//   * BatchNorm partial sums are written per (tile, wave row): no cross-wave reduction, hence no block barrier in the
//     epilogue (mmdyn_igemm_wsp_stat_tiles tells the host how many partial tiles a launch writes).
// The ring itself (LDS-DMA pieces of 8 rows x 128 B, XOR swizzle on the DMA source address and on the fragment read,
// counted vmcnt, ONE raw s_barrier per K-step, out-of-range offsets for padding rows) is igemm_ws.hip's.
#include "common.h"
#include "igemm_geom.h"
#include <cstdio>
#include <type_traits>

namespace {

constexpr int BK = 32;          // K-step: 32 fp32 channels (64 16-bit ones) = one 128-byte row segment
constexpr int RB = 128;         // bytes per tile row
constexpr int RPP = 8;          // rows per DMA piece (1 KiB per wave instruction)
constexpr int NL = 2;           // loader waves per block
constexpr int NRO = 4;          // depth of the row-offset table (segments the loaders may be ahead of the MFMA waves: <= S-1)
constexpr unsigned OOB = 0x80000000u;
constexpr int64_t MAX_BUFFER_BYTES = 0x7FFFFF00LL;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// raw block barrier (no vmcnt drain: LDS-DMA stays in flight across it); the loaders' LDS stores (row offsets) are complete
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// LAB build only (MMDYN_WSP_DIAG=1): cycle stamps of wave 0 (loader) and of the first MFMA wave of every block, summed over the
// launch -- where a K-step's time goes.  [0] MFMA wave: cycles waiting in the ring barrier, [1] in K loops (barrier included),
// [2] in epilogues / slab stores, [3] whole kernel; [4] loader: cycles in its counted vmcnt wait, [5] in the ring barrier,
// [6] issuing (row decode of the next segment included), [7] whole loop.  The stamps cost ~10 % themselves.
#ifdef MMDYN_LAB
__device__ unsigned long long wsp_diag[8];
#define WSP_STAMP() ((long long)__builtin_readcyclecounter())
#else
#define WSP_STAMP() 0LL
#endif

// Schedule of one launch: units = tiles x K-steps, block b owns units [b * per, (b + 1) * per).
struct WspSched {
  int tiles;       // M-tiles x N-tiles x parity classes
  int ksteps;      // K-steps per tile (TCONV_S1P0: of the heaviest tile; the count varies with the output pixel)
  int units;       // all (tile, K-step) units of the launch
  int per;         // units per block (the last block may get fewer)
  int si, ny;      // inner tiles per M-tile (N-tiles x classes), N-tiles
  int cin_steps;   // K-steps per filter tap
  int spg, ug;     // TCONV_S1P0: 128-sample tiles per group; units of one (group, sample tile): 400 taps x cin_steps
  float inv_hw, inv_w;
};

// TCONV_S1P0 (k4 s1 p0 transposed convolution, 5x5 -> 8x8): a tile is (group, 128 samples, ONE output pixel) and multiplies
// only the taps that reach the input for that pixel: nk(py) * nk(px) of them, nk = 1 2 3 4 4 3 2 1 along each axis (400 per
// sample tile in all).  The tiles are unequal, the unit ranges of the blocks are not: stream-K balances what the register-staged
// kernel balances with its four-pixel walk.
__device__ __forceinline__ int s1p0_nk(int i) { return min(i, 7 - i) + 1; }
__device__ __forceinline__ int s1p0_cum(int i) { return i <= 4 ? i * (i + 1) / 2 : 20 - (8 - i) * (9 - i) / 2; }   // taps before i

struct Seg {
  int t, ub, kt;   // tile, its first unit, its K-steps
};
template <int MODE> __device__ __forceinline__ Seg seg_of_tile(int t, const WspSched& sc) {
  Seg s;
  s.t = t;
  if constexpr (MODE == MMDYN_TCONV_S1P0) {
    const int q = t >> 6, py = (t >> 3) & 7, px = t & 7;
    s.ub = q * sc.ug + (20 * s1p0_cum(py) + s1p0_nk(py) * s1p0_cum(px)) * sc.cin_steps;
    s.kt = s1p0_nk(py) * s1p0_nk(px) * sc.cin_steps;
  } else {
    s.ub = t * sc.ksteps;
    s.kt = sc.ksteps;
  }
  return s;
}
template <int MODE> __device__ __forceinline__ Seg seg_at(int u, const WspSched& sc) {
  if constexpr (MODE == MMDYN_TCONV_S1P0) {
    const int q = u / sc.ug;
    const int ts = (u - q * sc.ug) / sc.cin_steps;       // tap-step inside the (group, sample tile): 0 .. 399
    int py = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) py = ts >= 20 * s1p0_cum(i) ? i : py;
    const int rem = ts - 20 * s1p0_cum(py), nky = s1p0_nk(py);
    int px = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) px = rem >= nky * s1p0_cum(i) ? i : px;
    return seg_of_tile<MODE>(q * 64 + py * 8 + px, sc);
  } else {
    return seg_of_tile<MODE>(u / sc.ksteps, sc);
  }
}

// quotient and remainder by a launch constant through a float reciprocal, exact for 0 <= n < 2^23 (checked by the launcher)
__device__ __forceinline__ void fdiv(int n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = n - q * d;
  if (r < 0) { q -= 1; r += d; }
  if (r >= d) { q += 1; r -= d; }
}

struct TileId {
  int grp, tile, cls, n0, ph, pw;      // TCONV_S1P0: tile = pixel * spg + sample tile (the partial-sum slot), ph / pw = the pixel
  int stile;                           // TCONV_S1P0: 128-sample tile inside the group
};
template <int MODE> __device__ __forceinline__ TileId tile_of(int t, const IgemmGeom& g, const WspSched& sc, int BN) {
  TileId id;
  if constexpr (MODE == MMDYN_TCONV_S1P0) {
    const int q = t >> 6, p = t & 63;
    id.grp = q / sc.spg;
    id.stile = q - id.grp * sc.spg;
    id.tile = p * sc.spg + id.stile;
    id.cls = 0;
    id.n0 = 0;
    id.ph = p >> 3;
    id.pw = p & 7;
    return id;
  }
  id.stile = 0;
  const int mx = t / sc.si, inner = t - mx * sc.si;
  id.grp = mx / g.tiles_per_group;
  id.tile = mx - id.grp * g.tiles_per_group;
  id.cls = inner / sc.ny;
  id.n0 = (inner - id.cls * sc.ny) * BN;
  id.ph = id.cls >> 1;
  id.pw = id.cls & 1;
  return id;
}

// row `ml` of group `grp` (class ph, pw) -> sample index, gather base (y0, x0), output offset (-1: the row does not exist)
template <int MODE>
__device__ __forceinline__ void decode_row(int ml, int Mg, const TileId& id, const IgemmGeom& g, const WspSched& sc, int& ib,
                                           int& y0, int& x0, int& ooff) {
  ib = -1;
  y0 = x0 = 0;
  ooff = -1;
  if constexpr (MODE == MMDYN_TCONV_S1P0) {              // ml: sample index inside the group; the row's pixel is the tile's
    if (ml < g.Bg) {
      ib = id.grp * g.Bg + ml;
      y0 = id.ph;
      x0 = id.pw;
      ooff = ((ib * g.Ho + id.ph) * g.Wo + id.pw) * g.ldc;
    }
    return;
  }
  if (ml < Mg) {
    int s, p, rr, cc;
    fdiv(ml, g.Hr * g.Wr, sc.inv_hw, s, p);
    fdiv(p, g.Wr, sc.inv_w, rr, cc);
    ib = id.grp * g.Bg + s;
    y0 = rr * g.rs + g.ro;
    x0 = cc * g.rs + g.ro;
    ooff = ((ib * g.Ho + rr * g.os + id.ph) * g.Wo + cc * g.os + id.pw) * g.ldc;
  }
}

// ---- epilogue of one finished tile, shared by the main kernel and the fix-up kernel (MFMA-wave lanes only) ----
// An accumulator register holds, over the wave, 4 rows x 16 columns of a 16x16 tile (element e: row 4*(l>>4) + e, column
// l&15): stored as it sits that is 32 global_store_dword per lane and tile, and the epilogue of a 128x128 tile took 15 000
// cycles -- store ISSUE, not bytes (cycle stamps: docs/LAB_NOTES.md E; cdna_hip_programming.md T21).  Every 16x16 tile
// therefore goes through a wave-private LDS patch ([16][TRLD] floats) and comes back ROW-major: lane l holds row l>>2,
// columns 4*(l&3) .. +3 -- one 16-byte store (and, for the BatchNorm / activation backward, one 16-byte load of the saved
// pre-activation) per tile and lane, 4x fewer memory instructions, the same bytes.
constexpr int TRLD = 20;        // patch row stride (floats): 16-byte aligned rows; the 4 rows of a store group 2-way conflict only

// the tile's saved pre-BN / pre-activation values in the transposed layout: yq[mt][nt] = row (l>>2) of tile (mt, nt), 4 columns
template <int MT, int NT, int B16>
__device__ __forceinline__ void wsp_fetch_y(f32x4v (&yq)[MT][NT], const int (&ooff)[MT], const IgemmGeom& g, int col0) {
  typedef typename std::conditional<B16 == 2, half_t, bf16_t>::type st16_t;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const size_t yo = (size_t)max(ooff[mt], 0) + col0 + nt * 16;
      if (B16 && g.bny_b16) yq[mt][nt] = ld4<st16_t>(reinterpret_cast<const st16_t*>(g.bn_y) + yo);
      else yq[mt][nt] = *reinterpret_cast<const f32x4v*>(g.bn_y + yo);
    }
}

// ooff[mt]: output offset of row (l>>2) of the wave's mt-th 16-row tile (-1: no such row); tr: this wave's LDS patch
template <int BM, int BN, int WM, int WN, int B16, bool PLOUT = false>
__device__ __forceinline__ void wsp_epilogue(f32x4v (&acc)[WM / 16][WN / 16], const int (&ooff)[WM / 16],
                                             const f32x4v (&yq)[WM / 16][WN / 16], const TileId& id, const IgemmGeom& g,
                                             const float* __restrict__ bias, float* __restrict__ C, float* __restrict__ C_act,
                                             float* __restrict__ stats, int wm, int wn, int lane, float* tr) {
  constexpr int TS = 16, MT = WM / TS, NT = WN / TS, WAVES_M = BM / WM;
  typedef typename std::conditional<B16 == 2, half_t, bf16_t>::type st16_t;
  const int h = lane >> 4, cl = lane & 15;
  const int trow = lane >> 2, c4 = (lane & 3) * 4;       // transposed layout: row inside the 16-row tile, first of 4 columns
  const bool bnbwd = g.bn_y != nullptr;
  const bool bn = bnbwd && g.bn_mean != nullptr;         // (activation-only backward: xhat = u, gamma = 1, beta = 0)
  f32x4v cs[NT], cq[NT], bn_m[NT], bn_r[NT], bn_g[NT], bn_b[NT], bia[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = id.n0 + wn * WN + nt * TS + c4;
    cs[nt] = cq[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    bn_m[nt] = bn ? *reinterpret_cast<const f32x4v*>(g.bn_mean + (size_t)id.grp * g.N + col) : f32x4v{0.f, 0.f, 0.f, 0.f};
    bn_r[nt] = bn ? *reinterpret_cast<const f32x4v*>(g.bn_rstd + (size_t)id.grp * g.N + col) : f32x4v{1.f, 1.f, 1.f, 1.f};
    bn_g[nt] = bn ? *reinterpret_cast<const f32x4v*>(g.bn_gamma + col) : f32x4v{1.f, 1.f, 1.f, 1.f};
    bn_b[nt] = bn ? *reinterpret_cast<const f32x4v*>(g.bn_beta + col) : f32x4v{0.f, 0.f, 0.f, 0.f};
    bia[nt] = g.has_bias ? *reinterpret_cast<const f32x4v*>(bias + id.grp * g.bias_group_stride + col) : f32x4v{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int oo = ooff[mt];
    const bool live = oo >= 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = id.n0 + wn * WN + nt * TS + c4;
#pragma unroll
      for (int e = 0; e < 4; ++e) tr[(4 * h + e) * TRLD + cl] = acc[mt][nt][e];
      f32x4v v = *reinterpret_cast<const f32x4v*>(&tr[trow * TRLD + c4]);
      if (bnbwd) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = live ? (yq[mt][nt][k] - bn_m[nt][k]) * bn_r[nt][k] : 0.f;
          v[k] = live ? v[k] * act_grad(bn_g[nt][k] * xh + bn_b[nt][k], g.bwd_act) : v[k];
          cs[nt][k] += v[k];
          cq[nt][k] += v[k] * xh;
        }
      } else {
        cs[nt] += v;
        cq[nt] += v * v;
      }
      if (live) {
        if (g.has_bias) v += bia[nt];
        f32x4v a = v;
        if (g.want_act_out) {
#pragma unroll
          for (int k = 0; k < 4; ++k) a[k] = apply_act(v[k], g.act);
        }
        if (B16 && g.c_b16) {
          st4<st16_t>(reinterpret_cast<st16_t*>(C) + (size_t)oo + col, v);
          if (g.want_act_out) st4<st16_t>(reinterpret_cast<st16_t*>(C_act) + (size_t)oo + col, a);
        } else {
          *reinterpret_cast<f32x4v*>(C + (size_t)oo + col) = v;
          if (g.want_act_out) {
            if (B16 && g.cact_b16) st4<st16_t>(reinterpret_cast<st16_t*>(C_act) + (size_t)oo + col, a);
            else if (PLOUT && !B16 && g.cact_planes) {     // (PLOUT: compiled into the DENSE instance of the plane kernel only)
              // the activated output as a PLANE tensor (rows of [plane][cact_planes] bf16, the exact three-term split; ldc == N): the GEMM that
              // consumes it takes its operand already split and no stand-alone split launch is needed (round 6)
              uint32_t h0, m0, l0, h1, m1, l1;
              split3_bf16(a[0], a[1], h0, m0, l0);
              split3_bf16(a[2], a[3], h1, m1, l1);
              typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
              // (plane rows of cact_planes channels: an output row of N columns is N / cact_planes consecutive plane rows -- the
              //  [B][hw*256 + c] output of the decoder's Linear layer read as [B*25 pixels][256 channels] by the layer above)
              const int cp = g.cact_planes, cin = col % cp;
              bf16_t* pr = reinterpret_cast<bf16_t*>(C_act) + (size_t)oo * 3 + (size_t)(col - cin) * 3 + cin;
              *reinterpret_cast<u32x2v*>(pr) = u32x2v{h0, h1};
              *reinterpret_cast<u32x2v*>(pr + cp) = u32x2v{m0, m1};
              *reinterpret_cast<u32x2v*>(pr + 2 * cp) = u32x2v{l0, l1};
            } else *reinterpret_cast<f32x4v*>(C_act + (size_t)oo + col) = a;
          }
        }
      }
    }
  }
  if (g.want_stats) {
    // one partial-sum tile per (M-tile, class, wave row): rows beyond the group are zero-filled operands -> contribute 0.
    // A column's 64 rows: 4 tiles (summed above) x the 16 lanes with the same l&3
    const int T = g.nclasses * g.tiles_per_group * WAVES_M;
    const int slot = (id.cls * g.tiles_per_group + id.tile) * WAVES_M + wm;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float sv = cs[nt][k], qv = cq[nt][k];
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) {
          sv += __shfl_xor(sv, m, 64);
          qv += __shfl_xor(qv, m, 64);
        }
        cs[nt][k] = sv;
        cq[nt][k] = qv;
      }
      if (lane < 4) {
        const size_t base = ((size_t)(id.grp * T + slot) * 2) * g.N + id.n0 + wn * WN + nt * TS + c4;
        *reinterpret_cast<f32x4v*>(stats + base) = cs[nt];
        *reinterpret_cast<f32x4v*>(stats + base + g.N) = cq[nt];
      }
    }
  }
}

// ---- split tiles finished INSIDE the launch (round 6) -- LAB build only ----
// Built, tested bit-identical to the fix-up launch (tests/test_kernels_aten_gpu.py::test_split_tiles_finished_inside_the_launch_*)
// and measured SLOWER on the step, same box, alternating runs (docs/LAB_NOTES.md H.a): 50.4-50.6 k samples/s against 51.2-51.7 k for the
// fix-up launch (v1 with an agent-scope acquire per finishing wave: 49.3-50.9 against 53.8 k) -- the last arriver's round trips
// (arrival word, pieces) sit at the END of a persistent block that holds a whole CU, where the separate launch's 13 us run beside the
// other lane's kernels; its code also costs the main loop 6-18 VGPRs.  The product library compiles it out and always takes the
// fix-up launch; arrival_flags is then ignored.
#ifdef MMDYN_LAB
#define MMDYN_INKERNEL_FINISH 1
#else
#define MMDYN_INKERNEL_FINISH 0
#endif
// A tile whose K range straddles the ranges of several blocks is finished by whichever of its pieces' owners ARRIVES LAST -- no
// block ever waits for another one, so there is no progress assumption about dispatch order or co-residency (the lanes' persistent
// kernels share the chip).  Per MFMA wave (the slab regions are per wave and the fragment layout is the same in every block, so
// wave w only needs wave w of the other blocks: no block-level hand-off, the ring's barrier count is untouched): the piece is stored
// WRITE-THROUGH (sc1 stores -- no release fence, which would write back the L2's dirty lines of whatever the other lane is
// running), drained (s_waitcnt vmcnt(0)) and counted on the tile's arrival word (agent-scope fetch-add).  The wave that draws the
// last ticket takes one agent-scope acquire, adds the pieces in piece (= K) order with agent-scope loads -- the additions of
// igemm_wsp_fixup_kernel, bit for bit, whoever arrives last --, puts the word back to zero
// (a replay finds it zero) and runs the epilogue.  `flags`: one word per (block range, MFMA wave), indexed by the range that owns
// the tile's FIRST piece; zero at launch (mmdyn_hip/ops.py hands out a zeroed block per launch site).
// Returns true when this wave finished the tile (`acc` then holds the whole tile and the caller runs the epilogue).
#ifdef MMDYN_LAB
template <int MT, int NT, int NM>
__device__ __forceinline__ bool wsp_arrive_and_sum(f32x4v (&acc)[MT][NT], float* slabs, unsigned* flags, const int ub, const int ue,
                                                   const int per, const int rb, const bool first_seg, const int mw, const int lane) {
  constexpr int SLAB = MT * NT * 256;
  const int b0 = ub / per, npieces = (ue - 1) / per - b0 + 1;
  unsigned* word = flags + b0 * NM + mw;
  // The owner of the tile's FIRST piece reaches it at the END of its range, the others computed theirs first thing: when it finds
  // every other piece counted already (the usual case) it is the last arriver without publishing anything -- its own piece stays
  // in registers and the others are added to it in place, p0 + p1 + ...
  bool head_last = false;
  if (b0 == rb) head_last = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(npieces - 1);
  if (!head_last) {
    float* mine = slabs + ((size_t)(rb * 2 + (first_seg ? 0 : 1)) * NM + mw) * SLAB;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) st_wt(mine + ((mt * NT + nt) * 64 + lane) * 4 + e, acc[mt][nt][e]);      // write-through (sc1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the piece has left the core before it is counted
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __shfl(t, 0, 64);
    if (t != (unsigned)(npieces - 1)) return false;
  }
  // NO acquire fence here: at agent scope it is `buffer_inv sc1`, which drops the XCD's L2 lines -- executed by every finishing wave
  // it kept evicting the operands the other blocks of the XCD were streaming (measured: +27 us per stream-K launch).  The pieces are
  // read with sc1 (agent-coherent) buffer loads instead, which do not take a stale line of an earlier launch's slab from this L2;
  // they are issued after the arrival word's value has returned (the branch above waits for it) and loads return in order.
  asm volatile("" ::: "memory");
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, 0x7FFFFF00, 0x00020000);
  // pieces in K order: 0 + p0 + p1 + ...; a piece of this very wave comes from its registers when it is the first one (head_last),
  // from its slab otherwise (one accumulator set whoever finishes)
  if (!head_last) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  }
  for (int u = head_last ? min(ue, (b0 + 1) * per) : ub; u < ue;) {
    const int b = u / per;
    const unsigned so = (unsigned)(((b * 2 + (u == b * per ? 0 : 1)) * NM + mw) * SLAB) * 4u;     // (slab workspaces are < 2 GB)
    // (four fragments = 16 registers in flight at a time: the kernel has no 32 to spare)
    typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
    constexpr int NF = MT * NT, HB = NF >= 4 ? 4 : NF;
#pragma unroll
    for (int f0 = 0; f0 < NF; f0 += HB) {
      u32x4b w[HB];
#pragma unroll
      for (int i = 0; i < HB; ++i)
        w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((f0 + i) * 64 + lane) * 16), so, 16 /* sc1 */);
#pragma unroll
      for (int i = 0; i < HB; ++i) acc[(f0 + i) / NT][(f0 + i) % NT] += __builtin_bit_cast(f32x4v, w[i]);
    }
    u = min(ue, (b + 1) * per);
  }
  if (lane == 0) __hip_atomic_store(word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
#endif

// X3 (B16 = 0: fp32 operands in HBM and in the ring): the MFMA waves split their fp32 fragments in registers into three bf16
// terms (split3_bf16, common.h) and multiply six of the nine plane pairs on v_mfma_f32_16x16x32_bf16 -- the fp32 product at 6/16 of
// the fp32 matrix time (see X3 at igemm_nt_kernel).  The fragment reads are the fp32 kernel's own: lane (row l&15, quarter h)
// holds the granules h and 4 + h of its row, i.e. eight channels of the 32-deep K-step, and the bf16 MFMA only needs A and B lanes
// of equal h to hold the SAME eight channels in the same order.  Loaders, ring, scheduling and epilogue are unchanged.
template <int MODE, int BM, int BN, int WM, int WN, int S, int B16, bool DIAG = false, bool X3 = false>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NL)) void igemm_wsp_kernel(
    const float* __restrict__ A, const float* __restrict__ Bp, const float* __restrict__ bias, float* __restrict__ C,
    float* __restrict__ C_act, float* __restrict__ stats, float* __restrict__ slabs, const IgemmGeom g, const WspSched sc,
    const unsigned a_bytes, const unsigned b_bytes) {
  constexpr int NM = (BM / WM) * (BN / WN);        // MFMA waves
  constexpr int PA = BM / RPP, PB = BN / RPP;      // DMA pieces per K-step
  static_assert(PA % NL == 0 && PB % NL == 0, "pieces split evenly over the loader waves");
  constexpr int PAL = PA / NL, PBL = PB / NL, PPL = PAL + PBL;
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  static_assert(S - 1 < NRO, "row-offset table deep enough for the loaders' lead");
  constexpr int SLOT = (BM + BN) * RB;
  constexpr int TS = 16, MT = WM / TS, NT = WN / TS;
  constexpr int WAVES_N = BN / WN;
  static_assert(!X3 || B16 == 0, "the three-term split is a variant of the fp32 kernel");
  constexpr int ESZ = B16 ? 2 : 4;
  constexpr int KB = RB / ESZ;                     // channels per K-step

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* rowoff = reinterpret_cast<int*>(smem + S * SLOT);       // [NRO][BM]: output offset of every tile row (-1: none)
  float* trans = reinterpret_cast<float*>(rowoff + NRO * BM);  // [NM][16][TRLD]: the MFMA waves' epilogue patches

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Mg = g.Bg * g.Hr * g.Wr;
  // XCD-aware range order for the k4 s1 p0 layer (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its
  // own L2; block b takes range (b & 7) * (blocks / 8) + (b >> 3), so the 32 blocks of one XCD walk 32 CONSECUTIVE ranges = one
  // (group, 128-sample tile), whose 3.2 MB of input then stay in that XCD's L2 instead of 25.6 MB passing through all eight:
  // HBM-side fetch of the launch 868 -> 438 MB per pair (PMC), time unchanged.  The convolution launches keep the plain order
  // (their ranges share nothing but the weights; the same remap measured 689 -> 930 MB there).
  const int rb = (MODE == MMDYN_TCONV_S1P0 && (gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3))
                                                                    : (int)blockIdx.x;
  const int u0 = rb * sc.per, u1 = min(sc.units, u0 + sc.per);
  if (u0 >= u1) return;
  const int nsteps = u1 - u0;
  const int cin_steps = sc.cin_steps;

  if (wave < NL) {
    // ===================================== loader wave =====================================
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)b_bytes, 0x00020000);
    const int prow = lane >> 3;                          // row of this lane inside a piece
    int rb[PAL], ry[PAL], rx[PAL];
    unsigned voffA[PAL], voffB[PBL];
    int su = u0, seg = -1, seg_end = u0;                 // next unit to issue; running segment; its last unit + 1
    int tap = 0, cstep = 0, ph = 0, pw = 0;
    unsigned sB = 0;
    auto settap = [&]() {
      int dh = 0, dw = 0, wi = 0;
      if (MODE == MMDYN_TCONV_S1P0) {                    // (ph, pw) = the tile's output pixel; tap counts its VALID taps
        const int kh0 = max(0, ph - (g.Hi - 1)), kw0 = max(0, pw - (g.Wi - 1));
        const int nkw = min(3, pw) - kw0 + 1;
        const int a = tap / nkw;
        const int kh = kh0 + a, kw = kw0 + (tap - a * nkw);
        dh = -kh;
        dw = -kw;
        wi = kh * 4 + kw;
      } else if (MODE == MMDYN_CONV) {
        // (stride 2: the four taps of one input-pixel class back to back, as in igemm_ws.hip)
        const int cls = tap >> 2, j = tap & 3;
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
        const int r = (wave + NL * i) * RPP + prow;
        const int y = ry[i] + dh, x = rx[i] + dw;
        const bool ok = (rb[i] >= 0) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
        const unsigned pix = (unsigned)((rb[i] * g.Hi + y) * g.Wi + x);
        voffA[i] = ok ? pix * (unsigned)(g.Cin * ESZ) + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16) : OOB;
      }
    };
    auto next_segment = [&]() {
      ++seg;
      const Seg sg = seg_at<MODE>(su, sc);
      const int kb = su - sg.ub;
      seg_end = min(u1, sg.ub + sg.kt);
      const TileId id = tile_of<MODE>(sg.t, g, sc, BN);
      ph = id.ph;
      pw = id.pw;
      const int row0 = (MODE == MMDYN_TCONV_S1P0 ? id.stile : id.tile) * BM;
#pragma unroll
      for (int i = 0; i < PAL; ++i) {
        const int r = (wave + NL * i) * RPP + prow;      // tile row
        int oo;
        decode_row<MODE>(row0 + r, Mg, id, g, sc, rb[i], ry[i], rx[i], oo);
        if ((lane & 7) == 0) rowoff[(seg & (NRO - 1)) * BM + r] = oo;
      }
#pragma unroll
      for (int j = 0; j < PBL; ++j) {
        const int r = (wave + NL * j) * RPP + prow;
        voffB[j] = (unsigned)((id.grp * g.b_group_stride + (id.n0 + r) * g.Cin) * ESZ + ((lane & 7) ^ ((r >> 1) & 7)) * 16);
      }
      tap = kb / cin_steps;
      cstep = kb - tap * cin_steps;
      settap();
    };
    int islot = 0;                                       // ring slot of the next issue
    auto issue = [&]() {
      if (su == seg_end) next_segment();
      char* slot = smem + islot * SLOT;
      const unsigned so = (unsigned)cstep * RB;
#pragma unroll
      for (int i = 0; i < PAL; ++i) dma16(rsA, slot + (wave + NL * i) * 1024, voffA[i], so);
#pragma unroll
      for (int j = 0; j < PBL; ++j) dma16(rsB, slot + BM * RB + (wave + NL * j) * 1024, voffB[j], sB + so);
      islot = islot + 1 == S ? 0 : islot + 1;
      ++su;
      if (++cstep == cin_steps) {
        cstep = 0;
        ++tap;
        if (su < seg_end) settap();
      }
    };
    for (int k = 0; k < S - 1 && k < nsteps; ++k) issue();
    long long t_wait = 0, t_bar = 0, t_iss = 0, t_all = 0;
    if (DIAG) t_all = -WSP_STAMP();
    for (int k = 0; k < nsteps; ++k) {
      if (DIAG) t_wait -= WSP_STAMP();
      if (k + S - 1 <= nsteps) wait_vmcnt<PPL*(S - 2)>(); else wait_vmcnt<0>();
      if (DIAG) { const long long c = WSP_STAMP(); t_wait += c; t_bar -= c; }
      ring_barrier();                                    // slot k is complete; slot k-1 has been read by every MFMA wave
      if (DIAG) { const long long c = WSP_STAMP(); t_bar += c; t_iss -= c; }
      if (k + S - 1 < nsteps) issue();
      if (DIAG) t_iss += WSP_STAMP();
    }
#ifdef MMDYN_LAB
    if (DIAG && wave == 0 && lane == 0) {
      t_all += WSP_STAMP();
      atomicAdd(&wsp_diag[4], (unsigned long long)t_wait);
      atomicAdd(&wsp_diag[5], (unsigned long long)t_bar);
      atomicAdd(&wsp_diag[6], (unsigned long long)t_iss);
      atomicAdd(&wsp_diag[7], (unsigned long long)t_all);
    }
#endif
    return;
  }

  // ===================================== MFMA waves =====================================
  const int mw = wave - NL;
  const int wm = mw / WAVES_N, wn = mw - wm * WAVES_N;
  const int h = lane >> 4, cl = lane & 15;
  const bool bnbwd = g.bn_y != nullptr;
  const int fr = (cl >> 1) & 7;
  const int foff0 = cl * RB + 16 * ((0 + h) ^ fr), foff1 = cl * RB + 16 * ((4 + h) ^ fr);
  const int abase = wm * WM * RB, bbase = BM * RB + wn * WN * RB;
  int cslot = 0;                                         // ring slot of the next K-step
  int seg = -1;
  long long d_bar = 0, d_loop = 0, d_epi = 0, d_all = 0;
  if (DIAG) d_all = -WSP_STAMP();
  for (int cu = u0; cu < u1;) {
    ++seg;
    const Seg sg = seg_at<MODE>(cu, sc);
    const int kb = cu - sg.ub;
    const int ke = min(sg.kt, kb + (u1 - cu));
    const bool full = kb == 0 && ke == sg.kt;
    const TileId id = tile_of<MODE>(sg.t, g, sc, BN);
    f32x4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    int ooff[MT];                                        // output offset of row (lane >> 2) of each 16-row tile of the wave
    f32x4v yq[MT][NT];                                   // BatchNorm / activation backward epilogue: the tile's saved values
    if (DIAG) d_loop -= WSP_STAMP();
    for (int k = kb; k < ke; ++k) {
      if (DIAG) d_bar -= WSP_STAMP();
      ring_barrier();
      if (DIAG) d_bar += WSP_STAMP();
      if (k == kb) {
        // the loaders published this segment's row offsets before its first K-step; the epilogue's operand is requested NOW
        // and lands under the K loop
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ooff[mt] = rowoff[(seg & (NRO - 1)) * BM + wm * WM + mt * TS + (lane >> 2)];
        if (full && bnbwd) wsp_fetch_y<MT, NT, B16>(yq, ooff, g, id.n0 + wn * WN + (lane & 3) * 4);
      }
      const char* sl = smem + cslot * SLOT;
      cslot = cslot + 1 == S ? 0 : cslot + 1;
      if constexpr (X3) {
        typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
        bf16x8v bp[3][NT];
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
        // one A fragment's planes live at a time (all of them at once: 256 registers and a spill on the 64x64 wave tile)
        constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (plane of A, plane of B), smallest first
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          bf16x8v ap[3];
          split_frag(sl + abase + mt * TS * RB, ap[0], ap[1], ap[2]);
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[order[t][0]], bp[order[t][1]][nt], acc[mt][nt], 0, 0, 0);
        }
      } else
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int fo = q ? foff1 : foff0;
        if constexpr (B16 != 0) {
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
    if (DIAG) { const long long c = WSP_STAMP(); d_loop += c; d_epi -= c; }
    bool finish = full;
    if (!full) {
#ifdef MMDYN_LAB
      if (g.flags != nullptr) {
        // a piece of a split tile, finished inside the launch by the piece that arrives last (wsp_arrive_and_sum; LAB build only)
        finish = wsp_arrive_and_sum<MT, NT, NM>(acc, slabs, g.flags, sg.ub, sg.ub + sg.kt, sc.per, rb, cu == u0, mw, lane);
        if (finish && bnbwd) wsp_fetch_y<MT, NT, B16>(yq, ooff, g, id.n0 + wn * WN + (lane & 3) * 4);
      } else
#endif
      {
        // a piece of a split tile: the accumulator fragments as they are, 16 bytes per lane (slot 0: the piece is this block's
        // first segment, slot 1: its last); igemm_wsp_fixup_kernel sums the pieces and runs the epilogue
        float* sb = slabs + ((size_t)(rb * 2 + (cu == u0 ? 0 : 1)) * NM + mw) * (MT * NT * 256);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4v*>(sb + ((mt * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
      }
    }
    if (finish)
      wsp_epilogue<BM, BN, WM, WN, B16>(acc, ooff, yq, id, g, bias, C, C_act, stats, wm, wn, lane, trans + mw * 16 * TRLD);
    if (DIAG) d_epi += WSP_STAMP();
    cu += ke - kb;
  }
#ifdef MMDYN_LAB
  if (DIAG && mw == 0 && lane == 0) {
    d_all += WSP_STAMP();
    atomicAdd(&wsp_diag[0], (unsigned long long)d_bar);
    atomicAdd(&wsp_diag[1], (unsigned long long)d_loop);
    atomicAdd(&wsp_diag[2], (unsigned long long)d_epi);
    atomicAdd(&wsp_diag[3], (unsigned long long)d_all);
  }
#endif
}

// ---- P3: the fp32x3 arithmetic on operands that ARRIVE SPLIT ("split once, not once per tile") ----
// A and Bp are the exact three-term bf16 split of fp32 tensors, kept in HBM by whoever produced them (mmdyn_split_planes, the
// pack plan, the producing kernels' epilogues): every row -- a pixel of A, an (tap, n) row of Bp -- is [plane][Cin] bf16, hi | mid
// | lo, 6 bytes per element.  A K-step is 32 channels: 64 bytes per plane and row.  Ring image of a K-step: per operand, 16-row
// blocks x 3 planes x 1-KiB pieces ([16 rows][64 B]); a piece is one DMA wave instruction (lane l -> row l >> 2, 16-byte position
// l & 3 holding source granule (l & 3) ^ f(row), f(r) = (r >> 2) & 2) and ONE conflict-free ds_read_b128 per 16x16x32 fragment
// (lane (row r, quarter h) reads position h ^ f(r)).  The MFMA waves execute ds_read_b128 + v_mfma_f32_16x16x32_bf16 only: six
// plane products per fragment pair, smallest first, fp32 accumulate -- the terms and the order of the X3 variant above (the two
// map channels to the k lanes of the 32-deep MFMA differently: equal to the last bits, not bit for bit), with no VALU work in the
// K loop and 48 KB of fills per 128x128x32 K-step instead of the fp32 ring's 32 KB.  Schedule, row decode, slabs, fix-up launch and epilogue are igemm_wsp_kernel's (B16 = 0: fp32 results).
// Measured as a dense GEMM against the split-in-the-kernel structures: tests/microbench/p3_ring_gemm.hip, profiles/r5/.
constexpr int P3_RB = 192;      // bytes per tile row in the ring: three planes of 32 bf16 channels
constexpr int P3_PIECE = 1024;  // 16 rows x 64 bytes
__device__ __forceinline__ int p3_swz(int r) { return (r >> 2) & 2; }

template <int MODE, int BM, int BN, int WM, int WN, int S, int NLD>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + NLD)) void igemm_wsp3_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bp, const float* __restrict__ bias, float* __restrict__ C,
    float* __restrict__ C_act, float* __restrict__ stats, float* __restrict__ slabs, const IgemmGeom g, const WspSched sc,
    const unsigned a_bytes, const unsigned b_bytes) {
  constexpr int NM = (BM / WM) * (BN / WN);        // MFMA waves
  constexpr int RBA = BM / 16, RBB = BN / 16;      // 16-row blocks per operand tile
  static_assert(RBA % NLD == 0 && RBB % NLD == 0, "row blocks split evenly over the loader waves");
  constexpr int RAL = RBA / NLD, RBL = RBB / NLD, PPL = 3 * (RAL + RBL);
  static_assert(PPL * (S - 2) <= 63, "vmcnt is a 6-bit counter");
  static_assert(S - 1 < NRO, "row-offset table deep enough for the loaders' lead");
  constexpr int SLOT = (BM + BN) * P3_RB;
  constexpr int TS = 16, MT = WM / TS, NT = WN / TS;
  constexpr int WAVES_N = BN / WN;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* rowoff = reinterpret_cast<int*>(smem + S * SLOT);       // [NRO][BM]: output offset of every tile row (-1: none)
  float* trans = reinterpret_cast<float*>(rowoff + NRO * BM);  // [NM][16][TRLD]: the MFMA waves' epilogue patches

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Mg = g.Bg * g.Hr * g.Wr;
  const int rb = (MODE == MMDYN_TCONV_S1P0 && (gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3))
                                                                    : (int)blockIdx.x;
  const int u0 = rb * sc.per, u1 = min(sc.units, u0 + sc.per);
  if (u0 >= u1) return;
  const int nsteps = u1 - u0;
  const int cin_steps = sc.cin_steps;

  if (wave < NLD) {
    // ===================================== loader wave =====================================
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)b_bytes, 0x00020000);
    const int prow = lane >> 2;                          // row of this lane inside a 16-row piece
    const unsigned gran = (unsigned)(((lane & 3) ^ p3_swz(prow)) * 16);
    const unsigned plane_b = (unsigned)(g.Cin * 2);      // bytes from one plane of a row to the next
    int rbs[RAL], ry[RAL], rx[RAL];
    unsigned voffA[RAL], voffB[RBL];
    int su = u0, seg = -1, seg_end = u0;
    int tap = 0, cstep = 0, ph = 0, pw = 0;
    unsigned sB = 0;
    auto settap = [&]() {
      int dh = 0, dw = 0, wi = 0;
      if (MODE == MMDYN_TCONV_S1P0) {
        const int kh0 = max(0, ph - (g.Hi - 1)), kw0 = max(0, pw - (g.Wi - 1));
        const int nkw = min(3, pw) - kw0 + 1;
        const int a = tap / nkw;
        const int kh = kh0 + a, kw = kw0 + (tap - a * nkw);
        dh = -kh;
        dw = -kw;
        wi = kh * 4 + kw;
      } else if (MODE == MMDYN_CONV) {
        const int cls = tap >> 2, j = tap & 3;
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
      sB = (unsigned)wi * (unsigned)(g.N * g.Cin) * 6u;
#pragma unroll
      for (int i = 0; i < RAL; ++i) {
        const int y = ry[i] + dh, x = rx[i] + dw;
        const bool ok = (rbs[i] >= 0) & ((unsigned)y < (unsigned)g.Hi) & ((unsigned)x < (unsigned)g.Wi);
        const unsigned pix = (unsigned)((rbs[i] * g.Hi + y) * g.Wi + x);
        voffA[i] = ok ? pix * (unsigned)(g.Cin * 6) + gran : OOB;
      }
    };
    auto next_segment = [&]() {
      ++seg;
      const Seg sg = seg_at<MODE>(su, sc);
      const int kb = su - sg.ub;
      seg_end = min(u1, sg.ub + sg.kt);
      const TileId id = tile_of<MODE>(sg.t, g, sc, BN);
      ph = id.ph;
      pw = id.pw;
      const int row0 = (MODE == MMDYN_TCONV_S1P0 ? id.stile : id.tile) * BM;
#pragma unroll
      for (int i = 0; i < RAL; ++i) {
        const int r = (wave + NLD * i) * 16 + prow;      // tile row
        int oo;
        decode_row<MODE>(row0 + r, Mg, id, g, sc, rbs[i], ry[i], rx[i], oo);
        if ((lane & 3) == 0) rowoff[(seg & (NRO - 1)) * BM + r] = oo;
      }
#pragma unroll
      for (int j = 0; j < RBL; ++j) {
        const int r = (wave + NLD * j) * 16 + prow;
        voffB[j] = (unsigned)(id.grp * g.b_group_stride + (id.n0 + r) * g.Cin) * 6u + gran;
      }
      tap = kb / cin_steps;
      cstep = kb - tap * cin_steps;
      settap();
    };
    int islot = 0;
    auto issue = [&]() {
      if (su == seg_end) next_segment();
      char* slot = smem + islot * SLOT;
      const unsigned so = (unsigned)cstep * 64u;
#pragma unroll
      for (int i = 0; i < RAL; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) dma16(rsA, slot + ((wave + NLD * i) * 3 + p) * P3_PIECE, voffA[i], so + (unsigned)p * plane_b);
#pragma unroll
      for (int j = 0; j < RBL; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          dma16(rsB, slot + BM * P3_RB + ((wave + NLD * j) * 3 + p) * P3_PIECE, voffB[j], sB + so + (unsigned)p * plane_b);
      islot = islot + 1 == S ? 0 : islot + 1;
      ++su;
      if (++cstep == cin_steps) {
        cstep = 0;
        ++tap;
        if (su < seg_end) settap();
      }
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
  const int mw = wave - NLD;
  const int wm = mw / WAVES_N, wn = mw - wm * WAVES_N;
  const int h = lane >> 4, cl = lane & 15;
  const bool bnbwd = g.bn_y != nullptr;
  const int foff = cl * 64 + ((h ^ p3_swz(cl)) * 16);
  const int abase = (wm * WM / 16) * 3 * P3_PIECE, bbase = BM * P3_RB + (wn * WN / 16) * 3 * P3_PIECE;
  int cslot = 0;
  int seg = -1;
  for (int cu = u0; cu < u1;) {
    ++seg;
    const Seg sg = seg_at<MODE>(cu, sc);
    const int kb = cu - sg.ub;
    const int ke = min(sg.kt, kb + (u1 - cu));
    const bool full = kb == 0 && ke == sg.kt;
    const TileId id = tile_of<MODE>(sg.t, g, sc, BN);
    f32x4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    int ooff[MT];
    f32x4v yq[MT][NT];
    for (int k = kb; k < ke; ++k) {
      ring_barrier();
      if (k == kb) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ooff[mt] = rowoff[(seg & (NRO - 1)) * BM + wm * WM + mt * TS + (lane >> 2)];
        if (full && bnbwd) wsp_fetch_y<MT, NT, 0>(yq, ooff, g, id.n0 + wn * WN + (lane & 3) * 4);
      }
      const char* sl = smem + cslot * SLOT;
      cslot = cslot + 1 == S ? 0 : cslot + 1;
      bf16x8v bp[3][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < 3; ++p) bp[p][nt] = *reinterpret_cast<const bf16x8v*>(sl + bbase + (nt * 3 + p) * P3_PIECE + foff);
      constexpr int order[6][2] = {{0, 2}, {2, 0}, {1, 1}, {0, 1}, {1, 0}, {0, 0}};      // (plane of A, plane of B), smallest first
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x8v ap[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ap[p] = *reinterpret_cast<const bf16x8v*>(sl + abase + (mt * 3 + p) * P3_PIECE + foff);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[order[t][0]], bp[order[t][1]][nt], acc[mt][nt], 0, 0, 0);
      }
    }
    bool finish = full;
    if (!full) {
#ifdef MMDYN_LAB
      if (g.flags != nullptr) {
        // a piece of a split tile, finished inside the launch by the piece that arrives last (wsp_arrive_and_sum; LAB build only)
        finish = wsp_arrive_and_sum<MT, NT, NM>(acc, slabs, g.flags, sg.ub, sg.ub + sg.kt, sc.per, rb, cu == u0, mw, lane);
        if (finish && bnbwd) wsp_fetch_y<MT, NT, 0>(yq, ooff, g, id.n0 + wn * WN + (lane & 3) * 4);
      } else
#endif
      {
        float* sb = slabs + ((size_t)(rb * 2 + (cu == u0 ? 0 : 1)) * NM + mw) * (MT * NT * 256);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4v*>(sb + ((mt * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
      }
    }
    if (finish)
      wsp_epilogue<BM, BN, WM, WN, 0, MODE == MMDYN_DENSE>(acc, ooff, yq, id, g, bias, C, C_act, stats, wm, wn, lane, trans + mw * 16 * TRLD);
    cu += ke - kb;
  }
}

// One block per tile; blocks of tiles that were computed whole return at once.  Sums the pieces of a split tile in piece
// (= K) order and runs the epilogue of the main kernel.
template <int MODE, int BM, int BN, int WM, int WN, int B16>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void igemm_wsp_fixup_kernel(
    const float* __restrict__ bias, float* __restrict__ C, float* __restrict__ C_act, float* __restrict__ stats,
    const float* __restrict__ slabs, const IgemmGeom g, const WspSched sc) {
  constexpr int NM = (BM / WM) * (BN / WN);
  constexpr int TS = 16, MT = WM / TS, NT = WN / TS, WAVES_N = BN / WN;
  const int t = blockIdx.x;
  const Seg sg = seg_of_tile<MODE>(t, sc);
  const int ub = sg.ub, ue = sg.ub + sg.kt;
  const int b0 = ub / sc.per;
  if (ue <= (b0 + 1) * sc.per) return;                   // the tile lies inside one block's range: computed whole
  const int lane = threadIdx.x & 63, mw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = mw / WAVES_N, wn = mw - wm * WAVES_N;
  const int Mg = g.Bg * g.Hr * g.Wr;
  const TileId id = tile_of<MODE>(t, g, sc, BN);
  f32x4v acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (int u = ub; u < ue;) {
    const int b = u / sc.per;
    const int pe = min(ue, (b + 1) * sc.per);            // end of this block's piece of the tile
    const float* sb = slabs + ((size_t)(b * 2 + (u == b * sc.per ? 0 : 1)) * NM + mw) * (MT * NT * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] += *reinterpret_cast<const f32x4v*>(sb + ((mt * NT + nt) * 64 + lane) * 4);
    u = pe;
  }
  __shared__ __attribute__((aligned(16))) float trans[NM * 16 * TRLD];
  int ooff[MT];
  f32x4v yq[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int ib, y0, x0;
    decode_row<MODE>((MODE == MMDYN_TCONV_S1P0 ? id.stile : id.tile) * BM + wm * WM + mt * TS + (lane >> 2), Mg, id, g, sc, ib, y0,
                     x0, ooff[mt]);
  }
  if (g.bn_y != nullptr) wsp_fetch_y<MT, NT, B16>(yq, ooff, g, id.n0 + wn * WN + (lane & 3) * 4);
  wsp_epilogue<BM, BN, WM, WN, B16, MODE == MMDYN_DENSE && B16 == 0>(acc, ooff, yq, id, g, bias, C, C_act, stats, wm, wn, lane,
                                                                   trans + mw * 16 * TRLD);
}

// ---- which launches this file serves, and how ----
struct WspPick {
  int bm, bn;      // 0: not served
  int bpc;         // resident blocks per CU the grid is sized for
};

static int device_cus() {
  static int cus[LdsOptIn::MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LdsOptIn::MAX_DEVICES) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t p;
    cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus[dev];
}

// Measured per shape against the one-tile-per-block ring kernels and the register-staged kernels, each launch alone on the
// chip (tests/microbench/ab_wsp.py, profiles/r4/ab_wsp_per_shape.txt).  Served: fp32 (and all-16-bit) CONV / TCONV_S2P1 launches
// with N % 128 == 0 and at least two 16-K-step tiles' worth of work per CU, and the k4 s1 p0 transposed convolution with N = 128.
static WspPick wsp_pick(int mode, int G, int Bg, int Hi, int Wi, int Hr, int Wr, int Cin, int N, int ncls, int splitk, bool b16,
                        int b_group_stride) {
  WspPick p{0, 0, 0};
  if (const char* e = lab_env("MMDYN_WSP"))
    if (e[0] == '0') return p;
  if (mode != MMDYN_CONV && mode != MMDYN_TCONV_S2P1 && mode != MMDYN_DENSE && mode != MMDYN_TCONV_S1P0) return p;
  if (splitk > 1 || b_group_stride) return p;
  // All-16-bit operands: built and tested (tests/test_kernels_gpu.py::test_igemm_all16_persistent) but measured SLOWER in the
  // 16-bit storage modes' steps (same box, profiles/r4/step_ab_wsp_configs.txt: bf16s bs 128 -0.6 %, bs 256 -2.6 %, 256 px -0.3 %):
  // their K loops are DMA round trips at one block per CU, where three 64x64 blocks per CU hide more.  LAB build only
  // (MMDYN_WSP_B16=1).
#ifdef MMDYN_LAB
  if (b16) {
    const char* e = lab_env("MMDYN_WSP_B16");
    if (!(e && e[0] == '1')) return p;
  }
#else
  if (b16) return p;
#endif
  const int kb = b16 ? 64 : BK;
  if (Cin % kb || N % 64) return p;
  const int esz = b16 ? 2 : 4;
  if ((int64_t)G * Bg * Hi * Wi * Cin * esz >= MAX_BUFFER_BYTES || (int64_t)16 * N * Cin * esz >= MAX_BUFFER_BYTES) return p;
  const int cus = device_cus();
  if (mode == MMDYN_TCONV_S1P0) {
    // tiles of 128 samples x ONE output pixel x all 128 channels, 1..16 taps each (stream-K balances them)
    if (const char* e = lab_env("MMDYN_WSP_S1P0"))
      if (e[0] == '0') return p;
    if (N != 128 || Hi != 5 || Wi != 5 || Hr != 8 || Wr != 8) return p;
    const long units = (long)G * ((Bg + 127) / 128) * 400 * (Cin / kb);
    long min_units = (b16 ? 1L : 2L) * 16 * cus;      // (16-bit: a K-step is latency, not arithmetic: even cuts pay earlier)
    if (const char* e = lab_env("MMDYN_WSP_MIN_UNITS")) min_units = atol(e);
    if (units < min_units || units >= (1L << 30)) return p;
    p.bm = p.bn = 128;
    p.bpc = 1;
    return p;
  }
  const long rows_g = (long)Bg * Hr * Wr;
  if (rows_g >= (1L << 23)) return p;                    // float-reciprocal row decode
  const int ksteps = (mode == MMDYN_CONV ? 16 : (mode == MMDYN_TCONV_S2P1 ? 4 : 1)) * (Cin / kb);
  // 128x128 tiles, one block per CU, where N allows: x1.03-1.15 against the one-tile-per-block kernels on the N = 128 / 256
  // launches (more where the tile count does not divide the CUs: 400 tiles x1.07, 256 tiles x1.15).  The 128x64 tile (two
  // blocks per CU) is built and tested but LOSES on the step's N = 64 launches (x0.86 / x0.98): their 16-K-step tiles spend
  // 17-21 % of a block's time in the epilogue and keep the loaders 36-74 % busy issuing (cycle stamps: docs/LAB_NOTES.md E),
  // so it is served only when forced (LAB build).
  if (N % 128 == 0) {
    p.bm = p.bn = 128;
    p.bpc = 1;
  } else {
    if (!lab_env("MMDYN_WSP_TILE")) return p;
    p.bm = 128;
    p.bn = 64;
    p.bpc = 2;
  }
  if (const char* e = lab_env("MMDYN_WSP_TILE")) {       // LAB build: force one tile (kernel experiments)
    int a = 0, b = 0;
    if (sscanf(e, "%d,%d", &a, &b) == 2 && a == 128 && (b == 64 || b == 128) && N % b == 0) {
      p.bn = b;
      p.bpc = b == 128 ? 1 : 2;
    }
  }
  const long tiles = (long)G * ((rows_g + p.bm - 1) / p.bm) * (N / p.bn) * ncls;
  if (tiles * ksteps >= (1L << 30)) return WspPick{0, 0, 0};
  long min_units = 2L * 16 * cus * p.bpc;                // at least two 16-K-step tiles' worth per resident block
  if (const char* e = lab_env("MMDYN_WSP_MIN_UNITS")) min_units = atol(e);
  if (tiles * ksteps < min_units || mode == MMDYN_DENSE) {
    if (!lab_env("MMDYN_WSP_TILE")) return WspPick{0, 0, 0};
  }
  return p;
}

static WspSched make_sched(const IgemmGeom& g, const WspPick& p, bool b16) {
  WspSched sc{};
  sc.cin_steps = g.Cin / (b16 ? 64 : BK);
  if (g.mode == MMDYN_TCONV_S1P0) {
    sc.spg = ceil_div(g.Bg, p.bm);
    sc.ug = 400 * sc.cin_steps;
    sc.ny = sc.si = 1;
    sc.tiles = g.G * sc.spg * 64;
    sc.ksteps = 16 * sc.cin_steps;
    sc.units = g.G * sc.spg * sc.ug;
  } else {
    const int tpg = ceil_div(g.Bg * g.Hr * g.Wr, p.bm);
    sc.ny = g.N / p.bn;
    sc.si = sc.ny * g.nclasses;
    sc.tiles = g.G * tpg * sc.si;
    sc.ksteps = (g.mode == MMDYN_CONV ? 16 : (g.mode == MMDYN_TCONV_S2P1 ? 4 : 1)) * sc.cin_steps;
    sc.units = sc.tiles * sc.ksteps;
  }
  const long nblk = (long)device_cus() * p.bpc;
  sc.per = (int)((sc.units + nblk - 1) / nblk);
  if (const char* e = lab_env("MMDYN_WSP_UNITS_PER_BLOCK")) {      // LAB build: force the cut (kernel tests: split tiles)
    const int v = atoi(e);
    if (v > 0 && (sc.units + v - 1) / v <= 65535L * 16) sc.per = v;
  }
  sc.inv_hw = 1.0f / (float)(g.Hr * g.Wr);
  sc.inv_w = 1.0f / (float)g.Wr;
  return sc;
}

// no tile straddles two blocks' ranges (then there are no slabs and no fix-up launch)
static bool has_split_tiles(const IgemmGeom& g, const WspSched& sc) { return g.mode == MMDYN_TCONV_S1P0 || sc.per % sc.ksteps != 0; }

template <int MODE, int BM, int BN, int WM, int WN, int S, int B16>
static int wsp_launch(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                      IgemmGeom g, const WspSched& sc, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  if constexpr (B16 == 0 && S == 3) {
    if (g.x3) {            // fp32 on the bf16 matrix cores (three-term split in the MFMA waves)
      constexpr int NMx = (BM / WM) * (BN / WN);
      g.tiles_per_group = MODE == MMDYN_TCONV_S1P0 ? 64 * sc.spg : ceil_div(g.Bg * g.Hr * g.Wr, BM);
      const int nblkx = (sc.units + sc.per - 1) / sc.per;
      const size_t smemx = (size_t)S * (BM + BN) * RB + (size_t)NRO * BM * sizeof(int) + (size_t)NMx * 16 * TRLD * sizeof(float);
      static LdsOptIn x3_opt_in;
      if (int e = x3_opt_in.ensure((const void*)igemm_wsp_kernel<MODE, BM, BN, WM, WN, S, 0, false, true>, (int)smemx)) return e;
      const bool splitx = has_split_tiles(g, sc);
      if (splitx && !slabs) return MMDYN_ERR_NULL;
      if (!splitx || nblkx * NMx > MMDYN_IGEMM_FLAG_WORDS || !MMDYN_INKERNEL_FINISH) g.flags = nullptr;     // (flags: split tiles are finished inside the launch)
      hipLaunchKernelGGL((igemm_wsp_kernel<MODE, BM, BN, WM, WN, S, 0, false, true>), dim3(nblkx), dim3(64 * (NMx + NL)), smemx, st, A,
                         Bp, bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes);
      if (splitx && !g.flags)
        hipLaunchKernelGGL((igemm_wsp_fixup_kernel<MODE, BM, BN, WM, WN, 0>), dim3(sc.tiles), dim3(64 * NMx), 0, st, bias, C, C_act,
                           stats, slabs, g, sc);
      MMDYN_LAUNCH_CHECK();
    }
  }
  constexpr int NM = (BM / WM) * (BN / WN);
  // (partial-sum slots per group: M-tiles x classes; TCONV_S1P0: output pixels x sample tiles)
  g.tiles_per_group = MODE == MMDYN_TCONV_S1P0 ? 64 * sc.spg : ceil_div(g.Bg * g.Hr * g.Wr, BM);
  const int nblk = (sc.units + sc.per - 1) / sc.per;
  const size_t smem = (size_t)S * (BM + BN) * RB + (size_t)NRO * BM * sizeof(int) + (size_t)NM * 16 * TRLD * sizeof(float);
  static LdsOptIn lds_opt_in;
  if (int e = lds_opt_in.ensure((const void*)igemm_wsp_kernel<MODE, BM, BN, WM, WN, S, B16>, (int)smem)) return e;
  const bool split = has_split_tiles(g, sc);
  if (split && !slabs) return MMDYN_ERR_NULL;
  if (!split || nblk * NM > MMDYN_IGEMM_FLAG_WORDS || !MMDYN_INKERNEL_FINISH) g.flags = nullptr;
#ifdef MMDYN_LAB
  if constexpr (B16 == 0 && MODE != MMDYN_DENSE) {
    const char* e = lab_env("MMDYN_WSP_DIAG");
    if (e && e[0] == '1') {
      static LdsOptIn diag_opt_in;
      if (int er = diag_opt_in.ensure((const void*)igemm_wsp_kernel<MODE, BM, BN, WM, WN, S, B16, true>, (int)smem)) return er;
      hipLaunchKernelGGL((igemm_wsp_kernel<MODE, BM, BN, WM, WN, S, B16, true>), dim3(nblk), dim3(64 * (NM + NL)), smem, st, A, Bp,
                         bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes);
      if (split && !g.flags)
        hipLaunchKernelGGL((igemm_wsp_fixup_kernel<MODE, BM, BN, WM, WN, B16>), dim3(sc.tiles), dim3(64 * NM), 0, st, bias, C, C_act,
                           stats, slabs, g, sc);
      MMDYN_LAUNCH_CHECK();
    }
  }
#endif
  hipLaunchKernelGGL((igemm_wsp_kernel<MODE, BM, BN, WM, WN, S, B16>), dim3(nblk), dim3(64 * (NM + NL)), smem, st, A, Bp, bias, C,
                     C_act, stats, slabs, g, sc, a_bytes, b_bytes);
  if (split && !g.flags)
    hipLaunchKernelGGL((igemm_wsp_fixup_kernel<MODE, BM, BN, WM, WN, B16>), dim3(sc.tiles), dim3(64 * NM), 0, st, bias, C, C_act,
                       stats, slabs, g, sc);
  MMDYN_LAUNCH_CHECK();
}

template <int MODE, int B16>
static int wsp_launch_mode(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                           const IgemmGeom& g, const WspPick& p, const WspSched& sc, unsigned a_bytes, unsigned b_bytes,
                           hipStream_t st) {
#ifdef MMDYN_LAB
  if constexpr (B16 == 0) {        // LAB experiment: a four-slot ring (three K-steps in flight, 140 KB of LDS)
    const char* e = lab_env("MMDYN_WSP_S");
    if (e && e[0] == '4' && p.bn == 128)
      return wsp_launch<MODE, 128, 128, 64, 32, 4, B16>(A, Bp, bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes, st);
  }
#endif
  if constexpr (B16 == 0) {
    // the split arithmetic runs FOUR MFMA waves of 64x64 instead of eight of 64x32: a wave then splits 64 fragment values for 96
    // MFMAs instead of 48 for 48 (3.7 instead of 5.5 VALU operations per MFMA; one MFMA wave per SIMD): x1.06-1.09 on the launches it
    // serves, step 5.81 -> 5.74 ms on one box (profiles/r4/step_ab_x3_persistent_wave_tile.txt).  LAB: MMDYN_X3_WSP_W64=0 = eight waves.
    const char* e = lab_env("MMDYN_X3_WSP_W64");
    if (g.x3 && p.bn == 128 && !(e && e[0] == '0'))
      return wsp_launch<MODE, 128, 128, 64, 64, 3, B16>(A, Bp, bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes, st);
  }
  if (p.bn == 128) return wsp_launch<MODE, 128, 128, 64, 32, 3, B16>(A, Bp, bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes, st);
  if constexpr (MODE != MMDYN_TCONV_S1P0)
    return wsp_launch<MODE, 128, 64, 64, 32, 3, B16>(A, Bp, bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes, st);
  return MMDYN_ERR_SHAPE;
}

template <int B16>
static int wsp_dispatch(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                        const IgemmGeom& g, const WspPick& p, const WspSched& sc, hipStream_t st) {
  const int esz = B16 ? 2 : 4;
  const unsigned a_bytes = (unsigned)((int64_t)g.G * g.Bg * g.Hi * g.Wi * g.Cin * esz);
  const unsigned b_bytes = (unsigned)((int64_t)(g.mode == MMDYN_DENSE ? 1 : 16) * g.N * g.Cin * esz);
  if (g.mode == MMDYN_DENSE) return wsp_launch_mode<MMDYN_DENSE, B16>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, a_bytes, b_bytes, st);
  if (g.mode == MMDYN_CONV) return wsp_launch_mode<MMDYN_CONV, B16>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, a_bytes, b_bytes, st);
  if (g.mode == MMDYN_TCONV_S1P0)
    return wsp_launch_mode<MMDYN_TCONV_S1P0, B16>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, a_bytes, b_bytes, st);
  return wsp_launch_mode<MMDYN_TCONV_S2P1, B16>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, a_bytes, b_bytes, st);
}

// P3 launch configurations (block tile, MFMA-wave tile, ring slots, resident blocks per CU the grid is sized for).  Four loader
// waves everywhere.  128x128: eight MFMA waves of 64x32 (two per SIMD: one wave's fragment reads land under its partner's MFMAs --
// tests/microbench/p3_ring_gemm.hip: LDS reads + MFMA alone 263 against 219 TFLOP/s for one 64x64 wave per SIMD), three ring slots
// of 48 KB.  N % 128 == 64 layers: 128x64 (36 KB per K-step: 14 flop per filled byte against 21) -- variants measured per shape in
// tests/microbench/ab_p3.py (profiles/r5/ab_p3_*.txt).
constexpr int P3_NLD = 4;
struct Wsp3Cfg {
  int bm, bn, s, bpc;    // bm == 0: not served
};
template <int MODE, int BM, int BN, int WM, int WN, int S>
static int wsp3_launch(const bf16_t* A, const bf16_t* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                       IgemmGeom g, const WspSched& sc, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  constexpr int NM = (BM / WM) * (BN / WN);
  g.tiles_per_group = MODE == MMDYN_TCONV_S1P0 ? 64 * sc.spg : ceil_div(g.Bg * g.Hr * g.Wr, BM);
  const int nblk = (sc.units + sc.per - 1) / sc.per;
  const size_t smem = (size_t)S * (BM + BN) * P3_RB + (size_t)NRO * BM * sizeof(int) + (size_t)NM * 16 * TRLD * sizeof(float);
  static LdsOptIn opt_in;
  if (int e = opt_in.ensure((const void*)igemm_wsp3_kernel<MODE, BM, BN, WM, WN, S, P3_NLD>, (int)smem)) return e;
  const bool split = has_split_tiles(g, sc);
  if (split && !slabs) return MMDYN_ERR_NULL;
  if (!split || nblk * NM > MMDYN_IGEMM_FLAG_WORDS || !MMDYN_INKERNEL_FINISH) g.flags = nullptr;      // (flags: split tiles are finished inside the launch)
  hipLaunchKernelGGL((igemm_wsp3_kernel<MODE, BM, BN, WM, WN, S, P3_NLD>), dim3(nblk), dim3(64 * (NM + P3_NLD)), smem, st, A, Bp,
                     bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes);
  if (split && !g.flags)
    hipLaunchKernelGGL((igemm_wsp_fixup_kernel<MODE, BM, BN, WM, WN, 0>), dim3(sc.tiles), dim3(64 * NM), 0, st, bias, C, C_act,
                       stats, slabs, g, sc);
  MMDYN_LAUNCH_CHECK();
}

// Which launches take their operands already split, and on which tile.  Every (tile, wave tile) pair has BM / WM = 2 wave rows, so
// a launch writes two partial-sum tiles per M-tile and class whatever the configuration (wsp3_stat_tiles).
static Wsp3Cfg wsp3_pick(int mode, int G, int Bg, int Hi, int Wi, int Hr, int Wr, int Cin, int N, int ncls, int splitk,
                         int b_group_stride) {
  Wsp3Cfg c{0, 0, 0, 0};
  if (mode != MMDYN_CONV && mode != MMDYN_TCONV_S2P1 && mode != MMDYN_TCONV_S1P0 && mode != MMDYN_DENSE) return c;
  if (splitk > 1 || b_group_stride || Cin % BK || N % 64) return c;
  if (mode == MMDYN_DENSE) {
    // FC-level GEMMs (round 6): 128x128 tiles of the plane ring, stream-K over all CUs with the split tiles finished inside the
    // launch.  Served where the tiles outnumber their K-steps' pieces: at least 4 K-steps of 128x128 work per CU and at most
    // ~4 pieces per tile (K <= 1024 at one tile per CU) -- the deep-K / few-tile shapes (6400 -> 512 / 256) would hand one block
    // the sum of 16-32 slabs and stay on split-K + reduce.
    if (N % 128 || Hi != 1 || Wi != 1 || Hr != 1 || Wr != 1) return c;
    const int cusd = device_cus();
    const long tiles = (long)G * ((Bg + 127) / 128) * (N / 128), ks = Cin / BK;
    if (tiles * ks < 4L * cusd || tiles * 4 < cusd || tiles * ks >= (1L << 30) || Bg >= (1 << 23)) return c;
    if ((int64_t)G * Bg * Cin * 6 >= MAX_BUFFER_BYTES || (int64_t)N * Cin * 6 >= MAX_BUFFER_BYTES) return c;
    return Wsp3Cfg{128, 128, 3, 1};
  }
  if ((int64_t)G * Bg * Hi * Wi * Cin * 6 >= MAX_BUFFER_BYTES || (int64_t)16 * N * Cin * 6 >= MAX_BUFFER_BYTES) return c;
  const int cus = device_cus();
  const int cin_steps = Cin / BK;
  // Threshold (in 128x128x32 K-step units): 8 per CU.  Measured per shape against the kernels that split inside the GEMM
  // (tests/microbench/ab_p3.py, profiles/r5/ab_p3_tiles_and_small_launches.txt): the plane ring wins on every convolution-level launch of
  // the bs 256 step down to the encoder's 4096-unit ones (x1.2-1.5; the one-group k4 s1 p0 input gradient x2.8 against the
  // register-staged quad walk) -- stream-K cuts even a 100-tile launch into equal ranges for all CUs.  Below ~8 K-steps per CU a
  // launch is all prologue; those keep the fp32-operand kernels.
  long units, min_units = 8L * cus;
  if (const char* e = lab_env("MMDYN_P3_MIN_UNITS")) min_units = atol(e);
  if (mode == MMDYN_TCONV_S1P0) {
    if (N != 128 || Hi != 5 || Wi != 5 || Hr != 8 || Wr != 8) return c;
    units = (long)G * ((Bg + 127) / 128) * 400 * cin_steps;
    if (units < min_units || units >= (1L << 30)) return c;
    return Wsp3Cfg{128, 128, 3, 1};
  }
  const long rows_g = (long)Bg * Hr * Wr;
  if (rows_g >= (1L << 23)) return c;                    // float-reciprocal row decode
  const int ksteps = (mode == MMDYN_CONV ? 16 : 4) * cin_steps;
  if (N % 128 == 0) {
    c = Wsp3Cfg{128, 128, 3, 1};
  } else {
    // N % 128 == 64: 128x64 tiles, three slots (36 KB per K-step).  Against the register-staged split kernels x1.25-1.5 on the
    // step's four N = 64 launches; two slots at two blocks per CU and 256x64 tiles at two slots measured slower on most of them.
    c = Wsp3Cfg{128, 64, 3, 1};
  }
  if (const char* e = lab_env("MMDYN_P3_TILE")) {        // LAB build: force one configuration "BM,BN,S"
    int a = 0, b = 0, sl = 0;
    if (sscanf(e, "%d,%d,%d", &a, &b, &sl) == 3 && N % b == 0) {
      if (a == 128 && b == 128 && sl == 3) c = Wsp3Cfg{128, 128, 3, 1};
      else if (a == 128 && b == 64 && sl == 3) c = Wsp3Cfg{128, 64, 3, 1};
      else if (a == 128 && b == 64 && sl == 2) c = Wsp3Cfg{128, 64, 2, 2};
      else if (a == 256 && b == 64 && sl == 2) c = Wsp3Cfg{256, 64, 2, 1};
    }
  }
  const long tiles = (long)G * ((rows_g + c.bm - 1) / c.bm) * (N / c.bn) * ncls;
  if (tiles * ksteps >= (1L << 30) || tiles * ksteps * (c.bn == 64 ? 1 : 2) < min_units * 2) return Wsp3Cfg{0, 0, 0, 0};
  return c;
}

static WspSched make_sched3(const IgemmGeom& g, const Wsp3Cfg& c) {
  WspPick p{c.bm, c.bn, c.bpc};
  WspSched sc = make_sched(g, p, false);
  if (g.mode == MMDYN_DENSE) {
    // FC-level launches: WHOLE tiles per block.  Their tiles are short (8-16 K-steps), so a stream-K cut splits nearly every tile
    // and the fix-up launch (13 us) plus the slab traffic cost more than the idle CUs of an uneven cut -- which the other lane's
    // kernels use anyway.
    const long nblk = (long)device_cus() * c.bpc;
    sc.per = (int)((sc.tiles + nblk - 1) / nblk) * sc.ksteps;
  }
  return sc;
}

// geometry of a launch as the queries below know it (shape only)
static IgemmGeom query_geom(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  IgemmGeom g{};
  g.mode = mode;
  g.G = G;
  g.Bg = Bg;
  g.Hi = Hi;
  g.Wi = Wi;
  g.Cin = Cin;
  g.N = N;
  g.Hr = Ho;
  g.Wr = Wo;
  g.nclasses = 1;
  if (mode == MMDYN_TCONV_S2P1) {
    g.Hr = Hi;
    g.Wr = Wi;
    g.nclasses = 4;
  }
  return g;
}

}  // namespace

// BatchNorm partial-sum tiles per group the persistent kernel writes for the shape (0: not served): one per M-tile, parity
// class and wave row (TCONV_S1P0: per output pixel, sample tile and wave row)
int mmdyn_igemm_wsp_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, bool b16) {
  const IgemmGeom g = query_geom(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  const WspPick p = wsp_pick(mode, G, Bg, Hi, Wi, g.Hr, g.Wr, Cin, N, g.nclasses, 1, b16, 0);
  if (!p.bm) return 0;
  if (mode == MMDYN_TCONV_S1P0) return 64 * ceil_div(Bg, p.bm) * 2;
  return g.nclasses * ceil_div(Bg * g.Hr * g.Wr, p.bm) * 2;        // both tiles are cut into waves of 64 x 32: two wave rows
}

// bytes of slab workspace the launch needs for its split tiles (0: none, or not served)
int64_t mmdyn_igemm_wsp_slab_bytes(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, bool b16) {
  const IgemmGeom g = query_geom(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  const WspPick p = wsp_pick(mode, G, Bg, Hi, Wi, g.Hr, g.Wr, Cin, N, g.nclasses, 1, b16, 0);
  if (!p.bm) return 0;
  const WspSched sc = make_sched(g, p, b16);
  if (!has_split_tiles(g, sc)) return 0;
  const long nblk = (sc.units + sc.per - 1) / sc.per;
  return (int64_t)nblk * 2 * p.bm * p.bn * 4;
}

// bf16_ops: the launch runs on the 16-bit matrix cores; served only when BOTH operands are 16-bit in HBM
int mmdyn_igemm_wsp_try(const float* A, const float* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                        const IgemmGeom& g_in, bool bf16_ops, hipStream_t st) {
  IgemmGeom g = g_in;
  g.tap_order = g.mode == MMDYN_CONV && g.rs == 2;
  if (const char* e = lab_env("MMDYN_WS_TAPORDER")) g.tap_order = g.tap_order && e[0] != '0';
  if (bf16_ops && (!g.a_b16 || !g.b_b16)) return 1;
  const WspPick p = wsp_pick(g.mode, g.G, g.Bg, g.Hi, g.Wi, g.Hr, g.Wr, g.Cin, g.N, g.nclasses, g.splitk, bf16_ops, g.b_group_stride);
  if (!p.bm) return 1;
  const WspSched sc = make_sched(g, p, bf16_ops);
#ifdef MMDYN_LAB
  if (bf16_ops && g.f16) return wsp_dispatch<2>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, st);
  if (bf16_ops) return wsp_dispatch<1>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, st);
#endif
  return wsp_dispatch<0>(A, Bp, bias, C, C_act, stats, slabs, g, p, sc, st);
}

// The launch with both operands ARRIVING as three-plane bf16 rows (igemm_wsp3_kernel).  Returns 1 when the shape is not served.
bool mmdyn_igemm_wsp3_serves(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  const IgemmGeom g = query_geom(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  return wsp3_pick(mode, G, Bg, Hi, Wi, g.Hr, g.Wr, Cin, N, g.nclasses, 1, 0).bm != 0;
}
// BatchNorm partial-sum tiles per group of a plane launch (0: not served): one per M-tile, parity class and wave row
int mmdyn_igemm_wsp3_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  const IgemmGeom g = query_geom(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  const Wsp3Cfg c = wsp3_pick(mode, G, Bg, Hi, Wi, g.Hr, g.Wr, Cin, N, g.nclasses, 1, 0);
  if (!c.bm) return 0;
  if (mode == MMDYN_TCONV_S1P0) return 64 * ceil_div(Bg, c.bm) * 2;
  return g.nclasses * ceil_div(Bg * g.Hr * g.Wr, c.bm) * 2;
}
int64_t mmdyn_igemm_wsp3_slab_bytes(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N) {
  const IgemmGeom g = query_geom(mode, G, Bg, Hi, Wi, Cin, Ho, Wo, N);
  const Wsp3Cfg c = wsp3_pick(mode, G, Bg, Hi, Wi, g.Hr, g.Wr, Cin, N, g.nclasses, 1, 0);
  if (!c.bm) return 0;
  const WspSched sc = make_sched3(g, c);
  if (!has_split_tiles(g, sc)) return 0;
  const long nblk = (sc.units + sc.per - 1) / sc.per;
  return (int64_t)nblk * 2 * c.bm * c.bn * 4;
}

int mmdyn_igemm_wsp3_try(const void* A, const void* Bp, const float* bias, float* C, float* C_act, float* stats, float* slabs,
                         const IgemmGeom& g_in, hipStream_t st) {
  IgemmGeom g = g_in;
  const Wsp3Cfg c = wsp3_pick(g.mode, g.G, g.Bg, g.Hi, g.Wi, g.Hr, g.Wr, g.Cin, g.N, g.nclasses, g.splitk, g.b_group_stride);
  if (!c.bm) return 1;
  g.tap_order = g.mode == MMDYN_CONV && g.rs == 2;
  const WspSched sc = make_sched3(g, c);
  const unsigned a_bytes = (unsigned)((int64_t)g.G * g.Bg * g.Hi * g.Wi * g.Cin * 6);
  const unsigned b_bytes = (unsigned)((int64_t)(g.mode == MMDYN_DENSE ? 1 : 16) * g.N * g.Cin * 6);
  const bf16_t* Ap = reinterpret_cast<const bf16_t*>(A);
  const bf16_t* Bq = reinterpret_cast<const bf16_t*>(Bp);
#define P3_GO(MODE_, BM_, BN_, WM_, WN_, S_) \
  return wsp3_launch<MODE_, BM_, BN_, WM_, WN_, S_>(Ap, Bq, bias, C, C_act, stats, slabs, g, sc, a_bytes, b_bytes, st)
#ifdef MMDYN_LAB
  // LAB, MMDYN_P3_W64=1: the 128x128 tile on FOUR MFMA waves of 64x64 (one per SIMD; 0.25 fragment reads per MFMA instead of 0.375)
  if (const char* e = lab_env("MMDYN_P3_W64"))
    if (e[0] == '1' && c.bn == 128 && c.bm == 128) {
      if (g.mode == MMDYN_TCONV_S1P0) P3_GO(MMDYN_TCONV_S1P0, 128, 128, 64, 64, 3);
      if (g.mode == MMDYN_CONV) P3_GO(MMDYN_CONV, 128, 128, 64, 64, 3);
      if (g.mode == MMDYN_TCONV_S2P1) P3_GO(MMDYN_TCONV_S2P1, 128, 128, 64, 64, 3);
    }
#endif
  if (g.mode == MMDYN_TCONV_S1P0) P3_GO(MMDYN_TCONV_S1P0, 128, 128, 64, 32, 3);
  if (g.mode == MMDYN_DENSE) P3_GO(MMDYN_DENSE, 128, 128, 64, 32, 3);
  if (g.mode == MMDYN_CONV) {
    if (c.bn == 128) P3_GO(MMDYN_CONV, 128, 128, 64, 32, 3);
    if (c.bm == 256) P3_GO(MMDYN_CONV, 256, 64, 128, 16, 2);
    if (c.s == 2) P3_GO(MMDYN_CONV, 128, 64, 64, 16, 2);
    P3_GO(MMDYN_CONV, 128, 64, 64, 16, 3);
  }
  if (c.bn == 128) P3_GO(MMDYN_TCONV_S2P1, 128, 128, 64, 32, 3);
  if (c.bm == 256) P3_GO(MMDYN_TCONV_S2P1, 256, 64, 128, 16, 2);
  if (c.s == 2) P3_GO(MMDYN_TCONV_S2P1, 128, 64, 64, 16, 2);
  P3_GO(MMDYN_TCONV_S2P1, 128, 64, 64, 16, 3);
#undef P3_GO
}

#ifdef MMDYN_LAB
// LAB build: read (and clear) the cycle stamps of the MMDYN_WSP_DIAG=1 launches since the last call (out: 8 x uint64, host)
extern "C" int mmdyn_lab_wsp_diag(unsigned long long* out) {
  if (!out) return MMDYN_ERR_NULL;
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(wsp_diag), sizeof(unsigned long long) * 8);
  if (e != hipSuccess) return (int)e;
  unsigned long long zero[8] = {};
  e = hipMemcpyToSymbol(HIP_SYMBOL(wsp_diag), zero, sizeof(zero));
  return e == hipSuccess ? MMDYN_OK : (int)e;
}
#endif
