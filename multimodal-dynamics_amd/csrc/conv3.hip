// Direct MFMA kernels for the two 3-channel layers of every image network: the encoder's first Conv2d(3, 32, k4 s2 p1)
// and the decoder's last ConvTranspose2d(32, 3, k4 s2 p1) (/root/reference/mmdyn/pytorch/models/vae.py:198, 277).
// Both reduce to the same geometry (MMDYN_IM2COL3): a GEMM row is an output pixel of the 32-channel side, its 48
// "K" entries are the k4 s2 p1 window of the NCHW 3-channel image, k = ci*16 + kh*4 + kw.
//
// These GEMMs are skinny (N = 32, K = 48, a million rows at G*Bg = 1024): 1.5 kFLOP per row against 256 B of
// activations, i.e. bound by HBM, not by the matrix cores.  The generic tiled kernels spend their time in the
// per-tile prologue (two K-steps only) and in 4-byte window gathers through LDS.  Here instead
//   * conv3_nt (forward of conv1 / input gradient of the last tconv): a wave owns one output image row (32 pixels
//     = the 32 rows of one 32x32 MFMA tile) and reads its A fragments STRAIGHT from the image -- with the K order
//     permuted so that lane half h takes kw = h, h+2, the 64 lanes of one load cover 64 consecutive floats of one
//     image row.  The weights live in 24 VGPRs for the life of the (persistent) block.  No LDS on the operand path.
//   * conv3_wgrad (weight gradient of both layers): the reduction runs over pixels, so the window is the MFMA B
//     operand and is gathered per lane.  Each wave stages the 12 image rows its output row touches in a private,
//     zero-padded LDS patch (coalesced 16-byte loads, double buffered, no block barrier in the loop) and gathers
//     from there conflict-free; the 32-channel operand is read from HBM directly in fragment order (64 lanes =
//     256 contiguous bytes).
// fp32 matrix cores in every precision mode: there is nothing to gain from bf16 operands on an HBM-bound layer.
#include "common.h"

namespace {

struct Conv3Geom {
  int G, Bg, Hi, Wi, Ho, ldc;
  int act, want_act_out, want_stats, w_b16;
  int tiles_per_group;   // Bg*Ho*32/128
  const void* bn_y;
  const float* bn_mean;
  const float* bn_rstd;
  const float* bn_gamma;
  const float* bn_beta;
};

// A wave-uniform pointer, pinned into a scalar register pair (and kept in the GLOBAL address space): the compiler then
// emits the (SGPR base + 32-bit VGPR offset) form of global_load / global_store instead of building a 64-bit address
// per access on the vector ALU
#define GLOBAL_AS __attribute__((address_space(1)))
typedef GLOBAL_AS char* gchar_p;
__device__ __forceinline__ gchar_p sgpr_ptr(const void* p) {
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return (gchar_p)(((uint64_t)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ float gld1(gchar_p p);
template <> __device__ __forceinline__ float gld1<float>(gchar_p p) { return *(GLOBAL_AS const float*)p; }
template <> __device__ __forceinline__ float gld1<bf16_t>(gchar_p p) {
  return __uint_as_float((uint32_t)(*(GLOBAL_AS const bf16_t*)p) << 16);
}
template <> __device__ __forceinline__ float gld1<half_t>(gchar_p p) { return h2f(*(GLOBAL_AS const uint16_t*)p); }
template <typename T> struct store16 { typedef bf16_t type; };      // (the 16-bit packing a store of type T uses)
template <> struct store16<half_t> { typedef half_t type; };

// slot j (0..23) of lane half h is the window element (ci, kh, kw) = (j>>3, (j>>1)&3, 2*(j&1)+h); MFMA j multiplies
// slot j of A and B, so any bijection works as long as both operands use it.
// The kernel is specialised to square images of 64 * SEGS pixels a side (SEGS = 1: the reference's 64x64; 2 / 4: the 128 /
// 256 pixel extensions of models/shapes.py), N = ldc = 32 channels.  The unit of work of a wave is one 32-pixel segment
// of an output image row (SEGS segments per row); units are numbered (sample, row, segment), which is also their order
// in the channels-last output, so unit u owns the 32 x 32 output block at byte offset u * OROW_B.
// BN: BatchNorm+Swish backward epilogue; ACT: -1 = no second output, else the activation of the second output
constexpr int C3_N = 32;
template <typename T, bool BN, int ACT, int SEGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3_nt_kernel(
    const float* __restrict__ img, const float* __restrict__ Wp, T* __restrict__ C, T* __restrict__ C_act,
    float* __restrict__ stats, const Conv3Geom g) {
  __shared__ float Ws[32 * 65];
  __shared__ float red[4][2][32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: row / sample arithmetic stays scalar
  const int x = lane & 31, h = lane >> 5;
  for (int i = tid; i < 512; i += 256) {           // Wp: [32 n][64 k] (k >= 48 zero, never read), fp32 or bf16
    const f32x4 v = g.w_b16 == 2 ? ld4<half_t>(reinterpret_cast<const half_t*>(Wp) + 4 * i)
                    : g.w_b16 ? ld4<bf16_t>(reinterpret_cast<const bf16_t*>(Wp) + 4 * i)
                              : reinterpret_cast<const f32x4*>(Wp)[i];
    const int n = i >> 4, k = (i & 15) * 4;
    Ws[n * 65 + k + 0] = v[0];
    Ws[n * 65 + k + 1] = v[1];
    Ws[n * 65 + k + 2] = v[2];
    Ws[n * 65 + k + 3] = v[3];
  }
  __syncthreads();
  // weight fragments: kept in 24 VGPRs for the life of the block, except in the BatchNorm-backward variant, which
  // needs the registers for the prefetched pre-BN values and re-reads them from LDS per tile instead
  float bw[24];
  auto load_bw = [&]() {
#pragma unroll
    for (int j = 0; j < 24; ++j) bw[j] = Ws[x * 65 + (j >> 3) * 16 + ((j >> 1) & 3) * 4 + 2 * (j & 1) + h];
  };
  if (!BN) load_bw();

  const float bn_g = BN ? g.bn_gamma[x] : 0.f, bn_b = BN ? g.bn_beta[x] : 0.f;
  const int ntiles = g.G * g.tiles_per_group;

  // Addressing is kept off the vector ALU: every access is (uniform 64-bit base, scalar instructions) + (one of three
  // per-lane BYTE offsets that never change) + (a compile-time immediate), the saddr form of global_load/store.
  constexpr int C3_HI = 64 * SEGS, C3_HO = 32 * SEGS;
  constexpr int UNITS = C3_HO * SEGS;                                  // units per sample
  int xl[2];                                                           // input column of the lane inside its segment
#pragma unroll
  for (int q = 0; q < 2; ++q) xl[q] = 2 * x - 1 + 2 * q + h;           // -1 .. 64
  const unsigned voffB = (4 * h * C3_N + x) * sizeof(T);              // pixel +4h of the unit, channel x
  constexpr unsigned IMG_B = 3 * C3_HI * C3_HI * sizeof(float);       // bytes per sample of the image
  constexpr unsigned OROW_B = 32 * C3_N * sizeof(T);                  // bytes per unit of output (32 px x 32 ch)

  // tile t = 128 GEMM rows = units 4t .. 4t+3 (wave w takes unit 4t + w); UNITS / 4 tiles per sample
  struct TileAt {
    gchar_p imgb;       // the sample's image, advanced to the unit's segment (+ seg * 64 columns)
    size_t orow;        // byte offset of the wave's output unit (the same in C, C_act and bn_y)
    int y, seg;
    unsigned xoB[2];    // per-lane byte offsets of the two column phases (clamped to a valid address when outside)
    bool xok[2];
  };
  auto locate = [&](int t) {
    TileAt p;
    const int u = t * 4 + wave;                                        // wave-uniform
    const int ib = u / UNITS, rem = u - ib * UNITS;
    p.y = rem / SEGS;
    p.seg = rem - p.y * SEGS;
    // (one float in front of the segment, so that the per-lane offsets below are never negative: the 32-bit lane offset of
    //  the scalar-base addressing form is unsigned)
    p.imgb = sgpr_ptr(reinterpret_cast<const char*>(img) + (size_t)ib * IMG_B + (size_t)p.seg * 256 - 4);
    p.orow = (size_t)u * OROW_B;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      p.xok[q] = (unsigned)(p.seg * 64 + xl[q]) < (unsigned)C3_HI;
      p.xoB[q] = (unsigned)((p.xok[q] ? xl[q] + 1 : 1) * 4);          // outside the image: the segment's first column
    }
    return p;
  };
  auto load_a = [&](const TileAt& p, int j) {         // window slot j of the wave's 32 pixels: one 256-byte load
    const int ci = j >> 3, yin = 2 * p.y - 1 + ((j >> 1) & 3);            // wave-uniform
    const bool yok = (unsigned)yin < (unsigned)C3_HI;
    const gchar_p rowp = p.imgb + (unsigned)((ci * C3_HI + (yok ? yin : 0)) * C3_HI) * 4u;
    return gld1<float>(rowp + p.xoB[j & 1]);
  };
  auto erow = [](int e) { return (unsigned)(((e & 3) + 8 * (e >> 2)) * C3_N * sizeof(T)); };   // immediate offsets

  // Software pipeline over the block's tiles: the loads of tile t+1 are issued behind the MFMA / epilogue step
  // that consumed the same register of tile t, so they fly during the rest of tile t (no second register set).
  float a[24], yv[16];
  int t = blockIdx.x;
  if (t >= ntiles) return;
  TileAt cur = locate(t);
#pragma unroll
  for (int j = 0; j < 24; ++j) a[j] = load_a(cur, j);
  if (BN) {
    const gchar_p yb = sgpr_ptr(reinterpret_cast<const char*>(g.bn_y) + cur.orow) + voffB;
#pragma unroll
    for (int e = 0; e < 16; ++e) yv[e] = gld1<T>(yb + erow(e));
  }
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x < ntiles ? t + gridDim.x : t;      // last tile: harmless re-read of its own data
    const TileAt nxt = locate(tn);
    const int grp = (BN || g.want_stats) ? __builtin_amdgcn_readfirstlane((t / (UNITS / 4)) / g.Bg) : 0;
    if (BN) load_bw();
    float bn_m = 0.f, bn_r = 0.f;
    if (BN) {
      bn_m = g.bn_mean[grp * 32 + x];
      bn_r = g.bn_rstd[grp * 32 + x];
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      const int yin = 2 * cur.y - 1 + ((j >> 1) & 3);
      const bool ok = ((unsigned)yin < (unsigned)C3_HI) & cur.xok[j & 1];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? a[j] : 0.f, bw[j], acc, 0, 0, 0);
      a[j] = load_a(nxt, j);
    }

    const gchar_p cb = sgpr_ptr(reinterpret_cast<const char*>(C) + cur.orow) + voffB;
    const gchar_p ab = sgpr_ptr(reinterpret_cast<const char*>(C_act) + cur.orow) + voffB;
    const gchar_p yb = sgpr_ptr(reinterpret_cast<const char*>(g.bn_y) + nxt.orow) + voffB;
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[e];
      if (BN) {
        const float xh = (yv[e] - bn_m) * bn_r;
        yv[e] = gld1<T>(yb + erow(e));
        v *= swish_gradf_(bn_g * xh + bn_b);
        cs += v;
        cq += v * xh;
      } else {
        cs += v;
        cq += v * v;
      }
      if (sizeof(T) == 2) {
        // adjacent channels sit in adjacent lanes: the even lane stores both as one dword
        const float vn = __shfl_down(v, 1, 64);
        if (!(x & 1)) {
          *(GLOBAL_AS uint32_t*)(cb + erow(e)) = pack2<typename store16<T>::type>(v, vn);
          if (ACT >= 0)
            *(GLOBAL_AS uint32_t*)(ab + erow(e)) = pack2<typename store16<T>::type>(apply_act(v, ACT), apply_act(vn, ACT));
        }
      } else {
        *(GLOBAL_AS float*)(cb + erow(e)) = v;
        if (ACT >= 0) *(GLOBAL_AS float*)(ab + erow(e)) = apply_act(v, ACT);
      }
    }
    if (g.want_stats) {
      cs += __shfl_xor(cs, 32, 64);
      cq += __shfl_xor(cq, 32, 64);
      if (h == 0) {
        red[wave][0][x] = cs;
        red[wave][1][x] = cq;
      }
      __syncthreads();
      if (tid < 32) {
        float s0 = 0.f, q0 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s0 += red[w][0][tid];
          q0 += red[w][1][tid];
        }
        const size_t sb = ((size_t)t * 2) * 32 + tid;      // slot = grp * tiles_per_group + tile = t
        stats[sb] = s0;
        stats[sb + 32] = q0;
      }
      __syncthreads();
    }
    cur = nxt;
  }
}

// ---- weight gradient --------------------------------------------------------------------------------------------
struct Conv3WgradGeom {
  int Bt, Hr, Hi, Wi;       // Wr = Hr = 32 * SEGS
  int img_rows;             // Bt*Hr*SEGS units (32-pixel segments of output image rows) = K-blocks of 32 pixels
  int rows_per_chunk;       // units per block
  // BNACT: the dense operand is swish(BatchNorm(D)) of the layer's pre-BatchNorm tensor, recomputed on the fetch
  const float* bn_mean;     // [G][32]
  const float* bn_rstd;
  const float* bn_gamma;    // [32]
  const float* bn_beta;
  int Bg;                   // samples per BatchNorm group
};

constexpr int PATCH_LD = 76;                 // 3 pad + 1 (x = -1) + 64 + 1 (x = 64) + pad; 76 % 32 = 12 spreads the
constexpr int PATCH_ROWS = 13;               // 8 rows of one fragment over all banks; row 12 stays zero (k >= 48)
constexpr int PATCH = PATCH_ROWS * PATCH_LD;

// SEGS as in conv3_nt_kernel: images of 64 * SEGS pixels a side; a K-block is one 32-pixel segment of an output row, its
// patch the 12 input rows x 64 columns under it plus one halo column on either side (zero at the image border)
// BNACT (round 4): D is the PRE-BatchNorm output y of the decoder's last BatchNorm layer and the kernel multiplies with
// a = swish(gamma * ((y - mean[g]) * rstd[g]) + beta), recomputed as the fragments arrive (a lane always holds channel lane & 31:
// four constants per lane and group; ~8 VALU instructions per element in the shadow of the 64-cycle MFMAs) -- the activated
// tensor the forward no longer writes (tconv_out3.hip, FUSED) is not needed here either.
template <typename TD, int SEGS, bool BNACT = false>
__global__ __launch_bounds__(256) void conv3_wgrad_kernel(const TD* __restrict__ D, const float* __restrict__ img,
                                                          float* __restrict__ partial, const Conv3WgradGeom g) {
  __shared__ __attribute__((aligned(16))) float smem[8192];       // 4 waves x 2 patches (7904 floats); reused as [4][32][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, kk = lane >> 5;
  for (int i = tid; i < 8192; i += 256) smem[i] = 0.f;
  __syncthreads();
  float* patch = smem + wave * 2 * PATCH;

  // gather offsets of the two B fragments (cg = n and cg = 32 + n) inside a patch, for pixel x = kk (+ 2j later)
  int lofs[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int cg = nt * 32 + n;
    const int r = cg < 48 ? (cg >> 2) : 12;
    lofs[nt] = r * PATCH_LD + 3 + (cg & 3) + 2 * kk;
  }
  const int it_begin = blockIdx.x * g.rows_per_chunk;
  const int it_end = min(g.img_rows, it_begin + g.rows_per_chunk);

  f32x4 ri[3];
  float rh = 0.f;                               // SEGS > 1: halo column (lanes 0..23: row lane>>1, side lane&1)
  float dc[16], dn[16];
  auto load_img = [&](int it) {                 // the 12 input rows (3 ci x 4 kh) under unit `it`
    const int b = it / (g.Hr * SEGS), rem = it - b * (g.Hr * SEGS);
    const int y = rem / SEGS, seg = rem - y * SEGS;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r = (lane >> 4) + 4 * i;        // r = ci*4 + kh with ci = i
      const int yin = 2 * y - 1 + (r & 3);
      const bool ok = (unsigned)yin < (unsigned)g.Hi;
      const f32x4 v = *reinterpret_cast<const f32x4*>(img + ((size_t)(b * 3 + i) * g.Hi + (ok ? yin : 0)) * g.Wi +
                                                      seg * 64 + (lane & 15) * 4);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      ri[i] = ok ? v : z;
    }
    if constexpr (SEGS > 1) {
      const int r = (lane >> 1) % 12, side = lane & 1;
      const int yin = 2 * y - 1 + (r & 3);
      const int xin = side ? seg * 64 + 64 : seg * 64 - 1;
      const bool ok = (lane < 24) & ((unsigned)yin < (unsigned)g.Hi) & ((unsigned)xin < (unsigned)g.Wi);
      const float v = img[((size_t)(b * 3 + (r >> 2)) * g.Hi + (ok ? yin : 0)) * g.Wi + (ok ? xin : 0)];
      rh = ok ? v : 0.f;
    }
  };
  auto store_img = [&](float* p) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      *reinterpret_cast<f32x4*>(p + ((lane >> 4) + 4 * i) * PATCH_LD + 4 + (lane & 15) * 4) = ri[i];
    if constexpr (SEGS > 1) {
      if (lane < 24) p[(lane >> 1) * PATCH_LD + ((lane & 1) ? 68 : 3)] = rh;
    }
  };
  float bn_m = 0.f, bn_r = 1.f, bn_g = 1.f, bn_b = 0.f;
  int bn_grp = -1;
  if constexpr (BNACT) {
    bn_g = g.bn_gamma[n];
    bn_b = g.bn_beta[n];
  }
  auto load_d = [&](int it, float* d) {         // fragment order: lane (cd = n, pixel 2j + kk) = 64 consecutive elements
    const TD* p = D + (size_t)it * 32 * 32 + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) d[j] = ld1<TD>(p + 64 * j);
    if constexpr (BNACT) {
      const int grp = (it / (g.Hr * SEGS)) / g.Bg;          // (wave-uniform; changes a handful of times per block)
      if (grp != bn_grp) {
        bn_grp = grp;
        bn_m = g.bn_mean[grp * 32 + n];
        bn_r = g.bn_rstd[grp * 32 + n];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) d[j] = swishf_(bn_g * ((d[j] - bn_m) * bn_r) + bn_b);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

  int it = it_begin + wave;
  int cur = 0;
  if (it < it_end) {
    load_img(it);
    load_d(it, dc);
    store_img(patch);
  }
  for (; it < it_end; it += 4) {
    const bool more = it + 4 < it_end;
    const int nx = more ? it + 4 : it;          // the last iteration re-reads its own rows (branch-free body)
    load_img(nx);
    load_d(nx, dn);
    const float* p = patch + cur * PATCH;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float b0 = p[lofs[0] + 4 * j], b1 = p[lofs[1] + 4 * j];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[j], b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[j], b1, acc[1], 0, 0, 0);
    }
    store_img(patch + (cur ^ 1) * PATCH);
#pragma unroll
    for (int j = 0; j < 16; ++j) dc[j] = dn[j];
    cur ^= 1;
  }

  // block reduction of the four waves' 32x64 accumulators, then one slab per block
  __syncthreads();
  float* redw = smem + wave * 2048;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) redw[((e & 3) + 8 * (e >> 2) + 4 * kk) * 64 + nt * 32 + n] = acc[nt][e];
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 2048;
  for (int i = tid; i < 512; i += 256) {
    f32x4 s = reinterpret_cast<const f32x4*>(smem)[i];
    s += reinterpret_cast<const f32x4*>(smem + 2048)[i];
    s += reinterpret_cast<const f32x4*>(smem + 4096)[i];
    s += reinterpret_cast<const f32x4*>(smem + 6144)[i];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}

// persistent grid: four blocks per CU stay resident (<= 128 VGPRs), each walks ntiles/1024 tiles (whole-step sweep of
// 512 / 768 / 1024 / 2048 blocks: 1024 is best in fp32 and bf16s)
constexpr int CONV3_GRID = 1024;

}  // namespace

// Returns MMDYN_OK / an error, or 1 when the shape is not one of this file's (the caller then takes the generic path).
int mmdyn_conv3_nt_try(const float* A, const float* Bp, const float* bias, void* C, void* C_act, float* stats, int G,
                       int Bg, int Hi, int Wi, int Ho, int Wo, int N, int ldc, int act, int splitk, const void* bn_y,
                       const float* bn_mean, const float* bn_rstd, const float* bn_gamma, const float* bn_beta,
                       int c_b16, int bny_b16, int b_b16, hipStream_t st) {
  if (N != C3_N || ldc != C3_N || Wo != Ho || Wi != Hi || Hi != 2 * Ho || (Hi != 64 && Hi != 128 && Hi != 256) || bias ||
      splitk != 1)
    return 1;
  const int segs = Hi / 64;
  if (bn_y && (c_b16 != bny_b16 || C_act)) return 1;
  if (C_act && act != MMDYN_ACT_NONE && act != MMDYN_ACT_SWISH && act != MMDYN_ACT_RELU) return 1;
  if ((int64_t)G * Bg * Ho * Ho * ldc >= (1LL << 31) || (int64_t)G * Bg * Ho * Ho / 32 >= (1LL << 29)) return 1;
  Conv3Geom g{};
  g.G = G;
  g.Bg = Bg;
  g.Hi = Hi;
  g.Wi = Wi;
  g.Ho = Ho;
  g.ldc = ldc;
  g.act = act;
  g.want_act_out = C_act != nullptr;
  g.want_stats = stats != nullptr;
  g.w_b16 = b_b16;
  g.tiles_per_group = Bg * Ho * Ho / 128;
  g.bn_y = bn_y;
  g.bn_mean = bn_mean;
  g.bn_rstd = bn_rstd;
  g.bn_gamma = bn_gamma;
  g.bn_beta = bn_beta;
  const int ntiles = G * g.tiles_per_group;
  const dim3 grid(ntiles < CONV3_GRID ? ntiles : CONV3_GRID);
  const int variant = bn_y ? 4 : (C_act ? act : -1);
#define CONV3_LAUNCH(T_, BN_, ACT_, S_)                                                                            \
  hipLaunchKernelGGL((conv3_nt_kernel<T_, BN_, ACT_, S_>), grid, dim3(256), 0, st, A, Bp, (T_*)C, (T_*)C_act, stats, g)
#define CONV3_CASE(V, BN_, ACT_)                                                                                   \
  if (variant == (V)) {                                                                                            \
    if (c_b16 == 2) {                                                                                              \
      if (segs == 1) CONV3_LAUNCH(half_t, BN_, ACT_, 1);                                                           \
      else if (segs == 2) CONV3_LAUNCH(half_t, BN_, ACT_, 2);                                                      \
      else CONV3_LAUNCH(half_t, BN_, ACT_, 4);                                                                     \
    } else if (c_b16) {                                                                                            \
      if (segs == 1) CONV3_LAUNCH(bf16_t, BN_, ACT_, 1);                                                           \
      else if (segs == 2) CONV3_LAUNCH(bf16_t, BN_, ACT_, 2);                                                      \
      else CONV3_LAUNCH(bf16_t, BN_, ACT_, 4);                                                                     \
    } else {                                                                                                       \
      if (segs == 1) CONV3_LAUNCH(float, BN_, ACT_, 1);                                                            \
      else if (segs == 2) CONV3_LAUNCH(float, BN_, ACT_, 2);                                                       \
      else CONV3_LAUNCH(float, BN_, ACT_, 4);                                                                      \
    }                                                                                                              \
  }
  CONV3_CASE(4, true, -1)
  CONV3_CASE(-1, false, -1)
  CONV3_CASE(MMDYN_ACT_NONE, false, MMDYN_ACT_NONE)
  CONV3_CASE(MMDYN_ACT_SWISH, false, MMDYN_ACT_SWISH)
  CONV3_CASE(MMDYN_ACT_RELU, false, MMDYN_ACT_RELU)
#undef CONV3_LAUNCH
#undef CONV3_CASE
  MMDYN_LAUNCH_CHECK();
}

// chunks = partial slabs = blocks; 1 when the shape is not this file's
// bn_mean != nullptr: D is the pre-BatchNorm tensor, the operand swish(BatchNorm(D)) (groups of Bg samples)
int mmdyn_conv3_wgrad_try(const void* D, const float* Gt, float* partial, int Bt, int Hr, int Wr, int Cd, int Hi,
                          int Wi, int Cg, int chunks, int d_b16, hipStream_t st, const float* bn_mean, const float* bn_rstd,
                          const float* bn_gamma, const float* bn_beta, int Bg) {
  if (Cd != 32 || Cg != 64 || Wr != Hr || Wi != Hi || Hi != 2 * Hr || (Hi != 64 && Hi != 128 && Hi != 256)) return 1;
  const int segs = Hi / 64;
  if ((int64_t)Bt * Hr * segs >= (1LL << 30)) return 1;
  Conv3WgradGeom g{};
  g.Bt = Bt;
  g.Hr = Hr;
  g.Hi = Hi;
  g.Wi = Wi;
  g.img_rows = Bt * Hr * segs;
  g.rows_per_chunk = ceil_div(g.img_rows, chunks);
  g.bn_mean = bn_mean;
  g.bn_rstd = bn_rstd;
  g.bn_gamma = bn_gamma;
  g.bn_beta = bn_beta;
  g.Bg = Bg > 0 ? Bg : Bt;
#define CONV3_WG(T_, S_)                                                                                                  \
  do {                                                                                                                    \
    if (bn_mean)                                                                                                          \
      hipLaunchKernelGGL((conv3_wgrad_kernel<T_, S_, true>), dim3(chunks), dim3(256), 0, st, (const T_*)D, Gt, partial, g); \
    else                                                                                                                  \
      hipLaunchKernelGGL((conv3_wgrad_kernel<T_, S_>), dim3(chunks), dim3(256), 0, st, (const T_*)D, Gt, partial, g);     \
  } while (0)
  if (d_b16 == 2) {
    if (segs == 1) CONV3_WG(half_t, 1);
    else if (segs == 2) CONV3_WG(half_t, 2);
    else CONV3_WG(half_t, 4);
  } else if (d_b16) {
    if (segs == 1) CONV3_WG(bf16_t, 1);
    else if (segs == 2) CONV3_WG(bf16_t, 2);
    else CONV3_WG(bf16_t, 4);
  } else {
    if (segs == 1) CONV3_WG(float, 1);
    else if (segs == 2) CONV3_WG(float, 2);
    else CONV3_WG(float, 4);
  }
#undef CONV3_WG
  MMDYN_LAUNCH_CHECK();
}
