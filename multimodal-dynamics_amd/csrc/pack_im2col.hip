// Weight packing and the 3-channel (NCHW) side of the first / last layer.
//   - canonical reference weight layouts (vae.py:198-206, 268-277: Conv2d [Cout][Cin][4][4],
//     ConvTranspose2d [Cin][Cout][4][4], Linear [out][in]) -> [tap][n][k] GEMM operands;
//   - im2col of the 3-channel NCHW image (input of Conv2d(3,32,4,2,1), vae.py:198; logit gradient of
//     ConvTranspose2d(32,3,4,2,1), vae.py:277);
//   - col2im (transposed-conv scatter written as a gather, so it is deterministic and atomic-free).
// All HBM-bound, fully coalesced on the write side.
#include "common.h"

namespace {

// TO = float or bf16_t: in the bf16 precision modes the GEMM operands are packed straight to bf16 (RNE), which halves
// what the implicit-GEMM blocks pull through L2 for the weight tiles
template <typename TO>
__global__ void pack_conv_weight_kernel(const float* __restrict__ Wc, TO* __restrict__ P, int d0, int d1,
                                        int swap) {
  const int64_t total = (int64_t)16 * d0 * d1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    // i indexes P[tap][x][y]
    int nx = swap ? d1 : d0, ny = swap ? d0 : d1;
    int y = (int)(i % ny);
    int64_t t = i / ny;
    int x = (int)(t % nx);
    int tap = (int)(t / nx);
    int a = swap ? y : x, b = swap ? x : y;  // canonical indices (d0, d1)
    st1<TO>(P + i, Wc[((int64_t)a * d1 + b) * 16 + tap]);
  }
}

__global__ void repack2d_kernel(const float* __restrict__ in, float* __restrict__ out, int rows_in,
                                int cols_in, int rows_out, int cols_out, int mode) {
  const int64_t total = (int64_t)rows_out * cols_out;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cols_out), r = (int)(i / cols_out);
    int ri, ci;
    switch (mode) {
      case 0: ri = r; ci = c; break;
      case 1: ri = c; ci = r; break;
      case 2: { int hw = c / 256, ch = c - hw * 256; ri = r; ci = ch * 25 + hw; } break;
      case 3: { int hw = r / 256, ch = r - hw * 256; ri = ch * 25 + hw; ci = c; } break;
      case 4: { int hw = r / 256, ch = r - hw * 256; ri = c; ci = ch * 25 + hw; } break;
      default: { int hw = c / 256, ch = c - hw * 256; ri = ch * 25 + hw; ci = r; } break;
    }
    float v = 0.f;
    if (ri < rows_in && ci < cols_in) v = in[(int64_t)ri * cols_in + ci];
    out[i] = v;
  }
}

// All weight repacks of a step in ONE launch: blockIdx.y = plan entry, grid-stride over its output elements.
__device__ __forceinline__ float repack_fetch(const float* __restrict__ in, int rows_in, int cols_in, int r, int c,
                                              int mode) {
  int ri, ci;
  switch (mode) {
    case 0: ri = r; ci = c; break;
    case 1: ri = c; ci = r; break;
    case 2: { int hw = c / 256, ch = c - hw * 256; ri = r; ci = ch * 25 + hw; } break;
    case 3: { int hw = r / 256, ch = r - hw * 256; ri = ch * 25 + hw; ci = c; } break;
    case 4: { int hw = r / 256, ch = r - hw * 256; ri = c; ci = ch * 25 + hw; } break;
    default: { int hw = c / 256, ch = c - hw * 256; ri = ch * 25 + hw; ci = r; } break;
  }
  return (ri < rows_in && ci < cols_in) ? in[(int64_t)ri * cols_in + ci] : 0.f;
}

template <typename TO>
__global__ void repack2d_ld_kernel(const float* __restrict__ in, TO* __restrict__ out, int rows_in, int cols_in,
                                   int rows_out, int cols_out, int ld_out, int mode) {
  const int64_t total = (int64_t)rows_out * cols_out;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cols_out), r = (int)(i / cols_out);
    st1<TO>(out + (int64_t)r * ld_out + c, repack_fetch(in, rows_in, cols_in, r, c, mode));
  }
}

// ---- LDS-tiled forms of the transposing / permuting packs ---------------------------------------------------------
// Every pack is a permutation (plus zero padding).  Done element-wise with coalesced WRITES, the reads of the transposing
// kinds are strided: each 4-byte read touches its own 64-byte sector and the step's packs fetched 883 MB to write 107 MB
// (profiles/r1/hbm_traffic_by_kernel_v5.json).  Here a block moves a tile whose footprint is contiguous runs on BOTH
// sides: coalesced 16-byte / 4-byte loads of whole runs into LDS, then coalesced stores that gather from LDS (odd LDS row
// strides: conflict-free).  Tiles are walked grid-stride, one plan entry per blockIdx.y as before.
constexpr int PACK_LDS_FLOATS = 6912;           // 27 KB (5 blocks per CU): the largest tiles below are 32 x 201 and 400 x 17 floats

// conv weight: Wc[d0][d1][16] -> P[tap][x][y], (x, y) = (a, b) or swapped: tile = TA values of a x TB values of b x 16 taps
// PLANES (TO = bf16_t): the destination is a plane tensor -- rows (tap, x) of [plane][ny] bf16, the exact three-term split of the
// packed weight (the Bp operand of the fp32x3 launches that take their operands already split, igemm_wsp3_kernel)
__device__ __forceinline__ void st1_planes(bf16_t* row_y, int ny, float v) {
  uint32_t h, m, l;
  split3_bf16(v, 0.f, h, m, l);
  row_y[0] = (bf16_t)(h & 0xffffu);
  row_y[ny] = (bf16_t)(m & 0xffffu);
  row_y[2 * ny] = (bf16_t)(l & 0xffffu);
}
// element (r, c) of a packed 2-D operand with leading dimension ld: a plain tensor, or (PL) a plane tensor -- rows of [plane][ld]
// bf16, the exact three-term split (round 6: the FC-level weights whose launches take their operands already split)
template <typename TO, bool PL>
__device__ __forceinline__ void put_rc(TO* dst, int64_t r, int ld, int c, float v) {
  if constexpr (PL) st1_planes(reinterpret_cast<bf16_t*>(dst) + r * 3 * ld + c, ld, v);
  else st1<TO>(dst + r * ld + c, v);
}
template <typename TO, bool PLANES = false>
__device__ __forceinline__ void pack_conv_tiled(const float* __restrict__ src, TO* __restrict__ dst, int d0, int d1, int swap,
                                                float* lds) {
  // runs on the destination are along y: b for the keep form, a for the swapped one
  const int TB = swap ? 8 : (d1 < 64 ? d1 : 64), TA = 256 / TB > d0 ? d0 : 256 / TB;
  const int tiles_b = (d1 + TB - 1) / TB, tiles_a = (d0 + TA - 1) / TA;
  const int LDT = 17;                                                // [pair][tap], padded
  for (int t = blockIdx.x; t < tiles_a * tiles_b; t += gridDim.x) {
    const int a0 = (t / tiles_b) * TA, b0 = (t % tiles_b) * TB;
    __syncthreads();
    // source: for each a, the run [b0, b0+TB) x 16 taps is contiguous
#pragma unroll 4
    for (int i = threadIdx.x; i < TA * TB * 4; i += blockDim.x) {
      const int q = i & 3, pr = i >> 2, ia = pr / TB, ib = pr - ia * TB;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a0 + ia < d0 && b0 + ib < d1) v = *reinterpret_cast<const f32x4*>(src + ((size_t)(a0 + ia) * d1 + b0 + ib) * 16 + q * 4);
      float* l = lds + (ia * TB + ib) * LDT + q * 4;
      l[0] = v[0]; l[1] = v[1]; l[2] = v[2]; l[3] = v[3];
    }
    __syncthreads();
    const int nx = swap ? d1 : d0, ny = swap ? d0 : d1;
    const int TY = swap ? TA : TB, TX = swap ? TB : TA;              // tile extent along y (fast) and x
#pragma unroll 4
    for (int i = threadIdx.x; i < 16 * TX * TY; i += blockDim.x) {
      const int iy = i % TY, r = i / TY, ix = r % TX, tap = r / TX;
      const int ia = swap ? iy : ix, ib = swap ? ix : iy;
      const int x = (swap ? b0 : a0) + ix, y = (swap ? a0 : b0) + iy;
      if (x < nx && y < ny) {
        if constexpr (PLANES) st1_planes(reinterpret_cast<bf16_t*>(dst) + ((size_t)tap * nx + x) * 3 * ny + y, ny, lds[(ia * TB + ib) * LDT + tap]);
        else st1<TO>(dst + ((size_t)tap * nx + x) * ny + y, lds[(ia * TB + ib) * LDT + tap]);
      }
    }
  }
}

// kind 2: out[r][hw*256+ch] = in[r][ch*25+hw] (a permutation inside every 6400-float row): one row per tile
template <typename TO, bool PL = false>
__device__ __forceinline__ void pack_rowperm_tiled(const mmdyn_pack_entry& e, float* lds) {
  const float* __restrict__ src = e.src;
  TO* __restrict__ dst = reinterpret_cast<TO*>(e.dst);
  const int W = e.cols_in;                                            // 6400
  for (int r = blockIdx.x; r < e.rows_out; r += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < W / 4; i += blockDim.x)
      reinterpret_cast<f32x4*>(lds)[i] = r < e.rows_in ? reinterpret_cast<const f32x4*>(src + (size_t)r * W)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 5
    for (int c = threadIdx.x; c < e.cols_out; c += blockDim.x) {
      const int hw = c >> 8, ch = c & 255;
      const int ci = ch * 25 + hw;
      put_rc<TO, PL>(dst, r, e.ld_out, c, ci < W ? lds[ci] : 0.f);
    }
  }
}

// kinds 1, 4, 5: out[r][c] = in[ci][ri] with (ri, ci) a function of (r, c): transposes, with the 25 x 256 flatten
// permutation on the output rows (4) or columns (5).  Tile = RT input rows x CT contiguous input columns.
//   kind 1: out[r][c] = in[c][r]
//   kind 4: out[hw*256+ch][c] = in[c][ch*25+hw]        input rows c, input columns ch*25+hw
//   kind 5: out[r][hw*256+ch] = in[ch*25+hw][r]        input rows ch*25+hw, input columns r
template <typename TO, bool PL = false>
__device__ __forceinline__ void pack_transpose_tiled(const mmdyn_pack_entry& e, float* lds) {
  const float* __restrict__ src = e.src;
  TO* __restrict__ dst = reinterpret_cast<TO*>(e.dst);
  const int kind = e.kind, RI = e.rows_in, CI = e.cols_in;
  // input tile: RT rows x CT columns (CT contiguous floats per row on the source side)
  const int RT = kind == 5 ? 400 : (kind == 4 ? 32 : 64), CT = kind == 4 ? 200 : (kind == 5 ? 16 : 64);
  const int LDT = CT + 1;
  const int tiles_r = (RI + RT - 1) / RT, tiles_c = (CI + CT - 1) / CT;
  for (int t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x) {
    const int r0 = (t / tiles_c) * RT, c0 = (t % tiles_c) * CT;
    __syncthreads();
    if ((CI & 3) == 0) {                        // 16-byte loads (CT and c0 are multiples of 4): several in flight per thread
      const int CV = CT >> 2;
#pragma unroll 4
      for (int i = threadIdx.x; i < RT * CV; i += blockDim.x) {
        const int ir = i / CV, iv = i - ir * CV;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + ir < RI && c0 + iv * 4 < CI) v = *reinterpret_cast<const f32x4*>(src + (size_t)(r0 + ir) * CI + c0 + iv * 4);
        float* l = lds + ir * LDT + iv * 4;
        l[0] = v[0]; l[1] = v[1]; l[2] = v[2]; l[3] = v[3];
      }
    } else {
      for (int i = threadIdx.x; i < RT * CT; i += blockDim.x) {
        const int ir = i / CT, ic = i - ir * CT;
        lds[ir * LDT + ic] = (r0 + ir < RI && c0 + ic < CI) ? src[(size_t)(r0 + ir) * CI + c0 + ic] : 0.f;
      }
    }
    __syncthreads();
    // destination: runs along the input-ROW index (that is what a transpose makes contiguous)
#pragma unroll 4
    for (int i = threadIdx.x; i < RT * CT; i += blockDim.x) {
      int ir, ic;
      if (kind == 5) {                          // input row = ch*25+hw: make ch (stride 25 rows) the fast index
        const int ch = i % 16, rest = i / 16, hw = rest % 25;
        ic = rest / 25;
        ir = ch * 25 + hw;                      // 16 channels x 25 positions = the tile's 400 rows
      } else {
        ir = i % RT;
        ic = i / RT;
      }
      const int ri = r0 + ir, ci = c0 + ic;     // input coordinates
      if (ri >= RI || ci >= CI) continue;
      int ro, co;
      if (kind == 1) {
        ro = ci;
        co = ri;
      } else if (kind == 4) {
        const int ch = ci / 25, hw = ci - ch * 25;
        ro = hw * 256 + ch;
        co = ri;
      } else {
        const int ch = ri / 25, hw = ri - ch * 25;
        ro = ci;
        co = hw * 256 + ch;
      }
      if (ro < e.rows_out && co < e.cols_out) put_rc<TO, PL>(dst, ro, e.ld_out, co, lds[ir * LDT + ic]);
    }
  }
}

template <typename TO, bool PL = false>
__device__ __forceinline__ void pack_plan_entry(const mmdyn_pack_entry& e, float* lds) {
  const float* __restrict__ src = e.src;
  TO* __restrict__ dst = reinterpret_cast<TO*>(e.dst);
  if (e.kind >= 100) {                       // conv weight: Wc[d0][d1][16] -> P[tap][x][y]
    pack_conv_tiled<TO, PL>(src, dst, e.rows_in, e.cols_in, e.kind - 100, lds);
  } else if (e.kind == 2 && e.cols_in == 6400 && e.cols_out == 6400 && e.cols_in % 4 == 0) {
    pack_rowperm_tiled<TO, PL>(e, lds);
  } else if (e.kind == 1 || ((e.kind == 4 || e.kind == 5) && (e.kind == 4 ? e.cols_in : e.rows_in) == 6400)) {
    // (zero padding beyond the transposed source is written by the element-wise pass: only when the shapes differ)
    const bool padded = e.kind == 1 ? (e.rows_out != e.cols_in || e.cols_out != e.rows_in)
                                    : (e.kind == 4 ? (e.rows_out != 6400 || e.cols_out != e.rows_in)
                                                   : (e.cols_out != 6400 || e.rows_out != e.cols_in));
    if (!padded) {
      pack_transpose_tiled<TO, PL>(e, lds);
      return;
    }
    const int64_t total = (int64_t)e.rows_out * e.cols_out;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int c = (int)(i % e.cols_out), r = (int)(i / e.cols_out);
      put_rc<TO, PL>(dst, r, e.ld_out, c, repack_fetch(src, e.rows_in, e.cols_in, r, c, e.kind));
    }
  } else {                                   // 2-D repack with output leading dimension
    const int64_t total = (int64_t)e.rows_out * e.cols_out;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int c = (int)(i % e.cols_out), r = (int)(i / e.cols_out);
      put_rc<TO, PL>(dst, r, e.ld_out, c, repack_fetch(src, e.rows_in, e.cols_in, r, c, e.kind));
    }
  }
}

__global__ __launch_bounds__(256) void pack_plan_kernel(const mmdyn_pack_entry* __restrict__ plan) {
  __shared__ __attribute__((aligned(16))) float lds[PACK_LDS_FLOATS];
  const mmdyn_pack_entry e = plan[blockIdx.y];
  if (e.dst_bf16 == 3) {          // (3: plane tensor -- rows of [plane][ld_out] bf16; conv kinds: rows (tap, x) of [plane][y])
    pack_plan_entry<bf16_t, true>(e, lds);
  } else if (e.dst_bf16 == 2)     // (2: IEEE half, the fp16-storage mode)
    pack_plan_entry<half_t>(e, lds);
  else if (e.dst_bf16)
    pack_plan_entry<bf16_t>(e, lds);
  else
    pack_plan_entry<float>(e, lds);
}

// one thread per (output pixel, ci*4+kh): writes one float4 = the 4 kw taps; 16 threads cover a 64-float row
__global__ void im2col_nchw3_kernel(const float* __restrict__ x, float* __restrict__ col, int Bt, int H,
                                    int W) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)Bt * Ho * Wo * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int q = (int)(i & 15);
    int64_t pix = i >> 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < 12) {
      int ci = q >> 2, kh = q & 3;
      int wo = (int)(pix % Wo);
      int64_t t = pix / Wo;
      int ho = (int)(t % Ho);
      int b = (int)(t / Ho);
      int y = 2 * ho - 1 + kh;
      if ((unsigned)y < (unsigned)H) {
        const float* row = x + (((int64_t)b * 3 + ci) * H + y) * W;
        int x0 = 2 * wo - 1;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
          int xx = x0 + kw;
          if ((unsigned)xx < (unsigned)W) v[kw] = row[xx];
        }
      }
    }
    reinterpret_cast<f32x4*>(col)[i] = v;
  }
}

// out(b, ho, wo, c) = sum over taps with (ho + p - kh) % s == 0: col[(b, hi, wi)][tap, c]
template <bool TAP_MAJOR>
__global__ void col2im_k4_kernel(const float* __restrict__ col, float* __restrict__ out, int Bt, int Hi,
                                 int Wi, int Ho, int Wo, int C, int ldcol, int s, int p) {
  const int64_t total = (int64_t)Bt * Ho * Wo * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int b, c, ho, wo;
    if (TAP_MAJOR) {  // NHWC output
      c = (int)(i % C);
      int64_t t = i / C;
      wo = (int)(t % Wo);
      t /= Wo;
      ho = (int)(t % Ho);
      b = (int)(t / Ho);
    } else {          // NCHW output
      wo = (int)(i % Wo);
      int64_t t = i / Wo;
      ho = (int)(t % Ho);
      t /= Ho;
      c = (int)(t % C);
      b = (int)(t / C);
    }
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      int ty = ho + p - kh;
      if (ty < 0 || (ty % s) != 0) continue;
      int hi = ty / s;
      if (hi >= Hi) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        int tx = wo + p - kw;
        if (tx < 0 || (tx % s) != 0) continue;
        int wi = tx / s;
        if (wi >= Wi) continue;
        int tap = kh * 4 + kw;
        int64_t row = ((int64_t)b * Hi + hi) * Wi + wi;
        int cc = TAP_MAJOR ? (tap * C + c) : (c * 16 + tap);
        acc += col[row * ldcol + cc];
      }
    }
    out[i] = acc;
  }
}

// The same sum for the NHWC / tap-major form with C % 4 == 0 (the fused step's k4 s1 p0 decoder layer on small batches):
// four channels per thread as 16-byte accesses, 32-bit index arithmetic (the element-wise kernel above spends its time in
// 64-bit divisions and 4-byte loads: 52 us for 60 MB).
__global__ __launch_bounds__(256) void col2im_k4_nhwc_vec_kernel(const float* __restrict__ col, float* __restrict__ out,
                                                                 int total4, int Hi, int Wi, int Ho, int Wo, int C, int ldcol,
                                                                 int s, int p) {
  const int C4 = C >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    int t = i / C4;
    const int wo = t % Wo;
    t /= Wo;
    const int ho = t % Ho, b = t / Ho;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int ty = ho + p - kh;
      if (ty < 0 || (ty % s) != 0) continue;
      const int hi = ty / s;
      if (hi >= Hi) continue;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int tx = wo + p - kw;
        if (tx < 0 || (tx % s) != 0) continue;
        const int wi = tx / s;
        if (wi >= Wi) continue;
        const size_t row = (size_t)(b * Hi + hi) * Wi + wi;
        acc += *reinterpret_cast<const f32x4*>(col + row * ldcol + (kh * 4 + kw) * C + c4 * 4);
      }
    }
    reinterpret_cast<f32x4*>(out)[i] = acc;
  }
}

// [B][C][HW] <-> [B][HW][C]; small tensors only (3-channel images), simple gather
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                    int HW) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    int64_t t = i / C;
    int p = (int)(t % HW);
    int b = (int)(t / HW);
    out[i] = in[((int64_t)b * C + c) * HW + p];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C,
                                    int HW) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int p = (int)(i % HW);
    int64_t t = i / HW;
    int c = (int)(t % C);
    int b = (int)(t / C);
    out[i] = in[((int64_t)b * HW + p) * C + c];
  }
}

}  // namespace

extern "C" const char* mmdyn_version(void) { return "mmdyn_hip 0.5 (gfx950)"; }
extern "C" int mmdyn_abi_version(void) { return MMDYN_ABI_VERSION; }

extern "C" int mmdyn_pack_conv_weight(const float* Wc, float* P, int d0, int d1, int swap, void* stream) {
  if (!Wc || !P) return MMDYN_ERR_NULL;
  if (d0 <= 0 || d1 <= 0) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(ew_grid((int64_t)16 * d0 * d1)), dim3(256), 0,
                     (hipStream_t)stream, Wc, P, d0, d1, swap);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_pack_conv_weight_b16(const float* Wc, void* P, int d0, int d1, int swap, int half, void* stream) {
  if (!Wc || !P) return MMDYN_ERR_NULL;
  if (d0 <= 0 || d1 <= 0) return MMDYN_ERR_SHAPE;
  if (half)
    hipLaunchKernelGGL(pack_conv_weight_kernel<half_t>, dim3(ew_grid((int64_t)16 * d0 * d1)), dim3(256), 0,
                       (hipStream_t)stream, Wc, (half_t*)P, d0, d1, swap);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(ew_grid((int64_t)16 * d0 * d1)), dim3(256), 0,
                       (hipStream_t)stream, Wc, (bf16_t*)P, d0, d1, swap);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_repack2d(const float* in, float* out, int rows_in, int cols_in, int rows_out,
                              int cols_out, int mode, void* stream) {
  if (!in || !out) return MMDYN_ERR_NULL;
  if (mode < 0 || mode > 5 || rows_out <= 0 || cols_out <= 0) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(repack2d_kernel, dim3(ew_grid((int64_t)rows_out * cols_out)), dim3(256), 0,
                     (hipStream_t)stream, in, out, rows_in, cols_in, rows_out, cols_out, mode);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_repack2d_ld(const float* in, float* out, int rows_in, int cols_in, int rows_out, int cols_out,
                                 int ld_out, int mode, void* stream) {
  if (!in || !out) return MMDYN_ERR_NULL;
  if (mode < 0 || mode > 5 || rows_out <= 0 || cols_out <= 0 || ld_out < cols_out) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(repack2d_ld_kernel<float>, dim3(ew_grid((int64_t)rows_out * cols_out)), dim3(256), 0,
                     (hipStream_t)stream, in, out, rows_in, cols_in, rows_out, cols_out, ld_out, mode);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_repack2d_ld_b16(const float* in, void* out, int rows_in, int cols_in, int rows_out, int cols_out,
                                     int ld_out, int mode, int half, void* stream) {
  if (!in || !out) return MMDYN_ERR_NULL;
  if (mode < 0 || mode > 5 || rows_out <= 0 || cols_out <= 0 || ld_out < cols_out) return MMDYN_ERR_SHAPE;
  if (half)
    hipLaunchKernelGGL(repack2d_ld_kernel<half_t>, dim3(ew_grid((int64_t)rows_out * cols_out)), dim3(256), 0,
                       (hipStream_t)stream, in, (half_t*)out, rows_in, cols_in, rows_out, cols_out, ld_out, mode);
  else
    hipLaunchKernelGGL(repack2d_ld_kernel<bf16_t>, dim3(ew_grid((int64_t)rows_out * cols_out)), dim3(256), 0,
                       (hipStream_t)stream, in, (bf16_t*)out, rows_in, cols_in, rows_out, cols_out, ld_out, mode);
  MMDYN_LAUNCH_CHECK();
}

// fp32 [rows][C] -> the exact three-term bf16 split, rows of [plane][C]: hi | mid | lo (split3_bf16, common.h), 6 bytes per
// element.  One thread moves 8 consecutive channels of a row: two 16-byte loads, three 16-byte stores.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, bf16_t* __restrict__ planes, int64_t granules,
                                                          int C) {
  const int gpr = C >> 3;        // 8-channel granules per row
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < granules; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / gpr;
    const int c0 = (int)(i - row * gpr) * 8;
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * C + c0));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + row * C + c0 + 4));
    uint32_t h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split3_bf16(a[0], a[1], h0, m0, l0);
    split3_bf16(a[2], a[3], h1, m1, l1);
    split3_bf16(b[0], b[1], h2, m2, l2);
    split3_bf16(b[2], b[3], h3, m3, l3);
    bf16_t* o = planes + row * 3 * (int64_t)C + c0;
    *reinterpret_cast<u32x4*>(o) = u32x4{h0, h1, h2, h3};
    *reinterpret_cast<u32x4*>(o + C) = u32x4{m0, m1, m2, m3};
    *reinterpret_cast<u32x4*>(o + 2 * C) = u32x4{l0, l1, l2, l3};
  }
}

extern "C" int mmdyn_split_planes(const float* x, void* planes, int64_t rows, int C, void* stream) {
  if (!x || !planes) return MMDYN_ERR_NULL;
  if (rows <= 0 || C <= 0 || C % 8) return MMDYN_ERR_SHAPE;
  const int64_t granules = rows * (C / 8);
  hipLaunchKernelGGL(split_planes_kernel, dim3(ew_grid(granules)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<bf16_t*>(planes), granules, C);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_pack_plan(const mmdyn_pack_entry* plan_dev, int n, void* stream) {
  if (!plan_dev) return MMDYN_ERR_NULL;
  if (n <= 0 || n > 65535) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(pack_plan_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, plan_dev);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_im2col_nchw3(const float* x, float* col, int Bt, int H, int W, void* stream) {
  if (!x || !col) return MMDYN_ERR_NULL;
  if (H % 2 || W % 2 || Bt <= 0) return MMDYN_ERR_SHAPE;
  int64_t total = (int64_t)Bt * (H / 2) * (W / 2) * 16;
  hipLaunchKernelGGL(im2col_nchw3_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, col,
                     Bt, H, W);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_col2im_k4(const float* col, float* out, int Bt, int Hi, int Wi, int Ho, int Wo, int C,
                               int ldcol, int stride, int pad, int tap_major, void* stream) {
  if (!col || !out) return MMDYN_ERR_NULL;
  if (stride < 1 || Ho != (Hi - 1) * stride - 2 * pad + 4 || Wo != (Wi - 1) * stride - 2 * pad + 4 ||
      ldcol < 16 * C)
    return MMDYN_ERR_SHAPE;
  int64_t total = (int64_t)Bt * Ho * Wo * C;
  if (tap_major && C % 4 == 0 && ldcol % 4 == 0 && total < (1LL << 31) && (int64_t)Bt * Hi * Wi * ldcol < (1LL << 40)) {
    hipLaunchKernelGGL(col2im_k4_nhwc_vec_kernel, dim3(ew_grid(total / 4)), dim3(256), 0, (hipStream_t)stream, col, out,
                       (int)(total / 4), Hi, Wi, Ho, Wo, C, ldcol, stride, pad);
    MMDYN_LAUNCH_CHECK();
  }
  if (tap_major)
    hipLaunchKernelGGL(col2im_k4_kernel<true>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, col,
                       out, Bt, Hi, Wi, Ho, Wo, C, ldcol, stride, pad);
  else
    hipLaunchKernelGGL(col2im_k4_kernel<false>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       col, out, Bt, Hi, Wi, Ho, Wo, C, ldcol, stride, pad);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, void* stream) {
  if (!in || !out) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((int64_t)B * C * HW)), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, C, HW);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, void* stream) {
  if (!in || !out) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((int64_t)B * C * HW)), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, C, HW);
  MMDYN_LAUNCH_CHECK();
}
