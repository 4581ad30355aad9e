// Image decode in front of the hot path: uint8 HWC frames resident in HBM -> anti-aliased bilinear resize ->
// float32 CHW in [0,1].  Replaces transforms.Resize(input_size) + transforms.ToTensor() applied per frame on the
// host by the reference's dataset (/root/reference/mmdyn/pytorch/utils/datasets.py:23-31, 375-385), whose
// arithmetic is Pillow's 8-bit resampler (libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// horizontal then vertical pass, each rounded to uint8) followed by /255.  Integer work: bit-exact.
//
// HBM-bound: one block stages a band of input rows in LDS with 16-byte loads (each input byte is fetched once
// per band; bands overlap by the filter support), runs the horizontal pass LDS->LDS, then the vertical pass
// straight to coalesced float stores.  A batch gathers frames by index, so a shuffled mini-batch is one launch.
#include <math.h>
#include <vector>
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Resample.c
constexpr int LDS_BUDGET = 60 * 1024;

struct AxisPlan {
  int ksize = 0;
  std::vector<int> bounds, coeffs;
};

// Resample.c precompute_coeffs + normalize_coeffs_8bpc, bilinear filter (support 1), box = the whole axis
AxisPlan make_plan(int in_size, int out_size) {
  AxisPlan p;
  double scale = (double)in_size / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  p.ksize = (int)ceil(support) * 2 + 1;
  p.bounds.assign((size_t)out_size * 2, 0);
  p.coeffs.assign((size_t)out_size * p.ksize, 0);
  if (in_size == out_size) {   // Image.resize copies; as a resampling pass that is the identity tap
    for (int xx = 0; xx < out_size; ++xx) {
      p.bounds[xx * 2] = xx;
      p.bounds[xx * 2 + 1] = 1;
      p.coeffs[(size_t)xx * p.ksize] = 1 << PRECISION_BITS;
    }
    return p;
  }
  std::vector<double> k(p.ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double w = a < 1.0 ? 1.0 - a : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = xmax; x < p.ksize; ++x) k[x] = 0.0;
    for (int x = 0; x < p.ksize; ++x)
      p.coeffs[(size_t)xx * p.ksize + x] =
          k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
    p.bounds[xx * 2] = xmin;
    p.bounds[xx * 2 + 1] = xmax;
  }
  return p;
}

__device__ __forceinline__ unsigned clip8(int acc) {
  int v = acc >> PRECISION_BITS;
  return (unsigned)min(max(v, 0), 255);
}

__global__ __launch_bounds__(256) void resize_u8_chw_f32_kernel(
    const uint8_t* __restrict__ src, const int* __restrict__ index, float* __restrict__ dst, int Hin, int Win,
    int Hout, int Wout, const int* __restrict__ xb, const int* __restrict__ xk, int xks, const int* __restrict__ yb,
    const int* __restrict__ yk, int yks, int band, int in_pitch, int tmp_pitch, int max_in_rows) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* in_rows = lds;                                   // [max_in_rows][in_pitch]
  uint8_t* tmp = lds + (size_t)max_in_rows * in_pitch;      // [max_in_rows][tmp_pitch]  (after the horizontal pass)
  int* tab = reinterpret_cast<int*>(tmp + (size_t)max_in_rows * tmp_pitch);
  int* s_xb = tab;                                          // [Wout][2]
  int* s_xk = s_xb + 2 * Wout;                              // [Wout][xks]
  int* s_yb = s_xk + Wout * xks;                            // [band][2]
  int* s_yk = s_yb + 2 * band;                              // [band][yks]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int64_t img = index ? index[b] : b;
  const int oy0 = blockIdx.x * band, oy1 = min(Hout, oy0 + band);
  const int iy0 = yb[oy0 * 2], iy1 = yb[(oy1 - 1) * 2] + yb[(oy1 - 1) * 2 + 1];
  const int nrows = iy1 - iy0, row_bytes = Win * 3;

  // ---- tap tables of this band into LDS (every output sample re-reads them ksize times) ----
  for (int i = tid; i < 2 * Wout; i += 256) s_xb[i] = xb[i];
  for (int i = tid; i < Wout * xks; i += 256) s_xk[i] = xk[i];
  for (int i = tid; i < 2 * (oy1 - oy0); i += 256) s_yb[i] = yb[oy0 * 2 + i];
  for (int i = tid; i < (oy1 - oy0) * yks; i += 256) s_yk[i] = yk[oy0 * yks + i];

  // ---- stage the band's input rows (one contiguous byte range of the HWC image) ----
  const uint8_t* p = src + (img * Hin + iy0) * (int64_t)row_bytes;
  if ((((uintptr_t)p) & 15) == 0 && (row_bytes & 15) == 0) {
    const int vec_per_row = row_bytes >> 4;
    for (int i = tid; i < nrows * vec_per_row; i += 256) {
      const int r = i / vec_per_row, v = i - r * vec_per_row;
      *reinterpret_cast<uint4*>(in_rows + (size_t)r * in_pitch + v * 16) =
          *reinterpret_cast<const uint4*>(p + (size_t)r * row_bytes + v * 16);
    }
  } else {
    for (int i = tid; i < nrows * row_bytes; i += 256) {
      const int r = i / row_bytes, v = i - r * row_bytes;
      in_rows[(size_t)r * in_pitch + v] = p[i];
    }
  }
  __syncthreads();

  // ---- horizontal pass: [nrows][Win][3] -> [nrows][Wout][3], rounded to uint8 like Pillow's temp image.
  // One thread = one output pixel (3 channels share the tap weights); taps beyond the count have weight 0 in the
  // table (Resample.c zero-fills them), so the loop runs the full ksize with a clamped, never out-of-row read.
  if (xks == 9) {
    // the 4x down-scale (256 -> 64): the 27 tap bytes of a pixel are one unaligned 27-byte run; fetch it as 8
    // aligned dwords + v_alignbyte instead of 27 byte reads (the LDS issue rate is what bounds this pass)
    for (int i = tid; i < nrows * Wout; i += 256) {
      const int r = i / Wout, ox = i - r * Wout;
      const int base = s_xb[ox * 2] * 3;
      const uint32_t* wp = reinterpret_cast<const uint32_t*>(in_rows + (size_t)r * in_pitch + (base & ~3));
      const unsigned sh = base & 3;
      uint32_t w[8], q[7];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = wp[j];
#pragma unroll
      for (int j = 0; j < 7; ++j) q[j] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh);
      const int* k = s_xk + ox * 9;
      int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int wt = k[t];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int j = t * 3 + c;
          acc[c] += (int)((q[j >> 2] >> ((j & 3) * 8)) & 0xffu) * wt;
        }
      }
      uint8_t* o = tmp + (size_t)r * tmp_pitch + ox * 3;
      o[0] = (uint8_t)clip8(acc[0]);
      o[1] = (uint8_t)clip8(acc[1]);
      o[2] = (uint8_t)clip8(acc[2]);
    }
  } else
  for (int i = tid; i < nrows * Wout; i += 256) {
    const int r = i / Wout, ox = i - r * Wout;
    const int lo = s_xb[ox * 2];
    const uint8_t* row = in_rows + (size_t)r * in_pitch;
    const int* k = s_xk + ox * xks;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < xks; ++t) {
      const int xi = min(lo + t, Win - 1) * 3, w = k[t];
      a0 += (int)row[xi] * w;
      a1 += (int)row[xi + 1] * w;
      a2 += (int)row[xi + 2] * w;
    }
    uint8_t* o = tmp + (size_t)r * tmp_pitch + ox * 3;
    o[0] = (uint8_t)clip8(a0);
    o[1] = (uint8_t)clip8(a1);
    o[2] = (uint8_t)clip8(a2);
  }
  __syncthreads();

  // ---- vertical pass + ToTensor: uint8 -> float32 / 255, CHW (one thread = one output pixel, 3 planes) ----
  const int per_c = (oy1 - oy0) * Wout;
  for (int i = tid; i < per_c; i += 256) {
    const int oyl = i / Wout, ox = i - oyl * Wout, oy = oy0 + oyl;
    const int lo = s_yb[oyl * 2] - iy0;
    const int* k = s_yk + oyl * yks;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < yks; ++t) {
      const uint8_t* px = tmp + (size_t)min(lo + t, nrows - 1) * tmp_pitch + ox * 3;
      const int w = k[t];
      a0 += (int)px[0] * w;
      a1 += (int)px[1] * w;
      a2 += (int)px[2] * w;
    }
    float* o = dst + (((int64_t)b * 3) * Hout + oy) * Wout + ox;
    const int64_t plane = (int64_t)Hout * Wout;
    o[0] = (float)clip8(a0) / 255.0f;
    o[plane] = (float)clip8(a1) / 255.0f;
    o[2 * plane] = (float)clip8(a2) / 255.0f;
  }
}

}  // namespace

extern "C" int mmdyn_resize_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return MMDYN_ERR_SHAPE;
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(fs) * 2 + 1;
}

extern "C" int mmdyn_resize_plan(int in_size, int out_size, int* bounds, int* coeffs) {
  if (!bounds || !coeffs) return MMDYN_ERR_NULL;
  if (in_size <= 0 || out_size <= 0) return MMDYN_ERR_SHAPE;
  AxisPlan p = make_plan(in_size, out_size);
  for (size_t i = 0; i < p.bounds.size(); ++i) bounds[i] = p.bounds[i];
  for (size_t i = 0; i < p.coeffs.size(); ++i) coeffs[i] = p.coeffs[i];
  return p.ksize;
}

extern "C" int mmdyn_resize_u8_to_chw_f32(const uint8_t* src, const int* index, float* dst, int n_out, int Hin,
                                          int Win, int Hout, int Wout, const int* xb, const int* xk, const int* yb,
                                          const int* yk, void* stream) {
  if (!src || !dst || !xb || !xk || !yb || !yk) return MMDYN_ERR_NULL;
  if (n_out <= 0 || n_out > 65535 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return MMDYN_ERR_SHAPE;
  const AxisPlan py = make_plan(Hin, Hout);          // host copy of the row plan: sizes the bands
  const int in_pitch = (Win * 3 + 15) / 16 * 16, tmp_pitch = (Wout * 3 + 15) / 16 * 16;
  const int xks = mmdyn_resize_ksize(Win, Wout);
  auto tab_bytes = [&](int bnd) { return (size_t)4 * (2 * Wout + (size_t)Wout * xks + 2 * bnd + (size_t)bnd * py.ksize); };
  int band = 0, max_rows = 0;
  for (int cand = 32; cand >= 1; cand >>= 1) {
    int worst = 0;
    for (int oy0 = 0; oy0 < Hout; oy0 += cand) {
      const int oy1 = (oy0 + cand < Hout ? oy0 + cand : Hout) - 1;
      const int rows = py.bounds[oy1 * 2] + py.bounds[oy1 * 2 + 1] - py.bounds[oy0 * 2];
      if (rows > worst) worst = rows;
    }
    if ((size_t)worst * (in_pitch + tmp_pitch) + tab_bytes(cand) <= (size_t)LDS_BUDGET) {
      band = cand;
      max_rows = worst;
      break;
    }
  }
  if (!band) return MMDYN_ERR_RANGE;                 // a single output row's taps do not fit the LDS budget
  const size_t smem = (size_t)max_rows * (in_pitch + tmp_pitch) + tab_bytes(band);   // (the 32-byte tap window of
  // the last staged row may run past its row into tmp / the tables: read-only, weighted 0)
  dim3 grid(ceil_div(Hout, band), n_out);
  hipLaunchKernelGGL(resize_u8_chw_f32_kernel, grid, dim3(256), smem, (hipStream_t)stream, src, index, dst, Hin, Win,
                     Hout, Wout, xb, xk, xks, yb, yk, py.ksize, band, in_pitch, tmp_pitch,
                     max_rows);
  MMDYN_LAUNCH_CHECK();
}
