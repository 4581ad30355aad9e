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
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int64_t img = index ? index[b] : b;
  const int oy0 = blockIdx.x * band, oy1 = min(Hout, oy0 + band);
  const int iy0 = yb[oy0 * 2], iy1 = yb[(oy1 - 1) * 2] + yb[(oy1 - 1) * 2 + 1];
  const int nrows = iy1 - iy0, row_bytes = Win * 3;

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

  // ---- horizontal pass: [nrows][Win][3] -> [nrows][Wout][3], rounded to uint8 like Pillow's temp image ----
  const int out_row = Wout * 3;
  for (int i = tid; i < nrows * out_row; i += 256) {
    const int r = i / out_row, rem = i - r * out_row;
    const int ox = rem / 3, c = rem - ox * 3;
    const int lo = xb[ox * 2], n = xb[ox * 2 + 1];
    const uint8_t* row = in_rows + (size_t)r * in_pitch + lo * 3 + c;
    const int* k = xk + (size_t)ox * xks;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < n; ++t) acc += (int)row[t * 3] * k[t];
    tmp[(size_t)r * tmp_pitch + rem] = (uint8_t)clip8(acc);
  }
  __syncthreads();

  // ---- vertical pass + ToTensor: uint8 -> float32 / 255, CHW ----
  const int per_c = (oy1 - oy0) * Wout;
  for (int i = tid; i < 3 * per_c; i += 256) {
    const int c = i / per_c, rem = i - c * per_c;
    const int oyl = rem / Wout, ox = rem - oyl * Wout, oy = oy0 + oyl;
    const int lo = yb[oy * 2] - iy0, n = yb[oy * 2 + 1];
    const int* k = yk + (size_t)oy * yks;
    const uint8_t* col = tmp + (size_t)lo * tmp_pitch + ox * 3 + c;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int t = 0; t < n; ++t) acc += (int)col[(size_t)t * tmp_pitch] * k[t];
    dst[(((int64_t)b * 3 + c) * Hout + oy) * Wout + ox] = (float)clip8(acc) / 255.0f;
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
  int band = 0, max_rows = 0;
  for (int cand = 32; cand >= 1; cand >>= 1) {
    int worst = 0;
    for (int oy0 = 0; oy0 < Hout; oy0 += cand) {
      const int oy1 = (oy0 + cand < Hout ? oy0 + cand : Hout) - 1;
      const int rows = py.bounds[oy1 * 2] + py.bounds[oy1 * 2 + 1] - py.bounds[oy0 * 2];
      if (rows > worst) worst = rows;
    }
    if ((size_t)worst * (in_pitch + tmp_pitch) <= (size_t)LDS_BUDGET) {
      band = cand;
      max_rows = worst;
      break;
    }
  }
  if (!band) return MMDYN_ERR_RANGE;                 // a single output row's taps do not fit the LDS budget
  const size_t smem = (size_t)max_rows * (in_pitch + tmp_pitch);
  dim3 grid(ceil_div(Hout, band), n_out);
  hipLaunchKernelGGL(resize_u8_chw_f32_kernel, grid, dim3(256), smem, (hipStream_t)stream, src, index, dst, Hin, Win,
                     Hout, Wout, xb, xk, mmdyn_resize_ksize(Win, Wout), yb, yk, py.ksize, band, in_pitch, tmp_pitch,
                     max_rows);
  MMDYN_LAUNCH_CHECK();
}
