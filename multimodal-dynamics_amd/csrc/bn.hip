// Train-mode BatchNorm2d (+ fused Swish) on channels-last [rows][C] activations, statistics per group.
// Replaces nn.BatchNorm2d + Swish forward/backward of the reference
// (/root/reference/mmdyn/pytorch/models/vae.py:201-208, 269-276, 331-334): batch mean / biased variance
// over (B, H, W), eps 1e-5, affine, running estimates with momentum 0.1 and the unbiased variance.
// All kernels are HBM-bound: 16-byte lane accesses along the channel axis, column sums by
// per-thread accumulation + one LDS pass, per-tile partials (no atomics, deterministic).
#include "common.h"

namespace {

// Rows per block of the column reductions: 512 for large tensors; small ones (the encoder's last two BatchNorm layers: 6400 and
// 16384 rows) get tiles that still give the launch ~200+ blocks -- at 512 rows they ran on 13 / 32 blocks (35 us for 13 MB).
// ONE function for the launchers and for mmdyn_colstats_tiles (the size of the partial-sum buffer).
static int tile_rows_for(int rows_per_group) {
  int t = rows_per_group / 256 / 32 * 32;
  return t < 32 ? 32 : (t > 512 ? 512 : t);
}

struct BnParams {
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
};

// column sums of two per-element quantities over a tile of rows -> partial[g][t][2][C]
// MODE 0: (y, y^2).  MODE 1: (du, du * xhat) with du = da * swish'(gamma*xhat+beta).
template <int MODE, typename TA>
__global__ __launch_bounds__(256) void colreduce_kernel(const TA* __restrict__ y,
                                                        const TA* __restrict__ da, BnParams bp,
                                                        float* __restrict__ partial, int rows_per_group,
                                                        int C, int T, int tile_rows) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int g = blockIdx.y, t = blockIdx.x;
  const int CV = C >> 2;        // float4 columns (8..64)
  const int RL = 256 / CV;      // row lanes
  const int rl = tid / CV, cv = tid - rl * CV;
  const int r_begin = t * tile_rows;
  const int r_end = min(rows_per_group, r_begin + tile_rows);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 mean4, rstd4, gam4, bet4;
  if (MODE == 1 && rl < RL) {
    mean4 = *reinterpret_cast<const f32x4*>(bp.mean + (size_t)g * C + cv * 4);
    rstd4 = *reinterpret_cast<const f32x4*>(bp.rstd + (size_t)g * C + cv * 4);
    gam4 = *reinterpret_cast<const f32x4*>(bp.gamma + cv * 4);
    bet4 = *reinterpret_cast<const f32x4*>(bp.beta + cv * 4);
  }
  if (rl < RL) {
    const size_t base = (size_t)g * rows_per_group;
    // 4 rows per iteration: 4 (MODE 0) or 8 (MODE 1) independent 16-byte loads in flight per thread
    for (int r = r_begin + rl; r < r_end; r += 4 * RL) {
      f32x4 v[4], d[4];
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const int rr = r + u4 * RL;
        const size_t off = (base + (rr < r_end ? rr : r)) * C + cv * 4;
        v[u4] = ld4<TA>(y + off);
        if (MODE == 1) d[u4] = ld4<TA>(da + off);
      }
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        if (r + u4 * RL >= r_end) continue;
        if (MODE == 0) {
          s0 += v[u4];
          s1 += v[u4] * v[u4];
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float xh = (v[u4][k] - mean4[k]) * rstd4[k];
            float u = gam4[k] * xh + bet4[k];
            float du = d[u4][k] * swish_gradf_(u);
            s0[k] += du;
            s1[k] += du * xh;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[tid * 8 + k] = s0[k];
    red[tid * 8 + 4 + k] = s1[k];
  }
  __syncthreads();
  if (tid < C) {
    const int cv2 = tid >> 2, k = tid & 3;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < RL; ++l) {
      a += red[(l * CV + cv2) * 8 + k];
      b += red[(l * CV + cv2) * 8 + 4 + k];
    }
    const size_t o = ((size_t)(g * T + t) * 2) * C + tid;
    partial[o] = a;
    partial[o + C] = b;
  }
}

// sums over the T tile partials of one group, 32 channels per block and one of S slices of the tiles per
// blockIdx.z (a 32-channel layer has only G x 1 channel blocks; the slices give the big layers 100+ blocks):
// out[s][g][2][C] (double accumulate); the finish kernels add the S slices in a fixed order.
constexpr int BN_MAX_SPLITS = 32;
// single-launch ("last block finishes") form: every block arrives on ONE counter (~12 ns per arrival), so it runs with few,
// fat slices -- at most 4 x groups x channel slabs <= 128 blocks -- instead of the 1024 the two-launch form spreads the sums over
constexpr int TICKET_MAX_SPLITS = 4;
static inline int bn_splits(int T) {
  int s = T / 128;
  return s < 1 ? 1 : (s > BN_MAX_SPLITS ? BN_MAX_SPLITS : s);
}
// (2 KB of LDS, not 4: the persistent plane-ring GEMM of the OTHER lane holds 156 of a CU's 160 KB, and a finalize launch that
//  does not fit beside it waits for that whole GEMM -- the two half-waves of a wave are combined by a shuffle first)
__global__ __launch_bounds__(256) void tile_sum_kernel(const float* __restrict__ partial,
                                                       double* __restrict__ out, int T, int C) {
  __shared__ double red[2][4][32];
  const int g = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), tl = threadIdx.x >> 5;
  const int per = (T + gridDim.z - 1) / gridDim.z;
  const int t_lo = blockIdx.z * per, t_hi = min(T, t_lo + per);
  out += (size_t)blockIdx.z * gridDim.y * 2 * C;
  double a = 0.0, b = 0.0;
  for (int t = t_lo + tl; t < t_hi; t += 8) {
    const size_t o = ((size_t)(g * T + t) * 2) * C + c;
    a += (double)partial[o];
    b += (double)partial[o + C];
  }
  a += __shfl_xor(a, 32, 64);
  b += __shfl_xor(b, 32, 64);
  if ((threadIdx.x & 32) == 0) {
    red[0][threadIdx.x >> 6][threadIdx.x & 31] = a;
    red[1][threadIdx.x >> 6][threadIdx.x & 31] = b;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, cc = threadIdx.x & 31;
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l) s += red[which][l][cc];
    out[((size_t)g * 2 + which) * C + blockIdx.x * 32 + cc] = s;
  }
}

// eval mode: the "statistics" are the running estimates
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ mean, float* __restrict__ rstd, int G, int C, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int c = i % C;
  mean[i] = running_mean[c];
  rstd[i] = 1.0f / sqrtf(running_var[c] + eps);
}

// adds the S slices of tile_sum_kernel: out[g][2][C] (the quantity a synchronised BatchNorm all-reduces)
__global__ void bn_collapse_kernel(const double* __restrict__ sliced, double* __restrict__ out, int n, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += sliced[(size_t)k * n + i];
  out[i] = s;
}

struct BnFinish {
  int kind;                 // 0: none (tile sums only), 1: forward statistics, 2: backward sums / parameter gradients
  float* mean;              // kind 1
  float* rstd;
  float* running_mean;
  float* running_var;
  int64_t* nbt;
  int n;
  float eps, momentum;
  int repeat;
  float* sums_f;            // kind 2
  float* dgamma;
  float* dbeta;
  float beta_acc, sums_scale;
};

__device__ __forceinline__ void bn_stats_finish_channel(const double* sums, float* mean, float* rstd, float* running_mean,
                                                        float* running_var, int64_t* nbt, int G, int C, int n, float eps,
                                                        float momentum, int repeat, int S, int c) {
  float rm = running_mean ? running_mean[c] : 0.f;
  float rv = running_var ? running_var[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
      s1 += sums[(((size_t)s * G + g) * 2 + 0) * C + c];
      s2 += sums[(((size_t)s * G + g) * 2 + 1) * C + c];
    }
    double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[(size_t)g * C + c] = (float)m;
    rstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    float unb = (float)(var * ((double)n / (double)(n > 1 ? n - 1 : 1)));
    for (int k = 0; k < repeat; ++k) {
      rm = (1.f - momentum) * rm + momentum * (float)m;
      rv = (1.f - momentum) * rv + momentum * unb;
    }
  }
  if (running_mean) running_mean[c] = rm;
  if (running_var) running_var[c] = rv;
  if (nbt && c == 0) *nbt += (int64_t)G * repeat;
}

__global__ void bn_stats_finish_kernel(const double* __restrict__ sums, float* __restrict__ mean,
                                       float* __restrict__ rstd, float* __restrict__ running_mean,
                                       float* __restrict__ running_var, int64_t* __restrict__ nbt, int G,
                                       int C, int n, float eps, float momentum, int repeat, int S) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_stats_finish_channel(sums, mean, rstd, running_mean, running_var, nbt, G, C, n, eps, momentum, repeat, S, c);
}

template <typename T>
__global__ void bn_swish_fwd_kernel(const T* __restrict__ y, BnParams bp, T* __restrict__ a,
                                    int64_t total4, int rows_per_group, int C) {
  constexpr int V = vecw<T>::n, NV = V / 4;          // one 16-byte access per thread and iteration
  const int CV = C / V;
  const int64_t totalv = total4 * 4 / V;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < totalv;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const int64_t row = i / CV;
    const int g = (int)(row / rows_per_group);
    f32x4 v[NV], o[NV];
    ldv_nt<T>(y + i * V, v);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c0 = cv * V + q * 4;
      f32x4 m = *reinterpret_cast<const f32x4*>(bp.mean + (size_t)g * C + c0);
      f32x4 r = *reinterpret_cast<const f32x4*>(bp.rstd + (size_t)g * C + c0);
      f32x4 ga = *reinterpret_cast<const f32x4*>(bp.gamma + c0);
      f32x4 be = *reinterpret_cast<const f32x4*>(bp.beta + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[q][k] = swishf_(ga[k] * ((v[q][k] - m[k]) * r[k]) + be[k]);
    }
    stv_nt<T>(a + i * V, o);
  }
}

__device__ __forceinline__ void bn_bwd_param_channel(const double* sums, float* sums_f, float* dgamma, float* dbeta, int G,
                                                     int C, float beta_acc, int S, float sums_scale, int c) {
  double sb = 0.0, sg = 0.0;
  for (int g = 0; g < G; ++g) {
    double a = 0.0, b = 0.0;
    for (int s = 0; s < S; ++s) {
      a += sums[(((size_t)s * G + g) * 2 + 0) * C + c];
      b += sums[(((size_t)s * G + g) * 2 + 1) * C + c];
    }
    if (sums_f) {
      sums_f[((size_t)g * 2 + 0) * C + c] = (float)(a * sums_scale);
      sums_f[((size_t)g * 2 + 1) * C + c] = (float)(b * sums_scale);
    }
    sb += a;
    sg += b;
  }
  if (dbeta) dbeta[c] = (beta_acc != 0.f ? beta_acc * dbeta[c] : 0.f) + (float)sb;
  if (dgamma) dgamma[c] = (beta_acc != 0.f ? beta_acc * dgamma[c] : 0.f) + (float)sg;
}

__global__ void bn_bwd_param_kernel(const double* __restrict__ sums, float* __restrict__ sums_f,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int G, int C,
                                    float beta_acc, int S, float sums_scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_bwd_param_channel(sums, sums_f, dgamma, dbeta, G, C, beta_acc, S, sums_scale, c);
}

// tile_sum_kernel + the finish of the layer in ONE launch: the block that arrives last (ticket counter) adds the slices
// in the same fixed order as the separate finish kernels -- bit-identical results, one dependent launch fewer per
// BatchNorm layer and direction (24 per train step of the cnn-mvae).
__global__ __launch_bounds__(256) void tile_sum_finish_kernel(const float* __restrict__ partial, double* out, int T, int C,
                                                              unsigned* ticket, BnFinish f) {
  __shared__ double red[2][4][32];
  __shared__ int last_flag;
  const int g = blockIdx.y, c = blockIdx.x * 32 + (threadIdx.x & 31), tl = threadIdx.x >> 5;
  const int per = (T + gridDim.z - 1) / gridDim.z;
  const int t_lo = blockIdx.z * per, t_hi = min(T, t_lo + per);
  double* myout = out + (size_t)blockIdx.z * gridDim.y * 2 * C;
  double a = 0.0, b = 0.0;
  for (int t = t_lo + tl; t < t_hi; t += 8) {
    const size_t o = ((size_t)(g * T + t) * 2) * C + c;
    a += (double)partial[o];
    b += (double)partial[o + C];
  }
  a += __shfl_xor(a, 32, 64);
  b += __shfl_xor(b, 32, 64);
  if ((threadIdx.x & 32) == 0) {
    red[0][threadIdx.x >> 6][threadIdx.x & 31] = a;
    red[1][threadIdx.x >> 6][threadIdx.x & 31] = b;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5, cc = threadIdx.x & 31;
    double s = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l) s += red[which][l][cc];
    st_wt(&myout[((size_t)g * 2 + which) * C + blockIdx.x * 32 + cc], s);
  }
  const unsigned nblocks = gridDim.x * gridDim.y * gridDim.z;
  if (!last_block_arrives(ticket, nblocks, &last_flag)) return;
  const int G = gridDim.y, S = gridDim.z;
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    if (f.kind == 1)
      bn_stats_finish_channel(out, f.mean, f.rstd, f.running_mean, f.running_var, f.nbt, G, C, f.n, f.eps, f.momentum,
                              f.repeat, S, ch);
    else
      bn_bwd_param_channel(out, f.sums_f, f.dgamma, f.dbeta, G, C, f.beta_acc, S, f.sums_scale, ch);
  }
}

template <typename T>
__global__ void bn_swish_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ y,
                                          BnParams bp, const float* __restrict__ sums,
                                          T* __restrict__ dy, int64_t total4, int rows_per_group, int C,
                                          int da_is_du) {
  constexpr int V = vecw<T>::n, NV = V / 4;
  const int CV = C / V;
  const int64_t totalv = total4 * 4 / V;
  const float inv_n = 1.f / (float)rows_per_group;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < totalv;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const int64_t row = i / CV;
    const int g = (int)(row / rows_per_group);
    f32x4 v[NV], d[NV], o[NV];
    ldv_nt<T>(y + i * V, v);
    ldv_nt<T>(da + i * V, d);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c0 = cv * V + q * 4;
      f32x4 m = *reinterpret_cast<const f32x4*>(bp.mean + (size_t)g * C + c0);
      f32x4 r = *reinterpret_cast<const f32x4*>(bp.rstd + (size_t)g * C + c0);
      f32x4 ga = *reinterpret_cast<const f32x4*>(bp.gamma + c0);
      f32x4 be = *reinterpret_cast<const f32x4*>(bp.beta + c0);
      f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + ((size_t)g * 2 + 0) * C + c0);
      f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + ((size_t)g * 2 + 1) * C + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xh = (v[q][k] - m[k]) * r[k];
        float u = ga[k] * xh + be[k];
        float du = da_is_du ? d[q][k] : d[q][k] * swish_gradf_(u);   // the dgrad epilogue may already have applied swish'
        o[q][k] = ga[k] * r[k] * (du - s0[k] * inv_n - xh * (s1[k] * inv_n));
      }
    }
    stv_nt<T>(dy + i * V, o);
  }
}

// The same two passes writing their result as a PLANE tensor (rows of [plane][C] bf16: the exact three-term split, common.h) for
// the fp32x3 GEMMs that take their operands already split -- the split rides on a pass that exists anyway (VERDICT r4 items 1 and
// 3a).  One thread = 8 channels of a row: two 16-byte loads per input, three 16-byte plane stores; the fp32 result is written as
// well when `a` / `dy` is not null (a consumer that still wants fp32).
__global__ void bn_swish_fwd_planes_kernel(const float* __restrict__ y, BnParams bp, float* __restrict__ a, bf16_t* __restrict__ ap,
                                           int64_t total8, int rows_per_group, int C) {
  const int CV = C / 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const int64_t row = i / CV;
    const int g = (int)(row / rows_per_group);
    f32x4 v[2], o[2];
    ldv_nt<float>(y + i * 8, v);
    ldv_nt<float>(y + i * 8 + 4, v + 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c0 = cv * 8 + q * 4;
      f32x4 m = *reinterpret_cast<const f32x4*>(bp.mean + (size_t)g * C + c0);
      f32x4 r = *reinterpret_cast<const f32x4*>(bp.rstd + (size_t)g * C + c0);
      f32x4 ga = *reinterpret_cast<const f32x4*>(bp.gamma + c0);
      f32x4 be = *reinterpret_cast<const f32x4*>(bp.beta + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[q][k] = swishf_(ga[k] * ((v[q][k] - m[k]) * r[k]) + be[k]);
    }
    if (a) {
      stv_nt<float>(a + i * 8, o);
      stv_nt<float>(a + i * 8 + 4, o + 1);
    }
    store_planes8(ap + row * 3 * (int64_t)C + cv * 8, C, o[0], o[1]);
  }
}

__global__ void bn_swish_bwd_apply_planes_kernel(const float* __restrict__ da, const float* __restrict__ y, BnParams bp,
                                                 const float* __restrict__ sums, float* __restrict__ dy, bf16_t* __restrict__ dyp,
                                                 int64_t total8, int rows_per_group, int C, int da_is_du) {
  const int CV = C / 8;
  const float inv_n = 1.f / (float)rows_per_group;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    const int64_t row = i / CV;
    const int g = (int)(row / rows_per_group);
    f32x4 v[2], d[2], o[2];
    ldv_nt<float>(y + i * 8, v);
    ldv_nt<float>(y + i * 8 + 4, v + 1);
    ldv_nt<float>(da + i * 8, d);
    ldv_nt<float>(da + i * 8 + 4, d + 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c0 = cv * 8 + q * 4;
      f32x4 m = *reinterpret_cast<const f32x4*>(bp.mean + (size_t)g * C + c0);
      f32x4 r = *reinterpret_cast<const f32x4*>(bp.rstd + (size_t)g * C + c0);
      f32x4 ga = *reinterpret_cast<const f32x4*>(bp.gamma + c0);
      f32x4 be = *reinterpret_cast<const f32x4*>(bp.beta + c0);
      f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + ((size_t)g * 2 + 0) * C + c0);
      f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + ((size_t)g * 2 + 1) * C + c0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xh = (v[q][k] - m[k]) * r[k];
        float u = ga[k] * xh + be[k];
        float du = da_is_du ? d[q][k] : d[q][k] * swish_gradf_(u);
        o[q][k] = ga[k] * r[k] * (du - s0[k] * inv_n - xh * (s1[k] * inv_n));
      }
    }
    if (dy) {
      stv_nt<float>(dy + i * 8, o);
      stv_nt<float>(dy + i * 8 + 4, o + 1);
    }
    store_planes8(dyp + row * 3 * (int64_t)C + cv * 8, C, o[0], o[1]);
  }
}

}  // namespace

extern "C" int mmdyn_colstats_tiles(int rows_per_group) { return ceil_div(rows_per_group, tile_rows_for(rows_per_group)); }

static bool bn_shape_ok(int G, int rows_per_group, int C) {
  return G > 0 && rows_per_group > 0 && C >= 32 && C <= 256 && (C % 32) == 0 && (256 % (C / 4)) == 0 &&
         true;
}

extern "C" int mmdyn_colstats(const float* y, float* partial, int G, int rows_per_group, int C, void* stream) {
  if (!y || !partial) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  const int TR = tile_rows_for(rows_per_group), T = ceil_div(rows_per_group, TR);
  BnParams bp{};
  hipLaunchKernelGGL((colreduce_kernel<0, float>), dim3(T, G), dim3(256), 0, (hipStream_t)stream, y,
                     (const float*)nullptr, bp, partial, rows_per_group, C, T, TR);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_finalize(const float* partial, float* mean, float* rstd, float* running_mean,
                                 float* running_var, int64_t* nbt, double* g_sums, int G, int T, int C,
                                 int rows_per_group, float eps, float momentum, int repeat, uint32_t* ticket, void* stream) {
  if (!partial || !mean || !rstd || !g_sums) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C) || T <= 0) return MMDYN_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int S = bn_splits(T);
  if (ticket) {           // one launch: the last-arriving block of the tile sums finishes the statistics
    if (S > TICKET_MAX_SPLITS) S = TICKET_MAX_SPLITS;
    BnFinish f{};
    f.kind = 1;
    f.mean = mean;
    f.rstd = rstd;
    f.running_mean = running_mean;
    f.running_var = running_var;
    f.nbt = nbt;
    f.n = rows_per_group;
    f.eps = eps;
    f.momentum = momentum;
    f.repeat = repeat;
    hipLaunchKernelGGL(tile_sum_finish_kernel, dim3(C / 32, G, S), dim3(256), 0, st, partial, g_sums, T, C, ticket, f);
    MMDYN_LAUNCH_CHECK();
  }
  // without a ticket: two launches (the tile sums want G x C/32 x S blocks, the finish one thread per channel)
  hipLaunchKernelGGL(tile_sum_kernel, dim3(C / 32, G, S), dim3(256), 0, st, partial, g_sums, T, C);
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, g_sums, mean, rstd,
                     running_mean, running_var, nbt, G, C, rows_per_group, eps, momentum, repeat, S);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* rstd,
                                   int G, int C, float eps, void* stream) {
  if (!running_mean || !running_var || !mean || !rstd) return MMDYN_ERR_NULL;
  if (G <= 0 || C <= 0) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(ceil_div(G * C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                     running_var, mean, rstd, G, C, eps);
  MMDYN_LAUNCH_CHECK();
}

/* ---- synchronised BatchNorm (multi-GPU option): the three pieces around the all-reduce of the per-channel sums -- */
extern "C" int mmdyn_bn_reduce_partials(const float* partial, double* sums, double* g_sums, int G, int T, int C,
                                        void* stream) {
  if (!partial || !sums || !g_sums) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, 1, C) || T <= 0) return MMDYN_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int S = bn_splits(T), n = G * 2 * C;
  hipLaunchKernelGGL(tile_sum_kernel, dim3(C / 32, G, S), dim3(256), 0, st, partial, g_sums, T, C);
  hipLaunchKernelGGL(bn_collapse_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, g_sums, sums, n, S);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_finalize_sums(const double* sums, float* mean, float* rstd, float* running_mean,
                                      float* running_var, int64_t* nbt, int G, int C, int n, float eps,
                                      float momentum, int repeat, void* stream) {
  if (!sums || !mean || !rstd) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, 1, C) || n <= 0 || repeat < 1) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, mean,
                     rstd, running_mean, running_var, nbt, G, C, n, eps, momentum, repeat, 1);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_bwd_finalize_sums(const double* sums, float* sums_f, float* dgamma, float* dbeta, int G, int C,
                                          float sums_scale, float beta_acc, void* stream) {
  if (!sums) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, 1, C)) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(bn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, sums_f,
                     dgamma, dbeta, G, C, beta_acc, 1, sums_scale);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_swish_fwd(const float* y, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, float* a, int G, int rows_per_group, int C,
                                  void* stream) {
  if (!y || !mean || !rstd || !gamma || !beta || !a) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  BnParams bp{mean, rstd, gamma, beta};
  int64_t total4 = (int64_t)G * rows_per_group * (C / 4);
  hipLaunchKernelGGL(bn_swish_fwd_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, y, bp, a,
                     total4, rows_per_group, C);
  MMDYN_LAUNCH_CHECK();
}

/* bf16 activation storage (BASELINE configs[2]): same kernels, y / a / da / dy are bf16 in HBM (fp32 in registers);
 * statistics, affine parameters and sums stay fp32 */
extern "C" int mmdyn_bn_swish_fwd_b16(const uint16_t* y, const float* mean, const float* rstd, const float* gamma,
                                      const float* beta, uint16_t* a, int G, int rows_per_group, int C, int half,
                                      void* stream) {
  if (!y || !mean || !rstd || !gamma || !beta || !a) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  BnParams bp{mean, rstd, gamma, beta};
  int64_t total4 = (int64_t)G * rows_per_group * (C / 4);
  if (half)
    hipLaunchKernelGGL(bn_swish_fwd_kernel<half_t>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)y, bp, (half_t*)a, total4, rows_per_group, C);
  else
    hipLaunchKernelGGL(bn_swish_fwd_kernel<bf16_t>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, y, bp, a,
                       total4, rows_per_group, C);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_swish_bwd_reduce_b16(const uint16_t* da, const uint16_t* y, const float* mean,
                                             const float* rstd, const float* gamma, const float* beta, float* partial,
                                             int G, int rows_per_group, int C, int half, void* stream) {
  if (!da || !y || !mean || !rstd || !gamma || !beta || !partial) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  const int TR = tile_rows_for(rows_per_group), T = ceil_div(rows_per_group, TR);
  BnParams bp{mean, rstd, gamma, beta};
  if (half)
    hipLaunchKernelGGL((colreduce_kernel<1, half_t>), dim3(T, G), dim3(256), 0, (hipStream_t)stream, (const half_t*)y,
                       (const half_t*)da, bp, partial, rows_per_group, C, T, TR);
  else
    hipLaunchKernelGGL((colreduce_kernel<1, bf16_t>), dim3(T, G), dim3(256), 0, (hipStream_t)stream, y, da, bp, partial,
                       rows_per_group, C, T, TR);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_swish_bwd_apply_b16(const uint16_t* da, const uint16_t* y, const float* mean, const float* rstd,
                                            const float* gamma, const float* beta, const float* sums, uint16_t* dy,
                                            int G, int rows_per_group, int C, int da_is_du, int half, void* stream) {
  if (!da || !y || !mean || !rstd || !gamma || !beta || !sums || !dy) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  BnParams bp{mean, rstd, gamma, beta};
  int64_t total4 = (int64_t)G * rows_per_group * (C / 4);
  if (half)
    hipLaunchKernelGGL(bn_swish_bwd_apply_kernel<half_t>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)da, (const half_t*)y, bp, sums, (half_t*)dy, total4, rows_per_group, C, da_is_du);
  else
    hipLaunchKernelGGL(bn_swish_bwd_apply_kernel<bf16_t>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, da,
                       y, bp, sums, dy, total4, rows_per_group, C, da_is_du);
  MMDYN_LAUNCH_CHECK();
}


extern "C" int mmdyn_bn_swish_bwd_reduce(const float* da, const float* y, const float* mean,
                                         const float* rstd, const float* gamma, const float* beta,
                                         float* partial, int G, int rows_per_group, int C, void* stream) {
  if (!da || !y || !mean || !rstd || !gamma || !beta || !partial) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  const int TR = tile_rows_for(rows_per_group), T = ceil_div(rows_per_group, TR);
  BnParams bp{mean, rstd, gamma, beta};
  hipLaunchKernelGGL((colreduce_kernel<1, float>), dim3(T, G), dim3(256), 0, (hipStream_t)stream, y, da, bp, partial,
                     rows_per_group, C, T, TR);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_bwd_finalize(const float* partial, float* sums, float* dgamma, float* dbeta,
                                     double* g_sums, int G, int T, int C, float beta_acc, uint32_t* ticket, void* stream) {
  if (!partial || !sums || !g_sums) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, 1, C) || T <= 0) return MMDYN_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int S = bn_splits(T);
  if (ticket) {
    if (S > TICKET_MAX_SPLITS) S = TICKET_MAX_SPLITS;
    BnFinish f{};
    f.kind = 2;
    f.sums_f = sums;
    f.dgamma = dgamma;
    f.dbeta = dbeta;
    f.beta_acc = beta_acc;
    f.sums_scale = 1.0f;
    hipLaunchKernelGGL(tile_sum_finish_kernel, dim3(C / 32, G, S), dim3(256), 0, st, partial, g_sums, T, C, ticket, f);
    MMDYN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(tile_sum_kernel, dim3(C / 32, G, S), dim3(256), 0, st, partial, g_sums, T, C);
  hipLaunchKernelGGL(bn_bwd_param_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, st, g_sums, sums, dgamma,
                     dbeta, G, C, beta_acc, S, 1.0f);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_swish_bwd_apply(const float* da, const float* y, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta,
                                        const float* sums, float* dy, int G, int rows_per_group, int C,
                                        int da_is_du, void* stream) {
  if (!da || !y || !mean || !rstd || !gamma || !beta || !sums || !dy) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  BnParams bp{mean, rstd, gamma, beta};
  int64_t total4 = (int64_t)G * rows_per_group * (C / 4);
  hipLaunchKernelGGL(bn_swish_bwd_apply_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, da,
                     y, bp, sums, dy, total4, rows_per_group, C, da_is_du);
  MMDYN_LAUNCH_CHECK();
}

/* mmdyn_bn_swish_fwd / mmdyn_bn_swish_bwd_apply with the result written as a PLANE tensor (rows of [plane][C] bf16, the exact
 * three-term split: mmdyn_split_planes) for the fp32x3 GEMMs that take their operands already split; `a` / `dy` may be NULL when no
 * consumer wants the fp32 tensor.  The fp32 values are bit-identical to the plain entry points'. */
extern "C" int mmdyn_bn_swish_fwd_planes(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                         float* a, void* a_planes, int G, int rows_per_group, int C, void* stream) {
  if (!y || !mean || !rstd || !gamma || !beta || !a_planes) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  BnParams bp{mean, rstd, gamma, beta};
  const int64_t total8 = (int64_t)G * rows_per_group * (C / 8);
  hipLaunchKernelGGL(bn_swish_fwd_planes_kernel, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, y, bp, a,
                     reinterpret_cast<bf16_t*>(a_planes), total8, rows_per_group, C);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bn_swish_bwd_apply_planes(const float* da, const float* y, const float* mean, const float* rstd,
                                               const float* gamma, const float* beta, const float* sums, float* dy, void* dy_planes,
                                               int G, int rows_per_group, int C, int da_is_du, void* stream) {
  if (!da || !y || !mean || !rstd || !gamma || !beta || !sums || !dy_planes) return MMDYN_ERR_NULL;
  if (!bn_shape_ok(G, rows_per_group, C)) return MMDYN_ERR_SHAPE;
  BnParams bp{mean, rstd, gamma, beta};
  const int64_t total8 = (int64_t)G * rows_per_group * (C / 8);
  hipLaunchKernelGGL(bn_swish_bwd_apply_planes_kernel, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, da, y, bp, sums, dy,
                     reinterpret_cast<bf16_t*>(dy_planes), total8, rows_per_group, C, da_is_du);
  MMDYN_LAUNCH_CHECK();
}
