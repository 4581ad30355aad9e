// Element-wise / small-reduction kernels of the cnn-mvae step: activations of the FC layers, dropout with
// injected keep-masks (vae.py:213), bias gradients, the tiny 7-DoF pose Linear layers (vae.py:117-123),
// counter-based random draws for throughput runs, and Adam (problems.py:137-138).
// HBM-bound: 16-byte lane accesses, grid-stride loops capped at 2048 blocks.
#include "common.h"

namespace {

__global__ void act_fwd_kernel(const float* __restrict__ u, float* __restrict__ h, int64_t n, int act) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(u)[i], o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = apply_act(v[k], act);
    reinterpret_cast<f32x4*>(h)[i] = o;
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    h[i] = apply_act(u[i], act);
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ u, T* __restrict__ du, int64_t n,
                               int act) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = ld4<T>(u + i * 4), d = ld4<T>(dh + i * 4), o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = d[k] * act_grad(v[k], act);
    st4<T>(du + i * 4, o);
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    st1<T>(du + i, ld1<T>(dh + i) * act_grad(ld1<T>(u + i), act));
}

__global__ void dropout_expand_kernel(const float* __restrict__ h, const uint8_t* __restrict__ masks,
                                      float* __restrict__ out, int P, int64_t bh, float scale) {
  const int64_t total = (int64_t)P * bh;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = i % bh;
    // same association as the reference: x * (mask / (1-p))
    out[i] = h[j] * ((float)masks[i] * scale);
  }
}

// u != nullptr: the backward of the activation in front of the dropout rides along: dh = (sum ...) * act'(u)
// dh_planes != nullptr (round 6): dh also as a PLANE tensor (rows of [plane][H] bf16, the exact three-term split) -- the operand of
// the FC layer's input-gradient GEMM where that launch takes its operands already split
__global__ void dropout_reduce_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ masks,
                                      float* __restrict__ dh, int P, int64_t bh, float scale,
                                      const float* __restrict__ u, int act, bf16_t* __restrict__ dh_planes, int H) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < bh;
       j += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += dout[(int64_t)p * bh + j] * ((float)masks[(int64_t)p * bh + j] * scale);
    const float v = u ? s * act_grad(u[j], act) : s;
    dh[j] = v;
    if (dh_planes) {
      const int64_t row = j / H;
      const int col = (int)(j - row * H);
      uint32_t h, m, lo;
      split3_bf16(v, 0.f, h, m, lo);
      bf16_t* pr = dh_planes + row * 3 * H + col;
      pr[0] = (bf16_t)(h & 0xffffu);
      pr[H] = (bf16_t)(m & 0xffffu);
      pr[2 * H] = (bf16_t)(lo & 0xffffu);
    }
  }
}

// Philox-4x32-10 counter-based generator (Salmon et al., SC'11): stateless, so a draw is a pure function
// of (seed, offset, element index) and a captured graph replays with a device-side offset bump.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0;
  c[1] = n1;
  c[2] = n2;
  c[3] = n3;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0];
  out[1] = c[1];
  out[2] = c[2];
  out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void random_masks_kernel(uint8_t* __restrict__ masks, int64_t n, float p_drop, uint64_t seed,
                                    uint64_t offset, const uint64_t* __restrict__ offset_dev) {
  if (offset_dev) offset += offset_dev[0];
  const int64_t n4 = (n + 3) >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(seed, offset + (uint64_t)i, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t j = i * 4 + k;
      if (j < n) masks[j] = (u01(r[k]) >= p_drop) ? 1 : 0;
    }
  }
}

__global__ void random_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset,
                                     const uint64_t* __restrict__ offset_dev) {
  if (offset_dev) offset += offset_dev[0];
  const int64_t n4 = (n + 3) >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(seed, offset + (uint64_t)i, r);
    float z[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // Box-Muller on two pairs
      float rad = sqrtf(-2.0f * __logf(u01(r[2 * k])));
      float ang = 6.28318530717958647692f * u01(r[2 * k + 1]);
      z[2 * k] = rad * __cosf(ang);
      z[2 * k + 1] = rad * __sinf(ang);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int64_t j = i * 4 + k;
      if (j < n) out[j] = z[k];
    }
  }
}

__global__ void counter_add_kernel(uint64_t* __restrict__ c, uint64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += inc;
}

// out[c] (+)= sum_r x[r][c], two deterministic stages.  Stage 1: grid (C/128, chunks); a block covers 128 channels
// (32 float4 columns) x one row chunk with 8 row lanes, 4 rows in flight per thread, LDS finish -> partial[chunk][C].
// Stage 2 adds the chunks (and applies the upsample-bias row permutation).  C % 4 == 0.
__device__ __forceinline__ void colsum_final_channel(const float* partial, float* out, int chunks, int C, int perm,
                                                     float beta, int c) {
  float t = 0.f;
  for (int k = 0; k < chunks; ++k) t += partial[(size_t)k * C + c];
  int o = c;
  if (perm == 2) {
    int hw = c / 256, ch = c - hw * 256;
    o = ch * 25 + hw;
  }
  out[o] = (beta != 0.f ? beta * out[o] : 0.f) + t;
}

// ticket != nullptr: the block that arrives last adds the chunks (same order as colsum_final_kernel): one launch
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* partial,
                                                             int rows, int C, int rows_per_chunk, unsigned* ticket,
                                                             float* out, int perm, float beta) {
  __shared__ f32x4 red[4][32];          // (2 KB: fits beside a persistent GEMM block of the other lane, see tile_sum_kernel)
  __shared__ int last_flag;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c4 = blockIdx.x * 32 + cl;                     // float4 column
  const int CV = C >> 2;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < CV) {
    for (int r = r0 + rl; r < r1; r += 32) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 8 * u;
        v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(rr < r1 ? rr : r) * C + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + 8 * u < r1) s += v[u];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], 32, 64);          // the two half-waves of a wave
  if ((rl & 1) == 0) red[rl >> 1][cl] = s;
  __syncthreads();
  if (rl == 0 && c4 < CV) {
    f32x4 t = red[0][cl];
#pragma unroll
    for (int l = 1; l < 4; ++l) t += red[l][cl];
    float* dst = partial + (size_t)blockIdx.y * C + c4 * 4;
    if (ticket) {
#pragma unroll
      for (int k = 0; k < 4; ++k) st_wt(dst + k, t[k]);
    } else {
      *reinterpret_cast<f32x4*>(dst) = t;
    }
  }
  if (!ticket) return;
  if (!last_block_arrives(ticket, gridDim.x * gridDim.y, &last_flag)) return;
  for (int c = threadIdx.x; c < C; c += 256) colsum_final_channel(partial, out, gridDim.y, C, perm, beta, c);
}

__global__ void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int chunks, int C,
                                    int perm, float beta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  colsum_final_channel(partial, out, chunks, C, perm, beta, c);
}

// Single-launch form for the row counts of the FC level (rows <= COLSUM_DIRECT_ROWS: bias gradients of the Linear layers,
// vae.py:211-222, 264-265): a block owns 32 channels and ALL rows -- 32 row lanes x 8 float4 columns, 4 rows in flight per thread,
// shuffle + 512 bytes of LDS -- so there is no partial table and no second launch (round 6: 20 -> 10 launches per step).
constexpr int COLSUM_DIRECT_ROWS = 4096;
__global__ __launch_bounds__(256) void colsum_direct_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int C,
                                                            int perm, float beta) {
  __shared__ f32x4 red[4][8];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c4 = blockIdx.x * 8 + cl;                      // float4 column
  const int CV = C >> 2;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < CV) {
    for (int r = rl; r < rows; r += 128) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 32 * u;
        v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(rr < rows ? rr : r) * C + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + 32 * u < rows) s += v[u];
    }
  }
#pragma unroll
  for (int m = 8; m < 64; m <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += __shfl_xor(s[k], m, 64);         // the 8 row lanes of a wave
  if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][cl] = s;
  __syncthreads();
  if (threadIdx.x < 8 && c4 < CV) {
    f32x4 t = red[0][cl];
#pragma unroll
    for (int l = 1; l < 4; ++l) t += red[l][cl];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c4 * 4 + k;
      int o = c;
      if (perm == 2) {
        const int hw = c / 256, ch = c - hw * 256;
        o = ch * 25 + hw;
      }
      out[o] = (beta != 0.f ? beta * out[o] : 0.f) + t[k];
    }
  }
}

// Up to MMDYN_COPY_MANY_MAX device-to-device copies as ONE launch: the batch of a replayed step (visual / tactile / pose inputs and
// their targets, problems.py:148-156) moves into the captured step's static buffers with one kernel instead of one runtime copy
// per tensor.  blockIdx.y = segment; 16-byte lanes where both ends are 16-byte aligned, bytes otherwise.
struct CopyMany {
  const char* src[MMDYN_COPY_MANY_MAX];
  char* dst[MMDYN_COPY_MANY_MAX];
  int64_t bytes[MMDYN_COPY_MANY_MAX];
};
__global__ __launch_bounds__(256) void copy_many_kernel(const CopyMany cm) {
  const int sgm = blockIdx.y;
  const char* __restrict__ src = cm.src[sgm];
  char* __restrict__ dst = cm.dst[sgm];
  const int64_t n = cm.bytes[sgm];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int64_t n16 = n >> 4;
    for (int64_t i = i0; i < n16; i += stride) reinterpret_cast<u32x4_t*>(dst)[i] = reinterpret_cast<const u32x4_t*>(src)[i];
    for (int64_t i = (n16 << 4) + i0; i < n; i += stride) dst[i] = src[i];
  } else {
    for (int64_t i = i0; i < n; i += stride) dst[i] = src[i];
  }
}

__global__ void sum_blocks_kernel(const float* __restrict__ x, float* __restrict__ out, int P, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += x[(int64_t)p * n + i];
    out[i] = s;
  }
}

__global__ void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                        const float* __restrict__ b, float* __restrict__ y, int rows, int K,
                                        int N, int act) {
  const int64_t total = (int64_t)rows * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % N), r = (int)(i / N);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(x[(size_t)r * K + k], W[(size_t)n * K + k], s);
    if (b) s += b[n];
    y[i] = apply_act(s, act);
  }
}
// narrow outputs over a long reduction (the 512 -> 7 pose head): one wavefront per row, lanes stride over K with
// coalesced reads of x and of the N weight rows, N accumulators per lane, then a 64-lane shuffle reduction
template <int NMAX>
__global__ __launch_bounds__(256) void linear_small_fwd_rowwave_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ W,
                                                                       const float* __restrict__ b,
                                                                       float* __restrict__ y, int rows, int K, int N,
                                                                       int act) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int r = wave; r < rows; r += nwaves) {
    float s[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) s[n] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float xv = x[(size_t)r * K + k];
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n < N) s[n] = fmaf(xv, W[(size_t)n * K + k], s[n]);
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n) s[n] = wave_sum(s[n]);
    if (lane < N) {
      float v = 0.f;
#pragma unroll
      for (int n = 0; n < NMAX; ++n)
        if (n == lane) v = s[n];
      if (b) v += b[lane];
      y[(size_t)r * N + lane] = apply_act(v, act);
    }
  }
}

__global__ void linear_small_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                       float* __restrict__ dx, int rows, int K, int N) {
  const int64_t total = (int64_t)rows * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % K), r = (int)(i / K);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dy[(size_t)r * N + n], W[(size_t)n * K + k], s);
    dx[i] = s;
  }
}
// one wavefront per weight element: lanes stride over rows, then a 64-lane shuffle reduction
__global__ __launch_bounds__(256) void linear_small_dw_kernel(const float* __restrict__ dy,
                                                              const float* __restrict__ x,
                                                              float* __restrict__ dW, float* __restrict__ db,
                                                              int rows, int K, int N, float beta) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t total = (int64_t)N * (K + 1);  // column K = bias
  for (int64_t i = wave; i < total; i += nwaves) {
    const int k = (int)(i % (K + 1)), n = (int)(i / (K + 1));
    float s = 0.f;
    for (int r = lane; r < rows; r += 64) {
      float g = dy[(size_t)r * N + n];
      s += (k < K) ? g * x[(size_t)r * K + k] : g;
    }
    s = wave_sum(s);
    if (lane == 0) {
      if (k < K) {
        float* o = dW + (size_t)n * K + k;
        *o = (beta != 0.f ? beta * *o : 0.f) + s;
      } else if (db) {
        db[n] = (beta != 0.f ? beta * db[n] : 0.f) + s;
      }
    }
  }
}

// out = x * s[0] with the scale read from device memory (autograd's upstream scalar gradient: no host sync)
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                 float* __restrict__ out, int64_t n) {
  const float k = s[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[i] * k;
}

// Gradient buckets on the wire in bf16 (16-bit storage modes, SURVEY section 8e: 28.1 instead of 56.2 MB per step): the
// flat fp32 gradient bucket is rounded (RNE, v_cvt_pk_bf16_f32) into a bf16 staging buffer in front of the all-reduce and
// widened back (exact) behind it; Adam keeps reading fp32.  16 bytes per lane on the fp32 side.
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    st4<bf16_t>(dst + 4 * i, ld4<float>(src + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) st1<bf16_t>(dst + 4 * n4 + threadIdx.x, src[4 * n4 + threadIdx.x]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    st4<float>(dst + 4 * i, ld4<bf16_t>(src + 4 * i));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[4 * n4 + threadIdx.x] = ld1<bf16_t>(src + 4 * n4 + threadIdx.x);
}

__global__ void adam_tick_kernel(double* __restrict__ state, float lr, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = state[0] + 1.0;
    state[0] = t;
    state[1] = (double)lr / (1.0 - pow((double)beta1, t));  // step size
    state[2] = sqrt(1.0 - pow((double)beta2, t));           // sqrt(bias_correction2)
  }
}

// Guarded form (the fp16 modes, whose scaled gradients can overflow on their way through the matrix cores): state has six
// doubles -- {step count, step size, sqrt(bias_correction2), "gradient holds inf / NaN" flag, skipped steps, skip this step}.
// grad_nonfinite_kernel raises the flag, the tick turns it into "skip" (no step count, no moment or parameter update: what
// torch.cuda.amp.GradScaler does with such a step) and counts it.
__global__ void grad_nonfinite_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ state) {
  const int64_t n4 = n >> 2;
  float s = 0.f;                        // x - x is 0 for finite x and NaN otherwise
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s += g[i] - g[i];
  if (!(s == 0.f)) state[3] = 1.0;      // (every writer stores the same value)
}

__global__ void adam_tick_guarded_kernel(double* __restrict__ state, float lr, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (state[3] != 0.0) {
      state[3] = 0.0;
      state[4] += 1.0;
      state[5] = 1.0;
    } else {
      const double t = state[0] + 1.0;
      state[0] = t;
      state[1] = (double)lr / (1.0 - pow((double)beta1, t));
      state[2] = sqrt(1.0 - pow((double)beta2, t));
      state[5] = 0.0;
    }
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, const double* __restrict__ state, int64_t n, float beta1,
                            float beta2, float eps, float grad_scale, int guarded) {
  if (guarded && state[5] != 0.0) return;
  const float step_size = (float)state[1], bc2s = (float)state[2];
  const int64_t n4 = n >> 2;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= grad_scale;
    mm = mm + (gg - mm) * (1.f - beta1);
    vv = vv * beta2 + (1.f - beta2) * gg * gg;
    float denom = sqrtf(vv) / bc2s + eps;
    pp = pp - step_size * (mm / denom);
  };
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = pp[k], b = mm[k], c = vv[k];
      upd(a, gg[k], b, c);
      pp[k] = a;
      mm[k] = b;
      vv[k] = c;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    upd(p[i], g[i], m[i], v[i]);
}

// torch.optim.SGD(lr, momentum, weight_decay) as the reference configures it (problems.py:132-136: momentum 0.9,
// weight decay 5e-4, no dampening, no Nesterov).  first != 0: the momentum buffer starts as the gradient.
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n,
                           float lr, float momentum, float weight_decay, float grad_scale, int first) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float d = g[i] * grad_scale + weight_decay * p[i];
    float b = first ? d : momentum * buf[i] + d;
    buf[i] = b;
    p[i] = p[i] - lr * b;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int mmdyn_act_fwd(const float* u, float* h, int64_t n, int act, void* stream) {
  if (!u || !h) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, u, h, n, act);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_act_bwd(const float* dh, const float* u, float* du, int64_t n, int act, void* stream) {
  if (!dh || !u || !du) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, dh, u, du, n, act);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_act_bwd_b16(const uint16_t* dh, const uint16_t* u, uint16_t* du, int64_t n, int act, int half,
                                 void* stream) {
  if (!dh || !u || !du) return MMDYN_ERR_NULL;
  if (half)
    hipLaunchKernelGGL(act_bwd_kernel<half_t>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, (const half_t*)dh, (const half_t*)u,
                       (half_t*)du, n, act);
  else
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, dh, u, du, n, act);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_dropout_expand(const float* h, const uint8_t* masks, float* out, int P, int B, int H,
                                    float p_drop, void* stream) {
  if (!h || !masks || !out) return MMDYN_ERR_NULL;
  if (p_drop < 0.f || p_drop >= 1.f) return MMDYN_ERR_SHAPE;
  int64_t bh = (int64_t)B * H;
  hipLaunchKernelGGL(dropout_expand_kernel, dim3(ew_grid(P * bh)), dim3(256), 0, ST, h, masks, out, P, bh,
                     1.0f / (1.0f - p_drop));
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_dropout_reduce(const float* dout, const uint8_t* masks, float* dh, int P, int B, int H,
                                    float p_drop, const float* u, int act, void* dh_planes, void* stream) {
  if (!dout || !masks || !dh) return MMDYN_ERR_NULL;
  if (p_drop < 0.f || p_drop >= 1.f) return MMDYN_ERR_SHAPE;
  int64_t bh = (int64_t)B * H;
  hipLaunchKernelGGL(dropout_reduce_kernel, dim3(ew_grid(bh)), dim3(256), 0, ST, dout, masks, dh, P, bh,
                     1.0f / (1.0f - p_drop), u, act, reinterpret_cast<bf16_t*>(dh_planes), H);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_random_masks(uint8_t* masks, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                                  const uint64_t* offset_dev, void* stream) {
  if (!masks) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(random_masks_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, masks, n, p_drop, seed,
                     offset, offset_dev);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_random_normal(float* out, int64_t n, uint64_t seed, uint64_t offset,
                                   const uint64_t* offset_dev, void* stream) {
  if (!out) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(random_normal_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, out, n, seed, offset,
                     offset_dev);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
  if (!counter) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, ST, counter, inc);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_colsum_chunks(int rows) {
  int c = rows / 64;
  return c < 1 ? 1 : (c > 32 ? 32 : c);
}
extern "C" int mmdyn_colsum(const float* x, float* out, float* scratch, int rows, int C, int perm, float beta,
                            uint32_t* ticket, void* stream) {
  if (!x || !out || !scratch) return MMDYN_ERR_NULL;
  if ((perm == 2 && C != 6400) || C % 4 || rows <= 0) return MMDYN_ERR_SHAPE;
  if (rows <= COLSUM_DIRECT_ROWS && !ticket) {
    hipLaunchKernelGGL(colsum_direct_kernel, dim3(ceil_div(C / 4, 8)), dim3(256), 0, ST, x, out, rows, C, perm, beta);
    MMDYN_LAUNCH_CHECK();
  }
  const int chunks = mmdyn_colsum_chunks(rows);
  const int rpc = ceil_div(rows, chunks);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(C / 4, 32), chunks), dim3(256), 0, ST, x, scratch, rows, C,
                     rpc, ticket, out, perm, beta);
  if (ticket) MMDYN_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, ST, scratch, out, chunks, C, perm,
                     beta);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream) {
  if (!src || !dst || !bytes) return MMDYN_ERR_NULL;
  if (n < 0 || n > MMDYN_COPY_MANY_MAX) return MMDYN_ERR_SHAPE;
  if (n == 0) return MMDYN_OK;
  CopyMany cm{};
  int64_t longest = 0;
  for (int i = 0; i < n; ++i) {
    if (bytes[i] < 0) return MMDYN_ERR_SHAPE;
    if (bytes[i] > 0 && (!src[i] || !dst[i])) return MMDYN_ERR_NULL;
    cm.src[i] = (const char*)src[i];
    cm.dst[i] = (char*)dst[i];
    cm.bytes[i] = bytes[i];
    longest = bytes[i] > longest ? bytes[i] : longest;
  }
  int g = ew_grid((longest + 15) / 16);
  if (g > 512) g = 512;
  hipLaunchKernelGGL(copy_many_kernel, dim3(g, n), dim3(256), 0, ST, cm);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_sum_blocks(const float* x, float* out, int P, int64_t n, void* stream) {
  if (!x || !out) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(sum_blocks_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, out, P, n);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_linear_small_fwd(const float* x, const float* W, const float* b, float* y, int rows,
                                      int K, int N, int act, void* stream) {
  if (!x || !W || !y) return MMDYN_ERR_NULL;
  if (N <= 8 && K >= 64) {
    hipLaunchKernelGGL(linear_small_fwd_rowwave_kernel<8>, dim3(ew_grid((int64_t)rows * 64)), dim3(256), 0, ST, x, W,
                       b, y, rows, K, N, act);
    MMDYN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(ew_grid((int64_t)rows * N)), dim3(256), 0, ST, x, W, b, y,
                     rows, K, N, act);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_linear_small_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW,
                                      float* db, int rows, int K, int N, float beta, void* stream) {
  if (!dy || !x || !W || !dW) return MMDYN_ERR_NULL;
  if (dx)
    hipLaunchKernelGGL(linear_small_dx_kernel, dim3(ew_grid((int64_t)rows * K)), dim3(256), 0, ST, dy, W, dx,
                       rows, K, N);
  hipLaunchKernelGGL(linear_small_dw_kernel, dim3(ew_grid((int64_t)N * (K + 1) * 64)), dim3(256), 0, ST, dy,
                     x, dW, db, rows, K, N, beta);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_scale_dev(const float* x, const float* s, float* out, int64_t n, void* stream) {
  if (!x || !s || !out) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, s, out, n);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst) return MMDYN_ERR_NULL;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return MMDYN_ERR_SHAPE;       // 16-byte / 8-byte accesses
  if (n <= 0) return MMDYN_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, ST, src, (bf16_t*)dst, n);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst) return MMDYN_ERR_NULL;
  if (((uintptr_t)dst & 15) || ((uintptr_t)src & 7)) return MMDYN_ERR_SHAPE;
  if (n <= 0) return MMDYN_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, ST, (const bf16_t*)src, dst, n);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                              float weight_decay, float grad_scale, int first, void* stream) {
  if (!p || !g || !momentum_buf) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, p, g, momentum_buf, n, lr, momentum, weight_decay,
                     grad_scale, first);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_adam_step(float* p, const float* g, float* m, float* v, double* state, int64_t n,
                               float lr, float beta1, float beta2, float eps, float grad_scale,
                               void* stream) {
  if (!p || !g || !m || !v || !state) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, ST, state, lr, beta1, beta2);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, p, g, m, v, state, n, beta1,
                     beta2, eps, grad_scale, 0);
  MMDYN_LAUNCH_CHECK();
}
extern "C" int mmdyn_adam_step_guarded(float* p, const float* g, float* m, float* v, double* state, int64_t n,
                                       float lr, float beta1, float beta2, float eps, float grad_scale,
                                       void* stream) {
  if (!p || !g || !m || !v || !state) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, g, n, state);
  hipLaunchKernelGGL(adam_tick_guarded_kernel, dim3(1), dim3(64), 0, ST, state, lr, beta1, beta2);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, ST, p, g, m, v, state, n, beta1,
                     beta2, eps, grad_scale, 1);
  MMDYN_LAUNCH_CHECK();
}
