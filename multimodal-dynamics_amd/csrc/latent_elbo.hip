// Latent-space and loss kernels of the cnn-mvae step:
//   - product of experts + reparametrisation + KL, all P modality-subset passes in one launch
//     (/root/reference/mmdyn/pytorch/models/vae.py:311-318, 52-61; problems.py:429);
//   - BCE-with-logits / MSE reconstruction sums with their gradients in the same pass
//     (problems.py:433-449), 64-lane shuffle reduction -> one double atomic per block;
//   - the final (sum recon + kl_weight * KL) / B assembly (problems.py:458).
#include "common.h"

namespace {

struct PoeArgs {
  mmdyn_pass_experts pass[MMDYN_MAX_PASSES];
};

constexpr float POE_EPS = 1e-8f;

__global__ __launch_bounds__(256) void poe_fwd_kernel(PoeArgs args, const float* __restrict__ eps_noise,
                                                      float* __restrict__ mu_out, float* __restrict__ lv_out,
                                                      float* __restrict__ z_out, double* __restrict__ kl_sum,
                                                      int with_prior, int B, int L) {
  const int p = blockIdx.y;
  const mmdyn_pass_experts& e = args.pass[p];
  const int64_t n = (int64_t)B * L;
  double kl = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (int64_t)b * L);
    // universal prior expert N(0, 1) first, then the present modalities in the reference's order
    float var0 = 1.f + POE_EPS;
    float sumT = with_prior ? 1.f / (var0 + POE_EPS) : 0.f, sumMuT = 0.f;
#pragma unroll
    for (int m = 0; m < MMDYN_MAX_EXPERTS; ++m) {
      if (e.mu[m]) {
        float mu_m = e.mu[m][(size_t)b * e.ld[m] + l];
        float lv_m = e.lv[m][(size_t)b * e.ld[m] + l];
        float var = expf(lv_m) + POE_EPS;
        float Tm = 1.f / (var + POE_EPS);
        sumT += Tm;
        sumMuT += mu_m * Tm;
      }
    }
    float pd_mu = sumMuT / sumT;
    float pd_var = 1.f / sumT;
    float pd_lv = logf(pd_var + POE_EPS);
    const size_t o = (size_t)p * n + i;
    mu_out[o] = pd_mu;
    lv_out[o] = pd_lv;
    if (z_out) {
      const float zv = eps_noise[o] * expf(0.5f * pd_lv) + pd_mu;
      z_out[o] = zv;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (e.zdst[k]) e.zdst[k][i] = zv;
        if (e.zpl[k]) {                  // ... and as a plane row block: hi | mid | lo of the exact three-term split
          uint32_t h, m, lo;
          split3_bf16(zv, 0.f, h, m, lo);
          bf16_t* pr = reinterpret_cast<bf16_t*>(e.zpl[k]) + (size_t)b * 3 * L + l;
          pr[0] = (bf16_t)(h & 0xffffu);
          pr[L] = (bf16_t)(m & 0xffffu);
          pr[2 * L] = (bf16_t)(lo & 0xffffu);
        }
      }
    }
    kl += (double)(1.f + pd_lv - pd_mu * pd_mu - expf(pd_lv));
  }
  if (kl_sum) {
    kl = wave_sum_d(kl);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&kl_sum[p], -0.5 * (red[0] + red[1] + red[2] + red[3]));
  }
}

__global__ __launch_bounds__(256) void poe_bwd_kernel(PoeArgs args, const float* __restrict__ eps_noise,
                                                      const float* __restrict__ mu_pd,
                                                      const float* __restrict__ lv_pd,
                                                      const float* __restrict__ dz,
                                                      const float* __restrict__ g_mu,
                                                      const float* __restrict__ g_lv, float kl_scale_arg,
                                                      const float* __restrict__ kl_weight_dev, int with_prior, int B, int L) {
  // kl_weight_dev (optional): the KL weight lives in device memory and multiplies kl_scale -- a captured launch then
  // follows the annealing schedule (problems.py:212-216) without being re-captured
  const float kl_scale = kl_weight_dev ? kl_scale_arg * kl_weight_dev[0] : kl_scale_arg;
  const int p = blockIdx.y;
  const mmdyn_pass_experts& e = args.pass[p];
  const int64_t n = (int64_t)B * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (int64_t)b * L);
    const size_t o = (size_t)p * n + i;
    const float mu = mu_pd[o], lv = lv_pd[o];
    float g = dz ? dz[o] : 0.f;
    bool any_dz = dz != nullptr;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (e.dz[k]) {
        g += e.dz[k][i];
        any_dz = true;
      }
    // z = eps * exp(lv/2) + mu ;  KL = -0.5 * sum(1 + lv - mu^2 - exp(lv))
    float dmu_pd = g + kl_scale * mu;
    float dlv_pd = -0.5f * kl_scale * (1.f - expf(lv));
    if (any_dz) dlv_pd += g * eps_noise[o] * 0.5f * expf(0.5f * lv);
    if (g_mu) dmu_pd += g_mu[o];
    if (g_lv) dlv_pd += g_lv[o];
    float Tm[MMDYN_MAX_EXPERTS], mum[MMDYN_MAX_EXPERTS], ex[MMDYN_MAX_EXPERTS];
    float var0 = 1.f + POE_EPS;
    float S = with_prior ? 1.f / (var0 + POE_EPS) : 0.f, N = 0.f;
#pragma unroll
    for (int m = 0; m < MMDYN_MAX_EXPERTS; ++m) {
      Tm[m] = 0.f;
      mum[m] = 0.f;
      ex[m] = 0.f;
      if (e.mu[m]) {
        mum[m] = e.mu[m][(size_t)b * e.ld[m] + l];
        ex[m] = expf(e.lv[m][(size_t)b * e.ld[m] + l]);
        Tm[m] = 1.f / (ex[m] + POE_EPS + POE_EPS);
        S += Tm[m];
        N += mum[m] * Tm[m];
      }
    }
    const float pd_var = 1.f / S;
    const float dvar = dlv_pd / (pd_var + POE_EPS);
    const float invS2 = pd_var * pd_var;
    const float dS = -dvar * invS2 - dmu_pd * N * invS2;
    const float dN = dmu_pd * pd_var;
#pragma unroll
    for (int m = 0; m < MMDYN_MAX_EXPERTS; ++m) {
      if (e.mu[m]) {
        const float dT = dS + dN * mum[m];
        e.dmu[m][(size_t)b * e.ld[m] + l] = dN * Tm[m];
        e.dlv[m][(size_t)b * e.ld[m] + l] = -dT * Tm[m] * Tm[m] * ex[m];
      }
    }
  }
}

// z = eps * exp(lv/2) + mu and/or KL(mu, lv); mu/lv rows of stride ld (vae.py:57-59, problems.py:406)
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mu,
                                                          const float* __restrict__ lv,
                                                          const float* __restrict__ eps_noise,
                                                          float* __restrict__ z, double* __restrict__ kl_sum,
                                                          int B, int L, int ld) {
  const int64_t n = (int64_t)B * L;
  double kl = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (int64_t)b * L);
    const float m = mu[(size_t)b * ld + l], v = lv[(size_t)b * ld + l];
    if (z) z[i] = eps_noise[i] * expf(0.5f * v) + m;
    kl += (double)(1.f + v - m * m - expf(v));
  }
  if (kl_sum) {
    kl = wave_sum_d(kl);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = kl;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(kl_sum, -0.5 * (red[0] + red[1] + red[2] + red[3]));
  }
}

__global__ void reparam_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                   const float* __restrict__ eps_noise, const float* __restrict__ dz,
                                   float kl_scale, float* __restrict__ dmu, float* __restrict__ dlv, int B,
                                   int L, int ld) {
  const int64_t n = (int64_t)B * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (int64_t)b * L);
    const float m = mu[(size_t)b * ld + l], v = lv[(size_t)b * ld + l];
    const float g = dz ? dz[i] : 0.f;
    float gm = g + kl_scale * m;
    float gv = -0.5f * kl_scale * (1.f - expf(v));
    if (dz) gv += g * eps_noise[i] * 0.5f * expf(0.5f * v);
    dmu[(size_t)b * ld + l] = gm;
    dlv[(size_t)b * ld + l] = gv;
  }
}

__device__ __forceinline__ void block_atomic_add(double v, double* dst) {
  v = wave_sum_d(v);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, red[0] + red[1] + red[2] + red[3]);
}

// (bce_elem: common.h -- shared with the last decoder layer's fused loss epilogue, tconv_out3.hip)

// The same reconstruction term for several decoder passes that share ONE target (the live passes of a modality in the
// multi-subset ELBO): logits [G][n], target [n], one loss slot per pass; blockIdx.y = pass.  A pass whose slot is negative
// is a discarded reconstruction: its logit gradient is zero and it adds nothing to the loss.
struct BceGroups {
  int slot[MMDYN_BCE_GROUPS_MAX];
};
template <bool MASKED>
__global__ __launch_bounds__(256) void bce_logits_groups_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ target,
                                                                const float* __restrict__ mask,
                                                                float* __restrict__ dlogit, double* __restrict__ loss,
                                                                double* __restrict__ unmasked, const BceGroups gs,
                                                                int64_t n, int chw, int hw, int mask_c, float grad_scale) {
  const int grp = blockIdx.y, slot = gs.slot[grp];
  const float* __restrict__ lg = logits + (size_t)grp * n;
  float* __restrict__ dl = dlogit ? dlogit + (size_t)grp * n : nullptr;
  const int64_t n4 = n >> 2;
  if (slot < 0) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (dl)
      for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(dl)[i] = zero;
    return;
  }
  double acc = 0.0, acc_u = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 xv = reinterpret_cast<const f32x4*>(lg)[i], t = reinterpret_cast<const f32x4*>(target)[i];
    f32x4 d;
    float part = 0.f;
    if constexpr (MASKED) {
      // the loss mask multiplies logits and target (problems.py:445-447): [B][1 or C][H][W] (mask_c == 1: broadcast over channels)
      const int64_t e0 = i * 4, b = e0 / chw;
      const int rem = (int)(e0 - b * chw);
      const int ch = rem / hw, pix = rem - ch * hw;
      const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + (b * mask_c + (mask_c == 1 ? 0 : ch)) * hw + pix);
      float part_u = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xm = xv[k] * mk[k], tm = t[k] * mk[k];
        float l, sg, lu, su;
        bce_elem(xm, tm, l, sg);
        bce_elem(xv[k], t[k], lu, su);
        part += l;
        d[k] = mk[k] * (sg - tm) * grad_scale;
        part_u += lu;
      }
      acc_u += (double)part_u;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float l, sg;
        bce_elem(xv[k], t[k], l, sg);
        part += l;
        d[k] = (sg - t[k]) * grad_scale;
      }
    }
    acc += (double)part;
    if (dl) reinterpret_cast<f32x4*>(dl)[i] = d;
  }
  block_atomic_add(acc, loss + slot);
  if constexpr (MASKED) {
    if (unmasked) {
      __syncthreads();          // block_atomic_add's scratch is reused
      block_atomic_add(acc_u, unmasked + slot);
    }
  }
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ r, const float* __restrict__ t,
                                                  float* __restrict__ dr, double* __restrict__ loss, int64_t n,
                                                  float grad_scale) {
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float d = r[i] - t[i];
    acc += (double)(d * d);
    if (dr) dr[i] = 2.f * d * grad_scale;
  }
  block_atomic_add(acc, loss);
}

// the pose term of several passes against ONE target (the pose-bearing subsets of the multi-subset ELBO): r / dr [G][n], t [n],
// one loss slot per pass; blockIdx.y = pass.  Same arithmetic per pass as mse_kernel.
__global__ __launch_bounds__(256) void mse_groups_kernel(const float* __restrict__ r, const float* __restrict__ t,
                                                         float* __restrict__ dr, double* __restrict__ loss, const BceGroups gs,
                                                         int64_t n, float grad_scale) {
  const int grp = blockIdx.y;
  r += (size_t)grp * n;
  if (dr) dr += (size_t)grp * n;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = r[i] - t[i];
    acc += (double)(d * d);
    if (dr) dr[i] = 2.f * d * grad_scale;
  }
  block_atomic_add(acc, loss + gs.slot[grp]);
}

__global__ void elbo_assemble_kernel(const double* __restrict__ bce, const double* __restrict__ mse,
                                     const double* __restrict__ kl, float* __restrict__ loss,
                                     float* __restrict__ partials, int P, int B, float kl_weight_arg,
                                     float pose_multiplier, const float* __restrict__ kl_weight_dev) {
  const float kl_weight = kl_weight_dev ? kl_weight_arg * kl_weight_dev[0] : kl_weight_arg;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double tot = 0.0;
    for (int p = 0; p < P; ++p) {
      double v = ((bce ? bce[p] : 0.0) + (double)pose_multiplier * (mse ? mse[p] : 0.0) +
                  (double)kl_weight * (kl ? kl[p] : 0.0)) /
                 (double)B;
      if (partials) partials[p] = (float)v;
      tot += v;
    }
    loss[0] = (float)tot;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

static int copy_passes(const mmdyn_pass_experts* passes, int P, PoeArgs* out) {
  if (!passes) return MMDYN_ERR_NULL;
  if (P < 1 || P > MMDYN_MAX_PASSES) return MMDYN_ERR_SHAPE;
  for (int p = 0; p < P; ++p) out->pass[p] = passes[p];
  return 0;
}

extern "C" int mmdyn_poe_fwd(const mmdyn_pass_experts* passes, const float* eps_noise, float* mu,
                             float* logvar, float* z, double* kl_sum, int with_prior, int P, int B, int L,
                             void* stream) {
  if (!mu || !logvar || (z && !eps_noise)) return MMDYN_ERR_NULL;
  PoeArgs a{};
  if (int e = copy_passes(passes, P, &a)) return e;
  for (int p = 0; p < P; ++p)
    for (int m = 0; m < MMDYN_MAX_EXPERTS; ++m)
      if ((a.pass[p].mu[m] != nullptr) != (a.pass[p].lv[m] != nullptr)) return MMDYN_ERR_NULL;
  int gx = ew_grid((int64_t)B * L);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(poe_fwd_kernel, dim3(gx, P), dim3(256), 0, ST, a, eps_noise, mu, logvar, z, kl_sum,
                     with_prior, B, L);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_poe_bwd(const mmdyn_pass_experts* passes, const float* eps_noise, const float* mu,
                             const float* logvar, const float* dz, const float* g_mu, const float* g_lv,
                             float kl_scale, int with_prior, int P, int B, int L, const float* kl_weight_dev,
                             void* stream) {
  if (!mu || !logvar || (dz && !eps_noise)) return MMDYN_ERR_NULL;
  PoeArgs a{};
  if (int e = copy_passes(passes, P, &a)) return e;
  for (int p = 0; p < P; ++p) {
    for (int m = 0; m < MMDYN_MAX_EXPERTS; ++m)
      if (a.pass[p].mu[m] && (!a.pass[p].lv[m] || !a.pass[p].dmu[m] || !a.pass[p].dlv[m])) return MMDYN_ERR_NULL;
    for (int k = 0; k < 3; ++k)
      if (a.pass[p].dz[k] && !eps_noise) return MMDYN_ERR_NULL;
  }
  int gx = ew_grid((int64_t)B * L);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(poe_bwd_kernel, dim3(gx, P), dim3(256), 0, ST, a, eps_noise, mu, logvar, dz, g_mu, g_lv,
                     kl_scale, kl_weight_dev, with_prior, B, L);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_reparam_fwd(const float* mu, const float* lv, const float* eps_noise, float* z,
                                 double* kl_sum, int B, int L, int ld, void* stream) {
  if (!mu || !lv || (z && !eps_noise)) return MMDYN_ERR_NULL;
  if (ld < L) return MMDYN_ERR_SHAPE;
  int gx = ew_grid((int64_t)B * L);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(gx), dim3(256), 0, ST, mu, lv, eps_noise, z, kl_sum, B, L, ld);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_reparam_bwd(const float* mu, const float* lv, const float* eps_noise, const float* dz,
                                 float kl_scale, float* dmu, float* dlv, int B, int L, int ld, void* stream) {
  if (!mu || !lv || !dmu || !dlv || (dz && !eps_noise)) return MMDYN_ERR_NULL;
  if (ld < L) return MMDYN_ERR_SHAPE;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_grid((int64_t)B * L)), dim3(256), 0, ST, mu, lv, eps_noise,
                     dz, kl_scale, dmu, dlv, B, L, ld);
  MMDYN_LAUNCH_CHECK();
}

static int bce_groups_launch(const float* logits, const float* target, const float* mask, float* dlogit,
                             double* loss_slots, double* unmasked_slots, const int* slot_of_group, int G, int64_t n,
                             int chw, int hw, int mask_channels, float grad_scale, void* stream);

/* one pass: the grouped kernel with a single group (the same arithmetic, to the last bit, as a multi-pass launch) */
extern "C" int mmdyn_bce_logits(const float* logits, const float* target, const float* mask, float* dlogit,
                                double* loss_sum, int64_t n, int chw, int hw, int mask_channels, float grad_scale,
                                void* stream) {
  const int slot0 = 0;
  return bce_groups_launch(logits, target, mask, dlogit, loss_sum, nullptr, &slot0, 1, n, chw, hw, mask_channels, grad_scale,
                           stream);
}

static int bce_groups_launch(const float* logits, const float* target, const float* mask, float* dlogit,
                             double* loss_slots, double* unmasked_slots, const int* slot_of_group, int G, int64_t n,
                             int chw, int hw, int mask_channels, float grad_scale, void* stream) {
  if (!logits || !target || !loss_slots || !slot_of_group) return MMDYN_ERR_NULL;
  if (G <= 0 || G > MMDYN_BCE_GROUPS_MAX || n <= 0 || n % 4) return MMDYN_ERR_SHAPE;
  if (mask && (hw <= 0 || hw % 4 || chw <= 0 || chw % hw || n % chw || (mask_channels != 1 && mask_channels != chw / hw)))
    return MMDYN_ERR_SHAPE;
  BceGroups gs{};
  for (int i = 0; i < G; ++i) gs.slot[i] = slot_of_group[i];
  int g = ew_grid(n / 4);
  if (g > 512) g = 512;
  if (mask)
    hipLaunchKernelGGL(bce_logits_groups_kernel<true>, dim3(g, G), dim3(256), 0, ST, logits, target, mask, dlogit,
                       loss_slots, unmasked_slots, gs, n, chw, hw, mask_channels, grad_scale);
  else
    hipLaunchKernelGGL(bce_logits_groups_kernel<false>, dim3(g, G), dim3(256), 0, ST, logits, target, mask, dlogit,
                       loss_slots, unmasked_slots, gs, n, 0, 0, 1, grad_scale);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_bce_logits_groups(const float* logits, const float* target, float* dlogit, double* loss_slots,
                                       const int* slot_of_group, int G, int64_t n, float grad_scale, void* stream) {
  return bce_groups_launch(logits, target, nullptr, dlogit, loss_slots, nullptr, slot_of_group, G, n, 0, 0, 1, grad_scale,
                           stream);
}

extern "C" int mmdyn_bce_logits_groups_masked(const float* logits, const float* target, const float* mask, float* dlogit,
                                              double* loss_slots, double* unmasked_slots, const int* slot_of_group, int G,
                                              int64_t n, int chw, int hw, int mask_channels, float grad_scale,
                                              void* stream) {
  if (!mask) return MMDYN_ERR_NULL;
  return bce_groups_launch(logits, target, mask, dlogit, loss_slots, unmasked_slots, slot_of_group, G, n, chw, hw,
                           mask_channels, grad_scale, stream);
}

extern "C" int mmdyn_mse(const float* r, const float* t, float* dr, double* loss_sum, int64_t n,
                         float grad_scale, void* stream) {
  if (!r || !t || !loss_sum) return MMDYN_ERR_NULL;
  int g = ew_grid(n);
  if (g > 256) g = 256;
  hipLaunchKernelGGL(mse_kernel, dim3(g), dim3(256), 0, ST, r, t, dr, loss_sum, n, grad_scale);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_mse_groups(const float* r, const float* t, float* dr, double* loss_slots, const int* slot_of_group, int G,
                                int64_t n, float grad_scale, void* stream) {
  if (!r || !t || !loss_slots || !slot_of_group) return MMDYN_ERR_NULL;
  if (G <= 0 || G > MMDYN_BCE_GROUPS_MAX || n <= 0) return MMDYN_ERR_SHAPE;
  BceGroups gs{};
  for (int i = 0; i < G; ++i) {
    if (slot_of_group[i] < 0) return MMDYN_ERR_SHAPE;
    gs.slot[i] = slot_of_group[i];
  }
  int g = ew_grid(n);
  if (g > 64) g = 64;
  hipLaunchKernelGGL(mse_groups_kernel, dim3(g, G), dim3(256), 0, ST, r, t, dr, loss_slots, gs, n, grad_scale);
  MMDYN_LAUNCH_CHECK();
}

extern "C" int mmdyn_elbo_assemble(const double* bce, const double* mse, const double* kl, float* loss,
                                   float* partials, int P, int B, float kl_weight, float pose_multiplier,
                                   const float* kl_weight_dev, void* stream) {
  if (!loss) return MMDYN_ERR_NULL;
  hipLaunchKernelGGL(elbo_assemble_kernel, dim3(1), dim3(64), 0, ST, bce, mse, kl, loss, partials, P, B,
                     kl_weight, pose_multiplier, kl_weight_dev);
  MMDYN_LAUNCH_CHECK();
}
