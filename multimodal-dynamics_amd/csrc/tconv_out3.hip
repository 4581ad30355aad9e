// Direct k4 s2 p1 transposed convolution to 3 output channels: the last decoder layer
// nn.ConvTranspose2d(32, 3, 4, 2, 1) (/root/reference/mmdyn/pytorch/models/vae.py:277), channels-last input,
// NCHW logits out.  With N = 3 an MFMA tile would be 90 % padding and the layer is HBM-bound anyway
// (131 KB in + 49 KB out per sample), so it is a VALU kernel designed around data movement:
//   * one block = a 16x16 tile of INPUT pixels (+1 halo) staged once in LDS (coalesced 128-byte pixel rows),
//     every input byte is read from HBM exactly once (1.27x with halo);
//   * one thread = one input pixel -> its 2x2 output pixels x 3 channels (12 accumulators); the 3x3x32
//     neighbourhood is read from LDS as 16-byte lanes (row stride 36 floats: conflict-free);
//   * the 1536 weights are wave-uniform: the compiler keeps them in SGPRs (s_load), no LDS/VGPR traffic;
//   * each thread stores 8-byte pairs, a wave writes full 128-byte lines of the NCHW planes.
#include "common.h"

namespace {

constexpr int TI = 16;          // input tile edge
constexpr int TH = TI + 2;      // with halo
constexpr int PX_LD = 20;       // floats per staged pixel ROW of one channel (18 + pad)
constexpr int CH_LD = TH * PX_LD;   // floats per channel plane
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Round 3: the arithmetic is packed.  The two output columns (2j, 2j+1) of an input pixel j read the input pixels
// (j+1-tw, j+2-tw) and the ADJACENT weights (kw = 1+2tw, 2tw), so one v_pk_fma_f32 serves both -- IF the two input pixels sit
// in one register pair.  The tile is therefore staged channel-major, [32 ch][18 rows][18 px]: a pixel pair of one channel
// is two neighbouring floats.  768 packed FMAs per thread and no register shuffles, against 1536 + ~1000 v_mov when the pairs
// had to be assembled from pixel-major 16-byte reads (82 -> 5x us per launch on 4 x 256 samples).
// Round 4: FUSED = the input is the layer's pre-BatchNorm tensor y and the staging loop applies the preceding
// BatchNorm2d + Swish (vae.py:275-276) on its way into LDS -- a = swish(gamma * ((y - mean[g]) * rstd[g]) + beta), the expression
// of bn_swish_fwd_kernel, once per staged element (1.27x with the halo) -- so the activated tensor is never written to HBM and
// the separate element-wise pass over the largest activation of the network disappears.
struct Out3Bn {
  const float* mean;      // [G][32]
  const float* rstd;      // [G][32]
  const float* gamma;     // [32]
  const float* beta;      // [32]
  int Bg;                 // samples per BatchNorm group
};
// Round 6: LOSS = the reconstruction term rides in the epilogue.  The logits of a pixel pair are in
// registers when the layer is done; binary_cross_entropy_with_logits against the pass's target (problems.py:433-437, 445-447 with
// a loss mask), its sum into the pass's loss slot and dlogit = (sigmoid(l) - t) * grad_scale are computed there, so the logits
// of the passes nobody reads are never written and read back (bce_logits_groups_kernel: 8 bytes read per logit) -- they are
// materialised only for the group the caller publishes (`logit_group`; -1: all).  Same element expression as the stand-alone
// loss kernel (bce_elem, common.h); per-thread fp32 sums of 12 elements, fp64 from there on.
struct Out3Loss {
  const float* target;    // [Bg][3][2Hi][2Wi]: every group is scored against the same target
  const float* mask;      // [Bg][mask_c][2Hi][2Wi] or null (--mask-loss)
  float* dlogit;          // [G*Bg][3][2Hi][2Wi] or null (evaluation)
  double* loss;           // loss[slot[g]] += sum over group g
  double* unmasked;       // with a mask: the plain sums as well (or null)
  int slot[MMDYN_BCE_GROUPS_MAX];      // < 0: a discarded pass (zero gradient, no loss)
  int mask_c;
  int logit_group;        // group whose logits go to `out` ([Bg][3][2Hi][2Wi]); -1: all groups ([G*Bg]...)
  float grad_scale;
};
template <typename TA, bool FUSED = false, bool LOSS = false>
__global__ __launch_bounds__(256) void tconv_out3_kernel(const TA* __restrict__ a,         // [Bt][Hi][Wi][32]
                                                         const float* __restrict__ w,      // [32][3][4][4]
                                                         float* __restrict__ out,          // [Bt][3][2Hi][2Wi]
                                                         int Hi, int Wi, const Out3Bn bn, const Out3Loss ls) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);           // [32][TH][PX_LD]
  const int tid = threadIdx.x;
  const int tiles_x = Wi / TI;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TI - 1, x0 = tx * TI - 1;           // top-left of the halo tile

  // stage: TH*TH pixels x 8 float4, transposed to channel planes.  All of a thread's loads are issued before the first
  // store: as a load -> store loop the eleven round trips were serial and the launch latency-bound (10 us of a block's 13)
  constexpr int NLD = (TH * TH * 8 + 255) / 256;
  f32x4 val[NLD];
  // (a thread stages the same four channels in every round: 256 % 8 == 0)
  f32x4 bm = {0.f, 0.f, 0.f, 0.f}, br = bm, bg = bm, bb = bm;
  if constexpr (FUSED) {
    const int c0 = (tid & 7) * 4, grp = b / bn.Bg;
    bm = *reinterpret_cast<const f32x4*>(bn.mean + grp * 32 + c0);
    br = *reinterpret_cast<const f32x4*>(bn.rstd + grp * 32 + c0);
    bg = *reinterpret_cast<const f32x4*>(bn.gamma + c0);
    bb = *reinterpret_cast<const f32x4*>(bn.beta + c0);
  }
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int idx = tid + 256 * k;
    const int p = idx >> 3, v = idx & 7;
    const int py = p / TH, px = p - py * TH;
    const int y = y0 + py, x = x0 + px;
    const bool ok = idx < TH * TH * 8 && (unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // (masked lanes read a valid dummy address: a predicated load would be sunk into its own branch)
    f32x4 r = ld4<TA>(a + (ok ? ((size_t)(b * Hi + y) * Wi + x) * 32 + v * 4 : (size_t)0));
    if constexpr (FUSED) {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = swishf_(bg[q] * ((r[q] - bm[q]) * br[q]) + bb[q]);
    }
    val[k] = ok ? r : z;                                  // (padding is zero AFTER the activation, as in the unfused layer)
  }
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int idx = tid + 256 * k;
    if (idx < TH * TH * 8) {
      const int p = idx >> 3, v = idx & 7;
      const int py = p / TH, px = p - py * TH;
      float* dst = tile + (size_t)(v * 4) * CH_LD + py * PX_LD + px;
      dst[0] = val[k][0];
      dst[CH_LD] = val[k][1];
      dst[2 * CH_LD] = val[k][2];
      dst[3 * CH_LD] = val[k][3];
    }
  }
  __syncthreads();

  const int li = tid >> 4, lj = tid & 15;                 // input pixel inside the tile
  f32x2 acc[2][3];                                        // [ph][co] = (output column 2j, 2j+1)
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int co = 0; co < 3; ++co) acc[ph][co] = (f32x2){0.f, 0.f};

  // output (2i+ph, 2j+pw) <- input (i+ph-th, j+pw-tw), kernel tap (kh, kw) = (1-ph+2th, 1-pw+2tw):
  // for a fixed tw the pair pw = (0, 1) reads the input pair (j+1-tw, j+2-tw) [tile columns lj+1-tw, lj+2-tw] and the weight
  // pair (kw = 1+2tw, kw = 2tw)
  const float* base = tile + li * PX_LD + lj;
#pragma unroll 4
  for (int ci = 0; ci < 32; ++ci) {
    const float* pc = base + ci * CH_LD;
    f32x2 in[3][2];                                       // [dy][tw]: tw = 0 -> columns (lj+1, lj+2), tw = 1 -> (lj, lj+1)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const float c0 = pc[dy * PX_LD], c1 = pc[dy * PX_LD + 1], c2 = pc[dy * PX_LD + 2];
      in[dy][0] = (f32x2){c1, c2};
      in[dy][1] = (f32x2){c0, c1};
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int th = 0; th < 2; ++th) {
        const int dy = 1 + ph - th, kh = 1 - ph + 2 * th;
#pragma unroll
        for (int tw = 0; tw < 2; ++tw)
#pragma unroll
          for (int co = 0; co < 3; ++co) {
            const float* wp = w + ((ci * 3 + co) * 4 + kh) * 4 + 2 * tw;       // (kw = 2tw, 2tw+1): (pw = 1, pw = 0)
            const f32x2 w2 = {wp[1], wp[0]};
            acc[ph][co] = __builtin_elementwise_fma(in[dy][tw], w2, acc[ph][co]);
          }
      }
  }

  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int oy = 2 * (ty * TI + li), ox = 2 * (tx * TI + lj);
  if constexpr (!LOSS) {
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
        *reinterpret_cast<f32x2*>(out + (((size_t)b * 3 + co) * Ho + oy + ph) * Wo + ox) = acc[ph][co];
  } else {
    const int grp = b / bn.Bg, smp = b - grp * bn.Bg, slot = ls.slot[grp];
    const bool keep = out != nullptr && (ls.logit_group < 0 || ls.logit_group == grp);
    const int ob = ls.logit_group < 0 ? b : smp;
    float part = 0.f, part_u = 0.f;
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const size_t row = (size_t)(oy + ph) * Wo + ox;
        const f32x2 x = acc[ph][co];
        if (keep) *reinterpret_cast<f32x2*>(out + ((size_t)ob * 3 + co) * Ho * Wo + row) = x;
        f32x2 d = {0.f, 0.f};
        if (slot >= 0) {
          const f32x2 t = *reinterpret_cast<const f32x2*>(ls.target + ((size_t)smp * 3 + co) * Ho * Wo + row);
          if (ls.mask) {
            const f32x2 mk = *reinterpret_cast<const f32x2*>(ls.mask + ((size_t)smp * ls.mask_c + (ls.mask_c == 1 ? 0 : co)) * Ho * Wo + row);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float xm = x[k] * mk[k], tm = t[k] * mk[k];
              float l, sg, lu, su;
              bce_elem(xm, tm, l, sg);
              bce_elem(x[k], t[k], lu, su);
              part += l;
              part_u += lu;
              d[k] = mk[k] * (sg - tm) * ls.grad_scale;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              float l, sg;
              bce_elem(x[k], t[k], l, sg);
              part += l;
              d[k] = (sg - t[k]) * ls.grad_scale;
            }
          }
        }
        if (ls.dlogit) *reinterpret_cast<f32x2*>(ls.dlogit + ((size_t)b * 3 + co) * Ho * Wo + row) = d;
      }
    if (slot >= 0) {          // (block-uniform)
      __shared__ double red[2][4];
      const double s = wave_sum_d((double)part), su = wave_sum_d((double)part_u);
      if ((tid & 63) == 0) {
        red[0][tid >> 6] = s;
        red[1][tid >> 6] = su;
      }
      __syncthreads();
      if (tid == 0) {
        atomicAdd(ls.loss + slot, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        if (ls.mask && ls.unmasked) atomicAdd(ls.unmasked + slot, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
      }
    }
  }
}

}  // namespace

extern "C" int mmdyn_tconv_out3_fwd(const float* a, const float* w, float* out, int Bt, int Hi, int Wi,
                                    void* stream) {
  if (!a || !w || !out) return MMDYN_ERR_NULL;
  if (Bt <= 0 || Hi % TI || Wi % TI || Bt > 65535) return MMDYN_ERR_SHAPE;
  if ((int64_t)Bt * Hi * Wi * 32 >= (1LL << 31)) return MMDYN_ERR_RANGE;
  dim3 grid((Hi / TI) * (Wi / TI), Bt);
  size_t smem = (size_t)32 * CH_LD * sizeof(float);
  hipLaunchKernelGGL(tconv_out3_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, a, w, out, Hi, Wi, Out3Bn{}, Out3Loss{});
  MMDYN_LAUNCH_CHECK();
}

/* The same layer fused with the BatchNorm2d + Swish in front of it: `y` is the pre-BatchNorm output of the layer below
 * ([G*Bg][Hi][Wi][32]; fp32, or 16-bit with b16 = 1 (bf16) / 2 (IEEE half)), mean / rstd [G][32] its batch statistics. */
extern "C" int mmdyn_tconv_out3_bn_fwd(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                       const float* w, float* out, int G, int Bg, int Hi, int Wi, int b16, void* stream) {
  if (!y || !mean || !rstd || !gamma || !beta || !w || !out) return MMDYN_ERR_NULL;
  const int64_t Bt = (int64_t)G * Bg;
  if (G <= 0 || Bg <= 0 || Hi % TI || Wi % TI || Bt > 65535 || b16 < 0 || b16 > 2) return MMDYN_ERR_SHAPE;
  if (Bt * Hi * Wi * 32 >= (1LL << 31)) return MMDYN_ERR_RANGE;
  dim3 grid((Hi / TI) * (Wi / TI), (unsigned)Bt);
  size_t smem = (size_t)32 * CH_LD * sizeof(float);
  const Out3Bn bn{mean, rstd, gamma, beta, Bg};
  if (b16 == 2)
    hipLaunchKernelGGL((tconv_out3_kernel<half_t, true>), grid, dim3(256), smem, (hipStream_t)stream, (const half_t*)y, w, out, Hi, Wi, bn, Out3Loss{});
  else if (b16 == 1)
    hipLaunchKernelGGL((tconv_out3_kernel<bf16_t, true>), grid, dim3(256), smem, (hipStream_t)stream, (const bf16_t*)y, w, out, Hi, Wi, bn, Out3Loss{});
  else
    hipLaunchKernelGGL((tconv_out3_kernel<float, true>), grid, dim3(256), smem, (hipStream_t)stream, (const float*)y, w, out, Hi, Wi, bn, Out3Loss{});
  MMDYN_LAUNCH_CHECK();
}

/* mmdyn_tconv_out3_bn_fwd with the reconstruction term in its epilogue (problems.py:433-437, 445-447): the logits never leave the
 * kernel except for group `logits_group` (-1: all; `logits` may be null: none).  See the header. */
extern "C" int mmdyn_tconv_out3_bn_bce(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                       const float* w, float* logits, int logits_group, const float* target, const float* mask,
                                       int mask_channels, float* dlogit, double* loss_slots, double* unmasked_slots,
                                       const int* slot_of_group, float grad_scale, int G, int Bg, int Hi, int Wi, int b16,
                                       void* stream) {
  if (!y || !mean || !rstd || !gamma || !beta || !w || !target || !loss_slots || !slot_of_group) return MMDYN_ERR_NULL;
  const int64_t Bt = (int64_t)G * Bg;
  if (G <= 0 || G > MMDYN_BCE_GROUPS_MAX || Bg <= 0 || Hi % TI || Wi % TI || Bt > 65535 || b16 < 0 || b16 > 2) return MMDYN_ERR_SHAPE;
  if (logits_group < -1 || logits_group >= G || (mask && mask_channels != 1 && mask_channels != 3)) return MMDYN_ERR_SHAPE;
  if (Bt * Hi * Wi * 32 >= (1LL << 31)) return MMDYN_ERR_RANGE;
  dim3 grid((Hi / TI) * (Wi / TI), (unsigned)Bt);
  size_t smem = (size_t)32 * CH_LD * sizeof(float);
  const Out3Bn bn{mean, rstd, gamma, beta, Bg};
  Out3Loss ls{};
  ls.target = target;
  ls.mask = mask;
  ls.dlogit = dlogit;
  ls.loss = loss_slots;
  ls.unmasked = unmasked_slots;
  for (int i = 0; i < G; ++i) ls.slot[i] = slot_of_group[i];
  ls.mask_c = mask ? mask_channels : 1;
  ls.logit_group = logits_group;
  ls.grad_scale = grad_scale;
  if (b16 == 2)
    hipLaunchKernelGGL((tconv_out3_kernel<half_t, true, true>), grid, dim3(256), smem, (hipStream_t)stream, (const half_t*)y, w, logits, Hi, Wi, bn, ls);
  else if (b16 == 1)
    hipLaunchKernelGGL((tconv_out3_kernel<bf16_t, true, true>), grid, dim3(256), smem, (hipStream_t)stream, (const bf16_t*)y, w, logits, Hi, Wi, bn, ls);
  else
    hipLaunchKernelGGL((tconv_out3_kernel<float, true, true>), grid, dim3(256), smem, (hipStream_t)stream, (const float*)y, w, logits, Hi, Wi, bn, ls);
  MMDYN_LAUNCH_CHECK();
}

/* bf16 activation storage: the input activations are bf16, weights and logits stay fp32 */
extern "C" int mmdyn_tconv_out3_fwd_b16(const uint16_t* a, const float* w, float* out, int Bt, int Hi, int Wi, int half,
                                        void* stream) {
  if (!a || !w || !out) return MMDYN_ERR_NULL;
  if (Bt <= 0 || Hi % TI || Wi % TI || Bt > 65535) return MMDYN_ERR_SHAPE;
  if ((int64_t)Bt * Hi * Wi * 32 >= (1LL << 31)) return MMDYN_ERR_RANGE;
  dim3 grid((Hi / TI) * (Wi / TI), Bt);
  size_t smem = (size_t)32 * CH_LD * sizeof(float);
  if (half)
    hipLaunchKernelGGL(tconv_out3_kernel<half_t>, grid, dim3(256), smem, (hipStream_t)stream, (const half_t*)a, w, out, Hi, Wi, Out3Bn{}, Out3Loss{});
  else
    hipLaunchKernelGGL(tconv_out3_kernel<bf16_t>, grid, dim3(256), smem, (hipStream_t)stream, a, w, out, Hi, Wi, Out3Bn{}, Out3Loss{});
  MMDYN_LAUNCH_CHECK();
}
