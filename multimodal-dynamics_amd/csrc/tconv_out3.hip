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
constexpr int PIX_LD = 36;      // floats per staged pixel (32 channels + 16-byte pad)

template <typename TA>
__global__ __launch_bounds__(256) void tconv_out3_kernel(const TA* __restrict__ a,         // [Bt][Hi][Wi][32]
                                                         const float* __restrict__ w,      // [32][3][4][4]
                                                         float* __restrict__ out,          // [Bt][3][2Hi][2Wi]
                                                         int Hi, int Wi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);           // [TH*TH][PIX_LD]
  const int tid = threadIdx.x;
  const int tiles_x = Wi / TI;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TI - 1, x0 = tx * TI - 1;           // top-left of the halo tile

  // stage: TH*TH pixels x 8 float4
  for (int idx = tid; idx < TH * TH * 8; idx += 256) {
    const int p = idx >> 3, v = idx & 7;
    const int py = p / TH, px = p - py * TH;
    const int y = y0 + py, x = x0 + px;
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi)
      val = ld4<TA>(a + ((size_t)(b * Hi + y) * Wi + x) * 32 + v * 4);
    *reinterpret_cast<f32x4*>(&tile[p * PIX_LD + v * 4]) = val;
  }
  __syncthreads();

  const int li = tid >> 4, lj = tid & 15;                 // input pixel inside the tile
  float acc[2][2][3];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int co = 0; co < 3; ++co) acc[ph][pw][co] = 0.f;

  // output (2i+ph, 2j+pw) <- input (i+ph-th, j+pw-tw), kernel tap (1-ph+2th, 1-pw+2tw)
#pragma unroll
  for (int c4 = 0; c4 < 8; ++c4) {
    f32x4 nb[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        nb[dy][dx] = *reinterpret_cast<const f32x4*>(&tile[((li + dy) * TH + (lj + dx)) * PIX_LD + c4 * 4]);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
          for (int tw = 0; tw < 2; ++tw) {
            const int dy = 1 + ph - th, dx = 1 + pw - tw;          // neighbour index 0..2
            const int kh = 1 - ph + 2 * th, kw = 1 - pw + 2 * tw;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ci = c4 * 4 + k;
              const float v = nb[dy][dx][k];
#pragma unroll
              for (int co = 0; co < 3; ++co)
                acc[ph][pw][co] = fmaf(v, w[((ci * 3 + co) * 4 + kh) * 4 + kw], acc[ph][pw][co]);
            }
          }
  }

  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int oy = 2 * (ty * TI + li), ox = 2 * (tx * TI + lj);
#pragma unroll
  for (int co = 0; co < 3; ++co)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      float2 v2 = make_float2(acc[ph][0][co], acc[ph][1][co]);
      *reinterpret_cast<float2*>(out + (((size_t)b * 3 + co) * Ho + oy + ph) * Wo + ox) = v2;
    }
}

}  // namespace

extern "C" int mmdyn_tconv_out3_fwd(const float* a, const float* w, float* out, int Bt, int Hi, int Wi,
                                    void* stream) {
  if (!a || !w || !out) return MMDYN_ERR_NULL;
  if (Bt <= 0 || Hi % TI || Wi % TI || Bt > 65535) return MMDYN_ERR_SHAPE;
  if ((int64_t)Bt * Hi * Wi * 32 >= (1LL << 31)) return MMDYN_ERR_RANGE;
  dim3 grid((Hi / TI) * (Wi / TI), Bt);
  size_t smem = (size_t)TH * TH * PIX_LD * sizeof(float);
  hipLaunchKernelGGL(tconv_out3_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, a, w, out, Hi, Wi);
  MMDYN_LAUNCH_CHECK();
}

/* bf16 activation storage: the input activations are bf16, weights and logits stay fp32 */
extern "C" int mmdyn_tconv_out3_fwd_b16(const uint16_t* a, const float* w, float* out, int Bt, int Hi, int Wi, int half,
                                        void* stream) {
  if (!a || !w || !out) return MMDYN_ERR_NULL;
  if (Bt <= 0 || Hi % TI || Wi % TI || Bt > 65535) return MMDYN_ERR_SHAPE;
  if ((int64_t)Bt * Hi * Wi * 32 >= (1LL << 31)) return MMDYN_ERR_RANGE;
  dim3 grid((Hi / TI) * (Wi / TI), Bt);
  size_t smem = (size_t)TH * TH * PIX_LD * sizeof(float);
  if (half)
    hipLaunchKernelGGL(tconv_out3_kernel<half_t>, grid, dim3(256), smem, (hipStream_t)stream, (const half_t*)a, w, out, Hi, Wi);
  else
    hipLaunchKernelGGL(tconv_out3_kernel<bf16_t>, grid, dim3(256), smem, (hipStream_t)stream, a, w, out, Hi, Wi);
  MMDYN_LAUNCH_CHECK();
}
