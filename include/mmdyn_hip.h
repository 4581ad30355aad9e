/*
 * mmdyn_hip.h -- C ABI of libmmdyn_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * cnn-mvae / cnn-vae training and inference hot path of SAIC-MONTREAL/multimodal-dynamics.
 *
 * The reference has no native boundary: the path sits behind torch.nn modules
 * (mmdyn/pytorch/models/vae.py) and torch.nn.functional losses (mmdyn/pytorch/problems/problems.py).
 * Every entry point below replaces the stock ATen op(s) named in its comment (reference file:line),
 * and is what a maintainer would bind with ctypes from those modules.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless stated; no allocation, no ownership transfer;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (graph-capturable);
 *   - return value: 0 = ok, MMDYN_ERR_* (<0) = argument error detected on the host before launch,
 *     >0 = hipError_t of the launch;
 *   - activations between layers are kept channels-last: a [rows][C] matrix whose row index is
 *     (sample, y, x); the reference's NCHW tensors appear only at the image input / logits output;
 *   - "groups": a batch of G*Bg samples made of G independent sub-batches (one per modality-subset
 *     pass of _evaluate_mvae, problems.py:473-546); train-mode BatchNorm statistics are per group.
 */
#ifndef MMDYN_HIP_H
#define MMDYN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMDYN_OK 0
#define MMDYN_ERR_SHAPE (-1)   /* dimension not supported by the kernel (see each function) */
#define MMDYN_ERR_NULL (-2)    /* required pointer is null */
#define MMDYN_ERR_RANGE (-3)   /* tensor too large for 32-bit element offsets */

#define MMDYN_ACT_NONE 0
#define MMDYN_ACT_SWISH 1      /* x * sigmoid(x), vae.py:331-334 */
#define MMDYN_ACT_RELU 2

/* implicit-GEMM modes (k = 4 everywhere, as in vae.py:198-206, 268-277) */
#define MMDYN_DENSE 0          /* plain rows x Cin matrix (nn.Linear, and the col-matrix GEMMs) */
#define MMDYN_CONV 1           /* gather form: out(r,c) <- in(r*s+o+kh, c*s+o+kw), 16 taps */
#define MMDYN_TCONV_S2P1 2     /* transposed k4 s2 p1, four output-parity classes of 4 taps each */
#define MMDYN_IM2COL3 3        /* k4 s2 p1 window of an NCHW 3-channel tensor gathered on the fly: virtual Cin = 64
                                  (k = ci*16 + kh*4 + kw, 48 real + 16 zero); rows = output pixels.  Lowers
                                  nn.Conv2d(3,32,4,2,1) (vae.py:198) and the backward of nn.ConvTranspose2d(32,3,4,2,1)
                                  (vae.py:277) onto the MFMA GEMMs without materialising an im2col matrix */

#define MMDYN_TCONV_S1P0 4      /* transposed k4 s1 p0 (Ho = Hi+3): rows ordered (output pixel, sample) inside a group so
                                  every tile covers one output pixel and only its 1..16 valid taps are multiplied --
                                  exactly the useful MACs, no zero padding, no column matrix */

const char* mmdyn_version(void);
/* ABI revision of this header.  It changes whenever an exported signature or a workspace requirement changes (round 4 added
 * the `ws` argument of mmdyn_igemm_nt_dgrad_act / _dgrad_bn and the slab workspace of large launches: revision 4; round 5 the
 * plane-packed weight kinds of the pack plan: revision 5; round 6 the `zdst` field of mmdyn_pass_experts, the arrival-flag words in
 * front of the slab workspace (mmdyn_igemm_slab_floats*) and new entry points: revision 6).  A binding checks mmdyn_abi_version() == MMDYN_ABI_VERSION right after
 * loading the library (mmdyn_hip/_lib.py does) so that a caller built against an older header fails at load time instead of
 * passing its stream handle where the library now expects a workspace pointer. */
#define MMDYN_ABI_VERSION 6
int mmdyn_abi_version(void);

/* ---- MFMA implicit GEMM, "NT" form ---------------------------------------------------------
 * C[row][n] = act( sum_{tap,ci} A_tap[row][ci] * Bp[widx(tap)][n][ci] + bias[n] )
 * Replaces: nn.Conv2d forward (vae.py:200,203,206), nn.ConvTranspose2d forward (vae.py:271,274),
 * their input-gradient kernels, and nn.Linear forward / input-gradient (vae.py:211,215,216,264).
 *   A      : NHWC activations [G*Bg][Hi][Wi][Cin]
 *   Bp     : packed weights [taps][N][Cin] (see mmdyn_pack_conv_weight)
 *   C      : NHWC output [G*Bg][Ho][Wo] rows of stride ldc (>= N); pre-activation (+bias)
 *   C_act  : optional second output = act(C) (null: none)
 *   stats  : optional per-tile BatchNorm partial sums [G][T][2][N], T = mmdyn_igemm_stat_tiles(...)
 *   splitk : >1 only for DENSE: partial products go to `ws` ([splitk][rows][N]) and
 *            mmdyn_splitk_reduce finishes (bias/act/second output are applied there).
 *   ws     : with splitk == 1 the LARGE launches (persistent stream-K kernel, csrc/igemm_wsp.hip) park the pieces of tiles
 *            that straddle two blocks there: ws must hold mmdyn_igemm_slab_floats(...) floats whenever that query answers
 *            > 0 (MMDYN_ERR_NULL otherwise); NULL is fine when it answers 0.  Same rule for mmdyn_igemm_nt_mx
 *            (mmdyn_igemm_slab_floats_mx).
 *   arrival_flags (mmdyn_igemm_nt_mx, _dgrad_bn, _dgrad_act; revision 6): MMDYN_IGEMM_FLAG_WORDS uint32 words that are ZERO when the
 *            launch starts and that the launch leaves zero, owned by this launch until it has completed (a launch captured
 *            into a graph owns them for the graph's life).  With them the LAB build of the library (make lab) finishes a tile
 *            whose K range straddles several blocks INSIDE the launch -- the piece that arrives last on the tile's word sums the
 *            pieces in K order and runs the epilogue: same results bit for bit, no fix-up launch.  The PRODUCT library accepts
 *            and ignores the argument: that form measured 3-6 % slower on the train step (docs/LAB_NOTES.md H.a), so slabs + the
 *            fix-up launch stay.  NULL: slabs + the fix-up launch in either build.
 * Requirements: Cin % 32 == 0, N % 32 == 0.  v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate. */
#define MMDYN_IGEMM_FLAG_WORDS 8192
int mmdyn_igemm_nt(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                   float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                   int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                   void* stream);
/* Input-gradient GEMM with the backward of a plain activation in its epilogue (replaces the separate element-wise
 * x * act'(u) pass of loss.backward() through Swish / ReLU, vae.py:14-19, 215, 267, 331-334):
 *     C = (A x Bp) * act'(u),   u = the layer's saved pre-activation at the C positions (row stride N),
 * act = MMDYN_ACT_SWISH or MMDYN_ACT_RELU (for ReLU u may be the activated output: same sign).  No bias, second output,
 * statistics or split-K.  flags: 0 = fp32 everywhere; otherwise as mmdyn_igemm_nt_mx (bit 3: u is bf16).
 * ws: workspace of mmdyn_igemm_slab_floats(...) floats (NULL when that is 0). */
int mmdyn_igemm_nt_dgrad_act(const void* A, const void* Bp, void* C, const void* u, int act, int mode, int G, int Bg,
                             int Hi, int Wi, int Cin, int Ho, int Wo, int N, int stride, int offset, int flags,
                             float* ws, uint32_t* arrival_flags, void* stream);
/* Input-gradient GEMM with the BatchNorm+Swish backward of the PRECEDING layer fused into its epilogue.
 * The tile of dL/d(activation) never reaches HBM as such: with y the layer's saved pre-BatchNorm output (same
 * rows/columns as C) and xhat = (y - mean[g]) * rstd[g], the kernel writes
 *     C = du = dL/da * swish'(gamma*xhat + beta)
 * and the per-tile column sums (du, du*xhat) into stats[G][T][2][N] (T = mmdyn_igemm_stat_tiles), which feed
 * mmdyn_bn_bwd_finalize directly -- the separate reduction pass (mmdyn_bn_swish_bwd_reduce: one more read of da
 * and y) disappears; mmdyn_bn_swish_bwd_apply(da_is_du = 1) finishes the layer.  bf16: 0 = fp32 matrix cores,
 * 1 = bf16, 2 = fp16 operands (T = mmdyn_igemm_stat_tiles_bf16 for both), 3 = fp32 arithmetic with the three-term split allowed
 * (flag bit 7 of mmdyn_igemm_nt_mx; T = mmdyn_igemm_stat_tiles_mx(..., 128)).  No bias / activation / split-K here.
 * ws: workspace of mmdyn_igemm_slab_floats(...) floats (NULL when that is 0). */
int mmdyn_igemm_nt_dgrad_bn(const float* A, const float* Bp, float* C, float* stats, const float* y,
                            const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N,
                            int stride, int offset, int bf16, float* ws, uint32_t* arrival_flags, void* stream);
/* Workspace (floats) the fp32 launch of a shape wants in its `ws` argument when it is not split over K: the persistent,
 * stream-K-scheduled ring kernel (csrc/igemm_wsp.hip) accumulates a tile whose K range straddles two blocks in pieces and
 * parks the pieces there for its fix-up launch.  0 = none (ws may be NULL). */
int mmdyn_igemm_slab_floats(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);

/* Same contract, bf16 matrix cores: the fp32 operands are rounded to bf16 (round-to-nearest-even) on their way
 * into the MFMA (v_mfma_f32_32x32x16_bf16), products are accumulated in fp32, everything in HBM stays fp32.
 * The reduced-precision mode of BASELINE configs[2] ("bf16"); never used by the fp32 path. */
int mmdyn_igemm_nt_bf16(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                        float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                        int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                        void* stream);
/* Same contract, fp16 matrix cores: operands rounded to IEEE half (RNE) on their way into v_mfma_f32_32x32x16_f16, fp32
 * accumulate, everything in HBM fp32 -- "MFMA fp16 conv with fp32 accumulate" of BASELINE configs[4].  Partial-sum tile
 * count: mmdyn_igemm_stat_tiles_bf16. */
int mmdyn_igemm_nt_f16(const float* A, const float* Bp, const float* bias, float* C, float* C_act,
                       float* stats, float* ws, int mode, int G, int Bg, int Hi, int Wi, int Cin,
                       int Ho, int Wo, int N, int ldc, int stride, int offset, int act, int splitk,
                       void* stream);
/* T of the `stats` argument for the shape: what mmdyn_igemm_nt / mmdyn_igemm_nt_dgrad_bn(bf16 = 0) write ... */
int mmdyn_igemm_stat_tiles(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
/* ... and what the bf16 matrix-core variants (mmdyn_igemm_nt_bf16, mmdyn_igemm_nt_mx, dgrad_bn(bf16 = 1)) write */
int mmdyn_igemm_stat_tiles_bf16(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
/* ... and of the mixed-storage entry point mmdyn_igemm_nt_mx for the given flags (launches whose two operands are both 16-bit
 * in HBM may run the persistent ring kernel, csrc/igemm_wsp.hip, which writes one partial tile per wave row); flags == 0:
 * mmdyn_igemm_stat_tiles.  mmdyn_igemm_slab_floats_mx: the `ws` workspace of such a launch (see mmdyn_igemm_slab_floats). */
int mmdyn_igemm_stat_tiles_mx(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int flags);
int mmdyn_igemm_slab_floats_mx(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int flags);
/* Grouped dense GEMM (round 4): G independent problems of ONE shape in one launch,
 *   C_g[rows][N] = A_g[rows][K] . Bp_g[N][K]^T (+ bias_g),   g = 0 .. G-1,
 * group g at A + g*rows*K, Bp + g*N*K, bias + g*N, C / C_act / u + g*rows*N.  Replaces the three nn.Linear pairs
 * linear_means | linear_log_var (vae.py:211-216, 239-240) of the visual, tactile and pose encoders at the product-of-experts
 * join, forward and input gradient: 1024 x 512 x 512 each, half a chip's worth of tiles when launched alone.
 * u != NULL: C = (A . Bp^T) * act'(u) (activation backward in the epilogue; bias and C_act must be NULL).
 * flags: as mmdyn_igemm_nt_mx (0 = fp32 everywhere).  K % 32 == 0, N % 32 == 0. */
int mmdyn_igemm_nt_grouped(const void* A, const void* Bp, const float* bias, void* C, void* C_act, const void* u, int G,
                           int rows, int K, int N, int act, int flags, void* stream);
int mmdyn_splitk_reduce(const float* ws, const float* bias, float* C, float* C_act, int splitk,
                        int rows, int N, int act, void* stream);

/* ---- MFMA weight gradient, "TN" form -------------------------------------------------------
 * partial[chunk][tap][cd][cg] = sum_{rows in chunk} D[row][cd] * G_tap[row][cg]
 * Replaces the weight-gradient kernels of Conv2d / ConvTranspose2d / Linear.
 *   D  : dense rows [Bt*Hr*Wr][Cd];  Gt : gathered operand, NHWC [Bt][Hi][Wi][Cg]
 *   mode DENSE: one tap, Gt rows == D rows.  mode CONV: 16 taps, pixel (r*s+o+kh, c*s+o+kw).
 * mmdyn_wgrad_reduce sums the chunks and writes the reference's canonical layout:
 *   perm 0: canon[cd][cg][tap] (conv [Cout][Cin][4][4] / convT [Cin][Cout][4][4]; Linear [out][in])
 *   perm 1: linear with permuted columns, cg = hw*256+c -> canon[cd][c*25+hw]  (Encoder fc_net.0)
 *   perm 2: linear with permuted rows,    cd = hw*256+c -> canon[c*25+hw][cg]  (Decoder upsample.0)
 *   cg_canon <= Cg drops zero-padded gathered columns (the 48 -> 64 padded first/last layers).
 * Requirements: Cd % 32 == 0, Cg % 32 == 0. */
int mmdyn_wgrad_tn(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr, int Wr,
                   int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks, void* stream);
/* bf16 matrix cores (v_mfma_f32_32x32x8_bf16), fp32 accumulate; see mmdyn_igemm_nt_bf16 */
int mmdyn_wgrad_tn_bf16(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr, int Wr,
                        int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks, void* stream);
/* fp16 matrix cores (v_mfma_f32_32x32x8_f16), fp32 accumulate; see mmdyn_igemm_nt_f16 */
int mmdyn_wgrad_tn_f16(const float* D, const float* Gt, float* partial, int mode, int Bt, int Hr, int Wr,
                       int Cd, int Hi, int Wi, int Cg, int stride, int offset, int chunks, void* stream);
/* recommended `chunks` (a multiple of 4) for mmdyn_wgrad_tn; partial must hold chunks*taps*Cd*Cg floats */
/* Grouped weight gradient (round 4; DENSE): group g owns rows [g*rows, (g+1)*rows) of D [G*rows][Cd] and Gt [G*rows][Cg];
 * partial is [chunks][G][Cd][Cg], so ONE mmdyn_wgrad_reduce(partial, canon, chunks, 1, G*Cd, Cg, Cg, 0, beta) writes the G
 * gradients [G][Cd][Cg] (the fused engine keeps the heads' weights of the three encoders adjacent).  chunks as
 * mmdyn_wgrad_chunks_mx(DENSE, rows, Cd, Cg, flags); flags as mmdyn_wgrad_tn_mx (not both operands 16-bit). */
int mmdyn_wgrad_tn_grouped(const void* D, const void* Gt, float* partial, int G, int rows, int Cd, int Cg, int chunks,
                           int flags, void* stream);
/* Round 4, SURVEY.md section 7 step 5 (BatchNorm-apply + Swish fused into the neighbouring kernels): the weight gradient of
 * the decoder's last layer nn.ConvTranspose2d(32, 3, 4, 2, 1) (vae.py:277) with the BatchNorm2d + Swish in front of it
 * (vae.py:275-276) recomputed on the operand fetch.  y [G*Bg][Hr][Hr][32]: the pre-BatchNorm tensor (fp32; y_b16 = 1 bf16,
 * 2 IEEE half); mean / rstd [G][32]; Gt: the NCHW logit gradient [G*Bg][3][2Hr][2Hr]; partial [chunks][32][64] as
 * mmdyn_wgrad_tn(MMDYN_IM2COL3) writes it (chunks = mmdyn_wgrad_chunks(MMDYN_IM2COL3, ...)).  Hr = 32, 64 or 128. */
int mmdyn_wgrad_out3_bn(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        const float* Gt, float* partial, int G, int Bg, int Hr, int chunks, int y_b16, void* stream);
int mmdyn_wgrad_chunks(int mode, int rows, int Cd, int Cg);
/* ... for the kernel the flags of mmdyn_wgrad_tn_mx select (both operands 16-bit in HBM: the all-16-bit kernels' tiles; bit 7
 * alone: the three-term-split kernels, whose LDS planes let fewer blocks share a CU) */
int mmdyn_wgrad_chunks_mx(int mode, int rows, int Cd, int Cg, int flags);
int mmdyn_wgrad_reduce(const float* partial, float* canon, int chunks, int taps, int Cd, int Cg,
                       int cg_canon, int perm, float beta, void* stream);

/* ---- weight packing (canonical reference layout -> GEMM operand layout) --------------------- */
/* Wc[d0][d1][16] -> P[tap][d0][d1] (swap=0) or P[tap][d1][d0] (swap=1) */
int mmdyn_pack_conv_weight(const float* Wc, float* P, int d0, int d1, int swap, void* stream);
/* generic 2-D repack: out[r][c] (rows_out x cols_out, zero padded) = in[...] with
 * mode 0: copy/pad in[r][c] (in is rows_in x cols_in)
 * mode 1: transpose  out[r][c] = in[c][r]
 * mode 2: column permute out[r][hw*256+ch] = in[r][ch*25+hw]
 * mode 3: row permute    out[hw*256+ch][c] = in[ch*25+hw][c]
 * mode 4: transpose of mode 2: out[hw*256+ch][c] = in[c][ch*25+hw]
 * mode 5: transpose of mode 3: out[r][hw*256+ch] = in[ch*25+hw][r] */
int mmdyn_repack2d(const float* in, float* out, int rows_in, int cols_in, int rows_out, int cols_out,
                   int mode, void* stream);

/* mmdyn_repack2d writing a rows_out x cols_out block into a wider matrix (row stride ld_out >= cols_out) */
int mmdyn_repack2d_ld(const float* in, float* out, int rows_in, int cols_in, int rows_out, int cols_out,
                      int ld_out, int mode, void* stream);
/* the same two packs with a 16-bit destination (RNE; half = 0: bf16, half = 1: IEEE half): in the 16-bit storage modes
 * the weights are packed straight to the matrix cores' operand type, which halves the weight bytes every implicit-GEMM
 * block pulls through L2 */
int mmdyn_pack_conv_weight_b16(const float* Wc, void* P, int d0, int d1, int swap, int half, void* stream);
int mmdyn_repack2d_ld_b16(const float* in, void* out, int rows_in, int cols_in, int rows_out, int cols_out,
                          int ld_out, int mode, int half, void* stream);
/* A whole step's weight repacks in one launch.  `plan_dev` is a DEVICE array of n entries (built once: the
 * parameter storage of a training run does not move).  kind 0..5 = mmdyn_repack2d modes with an output leading
 * dimension ld_out (>= cols_out); kind 100 / 101 = mmdyn_pack_conv_weight with swap 0 / 1 (rows_in = d0,
 * cols_in = d1). */
typedef struct {
  const float* src;
  float* dst;
  int kind, rows_in, cols_in, rows_out, cols_out, ld_out;
  int dst_bf16;   /* 1: dst is a bf16 tensor (ld_out in elements), 2: an IEEE-half tensor; the GEMM operands of the 16-bit modes;
                     3 (kinds 100 / 101 only): dst is a PLANE tensor, rows (tap, x) of [plane][y] bf16 -- the exact three-term split
                     of the packed weight (mmdyn_split_planes of the kind's fp32 output), the Bp of the plane launches */
} mmdyn_pack_entry;
int mmdyn_pack_plan(const mmdyn_pack_entry* plan_dev, int n, void* stream);

/* ---- first / last layer helpers (3-channel NCHW side) --------------------------------------- */
/* col[(b*Ho+ho)*Wo+wo][ci*16+kh*4+kw] = x[b][ci][2ho-1+kh][2wo-1+kw], columns 48..63 zero.
 * x: NCHW [Bt][3][H][W]; col: [Bt*(H/2)*(W/2)][64].  Lowers nn.Conv2d(3,32,4,2,1) (vae.py:198) and the
 * backward of nn.ConvTranspose2d(32,3,4,2,1) (vae.py:277) onto the MFMA GEMMs above. */
int mmdyn_im2col_nchw3(const float* x, float* col, int Bt, int H, int W, void* stream);
/* transposed-conv scatter as a gather: out(ho,wo,c) = sum_{kh,kw} col[(hi,wi)][...], hi=(ho+p-kh)/s.
 *   tap_major=1: col column = tap*C + c (NHWC out, [Bt][Ho][Wo][C]);
 *   tap_major=0: col column = c*16 + tap and the output is NCHW [Bt][C][Ho][Wo] (logits). */
int mmdyn_col2im_k4(const float* col, float* out, int Bt, int Hi, int Wi, int Ho, int Wo, int C,
                    int ldcol, int stride, int pad, int tap_major, void* stream);

/* nn.ConvTranspose2d(32, 3, 4, 2, 1) forward (vae.py:277) as a direct LDS-tiled VALU kernel: a is NHWC
 * [Bt][Hi][Wi][32], w the reference's [32][3][4][4], out NCHW logits [Bt][3][2Hi][2Wi]; Hi, Wi % 16 == 0. */
int mmdyn_tconv_out3_fwd(const float* a, const float* w, float* out, int Bt, int Hi, int Wi, void* stream);
/* ... fused with the BatchNorm2d + Swish in front of it (vae.py:275-277): y [G*Bg][Hi][Wi][32] is the pre-BatchNorm output of the
 * layer below (fp32; b16 = 1 bf16, 2 IEEE half), mean / rstd [G][32] its batch statistics; the activation is applied while
 * the input tile is staged, the activated tensor never exists in HBM. */
int mmdyn_tconv_out3_bn_fwd(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            const float* w, float* out, int G, int Bg, int Hi, int Wi, int b16, void* stream);
/* ... and with the reconstruction term of the ELBO in its epilogue (round 6): F.binary_cross_entropy_with_logits(recon, target,
 * reduction='sum') of problems.py:433-437 (with `mask`, [Bg][mask_channels][2Hi][2Wi], mask_channels 1 or 3: torch.mul of logits and
 * target with the loss mask first, problems.py:445-447; `unmasked_slots`, may be null, then also gets the plain sums) for the G
 * groups of the batch against ONE target [Bg][3][2Hi][2Wi]: loss_slots[slot_of_group[g]] += sum over group g (slot < 0: a discarded
 * pass -- zero gradient, no loss); dlogit [G*Bg][3][2Hi][2Wi] (null in evaluation) = (sigmoid(logit) - target) * grad_scale, the
 * gradient the backward of the layer consumes.  The logits themselves are written only for group `logits_group` (`logits`
 * [Bg][3][2Hi][2Wi]; -1: for all groups, [G*Bg]...; `logits` null: for none) -- the reconstruction the caller publishes
 * (problems.py:537-545).  Element arithmetic identical to mmdyn_bce_logits_groups. */
int mmdyn_tconv_out3_bn_bce(const void* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            const float* w, float* logits, int logits_group, const float* target, const float* mask,
                            int mask_channels, float* dlogit, double* loss_slots, double* unmasked_slots,
                            const int* slot_of_group, float grad_scale, int G, int Bg, int Hi, int Wi, int b16, void* stream);

/* ---- train-mode BatchNorm2d + Swish, channels-last, per group (vae.py:201-208, 269-276) ----- */
/* column sums of y and y*y over row chunks -> partial[G][T][2][C], T = mmdyn_colstats_tiles(rows_per_group) */
int mmdyn_colstats(const float* y, float* partial, int G, int rows_per_group, int C, void* stream);
int mmdyn_colstats_tiles(int rows_per_group);
/* partial -> mean/rstd [G][C]; running stats EMA (momentum 0.1, unbiased var) applied for the groups in
 * order, `repeat` times each (the reference re-runs identical encoder trunks: SURVEY.md 3.2);
 * num_batches_tracked (int64) += G*repeat.  running_* may be null. */
int mmdyn_bn_finalize(const float* partial, float* mean, float* rstd, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, double* scratch /* [32][G][2][C] */,
                      int G, int T, int C, int rows_per_group, float eps, float momentum, int repeat,
                      uint32_t* ticket, void* stream);
/* `ticket` (here, in mmdyn_bn_bwd_finalize and in mmdyn_colsum; nullable): a zero-initialised 32-bit word in device
 * memory that no other launch in flight uses.  With it the two stages run as ONE launch: every block publishes its
 * partial sums (agent-scope release), draws a ticket, and the block that arrives last (agent-scope acquire) adds the
 * partials in the same fixed order as the separate second kernel -- bit-identical results -- and puts the word back to
 * zero (a HIP-graph replay finds it zero again).  Without it: two launches. */
/* a = swish(gamma*(y-mean)*rstd + beta) */
int mmdyn_bn_swish_fwd(const float* y, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, float* a, int G, int rows_per_group, int C, void* stream);
/* backward, two launches: reduce -> sums[G][2][C] (+ dgamma/dbeta accumulated), then apply -> dy */
int mmdyn_bn_swish_bwd_reduce(const float* da, const float* y, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, float* partial, int G,
                              int rows_per_group, int C, void* stream);
int mmdyn_bn_bwd_finalize(const float* partial, float* sums, float* dgamma, float* dbeta,
                          double* scratch /* [32][G][2][C] */, int G, int T, int C, float beta_acc,
                          uint32_t* ticket, void* stream);
/* eval mode (model.eval(): nn.BatchNorm2d with training=False): mean[g][c] = running_mean[c],
 * rstd[g][c] = 1/sqrt(running_var[c] + eps); mmdyn_bn_swish_fwd then applies them.  No buffer update. */
int mmdyn_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* rstd, int G, int C,
                        float eps, void* stream);
/* Synchronised BatchNorm across data-parallel ranks (optional): the per-channel sums are
 * collapsed to double [G][2][C], all-reduced by the host (RCCL), and the statistics / backward coefficients are
 * finished from the GLOBAL sums.  reduce_partials: scratch like mmdyn_bn_finalize.  finalize_sums: n = global rows per
 * group.  bwd_finalize_sums: dgamma/dbeta (nullable) = local parameter gradients (the gradient all-reduce adds the
 * ranks); sums_f (nullable) = float copy of the sums times sums_scale (global sums x n_local/n_global, so that
 * mmdyn_bn_swish_bwd_apply, which divides by the local row count, applies the global means). */
int mmdyn_bn_reduce_partials(const float* partial, double* sums, double* scratch, int G, int T, int C, void* stream);
int mmdyn_bn_finalize_sums(const double* sums, float* mean, float* rstd, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, int G, int C, int n, float eps, float momentum, int repeat,
                           void* stream);
int mmdyn_bn_bwd_finalize_sums(const double* sums, float* sums_f, float* dgamma, float* dbeta, int G, int C,
                               float sums_scale, float beta_acc, void* stream);
/* da_is_du != 0: `da` already holds du = da * swish'(gamma*xhat+beta) (written by mmdyn_igemm_nt_dgrad_bn) */
int mmdyn_bn_swish_bwd_apply(const float* da, const float* y, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const float* sums, float* dy, int G,
                             int rows_per_group, int C, int da_is_du, void* stream);

/* ---- element-wise ---------------------------------------------------------------------------- */
int mmdyn_act_fwd(const float* u, float* h, int64_t n, int act, void* stream);
/* du = dh * act'(u) */
int mmdyn_act_bwd(const float* dh, const float* u, float* du, int64_t n, int act, void* stream);
/* nn.Dropout(p) with injected keep-masks (vae.py:213): out[p][b][:] = h[b][:] * mask[p][b][:] / (1-p_drop)
 * for P passes sharing one trunk output h[B][H];  backward sums the P masked gradients. */
int mmdyn_dropout_expand(const float* h, const uint8_t* masks, float* out, int P, int B, int H,
                         float p_drop, void* stream);
int mmdyn_dropout_reduce(const float* dout, const uint8_t* masks, float* dh, int P, int B, int H,
                         float p_drop, const float* u /* nullable [B][H]: dh *= act'(u), the activation in front of the
                         dropout (vae.py:213-216) */, int act, void* dh_planes /* nullable (revision 6): dh also as a plane tensor,
                         rows of [plane][H] bf16 -- the operand of a plane launch */, void* stream);
/* keep-masks / N(0,1) draws from a counter-based Philox-4x32-10 stream (throughput runs; parity runs inject
 * tensors).  The stream position is offset + *offset_dev (offset_dev may be null): keeping the running position
 * in device memory and bumping it with mmdyn_counter_add makes a captured HIP graph draw fresh numbers on replay. */
int mmdyn_random_masks(uint8_t* masks, int64_t n, float p_drop, uint64_t seed, uint64_t offset,
                       const uint64_t* offset_dev, void* stream);
int mmdyn_random_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                        void* stream);
int mmdyn_counter_add(uint64_t* counter, uint64_t inc, void* stream);
/* out[c] (+)= sum_r x[r][c]   (bias gradients); deterministic two-stage sum, scratch holds
 * mmdyn_colsum_chunks(rows) * C floats; perm 2 = the upsample-bias permutation hw*256+c -> c*25+hw; C % 4 == 0 */
int mmdyn_colsum(const float* x, float* out, float* scratch, int rows, int C, int perm, float beta, uint32_t* ticket,
                 void* stream);
int mmdyn_colsum_chunks(int rows);
/* out = x * s[0], s in device memory (chain rule through a scalar loss term without a host sync) */
int mmdyn_scale_dev(const float* x, const float* s, float* out, int64_t n, void* stream);
/* Gradient buckets on the wire in bf16 (data parallel, 16-bit storage modes; the reference has no
 * counterpart: problems.py:52, 388 train on one device): round the fp32 bucket to bf16 (RNE) in front of the all-reduce,
 * widen it back behind it.  src / dst 16-byte (fp32 side) and 8-byte (bf16 side) aligned. */
int mmdyn_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int mmdyn_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* n <= MMDYN_COPY_MANY_MAX device-to-device copies (src[i] -> dst[i], bytes[i] bytes; host arrays of device pointers) as ONE
 * launch: a batch (inputs + targets of problems.py:148-156) moved into the static buffers of a captured step.  Segments must not
 * overlap.  (The reference hands its batch to the model by reference; there is no counterpart call.) */
#define MMDYN_COPY_MANY_MAX 8
int mmdyn_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int n, void* stream);
/* sum of P row blocks: out[b][:] = sum_p x[p][b][:] */
int mmdyn_sum_blocks(const float* x, float* out, int P, int64_t n, void* stream);
/* tiny Linear layers of the 7-DoF pose MLP (K or N == 7; vae.py:117-123): y = x W^T + b */
int mmdyn_linear_small_fwd(const float* x, const float* W, const float* b, float* y, int rows, int K,
                           int N, int act, void* stream);
int mmdyn_linear_small_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW,
                           float* db, int rows, int K, int N, float beta, void* stream);

/* ---- product of experts + reparametrisation + KL (vae.py:311-318, 52-61; problems.py:429) --- */
#define MMDYN_MAX_PASSES 8
#define MMDYN_MAX_EXPERTS 4
typedef struct {
  const float* mu[MMDYN_MAX_EXPERTS];   /* expert means (visual, tactile, pose, spare); null = absent */
  const float* lv[MMDYN_MAX_EXPERTS];   /* expert log-variances */
  float* dmu[MMDYN_MAX_EXPERTS];        /* backward outputs (same shapes/strides) */
  float* dlv[MMDYN_MAX_EXPERTS];
  int ld[MMDYN_MAX_EXPERTS];            /* row stride of the expert tensors (heads are stored fused: [rows][2L]) */
  const float* dz[3];                   /* backward only: up to three [B][L] latent gradients of this pass (one per
                                           decoder that consumed z), summed by the kernel; null = none */
  float* zdst[3];                       /* forward only (revision 6): up to three further [B][L] destinations of this pass's z --
                                           its row block in the stacked input of each decoder that consumes it (the
                                           torch.cat of the subset passes, problems.py:478-529, without a copy launch);
                                           null = none */
  void* zpl[3];                         /* forward only (revision 6): the same row blocks as PLANE tensors (rows of [plane][L] bf16,
                                           the exact three-term split: mmdyn_split_planes) for a decoder whose first GEMM takes
                                           its operand already split; null = none */
} mmdyn_pass_experts;
/* P passes of [B][L].  with_prior=1 adds the universal N(0,1) expert first (vae.py:139, 321-328).
 * Outputs mu/logvar [P][B][L]; optional z = eps*exp(logvar/2)+mu; optional kl_sum[p] (double)
 * += -0.5*sum(1+lv-mu^2-e^lv).  eps added twice to each variance, as the reference does. */
int mmdyn_poe_fwd(const mmdyn_pass_experts* passes, const float* eps_noise, float* mu, float* logvar,
                  float* z, double* kl_sum, int with_prior, int P, int B, int L, void* stream);
/* upstream gradients: dz [P][B][L] (through z), g_mu / g_lv [P][B][L] (directly on the fused mu/logvar),
 * any may be null; kl_scale = kl_weight / B adds the KL term's gradient. */
int mmdyn_poe_bwd(const mmdyn_pass_experts* passes, const float* eps_noise, const float* mu,
                  const float* logvar, const float* dz, const float* g_mu, const float* g_lv, float kl_scale,
                  int with_prior, int P, int B, int L, const float* kl_weight_dev, void* stream);
/* single-expert path (VAE, vae.py:81-88): reparametrisation and/or KL on mu/lv rows of stride ld */
int mmdyn_reparam_fwd(const float* mu, const float* lv, const float* eps_noise, float* z, double* kl_sum,
                      int B, int L, int ld, void* stream);
int mmdyn_reparam_bwd(const float* mu, const float* lv, const float* eps_noise, const float* dz,
                      float kl_scale, float* dmu, float* dlv, int B, int L, int ld, void* stream);

/* ---- ELBO reconstruction terms (problems.py:433-449) ----------------------------------------- */
/* sum over all elements of BCE-with-logits(x, t) added to *loss_sum (double); optional dlogit =
 * (sigmoid(x) - t) * grad_scale.  With `mask` ([B][mask_channels][H][W]; mask_channels = 1: broadcast over the
 * channels, = C: elementwise, the dataset's 3-channel segmentation mask) both logits and targets are multiplied by it
 * first (problems.py:445-447); any other channel count is MMDYN_ERR_SHAPE. */
int mmdyn_bce_logits(const float* logits, const float* target, const float* mask, float* dlogit,
                     double* loss_sum, int64_t n, int chw, int hw, int mask_channels, float grad_scale, void* stream);
/* The unmasked term for G decoder passes that share one target, in ONE launch: logits / dlogit [G][n], target [n];
 * pass g adds its sum to loss_slots[slot_of_group[g]] (a host array, copied by value into the launch: capturable);
 * a negative slot marks a discarded reconstruction (zero gradient, no loss).  The multi-subset ELBO of
 * problems.py:462-545 compares every subset's reconstruction of a modality with the same target. */
#define MMDYN_BCE_GROUPS_MAX 8
int mmdyn_bce_logits_groups(const float* logits, const float* target, float* dlogit, double* loss_slots,
                            const int* slot_of_group, int G, int64_t n, float grad_scale, void* stream);
/* The same launch with the loss mask of --mask-loss (problems.py:445-447, 684: torch.mul(recon, mask) against
 * torch.mul(target, mask)); mask [B][mask_channels][H][W] as in mmdyn_bce_logits (n = B*chw per pass).  unmasked_slots (may be
 * null) receives the plain sums next to the masked ones: the reference's perf_measure of the single-modality passes is
 * computed without the mask (problems.py:495-505). */
int mmdyn_bce_logits_groups_masked(const float* logits, const float* target, const float* mask, float* dlogit,
                                   double* loss_slots, double* unmasked_slots, const int* slot_of_group, int G, int64_t n,
                                   int chw, int hw, int mask_channels, float grad_scale, void* stream);
/* sum (r-t)^2 added to *loss_sum; dr = 2 (r-t) grad_scale */
int mmdyn_mse(const float* r, const float* t, float* dr, double* loss_sum, int64_t n, float grad_scale,
              void* stream);
/* The same for G passes against ONE target (F.mse_loss of the pose term, problems.py:439-443, for every pose-bearing subset of
 * the multi-subset ELBO): r / dr [G][n], t [n], loss_slots[slot_of_group[g]] += the sum of pass g.  One launch. */
int mmdyn_mse_groups(const float* r, const float* t, float* dr, double* loss_slots, const int* slot_of_group, int G, int64_t n,
                     float grad_scale, void* stream);
/* loss[0] = (sum_p bce[p] + pose_multiplier * mse[p] + kl_weight * kl[p]) / B; partial[p] likewise.
 * kl_weight_dev (here and in mmdyn_poe_bwd; may be null): one float in device memory that multiplies kl_weight /
 * kl_scale -- the annealed KL weight of problems.py:212-216 kept on the device, so that a captured launch serves every
 * epoch of the schedule. */
int mmdyn_elbo_assemble(const double* bce, const double* mse, const double* kl, float* loss, float* partials,
                        int P, int B, float kl_weight, float pose_multiplier, const float* kl_weight_dev, void* stream);

/* ---- Adam (torch.optim.Adam defaults, problems.py:137-138) ----------------------------------- */
/* state: 3 doubles {step count, step size, sqrt(bias_correction2)}, advanced on the device by this call
 * (graph-replay safe); p/g/m/v: flat fp32 buffers of n elements; g is multiplied by grad_scale first */
int mmdyn_adam_step(float* p, const float* g, float* m, float* v, double* state, int64_t n, float lr,
                    float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The same step behind an overflow guard (the fp16 precision modes: loss-scaled gradients can overflow fp16 on their way
 * through the matrix cores).  state: SIX doubles {step count, step size, sqrt(bias_correction2), flag, skipped steps, skip};
 * a gradient buffer that holds an inf or NaN leaves parameters, moments and step count untouched and adds one to state[4]
 * -- the behaviour of torch.cuda.amp.GradScaler.step on such a step (the reference itself trains in fp32: problems.py:150-155). */
int mmdyn_adam_step_guarded(float* p, const float* g, float* m, float* v, double* state, int64_t n, float lr,
                            float beta1, float beta2, float eps, float grad_scale, void* stream);

/* torch.optim.SGD with momentum / weight decay as configured by problems.py:132-136 (momentum 0.9, wd 5e-4);
 * first != 0 on the first step (momentum buffer := gradient, like torch) */
int mmdyn_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                   float weight_decay, float grad_scale, int first, void* stream);

/* ---- image decode in front of the path --------------------------------------------------------
 * transforms.Resize(input_size) + transforms.ToTensor() of datasets.py:23-31 / 375-385 on frames kept as uint8
 * HWC in HBM: Pillow's 8-bit anti-aliased bilinear resampler (libImaging/Resample.c: 22-bit fixed-point taps,
 * horizontal then vertical pass, each rounded to uint8) followed by /255 -> float32 CHW.  Bit-exact.
 *   mmdyn_resize_ksize : taps per output sample for one axis (host only)
 *   mmdyn_resize_plan  : bounds[out][2] = (first tap, tap count), coeffs[out][ksize]; returns ksize (host only)
 *   mmdyn_resize_u8_to_chw_f32 : src uint8 [n][Hin][Win][3]; index (nullable) int32 [n_out] frame gather;
 *                        dst float32 [n_out][3][Hout][Wout]; xb/xk, yb/yk: the DEVICE copies of the two plans */
int mmdyn_resize_ksize(int in_size, int out_size);
int mmdyn_resize_plan(int in_size, int out_size, int* bounds, int* coeffs);
int mmdyn_resize_u8_to_chw_f32(const uint8_t* src, const int* index, float* dst, int n_out, int Hin, int Win,
                               int Hout, int Wout, const int* xb, const int* xk, const int* yb, const int* yk,
                               void* stream);

/* ---- bf16 activation storage (BASELINE configs[2]: "bf16 storage, fp32 accumulate / master weights") -----------
 * The activations between the convolution layers (pre-BatchNorm outputs, post-Swish activations and their
 * gradients) are bf16 in HBM (RNE on store, exact widening on load); weights, BatchNorm statistics and parameters,
 * the FC-level tensors, logits, losses and optimiser state stay fp32; all arithmetic is fp32 in registers except the
 * bf16 matrix-core products (fp32 accumulate).
 *   mmdyn_igemm_nt_mx : mmdyn_igemm_nt / _dgrad_bn with per-tensor storage flags --
 *       bit 0 bf16 matrix cores (required when any other bit is set), bit 1 A is bf16 (not IM2COL3), bit 2 C and
 *       C_act are bf16, bit 3 the BatchNorm-backward operand y is bf16, bit 4 the packed weights Bp are bf16
 *       (mmdyn_pack_conv_weight_b16 / mmdyn_repack2d_ld_b16 / a dst_bf16 plan entry), bit 6 C_act alone is bf16 (C stays
 *       fp32: the Linear layer whose activated output feeds the first transposed convolution, vae.py:264-271; not with
 *       bit 2 or split-K).  y == NULL: plain GEMM epilogue.  Split-K (fp32 workspace) is allowed with a bf16 A, not with a
 *       bf16 C.  Bit 5 together with bit 0: every tensor the other bits mark is IEEE HALF instead of bf16 and the product
 *       runs on the fp16 matrix cores (precision "fp16s": fp16 activation storage, BASELINE configs[4] arithmetic); bit 5
 *       with no storage bit is mmdyn_igemm_nt_f16.
 *       Bit 7 ALONE (flags == 128; also accepted by mmdyn_igemm_nt_dgrad_act / _grouped, mmdyn_wgrad_tn_mx / _grouped and the
 *       *_stat_tiles_mx / *_slab_floats_mx queries): an fp32 launch -- fp32 operands, fp32 results, every tensor fp32 in HBM --
 *       that MAY run on the bf16 matrix cores through the exact three-term split of its fp32 operands (precision "fp32x3" of the
 *       host side): x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (round-to-nearest, both
 *       subtractions exact), six of the nine cross products (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) on
 *       v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped products are below 2^-23 |a||b| together, less than one fp32
 *       rounding of the product, and the error against fp64 is no larger than the fp32 matrix cores' (csrc/common.h split3_bf16,
 *       csrc/igemm_nt.hip / wgrad_tn.hip X3; profiles/r4/ab_x3_*.txt).  The library decides per launch (shapes with >= 512 blocks
 *       of 64x64 or >= 384 of 128x128 outputs; every weight-gradient GEMM); the rest of such a step runs the fp32 matrix cores.
 *       Operand range (tests/test_kernels_aten_gpu.py::test_x3_operand_magnitudes_*, ::test_x3_non_finite_*): full fp32 accuracy
 *       for 2^-100 <= |x| < 0x1.ffp+127 (bit pattern 0x7F7F8000, ~3.39e38: from there to FLT_MAX bf16's round-to-nearest takes the
 *       high term to Inf and the split gives NaN, like an Inf operand).  Below 2^-100 the lower terms of the split leave bf16's normal range and are flushed by the matrix
 *       pipe: an operand under ~1e-33 keeps 16 significant bits, one under ~3e-36 keeps 8 -- an absolute error of at most
 *       2^-126 |w| per product.  An Inf operand gives NaN (inf - inf in the split) where the fp32 matrix cores would give Inf; a
 *       NaN operand gives NaN; no other row of the result is touched.  The library does not guard: non-finite operands do not occur
 *       on this path (the fp16 modes' overflow guard, mmdyn_adam_step_guarded, is not needed in fp32 storage).
 *       Bit 10 (with bits 7 + 8, DENSE launches; round 6): C_act is written as a plane tensor -- the activated output of a Linear
 *       layer handed to the next plane launch without a stand-alone split; C stays fp32, ldc == N.  Bits 16-27 = c / 8: the plane
 *       rows hold c channels, c dividing N (an output row is N / c consecutive plane rows: the decoder's Linear output
 *       [B][hw*256 + ch] read as [B*25][256] by the transposed convolution above it, vae.py:264-271); 0: rows of [plane][N].
 *       Bits 7 + 8 (flags == 384; mmdyn_igemm_nt_mx, mmdyn_igemm_nt_dgrad_act; mmdyn_igemm_nt_dgrad_bn(bf16 = 4)): the same
 *       arithmetic on operands that ARRIVE SPLIT -- A and Bp are rows of [plane][Cin] bf16 (hi | mid | lo, 6 bytes per element:
 *       mmdyn_split_planes, or written so by their producers) -- so the GEMM itself contains no split: LDS-DMA of the planes, six
 *       v_mfma_f32_16x16x32_bf16 per fragment pair (csrc/igemm_wsp.hip, igemm_wsp3_kernel; the 32-channel up-sampling layers:
 *       csrc/tconv_patch.hip, tconv_patch_kernel<..., P3>).  Same terms, same product order per
 *       K-step as the flags == 128 launch of the shape (results agree to the last bits; the channel -> k-lane map inside the 32-deep
 *       MFMA differs, so not bit for bit).  Only for shapes mmdyn_igemm_planes_served answers 1 for
 *       (MMDYN_ERR_SHAPE otherwise); partial-sum tile count and workspace: the *_stat_tiles_mx / *_slab_floats_mx queries with flags == 384.
 *   mmdyn_wgrad_tn_mx : bit 0 as above, bit 1 D is 16-bit, bit 2 Gt is 16-bit (not IM2COL3), bit 5 as above, bit 7 alone as above;
 *       with bit 7, bit 8: D arrives split, bit 9: Gt arrives split (plane tensors, rows of [plane][C] bf16; MMDYN_CONV only) -- that
 *       operand goes from HBM to the LDS planes as it is, the other is split in the kernel as with bit 7 alone; both bits: the
 *       plane-ring weight-gradient kernel where it serves the channel counts (csrc/wgrad_p3.hip; chunks from mmdyn_wgrad_chunks_mx
 *       with the same flags).
 *   *_b16             : the element-wise kernels on 16-bit activation tensors (half = 0: bf16, half = 1: IEEE half). */
int mmdyn_igemm_nt_mx(const void* A, const void* Bp, const float* bias, void* C, void* C_act, float* stats,
                      float* ws, const void* y, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N,
                      int ldc, int stride, int offset, int act, int splitk, int flags, uint32_t* arrival_flags, void* stream);
int mmdyn_wgrad_tn_mx(const void* D, const void* Gt, float* partial, int mode, int Bt, int Hr, int Wr, int Cd, int Hi,
                      int Wi, int Cg, int stride, int offset, int chunks, int flags, void* stream);
/* The exact three-term bf16 split of an fp32 matrix, stored for the plane launches above: x [rows][C] fp32 ->
 * planes [rows][3][C] bf16 with planes[r][0] = hi = bf16(x), [1] = mid = bf16(x - hi), [2] = lo = x - hi - mid (round-to-nearest-
 * even; hi + mid + lo == x bit for bit for finite x).  C % 8 == 0.  A channels-last activation tensor is such a matrix with one
 * row per pixel; packed weights [taps][N][Cin] are one with taps * N rows.  (Replaces nothing in the reference: it is the storage
 * format of the fp32x3 arithmetic's GEMM operands, the role nn.Conv2d's fp32 input tensor plays in vae.py:198-216, 264-277.) */
int mmdyn_split_planes(const float* x, void* planes, int64_t rows, int C, void* stream);
int mmdyn_igemm_planes_served(int mode, int G, int Bg, int Hi, int Wi, int Cin, int Ho, int Wo, int N);
/* mmdyn_bn_swish_fwd / mmdyn_bn_swish_bwd_apply (train-mode nn.BatchNorm2d + Swish forward, vae.py:200-208, 268-276, and the apply
 * pass of their backward) with the result ALSO (a / dy non-null) or ONLY (a / dy NULL) written as a plane tensor -- the split
 * rides on a pass that exists anyway, so the GEMMs that consume the tensor find their operand already split. */
int mmdyn_bn_swish_fwd_planes(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                              float* a, void* a_planes, int G, int rows_per_group, int C, void* stream);
int mmdyn_bn_swish_bwd_apply_planes(const float* da, const float* y, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, const float* sums, float* dy, void* dy_planes, int G, int rows_per_group,
                                    int C, int da_is_du, void* stream);
int mmdyn_bn_swish_fwd_b16(const uint16_t* y, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, uint16_t* a, int G, int rows_per_group, int C, int half, void* stream);
int mmdyn_bn_swish_bwd_reduce_b16(const uint16_t* da, const uint16_t* y, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float* partial, int G, int rows_per_group,
                                  int C, int half, void* stream);
int mmdyn_bn_swish_bwd_apply_b16(const uint16_t* da, const uint16_t* y, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, const float* sums, uint16_t* dy, int G,
                                 int rows_per_group, int C, int da_is_du, int half, void* stream);
int mmdyn_act_bwd_b16(const uint16_t* dh, const uint16_t* u, uint16_t* du, int64_t n, int act, int half, void* stream);
int mmdyn_tconv_out3_fwd_b16(const uint16_t* a, const float* w, float* out, int Bt, int Hi, int Wi, int half,
                             void* stream);

/* ---- misc ------------------------------------------------------------------------------------- */
int mmdyn_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, void* stream);
int mmdyn_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif
