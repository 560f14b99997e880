/*
 * downgan_hip.h — C ABI of libdowngan_hip.so: the MI355X (gfx950) kernels of the DoWnGAN WGAN-GP
 * train step.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (nannau/DoWnGAN) has no FFI of its own: its hot path is PyTorch ops called from
 * DoWnGAN/GAN/wasserstein.py and DoWnGAN/networks/{generator,critic}.py.  Each entry point below
 * names the reference op / call site it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - Activations are NHWC ("channels-last"), element type `dtype` (DG_F32 or DG_BF16), channel
 *    counts padded to a multiple of 16, `ld*` = distance between consecutive pixels in ELEMENTS
 *    (so a tensor may be a channel slice of a wider dense-block slab).  All base pointers and
 *    channel offsets are 16-byte aligned.
 *  - Conv weights are packed [Nout][9 taps][Cred] in `dtype` (see dg_repack_conv_weights): the
 *    forward pack has Nout=Cout, Cred=Cin (KRSC); the dgrad pack has Nout=Cin, Cred=Cout.
 *  - Master parameters, gradients, Adam moments, biases and all loss scalars are fp32.
 *  - Ownership: the caller owns every buffer (incl. workspaces); nothing here allocates or frees
 *    device memory.  Launches are asynchronous on `stream` (a hipStream_t passed as void*).
 *  - Errors: 0 = DG_OK, negative = dg_status; no exceptions cross the ABI.
 *  - Re-entrant.  The only mutable state is a per-kernel, per-device "function attributes configured" mask
 *    (hipFuncAttributeMaxDynamicSharedMemorySize is set once per device) and cached occupancy answers, both
 *    std::atomic; the device that is current when an entry point is called must be the device of `stream`.
 */
#ifndef DOWNGAN_HIP_H
#define DOWNGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_F32 0
#define DG_BF16 1

typedef enum dg_status {
  DG_OK = 0,
  DG_ERR_BAD_SHAPE = -1,
  DG_ERR_BAD_DTYPE = -2,
  DG_ERR_BAD_ARG = -3,
  DG_ERR_LAUNCH = -4
} dg_status;

/* Epilogue applied to every output element v (fp32 accumulator) before the store, in this order:
 *   v += bias[c]; if (has_act) v = v>0 ? v : v*act_slope;
 *   if (r1) v = v*s1 + r1[pixel,c];  if (r2) v = v*s2 + r2[pixel,c];
 *   if (mask) v *= (mask[pixel,c] > 0 ? 1 : mask_slope);     -- LeakyReLU'(saved activation)
 *   if (accumulate) v += y[pixel,c];
 * r1/r2/mask are tensors of the OUTPUT's shape in `dtype`.
 * mask_c0 / mask_last (both 0 = the order above, every channel): the mask applies to channels c >= mask_c0 only
 * (multiple of 16), and with mask_last != 0 it multiplies AFTER the accumulate -- the data gradient of a dense block's
 * conv k completes channel slice k-1 of the block's gradient slab and applies that slice's LeakyReLU' in the same pass. */
typedef struct dg_epilogue {
  const float* bias;
  int has_act;
  float act_slope;
  const void* r1; int64_t ldr1; float s1;
  const void* r2; int64_t ldr2; float s2;
  const void* mask; int64_t ldmask; float mask_slope;
  int accumulate;
  /* 1-bit form of the LeakyReLU' mask (16x fewer bytes than re-reading the activation; the critic's data-gradient
   * epilogues are HBM-bound on it): a tensor [pixel][Cout/64][4] of uint16, bit b of word (block, g) belonging to channel
   * 64*block + 16g + b -- a little-endian bit string over the channels of a pixel.  mask_bits: multiply by (bit ? 1 : mask_slope) (instead of `mask`); out_bits: written by this
   * launch as (stored value > 0).  Both need Cout % 64 == 0, Cout >= 128 and no pixel shuffle. */
  const void* mask_bits;
  void* out_bits;
  int mask_c0;
  int mask_last;
  /* MXFP8 copy of the stored output for the fp8 conv path (see dg_quant_mxfp8 below): out_q [pixel][ld of y] E4M3 bytes
   * (same pixel stride as y, counted in bytes) and out_qs [pixel][Cout/32] E8M0 scale bytes, bit-identical to
   * dg_quant_mxfp8 of the bf16 tensor this launch stores -- the next layer's fp8 conv reads them instead of a separate
   * quantisation pass.  Both or neither; bf16 launches with Cout % 64 == 0, Cout >= 128, no pixel shuffle.
   * ldqs: scale bytes per pixel of out_qs (0 = Cout/32, dense); a larger stride addresses a channel slice of a wider tensor
   * (dense-block slab: y, out_q and out_qs all point at the slice's first channel) and needs Cout/16 to be a power of two. */
  void* out_q;
  void* out_qs;
  int64_t ldqs;
  /* Second fp8 copy of the stored output for the fp8 WEIGHT GRADIENT (dg_conv3x3_wgrad_f8), whose contraction runs over pixels:
   * out_u [pixel][ld of y] E4M3 bytes = stored value / 2^(out_ue[channel / 32] - 127), saturated at +-448, with ONE exponent
   * byte per 32-channel block for the whole tensor (out_ue: Cout / 32 bytes of device memory, read by the launch; the caller
   * derives them from an earlier pass, dg_block_exp_max).  Needs out_q (the copy is formed from the same rounded values). */
  void* out_u;
  const void* out_ue;
  /* != 0: the output tensor y itself is NOT stored -- only its fp8 copies (out_q or out_u required) / mask bits are, for launches whose
   * bf16 result nobody reads (fp8 mode: the next conv reads out_q, the weight gradient out_u, the masks out_bits).  Not with accumulate. */
  int skip_y;
  /* First-layer launches (<= 2 real input channels, critic.py:21-24: an 8.6-GB output at configs[1], store-bound) may write out_u
   * WITHOUT out_q -- the next layer's MXFP8 conv then reads the uniform-scale copy with out_ue as ONE scale row for every pixel
   * (dg_f8_operands.ldxs < 0): half the bytes -- and keep the census the exponents of the next pass come from themselves:
   * out_amax [Cout / 32] uint32 (device, zeroed by the caller / by dg_exp_from_amax), atomically maxed with the bit pattern of the
   * largest |stored value| of each 32-channel block.  Other launches return DG_ERR_BAD_SHAPE for either. */
  void* out_amax;
} dg_epilogue;

/* Geometry of ONE reference nn.Conv2d(Cin, Cout, kernel_size=3, stride, padding=1) layer
 * (generator.py:24,62,66,70,77,80; critic.py:21-87), in its FORWARD orientation. */
typedef struct dg_conv_geom {
  int dtype;
  int N, H, W;          /* forward input: N images of H x W                                     */
  int Cin, Cout;        /* padded channel counts (multiples of 16; Cin may be a multiple of 8)   */
  int stride;           /* 1 or 2; output is Ho = H/stride, Wo = W/stride (H, W even if 2)       */
  int cin_real;         /* number of REAL (non-padding) input channels, 0 = unknown/all. Layers   */
                        /*    with <= 2 real channels (critic features.0, critic.py:21) take an    */
                        /*    im2col path that reads the big tensor once for all 9 taps.           */
  int pixel_shuffle;    /* 1: the layer is followed by LeakyReLU + nn.PixelShuffle(2)            */
                        /*    (generator.py:69-75). Output channels are packed (2i+j)*Cout/4 + c */
                        /*    and the activation tensor is stored shuffled: [N,2Ho,2Wo,Cout/4].  */
  int64_t ldx;          /* pixel stride of the layer INPUT tensor x  (or of dx in dgrad)         */
  int64_t ldy;          /* pixel stride of the layer OUTPUT tensor y (or of dy in dgrad/wgrad)   */
} dg_conv_geom;

/* The generic launch descriptor every conv forward / data-gradient is lowered to: a gather-GEMM
 *   Y[dst(m), n] = epilogue( sum_{t<ntaps} sum_{c<Cred} X[src(m,t), c] * Wp[n][tap_w[t]][c] )
 * over GEMM rows m = (img, gy, gx) of an Hg x Wg grid, src(m,t) = (img, gy*sy_mul+tap_dy[t],
 * gx*sx_mul+tap_dx[t]) (zero outside the source), dst(m) = (img, gy*dy_mul+dy_off, gx*dx_mul+dx_off).
 * Exposed so that the planner can be checked on a CPU (dg_conv3x3_plan needs no GPU). */
typedef struct dg_gg_desc {
  int dtype;
  int N, Hs, Ws, Cred; int64_t lds;   /* source tensor                                          */
  int src_ps;                          /* source is stored pixel-shuffled (Cred = 4*Cps)         */
  int Hg, Wg, sy_mul, sx_mul;
  int ntaps; int tap_dy[9], tap_dx[9], tap_w[9];
  int Nout; int64_t ldw;               /* packed weights: row stride in elements (9*Cred)        */
  int Hd, Wd; int64_t ldd;             /* destination tensor                                     */
  int dy_mul, dx_mul, dy_off, dx_off;
  int dst_ps;                          /* store pixel-shuffled (Nout = 4*Cps)                    */
} dg_gg_desc;

const char* dg_version(void);

/* ---- convolution family (replaces torch.nn.Conv2d forward/backward on the hot path) ---------- */

/* y = epilogue(conv3x3(x, w_fwd)).  Replaces nn.Conv2d.forward (+ fused LeakyReLU / residual /
 * PixelShuffle): generator.py:36-41,53,62-90; critic.py:101-102. */
int dg_conv3x3_fwd(const dg_conv_geom* g, const dg_epilogue* ep, const void* x, const void* w_fwd,
                   void* y, void* stream);

/* dx = epilogue(conv3x3_input_grad(dy, w_dgrad)).  Replaces the autograd input-gradient of
 * nn.Conv2d (wasserstein.py:52,80 backward; :100-106 autograd.grad for the penalty).  Stride 2 is
 * done as 4 output-parity classes (no zero insertion).  Epilogue tensors have dx's shape. */
int dg_conv3x3_dgrad(const dg_conv_geom* g, const dg_epilogue* ep, const void* dy,
                     const void* w_dgrad, void* dx, void* stream);

/* dw[Cout][9][Cin] (fp32) += sum_pixels dy (x) x   (atomic accumulate).  Replaces the autograd
 * weight-gradient of nn.Conv2d (wasserstein.py:52,80) and, with (x:=tangent, dy:=adjoint), the
 * double-backward term of the gradient penalty (wasserstein.py:100-117 under :52). */
int dg_conv3x3_wgrad(const dg_conv_geom* g, const void* x, const void* dy, float* dw, float* db,
                     void* stream);
  /* db (optional): bias gradient += sum_pixels dy */

/* Weight and bias gradients of ALL convs of a dense block (generator.py:14-41) in one launch.  Conv k = 1..nconv reads channels
 * [0, k*128) of the block's activation slab x [N,H,W,nconv*128] and its output adjoint is channels [(k-1)*128, k*128) of the
 * adjoint slab dy; dw[k-1] ([128][9][k*128] fp32) and db[k-1] ([128] fp32, db or any entry may be NULL) are accumulated into.
 * g: Cin = Cout = nconv*128, stride 1, ldx / ldy = pixel strides of the two slabs; bf16; W % 32 == 0. */
int dg_conv3x3_wgrad_dense(const dg_conv_geom* g, int nconv, const void* x, const void* dy,
                           float* const* dw, float* const* db, void* stream);

/* fp8 weight gradient (BASELINE.json configs[4]; autograd of DoWnGAN/networks/critic.py:34-88 under GAN/wasserstein.py:52):
 * dw[co][tap][ci] (fp32, accumulated into) += sum_p dy[p, co] * x[src(p, tap), ci] with E4M3 operands whose scale does not vary
 * along pixels -- the contraction index -- : x = xq * 2^(ex[ci / 32] - 127), dy = dyq * 2^(ey[co / 32] - 127), one E8M0 exponent
 * byte per 32-channel block of the whole tensor (device arrays of Cin / 32 and Cout / 32 bytes).  g->ldx / g->ldy are the pixel
 * strides of xq / dyq in bytes (= elements); g->dtype = DG_BF16 (the precision the fp8 forms stand for).  Stride 1, Cin and Cout
 * multiples of 128, W a multiple of 64, no pixel shuffle; anything else returns DG_ERR_BAD_SHAPE (callers keep the bf16 kernel). */
int dg_conv3x3_wgrad_f8(const dg_conv_geom* g, const void* xq, const void* ex, const void* dyq, const void* ey, float* dw,
                        void* stream);

/* dg_conv3x3_wgrad_dense on the fp8 kernel (autograd of DoWnGAN/networks/generator.py:24-41 under GAN/wasserstein.py:80 in fp8 mode):
 * xq / dyq = uniform-scale E4M3 copies of the block's activation and adjoint slabs [N,H,W,nconv*128] (pixel strides g->ldx / g->ldy
 * in bytes), ex / ey = their nconv*4 block exponents (as dg_conv3x3_wgrad_f8); dw[k-1] ([128][9][k*128] fp32) accumulated into.
 * Weight gradients only: the bias gradients are column sums of the adjoint slab (dg_colsum).  Stride 1, W a multiple of 64. */
int dg_conv3x3_wgrad_dense_f8(const dg_conv_geom* g, int nconv, const void* xq, const void* ex, const void* dyq, const void* ey,
                              float* const* dw, void* stream);

/* db[c] (fp32) += sum over rows of dy[row, c]  (bias gradient of a conv or Linear).  Row r is at
 * element offset (r / rows_inner)*ld_outer + (r % rows_inner)*ld, so one sub-position of a
 * pixel-shuffled tensor can be reduced (rows_outer x rows_inner rows in total). */
int dg_colsum(int dtype, const void* dy, int64_t rows_outer, int64_t ld_outer, int64_t rows_inner,
              int64_t ld, int C, float* db, void* stream);

/* One pass over a wide tensor, nseg (<= 8) destinations: db[k][c] (fp32) += sum over rows of dy[row, k*(C/nseg) + c]; rows of `ld`
 * elements.  The five bias gradients of a dense block (generator.py:24-41) from its adjoint slab when the weight gradients run on
 * dg_conv3x3_wgrad_dense_f8 (the bf16 dense launch sums them itself). */
int dg_colsum_multi(int dtype, const void* dy, int64_t rows, int64_t ld, int C, int nseg, float* const* db, void* stream);

/* Launches one gather-GEMM descriptor (what dg_conv3x3_fwd / _dgrad call after planning). */
int dg_gather_gemm(const dg_gg_desc* d, const dg_epilogue* ep, const void* x, const void* w, void* y,
                   void* stream);

/* Which kernel variants the calling thread's last dg_conv3x3_fwd / _dgrad launched (bit mask: 1 generic
 * gather-GEMM, 2 fast path, 8 halo-patch kernel, 16 im2col small-Cin kernel). Diagnostic. */
int dg_last_conv_kernels(void);

/* Host-only planner (no GPU needed): lowers a layer to its gather-GEMM descriptor(s).
 * kind 0 = forward (1 desc), kind 1 = dgrad (1 desc for stride 1, 4 parity classes for stride 2).
 * Returns the number of descriptors written to out[0..3], or a negative dg_status. */
int dg_conv3x3_plan(const dg_conv_geom* g, int kind, dg_gg_desc* out);

/* Host-only: how dg_conv3x3_dgrad launches geometry g -- 1 = the four parity classes of a stride-2 layer merged into one launch
 * of the halo kernel (or a stride-1 layer's single class), 4 = one launch per class (narrow layers, or rows too long for the halo
 * kernel's 24-bit row step), negative = dg_status. */
int dg_conv3x3_dgrad_launches(const dg_conv_geom* g);

/* Derives the compute-precision weight packs from the fp32 MASTER, which is kept forward-packed and
 * padded as [CoutP][9][CinP] (tap = r*3+s; for a pixel-shuffle layer the host has already moved
 * output channel co=4c+2i+j to row (2i+j)*Cout/4 + c, torch PixelShuffle order, generator.py:73):
 *   kind 0 (forward pack): same layout cast to `dtype`;  kind 1 (dgrad pack): dst[ci][tap][co]. */
int dg_repack_conv_weights(int dtype, int kind, const float* master, void* dst, int CoutP, int CinP,
                           void* stream);
/* Data-gradient packs of a dense block's stacked ("virtual") convs: masters[k-1] = conv k's fp32 weight [F][9][k*F], k = 1..nconv
 * (generator.py:14-41).  dst = for j = 0..nconv-1 the kind-1 pack of the conv that takes the adjoints of convs j+1..nconv
 * ((nconv-j)*F channels) to the adjoint of slab slice j (F channels): dst_j[ci][tap][(k-j-1)*F + co] = W_k[co][tap][j*F + ci];
 * the packs concatenated, 9*F*F*nconv*(nconv+1)/2 elements of `dtype`. */
int dg_repack_dense_dgrad(int dtype, const float* const* masters, int nconv, int F, void* dst, void* stream);
/* Two helpers for layers with <= 2 real OUTPUT channels (generator conv3.2, generator.py:80), whose backward is HBM-bound:
 * dg_repack_conv_weights kind 2 = kind 1 with mirrored taps, the pack with which that layer's data gradient is a forward conv of
 * dy over its 2 real channels (im2col kernel); dg_wgrad_unswap: dw[co][t][ci] += tmp[ci][8-t][co], which folds a weight-gradient
 * launch with swapped operand roles (x := dy with 2 real channels, dy := x) back into the layer's gradient layout. */
int dg_wgrad_unswap(const float* tmp, float* dw, int CoutP, int CinP, void* stream);

/* ---- Linear family (replaces nn.Linear in critic.py:94-105) ---------------------------------- */

/* y[b][o] (fp32, pre-zeroed, ldy) += sum_k x[b][k] * w[o][k]; split-K with atomics. */
int dg_linear_fwd(int dtype, const void* x, int64_t ldx, const void* w, int64_t ldw, float* y,
                  int ldy, int B, int O, int64_t K, void* stream);
/* dx[b][k] = (sum_o dy[b][o] * w[o][k]) * LeakyReLU'(mask[b][k]);  dy fp32 [B][ldo]; dx in
 * out_dtype (DG_F32 or `dtype`); mask (optional) in `dtype`, same shape as dx. */
int dg_linear_dx(int dtype, int out_dtype, const float* dy, int ldo, const void* w, int64_t ldw,
                 void* dx, int64_t lddx, const void* mask, int64_t ldmask, float mask_slope, int B,
                 int O, int64_t K, void* stream);
/* dw[o][k] (fp32) += sum_b dy[b][o] * x[b][k]. */
int dg_linear_dw(int dtype, const float* dy, int ldo, const void* x, int64_t ldx, float* dw,
                 int64_t lddw, int B, int O, int64_t K, void* stream);
/* The same product for up to 128 rows and O <= 112 in ONE sweep over dw: the rows of every pass that contributes to
 * critic_loss.backward() (wasserstein.py:52: real, fake, penalty tangent) concatenated, so the 1.9 GB FC1 gradient
 * (critic.py:94-96) is written once per iteration instead of read-modify-written once per pass.
 * accumulate != 0: dw += ...; accumulate == 0: dw = ... (no zero-fill needed).  K % 4 == 0. */
int dg_linear_dw_wide(int dtype, const float* dy, int ldo, const void* x, int64_t ldx, float* dw,
                      int64_t lddw, int B, int O, int64_t K, int accumulate, void* stream);

/* ---- elementwise / reductions ----------------------------------------------------------------- */

/* out[r][c] = act(in_f32[r][c] + bias[c]) cast to out_dtype; optional mask multiply instead of act
 * (tangent pass).  Rows x C small matrices (critic head, critic.py:96-98). */
int dg_bias_act(int out_dtype, const float* in, int ldi, const float* bias, void* out, int ldo,
                int rows, int C, int has_act, float slope, const void* mask, int ldmask,
                float mask_slope, void* stream);
/* u[p][c] *= LeakyReLU'(y[p][c]) in place over a [rows x C] channel slice. */
int dg_mask_mul(int dtype, void* u, int64_t ldu, const void* y, int64_t ldy, int64_t rows, int C,
                float slope, void* stream);
/* out = a*x + b*y over [rows x C] slices (residual adds, generator.py:41,53,87). y may be NULL. */
int dg_axpby(int dtype, void* out, int64_t ldo, const void* x, int64_t ldx, float a, const void* y,
             int64_t ldy, float b, int64_t rows, int C, void* stream);
/* xhat[b] = alpha[b]*real[b] + (1-alpha[b])*fake[b]   (wasserstein.py:94). per_img = elements/image */
int dg_gp_interp(int dtype, const void* real, const void* fake, const float* alpha, void* xhat,
                 int B, int64_t per_img, void* stream);
/* The same interpolate for fields stored `ld` channels wide of which the first TWO are real: writes the COMPACT [pixel][2]
 * forms the critic's first layer reads fastest (dg_conv3x3_fwd / _wgrad with cin_real <= 2 take either layout) -- xhat_c, and
 * (optional, may be NULL) compact copies real_c / fake_c of the two inputs.  pix_per_img % 4 == 0. */
int dg_gp_interp_c2(int dtype, const void* real, const void* fake, int64_t ld, const float* alpha,
                    void* xhat_c, void* real_c, void* fake_c, int B, int64_t pix_per_img, void* stream);
/* out_c[b][pixel][0..1] = coef[b] * g[b][pixel][0..1]  (the penalty's v0 = dGP/dg, wasserstein.py:110-117 backward), g stored
 * `ld` channels wide, out_c compact. */
int dg_scale_rows_c2(int dtype, const void* g, int64_t ld, const float* coef, void* out_c, int B,
                     int64_t pix_per_img, void* stream);
/* ss[b] (fp32, pre-zeroed) += sum of squares of image b   (wasserstein.py:114), wave-shuffle reduce */
int dg_sumsq_rows(int dtype, const void* g, int B, int64_t per_img, float* ss, void* stream);
/* From ss[b]: n_b = sqrt(ss+1e-12); scalars[0] = gp_lambda*mean((n_b-1)^2)  (wasserstein.py:117);
 * coef[b] = weight * gp_lambda * (2/B_global) * (n_b-1)/n_b  (d/dg of weight*gp_ret). */
int dg_gp_finish(const float* ss, int B, int B_global, float gp_lambda, float weight, float* coef,
                 float* scalars, void* stream);
/* out[b] = coef[b] * g[b]  (per-image scale) */
int dg_scale_rows(int dtype, const void* g, const float* coef, void* out, int B, int64_t per_img,
                  void* stream);
/* L1 content loss (losses.py:51-53): acc[0] (fp32, pre-zeroed) += sum |a-b| over [rows x C] with C
 * real channels of ld-strided pixels; if grad: grad = grad_scale*sign(a-b) (+ addend if given). */
int dg_l1(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int C,
          float* acc, void* grad, int64_t ldg, float grad_scale, const void* addend, int64_t ldadd,
          void* stream);
/* acc[0] (fp32, pre-zeroed) += sum (a-b)^2 over [rows x C]  (content_MSELoss metric, losses.py:58-70; metrics pass
 * mlflow_epoch.py:53-63) */
int dg_sqdiff(int dtype, const void* a, int64_t lda, const void* b, int64_t ldb, int64_t rows, int C,
              float* acc, void* stream);
/* out[0] = scale * sum_{i<n} in[i*stride]  (means of critic outputs, wasserstein.py:46-47,74) */
int dg_sum_strided(const float* in, int n, int stride, float scale, float* out, void* stream);
/* fill fp32 buffer with a constant on a strided column (grad_outputs=ones, wasserstein.py:103) */
int dg_fill_col(float* buf, int rows, int ld, int col, float value, void* stream);

/* Adam (stage.py:63-64: lr 2.5e-4, betas (0.9,0.99), eps 1e-8, no weight decay) over a flat fp32
 * buffer; also refreshes the bf16 shadow copy when shadow != NULL.  grad_scale multiplies g first
 * (1/world_size after a sum all-reduce). */
int dg_adam(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
            float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* Layout converters between the reference's NCHW fp32 tensors and native NHWC padded `dtype`. */
int dg_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, int H, int W, int Cpad,
                    void* stream);
int dg_nhwc_to_nchw(int dtype, const void* src, int64_t lds, float* dst, int N, int C, int H, int W,
                    void* stream);
/* fp32 -> dtype cast of n elements (weight shadows) */
int dg_cast(int dtype, const float* src, void* dst, int64_t n, void* stream);

/* ---- MS-SSIM of the per-step metrics pass.  Replaces `SSIM_Loss(x, y, device)`
 * (DoWnGAN/GAN/losses.py:12-38, called per batch from mlflow_tools/mlflow_epoch.py:53-63 at wasserstein.py:140), which
 * min-max normalises each channel over the batch and evaluates `pytorch_msssim.MS_SSIM(win_size=7, data_range=1,
 * channel=2)` (third-party, unpinned: the kernels follow its published algorithm; see oracle/msssim.py). */
#define DG_SSIM_MAX_CH 8
#define DG_SSIM_MAX_WIN 11
#define DG_SSIM_MAX_LEVELS 5
#define DG_MINMAX_PARTS 256
typedef struct dg_ssim_params {
  int win;                      /* Gaussian window length (reference: 7) */
  float g[DG_SSIM_MAX_WIN];     /* normalised 1-D Gaussian, sigma 1.5 (pytorch_msssim _fspecial_gauss_1d) */
  float C1, C2;                 /* (K1*data_range)^2, (K2*data_range)^2 with K = (0.01, 0.03) */
} dg_ssim_params;
typedef struct dg_msssim_combine {
  float inv_count[DG_SSIM_MAX_LEVELS];  /* 1 / ((H_l-win+1)*(W_l-win+1)) */
  float weight[DG_SSIM_MAX_LEVELS];     /* 0.0448, 0.2856, 0.3001, 0.2363, 0.1333 */
} dg_msssim_combine;
/* per-channel min / max over `pixels` ld-strided pixels (first C channels): partial[DG_MINMAX_PARTS][C][2]
 * (losses.py:15-18, 23-26: x[:, c].min() / .max() over the whole batch) */
int dg_minmax_partial(int dtype, const void* x, int64_t pixels, int64_t ld, int C, float* partial, void* stream);
/* minmax[C][2] = {min, max} over the partials */
int dg_minmax_finish(const float* partial, int C, float* minmax, void* stream);
/* out[n][c][h][w] (planar fp32) = (x[n][h][w][c] - min_c) / (max_c - min_c)   (losses.py:20-21, 28-29) */
int dg_normalise_planar(int dtype, const void* x, int N, int H, int W, int64_t ld, int C, const float* minmax,
                        float* out, void* stream);
/* one scale of pytorch_msssim `_ssim`: sums[plane][0] += sum of the SSIM map, sums[plane][1] += sum of the CS map over
 * the 'valid' (H-win+1)x(W-win+1) window positions of every HxW plane of X, Y (planar fp32) */
int dg_ssim_level(const float* X, const float* Y, int planes, int H, int W, const dg_ssim_params* p, float* sums,
                  void* stream);
/* avg_pool2d(kernel 2, stride 2, padding = size % 2 per dim, padded zeros counted) between scales (`ms_ssim`) */
int dg_avgpool2(const float* in, float* out, int planes, int H, int W, void* stream);
/* out[0] = sum over planes of prod_l relu(mean_l)^weight_l, mean_l = CS mean for l < levels-1, SSIM mean at the last
 * scale; sums is [levels][planes][2].  The caller divides by the (global) plane count (`size_average=True`). */
int dg_msssim_finish(const float* sums, int levels, int planes, const dg_msssim_combine* cmb, float* out, void* stream);

/* Finite-difference physics metrics `divergence_loss` / `vorticity_loss` (DoWnGAN/GAN/losses.py:119-193; known answers in
 * DoWnGAN/GAN/tests/test_losses.py:75-116).  sums[10] (double, pre-zeroed) += {sum r, sum r^2, sum f, sum f^2, sum r*f} of
 * the divergence (dudy + dvdx) of `hr` (r) and `fake` (f), then the same five of the vorticity (dvdx - dudy); channel 0 = u,
 * channel 1 = v, differences on the [1:, 1:] window.  The host forms MSE(r/std(r), f/std(f)) with the unbiased std. */
int dg_div_vort_sums(int dtype, const void* hr, int64_t ldhr, const void* fake, int64_t ldfake, int N, int H, int W,
                     double* sums, void* stream);

/* Data feed.  The reference keeps the whole train set on the device (stage.py:28-31) and lets a
 * shuffling `DataLoader` index it through `NetCDFSR.__getitem__` (dataloader.py:26-33, stage.py:69-72).  Here the set stays
 * resident in HBM as [n][H*W][c_real] in the compute dtype and one launch forms a minibatch in native layout:
 * dst[b][p][c] = src[idx[b]][p][c] for c < c_real, 0 for the padding channels up to c_pad.  idx: B int64 on the device. */
int dg_gather_samples(int dtype, const void* src, int64_t HW, int c_real, const int64_t* idx, int B, void* dst, int c_pad,
                      void* stream);

/* MXFP8 conv path (BASELINE.json configs[4]: "fp8 (CDNA4 fp8 MFMA) conv path"; reference math of the layers it serves:
 * DoWnGAN/networks/critic.py:25-88, the critic's seven 128..1024-channel convs).
 * dg_quant_mxfp8: rows x C values (C % 128 == 0, row stride `ld` elements, dtype DG_BF16 or DG_F32) -> OCP FP8 E4M3 bytes
 *   q[rows][ldq] + one E8M0 scale byte per block of 32 consecutive channels (OCP MX layout), scales[rows][C/32];
 *   scale = 2^(floor(log2 amax) - 8), elements = round-to-nearest-even(x / scale) saturated at +-448.
 *   Serves activations / adjoints (rows = pixels) and conv weight packs (rows = Nout * 9, C = Cred).
 * dg_conv3x3_fwd_f8 / dg_conv3x3_dgrad_f8: dg_conv3x3_fwd / _dgrad with both MFMA operands in that format and fp32
 *   accumulation; `g` describes the layer as for the bf16 calls (g->dtype = DG_BF16: the type of y / dx and of every
 *   epilogue operand), `q` carries the quantised source (xq, xs, pixel stride ldxq bytes) and weight pack (wq, ws: the
 *   forward pack [Cout][9][Cin] for _fwd, the data-gradient pack [Cin][9][Cout] for _dgrad).  Shapes: reduction channels a
 *   multiple of 128, more than 64 output channels, no pixel shuffle; anything else returns DG_ERR_BAD_SHAPE. */
typedef struct dg_f8_operands {
  const void* xq;   /* fp8 source, NHWC */
  const void* xs;   /* its scales [pixels][ldxs] (the first Cred/32 bytes of each row are used) */
  int64_t ldxq;     /* pixel stride of xq in bytes */
  int64_t ldxs;     /* scale bytes per pixel of xs; 0 = Cred/32 (dense); a multiple of 4; < 0 = xs is ONE row of Cred/32 exponents valid for every pixel (a uniform-scale source) */
  const void* wq;   /* fp8 weight pack [Nout][9][Cred] */
  const void* ws;   /* its scales [Nout][9][Cred/32] */
} dg_f8_operands;
/* out[b] = min(254, E8M0 exponent byte of the magnitude in amax[b] (floor(log2) - 8 + 127, >= 0) + margin), b < nblocks <= 64, and amax[b] = 0:
 * turns the census a first-layer launch kept (dg_epilogue.out_amax) into the exponents of the next pass's uniform-scale copy -- the same
 * values dg_block_exp_max derives from the MXFP8 scale bytes of that tensor. */
int dg_exp_from_amax(void* amax, int nblocks, int margin, void* out, void* stream);

/* out[b] = min(254, max over rows r of scales[r * ld + b] + margin), b < nblocks: the largest MXFP8 block exponent a tensor's
 * 32-channel block b reached anywhere (scales = the E8M0 bytes dg_quant_mxfp8 / dg_epilogue.out_qs wrote) -- the per-block exponent
 * of the uniform-scale copy (dg_epilogue.out_u) of the NEXT pass over the same tensor.  nblocks <= 64, a multiple of 4; `scratch`
 * = 64 * DG_EXP_BATCH_MAX dwords of device memory the caller owns, ZERO on entry and left zero (one per stream that calls this
 * concurrently). */
int dg_block_exp_max(const void* scales, int64_t rows, int64_t ld, int nblocks, int margin, void* out, void* scratch, void* stream);

/* The same for up to DG_EXP_BATCH_MAX tensors in one launch (the fp8 train step refreshes a dozen exponent tables per critic pass);
 * scratch = 64 * DG_EXP_BATCH_MAX dwords, zero on entry and left zero. */
#define DG_EXP_BATCH_MAX 16
typedef struct dg_exp_batch {
  const void* scales[DG_EXP_BATCH_MAX];
  int64_t rows[DG_EXP_BATCH_MAX];
  int64_t ld[DG_EXP_BATCH_MAX];
  int nblocks[DG_EXP_BATCH_MAX];
  void* out[DG_EXP_BATCH_MAX];
  int n;
} dg_exp_batch;
int dg_block_exp_max_batch(const dg_exp_batch* b, int margin, void* scratch, void* stream);

/* Stand-alone form of dg_epilogue.out_u: q[r][c] = E4M3(src[r][c] / 2^(exps[c / 32] - 127)) (saturated at +-448; a block holding a
 * NaN / Inf becomes 32 x NaN), for tensors that do not come out of a conv epilogue (the critic's last adjoint, written by the FC's
 * input gradient).  src bf16 / fp32 [rows][ld], C % 128 == 0, q [rows][ldq] bytes. */
int dg_quant_uniform(int src_dtype, const void* src, int64_t rows, int64_t ld, int C, const void* exps, void* q, int64_t ldq, void* stream);

int dg_quant_mxfp8(int src_dtype, const void* src, int64_t rows, int64_t ld, int C, void* q, int64_t ldq, void* scales,
                   int64_t ldqs, void* stream);   /* ldqs: scale bytes per row, 0 = C/32 */
int dg_conv3x3_fwd_f8(const dg_conv_geom* g, const dg_epilogue* ep, const dg_f8_operands* q, void* y, void* stream);
int dg_conv3x3_dgrad_f8(const dg_conv_geom* g, const dg_epilogue* ep, const dg_f8_operands* q, void* dx, void* stream);

/* Dataset preprocessing of the data feed.
 * dg_moments: acc[3] (double, pre-zeroed) += { sum, sum of squares, count } over the non-NaN elements of x[n] -- the
 *   moments behind `xr_standardize_array` (DoWnGAN/helpers/gen_experiment_datasets.py:195-201: da.mean(skipna=True),
 *   da.std(skipna=True), population std), accumulated chunk by chunk over a field's whole record.  x 16-byte aligned.
 * dg_stage_fields: dst[p][k] = (plane[k][p] - mean[k]) * inv_std[k] for p < npix, k < c -- the standardisation itself fused
 *   with the [time, var, lat, lon] staging of DoWnGAN/GAN/stage.py:28-31, written as the HBM-resident [n][H*W][c] store that
 *   dg_gather_samples reads.  A field that must not be standardised (the binary `land_sea_mask`,
 *   gen_experiment_datasets.py:208-209) is passed with mean 0 and inv_std 1.  NaNs stay NaNs, as in the reference. */
#define DG_MAX_FIELDS 8
typedef struct dg_field_planes {
  const float* plane[DG_MAX_FIELDS]; /* c device pointers, each [npix] fp32 */
  float mean[DG_MAX_FIELDS];
  float inv_std[DG_MAX_FIELDS];
  int c;
} dg_field_planes;
int dg_moments(const float* x, int64_t n, double* acc, void* stream);
int dg_stage_fields(int dtype, const dg_field_planes* f, int64_t npix, void* dst, void* stream);

/* Frequency-separation variant (DoWnGAN/GAN/wasserstein_fs.py:36-46,73-86, hyperparams.py:31-35):
 * low = AvgPool2d(5, stride 1)(ReplicationPad2d(2)(x)) and/or high = x - low of an NHWC tensor (C padded channels,
 * multiple of 8); either output may be NULL. */
int dg_lowpass5(int dtype, const void* x, int64_t ldx, int N, int H, int W, int C, void* low, int64_t ldl, void* high,
                int64_t ldh, void* stream);
/* out = low^T(g): the adjoint of the operator above (what autograd applies in the reference's `g_loss.backward()`,
 * wasserstein_fs.py:88, to reach `fake` through `fake_low` and `fake_high`). */
int dg_lowpass5_adjoint(int dtype, const void* g, int64_t ldg, int N, int H, int W, int C, void* out, int64_t ldo,
                        void* stream);

/* ---- debugging aids (off by default; csrc/debug.hip) ----------------------------------------------------------------
 * dg_set_deterministic_workspace: the split-K weight gradients (dg_conv3x3_wgrad, _wgrad_dense) and the small reductions
 *   (dg_colsum, dg_sumsq_rows, dg_l1, dg_sqdiff, dg_linear_fwd) accumulate partial results with fp32 atomics, whose order -- and
 *   so the last bits of every gradient -- differs from run to run.  With a caller-owned device workspace registered here
 *   (>= 1 MiB, 16-byte aligned; NULL switches the mode off again) each of those launches writes its partials side by side into
 *   the workspace and adds them to the target in a fixed order: results are bit-identical between runs.  Process-wide; the
 *   workspace must stay alive while registered and serves one stream at a time.  A launch that needs more room than the
 *   workspace has runs with fewer splits (slower, same guarantee).
 * dg_count_nonfinite: counts[i] = number of NaN / Inf elements of buffer i -- the stand-in for the reference's global
 *   torch.autograd.set_detect_anomaly(True) (DoWnGAN/GAN/wasserstein.py:13), run once per iteration by
 *   TrainEngine(check_finite=True) instead of after every op. */
#define DG_FINITE_MAX 8
typedef struct dg_finite_bufs {
  const void* ptr[DG_FINITE_MAX];
  int64_t n[DG_FINITE_MAX];      /* elements */
  int dtype[DG_FINITE_MAX];      /* DG_F32 or DG_BF16 */
  int nbuf;
} dg_finite_bufs;
int dg_set_deterministic_workspace(void* ws, int64_t bytes);
int dg_deterministic(void);      /* 1 while a workspace is registered */
int dg_count_nonfinite(const dg_finite_bufs* bufs, uint32_t* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DOWNGAN_HIP_H */
