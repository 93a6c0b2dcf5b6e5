/*
 * lic.h -- C ABI of liblic_hip.so: the MI355X (gfx950) implementation of the
 * analysis/synthesis + hyperprior + likelihood + rate-distortion hot path of
 * achraf-15/neural_image_compression.
 *
 * The reference has no FFI: its "operator API" for this path is torch's ATen ops reached from
 * the nn.Module surface.  Each entry point below names the reference call
 * site(s) whose device arithmetic it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *  - All pointers are DEVICE pointers to fp32 unless stated.  Activations are NHWC: element
 *    (b,h,w,c) of a tensor with row pitch `ld` (floats per pixel, ld >= C) lives at
 *    ((b*H + h)*W + w)*ld + c.  `ld` lets two producers write disjoint channel ranges of one
 *    buffer: the reference's torch.cat([phi, psi]) (Models.py:73) is not materialised in the fp32
 *    path -- the context conv and the hyper decoder's last conv write the two halves of one tensor
 *    (models.py; the backward's two gradient slices are still made contiguous by a copy each).
 *  - The library never allocates, frees or synchronises; every launch goes to `stream`
 *    (a hipStream_t passed as void*), so every call is hipGraph-capturable.  Workspaces are
 *    caller-owned; sizes come from the *_workspace_bytes queries.
 *  - Return value: LIC_OK (0) or a negative lic_status.  Nothing throws or exits.
 *  - Re-entrant: no global mutable state; safe to call from autograd's backward thread.
 */
#ifndef LIC_H
#define LIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIC_ABI_VERSION 3

typedef void* lic_stream_t; /* hipStream_t */

enum lic_status {
  LIC_OK = 0,
  LIC_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, ...) */
  LIC_ERR_UNSUPPORTED = -2, /* geometry outside what the kernels implement */
  LIC_ERR_LAUNCH = -3,      /* hipLaunchKernel failed; see lic_last_hip_error() */
  LIC_ERR_WORKSPACE = -4    /* workspace too small */
};

/* epilogues of lic_igemm (fused into the producing kernel) */
enum lic_epilogue {
  LIC_EPI_NONE = 0,           /* v = acc + bias                                              */
  LIC_EPI_LEAKY = 1,          /* v = leaky(acc + bias); with res: out2 = v + res            */
  LIC_EPI_MUL_LEAKY_MASK = 2, /* v = acc * (aux > 0 ? 1 : slope)  (grad through in-place LeakyReLU) */
  LIC_EPI_GDN = 3,            /* n = acc + bias; out2 = n; v = aux * rsqrt(n)                 */
  LIC_EPI_IGDN = 4,           /* n = acc + bias; out2 = n; v = aux * sqrt(n)                  */
  LIC_EPI_GDN_BWD = 5,        /* v = aux * rsqrt(aux3) + 2 * aux2 * acc                       */
  LIC_EPI_IGDN_BWD = 6,       /* v = aux * sqrt(aux3)  + 2 * aux2 * acc                       */
  /* convolution + the GDN / IGDN that follows it, in one kernel (lic_igemm_fused_gdn_supported):
   *   x = acc + bias -> out3 (may be NULL);  n = aux2 + x^2 . aux -> out2 (may be NULL);  out = x * rsqrt(n) | x * sqrt(n)
   *   aux = gamma_eff^T packed by lic_pack_weight(taps=1, K=Cout, N=Cout), aux2 = beta_eff [Cout].
   *   Bitwise identical to LIC_EPI_NONE followed by a prologue=1 / LIC_EPI_GDN contraction launch. */
  LIC_EPI_CONV_GDN = 7,
  LIC_EPI_CONV_IGDN = 8
  /* for every epilogue except LEAKY a non-null `res` is added to v before the store */
};

/* ------------------------------------------------------------------------------------------
 * lic_igemm -- implicit-GEMM convolution family on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   rows  = output pixels, cols = output channels, K = live taps x input channels.
 *   transposed == 0:  out[b,oh,ow,:] = sum_{r,s} in[b, oh*stride-pad+r, ow*stride-pad+s, :] . W[r,s]
 *   transposed == 1:  out[b,oy,ox,:] = sum_{r,s : (oy+pad-r) % stride == 0 ...}
 *                                       in[b,(oy+pad-r)/stride,(ox+pad-s)/stride,:] . W[r,s]
 *   W is the packed weight produced by lic_pack_weight: [kh*kw][ceil(Cin/16)][Npad/32][2][64][4], Npad = Cout
 *   rounded up to 64 (32 when Cout <= 32)
 *   (an opaque MFMA-operand order: element (k, n) of a chunk sits at lane (n%32) + 32*((k%16)/8),
 *   load q = (k%8)/4, float k%4)
 * Replaces: nn.Conv2d / nn.ConvTranspose2d forward and their input gradients
 *   (Components.py:10-16,39-45,69-73,99-103; Layers.py:21,38,40,43,74,76,99,101,103;
 *   ParametersModels.py:22-34; ContextModels.py:19-20 via tap_mask), the GDN/IGDN channel
 *   contraction (compressai GDN at Components.py:11,13,15,40,42,44; Layers.py:41,75) with
 *   prologue=1 (square the input), nn.LeakyReLU (Components.py:70,72,100,102; Layers.py:39,73,100;
 *   ParametersModels.py:23,25) and the residual adds (Layers.py:57,85,118) as epilogues.
 * ------------------------------------------------------------------------------------------ */
typedef struct lic_igemm_desc {
  const float* in;
  const float* w;
  const float* bias; /* [Cout] or NULL */
  float* out;
  float* out2; /* secondary output (GDN norm, LEAKY+res sum) or NULL */
  const float* aux;
  const float* aux2;
  const float* aux3;
  const float* res;
  int64_t in_ld, out_ld, out2_ld, aux_ld, aux2_ld, aux3_ld, res_ld;
  int32_t B, Hi, Wi, Cin;
  int32_t Ho, Wo, Cout;
  int32_t kh, kw, stride, pad;
  int32_t transposed;
  int32_t prologue; /* 0: none, 1: square the gathered input, 2 / 3: GDN / IGDN backward -- the 1x1
                     * contraction's operand is t = dL/dnorm built on the fly from in = g, aux2 = x,
                     * aux3 = norm (-0.5 g x norm^-3/2, or 0.5 g x norm^-1/2 for IGDN) and also
                     * written to out2 for the d-gamma / d-beta launches (replaces lic_gdn_dnorm) */
  int32_t epilogue; /* enum lic_epilogue */
  uint32_t tap_mask; /* bit (r*kw+s) set = tap is live; 0 = all taps (kh*kw <= 32) */
  float slope;       /* LeakyReLU negative slope */
  void* workspace;   /* optional (may be NULL): lets small layers split K across workgroups */
  size_t workspace_bytes;
  float* out3;       /* LIC_EPI_CONV_GDN / CONV_IGDN: the pre-normalisation conv output, or NULL */
  int64_t out3_ld;
  /* Overrides of the launch plan, 0 = automatic.  The automatic tile depends on the batch (a 128-row
   * tile needs >= 512 workgroups), so parity tests use these to put every kernel variant the full-size
   * workloads dispatch in front of the oracle at sizes the oracle finishes in seconds, and the entropy
   * coder pins one variant so that encoder and decoder build bit-identical tables whatever their batch.
   *   lic_igemm (fp32): force_bm in {64,128}, force_tn in {1,2,3} (both or neither; needs float4-aligned operands and
   *   Npad % (64*force_tn) == 0, else LIC_ERR_UNSUPPORTED); force_split >= 1: K splits
   *   (1 = never split; needs `workspace`). */
  int32_t force_bm, force_tn, force_split, reserved0;
} lic_igemm_desc;

/* 1 when lic_igemm can run LIC_EPI_CONV_GDN / LIC_EPI_CONV_IGDN for these channel counts
 * (all output channels in one full tile: Cout in {64,128,192}; Cin % 4 == 0) */
int lic_igemm_fused_gdn_supported(int32_t Cin, int32_t Cout);
/* 1 when fusing is also expected to be faster than two launches for this geometry (only the
 * shape fields of `d` are read) */
int lic_igemm_fused_gdn_preferred(const lic_igemm_desc* d);

/* workspace size that enables split-K for `d` (0 = the layer is large enough not to need it) */
size_t lic_igemm_workspace_bytes(const lic_igemm_desc* d);

int lic_igemm(const lic_igemm_desc* d, lic_stream_t stream);
/* name of the kernel variant lic_igemm launches for `d`, as rocprofv3 prints it (profiling aid) */
int lic_igemm_kernel_name(const lic_igemm_desc* d, char* buf, size_t n);
/* Weight packing for lic_igemm.  Logical element (tap, k, n) is read from
 * src[tap*s_tap + k*s_k + n*s_n]; dst holds lic_packed_weight_floats(taps, K, N) floats
 * (zero padded).  Any [Cout,Cin,kh,kw] / [Cin,Cout,kh,kw] weight, its transpose for the data
 * gradient, or a dense [K][N] matrix is expressed through the three strides. */
int64_t lic_packed_weight_floats(int32_t taps, int32_t K, int32_t N);
int lic_pack_weight(const float* src, float* dst, int32_t taps, int32_t K, int32_t N, int64_t s_tap,
                    int64_t s_k, int64_t s_n, lic_stream_t stream);
/* which workgroup tile lic_igemm will launch for `d` (kernel name igemm_kernel<BM,BN>) and how
 * many multiply-adds it will issue on live taps: lets a profiler attribute time and FLOPs. */
int lic_igemm_plan(const lic_igemm_desc* d, int32_t* BM, int32_t* BN, int64_t* live_macs);

/* ------------------------------------------------------------------------------------------
 * lic_wgrad -- weight-gradient contraction over pixels on fp32 MFMA, split-K + slab reduce.
 *   P ("plain") lives on the small grid [B,Hs,Ws,Cp]; G ("gathered") on the large grid
 *   [B,Hl,Wl,Cg] sampled at (hs*stride-pad+r, ws*stride-pad+s).
 *   R[tap][m][n] = sum_{b,hs,ws} (g_is_row ? G : P)[..m] * (g_is_row ? P : G)[..n]
 *   dst[m*dst_sm + n*dst_sn + tap*dst_stap] = scale * R[tap][m][n]
 * Replaces: autograd's weight gradients of every conv / convT / 1x1 on the path and the
 *   dgamma contraction of GDN (sq_g = 1 squares G on load).
 * ------------------------------------------------------------------------------------------ */
typedef struct lic_wgrad_desc {
  const float* p;
  const float* g;
  float* dst;
  int64_t p_ld, g_ld;
  int64_t dst_sm, dst_sn, dst_stap;
  int32_t B, Hs, Ws, Cp;
  int32_t Hl, Wl, Cg;
  int32_t kh, kw, stride, pad;
  int32_t g_is_row;
  int32_t sq_p, sq_g;
  float scale;
  /* overrides of the launch plan, 0 = automatic (see lic_igemm_desc): tile in 64-channel units, one of
   * (1,1) (1,3) (2,1) (2,2) (2,3) (3,3) -- (3,3) needs both channel counts % 192 == 0 -- and the number
   * of pixel splits */
  int32_t force_tm, force_tn, force_split;
} lic_wgrad_desc;

size_t lic_wgrad_workspace_bytes(const lic_wgrad_desc* d);
int lic_wgrad(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, lic_stream_t stream);
/* kernel variant wgrad_kernel<TM,TN,...> and split-K factor lic_wgrad will use (profiling aid) */
int lic_wgrad_plan(const lic_wgrad_desc* d, int32_t* TM, int32_t* TN, int32_t* splitk);
/* lic_wgrad in two steps, so a profiler can time the MFMA kernel apart from the slab reduction:
 * stage 1 = partial sums only, stage 2 = reduction only, 0 = both (== lic_wgrad) */
int lic_wgrad_stage(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, int32_t stage,
                    lic_stream_t stream);
/* name of the MFMA kernel lic_wgrad launches for `d`, as rocprofv3 prints it */
int lic_wgrad_kernel_name(const lic_wgrad_desc* d, char* buf, size_t n);

/* column sums over pixels: out[c] = scale * sum_p in[p*ld + c]  (bias gradients, GDN dbeta) */
size_t lic_colsum_workspace_bytes(int64_t P, int32_t C);
int lic_colsum(const float* in, int64_t ld, int64_t P, int32_t C, float scale, float* out,
               void* workspace, size_t workspace_bytes, lic_stream_t stream);

/* generic 3-D strided copy: dst[i*d0+j*d1+k*d2] = src[i*s0+j*s1+k*s2]  (weight re-packing) */
int lic_permute3(const float* src, float* dst, int32_t n0, int32_t n1, int32_t n2, int64_t s0,
                 int64_t s1, int64_t s2, int64_t d0, int64_t d1, int64_t d2, lic_stream_t stream);

/* 3-channel layers (image side): patches <-> columns, C small.
 * im2col:  col[(b,oh,ow)][(r*kw+s)*C + c] = x[b, oh*stride-pad+r, ow*stride-pad+s, c] (0 outside),
 *          columns [kh*kw*C, Kpad) are zero.       (stem conv Components.py:10, Layers.py:38,43)
 * col2im:  out[b,oy,ox,c] = bias[c] + sum_{r,s} col[(b,ih,iw)][(r*kw+s)*C + c] with
 *          oy = ih*stride-pad+r.                    (last convT Components.py:45, :60)            */
int lic_im2col(const float* x, float* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho,
               int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t Kpad,
               lic_stream_t stream);
int lic_col2im(const float* col, const float* bias, float* out, int32_t B, int32_t Hi, int32_t Wi,
               int32_t C, int32_t Ho, int32_t Wo, int32_t kh, int32_t kw, int32_t stride,
               int32_t pad, int32_t Kpad, lic_stream_t stream);

/* ---- elementwise ------------------------------------------------------------------------- */
/* w[i] *= mask[i]  (ContextModels.py:19, in place) */
int lic_mul_inplace(float* w, const float* mask, int64_t n, lic_stream_t stream);
/* dx = y > 0 ? dy : slope*dy  (LeakyReLU(inplace=True) backward from its output) */
int lic_leaky_bwd(const float* y, const float* dy, float* dx, int64_t n, float slope,
                  lic_stream_t stream);
/* out = max(p,bound)^2 - pedestal  (compressai NonNegativeParametrizer forward) */
int lic_gdn_reparam(const float* p, float* out, int64_t n, float bound, float pedestal,
                    lic_stream_t stream);
/* dp = pass ? dout*2*max(p,bound) : 0, pass = (p >= bound) | (that product < 0)  (LowerBound bwd) */
int lic_gdn_reparam_bwd(const float* p, const float* dout, float* dp, int64_t n, float bound,
                        lic_stream_t stream);
/* lic_gdn_reparam_bwd for beta [nb] and gamma [ng] of one GDN in one launch (either count may be 0) */
int lic_gdn_reparam_bwd2(const float* pb, const float* dob, float* dpb, int64_t nb, float bound_b, const float* pg,
                         const float* dog, float* dpg, int64_t ng, float bound_g, lic_stream_t stream);
/* t = dL/dnorm: inverse ? 0.5*g*x*rsqrt(n) : -0.5*g*x*rsqrt(n)/n   (dense [n] tensors) */
int lic_gdn_dnorm(const float* g, const float* x, const float* norm, float* t, int64_t n,
                  int32_t inverse, lic_stream_t stream);
/* training: out = v + (u - 0.5); eval: out = rint(v)   (Models.py:55-64) */
int lic_quantize(const float* v, const float* u, float* out, int64_t n, int32_t training,
                 lic_stream_t stream);

/* ---- entropy models (NHWC, P = B*h*w pixels) ------------------------------------------------ */
/* ParametersModels.py:43-64.  raw/out rows hold G*K*M channels (G = 2 if K == 1 else 3):
 * K==1: [mu | sigma]; K>1: [w | mu | sigma] with channel k*M+m inside each third. */
int lic_entropy_params_fwd(const float* raw, float* out, int64_t P, int32_t M, int32_t K,
                           lic_stream_t stream);
int lic_entropy_params_bwd(const float* raw, const float* out, const float* dout, float* draw,
                           int64_t P, int32_t M, int32_t K, lic_stream_t stream);
/* EntropyModels.py:188-233 + clamp :29-31 + log Models.py:87.  x,p,logp: [P][M]; params as above.
 * dp / dlogp may be NULL (treated as zero). */
int lic_gmm_likelihood_fwd(const float* x, const float* params, float* p, float* logp, int64_t P,
                           int32_t M, int32_t K, float bound, lic_stream_t stream);
int lic_gmm_likelihood_bwd(const float* x, const float* params, const float* dp,
                           const float* dlogp, float* dx, float* dparams, int64_t P, int32_t M,
                           int32_t K, float bound, lic_stream_t stream);
/* EntropyModels.py:49-151.  fe_params: [C][43] packed = matrices (3,9,9,3) | biases (3,3,3,1) |
 * factors (3,3,3), each in the reference's row-major (out,in) order.  x,p,logp: [P][C]. */
#define LIC_FE_NPARAM 43
int lic_factorized_fwd(const float* x, const float* fe_params, float* p, float* logp, int64_t P,
                       int32_t C, float bound, lic_stream_t stream);
int lic_factorized_bwd(const float* x, const float* fe_params, const float* dp, const float* dlogp,
                       float* dx, float* dfe_params, int64_t P, int32_t C, float bound,
                       lic_stream_t stream);
/* The factorised model's 11 parameter tensors (EntropyModels.py:62-86, each contiguous [C][out][in]) gathered into
 * the [C][43] operand above / its [C][43] gradient scattered into a parameter-major buffer (parameter k's [C][n_k]
 * block at offset C * prefix_k: each block has the parameter's own layout, so its views are the parameters'
 * gradients), one launch each.  params11_host: HOST array of the 11 device pointers in the order matrices, biases,
 * factors. */
int lic_fe_pack(const void* const* params11_host, float* packed, int32_t C, lic_stream_t stream);
int lic_fe_unpack(const float* dpacked, float* flat, int32_t C, lic_stream_t stream);
/* channel_logits_cumulative (EntropyModels.py:153-169): out[i] = L_ch(xs[i]) */
int lic_factorized_channel_logits(const float* fe_params, int32_t ch, const float* xs, float* out,
                                  int64_t n, lic_stream_t stream);

/* ---- rate-distortion loss (RateDistortionLoss.py:5-49) ---------------------------------------
 * Per-image element counts ny, nz, nx; images are contiguous blocks (layout-agnostic sums).
 * out[0..8] = loss, bpp_y, bpp_z, bpp_total, mse, psnr, bits_y, bits_z, bits_total; out[9..15] = 0 (every slot is written);
 * out[16 .. 16+B) = mse_per_image; out[16+B .. 16+2B) = psnr_per_image. */
size_t lic_rd_loss_workspace_bytes(int32_t B);
int lic_rd_loss_fwd(const float* logp_y, int64_t ny, const float* logp_z, int64_t nz,
                    const float* x_hat, const float* x, int64_t nx, int32_t B, int64_t num_pixels,
                    float lambda_rd, float* out, void* workspace, size_t workspace_bytes,
                    lic_stream_t stream);
/* gl: device pointer to the upstream gradient of `loss` (a single float) */
int lic_rd_loss_bwd(const float* x_hat, const float* x, int64_t ny, int64_t nz, int64_t nx,
                    int32_t B, int64_t num_pixels, float lambda_rd, const float* gl, float* dlogp_y,
                    float* dlogp_z, float* dx_hat, lic_stream_t stream);

/* ---- bf16-storage variants (BASELINE config 3) --------------------------------------------------
 * Activations / auxiliaries are bf16 NHWC (pitches and channel counts multiples of 8), weights are
 * packed to bf16 by lic_pack_weight_bf16 ([tap][ceil(K/32)][ceil64(N)/32][2][64 lanes][8], MFMA operand order), bias and all
 * accumulation are fp32 (v_mfma_f32_32x32x16_bf16).  lic_igemm_bf16 supports the NONE / LEAKY / GDN /
 * IGDN / GDN_BWD / IGDN_BWD epilogues (no residual) and tap_mask; `out` is bf16, or fp32 when out_f32 != 0.
 * Parameter gradients (lic_wgrad_bf16 dst, lic_colsum_bf16 out) are fp32. */
int64_t lic_packed_weight_bf16_elems(int32_t taps, int32_t K, int32_t N);
int lic_pack_weight_bf16(const float* src, void* dst, int32_t taps, int32_t K, int32_t N, int64_t s_tap,
                         int64_t s_k, int64_t s_n, lic_stream_t stream);
int lic_igemm_bf16(const lic_igemm_desc* d, int32_t out_f32, lic_stream_t stream);
/* workspace size that lets lic_igemm_bf16 split K across workgroups for `d` (0 = large enough not to need it);
 * pass the buffer in d->workspace / workspace_bytes.  The split and the choice between the tap-major tiles and the
 * chunk-major halo-resident variant (the two things that change the summation order of an output) are functions of
 * per-image geometry only: an image's bits do not depend on the batch it is computed in.  force_split of the
 * descriptor is honoured (1 = never, n > 1 = n splits); force_bm in {64, 128, 256, 512}: 256 = the 8-wave
 * ping-pong tile, 512 = the halo-resident 5x5 stride-2 kernel wherever a launch is eligible for it (other
 * launches keep their automatic tile); force_tn is not used on this path (the N tile follows from the channel
 * count; a value that contradicts it returns LIC_ERR_UNSUPPORTED). */
size_t lic_igemm_bf16_workspace_bytes(const lic_igemm_desc* d);
/* lic_igemm_bf16 also runs LIC_EPI_CONV_GDN / LIC_EPI_CONV_IGDN (the conv -> GDN pairs of Components.py:10-15,
 * 39-44 in one launch) when this returns 1 (Cout in {64,128,192}: one tile spans every output channel; bf16 `out`):
 * aux = gamma_eff^T packed by lic_pack_weight_bf16_kperm(taps=1, K=Cout, N=Cout), aux2 = beta_eff (fp32 [Cout]),
 * out3 / out2 = bf16 conv output / norm for the backward pass (either may be NULL: inference writes `out` only).
 * The rounding points are those of LIC_EPI_NONE followed by the prologue=1 / LIC_EPI_GDN launch (x and x^2 to
 * bf16, fp32 norm); the pool sums each group of 16 channels in a different order, so norms agree to fp32
 * rounding, not bitwise. */
int lic_igemm_bf16_fused_gdn_supported(int32_t Cin, int32_t Cout);
/* lic_stem_gdn_bf16 -- the RGB stem of the analysis transform and the GDN behind it (Components.py:10-11:
 * nn.Conv2d(3, C, 5, stride=2, padding=2) -> GDN(C)) in one launch that reads the fp32 NHWC image and writes the
 * normalised bf16 feature map (no column matrix).  w_packed: lic_pack_stem_weight_bf16 of the [C][3][5][5] weight
 * (lic_stem_weight_bf16_elems(C) bf16 elements); gamma_packed: gamma_eff^T by lic_pack_weight_bf16_kperm(taps 1,
 * K = N = C); beta_eff fp32 [C]; y / conv_out / norm: bf16 [B][ceil(H/2)][ceil(W/2)][C], the last two may be NULL
 * (they feed the backward pass).  Same rounding points as lic_igemm_bf16's LIC_EPI_CONV_GDN. */
int lic_stem_gdn_bf16_supported(int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad);
int64_t lic_stem_weight_bf16_elems(int32_t Cout);
int lic_pack_stem_weight_bf16(const float* w, void* dst, int32_t Cout, lic_stream_t stream);
int lic_stem_gdn_bf16(const float* x, const void* w_packed, const float* bias, const void* gamma_packed,
                      const float* beta_eff, void* y, void* conv_out, void* norm, int32_t B, int32_t H, int32_t W,
                      int32_t Cout, int32_t inverse, lic_stream_t stream);
/* lic_pack_weight_bf16 with the K index permuted inside every group of 16: slot (h = 0..1, e = 0..7) of a group
 * holds k = 8*(e/4) + 4*h + e%4 instead of 8*h + e -- the order in which a lane of the transposed accumulator
 * tile owns channels, so the fused pool reads its x^2 operand from registers */
int lic_pack_weight_bf16_kperm(const float* src, void* dst, int32_t taps, int32_t K, int32_t N, int64_t s_tap,
                               int64_t s_k, int64_t s_n, lic_stream_t stream);
/* names of the kernel variants lic_igemm_bf16 / lic_wgrad_bf16 launch for `d`, as rocprofv3 prints them
 * (force_bm of the descriptor is honoured; the N tile follows from the channel count) */
int lic_igemm_bf16_kernel_name(const lic_igemm_desc* d, char* buf, size_t n);
int lic_wgrad_bf16_kernel_name(const lic_wgrad_desc* d, char* buf, size_t n);
size_t lic_wgrad_bf16_workspace_bytes(const lic_wgrad_desc* d);
int lic_wgrad_bf16(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, lic_stream_t stream);
/* fp32 image -> bf16 columns; bf16 columns -> fp32 image (+ fp32 bias) */
int lic_im2col_bf16(const float* x, void* col, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho,
                    int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t Kpad,
                    lic_stream_t stream);
int lic_col2im_bf16(const void* col, const float* bias, float* out, int32_t B, int32_t Hi, int32_t Wi,
                    int32_t C, int32_t Ho, int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                    int32_t Kpad, lic_stream_t stream);
/* GDN / IGDN backward in bf16 storage as one sweep (replaces lic_gdn_dnorm_bf16 + lic_igemm_bf16(EPI_GDN_BWD) for
 * C in {64, 128}; compressai GDN at Components.py:11-44): t = dL/dnorm (written: the d-gamma / d-beta launches read
 * it), dx = g * norm^-1/2 + 2 x (t . gamma_eff) (inverse: norm^+1/2).  g, x, norm, dx, t: dense bf16 [P][C];
 * gamma_packed = lic_pack_weight_bf16_kperm(gamma_eff, taps 1, K = C, N = C, s_k = C, s_n = 1). */
int lic_gdn_bwd_bf16_supported(int32_t C);
/* colsum_*_partial (optional, both or neither): [lic_gdn_bwd_bf16_partial_rows(P)][C] fp32 per-workgroup column sums of t
 * and of dx -- d beta and the d bias of the convolution in front need one small reduction over these rows instead of a
 * pass over the two activations */
int64_t lic_gdn_bwd_bf16_partial_rows(int64_t P);
int lic_gdn_bwd_bf16(const void* g, const void* x, const void* norm, const void* gamma_packed, void* dx, void* t,
                     float* colsum_t_partial, float* colsum_dx_partial, int64_t P, int32_t C, int32_t inverse,
                     lic_stream_t stream);
/* ... with the pool RECOMPUTED instead of read: norm = beta_eff + x^2 . gamma_eff^T formed as the forward pass forms it
 * (x^2 and the result rounded to bf16), so the forward pass writes two tensors per GDN layer instead of three and this
 * launch reads two instead of three.  gammaT_packed = lic_pack_weight_bf16_kperm(gamma_eff, taps 1, K = C, N = C,
 * s_k = 1, s_n = C) -- the fused conv + GDN kernels' operand. */
int lic_gdn_bwd_bf16_recompute(const void* g, const void* x, const void* gamma_packed, const void* gammaT_packed,
                               const float* beta_eff, void* dx, void* t, float* colsum_t_partial,
                               float* colsum_dx_partial, int64_t P, int32_t C, int32_t inverse, lic_stream_t stream);
/* column sums of two bf16 [P][ld] matrices of one shape in one launch pair (workspace: twice the single size) */
int lic_colsum2_bf16(const void* in_a, const void* in_b, int64_t ld, int64_t P, int32_t C, float scale, float* out_a,
                     float* out_b, void* workspace, size_t workspace_bytes, lic_stream_t stream);
size_t lic_colsum_bf16_workspace_bytes(int64_t P, int32_t C);
int lic_colsum_bf16(const void* in, int64_t ld, int64_t P, int32_t C, float scale, float* out,
                    void* workspace, size_t workspace_bytes, lic_stream_t stream);
/* dx = dy * (y > 0 ? 1 : slope), bf16 tensors, n % 8 == 0: backward of lic_igemm_bf16's LIC_EPI_LEAKY */
int lic_leaky_bwd_bf16(const void* y, const void* dy, void* dx, int64_t n, float slope, lic_stream_t stream);
int lic_gdn_dnorm_bf16(const void* g, const void* x, const void* norm, void* t, int64_t n, int32_t inverse,
                       lic_stream_t stream);
/* lic_quantize (Models.py:55-64) that also writes the bf16 copies the bf16-storage consumers read: v_bf16 = bf16(v) (the
 * hyper-encoder's input), out_bf16 = bf16(out) (decoder, context model, hyper-decoder); either may be NULL; n % 4 == 0 */
int lic_quantize_bf16(const float* v, const float* u, float* out, void* v_bf16, void* out_bf16, int64_t n,
                      int32_t training, lic_stream_t stream);

/* ---- misc ---------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------
 * Stand-alone GDN / IGDN (compressai GDN at Components.py:11-44, Layers.py:41,75)
 * for C in {64,128,192} (lic_gdn_supported): one sweep over the activation per launch, the C x C pool
 * on MFMA out of an LDS tile.  x, y, norm, g, dx, t: dense [P][C] fp32 (NHWC activations, P = B*H*W).
 *   lic_gdn_fwd: norm = beta_eff + x^2 . gamma_eff^T (written unless `norm` is NULL: only the backward pass reads
 *                it); y = x * rsqrt(norm) (inverse: * sqrt(norm)) (+ res)
 *                gammaT_packed = lic_pack_weight(gamma_eff, taps=1, K=C, N=C, s_k=1, s_n=C)
 *   lic_gdn_bwd: t = dL/dnorm (written: the d-gamma / d-beta launches read it),
 *                dx = g * rsqrt(norm) + 2 x (t . gamma_eff)  (inverse: g * sqrt(norm) + ...)
 *                gamma_packed = lic_pack_weight(gamma_eff, taps=1, K=C, N=C, s_k=C, s_n=1)
 * Same chunk / k order as lic_igemm's prologue-1 / prologue-2 route, which serves other channel counts.
 * ------------------------------------------------------------------------------------------ */
int lic_gdn_supported(int32_t C);
int lic_gdn_fwd(const float* x, const float* gammaT_packed, const float* beta_eff, const float* res, float* y,
                float* norm, int64_t P, int32_t C, int32_t inverse, lic_stream_t stream);
/* colsum_*_partial (optional, may be NULL): [lic_gdn_bwd_partial_rows(P)][C] per-workgroup column sums of
 * t and of dx, so that d beta and the d bias of the convolution in front of the GDN need one
 * lic_colsum over a few thousand rows instead of a pass over the whole activation */
int64_t lic_gdn_bwd_partial_rows(int64_t P);
int lic_gdn_bwd(const float* g, const float* x, const float* norm, const float* gamma_packed, float* dx, float* t,
                float* colsum_t_partial, float* colsum_dx_partial, int64_t P, int32_t C, int32_t inverse,
                lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f).2 -- tables for an entropy coder (the reference has none), built on the
 * device from the same distributions the likelihood entries evaluate and consumed by the host range
 * coder of include/lic_codec.h.  A table has S symbols (index i <-> integer value lo + i) and S+1
 * uint32 entries  cum[i] = floor(F_i * (65536 - S)) + i  with F_0 = 0, F_S = 1 and
 * F_i = CDF(lo + i - 0.5): cum[0] = 0, cum[S] = 65536, every symbol keeps frequency >= 1; indices 0
 * and S-1 carry the tails (the coder escapes out-of-window values through them).
 *   lic_factorized_cdf_tables: out[C][S+1], F = sigmoid(L_c(.))         (EntropyModels.py:153-184)
 *   lic_gmm_cdf_tables: per latent element e of [P][M]: center[e] = rint(sum_k w_k mu_k), window
 *     lo = center - W, S = 2W+1, out[P*M][S+1], F = sum_k w_k Phi((x - mu_k)/sigma_k); `params` as
 *     lic_gmm_likelihood_fwd                                    (EntropyModels.py:192-233, utils.py:6-8)
 * ------------------------------------------------------------------------------------------ */
int lic_factorized_cdf_tables(const float* fe_params, int32_t C, int32_t lo, int32_t S, uint32_t* out,
                              lic_stream_t stream);
int lic_gmm_cdf_tables(const float* params, int64_t P, int32_t M, int32_t K, int32_t W, int32_t* center,
                       uint32_t* out, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f).1 -- evaluation metric on the device.
 * Multi-scale SSIM exactly as the reference's evaluator calls it (Evaluator.py:7,38,45:
 * `ms_ssim(recon, orig, data_range=1.0, size_average=True)` of the third-party pytorch-msssim==0.2.1,
 * requirements.txt:5 -- absent offline, PARITY UNPINNED; follows the package's published algorithm:
 * 11-tap sigma-1.5 Gaussian window, valid filtering, K=(0.01,0.03), 5 scales, 2x2 average pooling with
 * zero padding of odd sides, default weights).  x, y: element (b,c,h,w) at b*sb + c*sc + h*sh + w*sw
 * (any NCHW / channels_last view).  out[B*C] = per-image per-channel MS-SSIM (the package's value is
 * its mean); level_out[5][B*C][2] = per-scale (mean ssim, mean cs).  min(H, W) must exceed 160 (the
 * package's assertion) -> LIC_ERR_UNSUPPORTED otherwise.
 * ------------------------------------------------------------------------------------------ */
size_t lic_msssim_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);
int lic_msssim(const float* x, const float* y, int32_t B, int32_t C, int32_t H, int32_t W, int64_t sb,
               int64_t sc, int64_t sh, int64_t sw, float data_range, float* out, float* level_out,
               void* workspace, size_t workspace_bytes, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f).3 -- input pipeline and logging statistics.
 *   lic_u8_to_f32: out[i] = float(in[i]) / 255 (true division: bit-identical to torchvision
 *     ToTensor(), Dataloader.py:7-9) for uint8 NHWC shards; `in` 4-byte, `out` 16-byte aligned.
 *   lic_tensor_stats: stats[6] = count, sum, sum of squares, min, max, NaN count (fp64) and hist[nbins]
 *     (uint64) over [lo, hi] with outliers clamped into the edge bins -- the summary the reference's
 *     logging (Trainer.py:167-217: add_histogram / mean of whole latent tensors) needs, without the
 *     full-tensor D2H copy.  Reproducible (integer atomics + fixed-order fp64 reduction).
 * ------------------------------------------------------------------------------------------ */
int lic_u8_to_f32(const uint8_t* in, float* out, int64_t n, lic_stream_t stream);
size_t lic_tensor_stats_workspace_bytes(void);
int lic_tensor_stats(const float* x, int64_t n, int32_t nbins, float lo, float hi, double* stats,
                     uint64_t* hist, void* workspace, size_t workspace_bytes, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * lic_prep -- every parameter-derived buffer of a model refreshed by ONE launch per optimizer step.
 *   The reference keeps derived values implicit in ATen (cuDNN/oneDNN repack weights internally; compressai's
 *   GDN recomputes beta/gamma re-parametrisations in every forward, Components.py:11-44; ContextModels.py:19
 *   masks the weight in place in every forward).  Here they are explicit buffers: the packed MFMA operand
 *   of every conv / convT weight for the forward and for the data gradient (lic_pack_weight /
 *   lic_pack_weight_bf16 layouts), beta_eff and the two packed gamma_eff panels of every GDN, the
 *   column-matrix forms of the RGB stem / head weights.  A job describes one buffer:
 *     PACK_F32 / PACK_BF16: dst = packed operand of the logical [taps][K][N] tensor whose element (tap, k, n)
 *       is value(src[tap*s_tap + koff(k) + noff(n)]), koff(k) = kdiv ? (k/kdiv)*s_kq + (k%kdiv)*s_kr : k*s_kq
 *       (same for n): the split index expresses the (tap, channel) <-> column maps of the RGB layers;
 *     MAP: dst[i] = value(src[i]), i < N;   MASK_INPLACE: src[i] *= mask[i], i < N (ContextModels.py:19);
 *     value(v) = v * mask (when `mask` is set, same offset as src), then transform 1 = the GDN
 *       re-parametrisation max(v, bound)^2 - pedestal (lic_gdn_reparam).
 *   A pack job may read a tensor that a MASK_INPLACE job of the same launch is masking, provided it names the
 *   same mask: mask values are 0 or 1, so w*m and (w*m)*m are the same float whichever the read observes.
 *   Usage: fill jobs on the host, lic_prep_plan (fills the derived fields, returns the grid size), copy the
 *   array to device memory once, then lic_prep_run once per optimizer step.  Results are bit-identical to
 *   the stand-alone entry points.
 * ------------------------------------------------------------------------------------------ */
enum lic_prep_kind { LIC_PREP_PACK_F32 = 0, LIC_PREP_PACK_BF16 = 1, LIC_PREP_MAP = 2, LIC_PREP_MASK_INPLACE = 3,
                     LIC_PREP_PACK_BF16_KPERM = 4 /* lic_pack_weight_bf16_kperm's order */,
                     LIC_PREP_PACK_BF16_STEM = 5  /* lic_pack_stem_weight_bf16: src = [N][3][5][5], K = 80 */ };
typedef struct lic_prep_job {
  const float* src;
  void* dst;
  const float* mask;
  int64_t s_tap, s_kq, s_kr, s_nq, s_nr;
  int32_t kind; /* enum lic_prep_kind */
  int32_t taps, K, N, kdiv, ndiv;
  int32_t transform; /* 0 none, 1 GDN re-parametrisation */
  float bound, pedestal;
  /* derived by lic_prep_plan */
  int32_t cpt, npad, tiled, v4, block0, nblocks;
  int64_t total;
} lic_prep_job;
int64_t lic_prep_plan(lic_prep_job* jobs, int32_t njobs);
int lic_prep_run(const lic_prep_job* jobs_device, int32_t njobs, int64_t total_blocks, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * lic_adam -- optimizer.step() (Trainer.py:86; the reference trains with torch.optim.Adam, Main.ipynb) for every
 *   parameter of a model in ONE launch: g' = g + weight_decay*p; m += (1-beta1)(g'-m); v = beta2 v + (1-beta2) g'^2;
 *   p -= (lr / bias_correction1) * m / (sqrt(v)/sqrt(bias_correction2) + eps)   -- torch's non-amsgrad arithmetic,
 *   bias_correction{1,2} = 1 - beta{1,2}^step computed by the caller.  Jobs: fill p / m / v / n on the host,
 *   lic_adam_plan (fills block0 / nblocks, returns the grid size), keep a device copy of the array; per step
 *   lic_adam_run with `grads_host`, a HOST array of the njobs (<= 448) gradient device pointers in job order --
 *   gradients are fresh allocations every step, so their addresses travel as kernel arguments.
 * ------------------------------------------------------------------------------------------ */
typedef struct lic_adam_job {
  float* p;       /* parameter, updated in place */
  const float* g; /* unused (gradient pointers are passed to lic_adam_run) */
  float* m;       /* exp_avg */
  float* v;       /* exp_avg_sq */
  int64_t n;      /* elements */
  int32_t block0, nblocks; /* derived by lic_adam_plan */
} lic_adam_job;
int64_t lic_adam_plan(lic_adam_job* jobs, int32_t njobs);
int lic_adam_run(const lic_adam_job* jobs_device, int32_t njobs, int64_t total_blocks, const float* const* grads_host,
                 double lr, double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                 double bias_correction2, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * lic_reduce_batch -- every pending reduction of a backward pass (Trainer.py:84 `loss.backward()`) in ONE launch.
 *   The weight-gradient launches end in a slab reduction and the column sums (bias / GDN beta gradients) in a second
 *   stage: 41 launches of 5-15 us per config-3 step.  lic_wgrad_partial / lic_colsum_partial (fp32 operands),
 *   lic_wgrad_bf16_partial / lic_colsum_bf16_partial / lic_colsum2_bf16_partial launch the first stage only and fill a lic_reduce_job; lic_reduce_batch(jobs, n) -- `jobs` a
 *   HOST array, copied into the kernel arguments, LIC_REDUCE_MAX_JOBS per launch -- finishes them with the arithmetic
 *   and association order of the stand-alone second stages (bitwise the same results), optionally followed per element
 *   by the GDN re-parametrisation's backward (epilogue LIC_REDUCE_EPI_REPARAM: lic_gdn_reparam_bwd with `param`, `bound`
 *   on the reduced value, `param` indexed like `dst`) and with split destination indices (mdiv / ndiv as in lic_prep_job:
 *   offset(m) = mdiv ? (m / mdiv) * sm + (m % mdiv) * smr : m * sm; rows >= Mvalid / columns >= Nvalid are dropped):
 *   the (tap, channel) <-> column maps of the RGB layers' weight gradients.  Workspaces and outputs named by a job must
 *   stay alive until the batch has been launched; block0 / nblocks are filled by lic_reduce_batch.
 * ------------------------------------------------------------------------------------------ */
enum { LIC_REDUCE_SLABS = 0, LIC_REDUCE_COLUMNS = 1 };
enum { LIC_REDUCE_EPI_NONE = 0, LIC_REDUCE_EPI_REPARAM = 1 };
#define LIC_REDUCE_MAX_JOBS 32
typedef struct lic_reduce_job {
  const float* src;   /* SLABS: [splitk][ntaps * Cm * Cn]; COLUMNS: [splitk][Cn] */
  float* dst;
  const float* param; /* epilogue REPARAM */
  int64_t sm, smr, sn, snr, stap;
  int32_t kind, epilogue;
  int32_t splitk, ntaps, Cm, Cn, Mvalid, Nvalid, mdiv, ndiv;
  float scale, bound;
  int32_t block0, nblocks;
} lic_reduce_job;
int lic_wgrad_partial(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, lic_reduce_job* job,
                      lic_stream_t stream);      /* fp32 operands (lic_wgrad's first stage) */
int lic_colsum_partial(const float* in, int64_t ld, int64_t P, int32_t C, float scale, float* out, void* workspace,
                       size_t workspace_bytes, lic_reduce_job* job, lic_stream_t stream);
int lic_wgrad_bf16_partial(const lic_wgrad_desc* d, void* workspace, size_t workspace_bytes, lic_reduce_job* job,
                           lic_stream_t stream);
int lic_colsum_bf16_partial(const void* in, int64_t ld, int64_t P, int32_t C, float scale, float* out, void* workspace,
                            size_t workspace_bytes, lic_reduce_job* job, lic_stream_t stream);
int lic_colsum2_bf16_partial(const void* in_a, const void* in_b, int64_t ld, int64_t P, int32_t C, float scale,
                             float* out_a, float* out_b, void* workspace, size_t workspace_bytes, lic_reduce_job* jobs2,
                             lic_stream_t stream);
/* lic_leaky_bwd_bf16 on [P][C] matrices, with the column sums of dx (the bias gradient of the convolution in front of the LeakyReLU:
 * Components.py:69-73,99-103, ParametersModels.py:22-34) out of the same pass.  workspace: lic_colsum_bf16_workspace_bytes(P, C);
 * `job` NULL: out[C] is complete when the call's launches are; else the sums' second stage is left in *job for
 * lic_reduce_batch.  Same bits as lic_leaky_bwd_bf16 followed by lic_colsum_bf16 / lic_colsum_bf16_partial. */
int lic_leaky_bwd_colsum_bf16(const void* y, const void* dy, void* dx, int64_t P, int32_t C, float slope, float* out,
                              void* workspace, size_t workspace_bytes, lic_reduce_job* job, lic_stream_t stream);
int lic_reduce_batch(const lic_reduce_job* jobs, int32_t njobs, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RGB head in bf16 storage without column matrices (Components.py:45: ConvTranspose2d(C, 3, 5, stride 2, padding 2,
 *   output_padding 1); the column-matrix route of lic_igemm_bf16 + lic_col2im_bf16 / lic_im2col_bf16 remains for
 *   other geometries).
 *   lic_head_convt_bf16: out fp32 [B][2 Hi][2 Wi][3] = conv_transpose2d(x bf16 [B][Hi][Wi][Cin]) + bias; `w_packed` =
 *     lic_pack_weight_bf16(taps 1, K = Cin, N = 75..80) of the matrix w[ci][3 * (5 ky + kx) + colour] (the forward
 *     operand of the column-matrix route); Cin in {64, 128, 192}.  fp32 accumulation, nothing rounded between the
 *     channel contraction and the tap sum.
 *   lic_stem_conv_bf16: y bf16 [B][ceil(H/2)][ceil(W/2)][Cout] = conv2d(x fp32 [B][H][W][3], w, stride 2, padding 2)
 *     + bias, `w_packed` = lic_pack_stem_weight_bf16 of a [Cout][3][5][5] tensor -- the convolution of
 *     lic_stem_gdn_bf16 without the GDN.  The data gradient of the head is this convolution of dL/d(image) with the
 *     head's own weight tensor (ConvTranspose2d stores it as [Cin][3][5][5]).
 * ------------------------------------------------------------------------------------------ */
int lic_head_convt_bf16_supported(int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                                  int32_t out_pad);
int lic_head_convt_bf16(const void* x, const void* w_packed, const float* bias, float* out, int32_t B, int32_t Hi,
                        int32_t Wi, int32_t Cin, lic_stream_t stream);
int lic_stem_conv_bf16(const float* x, const void* w_packed, const float* bias, void* y, int32_t B, int32_t H, int32_t W,
                       int32_t Cout, lic_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * lic_plan -- one training step (Trainer.py:75-90: forward, loss, backward) replayed on up to THREE HIP streams
 *   without Python in the loop.  The step is captured ONCE as a HIP graph (torch.cuda.graph with keep_graph=True:
 *   torch's private pool keeps every pointer fixed); lic_plan_create reads the graph's kernel / memset / memcpy /
 *   empty nodes and its edges back into a list of operations, and a SCHEDULE puts them in an issue order, each on one
 *   of the streams, with one event record + wait per cross-stream edge that stream order does not already imply;
 *   lic_plan_replay issues the schedule with plain stream launches (this ROCm walks a multi-branch graph node by node
 *   from the host: 8 ms per step instead of 0.1 ms, and a single-branch graph gives up the decoder / latent-side and
 *   data-gradient / weight-gradient overlap).  Every schedule honours every edge of the capture, so a replay computes
 *   what the captured step computed, bit for bit.
 *   `hip_graph` is a hipGraph_t and must outlive the plan (kernel arguments stay in the graph's storage).
 *   lic_plan_replay(plan, main, sides, n_sides): work queued on `main` before the call precedes the plan, work queued
 *   on it afterwards follows ALL of the plan; `sides` = up to two more, otherwise idle streams of the same device
 *   (fewer than the schedule uses: the missing ones fold onto the last one given; n_sides = 0: one stream).
 *   info[0..6] = operations, kernels, memsets, memcpys, operations not on `main`, cross-stream events, 1 when a
 *   tuned schedule is in use.
 *   lic_plan_create's schedule is the capture's own: the capture's streams are chains of the graph, and where a chain
 *   forks the successor with the longest way to go stays on the stream.  lic_plan_tune(plan, main, sides, n_sides,
 *   result) times every operation (one-stream replays with an event between operations), list-schedules them
 *   (earliest possible start first, the more critical operation on ties; long kernels of >= 192 workgroups take turns),
 *   measures capture order and three candidates end to end and keeps a tuned one only if it is >= 2 % faster;
 *   result[0..3] = sum of operation times, capture-order step, best tuned step (all us), 1 if kept.  It replays the
 *   plan a few dozen times: the capture must be idempotent (no in-place update of its own inputs -- keep the
 *   optimizer outside).
 *   Errors: LIC_ERR_UNSUPPORTED for a node kind that cannot be replayed (host, event, child-graph, allocation nodes;
 *   1-D copy nodes, whose parameters this ROCm does not hand back); lic_plan_last_error() names it.
 * ------------------------------------------------------------------------------------------ */
typedef struct lic_plan lic_plan;
int lic_plan_create(void* hip_graph, lic_plan** out);
int lic_plan_info(const lic_plan* plan, int64_t* info);
int lic_plan_replay(lic_plan* plan, lic_stream_t main, const lic_stream_t* sides, int32_t n_sides);
int lic_plan_tune(lic_plan* plan, lic_stream_t main, const lic_stream_t* sides, int32_t n_sides, double* result);
void lic_plan_destroy(lic_plan* plan);
const char* lic_plan_last_error(void);

int lic_version(void);        /* LIC_ABI_VERSION */
int lic_last_hip_error(void); /* hipError_t of the most recent failed launch on this thread */
const char* lic_arch(void);   /* "gfx950" */

#ifdef __cplusplus
}
#endif
#endif /* LIC_H */
