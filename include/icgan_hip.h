/*
 * icgan_hip.h — C ABI of libicgan_hip.so: the MI355X (gfx950) kernels behind the
 * IC-GAN G+D forward/backward hot path.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocations);
 *     the library never allocates, never synchronises, keeps no mutable global state
 *   - activations are fp32 NHWC:  x[b][h][w][c]
 *   - `stream` is a hipStream_t (the caller's current stream)
 *   - return 0 on success, a negative ICG_ERR_* otherwise; icg_strerror() explains
 *   - the reference file:line each entry point replaces is cited next to it
 *     (paths relative to the reference repository root)
 */
#ifndef ICGAN_HIP_H
#define ICGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICG_OK 0
#define ICG_ERR_ARG (-1)      /* bad argument (null pointer, unsupported shape) */
#define ICG_ERR_LAUNCH (-2)   /* hipLaunch failed, see icg_last_hip_error() */
#define ICG_ERR_WORKSPACE (-3)/* workspace too small */

const char* icg_strerror(int code);
/* Library state, all of it: (1) this per-thread error slot, (2) the per-thread label of the last GEMM launch
 * (icg_gemm_last_variant) and (3) the opt-in, mutex-guarded timing records of icg_planes_timing -- (2) and (3) are read only
 * by bench.py.  No entry point's RESULT depends on any of them; everything else is passed in (buffers, workspaces, stream). */
/* the hipError_t behind the last ICG_ERR_LAUNCH returned to the CALLING THREAD (thread-local) */
int icg_last_hip_error(void);
int icg_version(void);

/* ---- flags for the fused convolution ---------------------------------- */
#define ICG_PRE_RELU 1u       /* a = max(a, 0) after the optional affine            */
#define ICG_PRE_AFFINE 2u     /* a = x*scale[b][c] + shift[b][c] (BN / ccbn apply)  */
#define ICG_UPSAMPLE2X 4u     /* conv input is nearest-upsampled x2 on read         */
#define ICG_RES_UPSAMPLE2X 8u /* residual is at half resolution, upsampled on read  */
#define ICG_RES_RELU_MASK 16u /* `residual` is NOT added: it is the input r of the ReLU in front of the layer whose data gradient
                               * this call computes, and out = (r > 0) ? value : 0 -- autograd's ReLU backward of
                               * DBlock.forward (`self.activation(x)`, layers.py:587-600) folded into the epilogue of the
                               * data-gradient convolution.  Accepted by icg_conv2d_fprop[_ws], icg_conv2d_wino_fprop and
                               * icg_conv2d_wino4_fprop; excludes ICG_RES_UPSAMPLE2X. */
#define ICG_WINO_KEEP_V 32u   /* F(4x4,3x3) forward entries (icg_conv2d_wino4_fprop, icg_conv2d_{up,down}_wino_fprop): the caller will read
                               * the transformed-input planes V back from the start of `workspace` (icg_conv2d_wino4_wgrad_from_v).  The
                               * three-kernel composite leaves them there anyway; the fused kernel (icg_fwino_applies) writes them only
                               * when this flag is set -- V is its by-product, not an intermediate. */

/*
 * Fused implicit-GEMM convolution, stride 1, pad R/2, R in {1,3}; also every Linear
 * (H = W = 1, R = 1, B = rows).  fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32.
 *
 *   out[b,h,w,co] = alpha * sum_{r,s,ci} act(x)[b, h+r-p, w+s-p, ci] * w[co][r][s][ci]
 *                   + bias[co] + residual[b,h,w,co]
 *   act(x) = relu?( x*scale[b][ci] + shift[b][ci] )?   (zero padding applies AFTER act)
 *
 * H, W are the OUTPUT spatial dims; with ICG_UPSAMPLE2X x is [B][H/2][W/2][Cin].
 * w is OHWI ([Cout][R][R][Cin]).  scale/shift are [ss_rows][Cin], ss_rows in {1, B}
 * (ss_bstride = 0 or Cin).
 * Data-gradient = the same call with dy as x, w = [Cin][R][R][Cout] tap-flipped, Cin<->Cout.
 *
 * Replaces: F.conv2d in SNConv2d.forward (BigGAN_PyTorch/layers.py:144-153), F.linear in
 * SNLinear.forward (layers.py:164-165), the BN apply + ReLU + F.interpolate + residual add of
 * GBlock.forward (layers.py:542-552) and the ReLU / residual add of DBlock.forward (587-613).
 */
int icg_conv2d_fprop(const float* x, const float* w, const float* bias, const float* residual,
                     float* out, const float* scale, const float* shift, int64_t ss_bstride,
                     int B, int H, int W, int Cin, int Cout, int R, unsigned flags, float alpha,
                     void* stream);

/*
 * Weight gradient of the same fused convolution (the activation prologue is re-applied to x):
 *   dw[r][s][ci][co] = sum_{b,h,w} act(x)[b,h+r-p,w+s-p,ci] * dy[b,h,w,co]        (HWIO layout)
 * Split-K over pixels with a deterministic two-stage reduction through `workspace`.
 * Replaces the weight-gradient half of autograd's ConvolutionBackward for layers.py:144-153.
 */
/*
 * The same with an optional split-K workspace: launches whose [pixels x Cout] output has too few 128x128 tiles to fill
 * 256 CUs (small batches, low resolutions) cut K into slices that run concurrently and are summed deterministically.
 * icg_conv2d_fprop_workspace_bytes returns 0 when split-K does not apply; workspace NULL / too small = single pass.
 */
size_t icg_conv2d_fprop_workspace_bytes(int B, int H, int W, int Cin, int Cout, int R, unsigned flags);
int icg_conv2d_fprop_ws(const float* x, const float* w, const float* bias, const float* residual, float* out,
                        const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin,
                        int Cout, int R, unsigned flags, float alpha, void* workspace, size_t workspace_bytes,
                        void* stream);
size_t icg_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int R);
int icg_conv2d_wgrad(const float* x, const float* dy, float* dw, const float* scale,
                     const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin, int Cout,
                     int R, unsigned flags, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Nearest-x2-upsample-fused 3x3 convolution in PHASE form (GBlock conv1: F.interpolate(scale_factor=2) followed by
 * SNConv2d, layers.py:545-548): each of the 4 output phases (al, be) is a 2x2-tap convolution of the SOURCE-resolution
 * tensor, 2.25x fewer multiply-adds than convolving the upsampled tensor.  x is [B][Hs][Ws][Cin]; out / dy are
 * [B][2Hs][2Ws][Cout].  wp = [4][Cout][2][2][Cin] and vd = [Cin][4][4][Cout] come from icg_sn_forward; the weight
 * gradient dwp = [4][2][2][Cin][Cout] goes to icg_sn_backward (dw_up), which folds it onto the 3x3 parameter.
 * up_dgrad returns the gradient w.r.t. act(x) at SOURCE resolution (upsample adjoint included).
 */
int icg_conv2d_up_fprop(const float* x, const float* wp, const float* bias, float* out, const float* scale,
                        const float* shift, int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout,
                        unsigned flags, void* stream);
int icg_conv2d_up_dgrad(const float* dy, const float* vd, float* da, int B, int Hs, int Ws, int Cin, int Cout,
                        void* stream);
size_t icg_conv2d_up_wgrad_workspace_bytes(int B, int Hs, int Ws, int Cin, int Cout);
int icg_conv2d_up_wgrad(const float* x, const float* dy, float* dwp, const float* scale, const float* shift,
                        int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout, unsigned flags,
                        void* workspace, size_t workspace_bytes, void* stream);

/*
 * 3x3 convolution followed by 2x2 average pooling (DBlock conv2 + nn.AvgPool2d(2): layers.py:603-606, BigGAN.py:528) as
 * ONE 4x4 / stride-2 convolution at the pooled resolution (2.25x fewer multiply-adds; the full-resolution conv output
 * never exists).  x is [B][2Hp][2Wp][Cin], out / dy / residual are [B][Hp][Wp][Cout].  vdn = [Cout][4][4][Cin] and
 * wq = [4][Cin][2][2][Cout] come from icg_sn_forward; dvdn = [4][4][Cin][Cout] goes to icg_sn_backward (dw_down).
 * Only ICG_PRE_RELU is accepted in flags.  down_dgrad returns the gradient w.r.t. act(x) at full resolution.
 */
int icg_conv2d_down_fprop(const float* x, const float* vdn, const float* bias, const float* residual, float* out,
                          int B, int Hp, int Wp, int Cin, int Cout, unsigned flags, void* stream);
int icg_conv2d_down_dgrad(const float* dy, const float* wq, float* da, int B, int Hp, int Wp, int Cin, int Cout,
                          void* stream);
/* down_dgrad with the ReLU backward of the layer's prologue in the epilogue: dx = (relu_in > 0) ? da : 0, relu_in = the
 * layer's input x [B][2Hp][2Wp][Cin] (see ICG_RES_RELU_MASK) */
int icg_conv2d_down_dgrad_relu(const float* dy, const float* wq, const float* relu_in, float* dx, int B, int Hp, int Wp,
                               int Cin, int Cout, void* stream);
size_t icg_conv2d_down_wgrad_workspace_bytes(int B, int Hp, int Wp, int Cin, int Cout);
int icg_conv2d_down_wgrad(const float* x, const float* dy, float* dvdn, int B, int Hp, int Wp, int Cin, int Cout,
                          unsigned flags, void* workspace, size_t workspace_bytes, void* stream);

/*
 * General strided / zero-inserted convolution, NHWC fp32 — the contraction behind StyleGAN2's
 * conv2d_gradfix.conv2d / conv_transpose2d (stylegan2_ada_pytorch/torch_utils/ops/conv2d_gradfix.py:43-99), their
 * data / weight gradients (conv2d_gradfix.py:139-272) and, through those, conv2d_resample (conv2d_resample.py:79-216)
 * and modulated_conv2d (training/networks.py:37-117).  Replaces cudnn_convolution(_transpose) and
 * cudnn_convolution_backward_weight.
 *
 *   out[b,oy,ox,co] = bias[co] + sum_{r,s,ci} src(b, oy*stride + r - pad, ox*stride + s - pad, ci) * w[co][r][s][ci]
 *   zero_insert = 0: src = x (zero outside [0,Hin) x [0,Win))               -> F.conv2d(stride, padding=pad)
 *   zero_insert = z: src = x zero-inserted by z (extent (Hin-1)*z+1), stride must be 1
 *                    -> F.conv_transpose2d(stride=z, padding=R-1-pad) with w = flipped, transposed weight
 * Hout/Wout are free (positions beyond the source read zeros): output_padding needs no separate argument.
 * w is [Cout][R][R][Cin]; bias may be NULL.
 */
int icg_conv2d_g_fprop(const float* x, const float* w, const float* bias, float* out, int B, int Hin, int Win,
                       int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zero_insert,
                       void* stream);
size_t icg_conv2d_g_fprop_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int R, int zero_insert);
int icg_conv2d_g_fprop_ws(const float* x, const float* w, const float* bias, float* out, int B, int Hin, int Win,
                          int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zero_insert,
                          void* workspace, size_t workspace_bytes, void* stream);
/*
 * The zero_insert = 2, R = 3, pad = 2 case of the above (conv_transpose2d(stride=2, padding=0) of the up-sampling
 * synthesis layers, conv2d_resample.py:163-186, and the data gradient of the discriminator's stride-2 convolutions) in
 * phase form: 4 launches-in-one of 2x2 taps at source resolution, 16 instead of 36 tap slots per 4 outputs.
 *   wp[al][be][co][u][v][ci]  (u, v in {0,1}; even coordinate: taps {0,2}, odd: {1, zero})
 *   out[b, 2m+al, 2n+be, co] = bias[co] + sum x[b, m-1+al+u, n-1+be+v, ci] * wp[al][be][co][u][v][ci],  rows/cols >= Hout/Wout dropped
 */
int icg_conv2d_tr2_fprop(const float* x, const float* wp, const float* bias, float* out, int B, int Hin, int Win,
                         int Cin, int Hout, int Wout, int Cout, void* stream);
/* icg_conv2d_g_wgrad (below) on fp16 operands (x, dy fp16; dw fp32 HWIO): v_mfma_f32_16x16x32_f16 with fp32 accumulation, split-K over
 * pixel slices into fp32 slabs in the caller's workspace + a fixed-order reduction -- cudnn_convolution_backward_weight on the fp16
 * blocks' tensors (conv2d_gradfix.py:139-272).  `_applies` -> 1 when Cin and Cout are multiples of 32. */
int icg_conv2d_g_wgrad_f16_applies(int Cin, int Cout, int R, int stride);
size_t icg_conv2d_g_wgrad_f16_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int R);
int icg_conv2d_g_wgrad_f16(const void* x, const void* dy, float* dw, int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout,
                           int R, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream);
/*
 * The same gather with fp16 storage and fp16-input MFMA (v_mfma_f32_16x16x32_f16: exact fp16 products, fp32 accumulation, one
 * rounding to fp16) -- the arithmetic of the reference's fp16 blocks, which cast activations AND weights to fp16 and convolve in
 * fp16 (training/networks.py:77-91, 581-601; `num_fp16_res`).  x [B][Hin][Win][Cin], w [Cout][R][R][Cin], out [B][Hout][Wout][Cout],
 * all fp16; no bias (bias_act applies it).  zero_insert in {0, 2}.  `_applies` -> 1 when the shape is served (Cin a multiple of
 * 32, Cout of 64 or 96); the caller keeps icg_conv2d_g_fprop + two casts for the rest (3-channel toRGB / fromRGB layers).
 */
int icg_conv2d_g_fprop_f16_applies(int Cin, int Cout, int R, int stride, int zero_insert);
int icg_conv2d_g_fprop_f16(const void* x, const void* w, void* out, int B, int Hin, int Win, int Cin, int Hout, int Wout,
                           int Cout, int R, int stride, int pad, int zero_insert, void* stream);
/* The same convolution (zero_insert = 0) with the StyleGAN2 layer epilogue applied to the accumulators before they leave the registers:
 *   c = fp16(acc)                                             stored when c != NULL (the demodulation gradient reads it)
 *   y = clamp(gain * act(c * d[b][n] + noise[b * noise_bstride + p] * strength[0] + bias[n]))     act 1 linear / 3 lrelu(alpha)
 * with the fp16 rounding points of the reference's separate operations (networks.py:86-94 fma, 432-442 bias_act) -- the results of
 * icg_conv2d_g_fprop_f16 followed by icg_sg2_act_fwd, in one pass.  d [B][Cout], bias [Cout] fp32 (16-byte aligned), noise fp32; each
 * may be NULL. */
int icg_conv2d_g_fprop_f16_act(const void* x, const void* w, void* c, void* y, const float* d, const float* noise, int64_t noise_bstride,
                               const float* strength, const float* bias, int act, float alpha, float gain, float clamp, int B, int Hin,
                               int Win, int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, void* stream);
/* modulated_conv2d (networks.py:37-117, the training form) in ONE launch: the style scale x * s (style [B][Cin] fp32, may be NULL) is
 * applied to the A fragments after their LDS read -- fp16(x * fp16(s)), the rounding of the reference's tensor, which is never
 * materialised -- then the contraction, then (y != NULL; zero_insert = 0 only) the epilogue of icg_conv2d_g_fprop_f16_act.  y == NULL:
 * the plain (modulated) convolution into c, zero_insert 0 or 2.  `_applies`: the fp16 kernel takes the shape, Cin <= 1024, and every
 * phase has >= 127 pixels per sample (a 128-pixel tile then meets at most two samples' styles, which it keeps in LDS). */
int icg_modconv2d_f16_applies(int Cin, int Cout, int R, int stride, int zero_insert, int Hout, int Wout);
int icg_modconv2d_f16(const void* x, const float* style, const void* w, void* c, void* y, const float* d, const float* noise,
                      int64_t noise_bstride, const float* strength, const float* bias, int act, float alpha, float gain, float clamp,
                      int B, int Hin, int Win, int Cin, int Hout, int Wout, int Cout, int R, int stride, int pad, int zero_insert,
                      void* stream);
/*   dw[r][s][ci][co] = sum_{b,oy,ox} x[b, oy*stride + r - pad, ox*stride + s - pad, ci] * dy[b,oy,ox,co]
 * (weight gradient of either direction: for the transposed convolution swap the roles of x and dy). */
size_t icg_conv2d_g_wgrad_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout, int R);
int icg_conv2d_g_wgrad(const float* x, const float* dy, float* dw, int B, int Hin, int Win, int Cin, int Hout,
                       int Wout, int Cout, int R, int stride, int pad, void* workspace, size_t workspace_bytes,
                       void* stream);

/*
 * Winograd F(2x2, 3x3) form of the stride-1 3x3 convolution (forward; data gradient with the dgrad-layout weight): 16/36 of the
 * multiply-adds of icg_conv2d_fprop at the same result up to fp32 rounding (~1e-6); pays for wide layers (Cin, Cout >= 256).
 *   icg_wino_weight_transform : w [N][3][3][K] (OHWI or dgrad layout)  ->  U [16][N][K] = G g G^T
 *   icg_conv2d_wino_fprop     : out = alpha * conv3x3(act(x), w) + bias + residual;  flags: ICG_PRE_RELU, ICG_PRE_AFFINE,
 *                               ICG_RES_UPSAMPLE2X;  H, W even, Cin % 4 == 0, Cout % 4 == 0;  workspace holds the transformed
 *                               input and the 16 GEMM outputs (icg_conv2d_wino_workspace_bytes)
 */
int icg_wino_weight_transform(const float* w, float* U, int N, int K, void* stream);
size_t icg_conv2d_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_conv2d_wino_fprop(const float* x, const float* U, const float* bias, const float* residual, float* out,
                          const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin,
                          int Cout, unsigned flags, float alpha, void* workspace, size_t workspace_bytes, void* stream);

/* F(4x4, 3x3) variant: 36 GEMMs over 1/16 of the pixels = 1/4 of the direct multiply-adds, transforms over 2.25x the activation
 * volume; H, W multiples of 4; agreement with the direct kernel ~1e-5 relative (interpolation points 0, +-1, +-2, inf) */
int icg_wino4_weight_transform(const float* w, float* U, int N, int K, void* stream);     /* U [36][N][K] */
size_t icg_conv2d_wino4_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_conv2d_wino4_fprop(const float* x, const float* U, const float* bias, const float* residual, float* out,
                           const float* scale, const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin,
                           int Cout, unsigned flags, float alpha, void* workspace, size_t workspace_bytes, void* stream);
/* weight gradient of the same layer through the Winograd domain (HWIO output like icg_conv2d_wgrad, 16/36 of its MACs):
 * dU[xi] = V[xi]^T (A dy A^T)[xi] as 16 long-K GEMMs (icg_gemm_tn_batched), dw = G^T dU G */
size_t icg_conv2d_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, const float* scale, const float* shift,
                          int64_t ss_bstride, int B, int H, int W, int Cin, int Cout, unsigned flags, void* workspace,
                          size_t workspace_bytes, void* stream);
/* the same through the F(4x4,3x3) domain (H, W multiples of 4): 36 GEMMs, 9/36 of the direct MACs, both transform passes over
 * 2.25x the activation volume; dw = G^T [ sum_tiles (A dy A^T) .* (B^T act(x) B) ] G with the F(4x4,3x3) matrices */
size_t icg_conv2d_wino4_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_conv2d_wino4_wgrad(const float* x, const float* dy, float* dw, const float* scale, const float* shift,
                           int64_t ss_bstride, int B, int H, int W, int Cin, int Cout, unsigned flags, void* workspace,
                           size_t workspace_bytes, void* stream);
/*
 * The resample-fused layers in the F(4x4,3x3) Winograd domain — same contracts as icg_conv2d_up_* (GBlock conv1: nearest x2
 * upsample -> SNConv2d 3x3, layers.py:545-548) and icg_conv2d_down_* (DBlock conv2 -> nn.AvgPool2d(2), layers.py:603-606,
 * BigGAN.py:528), for wide layers.  The 6-pixel window of an upsampled signal [l0 l1 l1 l2 l2 l3] has a vanishing third
 * transform component and the 2-pixel sums of the output transform do not read it, so only 25 of the 36 per-tile GEMMs
 * remain: 25/64 of the multiply-adds of the 2x2-phase / 4x4-stride-2 forms.  U [25][N][K] comes from
 * icg_wino4r_weight_transform applied to the PLAIN 3x3 layouts of icg_sn_forward (w_ohwi for fprop, w_dgrad for the data
 * gradients); the weight gradients are plain HWIO [3][3][Cin][Cout] (icg_sn_backward's dw_hwio).  Hs/Ws: source resolution
 * of the upsampling layer, Hp/Wp: pooled resolution of the downsampling layer; both must be even, Cin and Cout multiples
 * of 4.  Workspace queries take the FULL resolution (2Hs x 2Ws, 2Hp x 2Wp) and the (Cin, Cout) of the GEMM as called.
 */
int icg_wino4r_weight_transform(const float* w, float* U, int N, int K, void* stream);
/* The Winograd-domain copies of many weights in one launch (planes = 16: icg_wino_weight_transform, 36: icg_wino4_..., 25:
 * icg_wino4r_...; bit-identical to the single-tensor calls): a network's layers after their batched spectral-norm pass. */
typedef struct {
  const float* w;   /* [N][3][3][K] (OHWI or the data-gradient layout) */
  float* U;         /* [planes][N][K] */
  int N, K, planes, reserved;
} icg_wino_weight;
int icg_wino_weight_transform_multi(const icg_wino_weight* items, int n, void* stream);    /* U [25][N][K] */
size_t icg_conv2d_rs_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout);
size_t icg_conv2d_rs_wino_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int icg_conv2d_up_wino_fprop(const float* x, const float* U, const float* bias, float* out, const float* scale,
                             const float* shift, int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout,
                             unsigned flags, void* workspace, size_t workspace_bytes, void* stream);
int icg_conv2d_up_wino_dgrad(const float* dy, const float* U, float* da, int B, int Hs, int Ws, int Cin, int Cout,
                             void* workspace, size_t workspace_bytes, void* stream);
int icg_conv2d_up_wino_wgrad(const float* x, const float* dy, float* dw, const float* scale, const float* shift,
                             int64_t ss_bstride, int B, int Hs, int Ws, int Cin, int Cout, unsigned flags,
                             void* workspace, size_t workspace_bytes, void* stream);
int icg_conv2d_down_wino_fprop(const float* x, const float* U, const float* bias, const float* residual, float* out,
                               int B, int Hp, int Wp, int Cin, int Cout, unsigned flags, void* workspace,
                               size_t workspace_bytes, void* stream);
int icg_conv2d_down_wino_dgrad(const float* dy, const float* U, float* da, int B, int Hp, int Wp, int Cin, int Cout,
                               void* workspace, size_t workspace_bytes, void* stream);
/* ... with the ReLU backward of the layer's prologue in the output transform (see icg_conv2d_down_dgrad_relu) */
int icg_conv2d_down_wino_dgrad_relu(const float* dy, const float* U, const float* relu_in, float* dx, int B, int Hp, int Wp,
                                    int Cin, int Cout, void* workspace, size_t workspace_bytes, void* stream);
int icg_conv2d_down_wino_wgrad(const float* x, const float* dy, float* dw, int B, int Hp, int Wp, int Cin, int Cout,
                               unsigned flags, void* workspace, size_t workspace_bytes, void* stream);
/*
 * Fused form of the F(4x4,3x3) forward entries above for the NARROW layers (csrc/fwino.hip): input transform, the 36 / 25 plane
 * GEMMs, output transform and epilogue in one kernel -- neither V nor M goes through HBM (at Cin = 96 the plane GEMM sits on
 * the HBM ridge: 24 FLOP per byte of V + M).  icg_conv2d_wino4_fprop, icg_conv2d_{up,down}_wino_{fprop,dgrad[_relu]} take it on their own
 * when icg_fwino_applies says 1 (Cin % 32 == 0, Cin <= 192, Cout % 96 == 0, Cout <= 192, H and W of the Winograd domain multiples of
 * 16, >= 512 workgroups; ICG_FWINO=0 in the environment switches it off): same arguments, same workspace, same result up
 * to fp32 summation order (single-level chains over K <= 192).  V is written only under ICG_WINO_KEEP_V.
 *   icg_fwino_pack_weights: U [planes][Cout][Cin] (icg_wino4_weight_transform / icg_wino4r_weight_transform) -> the
 *   fragment-major copy the kernel streams (every wave-load 1 KiB contiguous); the entries above do it into their workspace.
 */
int icg_fwino_applies(int B, int H, int W, int Cin, int Cout);
size_t icg_fwino_weight_bytes(int planes, int Cin, int Cout);
int icg_fwino_pack_weights(const float* U, float* Uf, int planes, int Cin, int Cout, void* stream);
/* the kernel as an entry point of its own (tests, microbenchmarks): H, W = resolution of the Winograd domain (multiples of 16);
 * (in_up, out_pool) = (0,0): plain 3x3, 36 planes; (1,0): x is [B][H/2][W/2][Cin], nearest x2 on read, 25 planes; (0,1): out is
 * [B][H/2][W/2][Cout] = 2x2 sums x alpha, 25 planes.  flags: ICG_PRE_*, ICG_RES_*.  V: NULL or [planes][B H/4 W/4][Cin]. */
int icg_fwino_conv(const float* x, const float* Uf, const float* bias, const float* residual, float* out, const float* scale,
                   const float* shift, int64_t ss_bstride, int B, int H, int W, int Cin, int Cout, unsigned flags, float alpha,
                   int in_up, int out_pool, float* V, void* stream);
/* Weight gradient from the V planes the FORWARD pass of the same layer left in its workspace: icg_conv2d_wino4_fprop,
 * icg_conv2d_up_wino_fprop and icg_conv2d_down_wino_fprop write V = transform(act(x)) to the first
 * planes * T * Cin floats of the workspace (T = B * H/4 * W/4 at the FULL resolution H x W; planes = 36 / 25).  A caller that
 * keeps that region until the backward pass (memory for HBM passes: sized for 288 GB) skips the input transform.  dy_up = 1,
 * dy_alpha = 0.25 for the avgpool-fused layer (dy at the pooled resolution), else 0 / 1.  Same result as the *_wgrad entries. */
size_t icg_conv2d_wino4_wgrad_from_v_workspace_bytes(int B, int H, int W, int Cin, int Cout, int planes);
int icg_conv2d_wino4_wgrad_from_v(const float* V, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                  int planes, int dy_up, float dy_alpha, void* workspace, size_t workspace_bytes, void* stream);
/* The same with dbias [Cout] = column sums of dy over all of its pixels (the gradient of the layer's bias, layers.py:144-153
 * `F.conv2d(x, W, bias)`), produced by the dy-transform pass that reads dy anyway; deterministic two-stage reduction. */
size_t icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes(int B, int H, int W, int Cin, int Cout, int planes);
int icg_conv2d_wino4_wgrad_from_v_db(const float* V, const float* dy, float* dw, float* dbias, int B, int H, int W, int Cin,
                                     int Cout, int planes, int dy_up, float dy_alpha, void* workspace, size_t workspace_bytes,
                                     void* stream);
/* measurement hook (bench.py): with timing enabled every batched GEMM over Winograd planes (rocprofv3 name
 * icg_gemm_planes_kernel<AMODE, BMODE, TN>) is bracketed by HIP events on its launch stream.  drain() writes rows of
 * {amode, tn (+ 10 when the single-level kernel icg_gemm_planes1_kernel ran), planes, launches, total ms, total executed
 * flops, total operand bytes} and returns the row count.
 * enable = P > 0: every launch is counted per distinct shape (M, N, K, planes, NN / TN); every P-th launch of a shape is bracketed
 * (P = 1: all); "total ms" = sum over shapes of (mean bracketed launch) x launches -- a shape's launches do identical work.
 * Disabled (the default) the product path pays one relaxed atomic load per call; enabled, records are appended under a mutex. */
int icg_planes_timing(int enable);
int icg_planes_timing_drain(double* out, int max_rows);
/* C[b] = A[b]^T B[b], A [K][M], B [K][N], long K: batched with deterministic split-K (strideC must be M*N) */
size_t icg_gemm_tn_batched_workspace_bytes(int M, int N, int K, int batch);
int icg_gemm_tn_batched(const float* A, const float* B, float* C, int M, int N, int K, int64_t strideA, int64_t strideB,
                        int64_t strideC, int batch, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Batched fp32 GEMM  C[z] = alpha * op(A[z]) * op(B[z]) for the attention
 * contractions (layers.py:237-243: theta^T phi, g beta^T and their gradients).
 *   transA = 0: A is [M][K] row-major;  1: A is [K][M]
 *   transB = 0: B is [K][N] row-major;  1: B is [N][K]   (supported: (0,1) (0,0) (1,0))
 */
int icg_gemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int transA,
                     int transB, int64_t strideA, int64_t strideB, int64_t strideC, int batch,
                     float alpha, void* stream);

/*
 * The batched GEMM over Winograd planes on its own: C[z] = alpha * A[z] B[z]^T, A [planes][M][K] (tiles x channels),
 * B [planes][N][K] (Winograd-domain weights), C [planes][M][N], all dense -- launched exactly as the Winograd composites
 * (icg_conv2d_wino4_fprop, icg_conv2d_up_wino_fprop, ...) launch it, i.e. on the second-generation plane GEMM of
 * csrc/pgemm.hip where that kernel has a tile (N a multiple of 128 or 96, K a multiple of 32) and on the first-generation
 * kernels otherwise.  No reference counterpart (the reference calls cuDNN, layers.py:144-153); entry point of the kernel-level
 * parity tests and of tools/pgemm_bench.py.
 */
int icg_plane_gemm(const float* A, const float* B, float* C, int M, int N, int K, int planes, float alpha, void* stream);

/*
 * ... and its weight-gradient counterpart: C[z] = A[z]^T B[z], A [planes][K][M] (V planes: tiles x Cin), B [planes][K][N]
 * (transformed dy: tiles x Cout), C [planes][M][N]; K = tiles is long and is cut into slices whose partial slabs (workspace) are
 * summed in fixed order.  Launched as icg_conv2d_wino4_wgrad* launch it (second-generation kernel where M % 4 == 0, K % 32 == 0
 * and N is a multiple of 128 or 96).
 */
size_t icg_plane_gemm_tn_workspace_bytes(int M, int N, int K, int planes);
int icg_plane_gemm_tn(const float* A, const float* B, float* C, int M, int N, int K, int planes, void* workspace,
                      size_t workspace_bytes, void* stream);

/*
 * Measurement support (no reference counterpart): template arguments {AMODE, BMODE, TN, PATH} of the last
 * icg_gemm_kernel<AMODE, BMODE, TN, PATH> launched by the calling host thread through any of the conv / GEMM
 * entry points above ({-1,...} before the first launch; {-2, 0 fprop / 1 wgrad, Cout, Cin} when the direct
 * narrow-output kernels of narrow_conv.hip ran instead).  bench.py uses it to label its HIP-event timings with
 * exactly the kernel name rocprofv3 reports.
 */
int icg_gemm_last_variant(int* out4);

/* ---- BatchNorm / ccbn statistics  (layers.py:398-437 ccbn.forward, 485-503 bn.forward,
 *      sync_batchnorm/batchnorm.py:61-193 for the cross-replica variant) ---------------- */
/* number of float partial slots per channel pair produced by icg_bn_partial_stats */
size_t icg_bn_workspace_bytes(int64_t rows, int C);
/* stage 1: per-chunk shifted sums  S1 = sum(x-k[c]), S2 = sum((x-k[c])^2) over `rows` x C */
int icg_bn_partial_stats(const float* x, const float* shift_k, int64_t rows, int C, void* workspace,
                         size_t workspace_bytes, void* stream);
/* stage 2: reduce the partials in fp64 to sums[2][C] (double); all-reduce these for SyncBN */
int icg_bn_reduce_partials(const void* workspace, int64_t rows, int C, double* sums, void* stream);
/*
 * stage 3: finalize. count = total rows over all replicas.  training != 0: mean/var from sums,
 * running stats updated in place (momentum, unbiased var); training == 0: use running stats.
 * Writes mean[C], invstd[C] and the fused per-sample affine
 *   scale[b][c] = invstd[c]*gain[b][c],  shift[b][c] = bias[b][c] - mean[c]*scale[b][c]
 * gain/bias (and scale/shift) are [gb_rows][C] (gb_rows in {1,B}); gain_offset is added to gain
 * (ccbn: 1 + gain(y)).  shift_k may alias running_mean.
 */
/* Cross-replica BN (the all-reduce semantics of sync_batchnorm/batchnorm.py:148-193 re-expressed for one process per GPU):
 * payload double[2*C + 1] = [sum x | sum x^2 | n], un-shifted (common origin) so that replicas with different running
 * means can be summed; all-reduce(sum) it over the replicas, then call icg_bn_finalize(payload, NULL, 0.0, ...):
 * count <= 0 means "the element count is sums[2*C]" (stays on the device: no host sync, unequal per-replica batches allowed).
 * The same convention holds for icg_bn_bwd_coefs (count <= 0: chan_sums[2*C]). */
int icg_bn_sync_pack(const double* sums, const float* shift_k, double local_count, int C, double* payload, void* stream);
int icg_bn_finalize(const double* sums, const float* shift_k, double count, float* running_mean,
                    float* running_var, float momentum, float eps, int training, const float* gain,
                    const float* bias, int gb_rows, float gain_offset, int C, float* mean,
                    float* invstd, float* scale, float* shift, void* stream);
/* icg_bn_reduce_partials + icg_bn_finalize of the training-mode forward (single replica) in ONE launch; element count = rows.
 * Bit-identical to the two calls (the fp64 channel sums are not written out: no caller of this form reads them). */
int icg_bn_reduce_finalize(const void* workspace, int64_t rows, int C, const float* shift_k, float* running_mean,
                           float* running_var, float momentum, float eps, const float* gain, const float* bias, int gb_rows,
                           float gain_offset, float* mean, float* invstd, float* scale, float* shift, void* stream);

/* stand-alone apply  y = relu?(x*scale[b][c] + shift[b][c])  (ccbn / bn used outside a fused block) */
int icg_bn_apply(const float* x, const float* scale, const float* shift, int64_t ss_bstride, int B, int64_t HW,
                 int C, unsigned flags, float* y, void* stream);

/*
 * Backward of  a = relu?(x*scale[b][c] + shift[b][c])  feeding a convolution (x is [B][Hs][Ws][C]; `da`, the
 * data gradient of the convolution input, is [B][2Hs][2Ws][C] when ICG_UPSAMPLE2X is set — the adjoint of the
 * nearest upsample (2x2 sum) is folded in — else [B][Hs][Ws][C]).  With  dy = (sum_2x2 da) * relu-mask:
 *
 *   stage 1  icg_bn_bwd_reduce       Sd[b][c] = sum_hw dy,  Sxc[b][c] = sum_hw dy*(x - mean[c])
 *   stage 2  icg_bn_bwd_channel_sums chan[0][c] = sum_b g[b][c]*Sd,  chan[1][c] = sum_b g[b][c]*invstd[c]*Sxc
 *            (g = gain_offset + gain;  double[2][C]; all-reduce(sum) these across replicas for SyncBN)
 *   stage 3  icg_bn_bwd_coefs        dgain = invstd*Sxc, dbias = Sd   ([gb_rows][C]; summed over b if gb_rows==1)
 *                                    coefA[c] = invstd*chan[0]/count, coefB[c] = invstd^2*chan[1]/count
 *                                    (both 0 when batch_stats == 0, i.e. eval-mode running statistics)
 *   stage 4  icg_bn_bwd_apply        dx = dy*scale[b][c] - coefA[c] - coefB[c]*(x - mean[c])
 *
 * Without ICG_PRE_AFFINE (plain ReLU in front of a D convolution, layers.py:587-604) only stage 4 is needed:
 * dx = da * (x > 0), coefA = coefB = NULL.
 */
size_t icg_bn_bwd_workspace_bytes(int B, int Hs, int Ws, int C);
int icg_bn_bwd_reduce(const float* x, const float* da, const float* scale, const float* shift,
                      int64_t ss_bstride, const float* mean, int B, int Hs, int Ws, int C, unsigned flags,
                      void* workspace, size_t workspace_bytes, float* sum_dy, float* sum_dyx, void* stream);
int icg_bn_bwd_channel_sums(const float* sum_dy, const float* sum_dyx, const float* gain, int gb_rows,
                            float gain_offset, const float* invstd, int B, int C, double* chan_sums,
                            void* stream);
int icg_bn_bwd_coefs(const float* sum_dy, const float* sum_dyx, const double* chan_sums, const float* invstd,
                     double count, int batch_stats, int gb_rows, int B, int C, float* dgain, float* dbias,
                     float* coefA, float* coefB, void* stream);
int icg_bn_bwd_apply(const float* x, const float* da, const float* scale, const float* shift,
                     int64_t ss_bstride, const float* mean, const float* coefA, const float* coefB, int B,
                     int Hs, int Ws, int C, unsigned flags, float* dx, void* stream);

/* ---- spectral norm  (layers.py:39-61 power_iteration, 98-112 SN.W_) ------------------ */
/*
 * One power-iteration step and the normalised weight for ONE layer.  w is the parameter in its
 * PyTorch layout [rows = Cout][Cin][R][R]; u is [rows] (updated in place when training);
 * sv (1 float, optional) receives sigma when training; v_out [Cin*R*R], u_out [rows] and
 * sigma_out[1] are saved for backward.  w_ohwi = w/sigma as [Cout][R][R][Cin]; w_dgrad (optional)
 * = w/sigma as [Cin][R][R][Cout] with taps flipped.  w_up_fprop / w_up_dgrad (optional, R = 3 only): the phase
 * layouts of icg_conv2d_up_*; w_down_fprop / w_down_dgrad likewise for icg_conv2d_down_*.
 * scratch: icg_sn_scratch_bytes().
 */
size_t icg_sn_scratch_bytes(int rows, int Cin, int R);
int icg_sn_forward(const float* w, float* u, float* sv, int rows, int Cin, int R, float eps,
                   int training, float* v_out, float* u_out, float* sigma_out, float* w_ohwi,
                   float* w_dgrad, float* w_up_fprop, float* w_up_dgrad, float* w_down_fprop,
                   float* w_down_dgrad, void* scratch, size_t scratch_bytes, void* stream);
/*
 * Backward of w_ = w/sigma with u,v constant:  dw = (dw_ - <dw_, w_> u^T v) / sigma.
 * dw_ is the sum of the given pieces: HWIO ([R][R][Cin][Cout]), OHWI, and the phase form dw_up
 * ([4][2][2][Cin][Cout], R = 3), the pooled 4x4 form dw_down ([4][4][Cin][Cout], R = 3) — each may be NULL; dw is written / accumulated (accumulate != 0) in
 * the parameter layout.
 */
/*
 * The same for many layers at once (all spectrally normalised layers of a network before its forward): one launch per
 * stage for up to ICG_SN_PACK layers instead of 5-6 launches per layer.  Bit-identical to icg_sn_forward per layer.
 */
#define ICG_SN_PACK 16
typedef struct {
  const float* w;
  float* u;
  float* sv;
  float* v_out;
  float* u_out;
  float* sigma_out;
  float* w_ohwi;
  float* w_dgrad;
  float* w_up_fprop;
  float* w_up_dgrad;
  float* w_down_fprop;
  float* w_down_dgrad;
  void* scratch;
  size_t scratch_bytes;
  int rows, Cin, R, reserved;
} icg_sn_layer;
int icg_sn_forward_multi(const icg_sn_layer* layers, int n, float eps, int training, void* stream);

/* scratch for icg_sn_backward: 256 doubles (minimum; uncoalesced gather) or this size (coalesced two-pass form) */
size_t icg_sn_backward_scratch_bytes(int rows, int Cin, int R);
int icg_sn_backward(const float* dw_hwio, const float* dw_ohwi, const float* dw_up, const float* dw_down,
                    const float* w_ohwi,
                    const float* u_saved, const float* v_saved, const float* sigma, int rows, int Cin,
                    int R, float* dw, int accumulate, void* scratch, size_t scratch_bytes, void* stream);

/* icg_sn_backward for many layers in two launches per ICG_SN_PACK layers (the per-layer form is two launches of a few microseconds,
 * 150 per training step): the fields are icg_sn_backward's arguments; `scratch` / `scratch_bytes` per layer as there.  Bit-identical
 * to the per-layer calls.  Called from the backward of the autograd node that groups the layers of a network
 * (ic_gan_amd/ops.py SNGroupFn; the reference differentiates SN.W_() per layer, layers.py:98-112). */
typedef struct icg_sn_bwd_item {
  const float* dw_hwio;
  const float* dw_ohwi;
  const float* dw_up;
  const float* dw_down;
  const float* w_ohwi;
  const float* u;
  const float* v;
  const float* sigma;
  float* dw;
  void* scratch;
  size_t scratch_bytes;
  int rows, Cin, R, accumulate;
} icg_sn_bwd_item;
int icg_sn_backward_multi(const icg_sn_bwd_item* items, int n, void* stream);

/* A group of dense layers over the same few rows in ONE launch per direction: the four conditional-BN projections of a GBlock
 * (bn1.gain / bn1.bias / bn2.gain / bn2.bias applied to the same y, reference layers.py:367-374; 64 rows, K = 657) otherwise cost four
 * latency-bound launches each way.  M rows and K inputs are common to the items; N = the item's output width.
 *   mode 0 (forward):         out_i [M][N_i] = x_i [M][K] * w_i^T,   w_i = W / sigma as [N_i][K]
 *   mode 1 (weight gradient): out_i [K][N_i] = x_i^T * dy_i          (HWIO at R = 1, what icg_sn_backward takes as dw_hwio); N_i % 4 == 0
 *   mode 2 (data gradient):   items[0].out [M][K] = sum_i dy_i [M][N_i] * w_i^T,  w_i = W / sigma as [K][N_i] (the dgrad layout); N_i % 4 == 0
 * Modes 0 / 1 are bit-identical to icg_conv2d_fprop / icg_conv2d_wgrad on the same operands; mode 2 sums the items in one chain. */
#define ICG_LINEAR_GROUP_MAX 8
typedef struct icg_linear_item {
  const float* x;
  const float* w;
  const float* dy;
  float* out;
  int N;
  int reserved;
} icg_linear_item;
int icg_linear_group(const icg_linear_item* items, int n, int M, int K, int mode, void* stream);

/* ---- pointwise / pooling / softmax ---------------------------------------------------- */
int icg_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, void* stream);
int icg_nhwc_to_nchw(const float* x, float* y, int B, int C, int H, int W, void* stream);
/* y = tanh(x) (BigGAN.py:386); dx = dy*(1-y^2) */
int icg_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int icg_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
/* 2x2 average pool (nn.AvgPool2d(2), BigGAN.py:528) with optional fused add: y = pool(x) + add */
int icg_avgpool2_fwd(const float* x, const float* add, float* y, int B, int H, int W, int C, void* stream);
/* y = sum over the 2x2 window (adjoint of the nearest x2 upsample of a residual branch) */
int icg_sumpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
/* dx = 0.25 * dy broadcast over the 2x2 window */
int icg_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);
/* ... plus a running gradient of the pooled tensor's source: dx = 0.25 * dy broadcast + carry (carry [B][H][W][C]; gradient chain
 * of the DBlock shortcut, ic_gan_amd/layers.py: the main path's gradient of x is added here instead of by an elementwise pass) */
int icg_avgpool2_bwd_add(const float* dy, const float* carry, float* dx, int B, int H, int W, int C, void* stream);
/* 2x2 max pool (F.max_pool2d, layers.py:230-231); backward routes to the first maximum */
int icg_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int icg_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream);
/* row softmax over the last dim (layers.py:237) */
int icg_softmax_fwd(const float* x, float* y, int64_t rows, int cols, void* stream);
/* Attention scores with the softmax in the epilogue (layers.py:233-238: beta = softmax(bmm(theta^T, phi), -1)):
 *   beta[b][i][:] = softmax_j( sum_k theta[b][i][k] * phi[b][j][k] ),  theta [B][n][d], phi [B][m][d], beta [B][n][m]
 * -- the scores are never written: one pass over beta instead of three (icg_gemm_batched + icg_softmax_fwd).  `_applies` -> 1 for
 * n % 32 == 0, m % 128 == 0, m <= 1024, d in {8, 16, 24, 32, 48, 64}; the caller keeps the two-kernel form otherwise. */
int icg_attn_scores_softmax_applies(int n, int m, int d);
int icg_attn_scores_softmax(const float* theta, const float* phi, float* beta, int B, int n, int m, int d, void* stream);
int icg_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int cols, void* stream);
/* Backward of the attention block's softmax with the dbeta GEMM in front of it (autograd of layers.py:237-243):
 *   dS[b][i][j] = beta[b][i][j] * (dbeta[b][i][j] - sum_j' beta[b][i][j'] dbeta[b][i][j']),  dbeta[b][i][j] = sum_c dO[b][i][c] V[b][j][c]
 * in one kernel -- dbeta [B][n][m] (1 - 2 GiB) is never written.  dO [B][n][dv], V = g [B][m][dv], beta / dS [B][n][m];
 * `_applies` -> 1 for n % 32 == 0, m % 128 == 0, m <= 1024, dv in {96, 192} (the two attention blocks of the 64 x 64 models). */
int icg_attn_dscores_applies(int n, int m, int dv);
int icg_attn_dscores(const float* dO, const float* V, const float* beta, float* dS, int B, int n, int m, int dv, void* stream);
/* The attention block's three input projections as ONE 1x1 convolution with stacked weights (theta, phi, g of layers.py:217-231 read
 * the same x): y [B][H][W][2 d + dv] = that convolution's output.  `icg_attn_split_pool` splits it into theta [B][H W][d] and the 2x2
 * max-pooled phi [B][H W / 4][d], g [B][H W / 4][dv] (F.max_pool2d, layers.py:230-231); `_bwd` assembles dy [B][H][W][2 d + dv] from
 * dtheta and the pooled gradients, routed to the first maximum of each window as icg_maxpool2_bwd does.  H, W even; d, dv % 4 == 0;
 * 16-byte aligned pointers. */
int icg_attn_split_pool(const float* y, float* theta, float* phi_p, float* g_p, int B, int H, int W, int d, int dv, void* stream);
int icg_attn_split_pool_bwd(const float* y, const float* dtheta, const float* dphi_p, const float* dg_p, float* dy, int B, int H, int W,
                            int d, int dv, void* stream);
/* gamma of the attention block folded into its output projection (layers.py:242-244: gamma * o(...) + x becomes ONE convolution with
 * the weight gamma * W / sigma and x as the residual operand): `icg_attn_gamma_scale` writes ws_a = gamma[0] * w_a (and ws_b = gamma[0] *
 * w_b when w_b is given: the data-gradient layout), n elements each, gamma on the device; `icg_attn_gamma_bwd` turns the gradient dws of
 * the SCALED weight into dw = gamma[0] * dws and dgamma[0] = sum(dws * w) (fp64 accumulation, one workgroup, deterministic), w = W / sigma
 * in the layout of dws. */
int icg_attn_gamma_scale(const float* gamma, const float* w_a, float* ws_a, const float* w_b, float* ws_b, int64_t n, void* stream);
int icg_attn_gamma_bwd(const float* gamma, const float* dws, const float* w, float* dw, float* dgamma, int64_t n, void* stream);
/* h[b][c] = sum_hw relu(x[b,h,w,c])   (BigGAN.py:625) and its backward */
int icg_relu_sumpool_fwd(const float* x, float* y, int B, int HW, int C, void* stream);
int icg_relu_sumpool_bwd(const float* x, const float* dy, float* dx, int B, int HW, int C, void* stream);
/* out = gamma[0]*o + x (layers.py:244); backward: do_ = gamma*dout, dgamma = sum(dout*o) */
int icg_scale_add_fwd(const float* gamma, const float* o, const float* x, float* out, int64_t n, void* stream);
int icg_scale_add_bwd(const float* gamma, const float* o, const float* dout, float* d_o, float* dgamma,
                      int64_t n, void* scratch, size_t scratch_bytes, void* stream);
/* column sums: out[c] = sum_rows x[row][c]  (bias gradients) */
size_t icg_colsum_workspace_bytes(int64_t rows, int C);
int icg_colsum(const float* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);
/* relu forward/backward (stand-alone uses) */
int icg_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int icg_relu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* y = a + b */
int icg_add(const float* a, const float* b, float* y, int64_t n, void* stream);

/* ---- optimiser / EMA  (trainer.py:158-171 optim.Adam, utils.py:1055-1067 ema.update) --- */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
} icg_adam_tensor;
/* torch.optim.Adam semantics (weight_decay = 0, amsgrad = false); step is 1-based */
int icg_adam_multi(const icg_adam_tensor* tensors, int n, float lr, float beta1, float beta2, float eps,
                   int step, void* stream);
typedef struct {
  float* target;
  const float* source;
  int64_t numel;
} icg_ema_tensor;
/* target = target*decay + source*(1-decay) */
int icg_ema_multi(const icg_ema_tensor* tensors, int n, float decay, void* stream);

/* ---- exact L2 k-NN build of the instance-feature table  (data_utils/datasets_common.py:695-769) ------ */
/*
 * For every row q of feats [N][D] (fp32): idx[q][0..k) = the k rows of the SAME table nearest in L2, ascending distance, ties to
 * the lower index, the row itself forced first; d2[q][j] = squared distance (>= 0).  Callers pass k = k_nn + 1 and drop the
 * query like the reference (datasets_common.py:720-740: faiss IndexFlatL2.search(feats, k_nn + 1)).  k <= 64.
 * Inner products on the fp32 MFMA GEMM, selection = one wavefront per query row with a lane-sorted running top-k.
 */
size_t icg_knn_l2_workspace_bytes(int N, int D);
int icg_knn_l2(const float* feats, int N, int D, int k, int64_t* idx, float* d2, void* workspace, size_t workspace_bytes,
               void* stream);

/* ---- StyleGAN2 custom ops (stylegan2_ada_pytorch/torch_utils/ops) ---------------------- */
/*
 * bias_act: same contract as the reference plugin's bias_act(x,b,xref,yref,dy,grad,dim,act,alpha,
 * gain,clamp) (bias_act.cpp:35-100, bias_act.cu:26-150), on a flat contiguous fp32 buffer.
 * bias index of element i is (i / step_b) % size_b; empty inputs are NULL.
 */
int icg_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy,
                 float* y, int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain,
                 float clamp, void* stream);
/*
 * upfirdn2d: same contract as the reference plugin's upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,
 * pady0,pady1,flip,gain) (upfirdn2d.cpp:19-104, upfirdn2d.cu:32-203) for NCHW fp32 x [N][C][H][W]
 * and a 2-D filter f [fh][fw]; y is [N][C][outH][outW] with the reference's output-size formula.
 */
int icg_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw,
                  int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                  int flip, float gain, int outH, int outW, void* stream);

/*
 * Storage-typed forms of the two plugins (the reference templates them on the tensor dtype: bias_act.cu:155-167,
 * upfirdn2d.cu:208-344): dtype 0 = fp32 (forwards to the entries above), 1 = fp16, 2 = fp64.  Arithmetic runs in the
 * plugin's internal type -- fp32 for fp16 / fp32 storage, fp64 for fp64 (bias_act.cu:18-21) -- with one rounding into y.
 * b has the dtype of x (bias_act.cpp:46); the filter f is always fp32 (upfirdn2d.cpp:27).
 * channels_last != 0: x / y are [N][H][W][C] in memory (C % 8 == 0 for fp16, % 2 for fp64, % 4 for fp32).
 */
int icg_bias_act_typed(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                       int64_t n, int64_t step_b, int size_b, int grad, int act, float alpha, float gain, float clamp,
                       int dtype, void* stream);
int icg_upfirdn2d_typed(const void* x, const float* f, void* y, int N, int C, int H, int W, int fh, int fw, int upx,
                        int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                        int outH, int outW, int dtype, int channels_last, void* stream);
/* out[c] = sum_rows x[row][c] in fp32 for an fp16 channels-last tensor [rows][C] (the bias gradient of bias_act in the fp16 blocks,
 * bias_act.py:127); `_applies` -> 1 for C = 8 * 2^k <= 2048; fixed summation order (per-block partials in the workspace). */
int icg_colsum_f16_applies(int C);
size_t icg_colsum_f16_workspace_bytes(int64_t rows, int C);
int icg_colsum_f16(const void* x, int64_t rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* the same operation on channels-last data: x [N][H][W][C], y [N][outH][outW][C], C % 4 == 0 (what the NHWC
 * convolutions produce and consume: no layout change between conv, FIR resampling and bias_act) */
int icg_upfirdn2d_nhwc(const float* x, const float* f, float* y, int N, int C, int H, int W, int fh, int fw,
                       int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                       int flip, float gain, int outH, int outW, void* stream);

/* ---- gradient sanitising of the StyleGAN2 training loop  (training/training_loop.py:511-515) ---- */
/* data[i] <- nan_v for NaN, posinf / neginf for +inf / -inf, unchanged otherwise, in place, for n tensors in ceil(n / 64) launches:
 * what the reference's loop `torch.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad)` does per parameter. */
typedef struct {
  float* data;
  int64_t numel;
} icg_f32_buffer;
int icg_nan_to_num_multi(const icg_f32_buffer* tensors, int n, float nan_v, float posinf, float neginf, void* stream);


/* ==== fused StyleGAN2 layers: stylegan2_ada_pytorch/training/networks.py:37-117 (modulated_conv2d), 361-444
 * (SynthesisLayer), 450-486 (ToRGBLayer), 171-242 (Conv2dLayer), 121-165 (FullyConnectedLayer) =====================================
 * What those layers do AROUND their contraction, as a few launches per layer and pass.  Activations are NHWC [N][HW][C] with
 * storage `dtype` 0 = fp32 / 1 = fp16; per-sample vectors (styles s, demodulation coefficients d) and all reductions are fp32.
 * fp16 storage rounds where the reference's fp16 tensors do (x * s, the convolution output, x * d + noise, the activation and their
 * gradients).  Reductions use per-block partial sums in the caller's workspace and a fixed-order final pass (no atomics). */

/* Weight preparation of n layers in two launches (per <= 40 layers).  Per layer, from the parameter w [O][I][R][R] (R <= 3):
 *   scale[o] = prenorm ? gain * (1 / max_{i,k} |w[o]|) : gain     prenorm: the fp16 pre-normalisation of networks.py:57-63 with
 *                                                                  gain = 1 / sqrt(I R R); otherwise the equalised-lr gain
 *   w_fwd [O][R][R][I]  = w * scale, taps reversed when flip       the gather-convolution weight of icg_conv2d_g_fprop(_f16)
 *   w_adj [I][R][R][O]  = w_fwd with both roles swapped and taps reversed (its data-gradient weight); may be NULL
 *   wsq   [O][I]        = sum_k (w * scale)^2                      for the demodulation coefficients; may be NULL
 *   wscale [O], warg [O] (flat (i, k) index of the maximum; prenorm only)                       kept for icg_sg2_weight_bwd */
typedef struct {
  const float* w;
  void* w_fwd;
  void* w_adj;
  float* wsq;
  float* wscale;
  int* warg;
  int O, I, R, prenorm;
  float gain;
  int flip, dtype, reserved;
} icg_sg2_weight;
int icg_sg2_weight_prep_multi(const icg_sg2_weight* layers, int n, void* stream);

/* styles of one modulated convolution from its affine layer's GEMM output lin [N][I]  (networks.py:409, 57-75):
 *   s0 = (lin + bias * bias_gain) * post_gain;   s = prenorm ? s0 / max_i |s0| : s0   (smax [N] = s0 at the maximum, sarg [N] its index)
 *   d[n][o] = rsqrt(sum_i s[n][i]^2 wsq[o][i] + 1e-8)      when wsq != NULL (demodulate) */
int icg_sg2_style_prep(const float* lin, const float* bias, float bias_gain, float post_gain, const float* wsq, int N, int I, int O,
                       int prenorm, float* s, float* smax, int* sarg, float* d, void* stream);

/* 1 when the row kernels below take C channels at this storage type (C / (16 bytes) a power of two <= 256) */
int icg_sg2_rows_applies(int C, int dtype);
/* xs = x * s[n][c]   (networks.py:78) */
int icg_sg2_modulate(const void* x, const float* s, void* xs, int N, int64_t HW, int C, int dtype, void* stream);
/* y = clamp(gain * act(c * d[n][o] + noise[n * noise_bstride + p] * strength[0] + bias[o]))   act: 1 linear, 3 lrelu(alpha)
 * (networks.py:86-94 fma / add, 432-442 bias_act).  d, noise, bias may be NULL; clamp < 0: none. */
int icg_sg2_act_fwd(const void* c, const float* d, const float* noise, int64_t noise_bstride, const float* strength, const float* bias,
                    void* y, int N, int64_t HW, int O, int act, float alpha, float gain, float clamp, int dtype, void* stream);
/* FIR pass with up = down = 1 (upfirdn2d's blur after the transposed convolution of an up-sampling layer / before a strided one,
 * conv2d_resample.py:152-186, and their adjoints; filter f [fh][fw] fp32 applied as upfirdn2d(flip_filter=flip) does, times fgain) fused with
 * icg_sg2_act_fwd on its result: c = fir(x) (may be NULL), y = clamp(gain * act(c * d + noise * strength + bias)); x [N][H][W][C],
 * c / y [N][outH][outW][C].  With act = 1, gain = 1, no d / noise / bias / clamp it is the plain blur (2 columns x 4 rows per thread: 35 window
 * loads per 8 outputs where the general upfirdn2d kernel does 28 per 4). */
int icg_sg2_fir_act_fwd(const void* x, const float* f, void* c, void* y, const float* d, const float* noise, int64_t noise_bstride,
                        const float* strength, const float* bias, int N, int C, int H, int W, int fh, int fw, int padx0, int padx1,
                        int pady0, int pady1, int flip, float fgain, int outH, int outW, int act, float alpha, float gain, float clamp,
                        int dtype, void* stream);
size_t icg_sg2_rows_workspace_bytes(int N, int64_t HW, int C, int ncols, int dtype);
/* gradient of icg_sg2_act_fwd: dz = dy * gain * act'(y) [|y| < clamp] (bias_act.cu's grad = 1 pass);  dc = dz * d (may be NULL);
 * sums [N][2 O + 1] per sample and tot [2 O + 1] over the batch of (dz | dz * c | dz * noise):  d bias = tot[0 .. O),
 * d d[n][o] = sums[n][O + o], d strength = tot[2 O].  workspace: icg_sg2_rows_workspace_bytes(N, HW, O, 2 O + 1, dtype). */
int icg_sg2_act_bwd(const void* dy, const void* y, const void* c, const float* d, const float* noise, int64_t noise_bstride, void* dc,
                    float* sums, float* tot, int N, int64_t HW, int O, int act, float alpha, float gain, float clamp, int dtype,
                    void* workspace, size_t workspace_bytes, void* stream);
/* gradient of icg_sg2_modulate: dx = dxs * s (may be NULL), ds[n][c] = sum_p dxs * x.  workspace: (N, HW, C, C, dtype). */
int icg_sg2_modulate_bwd(const void* dxs, const void* x, const float* s, void* dx, float* ds, int N, int64_t HW, int C, int dtype,
                         void* workspace, size_t workspace_bytes, void* stream);
/* styles, backward through the demodulation: t[n][o] = -dd d^3;  g[n][i] = ds_mod[n][i] + s[n][i] sum_o t[n][o] wsq[o][i];
 * pdot[n][ceil(I / 64)] = partial sums of g s (for the pre-normalisation's gradient).  dd == NULL: no demodulation (g = ds_mod). */
int icg_sg2_style_bwd(const float* ds_mod, int64_t ds_stride, const float* dd, int64_t dd_stride, const float* d, const float* s,
                      const float* wsq, int N, int I, int O, float* g, float* pdot, float* t, void* stream);
/* backward of a fully-connected layer at a small batch (N <= 64):  lin = wgain x W^T + bias_gain b,  x [N][K], W [I][K].
 * dlin = post_gain * g, or through the style pre-normalisation when smax != NULL:
 *   dlin[n][i] = post_gain (g[n][i] - [i == sarg[n]] sign(smax[n]) sum_k pdot[n][k]) / |smax[n]|
 * dW [I][K] = wgain dlin^T x, db [I] = bias_gain sum_n dlin, dx [N][K] = wgain dlin W   (each may be NULL) */
int icg_sg2_fc_bwd(const float* g, const float* smax, const int* sarg, const float* pdot, int npdot, float post_gain, const float* x,
                   const float* W, int N, int I, int K, float wgain, float bias_gain, float* dW, float* db, float* dx, void* stream);
/* weight gradient of a layer from the convolution's weight gradient dw_conv (fp32; layout 0: [R][R][I][O] as icg_conv2d_g_wgrad(_f16)
 * writes it, 1: [R][R][O][I] -- the zero-inserted direction) plus the demodulation term w scale sum_n t[n][o] s[n][i]^2 (t may be
 * NULL), back through scale[o] (gain, or the pre-normalisation's c0 / max|w[o]|): dw [O][I][R][R].  round_f16: dw_conv is rounded to
 * fp16 first (the gradient of an fp16 weight tensor).  workspace (prenorm only): icg_sg2_weight_bwd_workspace_bytes(O, I). */
size_t icg_sg2_weight_bwd_workspace_bytes(int O, int I);
int icg_sg2_weight_bwd(const float* dw_conv, int layout, const float* t, const float* s, int N, const float* w, const float* wscale,
                       const int* warg, int prenorm, float c0, int round_f16, float* dw, int O, int I, int R, void* workspace,
                       size_t workspace_bytes, void* stream);
/* fromRGB (networks.py:831-836: Conv2dLayer 1x1 over the 3-channel image, bias, activation, clamp) as one pass: x [N][3][HW] planar and
 * w [O][3] (the prepared weight: gain folded in) in the storage type, y [N][HW][O] = clamp(gain * act(x . w + bias)); and its gradient:
 * tot [4 O] = (d w[o][0..2], d bias[o]) at 4 o + k with respect to the prepared weight, dimg [N][3][HW] (may be NULL).
 * workspace: icg_sg2_rows_workspace_bytes(N, HW, O, 4 O, dtype). */
int icg_sg2_fromrgb_applies(int O, int dtype);
int icg_sg2_fromrgb_fwd(const void* x, const void* w, const float* bias, void* y, int N, int64_t HW, int O, int act, float alpha,
                        float gain, float clamp, int dtype, void* stream);
int icg_sg2_fromrgb_bwd(const void* dy, const void* y, const void* x, const void* w, void* dimg, float* tot, int N, int64_t HW, int O,
                        int act, float alpha, float gain, float clamp, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* ---- second-order passes (path-length regularisation differentiates a synthesis layer's backward, loss.py:112-146): the adjoints of the first-order
 * kernels above, used by fused_layers' twice-differentiable layer nodes ---- */
/* u = x * a[n][c] + g * b[n][c]  (g / b may be NULL) */
int icg_sg2_mod2(const void* x, const float* a, const void* g, const float* b, void* u, int N, int64_t HW, int C, int dtype, void* stream);
/* adjoint of icg_sg2_act_bwd with respect to (dy, c, d): given cdc = cot(dc) and cdd = cot(dd) [N][O]:
 *   cdy = (cdc * d + cdd * c) * act'(y)-mask,  cc = cot(c) = cdd * dz (may be NULL),  sums [N][O] = sum_p cdc * dz  (+= cot(d));  dz = dy * mask
 * workspace: icg_sg2_rows_workspace_bytes(N, HW, O, O, dtype) */
int icg_sg2_act_bwd2(const void* dy, const void* y, const void* c, const void* cdc, const float* d, const float* cdd, void* cdy, void* cc,
                     float* sums, int N, int64_t HW, int O, int act, float alpha, float gain, float clamp, int dtype, void* workspace,
                     size_t workspace_bytes, void* stream);
/* icg_sg2_weight_bwd with a general demodulation-table cotangent: g += w scale Q[o][i]  (Q [O][I], may be NULL) */
int icg_sg2_weight_bwd_q(const float* dw_conv, int layout, const float* t, const float* s, int N, const float* Q, const float* w,
                         const float* wscale, const int* warg, int prenorm, float c0, int round_f16, float* dw, int O, int I, int R,
                         void* workspace, size_t workspace_bytes, void* stream);
/* adjoint of icg_sg2_torgb_bwd: a [N][C] = cot(ds), cdx = cot(dx) (may be NULL), cim = cot(d img_in) (may be NULL) ->
 *   cdimg [N][3][HW] = mask * sum_c (x a + cdx s)_c w[o][c] + cim,  cx = cot(x) = dxs * a (may be NULL),
 *   sums [N][4 C] (cot s = sums[:, 0 .. C)), tot [4 C] (cot w[o][c] = tot[(1 + o) C + c]);  workspace as icg_sg2_torgb_bwd */
int icg_sg2_torgb_bwd2(const float* dimg, const void* y, const void* x, const float* s, const float* w, const float* a, const void* cdx,
                       const float* cim, float clamp, int mask_clamp, float* cdimg, void* cx, float* sums, float* tot, int N, int64_t HW,
                       int C, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* ToRGB (networks.py:450-486) as one pass over x: y[n][p][o] = clamp(sum_c (x * s)[n][p][c] w[o][c] + bias[o]), o < 3, stored in the
 * activation type (kept for the backward) and accumulated into the fp32 NCHW image: img_out = img_in + y (img_in may be NULL). */
int icg_sg2_torgb_applies(int C, int dtype);
int icg_sg2_torgb_fwd(const void* x, const float* s, const float* w, const float* bias, float clamp, const float* img_in, float* img_out,
                      void* y, int N, int64_t HW, int C, int dtype, void* stream);
size_t icg_sg2_torgb_bwd_workspace_bytes(int N, int64_t HW, int C, int dtype);
/* its gradient from dimg [N][3][HW] fp32: dx (may be NULL), sums [N][4 C + 3] (ds = sums[:, 0 .. C)), tot [4 C + 3]
 * (dw[o][c] = tot[(1 + o) C + c], d bias[o] = tot[4 C + o]).  mask_clamp 0: the reference CUDA plugin's unmasked gradient of a
 * clamped linear bias_act (bias_act.py:262-266). */
int icg_sg2_torgb_bwd(const float* dimg, const void* y, const void* x, const float* s, const float* w, float clamp, int mask_clamp, void* dx,
                      float* sums, float* tot, int N, int64_t HW, int C, int dtype, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICGAN_HIP_H */
