/*
 * avsep.h — C ABI of libavsep_gfx950.so: the MI355X (gfx950) kernels behind the
 * mix-and-separate train step of abcqmars/audio-visual-sepatation-in-visual-agnostic-situtation.
 *
 * The reference has no native layer: every entry point below
 * replaces a group of stock PyTorch operators at the cited reference call site,
 * and is what a binding for the reference's Python would load with ctypes
 * (INTEGRATION.md shows the stub).  Conventions:
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the boundary;
 *   - every pointer is a DEVICE pointer unless the name says host; tensors are
 *     dense fp32 NCHW unless stated; the caller owns all buffers incl. workspaces;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous;
 *   - return value: 0 = ok, negative = error (avsep_strerror()).
 *   - the library keeps no mutable global state, reads no environment variable and is re-entrant per stream; what steers
 *     the dispatch of a convolution call (A/B and tuning overrides) travels in its descriptor (avsep_conv_desc.algo / .tune).
 */
#ifndef AVSEP_H
#define AVSEP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVSEP_OK 0
#define AVSEP_ERR_ARG (-1)     /* bad shape / null pointer / unsupported geometry */
#define AVSEP_ERR_LAUNCH (-2)  /* hipGetLastError() after a launch */
#define AVSEP_ERR_WORKSPACE (-3)

#define AVSEP_ACT_NONE 0
#define AVSEP_ACT_RELU 1
#define AVSEP_ACT_LRELU02 2    /* LeakyReLU(0.2): models/audio_net.py:64 */
#define AVSEP_ACT_SIGMOID 3
#define AVSEP_ACT_TANH 4
#define AVSEP_ACT_SOFTMAX2 5   /* softmax over a 2-channel dim (models/__init__.py:19-20) */

#define AVSEP_PREC_F32 0
#define AVSEP_PREC_BF16 1

/* Storage format of an activation / gradient tensor [N,C,H,W]:
 *   AVSEP_FMT_F32: dense fp32 NCHW (the reference's tensors);
 *   AVSEP_FMT_B16: bf16, channel-blocked  [N][C/16][H][W][16]  (C % 16 == 0): the 16 channels of a block at one position
 *                  are 32 contiguous bytes — exactly one position of the bf16 kernels' LDS patch, so an operand is staged
 *                  with one 16-byte load per (position, 8-channel half) and no conversion.  Used between the bf16 kernels
 *                  (BASELINE.json configs[2]: half the HBM bytes of fp32 activations). */
#define AVSEP_FMT_F32 0
#define AVSEP_FMT_B16 1

typedef void* avsep_stream_t;

int avsep_version(void);
const char* avsep_arch(void);          /* "gfx950" */
const char* avsep_strerror(int code);

/* ---------------------------------------------------------------------------
 * Convolution as implicit GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d forward/backward at models/audio_net.py:72-98,177-182 and the
 * torchvision ResNet-18 convs wrapped by models/vision_net.py:84-92.
 *
 * The conv input is a virtual tensor: channel-concat of up to two sources (the U-Net
 * skip concat, audio_net.py:122,203), each passed through an optional per-channel
 * affine (a folded BatchNorm, audio_net.py:65-67) and activation (LeakyReLU / ReLU,
 * audio_net.py:64,66) while it is gathered — none of those is materialised.
 * With up2x != 0 the virtual input is additionally the bilinear x2 upsample
 * (align_corners=True, audio_net.py:68-69) of that tensor; H,W are then the
 * UPSAMPLED sizes and the sources are [N,C,H/2,W/2].
 * ------------------------------------------------------------------------- */
/* avsep_conv_desc.algo */
#define AVSEP_ALGO_NO_WINOGRAD        1   /* 3x3/s1 forward + data gradient: direct form instead of Winograd F(2x2,3x3) / F(4x4,3x3) */
#define AVSEP_ALGO_NO_WINOGRAD_WGRAD  2   /* weight gradients: direct form instead of the Winograd-domain kernels */
#define AVSEP_ALGO_NO_FLAT            4   /* no flat-pixel tiles on the 14x14 / 7x7 maps */
#define AVSEP_ALGO_NO_MISC_PATCH      8   /* 3x3/s2, 1x1, stem: im2col kernel instead of the halo-patch kernel */
#define AVSEP_ALGO_NO_BF16_KERNELS   16   /* prec = bf16 ignored: every call runs its exact f32 kernel */
#define AVSEP_ALGO_NO_SMALLCI_WGRAD  32   /* stem / first U-Net conv weight gradient on the im2col kernel */
#define AVSEP_ALGO_NO_WINOGRAD4      64   /* Winograd layers stay on F(2x2,3x3): no F(4x4,3x3) kernel */

typedef struct avsep_conv_desc {
  int32_t N, Cin, H, W;        /* virtual input  [N,Cin,H,W]   */
  int32_t Cout, Ho, Wo;        /* output         [N,Cout,Ho,Wo] */
  int32_t KH, KW, stride, pad, dil;
  int32_t C0;                  /* channels from x0; Cin-C0 from x1 (0 => x1 unused) */
  int32_t act0, act1;          /* AVSEP_ACT_NONE/RELU/LRELU02, applied after the affine */
  int32_t up2x;
  int32_t prec;                /* AVSEP_PREC_F32 (0): exact f32 MFMA; AVSEP_PREC_BF16 (1): operands rounded to bf16 while they are
                                  staged, fp32 accumulation / statistics / outputs (BASELINE.json configs[2]); geometries without a
                                  bf16 kernel run in f32 */
  int32_t plan_n;              /* batch the launch heuristics are PLANNED for (0 = N).  Tile sizes, split-K and the kernel family
                                  of a call depend on how many workgroups its grid has, i.e. on N; with plan_n = 64 a batch-8
                                  call takes the decisions of the batch-64 call (grids and workspaces stay those of N).  The
                                  parity tests use it to run the benched dispatch at an oracle-sized batch; results never
                                  depend on it beyond the summation order of the chosen kernel. */
  int32_t xfmt, yfmt;          /* AVSEP_FMT_* of x0 (forward / weight gradient) and of y as avsep_conv2d_fwd writes it */
  int32_t dyfmt, dxfmt;        /* AVSEP_FMT_* of dy (data / weight gradient) and of dx as avsep_conv2d_dgrad writes it.
                                  avsep_conv_io_formats tells which formats a call takes; B16 pointers are passed as float* */
  int32_t algo;                /* bit set of AVSEP_ALGO_NO_*: kernel families this call must NOT use (0 = the library's choice).
                                  This is the ONLY way to steer the dispatch: the library reads no environment variable and keeps
                                  no process-wide switch, so pack / workspace / launch calls of one descriptor cannot disagree. */
  int32_t tune;                /* measurement tools only (tools/conv_bench.py), 0 = the library's heuristics: bits 0-3 group shape
                                  of the Winograd forward / data gradient + 1, bits 4-7 of the Winograd weight gradient + 1,
                                  bits 8-23 target workgroup count of its K-split */
  const float* x0;
  const float* x1;
  const float* scale0;         /* [C0] or NULL */
  const float* shift0;
  const float* scale1;         /* [Cin-C0] or NULL */
  const float* shift1;
} avsep_conv_desc;

/* Weight repack, once per optimizer step.  w: OIHW [Cout,Cin,KH,KW].
 * mode 0 -> forward operand  [Kpad][Mpad], k=(ci,kh,kw), Mpad=roundup(Cout,128), Kpad=roundup(K,32)
 * mode 1 -> dgrad operand    [KH*KW*Cout (padded to 32)][roundup(Cin,128)]            */
size_t avsep_conv_packed_floats(const avsep_conv_desc* d, int mode);
int avsep_conv_pack_weights(const avsep_conv_desc* d, const float* w, float* packed, int mode,
                            avsep_stream_t stream);

/* y = conv(virtual input) (+bias).  stats (optional, double[2*Cout], pre-zeroed):
 * per-channel sum and sum of squares of y, for the following BatchNorm.
 * workspace (avsep_conv2d_fwd_workspace_bytes; may be 0/NULL): split-K partial slabs for layers whose
 * output grid cannot fill the chip; without it the call falls back to an unsplit launch.        */
/* Formats of a call (mode 0 forward, 1 data gradient, 2 weight gradient) under d->prec and the geometry:
 * *in_fmt = the AVSEP_FMT_* its input tensors MUST have (x0 for mode 0; dy for mode 1; x0 and dy for mode 2),
 * *out_b16 = 1 when the output (y / dx) may be requested as AVSEP_FMT_B16 through d->yfmt / d->dxfmt (fp32 always may).
 * A call whose descriptor disagrees returns AVSEP_ERR_ARG. */
int avsep_conv_io_formats(const avsep_conv_desc* d, int32_t mode, int32_t* in_fmt, int32_t* out_b16);
size_t avsep_conv2d_fwd_workspace_bytes(const avsep_conv_desc* d);
int avsep_conv2d_fwd(const avsep_conv_desc* d, const float* w_packed, const float* bias, float* y,
                     double* stats, void* workspace, size_t workspace_bytes, avsep_stream_t stream);
/* dx = gradient w.r.t. the VIRTUAL input [N,Cin,H,W] (after affine/activation/upsample). */
size_t avsep_conv2d_dgrad_workspace_bytes(const avsep_conv_desc* d);
int avsep_conv2d_dgrad(const avsep_conv_desc* d, const float* w_packed_dgrad, const float* dy,
                       float* dx, void* workspace, size_t workspace_bytes, avsep_stream_t stream);
/* The data gradient taken THROUGH the activation (and BatchNorm / residual add) in front of the convolution's input, in the
 * producing kernel's epilogue: the backward of `relu(bn1(y1))` between the two convs of a ResNet BasicBlock and of the block
 * tail `relu(bn2(y2) + residual)` (vision_net.py:84-109 through torchvision's BasicBlock; main.py:557-569 runs it as
 * autograd's ReluBackward + NativeBatchNormBackward between the two ConvolutionBackward nodes).  The operands are those of
 * avsep_affine_act_bwd below with dz = the data gradient:
 *     dx = act'(scale*y + shift [+ res_scale*residual + res_shift]) * (dgrad(dy) [+ dz2]) [+ add]
 *     bstats += (sum dx, sum dx * (y - mean) * invstd)                                  per channel of dx, fp64
 * y, residual, dz2, add: [N,Cin,H,W] fp32 like dx; the per-channel rows [Cin].  Exactly avsep_conv2d_dgrad followed by
 * avsep_affine_act_bwd in place on dx — which is what the call runs for the kernel families without the epilogue
 * (avsep_conv2d_dgrad_act_fused(d) == 0); with it the gradient is never written unmasked and never re-read.
 * fp32 dx only (d->dxfmt == AVSEP_FMT_F32).  workspace as avsep_conv2d_dgrad.
 * Measured (MI355X, tools/conv_bench.py dgrad --act-epilogue): with y alone (the bn1 form) the call costs what the data
 * gradient alone did on the 14x14 ... 56x56 trunk maps and saves the pass (the host layer uses it there); every further
 * operand (residual, dz2, add) is a read the epilogue cannot hide: slower than the two launches below 32x32 maps,
 * 5-15 % faster on 64x64 and 128x128 ones. */
typedef struct avsep_act_bwd {
  const float* y;              /* the BatchNorm input whose activation the gradient passes            (required) */
  const float* scale;          /* folded BatchNorm rows of y (NULL: identity)                                     */
  const float* shift;
  const float* residual;       /* added to scale*y + shift before the activation (NULL: none)                     */
  const float* res_scale;      /* folded BatchNorm rows of the residual (NULL: plain add)                         */
  const float* res_shift;
  const float* dz2;            /* a second gradient branch reaching the same activation output (NULL: none)       */
  const float* add;            /* added after the activation gradient (NULL: none)                                */
  const float* mean;           /* batch statistics of y: required with bstats                                     */
  const float* invstd;
  double* bstats;              /* [2*Cin], accumulated (NULL: no sums)                                            */
  int32_t act;                 /* AVSEP_ACT_*                                                                     */
} avsep_act_bwd;
int32_t avsep_conv2d_dgrad_act_fused(const avsep_conv_desc* d);   /* 1: one kernel; 0: the two launches */
int avsep_conv2d_dgrad_act(const avsep_conv_desc* d, const float* w_packed_dgrad, const float* dy,
                           const avsep_act_bwd* e, float* dx, void* workspace, size_t workspace_bytes,
                           avsep_stream_t stream);
/* dw OIHW; dbias optional ([Cout]).  workspace from avsep_conv2d_wgrad_workspace_bytes(). */
size_t avsep_conv2d_wgrad_workspace_bytes(const avsep_conv_desc* d);
int avsep_conv2d_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias,
                       void* workspace, size_t workspace_bytes, avsep_stream_t stream);

/* Fused decoder head (audio_net.py:72-76, the outermost up path: ReLU -> Upsample x2 -> Conv3x3 -> num_mix logits).
 * When avsep_conv2d_head_applicable(d) is 1 (up2x=1, 3x3/s1/p1, Cout <= 4, ReLU on both sources, both channel counts
 * multiples of 8, Cin <= 256), avsep_conv2d_fwd / _wgrad contract the channels at LOW resolution (interpolation and
 * channel mix commute) instead of reading a materialised hi-res copy, and avsep_conv2d_dgrad_up2x takes the gradient
 * straight to the two low-res sources: the same results as avsep_conv2d_dgrad followed by avsep_relu_up2x_bwd
 * (g0 [N,C0,H/2,W/2] accumulated in place when acc0, g1 [N,C1,H/2,W/2], bstats1[2*C1] += BatchNorm-backward sums of
 * source 1).  w is OIHW.  All three need their workspace (9*Cout low-res planes per image). */
size_t avsep_conv2d_dgrad_up2x_workspace_bytes(const avsep_conv_desc* d);
int32_t avsep_conv2d_head_applicable(const avsep_conv_desc* d);
/* Name of the kernel family avsep_conv2d_fwd (mode 0, `with_stats` as it will be called), _dgrad (mode 1) or _wgrad
 * (mode 2) dispatches this descriptor to ("convbf_kernel", "wgradbf_kernel", "conv3x3_kernel", "wgrad3x3_kernel",
 * "igemm_kernel<fwd|dgrad|wgrad>", "head_*_kernel", "smallco_*", "smallci_dgrad"): measurement bookkeeping
 * (bench.py groups its HIP-event timings by it), static strings, never freed. */
const char* avsep_conv_kernel_name(const avsep_conv_desc* d, int32_t mode, int32_t with_stats);
/* The same plus every launch decision that depends on the size of the grid (tile shape, 64- / 128-row tiles, 256- / 512-
 * thread workgroups, split-K), e.g. "convbf_kernel:8x32,128x256", "conv3x3_kernel:4x32,BM64", "igemm_kernel<fwd>:BM64,split6":
 * two descriptors with equal variants run the same kernel instantiation.  The parity tests assert that a batch-8 step
 * planned for the bench batch (desc.plan_n) has, layer by layer, the variants of the batch-64 step bench.py times. */
int avsep_conv_kernel_variant(const avsep_conv_desc* d, int32_t mode, int32_t with_stats, char* buf, size_t cap);
int avsep_conv2d_dgrad_up2x(const avsep_conv_desc* d, const float* w, const float* dy, float* g0,
                            float* g1, const float* mean1, const float* invstd1, double* bstats1,
                            int32_t acc0, void* workspace, size_t workspace_bytes, avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * SoP++ attention module core (SoP++/attention_net.py:24-58: `att` + `av_infer_forward`, shared by AttModel / MatchAtt):
 * a [B,S,K] pooled audio queries (S <= 4, K <= 128), mix [B,K,HW] mixed visual map (HW <= 4096), att 0 = cos, 1 = sig.
 *   maps_raw [B,S,HW] similarity maps BEFORE the clamp; ctx [B,S,K] = mean_hw(mix * clamp(maps,0,1));
 *   match [B] = -sum_s mean_hw maps_raw (the module's match term is its batch mean).
 * Backward: dctx [B,S,K], dmaps [B,S,HW] (wrt the clamped maps; may be NULL), dmatch [B] (may be NULL) -> da, dmix.
 * ------------------------------------------------------------------------- */
int avsep_attmodel_infer_fwd(const float* a, const float* mix, int32_t B, int32_t S, int32_t K, int32_t HW, int32_t att,
                             float* maps_raw, float* ctx, float* match, avsep_stream_t stream);
int avsep_attmodel_infer_bwd(const float* a, const float* mix, const float* maps_raw, const float* dctx,
                             const float* dmaps, const float* dmatch, int32_t B, int32_t S, int32_t K, int32_t HW,
                             int32_t att, float* da, float* dmix, avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * BatchNorm2d pieces (train-mode batch statistics; nn.BatchNorm2d at audio_net.py:37,65,67).
 * ------------------------------------------------------------------------- */
/* stats[2*C] += per-channel (sum, sumsq) of x [N,C,HW]. */
int avsep_channel_stats(const float* x, int32_t N, int32_t C, int32_t HW, double* stats,
                        avsep_stream_t stream);
/* From (sum,sumsq) -> scale=gamma*invstd, shift=beta-mean*scale, mean, invstd; updates the
 * running buffers (momentum, unbiased var) when training!=0; with training==0 uses them.
 * `updates` >= 1: the running buffers receive that many momentum updates and num_batches_tracked
 * (nn.BatchNorm2d's int64 counter, may be NULL) is incremented by it — a shared encoder stands for
 * several identical forward passes of the reference (main.py:128-141). */
int avsep_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps,
                      int32_t C, int32_t training, float* scale, float* shift, float* mean,
                      float* invstd, int64_t* num_batches_tracked, int32_t updates, avsep_stream_t stream);
/* Backward of train-mode BN given dz (grad wrt BN output) sums: bstats[2*C] = (sum dz, sum dz*xhat).
 * Writes dgamma,dbeta and the coefficients of dy = p*dz + q*y + r (p,q,r: [C] each).     */
int avsep_bn_bwd_coeffs(const double* bstats, double count, const float* gamma, const float* mean,
                        const float* invstd, int32_t C, float* dgamma, float* dbeta, float* pqr,
                        avsep_stream_t stream);
/* dy = p[c]*dz + q[c]*y + r[c]  (in place on dz allowed). */
int avsep_bn_bwd_apply(const float* dz, const float* y, const float* pqr, int32_t N, int32_t C,
                       int32_t HW, float* dy, avsep_stream_t stream);
/* z = act(scale[c]*y + shift[c] [+ res_scale[c]*residual + res_shift[c]]): BatchNorm + residual + ReLU of a
 * ResNet BasicBlock in one pass (the residual carries its own folded BN when it comes from a downsample conv;
 * res_scale/res_shift NULL -> plain residual add). */
int avsep_affine_act(const float* y, const float* scale, const float* shift, const float* residual,
                     const float* res_scale, const float* res_shift, int32_t act, int32_t N, int32_t C,
                     int32_t HW, float* z, avsep_stream_t stream);
/* Gradient through that z:  dz_pre = act'(pre)*(dz [+ dz2]) (+ add);  also accumulates
 * bstats (sum dz_pre, sum dz_pre*xhat(y)) when bstats != NULL.  dz2 (may be NULL): a second gradient
 * branch reaching z (a BasicBlock output feeds the next block's conv1 AND its residual add), summed on the fly. */
int avsep_affine_act_bwd(const float* dz, const float* dz2, const float* y, const float* scale, const float* shift,
                         const float* residual, const float* res_scale, const float* res_shift,
                         const float* add, const float* mean, const float* invstd, int32_t act,
                         int32_t N, int32_t C, int32_t HW, float* dz_pre, double* bstats,
                         avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * U-Net decoder glue: ReLU + bilinear x2 (align_corners=True) of the concat
 * (audio_net.py:66-69,122,203) and its transpose.
 * Sources as in avsep_conv_desc; a source with bcast!=0 is a [N,C] vector tiled over HxW
 * (the fusion output, fusion_net.py:70-72).
 * ------------------------------------------------------------------------- */
typedef struct avsep_cat_desc {
  int32_t N, C0, C1, H, W;     /* low-res sizes; output is [N,C0+C1,2H,2W] */
  int32_t bcast0, bcast1;
  const float* x0;
  const float* x1;
  const float* scale0;
  const float* shift0;
  const float* scale1;
  const float* shift1;
} avsep_cat_desc;
int avsep_relu_up2x_fwd(const avsep_cat_desc* d, float* out, avsep_stream_t stream);
/* g0/g1: gradient wrt the pre-ReLU (post-affine) value of each source, masked by (value>0);
 * for a bcast source the gradient is summed over HxW into [N,C].  bstats1 (optional):
 * (sum g1, sum g1*xhat1) for the BN that produced source 1 (needs mean1/invstd1).
 * acc0 != 0: g0 += instead of =.                                                         */
int avsep_relu_up2x_bwd(const avsep_cat_desc* d, const float* dout, float* g0, float* g1,
                        const float* mean1, const float* invstd1, double* bstats1, int32_t acc0,
                        avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * prepare: +1e-10, log-frequency warp (grid_sample bilinear/zeros/align_corners=False on the
 * utils.py:12-26 grid), loss weight, ground-truth masks, log  (main.py:51-95).
 * mags: [S][B,1,Fin,T] contiguous as one buffer [S,B,Fin,T].  warp==0 -> Fout==Fin, no resample.
 * Outputs [B,1,Fout,T]: mag_mix_w, log_mag_mix, weight; [S,B,Fout,T]: mags_w, gt.
 * ------------------------------------------------------------------------- */
int avsep_prepare(const float* mag_mix, const float* mags, int32_t S, int32_t B, int32_t Fin,
                  int32_t T, int32_t Fout, int32_t warp, int32_t weighted, int32_t binary,
                  float* mag_mix_w, float* mags_w, float* log_mag_mix, float* weight, float* gt,
                  avsep_stream_t stream);
/* grid_sample of [B,C,Hin,Win] on warpgrid(B,Hout,Wout,warp) — the un-warp of main.py:216-220
 * (warp=0) and the generic resample. */
int avsep_warp(const float* x, int32_t BC, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout,
               int32_t warp, float* y, avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * Bottleneck fusion (models/fusion_net.py).  x: [B,2*Dc,F,T] bottleneck; v: [C=2][B,Dc,H,W].
 * kind: 0 hidsep/CoLoc, 1 CoLoc_Sel, 2 MixVis (v is then ONE map [B,Dc,H,W]);  att: 0 cos, 1 sig.
 * Outputs: feat [B,2*Dc] (the two broadcast vectors), att_maps [B,2,H,W], match_part [B]
 * (per-sample match-loss terms; the caller averages), best [B] (winning permutation),
 * pool_idx [B,2*Dc] argmax of the global max-pool, sel_idx [B,2*Dc] argmax used for feat.
 * bwd: dfeat [B,2*Dc]; dmaps optional [B,2,H,W]; the match-loss cotangent is the DEVICE scalar
 * *dmatch (NULL = 1) times the host factor dmatch_scale (1/B for the batch mean); the gradient
 * wrt x is ADDED into dx_accum [B,2*Dc,F,T]; dv0/dv1 [B,Dc,H,W] are overwritten.
 * ------------------------------------------------------------------------- */
int avsep_fusion_av_fwd(const float* x, const float* v0, const float* v1, int32_t B, int32_t Dc,
                        int32_t FT, int32_t HW, int32_t kind, int32_t att, float* a_pool,
                        int32_t* pool_idx, float* feat, int32_t* sel_idx, float* att_maps,
                        float* match_part, int32_t* best, avsep_stream_t stream);
int avsep_fusion_av_bwd(const float* x, const float* v0, const float* v1, int32_t B, int32_t Dc,
                        int32_t FT, int32_t HW, int32_t kind, int32_t att, const float* a_pool,
                        const int32_t* pool_idx, const int32_t* sel_idx, const float* att_maps,
                        const int32_t* best, const float* dfeat, const float* dmaps,
                        const float* dmatch, float dmatch_scale, float* dx_accum, float* dv0,
                        float* dv1, avsep_stream_t stream);
/* AO branch (fusion_net.py:93-104): feat = swapped global-max-pooled blocks. draws: uint8[B]. */
/* CoLoc (kind 0) for C = 2..4 sources — BUILD-DEFINED generalisation of fusion_net.py:35-72,93-104 (the reference
 * hard-codes C = 2; rules in csrc/fusion_n.hip and DESIGN.md §9): x [B,D,F,T], v[c] [B,Dc,H,W] with Dc = D / C, the
 * audio blocks are the first C*Dc pooled channels, all C! permutations in itertools order, first maximum wins;
 * feat [B,D] (remainder channels zero), att_maps [B,C,H,W], match_part [B], best [B] (permutation index).
 * v / dv: HOST arrays of C device pointers.  Audio-only: draws[b] = permutation index (itertools order) of sample b. */
int avsep_fusion_n_av_fwd(const float* x, const float* const* v, int32_t B, int32_t C, int32_t D, int32_t FT,
                          int32_t HW, int32_t att, float* a_pool, int32_t* pool_idx, float* feat, int32_t* sel_idx,
                          float* att_maps, float* match_part, int32_t* best, avsep_stream_t stream);
int avsep_fusion_n_av_bwd(const float* x, const float* const* v, int32_t B, int32_t C, int32_t D, int32_t FT,
                          int32_t HW, int32_t att, const float* a_pool, const int32_t* pool_idx,
                          const int32_t* sel_idx, const int32_t* best, const float* dfeat, const float* dmatch,
                          float dmatch_scale, float* dx_accum, float* const* dv, avsep_stream_t stream);
int avsep_fusion_n_ao_fwd(const float* x, const int32_t* draws, int32_t B, int32_t C, int32_t D, int32_t FT,
                          float* feat, int32_t* pool_idx, avsep_stream_t stream);
int avsep_fusion_n_ao_bwd(const int32_t* draws, int32_t B, int32_t C, int32_t D, int32_t FT,
                          const int32_t* pool_idx, const float* dfeat, float* dx_accum, avsep_stream_t stream);
int avsep_fusion_ao_fwd(const float* x, const uint8_t* draws, int32_t all_zero, int32_t B,
                        int32_t Dc, int32_t FT, float* feat, int32_t* pool_idx,
                        avsep_stream_t stream);
int avsep_fusion_ao_bwd(const uint8_t* draws, int32_t all_zero, int32_t B, int32_t Dc, int32_t FT,
                        const int32_t* pool_idx, const float* dfeat, float* dx_accum,
                        avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * Mask loss: activation + weighted BCE/L1/L2 (models/criterion.py:10-49) and the PIT loss
 * matrix (criterion.py:138-178) in one pass.  logits [B,S,FT]; gt [S,B,FT]; weight [B,FT] shared by
 * all targets (w_target_stride 0) or one map per target i at weight + i*w_target_stride.
 * pred [B,S,FT] = activation(logits).  sums: double[B*S*S], sums[b,i,j] = sum_ft w*l(pred_j, gt_i).
 * loss: 0 bce, 1 l1, 2 l2.
 * bwd: dlogits[b,j,ft] = sum_i coef[b,i,j] * w * dl/dlogit(pred_j, gt_i).
 * ------------------------------------------------------------------------- */
int avsep_mask_loss_fwd(const float* logits, const float* gt, const float* weight,
                        int64_t w_target_stride, int32_t B, int32_t S, int32_t FT, int32_t act,
                        int32_t loss, float* pred, double* sums, avsep_stream_t stream);
int avsep_mask_loss_bwd(const float* logits, const float* gt, const float* weight,
                        int64_t w_target_stride, const float* coef, int32_t B, int32_t S, int32_t FT,
                        int32_t act, int32_t loss, float* dlogits, avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * STFT / iSTFT (librosa semantics, dataset/base.py:142-147, utils.py:101-104): periodic Hann,
 * center=True, DFT as GEMM on the f32 MFMA.  wav [R,L]; mag/phase [R,n_fft/2+1,frames].
 * The bases (window folded in, cos then -sin) are conv operands built once by avsep_stft_basis
 * into caller buffers of avsep_stft_basis_floats() floats; reflect=1 -> librosa<0.10 pad mode,
 * 0 -> zeros (librosa>=0.10).  phase may be NULL.
 * ------------------------------------------------------------------------- */
size_t avsep_stft_basis_floats(int32_t n_fft, int32_t inverse);
int avsep_stft_basis(int32_t n_fft, float* fwd_basis, float* inv_basis, avsep_stream_t stream);
size_t avsep_stft_workspace_bytes(int32_t R, int32_t L, int32_t n_fft, int32_t hop);
int avsep_stft_mag(const float* wav, int32_t R, int32_t L, int32_t n_fft, int32_t hop,
                   int32_t reflect, const float* basis, float* mag, float* phase, void* workspace,
                   size_t workspace_bytes, avsep_stream_t stream);
size_t avsep_istft_workspace_bytes(int32_t R, int32_t n_fft, int32_t frames);
int avsep_istft(const float* mag, const float* phase, int32_t R, int32_t n_fft, int32_t hop,
                int32_t frames, const float* inv_basis, float* wav, int32_t out_len, void* workspace,
                size_t workspace_bytes, avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * Misc per-frame visual ops and the optimizer.
 * ------------------------------------------------------------------------- */
/* nn.MaxPool2d(3,2,1) of act(scale[c]*x + shift[c]) (the ResNet stem's BN + ReLU folded in; scale NULL = plain). */
int avsep_maxpool3x3s2_fwd(const float* x, const float* scale, const float* shift, int32_t act, int32_t C,
                           int32_t NC, int32_t H, int32_t W, float* y, int32_t* idx, avsep_stream_t stream);
int avsep_maxpool3x3s2_bwd(const float* dy, const int32_t* idx, int32_t NC, int32_t H, int32_t W,
                           float* dx, avsep_stream_t stream);
/* Space-to-depth of the frames for the stem (vision_net.py:111, resnet conv1 7x7/s2/p3 == a 4x4/s1/p0 conv over it):
 * xs [N, Cp, H/2+3, W/2+3], xs[n][(dy*2+dx)*C + c][i+2][j+2] = x[n][c][2i+dy][2j+dx], zero elsewhere; H, W even, Cp >= 4*C. */
int avsep_space_to_depth2(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp, float* xs,
                          avsep_stream_t stream);
/* Stem tail backward, fused (vision_net.py:111-117: conv1 -> bn1 -> relu -> maxpool): from g = dL/d(pooled), the forward's
 * arg-max positions idx and the RAW conv output y with its BatchNorm rows (scale, shift, mean, invstd):
 *   _stats: bstats[2*C] (pre-zeroed) += (sum dz, sum dz*xhat), dz = relu'(scale*y+shift) * maxpool_backward(g);
 *   avsep_bn_bwd_coeffs turns them into (p,q,r);  _apply: dy = p*dz + q*y + r.  dz is never materialised. */
int avsep_maxpool_bn_relu_bwd_stats(const float* g, const int32_t* idx, const float* y, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, int32_t N, int32_t C,
                                    int32_t H, int32_t W, double* bstats, avsep_stream_t stream);
int avsep_maxpool_bn_relu_bwd_apply(const float* g, const int32_t* idx, const float* y, const float* scale,
                                    const float* shift, const float* pqr, int32_t N, int32_t C, int32_t H, int32_t W,
                                    float* dy, avsep_stream_t stream);
/* y[b,c,hw] = mean_t x[b*T+t,c,hw]  (vision_net.py:134-135) and its transpose. */
int avsep_temporal_mean_fwd(const float* x, int32_t B, int32_t T, int32_t CHW, float* y,
                            avsep_stream_t stream);
int avsep_temporal_mean_bwd(const float* dy, int32_t B, int32_t T, int32_t CHW, float* dx,
                            avsep_stream_t stream);
/* SGD with momentum + weight decay (torch.optim.SGD semantics, main.py:547):
 * g += wd*p; buf = first ? g : mom*buf + g; p -= lr*buf.  grad_scale multiplies g first
 * (1/world after the RCCL sum). */
int avsep_sgd_momentum(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                       float weight_decay, float grad_scale, int32_t first, avsep_stream_t stream);
/* InnerProd / Bias synthesizer (models/synthesizer_net.py:12-19): z[b,hw] = sum_k img[b,k]*scale[k]*snd[b,k,hw] + bias */
int avsep_innerprod_fwd(const float* img, const float* snd, const float* scale, const float* bias,
                        int32_t B, int32_t K, int32_t HW, float* z, avsep_stream_t stream);
/* The synthesizer's inference helpers (models/synthesizer_net.py:21-38; scale NULL for `Bias`):
 * forward_nosum     z[b,k,hw] = img[b,k]*scale[k]*snd[b,k,hw] + bias                         z [B,K,HW]
 * forward_pixelwise z[b,p,hw] = sum_k imgs[b,k,p]*scale[k]*snd[b,k,hw] + bias  (imgs [B,K,P]) z [B,P,HW], K even:
 *                   the per-sample [P x K] x [K x HW] contraction on the f32 MFMA. */
int avsep_innerprod_nosum(const float* img, const float* snd, const float* scale, const float* bias,
                          int32_t B, int32_t K, int32_t HW, float* z, avsep_stream_t stream);
int avsep_innerprod_pixelwise(const float* imgs, const float* snd, const float* scale, const float* bias,
                              int32_t B, int32_t K, int32_t P, int32_t HW, float* z, avsep_stream_t stream);
/* backward: dsnd[b,k,hw] = img[b,k]*scale[k]*dz[b,hw] (optional) and r[b,k] = sum_hw snd[b,k,hw]*dz[b,hw],
 * from which the caller forms dimg = scale*r, dscale = sum_b img*r, dbias = sum dz. */
int avsep_innerprod_bwd(const float* img, const float* snd, const float* scale, const float* dz,
                        int32_t B, int32_t K, int32_t HW, float* dsnd, float* r, avsep_stream_t stream);

/* Separation metrics (main.py:260-266): per row r the fp64 inner products
 * sums[3r..] += (<est,ref>, <ref,ref>, <est,est>) over L samples; SI-SDR and SDR are ratios of them. */
int avsep_sdr_sums(const float* est, const float* ref, int32_t R, int32_t L, int64_t est_stride,
                   int64_t ref_stride, double* sums, avsep_stream_t stream);

/* ---------------------------------------------------------------------------
 * AVSEP_FMT_B16 images (bf16, [N][C/16][H][W][16]; csrc/b16.hip): what travels between the bf16 convolution kernels.
 * HW = H*W positions; every entry point is one HBM pass with 16-byte accesses.  Statistics buffers are pre-zeroed doubles
 * that are accumulated into, exactly as for the fp32 NCHW entry points of the same names below.
 * ------------------------------------------------------------------------- */
int avsep_f32_to_b16(const float* x, int32_t N, int32_t C, int32_t HW, void* out, avsep_stream_t stream);
int avsep_b16_to_f32(const void* x, int32_t N, int32_t C, int32_t HW, float* out, avsep_stream_t stream);
/* z = act(scale*y + shift [+ res_scale*residual + res_shift | + residual])   (BasicBlock tail; act NONE/RELU/LRELU02) */
int avsep_b16_affine_act(const void* y, const float* scale, const float* shift, const void* residual,
                         const float* res_scale, const float* res_shift, int32_t act, int32_t N, int32_t C, int32_t HW,
                         void* z, avsep_stream_t stream);
/* out = act'(scale*y + shift [+ residual term]) * (dz [+ dz2]) [+ add]  (out may alias dz; NULL = statistics only);
 * bstats[2*C] += (sum out, sum out * (y - mean) * invstd) */
int avsep_b16_affine_act_bwd(const void* dz, const void* dz2, const void* y, const float* scale, const float* shift,
                             const void* residual, const float* res_scale, const float* res_shift, const void* add,
                             const float* mean, const float* invstd, int32_t act, int32_t N, int32_t C, int32_t HW,
                             void* out, double* bstats, avsep_stream_t stream);
/* out = p*dz + q*y + r with pqr[3*C] from avsep_bn_bwd_coeffs (out may alias dz) */
int avsep_b16_bn_bwd_apply(const void* dz, const void* y, const float* pqr, int32_t N, int32_t C, int32_t HW, void* out,
                           avsep_stream_t stream);
/* Grid image of a batch of small maps — the 2x2 ... 8x8 maps of the deep U-Net levels, models/audio_net.py:64-69,75-76 —
 * (B16 out [1][C/16][HG][WG][16]): image n of x (fp32 NCHW or B16, `xfmt`) goes to rows
 * (n / GX) * PY .., columns (n % GX) * PX .. after the folded affine (scale / shift per channel, or NULL) and activation `act`
 * (AVSEP_ACT_NONE / RELU / LRELU02); every other position is zero.  HG = rows (a multiple of PY), WG = GX * PX.  With pitch
 * H + 1 (3x3 / pad 1; X and dY) or H + 2 for X and H/2 + 1 for dY (4x4 / stride 2 / pad 1) the separators are every image's zero
 * padding, and avsep_conv2d_wgrad over the two grid images (N = 1) returns the weight gradient of the batch. */
int avsep_b16_grid_pack(const void* x, int32_t xfmt, int32_t N, int32_t C, int32_t H, int32_t W, int32_t GX, int32_t PY, int32_t PX,
                        int32_t HG, int32_t WG, const float* scale, const float* shift, int32_t act, void* out, avsep_stream_t stream);
/* The inverse for a convolution RESULT computed over grid images: out (B16 image or fp32 NCHW, `ofmt`) [N][C][H][W] = the real
 * positions of the fp32 grid image [1][C][HG][WG] (image n at rows (n / GX) * PY, columns (n % GX) * PX); stats (or NULL) +=
 * (sum y, sum y^2) per channel of those fp32 values, as avsep_conv2d_fwd accumulates them. */
int avsep_grid_unpack(const float* grid, int32_t N, int32_t C, int32_t H, int32_t W, int32_t GX, int32_t PY, int32_t PX, int32_t HG,
                      int32_t WG, void* out, int32_t ofmt, double* stats, avsep_stream_t stream);
/* out (B16) = p*dz + q*y + r with dz, y fp32 NCHW: avsep_bn_bwd_apply + avsep_f32_to_b16 in one pass (the fp32 -> B16 boundary
 * below the fused decoder head) */
int avsep_bn_bwd_apply_to_b16(const float* dz, const float* y, const float* pqr, int32_t N, int32_t C, int32_t HW, void* out,
                              avsep_stream_t stream);
/* out [N][(C0+C1)/16][2H][2W][16] = bilinear x2 (align_corners) of relu(affine(cat(x0, x1)))   (audio_net.py:66-69,122) */
int avsep_b16_relu_up2x_fwd(const void* x0, const void* x1, const float* sc0, const float* sh0, const float* sc1,
                            const float* sh1, int32_t N, int32_t C0, int32_t C1, int32_t H, int32_t W, void* out,
                            avsep_stream_t stream);
/* its adjoint: g0 / g1 = gradient wrt the pre-ReLU affine values of the two low-res sources (acc0: add into g0);
 * bstats1[2*C1] += BatchNorm-backward sums of source 1 (needs mean1 / invstd1) */
int avsep_b16_relu_up2x_bwd(const void* x0, const void* x1, const float* sc0, const float* sh0, const float* sc1,
                            const float* sh1, int32_t N, int32_t C0, int32_t C1, int32_t H, int32_t W, const void* dout,
                            void* g0, void* g1, const float* mean1, const float* invstd1, double* bstats1, int32_t acc0,
                            avsep_stream_t stream);
/* z = MaxPool2d(3,2,1)(act(scale*y + shift)); idx: one byte per element (winning tap kh*3+kw), blocked like z */
int avsep_b16_maxpool3x3s2_fwd(const void* y, const float* scale, const float* shift, int32_t act, int32_t N, int32_t C,
                               int32_t H, int32_t W, void* z, void* idx, avsep_stream_t stream);
/* fused stem-tail backward (avsep_maxpool_bn_relu_bwd_stats / _apply on B16 images): dy == NULL -> pass 1 (bstats),
 * else pass 2 (dy = p*dz + q*y + r; dy_f32 != 0: dy is written as fp32 NCHW for the fp32 stem weight gradient);
 * g2 (may be NULL) is added to g on the fly */
int avsep_b16_maxpool_bn_relu_bwd(const void* g, const void* g2, const void* idx, const void* y, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, const float* pqr,
                                  int32_t N, int32_t C, int32_t H, int32_t W, double* bstats, void* dy, int32_t dy_f32,
                                  avsep_stream_t stream);
/* avsep_space_to_depth2 written as a one-block B16 image [N][1][H/2+3][W/2+3][16] (4*C <= 16) */
int avsep_b16_space_to_depth2(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, void* xs, avsep_stream_t stream);

/* ---- BSS-eval SDR / SIR / SAR (eval path; replaces asteroid -> mir_eval.separation.bss_eval_sources, main.py:260-266) ----------
 * float64 throughout, as mir_eval.  refs [B][S][L], ests [B][E][L] (E == S in the reference's use), flen <= 512 delayed copies.
 * avsep_bss_corr:    R [B][S][S][2*flen-1]: R[..][tau + flen - 1] = sum_t ref_i[t + tau] * ref_j[t]  (the block-Toeplitz Gram matrix),
 *                    D [B][E][S][flen]:     D[..][k] = sum_t ref_i[t - k] * est_e[t]                 (the right-hand sides).
 * avsep_bss_solve:   least-squares filters by LU with partial pivoting (numpy.linalg.solve's algorithm), one workgroup per system.
 *                    mode 0: all sources, C [B][S*flen][E];  mode 1: own source only (E == S), C [B*S][flen][1].
 *                    info[system] = 0, or k + 1 when pivot k is exactly zero (a silent source): C is zero there and the caller
 *                    solves that system by minimum-norm least squares, as mir_eval does.  workspace: avsep_bss_solve_workspace_bytes.
 * avsep_bss_project: out [B][E][L + flen - 1] = sum_i conv(C_i, ref_i) (mode 0) or conv(C_e, ref_e) (mode 1). */
int avsep_bss_corr(const double* refs, const double* ests, int32_t B, int32_t S, int32_t E, int32_t L, int32_t flen, double* R, double* D,
                   avsep_stream_t stream);
size_t avsep_bss_solve_workspace_bytes(int32_t B, int32_t S, int32_t flen, int32_t mode);
int avsep_bss_solve(const double* R, const double* D, int32_t B, int32_t S, int32_t E, int32_t flen, int32_t mode, double* workspace,
                    size_t workspace_bytes, double* C, int32_t* info, avsep_stream_t stream);
int avsep_bss_project(const double* refs, const double* C, int32_t B, int32_t S, int32_t E, int32_t L, int32_t flen, int32_t mode,
                      double* out, avsep_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AVSEP_H */
