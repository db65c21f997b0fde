/* C ABI of libavsep_nhwc_gfx950.so: channels-last BatchNorm / ReLU / residual glue for the MIOpen comparison tool
 * (tools/miopen_compare).  NOT part of the product library: the measured path launches no MIOpen kernel. */
#pragma once
#include "avsep.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---------------------------------------------------------------------------
 * The same BatchNorm / ReLU / residual pieces for channels-last activations ([N,H,W,C] viewed as [M, C];
 * C % 4 == 0, C/4 a divisor of 256).  They sit between the MIOpen NHWC convolutions of the "hybrid" visual backend
 * (torchvision resnet BasicBlock.forward: bn -> relu, bn2 + identity -> relu; vision_net.py:62-147).
 * ------------------------------------------------------------------------- */
/* statistics are two-stage (per-block partial rows in the workspace, then an fp64 reduce that OVERWRITES stats[2*C]:
 * unlike the NCHW entry points these do not accumulate, so the caller needs no zero fill) */
size_t avsep_nhwc_stats_workspace_bytes(int64_t M, int32_t C);
int avsep_nhwc_channel_stats(const float* x, int64_t M, int32_t C, double* stats, void* workspace,
                             size_t workspace_bytes, avsep_stream_t stream);
int avsep_nhwc_affine_act(const float* y, const float* scale, const float* shift, const float* residual,
                          const float* res_scale, const float* res_shift, int32_t act, int64_t M, int32_t C,
                          float* z, avsep_stream_t stream);
/* avsep_nhwc_channel_stats + avsep_bn_finalize (training mode, count = M) with the finalisation in the second stage;
 * num_batches_tracked (nn.BatchNorm2d's int64 counter, may be NULL) is incremented by one */
int avsep_nhwc_bn_train_stats(const float* x, int64_t M, int32_t C, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float momentum, float eps,
                              float* scale, float* shift, float* mean, float* invstd, void* workspace,
                              size_t workspace_bytes, avsep_stream_t stream);
/* dz_pre = act'(scale*y+shift [+ res_scale*residual+res_shift]) * (dz [+ dz2]) (dz2: the second incoming gradient of
 * a residual join, NULL if none; dz_pre may alias dz or be NULL = statistics only);
 * bstats[2*C] = (sum dz_pre, sum dz_pre*xhat(y)) when bstats != NULL; when pqr != NULL the second stage also emits
 * dgamma, dbeta and pqr[3*C] of bn(y) (= avsep_bn_bwd_coeffs with count = M; gamma NULL = ones). */
int avsep_nhwc_affine_act_bwd(const float* dz, const float* dz2, const float* y, const float* scale, const float* shift,
                              const float* residual, const float* res_scale, const float* res_shift,
                              const float* mean, const float* invstd, int32_t act, int64_t M, int32_t C,
                              float* dz_pre, double* bstats, const float* gamma, float* dgamma, float* dbeta,
                              float* pqr, void* workspace, size_t workspace_bytes, avsep_stream_t stream);
int avsep_nhwc_bn_bwd_apply(const float* dz, const float* y, const float* pqr, int64_t M, int32_t C,
                            float* out, avsep_stream_t stream);
/* ResNet stem tail (torchvision resnet: bn1 -> relu -> maxpool 3x3/s2/p1) on channels-last tensors without materialising
 * the activated map: out [N,Ho,Wo,C] = maxpool(relu(scale*y+shift)), taps = winning tap 0..8 per element (one byte each,
 * packed per channel quad).  Backward: pass 1 (dy NULL) -> dgamma, dbeta, pqr of the stem BatchNorm from the masked
 * pooled gradient; pass 2 (dy given, same pqr) -> dy = p*gm + q*y + r. */
int avsep_nhwc_maxpool_bn_relu_fwd(const float* y, const float* scale, const float* shift, int32_t N, int32_t H,
                                   int32_t W, int32_t C, float* out, uint32_t* taps, avsep_stream_t stream);
int avsep_nhwc_maxpool_bn_relu_bwd(const float* g, const uint32_t* taps, const float* y, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, const float* gamma,
                                   int32_t N, int32_t H, int32_t W, int32_t C, float* dgamma, float* dbeta,
                                   float* pqr, float* dy, void* workspace, size_t workspace_bytes,
                                   avsep_stream_t stream);

#ifdef __cplusplus
}
#endif
