// BatchNorm / ReLU / residual pieces for channels-last ([N,H,W,C] = [M rows, C columns]) activations.
// Used by the "hybrid" visual backend (models/vision_hybrid.py): the ResNet convolutions run on MIOpen's NHWC
// implicit-GEMM kernels, everything between them on these kernels — train-mode BatchNorm2d statistics, the fused
// normalise + residual + ReLU pass of a BasicBlock tail (torchvision resnet BasicBlock.forward), and the backward
// with the BatchNorm gradient folded as dy = p*dz + q*y + r (same algebra as the NCHW kernels in ops.hip).
// All are HBM-bound streaming kernels with 16-byte accesses; C % 4 == 0 and C/4 a divisor of 256.
#include "common.h"

// per-channel (sum a, sum a*b) of two [M, C] streams.  One thread owns one channel quad (column tid % C4) and every
// (256/C4)-th row of the block's row range; LDS reduce -> one partial row [2*C] per block in the workspace, summed in
// fp64 by nhwc_stats_reduce_kernel (every block would otherwise hit the same 2*C addresses with atomics: measured
// 3x slower than the two-stage form).
template <bool BWD>
__global__ __launch_bounds__(256) void nhwc_stats_kernel(const float* __restrict__ dz, const float* __restrict__ dz2,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ res, const float* __restrict__ rscale,
                                                         const float* __restrict__ rshift, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int act, long long M, int C,
                                                         float* out, float* __restrict__ stats) {
  __shared__ float red[8][256];
  const int C4 = C >> 2, tid = threadIdx.x;
  const int cq = tid % C4, rl = tid / C4, RL = 256 / C4;
  const long long rows_per = (M + gridDim.x - 1) / gridDim.x;
  const long long r_beg = blockIdx.x * rows_per, r_end = min(M, r_beg + rows_per);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, rs = sc, rh = sh, mu = sh, is = sc;
  if constexpr (BWD) {
    if (scale) { sc = reinterpret_cast<const f32x4*>(scale)[cq]; sh = reinterpret_cast<const f32x4*>(shift)[cq]; }
    if (rscale) { rs = reinterpret_cast<const f32x4*>(rscale)[cq]; rh = reinterpret_cast<const f32x4*>(rshift)[cq]; }
    if (mean) { mu = reinterpret_cast<const f32x4*>(mean)[cq]; is = reinterpret_cast<const f32x4*>(invstd)[cq]; }
  }
  // four independent rows per trip: the loads of one trip are all in flight together (a single dependent
  // load-accumulate chain per thread ran at 0.7 TB/s)
  constexpr int U = 4;
  for (long long r = r_beg + rl; r < r_end; r += (long long)U * RL) {
    f32x4 yv[U], dv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + (long long)u * RL;
      const long long o = (rr < r_end ? rr : r) * C4 + cq;          // clamped: the duplicate is masked below
      yv[u] = reinterpret_cast<const f32x4*>(y)[o];
      if constexpr (BWD) {
        dv[u] = reinterpret_cast<const f32x4*>(dz)[o];
        if (dz2) dv[u] += reinterpret_cast<const f32x4*>(dz2)[o];   // second branch of a residual join
        if (res) rv[u] = reinterpret_cast<const f32x4*>(res)[o];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long rr = r + (long long)u * RL;
      if (rr >= r_end) break;
      if constexpr (!BWD) {
        s1 += yv[u];
        s2 += yv[u] * yv[u];
      } else {   // g = act'(scale*y + shift [+ rscale*res + rshift]) * dz;   stats (sum g, sum g*xhat(y))
        f32x4 pre = yv[u] * sc + sh;
        if (res) pre += rv[u] * rs + rh;
        f32x4 g;
        g.x = act_grad(pre.x, act) * dv[u].x; g.y = act_grad(pre.y, act) * dv[u].y;
        g.z = act_grad(pre.z, act) * dv[u].z; g.w = act_grad(pre.w, act) * dv[u].w;
        if (out) reinterpret_cast<f32x4*>(out)[rr * C4 + cq] = g;
        s1 += g;
        s2 += g * ((yv[u] - mu) * is);
      }
    }
  }
  if (!stats) return;
  red[0][tid] = s1.x; red[1][tid] = s1.y; red[2][tid] = s1.z; red[3][tid] = s1.w;
  red[4][tid] = s2.x; red[5][tid] = s2.y; red[6][tid] = s2.z; red[7][tid] = s2.w;
  __syncthreads();
  for (int i = tid; i < 8 * C4; i += 256) {            // (component k, channel quad q): sum over the RL row lanes
    const int k = i / C4, q = i % C4;
    float s = 0.f;
    for (int j = 0; j < RL; ++j) s += red[k][j * C4 + q];
    stats[(long long)blockIdx.x * 2 * C + (k >> 2) * C + 4 * q + (k & 3)] = s;
  }
}

// stats[i] = sum_b partial[b][i]   (fp64, overwrites: no zero fill needed), i < n = 2*C.  Block = 64 columns x 16 row lanes (coalesced 256-byte rows,
// 16 independent chains per column), LDS tree over the row lanes.
__global__ __launch_bounds__(1024) void nhwc_stats_reduce_kernel(const float* __restrict__ partial, int nblocks, int n,
                                                                 double* __restrict__ stats) {
  __shared__ double red[16][64];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6, i = blockIdx.x * 64 + col;
  double s = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int b = rl; b < nblocks; b += 16) s += (double)partial[(long long)b * n + i];
  }
  red[rl][col] = s;
  __syncthreads();
  if (rl == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][col];
    stats[i] = t;
  }
}

__global__ __launch_bounds__(256) void nhwc_affine_act_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ res,
                                                              const float* __restrict__ rscale,
                                                              const float* __restrict__ rshift, int act, long long n4,
                                                              int C4, float* __restrict__ z) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int cq = (int)i & (C4 - 1);                      // C4 is a power of two (it divides 256)
    f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    if (scale) v = v * reinterpret_cast<const f32x4*>(scale)[cq] + reinterpret_cast<const f32x4*>(shift)[cq];
    if (res) {
      f32x4 r = reinterpret_cast<const f32x4*>(res)[i];
      if (rscale) r = r * reinterpret_cast<const f32x4*>(rscale)[cq] + reinterpret_cast<const f32x4*>(rshift)[cq];
      v += r;
    }
    v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
    reinterpret_cast<f32x4*>(z)[i] = v;
  }
}

// dy = p[c]*dz + q[c]*y + r[c]   (pqr = [3][C]); out may alias dz
__global__ __launch_bounds__(256) void nhwc_bn_bwd_apply_kernel(const float* dz, const float* __restrict__ y,
                                                                const float* __restrict__ pqr, long long n4, int C4,
                                                                float* out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int cq = (int)i & (C4 - 1);
    const f32x4 p = reinterpret_cast<const f32x4*>(pqr)[cq], q = reinterpret_cast<const f32x4*>(pqr)[C4 + cq],
                r = reinterpret_cast<const f32x4*>(pqr)[2 * C4 + cq];
    reinterpret_cast<f32x4*>(out)[i] = p * reinterpret_cast<const f32x4*>(dz)[i] + q * reinterpret_cast<const f32x4*>(y)[i] + r;
  }
}

static bool nhwc_ok(long long M, int C) {
  return M > 0 && C >= 4 && (C & 3) == 0 && C <= 1024 && 256 % (C >> 2) == 0;
}
static int nhwc_grid(long long M, int C, int cap) {
  long long blocks = (M * (C >> 2) + 256 * 8 - 1) / (256 * 8);     // >= 8 float4 per thread
  return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

constexpr int NHWC_STAT_BLOCKS = 1024;
extern "C" size_t avsep_nhwc_stats_workspace_bytes(int64_t M, int32_t C) {
  return nhwc_ok(M, C) ? (size_t)nhwc_grid(M, C, NHWC_STAT_BLOCKS) * 2 * C * sizeof(float) : 0;
}
static int nhwc_reduce(const float* partial, int nblocks, int C, double* stats, hipStream_t st) {
  hipLaunchKernelGGL(nhwc_stats_reduce_kernel, dim3(cdiv(2 * C, 64)), dim3(1024), 0, st, partial, nblocks, 2 * C, stats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

extern "C" int avsep_nhwc_channel_stats(const float* x, int64_t M, int32_t C, double* stats, void* workspace,
                                        size_t workspace_bytes, avsep_stream_t stream) {
  if (!x || !stats || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  if (!workspace || workspace_bytes < avsep_nhwc_stats_workspace_bytes(M, C)) return AVSEP_ERR_WORKSPACE;
  const int nb = nhwc_grid(M, C, NHWC_STAT_BLOCKS);
  hipLaunchKernelGGL(nhwc_stats_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, x, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, (long long)M, C, nullptr, (float*)workspace);
  AVSEP_LAUNCH_CHECK();
  return nhwc_reduce((const float*)workspace, nb, C, stats, (hipStream_t)stream);
}

extern "C" int avsep_nhwc_affine_act(const float* y, const float* scale, const float* shift, const float* residual,
                                     const float* res_scale, const float* res_shift, int32_t act, int64_t M, int32_t C,
                                     float* z, avsep_stream_t stream) {
  if (!y || !z || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr) || (res_scale == nullptr) != (res_shift == nullptr)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(nhwc_affine_act_kernel, dim3(nhwc_grid(M, C, 4096)), dim3(256), 0, (hipStream_t)stream, y, scale,
                     shift, residual, res_scale, res_shift, act, (long long)M * (C >> 2), C >> 2, z);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

extern "C" int avsep_nhwc_affine_act_bwd(const float* dz, const float* dz2, const float* y, const float* scale,
                                         const float* shift,
                                         const float* residual, const float* res_scale, const float* res_shift,
                                         const float* mean, const float* invstd, int32_t act, int64_t M, int32_t C,
                                         float* dz_pre, double* bstats, void* workspace, size_t workspace_bytes,
                                         avsep_stream_t stream) {
  if (!dz || !y || !nhwc_ok(M, C) || (!dz_pre && !bstats)) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr) || (res_scale == nullptr) != (res_shift == nullptr)) return AVSEP_ERR_ARG;
  if (bstats && (!mean || !invstd)) return AVSEP_ERR_ARG;
  if (bstats && (!workspace || workspace_bytes < avsep_nhwc_stats_workspace_bytes(M, C))) return AVSEP_ERR_WORKSPACE;
  const int nb = nhwc_grid(M, C, bstats ? NHWC_STAT_BLOCKS : 4096);
  hipLaunchKernelGGL(nhwc_stats_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dz, dz2, y, scale, shift, residual,
                     res_scale, res_shift, mean, invstd, act, (long long)M, C, dz_pre, bstats ? (float*)workspace : nullptr);
  AVSEP_LAUNCH_CHECK();
  return bstats ? nhwc_reduce((const float*)workspace, nb, C, bstats, (hipStream_t)stream) : AVSEP_OK;
}

extern "C" int avsep_nhwc_bn_bwd_apply(const float* dz, const float* y, const float* pqr, int64_t M, int32_t C, float* out,
                                       avsep_stream_t stream) {
  if (!dz || !y || !pqr || !out || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(nhwc_bn_bwd_apply_kernel, dim3(nhwc_grid(M, C, 4096)), dim3(256), 0, (hipStream_t)stream, dz, y, pqr,
                     (long long)M * (C >> 2), C >> 2, out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
