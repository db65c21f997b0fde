// BatchNorm / ReLU / residual pieces for channels-last ([N,H,W,C] = [M rows, C columns]) activations.
// Used by the "hybrid" visual backend (tools/miopen_compare/backend.py): the ResNet convolutions run on MIOpen's NHWC
// implicit-GEMM kernels, everything between them on these kernels — train-mode BatchNorm2d statistics, the fused
// normalise + residual + ReLU pass of a BasicBlock tail (torchvision resnet BasicBlock.forward), and the backward
// with the BatchNorm gradient folded as dy = p*dz + q*y + r (same algebra as the NCHW kernels in ops.hip).
// All are HBM-bound streaming kernels with 16-byte accesses; C % 4 == 0 and C/4 a divisor of 256.
#include "common.h"
#include "avsep_nhwc.h"

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

// Second stage: per channel, (S1, S2) = sum over the partial rows (fp64), written to stats[2*C]; optionally followed
// in the same launch by what the caller would do next with them:
//   mode 1 (forward):  BatchNorm2d finalisation (scale, shift, mean, invstd, running statistics) = avsep_bn_finalize
//   mode 2 (backward): dgamma, dbeta and the folded-gradient coefficients (p, q, r)               = avsep_bn_bwd_coeffs
// Block = 16 channels x 64 row lanes: the per-thread chain over the partial rows is the latency of this launch
// (measured 10 us with 16 lanes), the partials are L2-resident, so short 64-byte row segments are fine.
struct NhwcTail {
  int mode;
  double count;
  const float *gamma, *beta, *mean_in, *invstd_in;
  float *running_mean, *running_var, *scale, *shift, *mean_o, *invstd_o, *dgamma, *dbeta, *pqr;
  float momentum, eps;
  long long* counter;      // nn.BatchNorm2d.num_batches_tracked (mode 1), incremented by one thread
};
__global__ __launch_bounds__(1024) void nhwc_stats_reduce_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                                 double* __restrict__ stats, NhwcTail t) {
  __shared__ double red[2][64][17];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4, c = blockIdx.x * 16 + col;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int b = rl; b < nblocks; b += 64) {
      s1 += (double)partial[(long long)b * 2 * C + c];
      s2 += (double)partial[(long long)b * 2 * C + C + c];
    }
  }
  red[0][rl][col] = s1;
  red[1][rl][col] = s2;
  __syncthreads();
  for (int half = 32; half >= 1; half >>= 1) {             // tree over the 64 row lanes
    if (rl < half) { red[0][rl][col] += red[0][rl + half][col]; red[1][rl][col] += red[1][rl + half][col]; }
    __syncthreads();
  }
  if (t.counter && blockIdx.x == 0 && threadIdx.x == 0) *t.counter += 1;
  if (rl != 0 || c >= C) return;
  s1 = red[0][0][col];
  s2 = red[1][0][col];
  if (stats) { stats[c] = s1; stats[C + c] = s2; }
  if (t.mode == 1) {            // same arithmetic as bn_finalize_kernel (ops.hip), training mode
    const double mean = s1 / t.count;
    double var = s2 / t.count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (t.running_mean) {
      const double unb = t.count > 1.0 ? var * t.count / (t.count - 1.0) : var;
      t.running_mean[c] = (float)((1.0 - t.momentum) * t.running_mean[c] + t.momentum * mean);
      t.running_var[c] = (float)((1.0 - t.momentum) * t.running_var[c] + t.momentum * unb);
    }
    const float invstd = (float)(1.0 / sqrt(var + (double)t.eps));
    const float g = t.gamma ? t.gamma[c] : 1.f, b = t.beta ? t.beta[c] : 0.f, sc = g * invstd;
    t.scale[c] = sc;
    t.shift[c] = b - (float)mean * sc;
    t.mean_o[c] = (float)mean;
    t.invstd_o[c] = invstd;
  } else if (t.mode == 2) {     // same arithmetic as bn_bwd_coeffs_kernel (ops.hip)
    const double g = t.gamma ? t.gamma[c] : 1.0, is = t.invstd_in[c], mu = t.mean_in[c];
    const double p = g * is, q = -p * is * s2 / t.count, r = -p * s1 / t.count - q * mu;
    if (t.dgamma) t.dgamma[c] = (float)s2;
    if (t.dbeta) t.dbeta[c] = (float)s1;
    t.pqr[c] = (float)p;
    t.pqr[C + c] = (float)q;
    t.pqr[2 * C + c] = (float)r;
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

constexpr int NHWC_STAT_BLOCKS = 1024;
static bool nhwc_ok(long long M, int C) {
  return M > 0 && C >= 4 && (C & 3) == 0 && C <= 1024 && 256 % (C >> 2) == 0;
}
static int nhwc_grid(long long M, int C, int cap) {
  long long blocks = (M * (C >> 2) + 256 * 8 - 1) / (256 * 8);     // >= 8 float4 per thread
  return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

extern "C" size_t avsep_nhwc_stats_workspace_bytes(int64_t M, int32_t C) {
  return nhwc_ok(M, C) ? (size_t)nhwc_grid(M, C, NHWC_STAT_BLOCKS) * 2 * C * sizeof(float) : 0;
}
static int nhwc_reduce(const float* partial, int nblocks, int C, double* stats, const NhwcTail& t, hipStream_t st) {
  hipLaunchKernelGGL(nhwc_stats_reduce_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, st, partial, nblocks, C, stats, t);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
static int nhwc_launch_stats(const float* x, long long M, int C, void* workspace, hipStream_t st) {
  const int nb = nhwc_grid(M, C, NHWC_STAT_BLOCKS);
  hipLaunchKernelGGL(nhwc_stats_kernel<false>, dim3(nb), dim3(256), 0, st, nullptr, nullptr, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, 0, M, C, nullptr, (float*)workspace);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

extern "C" int avsep_nhwc_channel_stats(const float* x, int64_t M, int32_t C, double* stats, void* workspace,
                                        size_t workspace_bytes, avsep_stream_t stream) {
  if (!x || !stats || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  if (!workspace || workspace_bytes < avsep_nhwc_stats_workspace_bytes(M, C)) return AVSEP_ERR_WORKSPACE;
  int rc = nhwc_launch_stats(x, M, C, workspace, (hipStream_t)stream);
  if (rc) return rc;
  return nhwc_reduce((const float*)workspace, nhwc_grid(M, C, NHWC_STAT_BLOCKS), C, stats, NhwcTail{}, (hipStream_t)stream);
}

// statistics of x + BatchNorm2d finalisation in the second stage (training mode): avsep_nhwc_channel_stats followed by
// avsep_bn_finalize, one launch fewer
extern "C" int avsep_nhwc_bn_train_stats(const float* x, int64_t M, int32_t C, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                         float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                                         void* workspace, size_t workspace_bytes, avsep_stream_t stream) {
  if (!x || !scale || !shift || !mean || !invstd || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return AVSEP_ERR_ARG;
  if (!workspace || workspace_bytes < avsep_nhwc_stats_workspace_bytes(M, C)) return AVSEP_ERR_WORKSPACE;
  int rc = nhwc_launch_stats(x, M, C, workspace, (hipStream_t)stream);
  if (rc) return rc;
  NhwcTail t{};
  t.mode = 1; t.count = (double)M; t.gamma = gamma; t.beta = beta; t.running_mean = running_mean;
  t.running_var = running_var; t.momentum = momentum; t.eps = eps; t.scale = scale; t.shift = shift; t.mean_o = mean;
  t.invstd_o = invstd; t.counter = (long long*)num_batches_tracked;
  return nhwc_reduce((const float*)workspace, nhwc_grid(M, C, NHWC_STAT_BLOCKS), C, nullptr, t, (hipStream_t)stream);
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
                                         float* dz_pre, double* bstats, const float* gamma, float* dgamma, float* dbeta,
                                         float* pqr, void* workspace, size_t workspace_bytes, avsep_stream_t stream) {
  if (!dz || !y || !nhwc_ok(M, C) || (!dz_pre && !bstats && !pqr)) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr) || (res_scale == nullptr) != (res_shift == nullptr)) return AVSEP_ERR_ARG;
  const bool want = bstats || pqr;
  if (want && (!mean || !invstd)) return AVSEP_ERR_ARG;
  if (want && (!workspace || workspace_bytes < avsep_nhwc_stats_workspace_bytes(M, C))) return AVSEP_ERR_WORKSPACE;
  const int nb = nhwc_grid(M, C, want ? NHWC_STAT_BLOCKS : 4096);
  hipLaunchKernelGGL(nhwc_stats_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dz, dz2, y, scale, shift, residual,
                     res_scale, res_shift, mean, invstd, act, (long long)M, C, dz_pre, want ? (float*)workspace : nullptr);
  AVSEP_LAUNCH_CHECK();
  if (!want) return AVSEP_OK;
  NhwcTail t{};
  if (pqr) {   // BatchNorm-backward coefficients of bn(y) in the second stage (= avsep_bn_bwd_coeffs)
    t.mode = 2; t.count = (double)M; t.gamma = gamma; t.mean_in = mean; t.invstd_in = invstd; t.dgamma = dgamma;
    t.dbeta = dbeta; t.pqr = pqr;
  }
  return nhwc_reduce((const float*)workspace, nb, C, bstats, t, (hipStream_t)stream);
}

extern "C" int avsep_nhwc_bn_bwd_apply(const float* dz, const float* y, const float* pqr, int64_t M, int32_t C, float* out,
                                       avsep_stream_t stream) {
  if (!dz || !y || !pqr || !out || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(nhwc_bn_bwd_apply_kernel, dim3(nhwc_grid(M, C, 4096)), dim3(256), 0, (hipStream_t)stream, dz, y, pqr,
                     (long long)M * (C >> 2), C >> 2, out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ResNet stem tail, channels-last: MaxPool2d(3, 2, 1) of relu(scale*y + shift) without materialising the activated
// tensor (the stem output is the largest activation of the trunk: 308 MB at 96 frames), and its backward fused with
// the ReLU mask and the BatchNorm backward.  thread = one output (forward) / input (backward) pixel x channel quad.
// The winning tap (0..8, first maximum in scan order like nn.MaxPool2d) is kept as one byte per element.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nhwc_pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int N, int H, int W, int C4,
                                                            int Ho, int Wo, float* __restrict__ out,
                                                            uint32_t* __restrict__ taps) {
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % C4);
    long long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[cq], sh = reinterpret_cast<const f32x4*>(shift)[cq];
    f32x4 best = {-1.f, -1.f, -1.f, -1.f};                 // activations are >= 0: any valid tap beats it
    uint32_t bt = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = 2 * ho - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = 2 * wo - 1 + kw;
        if (w < 0 || w >= W) continue;
        f32x4 v = reinterpret_cast<const f32x4*>(y)[((n * H + h) * W + w) * C4 + cq] * sc + sh;
        const uint32_t tap = kh * 3 + kw;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (v.x > best.x) { best.x = v.x; bt = (bt & 0xffffff00u) | tap; }
        if (v.y > best.y) { best.y = v.y; bt = (bt & 0xffff00ffu) | (tap << 8); }
        if (v.z > best.z) { best.z = v.z; bt = (bt & 0xff00ffffu) | (tap << 16); }
        if (v.w > best.w) { best.w = v.w; bt = (bt & 0x00ffffffu) | (tap << 24); }
      }
    }
    reinterpret_cast<f32x4*>(out)[i] = best;
    taps[i] = bt;
  }
}

// gradient reaching input pixel (n,h,w) of the pooled map's cotangent g: the <= 4 windows that contain the pixel and
// chose it; then the ReLU mask of relu(scale*y+shift).
__device__ __forceinline__ f32x4 pool_gather(const float* __restrict__ g, const uint32_t* __restrict__ taps, long long n,
                                             int h, int w, int cq, int C4, int Ho, int Wo) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int ho1 = (h + 1) >> 1, wo1 = (w + 1) >> 1;        // window whose rows are 2*ho1-1 .. 2*ho1+1
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ho = ho1 - a;
    if (ho < 0 || ho >= Ho || (a == 1 && (h & 1) == 0)) continue;       // even rows belong to one window only
    const uint32_t kh = h - (2 * ho - 1);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int wo = wo1 - b;
      if (wo < 0 || wo >= Wo || (b == 1 && (w & 1) == 0)) continue;
      const uint32_t tap = kh * 3 + (w - (2 * wo - 1));
      const long long o = ((n * Ho + ho) * Wo + wo) * C4 + cq;
      const uint32_t bt = taps[o];
      const f32x4 gv = reinterpret_cast<const f32x4*>(g)[o];
      if ((bt & 0xffu) == tap) acc.x += gv.x;
      if (((bt >> 8) & 0xffu) == tap) acc.y += gv.y;
      if (((bt >> 16) & 0xffu) == tap) acc.z += gv.z;
      if ((bt >> 24) == tap) acc.w += gv.w;
    }
  }
  return acc;
}

// APPLY = false: per-block partial (sum gm, sum gm*xhat(y)) of the masked gradient gm (statistics pass, writes nothing
// else); APPLY = true: dy = p*gm + q*y + r.  Rows of the input image are the reduction rows (M = N*H*W).
template <bool APPLY>
__global__ __launch_bounds__(256) void nhwc_pool_bwd_kernel(const float* __restrict__ g, const uint32_t* __restrict__ taps,
                                                            const float* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ pqr,
                                                            int N, int H, int W, int C, int Ho, int Wo,
                                                            float* __restrict__ out, float* __restrict__ partial) {
  __shared__ float red[8][256];
  const int C4 = C >> 2, tid = threadIdx.x, cq = tid % C4, rl = tid / C4, RL = 256 / C4;
  const long long M = (long long)N * H * W, rows_per = (M + gridDim.x - 1) / gridDim.x;
  const long long r_beg = blockIdx.x * rows_per, r_end = min(M, r_beg + rows_per);
  const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[cq], sh = reinterpret_cast<const f32x4*>(shift)[cq];
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, p = mu, q = mu, r3 = mu, s1 = mu, s2 = mu;
  if constexpr (APPLY) {
    p = reinterpret_cast<const f32x4*>(pqr)[cq]; q = reinterpret_cast<const f32x4*>(pqr)[C4 + cq];
    r3 = reinterpret_cast<const f32x4*>(pqr)[2 * C4 + cq];
  } else {
    mu = reinterpret_cast<const f32x4*>(mean)[cq]; is = reinterpret_cast<const f32x4*>(invstd)[cq];
  }
  for (long long r = r_beg + rl; r < r_end; r += RL) {
    const int w = (int)(r % W), h = (int)((r / W) % H);
    const long long n = r / ((long long)W * H);
    const f32x4 yv = reinterpret_cast<const f32x4*>(y)[r * C4 + cq];
    f32x4 gm = pool_gather(g, taps, n, h, w, cq, C4, Ho, Wo);
    const f32x4 pre = yv * sc + sh;
    gm.x = pre.x > 0.f ? gm.x : 0.f; gm.y = pre.y > 0.f ? gm.y : 0.f;
    gm.z = pre.z > 0.f ? gm.z : 0.f; gm.w = pre.w > 0.f ? gm.w : 0.f;
    if constexpr (APPLY) {
      reinterpret_cast<f32x4*>(out)[r * C4 + cq] = p * gm + q * yv + r3;
    } else {
      s1 += gm;
      s2 += gm * ((yv - mu) * is);
    }
  }
  if constexpr (!APPLY) {
    red[0][tid] = s1.x; red[1][tid] = s1.y; red[2][tid] = s1.z; red[3][tid] = s1.w;
    red[4][tid] = s2.x; red[5][tid] = s2.y; red[6][tid] = s2.z; red[7][tid] = s2.w;
    __syncthreads();
    for (int i = tid; i < 8 * C4; i += 256) {
      const int k = i / C4, qq = i % C4;
      float s = 0.f;
      for (int j = 0; j < RL; ++j) s += red[k][j * C4 + qq];
      partial[(long long)blockIdx.x * 2 * C + (k >> 2) * C + 4 * qq + (k & 3)] = s;
    }
  }
}

extern "C" int avsep_nhwc_maxpool_bn_relu_fwd(const float* y, const float* scale, const float* shift, int32_t N, int32_t H,
                                              int32_t W, int32_t C, float* out, uint32_t* taps, avsep_stream_t stream) {
  if (!y || !scale || !shift || !out || !taps || N <= 0 || H <= 0 || W <= 0 || !nhwc_ok((long long)N * H * W, C))
    return AVSEP_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C >> 2);
  hipLaunchKernelGGL(nhwc_pool_fwd_kernel, dim3((unsigned)min((total + 255) / 256, (long long)8192)), dim3(256), 0,
                     (hipStream_t)stream, y, scale, shift, N, H, W, C >> 2, Ho, Wo, out, taps);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// g: cotangent of the pooled map [N,Ho,Wo,C].  Pass 1 (dy == NULL): dgamma, dbeta, pqr of the stem BatchNorm (two-stage
// sums + avsep_bn_bwd_coeffs tail).  Pass 2 (dy != NULL): dy = p*gm + q*y + r with the pqr of pass 1.
extern "C" int avsep_nhwc_maxpool_bn_relu_bwd(const float* g, const uint32_t* taps, const float* y, const float* scale,
                                              const float* shift, const float* mean, const float* invstd,
                                              const float* gamma, int32_t N, int32_t H, int32_t W, int32_t C, float* dgamma,
                                              float* dbeta, float* pqr, float* dy, void* workspace, size_t workspace_bytes,
                                              avsep_stream_t stream) {
  const long long M = (long long)N * H * W;
  if (!g || !taps || !y || !scale || !shift || !pqr || N <= 0 || H <= 0 || W <= 0 || !nhwc_ok(M, C)) return AVSEP_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  if (dy) {
    hipLaunchKernelGGL(nhwc_pool_bwd_kernel<true>, dim3(nhwc_grid(M, C, 8192)), dim3(256), 0, st, g, taps, y, scale, shift,
                       nullptr, nullptr, pqr, N, H, W, C, Ho, Wo, dy, nullptr);
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  if (!mean || !invstd) return AVSEP_ERR_ARG;
  if (!workspace || workspace_bytes < avsep_nhwc_stats_workspace_bytes(M, C)) return AVSEP_ERR_WORKSPACE;
  const int nb = nhwc_grid(M, C, NHWC_STAT_BLOCKS);
  hipLaunchKernelGGL(nhwc_pool_bwd_kernel<false>, dim3(nb), dim3(256), 0, st, g, taps, y, scale, shift, mean, invstd, nullptr,
                     N, H, W, C, Ho, Wo, nullptr, (float*)workspace);
  AVSEP_LAUNCH_CHECK();
  NhwcTail t{};
  t.mode = 2; t.count = (double)M; t.gamma = gamma; t.mean_in = mean; t.invstd_in = invstd; t.dgamma = dgamma;
  t.dbeta = dbeta; t.pqr = pqr;
  return nhwc_reduce((const float*)workspace, nb, C, nullptr, t, st);
}
