// HBM-bound kernels of the train step: BatchNorm pieces, decoder glue (ReLU + bilinear x2 and
// its transpose), prepare (log-frequency warp + masks + weights), fused mask loss, SGD, pooling.
// gfx950 only.  See include/avsep.h for the contract of every entry point.
#include "common.h"

// ============================================================================
// BatchNorm pieces
// ============================================================================
// grid (C, chunks): block reduces one channel over a slice of (n,hw)
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, int N, int C, int HW,
                                                            double* __restrict__ stats) {
  const int c = blockIdx.x;
  const long long total = (long long)N * HW;
  const long long per = (total + gridDim.y - 1) / gridDim.y;
  const long long beg = per * blockIdx.y, end = min(total, beg + per);
  float s = 0.f, q = 0.f;
  double ds = 0.0, dq = 0.0;
  int cnt = 0;
  for (long long i = beg + threadIdx.x; i < end; i += 256) {
    int n = (int)(i / HW), hw = (int)(i % HW);
    float v = x[((long long)n * C + c) * HW + hw];
    s += v;
    q += v * v;
    if (++cnt == 64) { ds += s; dq += q; s = q = 0.f; cnt = 0; }
  }
  ds += s;
  dq += q;
  ds = wave_sum_d(ds);
  dq = wave_sum_d(dq);
  __shared__ double sh[8];
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = ds; sh[4 + (threadIdx.x >> 6)] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[c], sh[0] + sh[1] + sh[2] + sh[3]);
    atomicAdd(&stats[C + c], sh[4] + sh[5] + sh[6] + sh[7]);
  }
}

extern "C" int avsep_channel_stats(const float* x, int32_t N, int32_t C, int32_t HW, double* stats,
                                   avsep_stream_t stream) {
  if (!x || !stats || N <= 0 || C <= 0 || HW <= 0) return AVSEP_ERR_ARG;
  long long total = (long long)N * HW;
  int chunks = (int)min((long long)cdiv(2048, C), (total + 4095) / 4096);
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(channel_stats_kernel, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, x, N, C, HW, stats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float momentum, float eps, int C, int training, float* scale, float* shift,
                                   float* mean_o, float* invstd_o, long long* num_batches_tracked, int updates) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && num_batches_tracked) *num_batches_tracked += updates;
  if (c >= C) return;
  double mean, var;
  if (training) {
    mean = stats[c] / count;
    var = stats[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      float rm = running_mean[c], rv = running_var[c];
      for (int u = 0; u < updates; ++u) {      // `updates` identical forward passes (shared encoder), rounded like separate calls
        rm = (float)((1.0 - momentum) * rm + momentum * mean);
        rv = (float)((1.0 - momentum) * rv + momentum * unb);
      }
      running_mean[c] = rm;
      running_var[c] = rv;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  float invstd = (float)(1.0 / sqrt(var + (double)eps));
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)mean * sc;
  if (mean_o) mean_o[c] = (float)mean;
  if (invstd_o) invstd_o[c] = invstd;
}

extern "C" int avsep_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, int32_t C,
                                 int32_t training, float* scale, float* shift, float* mean, float* invstd,
                                 int64_t* num_batches_tracked, int32_t updates, avsep_stream_t stream) {
  if (C <= 0 || !scale || !shift || updates < 1) return AVSEP_ERR_ARG;
  if (training ? !stats : (!running_mean || !running_var)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, stats, count, gamma,
                     beta, running_mean, running_var, momentum, eps, C, training, scale, shift, mean, invstd,
                     (long long*)num_batches_tracked, updates);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ void bn_bwd_coeffs_kernel(const double* __restrict__ bstats, double count, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ invstd, int C,
                                     float* dgamma, float* dbeta, float* pqr) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = bstats[c], s2 = bstats[C + c];
  double g = gamma ? gamma[c] : 1.0, is = invstd[c], mu = mean[c];
  double p = g * is;
  double q = -p * is * s2 / count;
  double r = -p * s1 / count - q * mu;
  if (dgamma) dgamma[c] = (float)s2;
  if (dbeta) dbeta[c] = (float)s1;
  pqr[c] = (float)p;
  pqr[C + c] = (float)q;
  pqr[2 * C + c] = (float)r;
}

extern "C" int avsep_bn_bwd_coeffs(const double* bstats, double count, const float* gamma, const float* mean,
                                   const float* invstd, int32_t C, float* dgamma, float* dbeta, float* pqr,
                                   avsep_stream_t stream) {
  if (!bstats || !mean || !invstd || !pqr || C <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, bstats, count, gamma,
                     mean, invstd, C, dgamma, dbeta, pqr);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// grid (chunks over hw, C, N) — float4 when HW % 4 == 0
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                           const float* __restrict__ pqr, int C, int HW,
                                                           float* __restrict__ dy) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float p = pqr[c], q = pqr[C + c], r = pqr[2 * C + c];
  const long long base = ((long long)n * C + c) * HW;
  if ((HW & 3) == 0) {
    const float4* dz4 = reinterpret_cast<const float4*>(dz + base);
    const float4* y4 = reinterpret_cast<const float4*>(y + base);
    float4* o4 = reinterpret_cast<float4*>(dy + base);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
      float4 a = dz4[i], b = y4[i], o;
      o.x = fmaf(p, a.x, fmaf(q, b.x, r));
      o.y = fmaf(p, a.y, fmaf(q, b.y, r));
      o.z = fmaf(p, a.z, fmaf(q, b.z, r));
      o.w = fmaf(p, a.w, fmaf(q, b.w, r));
      o4[i] = o;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256)
      dy[base + i] = fmaf(p, dz[base + i], fmaf(q, y[base + i], r));
  }
}

extern "C" int avsep_bn_bwd_apply(const float* dz, const float* y, const float* pqr, int32_t N, int32_t C, int32_t HW,
                                  float* dy, avsep_stream_t stream) {
  if (!dz || !y || !pqr || !dy || N <= 0 || C <= 0 || HW <= 0) return AVSEP_ERR_ARG;
  if (C > 65535 || N > 65535) return AVSEP_ERR_ARG;
  int gx = min(cdiv(HW, 1024), 64);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gx, C, N), dim3(256), 0, (hipStream_t)stream, dz, y, pqr, C, HW, dy);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ res, const float* __restrict__ rscale,
                                                         const float* __restrict__ rshift, int act, int C, int HW,
                                                         float* __restrict__ z) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
  const float rs = rscale ? rscale[c] : 1.f, rh = rscale ? rshift[c] : 0.f;
  const long long base = ((long long)n * C + c) * HW;
  if ((HW & 3) == 0) {                                   // 16-byte accesses (the scalar form ran at 3.0-3.7 TB/s)
    const float slope = act_slope(act);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y + base);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(res + base);
    f32x4* z4 = reinterpret_cast<f32x4*>(z + base);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
      const f32x4 a = y4[i];
      f32x4 v = {fmaf(a.x, sc, sh), fmaf(a.y, sc, sh), fmaf(a.z, sc, sh), fmaf(a.w, sc, sh)};
      if (res) {
        const f32x4 r = r4[i];
        v.x += fmaf(r.x, rs, rh); v.y += fmaf(r.y, rs, rh); v.z += fmaf(r.z, rs, rh); v.w += fmaf(r.w, rs, rh);
      }
      z4[i] = f32x4{act_by_slope(v.x, slope), act_by_slope(v.y, slope), act_by_slope(v.z, slope), act_by_slope(v.w, slope)};
    }
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float v = fmaf(y[base + i], sc, sh);
    if (res) v += fmaf(res[base + i], rs, rh);
    z[base + i] = act_apply(v, act);
  }
}

extern "C" int avsep_affine_act(const float* y, const float* scale, const float* shift, const float* residual,
                                const float* res_scale, const float* res_shift, int32_t act, int32_t N, int32_t C,
                                int32_t HW, float* z, avsep_stream_t stream) {
  if (!y || !z || N <= 0 || C <= 0 || HW <= 0 || C > 65535 || N > 65535) return AVSEP_ERR_ARG;
  if ((res_scale == nullptr) != (res_shift == nullptr)) return AVSEP_ERR_ARG;
  int gx = min(cdiv(HW, (HW & 3) == 0 ? 1024 : 256), 64);
  hipLaunchKernelGGL(affine_act_kernel, dim3(gx, C, N), dim3(256), 0, (hipStream_t)stream, y, scale, shift, residual,
                     res_scale, res_shift, act, C, HW, z);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// grid (C, chunks).  V = 4: HW % 4 == 0, 16-byte loads / stores and one 32-bit division per four elements (the scalar
// form spent a 64-bit division and a modulo per element: ~1.5 TB/s on the trunk's tensors)
template <int V>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ dz2,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ res,
                                                             const float* __restrict__ rscale,
                                                             const float* __restrict__ rshift,
                                                             const float* __restrict__ add,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int act, int N, int C,
                                                             int HW, float* __restrict__ out, double* bstats) {
  typedef float fv __attribute__((ext_vector_type(V)));
  const int c = blockIdx.x;
  const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
  const float rs = rscale ? rscale[c] : 1.f, rh = rscale ? rshift[c] : 0.f;
  const float mu = mean ? mean[c] : 0.f, is = invstd ? invstd[c] : 1.f;
  const int HWv = HW / V;
  const int total = N * HWv;                                   // host-checked < 2^31
  const int per = (total + gridDim.y - 1) / gridDim.y;
  const int beg = per * blockIdx.y, end = min(total, beg + per);
  float s1 = 0.f, s2 = 0.f;
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    const int n = i / HWv, q = i - n * HWv;
    const long long o = ((long long)n * C + c) * HW + (long long)q * V;
    const fv yv = *reinterpret_cast<const fv*>(y + o), dv = *reinterpret_cast<const fv*>(dz + o);
    fv rv, av, gv, d2;
    if (dz2) d2 = *reinterpret_cast<const fv*>(dz2 + o);
    if (res) rv = *reinterpret_cast<const fv*>(res + o);
    if (add) av = *reinterpret_cast<const fv*>(add + o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float yk = yv[k];
      float pre = fmaf(yk, sc, sh);
      if (res) pre += fmaf(rv[k], rs, rh);
      float gk = act_grad(pre, act) * (dz2 ? dv[k] + d2[k] : dv[k]);
      if (add) gk += av[k];
      gv[k] = gk;
      s1 += gk;
      s2 += gk * (yk - mu) * is;
    }
    *reinterpret_cast<fv*>(out + o) = gv;
  }
  if (bstats) {
    double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
    __shared__ double sh2[8];
    if ((threadIdx.x & 63) == 0) { sh2[threadIdx.x >> 6] = d1; sh2[4 + (threadIdx.x >> 6)] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&bstats[c], sh2[0] + sh2[1] + sh2[2] + sh2[3]);
      atomicAdd(&bstats[C + c], sh2[4] + sh2[5] + sh2[6] + sh2[7]);
    }
  }
}

extern "C" int avsep_affine_act_bwd(const float* dz, const float* dz2, const float* y, const float* scale, const float* shift,
                                    const float* residual, const float* res_scale, const float* res_shift,
                                    const float* add, const float* mean, const float* invstd, int32_t act, int32_t N,
                                    int32_t C, int32_t HW, float* dz_pre, double* bstats, avsep_stream_t stream) {
  if (!dz || !y || !dz_pre || N <= 0 || C <= 0 || HW <= 0) return AVSEP_ERR_ARG;
  if (bstats && (!mean || !invstd)) return AVSEP_ERR_ARG;
  long long total = (long long)N * HW;
  if (total > 0x7fffffffLL) return AVSEP_ERR_ARG;
  int chunks = (int)min((long long)cdiv(2048, C), (total + 2047) / 2048);
  if (chunks < 1) chunks = 1;
  if ((HW & 3) == 0)
    hipLaunchKernelGGL(affine_act_bwd_kernel<4>, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, dz, dz2, y, scale, shift,
                       residual, res_scale, res_shift, add, mean, invstd, act, N, C, HW, dz_pre, bstats);
  else
    hipLaunchKernelGGL(affine_act_bwd_kernel<1>, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, dz, dz2, y, scale, shift,
                       residual, res_scale, res_shift, add, mean, invstd, act, N, C, HW, dz_pre, bstats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ============================================================================
// decoder glue: out = up2x(relu(cat(T0(x0), T1(x1))))   and transpose
// ============================================================================
struct CatArgs {
  int N, C0, C1, H, W, b0, b1;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  float rh, rw;
};
__device__ __forceinline__ float cat_relu_val(const CatArgs& a, int n, int c, int h, int w) {
  float v;
  if (c < a.C0) {
    v = a.b0 ? a.x0[(long long)n * a.C0 + c] : a.x0[(((long long)n * a.C0 + c) * a.H + h) * a.W + w];
    if (a.sc0) v = fmaf(v, a.sc0[c], a.sh0[c]);
  } else {
    int c1 = c - a.C0;
    v = a.b1 ? a.x1[(long long)n * a.C1 + c1] : a.x1[(((long long)n * a.C1 + c1) * a.H + h) * a.W + w];
    if (a.sc1) v = fmaf(v, a.sc1[c1], a.sh1[c1]);
  }
  return v;
}

// value of relu(affine(src)) bilinearly upsampled at (ho, wo) of plane p (row pointers of the low-res plane)
__device__ __forceinline__ float up2x_at(const float* p, int W, int Hs, int Ws, float rh, float rw, float scv, float shv,
                                         int ho, int wo) {
  float fh = rh * (float)ho, fw = rw * (float)wo;
  int h0 = (int)fh, w0 = (int)fw;
  int h1 = h0 + (h0 < Hs - 1), w1 = w0 + (w0 < Ws - 1);
  float lh = fh - (float)h0, lw = fw - (float)w0;
  float v00 = fmaxf(fmaf(p[h0 * W + w0], scv, shv), 0.f), v01 = fmaxf(fmaf(p[h0 * W + w1], scv, shv), 0.f);
  float v10 = fmaxf(fmaf(p[h1 * W + w0], scv, shv), 0.f), v11 = fmaxf(fmaf(p[h1 * W + w1], scv, shv), 0.f);
  return (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
}

// 16-byte stores: thread = 4 consecutive output columns.  grid (ceil(Wo/4/Wb), ceil(Ho/rows), N*C)
__global__ __launch_bounds__(256) void relu_up2x_fwd4_kernel(CatArgs a, float* __restrict__ out, int lw) {
  const int Ho = 2 * a.H, Wo = 2 * a.W, C = a.C0 + a.C1;
  const int nc = blockIdx.z, n = nc / C, c = nc % C;
  const int wq = (blockIdx.x << lw) + (threadIdx.x & ((1 << lw) - 1));
  const int ho = blockIdx.y * (256 >> lw) + (threadIdx.x >> lw);
  if (4 * wq >= Wo || ho >= Ho) return;
  const bool first = c < a.C0;
  const int cs = first ? c : c - a.C0;
  const float* sc = first ? a.sc0 : a.sc1;
  const float* sh = first ? a.sh0 : a.sh1;
  const float scv = sc ? sc[cs] : 1.f, shv = sc ? sh[cs] : 0.f;
  const float* p = (first ? a.x0 : a.x1) + ((long long)n * (first ? a.C0 : a.C1) + cs) * a.H * a.W;
  f32x4 v;
  v.x = up2x_at(p, a.W, a.H, a.W, a.rh, a.rw, scv, shv, ho, 4 * wq);
  v.y = up2x_at(p, a.W, a.H, a.W, a.rh, a.rw, scv, shv, ho, 4 * wq + 1);
  v.z = up2x_at(p, a.W, a.H, a.W, a.rh, a.rw, scv, shv, ho, 4 * wq + 2);
  v.w = up2x_at(p, a.W, a.H, a.W, a.rh, a.rw, scv, shv, ho, 4 * wq + 3);
  *reinterpret_cast<f32x4*>(out + ((long long)nc * Ho + ho) * Wo + 4 * wq) = v;
}

// grid (ceil(Wo/Wb), ceil(Ho/rows), N*C) with Wb = 2^lw columns x rows = 256/Wb rows per block:
// no per-element div/mod, full 256-thread blocks on narrow planes too
__global__ __launch_bounds__(256) void relu_up2x_fwd_kernel(CatArgs a, float* __restrict__ out, int lw) {
  const int Ho = 2 * a.H, Wo = 2 * a.W, C = a.C0 + a.C1;
  const int nc = blockIdx.z, n = nc / C, c = nc % C;   // block-uniform
  const int wo = (blockIdx.x << lw) + (threadIdx.x & ((1 << lw) - 1));
  const int ho = blockIdx.y * (256 >> lw) + (threadIdx.x >> lw);
  if (wo >= Wo || ho >= Ho) return;
  const bool first = c < a.C0;
  const int cs = first ? c : c - a.C0;
  const float* sc = first ? a.sc0 : a.sc1;
  const float* sh = first ? a.sh0 : a.sh1;
  const float scv = sc ? sc[cs] : 1.f, shv = sc ? sh[cs] : 0.f;
  const bool bc = first ? a.b0 : a.b1;
  const float* src = first ? a.x0 : a.x1;
  const int Cs = first ? a.C0 : a.C1;
  float v;
  if (bc) {
    v = fmaxf(fmaf(src[(long long)n * Cs + cs], scv, shv), 0.f);
  } else {
    const float* p = src + ((long long)n * Cs + cs) * a.H * a.W;
    float fh = a.rh * (float)ho, fw = a.rw * (float)wo;
    int h0 = (int)fh, w0 = (int)fw;
    int h1 = h0 + (h0 < a.H - 1), w1 = w0 + (w0 < a.W - 1);
    float lh = fh - (float)h0, lw = fw - (float)w0;
    float v00 = fmaxf(fmaf(p[h0 * a.W + w0], scv, shv), 0.f), v01 = fmaxf(fmaf(p[h0 * a.W + w1], scv, shv), 0.f);
    float v10 = fmaxf(fmaf(p[h1 * a.W + w0], scv, shv), 0.f), v11 = fmaxf(fmaf(p[h1 * a.W + w1], scv, shv), 0.f);
    v = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
  }
  out[((long long)nc * Ho + ho) * Wo + wo] = v;
}


// ---------------------------------------------------------------------------------------------------------------
// "static pair" forms of the x2 align-corners upsample (the same observation as csrc/head.hip): hi-res index o
// interpolates low-res (lo, lo+1) with lo = (o-1)>>1 for every o (weights l = r*o - lo, 1-l; at the borders the
// out-of-range partner is read as 0 or gets weight ~0), so a lane that owns low-res columns 2q, 2q+1 builds hi-res
// columns 4q..4q+3 from (L0..L3) = columns 2q-1..2q+2: own float2 + one value from each neighbour lane.  No integer
// index arithmetic, no gathers: ~7 VALU per output instead of ~50.  Needs a power-of-two low-res width <= 128.
// ---------------------------------------------------------------------------------------------------------------
// thread = (plane, low-res row pair (r, r+1), q) -> hi-res rows 2r+1, 2r+2, columns 4q..4q+3 (two 16-byte stores)
__global__ __launch_bounds__(256) void relu_up2x_fwd_pair_kernel(CatArgs a, float* __restrict__ out, int lw) {
  const int C = a.C0 + a.C1, nc = blockIdx.z * gridDim.y + blockIdx.y;          // plane (n, c): block-uniform; grid (x, y, z)
  if (nc >= a.N * C) return;                                                    // y * z may overshoot the plane count
  const int n = nc / C, c = nc % C;
  const int t = blockIdx.x * 256 + threadIdx.x, q = t & ((1 << lw) - 1), r = (t >> lw) - 1;
  const bool live = r < a.H;                                                    // r = -1 .. H-1
  const bool first = c < a.C0;
  const int cs = first ? c : c - a.C0;
  const float* sc = first ? a.sc0 : a.sc1;
  const float* sh = first ? a.sh0 : a.sh1;
  const float scv = sc ? sc[cs] : 1.f, shv = sc ? sh[cs] : 0.f;
  const float* p = (first ? a.x0 : a.x1) + ((long long)n * (first ? a.C0 : a.C1) + cs) * a.H * a.W + 2 * q;
  float L[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = r + i;
    float l1 = 0.f, l2 = 0.f;
    if (live && rr >= 0 && rr < a.H) {
      const float2 v = *reinterpret_cast<const float2*>(p + (long long)rr * a.W);
      l1 = fmaxf(fmaf(v.x, scv, shv), 0.f);
      l2 = fmaxf(fmaf(v.y, scv, shv), 0.f);
    }
    const float l0 = __shfl_up(l2, 1, 64), l3 = __shfl_down(l1, 1, 64);
    L[i][0] = q > 0 ? l0 : 0.f; L[i][1] = l1; L[i][2] = l2; L[i][3] = 2 * q + 2 < a.W ? l3 : 0.f;
  }
  if (!live) return;
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  // columns 4q+j on pairs (L0,L1), (L1,L2), (L1,L2), (L2,L3)
  float cl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    cl[j] = __fmul_rn(a.rw, (float)(4 * q + j)) - (float)(2 * q - 1 + ((j + 1) >> 1));   // product rounded first, like the reference
    if (4 * q + j == Wo - 1) cl[j] = 0.f;                 // last column: exactly the last low-res value
  }
  float Hr[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    Hr[i][0] = (1.f - cl[0]) * L[i][0] + cl[0] * L[i][1];
    Hr[i][1] = (1.f - cl[1]) * L[i][1] + cl[1] * L[i][2];
    Hr[i][2] = (1.f - cl[2]) * L[i][1] + cl[2] * L[i][2];
    Hr[i][3] = (1.f - cl[3]) * L[i][2] + cl[3] * L[i][3];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ho = 2 * r + 1 + i;
    if (ho >= 0 && ho < Ho) {
      const float l = ho == Ho - 1 ? 0.f : __fmul_rn(a.rh, (float)ho) - (float)r;
      f32x4 v;
      v.x = (1.f - l) * Hr[0][0] + l * Hr[1][0];
      v.y = (1.f - l) * Hr[0][1] + l * Hr[1][1];
      v.z = (1.f - l) * Hr[0][2] + l * Hr[1][2];
      v.w = (1.f - l) * Hr[0][3] + l * Hr[1][3];
      *reinterpret_cast<f32x4*>(out + ((long long)nc * Ho + ho) * Wo + 4 * q) = v;
    }
  }
}

// transpose: thread = (plane, chunk of UPB_R low-res rows, q) sweeps down its rows; per low-res row two new hi-res
// rows of dout (16-byte load + one word from each neighbour lane) are reduced horizontally onto the lane's two
// low-res columns and spread onto rows (r, r+1).  Fused ReLU mask, skip-gradient accumulation and BatchNorm-backward
// sums as in relu_up2x_bwd_kernel.
constexpr int UPB_R = 8;
// MULTI: planes that need TP = (H / UPB_R) << lw <= 64 threads (16x16 / 32x32 low-res maps): 256 / TP planes per block, a
// plane inside one wave, per-plane statistics by a segmented shuffle reduce (one plane per block left 75-94 % of the lanes
// idle: 1.1 TB/s on the [64, 1024, 32, 32] gradient).  tp_log = log2(TP).
template <bool MULTI>
__global__ __launch_bounds__(256) void relu_up2x_bwd_sweep_kernel(CatArgs a, const float* __restrict__ dout,
                                                                  float* __restrict__ g0, float* __restrict__ g1,
                                                                  const float* __restrict__ mean1,
                                                                  const float* __restrict__ invstd1, double* bstats1,
                                                                  int acc0, int lw, int tp_log) {
  const int C = a.C0 + a.C1;
  int nc = blockIdx.z * gridDim.y + blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if constexpr (MULTI) {
    nc = (nc << (8 - tp_log)) + (threadIdx.x >> tp_log);
    t = threadIdx.x & ((1 << tp_log) - 1);
  }
  const bool plane_ok = nc < a.N * C;
  if (!MULTI && !plane_ok) return;
  if (!plane_ok) nc = a.N * C - 1;                      // MULTI: lanes past the last plane compute on it and store nothing
  const int n = nc / C, c = nc % C;
  const int q = t & ((1 << lw) - 1), r0 = (t >> lw) * UPB_R;
  const bool first = c < a.C0;
  const int cs = first ? c : c - a.C0, Cs = first ? a.C0 : a.C1;
  float* g = first ? g0 : g1;
  const bool live = r0 < a.H && g != nullptr && plane_ok;
  const float* sc = first ? a.sc0 : a.sc1;
  const float* sh = first ? a.sh0 : a.sh1;
  const float scv = sc ? sc[cs] : 1.f, shv = sc ? sh[cs] : 0.f;
  const bool stats = !first && bstats1;
  const float mu = stats ? mean1[cs] : 0.f, is = stats ? invstd1[cs] : 1.f;
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const float* dp = dout + (long long)nc * Ho * Wo + 4 * q;
  const long long pbase = ((long long)n * Cs + cs) * a.H * a.W + 2 * q;
  const float* xs = (first ? a.x0 : a.x1) + pbase;
  // hi-res columns 4q-1+k (k = 0..5) on pairs (L0,L1) x2, (L1,L2) x2, (L2,L3) x2; this lane owns L1, L2
  float cl[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int wo = 4 * q - 1 + k;
    cl[k] = __fmul_rn(a.rw, (float)wo) - (float)(2 * q - 1 + (k >> 1));
    if (wo < 0 || wo >= Wo) cl[k] = (k < 2) ? 0.f : 1.f;      // out-of-range column: drop its share of L1 / L2
    if (wo == Wo - 1) cl[k] = 0.f;                            // last column belongs to the last low-res value alone
  }
  // horizontally reduced hi-res row ho -> contributions (T0, T1) to low-res columns 2q, 2q+1
  auto hrow = [&](int ho, float& T0, float& T1) __attribute__((always_inline)) {
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    if (live && ho >= 0 && ho < Ho) m = *reinterpret_cast<const f32x4*>(dp + (long long)ho * Wo);
    float dl = __shfl_up(m.w, 1, 64), dr = __shfl_down(m.x, 1, 64);
    if (q == 0) dl = 0.f;
    if (4 * q + 4 >= Wo) dr = 0.f;
    // column k contributes (1-cl) to its lower partner and cl to its upper partner
    T0 = (cl[0] * dl + cl[1] * m.x) + ((1.f - cl[2]) * m.y + (1.f - cl[3]) * m.z);
    T1 = (cl[2] * m.y + cl[3] * m.z) + ((1.f - cl[4]) * m.w + (1.f - cl[5]) * dr);
  };
  float s1 = 0.f, s2 = 0.f;
  // rows 2r-1, 2r feed row r through pair (r-1, r) with weight l; rows 2r+1, 2r+2 through pair (r, r+1) with 1-l
  float Ta0, Ta1, Tb0, Tb1;
  hrow(2 * r0 - 1, Ta0, Ta1);
  hrow(2 * r0, Tb0, Tb1);
  float G0, G1;
  {
    const float la = __fmul_rn(a.rh, (float)(2 * r0 - 1)) - (float)(r0 - 1), lb = __fmul_rn(a.rh, (float)(2 * r0)) - (float)(r0 - 1);
    G0 = la * Ta0 + lb * Tb0;
    G1 = la * Ta1 + lb * Tb1;
  }
#pragma unroll
  for (int i = 0; i < UPB_R; ++i) {
    const int r = r0 + i;
    hrow(2 * r + 1, Ta0, Ta1);
    hrow(2 * r + 2, Tb0, Tb1);
    const float la = __fmul_rn(a.rh, (float)(2 * r + 1)) - (float)r, lb = __fmul_rn(a.rh, (float)(2 * r + 2)) - (float)r;
    // the last low-res row keeps the whole weight of hi-res row 2H-1 (its upper partner does not exist)
    const float wa = (r == a.H - 1) ? 1.f : 1.f - la;
    G0 += wa * Ta0 + (1.f - lb) * Tb0;
    G1 += wa * Ta1 + (1.f - lb) * Tb1;
    if (live && r < a.H) {
      const float2 v = *reinterpret_cast<const float2*>(xs + (long long)r * a.W);
      float gx = fmaf(v.x, scv, shv) > 0.f ? G0 : 0.f, gy = fmaf(v.y, scv, shv) > 0.f ? G1 : 0.f;
      s1 += gx + gy;
      s2 += gx * (v.x - mu) * is + gy * (v.y - mu) * is;
      float2* dst = reinterpret_cast<float2*>(g + pbase + (long long)r * a.W);
      if (acc0 && first) { const float2 old = *dst; gx += old.x; gy += old.y; }
      *dst = make_float2(gx, gy);
    }
    G0 = la * Ta0 + lb * Tb0;          // share of the same two rows that belongs to row r+1
    G1 = la * Ta1 + lb * Tb1;
  }
  if constexpr (MULTI) {                                 // `stats` differs between the planes of a wave: no early outs
    double d1 = stats && live ? (double)s1 : 0.0, d2 = stats && live ? (double)s2 : 0.0;
    for (int o = (1 << tp_log) >> 1; o > 0; o >>= 1) {
      d1 += __shfl_xor(d1, o, 64);
      d2 += __shfl_xor(d2, o, 64);
    }
    if (stats && plane_ok && t == 0) {
      atomicAdd(&bstats1[cs], d1);
      atomicAdd(&bstats1[a.C1 + cs], d2);
    }
  } else if (stats) {
    const double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
    __shared__ double sh2[8];
    if ((threadIdx.x & 63) == 0) { sh2[threadIdx.x >> 6] = d1; sh2[4 + (threadIdx.x >> 6)] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&bstats1[cs], sh2[0] + sh2[1] + sh2[2] + sh2[3]);
      atomicAdd(&bstats1[a.C1 + cs], sh2[4] + sh2[5] + sh2[6] + sh2[7]);
    }
  }
}

// log2 of the lanes per low-res row (W/2) when the static-pair kernels apply, else -1
static int up2x_pair_lw(const avsep_cat_desc* d) {
  if (d->bcast0 || d->bcast1 || d->W < 4 || d->W > 128 || (d->W & (d->W - 1)) || d->H < 2) return -1;
  int lw = 0;
  while ((2 << lw) < d->W) ++lw;
  return lw;
}

static CatArgs make_cat(const avsep_cat_desc* d) {
  CatArgs a{};
  a.N = d->N; a.C0 = d->C0; a.C1 = d->C1; a.H = d->H; a.W = d->W; a.b0 = d->bcast0; a.b1 = d->bcast1;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.rh = (float)(d->H - 1) / (float)(2 * d->H - 1);
  a.rw = (float)(d->W - 1) / (float)(2 * d->W - 1);
  return a;
}
static int check_cat(const avsep_cat_desc* d) {
  if (!d || d->N <= 0 || d->C0 <= 0 || d->C1 < 0 || d->H <= 0 || d->W <= 0 || !d->x0) return AVSEP_ERR_ARG;
  if (d->C1 > 0 && !d->x1) return AVSEP_ERR_ARG;
  return AVSEP_OK;
}

extern "C" int avsep_relu_up2x_fwd(const avsep_cat_desc* d, float* out, avsep_stream_t stream) {
  int rc = check_cat(d);
  if (rc || !out) return AVSEP_ERR_ARG;
  CatArgs a = make_cat(d);
  long long planes = (long long)d->N * (d->C0 + d->C1);
  if (planes > 0x7fffffffLL || 2 * d->H > 65535) return AVSEP_ERR_ARG;
  const int plw = up2x_pair_lw(d);
  if (plw >= 0 && planes <= 32768LL * 65535) {
    // planes over (y, z): [64, 1024, 16, 16] sources are 65536 planes, one more than grid.y holds (they fell back to the
    // generic kernel: 1.9 TB/s)
    const unsigned gy = (unsigned)(planes < 32768 ? planes : 32768), gz = (unsigned)cdiv(planes, gy);
    hipLaunchKernelGGL(relu_up2x_fwd_pair_kernel, dim3(cdiv((long long)(d->H + 1) << plw, 256), gy, gz), dim3(256), 0,
                       (hipStream_t)stream, a, out, plw);
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  if (!d->bcast0 && !d->bcast1 && d->W >= 8) {              // Wo % 4 == 0: four outputs per thread, 16-byte stores
    int l4 = 6;
    while (l4 > 2 && (1 << (l4 - 1)) >= d->W / 2) --l4;
    hipLaunchKernelGGL(relu_up2x_fwd4_kernel, dim3(cdiv(d->W / 2, 1 << l4), cdiv(2 * d->H, 256 >> l4), (unsigned)planes),
                       dim3(256), 0, (hipStream_t)stream, a, out, l4);
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  int lw = 8;
  while (lw > 2 && (1 << (lw - 1)) >= 2 * d->W) --lw;      // smallest power of two >= Wo, capped at 256
  hipLaunchKernelGGL(relu_up2x_fwd_kernel, dim3(cdiv(2 * d->W, 1 << lw), cdiv(2 * d->H, 256 >> lw), (unsigned)planes),
                     dim3(256), 0, (hipStream_t)stream, a, out, lw);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// Transpose of the bilinear x2 (align_corners=True) along one axis: the hi-res indices that read low-res
// index h lie in [2h-2, 2h+3] (src = dst*(H-1)/(2H-1) is slightly below dst/2); their weights go into six
// registers (static indices only - no scratch).  The forward's float arithmetic is repeated exactly.
__device__ __forceinline__ void up2x_taps6(int h, int Hin, float r, float (&wt)[6]) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int ho = 2 * h - 2 + k;
    float w = 0.f;
    if (ho >= 0 && ho < 2 * Hin) {
      float f = r * (float)ho;
      int h0 = (int)f;
      int h1 = h0 + (h0 < Hin - 1);
      float l = f - (float)h0;
      w = (h0 == h ? 1.f - l : 0.f) + (h1 == h ? l : 0.f);
    }
    wt[k] = w;
  }
}

// grid (tiles of 8x32 low-res pixels, C0+C1, batch slices).  Per image the (2*8+4) x (2*32+4) hi-res window of
// dout is staged in LDS with coalesced row loads; each thread then gathers its <= 6x6 transposed-bilinear taps
// from LDS.  The block reduces the BN-backward sums of its channel (source 1) and issues one pair of atomics.
// Broadcast sources ([N,C] vectors, the innermost level only) take the simple path below.
constexpr int UB_TH = 8, UB_TW = 32, UB_RH = 2 * UB_TH + 4, UB_RW = 2 * UB_TW + 4;

__global__ __launch_bounds__(256) void relu_up2x_bwd_kernel(CatArgs a, const float* __restrict__ dout,
                                                            float* __restrict__ g0, float* __restrict__ g1,
                                                            const float* __restrict__ mean1,
                                                            const float* __restrict__ invstd1, double* bstats1,
                                                            int acc0, int tilesX) {
  __shared__ float tile[UB_RH][UB_RW + 1];
  const int C = a.C0 + a.C1, c = blockIdx.y;
  const int Ho = 2 * a.H, Wo = 2 * a.W, HW = a.H * a.W;
  const bool first = c < a.C0;
  const int cs = first ? c : c - a.C0, Cs = first ? a.C0 : a.C1;
  const bool bc = first ? a.b0 : a.b1;
  float* g = first ? g0 : g1;
  float s1 = 0.f, s2 = 0.f;
  const float mu = (!first && mean1) ? mean1[cs] : 0.f, is = (!first && invstd1) ? invstd1[cs] : 1.f;
  const int n_per = (a.N + gridDim.z - 1) / gridDim.z, n_beg = blockIdx.z * n_per, n_end = min(a.N, n_beg + n_per);
  if (g) {
    if (bc) {  // one thread per sample: the bilinear weights of a constant map sum to 1 per output
      if (blockIdx.x == 0)
        for (int n = n_beg + threadIdx.x; n < n_end; n += 256) {
          float v = cat_relu_val(a, n, c, 0, 0), tot = 0.f;
          if (v > 0.f) {
            const float* p = dout + ((long long)n * C + c) * Ho * Wo;
            for (int i = 0; i < Ho * Wo; ++i) tot += p[i];
          }
          long long o = (long long)n * Cs + cs;
          g[o] = (acc0 && first) ? g[o] + tot : tot;
        }
    } else {
      const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
      const int h0 = (blockIdx.x / tilesX) * UB_TH, w0 = (blockIdx.x % tilesX) * UB_TW;
      const int h = h0 + ty, w = w0 + tx;
      const bool inside = h < a.H && w < a.W;
      float wh[6], ww[6];
      up2x_taps6(min(h, a.H - 1), a.H, a.rh, wh);
      up2x_taps6(min(w, a.W - 1), a.W, a.rw, ww);
      const float* xsrc = first ? a.x0 : a.x1;
      const float* scp = first ? a.sc0 : a.sc1;
      const float* shp = first ? a.sh0 : a.sh1;
      const float scv = scp ? scp[cs] : 1.f, shv = scp ? shp[cs] : 0.f;
      const int r0 = 2 * h0 - 2, c0w = 2 * w0 - 2;   // hi-res origin of the staged window
      for (int n = n_beg; n < n_end; ++n) {
        const float* p = dout + ((long long)n * C + c) * Ho * Wo;
        __syncthreads();
        for (int i = threadIdx.x; i < UB_RH * UB_RW; i += 256) {
          int rr = i / UB_RW, cc = i % UB_RW;          // compile-time divisor
          int ho = r0 + rr, wo = c0w + cc;
          tile[rr][cc] = ((unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo) ? p[(long long)ho * Wo + wo] : 0.f;
        }
        __syncthreads();
        if (inside) {
          const long long o = ((long long)n * Cs + cs) * HW + h * a.W + w;
          const float yv = xsrc[o];
          float tot = 0.f;
          if (fmaf(yv, scv, shv) > 0.f) {
#pragma unroll
            for (int y = 0; y < 6; ++y) {
              float row = 0.f;
#pragma unroll
              for (int x = 0; x < 6; ++x) row = fmaf(ww[x], tile[2 * ty + y][2 * tx + x], row);
              tot = fmaf(wh[y], row, tot);
            }
          }
          g[o] = (acc0 && first) ? g[o] + tot : tot;
          s1 += tot;
          s2 += tot * (yv - mu) * is;
        }
      }
    }
  }
  if (!first && bstats1) {
    double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
    __shared__ double sh2[8];
    if ((threadIdx.x & 63) == 0) { sh2[threadIdx.x >> 6] = d1; sh2[4 + (threadIdx.x >> 6)] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&bstats1[cs], sh2[0] + sh2[1] + sh2[2] + sh2[3]);
      atomicAdd(&bstats1[a.C1 + cs], sh2[4] + sh2[5] + sh2[6] + sh2[7]);
    }
  }
}

extern "C" int avsep_relu_up2x_bwd(const avsep_cat_desc* d, const float* dout, float* g0, float* g1,
                                   const float* mean1, const float* invstd1, double* bstats1, int32_t acc0,
                                   avsep_stream_t stream) {
  int rc = check_cat(d);
  if (rc || !dout) return AVSEP_ERR_ARG;
  if (bstats1 && (!mean1 || !invstd1 || d->bcast1 || !g1)) return AVSEP_ERR_ARG;
  CatArgs a = make_cat(d);
  int C = d->C0 + d->C1;
  const int plw = up2x_pair_lw(d);
  // (tiny planes: one block and one pair of atomics per plane would cost more than the LDS-tiled kernel's batch slices;
  // 16x16 planes — the [64, 1024, 32, 32] gradient of u4's input — ran at 1.2 TB/s on the tiled kernel)
  const long long planes = (long long)d->N * C;
  if (plw >= 0 && d->H >= 8 && planes <= 32768LL * 65535) {
    const int chunks = cdiv(d->H, UPB_R), tp = chunks << plw;                  // threads per plane
    if (tp <= 64 && (chunks & (chunks - 1)) == 0) {                            // several planes per block
      int tp_log = 0;
      while ((1 << tp_log) < tp) ++tp_log;
      const long long groups = cdiv(planes, 256 >> tp_log);
      const unsigned gy = (unsigned)(groups < 32768 ? groups : 32768), gz = (unsigned)cdiv(groups, gy);
      hipLaunchKernelGGL(relu_up2x_bwd_sweep_kernel<true>, dim3(1, gy, gz), dim3(256), 0, (hipStream_t)stream, a, dout, g0, g1, mean1,
                         invstd1, bstats1, acc0, plw, tp_log);
      AVSEP_LAUNCH_CHECK();
      return AVSEP_OK;
    }
    const unsigned gy = (unsigned)(planes < 32768 ? planes : 32768), gz = (unsigned)cdiv(planes, gy);
    hipLaunchKernelGGL(relu_up2x_bwd_sweep_kernel<false>, dim3(cdiv((long long)tp, 256), gy, gz),
                       dim3(256), 0, (hipStream_t)stream, a, dout, g0, g1, mean1, invstd1, bstats1, acc0, plw, 0);
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  int tilesX = cdiv(d->W, UB_TW), tiles = tilesX * cdiv(d->H, UB_TH);
  int chunks = min(cdiv(4096, C * tiles), d->N);   // batch slices: enough blocks to fill the chip, few atomics
  if (chunks < 1) chunks = 1;
  if (C > 65535) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(relu_up2x_bwd_kernel, dim3(tiles, C, chunks), dim3(256), 0, (hipStream_t)stream, a, dout, g0, g1,
                     mean1, invstd1, bstats1, acc0, tilesX);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ============================================================================
// prepare: eps + log-frequency warp + weight + GT masks + log   (main.py:51-95)
// ============================================================================
// numpy.linspace(-1, 1, n)[i] in float64
__device__ __forceinline__ double linspace_pm1(int i, int n) {
  if (n == 1) return -1.0;
  if (i == n - 1) return 1.0;
  return -1.0 + (double)i * (2.0 / (double)(n - 1));
}
// y coordinate of utils.py:warpgrid in float64, cast to fp32 like grid.astype(np.float32)
__device__ __forceinline__ float warp_gy(int f, int Fout, int warp) {
  double yv = linspace_pm1(f, Fout);
  double gy = warp ? (pow(21.0, (yv + 1.0) / 2.0) - 11.0) / 10.0 : log(yv * 10.0 + 11.0) / log(21.0) * 2.0 - 1.0;
  return (float)gy;
}

struct Bilin {
  int y0, x0;
  float wnw, wne, wsw, wse;
};
// F.grid_sample(bilinear, zeros, align_corners=False) coordinates for one output location
__device__ __forceinline__ Bilin grid_bilin(float gx, float gy, int Hin, int Win) {
  float ix = ((gx + 1.f) * (float)Win - 1.f) / 2.f;
  float iy = ((gy + 1.f) * (float)Hin - 1.f) / 2.f;
  float fx = floorf(ix), fy = floorf(iy);
  Bilin b;
  b.x0 = (int)fx;
  b.y0 = (int)fy;
  float ex = fx + 1.f - ix, ey = fy + 1.f - iy;  // distance to the east / south neighbour
  float wx = ix - fx, wy = iy - fy;
  b.wnw = ex * ey;
  b.wne = wx * ey;
  b.wsw = ex * wy;
  b.wse = wx * wy;
  return b;
}
__device__ __forceinline__ float sample_bilin(const float* __restrict__ p, const Bilin& b, int Hin, int Win,
                                              float eps) {
  float v = 0.f;
  bool y0 = (unsigned)b.y0 < (unsigned)Hin, y1 = (unsigned)(b.y0 + 1) < (unsigned)Hin;
  bool x0 = (unsigned)b.x0 < (unsigned)Win, x1 = (unsigned)(b.x0 + 1) < (unsigned)Win;
  if (y0 && x0) v += (p[b.y0 * Win + b.x0] + eps) * b.wnw;
  if (y0 && x1) v += (p[b.y0 * Win + b.x0 + 1] + eps) * b.wne;
  if (y1 && x0) v += (p[(b.y0 + 1) * Win + b.x0] + eps) * b.wsw;
  if (y1 && x1) v += (p[(b.y0 + 1) * Win + b.x0 + 1] + eps) * b.wse;
  return v;
}

// grid (Fout, B), block over t
__global__ __launch_bounds__(256) void prepare_kernel(const float* __restrict__ mag_mix, const float* __restrict__ mags,
                                                      int S, int B, int Fin, int T, int Fout, int warp, int weighted,
                                                      int binary, float* __restrict__ mix_w, float* __restrict__ mags_w,
                                                      float* __restrict__ logm, float* __restrict__ weight,
                                                      float* __restrict__ gt) {
  const int f = blockIdx.x, b = blockIdx.y;
  __shared__ float s_gy;
  if (warp && threadIdx.x == 0) s_gy = warp_gy(f, Fout, 1);
  __syncthreads();
  const long long in_b = (long long)b * Fin * T, out_row = ((long long)b * Fout + f) * T;
  for (int t = threadIdx.x; t < T; t += 256) {
    float mix, src[4];
    if (warp) {
      Bilin bl = grid_bilin((float)linspace_pm1(t, T), s_gy, Fin, T);
      mix = sample_bilin(mag_mix + in_b, bl, Fin, T, 1e-10f);
      for (int s = 0; s < S; ++s) src[s] = sample_bilin(mags + (long long)s * B * Fin * T + in_b, bl, Fin, T, 0.f);
    } else {
      mix = mag_mix[in_b + (long long)f * T + t] + 1e-10f;
      for (int s = 0; s < S; ++s) src[s] = mags[(long long)s * B * Fin * T + in_b + (long long)f * T + t];
    }
    mix_w[out_row + t] = mix;
    logm[out_row + t] = logf(mix);
    weight[out_row + t] = weighted ? fminf(fmaxf(log1pf(mix), 1e-3f), 10.f) : 1.f;
    for (int s = 0; s < S; ++s) {
      long long o = (long long)s * B * Fout * T + out_row + t;
      mags_w[o] = src[s];
      gt[o] = binary ? (src[s] > 0.5f * mix ? 1.f : 0.f) : fminf(fmaxf(src[s] / mix, 0.f), 5.f);
    }
  }
}

extern "C" int avsep_prepare(const float* mag_mix, const float* mags, int32_t S, int32_t B, int32_t Fin, int32_t T,
                             int32_t Fout, int32_t warp, int32_t weighted, int32_t binary, float* mag_mix_w,
                             float* mags_w, float* log_mag_mix, float* weight, float* gt, avsep_stream_t stream) {
  if (!mag_mix || !mags || !mag_mix_w || !mags_w || !log_mag_mix || !weight || !gt) return AVSEP_ERR_ARG;
  if (S < 1 || S > 4 || B <= 0 || Fin <= 0 || T <= 0 || Fout <= 0 || B > 65535) return AVSEP_ERR_ARG;
  if (!warp && Fout != Fin) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(prepare_kernel, dim3(Fout, B), dim3(256), 0, (hipStream_t)stream, mag_mix, mags, S, B, Fin, T, Fout,
                     warp, weighted, binary, mag_mix_w, mags_w, log_mag_mix, weight, gt);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ x, int Hin, int Win, int Hout, int Wout,
                                                   int warp, float* __restrict__ y) {
  const int f = blockIdx.x, bc = blockIdx.y;
  __shared__ float s_gy;
  if (threadIdx.x == 0) s_gy = warp_gy(f, Hout, warp);
  __syncthreads();
  const float* p = x + (long long)bc * Hin * Win;
  for (int t = threadIdx.x; t < Wout; t += 256) {
    Bilin bl = grid_bilin((float)linspace_pm1(t, Wout), s_gy, Hin, Win);
    y[((long long)bc * Hout + f) * Wout + t] = sample_bilin(p, bl, Hin, Win, 0.f);
  }
}

extern "C" int avsep_warp(const float* x, int32_t BC, int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t warp,
                          float* y, avsep_stream_t stream) {
  if (!x || !y || BC <= 0 || BC > 65535 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(warp_kernel, dim3(Hout, BC), dim3(256), 0, (hipStream_t)stream, x, Hin, Win, Hout, Wout, warp, y);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ============================================================================
// mask loss (activation + weighted BCE/L1/L2 + PIT matrix)
// ============================================================================
#define MAXS 4
__device__ __forceinline__ void activate_vec(const float* l, float* p, int S, int act) {
  if (act == AVSEP_ACT_SOFTMAX2) {
    float m = l[0];
    for (int s = 1; s < S; ++s) m = fmaxf(m, l[s]);
    float z = 0.f;
    for (int s = 0; s < S; ++s) { p[s] = expf(l[s] - m); z += p[s]; }
    for (int s = 0; s < S; ++s) p[s] /= z;
    return;
  }
  for (int s = 0; s < S; ++s) {
    float v = l[s];
    if (act == AVSEP_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    else if (act == AVSEP_ACT_TANH) v = tanhf(v);
    else if (act == AVSEP_ACT_RELU) v = fmaxf(v, 0.f);
    p[s] = v;
  }
}
__device__ __forceinline__ float loss_elem(float p, float t, int loss) {
  if (loss == 0) return -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
  float d = p - t;
  return loss == 1 ? fabsf(d) : d * d;
}
__device__ __forceinline__ float loss_grad(float p, float t, int loss) {  // d loss / d p
  if (loss == 0) return (p - t) / fmaxf((1.f - p) * p, 1e-12f);
  float d = p - t;
  return loss == 1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
}

// grid (chunks, B)
__global__ __launch_bounds__(256) void mask_loss_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                            const float* __restrict__ weight, long long wts, int B,
                                                            int S, int FT, int act, int loss,
                                                            float* __restrict__ pred, double* __restrict__ sums) {
  const int b = blockIdx.y;
  float acc[MAXS * MAXS];
  for (int i = 0; i < MAXS * MAXS; ++i) acc[i] = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < FT; i += gridDim.x * 256) {
    float l[MAXS], p[MAXS], t[MAXS];
    for (int s = 0; s < S; ++s) {
      l[s] = logits[((long long)b * S + s) * FT + i];
      t[s] = gt[((long long)s * B + b) * FT + i];
    }
    activate_vec(l, p, S, act);
    for (int s = 0; s < S; ++s) pred[((long long)b * S + s) * FT + i] = p[s];
    for (int ti = 0; ti < S; ++ti) {
      float w = weight ? weight[ti * wts + (long long)b * FT + i] : 1.f;
      for (int pj = 0; pj < S; ++pj) acc[ti * MAXS + pj] += w * loss_elem(p[pj], t[ti], loss);
    }
  }
  __shared__ double sh[4][MAXS * MAXS];
  for (int k = 0; k < S * S; ++k) {
    int ti = k / S, pj = k % S;
    double d = wave_sum_d((double)acc[ti * MAXS + pj]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = d;
  }
  __syncthreads();
  if (threadIdx.x < S * S)
    atomicAdd(&sums[(long long)b * S * S + threadIdx.x],
              sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

extern "C" int avsep_mask_loss_fwd(const float* logits, const float* gt, const float* weight, int64_t w_target_stride,
                                   int32_t B, int32_t S, int32_t FT, int32_t act, int32_t loss, float* pred,
                                   double* sums, avsep_stream_t stream) {
  if (!logits || !gt || !pred || !sums || B <= 0 || B > 65535 || S < 1 || S > MAXS || FT <= 0) return AVSEP_ERR_ARG;
  if (loss < 0 || loss > 2) return AVSEP_ERR_ARG;
  int gx = min(cdiv(FT, 1024), 64);
  hipLaunchKernelGGL(mask_loss_fwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, logits, gt, weight,
                     (long long)w_target_stride, B, S, FT, act, loss, pred, sums);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ gt,
                                                            const float* __restrict__ weight, long long wts,
                                                            const float* __restrict__ coef, int B, int S, int FT, int act,
                                                            int loss, float* __restrict__ dlogits) {
  const int b = blockIdx.y;
  float cf[MAXS * MAXS];
  for (int k = 0; k < S * S; ++k) cf[(k / S) * MAXS + (k % S)] = coef[(long long)b * S * S + k];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < FT; i += gridDim.x * 256) {
    float l[MAXS], p[MAXS], t[MAXS], g[MAXS];
    for (int s = 0; s < S; ++s) {
      l[s] = logits[((long long)b * S + s) * FT + i];
      t[s] = gt[((long long)s * B + b) * FT + i];
    }
    activate_vec(l, p, S, act);
    float w[MAXS];
    for (int ti = 0; ti < S; ++ti) w[ti] = weight ? weight[ti * wts + (long long)b * FT + i] : 1.f;
    for (int pj = 0; pj < S; ++pj) {
      float s = 0.f;
      for (int ti = 0; ti < S; ++ti) {
        float c = cf[ti * MAXS + pj];
        if (c != 0.f) s += c * w[ti] * loss_grad(p[pj], t[ti], loss);
      }
      g[pj] = s;  // d total / d pred_j
    }
    if (act == AVSEP_ACT_SOFTMAX2) {
      float dot = 0.f;
      for (int s = 0; s < S; ++s) dot += g[s] * p[s];
      for (int s = 0; s < S; ++s) dlogits[((long long)b * S + s) * FT + i] = p[s] * (g[s] - dot);
    } else {
      for (int s = 0; s < S; ++s) {
        float d = 1.f;
        if (act == AVSEP_ACT_SIGMOID) d = p[s] * (1.f - p[s]);
        else if (act == AVSEP_ACT_TANH) d = 1.f - p[s] * p[s];
        else if (act == AVSEP_ACT_RELU) d = l[s] > 0.f ? 1.f : 0.f;
        dlogits[((long long)b * S + s) * FT + i] = g[s] * d;
      }
    }
  }
}

extern "C" int avsep_mask_loss_bwd(const float* logits, const float* gt, const float* weight, int64_t w_target_stride,
                                   const float* coef, int32_t B, int32_t S, int32_t FT, int32_t act, int32_t loss,
                                   float* dlogits, avsep_stream_t stream) {
  if (!logits || !gt || !coef || !dlogits || B <= 0 || B > 65535 || S < 1 || S > MAXS || FT <= 0) return AVSEP_ERR_ARG;
  if (loss < 0 || loss > 2) return AVSEP_ERR_ARG;
  int gx = min(cdiv(FT, 1024), 64);
  hipLaunchKernelGGL(mask_loss_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, logits, gt, weight,
                     (long long)w_target_stride, coef, B, S, FT, act, loss, dlogits);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ============================================================================
// pooling, temporal mean, SGD, synthesizer
// ============================================================================
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act, int C, int NC, int H,
                                                          int W, int Ho, int Wo, float* __restrict__ y,
                                                          int* __restrict__ idx) {
  const long long total = (long long)NC * Ho * Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int wo = (int)(i % Wo), ho = (int)((i / Wo) % Ho);
    long long nc = i / ((long long)Wo * Ho);
    const int c = (int)(nc % C);
    const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;   // folded BatchNorm + activation of the stem
    const float* p = x + nc * H * W;
    float best = -INFINITY;
    int bi = -1;
    for (int kh = 0; kh < 3; ++kh) {
      int h = ho * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        int w = wo * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        float v = act_apply(fmaf(p[h * W + w], sc, sh), act);
        if (v > best || bi < 0) { best = v; bi = h * W + w; }
      }
    }
    y[i] = best;
    if (idx) idx[i] = bi;
  }
}
// W % 8 == 0, H even: a thread produces FOUR outputs of one row from 3 x (one scalar + two 16-byte) loads and stores them
// with two 16-byte stores (the generic kernel above: 36 scalar loads and 64-bit index arithmetic per four outputs, 2.2 TB/s
// on the stem's [192, 64, 112, 112] map).  Same scan order and strict '>' as above: the first maximum wins.
// grid (ceil(Ho * Wo/4 / 256), NC)
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, int C, int H, int W,
                                                           int Ho, int Wo, float* __restrict__ y, int* __restrict__ idx) {
  const int nc = blockIdx.y, c = nc % C, Wq = Wo >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Ho * Wq) return;
  const int ho = t / Wq, q = t - ho * Wq;
  const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f, slope = act_slope(act);
  const float* p = x + (long long)nc * H * W;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {-1, -1, -1, -1};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int h = 2 * ho - 1 + kh;
    if ((unsigned)h >= (unsigned)H) continue;
    const float* row = p + h * W + 8 * q;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row), b = *reinterpret_cast<const f32x4*>(row + 4);
    float v[9];
    v[0] = q > 0 ? row[-1] : 0.f;                          // column 8q - 1 (outside the map for q == 0: skipped below)
    v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w; v[5] = b.x; v[6] = b.y; v[7] = b.z; v[8] = b.w;
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = act_by_slope(fmaf(v[k], sc, sh), slope);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int k = 2 * j + kw;                           // input column 8q - 1 + k
        if (k == 0 && q == 0) continue;
        if (v[k] > best[j] || bi[j] < 0) { best[j] = v[k]; bi[j] = h * W + 8 * q - 1 + k; }
      }
  }
  const long long o = ((long long)nc * Ho + ho) * Wo + 4 * q;
  *reinterpret_cast<f32x4*>(y + o) = f32x4{best[0], best[1], best[2], best[3]};
  if (idx) *reinterpret_cast<int4*>(idx + o) = make_int4(bi[0], bi[1], bi[2], bi[3]);
}

extern "C" int avsep_maxpool3x3s2_fwd(const float* x, const float* scale, const float* shift, int32_t act, int32_t C,
                                      int32_t NC, int32_t H, int32_t W, float* y, int32_t* idx, avsep_stream_t stream) {
  if (!x || !y || NC <= 0 || H <= 0 || W <= 0 || C <= 0 || NC % C) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return AVSEP_ERR_ARG;
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if ((W & 7) == 0 && (H & 1) == 0 && NC <= 65535 && (long long)H * W < 0x7fffffffLL) {
    hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(cdiv(Ho * (Wo >> 2), 256), NC), dim3(256), 0, (hipStream_t)stream, x, scale, shift, act,
                       C, H, W, Ho, Wo, y, idx);
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  long long total = (long long)NC * Ho * Wo;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((int)min((total + 255) / 256, (long long)65536)), dim3(256), 0,
                     (hipStream_t)stream, x, scale, shift, act, C, NC, H, W, Ho, Wo, y, idx);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// gather form: each input pixel collects from the <=4 windows that contain it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, int NC,
                                                          int H, int W, int Ho, int Wo, float* __restrict__ dx) {
  const long long total = (long long)NC * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int w = (int)(i % W), h = (int)((i / W) % H);
    long long nc = i / ((long long)W * H);
    int self = h * W + w;
    float s = 0.f;
    // window ho covers input rows 2ho-1..2ho+1  =>  h/2 <= ho <= (h+1)/2
    for (int ho = h / 2; ho <= (h + 1) / 2 && ho < Ho; ++ho)
      for (int wo = w / 2; wo <= (w + 1) / 2 && wo < Wo; ++wo) {
        long long o = (nc * Ho + ho) * Wo + wo;
        if (idx[o] == self) s += dy[o];
      }
    dx[i] = s;
  }
}
extern "C" int avsep_maxpool3x3s2_bwd(const float* dy, const int32_t* idx, int32_t NC, int32_t H, int32_t W, float* dx,
                                      avsep_stream_t stream) {
  if (!dy || !idx || !dx || NC <= 0 || H <= 0 || W <= 0) return AVSEP_ERR_ARG;
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long long total = (long long)NC * H * W;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((int)min((total + 255) / 256, (long long)65536)), dim3(256), 0,
                     (hipStream_t)stream, dy, idx, NC, H, W, Ho, Wo, dx);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- stem as a stride-1 conv: space-to-depth of the frames ------------------------------------------------------------------
// conv 7x7 / stride 2 / pad 3 over [N,3,H,W] (vision_net.py:111: resnet conv1)  ==  conv 4x4 / stride 1 / pad 0 over
//   xs[n][(dy*2+dx)*C + c][i + 2][j + 2] = x[n][c][2i + dy][2j + dx]      ([N, Cp, H/2 + 3, W/2 + 3], zero elsewhere)
// with w'[co][(dy*2+dx)*C + c][a][b] = w[co][c][2a + dy - 1][2b + dx - 1] (zero where kh, kw leave 0..6): every tap of
// the 7x7 kernel appears exactly once.  Cp = 4*C rounded up to 16 gives the halo-patch kernels (f32 and bf16) a
// 16-channel K-tile instead of the im2col gather over 3 channels (38 TFLOP/s).
// One (n, phase channel q) plane per blockIdx.y: per-plane constants in scalar registers and 32-bit index arithmetic (the flat
// form spent four 64-bit divisions per element: 165 us per call for 270 MB at 192 frames).
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ x, int C, int Cp, int H, int W,
                                                              int planes, float* __restrict__ xs) {
  const int Hs = H / 2 + 3, Ws = W / 2 + 3, Hh = H / 2, Wh = W / 2;
  for (int plane = blockIdx.y; plane < planes; plane += gridDim.y) {
    const int q = plane % Cp, n = plane / Cp;
    float* const op = xs + (long long)plane * Hs * Ws;
    const bool live = q < 4 * C;
    const int c = q % C, dy = (q / C) >> 1, dx = (q / C) & 1;
    const float* const xp = x + ((long long)(n * C + (live ? c : 0)) * H + dy) * W + dx;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hs * Ws; i += gridDim.x * 256) {
      const int r = i / Ws, j = i - r * Ws, ii = r - 2, jj = j - 2;
      float v = 0.f;
      if (live && (unsigned)ii < (unsigned)Hh && (unsigned)jj < (unsigned)Wh) v = xp[2 * ii * W + 2 * jj];
      op[i] = v;
    }
  }
}
extern "C" int avsep_space_to_depth2(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cp, float* xs,
                                     avsep_stream_t stream) {
  if (!x || !xs || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || Cp < 4 * C) return AVSEP_ERR_ARG;
  if ((long long)N * Cp > 0x7fffffffLL || (long long)H * W > 0x3fffffffLL) return AVSEP_ERR_ARG;
  const int per = (H / 2 + 3) * (W / 2 + 3), gx = (per + 255) / 256 < 16 ? (per + 255) / 256 : 16;
  const int planes = N * Cp;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(gx, planes < 65535 ? planes : 65535), dim3(256), 0, (hipStream_t)stream, x, C, Cp,
                     H, W, planes, xs);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- stem tail backward, fused: MaxPool(3,2,1) backward + ReLU mask + BatchNorm backward of the stem conv output -----
// (vision_net.py:111-117 children 1-3 behind conv1).  g = dL/d(pooled) [N,C,Ho,Wo], idx = arg-max positions of the
// forward, y = RAW conv output [N,C,H,W], pre = scale*y + shift.  dz[pos] = [pre > 0] * sum of g over the windows that
// chose pos.  The separate kernels wrote dz (616 MB at 192 frames), re-read it for the ReLU mask + statistics and again
// for dy = p*dz + q*y + r: 1.8 ms per call; fused: the statistics are taken over the POOLED grid (linear in g), the
// apply pass writes dy directly: one read of y, one write of dy.
// pass 1: bstats[c] += sum dz, bstats[C+c] += sum dz * xhat   (grid: C x N, one plane per block)
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_stats_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                                        const float* __restrict__ y, const float* __restrict__ scale,
                                                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd, int C, int HW, int HoWo,
                                                                        double* __restrict__ bstats) {
  const int c = blockIdx.x, n = blockIdx.y;
  const long long plane = (long long)n * C + c;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  const float* gp = g + plane * HoWo;
  const int* ip = idx + plane * HoWo;
  const float* yp = y + plane * HW;
  float s1 = 0.f, s2 = 0.f;
  for (int o = threadIdx.x; o < HoWo; o += 256) {
    const float yv = yp[ip[o]];
    if (fmaf(yv, sc, sh) > 0.f) {
      const float gv = gp[o];
      s1 += gv;
      s2 = fmaf(gv, (yv - mu) * is, s2);
    }
  }
  double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
  __shared__ double sh_[8];
  if ((threadIdx.x & 63) == 0) { sh_[threadIdx.x >> 6] = d1; sh_[4 + (threadIdx.x >> 6)] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&bstats[c], sh_[0] + sh_[1] + sh_[2] + sh_[3]);
    atomicAdd(&bstats[C + c], sh_[4] + sh_[5] + sh_[6] + sh_[7]);
  }
}
// pass 2: dy[i] = p * ([pre > 0] * sum_{windows that chose i} g) + q * y[i] + r
// One (n, c) plane per blockIdx.y (per-channel constants in scalar registers, 32-bit indexing); a thread owns the pixel
// pair (h, 2j), (h, 2j+1): the even pixel lies in window column j only, the odd one in j and j+1, an even row in window row
// h/2 only, an odd one in (h-1)/2 and (h+1)/2 — at most four (index, gradient) pairs per thread, 8-byte loads and stores.
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_apply_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                                        const float* __restrict__ y, const float* __restrict__ scale,
                                                                        const float* __restrict__ shift, const float* __restrict__ pqr,
                                                                        int C, int H, int W, int Ho, int Wo, float* __restrict__ dy) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int nc = blockIdx.y, c = nc % C, Wh = W >> 1;
  const float sc = scale[c], sh = shift[c], cp = pqr[c], cq = pqr[C + c], cr = pqr[2 * C + c];
  const float* yp = y + (long long)nc * H * W;
  float* dp = dy + (long long)nc * H * W;
  const float* gp = g + (long long)nc * Ho * Wo;
  const int* ip = idx + (long long)nc * Ho * Wo;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * Wh; i += gridDim.x * 256) {
    const int h = i / Wh, j = i - h * Wh, w = 2 * j, self = h * W + w;
    const f32x2 yv = *reinterpret_cast<const f32x2*>(yp + self);
    float s0 = 0.f, s1 = 0.f;
    const int ho0 = h >> 1, nho = (h & 1) && ho0 + 1 < Ho ? 2 : 1;
    const bool right = j + 1 < Wo;
    for (int a = 0; a < nho; ++a) {
      const int o = (ho0 + a) * Wo + j;
      const int i0 = ip[o];
      const float g0 = gp[o];
      if (i0 == self) s0 += g0;
      if (i0 == self + 1) s1 += g0;
      if (right) {
        if (ip[o + 1] == self + 1) s1 += gp[o + 1];
      }
    }
    if (!(fmaf(yv[0], sc, sh) > 0.f)) s0 = 0.f;
    if (!(fmaf(yv[1], sc, sh) > 0.f)) s1 = 0.f;
    *reinterpret_cast<f32x2*>(dp + self) = f32x2{fmaf(cp, s0, fmaf(cq, yv[0], cr)), fmaf(cp, s1, fmaf(cq, yv[1], cr))};
  }
}
// W % 4 == 0: a thread owns the pixel quad (h, 4j .. 4j+3) — window columns 2j (pixels 0, 1), 2j+1 (pixels 1, 2, 3) and 2j+2
// (pixel 3): 16-byte loads and stores of y / dy (the pair form moves 1.5 GB per call at 3.3 TB/s on the stem's 112x112 maps)
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_apply4_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                                         const float* __restrict__ y, const float* __restrict__ scale,
                                                                         const float* __restrict__ shift, const float* __restrict__ pqr,
                                                                         int C, int H, int W, int Ho, int Wo, float* __restrict__ dy) {
  const int nc = blockIdx.y, c = nc % C, Wq = W >> 2;
  const float sc = scale[c], sh = shift[c], cp = pqr[c], cq = pqr[C + c], cr = pqr[2 * C + c];
  const float* yp = y + (long long)nc * H * W;
  float* dp = dy + (long long)nc * H * W;
  const float* gp = g + (long long)nc * Ho * Wo;
  const int* ip = idx + (long long)nc * Ho * Wo;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * Wq; i += gridDim.x * 256) {
    const int h = i / Wq, j = i - h * Wq, self = h * W + 4 * j;
    const f32x4 yv = *reinterpret_cast<const f32x4*>(yp + self);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int ho0 = h >> 1, nho = (h & 1) && ho0 + 1 < Ho ? 2 : 1;
    const bool third = 2 * j + 2 < Wo;
    for (int a = 0; a < nho; ++a) {
      const int o = (ho0 + a) * Wo + 2 * j;                        // (Wo = W / 2: columns 2j and 2j + 1 exist)
      const int i0 = ip[o], i1 = ip[o + 1];
      const float g0 = gp[o], g1 = gp[o + 1];
      if (i0 == self) s[0] += g0;
      if (i0 == self + 1) s[1] += g0;
      if (i1 == self + 1) s[1] += g1;
      if (i1 == self + 2) s[2] += g1;
      if (i1 == self + 3) s[3] += g1;
      if (third) {
        if (ip[o + 2] == self + 3) s[3] += gp[o + 2];
      }
    }
    f32x4 out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sk = fmaf(yv[k], sc, sh) > 0.f ? s[k] : 0.f;
      out[k] = fmaf(cp, sk, fmaf(cq, yv[k], cr));
    }
    *reinterpret_cast<f32x4*>(dp + self) = out;
  }
}
// odd widths: one pixel per thread
__global__ __launch_bounds__(256) void maxpool_bn_relu_bwd_apply1_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                                         const float* __restrict__ y, const float* __restrict__ scale,
                                                                         const float* __restrict__ shift, const float* __restrict__ pqr,
                                                                         int C, int NC, int H, int W, int Ho, int Wo,
                                                                         float* __restrict__ dy) {
  const long long total = (long long)NC * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int w = (int)(i % W), h = (int)((i / W) % H);
    const long long nc = i / ((long long)W * H);
    const int c = (int)(nc % C), self = h * W + w;
    const float yv = y[i];
    float s = 0.f;
    if (fmaf(yv, scale[c], shift[c]) > 0.f) {
      for (int ho = h / 2; ho <= (h + 1) / 2 && ho < Ho; ++ho)
        for (int wo = w / 2; wo <= (w + 1) / 2 && wo < Wo; ++wo) {
          const long long o = (nc * Ho + ho) * Wo + wo;
          if (idx[o] == self) s += g[o];
        }
    }
    dy[i] = fmaf(pqr[c], s, fmaf(pqr[C + c], yv, pqr[2 * C + c]));
  }
}
extern "C" int avsep_maxpool_bn_relu_bwd_stats(const float* g, const int32_t* idx, const float* y, const float* scale,
                                               const float* shift, const float* mean, const float* invstd, int32_t N, int32_t C,
                                               int32_t H, int32_t W, double* bstats, avsep_stream_t stream) {
  if (!g || !idx || !y || !scale || !shift || !mean || !invstd || !bstats) return AVSEP_ERR_ARG;
  if (N <= 0 || N > 65535 || C <= 0 || H <= 0 || W <= 0) return AVSEP_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bn_relu_bwd_stats_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, g, idx, y, scale, shift,
                     mean, invstd, C, H * W, Ho * Wo, bstats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
extern "C" int avsep_maxpool_bn_relu_bwd_apply(const float* g, const int32_t* idx, const float* y, const float* scale,
                                               const float* shift, const float* pqr, int32_t N, int32_t C, int32_t H, int32_t W,
                                               float* dy, avsep_stream_t stream) {
  if (!g || !idx || !y || !scale || !shift || !pqr || !dy || N <= 0 || C <= 0 || H <= 0 || W <= 0) return AVSEP_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * C * H * W;
  if ((W & 3) == 0 && (long long)N * C <= 65535 && (long long)H * W < 0x7fffffffLL) {
    const int per = H * (W / 4), gx = (per + 255) / 256 < 64 ? (per + 255) / 256 : 64;
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_apply4_kernel, dim3(gx, N * C), dim3(256), 0, (hipStream_t)stream, g, idx, y, scale,
                       shift, pqr, C, H, W, Ho, Wo, dy);
  } else if ((W & 1) == 0 && (long long)N * C <= 65535 && (long long)H * W < 0x7fffffffLL) {
    const int per = H * (W / 2), gx = (per + 255) / 256 < 64 ? (per + 255) / 256 : 64;
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_apply_kernel, dim3(gx, N * C), dim3(256), 0, (hipStream_t)stream, g, idx, y, scale,
                       shift, pqr, C, H, W, Ho, Wo, dy);
  } else {
    hipLaunchKernelGGL(maxpool_bn_relu_bwd_apply1_kernel, dim3((int)min((total + 255) / 256, (long long)262144)), dim3(256), 0,
                       (hipStream_t)stream, g, idx, y, scale, shift, pqr, C, N * C, H, W, Ho, Wo, dy);
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void temporal_mean_fwd_kernel(const float* __restrict__ x, int B, int T, long long CHW,
                                                                float* __restrict__ y) {
  const long long total = (long long)B * CHW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long b = i / CHW, r = i % CHW;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += x[(b * T + t) * CHW + r];
    y[i] = s / (float)T;
  }
}
extern "C" int avsep_temporal_mean_fwd(const float* x, int32_t B, int32_t T, int32_t CHW, float* y,
                                       avsep_stream_t stream) {
  if (!x || !y || B <= 0 || T <= 0 || CHW <= 0) return AVSEP_ERR_ARG;
  long long total = (long long)B * CHW;
  hipLaunchKernelGGL(temporal_mean_fwd_kernel, dim3((int)min((total + 255) / 256, (long long)65536)), dim3(256), 0,
                     (hipStream_t)stream, x, B, T, (long long)CHW, y);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
__global__ __launch_bounds__(256) void temporal_mean_bwd_kernel(const float* __restrict__ dy, int B, int T, long long CHW,
                                                                float* __restrict__ dx) {
  const long long total = (long long)B * T * CHW;
  const float inv = 1.f / (float)T;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    long long bt = i / CHW, r = i % CHW;
    dx[i] = dy[(bt / T) * CHW + r] * inv;
  }
}
extern "C" int avsep_temporal_mean_bwd(const float* dy, int32_t B, int32_t T, int32_t CHW, float* dx,
                                       avsep_stream_t stream) {
  if (!dy || !dx || B <= 0 || T <= 0 || CHW <= 0) return AVSEP_ERR_ARG;
  long long total = (long long)B * T * CHW;
  hipLaunchKernelGGL(temporal_mean_bwd_kernel, dim3((int)min((total + 255) / 256, (long long)65536)), dim3(256), 0,
                     (hipStream_t)stream, dy, B, T, (long long)CHW, dx);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  size_t n, float lr, float mom, float wd, float gs, int first) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float pv = p[i];
    float d = fmaf(wd, pv, g[i] * gs);
    float b = first ? d : fmaf(mom, buf[i], d);
    buf[i] = b;
    p[i] = pv - lr * b;
  }
}
extern "C" int avsep_sgd_momentum(float* p, const float* g, float* buf, size_t n, float lr, float momentum,
                                  float weight_decay, float grad_scale, int32_t first, avsep_stream_t stream) {
  if (!p || !g || !buf) return AVSEP_ERR_ARG;
  if (n == 0) return AVSEP_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)min((n + 255) / 256, (size_t)16384)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                     n, lr, momentum, weight_decay, grad_scale, first);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void innerprod_kernel(const float* __restrict__ img, const float* __restrict__ snd,
                                                        const float* __restrict__ scale, const float* __restrict__ bias,
                                                        int K, int HW, float* __restrict__ z) {
  const int b = blockIdx.y;
  extern __shared__ float s_w[];
  for (int k = threadIdx.x; k < K; k += 256) s_w[k] = img[(long long)b * K + k] * (scale ? scale[k] : 1.f);
  __syncthreads();
  const float bs = bias ? bias[0] : 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(s_w[k], snd[((long long)b * K + k) * HW + i], s);
    z[(long long)b * HW + i] = s + bs;
  }
}
extern "C" int avsep_innerprod_fwd(const float* img, const float* snd, const float* scale, const float* bias, int32_t B,
                                   int32_t K, int32_t HW, float* z, avsep_stream_t stream) {
  if (!img || !snd || !z || B <= 0 || B > 65535 || K <= 0 || K > 8192 || HW <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(innerprod_kernel, dim3(min(cdiv(HW, 256), 256), B), dim3(256), K * sizeof(float), (hipStream_t)stream,
                     img, snd, scale, bias, K, HW, z);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// backward of the synthesizer GEMV: dsnd[b,k,hw] = w[b,k]*dz[b,hw];  r[b,k] = sum_hw snd[b,k,hw]*dz[b,hw]
// (w = img*scale; the caller forms dimg = scale*r, dscale = sum_b img*r, dbias = sum dz from r and dz)
__global__ __launch_bounds__(256) void innerprod_bwd_kernel(const float* __restrict__ img, const float* __restrict__ snd,
                                                            const float* __restrict__ scale, const float* __restrict__ dz,
                                                            int K, int HW, float* __restrict__ dsnd, float* __restrict__ r) {
  const int k = blockIdx.x, b = blockIdx.y;
  const float w = img[(long long)b * K + k] * (scale ? scale[k] : 1.f);
  const float* sp = snd + ((long long)b * K + k) * HW;
  const float* dp = dz + (long long)b * HW;
  float* op = dsnd ? dsnd + ((long long)b * K + k) * HW : nullptr;
  float acc = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    float d = dp[i];
    acc = fmaf(sp[i], d, acc);
    if (op) op[i] = w * d;
  }
  acc = wave_sum(acc);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) r[(long long)b * K + k] = sh[0] + sh[1] + sh[2] + sh[3];
}
extern "C" int avsep_innerprod_bwd(const float* img, const float* snd, const float* scale, const float* dz, int32_t B,
                                   int32_t K, int32_t HW, float* dsnd, float* r, avsep_stream_t stream) {
  if (!img || !snd || !dz || !r || B <= 0 || B > 65535 || K <= 0 || HW <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(innerprod_bwd_kernel, dim3(K, B), dim3(256), 0, (hipStream_t)stream, img, snd, scale, dz, K, HW, dsnd, r);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- the synthesizer's inference helpers (models/synthesizer_net.py:21-38) ------------------------------------------------
// forward_nosum: z[b,k,hw] = img[b,k]*scale[k]*snd[b,k,hw] + bias   (one pass, 16-byte accesses when HW % 4 == 0)
__global__ __launch_bounds__(256) void innerprod_nosum_kernel(const float* __restrict__ img, const float* __restrict__ snd,
                                                              const float* __restrict__ scale, const float* __restrict__ bias,
                                                              int K, int HW, float* __restrict__ z) {
  const int bk = blockIdx.y, k = bk % K;
  const float w = img[bk] * (scale ? scale[k] : 1.f), bs = bias ? bias[0] : 0.f;
  const long long base = (long long)bk * HW;
  if ((HW & 3) == 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(snd + base);
    f32x4* z4 = reinterpret_cast<f32x4*>(z + base);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
      const f32x4 v = s4[i];
      z4[i] = f32x4{fmaf(w, v.x, bs), fmaf(w, v.y, bs), fmaf(w, v.z, bs), fmaf(w, v.w, bs)};
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) z[base + i] = fmaf(w, snd[base + i], bs);
  }
}
extern "C" int avsep_innerprod_nosum(const float* img, const float* snd, const float* scale, const float* bias, int32_t B, int32_t K,
                                     int32_t HW, float* z, avsep_stream_t stream) {
  if (!img || !snd || !z || B <= 0 || K <= 0 || HW <= 0 || (long long)B * K > 65535) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(innerprod_nosum_kernel, dim3(min(cdiv(HW, 1024), 64), B * K), dim3(256), 0, (hipStream_t)stream, img, snd, scale,
                     bias, K, HW, z);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// forward_pixelwise: z[b,p,hw] = sum_k imgs[b,k,p]*scale[k]*snd[b,k,hw] + bias — per sample a [P x K] x [K x HW] contraction
// (the visual-feature x audio-feature mask inner product) on v_mfma_f32_32x32x2_f32.  Workgroup = 128 audio positions of one
// sample (wave w: 32 of them); the scaled visual vectors [K][P] sit in LDS; a wave walks the 32-row tiles of P, 16 MFMAs
// (K / 2 k-steps) each, and stores 32 x 32 outputs as 128-byte row segments.  HBM-bound on the [B, P, HW] output.
__global__ __launch_bounds__(256) void innerprod_pixelwise_kernel(const float* __restrict__ imgs, const float* __restrict__ snd,
                                                                  const float* __restrict__ scale, const float* __restrict__ bias,
                                                                  int K, int P, int HW, float* __restrict__ z) {
  extern __shared__ float s_a[];                   // [K][PS], PS = roundup(P, 32) + 1
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int PT = (P + 31) / 32, PS = PT * 32 + 1;
  for (int i = tid; i < K * PT * 32; i += 256) {
    const int k = i / (PT * 32), p = i % (PT * 32);
    s_a[k * PS + p] = p < P ? imgs[((long long)b * K + k) * P + p] * (scale ? scale[k] : 1.f) : 0.f;
  }
  __syncthreads();
  const int hw = blockIdx.x * 128 + wave * 32 + li;
  const int hwc = min(hw, HW - 1);
  const float bs = bias ? bias[0] : 0.f;
  const float* sp = snd + (long long)b * K * HW + hwc;
  for (int pt = 0; pt < PT; ++pt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < K / 2; ++s) {
      const float av = s_a[(2 * s + lk) * PS + pt * 32 + li];
      const float bv = sp[(long long)(2 * s + lk) * HW];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (hw < HW) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = pt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (p < P) z[((long long)b * P + p) * HW + hw] = acc[r] + bs;
      }
    }
  }
}
extern "C" int avsep_innerprod_pixelwise(const float* imgs, const float* snd, const float* scale, const float* bias, int32_t B,
                                         int32_t K, int32_t P, int32_t HW, float* z, avsep_stream_t stream) {
  if (!imgs || !snd || !z || B <= 0 || B > 65535 || K <= 0 || (K & 1) || P <= 0 || HW <= 0) return AVSEP_ERR_ARG;
  const size_t smem = (size_t)K * (roundup(P, 32) + 1) * sizeof(float);
  if (smem > 160 * 1024) return AVSEP_ERR_ARG;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)innerprod_pixelwise_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(innerprod_pixelwise_kernel, dim3(cdiv(HW, 128), B), dim3(256), smem, (hipStream_t)stream, imgs, snd, scale, bias,
                     K, P, HW, z);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// Separation metrics (main.py:260-266 via asteroid.metrics.get_metrics): the three inner products every
// SDR-type ratio is made of, per row: sums[r] = (<est,ref>, <ref,ref>, <est,est>) in fp64.  grid (chunks, R)
__global__ __launch_bounds__(256) void sdr_sums_kernel(const float* __restrict__ est, const float* __restrict__ ref,
                                                       int L, long long est_stride, long long ref_stride,
                                                       double* __restrict__ sums) {
  const int r = blockIdx.y;
  const float* e = est + (long long)r * est_stride;
  const float* g = ref + (long long)r * ref_stride;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L; i += gridDim.x * 256) {
    double x = e[i], y = g[i];
    a += x * y; b += y * y; c += x * x;
  }
  a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
  __shared__ double sh[12];
  if ((threadIdx.x & 63) == 0) { int w = threadIdx.x >> 6; sh[w] = a; sh[4 + w] = b; sh[8 + w] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[r * 3 + 0], sh[0] + sh[1] + sh[2] + sh[3]);
    atomicAdd(&sums[r * 3 + 1], sh[4] + sh[5] + sh[6] + sh[7]);
    atomicAdd(&sums[r * 3 + 2], sh[8] + sh[9] + sh[10] + sh[11]);
  }
}
extern "C" int avsep_sdr_sums(const float* est, const float* ref, int32_t R, int32_t L, int64_t est_stride,
                              int64_t ref_stride, double* sums, avsep_stream_t stream) {
  if (!est || !ref || !sums || R <= 0 || R > 65535 || L <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(sdr_sums_kernel, dim3(min(cdiv(L, 2048), 64), R), dim3(256), 0, (hipStream_t)stream, est, ref, L,
                     (long long)est_stride, (long long)ref_stride, sums);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

extern "C" int avsep_version(void) { return 100; }
extern "C" const char* avsep_arch(void) { return "gfx950"; }
extern "C" const char* avsep_strerror(int code) {
  switch (code) {
    case AVSEP_OK: return "ok";
    case AVSEP_ERR_ARG: return "invalid argument (shape, null pointer or unsupported geometry)";
    case AVSEP_ERR_LAUNCH: return "kernel launch failed";
    case AVSEP_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}
