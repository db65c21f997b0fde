// Glue kernels on B16 images (AVSEP_FMT_B16: bf16, [N][C/16][H][W][16 channels], include/avsep.h) — the activations and
// gradients that travel between the bf16 convolution kernels of BASELINE.json configs[2].  Every kernel here is one
// HBM pass with 16-byte accesses: a thread owns one (position, 8-channel half) slot = one u32x4, consecutive threads own
// consecutive slots, a workgroup stays inside one 16-channel block so that the per-channel BatchNorm rows are eight
// registers per thread.  Arithmetic is fp32 on the unpacked values, statistics are fp32 per thread -> fp64 atomics.
//   f32 <-> B16 conversion (boundaries to the fp32 kernels: decoder head, fusion, tiny maps)
//   BasicBlock tail      z = relu(bn2(y2) + (bnd(yd) | z))                     torchvision BasicBlock.forward
//   its backward         g = relu'(.) * (dz [+ dz2]) [+ add], sums for the BatchNorm backward   (also LeakyReLU: audio_net.py:64)
//   BatchNorm backward   dy = p*dz + q*y + r                                     (folded form, see ops.hip)
//   ReLU + bilinear x2 + concat of the U-Net decoder and its adjoint             audio_net.py:66-69,122,203
//   stem tail            max-pool 3x3/s2 over relu(bn(y0)) and its fused backward  vision_net.py:111-117
//   space-to-depth of the frames straight into a one-block B16 image
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned b16_pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void b16_unpack8(u32x4 q, float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __builtin_bit_cast(float, q[k] << 16);
    v[2 * k + 1] = __builtin_bit_cast(float, q[k] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 b16_pack8(const float (&v)[8]) {
  return u32x4{b16_pack2(v[0], v[1]), b16_pack2(v[2], v[3]), b16_pack2(v[4], v[5]), b16_pack2(v[6], v[7])};
}
__device__ __forceinline__ void b16_row8(const float* p, int c0, float dflt, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = p ? p[c0 + j] : dflt;
}

// Per-channel sums of a workgroup whose threads own the 8-channel half (threadIdx.x & 1) of channel block `cb`:
// lanes of equal parity are summed by xor-shuffles, the four waves through LDS, then one fp64 atomic per channel and sum.
__device__ __forceinline__ void b16_block_stats(float (&s1)[8], float (&s2)[8], int cb, int C, double* bstats) {
  __shared__ float red[4][2][16];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) {
      s1[j] += __shfl_xor(s1[j], o, 64);
      s2[j] += __shfl_xor(s2[j], o, 64);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[wave][lane][j] = s1[j];
      red[wave][lane][8 + j] = s2[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int half = threadIdx.x >> 4, k = threadIdx.x & 15;
    const double t = (double)red[0][half][k] + (double)red[1][half][k] + (double)red[2][half][k] + (double)red[3][half][k];
    atomicAdd(&bstats[(k < 8 ? 0 : C) + cb * 16 + half * 8 + (k & 7)], t);
  }
}

// ---- conversions ------------------------------------------------------------------------------------------------------------
// grid (position chunks, C/16, N): a thread converts the 16 channels of one position (16 coalesced 4-byte streams <-> 32 bytes)
__global__ __launch_bounds__(256) void f32_to_b16_kernel(const float* __restrict__ x, int C, int HW, u32x4* __restrict__ out) {
  const int cb = blockIdx.y, CB = C >> 4, n = blockIdx.z;                 // one image per z (N <= 65535: host-checked)
  const float* xp = x + ((long long)n * C + cb * 16) * HW;
  u32x4* op = out + ((long long)n * CB + cb) * HW * 2;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = xp[(long long)j * HW + p];
    op[2 * p] = u32x4{b16_pack2(v[0], v[1]), b16_pack2(v[2], v[3]), b16_pack2(v[4], v[5]), b16_pack2(v[6], v[7])};
    op[2 * p + 1] = u32x4{b16_pack2(v[8], v[9]), b16_pack2(v[10], v[11]), b16_pack2(v[12], v[13]), b16_pack2(v[14], v[15])};
  }
}
__global__ __launch_bounds__(256) void b16_to_f32_kernel(const u32x4* __restrict__ x, int C, int HW, float* __restrict__ out) {
  const int cb = blockIdx.y, CB = C >> 4, n = blockIdx.z;
  const u32x4* xp = x + ((long long)n * CB + cb) * HW * 2;
  float* op = out + ((long long)n * C + cb * 16) * HW;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float a[8], b[8];
    b16_unpack8(xp[2 * p], a);
    b16_unpack8(xp[2 * p + 1], b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      op[(long long)j * HW + p] = a[j];
      op[(long long)(8 + j) * HW + p] = b[j];
    }
  }
}
// dy = p*dz + q*y + r on fp32 NCHW inputs, written as a B16 image: the BatchNorm backward at the fp32 -> B16 boundary below the
// fused decoder head (its fp32 result was written, read back and converted: two full passes more)
__global__ __launch_bounds__(256) void f32_bn_bwd_apply_to_b16_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                                      const float* __restrict__ pqr, int C, int HW,
                                                                      u32x4* __restrict__ out) {
  const int cb = blockIdx.y, CB = C >> 4, n = blockIdx.z;
  const long long base = ((long long)n * C + cb * 16) * HW;
  u32x4* op = out + ((long long)n * CB + cb) * HW * 2;
  float cp[16], cq[16], cr[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { cp[j] = pqr[cb * 16 + j]; cq[j] = pqr[C + cb * 16 + j]; cr[j] = pqr[2 * C + cb * 16 + j]; }
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = fmaf(cp[j], dz[base + (long long)j * HW + p], fmaf(cq[j], y[base + (long long)j * HW + p], cr[j]));
    op[2 * p] = u32x4{b16_pack2(v[0], v[1]), b16_pack2(v[2], v[3]), b16_pack2(v[4], v[5]), b16_pack2(v[6], v[7])};
    op[2 * p + 1] = u32x4{b16_pack2(v[8], v[9]), b16_pack2(v[10], v[11]), b16_pack2(v[12], v[13]), b16_pack2(v[14], v[15])};
  }
}
static bool b16_dims_ok(int N, int C, long long HW);
extern "C" int avsep_bn_bwd_apply_to_b16(const float* dz, const float* y, const float* pqr, int32_t N, int32_t C, int32_t HW, void* out,
                                         avsep_stream_t stream) {
  if (!dz || !y || !pqr || !out || !b16_dims_ok(N, C, HW)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(f32_bn_bwd_apply_to_b16_kernel, dim3(min(cdiv(HW, 256), 64), C / 16, N), dim3(256), 0, (hipStream_t)stream, dz, y,
                     pqr, C, HW, (u32x4*)out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
static bool b16_dims_ok(int N, int C, long long HW) {
  return N > 0 && N <= 65535 && C > 0 && C % 16 == 0 && C / 16 <= 65535 && HW > 0 && HW < (1LL << 30);
}
extern "C" int avsep_f32_to_b16(const float* x, int32_t N, int32_t C, int32_t HW, void* out, avsep_stream_t stream) {
  if (!x || !out || !b16_dims_ok(N, C, HW)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(f32_to_b16_kernel, dim3(min(cdiv(HW, 256), 64), C / 16, N), dim3(256), 0, (hipStream_t)stream, x, C, HW,
                     (u32x4*)out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
extern "C" int avsep_b16_to_f32(const void* x, int32_t N, int32_t C, int32_t HW, float* out, avsep_stream_t stream) {
  if (!x || !out || !b16_dims_ok(N, C, HW)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(b16_to_f32_kernel, dim3(min(cdiv(HW, 256), 64), C / 16, N), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)x, C, HW, out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- elementwise passes: grid (slot chunks, C/16, image slices) ---------------------------------------------------------------
struct B16Grid { dim3 g; };
static B16Grid b16_grid(int N, int C, long long HW) {
  const int gx = (int)min((HW * 2 + 255) / 256, (long long)32);
  int gz = N;
  const long long want = 4096;                        // enough workgroups for 256 CUs without one per image on big batches
  if ((long long)gx * (C / 16) * gz > want) gz = (int)max(1LL, want / ((long long)gx * (C / 16)));
  if (gz > N) gz = N;
  return B16Grid{dim3(gx, C / 16, gz)};
}

// z = act(scale*y + shift [+ res_scale*res + res_shift | + res])
__global__ __launch_bounds__(256) void b16_affine_act_kernel(const u32x4* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const u32x4* __restrict__ res,
                                                             const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                             int act, int N, int C, int HW, u32x4* __restrict__ z) {
  const int cb = blockIdx.y, CB = C >> 4, c0 = cb * 16 + (threadIdx.x & 1) * 8, S = HW * 2;
  float sc[8], sh[8], rs[8], rh[8];
  b16_row8(scale, c0, 1.f, sc); b16_row8(shift, c0, 0.f, sh); b16_row8(rscale, c0, 1.f, rs); b16_row8(rshift, c0, 0.f, rh);
  const float slope = act_slope(act);
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const long long base = ((long long)n * CB + cb) * S;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
      float v[8], r[8];
      b16_unpack8(y[base + s], v);
      if (res) b16_unpack8(res[base + s], r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float pre = fmaf(v[j], sc[j], sh[j]);
        if (res) pre += fmaf(r[j], rs[j], rh[j]);
        v[j] = act_by_slope(pre, slope);
      }
      z[base + s] = b16_pack8(v);
    }
  }
}
extern "C" int avsep_b16_affine_act(const void* y, const float* scale, const float* shift, const void* residual,
                                    const float* res_scale, const float* res_shift, int32_t act, int32_t N, int32_t C, int32_t HW,
                                    void* z, avsep_stream_t stream) {
  if (!y || !z || !b16_dims_ok(N, C, HW)) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr) || (res_scale == nullptr) != (res_shift == nullptr)) return AVSEP_ERR_ARG;
  if (act != AVSEP_ACT_NONE && act != AVSEP_ACT_RELU && act != AVSEP_ACT_LRELU02) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(b16_affine_act_kernel, b16_grid(N, C, HW).g, dim3(256), 0, (hipStream_t)stream, (const u32x4*)y, scale, shift,
                     (const u32x4*)residual, res_scale, res_shift, act, N, C, HW, (u32x4*)z);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// g = act'(scale*y + shift [+ residual term]) * (dz [+ dz2]) [+ add]  ->  out (may alias dz; NULL = statistics only);
// bstats += (sum g, sum g * (y - mean) * invstd)
__global__ __launch_bounds__(256) void b16_affine_act_bwd_kernel(const u32x4* __restrict__ dz, const u32x4* __restrict__ dz2,
                                                                 const u32x4* __restrict__ y, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const u32x4* __restrict__ res,
                                                                 const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                                 const u32x4* __restrict__ add, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, int act, int N, int C, int HW,
                                                                 u32x4* out, double* bstats) {
  const int cb = blockIdx.y, CB = C >> 4, c0 = cb * 16 + (threadIdx.x & 1) * 8, S = HW * 2;
  float sc[8], sh[8], rs[8], rh[8], mu[8], is[8];
  b16_row8(scale, c0, 1.f, sc); b16_row8(shift, c0, 0.f, sh); b16_row8(rscale, c0, 1.f, rs); b16_row8(rshift, c0, 0.f, rh);
  b16_row8(mean, c0, 0.f, mu); b16_row8(invstd, c0, 1.f, is);
  const float neg = act == AVSEP_ACT_RELU ? 0.f : (act == AVSEP_ACT_LRELU02 ? 0.2f : 1.f);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const long long base = ((long long)n * CB + cb) * S;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
      float v[8], d[8], d2[8], r[8], ad[8];
      b16_unpack8(y[base + s], v);
      b16_unpack8(dz[base + s], d);
      if (dz2) b16_unpack8(dz2[base + s], d2);
      if (res) b16_unpack8(res[base + s], r);
      if (add) b16_unpack8(add[base + s], ad);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float pre = fmaf(v[j], sc[j], sh[j]);
        if (res) pre += fmaf(r[j], rs[j], rh[j]);
        float g = d[j];
        if (dz2) g += d2[j];
        g *= pre > 0.f ? 1.f : neg;
        if (add) g += ad[j];
        d[j] = g;
        s1[j] += g;
        s2[j] = fmaf(g, (v[j] - mu[j]) * is[j], s2[j]);
      }
      if (out) out[base + s] = b16_pack8(d);
    }
  }
  if (bstats) b16_block_stats(s1, s2, cb, C, bstats);
}
extern "C" int avsep_b16_affine_act_bwd(const void* dz, const void* dz2, const void* y, const float* scale, const float* shift,
                                        const void* residual, const float* res_scale, const float* res_shift, const void* add,
                                        const float* mean, const float* invstd, int32_t act, int32_t N, int32_t C, int32_t HW,
                                        void* out, double* bstats, avsep_stream_t stream) {
  if (!dz || !y || (!out && !bstats) || !b16_dims_ok(N, C, HW)) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr) || (res_scale == nullptr) != (res_shift == nullptr)) return AVSEP_ERR_ARG;
  if (bstats && (!mean || !invstd)) return AVSEP_ERR_ARG;
  if (act != AVSEP_ACT_NONE && act != AVSEP_ACT_RELU && act != AVSEP_ACT_LRELU02) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(b16_affine_act_bwd_kernel, b16_grid(N, C, HW).g, dim3(256), 0, (hipStream_t)stream, (const u32x4*)dz,
                     (const u32x4*)dz2, (const u32x4*)y, scale, shift, (const u32x4*)residual, res_scale, res_shift,
                     (const u32x4*)add, mean, invstd, act, N, C, HW, (u32x4*)out, bstats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// dy = p*dz + q*y + r  (pqr = rows p, q, r of avsep_bn_bwd_coeffs)
__global__ __launch_bounds__(256) void b16_bn_bwd_apply_kernel(const u32x4* __restrict__ dz, const u32x4* __restrict__ y,
                                                               const float* __restrict__ pqr, int N, int C, int HW, u32x4* out) {
  const int cb = blockIdx.y, CB = C >> 4, c0 = cb * 16 + (threadIdx.x & 1) * 8, S = HW * 2;
  float cp[8], cq[8], cr[8];
  b16_row8(pqr, c0, 0.f, cp); b16_row8(pqr + C, c0, 0.f, cq); b16_row8(pqr + 2 * C, c0, 0.f, cr);
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const long long base = ((long long)n * CB + cb) * S;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
      float v[8], d[8];
      b16_unpack8(y[base + s], v);
      b16_unpack8(dz[base + s], d);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = fmaf(cp[j], d[j], fmaf(cq[j], v[j], cr[j]));
      out[base + s] = b16_pack8(d);
    }
  }
}
extern "C" int avsep_b16_bn_bwd_apply(const void* dz, const void* y, const float* pqr, int32_t N, int32_t C, int32_t HW, void* out,
                                      avsep_stream_t stream) {
  if (!dz || !y || !pqr || !out || !b16_dims_ok(N, C, HW)) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(b16_bn_bwd_apply_kernel, b16_grid(N, C, HW).g, dim3(256), 0, (hipStream_t)stream, (const u32x4*)dz,
                     (const u32x4*)y, pqr, N, C, HW, (u32x4*)out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// per-channel sum of a B16 image (bias gradient): out[c] = sum over n, hw
__global__ __launch_bounds__(256) void b16_channel_sum_kernel(const u32x4* __restrict__ x, int N, int C, int HW, double* acc) {
  const int cb = blockIdx.y, CB = C >> 4, S = HW * 2;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const long long base = ((long long)n * CB + cb) * S;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
      float v[8];
      b16_unpack8(x[base + s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s1[j] += v[j];
    }
  }
  b16_block_stats(s1, s2, cb, C, acc);
}
__global__ void b16_sum_finish_kernel(const double* acc, int C, float* out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) out[c] = (float)acc[c];
}
// acc: 2*C zeroed doubles of workspace
int b16_channel_sum(const void* x, int N, int C, int HW, double* acc, float* out, hipStream_t st) {
  if (hipMemsetAsync(acc, 0, (size_t)2 * C * sizeof(double), st) != hipSuccess) return AVSEP_ERR_LAUNCH;
  hipLaunchKernelGGL(b16_channel_sum_kernel, b16_grid(N, C, HW).g, dim3(256), 0, st, (const u32x4*)x, N, C, HW, acc);
  hipLaunchKernelGGL(b16_sum_finish_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, acc, C, out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- U-Net decoder glue: out = up2x(relu(affine(cat(x0, x1)))), bilinear, align_corners=True (audio_net.py:66-69,122) ------------
// grid (hi-res slot chunks, (C0 + C1)/16, image slices).  Both sources are B16 [N][C/16][H][W][16]; out [N][(C0+C1)/16][2H][2W][16].
__global__ __launch_bounds__(256) void b16_relu_up2x_fwd_kernel(const u32x4* __restrict__ x0, const u32x4* __restrict__ x1,
                                                                const float* __restrict__ sc0, const float* __restrict__ sh0,
                                                                const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                                int N, int C0, int C1, int H, int W, float rh, float rw,
                                                                u32x4* __restrict__ out) {
  const int CB0 = C0 >> 4, CB1 = C1 >> 4, cbo = blockIdx.y, half = threadIdx.x & 1;
  const bool first = cbo < CB0;
  const int cbs = first ? cbo : cbo - CB0, CBs = first ? CB0 : CB1;
  const u32x4* xs = first ? x0 : x1;
  float sc[8], sh[8];
  b16_row8(first ? sc0 : sc1, cbs * 16 + half * 8, 1.f, sc);
  b16_row8(first ? sh0 : sh1, cbs * 16 + half * 8, 0.f, sh);
  const int Ho = 2 * H, Wo = 2 * W, So = Ho * Wo * 2;
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const u32x4* xp = xs + ((long long)n * CBs + cbs) * H * W * 2;
    u32x4* op = out + ((long long)n * (CB0 + CB1) + cbo) * So;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < So; s += gridDim.x * 256) {
      const int p = s >> 1, ho = p / Wo, wo = p - ho * Wo;
      const float fh = rh * (float)ho, fw = rw * (float)wo;
      const int h0 = (int)fh, w0 = (int)fw;
      const int h1 = h0 + (h0 < H - 1), w1 = w0 + (w0 < W - 1);
      const float lh = fh - (float)h0, lw = fw - (float)w0;
      float a[8], b[8], c[8], d[8];
      b16_unpack8(xp[(h0 * W + w0) * 2 + half], a);
      b16_unpack8(xp[(h0 * W + w1) * 2 + half], b);
      b16_unpack8(xp[(h1 * W + w0) * 2 + half], c);
      b16_unpack8(xp[(h1 * W + w1) * 2 + half], d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float va = fmaxf(fmaf(a[j], sc[j], sh[j]), 0.f), vb = fmaxf(fmaf(b[j], sc[j], sh[j]), 0.f);
        const float vc = fmaxf(fmaf(c[j], sc[j], sh[j]), 0.f), vd = fmaxf(fmaf(d[j], sc[j], sh[j]), 0.f);
        const float top = va + lw * (vb - va), bot = vc + lw * (vd - vc);
        a[j] = top + lh * (bot - top);
      }
      op[s] = b16_pack8(a);
    }
  }
}
extern "C" int avsep_b16_relu_up2x_fwd(const void* x0, const void* x1, const float* sc0, const float* sh0, const float* sc1,
                                       const float* sh1, int32_t N, int32_t C0, int32_t C1, int32_t H, int32_t W, void* out,
                                       avsep_stream_t stream) {
  if (!x0 || !out || N <= 0 || N > 65535 || C0 <= 0 || C0 % 16 || C1 < 0 || C1 % 16 || (C1 > 0 && !x1) || H <= 0 || W <= 0 ||
      (long long)H * W >= (1LL << 27))
    return AVSEP_ERR_ARG;
  if ((sc0 == nullptr) != (sh0 == nullptr) || (sc1 == nullptr) != (sh1 == nullptr)) return AVSEP_ERR_ARG;
  const float rh = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  hipLaunchKernelGGL(b16_relu_up2x_fwd_kernel, b16_grid(N, C0 + C1, 4LL * H * W).g, dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)x0, (const u32x4*)x1, sc0, sh0, sc1, sh1, N, C0, C1, H, W, rh, rw, (u32x4*)out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// adjoint: for every low-res position the <= 6x6 hi-res taps of dout that interpolate from it, masked by relu'(affine(src)).
// g0 (source 0, optionally accumulated into: the shared-encoder AV step) and g1 (source 1) are B16; bstats1 += the
// BatchNorm-backward sums of source 1 (sum g1, sum g1 * xhat(x1)).
// Workgroup = a TH x TW tile of low-res positions of ONE 16-channel block of one image; the (2TH+4) x (2TW+4) hi-res window
// is staged in LDS (16-byte slots), then a thread = (position, half) gathers its 36 taps with ds_read_b128.
constexpr int B16U_TH = 8, B16U_TW = 16, B16U_RH = 2 * B16U_TH + 4, B16U_RW = 2 * B16U_TW + 4;   // 20 x 36 window (RW even)
__device__ __forceinline__ void b16_taps6(int h, int Hin, float r, float (&wt)[6]) {     // = up2x_taps6 of ops.hip
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int ho = 2 * h - 2 + k;
    float w = 0.f;
    if (ho >= 0 && ho < 2 * Hin) {
      const float f = r * (float)ho;
      const int h0 = (int)f, h1 = h0 + (h0 < Hin - 1);
      const float l = f - (float)h0;
      w = (h0 == h ? 1.f - l : 0.f) + (h1 == h ? l : 0.f);
    }
    wt[k] = w;
  }
}
__global__ __launch_bounds__(256) void b16_relu_up2x_bwd_kernel(const u32x4* __restrict__ x0, const u32x4* __restrict__ x1,
                                                                const float* __restrict__ sc0, const float* __restrict__ sh0,
                                                                const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                                int N, int C0, int C1, int H, int W, float rh, float rw,
                                                                const u32x4* __restrict__ dout, u32x4* g0, u32x4* g1,
                                                                const float* __restrict__ mean1, const float* __restrict__ invstd1,
                                                                double* bstats1, int acc0, int tilesX) {
  // hi-res window, even and odd columns in separate planes: the 32 lanes of a row read columns 2tx + x, i.e. CONSECUTIVE
  // slots of one plane (conflict-free ds_read_b128; one interleaved image is a 2-way conflict on every tap)
  constexpr int HWD = B16U_RW / 2, PLANE = B16U_RH * HWD * 2, SLOTS = B16U_RH * B16U_RW * 2, LE = (SLOTS + 255) / 256;
  __shared__ u32x4 tile[2 * PLANE];
  const int CB0 = C0 >> 4, CB1 = C1 >> 4, cbo = blockIdx.y, half = threadIdx.x & 1, pos = threadIdx.x >> 1;
  const bool first = cbo < CB0;
  const int cbs = first ? cbo : cbo - CB0, CBs = first ? CB0 : CB1, c0 = cbs * 16 + half * 8;
  const u32x4* xs = first ? x0 : x1;
  u32x4* g = first ? g0 : g1;
  float sc[8], sh[8], mu[8], is[8];
  b16_row8(first ? sc0 : sc1, c0, 1.f, sc);
  b16_row8(first ? sh0 : sh1, c0, 0.f, sh);
  b16_row8(first ? nullptr : mean1, c0, 0.f, mu);
  b16_row8(first ? nullptr : invstd1, c0, 1.f, is);
  const int ty = pos / B16U_TW, tx = pos % B16U_TW;
  const int h0 = (blockIdx.x / tilesX) * B16U_TH, w0 = (blockIdx.x % tilesX) * B16U_TW;
  const int h = h0 + ty, w = w0 + tx;
  const bool inside = h < H && w < W;
  float wh[6], ww[6];
  b16_taps6(min(h, H - 1), H, rh, wh);
  b16_taps6(min(w, W - 1), W, rw, ww);
  // a bilinear x2 source position receives from 4-5 of the 6 window rows / columns; which ones drifts slowly with the position,
  // so most waves skip a third of the taps in each direction: wave-uniform masks of the taps ANY lane needs
  unsigned ymask = 0, xmask = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    ymask |= (__builtin_amdgcn_ballot_w64(wh[k] != 0.f) != 0ull ? 1u : 0u) << k;
    xmask |= (__builtin_amdgcn_ballot_w64(ww[k] != 0.f) != 0ull ? 1u : 0u) << k;
  }
  const int Ho = 2 * H, Wo = 2 * W, r0 = 2 * h0 - 2, c0w = 2 * w0 - 2;
  // staging slots of this thread: window position (rr, cc), half -> global offset inside an image plane (or -1) and LDS slot
  int l_src[LE], l_dst[LE];
#pragma unroll
  for (int e = 0; e < LE; ++e) {
    const int i = threadIdx.x + 256 * e;
    const int hs = i & 1, pp = i >> 1, rr = pp / B16U_RW, cc = pp % B16U_RW;
    const int ho = r0 + rr, wo = c0w + cc;
    const bool ok = i < SLOTS && (unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo;
    l_src[e] = ok ? (ho * Wo + wo) * 2 + hs : -1;
    l_dst[e] = i < SLOTS ? (cc & 1) * PLANE + (rr * HWD + (cc >> 1)) * 2 + hs : -1;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const long long img = (long long)Ho * Wo * 2;
  u32x4 stg[LE];
  auto fetch = [&](int n) __attribute__((always_inline)) {
    const u32x4* dp = dout + ((long long)n * (CB0 + CB1) + cbo) * img;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < LE; ++e) stg[e] = l_src[e] >= 0 ? dp[l_src[e]] : z;
  };
  if ((int)blockIdx.z < N) fetch(blockIdx.z);
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < LE; ++e)
      if (l_dst[e] >= 0) tile[l_dst[e]] = stg[e];
    __syncthreads();
    if (n + (int)gridDim.z < N) fetch(n + gridDim.z);          // the next image's window is in flight during the gather
    if (inside) {
      const long long o = (((long long)n * CBs + cbs) * H * W + (long long)h * W + w) * 2 + half;
      float yv[8], tot[8];
      b16_unpack8(xs[o], yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) tot[j] = 0.f;
#pragma unroll
      for (int y = 0; y < 6; ++y) {
        if (!((ymask >> y) & 1u)) continue;                  // wave-uniform: no lane of this wave has a weight on window row y
        float row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = 0.f;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          if (!((xmask >> x) & 1u)) continue;
          float t[8];
          // window column 2tx + x: plane x & 1, column tx + (x >> 1)
          b16_unpack8(tile[(x & 1) * PLANE + ((2 * ty + y) * HWD + tx + (x >> 1)) * 2 + half], t);
#pragma unroll
          for (int j = 0; j < 8; ++j) row[j] = fmaf(ww[x], t[j], row[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tot[j] = fmaf(wh[y], row[j], tot[j]);
      }
      float prev[8];
      if (acc0 && first) b16_unpack8(g[o], prev);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!(fmaf(yv[j], sc[j], sh[j]) > 0.f)) tot[j] = 0.f;
        s1[j] += tot[j];
        s2[j] = fmaf(tot[j], (yv[j] - mu[j]) * is[j], s2[j]);
        if (acc0 && first) tot[j] += prev[j];
      }
      g[o] = b16_pack8(tot);
    }
  }
  if (!first && bstats1) b16_block_stats(s1, s2, cbs, C1, bstats1);
}
extern "C" int avsep_b16_relu_up2x_bwd(const void* x0, const void* x1, const float* sc0, const float* sh0, const float* sc1,
                                       const float* sh1, int32_t N, int32_t C0, int32_t C1, int32_t H, int32_t W, const void* dout,
                                       void* g0, void* g1, const float* mean1, const float* invstd1, double* bstats1, int32_t acc0,
                                       avsep_stream_t stream) {
  if (!x0 || !dout || !g0 || N <= 0 || N > 65535 || C0 <= 0 || C0 % 16 || C1 < 0 || C1 % 16 || (C1 > 0 && (!x1 || !g1)) ||
      H <= 0 || W <= 0 || (long long)H * W >= (1LL << 27))
    return AVSEP_ERR_ARG;
  if ((sc0 == nullptr) != (sh0 == nullptr) || (sc1 == nullptr) != (sh1 == nullptr)) return AVSEP_ERR_ARG;
  if (bstats1 && (!mean1 || !invstd1 || C1 == 0)) return AVSEP_ERR_ARG;
  const float rh = H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.f, rw = W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
  const int tx = cdiv(W, B16U_TW), tyc = cdiv(H, B16U_TH), CB = (C0 + C1) / 16;
  int gz = N;
  if ((long long)tx * tyc * CB * gz > 8192) gz = (int)max(1LL, 8192LL / ((long long)tx * tyc * CB));
  hipLaunchKernelGGL(b16_relu_up2x_bwd_kernel, dim3(tx * tyc, CB, gz), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x0,
                     (const u32x4*)x1, sc0, sh0, sc1, sh1, N, C0, C1, H, W, rh, rw, (const u32x4*)dout, (u32x4*)g0, (u32x4*)g1,
                     mean1, invstd1, bstats1, acc0, tx);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- stem tail: MaxPool2d(3,2,1) over act(scale*y + shift), B16 in / out, winning tap (kh*3+kw) as one byte per element ------------
// idx image: [N][C/16][Ho][Wo][16] bytes (same blocking).  First maximum in (kh, kw) scan order wins, like F.max_pool2d.
__global__ __launch_bounds__(256) void b16_maxpool_fwd_kernel(const u32x4* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act, int N, int C, int H, int W,
                                                              int Ho, int Wo, u32x4* __restrict__ z, uint2* __restrict__ idx) {
  const int cb = blockIdx.y, CB = C >> 4, half = threadIdx.x & 1, c0 = cb * 16 + half * 8, So = Ho * Wo * 2;
  float sc[8], sh[8];
  b16_row8(scale, c0, 1.f, sc); b16_row8(shift, c0, 0.f, sh);
  const float slope = act_slope(act);
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const u32x4* yp = y + ((long long)n * CB + cb) * H * W * 2;
    const long long ob = ((long long)n * CB + cb) * So;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < So; s += gridDim.x * 256) {
      const int p = s >> 1, ho = p / Wo, wo = p - ho * Wo;
      float best[8];
      unsigned bi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { best[j] = 0.f; bi[j] = 0xffu; }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = 2 * ho - 1 + kh;
        if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int wx = 2 * wo - 1 + kw;
          if ((unsigned)wx >= (unsigned)W) continue;
          float v[8];
          b16_unpack8(yp[(hh * W + wx) * 2 + half], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = act_by_slope(fmaf(v[j], sc[j], sh[j]), slope);
            if (a > best[j] || bi[j] == 0xffu) { best[j] = a; bi[j] = kh * 3 + kw; }
          }
        }
      }
      z[ob + s] = b16_pack8(best);
      if (idx)
        idx[ob + s] = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
    }
  }
}
extern "C" int avsep_b16_maxpool3x3s2_fwd(const void* y, const float* scale, const float* shift, int32_t act, int32_t N, int32_t C,
                                          int32_t H, int32_t W, void* z, void* idx, avsep_stream_t stream) {
  if (!y || !z || H <= 0 || W <= 0 || !b16_dims_ok(N, C, (long long)H * W)) return AVSEP_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return AVSEP_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(b16_maxpool_fwd_kernel, b16_grid(N, C, (long long)Ho * Wo).g, dim3(256), 0, (hipStream_t)stream, (const u32x4*)y,
                     scale, shift, act, N, C, H, W, Ho, Wo, (u32x4*)z, (uint2*)idx);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// Fused stem-tail backward (see ops.hip): g (+ g2) = dL/d(pooled) B16, idx = winning taps, y = RAW stem conv output B16.
// pass 1 (dy == NULL): bstats += (sum dz, sum dz * xhat) taken over the POOLED grid;  pass 2: dy = p*dz + q*y + r with
// dz[pos] = [scale*y+shift > 0] * sum of g over the windows whose winner is pos.
__global__ __launch_bounds__(256) void b16_maxpool_bwd_stats_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ g2,
                                                                    const uint2* __restrict__ idx, const u32x4* __restrict__ y,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    int N, int C, int H, int W, int Ho, int Wo, double* bstats) {
  const int cb = blockIdx.y, CB = C >> 4, half = threadIdx.x & 1, c0 = cb * 16 + half * 8, So = Ho * Wo * 2;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
  b16_row8(scale, c0, 1.f, sc); b16_row8(shift, c0, 0.f, sh); b16_row8(mean, c0, 0.f, mu); b16_row8(invstd, c0, 1.f, is);
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const unsigned short* yh = reinterpret_cast<const unsigned short*>(y);
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const long long yb = ((long long)n * CB + cb) * H * W, ob = ((long long)n * CB + cb) * So;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < So; s += gridDim.x * 256) {
      const int p = s >> 1, ho = p / Wo, wo = p - ho * Wo;
      float gv[8], g2v[8];
      b16_unpack8(g[ob + s], gv);
      if (g2) b16_unpack8(g2[ob + s], g2v);
      const uint2 t = idx[ob + s];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned tap = ((j < 4 ? t.x : t.y) >> (8 * (j & 3))) & 0xffu;
        const int hh = 2 * ho - 1 + (int)(tap / 3), wx = 2 * wo - 1 + (int)(tap % 3);
        const float yv = __builtin_bit_cast(float, (unsigned)yh[((yb + (long long)hh * W + wx) * 16) + half * 8 + j] << 16);
        if (fmaf(yv, sc[j], sh[j]) > 0.f) {
          const float gg = g2 ? gv[j] + g2v[j] : gv[j];
          s1[j] += gg;
          s2[j] = fmaf(gg, (yv - mu[j]) * is[j], s2[j]);
        }
      }
    }
  }
  b16_block_stats(s1, s2, cb, C, bstats);
}
// a thread = (input position, half): the <= 4 windows that contain the position
__global__ __launch_bounds__(256) void b16_maxpool_bwd_apply_kernel(const u32x4* __restrict__ g, const u32x4* __restrict__ g2,
                                                                    const uint2* __restrict__ idx, const u32x4* __restrict__ y,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ pqr, int N, int C, int H, int W, int Ho,
                                                                    int Wo, u32x4* __restrict__ dy, float* __restrict__ dy32) {
  const int cb = blockIdx.y, CB = C >> 4, half = threadIdx.x & 1, c0 = cb * 16 + half * 8, S = H * W * 2;
  float sc[8], sh[8], cp[8], cq[8], cr[8];
  b16_row8(scale, c0, 1.f, sc); b16_row8(shift, c0, 0.f, sh);
  b16_row8(pqr, c0, 0.f, cp); b16_row8(pqr + C, c0, 0.f, cq); b16_row8(pqr + 2 * C, c0, 0.f, cr);
  for (int n = blockIdx.z; n < N; n += gridDim.z) {
    const long long yb = ((long long)n * CB + cb) * S, ob = ((long long)n * CB + cb) * Ho * Wo * 2;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < S; s += gridDim.x * 256) {
      const int p = s >> 1, h = p / W, w = p - h * W;
      float yv[8], acc[8];
      b16_unpack8(y[yb + s], yv);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      // window ho covers input rows 2ho-1 .. 2ho+1  =>  h/2 <= ho <= (h+1)/2
      for (int ho = h >> 1; ho <= ((h + 1) >> 1) && ho < Ho; ++ho)
        for (int wo = w >> 1; wo <= ((w + 1) >> 1) && wo < Wo; ++wo) {
          const unsigned self = (unsigned)((h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1)));   // this position's tap inside window (ho, wo)
          const long long o = ob + ((long long)ho * Wo + wo) * 2 + half;
          const uint2 t = idx[o];
          float gv[8], g2v[8];
          b16_unpack8(g[o], gv);
          if (g2) b16_unpack8(g2[o], g2v);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned tap = ((j < 4 ? t.x : t.y) >> (8 * (j & 3))) & 0xffu;
            if (tap == self) acc[j] += g2 ? gv[j] + g2v[j] : gv[j];
          }
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dz = fmaf(yv[j], sc[j], sh[j]) > 0.f ? acc[j] : 0.f;
        acc[j] = fmaf(cp[j], dz, fmaf(cq[j], yv[j], cr[j]));
      }
      if (dy32) {                                      // fp32 NCHW: 32 lanes of equal half write 128 contiguous bytes per channel
#pragma unroll
        for (int j = 0; j < 8; ++j) dy32[((long long)n * C + c0 + j) * H * W + p] = acc[j];
      } else {
        dy[yb + s] = b16_pack8(acc);
      }
    }
  }
}
extern "C" int avsep_b16_maxpool_bn_relu_bwd(const void* g, const void* g2, const void* idx, const void* y, const float* scale,
                                             const float* shift, const float* mean, const float* invstd, const float* pqr,
                                             int32_t N, int32_t C, int32_t H, int32_t W, double* bstats, void* dy, int32_t dy_f32,
                                             avsep_stream_t stream) {
  if (!g || !idx || !y || !scale || !shift || H <= 0 || W <= 0 || !b16_dims_ok(N, C, (long long)H * W)) return AVSEP_ERR_ARG;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (!dy) {                        // pass 1
    if (!bstats || !mean || !invstd) return AVSEP_ERR_ARG;
    hipLaunchKernelGGL(b16_maxpool_bwd_stats_kernel, b16_grid(N, C, (long long)Ho * Wo).g, dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)g, (const u32x4*)g2, (const uint2*)idx, (const u32x4*)y, scale, shift, mean, invstd, N, C, H, W,
                       Ho, Wo, bstats);
  } else {                          // pass 2
    if (!pqr) return AVSEP_ERR_ARG;
    hipLaunchKernelGGL(b16_maxpool_bwd_apply_kernel, b16_grid(N, C, (long long)H * W).g, dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)g, (const u32x4*)g2, (const uint2*)idx, (const u32x4*)y, scale, shift, pqr, N, C, H, W, Ho, Wo,
                       dy_f32 ? nullptr : (u32x4*)dy, dy_f32 ? (float*)dy : nullptr);
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- space-to-depth of the frames as a ONE-block B16 image (see ops.hip space_to_depth2_kernel) -----------------------------------
//   xs[n][0][i + 2][j + 2][(dy*2+dx)*C + c] = x[n][c][2i + dy][2j + dx],  channels 4C .. 15 and the border are zero
__global__ __launch_bounds__(256) void b16_space_to_depth2_kernel(const float* __restrict__ x, int C, int H, int W, long long total,
                                                                  u32x4* __restrict__ xs) {
  const int Hs = H / 2 + 3, Ws = W / 2 + 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int half = (int)(i & 1);
    const long long pp = i >> 1;
    const int j = (int)(pp % Ws), r = (int)((pp / Ws) % Hs);
    const long long n = pp / ((long long)Ws * Hs);
    float v[8];
    const int ii = r - 2, jj = j - 2;
    const bool in = ii >= 0 && ii < H / 2 && jj >= 0 && jj < W / 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int q = half * 8 + k;
      float t = 0.f;
      if (in && q < 4 * C) {
        const int c = q % C, dy = (q / C) >> 1, dx = (q / C) & 1;
        t = x[((n * C + c) * H + 2 * ii + dy) * W + 2 * jj + dx];
      }
      v[k] = t;
    }
    xs[i] = b16_pack8(v);
  }
}
extern "C" int avsep_b16_space_to_depth2(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, void* xs, avsep_stream_t stream) {
  if (!x || !xs || N <= 0 || C <= 0 || 4 * C > 16 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return AVSEP_ERR_ARG;
  const long long total = (long long)N * (H / 2 + 3) * (W / 2 + 3) * 2;
  hipLaunchKernelGGL(b16_space_to_depth2_kernel, dim3((int)min((total + 255) / 256, (long long)262144)), dim3(256), 0,
                     (hipStream_t)stream, x, C, H, W, total, (u32x4*)xs);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- grid image of a batch of small maps -----------------------------------------------------------------------------------------
// The deep U-Net levels run on 2x2 ... 8x8 maps: a 256-pixel chunk of wgradb_kernel (a tile of ONE image) would be 6-25 % full.
// Here the N images of the batch are laid side by side in ONE image [1][C/16][HG][WG][16], image n at (n / GX * PY, n % GX * PX),
// everything else zero, with the conv's folded affine + activation applied on the way (so that the separators are true zeros).
// With the right pitch the separator rows / columns are the zero padding of every image at once, and wherever one
// operand of the weight gradient is zero the product vanishes: dW over the grid images of X and dY IS the batch's dW
// (3x3 / pad 1: pitch H + 1 for both; 4x4 / stride 2 / pad 1: pitch H + 2 for X, H/2 + 1 for dY).
__global__ __launch_bounds__(256) void b16_grid_pack_kernel(const void* __restrict__ x, int xf32, int N, int C, int H, int W, int GX,
                                                            int PY, int PX, int HG, int WG, const float* __restrict__ sc,
                                                            const float* __restrict__ sh, int act, u32x4* __restrict__ out) {
  const int cb = blockIdx.y, CB = C >> 4;
  const float slope = act_slope(act);
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= HG * WG * 2) return;
  const int half = slot & 1, pos = slot >> 1, Y = pos / WG, X = pos % WG;
  const int gy = Y / PY, r = Y % PY, gx = X / PX, c = X % PX, n = gy * GX + gx, c0 = cb * 16 + half * 8;
  u32x4 q = {0u, 0u, 0u, 0u};
  if (r < H && c < W && gx < GX && n < N) {
    float v[8];
    if (xf32) {
      const float* xp = reinterpret_cast<const float*>(x) + (((long long)n * C + c0) * H + r) * W + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)(__bf16)xp[(long long)j * H * W];   // as stored in a B16 image: rounded before the affine
    } else {
      b16_unpack8(reinterpret_cast<const u32x4*>(x)[((((long long)n * CB + cb) * H + r) * W + c) * 2 + half], v);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (sc) v[j] = fmaf(v[j], sc[c0 + j], sh[c0 + j]);
      v[j] = act_by_slope(v[j], slope);
    }
    q = b16_pack8(v);
  }
  out[((long long)cb * HG * WG + pos) * 2 + half] = q;
}
extern "C" int avsep_b16_grid_pack(const void* x, int32_t xfmt, int32_t N, int32_t C, int32_t H, int32_t W, int32_t GX, int32_t PY,
                                   int32_t PX, int32_t HG, int32_t WG, const float* scale, const float* shift, int32_t act, void* out,
                                   avsep_stream_t stream) {
  if (!x || !out || (xfmt != AVSEP_FMT_F32 && xfmt != AVSEP_FMT_B16) || N <= 0 || C <= 0 || C % 16 || H <= 0 || W <= 0 || GX <= 0 ||
      PY < H || PX < W || HG <= 0 || WG <= 0 || HG % PY || WG != GX * PX || (long long)(HG / PY) * GX < N ||
      (long long)HG * WG >= (1LL << 27) || (scale == nullptr) != (shift == nullptr) || act < 0 || act > 2)
    return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(b16_grid_pack_kernel, dim3(cdiv(HG * WG * 2, 256), C / 16), dim3(256), 0, (hipStream_t)stream, x,
                     (int)(xfmt == AVSEP_FMT_F32), N, C, H, W, GX, PY, PX, HG, WG, scale, shift, act, (u32x4*)out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// the inverse for a forward result: the real positions of an fp32 grid image [1][C][HG][WG] -> the batch's output (B16 image or
// fp32 NCHW) with the BatchNorm sums (sum y, sum y^2) of the fp32 values, as the conv epilogues take them
__global__ __launch_bounds__(256) void grid_unpack_kernel(const float* __restrict__ grid, int N, int C, int H, int W, int GX, int PY,
                                                          int PX, int HG, int WG, void* __restrict__ out, int of32,
                                                          double* __restrict__ stats) {
  const int cb = blockIdx.y, CB = C >> 4;
  const int slot = blockIdx.x * 256 + threadIdx.x, half = threadIdx.x & 1, c0 = cb * 16 + half * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (slot < N * H * W * 2) {
    const int pos = slot >> 1, n = pos / (H * W), r = (pos % (H * W)) / W, c = pos % W;
    const float* gp = grid + ((long long)c0 * HG + (n / GX) * PY + r) * WG + (n % GX) * PX + c;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = gp[(long long)j * HG * WG];
      s1[j] = v[j];
      s2[j] = v[j] * v[j];
    }
    if (of32) {
      float* op = reinterpret_cast<float*>(out) + (((long long)n * C + c0) * H + r) * W + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) op[(long long)j * H * W] = v[j];
    } else {
      reinterpret_cast<u32x4*>(out)[((((long long)n * CB + cb) * H + r) * W + c) * 2 + half] = b16_pack8(v);
    }
  }
  if (stats) b16_block_stats(s1, s2, cb, C, stats);
}
extern "C" int avsep_grid_unpack(const float* grid, int32_t N, int32_t C, int32_t H, int32_t W, int32_t GX, int32_t PY, int32_t PX,
                                 int32_t HG, int32_t WG, void* out, int32_t ofmt, double* stats, avsep_stream_t stream) {
  if (!grid || !out || (ofmt != AVSEP_FMT_F32 && ofmt != AVSEP_FMT_B16) || N <= 0 || C <= 0 || C % 16 || H <= 0 || W <= 0 || GX <= 0 ||
      PY < H || PX < W || HG <= 0 || WG <= 0 || HG % PY || WG != GX * PX || (long long)(HG / PY) * GX < N ||
      (long long)HG * WG >= (1LL << 27) || (long long)N * H * W >= (1LL << 27))
    return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(grid_unpack_kernel, dim3(cdiv(N * H * W * 2, 256), C / 16), dim3(256), 0, (hipStream_t)stream, grid, N, C, H, W, GX,
                     PY, PX, HG, WG, out, (int)(ofmt == AVSEP_FMT_F32), stats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
