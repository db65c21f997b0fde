// Direct (non-MFMA) kernels for the convolutions whose GEMM M dimension is too small for a 32-row
// MFMA tile: the U-Net's last conv (128 -> num_mix=2 channels, 3x3, audio_net.py:75-76) forward and
// weight gradient.  Both are HBM/LDS-bound: the input is staged once per channel chunk as an LDS
// halo patch (coalesced rows), the few output channels live in registers, and the weights are read
// through the scalar unit (wave-uniform addresses).  gfx950, wave64.
#include "common.h"

#define SC_MAXCO 4

// ---------------------------------------------------------------------------
// forward: y[n,co,h,w] = bias[co] + sum_{ci,kh,kw} w[co,ci,kh,kw] * x[n,ci,h+kh-1,w+kw-1]
// block = 8 rows x 128 cols of one image; thread = 1 row x 4 consecutive cols, all Cout channels
// ---------------------------------------------------------------------------
constexpr int F_TH = 8, F_TW = 128, F_CH = 8, F_PH = F_TH + 2, F_PW = F_TW + 8;  // col j <-> w0 + j - 4: interior 16-B aligned

template <int COUT>
__global__ __launch_bounds__(256) void smallco_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          int wp_ld, const float* __restrict__ bias,
                                                          float* __restrict__ y, int Cin, int H, int W) {
  __shared__ __attribute__((aligned(16))) float patch[F_CH][F_PH][F_PW];
  const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
  const int n = blockIdx.z, h0 = blockIdx.y * F_TH, w0 = blockIdx.x * F_TW;
  float acc[COUT][4];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;
  const long long HW = (long long)H * W;
  for (int c0 = 0; c0 < Cin; c0 += F_CH) {
    __syncthreads();
    // halo patch rows h0-1 .. h0+TH; interior columns as aligned 16-byte loads (dword loads cap a streaming
    // kernel at ~1.3 TB/s on this chip), the two halo columns as scalars
    for (int i = tid; i < F_CH * F_PH * (F_TW / 4); i += 256) {
      int q = i % (F_TW / 4), row = i / (F_TW / 4), r = row % F_PH, ch = row / F_PH;
      int gh = h0 - 1 + r, gw = w0 + 4 * q, c = c0 + ch;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < Cin && (unsigned)gh < (unsigned)H && gw < W)
        v = *reinterpret_cast<const f32x4*>(x + ((long long)n * Cin + c) * HW + (long long)gh * W + gw);
      *reinterpret_cast<f32x4*>(&patch[ch][r][4 + 4 * q]) = v;
    }
    for (int i = tid; i < F_CH * F_PH * 2; i += 256) {
      int side = i & 1, row = i >> 1, r = row % F_PH, ch = row / F_PH;
      int gh = h0 - 1 + r, gw = side ? w0 + F_TW : w0 - 1, c = c0 + ch;
      float v = 0.f;
      if (c < Cin && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
        v = x[((long long)n * Cin + c) * HW + (long long)gh * W + gw];
      patch[ch][r][side ? F_TW + 4 : 3] = v;
    }
    __syncthreads();
    const int nch = min(F_CH, Cin - c0);
    for (int ch = 0; ch < nch; ++ch) {
      float xv[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(&patch[ch][ty + r][4 * tx + 4]);
        xv[r][0] = patch[ch][ty + r][4 * tx + 3];
        xv[r][1] = m.x; xv[r][2] = m.y; xv[r][3] = m.z; xv[r][4] = m.w;
        xv[r][5] = patch[ch][ty + r][4 * tx + 8];
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        // packed forward operand [k=(ci,kh,kw)][co]; wave-uniform address -> scalar loads
        const float* wc = wp + (long long)(c0 + ch) * 9 * wp_ld + co;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float wv = wc[(kh * 3 + kw) * wp_ld];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv, xv[kh][p + kw], acc[co][p]);
          }
      }
    }
  }
  const int h = h0 + ty, wq = w0 + 4 * tx;
  if (h < H) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float b = bias ? bias[co] : 0.f;
      float* o = y + ((long long)n * COUT + co) * HW + (long long)h * W + wq;
      if (wq + 3 < W && (W & 3) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[co][0] + b, acc[co][1] + b, acc[co][2] + b, acc[co][3] + b);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (wq + p < W) o[p] = acc[co][p] + b;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dw[co,ci,kh,kw] = sum_{n,h,w} dy[n,co,h,w] * x[n,ci,h+kh-1,w+kw-1]  (+ dbias)
// block = (16-channel chunk, image n).  Worker thread (cg, cl, kh) owns dw[:, c0+cl, kh, 0..2] for the
// column strip cg of the image and sweeps it in 4-row tiles staged in LDS (float4 rows, zero halo
// columns because a tile spans the full width).  Per 4 pixels: 1 b128 + 2 b32 reads of x and COUT b128
// reads of dy feed 12*COUT FMAs.  One partial slab per (image, strip), summed by a second kernel
// (deterministic).  The 4th wave accumulates sum(dy) for dbias.
// ---------------------------------------------------------------------------
constexpr int G_CC = 16, G_TH = 4, G_WMAX = 256, G_STRIPS = 4;

template <int COUT>
__global__ __launch_bounds__(256) void smallco_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ part, float* __restrict__ bpart,
                                                            int Cin, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int PW = W + 8;                                 // col j <-> w = j-4: interior 16-B aligned, halo at j=3 and j=W+4
  float* sx = smem;                                     // [G_CC][G_TH+2][PW]
  float* sdy = smem + G_CC * (G_TH + 2) * PW;           // [COUT][G_TH][W]
  const int tid = threadIdx.x, c0 = blockIdx.x * G_CC, n = blockIdx.y;
  const bool worker = tid < 192;
  const int cg = tid / 48, rem = tid % 48, cl = rem / 3, kh = rem % 3;
  const int strip = W / G_STRIPS;                       // W % 16 == 0 is checked on the host
  float acc[COUT][3];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[co][k] = 0.f;
  float bacc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) bacc[co] = 0.f;
  const long long HW = (long long)H * W;
  const int W4 = W >> 2;
  for (int i = tid; i < G_CC * (G_TH + 2) * PW; i += 256) sx[i] = 0.f;   // halo columns stay zero for the whole sweep
  for (int h0 = 0; h0 < H; h0 += G_TH) {
    __syncthreads();
    for (int i = tid; i < G_CC * (G_TH + 2) * W4; i += 256) {
      const int q = i % W4, row = i / W4;               // row = ch*(G_TH+2) + r
      const int ch = row / (G_TH + 2), r = row % (G_TH + 2);
      const int gh = h0 - 1 + r, c = c0 + ch;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < Cin && (unsigned)gh < (unsigned)H)
        v = *reinterpret_cast<const float4*>(x + ((long long)n * Cin + c) * HW + (long long)gh * W + 4 * q);
      *reinterpret_cast<float4*>(&sx[row * PW + 4 + 4 * q]) = v;
    }
    for (int i = tid; i < COUT * G_TH * W4; i += 256) {
      const int q = i % W4, row = i / W4;               // row = co*G_TH + r
      const int co = row / G_TH, gh = h0 + row % G_TH;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh < H) v = *reinterpret_cast<const float4*>(dy + ((long long)n * COUT + co) * HW + (long long)gh * W + 4 * q);
      *reinterpret_cast<float4*>(&sdy[row * W + 4 * q]) = v;
    }
    __syncthreads();
    if (worker) {
#pragma unroll
      for (int r = 0; r < G_TH; ++r) {
        const float* xr = sx + (cl * (G_TH + 2) + r + kh) * PW + 4;   // xr[w] = x(.., w), xr[-1] / xr[W] = zero halo
#pragma unroll 2
        for (int wq = cg * strip; wq < (cg + 1) * strip; wq += 4) {
          float xv[6];
          const float4 m = *reinterpret_cast<const float4*>(xr + wq);
          xv[0] = xr[wq - 1]; xv[1] = m.x; xv[2] = m.y; xv[3] = m.z; xv[4] = m.w; xv[5] = xr[wq + 4];
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float4 d = *reinterpret_cast<const float4*>(&sdy[(co * G_TH + r) * W + wq]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              acc[co][k] = fmaf(d.x, xv[k], acc[co][k]);
              acc[co][k] = fmaf(d.y, xv[k + 1], acc[co][k]);
              acc[co][k] = fmaf(d.z, xv[k + 2], acc[co][k]);
              acc[co][k] = fmaf(d.w, xv[k + 3], acc[co][k]);
            }
          }
        }
      }
    } else if (bpart && blockIdx.x == 0) {
      const int lane = tid - 192;
#pragma unroll
      for (int co = 0; co < COUT; ++co)
        for (int i = lane; i < G_TH * W; i += 64) bacc[co] += sdy[co * G_TH * W + i];
    }
  }
  if (worker && c0 + cl < Cin) {
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        part[((((long long)n * G_STRIPS + cg) * COUT + co) * Cin + c0 + cl) * 9 + kh * 3 + k] = acc[co][k];
  }
  if (!worker && bpart && blockIdx.x == 0) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float s = wave_sum(bacc[co]);
      if (tid == 192) bpart[n * COUT + co] = s;
    }
  }
}

__global__ void smallco_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int S) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < S; ++z) s += part[(long long)z * n + i];
  out[i] = s;
}

// ---------------------------------------------------------------------------
// data gradient for a conv with <= 4 INPUT channels (the U-Net's first conv, 1 -> 64, 4x4 s2: its input
// gradient is only needed for the bn0 parameter gradients).  M = Cin is too small for an MFMA tile:
// thread = one input pixel, all Cin channels; dy rows are read coalesced, weights through the scalar unit.
//   dx[n,ci,hi,wi] = sum_{co,kh,kw} w[co,ci,kh,kw] * dy[n,co,(hi+p-kh)/s,(wi+p-kw)/s]   (exact divisions only)
// ---------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void smallci_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int Cout, int H, int W, int Ho,
                                                            int Wo, int KH, int KW, int stride, int pad) {
  const int n = blockIdx.z, hi = blockIdx.y, wi = blockIdx.x * 256 + threadIdx.x;
  if (wi >= W) return;
  float acc[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
  const long long HoWo = (long long)Ho * Wo;
  for (int kh = 0; kh < KH; ++kh) {
    const int th = hi + pad - kh;                       // block-uniform
    if (th < 0 || th % stride) continue;
    const int ho = th / stride;
    if (ho >= Ho) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int tw = wi + pad - kw;
      const bool ok = tw >= 0 && tw % stride == 0 && tw / stride < Wo;
      const int wo = ok ? tw / stride : 0;
      const float* dp = dy + (long long)n * Cout * HoWo + (long long)ho * Wo + wo;
      for (int co = 0; co < Cout; ++co) {
        const float d = ok ? dp[co * HoWo] : 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[c] = fmaf(w[((co * CIN + c) * KH + kh) * KW + kw], d, acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CIN; ++c) dx[(((long long)n * CIN + c) * H + hi) * W + wi] = acc[c];
}

// The U-Net's first conv (1 -> 64, 4x4 s2 p1, audio_net.py:57) at full resolution: thread = a 2-row x 8-column block
// of dx (input rows 2a, 2a+1; columns 8q..8q+7) that needs, per output channel, only the 3 x 6 patch of dy around
// (a, 4q): three 16-byte loads + six halo words feed 64 FMAs (the generic kernel above issues one 4-byte load per
// FMA and re-fetched dy 16x from HBM: 2.1 GB for a 134 MB tensor, measured with FETCH_SIZE).
//   rows:  hi = 2a   <- (kh = 1, dy row a), (kh = 3, row a-1);   hi = 2a+1 <- (kh = 0, row a+1), (kh = 2, row a)
//   same for columns.  Weights through the scalar unit.
template <int CIN>
__global__ __launch_bounds__(256) void k4s2_smallci_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                                 float* __restrict__ dx, int Cout, int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int q = blockIdx.x * 32 + (threadIdx.x & 31), a = blockIdx.y * 8 + (threadIdx.x >> 5), n = blockIdx.z;
  if (4 * q >= Wo || a >= Ho) return;
  float acc[CIN][2][8];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i >> 3][i & 7] = 0.f;
  const long long HoWo = (long long)Ho * Wo;
  const float* base = dy + (long long)n * Cout * HoWo + 4 * q;
  const bool left = q > 0, right = 4 * q + 4 < Wo;
  for (int co = 0; co < Cout; ++co) {
    float d[3][6];                                        // dy rows a-1..a+1, columns 4q-1..4q+4
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ho = a - 1 + r;
      if (ho >= 0 && ho < Ho) {
        const float* row = base + co * HoWo + (long long)ho * Wo;
        const f32x4 m = *reinterpret_cast<const f32x4*>(row);
        d[r][0] = left ? row[-1] : 0.f;
        d[r][1] = m.x; d[r][2] = m.y; d[r][3] = m.z; d[r][4] = m.w;
        d[r][5] = right ? row[4] : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < 6; ++e) d[r][e] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float* wc = w + ((long long)co * CIN + c) * 16;   // wave-uniform -> scalar loads
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int th = 0; th < 2; ++th) {
          const int kh = ph ? 2 - 2 * th : 3 - 2 * th;        // th = 0 is the upper dy row of the class
          const int r = ph ? 1 + th : th;                     // index into d[]: rows (a-1, a) | (a, a+1)
#pragma unroll
          for (int j = 0; j < 4; ++j) {                       // b = 4q + j
            // even column 2b <- (kw = 1, dy col b), (kw = 3, col b-1);  odd column 2b+1 <- (kw = 0, b+1), (kw = 2, b)
            acc[c][ph][2 * j] = fmaf(wc[kh * 4 + 3], d[r][j], fmaf(wc[kh * 4 + 1], d[r][j + 1], acc[c][ph][2 * j]));
            acc[c][ph][2 * j + 1] = fmaf(wc[kh * 4 + 2], d[r][j + 1], fmaf(wc[kh * 4 + 0], d[r][j + 2], acc[c][ph][2 * j + 1]));
          }
        }
    }
  }
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      float* o = dx + (((long long)n * CIN + c) * H + 2 * a + ph) * W + 8 * q;
      *reinterpret_cast<f32x4*>(o) = f32x4{acc[c][ph][0], acc[c][ph][1], acc[c][ph][2], acc[c][ph][3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[c][ph][4], acc[c][ph][5], acc[c][ph][6], acc[c][ph][7]};
    }
}

// ---------------------------------------------------------------------------
// host dispatch (called from conv.hip)
// ---------------------------------------------------------------------------
bool smallci_applicable(const avsep_conv_desc* d) {
  return d->Cin <= 4 && d->dil == 1 && !d->up2x && d->H <= 65535 && d->N <= 65535;
}
int smallci_dgrad(const avsep_conv_desc* d, const float* w_oihw, const float* dy, float* dx, hipStream_t st) {
  if (d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && (d->W & 7) == 0 && (d->H & 1) == 0) {
    dim3 g2(cdiv(d->W / 8, 32), cdiv(d->H / 2, 8), d->N);
    switch (d->Cin) {
      case 1: hipLaunchKernelGGL(k4s2_smallci_dgrad_kernel<1>, g2, dim3(256), 0, st, w_oihw, dy, dx, d->Cout, d->H, d->W); break;
      case 2: hipLaunchKernelGGL(k4s2_smallci_dgrad_kernel<2>, g2, dim3(256), 0, st, w_oihw, dy, dx, d->Cout, d->H, d->W); break;
      case 3: hipLaunchKernelGGL(k4s2_smallci_dgrad_kernel<3>, g2, dim3(256), 0, st, w_oihw, dy, dx, d->Cout, d->H, d->W); break;
      default: hipLaunchKernelGGL(k4s2_smallci_dgrad_kernel<4>, g2, dim3(256), 0, st, w_oihw, dy, dx, d->Cout, d->H, d->W); break;
    }
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  dim3 grid(cdiv(d->W, 256), d->H, d->N);
#define LAUNCH_CI(CI)                                                                                              \
  hipLaunchKernelGGL(smallci_dgrad_kernel<CI>, grid, dim3(256), 0, st, w_oihw, dy, dx, d->Cout, d->H, d->W, d->Ho, \
                     d->Wo, d->KH, d->KW, d->stride, d->pad)
  switch (d->Cin) {
    case 1: LAUNCH_CI(1); break;
    case 2: LAUNCH_CI(2); break;
    case 3: LAUNCH_CI(3); break;
    default: LAUNCH_CI(4); break;
  }
#undef LAUNCH_CI
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

bool smallco_applicable(const avsep_conv_desc* d) {
  return d->Cout <= SC_MAXCO && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && !d->up2x &&
         d->C0 == d->Cin && !d->scale0 && d->act0 == AVSEP_ACT_NONE && d->W <= G_WMAX && (d->W & 15) == 0 && d->N <= 65535;
}

int smallco_fwd(const avsep_conv_desc* d, const float* wp, int wp_ld, const float* bias, float* y, hipStream_t st) {
  dim3 grid(cdiv(d->W, F_TW), cdiv(d->H, F_TH), d->N);
  switch (d->Cout) {
    case 1: hipLaunchKernelGGL(smallco_fwd_kernel<1>, grid, dim3(256), 0, st, d->x0, wp, wp_ld, bias, y, d->Cin, d->H, d->W); break;
    case 2: hipLaunchKernelGGL(smallco_fwd_kernel<2>, grid, dim3(256), 0, st, d->x0, wp, wp_ld, bias, y, d->Cin, d->H, d->W); break;
    case 3: hipLaunchKernelGGL(smallco_fwd_kernel<3>, grid, dim3(256), 0, st, d->x0, wp, wp_ld, bias, y, d->Cin, d->H, d->W); break;
    default: hipLaunchKernelGGL(smallco_fwd_kernel<4>, grid, dim3(256), 0, st, d->x0, wp, wp_ld, bias, y, d->Cin, d->H, d->W); break;
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

size_t smallco_wgrad_workspace_floats(const avsep_conv_desc* d) {
  return (size_t)d->N * G_STRIPS * d->Cout * d->Cin * 9 + (size_t)d->N * d->Cout;
}

int smallco_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias, float* ws, hipStream_t st) {
  float* part = ws;
  float* bpart = ws + (size_t)d->N * G_STRIPS * d->Cout * d->Cin * 9;
  dim3 grid(cdiv(d->Cin, G_CC), d->N);
  size_t smem = ((size_t)G_CC * (G_TH + 2) * (d->W + 8) + (size_t)d->Cout * G_TH * d->W) * sizeof(float);
#define LAUNCH_WG(CO)                                                                                                   \
  do {                                                                                                                  \
    if (smem > 64 * 1024)                                                                                               \
      (void)hipFuncSetAttribute((const void*)smallco_wgrad_kernel<CO>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)smem);                                                                             \
    hipLaunchKernelGGL(smallco_wgrad_kernel<CO>, grid, dim3(256), smem, st, d->x0, dy, part, dbias ? bpart : nullptr,   \
                       d->Cin, d->H, d->W);                                                                             \
  } while (0)
  switch (d->Cout) {
    case 1: LAUNCH_WG(1); break;
    case 2: LAUNCH_WG(2); break;
    case 3: LAUNCH_WG(3); break;
    default: LAUNCH_WG(4); break;
  }
#undef LAUNCH_WG
  AVSEP_LAUNCH_CHECK();
  int nw = d->Cout * d->Cin * 9;
  hipLaunchKernelGGL(smallco_reduce_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, st, part, dw, nw, d->N * G_STRIPS);
  AVSEP_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(smallco_reduce_kernel, dim3(1), dim3(64), 0, st, bpart, dbias, d->Cout, d->N);
    AVSEP_LAUNCH_CHECK();
  }
  return AVSEP_OK;
}
