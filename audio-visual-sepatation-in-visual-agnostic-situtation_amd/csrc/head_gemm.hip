// Decoder head (audio_net.py:72-76: ReLU -> bilinear x2 (align_corners) -> conv 3x3 p1 -> num_mix logits on the concat
// of the skip tensor and the inner block's BatchNorm'd output) with the channel contraction taken at LOW resolution.
//
// The x2 interpolation acts per channel and the channel mix acts per pixel: they commute.  With R = relu(affine(cat))
// ([C, Hl, Wl]), up() the interpolation and k = (co, kh, kw) one of the KT = 9*Cout (output channel, tap) pairs:
//   forward   y[co][p]  = bias + sum_tap [p+tap inside] * up(T[k])[p + tap - 1],   T[k][P] = sum_c w[co][c][tap] * R[c][P]
//   adjoint   S[k][P]   = up^T(shift_tap(dy[co]))[P]                 (zero padding of the hi-res map included)
//   dgrad     g[c][P]   = relu'(.) * sum_k w[k][c] * S[k][P]
//   wgrad     dw[k][c]  = sum_{n,P} S[k][P] * R[c][P]
// so the 128-channel tensor is touched once per pass at 1/4 of the hi-res pixel count and only 9*Cout planes ever exist at
// 256x256.  Against the register-rebuild form (head.hip: every wave re-interpolates the hi-res rows it needs; 2304 vector
// FMAs per hi-res pixel) the contraction is 2 x 128 x KT flops per LOW-res pixel: a 32-row f32 MFMA tile for the two GEMMs
// (T and dw), plain vector FMAs with wave-uniform weights for g (its K = KT is 18..36).  All four kernels are bound by the
// HBM pass over the low-res sources.
#include <stdlib.h>
#include "common.h"

struct HeadArgs {
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  int N, C0, C1, Hl, Wl;   // low-res geometry; hi-res is 2*Hl x 2*Wl
  float rh, rw;            // (L-1)/(2L-1) per axis
};

// hi-res index o on an axis of low-res length L: the reference's (i0, i1, lambda) in float arithmetic
struct HLerp { int i0, i1; float l; };
__device__ __forceinline__ HLerp head_lerp(int o, int L, float r) {
  HLerp q;
  const float f = r * (float)o;
  q.i0 = (int)f;
  q.l = f - (float)q.i0;
  q.i1 = q.i0 + (q.i0 < L - 1);
  return q;
}
// weight of low-res index i in hi-res index o (0 when o is outside the map: the conv's zero padding)
__device__ __forceinline__ float head_coef(int o, int i, int L, float r) {
  if (o < 0 || o >= 2 * L) return 0.f;
  const HLerp q = head_lerp(o, L, r);
  return (q.i0 == i ? 1.f - q.l : 0.f) + (q.i1 == i ? q.l : 0.f);
}

__device__ __forceinline__ float lane_value(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// ---------------------------------------------------------------------------------------------------------------
// forward, step 1: T[n][k][P] = sum_c w[co][c][tap] * relu(affine(x[n][c][P]))       (k = co*9 + tap)
// GEMM per image: M = k (one or two 32-row tiles), N = pixels (a wave owns 64: two 32-column tiles), K = channels.
// v_mfma_f32_32x32x2_f32 takes ONE value per lane: lane (i, h) = (l % 32, l / 32) supplies A[m = i][k = h] and
// B[k = h][n = i], so a B operand is a single dword load of channel 2s+h at pixel P0+i (two 128-byte runs per wave) and
// the A operands (the weights, constant for the kernel) sit in LDS in lane order.
// ---------------------------------------------------------------------------------------------------------------
// GENERIC: source channel counts that are not multiples of 8 or a total that is not a multiple of 16 (tests; a pair of channels may straddle the sources, the last step may
// be half empty): per-lane plane pointers instead of one wave-uniform base.
template <int MT, bool GENERIC>
__global__ __launch_bounds__(256) void head_fwd_gemm_kernel(HeadArgs a, const float* __restrict__ wp, int wp_ld, int KT,
                                                            float* __restrict__ T) {
  extern __shared__ float sm[];            // [MT][KS][64] weights in lane order | [2*KS] scale | [2*KS] shift; KS = channel pairs (x8)
  const int C = a.C0 + a.C1, KS = (C + 15) / 16 * 8;
  float* sw = sm;
  float* ssc = sm + MT * KS * 64;
  float* ssh = ssc + 2 * KS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
#pragma unroll 4
  for (int e = tid; e < KS * 64; e += 256) {
    const int s = e >> 6, l = e & 63, cr = 2 * s + (l >> 5), c = min(cr, C - 1);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int k = mt * 32 + (l & 31);
      sw[(mt * KS + s) * 64 + l] = (k < KT && cr < C) ? wp[(long long)(c * 9 + k % 9) * wp_ld + k / 9] : 0.f;
    }
  }
  for (int cr = tid; cr < 2 * KS; cr += 256) {
    const int c = min(cr, C - 1);
    const bool first = c < a.C0;
    const float* sc = first ? a.sc0 : a.sc1;
    const float* sh = first ? a.sh0 : a.sh1;
    const int cs = first ? c : c - a.C0;
    ssc[cr] = sc ? sc[cs] : 1.f;
    ssh[cr] = sc ? sh[cs] : 0.f;
  }
  __syncthreads();
  const long long HWl = (long long)a.Hl * a.Wl;
  const int KS0 = a.C0 / 2;
  const long long tpi = (HWl + 63) / 64, total = tpi * a.N;        // 64-pixel tiles per image / in all
  // persistent: the grid is one round of resident blocks, every wave strides over the (image, tile) list
  for (long long t = (long long)blockIdx.x * 4 + wave; t < total; t += (long long)gridDim.x * 4) {
    const int n = (int)(t / tpi);
    const long long P0 = (t - (long long)n * tpi) * 64;
    const float* x0n = a.x0 + (long long)n * a.C0 * HWl;
    const float* x1n = a.x1 ? a.x1 + (long long)n * a.C1 * HWl : a.x0;
    const float* b0 = x0n + lh * HWl;
    const float* b1 = x1n + lh * HWl;
    float* Tn = T + (long long)n * KT * HWl;
    // two pixels per lane (tiles P0.. and P0+32..); a pixel past the plane re-reads the last one and is never stored
    const long long pa = min(P0 + li, HWl - 1), pb = min(P0 + 32 + li, HWl - 1);
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;
    constexpr int U = 4;
    // the loads of trip s0 + U are issued before trip s0's MFMAs (a trip's 8 MFMAs are 512 cycles, a loaded round trip more)
    auto fetch = [&](int s0, float (&va)[U], float (&vb)[U]) __attribute__((always_inline)) {
      if constexpr (!GENERIC) {            // C0, C1 multiples of 8: the four pairs of a trip sit in one source
        const float* base = s0 < KS0 ? b0 + (long long)(2 * s0) * HWl : b1 + (long long)(2 * (s0 - KS0)) * HWl;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          va[u] = base[(long long)(2 * u) * HWl + pa];
          vb[u] = base[(long long)(2 * u) * HWl + pb];
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = min(2 * (s0 + u) + lh, C - 1);               // a channel past C carries zero weights
          const float* plane = c < a.C0 ? x0n + (long long)c * HWl : x1n + (long long)(c - a.C0) * HWl;
          va[u] = plane[pa];
          vb[u] = plane[pb];
        }
      }
    };
    auto trip = [&](int s0, const float (&va)[U], const float (&vb)[U]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float scv = ssc[2 * (s0 + u) + lh], shv = ssh[2 * (s0 + u) + lh];
        const float ra = fmaxf(fmaf(va[u], scv, shv), 0.f), rb = fmaxf(fmaf(vb[u], scv, shv), 0.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float wv = sw[(mt * KS + s0 + u) * 64 + lane];
          acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, ra, acc[mt][0], 0, 0, 0);
          acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, rb, acc[mt][1], 0, 0, 0);
        }
      }
    };
    // two register sets, two trips per iteration (KS is a multiple of 8); the scheduling barriers keep each fetch ABOVE the
    // MFMAs it is meant to overlap (left alone the compiler sinks the loads to the bottom of the loop body)
    float xa[U], xb[U], ya[U], yb[U];
    fetch(0, xa, xb);
    for (int s0 = 0; s0 < KS; s0 += 2 * U) {
      fetch(s0 + U, ya, yb);
      __builtin_amdgcn_sched_barrier(0);
      trip(s0, xa, xb);
      __builtin_amdgcn_sched_barrier(0);
      fetch(min(s0 + 2 * U, KS - U), xa, xb);                         // the last iteration re-reads its own second trip
      __builtin_amdgcn_sched_barrier(0);
      trip(s0 + U, ya, yb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // D[m = k][n = pixel]: lane column = pixel, register r = row (r & 3) + 8 * (r >> 2) + 4 * lh
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (k < KT) {
          if (P0 + li < HWl) Tn[(long long)k * HWl + P0 + li] = acc[mt][0][r];
          if (P0 + 32 + li < HWl) Tn[(long long)k * HWl + P0 + 32 + li] = acc[mt][1][r];
        }
      }
  }
}

// forward, step 2: y[n][co][h][w] = bias[co] + sum_{kh,kw} [inside] up(T[n][co*9+kh*3+kw])[h+kh-1][w+kw-1]
template <int COUT>
__global__ __launch_bounds__(256) void head_fwd_lerp_kernel(HeadArgs a, const float* __restrict__ T, const float* __restrict__ bias,
                                                            float* __restrict__ y) {
  const int H = 2 * a.Hl, W = 2 * a.Wl;
  const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
  if (w >= W || h >= H) return;
  const long long HWl = (long long)a.Hl * a.Wl;
  // per axis and tap offset: the two low-res indices and their weights, ZERO for a tap outside the hi-res map (the conv's
  // padding) — no branches, so that all 36 loads of an output channel are in flight together (the branchy form waited for
  // each tap's four loads in turn: 132 us for 108 MB of traffic)
  int r0[3], r1[3], c0[3], c1[3];
  float ur0[3], ur1[3], uc0[3], uc1[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int ho = h + d - 1, wo = w + d - 1;
    const bool inr = ho >= 0 && ho < H, inc = wo >= 0 && wo < W;
    const HLerp qr = head_lerp(inr ? ho : 0, a.Hl, a.rh), qc = head_lerp(inc ? wo : 0, a.Wl, a.rw);
    r0[d] = qr.i0 * a.Wl; r1[d] = qr.i1 * a.Wl;
    c0[d] = qc.i0; c1[d] = qc.i1;
    ur0[d] = inr ? 1.f - qr.l : 0.f; ur1[d] = inr ? qr.l : 0.f;
    uc0[d] = inc ? 1.f - qc.l : 0.f; uc1[d] = inc ? qc.l : 0.f;
  }
  const float* Tn = T + (long long)n * (9 * COUT) * HWl;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    float t[9][4];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float* tp = Tn + (long long)(co * 9 + kh * 3 + kw) * HWl;
        t[kh * 3 + kw][0] = tp[r0[kh] + c0[kw]]; t[kh * 3 + kw][1] = tp[r0[kh] + c1[kw]];
        t[kh * 3 + kw][2] = tp[r1[kh] + c0[kw]]; t[kh * 3 + kw][3] = tp[r1[kh] + c1[kw]];
      }
    float acc = bias ? bias[co] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float top = uc0[kw] * t[kh * 3 + kw][0] + uc1[kw] * t[kh * 3 + kw][1];
        const float bot = uc0[kw] * t[kh * 3 + kw][2] + uc1[kw] * t[kh * 3 + kw][3];
        acc += ur0[kh] * top + ur1[kh] * bot;
      }
    y[(((long long)n * COUT + co) * H + h) * W + w] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// adjoint of (interpolate, shift by the tap, zero-pad): S[n][co*9+kh*3+kw][r][x] =
//     sum_{ho, wo inside the hi-res map} rowcoef(ho, r) * colcoef(wo, x) * dy[n][co][ho+1-kh][wo+1-kw]      (dy = 0 outside)
// The x2 align_corners map has slope < 1/2: low-res row r is touched by the hi-res rows 2r-1 .. 2r+2 only (as the upper
// member of the pair (r-1, r) by 2r-1, 2r and as the lower member of (r, r+1) by 2r+1, 2r+2; a term outside them can only
// carry a rounding-sized weight at the last index and is dropped, as in relu_up2x_bwd).  A thread owns one (co, r, x): a
// 6x6 window of dy, column pass then row pass.  The bias gradient (sum of dy) rides along on each thread's own 2x2 pixels.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_adj_kernel(HeadArgs a, int COUT, const float* __restrict__ dy, float* __restrict__ S,
                                                       float* __restrict__ bpart) {
  const int H = 2 * a.Hl, W = 2 * a.Wl;
  const long long HWl = (long long)a.Hl * a.Wl;
  const int co = blockIdx.y, n = blockIdx.z;
  const long long P = (long long)blockIdx.x * 256 + threadIdx.x;
  float own = 0.f;
  if (P < HWl) {
    const int r = (int)(P / a.Wl), x = (int)(P % a.Wl);
    float ar[4], bc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ar[j] = head_coef(2 * r - 1 + j, r, a.Hl, a.rh);
      bc[j] = head_coef(2 * x - 1 + j, x, a.Wl, a.rw);
    }
    const float* plane = dy + ((long long)n * COUT + co) * H * W;
    float Cq[6][3];                                     // column pass: Cq[jj][kw] = sum_i bc[i] * D[jj][i - kw + 2]
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      const int hh = 2 * r - 2 + jj;
      float D[6];
#pragma unroll
      for (int q = 0; q < 3; ++q) {                     // columns 2x-2+2q, +1: W is even, a pair is all-in or all-out
        const int ww = 2 * x - 2 + 2 * q;
        float2 v = make_float2(0.f, 0.f);
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = *reinterpret_cast<const float2*>(plane + (long long)hh * W + ww);
        D[2 * q] = v.x; D[2 * q + 1] = v.y;
      }
      if (jj == 2 || jj == 3) own += D[2] + D[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) v = fmaf(bc[i], D[i - kw + 2], v);
        Cq[jj][kw] = v;
      }
    }
    float* Sn = S + ((long long)n * COUT + co) * 9 * HWl + P;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) v = fmaf(ar[j], Cq[j - kh + 2][kw], v);
        Sn[(long long)(kh * 3 + kw) * HWl] = v;
      }
  }
  if (bpart) {                                          // block total of dy -> one partial per (n, block, co)
    __shared__ float red[4];
    const float v = wave_sum(own);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) bpart[((long long)n * gridDim.x + blockIdx.x) * COUT + co] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient to the low-res sources: g[c][P] = relu'(affine(x[c][P])) * sum_k w[k][c] * S[k][P]  (+ g0 when acc0);
// source-1 channels also reduce the BatchNorm-backward sums (sum g, sum g * xhat).  Pure streaming over flat pixels: a lane
// holds the KT values of S for its four pixels, the channel loop runs with wave-uniform weights (scalar loads), x of the
// next channels is in flight while this channel's FMAs run.  blockIdx.z = source; a block sweeps `tpb` pixel tiles of 256
// and adds its statistics to the global sums once.
// ---------------------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void head_dgrad_kernel(HeadArgs a, const float* __restrict__ w, const float* __restrict__ S,
                                                         float* __restrict__ g0, float* __restrict__ g1,
                                                         const float* __restrict__ mean1, const float* __restrict__ invstd1,
                                                         double* bstats1, int acc0, int tpb) {
  constexpr int KT = 9 * COUT;
  extern __shared__ double sstat[];       // [2][C1] (source 1 with statistics only)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y, src = blockIdx.z;
  const int C = a.C0 + a.C1, Cs = src ? a.C1 : a.C0, cofs = src ? a.C0 : 0;
  float* g = src ? g1 : g0;
  if (!g || Cs == 0) return;
  const bool stats = src && bstats1;
  if (stats) {
    for (int e = tid; e < 2 * a.C1; e += 256) sstat[e] = 0.0;
    __syncthreads();
  }
  const long long HWl = (long long)a.Hl * a.Wl;
  const float* xs = (src ? a.x1 : a.x0) + (long long)n * Cs * HWl;
  float* gs = g + (long long)n * Cs * HWl;
  const float* sc = src ? a.sc1 : a.sc0;
  const float* sh = src ? a.sh1 : a.sh0;
  const float* Sn = S + (long long)n * KT * HWl;
  const bool accum = acc0 && !src;
  const long long tiles = (HWl + 255) / 256;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int it = wave; it < tpb; it += 4) {
    const long long tile = (long long)blockIdx.x * tpb + it;
    if (tile >= tiles) break;                           // wave-uniform
    const long long P = tile * 256 + 4 * lane;
    const bool ok = P < HWl;                            // HWl % 4 == 0: a quad is all-in or all-out
    const long long Pc = ok ? P : 0;
    f32x4 sv[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      sv[k] = *reinterpret_cast<const f32x4*>(Sn + (long long)k * HWl + Pc);
      if (!ok) sv[k] = z4;
    }
    // x of channels c+1 and c+2 is in flight while channel c's FMAs run (one 1 KB load per wave and channel)
    f32x4 xq0 = *reinterpret_cast<const f32x4*>(xs + Pc);
    f32x4 xq1 = *reinterpret_cast<const f32x4*>(xs + (long long)min(1, Cs - 1) * HWl + Pc);
    for (int c = 0; c < Cs; ++c) {
      const f32x4 xv = xq0;
      xq0 = xq1;
      xq1 = *reinterpret_cast<const f32x4*>(xs + (long long)min(c + 2, Cs - 1) * HWl + Pc);
      f32x4 old = z4;
      if (accum) old = *reinterpret_cast<const f32x4*>(gs + (long long)c * HWl + Pc);
      const float* wc = w + (long long)(cofs + c) * 9;
      f32x4 gv = z4;
#pragma unroll
      for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float wv = wc[(long long)co * C * 9 + t];
          gv += wv * sv[co * 9 + t];
        }
      const float scv = sc ? sc[c] : 1.f, shv = sc ? sh[c] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) gv[e] = fmaf(xv[e], scv, shv) > 0.f ? gv[e] : 0.f;
      if (stats) {
        const float mu = mean1[c], is = invstd1[c];
        // 32-lane sums on the DPP path, valid in lanes 31 and 63 (common.h); the two halves meet in scalar registers
        const float h1 = half_sum_hi((gv[0] + gv[1]) + (gv[2] + gv[3]));
        const float h2 = half_sum_hi((gv[0] * (xv[0] - mu) + gv[1] * (xv[1] - mu)) + (gv[2] * (xv[2] - mu) + gv[3] * (xv[3] - mu)));
        const float s1 = lane_value(h1, 31) + lane_value(h1, 63);
        const float s2 = (lane_value(h2, 31) + lane_value(h2, 63)) * is;
        if (lane == 0) {
          atomicAdd(&sstat[c], (double)s1);
          atomicAdd(&sstat[a.C1 + c], (double)s2);
        }
      }
      if (ok) *reinterpret_cast<f32x4*>(gs + (long long)c * HWl + P) = gv + old;
    }
  }
  if (stats) {
    __syncthreads();
    for (int e = tid; e < 2 * a.C1; e += 256) atomicAdd(&bstats1[e], sstat[e]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dw[k][c] = sum_{n,P} S[n][k][P] * relu(affine(x[n][c][P])).  GEMM with K = pixels: M = channels (a wave
// owns 32), N = k (NT 32-column tiles).  Both operands are pixel-contiguous in memory and the MFMA takes one value per lane
// and k, so lane (i, h) loads a float4 of ITS row (channel i / plane i) at pixels p + 4h .. p + 4h + 3 and the four
// components are four k-steps (the k <-> pixel pairing only has to agree between A and B).  No LDS; the four waves of a
// block (four channel tiles) share the S rows through L1.  One partial slab per (n, pixel segment), reduced afterwards.
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void head_wgrad_kernel(HeadArgs a, int KT, const float* __restrict__ S, float* __restrict__ part,
                                                         int SEG) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int C = a.C0 + a.C1, n = blockIdx.z, seg = blockIdx.y;
  const int m0 = (blockIdx.x * 4 + wave) * 32;
  if (m0 >= C) return;
  const long long HWl = (long long)a.Hl * a.Wl;
  const long long seglen = ((HWl / 4 + SEG - 1) / SEG) * 4, p_beg = seg * seglen, p_end = min(HWl, p_beg + seglen);
  const int c = min(m0 + li, C - 1);
  const bool cok = m0 + li < C, first = c < a.C0;
  const int cs = first ? c : c - a.C0;
  const float* scp = first ? a.sc0 : a.sc1;
  const float* shp = first ? a.sh0 : a.sh1;
  const float scv = scp ? scp[cs] : 1.f, shv = scp ? shp[cs] : 0.f;
  const float* xrow = (first ? a.x0 + ((long long)n * a.C0 + cs) * HWl : a.x1 + ((long long)n * a.C1 + cs) * HWl);
  const float* srow[NT];
  bool kok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = t * 32 + li;
    kok[t] = k < KT;
    srow[t] = S + ((long long)n * KT + min(k, KT - 1)) * HWl;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  // a trip is 32 pixels: lane half h takes pixels 16h .. 16h+15 of its row as four float4 (64 contiguous bytes: a row's
  // 128-byte line is consumed within the trip, nothing has to survive in L1 between trips); component j of quad q is
  // k-step 4q + j.  Out-of-range quads load as zeros on the S side, which zeroes the product (relu(affine(0)) is not 0).
  constexpr int Q = 4;
  auto load_a = [&](long long p, f32x4 (&v)[Q]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = (cok && p + 4 * q < p_end) ? *reinterpret_cast<const f32x4*>(xrow + p + 4 * q) : z4;
  };
  auto load_b = [&](int t, long long p, f32x4 (&v)[Q]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = (kok[t] && p + 4 * q < p_end) ? *reinterpret_cast<const f32x4*>(srow[t] + p + 4 * q) : z4;
  };
  long long p = p_beg + 16 * lh;
  f32x4 na[Q], nb[NT][Q];
  load_a(p, na);
#pragma unroll
  for (int t = 0; t < NT; ++t) load_b(t, p, nb[t]);
  for (long long q0 = p_beg; q0 < p_end; q0 += 32) {
    f32x4 ca[Q], cb[NT][Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      ca[q] = na[q];
#pragma unroll
      for (int t = 0; t < NT; ++t) cb[t][q] = nb[t][q];
    }
    p += 32;
    load_a(p, na);                                       // the next trip's loads fly during this trip's 16 * NT MFMAs
#pragma unroll
    for (int t = 0; t < NT; ++t) load_b(t, p, nb[t]);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r = fmaxf(fmaf(ca[q][j], scv, shv), 0.f);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r, cb[t][q][j], acc[t], 0, 0, 0);
      }
  }
  // D[m = channel][n = k]: lane column = k, register r = channel (r & 3) + 8 * (r >> 2) + 4 * lh
  float* slab = part + ((long long)n * SEG + seg) * KT * C;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int k = t * 32 + li;
    if (k < KT) {
      const int co = k / 9, tap = k % 9;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cc = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (cc < C) slab[((long long)co * C + cc) * 9 + tap] = acc[t][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------------------------------------------
int reduce_slabs_strided(const float* ws, float* out, long long n, int S, long long stride, hipStream_t st);   // conv.hip

#define HD_MAXCO 4
bool head_applicable(const avsep_conv_desc* d) {
  return d->up2x && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && d->Cout <= HD_MAXCO &&
         d->act0 == AVSEP_ACT_RELU && (d->C0 == d->Cin || d->act1 == AVSEP_ACT_RELU) && (d->W & 3) == 0 && (d->H & 3) == 0 &&
         d->Cin <= 256 && d->N <= 65535 && d->H / 4 <= 65535 &&
         (long long)d->Cin * (d->H / 2) * (d->W / 2) < 0x7fffffffLL;
}
static HeadArgs head_args(const avsep_conv_desc* d) {
  HeadArgs a{};
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Hl = d->H / 2; a.Wl = d->W / 2;
  a.rh = (float)(a.Hl - 1) / (float)(d->H - 1);
  a.rw = (float)(a.Wl - 1) / (float)(d->W - 1);
  return a;
}
static inline size_t head_planes_floats(const avsep_conv_desc* d) {       // T or S: [N][9*Cout][Hl][Wl]
  return (size_t)d->N * 9 * d->Cout * (d->H / 2) * (d->W / 2);
}
size_t head_fwd_workspace_floats(const avsep_conv_desc* d) { return head_planes_floats(d); }
size_t head_dgrad_workspace_floats(const avsep_conv_desc* d) { return head_planes_floats(d); }

int head_fwd(const avsep_conv_desc* d, const float* wp, int wp_ld, const float* bias, float* y, float* ws, hipStream_t st) {
  HeadArgs a = head_args(d);
  const long long HWl = (long long)a.Hl * a.Wl;
  const int KT = 9 * d->Cout, MT = (KT + 31) / 32, KS = (d->Cin + 15) / 16 * 8;
  const bool generic = (a.C0 & 7) || (a.C1 & 7) || (d->Cin & 15);
  const long long total = (HWl + 63) / 64 * d->N;                            // 64-pixel wave tiles
  const size_t lds = ((size_t)MT * KS * 64 + 4 * KS) * sizeof(float);       // <= 133 KB at Cin = 256, two row tiles
  // one round of resident blocks (4 per CU by registers at 16 KB of LDS, fewer when the weights take more)
  int per_cu = (int)(160 * 1024 / (lds + 1024));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  long long nb = (long long)cu_count() * per_cu;
  if (nb > (total + 3) / 4) nb = (total + 3) / 4;
  dim3 g1((unsigned)nb);
  if (lds > 64 * 1024) {   // set per call for the instantiation about to launch (no process state: any thread, any device)
    const void* f = MT == 1 ? (generic ? (const void*)head_fwd_gemm_kernel<1, true> : (const void*)head_fwd_gemm_kernel<1, false>)
                            : (generic ? (const void*)head_fwd_gemm_kernel<2, true> : (const void*)head_fwd_gemm_kernel<2, false>);
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AVSEP_ERR_LAUNCH;
  }
#define HEAD_FG(MT_, G_) hipLaunchKernelGGL((head_fwd_gemm_kernel<MT_, G_>), g1, dim3(256), lds, st, a, wp, wp_ld, KT, ws)
  if (MT == 1) { if (generic) HEAD_FG(1, true); else HEAD_FG(1, false); }
  else { if (generic) HEAD_FG(2, true); else HEAD_FG(2, false); }
#undef HEAD_FG
  AVSEP_LAUNCH_CHECK();
  dim3 g2(cdiv(d->W, 64), cdiv(d->H, 4), d->N);
  switch (d->Cout) {
    case 1: hipLaunchKernelGGL(head_fwd_lerp_kernel<1>, g2, dim3(256), 0, st, a, ws, bias, y); break;
    case 2: hipLaunchKernelGGL(head_fwd_lerp_kernel<2>, g2, dim3(256), 0, st, a, ws, bias, y); break;
    case 3: hipLaunchKernelGGL(head_fwd_lerp_kernel<3>, g2, dim3(256), 0, st, a, ws, bias, y); break;
    default: hipLaunchKernelGGL(head_fwd_lerp_kernel<4>, g2, dim3(256), 0, st, a, ws, bias, y); break;
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

static int head_adjoint(const avsep_conv_desc* d, const HeadArgs& a, const float* dy, float* S, float* bpart, hipStream_t st) {
  const long long HWl = (long long)a.Hl * a.Wl;
  hipLaunchKernelGGL(head_adj_kernel, dim3(cdiv(HWl, 256), d->Cout, d->N), dim3(256), 0, st, a, d->Cout, dy, S, bpart);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// pixel segments of the weight-gradient GEMM: enough waves for two rounds of the chip, at least 512 pixels each
static int head_wgrad_segments(const avsep_conv_desc* d) {
  const long long HWl = (long long)(d->H / 2) * (d->W / 2);
  const long long waves = (long long)cdiv(d->Cin, 32) * d->N;
  int seg = 1;
  while (seg < 64 && waves * seg < 2048 && HWl / (2 * seg) >= 512) seg *= 2;
  return seg;
}
size_t head_wgrad_workspace_floats(const avsep_conv_desc* d) {
  const long long HWl = (long long)(d->H / 2) * (d->W / 2);
  const int SEG = head_wgrad_segments(d);
  return head_planes_floats(d) + (size_t)d->N * SEG * d->Cout * d->Cin * 9 + (size_t)d->N * cdiv(HWl, 256) * d->Cout;
}
int head_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias, float* ws, hipStream_t st) {
  HeadArgs a = head_args(d);
  const long long HWl = (long long)a.Hl * a.Wl;
  const int KT = 9 * d->Cout, NT = (KT + 31) / 32, SEG = head_wgrad_segments(d), nw = d->Cout * d->Cin * 9;
  float* S = ws;
  float* part = S + head_planes_floats(d);
  float* bpart = part + (size_t)d->N * SEG * nw;
  int rc = head_adjoint(d, a, dy, S, dbias ? bpart : nullptr, st);
  if (rc) return rc;
  dim3 grid(cdiv(d->Cin, 128), SEG, d->N);
  if (NT == 1) hipLaunchKernelGGL(head_wgrad_kernel<1>, grid, dim3(256), 0, st, a, KT, S, part, SEG);
  else hipLaunchKernelGGL(head_wgrad_kernel<2>, grid, dim3(256), 0, st, a, KT, S, part, SEG);
  AVSEP_LAUNCH_CHECK();
  rc = reduce_slabs_strided(part, dw, nw, d->N * SEG, nw, st);
  if (rc) return rc;
  if (dbias) return reduce_slabs_strided(bpart, dbias, d->Cout, d->N * cdiv(HWl, 256), d->Cout, st);
  return AVSEP_OK;
}

int head_dgrad(const avsep_conv_desc* d, const float* w, const float* dy, float* g0, float* g1, const float* mean1,
               const float* invstd1, double* bstats1, int acc0, float* ws, hipStream_t st) {
  HeadArgs a = head_args(d);
  const long long HWl = (long long)a.Hl * a.Wl;
  int rc = head_adjoint(d, a, dy, ws, nullptr, st);
  if (rc) return rc;
  const long long tiles = (HWl + 255) / 256;          // a wave streams 256 pixels (one float4 per lane) per tile
  const int tpb = tiles >= 32 ? 8 : 4;
  dim3 grid(cdiv(tiles, tpb), d->N, 2);
  const size_t lds = bstats1 ? (size_t)2 * a.C1 * sizeof(double) : 0;
#define HEAD_DG(CO_) hipLaunchKernelGGL(head_dgrad_kernel<CO_>, grid, dim3(256), lds, st, a, w, (const float*)ws, g0, g1, mean1, invstd1, bstats1, acc0, tpb)
  switch (d->Cout) {
    case 1: HEAD_DG(1); break;
    case 2: HEAD_DG(2); break;
    case 3: HEAD_DG(3); break;
    default: HEAD_DG(4); break;
  }
#undef HEAD_DG
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
