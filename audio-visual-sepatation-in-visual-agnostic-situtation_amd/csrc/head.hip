// Fused decoder head: the U-Net's outermost up path (audio_net.py:72-76: ReLU -> bilinear x2 (align_corners) ->
// conv 3x3 p1 -> num_mix logits) on the concat of the skip tensor and the inner block's BatchNorm'd output.
//
// The hi-res input U = up2x(relu(affine(cat(x0, x1)))) has 128 channels at 256x256 (33.5 MB per sample) but feeds
// only <= 4 output channels: materialising it costs a 1 GB write + two 1 GB reads per pass at batch 32, and its
// gradient the same again.  These three kernels never materialise U or dU: every wave rebuilds the rows of U it
// needs from the LOW-RES sources in registers while it sweeps down the image.
//
// Register layout (no LDS, no barriers in the sweeps): lane q owns low-res columns 2q, 2q+1 = hi-res columns
// 4q..4q+3 (W <= 256 fits one wave).  Because the x2 align_corners map is monotone with slope < 1/2, hi-res column
// 4q+j always interpolates a STATIC pair of the four low-res values L0..L3 = columns 2q-1..2q+2:
//   j=-1,0 -> (L0,L1)   j=1,2 -> (L1,L2)   j=3,4 -> (L2,L3)          (same for rows: 2r+1, 2r+2 -> (r, r+1))
// so a lane builds the six hi-res columns 4q-1..4q+4 (its 3x3 halo included) from its own float2 plus one value
// from each neighbour lane, with per-lane constant coefficients.  M = Cout <= 4 is far too small for an MFMA
// tile: the FMAs run on the vector ALU, weights come through the scalar unit (wave-uniform addresses).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#define HD_MAXCO 4

// compile-time slot ids of the rolling 3-row windows (a runtime rotation would cost a register copy per row)
using Slot0 = std::integral_constant<int, 0>;
using Slot1 = std::integral_constant<int, 1>;
using Slot2 = std::integral_constant<int, 2>;
using Slot3 = std::integral_constant<int, 3>;

struct HeadArgs {
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  int N, C0, C1, Hl, Wl;   // low-res geometry; hi-res is 2*Hl x 2*Wl
  float rh, rw;            // (L-1)/(2L-1) per axis
};

// Coefficients of hi-res index o (axis of low-res length L, ratio r) on the static low-res pair (lo, lo+1).
// The reference value is (1-l)*v[i0] + l*v[i1] with i0 = floor(r*o), i1 = min(i0+1, L-1) (float arithmetic as in
// the unfused kernels); a term that falls outside the pair (only possible with weight ~0 from float rounding at the
// last index) is dropped.  o outside [0, 2L) is the conv's zero padding: both coefficients 0.
__device__ __forceinline__ void lerp_pair(int o, int lo, int L, float r, float& ca, float& cb) {
  const float f = r * (float)o;
  const int i0 = (int)f;
  const float l = f - (float)i0;
  const int i1 = i0 + (i0 < L - 1);
  const bool in = o >= 0 && o < 2 * L;
  ca = in ? ((i0 == lo ? 1.f - l : 0.f) + (i1 == lo ? l : 0.f)) : 0.f;
  cb = in ? ((i0 == lo + 1 ? 1.f - l : 0.f) + (i1 == lo + 1 ? l : 0.f)) : 0.f;
}

// column coefficients of hi-res columns 4q-1+k (k = 0..5) on their static pair; plain register arrays
#define HEAD_LANE_COEFFS(a, lane, ca, cb)                                                          \
  float ca[6], cb[6];                                                                              \
  _Pragma("unroll") for (int k_ = 0; k_ < 6; ++k_) {                                              \
    float ca_, cb_;                                                                                \
    lerp_pair(4 * (lane) - 1 + k_, 2 * (lane) - 1 + (k_ >> 1), (a).Wl, (a).rw, ca_, cb_);         \
    ca[k_] = ca_; cb[k_] = cb_;                                                                    \
  }

// Horizontally interpolated low-res row r of one channel plane: Hout[k] = hi-res column 4q-1+k (before the row lerp).
__device__ __forceinline__ void head_hrow(const float* __restrict__ plane, int r, int Hl, int Wl, bool active,
                                          const float (&ca)[6], const float (&cb)[6], float scv, float shv, int lane,
                                          float (&Hout)[6]) {
  float l1 = 0.f, l2 = 0.f;
  if (active && r >= 0 && r < Hl) {
    const float2 v = *reinterpret_cast<const float2*>(plane + (long long)r * Wl + 2 * lane);
    l1 = fmaxf(fmaf(v.x, scv, shv), 0.f);
    l2 = fmaxf(fmaf(v.y, scv, shv), 0.f);
  }
  float l0 = __shfl_up(l2, 1, 64), l3 = __shfl_down(l1, 1, 64);
  if (lane == 0) l0 = 0.f;
  if (lane == 63) l3 = 0.f;                       // (an inactive neighbour already holds 0)
  Hout[0] = ca[0] * l0 + cb[0] * l1;
  Hout[1] = ca[1] * l0 + cb[1] * l1;
  Hout[2] = ca[2] * l1 + cb[2] * l2;
  Hout[3] = ca[3] * l1 + cb[3] * l2;
  Hout[4] = ca[4] * l2 + cb[4] * l3;
  Hout[5] = ca[5] * l2 + cb[5] * l3;
}

// the same in two halves, so that a sweep can issue the load of row r+1 one iteration before it needs it: the kernels below
// run at two waves per SIMD and a load consumed right after its issue left them parked on s_waitcnt for half of their
// cycles (SQ_WAIT_ANY 0.46-0.58 of SQ_WAVE_CYCLES)
__device__ __forceinline__ float2 head_hrow_load(const float* __restrict__ plane, int r, int Hl, int Wl, bool active, int lane) {
  float2 v = make_float2(0.f, 0.f);
  if (active && r >= 0 && r < Hl) v = *reinterpret_cast<const float2*>(plane + (long long)r * Wl + 2 * lane);
  return v;
}
__device__ __forceinline__ void head_hrow_build(float2 v, int r, int Hl, bool active, const float (&ca)[6], const float (&cb)[6],
                                                float scv, float shv, int lane, float (&Hout)[6]) {
  float l1 = 0.f, l2 = 0.f;
  if (active && r >= 0 && r < Hl) {
    l1 = fmaxf(fmaf(v.x, scv, shv), 0.f);
    l2 = fmaxf(fmaf(v.y, scv, shv), 0.f);
  }
  float l0 = __shfl_up(l2, 1, 64), l3 = __shfl_down(l1, 1, 64);
  if (lane == 0) l0 = 0.f;
  if (lane == 63) l3 = 0.f;
  Hout[0] = ca[0] * l0 + cb[0] * l1;
  Hout[1] = ca[1] * l0 + cb[1] * l1;
  Hout[2] = ca[2] * l1 + cb[2] * l2;
  Hout[3] = ca[3] * l1 + cb[3] * l2;
  Hout[4] = ca[4] * l2 + cb[4] * l3;
  Hout[5] = ca[5] * l2 + cb[5] * l3;
}

struct HeadChan { const float* plane; float scv, shv; bool first; int cs; };
__device__ __forceinline__ HeadChan head_chan(const HeadArgs& a, int n, int c) {
  HeadChan h;
  h.first = c < a.C0;
  h.cs = h.first ? c : c - a.C0;
  const float* sc = h.first ? a.sc0 : a.sc1;
  const float* sh = h.first ? a.sh0 : a.sh1;
  h.scv = sc ? sc[h.cs] : 1.f;
  h.shv = sc ? sh[h.cs] : 0.f;
  h.plane = (h.first ? a.x0 + ((long long)n * a.C0 + h.cs) * a.Hl * a.Wl
                     : a.x1 + ((long long)n * a.C1 + h.cs) * a.Hl * a.Wl);
  return h;
}

// ---------------------------------------------------------------------------------------------------------------
// forward: y[n,co,h,w] = bias[co] + sum_{c,kh,kw} w[co,c,kh,kw] * U[n,c,h+kh-1,w+kw-1]
// block = R hi-res rows of one image; its 4 waves split the channels and are summed through LDS at the end.
// ---------------------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256, 4) void head_fwd_kernel(HeadArgs a, const float* __restrict__ wp, int wp_ld,
                                                       const float* __restrict__ bias, float* __restrict__ y) {
  constexpr int R = COUT <= 2 ? 4 : 2, NACC = R * 4 * COUT, NU = R + 2;
  __shared__ float red[2][NACC][64];
  __shared__ float rwa[NU], rwb[NU];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.y, g = blockIdx.x;
  const int H = 2 * a.Hl, W = 2 * a.Wl, C = a.C0 + a.C1;
  const bool active = 2 * lane < a.Wl;
  HEAD_LANE_COEFFS(a, lane, ca, cb)
  const int r_first = (R * g) / 2 - 1;                    // U row R*g-1 = 2*r_first+1
  if (threadIdx.x < NU) {
    const int u = threadIdx.x;
    lerp_pair(R * g - 1 + u, r_first + (u >> 1), a.Hl, a.rh, rwa[u], rwb[u]);
  }
  __syncthreads();
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  // channel loop, software-pipelined: the NROW low-res rows and the 9*COUT weights of channel c+1 are loaded while channel
  // c's FMAs run (a row consumed right after its load left the waves parked on s_waitcnt half of the time)
  constexpr int NROW = NU / 2 + 1;
  const int cper = (C + 3) / 4, c_beg = wave * cper, c_end = min(C, (wave + 1) * cper);
  float2 raw[NROW];
  float wn[COUT][9];
  auto fetch = [&](int c) __attribute__((always_inline)) {
    const HeadChan ch = head_chan(a, n, min(c, C - 1));
#pragma unroll
    for (int q = 0; q < NROW; ++q) raw[q] = head_hrow_load(ch.plane, r_first + q, a.Hl, a.Wl, active, lane);
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) wn[co][t] = wp[(long long)(min(c, C - 1) * 9 + t) * wp_ld + co];
  };
  if (c_beg < c_end) fetch(c_beg);
  for (int c = c_beg; c < c_end; ++c) {
    const HeadChan ch = head_chan(a, n, c);
    float wv[COUT][9];
    float2 rw[NROW];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) wv[co][t] = wn[co][t];
#pragma unroll
    for (int q = 0; q < NROW; ++q) rw[q] = raw[q];
    fetch(c + 1);                                          // clamped: the last trip re-reads channel C-1
    float Hp[6], Hc[6];
    head_hrow_build(rw[0], r_first, a.Hl, active, ca, cb, ch.scv, ch.shv, lane, Hp);
#pragma unroll
    for (int p = 0; p < NU / 2; ++p) {
      head_hrow_build(rw[1 + p], r_first + 1 + p, a.Hl, active, ca, cb, ch.scv, ch.shv, lane, Hc);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = 2 * p + i;
        const float wa = rwa[u], wb = rwb[u];
        float U[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) U[k] = wa * Hp[k] + wb * Hc[k];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int r = u - kh;                           // output row R*g + r reads U row u through tap kh
          if (r >= 0 && r < R) {
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int px = 0; px < 4; ++px)
                  acc[(r * 4 + px) * COUT + co] = fmaf(wv[co][kh * 3 + kw], U[px + kw], acc[(r * 4 + px) * COUT + co]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) Hp[k] = Hc[k];
    }
  }
  // 4 -> 2 -> 1 waves
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) red[wave - 2][i][lane] = acc[i];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] += red[wave][i][lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) red[0][i][lane] = acc[i];
  }
  __syncthreads();
  if (wave == 0 && active) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int h = R * g + r;
      if (h < H) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float b = bias ? bias[co] : 0.f;
          f32x4 o;
          o.x = acc[(r * 4 + 0) * COUT + co] + red[0][(r * 4 + 0) * COUT + co][lane] + b;
          o.y = acc[(r * 4 + 1) * COUT + co] + red[0][(r * 4 + 1) * COUT + co][lane] + b;
          o.z = acc[(r * 4 + 2) * COUT + co] + red[0][(r * 4 + 2) * COUT + co][lane] + b;
          o.w = acc[(r * 4 + 3) * COUT + co] + red[0][(r * 4 + 3) * COUT + co][lane] + b;
          *reinterpret_cast<f32x4*>(y + (((long long)n * COUT + co) * H + h) * W + 4 * lane) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dw[co,c,kh,kw] = sum_{n,h,w} dy[n,co,h,w] * U[n,c,h+kh-1,w+kw-1]   (+ dbias = sum dy)
// wave = (CPW consecutive channels, image n, segment of SR hi-res output rows); a rolling window of three dy rows —
// loaded ONCE per wave and row, shared by its CPW channels (with one channel per wave every dy row was re-read by all
// 128 channel waves: 4.3 GB of L2 traffic per call at batch 64) — meets each rebuilt U row; 9*COUT accumulators per
// lane and channel, wave-reduced into one partial slab per (n, segment).
// ---------------------------------------------------------------------------------------------------------------
template <int COUT, int CPW>
__global__ __launch_bounds__(256, COUT * CPW <= 2 ? 3 : 2) void head_wgrad_kernel(HeadArgs a, const float* __restrict__ dy,
                                                                            float* __restrict__ part, float* __restrict__ bpart,
                                                                            int S) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C0 + a.C1, c0 = (blockIdx.x * 4 + wave) * CPW, n = blockIdx.y, seg = blockIdx.z;
  if (c0 >= C) return;
  const int H = 2 * a.Hl, W = 2 * a.Wl, SR = H / S, s0 = seg * SR, s1 = s0 + SR;
  const bool active = 2 * lane < a.Wl;
  HEAD_LANE_COEFFS(a, lane, ca, cb)
  HeadChan ch[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) ch[j] = head_chan(a, n, min(c0 + j, C - 1));      // a channel past C re-reads the last one, never stored
  float acc[CPW][COUT][9], bacc[COUT];
  f32x4 Dm[COUT], D0[COUT], Dp[COUT];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    bacc[co] = 0.f;
    Dm[co] = zero4; D0[co] = zero4;
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[j][co][t] = 0.f;
  }
  // dy rows: loaded ONE U ROW ahead of their first use (Dn is in flight while the row's FMAs run), summed for the bias
  // gradient when they enter the window
  auto load_dy = [&](int h, f32x4 (&D)[COUT]) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      D[co] = zero4;
      if (active && h >= s0 && h < s1)
        D[co] = *reinterpret_cast<const f32x4*>(dy + (((long long)n * COUT + co) * H + h) * W + 4 * lane);
    }
  };
  f32x4 Dn[COUT];
  load_dy(s0, Dp);
  load_dy(s0 + 1, Dn);
#pragma unroll
  for (int co = 0; co < COUT; ++co) bacc[co] += (Dp[co].x + Dp[co].y) + (Dp[co].z + Dp[co].w);
  int r = s0 / 2 - 1;
  float Hp[CPW][6], Hc[CPW][6];
  float2 raw[CPW];                                         // low-res row r+1 of each channel, loaded one pair ahead
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    head_hrow(ch[j].plane, r, a.Hl, a.Wl, active, ca, cb, ch[j].scv, ch[j].shv, lane, Hp[j]);
    raw[j] = head_hrow_load(ch[j].plane, r + 1, a.Hl, a.Wl, active, lane);
  }
  for (int p = 0; p <= SR / 2; ++p, ++r) {                 // pair (r, r+1) -> U rows 2r+1, 2r+2
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      head_hrow_build(raw[j], r + 1, a.Hl, active, ca, cb, ch[j].scv, ch[j].shv, lane, Hc[j]);
      raw[j] = head_hrow_load(ch[j].plane, r + 2, a.Hl, a.Wl, active, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ho = 2 * r + 1 + i;
      float wa, wb;
      lerp_pair(ho, r, a.Hl, a.rh, wa, wb);
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        float U[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) U[k] = wa * Hp[j][k] + wb * Hc[j][k];
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            // tap kh pairs U row ho with output row ho+1-kh: kh=0 -> Dp, 1 -> D0, 2 -> Dm
            acc[j][co][kw] += (Dp[co].x * U[kw] + Dp[co].y * U[kw + 1]) + (Dp[co].z * U[kw + 2] + Dp[co].w * U[kw + 3]);
            acc[j][co][3 + kw] += (D0[co].x * U[kw] + D0[co].y * U[kw + 1]) + (D0[co].z * U[kw + 2] + D0[co].w * U[kw + 3]);
            acc[j][co][6 + kw] += (Dm[co].x * U[kw] + Dm[co].y * U[kw + 1]) + (Dm[co].z * U[kw + 2] + Dm[co].w * U[kw + 3]);
          }
      }
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        Dm[co] = D0[co]; D0[co] = Dp[co]; Dp[co] = Dn[co];
        bacc[co] += (Dp[co].x + Dp[co].y) + (Dp[co].z + Dp[co].w);
      }
      load_dy(ho + 3, Dn);
    }
#pragma unroll
    for (int j = 0; j < CPW; ++j)
#pragma unroll
      for (int k = 0; k < 6; ++k) Hp[j][k] = Hc[j][k];
  }
  const long long slab = ((long long)n * S + seg) * COUT * C * 9;
#pragma unroll
  for (int j = 0; j < CPW; ++j)
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float v = wave_sum(acc[j][co][t]);
        if (lane == 0 && c0 + j < C) part[slab + ((long long)co * C + c0 + j) * 9 + t] = v;
      }
  if (bpart && c0 == 0) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float v = wave_sum(bacc[co]);
      if (lane == 0) bpart[((long long)n * S + seg) * COUT + co] = v;
    }
  }
}

int reduce_slabs_strided(const float* ws, float* out, long long n, int S, long long stride, hipStream_t st);   // conv.hip

// ---------------------------------------------------------------------------------------------------------------
// data gradient straight to the low-res sources:
//   dU[c,ho,wo] = sum_{co,kh,kw} w[co,c,kh,kw] * dy[co,ho+1-kh,wo+1-kw]
//   g[c,r,x]    = relu'(affine(src[c,r,x])) * sum_{ho,wo} rowcoef(ho,r) * colcoef(wo,x) * dU[c,ho,wo]
// wave = (CPW consecutive channels, image n, segment of low-res rows).  A lane forms dU at the six hi-res columns
// 4q-1..4q+4 that touch its own two low-res columns (no cross-lane scatter) from a rolling 3-row window of dy (8 columns
// wide) that the CPW channels share.  Source-1 channels also reduce the BatchNorm-backward sums (sum g, sum g*xhat)
// like relu_up2x_bwd.
// ---------------------------------------------------------------------------------------------------------------
template <int COUT, int CPW>
__global__ __launch_bounds__(256, COUT * CPW <= 2 ? 3 : 2) void head_dgrad_kernel(HeadArgs a, const float* __restrict__ w,
                                                                            const float* __restrict__ dy, float* __restrict__ g0,
                                                                            float* __restrict__ g1, const float* __restrict__ mean1,
                                                                            const float* __restrict__ invstd1, double* bstats1,
                                                                            int acc0, int S) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C0 + a.C1, c0 = (blockIdx.x * 4 + wave) * CPW, n = blockIdx.y, seg = blockIdx.z;
  if (c0 >= C) return;
  const int H = 2 * a.Hl, W = 2 * a.Wl, RL = a.Hl / S, r0 = seg * RL, r1 = r0 + RL;
  const bool active = 2 * lane < a.Wl;
  HEAD_LANE_COEFFS(a, lane, ca, cb)
  const long long HWl = (long long)a.Hl * a.Wl;
  HeadChan ch[CPW];
  float* gp[CPW];                                         // destination plane of channel j (null: not wanted / past C)
  const float* xp[CPW];
  float mu[CPW], is[CPW], wv[CPW][COUT][9];
  bool stats[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int c = min(c0 + j, C - 1);
    ch[j] = head_chan(a, n, c);
    const long long pbase = ((long long)n * (ch[j].first ? a.C0 : a.C1) + ch[j].cs) * HWl;
    float* g = ch[j].first ? g0 : g1;
    gp[j] = (g && c0 + j < C) ? g + pbase : nullptr;
    xp[j] = (ch[j].first ? a.x0 : a.x1) + pbase;
    stats[j] = !ch[j].first && bstats1 && gp[j];
    mu[j] = stats[j] ? mean1[ch[j].cs] : 0.f;
    is[j] = stats[j] ? invstd1[ch[j].cs] : 1.f;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int t = 0; t < 9; ++t) wv[j][co][t] = w[((long long)co * C + c) * 9 + t];
  }
  float E[4][COUT][8];                                    // rolling window of dy rows (columns 4q-2..4q+5): three in use, one in flight
  const int lq = min(lane, W / 4 - 1);
  const bool has_right = 4 * lq + 4 < W;
  auto load_e = [&](int h, auto slot) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot)::value;
    if (h >= 0 && h < H) {                                 // wave-uniform: interior rows take no zero fills
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        // lanes past the row (W < 256) read the last quad again: they never store
        const float* row = dy + (((long long)n * COUT + co) * H + h) * W + 4 * lq;
        const f32x4 m = *reinterpret_cast<const f32x4*>(row);
        const float2 l = *reinterpret_cast<const float2*>(row - (lq > 0 ? 2 : 0));
        const float2 h2 = *reinterpret_cast<const float2*>(row + (has_right ? 4 : 2));
        E[SL][co][0] = lq > 0 ? l.x : 0.f; E[SL][co][1] = lq > 0 ? l.y : 0.f;
        E[SL][co][2] = m.x; E[SL][co][3] = m.y; E[SL][co][4] = m.z; E[SL][co][5] = m.w;
        E[SL][co][6] = has_right ? h2.x : 0.f; E[SL][co][7] = has_right ? h2.y : 0.f;
      }
    } else {
#pragma unroll
      for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int e = 0; e < 8; ++e) E[SL][co][e] = 0.f;
    }
  };
  float Glo[CPW][2], Ghi[CPW][2], s1[CPW], s2[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) { Glo[j][0] = Glo[j][1] = Ghi[j][0] = Ghi[j][1] = 0.f; s1[j] = s2[j] = 0.f; }
  // U row ho of pair (r, r+1): kh = 0 <- dy row ho+1 (slot SP), kh = 1 <- ho (S0), kh = 2 <- ho-1 (SM); dy row ho+2, the
  // next U row's SP, is loaded into the fourth slot FIRST, so that it is in flight while this row's FMAs run
  auto urow = [&](auto sm_, auto s0_, auto sp_, auto sn_, int ho, int r) __attribute__((always_inline)) {
    constexpr int SM = decltype(sm_)::value, SZ = decltype(s0_)::value, SP = decltype(sp_)::value;
    load_e(ho + 2, sn_);
    float wa, wb;
    lerp_pair(ho, r, a.Hl, a.rh, wa, wb);
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      float dU[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        float v = 0.f;
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            v = fmaf(wv[j][co][kw], E[SP][co][k - kw + 2], v);
            v = fmaf(wv[j][co][3 + kw], E[SZ][co][k - kw + 2], v);
            v = fmaf(wv[j][co][6 + kw], E[SM][co][k - kw + 2], v);
          }
        dU[k] = v;
      }
      const float T0 = (cb[0] * dU[0] + cb[1] * dU[1]) + (ca[2] * dU[2] + ca[3] * dU[3]);   // low-res column 2q
      const float T1 = (cb[2] * dU[2] + cb[3] * dU[3]) + (ca[4] * dU[4] + ca[5] * dU[5]);   // low-res column 2q+1
      Glo[j][0] = fmaf(wa, T0, Glo[j][0]); Glo[j][1] = fmaf(wa, T1, Glo[j][1]);
      Ghi[j][0] = fmaf(wb, T0, Ghi[j][0]); Ghi[j][1] = fmaf(wb, T1, Ghi[j][1]);
    }
  };
  // slots (a, b, c | d) = dy rows (ho-1, ho, ho+1 | ho+2 in flight) of the pair's first U row; the second sees (b, c, d | a)
  auto pair = [&](auto a_, auto b_, auto c_, auto d_, int r) __attribute__((always_inline)) {
    const bool store = r >= r0 && r < r1 && active;        // low-res row r receives both of its pairs here
    const long long o = (long long)r * a.Wl + 2 * lane;
    float2 xv[CPW];                                        // the source values of the ReLU mask, loaded before the FMAs
#pragma unroll
    for (int j = 0; j < CPW; ++j) xv[j] = (store && gp[j]) ? *reinterpret_cast<const float2*>(xp[j] + o) : make_float2(0.f, 0.f);
    urow(a_, b_, c_, d_, 2 * r + 1, r);
    urow(b_, c_, d_, a_, 2 * r + 2, r);
    if (store) {
#pragma unroll
      for (int j = 0; j < CPW; ++j) {
        if (!gp[j]) continue;
        const float2 v = xv[j];
        float gx = fmaf(v.x, ch[j].scv, ch[j].shv) > 0.f ? Glo[j][0] : 0.f;
        float gy = fmaf(v.y, ch[j].scv, ch[j].shv) > 0.f ? Glo[j][1] : 0.f;
        s1[j] += gx + gy;
        s2[j] += gx * (v.x - mu[j]) * is[j] + gy * (v.y - mu[j]) * is[j];
        float2* dst = reinterpret_cast<float2*>(gp[j] + o);
        if (acc0 && ch[j].first) { const float2 old = *dst; gx += old.x; gy += old.y; }
        *dst = make_float2(gx, gy);
      }
    }
#pragma unroll
    for (int j = 0; j < CPW; ++j) { Glo[j][0] = Ghi[j][0]; Glo[j][1] = Ghi[j][1]; Ghi[j][0] = 0.f; Ghi[j][1] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
  };
  load_e(2 * r0 - 2, Slot0{});
  load_e(2 * r0 - 1, Slot1{});
  load_e(2 * r0, Slot2{});
  for (int r = r0 - 1; r < r1; r += 2) {                   // pairs (r, r+1) <- U rows 2r+1, 2r+2; two per trip (4 slots)
    pair(Slot0{}, Slot1{}, Slot2{}, Slot3{}, r);
    pair(Slot2{}, Slot3{}, Slot0{}, Slot1{}, r + 1);
  }
#pragma unroll
  for (int j = 0; j < CPW; ++j)
    if (stats[j]) {
      const double d1 = wave_sum_d((double)s1[j]), d2 = wave_sum_d((double)s2[j]);
      if (lane == 0) {
        atomicAdd(&bstats1[ch[j].cs], d1);
        atomicAdd(&bstats1[a.C1 + ch[j].cs], d2);
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------------------------------------------
bool head_applicable(const avsep_conv_desc* d) {
  return d->up2x && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && d->Cout <= HD_MAXCO &&
         d->act0 == AVSEP_ACT_RELU && (d->C0 == d->Cin || d->act1 == AVSEP_ACT_RELU) && d->W <= 256 &&
         (d->W & 3) == 0 && (d->H & 3) == 0 && d->N <= 65535 && d->H / 4 <= 65535;
}
static HeadArgs head_args(const avsep_conv_desc* d) {
  HeadArgs a{};
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.N = d->N; a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.Hl = d->H / 2; a.Wl = d->W / 2;
  a.rh = (float)(a.Hl - 1) / (float)(d->H - 1);
  a.rw = (float)(a.Wl - 1) / (float)(d->W - 1);
  return a;
}
// channels per wave of the weight / data gradient kernels (they share every dy row): 4 for Cout <= 2, 2 above (registers);
// AVSEP_HEAD_CPW (1 | 2 | 4, read once) overrides for tuning
static int head_cpw(const avsep_conv_desc* d, bool wgrad) {
  static const char* e = getenv("AVSEP_HEAD_CPW");
  static const int forced = e ? atoi(e) : 0;
  int cpw = forced >= 1 && forced <= 4 ? forced : (d->Cout <= 2 ? (wgrad ? 3 : 4) : 1)   /* measured at batch 64: wgrad 0.92 / 0.85 / 1.08 ms, dgrad 0.97 / 0.93 / 0.87 ms for 2 / 3 / 4 */;
  if (d->Cout > 2 && cpw > 2) cpw = 2;
  while (cpw > 1 && d->Cin < 4 * cpw) --cpw;              // few channels: keep all four waves of a block busy
  return cpw;
}
static int head_segments(int rows, int min_rows, int max_s) {   // power-of-two split with even segments
  int S = 1;
  while (S < max_s && rows % (4 * S) == 0 && rows / (2 * S) >= min_rows) S *= 2;
  return S;
}

int head_fwd(const avsep_conv_desc* d, const float* wp, int wp_ld, const float* bias, float* y, hipStream_t st) {
  HeadArgs a = head_args(d);
  const int R = d->Cout <= 2 ? 4 : 2;
  dim3 grid(cdiv(d->H, R), d->N);
  switch (d->Cout) {
    case 1: hipLaunchKernelGGL(head_fwd_kernel<1>, grid, dim3(256), 0, st, a, wp, wp_ld, bias, y); break;
    case 2: hipLaunchKernelGGL(head_fwd_kernel<2>, grid, dim3(256), 0, st, a, wp, wp_ld, bias, y); break;
    case 3: hipLaunchKernelGGL(head_fwd_kernel<3>, grid, dim3(256), 0, st, a, wp, wp_ld, bias, y); break;
    default: hipLaunchKernelGGL(head_fwd_kernel<4>, grid, dim3(256), 0, st, a, wp, wp_ld, bias, y); break;
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

size_t head_wgrad_workspace_floats(const avsep_conv_desc* d) {
  const int S = head_segments(d->H, 32, 8);
  return (size_t)d->N * S * d->Cout * d->Cin * 9 + (size_t)d->N * S * d->Cout;
}
int head_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias, float* ws, hipStream_t st) {
  HeadArgs a = head_args(d);
  const int S = head_segments(d->H, 32, 8);
  float* part = ws;
  float* bpart = ws + (size_t)d->N * S * d->Cout * d->Cin * 9;
  const int cpw = head_cpw(d, true);
  dim3 grid(cdiv(d->Cin, 4 * cpw), d->N, S);
#define HEAD_WG(CO_, CPW_) hipLaunchKernelGGL((head_wgrad_kernel<CO_, CPW_>), grid, dim3(256), 0, st, a, dy, part, dbias ? bpart : nullptr, S)
  switch (d->Cout) {
    case 1: if (cpw == 4) HEAD_WG(1, 4); else if (cpw == 3) HEAD_WG(1, 3); else if (cpw == 2) HEAD_WG(1, 2); else HEAD_WG(1, 1); break;
    case 2: if (cpw == 4) HEAD_WG(2, 4); else if (cpw == 3) HEAD_WG(2, 3); else if (cpw == 2) HEAD_WG(2, 2); else HEAD_WG(2, 1); break;
    case 3: if (cpw >= 2) HEAD_WG(3, 2); else HEAD_WG(3, 1); break;
    default: if (cpw >= 2) HEAD_WG(4, 2); else HEAD_WG(4, 1); break;
  }
#undef HEAD_WG
  AVSEP_LAUNCH_CHECK();
  const int nw = d->Cout * d->Cin * 9;
  int rc = reduce_slabs_strided(part, dw, nw, d->N * S, nw, st);       // 512 slabs at batch 64: four slab groups per element
  if (rc) return rc;
  if (dbias) return reduce_slabs_strided(bpart, dbias, d->Cout, d->N * S, d->Cout, st);
  return AVSEP_OK;
}

int head_dgrad(const avsep_conv_desc* d, const float* w, const float* dy, float* g0, float* g1, const float* mean1,
               const float* invstd1, double* bstats1, int acc0, hipStream_t st) {
  HeadArgs a = head_args(d);
  const int S = head_segments(a.Hl, 16, 4);
  const int cpw = head_cpw(d, false);
  dim3 grid(cdiv(d->Cin, 4 * cpw), d->N, S);
#define HEAD_DG(CO_, CPW_) hipLaunchKernelGGL((head_dgrad_kernel<CO_, CPW_>), grid, dim3(256), 0, st, a, w, dy, g0, g1, mean1, invstd1, bstats1, acc0, S)
  switch (d->Cout) {
    case 1: if (cpw == 4) HEAD_DG(1, 4); else if (cpw == 3) HEAD_DG(1, 3); else if (cpw == 2) HEAD_DG(1, 2); else HEAD_DG(1, 1); break;
    case 2: if (cpw == 4) HEAD_DG(2, 4); else if (cpw == 3) HEAD_DG(2, 3); else if (cpw == 2) HEAD_DG(2, 2); else HEAD_DG(2, 1); break;
    case 3: if (cpw >= 2) HEAD_DG(3, 2); else HEAD_DG(3, 1); break;
    default: if (cpw >= 2) HEAD_DG(4, 2); else HEAD_DG(4, 1); break;
  }
#undef HEAD_DG
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
