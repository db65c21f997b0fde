// LDS halo-patch convolution kernel on the f32 MFMA (shared by conv3x3.hip and conv_flat.hip; see conv3x3.hip's header).
#pragma once
#include "common.h"

struct C3Args {
  int N, Cin, H, W, Cout;
  int C0, C1, act0, act1, up2x, Hs, Ws;
  float rh, rw;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* wp;
  int wp_ld;
  float* out;
  const float* bias;
  double* stats;
  int tilesX, tilesY, gridM;
  // generalised geometry (3x3/s1/p1: Ho = H, Wo = W, pad 1, unit store stride):
  int Ho, Wo;            // output tile space
  int padh, padw;        // patch origin = tile origin * S - pad
  int os, ooh, oow;      // store position = (oh*os + ooh, ow*os + oow) in an [OHs x OWs] plane
  int OHs, OWs;
  // split-K (bf16 kernels on layers whose output grid cannot fill the chip): blockIdx.y reduces K-tiles
  // [y*kts, (y+1)*kts) into partial slab y of `out` (slab elements apart); 0 = no split
  int kts;
  long long slab;
  int planN;             // host only: batch the tile-size heuristics are planned for (avsep_conv_desc.plan_n; 0 = N)
  int algo;              // host only: avsep_conv_desc.algo
  int out16;             // bf16 kernels: `out` is a B16 image ([N][Cout/16][OHs][OWs][16] bf16) instead of fp32 NCHW
};
static inline long long c3_plan_n(const C3Args& a) { return a.planN > 0 ? a.planN : a.N; }
// 64-row instead of 128-row tiles (fp32 halo-patch kernels): small GEMM M, too few 128-row workgroups for the 256 CUs, or
// (3x3/s1 only, `quantise`) a workgroup count that quantises badly over them; wg128 is counted on the PLANNED batch
static inline bool c3_narrow_rule(int cout, long long wg128, bool quantise) {
  return cout <= 64 || wg128 < 384 || (quantise && wg128 < 1024 && wg128 * 5 < ((wg128 + 255) / 256) * 256 * 4);
}

constexpr int C3_CK = 4;            // input channels per K-tile of the 3x3 path (host-side packing constant)

__device__ __forceinline__ float c3_src(const C3Args& a, int n, int c, int hs, int ws) {
  float v;
  if (c < a.C0) {
    v = a.x0[(((long long)n * a.C0 + c) * a.Hs + hs) * a.Ws + ws];
    if (a.sc0) v = fmaf(v, a.sc0[c], a.sh0[c]);
    v = act_apply(v, a.act0);
  } else {
    int c1 = c - a.C0;
    v = a.x1[(((long long)n * a.C1 + c1) * a.Hs + hs) * a.Ws + ws];
    if (a.sc1) v = fmaf(v, a.sc1[c1], a.sh1[c1]);
    v = act_apply(v, a.act1);
  }
  return v;
}

// KS x KS taps, stride S (1 or 2), dilation DIL, CK input channels per K-tile.  With S == 2 the patch columns are
// stored de-interleaved (even columns, then odd columns) so that the 32 pixels of an MFMA column tile still read
// consecutive LDS words (a stride-2 read would be a 2-way bank conflict on ds_read_b32).
//
// FW > 0 selects the FLAT-PIXEL tile for small maps of width FW (ResNet: 56 / 28 / 14 / 7) with 'same' padding: the
// 128 MFMA columns are 128 CONSECUTIVE pixels of the flattened (n, h, w) index, so no column is wasted on a partial
// tile (8x16 tiles cover a 14x14 map at 77 %).  The images are thought of as stacked vertically with PADH shared zero
// rows between them (the bottom padding of image n is the top padding of image n+1); the patch holds every virtual
// row the 128 pixels touch plus the halo, full padded width.  Only the (pixel -> LDS base) and (patch element ->
// global address) maps differ from the rectangular tile, and both are computed once in the prologue.
template <int TH, int TW, int BM, bool UP2X, int KS = 3, int S = 1, int DIL = 1, int CK = 4, int KH_ = KS, int KW_ = KS,
          int FW = 0>
__global__ __launch_bounds__(256) void conv3x3_kernel(C3Args a) {
  constexpr bool FLAT = FW > 0;
  static_assert(!FLAT || (S == 1 && !UP2X && (KH_ & 1) && (KW_ & 1)), "flat tiles: stride 1, odd taps, no fused upsample");
  constexpr int NT = KH_ * KW_, C3_KT = CK * NT, C3_CK = CK;
  constexpr int PADH = DIL * (KH_ - 1) / 2, PADW = DIL * (KW_ - 1) / 2;     // flat mode: pad = dil * (k - 1) / 2
  constexpr int F_NR = (FW + 126 + (FLAT ? FW : 1)) / (FLAT ? FW : 1);      // rows touched by 128 consecutive pixels
  constexpr int F_NC = 127 / (FLAT ? FW * FW : 1) + 1;                      // image boundaries crossed (H >= FW on the host)
  constexpr int PH = FLAT ? F_NR + F_NC * PADH + 2 * PADH : (TH - 1) * S + (KH_ - 1) * DIL + 1;
  constexpr int PWR = FLAT ? FW + 2 * PADW : (TW - 1) * S + (KW_ - 1) * DIL + 1;
  constexpr int PW = (S == 2) ? (PWR + 1) / 2 * 2 : PWR, PWH = PW / 2, PS = PH * PW;   // patch per channel
  constexpr int LDA = BM + 4;
  constexpr int NPATCH = C3_CK * PS;
  constexpr int PE = (NPATCH + 255) / 256;                // patch elements per thread
  constexpr int A4 = BM / 4, NA4 = C3_KT * A4, AE = (NA4 + 255) / 256;
  constexpr int WTM = BM / 2, TM = WTM / 32;              // waves 2 (M) x 2 (N); wave N-tile = 64 pixels
  __shared__ __attribute__((aligned(16))) float As[2][C3_KT][LDA];
  __shared__ float Ps[2][NPATCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  // block -> (pixel tile, image, M tile); consecutive logical ids share the pixel tile (same XCD L2)
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = t % a.gridM; t /= a.gridM;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int n = t / a.tilesY;
  const int h0 = ty * TH, w0 = tx * TW, m0 = mt * BM;
  // flat mode: tile tx covers pixels [p0, p0 + 128) of the flattened (n, h, w) index; vr0 = its first virtual row
  const int f_HW = a.H * a.W, f_Hp = a.H + PADH, f_P = a.N * f_HW;
  const int p0 = tx * 128;
  const int vr0 = FLAT ? (p0 / f_HW) * f_Hp + (p0 % f_HW) / FW : 0;

  // ---- loader state -------------------------------------------------------------------------------
  // The f32 MFMA executes on the vector ALUs, so every VALU instruction inside the K loop is taken from
  // it (measured: this loop without staging runs at 140 TFLOP/s).  All index decoding therefore happens
  // ONCE here: each thread keeps a pointer per staged element that simply advances by one K-tile.
  //   issue():  unconditional loads (clamped addresses) into registers + pointer bumps;
  //   finish(): affine/activation (+ bilinear blend) + zero masking + LDS stores, after the MFMA loop.
  const bool has0 = a.sc0 != nullptr, has1 = a.sc1 != nullptr;
  const long long sHW = (long long)a.Hs * a.Ws;
  f32x4 areg[AE];   // native vector type: an array of float4 structs is not promoted out of scratch memory
  const float* aptr[AE];
  int a_lds[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    int idx = min(tid + 256 * e, NA4 - 1);
    int row = idx / A4, c4 = idx % A4;
    aptr[e] = a.wp + (long long)row * a.wp_ld + m0 + c4 * 4;
    a_lds[e] = row * LDA + c4 * 4;
  }
  const long long a_step = (long long)C3_KT * a.wp_ld;

  constexpr int NRAW = UP2X ? 4 : 1;
  float praw[PE][NRAW], psc[PE], psh[PE], plh[UP2X ? PE : 1], plw[UP2X ? PE : 1];
  const float* pptr[PE][NRAW];      // element source pointers for the current K-tile (source 0 first)
  long long poff1[PE][NRAW];        // offsets of the same elements inside source 1 (its channel 0 + cc)
  int p_cc[PE];
  int p_lds[S == 2 ? PE : 1];       // LDS slot of the element (identity for S == 1)
  unsigned pok = 0;
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    int idx = min(tid + 256 * e, NPATCH - 1);
    int cc = idx / PS, r = (idx % PS) / PW, col = idx % PW;
    int gh = h0 * S - a.padh + r, gw = w0 * S - a.padw + col;
    int ne = n;                              // image of this patch element
    if constexpr (FLAT) {                    // virtual row -> (image, row); rows H .. H+PADH-1 of a period are zero rows
      const int vr = vr0 - PADH + r;
      ne = vr >= 0 ? vr / f_Hp : a.N;
      gh = vr >= 0 ? vr % f_Hp : -1;
      gw = col - PADW;
      if (ne >= a.N) { ne = a.N - 1; gh = -1; }
    }
    if constexpr (S == 2) p_lds[e] = cc * PS + r * PW + (col & 1) * PWH + (col >> 1);
    bool ok = (PE * 256 == NPATCH || tid + 256 * e < NPATCH) && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
    int ghc = min(max(gh, 0), a.H - 1), gwc = min(max(gw, 0), a.W - 1);
    long long o[NRAW];
    if constexpr (!UP2X) {
      o[0] = (long long)ghc * a.Ws + gwc;
    } else {  // nn.Upsample(x2, bilinear, align_corners=True): src = dst*(in-1)/(out-1)
      float fh = a.rh * (float)ghc, fw = a.rw * (float)gwc;
      int hh0 = (int)fh, ww0 = (int)fw;
      int hh1 = hh0 + (hh0 < a.Hs - 1), ww1 = ww0 + (ww0 < a.Ws - 1);
      plh[e] = fh - (float)hh0;
      plw[e] = fw - (float)ww0;
      o[0] = (long long)hh0 * a.Ws + ww0; o[1] = (long long)hh0 * a.Ws + ww1;
      o[2] = (long long)hh1 * a.Ws + ww0; o[3] = (long long)hh1 * a.Ws + ww1;
    }
#pragma unroll
    for (int q = 0; q < NRAW; ++q) {
      pptr[e][q] = a.x0 + ((long long)ne * a.C0 + cc) * sHW + o[q];
      poff1[e][q] = ((long long)ne * a.C1 + cc) * sHW + o[q];
    }
    p_cc[e] = cc;
    pok |= (unsigned)ok << e;
  }
  const long long p_step = (long long)C3_CK * sHW;
  const int kt_switch = a.C0 / C3_CK;          // first K-tile that reads source 1 (C0 % C3_CK == 0)
  bool cur_has = has0;
  const float *scp = a.sc0, *shp = a.sh0;      // affine rows of the current source, advanced per K-tile

  auto issue = [&](int kt) __attribute__((always_inline)) {
    if (kt == kt_switch && a.C1 > 0) {         // block-uniform: switch every element pointer to source 1
#pragma unroll
      for (int e = 0; e < PE; ++e)
#pragma unroll
        for (int q = 0; q < NRAW; ++q) pptr[e][q] = a.x1 + poff1[e][q];
      cur_has = has1;
      scp = a.sc1;
      shp = a.sh1;
    }
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      areg[e] = *reinterpret_cast<const f32x4*>(aptr[e]);
      aptr[e] += a_step;
    }
#pragma unroll
    for (int e = 0; e < PE; ++e) {
#pragma unroll
      for (int q = 0; q < NRAW; ++q) {
        praw[e][q] = *pptr[e][q];
        pptr[e][q] += p_step;
      }
      if (cur_has) {
        psc[e] = scp[p_cc[e]];
        psh[e] = shp[p_cc[e]];
      }
    }
    if (cur_has) {
      scp += C3_CK;
      shp += C3_CK;
    }
  };
  // the affine/activation flags that belong to the tile held in registers (issue() may already have switched)
  bool fin_has = has0;
  float fin_slope = act_slope(a.act0);      // branch-free activation max(v, slope * v): a run-time act switch costs two
                                            // scalar branches per staged element, and the f32 MFMA shares the vector ALU
  auto finish = [&](int buf, int kt) __attribute__((always_inline)) {
    if (kt == kt_switch && a.C1 > 0) {
      fin_has = has1;
      fin_slope = act_slope(a.act1);
    }
    float* Ab = &As[buf][0][0];
#pragma unroll
    for (int e = 0; e < AE; ++e)
      if (AE * 256 == NA4 || tid + 256 * e < NA4) *reinterpret_cast<f32x4*>(Ab + a_lds[e]) = areg[e];
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      float v;
      if constexpr (!UP2X) {
        v = praw[e][0];
        if (fin_has) v = fmaf(v, psc[e], psh[e]);
        v = act_by_slope(v, fin_slope);
      } else {
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = praw[e][k];
          if (fin_has) t = fmaf(t, psc[e], psh[e]);
          q[k] = act_by_slope(t, fin_slope);
        }
        v = (1.f - plh[e]) * ((1.f - plw[e]) * q[0] + plw[e] * q[1]) + plh[e] * ((1.f - plw[e]) * q[2] + plw[e] * q[3]);
      }
      if (PE * 256 == NPATCH || tid + 256 * e < NPATCH) {
        if constexpr (S == 2) Ps[buf][p_lds[e]] = ((pok >> e) & 1u) ? v : 0.f;
        else Ps[buf][tid + 256 * e] = ((pok >> e) & 1u) ? v : 0.f;
      }
    }
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // B operand lane base for the wave's two 32-pixel MFMA column tiles (+ channel parity from the lane half)
  int lb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int p = wn * 64 + j * 32 + li;
    if constexpr (FLAT) {
      const int pg = min(p0 + p, f_P - 1);                   // columns past the last pixel compute garbage, never stored
      lb[j] = ((pg / f_HW) * f_Hp + (pg % f_HW) / FW - vr0) * PW + pg % FW + lk * PS;
    } else {
      lb[j] = (p / TW) * S * PW + (p % TW) + lk * PS;      // S == 2: the column index is halved by the de-interleave
    }
  }
  const int nK = a.Cin / C3_CK;
  issue(0);
  finish(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nK; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nK) issue(kt + 1);
    const float* P = Ps[buf];
    // operands of k-step k2+1 are read into a second register set before the MFMAs of k-step k2 issue,
    // so the LDS latency hides behind 4 x 64 MFMA cycles instead of stalling the wave every step
    float av[2][TM], bv[2][2];
    auto read_ops = [&](int k2, int slot) __attribute__((always_inline)) {
      const int cp = k2 / NT, tap = k2 % NT, kh = tap / KW_, kw = tap % KW_;   // compile-time after unrolling
      const int koff = (2 * cp) * PS + kh * DIL * PW + (S == 2 ? (kw & 1) * PWH + (kw >> 1) : kw * DIL);
#pragma unroll
      for (int i = 0; i < TM; ++i) av[slot][i] = As[buf][2 * k2 + lk][wm * WTM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[slot][j] = P[lb[j] + koff];
    };
    read_ops(0, 0);
#pragma unroll
    for (int k2 = 0; k2 < C3_KT / 2; ++k2) {
      const int cur = k2 & 1;
      if (k2 + 1 < C3_KT / 2) read_ops(k2 + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nK) finish(buf ^ 1, kt + 1);
    __syncthreads();
  }

  // ---- epilogue (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ----
  const long long HW = (long long)a.OHs * a.OWs;
  long long cbase[2];
  bool cok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int p = wn * 64 + j * 32 + li;
    if constexpr (FLAT) {
      const int pg = p0 + p;
      cok[j] = pg < f_P;
      cbase[j] = (long long)(pg / f_HW) * a.Cout * HW + pg % f_HW;
    } else {
      int gh = h0 + p / TW, gw = w0 + p % TW;
      cok[j] = gh < a.Ho && gw < a.Wo;
      cbase[j] = (long long)n * a.Cout * HW + (long long)(gh * a.os + a.ooh) * a.OWs + (gw * a.os + a.oow);
    }
  }
  const bool want_stats = a.stats != nullptr;
  float* s_sum = &As[0][0][0];   // [2][BM] per-wave-column partial sums (operand tiles are dead now)
  float* s_sq = &As[1][0][0];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lrow = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      int row = m0 + lrow;
      bool rok = row < a.Cout;
      float bias = (a.bias && rok) ? a.bias[row] : 0.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v = acc[i][j][r] + bias;
        if (rok && cok[j]) {
          a.out[cbase[j] + (long long)row * HW] = v;
          s += v;
          q += v * v;
        }
      }
      if (want_stats) {
        s = half_sum_hi(s);
        q = half_sum_hi(q);
        if (li == 31) {
          s_sum[wn * BM + lrow] = s;
          s_sq[wn * BM + lrow] = q;
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int rr = tid; rr < BM; rr += 256) {
      int row = m0 + rr;
      if (row < a.Cout) {
        atomicAdd(&a.stats[row], (double)(s_sum[rr] + s_sum[BM + rr]));
        atomicAdd(&a.stats[a.Cout + row], (double)(s_sq[rr] + s_sq[BM + rr]));
      }
    }
  }
}

