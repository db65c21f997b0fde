// Implicit-GEMM convolution (forward / data-gradient / weight-gradient) on the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32) for gfx950.  NCHW fp32.  See include/avsep.h.
//
// GEMM view (D[M][N], N on the lanes so global stores are coalesced):
//   fwd   : D[co][pix]   = sum_k  Wp[k][co]      * X(k, pix)        k = (ci,kh,kw)
//   dgrad : D[ci][pixin] = sum_k' Wd[k'][ci]     * dY(k', pixin)    k' = (tap,co), one launch
//           slice (blockIdx.y) per input-pixel parity class so strided convs waste no MFMA work
//   wgrad : D[co][k]     = sum_pix dY[co][pix]   * X(k, pix)        split over pixel chunks
// X(k,pix) is the *virtual* conv input: two-source channel concat, per-channel affine (folded
// BatchNorm), activation and optional bilinear x2 upsample are applied while gathering.
// Both operands are staged global -> registers -> LDS (K-major, double buffered, one barrier
// per K-tile); each wave owns a (BM/WM)x(BN/WN) block of 32x32 MFMA tiles.
#include "common.h"
#include <stdio.h>
#include <string.h>

enum { M_FWD = 0, M_DGRAD = 1, M_WGRAD = 2 };

struct CArgs {
  int N, Cin, H, W, Cout, Ho, Wo, KH, KW, stride, pad, dil;
  int C0, C1, act0, act1, up2x;
  int Hs, Ws;  // source spatial size (H/2, W/2 when up2x)
  float rh, rw;
  const float *x0, *x1, *sc0, *sh0, *sc1, *sh1;
  const float* wp;
  int wp_ld;
  const float* dy;
  float* out;
  const float* bias;
  double* stats;
  int M, Ncols, K;
  int gridM;
  int chunk, P;  // wgrad: pixels per split, total pixels
  int kts;       // fwd/dgrad split-K: K-tiles per split (0 = no split)
  long long slab;  // fwd/dgrad split-K: elements of one partial output slab
  int lstride;   // log2(stride) (dgrad)
};

__device__ __forceinline__ float load_src(const CArgs& a, long long nb0, long long nb1, int ci, int off) {
  float v;
  if (ci < a.C0) {
    v = a.x0[nb0 + (long long)ci * (a.Hs * a.Ws) + off];
    if (a.sc0) v = fmaf(v, a.sc0[ci], a.sh0[ci]);
    v = act_apply(v, a.act0);
  } else {
    int c1 = ci - a.C0;
    v = a.x1[nb1 + (long long)c1 * (a.Hs * a.Ws) + off];
    if (a.sc1) v = fmaf(v, a.sc1[c1], a.sh1[c1]);
    v = act_apply(v, a.act1);
  }
  return v;
}

// value of the virtual input at (ci, hi, wi), 0<=hi<H, 0<=wi<W
__device__ __forceinline__ float load_virtual(const CArgs& a, long long nb0, long long nb1, int ci, int hi,
                                              int wi) {
  if (!a.up2x) return load_src(a, nb0, nb1, ci, hi * a.Ws + wi);
  // nn.Upsample(scale_factor=2, bilinear, align_corners=True): src = dst*(in-1)/(out-1)
  float fh = a.rh * (float)hi, fw = a.rw * (float)wi;
  int h0 = (int)fh, w0 = (int)fw;
  int h1 = h0 + (h0 < a.Hs - 1), w1 = w0 + (w0 < a.Ws - 1);
  float lh = fh - (float)h0, lw = fw - (float)w0;
  float v00 = load_src(a, nb0, nb1, ci, h0 * a.Ws + w0), v01 = load_src(a, nb0, nb1, ci, h0 * a.Ws + w1);
  float v10 = load_src(a, nb0, nb1, ci, h1 * a.Ws + w0), v11 = load_src(a, nb0, nb1, ci, h1 * a.Ws + w1);
  return (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
}

// UP2X: the fused bilinear x2 gather is compiled in only where it is used (as a run-time flag it put four corner loads,
// their interpolation and a scalar branch per element into every instantiation's K loop).
template <int MODE, int BM, int BN, int BK, int WM, int WN, bool UP2X = false>
__global__ __launch_bounds__(256) void igemm_kernel(CArgs a) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + ((MODE == M_WGRAD) ? 1 : 4);
  constexpr int LDB = BN + ((MODE == M_WGRAD) ? 1 : 0);
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ float Bs[2][BK][LDB];
  __shared__ int s_tab[(MODE == M_WGRAD) ? BN : 24];
  __shared__ float s_sc[(MODE == M_WGRAD) ? BN : 1], s_sh[(MODE == M_WGRAD) ? BN : 1];
  __shared__ int s_code[(MODE == M_WGRAD) ? BN : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int HoWo = a.Ho * a.Wo, KHW = a.KH * a.KW, HW = a.H * a.W;
  const long long srcHW = (long long)a.Hs * a.Ws;

  int M = a.M, Ncols = a.Ncols, K = a.K;
  int ph = 0, pw = 0, Hc = 0, Wc = 0, ntw = 1;
  if (MODE == M_DGRAD) {
    const int s = a.stride;
    ph = blockIdx.y / s;
    pw = blockIdx.y % s;
    Hc = (a.H > ph) ? (a.H - ph + s - 1) / s : 0;
    Wc = (a.W > pw) ? (a.W - pw + s - 1) / s : 0;
    if (tid == 0) {
      int nh = 0, nw = 0;
      for (int kh = 0; kh < a.KH; ++kh) {
        int t = ph + a.pad - kh * a.dil;
        if (((t % s) + s) % s == 0) s_tab[2 + nh++] = kh;
      }
      for (int kw = 0; kw < a.KW; ++kw) {
        int t = pw + a.pad - kw * a.dil;
        if (((t % s) + s) % s == 0) s_tab[12 + nw++] = kw;
      }
      s_tab[0] = nh;
      s_tab[1] = nw;
    }
    __syncthreads();
    ntw = s_tab[1];
    Ncols = a.N * Hc * Wc;
    K = s_tab[0] * ntw * a.Cout;
  }
  const int t_lin = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = (t_lin / a.gridM) * BN, m0 = (t_lin % a.gridM) * BM;
  if (MODE == M_DGRAD && n0 >= Ncols) return;  // block-uniform

  int p_begin = 0, p_end = 0, nK;
  if (MODE == M_WGRAD) {
    p_begin = blockIdx.y * a.chunk;
    p_end = min(a.P, p_begin + a.chunk);
    nK = (p_end - p_begin + BK - 1) / BK;
    // per-column (k) decode table: ci | dh<<16 | dw<<24
    for (int c = tid; c < BN; c += 256) {
      int k = n0 + c, v = -1, code = 0;
      float sc = 1.f, sh = 0.f;
      if (k < Ncols) {
        int ci = k / KHW, r = k % KHW;
        v = ci | ((r / a.KW) * a.dil << 16) | ((r % a.KW) * a.dil << 24);
        bool first = ci < a.C0;
        int cs = first ? ci : ci - a.C0;
        const float* sp = first ? a.sc0 : a.sc1;
        const float* hp = first ? a.sh0 : a.sh1;
        if (sp) { sc = sp[cs]; sh = hp[cs]; code = 1; }
        code |= (first ? a.act0 : a.act1) << 1;
      }
      s_tab[c] = v;
      s_sc[c] = sc;
      s_sh[c] = sh;
      s_code[c] = code;
    }
    __syncthreads();
  } else {
    nK = (K + BK - 1) / BK;
  }
  // split-K (fwd: blockIdx.y, dgrad: blockIdx.z): this block reduces K-tiles [kt0, nK) of its slice
  int kt0 = 0, ksplit = 0;
  if (MODE != M_WGRAD && a.kts > 0) {
    ksplit = (MODE == M_FWD) ? blockIdx.y : blockIdx.z;
    kt0 = ksplit * a.kts;
    nK = min(nK, kt0 + a.kts);
  }

  // ---------------- per-thread loader state ----------------
  // fwd/dgrad: thread owns one column (bcol) and BROWS consecutive k rows.
  // Loads are split in two phases so that their latency hides behind the MFMA loop:
  //   issue(kt+1): address math + UNCONDITIONAL global loads (clamped addresses) into raw registers
  //   finish():    affine + activation + zero-masking, then the LDS stores, after the MFMA loop.
  constexpr int BGROUPS = 256 / BN, BROWS = (MODE == M_WGRAD) ? 1 : BK / BGROUPS;
  constexpr int A4 = BM / 4, AV = (MODE == M_WGRAD) ? 1 : (BK * A4) / 256;
  constexpr int PG = 256 / BK, AE = BM / PG, BE = BN / PG;  // wgrad
  constexpr int NBR = (MODE == M_WGRAD) ? BE : BROWS;
  constexpr int NSC = (MODE == M_FWD) ? BROWS : 1;
  // BN >= 64, so all lanes of a wave own the same k rows: keep everything k-dependent on the scalar unit
  const int bcol = tid % BN, brow0 = __builtin_amdgcn_readfirstlane((tid / BN) * BROWS);
  bool cvalid = false;
  long long nb0 = 0, nb1 = 0, nbdy = 0;
  int hi0 = 0, wi0 = 0;
  if (MODE == M_FWD) {
    int j = n0 + bcol;
    cvalid = j < Ncols;
    int jj = cvalid ? j : 0;
    int n = jj / HoWo, hw = jj % HoWo;
    hi0 = (hw / a.Wo) * a.stride - a.pad;
    wi0 = (hw % a.Wo) * a.stride - a.pad;
    nb0 = (long long)n * a.C0 * srcHW;
    nb1 = (long long)n * a.C1 * srcHW;
  } else if (MODE == M_DGRAD) {
    int q = n0 + bcol;
    cvalid = q < Ncols;
    int qq = cvalid ? q : 0;
    int n = qq / (Hc * Wc), r = qq % (Hc * Wc);
    hi0 = (r / Wc) * a.stride + ph + a.pad;  // hi + pad
    wi0 = (r % Wc) * a.stride + pw + a.pad;
    nbdy = (long long)n * a.Cout * HoWo;
  }
  const bool has0 = a.sc0 != nullptr, has1 = a.sc1 != nullptr;
  const float slope0 = act_slope(a.act0), slope1 = act_slope(a.act1);   // branch-free activation (common.h)
  const int pixoff = hi0 * a.Ws + wi0;   // fwd: offset of the (possibly out-of-range) window origin

  float braw[NBR], bsc[NSC], bsh[NSC];
  f32x4 areg4[AV];   // native vector type (a float4 struct array would live in scratch memory)
  float areg[(MODE == M_WGRAD) ? AE : 1];
  unsigned bmask = 0, bfirst = 0, amask = 0;

  auto issue = [&](int kt) __attribute__((always_inline)) {
    if (MODE == M_FWD) {
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        int idx = tid + 256 * e, row = idx / A4, c4 = idx % A4;
        areg4[e] = *reinterpret_cast<const f32x4*>(a.wp + (long long)(kt * BK + row) * a.wp_ld + m0 + c4 * 4);
      }
      const int k = kt * BK + brow0;                       // wave-uniform
      int ci = k / KHW, r = k % KHW, kh = r / a.KW, kw = r % a.KW;
      bmask = 0;
      bfirst = 0;
#pragma unroll
      for (int e = 0; e < BROWS; ++e) {
        const int dh = kh * a.dil, dw = kw * a.dil;          // uniform
        const int hi = hi0 + dh, wi = wi0 + dw;
        const bool ok = cvalid && (k + e < K) && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        const int cc = min(ci, a.Cin - 1);
        if constexpr (UP2X) {  // the bilinear gather finishes here (raw value is final)
          braw[e] = ok ? load_virtual(a, nb0, nb1, cc, hi, wi) : 0.f;
          bsc[e] = 1.f;
          bsh[e] = 0.f;
        } else {
          const bool first = cc < a.C0;                      // uniform
          const int c = first ? cc : cc - a.C0;
          const float* xc = (first ? a.x0 : a.x1) + (long long)c * srcHW + (dh * a.Ws + dw);   // uniform pointer
          const long long lane_off = (first ? nb0 : nb1) + (ok ? pixoff : -(dh * a.Ws + dw));
          braw[e] = xc[lane_off];
          const bool has = first ? has0 : has1;
          bsc[e] = has ? (first ? a.sc0 : a.sc1)[c] : 1.f;   // uniform -> scalar loads
          bsh[e] = has ? (first ? a.sh0 : a.sh1)[c] : 0.f;
          bfirst |= (unsigned)first << e;
        }
        bmask |= (unsigned)ok << e;
        if (++kw == a.KW) {
          kw = 0;
          if (++kh == a.KH) { kh = 0; ++ci; }
        }
      }
    } else if (MODE == M_DGRAD) {
      amask = 0;
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        int idx = tid + 256 * e, row = idx / A4, c4 = idx % A4;
        int k = kt * BK + row;
        bool ok = k < K;
        int kk = ok ? k : 0;
        int co = kk % a.Cout, t = kk / a.Cout;
        int kh = s_tab[2 + t / ntw], kw = s_tab[12 + t % ntw];
        areg4[e] = *reinterpret_cast<const f32x4*>(a.wp + (long long)((kh * a.KW + kw) * a.Cout + co) * a.wp_ld + m0 +
                                                   c4 * 4);
        amask |= (unsigned)ok << e;
      }
      const int k = kt * BK + brow0;                       // wave-uniform
      int co = k % a.Cout, t = k / a.Cout;
      bmask = 0;
#pragma unroll
      for (int e = 0; e < BROWS; ++e) {
        const bool kok = k + e < K;
        const int tt = kok ? t : 0;
        const int kh = __builtin_amdgcn_readfirstlane(s_tab[2 + tt / ntw]);
        const int kw = __builtin_amdgcn_readfirstlane(s_tab[12 + tt % ntw]);
        const int th = hi0 - kh * a.dil, tw = wi0 - kw * a.dil;
        const int ho = th >> a.lstride, wo = tw >> a.lstride;   // exact: the parity class makes th,tw multiples of stride
        const bool ok = cvalid && kok && th >= 0 && tw >= 0 && ho < a.Ho && wo < a.Wo;
        const float* yc = a.dy + (long long)co * HoWo;       // uniform pointer
        braw[e] = yc[nbdy + (ok ? ho * a.Wo + wo : 0)];
        bmask |= (unsigned)ok << e;
        if (++co == a.Cout) { co = 0; ++t; }
      }
    } else {  // WGRAD: thread owns one pixel row (p_local) and AE/BE strided columns
      int p = p_begin + kt * BK + (tid % BK);
      bool pv = p < p_end;
      int pp = pv ? p : 0;
      int n = pp / HoWo, hw = pp % HoWo;
      int hb = (hw / a.Wo) * a.stride - a.pad, wb = (hw % a.Wo) * a.stride - a.pad;
      long long b0 = (long long)n * a.C0 * srcHW, b1 = (long long)n * a.C1 * srcHW;
      long long bdy = (long long)n * a.Cout * HoWo + hw;
      const int grp = tid / BK;
      amask = pv ? 0xffffffffu : 0u;
#pragma unroll
      for (int e = 0; e < AE; ++e) {
        int co = min(m0 + grp + PG * e, M - 1);   // rows >= M are never stored by the epilogue
        areg[e] = a.dy[bdy + (long long)co * HoWo];
      }
      bmask = 0;
#pragma unroll
      for (int e = 0; e < BE; ++e) {
        int tab = s_tab[grp + PG * e];
        int tt = tab >= 0 ? tab : 0;
        int ci = tt & 0xffff, hi = hb + ((tt >> 16) & 0xff), wi = wb + ((tt >> 24) & 0xff);
        bool ok = pv && tab >= 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
        if constexpr (UP2X) {
          braw[e] = ok ? load_virtual(a, b0, b1, ci, hi, wi) : 0.f;
        } else {
          bool first = ci < a.C0;
          int c = first ? ci : ci - a.C0;
          const float* xb = first ? a.x0 + b0 : a.x1 + b1;
          int off = ok ? hi * a.Ws + wi : 0;
          braw[e] = xb[(long long)c * srcHW + off];
        }
        bmask |= (unsigned)ok << e;
      }
    }
  };

  auto finish = [&](int buf) __attribute__((always_inline)) {
    if (MODE == M_FWD) {
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        int idx = tid + 256 * e, row = idx / A4, c4 = idx % A4;
        *reinterpret_cast<f32x4*>(&As[buf][row][c4 * 4]) = areg4[e];
      }
#pragma unroll
      for (int e = 0; e < BROWS; ++e) {
        float v = braw[e];
        if constexpr (!UP2X) {
          bool first = (bfirst >> e) & 1u;
          if (first ? has0 : has1) v = fmaf(v, bsc[e], bsh[e]);
          v = act_by_slope(v, first ? slope0 : slope1);
        }
        Bs[buf][brow0 + e][bcol] = ((bmask >> e) & 1u) ? v : 0.f;
      }
    } else if (MODE == M_DGRAD) {
#pragma unroll
      for (int e = 0; e < AV; ++e) {
        int idx = tid + 256 * e, row = idx / A4, c4 = idx % A4;
        f32x4 v = areg4[e];
        if (!((amask >> e) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&As[buf][row][c4 * 4]) = v;
      }
#pragma unroll
      for (int e = 0; e < BROWS; ++e) Bs[buf][brow0 + e][bcol] = ((bmask >> e) & 1u) ? braw[e] : 0.f;
    } else {
      const int pl = tid % BK, grp = tid / BK;
#pragma unroll
      for (int e = 0; e < AE; ++e) As[buf][pl][grp + PG * e] = amask ? areg[e] : 0.f;
#pragma unroll
      for (int e = 0; e < BE; ++e) {
        float v = braw[e];
        if constexpr (!UP2X) {
          int col = grp + PG * e;
          int code = s_code[col];   // bit0: affine present, bits1-2: activation
          if (code & 1) v = fmaf(v, s_sc[col], s_sh[col]);
          v = act_by_slope(v, act_slope(code >> 1));
        }
        Bs[buf][pl][grp + PG * e] = ((bmask >> e) & 1u) ? v : 0.f;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nK > kt0) {
    issue(kt0);
    finish(kt0 & 1);
  }
  __syncthreads();
  const int li = lane & 31, lk = lane >> 5;
  for (int kt = kt0; kt < nK; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nK) issue(kt + 1);
    // operands of k-step k2+1 are read before the MFMAs of k-step k2 issue (order pinned with sched_barrier):
    // the LDS latency hides behind TM*TN*64 MFMA cycles instead of stalling the wave every step
    float av[2][TM], bv[2][TN];
    auto read_ops = [&](int k2, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[slot][i] = As[buf][2 * k2 + lk][wm * WTM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[slot][j] = Bs[buf][2 * k2 + lk][wn * WTN + j * 32 + li];
    };
    read_ops(0, 0);
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; ++k2) {
      const int cur = k2 & 1;
      if (k2 + 1 < BK / 2) read_ops(k2 + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nK) finish(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  long long cbase[TN];
  bool cok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int col = n0 + wn * WTN + j * 32 + li;
    cok[j] = col < Ncols;
    int cc = cok[j] ? col : 0;
    if (MODE == M_FWD) {
      int n = cc / HoWo, hw = cc % HoWo;
      cbase[j] = (long long)ksplit * a.slab + (long long)n * a.Cout * HoWo + hw;
    } else if (MODE == M_DGRAD) {
      int n = cc / (Hc * Wc), r = cc % (Hc * Wc);
      cbase[j] = (long long)ksplit * a.slab + (long long)n * a.Cin * HW + ((r / Wc) * a.stride + ph) * a.W +
                 (r % Wc) * a.stride + pw;
    } else {
      cbase[j] = (long long)blockIdx.y * M * Ncols + cc;
    }
  }
  const long long rstride = (MODE == M_FWD) ? HoWo : (MODE == M_DGRAD ? HW : Ncols);
  const bool want_stats = (MODE == M_FWD) && a.stats != nullptr && a.kts == 0;
  // per-row partial sums of this wave go to LDS (the operand tiles are dead after the last barrier),
  // so each workgroup issues ONE pair of double atomics per output channel
  float* s_sum = &As[0][0][0];   // [WN][BM]
  float* s_sq = &Bs[0][0][0];    // [WN][BM]
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lrow = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      int row = m0 + lrow;
      bool rok = row < M;
      float bias = (MODE == M_FWD && a.bias && rok && a.kts == 0) ? a.bias[row] : 0.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v = acc[i][j][r] + bias;
        if (rok && cok[j]) {
          a.out[cbase[j] + (long long)row * rstride] = v;
          s += v;
          q += v * v;
        }
      }
      if (want_stats) {
        s = half_sum(s);
        q = half_sum(q);
        if (li == 0) {
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
      if (row < M) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) { s += s_sum[w * BM + rr]; q += s_sq[w * BM + rr]; }
        atomicAdd(&a.stats[row], (double)s);
        atomicAdd(&a.stats[M + row], (double)q);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// small helper kernels
// ---------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH,
                                    int KW, int rows, int ld, int mode) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)rows * ld;
  if (i >= total) return;
  int row = (int)(i / ld), col = (int)(i % ld);
  int KHW = KH * KW;
  float v = 0.f;
  if (mode == 0) {  // row = k = (ci,kh,kw), col = co
    if (row < Cin * KHW && col < Cout) v = w[(long long)col * Cin * KHW + row];
  } else {  // row = tap*Cout + co, col = ci
    if (row < KHW * Cout && col < Cin) {
      int tap = row / Cout, co = row % Cout;
      v = w[((long long)co * Cin + col) * KHW + tap];
    }
  }
  out[i] = v;
}

// out[i] = sum over S slabs (`stride` elements apart) of ws[z][i], i < n.  Block = 64 consecutive elements x 4 slab groups:
// thread (e, q) sums slabs z = q, q + 4, ... with four independent loads in flight, the four partials are added in a fixed
// order through LDS (deterministic).  The one-thread-per-element form walked its S slabs as one dependent chain: 0.15 TB/s
// on a [256 x 1152] gradient with 32 slabs.
// E elements x G = 256 / E slab groups per block: 64 x 4 for large gradients; 16 x 16 when n is small and S large (the decoder
// head's [2 x 128 x 9] gradient has 512 slabs: 36 blocks of the 64-element form left 220 CUs idle, 106 us).
template <int E>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ out, long long n,
                                                           int S, long long stride) {
  constexpr int G = 256 / E;
  __shared__ float part[G][E];
  const int e = threadIdx.x % E, q = threadIdx.x / E;
  const long long i = (long long)blockIdx.x * E + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int z = q;
    for (; z + 3 * G < S; z += 4 * G) {
      s0 += ws[(long long)z * stride + i];
      s1 += ws[(long long)(z + G) * stride + i];
      s2 += ws[(long long)(z + 2 * G) * stride + i];
      s3 += ws[(long long)(z + 3 * G) * stride + i];
    }
    for (; z < S; z += G) s0 += ws[(long long)z * stride + i];
  }
  part[q][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && i < n) {
    float v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = part[g][e];
#pragma unroll
    for (int w = G / 2; w > 0; w >>= 1)        // fixed pairwise order: deterministic
#pragma unroll
      for (int g = 0; g < w; ++g) v[g] = v[2 * g] + v[2 * g + 1];
    out[i] = v[0];
  }
}
static int launch_reduce_slabs(const float* ws, float* out, long long n, int S, long long stride, hipStream_t st) {
  if (n <= 16384 && S >= 64) hipLaunchKernelGGL(reduce_slabs_kernel<16>, dim3(cdiv(n, 16)), dim3(256), 0, st, ws, out, n, S, stride);
  else hipLaunchKernelGGL(reduce_slabs_kernel<64>, dim3(cdiv(n, 64)), dim3(256), 0, st, ws, out, n, S, stride);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// split-K forward combine: y = sum of slabs (+ bias); per-channel (sum, sumsq) for the following BatchNorm.
// grid (C, chunks); a block owns one channel over a slice of (n, hw)
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ ws, long long slab, int S, int N, int C,
                                                             int HW, const float* __restrict__ bias,
                                                             float* __restrict__ y, double* stats) {
  const int c = blockIdx.x;
  const float b = bias ? bias[c] : 0.f;
  const int n_per = (N + gridDim.y - 1) / gridDim.y, n_beg = blockIdx.y * n_per, n_end = min(N, n_beg + n_per);
  float s1 = 0.f, s2 = 0.f;
  // the block's (image, pixel) pairs as ONE index range: the maps that need split-K are 2x2 ... 8x8, and a loop over the
  // pixels of one image left 75-98 % of the threads idle; two slab chains in flight per element
  const int total = (n_end - n_beg) * HW;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int n = n_beg + e / HW, i = e - (e / HW) * HW;
    const long long base = ((long long)n * C + c) * HW + i;
    float v0 = b, v1 = 0.f;
    int z = 0;
    for (; z + 1 < S; z += 2) {
      v0 += ws[(long long)z * slab + base];
      v1 += ws[(long long)(z + 1) * slab + base];
    }
    if (z < S) v0 += ws[(long long)z * slab + base];
    const float v = v0 + v1;
    y[base] = v;
    s1 += v;
    s2 += v * v;
  }
  if (stats) {
    double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
    __shared__ double sh[8];
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = d1; sh[4 + (threadIdx.x >> 6)] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&stats[c], sh[0] + sh[1] + sh[2] + sh[3]);
      atomicAdd(&stats[C + c], sh[4] + sh[5] + sh[6] + sh[7]);
    }
  }
}

// dbias[c] = sum over n,hw of dy[n,c,hw]; one block per channel
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, int N, int C, int HW,
                                                          float* __restrict__ out) {
  int c = blockIdx.x;
  double s = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = x + ((long long)n * C + c) * HW;
    float t = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) t += p[i];
    s += (double)t;
  }
  __shared__ double sh[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// direct.hip: VALU kernels for convolutions with <= 4 output channels (3x3, stride 1)
bool smallco_applicable(const avsep_conv_desc* d);
int smallco_fwd(const avsep_conv_desc* d, const float* wp, int wp_ld, const float* bias, float* y, hipStream_t st);
size_t smallco_wgrad_workspace_floats(const avsep_conv_desc* d);
int smallco_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias, float* ws, hipStream_t st);
bool head_applicable(const avsep_conv_desc* d);
int head_fwd(const avsep_conv_desc* d, const float* wp, int wp_ld, const float* bias, float* y, float* ws, hipStream_t st);
size_t head_fwd_workspace_floats(const avsep_conv_desc* d);
size_t head_dgrad_workspace_floats(const avsep_conv_desc* d);
size_t head_wgrad_workspace_floats(const avsep_conv_desc* d);
int head_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias, float* ws, hipStream_t st);
int head_dgrad(const avsep_conv_desc* d, const float* w, const float* dy, float* g0, float* g1, const float* mean1,
               const float* invstd1, double* bstats1, int acc0, float* ws, hipStream_t st);
bool smallci_applicable(const avsep_conv_desc* d);
int smallci_dgrad(const avsep_conv_desc* d, const float* w_oihw, const float* dy, float* dx, hipStream_t st);
// conv_wino.hip: Winograd F(2x2, 3x3) form of the 3x3 / stride 1 / 'same' convs (forward and dgrad), fp32
// conv_wino4.hip: Winograd F(4x4, 3x3) for the maps that tile by 4 (asked before F(2x2, 3x3))
bool w4_applicable(const avsep_conv_desc* d, int mode);
size_t w4_packed_floats(const avsep_conv_desc* d, int mode);
int w4_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st);
int w4_fwd(const avsep_conv_desc* d, const float* up, const float* bias, float* y, double* stats, hipStream_t st);
int w4_dgrad(const avsep_conv_desc* d, const float* up, const float* dy, float* dx, const avsep_act_bwd* e, hipStream_t st);
void w4_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap);
bool wn_applicable(const avsep_conv_desc* d, int mode);
size_t wn_packed_floats(const avsep_conv_desc* d, int mode);
int wn_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st);
int wn_fwd(const avsep_conv_desc* d, const float* up, const float* bias, float* y, double* stats, hipStream_t st);
int wn_dgrad(const avsep_conv_desc* d, const float* up, const float* dy, float* dx, hipStream_t st);
// conv3x3.hip: LDS-halo-patch kernel for 3x3 / stride 1 / pad 1 (forward, and dgrad through flipped weights)
bool c3_applicable(const avsep_conv_desc* d, int mode);
size_t c3_packed_floats(const avsep_conv_desc* d, int mode);
int c3_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st);
int c3_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, hipStream_t st);
int c3_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, hipStream_t st);
bool w3_applicable(const avsep_conv_desc* d);
// wgrad_wino.hip: Winograd form of the 3x3 / stride 1 weight gradient, fp32
// wgrad_wino4.hip: Winograd F(4x4, 3x3) weight gradient (asked before the F(2x2) form)
bool x4_applicable(const avsep_conv_desc* d);
size_t x4_workspace_floats(const avsep_conv_desc* d);
int x4_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st);
void x4_variant(const avsep_conv_desc* d, char* buf, size_t cap);
bool ww_applicable(const avsep_conv_desc* d);
size_t ww_workspace_floats(const avsep_conv_desc* d);
int ww_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st);
bool w4d_applicable(const avsep_conv_desc* d);      // 4x4 / stride 2 on the same skeleton (direct form)
size_t w4d_workspace_floats(const avsep_conv_desc* d);
int w4d_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st);
bool c4_applicable(const avsep_conv_desc* d, int mode);
size_t c4_packed_floats(const avsep_conv_desc* d, int mode);
int c4_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st);
int c4_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, hipStream_t st);
int c4_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, hipStream_t st);
// conv_misc.hip: 3x3/s2 and 1x1 convolutions of the visual trunk on the halo-patch kernel (fp32)
bool cm_applicable(const avsep_conv_desc* d, int mode);
size_t cm_packed_floats(const avsep_conv_desc* d, int mode);
int cm_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st);
int cm_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, hipStream_t st);
int cm_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, hipStream_t st);
size_t w3_workspace_floats(const avsep_conv_desc* d);
// conv_bf16.hip: bf16-operand halo-patch kernels (desc.prec == AVSEP_PREC_BF16)
bool bf_applicable(const avsep_conv_desc* d, int mode);
size_t bf_workspace_bytes(const avsep_conv_desc* d, int mode);
size_t bf_packed_floats(const avsep_conv_desc* d, int mode);
int bf_pack(const avsep_conv_desc* d, const float* w, float* packed, int mode, hipStream_t st);
int bf_fwd(const avsep_conv_desc* d, const float* wp, const float* bias, float* y, double* stats, void* ws, size_t ws_bytes,
           hipStream_t st);
int bf_dgrad(const avsep_conv_desc* d, const float* wp, const float* dy, float* dx, void* ws, size_t ws_bytes, hipStream_t st);
bool bf_out_b16(const avsep_conv_desc* d, int mode);
// wgrad_b16.hip: bf16 weight gradient over B16 images (transposed LDS reads)
bool wbn_applicable(const avsep_conv_desc* d);
size_t wbn_workspace_floats(const avsep_conv_desc* d);
int wbn_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st);
void wbn_variant(const avsep_conv_desc* d, char* buf, size_t cap);
int b16_channel_sum(const void* x, int N, int C, int HW, double* acc, float* out, hipStream_t st);   // b16.hip
int w3_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st);
// wgrad_smallci.hip
bool scw_applicable(const avsep_conv_desc* d);
size_t scw_workspace_floats(const avsep_conv_desc* d);
int scw_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st);
static int check_desc(const avsep_conv_desc* d, bool fwd_only = false) {
  if (!d || !d->x0) return AVSEP_ERR_ARG;
  if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0) return AVSEP_ERR_ARG;
  if (d->KH <= 0 || d->KW <= 0 || d->stride <= 0 || d->dil <= 0) return AVSEP_ERR_ARG;
  // the dgrad tap tables and the wgrad column table hold at most 8 taps per axis / 8-bit offsets
  if (!fwd_only && (d->KH > 8 || d->KW > 8 || (d->KH - 1) * d->dil > 255 || (d->KW - 1) * d->dil > 255 ||
                    d->Cin > 65535))
    return AVSEP_ERR_ARG;
  if (d->C0 <= 0 || d->C0 > d->Cin || (d->C0 < d->Cin && !d->x1)) return AVSEP_ERR_ARG;
  int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  if (ho != d->Ho || wo != d->Wo) return AVSEP_ERR_ARG;
  if (d->up2x && ((d->H & 1) || (d->W & 1))) return AVSEP_ERR_ARG;
  if ((d->scale0 == nullptr) != (d->shift0 == nullptr)) return AVSEP_ERR_ARG;
  if ((d->scale1 == nullptr) != (d->shift1 == nullptr)) return AVSEP_ERR_ARG;
  if (d->prec != AVSEP_PREC_F32 && d->prec != AVSEP_PREC_BF16) return AVSEP_ERR_ARG;
  return AVSEP_OK;
}

static CArgs make_args(const avsep_conv_desc* d) {
  CArgs a{};
  a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
  a.C0 = d->C0; a.C1 = d->Cin - d->C0; a.act0 = d->act0; a.act1 = d->act1; a.up2x = d->up2x;
  a.Hs = d->up2x ? d->H / 2 : d->H;
  a.Ws = d->up2x ? d->W / 2 : d->W;
  a.rh = (d->up2x && d->H > 1) ? (float)(a.Hs - 1) / (float)(d->H - 1) : 0.f;
  a.rw = (d->up2x && d->W > 1) ? (float)(a.Ws - 1) / (float)(d->W - 1) : 0.f;
  a.x0 = d->x0; a.x1 = d->x1; a.sc0 = d->scale0; a.sh0 = d->shift0; a.sc1 = d->scale1; a.sh1 = d->shift1;
  return a;
}

static inline int packed_rows(const avsep_conv_desc* d, int mode) {
  return mode == 0 ? roundup(d->Cin * d->KH * d->KW, 32) : roundup(d->KH * d->KW * d->Cout, 32);
}
static inline int packed_ld(const avsep_conv_desc* d, int mode) { return roundup(mode == 0 ? d->Cout : d->Cin, 128); }

static bool wgrad_takes_b16(const avsep_conv_desc* d) {
  return !smallco_applicable(d) && !head_applicable(d) && wbn_applicable(d);
}
extern "C" int avsep_conv_io_formats(const avsep_conv_desc* d, int32_t mode, int32_t* in_fmt, int32_t* out_b16) {
  if (!in_fmt || !out_b16 || mode < 0 || mode > 2) return AVSEP_ERR_ARG;
  int rc = check_desc(d, mode == 0);
  if (rc) return rc;
  *in_fmt = AVSEP_FMT_F32;
  *out_b16 = 0;
  if (mode == 0) {
    if (!smallco_applicable(d) && !head_applicable(d) && bf_applicable(d, 0)) { *in_fmt = AVSEP_FMT_B16; *out_b16 = bf_out_b16(d, 0); }
  } else if (mode == 1) {
    if (!smallci_applicable(d) && !head_applicable(d) && bf_applicable(d, 1)) { *in_fmt = AVSEP_FMT_B16; *out_b16 = bf_out_b16(d, 1); }
  } else if (wgrad_takes_b16(d)) {
    *in_fmt = AVSEP_FMT_B16;
  }
  return AVSEP_OK;
}

extern "C" size_t avsep_conv_packed_floats(const avsep_conv_desc* d, int mode) {
  if (!d || (mode != 0 && mode != 1)) return 0;
  if (mode == 1 && smallci_applicable(d)) return (size_t)d->Cout * d->Cin * d->KH * d->KW;   // OIHW as is
  if (bf_applicable(d, mode)) return bf_packed_floats(d, mode);
  if (w4_applicable(d, mode)) return w4_packed_floats(d, mode);
  if (wn_applicable(d, mode)) return wn_packed_floats(d, mode);
  if (c3_applicable(d, mode)) return c3_packed_floats(d, mode);
  if (c4_applicable(d, mode)) return c4_packed_floats(d, mode);
  if (cm_applicable(d, mode)) return cm_packed_floats(d, mode);
  return (size_t)packed_rows(d, mode) * packed_ld(d, mode);
}

extern "C" int avsep_conv_pack_weights(const avsep_conv_desc* d, const float* w, float* packed, int mode,
                                       avsep_stream_t stream) {
  if (!d || !w || !packed || (mode != 0 && mode != 1)) return AVSEP_ERR_ARG;
  if (mode == 1 && smallci_applicable(d)) {
    if (hipMemcpyAsync(packed, w, (size_t)d->Cout * d->Cin * d->KH * d->KW * sizeof(float), hipMemcpyDeviceToDevice,
                       (hipStream_t)stream) != hipSuccess)
      return AVSEP_ERR_LAUNCH;
    return AVSEP_OK;
  }
  if (bf_applicable(d, mode)) return bf_pack(d, w, packed, mode, (hipStream_t)stream);
  if (w4_applicable(d, mode)) return w4_pack(d, w, packed, mode, (hipStream_t)stream);
  if (wn_applicable(d, mode)) return wn_pack(d, w, packed, mode, (hipStream_t)stream);
  if (c3_applicable(d, mode)) return c3_pack(d, w, packed, mode, (hipStream_t)stream);
  if (c4_applicable(d, mode)) return c4_pack(d, w, packed, mode, (hipStream_t)stream);
  if (cm_applicable(d, mode)) return cm_pack(d, w, packed, mode, (hipStream_t)stream);
  int rows = packed_rows(d, mode), ld = packed_ld(d, mode);
  long long total = (long long)rows * ld;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, packed,
                     d->Cout, d->Cin, d->KH, d->KW, rows, ld, mode);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// big tiles when they still fill the chip, else 64x64
static inline bool use_big(int M, long long ncols) {
  return M > 64 && (long long)cdiv(M, 128) * cdiv(ncols, 128) >= 64;   // few big tiles are topped up by split-K
}

// split-K plan for layers whose output grid cannot fill the chip (deep U-Net levels: 2x2 .. 8x8 maps, K ~ 8-9k)
struct SplitPlan { int splits, kts; };
static SplitPlan splitk_plan(long long tiles, int K) {
  SplitPlan p{1, 0};
  int nK = cdiv(K, 16);
  if (tiles >= 640 || nK < 64) return p;
  int want = cdiv(768, tiles), maxs = nK / 32;
  int s = want < maxs ? want : maxs;
  if (s > 32) s = 32;
  if (s < 2) return p;
  p.kts = cdiv(nK, s);
  p.splits = cdiv(nK, p.kts);
  if (p.splits < 2) { p.splits = 1; p.kts = 0; }
  return p;
}
static bool fwd_uses_igemm(const avsep_conv_desc* d, const double* stats) {
  return !((!stats && (smallco_applicable(d) || head_applicable(d))) || bf_applicable(d, 0) || w4_applicable(d, 0) || wn_applicable(d, 0) ||
           c3_applicable(d, 0) || c4_applicable(d, 0) || cm_applicable(d, 0));
}
// the im2col kernel's tile size and split-K are decided on the planned batch (plan_desc), like every launch heuristic
static bool fwd_big(const avsep_conv_desc* d) { return use_big(d->Cout, (long long)plan_batch(d) * d->Ho * d->Wo); }
static bool dgrad_big(const avsep_conv_desc* d) {
  const int s = d->stride;
  return use_big(d->Cin, (long long)plan_batch(d) * cdiv(d->H, s) * cdiv(d->W, s) * s * s);
}
static SplitPlan fwd_split(const avsep_conv_desc* d) {
  long long ncols = (long long)plan_batch(d) * d->Ho * d->Wo;
  bool big = use_big(d->Cout, ncols);
  long long tiles = big ? (long long)cdiv(d->Cout, 128) * cdiv(ncols, 128) : (long long)cdiv(d->Cout, 64) * cdiv(ncols, 64);
  return splitk_plan(tiles, d->Cin * d->KH * d->KW);
}
static SplitPlan dgrad_split(const avsep_conv_desc* d) {
  const int s = d->stride;
  long long ncols = (long long)plan_batch(d) * cdiv(d->H, s) * cdiv(d->W, s);
  bool big = use_big(d->Cin, ncols * s * s);
  long long tiles = (big ? (long long)cdiv(d->Cin, 128) * cdiv(ncols, 128) : (long long)cdiv(d->Cin, 64) * cdiv(ncols, 64)) * s * s;
  int taps = cdiv(d->KH, s) * cdiv(d->KW, s);               // taps of the fullest parity class
  return splitk_plan(tiles, taps * d->Cout);
}

// split-K combines, shared with conv_bf16.hip
int splitk_combine(const float* ws, long long slab, int S, const avsep_conv_desc* d, const float* bias, float* y, double* stats,
                   hipStream_t st) {
  int chunks = min(cdiv(1024, d->Cout), d->N);
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(splitk_combine_kernel, dim3(d->Cout, chunks), dim3(256), 0, st, ws, slab, S, d->N, d->Cout, d->Ho * d->Wo,
                     bias, y, stats);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
int reduce_slabs_strided(const float* ws, float* out, long long n, int S, long long stride, hipStream_t st) {
  return launch_reduce_slabs(ws, out, n, S, stride, st);
}
int reduce_slabs(const float* ws, float* out, long long n, int S, hipStream_t st) { return launch_reduce_slabs(ws, out, n, S, n, st); }

static size_t fwd_workspace_no_head(const avsep_conv_desc* d) {
  if (!check_desc(d, true) && bf_applicable(d, 0)) return bf_workspace_bytes(d, 0);
  if (check_desc(d, true) || !fwd_uses_igemm(d, (const double*)1)) return 0;
  SplitPlan p = fwd_split(d);
  return p.splits > 1 ? (size_t)p.splits * d->N * d->Cout * d->Ho * d->Wo * sizeof(float) : 0;
}
extern "C" size_t avsep_conv2d_fwd_workspace_bytes(const avsep_conv_desc* d) {
  size_t need = fwd_workspace_no_head(d);
  if (!check_desc(d, true) && head_applicable(d)) {      // the head kernels serve the call without statistics: cover both
    const size_t h = head_fwd_workspace_floats(d) * sizeof(float);
    if (h > need) need = h;
  }
  return need;
}
extern "C" size_t avsep_conv2d_dgrad_workspace_bytes(const avsep_conv_desc* d) {
  if (!check_desc(d) && !smallci_applicable(d) && bf_applicable(d, 1)) return bf_workspace_bytes(d, 1);
  if (check_desc(d) || bf_applicable(d, 1) || w4_applicable(d, 1) || wn_applicable(d, 1) || c3_applicable(d, 1) || smallci_applicable(d) || c4_applicable(d, 1) ||
      cm_applicable(d, 1))
    return 0;
  SplitPlan p = dgrad_split(d);
  return p.splits > 1 ? (size_t)p.splits * d->N * d->Cin * d->H * d->W * sizeof(float) : 0;
}

extern "C" int avsep_conv2d_fwd(const avsep_conv_desc* d, const float* w_packed, const float* bias, float* y,
                                double* stats, void* workspace, size_t workspace_bytes, avsep_stream_t stream) {
  int rc = check_desc(d, true);
  if (rc) return rc;
  if (!w_packed || !y) return AVSEP_ERR_ARG;
  if (!stats && smallco_applicable(d)) return smallco_fwd(d, w_packed, packed_ld(d, 0), bias, y, (hipStream_t)stream);
  if (!stats && head_applicable(d)) {
    if (!workspace || workspace_bytes < head_fwd_workspace_floats(d) * sizeof(float)) return AVSEP_ERR_WORKSPACE;
    return head_fwd(d, w_packed, packed_ld(d, 0), bias, y, (float*)workspace, (hipStream_t)stream);
  }
  if (bf_applicable(d, 0)) return bf_fwd(d, w_packed, bias, y, stats, workspace, workspace_bytes, (hipStream_t)stream);
  if (w4_applicable(d, 0)) return w4_fwd(d, w_packed, bias, y, stats, (hipStream_t)stream);
  if (wn_applicable(d, 0)) return wn_fwd(d, w_packed, bias, y, stats, (hipStream_t)stream);
  if (c3_applicable(d, 0)) return c3_fwd(d, w_packed, bias, y, stats, (hipStream_t)stream);
  if (c4_applicable(d, 0)) return c4_fwd(d, w_packed, bias, y, stats, (hipStream_t)stream);
  if (cm_applicable(d, 0)) return cm_fwd(d, w_packed, bias, y, stats, (hipStream_t)stream);
  CArgs a = make_args(d);
  a.wp = w_packed; a.wp_ld = packed_ld(d, 0); a.out = y; a.bias = bias; a.stats = stats;
  a.M = d->Cout; a.K = d->Cin * d->KH * d->KW;
  long long ncols = (long long)d->N * d->Ho * d->Wo;
  if (ncols > 0x7fffffffLL) return AVSEP_ERR_ARG;
  a.Ncols = (int)ncols;
  hipStream_t st = (hipStream_t)stream;
  SplitPlan sp = fwd_split(d);
  if (sp.splits > 1 && (!workspace || workspace_bytes < avsep_conv2d_fwd_workspace_bytes(d))) sp = SplitPlan{1, 0};
  if (sp.splits > 1) {
    a.kts = sp.kts;
    a.slab = (long long)d->N * d->Cout * d->Ho * d->Wo;
    a.out = (float*)workspace;
  }
  if (fwd_big(d)) {
    a.gridM = cdiv(a.M, 128);
    if (d->up2x) hipLaunchKernelGGL((igemm_kernel<M_FWD, 128, 128, 16, 2, 2, true>), dim3(a.gridM * cdiv(ncols, 128), sp.splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_kernel<M_FWD, 128, 128, 16, 2, 2, false>), dim3(a.gridM * cdiv(ncols, 128), sp.splits), dim3(256), 0, st, a);
  } else {
    a.gridM = cdiv(a.M, 64);
    if (d->up2x) hipLaunchKernelGGL((igemm_kernel<M_FWD, 64, 64, 16, 2, 2, true>), dim3(a.gridM * cdiv(ncols, 64), sp.splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_kernel<M_FWD, 64, 64, 16, 2, 2, false>), dim3(a.gridM * cdiv(ncols, 64), sp.splits), dim3(256), 0, st, a);
  }
  AVSEP_LAUNCH_CHECK();
  if (sp.splits > 1) {
    int chunks = min(cdiv(1024, d->Cout), d->N);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(splitk_combine_kernel, dim3(d->Cout, chunks), dim3(256), 0, st, (const float*)workspace, a.slab,
                       sp.splits, d->N, d->Cout, d->Ho * d->Wo, bias, y, stats);
    AVSEP_LAUNCH_CHECK();
  }
  return AVSEP_OK;
}

extern "C" int avsep_conv2d_dgrad(const avsep_conv_desc* d, const float* w_packed_dgrad, const float* dy, float* dx,
                                  void* workspace, size_t workspace_bytes, avsep_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!w_packed_dgrad || !dy || !dx) return AVSEP_ERR_ARG;
  if (smallci_applicable(d)) return smallci_dgrad(d, w_packed_dgrad, dy, dx, (hipStream_t)stream);
  if (bf_applicable(d, 1)) return bf_dgrad(d, w_packed_dgrad, dy, dx, workspace, workspace_bytes, (hipStream_t)stream);
  if (w4_applicable(d, 1)) return w4_dgrad(d, w_packed_dgrad, dy, dx, nullptr, (hipStream_t)stream);
  if (wn_applicable(d, 1)) return wn_dgrad(d, w_packed_dgrad, dy, dx, (hipStream_t)stream);
  if (c3_applicable(d, 1)) return c3_dgrad(d, w_packed_dgrad, dy, dx, (hipStream_t)stream);
  if (c4_applicable(d, 1)) return c4_dgrad(d, w_packed_dgrad, dy, dx, (hipStream_t)stream);
  if (cm_applicable(d, 1)) return cm_dgrad(d, w_packed_dgrad, dy, dx, (hipStream_t)stream);
  CArgs a = make_args(d);
  a.wp = w_packed_dgrad; a.wp_ld = packed_ld(d, 1); a.dy = dy; a.out = dx;
  a.M = d->Cin;
  const int s = d->stride;
  if (s != 1 && s != 2 && s != 4) return AVSEP_ERR_ARG;
  a.lstride = s == 1 ? 0 : (s == 2 ? 1 : 2);
  long long ncols = (long long)d->N * cdiv(d->H, s) * cdiv(d->W, s);  // largest parity class
  if ((long long)d->N * d->H * d->W > 0x7fffffffLL) return AVSEP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  SplitPlan sp = dgrad_split(d);
  if (sp.splits > 1 && (!workspace || workspace_bytes < avsep_conv2d_dgrad_workspace_bytes(d))) sp = SplitPlan{1, 0};
  if (sp.splits > 1) {
    a.kts = sp.kts;
    a.slab = (long long)d->N * d->Cin * d->H * d->W;
    a.out = (float*)workspace;
  }
  if (dgrad_big(d)) {
    a.gridM = cdiv(a.M, 128);
    hipLaunchKernelGGL((igemm_kernel<M_DGRAD, 128, 128, 16, 2, 2>), dim3(a.gridM * cdiv(ncols, 128), s * s, sp.splits),
                       dim3(256), 0, st, a);
  } else {
    a.gridM = cdiv(a.M, 64);
    hipLaunchKernelGGL((igemm_kernel<M_DGRAD, 64, 64, 16, 2, 2>), dim3(a.gridM * cdiv(ncols, 64), s * s, sp.splits),
                       dim3(256), 0, st, a);
  }
  AVSEP_LAUNCH_CHECK();
  if (sp.splits > 1) {
    hipLaunchKernelGGL(reduce_slabs_kernel<64>, dim3(cdiv(a.slab, 64)), dim3(256), 0, st, (const float*)workspace, dx, a.slab,
                       sp.splits, a.slab);
    AVSEP_LAUNCH_CHECK();
  }
  return AVSEP_OK;
}

struct WgradPlan {
  bool big;
  int splits, chunk, tiles;
};
static WgradPlan wgrad_plan(const avsep_conv_desc* d) {
  WgradPlan p;
  int M = d->Cout, Kc = d->Cin * d->KH * d->KW;
  long long P = (long long)d->N * d->Ho * d->Wo;
  p.big = M > 64 && Kc > 64;
  int bm = p.big ? 128 : 64;
  p.tiles = cdiv(M, bm) * cdiv(Kc, bm);
  int want = cdiv(1024, p.tiles);                       // aim for ~1024 workgroups
  long long maxs = (P + 255) / 256;                      // at least 256 pixels per split
  int splits = (int)((want < maxs) ? want : maxs);
  if (splits < 1) splits = 1;
  long long chunk = (P + splits - 1) / splits;
  chunk = (chunk + 31) / 32 * 32;
  p.chunk = (int)chunk;
  p.splits = (int)((P + chunk - 1) / chunk);
  return p;
}

// the data gradient through the activation in front of the conv's input (include/avsep.h): in the F(4x4) Winograd kernel's
// epilogue, else as the two launches it stands for
static bool dgrad_act_fused(const avsep_conv_desc* d) {
  return !smallci_applicable(d) && !bf_applicable(d, 1) && w4_applicable(d, 1);
}
extern "C" int32_t avsep_conv2d_dgrad_act_fused(const avsep_conv_desc* d) { return (!check_desc(d) && dgrad_act_fused(d)) ? 1 : 0; }
extern "C" int avsep_conv2d_dgrad_act(const avsep_conv_desc* d, const float* w_packed_dgrad, const float* dy,
                                      const avsep_act_bwd* e, float* dx, void* workspace, size_t workspace_bytes,
                                      avsep_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!w_packed_dgrad || !dy || !dx || !e || !e->y) return AVSEP_ERR_ARG;
  if (d->dxfmt != AVSEP_FMT_F32) return AVSEP_ERR_ARG;
  if ((e->scale == nullptr) != (e->shift == nullptr) || (e->res_scale == nullptr) != (e->res_shift == nullptr) ||
      (e->res_scale && !e->residual) || (e->bstats && (!e->mean || !e->invstd)))
    return AVSEP_ERR_ARG;
  if (e->act != AVSEP_ACT_NONE && e->act != AVSEP_ACT_RELU && e->act != AVSEP_ACT_LRELU02) return AVSEP_ERR_ARG;
  if (dgrad_act_fused(d)) return w4_dgrad(d, w_packed_dgrad, dy, dx, e, (hipStream_t)stream);
  rc = avsep_conv2d_dgrad(d, w_packed_dgrad, dy, dx, workspace, workspace_bytes, stream);
  if (rc) return rc;
  return avsep_affine_act_bwd(dx, e->dz2, e->y, e->scale, e->shift, e->residual, e->res_scale, e->res_shift, e->add, e->mean,
                              e->invstd, e->act, d->N, d->Cin, d->H * d->W, dx, e->bstats, stream);
}

extern "C" size_t avsep_conv2d_wgrad_workspace_bytes(const avsep_conv_desc* d) {
  if (check_desc(d)) return 0;
  if (smallco_applicable(d)) return smallco_wgrad_workspace_floats(d) * sizeof(float);
  if (head_applicable(d)) return head_wgrad_workspace_floats(d) * sizeof(float);
  if (wbn_applicable(d)) return wbn_workspace_floats(d) * sizeof(float) + (size_t)2 * d->Cout * sizeof(double);
  if (x4_applicable(d)) return x4_workspace_floats(d) * sizeof(float);
  if (ww_applicable(d)) return ww_workspace_floats(d) * sizeof(float);
  if (w4d_applicable(d)) return w4d_workspace_floats(d) * sizeof(float);
  if (w3_applicable(d)) return w3_workspace_floats(d) * sizeof(float);
  if (scw_applicable(d)) return scw_workspace_floats(d) * sizeof(float);
  WgradPlan p = wgrad_plan(d);
  if (p.splits <= 1) return 0;
  return (size_t)p.splits * d->Cout * d->Cin * d->KH * d->KW * sizeof(float);
}

extern "C" int avsep_conv2d_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* dbias, void* workspace,
                                  size_t workspace_bytes, avsep_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dy || !dw) return AVSEP_ERR_ARG;
  WgradPlan p = wgrad_plan(d);
  size_t need = avsep_conv2d_wgrad_workspace_bytes(d);
  if (need > workspace_bytes || (need && !workspace)) return AVSEP_ERR_WORKSPACE;
  if (smallco_applicable(d)) return smallco_wgrad(d, dy, dw, dbias, (float*)workspace, (hipStream_t)stream);
  if (head_applicable(d)) return head_wgrad(d, dy, dw, dbias, (float*)workspace, (hipStream_t)stream);
  if (wbn_applicable(d)) {
    int rcw = wbn_wgrad(d, dy, dw, (float*)workspace, (hipStream_t)stream);
    if (rcw || !dbias) return rcw;
    double* acc = reinterpret_cast<double*>((float*)workspace + wbn_workspace_floats(d));   // behind the slabs (8-byte aligned: slab sizes are multiples of 64*64)
    return b16_channel_sum(dy, d->N, d->Cout, d->Ho * d->Wo, acc, dbias, (hipStream_t)stream);
  }
  if (x4_applicable(d) || ww_applicable(d) || w4d_applicable(d) || w3_applicable(d) || scw_applicable(d)) {
    int rc3 = x4_applicable(d) ? x4_wgrad(d, dy, dw, (float*)workspace, (hipStream_t)stream)
              : ww_applicable(d) ? ww_wgrad(d, dy, dw, (float*)workspace, (hipStream_t)stream)
              : w4d_applicable(d) ? w4d_wgrad(d, dy, dw, (float*)workspace, (hipStream_t)stream)
              : w3_applicable(d) ? w3_wgrad(d, dy, dw, (float*)workspace, (hipStream_t)stream)
                                 : scw_wgrad(d, dy, dw, (float*)workspace, (hipStream_t)stream);
    if (rc3) return rc3;
    if (dbias) {
      hipLaunchKernelGGL(channel_sum_kernel, dim3(d->Cout), dim3(256), 0, (hipStream_t)stream, dy, d->N, d->Cout,
                         d->Ho * d->Wo, dbias);
      AVSEP_LAUNCH_CHECK();
    }
    return AVSEP_OK;
  }
  CArgs a = make_args(d);
  a.dy = dy;
  a.M = d->Cout; a.Ncols = d->Cin * d->KH * d->KW; a.K = 0;
  long long P = (long long)d->N * d->Ho * d->Wo;
  if (P > 0x7fffffffLL) return AVSEP_ERR_ARG;
  a.P = (int)P; a.chunk = p.chunk;
  a.out = (p.splits > 1) ? (float*)workspace : dw;
  hipStream_t st = (hipStream_t)stream;
  if (p.big) {
    a.gridM = cdiv(a.M, 128);
    if (d->up2x) hipLaunchKernelGGL((igemm_kernel<M_WGRAD, 128, 128, 32, 2, 2, true>), dim3(a.gridM * cdiv(a.Ncols, 128), p.splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_kernel<M_WGRAD, 128, 128, 32, 2, 2, false>), dim3(a.gridM * cdiv(a.Ncols, 128), p.splits), dim3(256), 0, st, a);
  } else {
    a.gridM = cdiv(a.M, 64);
    if (d->up2x) hipLaunchKernelGGL((igemm_kernel<M_WGRAD, 64, 64, 32, 2, 2, true>), dim3(a.gridM * cdiv(a.Ncols, 64), p.splits), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_kernel<M_WGRAD, 64, 64, 32, 2, 2, false>), dim3(a.gridM * cdiv(a.Ncols, 64), p.splits), dim3(256), 0, st, a);
  }
  AVSEP_LAUNCH_CHECK();
  if (p.splits > 1) {
    long long n = (long long)a.M * a.Ncols;
    hipLaunchKernelGGL(reduce_slabs_kernel<64>, dim3(cdiv(n, 64)), dim3(256), 0, st, (const float*)workspace, dw, n, p.splits, n);
    AVSEP_LAUNCH_CHECK();
  }
  if (dbias) {
    hipLaunchKernelGGL(channel_sum_kernel, dim3(d->Cout), dim3(256), 0, st, dy, d->N, d->Cout, d->Ho * d->Wo, dbias);
    AVSEP_LAUNCH_CHECK();
  }
  return AVSEP_OK;
}

extern "C" const char* avsep_conv_kernel_name(const avsep_conv_desc* d, int32_t mode, int32_t with_stats) {
  if (check_desc(d, mode == 0)) return "invalid";
  if (mode == 0) {
    if (!with_stats && smallco_applicable(d)) return "smallco_fwd";
    if (!with_stats && head_applicable(d)) return "head_fwd_kernel";
    if (bf_applicable(d, 0)) return "convbf_kernel";
    if (w4_applicable(d, 0)) return "wino4_kernel";
    if (wn_applicable(d, 0)) return "wino_kernel";
    if (c3_applicable(d, 0) || c4_applicable(d, 0) || cm_applicable(d, 0)) return "conv3x3_kernel";
    return "igemm_kernel<fwd>";
  }
  if (mode == 1) {
    if (head_applicable(d)) return "head_dgrad_kernel";
    if (smallci_applicable(d)) return "smallci_dgrad";
    if (bf_applicable(d, 1)) return "convbf_kernel";
    if (w4_applicable(d, 1)) return "wino4_kernel";
    if (wn_applicable(d, 1)) return "wino_kernel";
    if (c3_applicable(d, 1) || c4_applicable(d, 1) || cm_applicable(d, 1)) return "conv3x3_kernel";
    return "igemm_kernel<dgrad>";
  }
  if (smallco_applicable(d)) return "smallco_wgrad";
  if (head_applicable(d)) return "head_wgrad_kernel";
  if (wbn_applicable(d)) return "wgradb_kernel";
  if (x4_applicable(d)) return "winow4_kernel";
  if (ww_applicable(d)) return "winow_kernel";
  if (w4d_applicable(d)) return "wgrad4d_kernel";
  if (w3_applicable(d)) return "wgrad3x3_kernel";
  if (scw_applicable(d)) return "smallci_wgrad_kernel";
  return "igemm_kernel<wgrad>";
}

void bf_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap);      // conv_bf16.hip
void c3_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap);      // conv3x3.hip
void c4_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap);
void cm_variant(const avsep_conv_desc* d, int mode, char* buf, size_t cap);      // conv_misc.hip

extern "C" int avsep_conv_kernel_variant(const avsep_conv_desc* d, int32_t mode, int32_t with_stats, char* buf, size_t cap) {
  if (!buf || cap < 8) return AVSEP_ERR_ARG;
  const char* fam = avsep_conv_kernel_name(d, mode, with_stats);
  char tail[64] = "";
  if (!strcmp(fam, "convbf_kernel")) bf_variant(d, mode, tail, sizeof(tail));
  else if (!strcmp(fam, "wgradb_kernel")) wbn_variant(d, tail, sizeof(tail));
  else if (!strcmp(fam, "wino4_kernel")) w4_variant(d, mode, tail, sizeof(tail));
  else if (!strcmp(fam, "winow4_kernel")) x4_variant(d, tail, sizeof(tail));
  else if (!strcmp(fam, "conv3x3_kernel")) {
    if (c3_applicable(d, mode)) c3_variant(d, mode, tail, sizeof(tail));
    else if (c4_applicable(d, mode)) c4_variant(d, mode, tail, sizeof(tail));
    else cm_variant(d, mode, tail, sizeof(tail));
  } else if (!strcmp(fam, "igemm_kernel<fwd>")) {
    snprintf(tail, sizeof(tail), "BM%d,split%d", fwd_big(d) ? 128 : 64, fwd_split(d).splits);
  } else if (!strcmp(fam, "igemm_kernel<dgrad>")) {
    snprintf(tail, sizeof(tail), "BM%d,split%d", dgrad_big(d) ? 128 : 64, dgrad_split(d).splits);
  }
  snprintf(buf, cap, tail[0] ? "%s:%s" : "%s", fam, tail);
  return AVSEP_OK;
}

// ---------------------------------------------------------------------------
// fused decoder head (head_gemm.hip): dgrad of a conv over the virtual up2x(relu(affine(cat))) input, taken straight to
// the two low-res sources
// ---------------------------------------------------------------------------
extern "C" int32_t avsep_conv2d_head_applicable(const avsep_conv_desc* d) {
  return (check_desc(d) == AVSEP_OK && head_applicable(d)) ? 1 : 0;
}
extern "C" size_t avsep_conv2d_dgrad_up2x_workspace_bytes(const avsep_conv_desc* d) {
  return (check_desc(d) == AVSEP_OK && head_applicable(d)) ? head_dgrad_workspace_floats(d) * sizeof(float) : 0;
}
extern "C" int avsep_conv2d_dgrad_up2x(const avsep_conv_desc* d, const float* w, const float* dy, float* g0, float* g1,
                                       const float* mean1, const float* invstd1, double* bstats1, int32_t acc0,
                                       void* workspace, size_t workspace_bytes, avsep_stream_t stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!w || !dy || (!g0 && !g1)) return AVSEP_ERR_ARG;
  if (bstats1 && (!mean1 || !invstd1 || !g1)) return AVSEP_ERR_ARG;
  if (!head_applicable(d)) return AVSEP_ERR_ARG;   // unsupported geometry: use avsep_conv2d_dgrad + avsep_relu_up2x_bwd
  if (!workspace || workspace_bytes < head_dgrad_workspace_floats(d) * sizeof(float)) return AVSEP_ERR_WORKSPACE;
  return head_dgrad(d, w, dy, g0, g1, mean1, invstd1, bstats1, acc0, (float*)workspace, (hipStream_t)stream);
}
