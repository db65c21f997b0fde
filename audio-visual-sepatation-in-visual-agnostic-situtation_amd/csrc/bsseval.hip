// BSS-eval SDR / SIR / SAR (the reference scores with asteroid -> mir_eval.separation.bss_eval_sources,
// main.py:260-266): each estimate is projected, by least squares, on the span of `flen` = 512 delayed copies of (a) its own
// true source and (b) all true sources (Vincent et al. 2006, bss_decomp_mtifilt).  Everything is float64, as in mir_eval.
//   1. bss_corr_kernel     lagged correlations of the references with one another (the block-Toeplitz Gram matrix is made
//                          of them) and with the estimates (the right-hand sides): direct sums — no FFT, 1023 + 512 lags
//   2. bss_solve_kernel    per system (one workgroup): build the M x M Gram matrix (M = flen or S * flen) column-major in the
//                          workspace, LU with partial pivoting (what numpy.linalg.solve does), forward / back substitution
//   3. bss_project_kernel  the projection itself: sum_i conv(C_i, ref_i) over the L + flen - 1 output samples
// The residual energies / log10 of the three ratios are a few elementwise torch ops on the host side (bss_eval.py).
#include "common.h"

// ---- 1. correlations -------------------------------------------------------------------------------------------------------------
// R[b][i][j][u], u = tau + flen - 1, tau in (-flen, flen):  sum_t ref_i[t + tau] * ref_j[t]
// D[b][e][i][k], k in [0, flen):                            sum_t ref_i[t - k]   * est_e[t]
// grid (chunks of BC_CH samples, S*S + E*S pairs, B); a thread owns one lag and sums its chunk from LDS; fp64 atomics combine.
constexpr int BC_CH = 1024;
__global__ __launch_bounds__(1024) void bss_corr_kernel(const double* __restrict__ refs, const double* __restrict__ ests, int S, int E,
                                                        int L, int flen, double* __restrict__ R, double* __restrict__ D) {
  extern __shared__ double bc_smem[];
  double* const aw = bc_smem;                       // ref_i[t0 - (flen-1) .. t0 + BC_CH + flen - 1)
  double* const bw = bc_smem + BC_CH + 2 * (flen - 1);
  const int b = blockIdx.z, pair = blockIdx.y, t0 = blockIdx.x * BC_CH, tid = threadIdx.x;
  const bool rr = pair < S * S;
  const int i = rr ? pair / S : (pair - S * S) % S, other = rr ? pair % S : (pair - S * S) / S;
  const double* const a = refs + ((long long)b * S + i) * L;
  const double* const bb = rr ? refs + ((long long)b * S + other) * L : ests + ((long long)b * E + other) * L;
  const int AW = BC_CH + 2 * (flen - 1);
  for (int x = tid; x < AW; x += 1024) {
    const int t = t0 - (flen - 1) + x;
    aw[x] = (t >= 0 && t < L) ? a[t] : 0.0;
  }
  for (int x = tid; x < BC_CH; x += 1024) bw[x] = (t0 + x < L) ? bb[t0 + x] : 0.0;
  __syncthreads();
  const int nl = rr ? 2 * flen - 1 : flen;
  if (tid < nl) {
    // rr: a index = t + tau = t + (tid - (flen-1))  -> window offset x = (t - t0) + tid
    // er: a index = t - k                            -> window offset x = (t - t0) + (flen-1) - tid
    const double* ap = aw + (rr ? tid : flen - 1 - tid);
    double acc = 0.0;
#pragma unroll 8
    for (int x = 0; x < BC_CH; ++x) acc = fma(ap[x], bw[x], acc);
    double* dst = rr ? R + (((long long)b * S + i) * S + other) * (2 * flen - 1) + tid
                     : D + (((long long)b * E + other) * S + i) * flen + tid;
    atomicAdd(dst, acc);
  }
}
extern "C" int avsep_bss_corr(const double* refs, const double* ests, int32_t B, int32_t S, int32_t E, int32_t L, int32_t flen,
                              double* R, double* D, avsep_stream_t stream) {
  if (!refs || !ests || !R || !D || B <= 0 || B > 65535 || S <= 0 || S > 8 || E <= 0 || E > 8 || L <= 0 || flen <= 0 || flen > 512)
    return AVSEP_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(R, 0, sizeof(double) * (size_t)B * S * S * (2 * flen - 1), st) != hipSuccess) return AVSEP_ERR_LAUNCH;
  if (hipMemsetAsync(D, 0, sizeof(double) * (size_t)B * E * S * flen, st) != hipSuccess) return AVSEP_ERR_LAUNCH;
  const size_t lds = sizeof(double) * (2 * BC_CH + 2 * (flen - 1));
  hipLaunchKernelGGL(bss_corr_kernel, dim3(cdiv(L, BC_CH), S * S + E * S, B), dim3(1024), lds, st, refs, ests, S, E, L, flen, R, D);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- 2. the least-squares systems ---------------------------------------------------------------------------------------------------
// mode 0 ("all sources"): system b, M = S * flen, A[(i,a)][(j,c)] = R[b][i][j][c - a + flen - 1], right-hand side e: D[b][e][i][a]
// mode 1 ("own source"):  system b * S + j, M = flen, A[a][c] = R[b][j][j][c - a + flen - 1], one right-hand side D[b][j][j][a]
// One workgroup of 1024 threads per system; A column-major in `work` (a thread owns rows tid, tid + 1024, ...: every access to a
// column is coalesced); LU with partial pivoting exactly as LAPACK getrf / numpy.linalg.solve (row swaps applied at once), then
// the two triangular solves on the right-hand sides held in LDS.  info[sys] = k + 1 when the k-th pivot is exactly zero (a silent
// source: the caller falls back to a minimum-norm least-squares solve, as mir_eval does).
constexpr int BS_MAXROWS = 2;                 // M <= 2048: rows per thread
constexpr int BS_MAXRHS = 4;
__global__ __launch_bounds__(1024) void bss_solve_kernel(const double* __restrict__ R, const double* __restrict__ D, int S, int E, int flen,
                                                         int mode, double* __restrict__ work, double* __restrict__ C,
                                                         int* __restrict__ info) {
  extern __shared__ double bs_smem[];
  const int sys = blockIdx.x, tid = threadIdx.x;
  const int b = mode == 0 ? sys : sys / S, own = mode == 0 ? 0 : sys % S;
  const int M = mode == 0 ? S * flen : flen, nrhs = mode == 0 ? E : 1, NL = 2 * flen - 1;
  double* const A = work + (long long)sys * M * M;
  double* const rowk = bs_smem;                     // [M]  row k of U during step k
  double* const x = bs_smem + M;                    // [nrhs][M]
  __shared__ double red_v[16];
  __shared__ int red_i[16];
  __shared__ int s_piv;
  __shared__ double s_pivval;

  // build A (column-major) and the right-hand sides
  for (int c = 0; c < M; ++c) {
    const int j = mode == 0 ? c / flen : own, cc = c % flen;
    for (int r = tid; r < M; r += 1024) {
      const int i = mode == 0 ? r / flen : own, a = r % flen;
      A[(long long)c * M + r] = R[(((long long)b * S + i) * S + j) * NL + (cc - a + flen - 1)];
    }
  }
  for (int q = 0; q < nrhs; ++q)
    for (int r = tid; r < M; r += 1024) {
      const int i = mode == 0 ? r / flen : own, a = r % flen, e = mode == 0 ? q : own;
      x[q * M + r] = D[(((long long)b * E + e) * S + i) * flen + a];
    }
  __syncthreads();

  bool singular = false;
  for (int k = 0; k < M; ++k) {
    // pivot search in column k
    double best = -1.0;
    int bi = k;
    for (int r = tid; r < M; r += 1024)
      if (r >= k) {
        const double v = fabs(A[(long long)k * M + r]);
        if (v > best) { best = v; bi = r; }
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      double bv = red_v[0];
      int bx = red_i[0];
      for (int w = 1; w < 16; ++w)
        if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bx)) { bv = red_v[w]; bx = red_i[w]; }
      s_piv = bx;
      s_pivval = bv;
    }
    __syncthreads();
    const int p = s_piv;
    if (!(s_pivval > 0.0)) { singular = true; if (tid == 0) info[sys] = k + 1; break; }   // exactly singular (or NaN)
    // swap rows k and p in every column and in the right-hand sides; stage row k of U
    for (int c = tid; c < M; c += 1024) {
      double vk = A[(long long)c * M + k];
      if (p != k) {
        const double vp = A[(long long)c * M + p];
        A[(long long)c * M + p] = vk;
        A[(long long)c * M + k] = vp;
        vk = vp;
      }
      rowk[c] = vk;
    }
    if (p != k && tid < nrhs) {
      const double t = x[tid * M + k];
      x[tid * M + k] = x[tid * M + p];
      x[tid * M + p] = t;
    }
    __syncthreads();
    const double inv = 1.0 / rowk[k];
    // column k of L, trailing update, and the forward substitution of the right-hand sides folded into the same sweep
    for (int r = tid; r < M; r += 1024)
      if (r > k) {
        const double l = A[(long long)k * M + r] * inv;
        A[(long long)k * M + r] = l;
        // the sweep is latency-bound (an 8 MB matrix per system, one row element per column): 16 independent loads in flight
        int c = k + 1;
        for (; c + 16 <= M; c += 16) {
          double v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = A[(long long)(c + u) * M + r];
#pragma unroll
          for (int u = 0; u < 16; ++u) A[(long long)(c + u) * M + r] = fma(-l, rowk[c + u], v[u]);
        }
        for (; c < M; ++c) A[(long long)c * M + r] = fma(-l, rowk[c], A[(long long)c * M + r]);
        for (int q = 0; q < nrhs; ++q) x[q * M + r] = fma(-l, x[q * M + k], x[q * M + r]);
      }
    __syncthreads();
  }
  if (singular) {
    for (int q = 0; q < nrhs; ++q)
      for (int r = tid; r < M; r += 1024) C[((long long)sys * M + r) * nrhs + q] = 0.0;
    return;
  }
  if (tid == 0) info[sys] = 0;
  // back substitution with U (the forward half was done on the fly)
  for (int k = M - 1; k >= 0; --k) {
    if (tid < nrhs) x[tid * M + k] /= A[(long long)k * M + k];
    __syncthreads();
    for (int r = tid; r < k; r += 1024) {
      const double u = A[(long long)k * M + r];
      for (int q = 0; q < nrhs; ++q) x[q * M + r] = fma(-u, x[q * M + k], x[q * M + r]);
    }
    __syncthreads();
  }
  for (int q = 0; q < nrhs; ++q)
    for (int r = tid; r < M; r += 1024) C[((long long)sys * M + r) * nrhs + q] = x[q * M + r];
}
extern "C" size_t avsep_bss_solve_workspace_bytes(int32_t B, int32_t S, int32_t flen, int32_t mode) {
  if (B <= 0 || S <= 0 || flen <= 0) return 0;
  const size_t M = mode == 0 ? (size_t)S * flen : (size_t)flen, nsys = mode == 0 ? (size_t)B : (size_t)B * S;
  return nsys * M * M * sizeof(double);
}
extern "C" int avsep_bss_solve(const double* R, const double* D, int32_t B, int32_t S, int32_t E, int32_t flen, int32_t mode,
                               double* workspace, size_t workspace_bytes, double* C, int32_t* info, avsep_stream_t stream) {
  if (!R || !D || !workspace || !C || !info || B <= 0 || S <= 0 || S > 8 || E <= 0 || E > BS_MAXRHS || flen <= 0 || flen > 512 ||
      (mode != 0 && mode != 1) || (mode == 1 && E != S))
    return AVSEP_ERR_ARG;
  const int M = mode == 0 ? S * flen : flen, nrhs = mode == 0 ? E : 1, nsys = mode == 0 ? B : B * S;
  if (M > 1024 * BS_MAXROWS) return AVSEP_ERR_ARG;
  if (workspace_bytes < avsep_bss_solve_workspace_bytes(B, S, flen, mode)) return AVSEP_ERR_WORKSPACE;
  const size_t lds = sizeof(double) * (size_t)M * (1 + nrhs);
  if (lds > 150 * 1024) return AVSEP_ERR_ARG;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)bss_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return AVSEP_ERR_LAUNCH;
  hipLaunchKernelGGL(bss_solve_kernel, dim3(nsys), dim3(1024), lds, (hipStream_t)stream, R, D, S, E, flen, mode, workspace, C, info);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---- 3. projection -----------------------------------------------------------------------------------------------------------------
// mode 0: out[b][e][t] = sum_i sum_k C[b][(i,k)][e] * ref_i[t - k]      (C [B][S*flen][E])
// mode 1: out[b][j][t] = sum_k C[b*S + j][k] * ref_j[t - k]               (C [B*S][flen][1])
// for t in [0, L + flen - 1).  grid (chunks of 1024 outputs, E, B); coefficients and the reference windows in LDS.
__global__ __launch_bounds__(1024) void bss_project_kernel(const double* __restrict__ refs, const double* __restrict__ C, int S, int E, int L,
                                                           int flen, int mode, double* __restrict__ out) {
  extern __shared__ double bp_smem[];
  const int b = blockIdx.z, e = blockIdx.y, t0 = blockIdx.x * 1024, tid = threadIdx.x;
  const int ns = mode == 0 ? S : 1, Lp = L + flen - 1, WN = 1024 + flen - 1;
  double* const cf = bp_smem;                       // [ns][flen]
  double* const rw = bp_smem + ns * flen;           // [ns][WN]: ref[t0 - (flen-1) .. t0 + 1024)
  for (int x = tid; x < ns * flen; x += 1024) {
    const int i = x / flen, k = x % flen;
    cf[x] = mode == 0 ? C[(((long long)b * S + i) * flen + k) * E + e] : C[((long long)b * S + e) * flen + k];
  }
  for (int x = tid; x < ns * WN; x += 1024) {
    const int i = x / WN, t = t0 - (flen - 1) + x % WN, src = mode == 0 ? i : e;
    rw[x] = (t >= 0 && t < L) ? refs[((long long)b * S + src) * L + t] : 0.0;
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t >= Lp) return;
  double acc = 0.0;
  for (int i = 0; i < ns; ++i) {
    const double* rp = rw + i * WN + tid + (flen - 1);      // ref[t - k] = rp[-k]
    const double* cp = cf + i * flen;
#pragma unroll 8
    for (int k = 0; k < flen; ++k) acc = fma(cp[k], rp[-k], acc);
  }
  out[((long long)b * E + e) * Lp + t] = acc;
}
extern "C" int avsep_bss_project(const double* refs, const double* C, int32_t B, int32_t S, int32_t E, int32_t L, int32_t flen,
                                 int32_t mode, double* out, avsep_stream_t stream) {
  if (!refs || !C || !out || B <= 0 || B > 65535 || S <= 0 || S > 8 || E <= 0 || E > 8 || L <= 0 || flen <= 0 || flen > 512 ||
      (mode != 0 && mode != 1) || (mode == 1 && E != S))
    return AVSEP_ERR_ARG;
  const int ns = mode == 0 ? S : 1;
  const size_t lds = sizeof(double) * ((size_t)ns * flen + (size_t)ns * (1024 + flen - 1));
  if (lds > 150 * 1024) return AVSEP_ERR_ARG;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)bss_project_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return AVSEP_ERR_LAUNCH;
  hipLaunchKernelGGL(bss_project_kernel, dim3(cdiv(L + flen - 1, 1024), E, B), dim3(1024), lds, (hipStream_t)stream, refs, C, S, E, L,
                     flen, mode, out);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
