// CoLoc bottleneck fusion for C = 2..4 sources (BASELINE.json configs[4]: 3 sources): one workgroup per sample, one
// launch forward and one backward, like fusion.hip's two-source kernels.
//
// The reference hard-codes two sources (models/fusion_net.py:35,43-46: C = P = 2).  The generalisation is BUILD-DEFINED
// (restated on the CPU for the tests in oracle/nets.py:Fusion._coloc_n):
//   * Dc = D / C channels per audio block; the blocks are the FIRST C*Dc channels of the global max-pool of x, the
//     D - C*Dc remainder channels take no part and their output channels are zero;
//   * maps m[k][c][hw] = att(a_k, v_c[:, hw]) (cos: F.cosine_similarity eps 1e-8, fusion_net.py:27-29; sig:
//     sigmoid(dot / sqrt(Dc)), :31-32) for every audio block k and visual map c;
//   * all C! permutations in itertools (lexicographic) order: score_p = sum_c max_hw m[perm_p[c]][c], the FIRST maximum
//     wins (torch.sort(descending) / argmax keep the lowest index), match term = -score_best + sum of the others (:57-60);
//   * att_c = m[perm_best[c]][c] (:64), f_c[d] = max_hw v_c[d,hw] * att_c[hw] (:66-68).
// With C = 2 this is exactly fusion.hip's kind 0 (asserted by tests/test_gpu_ops.py::test_fusion_n_kernel).
#include "common.h"

#define FN_EPS 1e-8f
constexpr int FN_MAXC = 4;

struct FnArgs {
  const float* x;
  const float* v[FN_MAXC];
  int B, C, Dc, D, FT, HW, att;
};

// permutation p (lexicographic order = itertools.permutations(range(C))) -> out[0..C)
__device__ __forceinline__ void fn_perm(int p, int C, int* out) {
  int avail[FN_MAXC] = {0, 1, 2, 3};
  int f = 1;
  for (int i = 2; i < C; ++i) f *= i;            // (C-1)!
  for (int i = 0; i < C; ++i) {
    const int idx = p / f;
    p -= idx * f;
    out[i] = avail[idx];
    for (int j = idx; j + 1 < FN_MAXC; ++j) avail[j] = avail[j + 1];
    if (C - 1 - i > 0) f /= (C - 1 - i);
  }
}
__device__ __forceinline__ int fn_fact(int C) { return C == 2 ? 2 : (C == 3 ? 6 : 24); }

__device__ __forceinline__ void fn_wave_argmax(const float* s, int n, int lane, float& best, int& arg) {
  best = -INFINITY;
  arg = 0x7fffffff;
  for (int i = lane; i < n; i += 64) {
    const float v = s[i];
    if (v > best) { best = v; arg = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
}

// LDS layout shared by forward and backward
struct FnLds {
  float *a, *m, *nv, *na, *mx, *E, *S;
  int *arg, *perm;
};
__device__ __forceinline__ FnLds fn_lds(float* sm, int C, int D, int HW) {
  FnLds l;
  l.a = sm;                              // [D]
  l.m = l.a + D;                         // [C*C*HW]  m[(k*C + c)*HW + hw]
  l.nv = l.m + C * C * HW;               // [C*HW]    |v_c[:, hw]|
  l.E = l.nv + C * HW;                   // [C*C*HW]  (backward)
  l.na = l.E + C * C * HW;               // [C]
  l.mx = l.na + FN_MAXC;                 // [C*C]
  l.S = l.mx + FN_MAXC * FN_MAXC;        // [C]
  l.arg = (int*)(l.S + FN_MAXC);         // [C*C]
  l.perm = l.arg + FN_MAXC * FN_MAXC;    // [C] winning permutation, [C] = best index
  return l;
}
static size_t fn_smem(int C, int D, int HW) {
  return (size_t)(D + 2 * C * C * HW + C * HW + 2 * FN_MAXC + 2 * FN_MAXC * FN_MAXC + FN_MAXC + 1 + 16) * sizeof(float);
}

// maps + norms; s_a must be filled.  All threads call; ends with a barrier.
__device__ void fn_maps(const FnArgs& a, int b, const FnLds& l) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = a.C, Dc = a.Dc, HW = a.HW;
  const float inv_sqrt = 1.f / sqrtf((float)Dc);
  if (a.att == 0) {
    for (int k = wave; k < C; k += 4) {
      float q = 0.f;
      for (int d = lane; d < Dc; d += 64) q = fmaf(l.a[k * Dc + d], l.a[k * Dc + d], q);
      q = wave_sum(q);
      if (lane == 0) l.na[k] = sqrtf(q);
    }
    __syncthreads();
  }
  for (int i = tid; i < C * HW; i += 256) {
    const int c = i / HW, hw = i % HW;
    const float* vp = a.v[c] + (long long)b * Dc * HW + hw;
    float dot[FN_MAXC] = {0.f, 0.f, 0.f, 0.f}, nv = 0.f;
    for (int d = 0; d < Dc; ++d) {
      const float vv = vp[(long long)d * HW];
#pragma unroll
      for (int k = 0; k < FN_MAXC; ++k)
        if (k < C) dot[k] = fmaf(l.a[k * Dc + d], vv, dot[k]);
      nv = fmaf(vv, vv, nv);
    }
    nv = sqrtf(nv);
    l.nv[c * HW + hw] = nv;
#pragma unroll
    for (int k = 0; k < FN_MAXC; ++k)
      if (k < C)
        l.m[(k * C + c) * HW + hw] = a.att == 1 ? 1.f / (1.f + expf(-dot[k] * inv_sqrt))
                                                : dot[k] / (fmaxf(l.na[k], FN_EPS) * fmaxf(nv, FN_EPS));
  }
  __syncthreads();
  for (int q = wave; q < C * C; q += 4) {      // per-map max / argmax (first maximum)
    float mx;
    int am;
    fn_wave_argmax(l.m + q * HW, HW, lane, mx, am);
    if (lane == 0) { l.mx[q] = mx; l.arg[q] = am; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void fusion_n_av_fwd_kernel(FnArgs a, float* __restrict__ a_pool, int* __restrict__ pool_idx,
                                                              float* __restrict__ feat, int* __restrict__ sel_idx,
                                                              float* __restrict__ att_maps, float* __restrict__ match_part,
                                                              int* __restrict__ best_out) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, C = a.C, Dc = a.Dc, D = a.D, HW = a.HW, FT = a.FT, CD = C * Dc;
  const FnLds l = fn_lds(sm, C, D, HW);
  for (int d = tid; d < D; d += 256) {           // global max-pool over F x T with argmax (first maximum)
    const float* p = a.x + ((long long)b * D + d) * FT;
    float m = p[0];
    int am = 0;
    for (int i = 1; i < FT; ++i) {
      const float v = p[i];
      if (v > m) { m = v; am = i; }
    }
    l.a[d] = m;
    a_pool[(long long)b * D + d] = m;
    pool_idx[(long long)b * D + d] = am;
  }
  __syncthreads();
  fn_maps(a, b, l);
  if (tid == 0) {
    const int P = fn_fact(C);
    float sc[24], sbest = -INFINITY;
    int best = 0;
    for (int p = 0; p < P; ++p) {
      int pm[FN_MAXC];
      fn_perm(p, C, pm);
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += l.mx[pm[c] * C + c];
      sc[p] = s;
      if (s > sbest) { sbest = s; best = p; }
    }
    float others = 0.f;                            // summed without the winner: exactly p_other - p_best for C = 2
    for (int p = 0; p < P; ++p)
      if (p != best) others += sc[p];
    int pm[FN_MAXC];
    fn_perm(best, C, pm);
    for (int c = 0; c < C; ++c) l.perm[c] = pm[c];
    best_out[b] = best;
    match_part[b] = others - sbest;
  }
  __syncthreads();
  for (int i = tid; i < C * HW; i += 256) {
    const int c = i / HW, hw = i % HW;
    att_maps[((long long)b * C + c) * HW + hw] = l.m[(l.perm[c] * C + c) * HW + hw];
  }
  for (int i = tid; i < D; i += 256) {
    float f = 0.f;
    int arg = 0;
    if (i < CD) {
      const int c = i / Dc, d = i % Dc;
      const float* vp = a.v[c] + ((long long)b * Dc + d) * HW;
      const float* at = l.m + (l.perm[c] * C + c) * HW;
      f = vp[0] * at[0];
      for (int hw = 1; hw < HW; ++hw) {
        const float v = vp[hw] * at[hw];
        if (v > f) { f = v; arg = hw; }
      }
    }
    feat[(long long)b * D + i] = f;             // the D - C*Dc remainder channels are zero
    sel_idx[(long long)b * D + i] = arg;
  }
}

__global__ __launch_bounds__(256) void fusion_n_av_bwd_kernel(FnArgs a, const float* __restrict__ a_pool,
                                                              const int* __restrict__ pool_idx, const int* __restrict__ sel_idx,
                                                              const int* __restrict__ best_in, const float* __restrict__ dfeat,
                                                              const float* __restrict__ dmatch_ptr, float dmatch_scale,
                                                              float* __restrict__ dx, float* dv0, float* dv1, float* dv2,
                                                              float* dv3) {
  extern __shared__ float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, C = a.C, Dc = a.Dc, D = a.D, HW = a.HW, FT = a.FT, CD = C * Dc;
  const FnLds l = fn_lds(sm, C, D, HW);
  float* const dvs[FN_MAXC] = {dv0, dv1, dv2, dv3};
  for (int d = tid; d < D; d += 256) l.a[d] = a_pool[(long long)b * D + d];
  for (int i = tid; i < C * C * HW; i += 256) l.E[i] = 0.f;
  __syncthreads();
  fn_maps(a, b, l);
  const int best = best_in[b];
  const float dmatch = (dmatch_ptr ? dmatch_ptr[0] : 1.f) * dmatch_scale;
  if (tid == 0) {
    int pm[FN_MAXC];
    fn_perm(best, C, pm);
    for (int c = 0; c < C; ++c) l.perm[c] = pm[c];
  }
  // match term: score_p = sum_c max m[perm_p[c]][c]; coefficient -1 for the winner, +1 for the others
  if (tid < C * C) {
    const int k = tid / C, c = tid % C, P = fn_fact(C);
    float w = 0.f;
    for (int p = 0; p < P; ++p) {
      int pm[FN_MAXC];
      fn_perm(p, C, pm);
      if (pm[c] == k) w += (p == best ? -1.f : 1.f);
    }
    l.E[tid * HW + l.arg[tid]] = w * dmatch;       // one writer per map; the attended-vector terms are added after the barrier
  }
  __syncthreads();
  for (int i = tid; i < CD; i += 256) {          // f_c[d] = v_c[d, h*] * att_c[h*]
    const int c = i / Dc, d = i % Dc, h = sel_idx[(long long)b * D + i];
    const float vv = a.v[c][((long long)b * Dc + d) * HW + h];
    atomicAdd(&l.E[(l.perm[c] * C + c) * HW + h], dfeat[(long long)b * D + i] * vv);
  }
  __syncthreads();
  const float inv_sqrt = 1.f / sqrtf((float)Dc);
  if (a.att == 1) {
    for (int i = tid; i < C * C * HW; i += 256) {
      const float m = l.m[i];
      l.E[i] *= m * (1.f - m) * inv_sqrt;
    }
  } else {                                        // cos: S_k = sum_{c,hw} E * m
    const int lane = tid & 63, wave = tid >> 6;
    for (int k = wave; k < C; k += 4) {
      float q = 0.f;
      for (int i = lane; i < C * HW; i += 64) q = fmaf(l.E[k * C * HW + i], l.m[k * C * HW + i], q);
      q = wave_sum(q);
      if (lane == 0) l.S[k] = q;
    }
  }
  __syncthreads();
  for (int i = tid; i < CD; i += 256) {          // gradient to the audio vectors -> arg-max position of the global max-pool
    const int k = i / Dc, d = i % Dc;
    float g = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* vp = a.v[c] + ((long long)b * Dc + d) * HW;
      const float* E = l.E + (k * C + c) * HW;
      if (a.att == 1) {
        for (int hw = 0; hw < HW; ++hw) g = fmaf(E[hw], vp[hw], g);
      } else {
        const float* nv = l.nv + c * HW;
        for (int hw = 0; hw < HW; ++hw) g = fmaf(E[hw] / fmaxf(nv[hw], FN_EPS), vp[hw], g);
      }
    }
    if (a.att == 0) {
      const float na = l.na[k];
      g = g / fmaxf(na, FN_EPS);
      if (na > FN_EPS) g -= l.S[k] * l.a[i] / (na * na);
    }
    if (dx) dx[((long long)b * D + i) * FT + pool_idx[(long long)b * D + i]] += g;
  }
  for (int c = 0; c < C; ++c) {                  // gradient to the visual maps (every element written)
    float* dv = dvs[c];
    if (!dv) continue;
    const float* vsrc = a.v[c] + (long long)b * Dc * HW;
    const float* at = l.m + (l.perm[c] * C + c) * HW;
    for (int i = tid; i < Dc * HW; i += 256) {
      const int d = i / HW, hw = i % HW;
      float g = 0.f, em = 0.f;
      for (int k = 0; k < C; ++k) {
        const float e = l.E[(k * C + c) * HW + hw];
        if (a.att == 1) g = fmaf(e, l.a[k * Dc + d], g);
        else {
          g = fmaf(e, l.a[k * Dc + d] / fmaxf(l.na[k], FN_EPS), g);
          em = fmaf(e, l.m[(k * C + c) * HW + hw], em);
        }
      }
      if (a.att == 0) {
        const float nv = l.nv[c * HW + hw];
        g /= fmaxf(nv, FN_EPS);
        if (nv > FN_EPS) g -= em * vsrc[i] / (nv * nv);
      }
      if (hw == sel_idx[(long long)b * D + c * Dc + d]) g += dfeat[(long long)b * D + c * Dc + d] * at[hw];
      dv[(long long)b * Dc * HW + i] = g;
    }
  }
}

static int fn_check(int B, int C, int D, int FT, int HW, int att) {
  if (B <= 0 || C < 2 || C > FN_MAXC || D < C || FT <= 0 || HW <= 0 || (att != 0 && att != 1)) return AVSEP_ERR_ARG;
  if (fn_smem(C, D, HW) > 160 * 1024) return AVSEP_ERR_ARG;
  return AVSEP_OK;
}

extern "C" int avsep_fusion_n_av_fwd(const float* x, const float* const* v, int32_t B, int32_t C, int32_t D, int32_t FT,
                                     int32_t HW, int32_t att, float* a_pool, int32_t* pool_idx, float* feat, int32_t* sel_idx,
                                     float* att_maps, float* match_part, int32_t* best, avsep_stream_t stream) {
  if (!x || !v || !a_pool || !pool_idx || !feat || !sel_idx || !att_maps || !match_part || !best) return AVSEP_ERR_ARG;
  int rc = fn_check(B, C, D, FT, HW, att);
  if (rc) return rc;
  FnArgs a{};
  a.x = x; a.B = B; a.C = C; a.Dc = D / C; a.D = D; a.FT = FT; a.HW = HW; a.att = att;
  for (int c = 0; c < C; ++c) {
    if (!v[c]) return AVSEP_ERR_ARG;
    a.v[c] = v[c];
  }
  const size_t smem = fn_smem(C, D, HW);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)fusion_n_av_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(fusion_n_av_fwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, a, a_pool, pool_idx, feat, sel_idx,
                     att_maps, match_part, best);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

extern "C" int avsep_fusion_n_av_bwd(const float* x, const float* const* v, int32_t B, int32_t C, int32_t D, int32_t FT,
                                     int32_t HW, int32_t att, const float* a_pool, const int32_t* pool_idx, const int32_t* sel_idx,
                                     const int32_t* best, const float* dfeat, const float* dmatch, float dmatch_scale,
                                     float* dx_accum, float* const* dv, avsep_stream_t stream) {
  if (!x || !v || !a_pool || !pool_idx || !sel_idx || !best || !dfeat || !dv) return AVSEP_ERR_ARG;
  int rc = fn_check(B, C, D, FT, HW, att);
  if (rc) return rc;
  FnArgs a{};
  a.x = x; a.B = B; a.C = C; a.Dc = D / C; a.D = D; a.FT = FT; a.HW = HW; a.att = att;
  float* dvs[FN_MAXC] = {nullptr, nullptr, nullptr, nullptr};
  for (int c = 0; c < C; ++c) {
    if (!v[c]) return AVSEP_ERR_ARG;
    a.v[c] = v[c];
    dvs[c] = dv[c];
  }
  const size_t smem = fn_smem(C, D, HW);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)fusion_n_av_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(fusion_n_av_bwd_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, a, a_pool, pool_idx, sel_idx, best,
                     dfeat, dmatch, dmatch_scale, dx_accum, dvs[0], dvs[1], dvs[2], dvs[3]);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

// ---------------------------------------------------------------------------
// audio-only branch: the pooled blocks in a per-sample permutation (two sources: the reference's coin, fusion_net.py:93-104;
// more: a permutation index in itertools order)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_n_ao_fwd_kernel(const float* __restrict__ x, const int* __restrict__ draws, int C,
                                                              int Dc, int D, int FT, float* __restrict__ feat,
                                                              int* __restrict__ pool_idx) {
  const int b = blockIdx.x;
  extern __shared__ float s_a[];
  for (int d = threadIdx.x; d < D; d += 256) {
    const float* p = x + ((long long)b * D + d) * FT;
    float m = p[0];
    int am = 0;
    for (int i = 1; i < FT; ++i)
      if (p[i] > m) { m = p[i]; am = i; }
    s_a[d] = m;
    pool_idx[(long long)b * D + d] = am;
  }
  __syncthreads();
  int pm[FN_MAXC];
  fn_perm(draws[b], C, pm);
  for (int i = threadIdx.x; i < D; i += 256) {
    const int slot = i / Dc, d = i % Dc;
    feat[(long long)b * D + i] = i < C * Dc ? s_a[pm[slot] * Dc + d] : 0.f;
  }
}
__global__ __launch_bounds__(256) void fusion_n_ao_bwd_kernel(const int* __restrict__ draws, int C, int Dc, int D, int FT,
                                                              const int* __restrict__ pool_idx, const float* __restrict__ dfeat,
                                                              float* __restrict__ dx) {
  const int b = blockIdx.x;
  int pm[FN_MAXC];
  fn_perm(draws[b], C, pm);
  for (int i = threadIdx.x; i < C * Dc; i += 256) {      // i indexes the OUTPUT slot element; its source is block pm[slot]
    const int slot = i / Dc, d = i % Dc, src = pm[slot] * Dc + d;
    dx[((long long)b * D + src) * FT + pool_idx[(long long)b * D + src]] += dfeat[(long long)b * D + i];
  }
}

extern "C" int avsep_fusion_n_ao_fwd(const float* x, const int32_t* draws, int32_t B, int32_t C, int32_t D, int32_t FT, float* feat,
                                     int32_t* pool_idx, avsep_stream_t stream) {
  if (!x || !draws || !feat || !pool_idx || B <= 0 || C < 2 || C > FN_MAXC || D < C || D > 16384 || FT <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(fusion_n_ao_fwd_kernel, dim3(B), dim3(256), D * sizeof(float), (hipStream_t)stream, x, draws, C, D / C, D, FT,
                     feat, pool_idx);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
extern "C" int avsep_fusion_n_ao_bwd(const int32_t* draws, int32_t B, int32_t C, int32_t D, int32_t FT, const int32_t* pool_idx,
                                     const float* dfeat, float* dx_accum, avsep_stream_t stream) {
  if (!draws || !pool_idx || !dfeat || !dx_accum || B <= 0 || C < 2 || C > FN_MAXC || D < C || FT <= 0) return AVSEP_ERR_ARG;
  hipLaunchKernelGGL(fusion_n_ao_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, draws, C, D / C, D, FT, pool_idx, dfeat,
                     dx_accum);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
