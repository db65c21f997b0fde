// Shared device helpers for libavsep_gfx950 (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "avsep.h"

#define AVSEP_LAUNCH_CHECK()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return AVSEP_ERR_LAUNCH;            \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }
// The descriptor with the batch its launch heuristics are planned for (avsep_conv_desc.plan_n, 0 = N): kernel-family,
// tile-size and split-K DECISIONS are taken on this copy, grids and workspace sizes on the real batch.
static inline avsep_conv_desc plan_desc(const avsep_conv_desc* d) {
  avsep_conv_desc e = *d;
  if (d->plan_n > 0) e.N = d->plan_n;
  return e;
}
static inline int plan_batch(const avsep_conv_desc* d) { return d->plan_n > 0 ? d->plan_n : d->N; }
// compute units of the current device (256 on MI355X; 256 when no device is visible, e.g. the CPU-only build check)
static inline int cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == AVSEP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == AVSEP_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
  return v;
}
// branch-free form for the staging loops: act(v) = max(v, slope * v) with slope = 1 (none), 0 (ReLU), 0.2 (LeakyReLU).
// A run-time `act` inside act_apply() becomes two scalar branches PER ELEMENT, which serialises the staging code around
// them (measured on the bf16 kernels: 13 VALU + 2 branches per MFMA in the K loop).
__device__ __forceinline__ float act_slope(int act) {
  return act == AVSEP_ACT_RELU ? 0.f : (act == AVSEP_ACT_LRELU02 ? 0.2f : 1.f);
}
__device__ __forceinline__ float act_by_slope(float v, float slope) { return fmaxf(v, v * slope); }
static inline float act_slope_host(int act) { return act == AVSEP_ACT_RELU ? 0.f : (act == AVSEP_ACT_LRELU02 ? 0.2f : 1.f); }
// derivative of act at pre-activation value v
__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == AVSEP_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (act == AVSEP_ACT_LRELU02) return v > 0.f ? 1.f : 0.2f;
  return 1.f;
}

// sum over the 64 lanes of a wave (all lanes get the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum within each 32-lane half of the wave
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Same sum on the DPP path of the vector ALU (no LDS traffic, no ds_bpermute latency): four row rotations leave every lane
// of a 16-lane row with the row's total, a row broadcast then adds row 0 into row 1 (and row 2 into row 3).  The result
// is valid in lanes 16-31 and 48-63 ONLY: the caller lets lane (lane & 31) == 31 write it.
__device__ __forceinline__ float half_sum_hi(float v) {
#define AVSEP_DPP_ADD(ctrl, rmask)                                                                                     \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  AVSEP_DPP_ADD(0x128, 0xf);   // row_ror:8
  AVSEP_DPP_ADD(0x124, 0xf);   // row_ror:4
  AVSEP_DPP_ADD(0x122, 0xf);   // row_ror:2
  AVSEP_DPP_ADD(0x121, 0xf);   // row_ror:1
  AVSEP_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3 (rows 0 and 2 add the zero of `old`)
#undef AVSEP_DPP_ADD
  return v;
}

// Bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (id % 8 equal under
// round-robin dispatch) receive CONSECUTIVE logical ids, so tiles that re-read the same
// operand sit behind one L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, s = bid >> 3;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + s;
}
