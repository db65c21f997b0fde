// Weight gradient of a convolution over FEW input channels (Cin <= 4): the ResNet stem conv 7x7/s2/p3 over the RGB frames
// (torchvision resnet18 conv1 via models/vision_net.py:84-89) and the U-Net's first down conv 4x4/s2/p1 over the
// one-channel spectrogram (audio_net.py:57-58).  On the im2col kernel these ran at 38 and 5 TFLOP/s (2.4 + 0.4 ms of
// the batch-64 step): K = Cin*KH*KW is tiny, every operand element went through a gather.
//
//   dW[co][n] = sum_pix dY[co][pix] * X[pix][n],      n = (ci, kh, kw),  pix = (image, oh, ow)
//
// as a GEMM with M = 64 output channels per workgroup, N = Cin*KH*KW columns (padded to 32-column MFMA tiles) and
// K = pixels, on v_mfma_f32_32x32x2_f32 (exact f32).  A tile = ONE output row of one image: its dY row block
// [64][Wo] and the KH input rows of every channel it touches are staged once into LDS; the B operand of pixel ow and
// column n is X_lds[(ci*KH + kh)*XW + kw + S*ow], i.e. a per-lane constant plus a wave-uniform offset, so the K loop
// has no vector ALU work at all: per 2-pixel k-step a wave issues one A read, NTW B reads and NTW MFMAs.  Wave
// (mt, nh) owns M-tile mt (32 channels) and the nh-th half of the N tiles.  A workgroup walks tiles t, t + grid, ... with
// its accumulators in registers and writes one partial slab at the end (reduce_slabs, deterministic).
#include <stdlib.h>

#include "common.h"

struct ScwArgs {
  const float* x;
  const float* dy;
  float* part;         // [slabs][Cout][NC]
  int N, H, W, Cout, Ho, Wo, ntiles;
  int DS, XW;          // LDS row strides of the dY block and of the input rows (both odd, host-computed)
};

template <int CIN, int KH, int KW, int S, int PAD>
__global__ __launch_bounds__(256, 2) void smallci_wgrad_kernel(ScwArgs a) {
  constexpr int NC = CIN * KH * KW, NT = (NC + 31) / 32, NTW = (NT + 1) / 2;
  extern __shared__ float sm[];
  const int DS = a.DS;                             // dY row stride: odd, so that the 32 channel rows of an A read hit 32 banks
  const int XW = a.XW;                             // X row stride: odd, >= PAD + W and > the last column a B read touches
  float* const Ds = sm;                            // [64][DS]
  float* const Xs = sm + 64 * DS;                  // [CIN*KH][XW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int mt = wave & 1, nh = wave >> 1, co0 = blockIdx.y * 64;
  f32x16 acc[NTW];
#pragma unroll
  for (int q = 0; q < NTW; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  // per-lane B base of the N tiles this wave owns (a column past NC reads column 0: its products are never stored)
  int bbase[NTW];
#pragma unroll
  for (int q = 0; q < NTW; ++q) {
    int n = (nh * NTW + q) * 32 + li;
    if (n >= NC) n = 0;
    const int ci = n / (KH * KW), kh = (n / KW) % KH, kw = n % KW;
    bbase[q] = (ci * KH + kh) * XW + kw + S * lk;
  }
  const int abase = (mt * 32 + li) * DS + lk;
  const int W4 = a.W >> 2, Wo4 = a.Wo >> 2;        // host-checked: W % 4 == 0, Wo % 4 == 0, Wo <= 128, W <= 256
  // Staging is split in two halves so that the global loads of tile t + 1 are IN FLIGHT during the MFMA loop of tile t (they
  // were consumed right after their issue before: 12 exposed memory round trips per tile, longer than the tile's MFMAs):
  //   gload: dY rows of the 64 channels + the KH input rows of every channel -> registers (zero where outside the tensors)
  //   lstore: registers -> LDS (scalar stores: odd row strides); LDS column j of an input row = input column j - PAD
  constexpr int ND = 8, NX = (CIN * KH * 64 + 255) / 256;
  f32x4 rd[ND], rx[NX];
  auto gload = [&](int t) __attribute__((always_inline)) {
    const int n = t / a.Ho, oh = t % a.Ho;
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int i = tid + 256 * e, c = i / Wo4, q = i % Wo4;
      rd[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 64 * Wo4 && co0 + c < a.Cout)
        rd[e] = *reinterpret_cast<const f32x4*>(a.dy + (((long long)n * a.Cout + co0 + c) * a.Ho + oh) * a.Wo + 4 * q);
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int i = tid + 256 * e, row = i / W4, q = i % W4, ci = row / KH, kh = row % KH, ih = S * oh - PAD + kh;
      rx[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < CIN * KH * W4 && ih >= 0 && ih < a.H)
        rx[e] = *reinterpret_cast<const f32x4*>(a.x + (((long long)n * CIN + ci) * a.H + ih) * a.W + 4 * q);
    }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int i = tid + 256 * e, c = i / Wo4, q = i % Wo4;
      if (i < 64 * Wo4) {
        float* d = Ds + c * DS + 4 * q;
        d[0] = rd[e].x; d[1] = rd[e].y; d[2] = rd[e].z; d[3] = rd[e].w;
      }
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int i = tid + 256 * e, row = i / W4, q = i % W4;
      if (i < CIN * KH * W4) {
        float* d = Xs + row * XW + PAD + 4 * q;
        d[0] = rx[e].x; d[1] = rx[e].y; d[2] = rx[e].z; d[3] = rx[e].w;
      }
    }
  };
  for (int i = tid; i < CIN * KH * (XW - a.W); i += 256) {        // the zero borders (PAD columns left, the rest right): once
    const int row = i / (XW - a.W), j = i % (XW - a.W);
    Xs[row * XW + (j < PAD ? j : a.W + j)] = 0.f;
  }
  if ((int)blockIdx.x < a.ntiles) gload(blockIdx.x);
  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    __syncthreads();                               // the previous tile's operand reads are done
    lstore();
    __syncthreads();
    if (t + (int)gridDim.x < a.ntiles) gload(t + gridDim.x);
    // k-step = output pixels (p, p + 1); the operands of step p + 2 are read while the MFMAs of step p run (the reads of the
    // last trip fall past the row ends: inside the allocation, never used).  (Odd tile count: the second half's last tile
    // is a dummy, never stored.)
    const float* Ap = Ds + abase;
    float av = Ap[0], bv[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) bv[q] = Xs[bbase[q]];
    for (int p = 0; p < a.Wo; p += 2) {
      float an = 0.f, bn[NTW];
#pragma unroll
      for (int q = 0; q < NTW; ++q) {              // every LDS read sits behind one MFMA (the scheduler would group them otherwise)
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[q], acc[q], 0, 0, 0);
        if (q == 0) an = Ap[p + 2];
        bn[q] = Xs[bbase[q] + S * (p + 2)];
        __builtin_amdgcn_sched_barrier(0);
      }
      av = an;
#pragma unroll
      for (int q = 0; q < NTW; ++q) bv[q] = bn[q];
    }
  }
  // partial slab of this workgroup: C/D map row = (r & 3) + 8 * (r >> 2) + 4 * lk, column = li
  float* out = a.part + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * 64 * NC;
#pragma unroll
  for (int q = 0; q < NTW; ++q) {
    const int n = (nh * NTW + q) * 32 + li;
    if (n >= NC || (nh * NTW + q) >= NT) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      out[(long long)row * NC + n] = acc[q][r];
    }
  }
}

int reduce_slabs_strided(const float* ws, float* out, long long n, int S, long long stride, hipStream_t st);   // conv.hip

// ---------------------------------------------------------------------------
// host side (called from conv.hip)
// ---------------------------------------------------------------------------
static int scw_class(const avsep_conv_desc* d) {
  if (d->up2x || d->dil != 1 || d->C0 != d->Cin || d->scale0 || d->act0 != AVSEP_ACT_NONE) return 0;
  if (d->Cin == 3 && d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3) return 1;      // ResNet stem
  if (d->Cin == 1 && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1) return 2;      // U-Net first down conv
  return 0;
}
static int scw_ds(const avsep_conv_desc* d) { return d->Wo | 1; }
static int scw_xw(const avsep_conv_desc* d) {
  const int reach = d->stride * (d->Wo - 1) + d->KW, stored = d->pad + d->W;      // last column read + 1, last column stored + 1
  return ((reach > stored ? reach : stored) + 2) | 1;
}
static size_t scw_smem(const avsep_conv_desc* d) {
  return (size_t)(64 * scw_ds(d) + d->Cin * d->KH * scw_xw(d) + 4 * d->stride + 8) * sizeof(float);   // + the prefetch overshoot
}
static int scw_slabs(const avsep_conv_desc* d) {
  const long long tiles = (long long)d->N * d->Ho;
  const int want = 2 * cu_count() / cdiv(d->Cout, 64);            // two workgroups per CU
  return (int)(tiles < want ? tiles : (want < 1 ? 1 : want));
}
bool scw_applicable(const avsep_conv_desc* d) {
  if ((d->algo & AVSEP_ALGO_NO_SMALLCI_WGRAD) || !scw_class(d)) return false;
  if ((d->W & 3) || (d->Wo & 3) || d->Wo < 8 || d->Wo > 128 || d->W > 256) return false;      // register-staged rows: ND / NX pieces per thread
  if ((long long)d->N * d->Ho > 0x7fffffffLL || d->Cout > 65535 * 64) return false;
  return scw_smem(d) <= 72 * 1024;                                 // two workgroups per CU
}
size_t scw_workspace_floats(const avsep_conv_desc* d) {
  return (size_t)scw_slabs(d) * cdiv(d->Cout, 64) * 64 * d->Cin * d->KH * d->KW;
}
int scw_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  ScwArgs a{};
  a.x = d->x0; a.dy = dy; a.part = ws;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ntiles = d->N * d->Ho;
  a.DS = scw_ds(d); a.XW = scw_xw(d);
  const int slabs = scw_slabs(d), gridM = cdiv(d->Cout, 64);
  const size_t smem = scw_smem(d);
  dim3 grid(slabs, gridM);
  if (scw_class(d) == 1) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)smallci_wgrad_kernel<3, 7, 7, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((smallci_wgrad_kernel<3, 7, 7, 2, 3>), grid, dim3(256), smem, st, a);
  } else {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)smallci_wgrad_kernel<1, 4, 4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((smallci_wgrad_kernel<1, 4, 4, 2, 1>), grid, dim3(256), smem, st, a);
  }
  AVSEP_LAUNCH_CHECK();
  // a slab = [gridM * 64 rows][NC]: its first Cout * NC elements are dw's layout; summed over the workgroups in a fixed order
  const long long NC = (long long)d->Cin * d->KH * d->KW;
  return reduce_slabs_strided(ws, dw, d->Cout * NC, slabs, (long long)gridM * 64 * NC, st);
}
