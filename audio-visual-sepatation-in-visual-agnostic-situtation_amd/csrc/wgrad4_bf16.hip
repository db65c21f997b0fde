// Weight gradient of the 4x4 / stride 2 / pad 1 convolutions (the U-Net encoder, models/audio_net.py:57-58,170-171) with bf16
// operands and fp32 accumulation (avsep_conv_desc.prec == AVSEP_PREC_BF16; fp32: the im2col kernel of conv.hip):
//     dW[co][ci][kh][kw] = sum_{n,oh,ow} dY[n][co][oh][ow] * X[n][ci][2*oh + kh - 1][2*ow + kw - 1]
// Same GEMM as wgrad_bf16.hip: M = co (128 per workgroup), N = (tap, ci) (16 taps x 32 channels), K = OUTPUT pixels, a
// k-step = 16 consecutive output columns of one output row.  The stride-2 input columns of a k-step are made
// contiguous by de-interleaving the staged input rows into an EVEN plane E[i] = X[2i] and an ODD plane O[i] = X[2i+1]
// (columns relative to 2*ow0):  kw = 0 -> O[m-1], kw = 1 -> E[m], kw = 2 -> O[m], kw = 3 -> E[m+1]  for output column m,
// so two fragments per plane are one aligned ds_read_b128 and the other two are that read plus one neighbouring dword,
// assembled with 4 v_alignbit_b32 each.  The folded BatchNorm affine + LeakyReLU of the level below are applied while the
// planes are staged (zero padding AFTER the activation).  4 waves, wave = 32 co x 32 ci x 16 taps (256 accumulator
// registers, one wave per SIMD), tiles double-buffered in LDS, tap-major partial slabs + a deterministic fp32 reduce.
#include <stdlib.h>

#include "halo_bf16.h"

struct W4Args {
  int N, Cin, H, W, Cout, Ho, Wo;
  int act0;
  const float *x0, *sc0, *sh0;
  const float* dy;
  float* out;
  int tilesX, tilesY, gridM, gridC, tiles_per_split;
};

constexpr int W4_BM = 128, W4_BC = 32;

// KS = 4: the U-Net encoder convs; KS = 3: the 3x3 / stride 2 / pad 1 convs of ResNet layer2.0 / layer3.0 (same planes,
// taps kw = 0..2 and kh = 0..2 only).
template <int TH, int TW, int KS = 4>
__global__ __launch_bounds__(256) void wgrad4bf_kernel(W4Args a) {
  constexpr int NT = 256, NTAP = KS * KS;
  constexpr int NPIX = TH * TW, PR = 2 * TH + KS - 2;                   // output pixels / input rows of a tile
  constexpr int A_ROW = NPIX * 2 + 16;                                  // bytes per co row: 16 x odd
  constexpr int PL_EL = TW + 16, PLB = PL_EL * 2, ROWB = 2 * PLB;       // plane: 8 | TW | 8 elements; row = E plane, O plane
  constexpr int CH_RAW = PR * ROWB;
  constexpr int CH = (CH_RAW / 16) % 2 == 1 ? CH_RAW : CH_RAW + 16;     // bytes per ci: 16 x odd (lanes = ci)
  constexpr int A_BYTES = W4_BM * A_ROW, B_BYTES = W4_BC * CH;
  constexpr int AQ = W4_BM * NPIX / 4, AE = AQ / NT;
  constexpr int NG = PL_EL / 8, BU = W4_BC * PR * NG, BE = (BU + NT - 1) / NT;    // units: 16 input columns -> 8 + 8 plane elements
  static_assert(AQ % NT == 0 && TW % 16 == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int mt = blockIdx.x % a.gridM, ct = blockIdx.x / a.gridM, split = blockIdx.y;
  const int m0 = mt * W4_BM, c0 = ct * W4_BC;
  const int tiles_img = a.tilesX * a.tilesY, tiles_all = tiles_img * a.N;
  const int t_begin = split * a.tiles_per_split, t_end = min(tiles_all, t_begin + a.tiles_per_split);
  const long long HW = (long long)a.H * a.W, HoWo = (long long)a.Ho * a.Wo;
  const bool has_aff = a.sc0 != nullptr;
  const float slope = act_slope(a.act0);

  f32x16 acc[NTAP];
#pragma unroll
  for (int j = 0; j < NTAP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // dY quad e: LDS byte offset (bits 0-16) | tile row (17-20) | tile column (21-26) | co ok (31)
  int a_goff[AE];
  unsigned a_pk[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    const int idx = tid + NT * e;
    const int q = idx % (TW / 4), r = (idx / (TW / 4)) % TH, co = idx / (NPIX / 4);
    a_goff[e] = min(m0 + co, a.Cout - 1) * (int)HoWo + r * a.Wo + 4 * q;                 // Cout*Ho*Wo < 2^31: host check
    a_pk[e] = (unsigned)(co * A_ROW + (r * TW + 4 * q) * 2) | (unsigned)r << 17 | (unsigned)(4 * q) << 21 |
              (m0 + co < a.Cout ? 0x80000000u : 0u);
  }
  // input unit e: (ci, input row pr, group g) = input columns 2*ow0 + 16*(g-1) .. +15 -> elements 8g .. 8g+7 of both planes
  int b_goff[BE];
  unsigned b_pk[BE];
  float b_sc[BE], b_sh[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int idx = min(tid + NT * e, BU - 1);
    const int g = idx % NG, pr = (idx / NG) % PR, cc = idx / (NG * PR);
    const bool chok = (BE * NT == BU || tid + NT * e < BU) && c0 + cc < a.Cin;
    const int cs = min(c0 + cc, a.Cin - 1);
    b_goff[e] = cs * (int)HW + (pr - 1) * a.W + 16 * (g - 1);                              // Cin*H*W < 2^31: host check
    b_pk[e] = (unsigned)(cc * CH + pr * ROWB + g * 16) | (unsigned)pr << 17 | (unsigned)g << 21 | (chok ? 0x80000000u : 0u);
    b_sc[e] = has_aff ? a.sc0[cs] : 1.f;
    b_sh[e] = has_aff ? a.sh0[cs] : 0.f;
  }
  f32x4 areg[AE];
  f32x4 breg[BE][4];
  unsigned amask = 0;
  unsigned bmask[BE];              // per unit: 4 quad-valid bits (W % 4 == 0: quads are all-in or all-out)

  auto issue = [&](int t) __attribute__((always_inline)) {
    const int n = t / tiles_img, tt = t % tiles_img, oh0 = (tt / a.tilesX) * TH, ow0 = (tt % a.tilesX) * TW;
    const float* dyb = a.dy + (long long)n * a.Cout * HoWo + (long long)oh0 * a.Wo + ow0;
    const float* xb = a.x0 + (long long)n * a.Cin * HW + (long long)(2 * oh0) * a.W + 2 * ow0;
    amask = 0;
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      unsigned pk = a_pk[e];
      asm volatile("" : "+v"(pk));
      const int ar = (pk >> 17) & 15, ac = (pk >> 21) & 63;
      const bool ok = (pk >> 31) && oh0 + ar < a.Ho && ow0 + ac < a.Wo;
      areg[e] = *reinterpret_cast<const f32x4*>(ok ? dyb + a_goff[e] : a.dy);
      amask |= (unsigned)ok << e;
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      unsigned pk = b_pk[e];
      asm volatile("" : "+v"(pk));
      const int ih = 2 * oh0 - 1 + (int)((pk >> 17) & 15), col = 2 * ow0 + 16 * ((int)((pk >> 21) & 15) - 1);
      const bool rok = (pk >> 31) && (unsigned)ih < (unsigned)a.H;
      const float* src = xb + b_goff[e];
      unsigned m = 0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bool ok = rok && (unsigned)(col + 4 * p) < (unsigned)a.W;
        breg[e][p] = *reinterpret_cast<const f32x4*>(ok ? src + 4 * p : a.x0);
        m |= (unsigned)ok << p;
      }
      bmask[e] = m;
    }
  };
  auto finish = [&](int buf) __attribute__((always_inline)) {
    unsigned char* Ab = smem + buf * (A_BYTES + B_BYTES);
    unsigned char* Bb = Ab + A_BYTES;
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      unsigned pk = a_pk[e];
      asm volatile("" : "+v"(pk));
      const bool ok = (amask >> e) & 1u;
      const f32x4 v = areg[e];
      uint2 o;
      o.x = ok ? bf_pack2(v.x, v.y) : 0u;
      o.y = ok ? bf_pack2(v.z, v.w) : 0u;
      *reinterpret_cast<uint2*>(Ab + (pk & 0x1ffffu)) = o;
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      unsigned pk = b_pk[e];
      asm volatile("" : "+v"(pk));
      float ev[8], od[8];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bool ok = (bmask[e] >> p) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = act_by_slope(fmaf(breg[e][p][j], b_sc[e], b_sh[e]), slope);
          x = ok ? x : 0.f;                                  // zero padding of the ACTIVATED tensor
          if (j & 1) od[2 * p + (j >> 1)] = x;
          else ev[2 * p + (j >> 1)] = x;
        }
      }
      if (BE * NT == BU || tid + NT * e < BU) {
        u32x4 oe = {bf_pack2(ev[0], ev[1]), bf_pack2(ev[2], ev[3]), bf_pack2(ev[4], ev[5]), bf_pack2(ev[6], ev[7])};
        u32x4 oo = {bf_pack2(od[0], od[1]), bf_pack2(od[2], od[3]), bf_pack2(od[4], od[5]), bf_pack2(od[6], od[7])};
        *reinterpret_cast<u32x4*>(Bb + (pk & 0x1ffffu)) = oe;
        *reinterpret_cast<u32x4*>(Bb + (pk & 0x1ffffu) + PLB) = oo;
      }
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    finish(0);
  }
  __syncthreads();
  const int a_lane = (wave * 32 + li) * A_ROW + lk * 16;
  const int b_lane = li * CH + lk * 16;
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    if (t + 1 < t_end) issue(t + 1);
    const unsigned char* Ap = smem + buf * (A_BYTES + B_BYTES) + a_lane;
    const unsigned char* Bp = Ap - a_lane + A_BYTES + b_lane;
#pragma unroll
    for (int r = 0; r < TH; ++r) {
#pragma unroll
      for (int q = 0; q < TW / 16; ++q) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ap + (r * TW + 16 * q) * 2);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          // plane elements around m = 16q + 8*lk of input row 2r + kh; interior plane index i sits at element 8 + i
          const unsigned char* row = Bp + (2 * r + kh) * ROWB + (16 * q) * 2;
          const u32x4 e_mid = *reinterpret_cast<const u32x4*>(row + 16);            // E[m .. m+7]        (kw = 1)
          const uint2 e_hi = *reinterpret_cast<const uint2*>(row + 32);             // E[m+8], E[m+9]
          const uint2 o_lo = *reinterpret_cast<const uint2*>(row + PLB + 8);        // O[m-4 .. m-1]
          const u32x4 o_mid = *reinterpret_cast<const u32x4*>(row + PLB + 16);      // O[m .. m+7]        (kw = 2)
          const u32x4 f3 = {__builtin_amdgcn_alignbit(e_mid.y, e_mid.x, 16), __builtin_amdgcn_alignbit(e_mid.z, e_mid.y, 16),
                            __builtin_amdgcn_alignbit(e_mid.w, e_mid.z, 16), __builtin_amdgcn_alignbit(e_hi.x, e_mid.w, 16)};   // E[m+1 ..]
          const u32x4 f0 = {__builtin_amdgcn_alignbit(o_mid.x, o_lo.y, 16), __builtin_amdgcn_alignbit(o_mid.y, o_mid.x, 16),
                            __builtin_amdgcn_alignbit(o_mid.z, o_mid.y, 16), __builtin_amdgcn_alignbit(o_mid.w, o_mid.z, 16)};  // O[m-1 ..]
          acc[kh * KS + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f0), acc[kh * KS + 0], 0, 0, 0);
          acc[kh * KS + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, e_mid), acc[kh * KS + 1], 0, 0, 0);
          acc[kh * KS + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, o_mid), acc[kh * KS + 2], 0, 0, 0);
          if constexpr (KS == 4)
            acc[kh * KS + 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f3), acc[kh * KS + 3], 0, 0, 0);
        }
      }
    }
    if (t + 1 < t_end) finish(buf ^ 1);
    __syncthreads();
  }
  const int ci = c0 + li;
#pragma unroll
  for (int j = 0; j < NTAP; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (co < a.Cout && ci < a.Cin) a.out[(((long long)split * NTAP + j) * a.Cout + co) * a.Cin + ci] = acc[j][r];
    }
  }
}

// dw[cc][tap] = sum_z ws[z][tap][cc], cc = co*Cin + ci (16 taps)
__global__ void w4_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long P, int S) {
  const long long cc = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cc >= P) return;
  float s[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll 2
  for (int z = 0; z < S; ++z)
#pragma unroll
    for (int t = 0; t < 16; ++t) s[t] += ws[((long long)z * 16 + t) * P + cc];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    reinterpret_cast<float4*>(out + cc * 16)[t] = float4{s[4 * t], s[4 * t + 1], s[4 * t + 2], s[4 * t + 3]};
}

int w4_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st) {
  hipLaunchKernelGGL(w4_reduce_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, ws, dw, P, splits);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

static inline bool w4_enabled() {
  static const bool on = getenv("AVSEP_NO_BF16_KERNELS") == nullptr && getenv("AVSEP_NO_BF16_WGRAD") == nullptr;
  return on;
}

bool w4b_applicable(const avsep_conv_desc* d) {
  if (d->prec != AVSEP_PREC_BF16 || !w4_enabled()) return false;
  const bool k4 = d->KH == 4 && d->KW == 4, k3 = d->KH == 3 && d->KW == 3;
  if (!((k4 || k3) && d->stride == 2 && d->pad == 1 && d->dil == 1) || d->up2x || d->C0 != d->Cin) return false;
  return d->Wo >= 16 && d->Ho >= 2 && (d->W & 3) == 0 && (d->Wo & 3) == 0 && d->Cout >= 32 && d->Cin >= 32 && d->N <= 65535 &&
         (long long)d->Cout * d->Ho * d->Wo < 0x7fffffffLL && (long long)d->Cin * d->H * d->W < 0x7fffffffLL;
}
struct W4Plan { int tilesX, tilesY, gridM, gridC, splits, tps; bool wide; };
static W4Plan w4_plan(const avsep_conv_desc* d) {
  W4Plan p;
  p.wide = d->Wo > 16;
  p.tilesX = cdiv(d->Wo, p.wide ? 32 : 16);
  p.tilesY = cdiv(d->Ho, p.wide ? 2 : 4);
  p.gridM = cdiv(d->Cout, W4_BM);
  p.gridC = cdiv(d->Cin, W4_BC);
  const long long tiles = (long long)p.tilesX * p.tilesY * d->N;
  const int want = cdiv(768, p.gridM * p.gridC);
  const long long maxs = tiles / 4 > 0 ? tiles / 4 : 1;
  int splits = (int)(want < maxs ? want : maxs);
  if (splits < 1) splits = 1;
  p.tps = (int)((tiles + splits - 1) / splits);
  p.splits = (int)((tiles + p.tps - 1) / p.tps);
  return p;
}
size_t w4b_workspace_floats(const avsep_conv_desc* d) { return (size_t)w4_plan(d).splits * d->Cout * d->Cin * d->KH * d->KW; }
int w3_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st);   // conv3x3.hip (9 taps)

int w4b_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  W4Plan p = w4_plan(d);
  W4Args a{};
  a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo;
  a.act0 = d->act0; a.x0 = d->x0; a.sc0 = d->scale0; a.sh0 = d->shift0;
  a.dy = dy; a.out = ws;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.gridM = p.gridM; a.gridC = p.gridC; a.tiles_per_split = p.tps;
  dim3 grid(p.gridM * p.gridC, p.splits);
  const long long P = (long long)d->Cout * d->Cin;
  if (d->KH == 3) {
    if (p.wide) hipLaunchKernelGGL((wgrad4bf_kernel<2, 32, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad4bf_kernel<4, 16, 3>), grid, dim3(256), 0, st, a);
    AVSEP_LAUNCH_CHECK();
    return w3_reduce(ws, dw, P, p.splits, st);
  }
  if (p.wide) hipLaunchKernelGGL((wgrad4bf_kernel<2, 32>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad4bf_kernel<4, 16>), grid, dim3(256), 0, st, a);
  AVSEP_LAUNCH_CHECK();
  hipLaunchKernelGGL(w4_reduce_kernel, dim3(cdiv(P, 256)), dim3(256), 0, st, ws, dw, P, p.splits);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
