// Weight gradient of the 3x3 / stride 1 / pad = dil convolutions with bf16 operands and fp32 accumulation
// (avsep_conv_desc.prec == AVSEP_PREC_BF16; the fp32 counterpart is wgrad3x3_kernel in conv3x3.hip):
//     dW[co][ci][kh][kw] = sum_{n,h,w} dY[n][co][h][w] * X[n][ci][h + (kh-1)*dil][w + (kw-1)*dil]
// (U-Net decoder convs, models/audio_net.py:75-76,85-87,96-98,180-182; ResNet BasicBlock convs, models/vision_net.py:84-92).
// GEMM: M = co (128 per workgroup), N = (tap, ci) (9 taps x 32 input channels per workgroup), K = pixels.
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE k per lane: a k-step is 16 consecutive pixels of one image row
// (lane half h takes columns 8h .. 8h+7), which is contiguous in NCHW for both operands:
//   A = dY tile  [co][TH*TW pixels] bf16 in LDS (row stride 16 B x odd: conflict-free ds_read_b128 with lanes = co);
//   B = X patch  [ci][TH+2*dil rows][8 | TW | 8 columns] bf16 with the folded BatchNorm affine + ReLU applied while it
//       is staged; interior column w sits at element 8+w so that staging writes whole 16-byte groups.  Tap kw needs
//       columns w + (kw-1)*dil .. +7, i.e. elements 8 + w + (kw-1)*dil ..: ONE aligned ds_read_b128 plus the two
//       neighbouring dwords give all three kw fragments of a row — for dil = 1 the kw = 0 / 2 fragments straddle dwords
//       and are assembled with 4 v_alignbit_b32 each (VALU has idle issue slots beside the bf16 MFMA), for dil = 2
//       every fragment is a register selection.
// 4 waves, one per SIMD: wave wr owns 32 co x 32 ci x 9 taps = 144 accumulator registers; with the register-staged
// tile of the software pipeline that is ~340 registers per lane, which only a one-wave-per-SIMD workgroup can hold
// (512-thread workgroups are capped at 256 and spilled 330 bytes per lane).  Pixel tiles are double-buffered in
// LDS and software-pipelined through registers (issue(t+1) -> MFMAs(t) -> finish(t+1)), one barrier per tile; a
// workgroup sweeps `tiles_per_split` tiles and writes one tap-major partial slab [split][tap][Cout][Cin]; the fp32
// reduce kernel of conv3x3.hip sums the slabs deterministically.
#include <stdlib.h>

#include "halo_bf16.h"

struct WBArgs {
  int N, Cin, H, W, Cout;
  int act0;
  const float *x0, *sc0, *sh0;
  const float* dy;
  float* out;
  int tilesX, tilesY, gridM, gridC, tiles_per_split;
};

constexpr int WB_BM = 128, WB_BC = 32;

// RAW: the input needs no affine / activation (the U-Net decoder's materialised ReLU+upsample tensor): these kernels
// are VALU-bound in their staging (SQ counters: 15-19 VALU per MFMA before this), so that work is compiled out.
template <int TH, int TW, int DIL, bool A2, bool RAW = false>
__global__ __launch_bounds__(256) void wgradbf_kernel(WBArgs a) {
  constexpr int NT = 256;
  constexpr int NPIX = TH * TW, PH = TH + 2 * DIL;
  constexpr int A_ROW = NPIX * 2 + 16;                                  // bytes per co row: 16 x odd
  constexpr int ROW_EL = TW + 16, ROWB = ROW_EL * 2;                    // patch row: 8 | TW | 8 elements
  constexpr int CH_RAW = PH * ROWB;
  constexpr int CH = (CH_RAW / 16) % 2 == 1 ? CH_RAW : CH_RAW + 16;     // bytes per ci: 16 x odd (lanes = ci)
  constexpr int A_BYTES = WB_BM * A_ROW, B_BYTES = WB_BC * CH;
  constexpr int AQ = WB_BM * NPIX / 4, AE = AQ / NT;                    // dY quads per thread
  constexpr int NG = TW / 8 + 2, BU = WB_BC * PH * NG, BE = (BU + NT - 1) / NT;   // 8-element groups of the patch
  static_assert(AQ % NT == 0 && TW % 16 == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int wr = wave, wc = 0;
  const int mt = blockIdx.x % a.gridM, ct = blockIdx.x / a.gridM, split = blockIdx.y;
  const int m0 = mt * WB_BM, c0 = ct * WB_BC;
  const int tiles_img = a.tilesX * a.tilesY, tiles_all = tiles_img * a.N;
  const int t_begin = split * a.tiles_per_split, t_end = min(tiles_all, t_begin + a.tiles_per_split);
  const long long HW = (long long)a.H * a.W;
  const bool has_aff = !RAW && a.sc0 != nullptr;
  const float slope = act_slope(a.act0);

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- staging state (decoded once; per tile only (n, h0, w0) change) ------------------------------------------------
  // dY quad e: (co, tile row, column quad) -> packed word: LDS byte offset (bits 0-16) | row (17-20) | col (21-26) | co ok (31)
  int a_goff[AE];
  unsigned a_pk[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    const int idx = tid + NT * e;
    const int q = idx % (TW / 4), r = (idx / (TW / 4)) % TH, co = idx / (NPIX / 4);
    a_goff[e] = min(m0 + co, a.Cout - 1) * (int)HW + r * a.W + 4 * q;                    // Cout*H*W < 2^31: host check
    a_pk[e] = (unsigned)(co * A_ROW + (r * TW + 4 * q) * 2) | (unsigned)r << 17 | (unsigned)(4 * q) << 21 |
              (m0 + co < a.Cout ? 0x80000000u : 0u);
  }
  // patch group e: (ci, patch row, group g) -> LDS byte offset | row | group | channel ok
  int b_goff[BE];
  unsigned b_pk[BE];
  float b_sc[BE], b_sh[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int idx = min(tid + NT * e, BU - 1);
    const int g = idx % NG, pr = (idx / NG) % PH, cc = idx / (NG * PH);
    const bool chok = (BE * NT == BU || tid + NT * e < BU) && c0 + cc < a.Cin;
    const int cs = min(c0 + cc, a.Cin - 1);
    b_goff[e] = cs * (int)HW + (pr - DIL) * a.W + (8 * g - 8);                            // Cin*H*W < 2^31: host check
    b_pk[e] = (unsigned)(cc * CH + pr * ROWB + g * 16) | (unsigned)pr << 17 | (unsigned)g << 21 | (chok ? 0x80000000u : 0u);
    b_sc[e] = has_aff ? a.sc0[cs] : 1.f;
    b_sh[e] = has_aff ? a.sh0[cs] : 0.f;
  }
  f32x4 areg[AE];
  f32x4 breg[BE][2];
  unsigned amask = 0;
  unsigned bmask[BE];              // per group: 8 element-valid bits

  auto issue = [&](int t) __attribute__((always_inline)) {
    const int n = t / tiles_img, tt = t % tiles_img, h0 = (tt / a.tilesX) * TH, w0 = (tt % a.tilesX) * TW;
    const float* dyb = a.dy + (long long)n * a.Cout * HW + (long long)h0 * a.W + w0;
    const float* xb = a.x0 + (long long)n * a.Cin * HW + (long long)h0 * a.W + w0;
    amask = 0;
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      unsigned pk = a_pk[e];
      asm volatile("" : "+v"(pk));          // keep the decode inside the tile loop (see wgrad3x3_kernel)
      const int ar = (pk >> 17) & 15, ac = (pk >> 21) & 63;
      const bool ok = (pk >> 31) && h0 + ar < a.H && w0 + ac < a.W;
      if constexpr (!A2) {
        areg[e] = *reinterpret_cast<const f32x4*>(ok ? dyb + a_goff[e] : a.dy);
      } else {                               // rows only 8-byte aligned (W % 4 == 2): two float2, the second may be past the row
        const bool ok2 = ok && w0 + ac + 2 < a.W;
        const float2 lo = *reinterpret_cast<const float2*>(ok ? dyb + a_goff[e] : a.dy);
        const float2 hi = *reinterpret_cast<const float2*>(ok2 ? dyb + a_goff[e] + 2 : a.dy);
        areg[e] = f32x4{lo.x, lo.y, ok2 ? hi.x : 0.f, ok2 ? hi.y : 0.f};
      }
      amask |= (unsigned)ok << e;
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      unsigned pk = b_pk[e];
      asm volatile("" : "+v"(pk));
      const int pr = (int)((pk >> 17) & 15) - DIL, gc = (int)((pk >> 21) & 15) * 8 - 8;   // row / first column rel. to the tile
      const bool rok = (pk >> 31) && (unsigned)(h0 + pr) < (unsigned)a.H;
      const int col = w0 + gc;
      unsigned m = 0;                       // one bit per PAIR (bits 0, 2, 4, 6): W and col are even, a pair is all-in or all-out
#pragma unroll
      for (int p = 0; p < 4; ++p) m |= (unsigned)(rok && (unsigned)(col + 2 * p) < (unsigned)a.W) << (2 * p);
      bmask[e] = m;
      const float* src = xb + b_goff[e];
      if constexpr (!A2) {
        breg[e][0] = *reinterpret_cast<const f32x4*>((m & 0x01u) ? src : a.x0);           // quads are all-in or all-out (W % 4 == 0)
        breg[e][1] = *reinterpret_cast<const f32x4*>((m & 0x10u) ? src + 4 : a.x0);
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {        // pairs are all-in or all-out (W % 2 == 0)
          const float2 v = *reinterpret_cast<const float2*>(((m >> (2 * p)) & 1u) ? src + 2 * p : a.x0);
          breg[e][p >> 1][(2 * p) & 3] = v.x;
          breg[e][p >> 1][((2 * p) & 3) + 1] = v.y;
        }
      }
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
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = RAW ? breg[e][j >> 2][j & 3] : act_by_slope(fmaf(breg[e][j >> 2][j & 3], b_sc[e], b_sh[e]), slope);
      }
      if (BE * NT == BU || tid + NT * e < BU) {
        // zero padding of the ACTIVATED tensor, on the packed pairs (W is even: a pair is all-in or all-out)
        const unsigned m = bmask[e];
        u32x4 o = {(m & 1u) ? bf_pack2(v[0], v[1]) : 0u, (m & 4u) ? bf_pack2(v[2], v[3]) : 0u,
                   (m & 16u) ? bf_pack2(v[4], v[5]) : 0u, (m & 64u) ? bf_pack2(v[6], v[7]) : 0u};
        *reinterpret_cast<u32x4*>(Bb + (pk & 0x1ffffu)) = o;
      }
    }
  };

  if (t_begin < t_end) {
    issue(t_begin);
    finish(0);
  }
  __syncthreads();
  const int a_lane = (wr * 32 + li) * A_ROW + lk * 16;
  const int b_lane = (wc * 32 + li) * CH + lk * 16;
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
        for (int kh = 0; kh < 3; ++kh) {
          // elements 6+w .. 17+w of patch row r + kh*DIL (w = 16q + 8*lk): dwords d0 .. d5
          const unsigned char* row = Bp + (r + kh * DIL) * ROWB + (16 * q) * 2;
          const uint2 lo = *reinterpret_cast<const uint2*>(row + 8);        // elements 4..7   (d0 = lo.y)
          const u32x4 mid = *reinterpret_cast<const u32x4*>(row + 16);      // elements 8..15  (d1..d4)
          const uint2 hi = *reinterpret_cast<const uint2*>(row + 32);       // elements 16..19 (d5 = hi.x)
          u32x4 f0, f2;
          if constexpr (DIL == 1) {
            f0 = u32x4{__builtin_amdgcn_alignbit(mid.x, lo.y, 16), __builtin_amdgcn_alignbit(mid.y, mid.x, 16),
                       __builtin_amdgcn_alignbit(mid.z, mid.y, 16), __builtin_amdgcn_alignbit(mid.w, mid.z, 16)};
            f2 = u32x4{__builtin_amdgcn_alignbit(mid.y, mid.x, 16), __builtin_amdgcn_alignbit(mid.z, mid.y, 16),
                       __builtin_amdgcn_alignbit(mid.w, mid.z, 16), __builtin_amdgcn_alignbit(hi.x, mid.w, 16)};
          } else {
            f0 = u32x4{lo.y, mid.x, mid.y, mid.z};
            f2 = u32x4{mid.y, mid.z, mid.w, hi.x};
          }
          acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f0), acc[kh * 3 + 0], 0, 0, 0);
          acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, mid), acc[kh * 3 + 1], 0, 0, 0);
          acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, f2), acc[kh * 3 + 2], 0, 0, 0);
        }
      }
    }
    if (t + 1 < t_end) finish(buf ^ 1);
    __syncthreads();
  }
  // epilogue: row = output channel, MFMA column (lane) = input channel, accumulator = tap; tap-major slab
  const int ci = c0 + wc * 32 + li;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (co < a.Cout && ci < a.Cin) a.out[(((long long)split * 9 + j) * a.Cout + co) * a.Cin + ci] = acc[j][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static inline bool wb_enabled() {
  static const bool on = getenv("AVSEP_NO_BF16_KERNELS") == nullptr && getenv("AVSEP_NO_BF16_WGRAD") == nullptr;
  return on;
}

bool wb_applicable(const avsep_conv_desc* d) {
  if (d->prec != AVSEP_PREC_BF16 || !wb_enabled()) return false;
  if (!(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil)) return false;
  if (d->up2x || d->C0 != d->Cin) return false;
  // maps narrower than a 16-pixel k-step (U-Net u6 / u7: 8 / 4 wide) run with the unused columns zero-masked
  return d->W >= 4 && d->H >= 2 && (d->W & 1) == 0 && d->Cout >= 32 && d->Cin >= 32 && d->N <= 65535 &&
         (long long)(d->Cout > d->Cin ? d->Cout : d->Cin) * d->H * d->W < 0x7fffffffLL;
}

struct WBPlan { int tilesX, tilesY, gridM, gridC, splits, tps; bool wide; };
static WBPlan wb_plan(const avsep_conv_desc* d) {
  WBPlan p;
  p.wide = d->W > 16;
  p.tilesX = cdiv(d->W, p.wide ? 32 : 16);
  // 16-wide tiles are 8 rows tall: twice the MFMA work per staged tile (the loads of tile t+1 have one tile's MFMAs to
  // arrive in, and a 4x16 tile's 36 MFMAs are shorter than an HBM round trip); the dilated 32-wide patch only fits
  // with 2-row tiles
  p.tilesY = cdiv(d->H, p.wide ? (d->dil == 2 ? 2 : 4) : 8);
  p.gridM = cdiv(d->Cout, WB_BM);
  p.gridC = cdiv(d->Cin, WB_BC);
  const long long tiles = (long long)p.tilesX * p.tilesY * d->N;
  // one workgroup per CU (107 KB of LDS) and nothing overlaps a workgroup's prologue / epilogue: ONE round of workgroups
  // (measured 256 / 512 / 768 / 1024: 231 / 201 / 175 / 154 TFLOP/s at 64 -> 64 @ 56x56, 573 / 543 / 506 / 494 at 1024 -> 512 @ 16x16)
  static const char* tw = getenv("AVSEP_WB_WGS");
  int want = (tw ? atoi(tw) : cu_count()) / (p.gridM * p.gridC);
  if (want < 1) want = 1;
  const long long maxs = tiles / 4 > 0 ? tiles / 4 : 1;     // at least 4 pixel tiles per slab
  int splits = (int)(want < maxs ? want : maxs);
  if (splits < 1) splits = 1;
  p.tps = (int)((tiles + splits - 1) / splits);
  p.splits = (int)((tiles + p.tps - 1) / p.tps);
  return p;
}
size_t wb_workspace_floats(const avsep_conv_desc* d) {
  WBPlan p = wb_plan(d);
  return (size_t)p.splits * d->Cout * d->Cin * 9;          // always through slabs (tap-major) + the transposing reduce
}
int w3_reduce(const float* ws, float* dw, long long P, int splits, hipStream_t st);   // conv3x3.hip

int wb_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  WBPlan p = wb_plan(d);
  WBArgs a{};
  a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout;
  a.act0 = d->act0; a.x0 = d->x0; a.sc0 = d->scale0; a.sh0 = d->shift0;
  a.dy = dy; a.out = ws;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.gridM = p.gridM; a.gridC = p.gridC; a.tiles_per_split = p.tps;
  dim3 grid(p.gridM * p.gridC, p.splits);
  const bool a2 = (d->W & 3) != 0;
  const bool raw = d->scale0 == nullptr && d->act0 == AVSEP_ACT_NONE;
#define WB_L(TW_, DIL_, A2_)                                                                                                   \
  do {                                                                                                                         \
    if (raw) hipLaunchKernelGGL((wgradbf_kernel<TW_ == 16 ? 8 : (DIL_ == 2 ? 2 : 4), TW_, DIL_, A2_, true>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((wgradbf_kernel<TW_ == 16 ? 8 : (DIL_ == 2 ? 2 : 4), TW_, DIL_, A2_, false>), grid, dim3(256), 0, st, a); \
  } while (0)
  if (d->dil == 1) {
    if (p.wide) { if (a2) WB_L(32, 1, true); else WB_L(32, 1, false); }
    else { if (a2) WB_L(16, 1, true); else WB_L(16, 1, false); }
  } else {
    if (p.wide) { if (a2) WB_L(32, 2, true); else WB_L(32, 2, false); }
    else { if (a2) WB_L(16, 2, true); else WB_L(16, 2, false); }
  }
#undef WB_L
  AVSEP_LAUNCH_CHECK();
  return w3_reduce(ws, dw, (long long)d->Cout * d->Cin, p.splits, st);
}
