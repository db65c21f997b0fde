// Weight gradient with bf16 operands read from B16 images (AVSEP_FMT_B16, include/avsep.h), fp32 accumulation on
// v_mfma_f32_32x32x16_bf16:  dW[co][ci][kh][kw] = sum over (n, oh, ow) of dY[n][co][oh][ow] * act(affine(X))[n][ci][oh*S-pad+kh*DIL][..]
// for the 3x3 (dil 1, 2), 4x4/s2, 3x3/s2 and 1x1 convolutions of the U-Net (audio_net.py:57-58,75-98) and of the ResNet
// trunk (vision_net.py:84-109), replacing wgradbf_kernel / wgrad4bf_kernel (which read fp32 NCHW and rebuilt their
// operands with 8 loads + 4 converts + pre-shifted copies per 8 values).
//
// The GEMM: M = co, N = ci (one accumulator tile per tap), K = PIXELS.  In a B16 image the 16 channels of a position are 32
// contiguous bytes, so a (position, 8-channel half) slot travels HBM -> register -> LDS as one 16-byte piece with no
// conversion (X: the folded BatchNorm affine + activation is applied on the way when the conv has one), and the LDS images
// are simply [channel block][position][16 ch].  K = pixels means an MFMA lane needs 8 consecutive PIXELS of one channel —
// the transpose of that image — which gfx950's ds_read_b64_tr_b16 delivers for free: per 16-lane group it reads a
// 4 (positions) x 16 (channels) block and hands lane i channel i of the 4 positions.  Two such reads = one bf16x8 operand.
// A tap (kh, kw) is a constant byte offset on the X read, a stride-2 conv doubles the position step: every geometry is
// address arithmetic folded into the `offset:` immediates of fully unrolled reads — no shifted copies, no v_alignbit.
//
// Workgroup = 64 co x 64 ci x all taps, 8 waves as 2 (co) x 2 (ci) x 2 tap groups (4 waves for 1x1), <= 5 accumulator tiles
// of 16 registers per wave, two waves per SIMD; it walks its share of pixel chunks (TH x TW output pixels of one image;
// TW % 16 == 0 so that a k-step of 16 pixels stays in one row; columns past the map are zero dY); the next chunk's slots are
// in flight in registers while the current one is multiplied out of LDS.  K-split over chunks (one workgroup per CU, one
// round) -> tap-major slabs [split][tap][co][ci] + a split-parallel deterministic reduce that writes OIHW.
// LDS plane strides are padded so that the two 16-lane groups of a 32-lane half hit disjoint banks (stride 1: planes 128 B
// apart mod 256; stride 2: 32 B apart mod 64).
#include <type_traits>
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct WbArgs {
  const u32x4* x;
  const u32x4* dy;
  const float* sc;
  const float* sh;
  int act;
  int N, Cin, H, W, Cout, Ho, Wo, pad;
  int tilesX, tilesY, chunks, per_split, gridCi, gridCo;
  float* out;
  long long slab;
  int CoutP, CinP;
};

__device__ __forceinline__ unsigned wb_pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ bf16x8 wb_tr8(const unsigned char* p, int off0, int off1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + off0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + off1));
  const s16x8 t = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, t);
}

// TG tap groups: the workgroup has TG x 4 waves; group tg multiplies taps [tg * NTW, (tg + 1) * NTW) of the same 64 x 64 tile out
// of the same LDS images.  Two groups = two waves per SIMD at <= 256 registers: one wave's staging / LDS waits overlap the
// other's MFMAs (one 4-wave workgroup per CU left the matrix pipe idle for ~60 % of a chunk: stage + barriers + issue).
template <int KH, int KW, int S, int DIL, int TH, int TW, bool RAW>
__global__ __launch_bounds__(KH * KW > 1 ? 512 : 256) void wgradb_kernel(WbArgs a) {
  constexpr int NT = KH * KW, NPIX = TH * TW, KST = NPIX / 16;
  constexpr int TG = NT > 1 ? 2 : 1, NTHR = 256 * TG, NTW = (NT + TG - 1) / TG;
  static_assert(TW % 16 == 0, "a k-step of 16 pixels stays inside one tile row");
  constexpr int PH = (TH - 1) * S + (KH - 1) * DIL + 1, PW = (TW - 1) * S + (KW - 1) * DIL + 1, PS = PH * PW;
  constexpr int XRES = S == 1 ? 128 : 32;
  constexpr int XPL = PS * 32 + ((XRES - (PS * 32) % 256) + 256) % 256;          // bytes per channel-block plane of the X patch
  constexpr int YPL = NPIX * 32 + ((128 - (NPIX * 32) % 256) + 256) % 256;       // ... of the dY tile
  constexpr int XSL = 4 * PS * 2, YSL = 4 * NPIX * 2;                            // 16-byte slots (4 channel blocks each)
  constexpr int XE = (XSL + NTHR - 1) / NTHR, YE = (YSL + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * XPL + 4 * YPL];
  unsigned char* const Xs = smem;
  unsigned char* const Ys = smem + 4 * XPL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  // 1-D grid, XCD-aware (round 5): the gco * gci workgroups of one K-split read the same pixels; consecutive LOGICAL ids
  // share an XCD, so those pixels reach one L2 instead of eight
  const int t_ = xcd_remap(blockIdx.x, gridDim.x), per_split_ = a.gridCo * a.gridCi;
  const int zsplit = t_ / per_split_, lin_ = t_ % per_split_;
  const int cot = lin_ / a.gridCi, cit = lin_ % a.gridCi;
  const int CBi = a.Cin >> 4, CBo = a.Cout >> 4;
  const float slope = act_slope(a.act);
  const bool has_aff = !RAW && a.sc != nullptr;

  // folded BatchNorm rows of this workgroup's 64 input channels -> LDS once (read back per slot while staging)
  __shared__ __attribute__((aligned(16))) float aff[2][64];
  if constexpr (!RAW) {
    if (tid < 64) {
      const int c = cit * 64 + tid;
      aff[0][tid] = (has_aff && c < a.Cin) ? a.sc[c] : 1.f;
      aff[1][tid] = (has_aff && c < a.Cin) ? a.sh[c] : 0.f;
    }
  }

  // ---- per-thread slot tables (constant over chunks): 3 registers per slot ------------------------------------------------------
  int x_rc[XE], x_lds[XE], x_g[XE];        // (row << 16 | col) inside the patch; LDS byte; 16-byte unit inside one image (-1: no such channel block)
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int i = min(tid + NTHR * e, XSL - 1);
    const int cbx = i / (PS * 2), rem = i % (PS * 2), pos = rem >> 1, half = rem & 1;
    x_rc[e] = ((pos / PW) << 16) | (pos % PW);
    x_lds[e] = cbx * XPL + pos * 32 + half * 16;
    const int cb = cit * 4 + cbx;
    x_g[e] = (cb < CBi && (XE * NTHR == XSL || tid + NTHR * e < XSL)) ? ((cb * a.H + pos / PW) * a.W + pos % PW) * 2 + half : -1;
  }
  int y_rc[YE], y_lds[YE], y_g[YE];
#pragma unroll
  for (int e = 0; e < YE; ++e) {
    const int i = min(tid + NTHR * e, YSL - 1);
    const int cby = i / (NPIX * 2), rem = i % (NPIX * 2), pix = rem >> 1, half = rem & 1;
    y_rc[e] = ((pix / TW) << 16) | (pix % TW);
    y_lds[e] = cby * YPL + pix * 32 + half * 16;
    const int cb = cot * 4 + cby;
    y_g[e] = (cb < CBo && (YE * NTHR == YSL || tid + NTHR * e < YSL)) ? ((cb * a.Ho + pix / TW) * a.Wo + pix % TW) * 2 + half : -1;
  }

  u32x4 xr[XE], yr[YE];
  unsigned xok = 0;
  // per chunk and slot: one add per coordinate, two compares, one select, one add for the address (the slot's offset
  // inside its image relative to the patch origin is a per-thread constant: x_g[e] already holds (cb*H + r)*W + c)
  auto issue = [&](int c) __attribute__((always_inline)) {
    const int tx = c % a.tilesX, t2 = c / a.tilesX, ty = t2 % a.tilesY, n = t2 / a.tilesY;
    const int gh0 = ty * TH * S - a.pad, gw0 = tx * TW * S - a.pad;
    const u32x4* const xn = a.x + ((long long)n * CBi * a.H * a.W + (long long)gh0 * a.W + gw0) * 2;   // may point before the image: only valid slots load
    const u32x4* const yn = a.dy + ((long long)n * CBo * a.Ho * a.Wo + (long long)(ty * TH) * a.Wo + tx * TW) * 2;
    xok = 0;
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const bool ok = x_g[e] >= 0 && (unsigned)(gh0 + (x_rc[e] >> 16)) < (unsigned)a.H && (unsigned)(gw0 + (x_rc[e] & 0xffff)) < (unsigned)a.W;
      const u32x4 z = {0u, 0u, 0u, 0u};
      xr[e] = ok ? xn[x_g[e]] : z;
      xok |= (unsigned)ok << e;
    }
#pragma unroll
    for (int e = 0; e < YE; ++e) {
      const bool ok = y_g[e] >= 0 && ty * TH + (y_rc[e] >> 16) < a.Ho && tx * TW + (y_rc[e] & 0xffff) < a.Wo;
      const u32x4 z = {0u, 0u, 0u, 0u};
      yr[e] = ok ? yn[y_g[e]] : z;
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      u32x4 q = xr[e];
      if constexpr (!RAW) {
        // channel offset of the slot inside the workgroup's 64: (channel block, half) are bits of the LDS address
        const int ch8 = (x_lds[e] / XPL) * 16 + ((x_lds[e] >> 4) & 1) * 8;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(&aff[0][ch8]), s1 = *reinterpret_cast<const f32x4*>(&aff[0][ch8 + 4]);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(&aff[1][ch8]), h1 = *reinterpret_cast<const f32x4*>(&aff[1][ch8 + 4]);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x = __builtin_bit_cast(float, (j & 1) ? (q[j >> 1] & 0xffff0000u) : (q[j >> 1] << 16));
          x = fmaf(x, j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? h0[j & 3] : h1[j & 3]);
          v[j] = act_by_slope(x, slope);
        }
        q = u32x4{wb_pack2(v[0], v[1]), wb_pack2(v[2], v[3]), wb_pack2(v[4], v[5]), wb_pack2(v[6], v[7])};
      }
      if (XE * NTHR == XSL || tid + NTHR * e < XSL) *reinterpret_cast<u32x4*>(Xs + x_lds[e]) = (RAW || ((xok >> e) & 1u)) ? q : z;   // raw invalid slots loaded as 0
    }
#pragma unroll
    for (int e = 0; e < YE; ++e)
      if (YE * NTHR == YSL || tid + NTHR * e < YSL) *reinterpret_cast<u32x4*>(Ys + y_lds[e]) = yr[e];
  };

  // ---- operand addresses of the transposed reads: lane 4q + p of a 16-lane group supplies row q (a position), 8-byte column p --
  const int g16 = lane >> 4, blk = g16 & 1, kg = g16 >> 1, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* const a_lane = Ys + (2 * wm + blk) * YPL + (8 * kg + q) * 32 + p * 8;
  const unsigned char* const b_lane = Xs + (2 * wn + blk) * XPL + (8 * kg + q) * S * 32 + p * 8;

  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int c_beg = zsplit * a.per_split, c_end = min(a.chunks, c_beg + a.per_split);
  if (c_beg < c_end) issue(c_beg);
  for (int c = c_beg; c < c_end; ++c) {
    __syncthreads();                         // every wave is done reading the previous chunk
    stage();
    __syncthreads();
    if (c + 1 < c_end) issue(c + 1);         // in flight while this chunk is multiplied
    auto ksteps = [&](auto tgc) __attribute__((always_inline)) {
      constexpr int T0 = decltype(tgc)::value * NTW;
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        const int row = (ks * 16) / TW, col0 = (ks * 16) % TW;
        const bf16x8 av = wb_tr8(a_lane, ks * 16 * 32, ks * 16 * 32 + 4 * 32);
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          if (T0 + t < NT) {
            const int kh = (T0 + t) / KW, kw = (T0 + t) % KW;
            const int off = ((row * S + kh * DIL) * PW + col0 * S + kw * DIL) * 32;
            const bf16x8 bv = wb_tr8(b_lane, off, off + 4 * S * 32);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
          }
        }
      }
    };
    if (TG == 1 || tg == 0) ksteps(std::integral_constant<int, 0>{});
    else ksteps(std::integral_constant<int, TG - 1>{});
  }

  // ---- partial slab [tap][CoutP][CinP]: C/D map col = lane & 31 (ci), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (co) ----------
  float* const op = a.out + (long long)zsplit * a.slab;
  const int li = lane & 31, lk = lane >> 5;
  const int ci = cit * 64 + wn * 32 + li;
#pragma unroll
  for (int t = 0; t < NTW; ++t)
    if (tg * NTW + t < NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        op[((long long)(tg * NTW + t) * a.CoutP + co) * a.CinP + ci] = acc[t][r];
      }
    }
}


// ---- Cin == 16: the ResNet stem as a 4x4 / stride-1 conv over the space-to-depth frames (one 16-channel block) -----------------
// With a single input block the 64 x 64 tile above would multiply 48 zero columns.  Here N = (tap, ci): a 32-column MFMA tile is
// TWO taps x the 16 channels — the two 16-lane groups of a transposed read simply take different tap offsets — so all 16 taps
// are 8 full tiles: 8 waves = 2 (co halves) x 4 (tap quads), two accumulator tiles per wave, every MFMA column useful.
template <int KH, int KW, int TH, int TW>
__global__ __launch_bounds__(512) void wgradb_ci16_kernel(WbArgs a) {
  constexpr int NT = KH * KW, NPIX = TH * TW, KST = NPIX / 16, NTHR = 512;
  static_assert(NT == 16 && TW % 16 == 0, "16 taps = 4 quads");
  constexpr int PH = TH + KH - 1, PW = TW + KW - 1, PS = PH * PW;
  constexpr int YPL = NPIX * 32 + ((128 - (NPIX * 32) % 256) + 256) % 256;
  constexpr int XSL = PS * 2, YSL = 4 * NPIX * 2;
  constexpr int XE = (XSL + NTHR - 1) / NTHR, YE = (YSL + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PS * 32 + 4 * YPL];
  unsigned char* const Xs = smem;
  unsigned char* const Ys = smem + PS * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tq = wave >> 1, wm = wave & 1;
  const int cot = blockIdx.x, CBo = a.Cout >> 4;

  int x_rc[XE], x_lds[XE], x_g[XE];
#pragma unroll
  for (int e = 0; e < XE; ++e) {
    const int i = min(tid + NTHR * e, XSL - 1), pos = i >> 1, half = i & 1;
    x_rc[e] = ((pos / PW) << 16) | (pos % PW);
    x_lds[e] = pos * 32 + half * 16;
    x_g[e] = (XE * NTHR == XSL || tid + NTHR * e < XSL) ? ((pos / PW) * a.W + pos % PW) * 2 + half : -1;
  }
  int y_rc[YE], y_lds[YE], y_g[YE];
#pragma unroll
  for (int e = 0; e < YE; ++e) {
    const int i = min(tid + NTHR * e, YSL - 1);
    const int cby = i / (NPIX * 2), rem = i % (NPIX * 2), pix = rem >> 1, half = rem & 1;
    y_rc[e] = ((pix / TW) << 16) | (pix % TW);
    y_lds[e] = cby * YPL + pix * 32 + half * 16;
    const int cb = cot * 4 + cby;
    y_g[e] = (cb < CBo && (YE * NTHR == YSL || tid + NTHR * e < YSL)) ? ((cb * a.Ho + pix / TW) * a.Wo + pix % TW) * 2 + half : -1;
  }
  u32x4 xr[XE], yr[YE];
  auto issue = [&](int c) __attribute__((always_inline)) {
    const int tx = c % a.tilesX, t2 = c / a.tilesX, ty = t2 % a.tilesY, n = t2 / a.tilesY;
    const int gh0 = ty * TH - a.pad, gw0 = tx * TW - a.pad;
    const u32x4* const xn = a.x + ((long long)n * a.H * a.W + (long long)gh0 * a.W + gw0) * 2;
    const u32x4* const yn = a.dy + ((long long)n * CBo * a.Ho * a.Wo + (long long)(ty * TH) * a.Wo + tx * TW) * 2;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < XE; ++e) {
      const bool ok = x_g[e] >= 0 && (unsigned)(gh0 + (x_rc[e] >> 16)) < (unsigned)a.H && (unsigned)(gw0 + (x_rc[e] & 0xffff)) < (unsigned)a.W;
      xr[e] = ok ? xn[x_g[e]] : z;
    }
#pragma unroll
    for (int e = 0; e < YE; ++e) {
      const bool ok = y_g[e] >= 0 && ty * TH + (y_rc[e] >> 16) < a.Ho && tx * TW + (y_rc[e] & 0xffff) < a.Wo;
      yr[e] = ok ? yn[y_g[e]] : z;
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < XE; ++e)
      if (XE * NTHR == XSL || tid + NTHR * e < XSL) *reinterpret_cast<u32x4*>(Xs + x_lds[e]) = xr[e];
#pragma unroll
    for (int e = 0; e < YE; ++e)
      if (YE * NTHR == YSL || tid + NTHR * e < YSL) *reinterpret_cast<u32x4*>(Ys + y_lds[e]) = yr[e];
  };
  const int g16 = lane >> 4, blk = g16 & 1, kg = g16 >> 1, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* const a_lane = Ys + (2 * wm + blk) * YPL + (8 * kg + q) * 32 + p * 8;
  const unsigned char* b_lane[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tap = tq * 4 + 2 * j + blk;                       // this 16-lane group's tap of N-tile j
    b_lane[j] = Xs + ((tap / KW) * PW + tap % KW + 8 * kg + q) * 32 + p * 8;
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int c_beg = blockIdx.z * a.per_split, c_end = min(a.chunks, c_beg + a.per_split);
  if (c_beg < c_end) issue(c_beg);
  for (int c = c_beg; c < c_end; ++c) {
    __syncthreads();
    stage();
    __syncthreads();
    if (c + 1 < c_end) issue(c + 1);
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) {
      const int off = (((ks * 16) / TW) * PW + (ks * 16) % TW) * 32;
      const bf16x8 av = wb_tr8(a_lane, ks * 16 * 32, ks * 16 * 32 + 4 * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bv = wb_tr8(b_lane[j], off, off + 4 * 32);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
      }
    }
  }
  float* const op = a.out + (long long)blockIdx.z * a.slab;
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tap = tq * 4 + 2 * j + (li >> 4), ci = li & 15;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cot * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      op[((long long)tap * a.CoutP + co) * a.CinP + ci] = acc[j][r];
    }
  }
}

// dW[co][ci][tap] = sum over splits of slab[s][tap][co][ci].  Block = 4 split groups x 64 input channels of one output channel:
// a thread sums every 4th split (two chains in flight), the four partial sums meet in LDS in a fixed order (deterministic).
// One thread per (co, ci) walking all splits alone left 64-channel layers (256 splits, 64 active threads per block) latency-bound.
template <int NT>
__global__ __launch_bounds__(256) void wgradb_reduce_kernel(const float* __restrict__ ws, long long slab, int splits, int Cout,
                                                            int Cin, int CoutP, int CinP, float* __restrict__ dw) {
  __shared__ float part[3][NT][64];
  const int sg = threadIdx.x >> 6, ci = blockIdx.x * 64 + (threadIdx.x & 63), co = blockIdx.y;
  const bool live = ci < Cin;
  float s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s[t] = 0.f;
  if (live) {
    const long long tap_stride = (long long)CoutP * CinP;
    for (int z = sg; z < splits; z += 4) {
      const float* p = ws + (long long)z * slab + (long long)co * CinP + ci;
#pragma unroll
      for (int t = 0; t < NT; ++t) s[t] += p[(long long)t * tap_stride];
    }
  }
  if (sg > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) part[sg - 1][t][threadIdx.x & 63] = s[t];
  }
  __syncthreads();
  if (sg == 0 && live) {
    float* o = dw + ((long long)co * Cin + ci) * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = (s[t] + part[0][t][threadIdx.x]) + (part[1][t][threadIdx.x] + part[2][t][threadIdx.x]);
  }
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
static int wbn_class(const avsep_conv_desc* d) {
  if (d->up2x || d->C0 != d->Cin || d->prec != AVSEP_PREC_BF16) return 0;
  if (d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->dil == 1 || d->dil == 2) && d->pad == d->dil) return d->dil == 1 ? 1 : 2;
  if (d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->dil == 1) return 3;
  if (d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1 && d->dil == 1) return 4;
  if (d->KH == 1 && d->KW == 1 && d->pad == 0 && d->stride == 1) return 5;
  if (d->KH == 1 && d->KW == 1 && d->pad == 0 && d->stride == 2) return 6;
  if (d->KH == 4 && d->KW == 4 && d->stride == 1 && d->pad == 0 && d->dil == 1 && d->Cin == 16 && !d->scale0 && d->act0 == AVSEP_ACT_NONE)
    return 7;                                              // the stem over space-to-depth frames (raw input)
  return 0;
}
static inline bool wbn_enabled(const avsep_conv_desc* d) { return !(d->algo & AVSEP_ALGO_NO_BF16_KERNELS); }
bool wbn_applicable(const avsep_conv_desc* d) {
  if (!wbn_enabled(d) || !wbn_class(d)) return false;
  if (d->Cin % 16 || d->Cout % 16 || d->Cin < 16 || d->Cout < 16) return false;
  if (d->Wo < 8 || d->Ho < 4 || d->N > 65535) return false;
  if (wbn_class(d) == 7 && d->Wo <= 16) return false;          // the Cin == 16 form has the 8 x 32 chunk only
  if ((long long)d->N * d->Cin * d->H * d->W >= (1LL << 34) || (long long)d->N * d->Cout * d->Ho * d->Wo >= (1LL << 34)) return false;
  return true;
}
struct WbPlan { bool wide; int th, tw, tilesX, tilesY, chunks, splits, per_split, gco, gci, CoutP, CinP, plan_splits; };
static WbPlan wbn_plan(const avsep_conv_desc* d) {
  WbPlan p;
  const int cls = wbn_class(d);
  const bool s2 = cls == 3 || cls == 4 || cls == 6;
  p.wide = d->Wo > 16;
  p.tw = p.wide ? 32 : 16;
  p.th = cls == 3 ? (p.wide ? 2 : 4) : s2 ? (p.wide ? 4 : 8) : (p.wide ? 8 : 16);   // 16 taps (128 accumulator registers per wave): smaller chunks
  p.tilesX = cdiv(d->Wo, p.tw);
  p.tilesY = cdiv(d->Ho, p.th);
  p.gco = cdiv(d->Cout, 64);
  p.gci = cdiv(d->Cin, 64);
  p.CoutP = p.gco * 64;
  p.CinP = p.gci * 64;
  p.chunks = d->N * p.tilesX * p.tilesY;
  const long long plan_chunks = (long long)plan_batch(d) * p.tilesX * p.tilesY;
  int s = cdiv(cu_count(), p.gco * p.gci);            // one 512-thread workgroup per CU, one round
  if (s > plan_chunks) s = (int)plan_chunks;
  if (s < 1) s = 1;
  p.plan_splits = s;
  if (s > p.chunks) s = p.chunks;
  // the split count follows the PLANNED batch; chunks per split follow the real one
  p.per_split = cdiv(p.chunks, s);
  p.splits = cdiv(p.chunks, p.per_split);
  return p;
}
size_t wbn_workspace_floats(const avsep_conv_desc* d) {
  const WbPlan p = wbn_plan(d);
  return (size_t)p.splits * d->KH * d->KW * p.CoutP * p.CinP;
}
void wbn_variant(const avsep_conv_desc* d, char* buf, size_t cap) {
  const WbPlan p = wbn_plan(d);
  snprintf(buf, cap, "%dx%d,split%d", p.th, p.tw, p.plan_splits);
}

template <int KH, int KW, int S, int DIL>
static int wbn_launch(WbArgs& a, const WbPlan& p, bool raw, hipStream_t st) {
  dim3 grid(p.gco * p.gci * p.splits);
  constexpr int THW = KH * KW == 16 ? 2 : S == 2 ? 4 : 8, THN = KH * KW == 16 ? 4 : S == 2 ? 8 : 16;
#define WB_L(TH_, TW_)                                                                                                    \
  do {                                                                                                                    \
    if (raw) hipLaunchKernelGGL((wgradb_kernel<KH, KW, S, DIL, TH_, TW_, true>), grid, dim3(KH * KW > 1 ? 512 : 256), 0, st, a);  \
    else hipLaunchKernelGGL((wgradb_kernel<KH, KW, S, DIL, TH_, TW_, false>), grid, dim3(KH * KW > 1 ? 512 : 256), 0, st, a);     \
  } while (0)
  if (p.wide) WB_L(THW, 32); else WB_L(THN, 16);
#undef WB_L
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

int wbn_wgrad(const avsep_conv_desc* d, const float* dy, float* dw, float* ws, hipStream_t st) {
  if (d->xfmt != AVSEP_FMT_B16 || d->dyfmt != AVSEP_FMT_B16) return AVSEP_ERR_ARG;
  const WbPlan p = wbn_plan(d);
  WbArgs a{};
  a.x = (const u32x4*)d->x0; a.dy = (const u32x4*)dy; a.sc = d->scale0; a.sh = d->shift0; a.act = d->act0;
  a.N = d->N; a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Cout = d->Cout; a.Ho = d->Ho; a.Wo = d->Wo; a.pad = d->pad;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.chunks = p.chunks; a.per_split = p.per_split; a.gridCi = p.gci; a.gridCo = p.gco;
  a.out = ws; a.CoutP = p.CoutP; a.CinP = p.CinP;
  a.slab = (long long)d->KH * d->KW * p.CoutP * p.CinP;
  const bool raw = d->scale0 == nullptr && d->act0 == AVSEP_ACT_NONE;
  int rc;
  switch (wbn_class(d)) {
    case 7:
      hipLaunchKernelGGL((wgradb_ci16_kernel<4, 4, 8, 32>), dim3(p.gco, 1, p.splits), dim3(512), 0, st, a);
      rc = hipGetLastError() == hipSuccess ? AVSEP_OK : AVSEP_ERR_LAUNCH;
      break;
    case 1: rc = wbn_launch<3, 3, 1, 1>(a, p, raw, st); break;
    case 2: rc = wbn_launch<3, 3, 1, 2>(a, p, raw, st); break;
    case 3: rc = wbn_launch<4, 4, 2, 1>(a, p, raw, st); break;
    case 4: rc = wbn_launch<3, 3, 2, 1>(a, p, raw, st); break;
    case 5: rc = wbn_launch<1, 1, 1, 1>(a, p, raw, st); break;
    default: rc = wbn_launch<1, 1, 2, 1>(a, p, raw, st); break;
  }
  if (rc) return rc;
  dim3 rg(cdiv(d->Cin, 64), d->Cout);
  switch (d->KH * d->KW) {
    case 9: hipLaunchKernelGGL(wgradb_reduce_kernel<9>, rg, dim3(256), 0, st, ws, a.slab, p.splits, d->Cout, d->Cin, p.CoutP, p.CinP, dw); break;
    case 16: hipLaunchKernelGGL(wgradb_reduce_kernel<16>, rg, dim3(256), 0, st, ws, a.slab, p.splits, d->Cout, d->Cin, p.CoutP, p.CinP, dw); break;
    default: hipLaunchKernelGGL(wgradb_reduce_kernel<1>, rg, dim3(256), 0, st, ws, a.slab, p.splits, d->Cout, d->Cin, p.CoutP, p.CinP, dw); break;
  }
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
