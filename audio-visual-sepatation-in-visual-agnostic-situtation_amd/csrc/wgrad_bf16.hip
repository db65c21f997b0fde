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

constexpr int WB_BC = 32;

// RAW: the input needs no affine / activation (the U-Net decoder's materialised ReLU+upsample tensor): these kernels
// are VALU-bound in their staging (SQ counters: 15-19 VALU per MFMA before this), so that work is compiled out.
// HALF (Cout <= 64): the tile holds 64 output channels and the four waves are 2 channel blocks x 2 halves of the tile's rows
// (each half writes its own slab) instead of two waves multiplying zero rows: 256 -> 64 @ 128x128 ran at 330 TFLOP/s
// against 580 for the same work at 512 -> 128 @ 64x64.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned WB_OOB = 0x80000000u;     // a byte offset no buffer covers: the load returns zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wb_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fff0000, 0x00020000);   // offsets derived from WB_OOB (-8 .. +32) stay outside
}
__device__ __forceinline__ f32x4 wb_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ float2 wb_ld2(__amdgpu_buffer_rsrc_t r, unsigned off) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  const f32x2v v = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ float wb_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

template <int TH, int TW, int DIL, bool A2, bool RAW, bool HALF>
__global__ __launch_bounds__(256) void wgradbf_kernel(WBArgs a) {
  constexpr int NT = 256, WB_BM = HALF ? 64 : 128, RH = HALF ? TH / 2 : TH;
  static_assert(!HALF || TH % 2 == 0, "row halves");
  constexpr int NPIX = TH * TW, PH = TH + 2 * DIL;
  constexpr int A_ROW = NPIX * 2 + 16;                                  // bytes per co row: 16 x odd
  constexpr int COPYB = TW * 2, RSB = 3 * COPYB;                        // patch row: X | X shifted by -DIL | X shifted by +DIL
  constexpr int CH_RAW = PH * RSB;
  constexpr int CH = (CH_RAW / 16) % 2 == 1 ? CH_RAW : CH_RAW + 16;     // bytes per ci: 16 x odd (lanes = ci)
  constexpr int A_BYTES = WB_BM * A_ROW, B_BYTES = WB_BC * CH;
  constexpr int CO_PER = NPIX / 4, COSTEP = NT / CO_PER, AE = WB_BM / COSTEP;   // dY quads: a thread keeps its (row, quad), e steps co
  constexpr int NG = TW / 8, BU = WB_BC * PH * NG, BE = (BU + NT - 1) / NT;     // 8-element groups of the patch
  static_assert(NT % CO_PER == 0 && WB_BM % COSTEP == 0 && TW % 16 == 0, "tile shape");
  static_assert(2 * (A_BYTES + B_BYTES) <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int wr = HALF ? (wave & 1) : wave, wc = 0, hf = HALF ? (wave >> 1) : 0;
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

  // ---- staging state ------------------------------------------------------------------------------------------------
  // Loads are buffer loads relative to a per-tile base (dY: the tile's first pixel of channel m0; X: DIL rows above and 8
  // columns left of it, channel c0), so a thread's byte offsets are constants and everything that must read as zero
  // (rows / columns outside the image, channels past Cout / Cin) is ONE select of the offset against WB_OOB: no masks, no
  // zero selects after the load (the masked form spent 10-15 vector instructions per MFMA on its staging).
  const int tq = tid % (TW / 4), tr = (tid / (TW / 4)) % TH, tco = tid / CO_PER;
  unsigned a_voff[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    const int co = tco + COSTEP * e;
    a_voff[e] = m0 + co < a.Cout ? (unsigned)((co * (int)HW + tr * a.W + 4 * tq) * 4) : WB_OOB;   // WB_BM*H*W*4 < 2^31: host check
  }
  const int a_lds = tco * A_ROW + (tr * TW + 4 * tq) * 2;
  unsigned b_voff[BE];
  unsigned b_pk[BE];                 // LDS byte offset (bits 0-16) | patch row (17-20) | group (21-23)
  float b_sc[RAW ? 1 : BE], b_sh[RAW ? 1 : BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int idx = min(tid + NT * e, BU - 1);
    const int g = idx % NG, pr = (idx / NG) % PH, cc = idx / (NG * PH);
    const bool uok = (BE * NT == BU || tid + NT * e < BU) && c0 + cc < a.Cin;
    b_voff[e] = uok ? (unsigned)((cc * (int)HW + pr * a.W + 8 * g + 8) * 4) : WB_OOB;             // 32*H*W*4 < 2^31
    b_pk[e] = (unsigned)(cc * CH + pr * RSB + g * 16) | (unsigned)pr << 17 | (unsigned)g << 21;
    if constexpr (!RAW) {
      const int cs = min(c0 + cc, a.Cin - 1);
      b_sc[e] = has_aff ? a.sc0[cs] : 1.f;
      b_sh[e] = has_aff ? a.sh0[cs] : 0.f;
    }
  }
  f32x4 areg[AE];
  float bmain[BE][8], bl[BE][DIL], br[BE][DIL];
  unsigned bmask[RAW ? 1 : BE];      // quad-valid bits of a unit: main quads 0, 4 (pairs 0, 2, 4, 6 when A2), left extra 8, right extra 12

  // ---- rolling software pipeline -------------------------------------------------------------------------------------
  // ONE register set: while tile i's MFMAs run, every staging register is converted and written to LDS (tile i+1, loaded
  // during the previous trip) and immediately re-loaded for tile i+2, so a load has a whole trip to arrive and nothing
  // waits on it.  With one wave per SIMD no other wave hides a phase: the serial form (issue -> MFMAs -> wait for the
  // loads -> convert -> barrier) ran the matrix pipe 26 % of the time.
  // validity as an OR mask on the offset (bit 31 set = outside every buffer): plain integer arithmetic, because a
  // `valid ? offset : WB_OOB` select in front of a load is turned into a branch around two copies of the load
  auto oob = [](bool valid) __attribute__((always_inline)) {
    unsigned m = valid ? 0u : WB_OOB;
    asm("" : "+v"(m));                      // (not volatile: only opaque to the select -> branch rewrite, free to be scheduled)
    return m;
  };
  struct TileRef { __amdgpu_buffer_rsrc_t ra, rb; int h0, w0; unsigned m1, m2; };
  // tiles are visited in order: (image, tile row, tile column) advance as counters (a division per trip was ~100 scalar
  // instructions in front of the first MFMA); past the sweep the counters stop on the last tile (loaded again, never multiplied)
  int cur_t = t_begin, cur_n = t_begin / tiles_img, cur_ty = (t_begin % tiles_img) / a.tilesX, cur_tx = (t_begin % tiles_img) % a.tilesX;
  auto next_tile = [&]() __attribute__((always_inline)) {
    TileRef R;
    R.h0 = cur_ty * TH; R.w0 = cur_tx * TW;
    R.ra = wb_rsrc(a.dy + ((long long)cur_n * a.Cout + m0) * HW + (long long)R.h0 * a.W + R.w0);
    R.rb = wb_rsrc(a.x0 + ((long long)cur_n * a.Cin + c0) * HW + (long long)(R.h0 - DIL) * a.W + (R.w0 - 8));
    const bool rok = R.h0 + tr < a.H;
    R.m1 = oob(rok && R.w0 + 4 * tq < a.W);
    R.m2 = oob(rok && R.w0 + 4 * tq + 2 < a.W);
    if (cur_t + 1 < t_end) {
      ++cur_t;
      if (++cur_tx == a.tilesX) { cur_tx = 0; if (++cur_ty == a.tilesY) { cur_ty = 0; ++cur_n; } }
    }
    return R;
  };
  auto load_a = [&](const TileRef& R, int e) __attribute__((always_inline)) {
    if constexpr (!A2) {
      areg[e] = wb_ld4(R.ra, a_voff[e] | R.m1);                      // W % 4 == 0: a quad is all-in or all-out
    } else {                                                         // rows 8-byte aligned (W % 4 == 2): two pairs
      const float2 lo = wb_ld2(R.ra, a_voff[e] | R.m1), hi = wb_ld2(R.ra, (a_voff[e] + 8) | R.m2);
      areg[e] = f32x4{lo.x, lo.y, hi.x, hi.y};
    }
  };
  auto load_b = [&](const TileRef& R, int e) __attribute__((always_inline)) {
    const unsigned pk = b_pk[e];
    const int pr = (int)((pk >> 17) & 15), col = R.w0 + (int)((pk >> 21) & 7) * 8;          // first column of the group
    const bool rok = (unsigned)(R.h0 + pr - DIL) < (unsigned)a.H;
    const unsigned v0 = b_voff[e];
    const bool mL = rok && col > 0 && col <= a.W, mR = rok && col + 8 < a.W;
    unsigned m = 0;
    if constexpr (!A2) {
      const bool q0 = rok && col < a.W, q1 = rok && col + 4 < a.W;
      const f32x4 x0 = wb_ld4(R.rb, v0 | oob(q0)), x1 = wb_ld4(R.rb, (v0 + 16) | oob(q1));
#pragma unroll
      for (int j = 0; j < 4; ++j) { bmain[e][j] = x0[j]; bmain[e][4 + j] = x1[j]; }
      m = (q0 ? 0x01u : 0u) | (q1 ? 0x10u : 0u);
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {          // pairs are all-in or all-out (W % 2 == 0)
        const bool pp = rok && col + 2 * p < a.W;
        const float2 v = wb_ld2(R.rb, (v0 + 8 * p) | oob(pp));
        bmain[e][2 * p] = v.x; bmain[e][2 * p + 1] = v.y;
        m |= pp ? 1u << (2 * p) : 0u;
      }
    }
    if constexpr (DIL == 1) {
      bl[e][0] = wb_ld1(R.rb, (v0 - 4) | oob(mL));
      br[e][0] = wb_ld1(R.rb, (v0 + 32) | oob(mR));
    } else {
      const float2 l = wb_ld2(R.rb, (v0 - 8) | oob(mL)), r = wb_ld2(R.rb, (v0 + 32) | oob(mR));
      bl[e][0] = l.x; bl[e][1] = l.y; br[e][0] = r.x; br[e][1] = r.y;
    }
    if constexpr (!RAW) bmask[e] = m | (mL ? 0x100u : 0u) | (mR ? 0x1000u : 0u);
  };
  auto store_a = [&](int buf, int e) __attribute__((always_inline)) {
    const f32x4 v = areg[e];
    *reinterpret_cast<uint2*>(smem + buf * (A_BYTES + B_BYTES) + a_lds + e * COSTEP * A_ROW) =
        make_uint2(bf_pack2(v.x, v.y), bf_pack2(v.z, v.w));
  };
  auto store_b = [&](int buf, int e) __attribute__((always_inline)) {
    unsigned char* Bb = smem + buf * (A_BYTES + B_BYTES) + A_BYTES;
    const unsigned pk = b_pk[e];
    float v[8], l[DIL], r[DIL];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bmain[e][j];
#pragma unroll
    for (int j = 0; j < DIL; ++j) { l[j] = bl[e][j]; r[j] = br[e][j]; }
    if constexpr (!RAW) {
      // zero padding of the ACTIVATED tensor (act(affine(0)) is not 0): the validity of a quad / pair is folded into its
      // scale and shift (both 0 -> act(0) = 0) instead of a select per element
      const unsigned m = bmask[e];
      constexpr int NQ = A2 ? 4 : 2, QE = 8 / NQ;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool ok = (m >> (q * QE)) & 1u;
        const float sq = ok ? b_sc[e] : 0.f, hq = ok ? b_sh[e] : 0.f;
#pragma unroll
        for (int j = 0; j < QE; ++j) v[q * QE + j] = act_by_slope(fmaf(v[q * QE + j], sq, hq), slope);
      }
      const bool okl = m & 0x100u, okr = m & 0x1000u;
      const float sl = okl ? b_sc[e] : 0.f, hl = okl ? b_sh[e] : 0.f, sr = okr ? b_sc[e] : 0.f, hr = okr ? b_sh[e] : 0.f;
#pragma unroll
      for (int j = 0; j < DIL; ++j) {
        l[j] = act_by_slope(fmaf(l[j], sl, hl), slope);
        r[j] = act_by_slope(fmaf(r[j], sr, hr), slope);
      }
    }
    if (BE * NT == BU || tid + NT * e < BU) {
      u32x4 X, S1, S2;
      if constexpr (DIL == 1) {
        const unsigned o12 = bf_pack2(v[1], v[2]), o34 = bf_pack2(v[3], v[4]), o56 = bf_pack2(v[5], v[6]);
        X = u32x4{bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
        S1 = u32x4{bf_pack2(l[0], v[0]), o12, o34, o56};
        S2 = u32x4{o12, o34, o56, bf_pack2(v[7], r[0])};
      } else {
        X = u32x4{bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
        S1 = u32x4{bf_pack2(l[0], l[1]), X.x, X.y, X.z};
        S2 = u32x4{X.y, X.z, X.w, bf_pack2(r[0], r[1])};
      }
      unsigned char* dst = Bb + (pk & 0x1ffffu);
      *reinterpret_cast<u32x4*>(dst) = X;
      *reinterpret_cast<u32x4*>(dst + COPYB) = S1;
      *reinterpret_cast<u32x4*>(dst + 2 * COPYB) = S2;
    }
  };
  // the tile barrier: LDS traffic only.  __syncthreads() is a workgroup fence and compiles to s_waitcnt vmcnt(0), which
  // would drain the loads of the next tile issued during this trip
  auto lds_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // (a row half's offset is folded into the lane base: every fragment address below is base + immediate)
  const int a_lane = (wr * 32 + li) * A_ROW + lk * 16 + hf * RH * TW * 2;
  const int b_lane = (wc * 32 + li) * CH + lk * 16 + hf * RH * RSB;
  // One trip: tile i's MFMAs from LDS buffer `buf`, in NGRP groups (one patch row of one 16-pixel column block each), with
  // the staging work (registers of tile i+1 -> LDS buffer buf^1, reload for tile i+2) spread over the groups IN
  // SOURCE ORDER and pinned there — left to itself the scheduler puts all of it before or after the 72 MFMAs, and with one
  // wave per SIMD that is serial time (matrix pipe busy 26-30 %).
  constexpr int NPR = RH + 2 * DIL, NGRP = (TW / 16) * NPR, NGRP_A = NGRP - BE;
  static_assert(NGRP_A >= 1, "groups");
  auto trip = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* Ap = smem + buf * (A_BYTES + B_BYTES) + a_lane;
    const unsigned char* Bp = Ap - a_lane + A_BYTES + b_lane;
    const TileRef R = next_tile();
    // patch row p meets output rows p, p - DIL, p - 2 DIL through taps kh = 0, 1, 2: its three kw fragments (one aligned
    // ds_read_b128 each, from the three pre-shifted copies) are read ONCE and the dY fragments of the RH rows stay in
    // registers, instead of three reads of every patch row plus funnel shifts for the odd alignments.  The fragments of
    // group g+1 are read before group g's MFMAs (the order is pinned, so nothing else hides the LDS latency).
    bf16x8 af[2][RH], fb[2][3];
    auto read_a = [&](int q, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < RH; ++r) af[slot][r] = *reinterpret_cast<const bf16x8*>(Ap + (r * TW + 16 * q) * 2);
    };
    auto read_b = [&](int g, int slot) __attribute__((always_inline)) {
      const unsigned char* row = Bp + (g % NPR) * RSB + (16 * (g / NPR)) * 2;
      fb[slot][1] = *reinterpret_cast<const bf16x8*>(row);
      fb[slot][0] = *reinterpret_cast<const bf16x8*>(row + COPYB);
      fb[slot][2] = *reinterpret_cast<const bf16x8*>(row + 2 * COPYB);
    };
    read_a(0, 0);
    read_b(0, 0);
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
      const int q = g / NPR, p = g % NPR, cur = g & 1;
      if (g + 1 < NGRP) {
        if ((g + 1) % NPR == 0) read_a(q + 1, (q + 1) & 1);
        read_b(g + 1, cur ^ 1);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int r = p - kh * DIL;
        if (r >= 0 && r < RH) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][r], fb[cur][kw], acc[kh * 3 + kw], 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < AE; ++e)
        if (e * NGRP_A / AE == g) { store_a(buf ^ 1, e); load_a(R, e); }
#pragma unroll
      for (int e = 0; e < BE; ++e)
        if (NGRP_A + e == g) { store_b(buf ^ 1, e); load_b(R, e); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const int nt = t_end - t_begin;
  if (nt > 0) {
    {
      const TileRef R = next_tile();
#pragma unroll
      for (int e = 0; e < AE; ++e) load_a(R, e);
#pragma unroll
      for (int e = 0; e < BE; ++e) load_b(R, e);
    }
    {
      const TileRef R = next_tile();
#pragma unroll
      for (int e = 0; e < AE; ++e) { store_a(0, e); load_a(R, e); }
#pragma unroll
      for (int e = 0; e < BE; ++e) { store_b(0, e); load_b(R, e); }
    }
    lds_barrier();
    for (int i = 0; i < nt; ++i) {          // tile i from LDS buffer i & 1; the registers hold tile i+1 and receive tile i+2
      trip(i & 1);
      lds_barrier();
    }
  }
  // epilogue: row = output channel, MFMA column (lane) = input channel, accumulator = tap; tap-major slab
  const int ci = c0 + wc * 32 + li;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (co < a.Cout && ci < a.Cin) a.out[(((long long)(split * (HALF ? 2 : 1) + hf) * 9 + j) * a.Cout + co) * a.Cin + ci] = acc[j][r];
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
         (long long)(d->Cout > d->Cin ? d->Cout : d->Cin) * d->H * d->W < 0x7fffffffLL &&
         (long long)128 * d->H * d->W * 4 < 0x7fff0000LL;      // a workgroup's channel block as 32-bit byte offsets of its buffer loads
}

struct WBPlan { int tilesX, tilesY, gridM, gridC, splits, tps; bool wide, half; };
static WBPlan wb_plan(const avsep_conv_desc* d) {
  WBPlan p;
  p.wide = d->W > 16;
  p.tilesX = cdiv(d->W, p.wide ? 32 : 16);
  // 16-wide tiles are 8 rows tall: twice the MFMA work per staged tile (the loads of tile t+1 have one tile's MFMAs to
  // arrive in, and a 4x16 tile's 36 MFMAs are shorter than an HBM round trip); the dilated 32-wide patch only fits
  // with 2-row tiles
  p.tilesY = cdiv(d->H, p.wide ? (d->dil == 2 ? 2 : 4) : 8);
  p.half = d->Cout <= 64;
  p.gridM = p.half ? 1 : cdiv(d->Cout, 128);
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
  return (size_t)p.splits * (p.half ? 2 : 1) * d->Cout * d->Cin * 9;          // always through slabs (tap-major) + the transposing reduce
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
#define WB_K(TW_, DIL_, A2_, RAW_, HALF_) \
  hipLaunchKernelGGL((wgradbf_kernel<TW_ == 16 ? 8 : (DIL_ == 2 ? 2 : 4), TW_, DIL_, A2_, RAW_, HALF_>), grid, dim3(256), 0, st, a)
#define WB_L(TW_, DIL_, A2_)                                                                   \
  do {                                                                                         \
    if (raw) { if (p.half) WB_K(TW_, DIL_, A2_, true, true); else WB_K(TW_, DIL_, A2_, true, false); }     \
    else { if (p.half) WB_K(TW_, DIL_, A2_, false, true); else WB_K(TW_, DIL_, A2_, false, false); }       \
  } while (0)
  if (d->dil == 1) {
    if (p.wide) { if (a2) WB_L(32, 1, true); else WB_L(32, 1, false); }
    else { if (a2) WB_L(16, 1, true); else WB_L(16, 1, false); }
  } else {
    if (p.wide) { if (a2) WB_L(32, 2, true); else WB_L(32, 2, false); }
    else { if (a2) WB_L(16, 2, true); else WB_L(16, 2, false); }
  }
#undef WB_L
#undef WB_K
  AVSEP_LAUNCH_CHECK();
  return w3_reduce(ws, dw, (long long)d->Cout * d->Cin, p.splits * (p.half ? 2 : 1), st);
}
