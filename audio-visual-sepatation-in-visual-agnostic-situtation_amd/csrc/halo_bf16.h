// LDS halo-patch convolution with bf16 operands and fp32 accumulation on v_mfma_f32_32x32x16_bf16 (BASELINE.json
// configs[2]: "bf16, full HIP path").  Same problem statement as halo_kernel.h — activations stay dense fp32 NCHW in
// HBM, the folded BatchNorm affine + ReLU/LeakyReLU are applied while the input halo patch of the output tile is
// staged, BatchNorm statistics of the fp32 result are taken in the epilogue — but the operands are rounded to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS and the matrix pipe runs at 16x the f32 rate, so the
// kernel is organised around LDS bandwidth instead of VALU issue slots:
//   * K order = (16-channel chunk, tap): one MFMA k-step = the 16 channels of a chunk at ONE tap, so the lane's 8
//     consecutive k values (channels 8h .. 8h+7) are ONE ds_read_b128 from a channel-innermost patch
//     Ps[position][16 ch] (32 B per position) — the same halo-patch idea, transposed for the bf16 fragment layout;
//   * the two 16-byte halves of a position (and of a weight row) are swapped when bit 3 of the position (row) index is
//     set: lanes l and l+8 of a ds_read_b128 group would otherwise hit the same 4 banks (2-way conflict on every read);
//     the per-tap B addresses (position + tap offset, swizzled) are computed once per lane and kept in registers;
//   * the weight tile arrives by LDS-DMA (global_load_lds_dwordx4) straight from a pre-swizzled bf16 image written by
//     the pack kernel, so it costs no registers, no VALU and no ds_write; only the patch goes through registers
//     (8 fp32 loads -> affine/activation -> 4 packed converts -> one ds_write_b128).
// Single source, no fused upsample (the U-Net decoder materialises relu+upsample, the fused head stays fp32).
//
// Round 4: the staged tensor lives in HBM as bf16 in the channel-blocked layout the patch already has,
//   B16 = [N][C/16][H][W][16 ch]  (32 bytes per position and 16-channel block),
// so a (position, 8-channel half) slot of the patch is ONE 16-byte load and, for an operand that needs no affine /
// activation, one ds_write_b128 of the very same bits: no 8 strided fp32 loads, no converts.  The output is written either
// fp32 NCHW or B16 (C3Args.out16): the packed bf16 quads of a lane pair are exchanged with v_permlane32_swap so that every
// lane stores 16 contiguous bytes (8 channels of one position).  BatchNorm statistics are taken from the fp32 accumulators.
#pragma once
#include <type_traits>
#include "common.h"
#include "halo_kernel.h"   // C3Args (shared argument block)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BF_CK = 16;   // input channels per K-tile = k extent of one 32x32x16 MFMA
constexpr int BF_AFF_MAX = 1024;   // channels of a folded affine kept in LDS (host-checked)

__device__ __forceinline__ unsigned bf_pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  return __builtin_bit_cast(unsigned, v);
}

// Template parameters as halo_kernel.h: TH x TW = NPX output pixels (or FW > 0: NPX consecutive flat pixels of maps of
// width FW), BM output channels per workgroup, KH_ x KW_ taps, stride S, dilation DIL.  NWN = waves along the pixel
// dimension (2 or 4): NPX = 64 * NWN pixels, 128 * NWN threads.  The 512-thread form (BM x 256 tile) halves the weight
// traffic per flop and puts two waves on every SIMD, so one wave's staging overlaps the other's MFMAs.
// RAW: the staged tensor needs no affine and no activation (every data gradient; forward convs over a materialised
// input such as the U-Net decoder's ReLU+upsample tensor): the kernels are VALU-bound in their staging loops (SQ counters:
// 8-18 VALU per MFMA), so the two instructions per element are compiled out rather than multiplied by one.
template <int TH, int TW, int BM, int KH_, int KW_, int S, int DIL, int FW = 0, int NWN = 2, bool RAW = false>
__global__ __launch_bounds__(128 * NWN) void convbf_kernel(C3Args a) {
  constexpr int NTHR = 128 * NWN, NPX = 64 * NWN;
  static_assert(FW > 0 || TH * TW == NPX, "tile = NPX pixels");
  constexpr bool FLAT = FW > 0;
  static_assert(!FLAT || (S == 1 && (KH_ & 1) && (KW_ & 1)), "flat tiles: stride 1, odd taps");
  constexpr int NT = KH_ * KW_;
  constexpr int PADH = DIL * (KH_ - 1) / 2, PADW = DIL * (KW_ - 1) / 2;
  constexpr int F_NR = (FW + NPX - 2 + (FLAT ? FW : 1)) / (FLAT ? FW : 1);
  constexpr int F_NC = (NPX - 1) / (FLAT ? FW * FW : 1) + 1;
  constexpr int PH = FLAT ? F_NR + F_NC * PADH + 2 * PADH : (TH - 1) * S + (KH_ - 1) * DIL + 1;
  constexpr int PWR = FLAT ? FW + 2 * PADW : (TW - 1) * S + (KW_ - 1) * DIL + 1;
  constexpr int PW = (S == 2) ? (PWR + 1) / 2 * 2 : PWR, PWH = PW / 2, PS = PH * PW;   // positions of the patch
  constexpr int NSLOT = 2 * PS, PE = (NSLOT + NTHR - 1) / NTHR;          // (position, 8-channel half) slots per thread
  constexpr int A_BYTES = NT * BM * 32, P_BYTES = PS * 32;          // one buffer of each image
  constexpr int A_CH = A_BYTES / 16, AE = (A_CH + NTHR - 1) / NTHR;       // 16-byte DMA pieces of the weight tile
  static_assert(A_CH % 64 == 0, "whole waves of LDS-DMA pieces");
  constexpr int WTM = BM / 2, TM = WTM / 32;                        // waves 2 (M) x NWN (N); wave N-tile = 64 pixels
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * P_BYTES + 2 * A_BYTES + 2 * BF_AFF_MAX * 4];
  unsigned char* const Pb = smem;                                   // patch first: its per-tap offsets stay small
  unsigned char* const Ab = smem + 2 * P_BYTES;
  float* const aff_sc = reinterpret_cast<float*>(smem + 2 * P_BYTES + 2 * A_BYTES);   // folded BatchNorm rows of all Cin channels
  float* const aff_sh = aff_sc + BF_AFF_MAX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN, li = lane & 31, lk = lane >> 5;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = t % a.gridM; t /= a.gridM;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int n = t / a.tilesY;
  const int h0 = ty * TH, w0 = tx * TW, m0 = mt * BM;
  const int f_HW = a.H * a.W, f_Hp = a.H + PADH, f_P = a.N * f_HW;
  const int p0 = tx * NPX;
  const int vr0 = FLAT ? (p0 / f_HW) * f_Hp + (p0 % f_HW) / FW : 0;
  const long long sHW = (long long)a.Hs * a.Ws;
  const bool has_aff = !RAW && a.sc0 != nullptr;
  const float slope = act_slope(a.act0);

  // ---- patch loader state: slot = (position, channel half g); 8 channels of one position per slot ----------------
  unsigned p_off[PE];                       // 16-byte unit of the slot inside the B16 image of K-tile 0 (< 2^28: checked on the host)
  int p_lds[PE], p_g[PE];
  unsigned pok = 0;
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    const int idx = min(tid + NTHR * e, NSLOT - 1);
    const int g = idx / PS, pos = idx % PS, r = pos / PW, col = pos % PW;
    int gh = h0 * S - a.padh + r, gw = w0 * S - a.padw + col, ne = n;
    if constexpr (FLAT) {
      const int vr = vr0 - PADH + r;
      ne = vr >= 0 ? vr / f_Hp : a.N;
      gh = vr >= 0 ? vr % f_Hp : -1;
      gw = col - PADW;
      if (ne >= a.N) { ne = a.N - 1; gh = -1; }
    }
    const bool ok = (PE * NTHR == NSLOT || tid + NTHR * e < NSLOT) && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
    const int ghc = min(max(gh, 0), a.H - 1), gwc = min(max(gw, 0), a.W - 1);
    p_off[e] = (unsigned)((((long long)ne * (a.C0 >> 4)) * sHW + (long long)ghc * a.Ws + gwc) * 2 + g);
    // stride 2: columns de-interleaved (even | odd) so that the 32 pixels of an MFMA column tile read consecutive positions
    const int lpos = (S == 2) ? r * PW + (col & 1) * PWH + (col >> 1) : pos;
    p_lds[e] = lpos * 32 + ((g ^ (lpos >> 3)) & 1) * 16;
    p_g[e] = g;
    pok |= (unsigned)ok << e;
  }
  // TWO register sets for the patch: the global loads of K-tile kt + 2 are in flight while K-tile kt computes and kt + 1 is
  // converted into LDS.  With one set the loads of kt + 1 were issued at the top of iteration kt and consumed at its bottom:
  // one K-tile of MFMAs (18-36 x 32 cycles) is shorter than an HBM round trip under load, and the waves sat parked on
  // s_waitcnt for 38-50 % of their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r03_convbf_sq_before_prefetch2.txt)
  u32x4 praw[2][PE];
  if (has_aff) {                            // scale / shift rows -> LDS once (kept out of the register pipeline)
    for (int c = tid; c < a.Cin; c += NTHR) {
      aff_sc[c] = a.sc0[c];
      aff_sh[c] = a.sh0[c];
    }
    __syncthreads();
  }

  // weight tile: piece c of K-tile kt lives at  wp + ((kt*NT + c / (2*BM)) * ld + m0) * 32 B + (c % (2*BM)) * 16 B
  const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(a.wp);
  long long a_goff[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    const int c = min(tid + NTHR * e, A_CH - 1), tap = c / (2 * BM), rem = c % (2 * BM);
    a_goff[e] = ((long long)tap * a.wp_ld + m0) * 32 + rem * 16;
  }
  const long long a_step = (long long)NT * a.wp_ld * 32;

  auto issue_w = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      if (AE * NTHR == A_CH || (wave * 64 + NTHR * e) < A_CH) {      // wave-uniform: whole waves only (A_CH % 64 == 0)
        const unsigned char* src = wbase + (long long)kt * a_step + a_goff[e];
        unsigned char* dst = Ab + buf * A_BYTES + (NTHR * e + wave * 64) * 16;    // wave-uniform base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
      }
    }
  };
  auto issue_p = [&](int kt, auto set) __attribute__((always_inline)) {
    constexpr int R = decltype(set)::value;
    const u32x4* xk = reinterpret_cast<const u32x4*>(a.x0) + (long long)kt * sHW * 2;   // uniform: scalar base + 32-bit lane offset
#pragma unroll
    for (int e = 0; e < PE; ++e) praw[R][e] = xk[p_off[e]];
  };
  auto finish = [&](int kt, int buf, auto set) __attribute__((always_inline)) {
    constexpr int R = decltype(set)::value;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      const bool ok = (pok >> e) & 1u;
      u32x4 q = praw[R][e];
      if constexpr (!RAW) {
        f32x4 sc[2], sh[2];
        if (has_aff) {
          const f32x4* sp = reinterpret_cast<const f32x4*>(aff_sc + kt * BF_CK + 8 * p_g[e]);
          const f32x4* hp = reinterpret_cast<const f32x4*>(aff_sh + kt * BF_CK + 8 * p_g[e]);
          sc[0] = sp[0]; sc[1] = sp[1]; sh[0] = hp[0]; sh[1] = hp[1];
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // bf16 -> f32 is a shift: element 2k sits in the low half of word k, element 2k + 1 in the high half
          float x = __builtin_bit_cast(float, (j & 1) ? (q[j >> 1] & 0xffff0000u) : (q[j >> 1] << 16));
          if (has_aff) x = fmaf(x, sc[j >> 2][j & 3], sh[j >> 2][j & 3]);
          v[j] = act_by_slope(x, slope);
        }
        q = u32x4{bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
      }
      if (PE * NTHR == NSLOT || tid + NTHR * e < NSLOT) {
        // zero padding AFTER the activation, on the packed words (the 8 channels of a slot share one position)
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(Pb + buf * P_BYTES + p_lds[e]) = ok ? q : z;
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

  // per-lane operand addresses (bytes).  A: row m = wm*WTM + i*32 + li, half lk, swizzled by bit 3 of the row.
  const int a_lane = (wm * WTM + li) * 32 + ((lk ^ (li >> 3)) & 1) * 16;       // + i*32*32 + tap*BM*32 (immediates)
  int b_addr[NT][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = wn * 64 + j * 32 + li;
    int lb;
    if constexpr (FLAT) {
      const int pg = min(p0 + p, f_P - 1);
      lb = ((pg / f_HW) * f_Hp + (pg % f_HW) / FW - vr0) * PW + pg % FW;
    } else {
      lb = (p / TW) * S * PW + (p % TW);
    }
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int kh = tap / KW_, kw = tap % KW_;
      const int pos = lb + kh * DIL * PW + (S == 2 ? (kw & 1) * PWH + (kw >> 1) : kw * DIL);
      b_addr[tap][j] = pos * 32 + ((lk ^ (pos >> 3)) & 1) * 16;
    }
  }

  int kt0 = 0, nK = a.Cin / BF_CK;
  if (a.kts > 0) {                                  // split-K: this block reduces K-tiles [kt0, nK) of its slice
    kt0 = blockIdx.y * a.kts;
    nK = min(nK, kt0 + a.kts);
  }
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  if (kt0 < nK) {
    issue_w(kt0, 0);
    issue_p(kt0, Set0{});
    if (kt0 + 1 < nK) issue_p(kt0 + 1, Set1{});
    finish(kt0, 0, Set0{});
  }
  __syncthreads();
  // one K-tile: `nxt` holds the patch of kt + 1 (loaded an iteration ago), `far` receives the patch of kt + 2
  auto ktile = [&](int kt, auto nxt, auto far) __attribute__((always_inline)) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < nK) issue_w(kt + 1, buf ^ 1);
    const unsigned char* Ak = Ab + buf * A_BYTES + a_lane;
    const unsigned char* Pk = Pb + buf * P_BYTES;
    bf16x8 av[2][TM], bv[2][2];
    auto read_ops = [&](int tap, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[slot][i] = *reinterpret_cast<const bf16x8*>(Ak + tap * BM * 32 + i * 32 * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[slot][j] = *reinterpret_cast<const bf16x8*>(Pk + b_addr[tap][j]);
    };
    read_ops(0, 0);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < NT) read_ops(tap + 1, cur ^ 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nK) finish(kt + 1, buf ^ 1, nxt);
    // The barrier of the K-tile.  __syncthreads() is a workgroup-scope fence: on gfx9 loads and stores share vmcnt, so it
    // compiles to s_waitcnt vmcnt(0) and would drain the far patch loads issued just now.  What has to be complete here is
    // (a) this wave's LDS stores (lgkmcnt) and (b) its LDS-DMA pieces of the next weight tile, which are OLDER than the
    // PE far loads: a counted vmcnt leaves exactly those in flight.
    if (kt + 2 < nK) {
      issue_p(kt + 2, far);                                // `far` is the set the previous K-tile's finish() freed
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PE) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  };
  for (int kt = kt0; kt < nK; kt += 2) {
    ktile(kt, Set1{}, Set0{});
    if (kt + 1 < nK) ktile(kt + 1, Set0{}, Set1{});
  }

  // ---- epilogue: identical to halo_kernel.h (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) ----
  const long long HW = (long long)a.OHs * a.OWs;
  long long cbase[2];
  bool cok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = wn * 64 + j * 32 + li;
    if constexpr (FLAT) {
      const int pg = p0 + p;
      cok[j] = pg < f_P;
      cbase[j] = (long long)(pg / f_HW) * a.Cout * HW + pg % f_HW;
    } else {
      const int gh = h0 + p / TW, gw = w0 + p % TW;
      cok[j] = gh < a.Ho && gw < a.Wo;
      cbase[j] = (long long)n * a.Cout * HW + (long long)(gh * a.os + a.ooh) * a.OWs + (gw * a.os + a.oow);
    }
  }
  const bool want_stats = a.stats != nullptr && a.kts == 0;
  const bool out16 = a.out16 && a.kts == 0;                  // split-K slabs stay fp32 (the combine decides the final format)
  float* const outp = a.out + (a.kts > 0 ? (long long)blockIdx.y * a.slab : 0);   // partial slab: bias / statistics in the combine
  float* s_sum = reinterpret_cast<float*>(Ab);               // [NWN][BM] per-wave-column partial sums (operands are dead)
  float* s_sq = s_sum + NWN * BM;
  // B16 output: 32-byte unit index of (image, position) in the block-0 plane; + (row >> 4) * HW per 16-channel block
  long long cb16[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = wn * 64 + j * 32 + li;
    if constexpr (FLAT) {
      const int pg = p0 + p;
      cb16[j] = (long long)(pg / f_HW) * (a.Cout >> 4) * HW + pg % f_HW;
    } else {
      const int gh = h0 + p / TW, gw = w0 + p % TW;
      cb16[j] = (long long)n * (a.Cout >> 4) * HW + (long long)(gh * a.os + a.ooh) * a.OWs + (gw * a.os + a.oow);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float vv[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lrow = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int row = m0 + lrow;
      const bool rok = row < a.Cout;
      const float bias = (a.bias && rok && a.kts == 0) ? a.bias[row] : 0.f;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v = acc[i][j][r] + bias;
        vv[r][j] = v;
        if (rok && cok[j]) {
          if (!out16) outp[cbase[j] + (long long)row * HW] = v;
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
    if (out16) {
      // rows of a 32-row block: lane half lk holds (r & 3) + 8 * (r >> 2) + 4 * lk, i.e. per 16-channel block hb the quads
      // [4 lk .. 4 lk + 3] (r = 8 hb + 0..3) and [8 + 4 lk ..] (r = 8 hb + 4..7).  permlane32_swap(vdst = first quad,
      // src = second quad) leaves lanes 0-31 with channels 0-7 and lanes 32-63 with channels 8-15 of the block: one
      // 16-byte store per lane, 32 contiguous bytes per lane pair (cdna_hip_programming.md T21)
      unsigned char* const ob = reinterpret_cast<unsigned char*>(a.out);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int row0 = m0 + wm * WTM + i * 32 + 16 * hb;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned a0 = bf_pack2(vv[8 * hb + 0][j], vv[8 * hb + 1][j]), a1 = bf_pack2(vv[8 * hb + 2][j], vv[8 * hb + 3][j]);
          unsigned b0 = bf_pack2(vv[8 * hb + 4][j], vv[8 * hb + 5][j]), b1 = bf_pack2(vv[8 * hb + 6][j], vv[8 * hb + 7][j]);
          auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          if (row0 < a.Cout && cok[j]) {
            const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
            *reinterpret_cast<u32x4*>(ob + ((cb16[j] + (long long)(row0 >> 4) * HW) * 32 + lk * 16)) = o;
          }
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    for (int rr = tid; rr < BM; rr += NTHR) {
      const int row = m0 + rr;
      if (row < a.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NWN; ++w) { s += s_sum[w * BM + rr]; q += s_sq[w * BM + rr]; }
        atomicAdd(&a.stats[row], (double)s);
        atomicAdd(&a.stats[a.Cout + row], (double)q);
      }
    }
  }
}
