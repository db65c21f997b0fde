// convbf_res_kernel: the bf16 halo-patch convolution for SHORT reductions with <= 64 output rows — the 64 -> 64 3x3 convs of
// ResNet layer1 (forward and data gradient: torchvision BasicBlock via vision_net.py:84-92) and the stem in its 4x4 / stride-1
// form over the space-to-depth frames (Cin = 16) — as a PERSISTENT workgroup with the whole weight set resident in LDS.
// In convbf_kernel (halo_bf16.h) these calls have 1-4 K-tiles: a workgroup lives for 36-72 MFMAs per wave and spends the rest
// of its life on index setup, its first memory round trip, the K-tile barriers and the epilogue (SQ: 0.12-0.39 MFMA-busy,
// 8-30 VALU per MFMA).  Here a workgroup
//   * copies ALL K-tiles of the (pre-swizzled, bf16) weight image into LDS once (<= 72 KB),
//   * walks pixel tiles t = blockIdx.x, + gridDim.x, ...: the patch of ALL input channels of a tile is one LDS image
//     ([K-tile][position][16 ch], same swizzle as halo_bf16.h), so the MFMA loop over (K-tile, tap) runs without a barrier;
//   * has the NEXT tile's patch in flight in registers while the current one is multiplied (one barrier pair per tile);
//   * keeps the BatchNorm sums of its tiles in registers and issues one pair of atomics per channel at the very end.
// Operand layout, swizzle, B16 input / output and the C/D handling are those of halo_bf16.h.
#pragma once
#include "halo_bf16.h"

template <int TH, int TW, int KH_, int KW_, int NKT, bool RAW>
__global__ __launch_bounds__(512) void convbf_res_kernel(C3Args a) {
  constexpr int NTHR = 512, BM = 64, NT = KH_ * KW_, NPX = 256;
  static_assert(TH * TW == NPX, "tile = 256 pixels: 8 waves = 2 (rows) x 4 (64 pixels)");
  constexpr int PH = TH + KH_ - 1, PW = TW + KW_ - 1, PS = PH * PW;
  constexpr int P_BYTES = PS * 32, A_BYTES = NT * BM * 32;                 // per K-tile
  constexpr int NSLOT = NKT * PS * 2, PE = (NSLOT + NTHR - 1) / NTHR;
  constexpr int A_CH = NKT * A_BYTES / 16, AE = (A_CH + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NKT * P_BYTES + NKT * A_BYTES + 2 * 64 * NKT * 4 + 2 * 4 * BM * 4];
  unsigned char* const Pb = smem;
  unsigned char* const Ab = smem + NKT * P_BYTES;
  float* const aff_sc = reinterpret_cast<float*>(Ab + NKT * A_BYTES);     // [16 * NKT] (padded to 64 * NKT floats)
  float* const aff_sh = aff_sc + 64 * NKT;
  float* const s_sum = aff_sh + 64 * NKT;                                  // [4][BM] per-wave-column sums (end of kernel)
  float* const s_sq = s_sum + 4 * BM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lk = lane >> 5;
  const bool has_aff = !RAW && a.sc0 != nullptr;
  const float slope = act_slope(a.act0);
  const int CBi = a.Cin >> 4;

  // ---- weights: K-tile kt, tap, row m of the packed image live at wp + ((kt*NT + tap) * ld + m) * 32 B ------------------------
  {
    const u32x4* const wsrc = reinterpret_cast<const u32x4*>(a.wp);
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const int c = tid + NTHR * e;
      if (AE * NTHR == A_CH || c < A_CH) {
        const int kt_tap = c / (2 * BM), rem = c % (2 * BM);              // 2*BM 16-byte pieces per (K-tile, tap)
        reinterpret_cast<u32x4*>(Ab)[c] = wsrc[(long long)kt_tap * a.wp_ld * 2 + rem];
      }
    }
    if (has_aff)
      for (int c = tid; c < a.Cin; c += NTHR) { aff_sc[c] = a.sc0[c]; aff_sh[c] = a.sh0[c]; }
  }

  // ---- patch slots of this thread: (K-tile, position, half) -> offset relative to (image, patch origin), LDS byte ------------------
  int p_rc[PE], p_lds[PE], p_g[PE];
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    const int i = min(tid + NTHR * e, NSLOT - 1);
    const int kt = i / (PS * 2), rem = i % (PS * 2), pos = rem >> 1, g = rem & 1;
    p_rc[e] = ((pos / PW) << 16) | (pos % PW);
    p_lds[e] = kt * P_BYTES + pos * 32 + ((g ^ (pos >> 3)) & 1) * 16;
    p_g[e] = (PE * NTHR == NSLOT || tid + NTHR * e < NSLOT) ? ((kt * a.H + pos / PW) * a.W + pos % PW) * 2 + g : -1;
  }
  u32x4 praw[PE];
  unsigned pok = 0;
  const int tiles_per_img = a.tilesX * a.tilesY, T = a.N * tiles_per_img;
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int tx = t % a.tilesX, t2 = t / a.tilesX, ty = t2 % a.tilesY, n = t2 / a.tilesY;
    const int gh0 = ty * TH - a.padh, gw0 = tx * TW - a.padw;
    const u32x4* const xn = reinterpret_cast<const u32x4*>(a.x0) + ((long long)n * CBi * a.H * a.W + (long long)gh0 * a.W + gw0) * 2;
    const u32x4 z = {0u, 0u, 0u, 0u};
    pok = 0;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      const bool ok = p_g[e] >= 0 && (unsigned)(gh0 + (p_rc[e] >> 16)) < (unsigned)a.H && (unsigned)(gw0 + (p_rc[e] & 0xffff)) < (unsigned)a.W;
      praw[e] = ok ? xn[p_g[e]] : z;
      pok |= (unsigned)ok << e;
    }
  };
  auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      u32x4 q = praw[e];
      if constexpr (!RAW) {
        const int c8 = (p_lds[e] / P_BYTES) * 16 + ((p_g[e] & 1) * 8);   // first channel of the slot
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(aff_sc + c8), s1 = *reinterpret_cast<const f32x4*>(aff_sc + c8 + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(aff_sh + c8), h1 = *reinterpret_cast<const f32x4*>(aff_sh + c8 + 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x = __builtin_bit_cast(float, (j & 1) ? (q[j >> 1] & 0xffff0000u) : (q[j >> 1] << 16));
          if (has_aff) x = fmaf(x, j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? h0[j & 3] : h1[j & 3]);
          v[j] = act_by_slope(x, slope);
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
        q = ((pok >> e) & 1u) ? u32x4{bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])} : z;
      }
      if (PE * NTHR == NSLOT || tid + NTHR * e < NSLOT) *reinterpret_cast<u32x4*>(Pb + p_lds[e]) = q;
    }
  };

  // ---- operand addresses (halo_bf16.h): A row m = wm*32 + li, half lk swizzled by bit 3 of the row; B position + tap -----------
  const int a_lane = (wm * 32 + li) * 32 + ((lk ^ (li >> 3)) & 1) * 16;
  int b_addr[NT][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = wn * 64 + j * 32 + li;
    const int lb = (p / TW) * PW + (p % TW);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int pos = lb + (tap / KW_) * PW + tap % KW_;
      b_addr[tap][j] = pos * 32 + ((lk ^ (pos >> 3)) & 1) * 16;
    }
  }
  const bool want_stats = a.stats != nullptr;
  const long long HW = (long long)a.OHs * a.OWs;
  float st_s[16], st_q[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) st_s[r] = st_q[r] = 0.f;

  int t = blockIdx.x;
  if (t < T) issue(t);
  for (; t < T; t += gridDim.x) {
    __syncthreads();                       // every wave is done with the previous tile's patch (and, first trip, the weights are written)
    stage();
    __syncthreads();
    const int tcur = t;
    if (t + (int)gridDim.x < T) issue(t + gridDim.x);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(Ab + (kt * NT + tap) * BM * 32 + a_lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(Pb + kt * P_BYTES + b_addr[tap][j]);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
        }
      }
    }
    // ---- epilogue of the tile (C/D map: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) ---------------------------
    const int tx = tcur % a.tilesX, t2 = tcur / a.tilesX, ty = t2 % a.tilesY, n = t2 / a.tilesY;
    bool cok[2];
    long long cbase[2], cb16[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = wn * 64 + j * 32 + li;
      const int gh = ty * TH + p / TW, gw = tx * TW + p % TW;
      cok[j] = gh < a.Ho && gw < a.Wo;
      cbase[j] = (long long)n * a.Cout * HW + (long long)gh * a.OWs + gw;
      cb16[j] = (long long)n * (a.Cout >> 4) * HW + (long long)gh * a.OWs + gw;
    }
    float vv[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const bool rok = row < a.Cout;
      const float bias = (a.bias && rok) ? a.bias[row] : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v = acc[j][r] + bias;
        vv[r][j] = v;
        if (rok && cok[j]) {
          if (!a.out16) a.out[cbase[j] + (long long)row * HW] = v;
          st_s[r] += v;
          st_q[r] = fmaf(v, v, st_q[r]);
        }
      }
    }
    if (a.out16) {
      unsigned char* const ob = reinterpret_cast<unsigned char*>(a.out);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int row0 = wm * 32 + 16 * hb;
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
  // ---- BatchNorm sums of all the tiles this workgroup computed: lanes of a 32-lane half -> LDS -> one atomic pair per channel -----
  if (want_stats) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s = half_sum_hi(st_s[r]), q = half_sum_hi(st_q[r]);
      if (li == 31) {
        const int lrow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        s_sum[wn * BM + lrow] = s;
        s_sq[wn * BM + lrow] = q;
      }
    }
    __syncthreads();
    for (int rr = tid; rr < BM; rr += NTHR)
      if (rr < a.Cout) {
        const float s = (s_sum[rr] + s_sum[BM + rr]) + (s_sum[2 * BM + rr] + s_sum[3 * BM + rr]);
        const float q = (s_sq[rr] + s_sq[BM + rr]) + (s_sq[2 * BM + rr] + s_sq[3 * BM + rr]);
        atomicAdd(&a.stats[rr], (double)s);
        atomicAdd(&a.stats[a.Cout + rr], (double)q);
      }
  }
}
