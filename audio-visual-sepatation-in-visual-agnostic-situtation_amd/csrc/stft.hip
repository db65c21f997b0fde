// STFT / iSTFT with librosa semantics (dataset/base.py:142-147, utils.py:101-104) as a strided
// 1-D convolution of the padded waveform with the windowed DFT basis: a [2*bins x n_fft] x
// [n_fft x frames] GEMM per row on the f32 MFMA (conv.hip), fused pad / magnitude-phase kernels.
#include "common.h"

static inline int bins_of(int n_fft) { return n_fft / 2 + 1; }

extern "C" size_t avsep_stft_basis_floats(int32_t n_fft, int32_t inverse) {
  if (n_fft < 2 || (n_fft & 1)) return 0;
  int b2 = 2 * bins_of(n_fft);
  return inverse ? (size_t)roundup(b2, 32) * roundup(n_fft, 128) : (size_t)roundup(n_fft, 32) * roundup(b2, 128);
}

// forward: packed conv operand [k = sample n][m = bin | bins+bin] = hann[n] * (cos | -sin)(2 pi m n / N)
// inverse: packed 1x1-conv operand [k = bin | bins+bin][m = sample n] = hann[n] * c_k * (cos | -sin) / N
__global__ void stft_basis_kernel(int n_fft, int bins, int rows, int ld, int inverse, float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ld) return;
  int row = (int)(i / ld), col = (int)(i % ld);
  int n = inverse ? col : row, m = inverse ? row : col;
  float v = 0.f;
  if (n < n_fft && m < 2 * bins) {
    int bin = m < bins ? m : m - bins;
    double win = 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)n_fft);
    long long prod = ((long long)bin * n) % n_fft;  // exact phase reduction
    double ang = 2.0 * M_PI * (double)prod / (double)n_fft;
    double tr = (m < bins) ? cos(ang) : -sin(ang);
    if (inverse) {
      double ck = (bin == 0 || bin == n_fft / 2) ? 1.0 : 2.0;
      if (m >= bins && (bin == 0 || bin == n_fft / 2)) ck = 0.0;  // irfft ignores imag of DC / Nyquist
      tr *= ck / (double)n_fft;
    }
    v = (float)(win * tr);
  }
  out[i] = v;
}

extern "C" int avsep_stft_basis(int32_t n_fft, float* fwd_basis, float* inv_basis, avsep_stream_t stream) {
  if (n_fft < 2 || (n_fft & 1)) return AVSEP_ERR_ARG;
  int bins = bins_of(n_fft), b2 = 2 * bins;
  if (fwd_basis) {
    int rows = roundup(n_fft, 32), ld = roundup(b2, 128);
    hipLaunchKernelGGL(stft_basis_kernel, dim3(cdiv((long long)rows * ld, 256)), dim3(256), 0, (hipStream_t)stream, n_fft,
                       bins, rows, ld, 0, fwd_basis);
    AVSEP_LAUNCH_CHECK();
  }
  if (inv_basis) {
    int rows = roundup(b2, 32), ld = roundup(n_fft, 128);
    hipLaunchKernelGGL(stft_basis_kernel, dim3(cdiv((long long)rows * ld, 256)), dim3(256), 0, (hipStream_t)stream, n_fft,
                       bins, rows, ld, 1, inv_basis);
    AVSEP_LAUNCH_CHECK();
  }
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void stft_pad_kernel(const float* __restrict__ wav, int L, int pad, int reflect,
                                                       float* __restrict__ out) {
  const int r = blockIdx.y, Lp = L + 2 * pad;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Lp; i += gridDim.x * 256) {
    int j = i - pad;
    float v = 0.f;
    if (j >= 0 && j < L) v = wav[(long long)r * L + j];
    else if (reflect) {
      if (j < 0) j = -j;                 // numpy 'reflect': edge sample not repeated
      if (j >= L) j = 2 * (L - 1) - j;
      if (j >= 0 && j < L) v = wav[(long long)r * L + j];
    }
    out[(long long)r * Lp + i] = v;
  }
}

// hop-transposed padded waveforms for the 1x4-conv form: XT[c][r][j] = padded_r[hop*j + c] (reflect / zero padding
// applied on the fly, 0 past the end).  Block = (row r, 32 hops): coalesced 32*hop-sample read, LDS transpose,
// 128-byte writes along j.
__global__ __launch_bounds__(256) void stft_pad_t_kernel(const float* __restrict__ wav, int L, int pad, int reflect, int hop,
                                                         int R, int NH, float* __restrict__ xt) {
  extern __shared__ float tile[];                         // [32][hop + 1]
  const int r = blockIdx.y, j0 = blockIdx.x * 32, Lp = L + 2 * pad, ldt = hop + 1;
  for (int i = threadIdx.x; i < 32 * hop; i += 256) {
    const int jj = i / hop, c = i % hop, pos = (j0 + jj) * hop + c;
    float v = 0.f;
    if (pos < Lp) {
      int s = pos - pad;
      if (s >= 0 && s < L) v = wav[(long long)r * L + s];
      else if (reflect) {
        if (s < 0) s = -s;
        if (s >= L) s = 2 * (L - 1) - s;
        if (s >= 0 && s < L) v = wav[(long long)r * L + s];
      }
    }
    tile[jj * ldt + c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * hop; i += 256) {
    const int c = i / 32, jj = i % 32;
    if (j0 + jj < NH) xt[((long long)c * R + r) * NH + j0 + jj] = tile[jj * ldt + c];
  }
}

// basis [k = sample][m] -> halo-patch operand rows ((c/2 * NTAP + j) * 2 + c%2), k = hop*j + c
__global__ void stft_basis_repack_kernel(const float* __restrict__ basis, int n_fft, int hop, int ntap, int ld,
                                         float* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)hop * ntap * ld) return;
  const int row = (int)(i / ld), col = (int)(i % ld);
  const int par = row & 1, q = row >> 1, j = q % ntap, c = 2 * (q / ntap) + par, k = hop * j + c;
  out[i] = k < n_fft ? basis[(long long)k * ld + col] : 0.f;
}

// co_major: spec is [2*bins][R][frames] (1x4-conv form) instead of [R][2*bins][frames]
__global__ __launch_bounds__(256) void stft_magphase_kernel(const float* __restrict__ spec, int bins, int frames, int R,
                                                            int co_major, float* __restrict__ mag,
                                                            float* __restrict__ phase) {
  const int r = blockIdx.y;
  const long long n = (long long)bins * frames;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float a, b;
    if (co_major) {
      const long long m = i / frames, f = i % frames;
      a = spec[(m * R + r) * frames + f];
      b = spec[((bins + m) * R + r) * frames + f];
    } else {
      a = spec[(long long)r * 2 * n + i];
      b = spec[(long long)r * 2 * n + n + i];
    }
    mag[(long long)r * n + i] = sqrtf(a * a + b * b);
    if (phase) phase[(long long)r * n + i] = atan2f(b, a);
  }
}

int c1x4_stft_fwd(const float* xt, const float* wp, float* out, int R, int NH, int hop, int cout, int frames,
                  hipStream_t st);   // conv3x3.hip

// the 1x4-conv form needs n_fft <= 4 hops (librosa's default 1022/256 does), whole channel pairs and room for a tile
static bool stft_fast(int R, int L, int n_fft, int hop) {
  return n_fft > 3 * hop && n_fft <= 4 * hop && (hop & 3) == 0 && hop <= 1024 && 1 + L / hop >= 32 && R >= 1;
}
extern "C" size_t avsep_stft_workspace_bytes(int32_t R, int32_t L, int32_t n_fft, int32_t hop) {
  if (R <= 0 || L <= 0 || n_fft < 2 || hop <= 0) return 0;
  size_t frames = 1 + L / hop;
  size_t spec = (size_t)R * 2 * bins_of(n_fft) * frames;
  if (stft_fast(R, L, n_fft, hop))   // hop-transposed input + repacked basis + spectrum
    return ((size_t)hop * R * (frames + 3) + (size_t)hop * 4 * roundup(2 * bins_of(n_fft), 128) + spec) * sizeof(float);
  return ((size_t)R * (L + n_fft) + spec) * sizeof(float);
}

extern "C" int avsep_stft_mag(const float* wav, int32_t R, int32_t L, int32_t n_fft, int32_t hop, int32_t reflect,
                              const float* basis, float* mag, float* phase, void* workspace, size_t workspace_bytes,
                              avsep_stream_t stream) {
  if (!wav || !basis || !mag || R <= 0 || R > 65535 || L <= n_fft / 2 || n_fft < 2 || (n_fft & 1) || hop <= 0)
    return AVSEP_ERR_ARG;
  if (!workspace || workspace_bytes < avsep_stft_workspace_bytes(R, L, n_fft, hop)) return AVSEP_ERR_WORKSPACE;
  const int pad = n_fft / 2, Lp = L + 2 * pad, bins = bins_of(n_fft), frames = 1 + L / hop;
  hipStream_t st = (hipStream_t)stream;
  if (stft_fast(R, L, n_fft, hop)) {
    const int NH = frames + 3, ld = roundup(2 * bins, 128);
    float* xt = (float*)workspace;
    float* wp = xt + (size_t)hop * R * NH;
    float* spec = wp + (size_t)hop * 4 * ld;
    hipLaunchKernelGGL(stft_pad_t_kernel, dim3(cdiv(NH, 32), R), dim3(256), (size_t)32 * (hop + 1) * sizeof(float), st, wav, L,
                       pad, reflect, hop, R, NH, xt);
    AVSEP_LAUNCH_CHECK();
    hipLaunchKernelGGL(stft_basis_repack_kernel, dim3(cdiv((long long)hop * 4 * ld, 256)), dim3(256), 0, st, basis, n_fft, hop,
                       4, ld, wp);
    AVSEP_LAUNCH_CHECK();
    int rc = c1x4_stft_fwd(xt, wp, spec, R, NH, hop, 2 * bins, frames, st);
    if (rc) return rc;
    long long n = (long long)bins * frames;
    hipLaunchKernelGGL(stft_magphase_kernel, dim3((int)min((n + 255) / 256, (long long)1024), R), dim3(256), 0, st, spec, bins,
                       frames, R, 1, mag, phase);
    AVSEP_LAUNCH_CHECK();
    return AVSEP_OK;
  }
  float* padded = (float*)workspace;
  float* spec = padded + (size_t)R * Lp;
  hipLaunchKernelGGL(stft_pad_kernel, dim3(min(cdiv(Lp, 256), 1024), R), dim3(256), 0, st, wav, L, pad, reflect, padded);
  AVSEP_LAUNCH_CHECK();
  avsep_conv_desc d{};
  d.N = R; d.Cin = 1; d.H = 1; d.W = Lp; d.Cout = 2 * bins; d.Ho = 1; d.Wo = frames;
  d.KH = 1; d.KW = n_fft; d.stride = hop; d.pad = 0; d.dil = 1; d.C0 = 1; d.x0 = padded;
  int rc = avsep_conv2d_fwd(&d, basis, nullptr, spec, nullptr, nullptr, 0, stream);
  if (rc) return rc;
  long long n = (long long)bins * frames;
  hipLaunchKernelGGL(stft_magphase_kernel, dim3((int)min((n + 255) / 256, (long long)1024), R), dim3(256), 0, st, spec, bins,
                     frames, R, 0, mag, phase);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}

__global__ __launch_bounds__(256) void istft_spec_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                         long long n, float* __restrict__ spec) {
  const int r = blockIdx.y;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float m = mag[(long long)r * n + i], p = phase[(long long)r * n + i];
    float s, c;
    sincosf(p, &s, &c);
    spec[(long long)r * 2 * n + i] = m * c;
    spec[(long long)r * 2 * n + n + i] = m * s;
  }
}

// overlap-add with window-sum-square normalisation (librosa.istft), centre trim
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ td, int n_fft, int hop, int frames,
                                                        int out_len, float* __restrict__ wav) {
  const int r = blockIdx.y, pad = n_fft / 2;
  const float* p = td + (long long)r * n_fft * frames;  // [n_fft][frames]
  for (int t = blockIdx.x * 256 + threadIdx.x; t < out_len; t += gridDim.x * 256) {
    int g = t + pad;  // position in the untrimmed signal
    int f_hi = min(frames - 1, g / hop), f_lo = max(0, (g - n_fft + hop) / hop);
    float acc = 0.f, wss = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      int n = g - f * hop;
      if (n < 0 || n >= n_fft) continue;
      float w = 0.5f - 0.5f * cosf(2.f * (float)M_PI * (float)n / (float)n_fft);
      acc += p[(long long)n * frames + f];  // synthesis window already folded into the basis
      wss += w * w;
    }
    wav[(long long)r * out_len + t] = wss > 1.17549435e-38f ? acc / wss : acc;
  }
}

extern "C" size_t avsep_istft_workspace_bytes(int32_t R, int32_t n_fft, int32_t frames) {
  if (R <= 0 || n_fft < 2 || frames <= 0) return 0;
  return ((size_t)R * 2 * bins_of(n_fft) * frames + (size_t)R * n_fft * frames) * sizeof(float);
}

extern "C" int avsep_istft(const float* mag, const float* phase, int32_t R, int32_t n_fft, int32_t hop, int32_t frames,
                           const float* inv_basis, float* wav, int32_t out_len, void* workspace, size_t workspace_bytes,
                           avsep_stream_t stream) {
  if (!mag || !phase || !inv_basis || !wav || R <= 0 || R > 65535 || n_fft < 2 || (n_fft & 1) || hop <= 0 || frames <= 0)
    return AVSEP_ERR_ARG;
  if (out_len <= 0 || out_len > hop * (frames - 1)) return AVSEP_ERR_ARG;
  if (!workspace || workspace_bytes < avsep_istft_workspace_bytes(R, n_fft, frames)) return AVSEP_ERR_WORKSPACE;
  const int bins = bins_of(n_fft);
  float* spec = (float*)workspace;
  float* td = spec + (size_t)R * 2 * bins * frames;
  hipStream_t st = (hipStream_t)stream;
  long long n = (long long)bins * frames;
  hipLaunchKernelGGL(istft_spec_kernel, dim3((int)min((n + 255) / 256, (long long)1024), R), dim3(256), 0, st, mag, phase, n,
                     spec);
  AVSEP_LAUNCH_CHECK();
  avsep_conv_desc d{};
  d.N = R; d.Cin = 2 * bins; d.H = 1; d.W = frames; d.Cout = n_fft; d.Ho = 1; d.Wo = frames;
  d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0; d.dil = 1; d.C0 = 2 * bins; d.x0 = spec;
  int rc = avsep_conv2d_fwd(&d, inv_basis, nullptr, td, nullptr, nullptr, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(min(cdiv(out_len, 256), 1024), R), dim3(256), 0, st, td, n_fft, hop, frames,
                     out_len, wav);
  AVSEP_LAUNCH_CHECK();
  return AVSEP_OK;
}
