// pcgmix_logmel.hip — per-cycle STFT -> log-mel front end on gfx950.
//
// Replaces the offline librosa pipeline of databuilder.ipynb cell 6:81-101, 127-142
// (melspectrogram(n_fft = 4*hop, hop, n_mels, fmin, fmax) -> power_to_db(ref=max) ->
// (x - mean)/std -> keep the cycle's columns -> zero-pad to W), librosa 0.9.2 semantics restated:
// centred frames with reflect padding, periodic Hann of n_fft, float64 transform rounded to
// complex64, |.|^2 in float32, Slaney mel filter bank (float32 weights), float32 dB with
// amin = 1e-10 and top_db = 80.  One difference is inherent to doing this per batch item: `ref`
// is the maximum over the item's own spectrogram, not over the whole recording.
//
// The STFT is a GEMM and librosa evaluates it in float64 — so it runs on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64).  The input is real and the periodic Hann window is symmetric
// (win[N-k] = win[k], win[0] = 0), so with k = 1..N/2
//     re[bin][frame] = sum_k  win[k] cos(2 pi bin k / N) * (x[f*hop + k] + x[f*hop + N - k])
//     im[bin][frame] = sum_k -win[k] sin(2 pi bin k / N) * (x[f*hop + k] - x[f*hop + N - k])
// (the k = N/2 term carries weight 1/2 because its partner is itself): two GEMMs with M = bins,
// N = frames and K = N/2 instead of one with K = N — half the matrix-core work.
// Everything that does not depend on the data (the windowed twiddle matrices already in MFMA
// A-fragment order, the mel filter bank, each filter's non-zero span) is a constant table built
// once on the host (pcgmix_logmel_tables) and read through L2.  One block per sample; LDS holds the reflect-padded row, the power spectrogram and
// the whole dB image, so HBM sees the input once and the output once:
// 4*T + 4*n_mels*W algorithmic bytes per sample.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kMelThreads = 1024;
constexpr int kMelWaves = kMelThreads / 64;
constexpr int kNGroup = 2;  // 16-frame tiles per work unit of a wave (re + im accumulators: 4 x v4f64)

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- constant tables (host-built blob) ---------------------------------------------------------
//   [0]            double afrag[m_tiles][ksteps][2][64]   A operands (re, im), one double per lane:
//                  lane l -> bin = 16*mt + (l & 15), k = 4*ks + (l >> 4) + 1  (k = 1 .. n_fft/2)
//                  re:  win[k] * cos(2 pi bin k / n_fft) * (k == n_fft/2 ? 0.5 : 1)
//                  im: -win[k] * sin(2 pi bin k / n_fft);   0 for padded bins / k > n_fft/2
//   [off_wts]      float  wts[n_mels][n_bins]             librosa.filters.mel, slaney, float32
//   [off_krange]   int32  krange[n_mels][2]               first / last non-zero bin
struct MelTables {
  int m_tiles, ksteps, n_bins;
  size_t off_wts, off_krange, total;
};
__host__ __device__ inline MelTables mel_tables(int n_fft, int n_mels) {
  MelTables t;
  t.n_bins = n_fft / 2 + 1;
  t.m_tiles = (t.n_bins + 15) / 16;
  t.ksteps = (n_fft / 2 + 3) / 4;
  size_t o = (size_t)t.m_tiles * t.ksteps * 2 * 64 * sizeof(double);
  t.off_wts = o;
  o += (size_t)n_mels * t.n_bins * sizeof(float);
  o = (o + 7) & ~(size_t)7;
  t.off_krange = o;
  o += (size_t)n_mels * 2 * sizeof(int32_t);
  t.total = (o + 15) & ~(size_t)15;
  return t;
}

struct MelLayout {  // byte offsets into dynamic LDS
  int xrow, ps, img, melw, total;
  int nfp, xr;
};
__host__ __device__ inline MelLayout mel_layout(int T, int n_fft, int hop, int n_mels, int W) {
  const MelTables tb = mel_tables(n_fft, n_mels);
  const int n_frames = 1 + T / hop;
  MelLayout L;
  L.nfp = ((n_frames + 16 * kNGroup - 1) / (16 * kNGroup)) * (16 * kNGroup);  // frames, padded
  L.xr = (L.nfp - 1) * hop + n_fft + 8;      // padded-row samples the GEMM may touch
  if (L.xr < T + n_fft) L.xr = T + n_fft;
  L.xr = (L.xr + 3) & ~3;
  int o = 0;
  L.xrow = o; o += L.xr * 4;                    // reflect-padded waveform, zero beyond (float)
  L.ps = o;   o += tb.m_tiles * 16 * L.nfp * 4; // power spectrogram [bin][frame]        (float)
  L.img = o;  o += n_mels * W * 4;              // dB image                              (float)
  L.melw = o; o += n_mels * 8 * 4;              // per band: klo, khi, 4 weights (+2 pad) (32 B)
  L.total = o;
  return L;
}

__global__ __launch_bounds__(kMelThreads) void logmel_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ frames,
    const unsigned char* __restrict__ tables, float* __restrict__ spec,
    int32_t* __restrict__ frames_out, int B, int T, int n_fft, int hop, int n_mels, float mean,
    float stdv, int W) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ float red[kMelWaves];
  const MelTables tb = mel_tables(n_fft, n_mels);
  const MelLayout L = mel_layout(T, n_fft, hop, n_mels, W);
  const double* afrag = reinterpret_cast<const double*>(tables);
  const float* wts = reinterpret_cast<const float*>(tables + tb.off_wts);
  const int32_t* krange = reinterpret_cast<const int32_t*>(tables + tb.off_krange);
  float* xrow = reinterpret_cast<float*>(smem + L.xrow);
  float* ps = reinterpret_cast<float*>(smem + L.ps);
  float* img = reinterpret_cast<float*>(smem + L.img);
  float* melw = reinterpret_cast<float*>(smem + L.melw);

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_bins = tb.n_bins, pad = n_fft / 2;
  const int n_frames = 1 + T / hop;  // centred: 1 + (T + 2*pad - n_fft) / hop

  // reflect-padded row (numpy.pad mode='reflect': the edge sample is not repeated), zero beyond.
  // The body is a straight coalesced copy (all loads independent, issued back to back); only the
  // 2*pad edge samples take the mirrored index.
  const float* xg = x + (size_t)b * T;
  // per-band filter span and its first 4 weights -> LDS (global latency overlaps the row copy)
  for (int m = tid; m < n_mels; m += kMelThreads) {
    const int klo = krange[2 * m], khi = krange[2 * m + 1];
    reinterpret_cast<int*>(melw)[8 * m] = klo;
    reinterpret_cast<int*>(melw)[8 * m + 1] = khi;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      melw[8 * m + 2 + j] = (klo + j <= khi) ? wts[m * n_bins + klo + j] : 0.f;
  }
  for (int i = tid; i < T; i += kMelThreads) xrow[pad + i] = xg[i];
  for (int i = tid; i < pad; i += kMelThreads) {
    int sl = pad - i;                       // left edge: index -(i - pad)
    sl = sl >= T ? T - 1 : sl;
    xrow[i] = xg[sl];
    int sr = 2 * (T - 1) - (T + i);         // right edge: index T + i mirrored about T - 1
    sr = sr < 0 ? 0 : sr;
    xrow[pad + T + i] = xg[sr];
  }
  for (int i = T + n_fft + tid; i < L.xr; i += kMelThreads) xrow[i] = 0.f;
  __syncthreads();

  // ---- STFT power on the f64 matrix cores -----------------------------------------------------
  // C layout of v_mfma_f64_16x16x4_f64: row = (lane>>4) + 4*reg, col = lane&15.  The re and the
  // im GEMM use the same row -> bin map, so a lane holds re and im of the same (bin, frame) in
  // the same register slot of its two accumulators: |.|^2 needs no cross-lane traffic.
  const int n_units = tb.m_tiles * (L.nfp / (16 * kNGroup));
  const int n_groups = L.nfp / (16 * kNGroup);
  const int col = lane & 15, kq = lane >> 4;
  for (int unit = wave; unit < n_units; unit += kMelWaves) {
    const int mt = unit / n_groups, ng = unit - mt * n_groups;
    d4 are[kNGroup], aim[kNGroup];
#pragma unroll
    for (int i = 0; i < kNGroup; ++i) are[i] = aim[i] = d4{0.0, 0.0, 0.0, 0.0};
    const double* ap = afrag + (size_t)mt * tb.ksteps * 128 + lane;   // [ks][re|im][64]
    // B operands: x[f*hop + k] +- x[f*hop + n_fft - k], frame f = 16*(kNGroup*ng + i) + col
    const float* xlo = xrow + (16 * kNGroup * ng + col) * hop + kq + 1;
    const float* xhi = xrow + (16 * kNGroup * ng + col) * hop + n_fft - kq - 1;
    double a_re = ap[0], a_im = ap[64];
    for (int ks = 0; ks < tb.ksteps; ++ks) {
      const double cr = a_re, ci = a_im;
      if (ks + 1 < tb.ksteps) {                      // prefetch the next k-step's A fragments
        a_re = ap[(size_t)(ks + 1) * 128];
        a_im = ap[(size_t)(ks + 1) * 128 + 64];
      }
      double bs[kNGroup], bd[kNGroup];
#pragma unroll
      for (int i = 0; i < kNGroup; ++i) {
        const double lo = (double)xlo[(16 * i) * hop + 4 * ks];
        const double hi = (double)xhi[(16 * i) * hop - 4 * ks];
        bs[i] = lo + hi;                             // exact in float64
        bd[i] = lo - hi;
      }
#pragma unroll
      for (int i = 0; i < kNGroup; ++i) {
        are[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(cr, bs[i], are[i], 0, 0, 0);
        aim[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ci, bd[i], aim[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < kNGroup; ++i) {
      const int fcol = 16 * (kNGroup * ng + i) + col;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float fr = (float)are[i][rr], fi = (float)aim[i][rr];  // complex64
        const float mag = hypotf(fr, fi);                             // np.abs
        ps[(16 * mt + kq + 4 * rr) * L.nfp + fcol] = mag * mag;       // ** 2
      }
    }
  }
  __syncthreads();

  // ---- mel projection, dB, item maximum --------------------------------------------------------
  float vmax = -INFINITY;  // max over the item of 10*log10(max(amin, S)), all n_frames columns
  // wave -> mel band (its filter span and weights are wave-uniform, fetched once per band),
  // lane -> frame: the power-spectrogram reads and the image writes are unit-stride in LDS
  for (int m = wave; m < n_mels; m += kMelWaves) {
    const int klo = reinterpret_cast<const int*>(melw)[8 * m];
    const int khi = reinterpret_cast<const int*>(melw)[8 * m + 1];
    // triangular filters span a handful of bins: up to 4 weights come from the LDS record so that
    // the frame loop never waits on a global load
    float w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = melw[8 * m + 2 + j];
    const bool small = khi - klo < 4;
    for (int t = lane; t < n_frames; t += 64) {
      float accm = 0.f;
      if (small) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (klo + j <= khi) accm = fmaf(w4[j], ps[(klo + j) * L.nfp + t], accm);
      } else {
        for (int k = klo; k <= khi; ++k) accm = fmaf(wts[m * n_bins + k], ps[k * L.nfp + t], accm);
      }
      const float db = 10.0f * log10f(fmaxf(1e-10f, accm));  // power_to_db, amin = 1e-10
      vmax = fmaxf(vmax, db);
      if (t < W) img[m * W + t] = db;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  if (lane == 0) red[wave] = vmax;
  __syncthreads();
  float ref_db = red[0];
  for (int i = 1; i < kMelWaves; ++i) ref_db = fmaxf(ref_db, red[i]);

  // ---- column boundaries, dB referencing, top_db clip, normalisation, crop --------------------
  int col_end = W;
  {
    // round(f * n_frames / len(y)) with Python's round-half-even (databuilder.ipynb cell 6:101)
    const int f4 = frames[b * 5 + 4];
    const double v = (double)((long long)f4 * n_frames) / (double)T;
    const int c4 = (int)rint(v);
    col_end = c4 < 0 ? 0 : (c4 > W ? W : c4);
    if (col_end > n_frames) col_end = n_frames;
    if (frames_out && tid < 5) {
      const double vv = (double)((long long)frames[b * 5 + tid] * n_frames) / (double)T;
      frames_out[b * 5 + tid] = (int)rint(vv);
    }
  }
  // log_spec = db - ref_db; its maximum is (ref_db - ref_db) = 0, so top_db clips at -80
  const float floor_db = (ref_db - ref_db) - 80.0f;
  float* out = spec + (size_t)b * n_mels * W;
  const float inv_guard = stdv;  // IEEE division kept: (x - mean) / std as numpy evaluates it
  if (kMelThreads % W == 0) {    // each thread keeps one column: no per-element modulo
    const int t = tid % W;
    const bool inside = t < col_end;
#pragma unroll 4
    for (int i = tid; i < n_mels * W; i += kMelThreads) {  // coalesced rows of the image
      float v = 0.f;  // zero padding is applied AFTER normalisation (cell 6:99, 141-142)
      if (inside) v = (fmaxf(img[i] - ref_db, floor_db) - mean) / inv_guard;
      out[i] = v;
    }
  } else {
    for (int i = tid; i < n_mels * W; i += kMelThreads) {
      const int t = i % W;
      float v = 0.f;
      if (t < col_end) v = (fmaxf(img[i] - ref_db, floor_db) - mean) / inv_guard;
      out[i] = v;
    }
  }
}

// librosa.hz_to_mel / mel_to_hz, htk=False (Slaney): linear below 1 kHz, log above.
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0;
  const double min_log_mel = (min_log_hz - 0.0) / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : (f - 0.0) / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0;
  const double min_log_mel = (min_log_hz - 0.0) / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : 0.0 + f_sp * m;
}

}  // namespace pcgmix

extern "C" long long pcgmix_logmel_tables_size(int n_fft, int n_mels) {
  if (n_fft < 2 || (n_fft & 1) || n_mels < 1) return 0;
  return (long long)pcgmix::mel_tables(n_fft, n_mels).total;
}

extern "C" int pcgmix_logmel_tables(int n_fft, int n_mels, float fmin, float fmax, float sr,
                                    void* out) {
  using namespace pcgmix;
  if (!out || n_fft < 2 || (n_fft & 1) || n_mels < 1 || !(fmax > fmin) || !(sr > 0))
    return hipErrorInvalidValue;
  const MelTables tb = mel_tables(n_fft, n_mels);
  unsigned char* base = static_cast<unsigned char*>(out);
  // periodic Hann (scipy.signal.get_window('hann', n_fft, fftbins=True)) and twiddles
  std::vector<double> win(n_fft), cs(n_fft), sn(n_fft);
  for (int j = 0; j < n_fft; ++j) {
    const double ang = 2.0 * M_PI * (double)j / (double)n_fft;
    cs[j] = std::cos(ang);
    sn[j] = std::sin(ang);
    win[j] = 0.5 - 0.5 * cs[j];
  }
  double* afrag = reinterpret_cast<double*>(base);
  const int nh = n_fft / 2;
  for (int mt = 0; mt < tb.m_tiles; ++mt)
    for (int ks = 0; ks < tb.ksteps; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int bin = 16 * mt + (l & 15), k = 4 * ks + (l >> 4) + 1;
        double vr = 0.0, vi = 0.0;
        if (bin < tb.n_bins && k <= nh) {
          const int idx = (int)(((long long)bin * k) % n_fft);
          vr = win[k % n_fft] * cs[idx] * (k == nh ? 0.5 : 1.0);
          vi = -win[k % n_fft] * sn[idx];
          if (k == nh) vi = 0.0;                     // its partner is itself: x[k] - x[N-k] == 0
        }
        double* dst = afrag + ((size_t)mt * tb.ksteps + ks) * 128;
        dst[l] = vr;
        dst[64 + l] = vi;
      }
  // librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney', dtype=float32)
  std::vector<double> melf(n_mels + 2);
  const double m0 = hz_to_mel((double)fmin), m1 = hz_to_mel((double)fmax);
  const double step = (m1 - m0) / (double)(n_mels + 1);  // numpy.linspace
  for (int i = 0; i < n_mels + 2; ++i) melf[i] = mel_to_hz(i == n_mels + 1 ? m1 : m0 + i * step);
  const double fft_step = 1.0 / ((double)n_fft * (1.0 / (double)sr));  // np.fft.rfftfreq
  float* wts = reinterpret_cast<float*>(base + tb.off_wts);
  int32_t* krange = reinterpret_cast<int32_t*>(base + tb.off_krange);
  for (int m = 0; m < n_mels; ++m) {
    int lo = tb.n_bins, hi = -1;
    const double enorm = 2.0 / (melf[m + 2] - melf[m]);
    for (int k = 0; k < tb.n_bins; ++k) {
      const double f = (double)k * fft_step;
      const double lower = -(melf[m] - f) / (melf[m + 1] - melf[m]);
      const double upper = (melf[m + 2] - f) / (melf[m + 2] - melf[m + 1]);
      const float w = (float)std::fmax(0.0, std::fmin(lower, upper));  // float32 weights array
      const float wn = (float)((double)w * enorm);                     // weights *= enorm
      wts[(size_t)m * tb.n_bins + k] = wn;
      if (wn != 0.f) {
        lo = k < lo ? k : lo;
        hi = k;
      }
    }
    krange[2 * m] = lo;
    krange[2 * m + 1] = hi;
  }
  return hipSuccess;
}

extern "C" int pcgmix_logmel_f32(const float* x, const int32_t* frames, const void* tables,
                                 float* spec, int32_t* frames_out, int B, int T, int n_fft,
                                 int hop, int n_mels, float mean, float std, int W,
                                 pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !frames || !tables || !spec) return hipErrorInvalidValue;
  if (B < 0 || T < 2 || n_fft < 2 || (n_fft & 1) || hop < 1 || n_mels < 1 || W < 1 ||
      !(std != 0.f) || n_fft / 2 >= T)
    return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  const MelLayout L = mel_layout(T, n_fft, hop, n_mels, W);
  if (L.total > 158 * 1024) return hipErrorInvalidValue;
  static unsigned long long lds_ok = 0;
  if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(logmel_kernel), &lds_ok,
                                     158 * 1024))
    return (int)e;
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)B), dim3(kMelThreads), (size_t)L.total,
                     reinterpret_cast<hipStream_t>(stream), x, frames,
                     static_cast<const unsigned char*>(tables), spec, frames_out, B, T, n_fft, hop,
                     n_mels, mean, std, W);
  return (int)hipGetLastError();
}
