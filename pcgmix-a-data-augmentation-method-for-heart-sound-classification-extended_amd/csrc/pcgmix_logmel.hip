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
// One block per sample.  LDS holds the reflect-padded row, the DFT twiddles, the mel filter
// bank and the whole (n_mels x W) dB image, so the input is read from HBM once and the output is
// written once, in full 512-byte rows: 4*T + 4*n_mels*W algorithmic bytes per sample.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kMelThreads = 256;
constexpr int kMelWaves = kMelThreads / 64;

// librosa.hz_to_mel / mel_to_hz, htk=False (Slaney): linear below 1 kHz, log above.
__device__ __forceinline__ double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0;
  const double min_log_mel = (min_log_hz - 0.0) / f_sp, logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : (f - 0.0) / f_sp;
}
__device__ __forceinline__ double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0;
  const double min_log_mel = (min_log_hz - 0.0) / f_sp, logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : 0.0 + f_sp * m;
}

struct MelLayout {  // byte offsets into dynamic LDS
  int xrow, tw, win, xw, melf, wts, pw, img, total;
};

__host__ __device__ inline MelLayout mel_layout(int T, int n_fft, int n_mels, int W) {
  const int n_bins = n_fft / 2 + 1;
  MelLayout L;
  int o = 0;
  L.tw = o;   o += 2 * n_fft * 8;                 // cos, sin of 2*pi*j/n_fft      (double)
  L.win = o;  o += n_fft * 8;                     // periodic Hann                 (double)
  L.xw = o;   o += kMelWaves * n_fft * 8;         // windowed frame, one per wave  (double)
  L.melf = o; o += (n_mels + 2) * 8;              // mel band edges in Hz          (double)
  L.xrow = o; o += (T + n_fft) * 4;               // reflect-padded waveform       (float)
  L.wts = o;  o += n_mels * n_bins * 4;           // filter bank                   (float)
  L.pw = o;   o += kMelWaves * n_bins * 4;        // power spectrum, one per wave  (float)
  L.img = o;  o += n_mels * W * 4;                // dB image                      (float)
  L.total = o;
  return L;
}

__global__ __launch_bounds__(kMelThreads) void logmel_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ frames, float* __restrict__ spec,
    int32_t* __restrict__ frames_out, int B, int T, int n_fft, int hop, int n_mels, float f_lo,
    float f_hi, float sr, float mean, float stdv, int W) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ float red[kMelWaves];
  const MelLayout L = mel_layout(T, n_fft, n_mels, W);
  double* tw = reinterpret_cast<double*>(smem + L.tw);
  double* win = reinterpret_cast<double*>(smem + L.win);
  double* xw_all = reinterpret_cast<double*>(smem + L.xw);
  double* melf = reinterpret_cast<double*>(smem + L.melf);
  float* xrow = reinterpret_cast<float*>(smem + L.xrow);
  float* wts = reinterpret_cast<float*>(smem + L.wts);
  float* pw_all = reinterpret_cast<float*>(smem + L.pw);
  float* img = reinterpret_cast<float*>(smem + L.img);

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_bins = n_fft / 2 + 1, pad = n_fft / 2;
  const int n_frames = 1 + T / hop;  // centred: 1 + (T + 2*pad - n_fft) / hop

  // ---- tables ---------------------------------------------------------------------------------
  for (int j = tid; j < n_fft; j += kMelThreads) {
    double s, c;
    sincospi(2.0 * (double)j / (double)n_fft, &s, &c);
    tw[2 * j] = c;
    tw[2 * j + 1] = s;
    win[j] = 0.5 - 0.5 * c;  // scipy.signal.get_window('hann', n_fft, fftbins=True)
  }
  for (int i = tid; i < n_mels + 2; i += kMelThreads) {
    // librosa.mel_frequencies: linspace in mel between hz_to_mel(fmin) and hz_to_mel(fmax)
    const double m0 = hz_to_mel((double)f_lo), m1 = hz_to_mel((double)f_hi);
    const double step = (m1 - m0) / (double)(n_mels + 1);
    const double m = (i == n_mels + 1) ? m1 : m0 + (double)i * step;
    melf[i] = mel_to_hz(m);
  }
  // reflect-padded row (numpy.pad mode='reflect': the edge sample is not repeated)
  for (int i = tid; i < T + n_fft; i += kMelThreads) {
    int s = i - pad;
    if (s < 0) s = -s;
    if (s >= T) s = 2 * (T - 1) - s;
    s = s < 0 ? 0 : (s >= T ? T - 1 : s);
    xrow[i] = x[(size_t)b * T + s];
  }
  __syncthreads();
  // librosa.filters.mel(norm='slaney', dtype=float32)
  const double fft_step = 1.0 / ((double)n_fft * (1.0 / (double)sr));  // np.fft.rfftfreq
  for (int i = tid; i < n_mels * n_bins; i += kMelThreads) {
    const int m = i / n_bins, k = i - m * n_bins;
    const double f = (double)k * fft_step;
    const double lower = -(melf[m] - f) / (melf[m + 1] - melf[m]);
    const double upper = (melf[m + 2] - f) / (melf[m + 2] - melf[m + 1]);
    const float w = (float)fmax(0.0, fmin(lower, upper));
    const double enorm = 2.0 / (melf[m + 2] - melf[m]);
    wts[i] = (float)((double)w * enorm);
  }
  for (int i = tid; i < n_mels * W; i += kMelThreads) img[i] = 0.f;
  __syncthreads();

  // ---- one wave per STFT frame ----------------------------------------------------------------
  double* xw = xw_all + wave * n_fft;
  float* pw = pw_all + wave * n_bins;
  float vmax = -INFINITY;  // max over the item of 10*log10(max(amin, S))
  for (int t0 = 0; t0 < n_frames; t0 += kMelWaves) {  // uniform trip count: barriers inside
    const int t = t0 + wave;
    const bool active = t < n_frames;
    if (active)
      for (int n = lane; n < n_fft; n += 64) xw[n] = win[n] * (double)xrow[t * hop + n];
    __syncthreads();
    if (active)
      for (int k = lane; k < n_bins; k += 64) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
          const double v = xw[n];
          re = fma(v, tw[2 * idx], re);
          im = fma(-v, tw[2 * idx + 1], im);
          idx += k;
          idx = idx >= n_fft ? idx - n_fft : idx;
        }
        const float fr = (float)re, fi = (float)im;  // complex64, as librosa stores the STFT
        const float mag = hypotf(fr, fi);            // np.abs(complex64)
        pw[k] = mag * mag;                           // ** 2
      }
    __syncthreads();
    if (active)
      for (int m = lane; m < n_mels; m += 64) {
        float acc = 0.f;
        for (int k = 0; k < n_bins; ++k) acc = fmaf(wts[m * n_bins + k], pw[k], acc);
        const float db = 10.0f * log10f(fmaxf(1e-10f, acc));  // power_to_db, amin = 1e-10
        vmax = fmaxf(vmax, db);
        if (t < W) img[m * W + t] = db;
      }
  }
  // item maximum (ref = np.max): wave shuffle, then across the block's waves
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
    if (frames_out && tid < 5) {
      const double vv = (double)((long long)frames[b * 5 + tid] * n_frames) / (double)T;
      frames_out[b * 5 + tid] = (int)rint(vv);
    }
  }
  // log_spec = db - ref_db; its maximum is (ref_db - ref_db) = 0, so top_db clips at -80
  const float floor_db = (ref_db - ref_db) - 80.0f;
  for (int i = tid; i < n_mels * W; i += kMelThreads) {
    const int t = i % W;
    float v = 0.f;  // zero padding is applied AFTER normalisation (cell 6:99, 141-142)
    if (t < col_end) {
      v = fmaxf(img[i] - ref_db, floor_db);
      v = (v - mean) / stdv;
    }
    spec[(size_t)b * n_mels * W + i] = v;
  }
}

}  // namespace pcgmix

extern "C" int pcgmix_logmel_f32(const float* x, const int32_t* frames, float* spec,
                                 int32_t* frames_out, int B, int T, int n_fft, int hop,
                                 int n_mels, float fmin, float fmax, float sr, float mean,
                                 float std, int W, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !frames || !spec) return hipErrorInvalidValue;
  if (B < 0 || T < 2 || n_fft < 2 || (n_fft & 1) || hop < 1 || n_mels < 1 || W < 1 ||
      !(fmax > fmin) || !(sr > 0) || !(std != 0.f) || n_fft / 2 >= T)
    return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  const MelLayout L = mel_layout(T, n_fft, n_mels, W);
  if (L.total > 158 * 1024) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)B), dim3(kMelThreads), (size_t)L.total,
                     reinterpret_cast<hipStream_t>(stream), x, frames, spec, frames_out, B, T,
                     n_fft, hop, n_mels, fmin, fmax, sr, mean, std, W);
  return (int)hipGetLastError();
}
