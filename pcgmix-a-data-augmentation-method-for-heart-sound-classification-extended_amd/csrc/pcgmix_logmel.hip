// pcgmix_logmel.hip — STFT -> log-mel front end on gfx950.
//
// Replaces the offline librosa pipeline of databuilder.ipynb cell 6:81-101, 127-142
// (melspectrogram(n_fft = 4*hop, hop, n_mels, fmin, fmax) -> power_to_db(ref=max) ->
// (x - mean)/std -> keep the cycle's columns -> zero-pad to W), librosa 0.9.2 semantics restated:
// centred frames (edge padding selectable: zeros = numpy 'constant', or 'reflect' — which of the
// two librosa 0.9.2 defaults to could not be checked offline), periodic Hann of n_fft, float64
// transform rounded to complex64, |.|^2 in float32, Slaney mel filter bank (float32 weights),
// float32 dB with amin = 1e-10 and top_db = 80.
// Two granularities:
//   per cycle      (pcgmix_logmel_f32)  one STFT per heart-cycle item, `ref` = the item's own
//                  maximum — what a per-batch transform of already cut cycles can see;
//   per recording  (pcgmix_logmel_recordings_f32)  the reference's own order of operations: one
//                  STFT over the WHOLE recording, `ref` = the recording's maximum, then every
//                  cycle's columns are sliced out of it (cell 6:93, 101, 134) and zero-padded.
//
// The STFT is a GEMM and librosa evaluates it in float64 — so it runs on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64).  The input is real and the periodic Hann window is symmetric
// (win[N-n] = win[n], win[0] = 0): with a[n] = x[n] + x[N-n], d[n] = x[n] - x[N-n] (exact in
// float64), H = N/2,
//     re[bin] = sum_{n=1..H}   g_n win[n] cos(2 pi bin n / N) a[n]        (g_H = 1/2, else 1)
//     im[bin] = sum_{n=1..H-1}    -win[n] sin(2 pi bin n / N) d[n]
// — K = N/2 instead of N (rounds 1-3).  Round 4 folds once more.  With the window applied on the
// DATA side, u[n] = win[n] a[n], v[n] = win[n] d[n], the coefficient matrices are pure twiddles, and
// n_fft = 4 hop makes H even: cos(theta(bin, H-n)) = s cos(theta(bin, n)), sin(theta(bin, H-n)) =
// -s sin(theta(bin, n)) with s = (-1)^bin.  With Q = H/2 = hop:
//     re[bin] =  sum_{n=0..Q} c_n cos(2 pi bin n / N) (u[n] + s u[H-n])    (c_0 = c_Q = 1/2, else 1)
//     im[bin] = -sum_{n=0..Q} c_n sin(2 pi bin n / N) (v[n] - s v[H-n])
// (the n = 0 column carries the old n = H term: u[0] = 0).  Even and odd bins are separate GEMMs
// with K = Q + 1 = 35 instead of 68: 9 k-steps instead of 17, half the matrix instructions again —
// the f64 matrix pipe (150 cycles per instruction and SIMD) is what bounds this kernel
// (profiles/r4_logmel_phases.txt).  Tile pair t holds the bins bin_lo + 2 (16 t + row) (+ 1), bin_lo = the
// lowest bin any mel filter reads, rounded down to even (2 at the reference's bank).
// Everything that does not depend on the data (the windowed twiddle matrices already in MFMA
// A-fragment order, the mel filter bank, each filter's non-zero span) is a constant table built
// once on the host (pcgmix_logmel_tables) and read through L2.  One block per sample; LDS holds the reflect-padded row, the power spectrogram and
// the whole dB image, so HBM sees the input once and the output once:
// 4*T + 4*n_mels*W algorithmic bytes per sample.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <cmath>
#include <type_traits>
#include <vector>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kMelThreads = 1024;
constexpr int kMelWaves = kMelThreads / 64;
constexpr int kLeftPad = 8;    // zero columns behind their coefficient rows (k parts need not divide n_fft/2)

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- constant tables (host-built blob) ---------------------------------------------------------
//   [0]            double afrag[tpp][ksteps][4][64]       A operands of tile pair t (even re, even im,
//                  odd re, odd im), one double per lane:
//                  lane l -> row = l & 15, n = 4*ks + (l >> 4)  (n = 0 .. Q; beyond: 0),
//                  even bin = bin_lo + 2 (16 t + row), odd bin = even bin + 1
//                  re:  c_n cos(2 pi bin n / n_fft);   im: -c_n sin(2 pi bin n / n_fft)
//   [off_wts]      float  wts[n_mels][n_bins]             librosa.filters.mel, slaney, float32
//   [off_krange]   int32  krange[n_mels][2]               first / last non-zero bin
//   [off_left]     double left[n_left][n_fft/2 + 8][2]    (re, im) coefficients (window included,
//                  single fold) of the bins beyond the last full PAIR of 16-bin tiles,
//                  n = 1 .. n_fft/2: those few rows (5 of 69 at n_fft = 136: bins 64..68) run on the
//                  float64 VALU instead of costing two more matrix tiles
//   [off_win]      double win[n_fft/2 + 1]                periodic Hann
//   [off_meta]     int32  n_left_used, bin_lo             the matrix tiles start at bin_lo (bins 0 and 1
//                  of the reference's bank carry no filter weight: the 64 tile rows are bins 2..65)
//                  and of the VALU rows bin_lo + 16*m_mfma + lb only the first n_left_used are read by
//                  a filter (66, 67; the Nyquist bin 68 carries no weight either)
// m_tiles counts the 16-row tiles the power spectrogram has room for; m_mfma = 2 * tpp of them (tpp
// tile pairs: even bins, odd bins) go through the matrix cores.
struct MelTables {
  int m_tiles, m_mfma, n_left, ksteps, n_bins;
  size_t off_wts, off_krange, off_left, off_win, off_meta, total;
};
__host__ __device__ inline MelTables mel_tables(int n_fft, int n_mels) {
  MelTables t;
  t.n_bins = n_fft / 2 + 1;
  int tpp = t.n_bins / 32;
  const int rem = t.n_bins - 32 * tpp;
  if (tpp >= 1 && rem <= 8) {
    t.n_left = rem;
  } else {
    tpp = (t.n_bins + 31) / 32;
    t.n_left = 0;
  }
  t.m_mfma = 2 * tpp;
  const int need_tiles = (t.n_bins + 15) / 16;
  t.m_tiles = t.m_mfma > need_tiles ? t.m_mfma : need_tiles;
  t.ksteps = (n_fft / 4 + 1 + 3) / 4;               // columns n = 0 .. n_fft/4
  size_t o = (size_t)t.m_mfma * t.ksteps * 2 * 64 * sizeof(double);
  t.off_wts = o;
  o += (size_t)n_mels * t.n_bins * sizeof(float);
  o = (o + 7) & ~(size_t)7;
  t.off_krange = o;
  o += (size_t)n_mels * 2 * sizeof(int32_t);
  o = (o + 7) & ~(size_t)7;
  t.off_left = o;
  o += (size_t)t.n_left * (n_fft / 2 + kLeftPad) * 2 * sizeof(double);
  t.off_win = o;
  o += (size_t)(n_fft / 2 + 1) * sizeof(double);
  t.off_meta = o;                                   // int32 n_left_used, bin_lo
  o += 8;
  t.total = (o + 15) & ~(size_t)15;
  return t;
}

struct MelLayout {  // byte offsets into dynamic LDS
  int xrow, ps, img, melw, left, win, part, part_bytes, total;
  int nfp, xr;
};
// n_frames = frames one block transforms: 1 + T/hop of a heart-cycle item, or the tile size of
// the per-recording pass; img_rows = n_mels when the block keeps a dB image in LDS, 0 otherwise.
__host__ __device__ inline MelLayout mel_layout(int n_frames, int n_fft, int hop, int n_mels,
                                                int W, bool image = true) {
  const MelTables tb = mel_tables(n_fft, n_mels);
  MelLayout L;
  L.nfp = ((n_frames + 31) / 32) * 32;          // frames, padded: whole 16-frame groups, rows 128-byte aligned
  L.xr = (L.nfp - 1) * hop + n_fft + 8;      // padded-row samples the GEMM may touch
  L.xr = (L.xr + 3) & ~3;
  int o = 0;
  L.xrow = o; o += L.xr * 4;                    // reflect-padded waveform, zero beyond (float)
  L.ps = o;   o += tb.m_tiles * 16 * L.nfp * 4; // power spectrogram [bin][frame]        (float)
  L.img = o;  o += (image ? n_mels * W : 0) * 4;  // dB image                            (float)
  L.melw = o; o += n_mels * 8 * 4;              // per band: klo, khi, 4 weights (+2 pad) (32 B)
  L.left = o; o += tb.n_left * (n_fft / 2 + kLeftPad) * 2 * 8;   // coefficients of the VALU bins (double)
  L.win = o;  o += (n_fft / 2 + 1) * 8;         // periodic Hann, n = 0 .. n_fft/2          (double)
  // partial sums of the VALU bins [wave job][bin][lane][re, im] (double): over the dB image, which
  // is written only after they are consumed, when that is large enough; else on their own
  // (the kernel takes as many k parts as fit: one wave job = n_left_used rows x 64 lanes x 16 bytes)
  if (image) {
    L.part = L.img;
    L.part_bytes = n_mels * W * 4;
  } else {
    o = (o + 15) & ~15;
    L.part = o;
    L.part_bytes = kMelWaves * tb.n_left * 64 * 16;
    o += L.part_bytes;
  }
  L.total = o;
  return L;
}

// power_to_db's 10 * log10(max(amin, S)), amin = 1e-10, as 10 log10(2) * v_log_f32: the hardware
// base-2 logarithm is good to 1 ulp, i.e. <= 3e-5 dB at the -100 dB end of the range, against the
// ~40 instructions of the library log10f — the mel/dB phase was 8.2 us of a 35 us block and bound
// by exactly that (profiles/r4_logmel_phases.txt).  The per-recording slice pass references its
// columns with the same function, so the recording's maximum still maps to exactly 0 dB.
__device__ __forceinline__ float power_db(float p) {
  return 3.01029995663981195f * __builtin_amdgcn_logf(fmaxf(1e-10f, p));
}

constexpr int kPadConstant = 0, kPadReflect = 1;
// End of every cycle (frames[b][4]) as int16 in the kernel ARGUMENTS (pcgmix_logmel_hostframes_f32):
// the only boundary the per-cycle kernel needs when the caller converts the others itself — no
// upload, no copy kernel in front of the launch.  n == 0: not in use, the kernel reads `frames`.
constexpr int kMelEndsMax = 1024;
struct MelEnds {
  int n;
  int16_t v[kMelEndsMax];
};
struct MelNoEnds { int n; };
constexpr int kTileFrames = 128;   // frames per block of the per-recording pass

// RECORD = false: one block per heart-cycle item b of x (B, T); image to `spec`.
// RECORD = true : one block per tile of a recording (`tiles[b]` = {recording, first frame, frames
//                 in this tile, absolute scratch column of the first frame}); un-referenced dB
//                 columns to the scratch `spec` (n_mels, scratch_cols), the tile's maximum mel
//                 power folded into ref_pow[recording] (non-negative floats order like their bit
//                 patterns, so an unsigned atomicMax is a float max).
// KS > 0: the number of k-steps is the compile-time constant KS (9 at n_fft = 136) and a wave keeps a
// ring of kRing A fragments in registers (slot = k-step mod kRing): at step ks it consumes slot
// ks % kRing and refills it with step ks + kRing of the same unit or, once that is past the end,
// with step ks % kRing of the wave's NEXT unit — kRing k-steps of distance throughout, also across
// units.  KS == 0: any n_fft, one k-step of prefetch.  Round 4 (profiles/r4_logmel_phases.txt, wall_clock64 around the phases of
// one block): the matrix phase was 22-23 us of a 35 us block whether it held 476 or 340 matrix
// instructions on its busiest SIMD — every k-step waited for its fragment's L2 round trip
// (~650 ns, prefetch distance one k-step), 34 of them in a row on the waves with two units.
// Probe builds only (-DPCGMIX_PHASE_CLOCK, profiles/probes/logmel_phase_clock.py): every block of the
// per-cycle launch leaves wall_clock64 (100 MHz) at its phase boundaries (+ XCC_ID, HW_ID) in g_logmel_clock.
#ifdef PCGMIX_PHASE_CLOCK
constexpr int kMelClockBlocks = 1024;
__device__ long long g_logmel_clock[kMelClockBlocks * 8];
#define PCGMIX_CLOCK(i)                                                          \
  do {                                                                           \
    if (!RECORD && blockIdx.x < kMelClockBlocks && threadIdx.x == 0) {           \
      g_logmel_clock[blockIdx.x * 8 + (i)] = wall_clock64();                     \
      if ((i) == 0) g_logmel_clock[blockIdx.x * 8 + 5] =                         \
          ((long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | __builtin_amdgcn_s_getreg(0xF804); \
    }                                                                            \
  } while (0)
__device__ long long g_logmel_wave_clock[16 * 4];
#define PCGMIX_WCLOCK(i)                                                         \
  do {                                                                           \
    if (!RECORD && blockIdx.x == 7 && (threadIdx.x & 63) == 0)                   \
      g_logmel_wave_clock[(threadIdx.x >> 6) * 4 + (i)] = wall_clock64();        \
  } while (0)
#else
#define PCGMIX_WCLOCK(i) do { } while (0)
#define PCGMIX_CLOCK(i) do { } while (0)
#endif

template <bool RECORD, int KS, bool ENDS = false>
__global__ __launch_bounds__(kMelThreads) void logmel_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ frames,
    const long long* __restrict__ rec_off, const int32_t* __restrict__ rec_len,
    const int4* __restrict__ tiles, const unsigned char* __restrict__ tables,
    float* __restrict__ spec, unsigned* __restrict__ ref_pow, long long scratch_cols,
    int32_t* __restrict__ frames_out, int B, int T, int n_fft, int hop, int n_mels, float mean,
    float stdv, int W, int pad_mode, const std::conditional_t<ENDS, MelEnds, MelNoEnds> ends) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ float red[kMelWaves];
  const MelTables tb = mel_tables(n_fft, n_mels);
  // (read here, used behind the staging barrier: a load issued there costs the phase its ~1.5 us)
  const int n_left_used = tb.n_left ? reinterpret_cast<const int32_t*>(tables + tb.off_meta)[0] : 0;
  const int bin_lo = reinterpret_cast<const int32_t*>(tables + tb.off_meta)[1];
  PCGMIX_CLOCK(0);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_bins = tb.n_bins, pad = n_fft / 2;
  // what this block transforms: `n_frames` centred frames of the signal xg[0..len), the first
  // one centred on sample s0 + pad
  const float* xg;
  int len, s0, n_frames, rec = 0;
  long long out_col = 0;
  if (RECORD) {
    const int4 tl = tiles[b];
    rec = tl.x;
    xg = x + rec_off[rec];
    len = rec_len[rec];
    s0 = tl.y * hop - pad;
    n_frames = tl.z;
    out_col = tl.w;
  } else {
    xg = x + (size_t)b * T;
    len = T;
    s0 = -pad;
    n_frames = 1 + T / hop;  // centred: 1 + (T + 2*pad - n_fft) / hop
  }
  const MelLayout L = mel_layout(RECORD ? kTileFrames : n_frames, n_fft, hop, n_mels, W, !RECORD);
  const double* afrag = reinterpret_cast<const double*>(tables);
  const float* wts = reinterpret_cast<const float*>(tables + tb.off_wts);
  const int32_t* krange = reinterpret_cast<const int32_t*>(tables + tb.off_krange);
  float* xrow = reinterpret_cast<float*>(smem + L.xrow);
  float* ps = reinterpret_cast<float*>(smem + L.ps);
  float* img = reinterpret_cast<float*>(smem + L.img);
  float* melw = reinterpret_cast<float*>(smem + L.melw);
  double* leftc = reinterpret_cast<double*>(smem + L.left);
  double* winl = reinterpret_cast<double*>(smem + L.win);
  double* lpart = reinterpret_cast<double*>(smem + L.part);
  {
    const double* lg = reinterpret_cast<const double*>(tables + tb.off_left);
    for (int i = tid; i < tb.n_left * (n_fft / 2 + kLeftPad) * 2; i += kMelThreads) leftc[i] = lg[i];
    const double* wg = reinterpret_cast<const double*>(tables + tb.off_win);
    for (int i = tid; i <= n_fft / 2; i += kMelThreads) winl[i] = wg[i];
  }

  // per-band filter span and its first 4 weights -> LDS (global latency overlaps the row copy)
  for (int m = tid; m < n_mels; m += kMelThreads) {
    const int klo = krange[2 * m], khi = krange[2 * m + 1];
    reinterpret_cast<int*>(melw)[8 * m] = klo;
    reinterpret_cast<int*>(melw)[8 * m + 1] = khi;
    // (clamped index, unconditional loads, zero applied behind them: a predicated load is a branch with
    // its own wait — four memory round trips in a row in front of the staging barrier)
    float wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int k = klo + j <= khi ? klo + j : khi;
      k = k < 0 ? 0 : (k >= n_bins ? n_bins - 1 : k);
      wv[j] = wts[m * n_bins + k];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) melw[8 * m + 2 + j] = (klo + j <= khi) ? wv[j] : 0.f;
  }
  // The padded row: LDS index i holds signal sample s = s0 + i.  Outside [0, len): zero
  // (numpy.pad 'constant') or the mirror image about the edge sample (numpy.pad 'reflect': the
  // edge sample itself is not repeated).  Every load goes to a clamped in-range address and is
  // unconditional (a predicated load would serialise the requests); the pad rule is a select.
  // Samples that only frames beyond n_frames would touch are zero.
  const int need = (n_frames - 1) * hop + n_fft;     // samples the valid frames touch
  // Eight samples per thread and pass, all eight loads issued before the first LDS store (a loop
  // with a runtime bound compiles to load, wait, store, load: ~20 serialised round trips here).
  constexpr int kStagePer = 8;
  for (int base = 0; base < L.xr; base += kStagePer * kMelThreads) {
    float v[kStagePer];
#pragma unroll
    for (int u = 0; u < kStagePer; ++u) {
      const int sidx = s0 + base + u * kMelThreads + tid;
      int src = sidx < 0 ? -sidx : (sidx >= len ? 2 * (len - 1) - sidx : sidx);
      src = src < 0 ? 0 : (src > len - 1 ? len - 1 : src);
      v[u] = xg[src];
    }
#pragma unroll
    for (int u = 0; u < kStagePer; ++u) {
      const int i = base + u * kMelThreads + tid, sidx = s0 + i;
      const bool inside = sidx >= 0 && sidx < len;
      if (i < L.xr) xrow[i] = (i >= need || (!inside && pad_mode == kPadConstant)) ? 0.f : v[u];
    }
  }
  __syncthreads();
  PCGMIX_CLOCK(1);

  // ---- STFT power on the f64 matrix cores -----------------------------------------------------
  // C layout of v_mfma_f64_16x16x4_f64: row = (lane>>4) + 4*reg, col = lane&15.  The re and the
  // im GEMM use the same row -> bin map, so a lane holds re and im of the same (bin, frame) in
  // the same register slot of its two accumulators: |.|^2 needs no cross-lane traffic.
  // A work unit is one PAIR of tiles (the even bins and the odd bins next to them) times one group
  // of 16 frames: the four B operands of a k-step — u[n] +- u[H-n], v[n] -+ v[H-n] — come from the
  // same four samples, two window values and sixteen float64 VALU operations, and feed four matrix
  // instructions.  (Until the units were paired an even tile and an odd tile each prepared their
  // own operands for two frame groups: 36 per four matrix instructions.  The phase is bound by the
  // instructions a SIMD issues, matrix and other, one after the other — profiles/r4_logmel_phases.txt.)
  const int tpp = tb.m_mfma >> 1;
  const int n_groups = L.nfp / 16;
  const int n_units = tpp * n_groups;
  const int col = lane & 15, kq = lane >> 4;
  // The bins beyond the last full tile pair on the float64 VALU (single fold, window in the
  // coefficients), by ALL waves, with barriers around them; only the rows some mel filter reads
  // (n_left_used: bins 66, 67 at the reference's bank).  A lane takes one FRAME, a wave one part of
  // the k range for 64 frames: the sample pair of a k is read, converted, added and subtracted once
  // for all rows, the coefficients are wave-uniform, nothing is clamped or masked (the rows end in
  // zero columns); the parts meet in LDS and are added in a fixed order.  Per-wave stamps
  // (profiles/r4_logmel_phases.txt) showed what the older form cost: one (bin, frame) per lane over
  // the whole k range, by the waves without a second unit, "beside" the first matrix units — those
  // waves came out of it 4 to 13 us into the phase: the pass is bound by its instruction count, and
  // every instruction it issues is a slot the matrix waves of the same SIMD do not get.
  if (n_left_used > 0) {                             // block-uniform
    const int nh = n_fft / 2, ncol = nh + kLeftPad;
    const int n_fg = (L.nfp + 63) / 64;              // 64-frame groups
    int parts = kMelWaves / n_fg;                    // k parts, one wave each per frame group
    const int fit = L.part_bytes / (n_left_used * 64 * 16) / n_fg;   // ... as far as their sums fit
    parts = parts > fit ? fit : parts;
    parts = parts < 1 ? 1 : (parts > kLeftPad ? kLeftPad : parts);
    const int kper = (nh + parts - 1) / parts;       // parts * kper <= nh + parts - 1 < nh + kLeftPad
    // the row count as a compile-time constant: straight-line code per k (with a run-time count every
    // row became its own branch with its own LDS wait)
    auto rows_pass = [&](auto nr_tag) {
      constexpr int NR = decltype(nr_tag)::value;
      for (int job = wave; job < n_fg * parts; job += kMelWaves) {
        const int fg = job % n_fg, part = job / n_fg;
        const int f = fg * 64 + lane, fc = f < L.nfp ? f : L.nfp - 1;
        const float* xl = xrow + fc * hop;
        const float* xh = xrow + fc * hop + n_fft;
        const int k0 = 1 + part * kper;
        double re[NR], im[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) re[r] = im[r] = 0.0;
#pragma unroll 2
        for (int k = k0; k < k0 + kper; ++k) {
          const double lo = (double)xl[k], hi = (double)xh[-k];
          const double sm = lo + hi, df = lo - hi;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const double* cf = leftc + ((size_t)r * ncol + (k - 1)) * 2;
            re[r] = fma(cf[0], sm, re[r]);
            im[r] = fma(cf[1], df, im[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          double* dst = lpart + (((size_t)job * NR + r) * 64 + lane) * 2;
          dst[0] = re[r];
          dst[1] = im[r];
        }
      }
    };
    switch (n_left_used) {
      case 1: rows_pass(std::integral_constant<int, 1>{}); break;
      case 2: rows_pass(std::integral_constant<int, 2>{}); break;
      case 3: rows_pass(std::integral_constant<int, 3>{}); break;
      case 4: rows_pass(std::integral_constant<int, 4>{}); break;
      case 5: rows_pass(std::integral_constant<int, 5>{}); break;
      case 6: rows_pass(std::integral_constant<int, 6>{}); break;
      case 7: rows_pass(std::integral_constant<int, 7>{}); break;
      default: rows_pass(std::integral_constant<int, 8>{}); break;
    }
    PCGMIX_WCLOCK(3);      // this wave's part of the VALU rows summed
    __syncthreads();
    for (int o = tid; o < n_left_used * L.nfp; o += kMelThreads) {
        const int r = o / L.nfp, f = o - r * L.nfp;
        const int fg = f >> 6, ln = f & 63;
        double re = 0.0, im = 0.0;
        for (int part = 0; part < parts; ++part) {
          const double* src = lpart + (((size_t)(part * n_fg + fg) * n_left_used + r) * 64 + ln) * 2;
          re += src[0];
          im += src[1];
        }
        const float fr = (float)re, fi = (float)im;
        const float mag = hypotf(fr, fi);
        ps[(bin_lo + 16 * tb.m_mfma + r) * L.nfp + f] = mag * mag;
      }
    __syncthreads();   // the partial sums may lie over the dB image: consumed before anything else starts
  }
  PCGMIX_WCLOCK(0);      // VALU rows done (or nothing to do)
  // Units are dealt in rounds of 16.  In a round that is not full the waves that get a unit must
  // sit on different SIMDs, and how a block's 16 waves map to the CU's four SIMDs is not something
  // to rely on: with `unit = wave + 16 r` the four second-round units of the 20 went to waves 0-3,
  // and the matrix phase took exactly as long as with 25 units (waves 0-3 share a SIMD, it seems:
  // 8 units on it either way).  Round r >= 1 hands unit 16 r + j to wave 5 j mod 16 — 0, 5, 10, 15,
  // 4, 9, ... — which spreads any prefix over both plausible mappings (wave mod 4 and wave / 4).
  auto unit_of = [&](int r) { return r == 0 ? wave : 16 * r + ((13 * wave) & 15); };
  const int half = n_fft / 2, quarter = n_fft / 4;
  // B operands of k-step ks for the frames 16*ng + col: column n = 4*ks + kq of the doubly folded
  // transform (header) with u = win * (lo + hi), v = win * (lo - hi):
  //   bo[0] = u[n] + u[H-n] (even re)   bo[1] = v[n] - v[H-n] (even im)
  //   bo[2] = u[n] - u[H-n] (odd re)    bo[3] = v[n] + v[H-n] (odd im)
  // Columns beyond Q are padding (zero coefficients): they read column 0.  Column 0 pairs x[0]
  // with itself (win[0] = 0 makes it vanish; x[N] is not part of the frame).
  auto b_operands = [&](int ks, int ng, double (&bo)[4]) {
    int n = 4 * ks + kq;
    // only the last k-step can hold padding columns and only the first one column 0: with ks a
    // constant of the unrolled loop the other steps' addresses stay base + immediate
    if (KS == 0 || ks == KS - 1) n = n > quarter ? 0 : n;
    const int m = half - n;
    const double wn = winl[n], wm = winl[m];
    const int hi_n = ((KS == 0 || ks == 0 || ks == KS - 1) && n == 0) ? 0 : n_fft - n;
    const float* xf = xrow + (16 * ng + col) * hop;
    const double lo = (double)xf[n], hi = (double)xf[hi_n];
    const double lo2 = (double)xf[m], hi2 = (double)xf[half + n];
    const double u = wn * (lo + hi), u2 = wm * (lo2 + hi2);      // the sums are exact in float64
    const double v = wn * (lo - hi), v2 = wm * (lo2 - hi2);
    bo[0] = u + u2;
    bo[1] = v - v2;
    bo[2] = u - u2;
    bo[3] = v + v2;
  };
  // KS > 0: the fragment ring.  Two slots of four fragments at KS = 9: a k-step is 4 matrix
  // instructions plus ~60 others, and four or five waves share the SIMD, so two steps ahead covers
  // the ~650 ns L2 round trip within the 128-VGPR budget of a 1024-thread block.
  constexpr int kRing = KS >= 16 ? 4 : 2;
  static_assert(KS == 0 || KS > kRing, "the ring must be shorter than a unit");
  double fr[kRing][4];
  if (KS > 0 && wave < n_units) {
    const double* ap0 = afrag + (size_t)(wave / n_groups) * KS * 256 + lane;
#pragma unroll
    for (int ks = 0; ks < kRing; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) fr[ks][q] = ap0[(size_t)ks * 256 + 64 * q];
  }
  for (int round = 0, unit = wave; unit < n_units; ++round, unit = unit_of(round)) {
    const int tp = unit / n_groups, ng = unit - tp * n_groups;
    d4 acc[4];                                        // even re, even im, odd re, odd im
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    const double* ap = afrag + (size_t)tp * tb.ksteps * 256 + lane;   // [ks][even re|even im|odd re|odd im][64]
    if (KS > 0) {
      const int nxt = unit_of(round + 1);
      const bool more = nxt < n_units;                // wave-uniform
      const double* apn = afrag + (size_t)((more ? nxt : unit) / n_groups) * KS * 256 + lane;
#pragma unroll
      for (int ks = 0; ks < (KS > 0 ? KS : 1); ++ks) {
        constexpr int kM = kRing - 1;
        double cf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cf[q] = fr[ks & kM][q];
        if (ks + kRing < KS) {                       // kRing steps ahead in this unit
#pragma unroll
          for (int q = 0; q < 4; ++q) fr[ks & kM][q] = ap[(size_t)(ks + kRing) * 256 + 64 * q];
        } else if (more && KS - ks <= kRing) {       // ... or the next unit's step ks % kRing
#pragma unroll
          for (int q = 0; q < 4; ++q) fr[ks & kM][q] = apn[(size_t)(ks & kM) * 256 + 64 * q];
        }
        double bo[4];
        b_operands(ks, ng, bo);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[q], bo[q], acc[q], 0, 0, 0);
        // keep the scheduler from hoisting several steps' operand loads above this step's matrix
        // instructions (the fully unrolled loop spilled at 128 VGPRs without it)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      double cn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cn[q] = ap[64 * q];
      for (int ks = 0; ks < tb.ksteps; ++ks) {
        double cf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cf[q] = cn[q];
        if (ks + 1 < tb.ksteps) {                    // prefetch the next k-step's A fragments
#pragma unroll
          for (int q = 0; q < 4; ++q) cn[q] = ap[(size_t)(ks + 1) * 256 + 64 * q];
        }
        double bo[4];
        b_operands(ks, ng, bo);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[q], bo[q], acc[q], 0, 0, 0);
      }
    }
    const int fcol = 16 * ng + col;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float fre = (float)acc[2 * par][rr], fim = (float)acc[2 * par + 1][rr];   // complex64
        const float mag = hypotf(fre, fim);                                            // np.abs
        ps[(bin_lo + 2 * (16 * tp + kq + 4 * rr) + par) * L.nfp + fcol] = mag * mag;   // ** 2
      }
#ifdef PCGMIX_PHASE_CLOCK
    if (round == 0) PCGMIX_WCLOCK(1); else PCGMIX_WCLOCK(2);
#endif
  }
  __syncthreads();

  PCGMIX_CLOCK(2);
  // ---- mel projection, dB, maximum ------------------------------------------------------------
  float vmax = -INFINITY;  // cycle mode: max over the item of 10*log10(max(amin, S)), all columns
  float pmax = 0.f;        // recording mode: max mel power of this tile's valid frames
  // wave -> mel band (its filter span and weights are wave-uniform, fetched once per band),
  // lane -> frame: the power-spectrogram reads and the image writes are unit-stride in LDS
  // The band's record is wave-uniform but comes out of LDS: without readfirstlane the compiler treats
  // the span as per-lane data and turns the taps into exec-masked branches, each with its own LDS
  // read and s_waitcnt.  The NEXT band's record is fetched before this band's frames are walked, and
  // up to three 64-frame passes of a band run side by side (independent read -> fma -> v_log ->
  // store chains).  Phase: 7.0 -> 5.8 us of the block (profiles/r4_logmel_phases.txt).
  struct BandRec { int klo, khi; float w[4]; };
  auto fetch = [&](int m) {
    BandRec r;
    r.klo = reinterpret_cast<const int*>(melw)[8 * m];
    r.khi = reinterpret_cast<const int*>(melw)[8 * m + 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) r.w[j] = melw[8 * m + 2 + j];
    return r;
  };
  auto emit = [&](int m, int t, float acc) {
    const float db = power_db(acc);                          // power_to_db, amin = 1e-10
    if (RECORD) {
      pmax = fmaxf(pmax, acc);
      spec[(size_t)m * scratch_cols + out_col + t] = db;     // 256-byte runs per wave
    } else {
      vmax = fmaxf(vmax, db);
      if (t < W) img[m * W + t] = db;
    }
  };
  BandRec nxt = fetch(wave < n_mels ? wave : 0);
  for (int m = wave; m < n_mels; m += kMelWaves) {
    const int klo = __builtin_amdgcn_readfirstlane(nxt.klo), khi = __builtin_amdgcn_readfirstlane(nxt.khi);
    // triangular filters span a handful of bins: up to 4 weights come from the LDS record so that
    // the frame loop never waits on a global load; weights beyond the span are 0 and their taps read
    // the span's last row, so the taps are unconditional: fmaf(0, finite, acc) == acc, the sum is
    // bit for bit the one over the span alone
    float w4[4];
    int row[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w4[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nxt.w[j])));
      int r = klo + j <= khi ? klo + j : khi;
      row[j] = (r < 0 ? 0 : r) * L.nfp;
    }
    if (m + kMelWaves < n_mels) nxt = fetch(m + kMelWaves);
    const int span = khi - klo;       // 0 for 125 of the reference's 128 filters (one FFT bin)
    if (span < 4) {
      constexpr int kTP = 3;
      for (int t0 = lane; t0 < n_frames; t0 += 64 * kTP) {
        float acc[kTP];
        if (span == 0) {
          float pv[kTP];
#pragma unroll
          for (int u = 0; u < kTP; ++u) {
            const int t = t0 + 64 * u;
            pv[u] = ps[row[0] + (t < n_frames ? t : n_frames - 1)];       // clamped: unconditional
          }
#pragma unroll
          for (int u = 0; u < kTP; ++u) acc[u] = fmaf(w4[0], pv[u], 0.f);
        } else {
          float pv[kTP][4];
#pragma unroll
          for (int u = 0; u < kTP; ++u) {
            const int t = t0 + 64 * u, tc = t < n_frames ? t : n_frames - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[u][j] = ps[row[j] + tc];
          }
#pragma unroll
          for (int u = 0; u < kTP; ++u) {
            acc[u] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[u] = fmaf(w4[j], pv[u][j], acc[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < kTP; ++u)
          if (t0 + 64 * u < n_frames) emit(m, t0 + 64 * u, acc[u]);
      }
    } else {
      for (int t = lane; t < n_frames; t += 64) {
        float accm = 0.f;
        for (int k = klo; k <= khi; ++k) accm = fmaf(wts[m * n_bins + k], ps[k * L.nfp + t], accm);
        emit(m, t, accm);
      }
    }
  }
  if (RECORD) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, o, 64));
    if (lane == 0) red[wave] = pmax;
    __syncthreads();
    if (tid == 0) {
      float p = red[0];
      for (int i = 1; i < kMelWaves; ++i) p = fmaxf(p, red[i]);
      atomicMax(ref_pow + rec, __float_as_uint(p));
    }
    return;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  if (lane == 0) red[wave] = vmax;
  __syncthreads();
  float ref_db = red[0];
  for (int i = 1; i < kMelWaves; ++i) ref_db = fmaxf(ref_db, red[i]);

  PCGMIX_CLOCK(3);
  // ---- column boundaries, dB referencing, top_db clip, normalisation, crop --------------------
  int col_end = W;
  {
    // round(f * n_frames / len(y)) with Python's round-half-even (databuilder.ipynb cell 6:101)
    int f4;
    if constexpr (ENDS) f4 = ends.v[b];
    else f4 = frames[b * 5 + 4];
    const double v = (double)((long long)f4 * n_frames) / (double)T;
    const int c4 = (int)rint(v);
    col_end = c4 < 0 ? 0 : (c4 > W ? W : c4);
    if (col_end > n_frames) col_end = n_frames;
    if (!ENDS && frames_out && tid < 5) {
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
      __builtin_nontemporal_store(v, out + i);
    }
  } else {
    for (int i = tid; i < n_mels * W; i += kMelThreads) {
      const int t = i % W;
      float v = 0.f;
      if (t < col_end) v = (fmaxf(img[i] - ref_db, floor_db) - mean) / inv_guard;
      out[i] = v;
    }
  }
#ifdef PCGMIX_PHASE_CLOCK
  __syncthreads();
  PCGMIX_CLOCK(4);
#endif
}

__global__ void zero_u32_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// Second pass of the per-recording front end: cycle c keeps columns [col0, col0 + n) of its
// recording's dB spectrogram (cell 6:134), referenced to the recording's maximum, clipped at
// -80 dB (power_to_db top_db), normalised, zero-padded to W columns after normalisation
// (cell 6:99, 141-142).  cycles[c] = {recording, absolute scratch column, n columns, unused}.
__global__ __launch_bounds__(256) void logmel_slice_kernel(
    const float* __restrict__ db, long long scratch_cols, const unsigned* __restrict__ ref_pow,
    const int4* __restrict__ cycles, float* __restrict__ spec, int n_mels, float mean, float stdv,
    int W) {
  const int4 cy = cycles[blockIdx.x];
  const float ref_db = power_db(__uint_as_float(ref_pow[cy.x]));
  const float floor_db = (ref_db - ref_db) - 80.0f;
  const int n = cy.z < 0 ? 0 : (cy.z > W ? W : cy.z);
  float* out = spec + (size_t)blockIdx.x * n_mels * W;
  if (!(W & 3) && !(reinterpret_cast<uintptr_t>(spec) & 15)) {
    // 16-byte stores (the zero padding is most of the image: a 0.8 s cycle keeps ~47 of 128
    // columns), four rows of loads in flight per thread; the scratch columns start anywhere, so
    // the loads stay 4-byte (consecutive lanes still read consecutive addresses)
    const int W4 = W >> 2, quads = n_mels * W4;
    const float* src = db + cy.y;
    constexpr int kUnroll = 4;
    for (int q0 = threadIdx.x; q0 < quads; q0 += kUnroll * blockDim.x) {
      float v[kUnroll][4];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * blockDim.x, m = q / W4, t = 4 * (q - m * W4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[u][e] = (q < quads && t + e < n) ? src[(size_t)m * scratch_cols + t + e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int q = q0 + u * blockDim.x, m = q / W4, t = 4 * (q - m * W4);
        if (q >= quads) break;
        float4 o;
        o.x = t < n ? (fmaxf(v[u][0] - ref_db, floor_db) - mean) / stdv : 0.f;
        o.y = t + 1 < n ? (fmaxf(v[u][1] - ref_db, floor_db) - mean) / stdv : 0.f;
        o.z = t + 2 < n ? (fmaxf(v[u][2] - ref_db, floor_db) - mean) / stdv : 0.f;
        o.w = t + 3 < n ? (fmaxf(v[u][3] - ref_db, floor_db) - mean) / stdv : 0.f;
        *reinterpret_cast<float4*>(out + 4 * (size_t)q) = o;
      }
    }
    return;
  }
  for (int i = threadIdx.x; i < n_mels * W; i += blockDim.x) {
    const int m = i / W, t = i - m * W;
    float v = 0.f;
    if (t < n) v = (fmaxf(db[(size_t)m * scratch_cols + cy.y + t] - ref_db, floor_db) - mean) / stdv;
    out[i] = v;
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

#ifdef PCGMIX_PHASE_CLOCK
extern "C" int pcgmix_logmel_wave_clock(long long* out64) {
  return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(pcgmix::g_logmel_wave_clock), 64 * sizeof(long long));
}
extern "C" int pcgmix_logmel_phase_clock(long long* out, int n_blocks) {
  if (n_blocks > pcgmix::kMelClockBlocks) return hipErrorInvalidValue;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcgmix::g_logmel_clock), (size_t)n_blocks * 8 * sizeof(long long));
}
#endif

extern "C" long long pcgmix_logmel_tables_size(int n_fft, int n_mels) {
  if (n_fft < 4 || (n_fft & 3) || n_mels < 1) return 0;
  return (long long)pcgmix::mel_tables(n_fft, n_mels).total;
}

extern "C" int pcgmix_logmel_tables(int n_fft, int n_mels, float fmin, float fmax, float sr,
                                    void* out) {
  using namespace pcgmix;
  if (!out || n_fft < 4 || (n_fft & 3) || n_mels < 1 || !(fmax > fmin) || !(sr > 0))
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
  // The matrix tiles start at the lowest bin a filter reads, rounded down to even (every bin a filter
  // reads must stay covered by the tiles plus the VALU rows, and the rows must exist in `ps`).
  int bin_lo = 0, used = 0;
  {
    int low = tb.n_bins, top = -1;
    for (int m = 0; m < n_mels; ++m) {
      if (krange[2 * m + 1] < krange[2 * m]) continue;           // empty filter
      low = krange[2 * m] < low ? krange[2 * m] : low;
      top = krange[2 * m + 1] > top ? krange[2 * m + 1] : top;
    }
    if (top >= 0) {
      bin_lo = low & ~1;
      const int cover = 16 * tb.m_mfma + tb.n_left;              // bins the tiles + VALU rows span
      if (bin_lo + cover > 16 * tb.m_tiles) bin_lo = (16 * tb.m_tiles - cover) & ~1;
      if (bin_lo < 0 || bin_lo + cover <= top) bin_lo = 0;       // cannot shift: the plain layout
      used = top - (bin_lo + 16 * tb.m_mfma) + 1;
      used = used < 0 ? 0 : (used > tb.n_left ? tb.n_left : used);
    }
    int32_t* meta = reinterpret_cast<int32_t*>(base + tb.off_meta);
    meta[0] = used;
    meta[1] = bin_lo;
  }
  double* afrag = reinterpret_cast<double*>(base);
  const int nh = n_fft / 2, nq = n_fft / 4;
  for (int tp = 0; tp < tb.m_mfma / 2; ++tp)
    for (int ks = 0; ks < tb.ksteps; ++ks)
      for (int par = 0; par < 2; ++par)
        for (int l = 0; l < 64; ++l) {
          const int bin = bin_lo + 2 * (16 * tp + (l & 15)) + par, n = 4 * ks + (l >> 4);
          double vr = 0.0, vi = 0.0;
          if (bin < tb.n_bins && n <= nq) {
            const int idx = (int)(((long long)bin * n) % n_fft);
            const double c = (n == 0 || n == nq) ? 0.5 : 1.0;
            vr = c * cs[idx];
            vi = -c * sn[idx];
          }
          double* dst = afrag + ((size_t)tp * tb.ksteps + ks) * 256 + 128 * par;
          dst[l] = vr;
          dst[64 + l] = vi;
        }
  {
    double* wt = reinterpret_cast<double*>(base + tb.off_win);
    for (int n = 0; n <= nh; ++n) wt[n] = win[n];
  }
  {
    double* left = reinterpret_cast<double*>(base + tb.off_left);
    const int ncol = nh + kLeftPad;
    for (int lb = 0; lb < tb.n_left; ++lb)
      for (int k = 1; k <= ncol; ++k) {
        const int bin = bin_lo + 16 * tb.m_mfma + lb;
        const bool real = bin < tb.n_bins && k <= nh;             // zero columns behind the row
        const int idx = (int)(((long long)bin * (k <= nh ? k : 0)) % n_fft);
        left[((size_t)lb * ncol + (k - 1)) * 2] = real ? win[k % n_fft] * cs[idx] * (k == nh ? 0.5 : 1.0) : 0.0;
        left[((size_t)lb * ncol + (k - 1)) * 2 + 1] = (!real || k == nh) ? 0.0 : -win[k % n_fft] * sn[idx];
      }
  }
  return hipSuccess;
}

namespace pcgmix {
// The per-cycle launch; ends != nullptr: the cycle ends travel in the kernel arguments.
static int launch_logmel_cycles(const float* x, const int32_t* frames, const MelEnds* ends, const void* tables,
                                float* spec, int32_t* frames_out, int B, int T, int n_fft, int hop,
                                int n_mels, float mean, float std, int W, int pad_mode, hipStream_t s) {
  const MelLayout L = mel_layout(1 + T / hop, n_fft, hop, n_mels, W);
  if (L.total > 158 * 1024) return hipErrorInvalidValue;
  static unsigned long long lds_ok[4] = {0, 0, 0, 0};
  const bool ks9 = mel_tables(n_fft, n_mels).ksteps == 9;      // the reference's n_fft = 136
  const unsigned char* tb = static_cast<const unsigned char*>(tables);
#define PCGMIX_MEL(KSV, EN, SLOT, ENDARG)                                                              \
  do {                                                                                               \
    auto kern = logmel_kernel<false, KSV, EN>;                                                       \
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(kern), &lds_ok[SLOT], 158 * 1024)) \
      return (int)e;                                                                                 \
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(kMelThreads), (size_t)L.total, s, x, frames, nullptr, \
                       nullptr, nullptr, tb, spec, nullptr, 0LL, frames_out, B, T, n_fft, hop, n_mels, mean, \
                       std, W, pad_mode, ENDARG);                                                    \
  } while (0)
  if (ends) {
    if (ks9) PCGMIX_MEL(9, true, 2, *ends); else PCGMIX_MEL(0, true, 3, *ends);
  } else {
    if (ks9) PCGMIX_MEL(9, false, 0, MelNoEnds{0}); else PCGMIX_MEL(0, false, 1, MelNoEnds{0});
  }
#undef PCGMIX_MEL
  return (int)hipGetLastError();
}
}  // namespace pcgmix

extern "C" int pcgmix_logmel_f32(const float* x, const int32_t* frames, const void* tables,
                                 float* spec, int32_t* frames_out, int B, int T, int n_fft,
                                 int hop, int n_mels, float mean, float std, int W, int pad_mode,
                                 pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !frames || !tables || !spec) return hipErrorInvalidValue;
  if (B < 0 || T < 2 || n_fft < 4 || (n_fft & 3) || hop < 1 || n_mels < 1 || W < 1 ||
      !(std != 0.f) || n_fft / 2 >= T || (pad_mode != kPadConstant && pad_mode != kPadReflect))
    return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  return launch_logmel_cycles(x, frames, nullptr, tables, spec, frames_out, B, T, n_fft, hop, n_mels, mean,
                              std, W, pad_mode, reinterpret_cast<hipStream_t>(stream));
}

// pcgmix_logmel_f32 for a caller that holds the boundaries on the HOST (the reference cuts its
// spectrograms from numpy arrays, databuilder.ipynb cell 6:101): frames_host (B,5) int32 in host
// memory.  Up to 1024 items and T <= 32767 the cycle ends ride in the kernel arguments — no upload,
// no copy kernel ahead of the launch; beyond that: hipErrorInvalidValue (upload and call
// pcgmix_logmel_f32).  The boundaries in spectrogram columns are the caller's to compute
// (frontend.spec_frames: round(f * n_frames / T), round-half-even).
extern "C" int pcgmix_logmel_hostframes_f32(const float* x, const int32_t* frames_host, const void* tables,
                                            float* spec, int B, int T, int n_fft, int hop, int n_mels,
                                            float mean, float std, int W, int pad_mode,
                                            pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !frames_host || !tables || !spec) return hipErrorInvalidValue;
  if (B < 0 || B > kMelEndsMax || T < 2 || T > 32767 || n_fft < 4 || (n_fft & 3) || hop < 1 || n_mels < 1 ||
      W < 1 || !(std != 0.f) || n_fft / 2 >= T || (pad_mode != kPadConstant && pad_mode != kPadReflect))
    return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  MelEnds ends;
  ends.n = B;
  for (int b = 0; b < B; ++b) {
    const int32_t f4 = frames_host[(size_t)b * 5 + 4];
    ends.v[b] = (int16_t)(f4 < -32768 ? -32768 : (f4 > 32767 ? 32767 : f4));
  }
  return launch_logmel_cycles(x, nullptr, &ends, tables, spec, nullptr, B, T, n_fft, hop, n_mels, mean, std, W,
                              pad_mode, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pcgmix_logmel_tile_frames(void) { return pcgmix::kTileFrames; }

extern "C" int pcgmix_logmel_recordings_f32(
    const float* y, const int64_t* rec_off, const int32_t* rec_len, int R, const int32_t* tiles,
    int n_tiles, const int32_t* cycles, int n_cycles, const void* tables, float* db_scratch,
    long long scratch_cols, uint32_t* ref_pow, float* spec, int n_fft, int hop, int n_mels,
    float mean, float std, int W, int pad_mode, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!y || !rec_off || !rec_len || !tiles || !cycles || !tables || !db_scratch || !ref_pow ||
      !spec)
    return hipErrorInvalidValue;
  if (R < 1 || n_tiles < 1 || n_cycles < 0 || scratch_cols < 1 || n_fft < 4 || (n_fft & 3) ||
      hop < 1 || n_mels < 1 || W < 1 || !(std != 0.f) ||
      (pad_mode != kPadConstant && pad_mode != kPadReflect))
    return hipErrorInvalidValue;
  const MelLayout L = mel_layout(kTileFrames, n_fft, hop, n_mels, W, false);
  if (L.total > 158 * 1024) return hipErrorInvalidValue;
  static unsigned long long lds_ok = 0, lds_ok17 = 0;
  const bool ks17 = mel_tables(n_fft, n_mels).ksteps == 9;
  if (hipError_t e = ks17 ? allow_large_lds(reinterpret_cast<const void*>(logmel_kernel<true, 9>),
                                            &lds_ok17, 158 * 1024)
                          : allow_large_lds(reinterpret_cast<const void*>(logmel_kernel<true, 0>),
                                            &lds_ok, 158 * 1024))
    return (int)e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, ref_pow, R);
  static_assert(sizeof(long long) == sizeof(int64_t), "rec_off is read as long long");
  if (ks17)
    hipLaunchKernelGGL((logmel_kernel<true, 9>), dim3((unsigned)n_tiles), dim3(kMelThreads),
                       (size_t)L.total, s, y, nullptr, reinterpret_cast<const long long*>(rec_off),
                       rec_len, reinterpret_cast<const int4*>(tiles),
                       static_cast<const unsigned char*>(tables), db_scratch, ref_pow, scratch_cols,
                       nullptr, n_tiles, 0, n_fft, hop, n_mels, mean, std, W, pad_mode, MelNoEnds{0});
  else
    hipLaunchKernelGGL((logmel_kernel<true, 0>), dim3((unsigned)n_tiles), dim3(kMelThreads),
                       (size_t)L.total, s, y, nullptr, reinterpret_cast<const long long*>(rec_off),
                       rec_len, reinterpret_cast<const int4*>(tiles),
                       static_cast<const unsigned char*>(tables), db_scratch, ref_pow, scratch_cols,
                       nullptr, n_tiles, 0, n_fft, hop, n_mels, mean, std, W, pad_mode, MelNoEnds{0});
  if (n_cycles > 0)
    hipLaunchKernelGGL(logmel_slice_kernel, dim3((unsigned)n_cycles), dim3(256), 0, s, db_scratch,
                       scratch_cols, ref_pow, reinterpret_cast<const int4*>(cycles), spec, n_mels,
                       mean, std, W);
  return (int)hipGetLastError();
}
