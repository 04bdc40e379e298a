// pcgmix_saliency.hip — saliency post-processing and saliency-optimal displacement search (gfx950).
//
//  saliency_post_kernel   saliency.py:63-91   |grad| -> zero tail -> sum channels -> Gaussian ->
//                                             zero tail -> per-row min/max normalisation
//  salopt_disp_kernel     augmentations.py:60-128, 210-287   first strict argmax over the
//                                             displacement d of a float32 objective summed in
//                                             numpy's pairwise order
//
// Both are tiny next to the splice (4*C*T + 4*T and 8*T algorithmic bytes per sample); they exist
// so that a saliency-guided step needs no host round trip: the displacement table they produce is
// consumed directly by pcgmix_mix_warp_f32 as its `off` argument.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <math.h>
#include <stdint.h>

#include <stdlib.h>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kSalThreads = 640;   // saliency_post: one block per row (625 x 8 outputs at T = 5000)
constexpr int kDispThreads = 256;  // displacement scan: 4 waves per block, every one with work
constexpr int kMaxTaps = 255;

struct Taps {
  float w[kMaxTaps];
};

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  // wave64 shuffle reduction, then one LDS hop across the 4 waves of the block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float other = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, other) : fminf(v, other);
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < kSalThreads / 64; ++i) r = is_max ? fmaxf(r, red[i]) : fminf(r, red[i]);
  return r;
}

// One block per row b.  LDS: a[T + ksize - 1 (+pad)] (channel-summed |grad| with zero halo), s[T].
// KS > 0: the tap count is a compile-time constant and every thread produces 8 consecutive
// outputs from one 8+KS-1 sample window held in registers (27 ds_read_b128 for 808 FMAs with the
// reference's 101 taps, instead of one ds_read_b32 per multiply-add): 25 -> ~10 us at bs=256.
// KS == 0: any odd ksize <= 255, one output per thread per round.  Taps are applied in ascending
// order either way.
// Probe builds only (-DPCGMIX_PHASE_CLOCK, profiles/probes/salpost_phase_clock.py).
#ifdef PCGMIX_PHASE_CLOCK
__device__ long long g_salpost_clock[1024 * 8];
#define PCGMIX_SCLOCK(i)                                                                     \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_salpost_clock[blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define PCGMIX_SCLOCK(i) do { } while (0)
#endif

template <int KS>
__global__ __launch_bounds__(kSalThreads) void saliency_post_kernel(
    const float* __restrict__ grad, const int32_t* __restrict__ frames, float* __restrict__ sal,
    Taps taps, int ksize, int B, int C, int T) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float red[kSalThreads / 64];
  const int b = blockIdx.x;
  const int half = ksize / 2;
  const int a_len = (T + ksize - 1 + 8 + 3) & ~3;   // +8: the last window may start up to 7 past T
  float* a = smem;                  // a[half + t]
  float* s = smem + a_len;          // s[t]
  PCGMIX_SCLOCK(0);
  int f4 = frames[b * 5 + 4];
  f4 = f4 < 0 ? 0 : (f4 > T ? T : f4);

  if (C == 4) {
    // The four band channels of the Potes input, up to 8 positions per thread: all 32 loads of a
    // lane are issued before the first is used.  (The generic loop below compiles to load, wait,
    // add per channel: 32 serialised L2 round trips per thread with one block per CU — half of
    // this kernel's time.)  Same order of additions: ((0 + |g0|) + |g1|) + |g2|) + |g3|.
    constexpr int kPer = 8;
    for (int base = 0; base < a_len; base += kPer * kSalThreads) {
      float v[kPer][4];
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int t = base + u * kSalThreads + (int)threadIdx.x - half;
        // clamped address, UNCONDITIONAL load, zero applied below: `in ? g[..] : 0` compiles to a branch
        // around every load with an s_waitcnt inside it (31 of them in this loop's ISA)
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
        const float* g = grad + ((size_t)b * 4) * T + tc;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = g[(size_t)c * T];
      }
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int i = base + u * kSalThreads + (int)threadIdx.x;
        const int t = i - half;
        const bool in = t >= 0 && t < f4;      // saliency.py:66-67 zeroes t >= f[-1] before the sum
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __fadd_rn(acc, fabsf(v[u][c]));
        if (i < a_len) a[i] = in ? acc : 0.f;
      }
    }
  } else {
    for (int i = threadIdx.x; i < a_len; i += kSalThreads) {
      const int t = i - half;
      float acc = 0.f;
      if (t >= 0 && t < f4) {  // saliency.py:66-67 zeroes t >= f[-1] before the channel sum
        const float* g = grad + ((size_t)b * C) * T + t;
        for (int c = 0; c < C; ++c) acc = __fadd_rn(acc, fabsf(g[(size_t)c * T]));
      }
      a[i] = acc;
    }
  }
  __syncthreads();
  PCGMIX_SCLOCK(1);

  float lmin = INFINITY;
  if (KS > 0) {
    constexpr int kWin = (KS > 0 ? KS : 1) + 7;                  // samples feeding 8 outputs
    constexpr int kWin4 = (kWin + 3) / 4;
    for (int t0 = 8 * threadIdx.x; t0 < T; t0 += 8 * kSalThreads) {
      float win[4 * kWin4];
#pragma unroll
      for (int q = 0; q < kWin4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(a + t0 + 4 * q);
        win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
      }
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < kWin; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j - u >= 0 && j - u < KS) acc[u] = fmaf(taps.w[j - u], win[j], acc[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u;
        if (t < T) {
          const float v = t < f4 ? acc[u] : 0.f;   // saliency.py:78-79 zeroes the tail again
          s[t] = v;
          lmin = fminf(lmin, v);
        }
      }
    }
  } else {
    for (int t = threadIdx.x; t < T; t += kSalThreads) {
      float acc = 0.f;
      if (t < f4)
        for (int j = 0; j < ksize; ++j) acc = fmaf(taps.w[j], a[t + j], acc);
      s[t] = acc;
      lmin = fminf(lmin, acc);
    }
  }
  PCGMIX_SCLOCK(2);
  const float rmin = block_reduce(lmin, red, false);
  PCGMIX_SCLOCK(3);
  float lmax = -INFINITY;
  for (int t = threadIdx.x; t < T; t += kSalThreads) {
    const float v = __fsub_rn(s[t], rmin);  // saliency.py:83
    s[t] = v;
    lmax = fmaxf(lmax, v);
  }
  const float rmax = block_reduce(lmax, red, true);
  PCGMIX_SCLOCK(4);
  for (int t = threadIdx.x; t < T; t += kSalThreads) {
    float v = __fdiv_rn(s[t], rmax);  // saliency.py:84; 0/0 -> NaN -> 0 (saliency.py:87)
    if (v != v) v = 0.f;
    __builtin_nontemporal_store(v, sal + (size_t)b * T + t);
  }
  PCGMIX_SCLOCK(5);
}

// Spectrogram saliency (saliency.py:93-113, dim = 2): |grad| of a (F, W) image -> zero the columns
// t >= f[4] -> sum over the frequency rows -> `ksize`-tap Gaussian along time (zero 'same' padding)
// -> zero the tail again -> min/max normalisation over the cycle's OWN columns [0, f[4]) only (the
// 1D branch normalises over the whole row; here the tail is not part of the minimum) -> NaN -> 0.
// One block per image; the row sum is split over G = 256 / W thread groups (rows g, g + G, ...)
// whose partial columns are added in group order.  W <= 1024, ksize <= 255.
constexpr int kSal2dThreads = 256;
__global__ __launch_bounds__(kSal2dThreads) void saliency_post2d_kernel(
    const float* __restrict__ grad, const int32_t* __restrict__ frames, float* __restrict__ sal,
    Taps taps, int ksize, int B, int F, int W) {
  extern __shared__ __align__(16) float smem[];
  __shared__ float red[kSal2dThreads / 64];
  const int b = blockIdx.x, half = ksize / 2;
  const int G = W >= kSal2dThreads ? 1 : kSal2dThreads / W;
  float* part = smem;                       // part[g * W + t]
  float* a = smem + (size_t)G * W;          // a[half + t], zero halo
  float* s = a + W + ksize - 1;             // s[t]
  int f4 = frames[b * 5 + 4];
  f4 = f4 < 0 ? 0 : (f4 > W ? W : f4);
  const float* img = grad + (size_t)b * F * W;
  for (int i = threadIdx.x; i < G * W; i += kSal2dThreads) {
    const int g = i / W, t = i - g * W;
    float acc = 0.f;
    if (t < f4)
      for (int r = g; r < F; r += G) acc = __fadd_rn(acc, fabsf(img[(size_t)r * W + t]));
    part[i] = acc;
  }
  for (int i = threadIdx.x; i < W + ksize - 1; i += kSal2dThreads) a[i] = 0.f;
  __syncthreads();
  for (int t = threadIdx.x; t < W; t += kSal2dThreads) {
    float acc = 0.f;
    for (int g = 0; g < G; ++g) acc = __fadd_rn(acc, part[g * W + t]);
    a[half + t] = acc;
  }
  __syncthreads();
  auto reduce = [&](float v, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float other = __shfl_xor(v, o, 64);
      v = is_max ? fmaxf(v, other) : fminf(v, other);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < kSal2dThreads / 64; ++i) r = is_max ? fmaxf(r, red[i]) : fminf(r, red[i]);
    return r;
  };
  float lmin = INFINITY;
  for (int t = threadIdx.x; t < W; t += kSal2dThreads) {
    float acc = 0.f;
    if (t < f4) {
      for (int j = 0; j < ksize; ++j) acc = fmaf(taps.w[j], a[t + j], acc);
      lmin = fminf(lmin, acc);
    }
    s[t] = acc;
  }
  const float rmin = reduce(lmin, false);
  float lmax = -INFINITY;
  for (int t = threadIdx.x; t < f4; t += kSal2dThreads) {
    const float v = __fsub_rn(s[t], rmin);  // saliency.py:108
    s[t] = v;
    lmax = fmaxf(lmax, v);
  }
  const float rmax = reduce(lmax, true);
  for (int t = threadIdx.x; t < W; t += kSal2dThreads) {
    float v = t < f4 ? __fdiv_rn(s[t], rmax) : 0.f;  // saliency.py:109; 0/0 -> NaN -> 0 (:111)
    if (v != v) v = 0.f;
    sal[(size_t)b * W + t] = v;
  }
}

// ---- numpy's pairwise float32 summation (numpy/_core/src/umath/loops_utils.h.src), restated ----
// np.sum over a contiguous float32 array of n elements = 0 + pairwise(a, n) where
//   n < 8      sequential from 0
//   n <= 128   8 strided accumulators, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
//              n % 8 tail added sequentially
//   n > 128    n2 = n/2 - (n/2) % 8;  pairwise(a, n2) + pairwise(a + n2, n - n2)
// Every operation is a separately rounded float32 add.  `elem(i)` yields element i.
// 16-byte LDS vector whose address is only 4-byte aligned (a lane's window starts at an
// arbitrary sample): gfx950 serves it with one ds_read_b128.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4al __attribute__((ext_vector_type(4), aligned(16)));

// `elem.get(i)` yields element i, `elem.get8(i, v)` the eight elements i .. i+7.
template <class F>
__device__ __forceinline__ float pw_leaf(F elem, int start, int n) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = __fadd_rn(res, elem.get(start + i));
    return res;
  }
  float r[8];
  elem.get8(start, r);
  int i = 8;
  const int nn = n - (n % 8);
  // Four 8-element groups per round, every load of the round issued before its first add: the
  // adds of one accumulator stay in numpy's order (r[j] += a[i+j], then a[i+8+j], ...), but a lane
  // now waits for LDS once per 32 elements instead of once per 8 — the launch is as long as its
  // longest lane's chain of such waits.
  for (; i + 24 < nn; i += 32) {
    float v0[8], v1[8], v2[8], v3[8];
    elem.get8(start + i, v0);
    elem.get8(start + i + 8, v1);
    elem.get8(start + i + 16, v2);
    elem.get8(start + i + 24, v3);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v0[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v1[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v2[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v3[j]);
  }
  for (; i < nn; i += 8) {
    float v[8];
    elem.get8(start + i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v[j]);
  }
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, elem.get(start + i));
  return res;
}

// AL: p is 16-byte aligned -> two ds_read_b128.  Otherwise (4-byte aligned window) the compiler
// emits four ds_read2_b32 — an unaligned b128 is split by the hardware and slower still.  The
// displacement search keeps the longer segment in four copies, copy c shifted by c samples, so
// that a window starting at ANY sample d is read 16-byte aligned from copy d & 3 (see
// salopt_disp_kernel); only segments too long for four copies in LDS take the unaligned form.
template <bool AL>
__device__ __forceinline__ void lds_get8(const float* p, float (&v)[8]) {
  if (AL) {
    const f4al a = *reinterpret_cast<const f4al*>(p), b = *reinterpret_cast<const f4al*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const f4u a = *reinterpret_cast<const f4u*>(p), b = *reinterpret_cast<const f4u*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}

// Accessors of the three summed sequences.
template <bool AL>
struct SeqPlain {               // a[i]
  const float* a;
  __device__ __forceinline__ float get(int i) const { return a[i]; }
  __device__ __forceinline__ void get8(int i, float (&v)[8]) const { lds_get8<AL>(a + i, v); }
};
template <int MODE, bool AL>
struct SeqMid {                 // op(l[i], s[i]); own_longer decides which one is the OWN saliency
  const float* l;
  const float* s;
  float lam, oml;
  bool own_longer;
  __device__ __forceinline__ float op(float lv, float sv) const {
    if (MODE == 0) {
      // np.maximum on finite values.  fmaxf() compiles to v_max_f32 PLUS one canonicalising
      // v_max_f32 x, x per operand (IEEE quieting of signalling NaNs): three instructions where
      // one does the work — the saliency maps hold no NaN (saliency.py:87 replaces them by 0).
      float r;
      asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(lv), "v"(sv));
      return r;
    }
    const float s1 = own_longer ? lv : sv, s2 = own_longer ? sv : lv;
    return __fadd_rn(__fmul_rn(s1, lam), __fmul_rn(s2, oml));   // s1*lam + s2*(1-lam)
  }
  __device__ __forceinline__ float get(int i) const { return op(l[i], s[i]); }
  __device__ __forceinline__ void get8(int i, float (&v)[8]) const {
    float a[8], b[8];
    lds_get8<AL>(l + i, a);
    lds_get8<true>(s + i, b);          // the shorter segment starts 16-byte aligned
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = op(a[j], b[j]);
  }
};

// Middle sum of the lambda-weighted objective with both segments scaled ONCE at staging
// (lng * lam or (1-lam), sht * the other): the reference forms lam*s1 + (1-lam)*s2 element by
// element for every candidate (augmentations.py:111); the two products are rounded on their own,
// so scaling each sample once and adding per candidate gives the same bits (the add is
// commutative) with one operation per element instead of three.
struct SeqAdd {
  const float* l;
  const float* s;
  __device__ __forceinline__ float get(int i) const { return __fadd_rn(l[i], s[i]); }
  __device__ __forceinline__ void get8(int i, float (&v)[8]) const {
    float a[8], b[8];
    lds_get8<true>(l + i, a);
    lds_get8<true>(s + i, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __fadd_rn(a[j], b[j]);
  }
};

// The split tree is walked leaf by leaf, in order, without a memory stack.  A leaf is a PATH from
// the root — DEPTH bits, MSB first, 0 = left child — and its range follows from the path by DEPTH
// unrolled split steps (a node splits while it holds more than 128 elements, so subtrees end at
// different depths; the unused low bits of a path are zero).  After a leaf's sum the walk climbs:
// while the node is a right child its parked left sibling is added in front of it (acc[level],
// statically indexed registers); the first time it is a left child it is parked and the walk
// moves to the leftmost leaf of the right sibling — which is simply path + (1 << (DEPTH - depth)):
// the carry runs through the right-child bits just consumed.  ~40 instructions per leaf at
// DEPTH = 4; the first version recomputed every node's range from the root on the way down AND up
// with compare-select register arrays, ~150 instructions per node visit, and those walks — not the
// leaf sums — were most of a candidate's chain.
// DEPTH = 4 covers n <= 1928 (brute-forced: the deepest leaf stays <= 128 elements), 8 covers
// 128 * 2^8 = 32768 (T is capped at 19000); a node that is still larger at depth DEPTH is summed
// as one leaf by pw_leaf, exactly as before.
constexpr int kPwShallowN = 1928;

__device__ __forceinline__ int pw_split(int n) {
  int n2 = n / 2;
  return n2 - (n2 % 8);
}

template <int DEPTH, class F>
__device__ __forceinline__ float pw_sum(F elem, int n) {
  if (n <= 128) return pw_leaf(elem, 0, n);
  float acc[DEPTH];
#pragma unroll
  for (int l = 0; l < DEPTH; ++l) acc[l] = 0.f;
  unsigned path = 0;
  for (;;) {
    int start = 0, len = n, depth = 0;
#pragma unroll
    for (int lvl = 0; lvl < DEPTH; ++lvl) {
      if (len > 128) {
        const int n2 = pw_split(len);
        if ((path >> (DEPTH - 1 - lvl)) & 1u) { start += n2; len -= n2; } else { len = n2; }
        depth = lvl + 1;
      }
    }
    float v = pw_leaf(elem, start, len);
    bool parked = false;
#pragma unroll
    for (int l = DEPTH; l >= 1; --l) {
      if (l <= depth && !parked) {
        if ((path >> (DEPTH - l)) & 1u) {
          v = __fadd_rn(acc[l - 1], v);       // right child: left sibling + this
        } else {
          acc[l - 1] = v;                     // left child: wait for the right sibling
          parked = true;
        }
      }
    }
    if (!parked) return v;                    // climbed through the root
    path += 1u << (DEPTH - depth);
  }
}

// kDispSplit blocks of 256 threads per (state k, sample b): block z takes the candidates
// d = 256 z + lane, + 1024, ...; a block without a candidate exits at once.  Round 1 ran ONE
// 1024-thread block per pair with 2*T floats of LDS: two blocks per CU, and of their 32 waves
// only the ~10 that held a candidate worked — the rest sat at the closing barrier and kept the
// wave slots (SQ counters, profiles/r2_disp_sq_counters.json: 75 % of all wave-cycles waiting,
// VALU issue 20 % of the launch).  Small blocks whose LDS is sized by the longest heart state of
// the batch (`max_len`, known on the host) put eight working blocks on a CU.  Every block
// writes its first strict maximum (value, displacement) to `part`; salopt_finalize_kernel picks
// the greatest value, smallest displacement on ties — the first strict maximum of the reference's
// ascending scan (augmentations.py:76).
// LDS: lng[nL] (the longer state's saliency), sht[nS] (the shorter one's).
// (kDispSplit = 4 blocks per pair: pcgmix_kernels.h — the splice kernel can read `part` itself.)

__host__ __device__ inline int disp_copy_stride(int max_len) {   // floats; = 16 (mod 64), >= max_len
  return ((max_len + 63) & ~63) + 16;
}

// COPIES: four shifted copies of the longer segment (aligned 16-byte LDS reads); false: one copy.
// DEPTH: levels of numpy's split tree the walk provides for (4 when max_len <= kPwShallowN, else 8).
// Probe builds only (-DPCGMIX_PHASE_CLOCK, profiles/probes/disp_phase_clock.py): every block leaves
// wall_clock64 (100 MHz) at its phase boundaries and its HW_ID / XCC_ID in g_disp_clock.
#ifdef PCGMIX_PHASE_CLOCK
constexpr int kDispClockBlocks = 8192;
__device__ long long g_disp_clock[kDispClockBlocks * 8];
#define PCGMIX_DCLOCK(i)                                                                            \
  do {                                                                                              \
    const unsigned lin_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;           \
    if (threadIdx.x == 0 && lin_ < kDispClockBlocks) g_disp_clock[lin_ * 8 + (i)] = wall_clock64(); \
  } while (0)
#define PCGMIX_DCLOCK_ID()                                                                          \
  do {                                                                                              \
    const unsigned lin_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;           \
    if (threadIdx.x == 0 && lin_ < kDispClockBlocks) {                                              \
      g_disp_clock[lin_ * 8 + 6] = __builtin_amdgcn_s_getreg(0xF804);                               \
      g_disp_clock[lin_ * 8 + 7] = __builtin_amdgcn_s_getreg(0xF814);                               \
    }                                                                                               \
  } while (0)
// shader clocks (s_memtime) across the scan, kept in the upper half of slot 7
#define PCGMIX_DCLOCK_SH(var) const long long var = clock64()
#define PCGMIX_DCLOCK_SH_END(var)                                                                   \
  do {                                                                                              \
    const unsigned lin_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;           \
    if (threadIdx.x == 0 && lin_ < kDispClockBlocks)                                                \
      g_disp_clock[lin_ * 8 + 7] |= (clock64() - var) << 8;                                         \
  } while (0)
#else
#define PCGMIX_DCLOCK(i) do { } while (0)
#define PCGMIX_DCLOCK_ID() do { } while (0)
#define PCGMIX_DCLOCK_SH(var) do { } while (0)
#define PCGMIX_DCLOCK_SH_END(var) do { } while (0)
#endif

// PLAN: the grid is the host's list of blocks (DispPlan, in the kernel arguments), heaviest first.
template <int MODE, bool COPIES, int DEPTH, bool PLAN = false>  // MODE 0: envelope (max), 1: lambda-weighted sum
__global__ __launch_bounds__(kDispThreads) void salopt_disp_kernel(
    const float* __restrict__ sal, const int32_t* __restrict__ frames,
    const int32_t* __restrict__ mix_idx, float lam, float oml, float2* __restrict__ part, int B,
    int T, int max_len, const uint4* __restrict__ pay_src, uint4* __restrict__ pay_dst,
    int pay_n16, const PartnerPack pk, const std::conditional_t<PLAN, DispPlan, DispNoPlan> plan) {
  extern __shared__ __align__(16) float smem[];
  PCGMIX_DCLOCK(0);
  PCGMIX_DCLOCK_ID();
  // Side job of the LAST block in launch order (shortest state, last candidate slice: it almost
  // never has candidates of its own): copy pay_n16 16-byte words from pay_src — host memory the
  // device can read, e.g. the warp knots in the step context's pinned slot — to pay_dst.  The
  // splice kernel launched behind this one reads them from device memory; a hipMemcpyAsync of
  // this size (49 KB at bs 256) takes the SDMA path and stalls the stream for ~25 us.
  if (pay_n16 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1)
    for (int i = threadIdx.x; i < pay_n16; i += kDispThreads) pay_dst[i] = pay_src[i];
  __shared__ float best_v[kDispThreads / 64];
  __shared__ int best_d[kDispThreads / 64];
  // Sample index fastest, states ordered by expected work (diastole, systole, S1, S2): with the
  // state as the fast index and a batch that is a multiple of 4, round-robin dispatch handed every
  // diastole block (the long ones) to the same quarter of the CUs.
  // Blocks are dealt to the CUs in launch order, and the kDispSplit blocks of one pair are
  // gridDim.x * 4 apart — a multiple of the CU count at the benchmark batch, i.e. the SAME CU,
  // whose LDS pipe the heaviest pair then saturates alone.  Rotating the sample index by a
  // z-dependent offset puts them on different CUs.
  // (Round 4 first tried the PAIRS sorted by chain length with the slices still outermost: no gain,
  // profiles/r4_disp_dispatch_order_null.txt — the per-block clock later showed why: what ends the
  // launch are the slices z >= 1 of the long pairs, and they entered last in either order.)
  // Launch order: sample, then slice, then state.  (Until round 4: sample, state, slice — the
  // slices z >= 1 of the long diastole pairs, the blocks with the longest chains, then ENTERED
  // 10-12 us into the launch, behind 2048 other blocks at ~60 dispatcher cycles each:
  // profiles/r4_disp_phase_clock.txt.)
  int z, b, k;
  if constexpr (PLAN) {
    const unsigned id = plan.e[blockIdx.x];
    b = (int)(id >> 4);
    k = (int)(id >> 2) & 3;
    z = (int)id & 3;
  } else {
    z = blockIdx.y;
    b = (int)((blockIdx.x + (unsigned)z * (gridDim.x / kDispSplit + 3)) % gridDim.x);
    k = (0x2013 >> (4 * blockIdx.z)) & 3;   // blockIdx.z 0,1,2,3 -> state 3,1,0,2
  }
  float2* out = part + ((size_t)b * 4 + k) * kDispSplit + z;
  int m = pk.n ? partner_get(pk, b) : mix_idx[b];
  m = (m < 0 || m >= B) ? b : m;
  int a1 = frames[b * 5 + k], e1 = frames[b * 5 + k + 1];
  int a2 = frames[m * 5 + k], e2 = frames[m * 5 + k + 1];
  a1 = a1 < 0 ? 0 : (a1 > T ? T : a1);
  e1 = e1 < a1 ? a1 : (e1 > T ? T : e1);
  a2 = a2 < 0 ? 0 : (a2 > T ? T : a2);
  e2 = e2 < a2 ? a2 : (e2 > T ? T : e2);
  e1 = e1 - a1 > max_len ? a1 + max_len : e1;       // memory safety: LDS holds 2 * max_len floats
  e2 = e2 - a2 > max_len ? a2 + max_len : e2;       // (the caller passes the true maximum)
  const int n1 = e1 - a1, n2 = e2 - a2;
  const bool own_longer = n1 > n2;
  const int nL = own_longer ? n1 : n2, nS = own_longer ? n2 : n1;
  if (PLAN && (n1 == n2 || (z + 1) * kDispThreads > nL - nS)) {
    // the plan lists only slices with candidates: the last of them marks the rest of its pair empty
    if ((int)threadIdx.x > z && threadIdx.x < kDispSplit)
      out[(int)threadIdx.x - z] = float2{-INFINITY, __int_as_float(0x7fffffff)};
  }
  if (n1 == n2 || z * kDispThreads > nL - nS) {  // no search (:226-229) / no candidate for this block
    if (threadIdx.x == 0) *out = float2{-INFINITY, __int_as_float(0x7fffffff)};
    PCGMIX_DCLOCK(5);
    return;
  }
  PCGMIX_DCLOCK(1);
  const float* gl = sal + (size_t)(own_longer ? b : m) * T + (own_longer ? a1 : a2);
  const float* gs = sal + (size_t)(own_longer ? m : b) * T + (own_longer ? a2 : a1);
  // LDS: sht[nS] (16-byte aligned), then four copies of the longer segment, copy c holding
  // lng[c ..] at its start: a window lng[d .. d+7] is copy (d & 3) at offset d - (d & 3), a
  // multiple of 4 floats.  Copies are `cs` floats apart with cs = 16 (mod 64): the 16 lanes of a
  // ds_read_b128 pass (4 consecutive offsets x 4 copies) then cover all 64 banks exactly once.
  const int cs = COPIES ? disp_copy_stride(max_len) : 0;
  // MODE 1 with copies: the four copies and sht hold the segments already multiplied by lambda /
  // (1 - lambda) (SeqAdd); the head and tail sums of the longer-is-own case need the unscaled
  // segment, kept once in front of the copies (`raw`; its tail windows are read unaligned).
  constexpr bool kPrescale = MODE == 1 && COPIES;
  const float fl = own_longer ? lam : oml, fs = own_longer ? oml : lam;   // lng / sht multiplier
  float* sht = smem;
  float* raw = smem + ((max_len + 3) & ~3);
  float* lng = kPrescale ? raw + cs : raw;           // copy 0 == the (scaled) segment itself
  {  // both segments staged with all of a lane's loads in flight together (a loop with runtime
     // bounds compiles to load, wait, store, next load: up to eight serialised L2 round trips)
    constexpr int kMaxPer = 8;                       // covers states up to 2048 samples per pass
    for (int base = 0; base < nL + nS; base += kMaxPer * kDispThreads) {
      float v[kMaxPer];
#pragma unroll
      for (int u = 0; u < kMaxPer; ++u) {
        // clamped, never predicated: a predicated load compiles to a branch with its own
        // s_waitcnt vmcnt(0) and the eight L2 round trips of a lane run one after the other
        int i = base + u * kDispThreads + threadIdx.x;
        i = i < nL + nS ? i : nL + nS - 1;
        const float* p = i < nL ? gl + i : gs + (i - nL);
        v[u] = *p;
      }
#pragma unroll
      for (int u = 0; u < kMaxPer; ++u) {
        const int i = base + u * kDispThreads + threadIdx.x;
        if (i < nL) {
          const float sv = kPrescale ? __fmul_rn(v[u], fl) : v[u];
          if (kPrescale) raw[i] = v[u];
          lng[i] = sv;
          if (COPIES) {
#pragma unroll
            for (int c = 1; c < 4; ++c)
              if (i >= c) lng[c * cs + i - c] = sv;
          }
        } else if (i < nL + nS) {
          sht[i - nL] = kPrescale ? __fmul_rn(v[u], fs) : v[u];
        }
      }
    }
  }
  __syncthreads();
  PCGMIX_DCLOCK(2);
  PCGMIX_DCLOCK_SH(sh0_);

  float bv = -INFINITY;
  int bd = 0x7fffffff;
  for (int d = z * kDispThreads + threadIdx.x; d <= nL - nS; d += kDispSplit * kDispThreads) {
    const float* win = COPIES ? lng + (d & 3) * cs + (d & ~3) : lng + d;
    float cur;
    if (kPrescale) cur = pw_sum<DEPTH>(SeqAdd{win, sht}, nS);
    else cur = pw_sum<DEPTH>(SeqMid<MODE, COPIES>{win, sht, lam, oml, own_longer}, nS);
    if (own_longer) {  // np.sum(s1[:d]) + np.sum(mid) + np.sum(s1[d+n2:])   (:76-78, :111-113)
      const int t0 = d + nS;
      const float head = pw_sum<DEPTH>(SeqPlain<true>{raw}, d);
      float tail;
      if (kPrescale) tail = pw_sum<DEPTH>(SeqPlain<false>{raw + t0}, nL - nS - d);
      else tail = pw_sum<DEPTH>(SeqPlain<COPIES>{COPIES ? lng + (t0 & 3) * cs + (t0 & ~3) : lng + t0},
                                nL - nS - d);
      cur = __fadd_rn(__fadd_rn(head, cur), tail);
    }
    if (cur > bv) {  // ascending d per lane: strict '>' keeps the first maximum
      bv = cur;
      bd = d;
    }
  }
  PCGMIX_DCLOCK(3);       // thread 0's wave is done with its candidates
  PCGMIX_DCLOCK_SH_END(sh0_);
  // arg-max across the block; ties go to the smallest d (= first strict maximum of the scan)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int od = __shfl_xor(bd, o, 64);
    if (ov > bv || (ov == bv && od < bd)) {
      bv = ov;
      bd = od;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    best_v[threadIdx.x >> 6] = bv;
    best_d[threadIdx.x >> 6] = bd;
  }
  __syncthreads();
  PCGMIX_DCLOCK(4);       // every wave of the block is done
  if (threadIdx.x == 0) {
    for (int i = 1; i < kDispThreads / 64; ++i)
      if (best_v[i] > bv || (best_v[i] == bv && best_d[i] < bd)) {
        bv = best_v[i];
        bd = best_d[i];
      }
    *out = float2{bv, __int_as_float(bd)};
  }
  PCGMIX_DCLOCK(5);
}

__global__ void salopt_finalize_kernel(const float2* __restrict__ part, int32_t* __restrict__ disp,
                                       int n /* B * 4 */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float bv = -INFINITY;
  int bd = 0x7fffffff;
#pragma unroll
  for (int z = 0; z < kDispSplit; ++z) {
    const float2 p = part[(size_t)i * kDispSplit + z];
    const int d = __float_as_int(p.y);
    if (p.x > bv || (p.x == bv && d < bd)) {
      bv = p.x;
      bd = d;
    }
  }
  disp[i] = (bd == 0x7fffffff) ? 0 : bd;  // equal lengths, or an all-NaN objective: the reference keeps 0
}

}  // namespace pcgmix

#ifdef PCGMIX_PHASE_CLOCK
extern "C" int pcgmix_salpost_phase_clock(long long* out, int n_blocks) {
  if (n_blocks > 1024) return hipErrorInvalidValue;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcgmix::g_salpost_clock), (size_t)n_blocks * 8 * sizeof(long long));
}
extern "C" int pcgmix_disp_phase_clock(long long* out, int n_blocks) {
  if (n_blocks > pcgmix::kDispClockBlocks) return hipErrorInvalidValue;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcgmix::g_disp_clock), (size_t)n_blocks * 8 * sizeof(long long));
}
#endif

extern "C" int pcgmix_saliency_post_f32(const float* grad, const int32_t* frames, float* sal,
                                        int ksize, double sigma, int B, int C, int T,
                                        pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!grad || !frames || !sal) return hipErrorInvalidValue;
  if (B < 0 || C <= 0 || T <= 0 || ksize < 1 || ksize > kMaxTaps || !(ksize & 1) || !(sigma > 0))
    return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  const size_t a_len = ((size_t)T + ksize - 1 + 8 + 3) & ~(size_t)3;
  const size_t lds = sizeof(float) * (a_len + (size_t)T);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  // gaussian_kernel(), saliency.py:15-18: Python float64 arithmetic, then torch.FloatTensor
  Taps taps;
  const int half = ksize / 2;
  for (int j = 0; j < ksize; ++j) {
    const double xr = (double)(j - half);
    const double w = 1.0 / (sigma * sqrt(2.0 * M_PI)) * exp(-(xr * xr) / (2.0 * (sigma * sigma)));
    taps.w[j] = (float)w;
  }
  for (int j = ksize; j < kMaxTaps; ++j) taps.w[j] = 0.f;
  static unsigned long long lds_ok101 = 0, lds_ok0 = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ksize == 101) {             // the reference's gauss_k_n (augmentations.py:966)
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(saliency_post_kernel<101>),
                                       &lds_ok101, 150 * 1024))
      return (int)e;
    hipLaunchKernelGGL(saliency_post_kernel<101>, dim3((unsigned)B), dim3(kSalThreads), lds, st, grad,
                       frames, sal, taps, ksize, B, C, T);
  } else {
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(saliency_post_kernel<0>),
                                       &lds_ok0, 150 * 1024))
      return (int)e;
    hipLaunchKernelGGL(saliency_post_kernel<0>, dim3((unsigned)B), dim3(kSalThreads), lds, st, grad,
                       frames, sal, taps, ksize, B, C, T);
  }
  return (int)hipGetLastError();
}

extern "C" int pcgmix_saliency_post2d_f32(const float* grad, const int32_t* frames, float* sal,
                                          int ksize, double sigma, int B, int F, int W,
                                          pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!grad || !frames || !sal) return hipErrorInvalidValue;
  if (B < 0 || F <= 0 || W <= 0 || W > 1024 || ksize < 1 || ksize > kMaxTaps || !(ksize & 1) ||
      !(sigma > 0))
    return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  Taps taps;
  const int half = ksize / 2;
  for (int j = 0; j < ksize; ++j) {                  // gaussian_kernel(), saliency.py:15-18
    const double xr = (double)(j - half);
    const double w = 1.0 / (sigma * sqrt(2.0 * M_PI)) * exp(-(xr * xr) / (2.0 * (sigma * sigma)));
    taps.w[j] = (float)w;
  }
  for (int j = ksize; j < kMaxTaps; ++j) taps.w[j] = 0.f;
  const int G = W >= kSal2dThreads ? 1 : kSal2dThreads / W;
  const size_t lds = sizeof(float) * ((size_t)G * W + (size_t)W + ksize - 1 + (size_t)W);
  hipLaunchKernelGGL(saliency_post2d_kernel, dim3((unsigned)B), dim3(kSal2dThreads), lds,
                     reinterpret_cast<hipStream_t>(stream), grad, frames, sal, taps, ksize, B, F, W);
  return (int)hipGetLastError();
}

extern "C" long long pcgmix_salopt_workspace_bytes(int B) {
  return B <= 0 ? 0 : (long long)B * 4 * pcgmix::kDispSplit * (long long)sizeof(float2);
}

extern "C" int pcgmix_salopt_disp_f32(const float* sal, const int32_t* frames,
                                      const int32_t* mix_idx, float lam, int mode, int32_t* disp,
                                      void* workspace, int max_len, int B, int T,
                                      pcgmix_stream_t stream) {
  if (!disp) return hipErrorInvalidValue;
  return pcgmix::launch_salopt_search(sal, frames, mix_idx, lam, mode, disp, workspace, max_len, B, T,
                                      reinterpret_cast<hipStream_t>(stream));
}

// The saliency-guided splice in one call: the displacement search, then the fused splice(+warp)
// kernel, whose blocks reduce the search's per-block results for their own sample themselves —
// no finalize launch between the two.  disp_out (optional): the displacements as int32 (B,4),
// written by a finalize launch BEHIND the splice (off the critical path).
extern "C" int pcgmix_salopt_mix_warp_f32(const float* x, float* y, const float* sal,
                                          const int32_t* frames, const int32_t* mix_idx, float lam,
                                          int mode, const double* knots, const double* spline_op,
                                          int n_knots, void* workspace, int max_len,
                                          int32_t* disp_out, int B, int C, int T,
                                          pcgmix_stream_t stream) {
  using namespace pcgmix;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!x || !y || B < 0) return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  int err = launch_salopt_search(sal, frames, mix_idx, lam, mode, nullptr, workspace, max_len, B, T, s);
  if (err) return err;
  const float2* part = static_cast<const float2*>(workspace);
  err = launch_mix_warp(x, y, frames, mix_idx, nullptr, lam, knots, spline_op, n_knots, nullptr, B, C,
                        T, s, nullptr, nullptr, 0, part);
  if (err || !disp_out) return err;
  hipLaunchKernelGGL(salopt_finalize_kernel, dim3((unsigned)((B * 4 + 255) / 256)), dim3(256), 0, s,
                     part, disp_out, B * 4);
  return (int)hipGetLastError();
}

namespace pcgmix {
template <int MODE, bool COPIES, int DEPTH, bool PLAN>
static int launch_disp_variant(unsigned long long* lds_ok, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                               const float* sal, const int32_t* frames, const int32_t* mix_idx, float lam,
                               float oml, float2* part, int B, int T, int max_len, const uint4* pay_src,
                               uint4* pay_dst, int pay_n16, const PartnerPack& pk, const DispPlan* plan) {
  auto kern = salopt_disp_kernel<MODE, COPIES, DEPTH, PLAN>;
  if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(kern), lds_ok, 150 * 1024)) return (int)e;
  if constexpr (PLAN)
    hipLaunchKernelGGL(kern, grid, block, lds, s, sal, frames, mix_idx, lam, oml, part, B, T, max_len, pay_src,
                       pay_dst, pay_n16, pk, *plan);
  else
    hipLaunchKernelGGL(kern, grid, block, lds, s, sal, frames, mix_idx, lam, oml, part, B, T, max_len, pay_src,
                       pay_dst, pay_n16, pk, DispNoPlan{0});
  return 0;
}
}  // namespace pcgmix

bool pcgmix::plan_salopt_blocks(const int32_t* frames_h, const int32_t* mix_h, const int16_t* mix16,
                                int B, int T, int max_len, DispPlan* out) {
  out->n = 0;
  if (!frames_h || (!mix_h && !mix16) || B <= 0 || B > kPackB || T <= 0) return false;
  if (max_len <= 0 || max_len > T) max_len = T;
  auto len = [&](int b, int k) {     // the kernel's clamps
    int a = frames_h[b * 5 + k], e = frames_h[b * 5 + k + 1];
    a = a < 0 ? 0 : (a > T ? T : a);
    e = e < a ? a : (e > T ? T : e);
    const int n = e - a;
    return n > max_len ? max_len : n;
  };
  // counting sort by chain length (own state's length x passes of the busiest lane) in steps of 16
  constexpr int kBuckets = 512;
  uint16_t cnt[kBuckets + 1] = {};
  uint16_t key[kPackB * 16];
  uint16_t ids[kPackB * 16];
  int n = 0;
  for (int b = 0; b < B; ++b) {
    int m = mix16 ? (int)mix16[b] : mix_h[b];
    m = (m < 0 || m >= B) ? b : m;
    for (int k = 0; k < 4; ++k) {
      const int n1 = len(b, k), n2 = len(m, k);
      const int dm = n1 > n2 ? n1 - n2 : n2 - n1;             // last candidate
      const unsigned base = ((unsigned)b << 4) | ((unsigned)k << 2);
      if (dm == 0) {
        key[n] = 0;
        ids[n++] = (uint16_t)base;
        continue;
      }
      for (int z = 0; z < kDispSplit && z * kDispThreads <= dm; ++z) {
        const int passes = (dm - z * kDispThreads) / (kDispSplit * kDispThreads) + 1;
        int c = (n1 * passes) >> 4;
        key[n] = (uint16_t)(c >= kBuckets ? kBuckets - 1 : c);
        ids[n++] = (uint16_t)(base | (unsigned)z);
      }
    }
  }
  if (n > kDispPlanMax) return false;
  for (int i = 0; i < n; ++i) ++cnt[kBuckets - 1 - key[i]];  // descending
  int pos = 0;
  for (int i = 0; i < kBuckets; ++i) {
    const int c = cnt[i];
    cnt[i] = (uint16_t)pos;
    pos += c;
  }
  for (int i = 0; i < n; ++i) out->e[cnt[kBuckets - 1 - key[i]]++] = ids[i];
  out->n = n;
  return true;
}

// Host only: the launch plan of pcgmix_salopt_disp_hosted_f32 for inspection and tests.  ids_out
// receives up to `cap` block ids ((sample << 4) | (state << 2) | slice) in launch order; returns
// their number, 0 when no plan is made for this shape (B > 256, more blocks than a plan holds),
// or a negative value for bad arguments.
extern "C" int pcgmix_salopt_plan(const int32_t* frames_host, const int32_t* mix_host, int B, int T,
                                  int max_len, uint16_t* ids_out, int cap) {
  if (!frames_host || !mix_host || !ids_out || B <= 0 || T <= 0 || cap < 0) return -1;
  pcgmix::DispPlan plan;
  if (!pcgmix::plan_salopt_blocks(frames_host, mix_host, nullptr, B, T, max_len, &plan)) return 0;
  const int n = plan.n < cap ? plan.n : cap;
  for (int i = 0; i < n; ++i) ids_out[i] = plan.e[i];
  return plan.n;
}

// pcgmix_salopt_disp_f32 for a caller that also holds the boundaries and the partners on the HOST
// (the reference's own situation: augmentations.py:210-287 receives them as CPU arrays): the
// launch then consists of the blocks that have candidates, longest chain first.
extern "C" int pcgmix_salopt_disp_hosted_f32(const float* sal, const int32_t* frames,
                                             const int32_t* mix_idx, float lam, int mode,
                                             int32_t* disp, void* workspace, int max_len, int B, int T,
                                             pcgmix_stream_t stream, const int32_t* frames_host,
                                             const int32_t* mix_host) {
  if (!disp) return hipErrorInvalidValue;
  pcgmix::DispPlan plan;
  plan.n = 0;
  if (frames_host && mix_host) pcgmix::plan_salopt_blocks(frames_host, mix_host, nullptr, B, T, max_len, &plan);
  return pcgmix::launch_salopt_search(sal, frames, mix_idx, lam, mode, disp, workspace, max_len, B, T,
                                      reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, 0, nullptr,
                                      plan.n ? &plan : nullptr);
}

// disp == nullptr: the per-block results stay in `workspace` (no finalize launch).
int pcgmix::launch_salopt_search(const float* sal, const int32_t* frames, const int32_t* mix_idx,
                                 float lam, int mode, int32_t* disp, void* workspace, int max_len,
                                 int B, int T, hipStream_t s, const void* pay_src_v, void* pay_dst_v,
                                 int pay_n16, const int16_t* partners16, const DispPlan* plan) {
  using namespace pcgmix;
  const PartnerPack pk = make_partner_pack(partners16, B);

  const uint4* pay_src = static_cast<const uint4*>(pay_src_v);
  uint4* pay_dst = static_cast<uint4*>(pay_dst_v);
  if (pay_n16 < 0 || (pay_n16 > 0 && (!pay_src || !pay_dst ||
                                      ((reinterpret_cast<uintptr_t>(pay_src) |
                                        reinterpret_cast<uintptr_t>(pay_dst)) & 15))))
    return hipErrorInvalidValue;
  if (!sal || !frames || (!mix_idx && !pk.n) || !workspace) return hipErrorInvalidValue;
  if (B < 0 || B > 65535 || T <= 0 || (mode != 0 && mode != 1)) return hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(workspace) & 7) return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  if (max_len <= 0 || max_len > T) max_len = T;
  const size_t seg = (size_t)((max_len + 3) & ~3);
  // mode 1 keeps one unscaled copy of the longer segment in front of the four scaled ones
  size_t lds = sizeof(float) * (seg + (mode == 1 ? 5 : 4) * (size_t)disp_copy_stride(max_len));
  const bool copies = lds <= 96 * 1024;              // longer segments: one copy, unaligned windows
  if (!copies) lds = sizeof(float) * 2 * seg;
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  const bool shallow = max_len <= kPwShallowN;
  const bool planned = plan && plan->n > 0 && plan->n <= kDispPlanMax;
  const float oml = 1.0f - lam;
  float2* part = static_cast<float2*>(workspace);
  const dim3 block(kDispThreads);
  const dim3 grid = planned ? dim3((unsigned)plan->n, 1, 1) : dim3((unsigned)B, kDispSplit, 4);
  static unsigned long long lds_ok[16] = {};
#define PCGMIX_DISP(M, CP, DP, PL, SLOT)                                                               \
  do {                                                                                               \
    if (int e_ = launch_disp_variant<M, CP, DP, PL>(&lds_ok[SLOT], grid, block, lds, s, sal, frames, mix_idx, \
                                                    lam, oml, part, B, T, max_len, pay_src, pay_dst, \
                                                    pay_n16, pk, plan))                              \
      return e_;                                                                                     \
  } while (0)
#define PCGMIX_DISP_P(M, CP, DP, SLOT) \
  do { if (planned) PCGMIX_DISP(M, CP, DP, true, SLOT + 8); else PCGMIX_DISP(M, CP, DP, false, SLOT); } while (0)
#define PCGMIX_DISP_D(M, CP, SLOT) \
  do { if (shallow) PCGMIX_DISP_P(M, CP, 4, SLOT); else PCGMIX_DISP_P(M, CP, 8, SLOT + 1); } while (0)
  if (mode == 0) { if (copies) PCGMIX_DISP_D(0, true, 0); else PCGMIX_DISP_D(0, false, 2); }
  else { if (copies) PCGMIX_DISP_D(1, true, 4); else PCGMIX_DISP_D(1, false, 6); }
#undef PCGMIX_DISP_D
#undef PCGMIX_DISP_P
#undef PCGMIX_DISP
  if (disp)
    hipLaunchKernelGGL(salopt_finalize_kernel, dim3((unsigned)((B * 4 + 255) / 256)), dim3(256), 0, s,
                       part, disp, B * 4);
  return (int)hipGetLastError();
}
