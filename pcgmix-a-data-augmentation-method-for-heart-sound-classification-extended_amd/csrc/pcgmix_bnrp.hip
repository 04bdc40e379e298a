// pcgmix_bnrp.hip — BatchNorm (training) + ReLU + MaxPool of the ResNet9 blocks as fused NHWC
// kernels (gfx950).
//
// Reference: models.py:468-473 / models2d.py:13-19 (conv_block: Conv + BatchNorm + ReLU [+ MaxPool]).
// The convolutions run at ~130 TFLOP/s fp32 through MIOpen's implicit-GEMM kernels (84 % of the
// matrix peak); what is left of a ResNet9 step is memory-bound elementwise work on activations of
// 0.3-0.65 GB each: BatchNorm forward (2 kernels), ReLU, pooling, and their backward twins
// (pool, ReLU, two BatchNorm kernels) — nine to ten passes over the activation per block, each
// through HBM.  Here, with y = conv output (rows x C, channels innermost):
//
//   forward   bn_stats_kernel        one read of y  -> per-block (sum, sum of squares) per channel
//             bn_finalize_kernel     mean, 1/std, running-stat update (float64 combination)
//             bnrp_apply_kernel      one read of y  -> z = maxpool(relu(gamma * xhat + beta))
//   backward  bnrp_bwd_reduce_kernel one read of y, dz -> sums of dy and dy * xhat per channel
//             bn_bwd_finalize_kernel dgamma, dbeta, the two means BatchNorm's dx needs
//             bnrp_bwd_apply_kernel  one read of y, dz -> dx (one write)
//
// The ReLU mask and the pooling arg-max are recomputed from y (first maximum wins, as in torch's
// max_pool), so nothing but y, mean and 1/std is kept for backward.  Five passes instead of ten.
// All reductions are fixed-order (deterministic).  HBM-bound: 4 bytes per element per pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <cmath>

#include "pcgmix_kernels.h"

namespace pcgmix {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = 1024;
constexpr int kFinCh = 16;            // channels per finalize block (x 16 partial-row lanes)

struct BnShape {
  int B, H, W, C, ph, pw, Ho, Wo, Q;   // Q = C / 4 float4 per row; kBnThreads % Q == 0
};

__device__ __forceinline__ f4 f4_fma(f4 a, f4 b, f4 c) {
  return f4{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}

// Sum the per-thread float4 values of the threads that share a channel quad (tid % Q) in a fixed
// order and let thread q < Q write the result for quad q.
__device__ __forceinline__ void quad_reduce_store(f4 a, f4 b, int Q, f4* lds, float* out0,
                                                  float* out1) {
  const int tid = threadIdx.x;
  lds[tid] = a;
  lds[kBnThreads + tid] = b;
  __syncthreads();
  if (tid < Q) {
    f4 sa = lds[tid], sb = lds[kBnThreads + tid];
    for (int j = tid + Q; j < kBnThreads; j += Q) {
      sa += lds[j];
      sb += lds[kBnThreads + j];
    }
    *reinterpret_cast<f4*>(out0 + 4 * tid) = sa;
    *reinterpret_cast<f4*>(out1 + 4 * tid) = sb;
  }
}

// ------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const f4* __restrict__ y,
                                                              long long n4, int Q,
                                                              float* __restrict__ partial, int C) {
  __shared__ f4 lds[2 * kBnThreads];
  f4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  const long long stride = (long long)gridDim.x * kBnThreads;     // multiple of Q
  long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {   // four loads in flight per lane (a pure reader
    const f4 v0 = y[i], v1 = y[i + stride], v2 = y[i + 2 * stride], v3 = y[i + 3 * stride];
    s += v0; ss = f4_fma(v0, v0, ss);               // has no store to hide a load-wait-load chain behind)
    s += v1; ss = f4_fma(v1, v1, ss);
    s += v2; ss = f4_fma(v2, v2, ss);
    s += v3; ss = f4_fma(v3, v3, ss);
  }
  for (; i < n4; i += stride) {
    const f4 v = y[i];
    s += v;
    ss = f4_fma(v, v, ss);
  }
  float* p = partial + (size_t)blockIdx.x * 2 * C;
  quad_reduce_store(s, ss, Q, lds, p, p + C);
}

// Sum of the per-block partials of 16 channels: 16 lanes per channel stride over the blocks in
// float64, then a fixed-order combination through LDS.  (One thread per channel walking 1024+
// partials serially took 0.5 ms per call — more than the pass over the activation itself.)
__device__ __forceinline__ void partial_sums(const float* __restrict__ partial, int nblk, int C,
                                             int c, int lane, double (*lds)[kFinCh][16], double* s0,
                                             double* s1) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    int k = lane;
    for (; k + 7 * 16 < nblk; k += 8 * 16) {       // eight pairs of loads in flight, added in order
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        va[u] = partial[(size_t)(k + 16 * u) * 2 * C + c];
        vb[u] = partial[(size_t)(k + 16 * u) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += (double)va[u];
        b += (double)vb[u];
      }
    }
    for (; k < nblk; k += 16) {
      a += (double)partial[(size_t)k * 2 * C + c];
      b += (double)partial[(size_t)k * 2 * C + C + c];
    }
  }
  const int cl = threadIdx.x % kFinCh;
  lds[0][cl][lane] = a;
  lds[1][cl][lane] = b;
  __syncthreads();
  a = b = 0.0;
  for (int k = 0; k < 16; ++k) {
    a += lds[0][cl][k];
    b += lds[1][cl][k];
  }
  *s0 = a;
  *s1 = b;
}

// var is the biased batch variance (normalisation); the running variance gets the unbiased one,
// as torch.nn.BatchNorm does.  Block = 16 channels x 16 lanes.
__global__ __launch_bounds__(kFinCh * 16) void bn_finalize_kernel(
    const float* __restrict__ partial, int nblk, int C, double n_rows, float eps, float momentum,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, const float* __restrict__ mean_shift,
    long long* __restrict__ batches_tracked) {
  // mean_shift: a per-channel constant that was left out of y (the convolution's bias, which the
  // normalisation cancels): it belongs in the running mean.  batches_tracked: nn.BatchNorm's counter.
  if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) batches_tracked[0] += 1;
  __shared__ double lds[2][kFinCh][16];
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, lane = threadIdx.x / kFinCh;
  double s, ss;
  partial_sums(partial, nblk, C, c, lane, lds, &s, &ss);
  if (c >= C || lane != 0) return;
  const double m = s / n_rows;
  double var = ss / n_rows - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean)
    running_mean[c] = (1.f - momentum) * running_mean[c] +
                      momentum * ((float)m + (mean_shift ? mean_shift[c] : 0.f));
  if (running_var) {
    const double unbiased = n_rows > 1.0 ? var * n_rows / (n_rows - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// relu(scale * y + shift) at the ph x pw window of pooled position (b, ho, wo), channel quad q;
// returns the maximum and (through *arg) the index i * pw + j of its FIRST occurrence.
// If `raw` is given it receives y itself at that first maximum (what the backward's xhat needs:
// re-reading it by index is a dependent scalar gather per channel).
__device__ __forceinline__ f4 window_max(const f4* __restrict__ y, const BnShape& s, long long b,
                                         int ho, int wo, int q, f4 scale, f4 shift, int arg[4],
                                         f4* raw = nullptr) {
  f4 best = {-1.f, -1.f, -1.f, -1.f};                 // relu output is >= 0: any value beats this
  f4 vb = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < s.ph; ++i)
    for (int j = 0; j < s.pw; ++j) {
      const long long row = (b * s.H + (ho * s.ph + i)) * s.W + (wo * s.pw + j);
      const f4 v = y[row * s.Q + q];
      f4 a = f4_fma(v, scale, shift);
      const int idx = i * s.pw + j;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r = a[e] > 0.f ? a[e] : 0.f;
        if (r > best[e]) {
          best[e] = r;
          arg[e] = idx;
          vb[e] = v[e];
        }
      }
    }
  if (raw) *raw = vb;
  return best;
}

__global__ __launch_bounds__(kBnThreads) void bnrp_apply_kernel(
    const f4* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ invstd, const f4* __restrict__ skip,
    f4* __restrict__ z, BnShape s) {
  const int q = threadIdx.x % s.Q;                    // fixed per thread: strides are multiples of Q
  const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * q), bt = *reinterpret_cast<const f4*>(beta + 4 * q);
  const f4 mu = *reinterpret_cast<const f4*>(mean + 4 * q), is = *reinterpret_cast<const f4*>(invstd + 4 * q);
  const f4 scale = g * is, shift = bt - mu * scale;
  const long long n_out = (long long)s.B * s.Ho * s.Wo * s.Q;
  const long long stride = (long long)gridDim.x * kBnThreads;
  for (long long o = (long long)blockIdx.x * kBnThreads + threadIdx.x; o < n_out; o += stride) {
    const long long orow = o / s.Q;
    const int wo = (int)(orow % s.Wo);
    const long long t = orow / s.Wo;
    const int ho = (int)(t % s.Ho);
    const long long b = t / s.Ho;
    int arg[4];
    f4 v = window_max(y, s, b, ho, wo, q, scale, shift, arg);
    if (skip) v += skip[o];                           // residual connection (models.py:577, 581)
    z[o] = v;
  }
}

// ------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(kBnThreads) void bnrp_bwd_reduce_kernel(
    const f4* __restrict__ y, const f4* __restrict__ dz, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, BnShape s) {
  __shared__ f4 lds[2 * kBnThreads];
  const int q = threadIdx.x % s.Q;
  const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * q), bt = *reinterpret_cast<const f4*>(beta + 4 * q);
  const f4 mu = *reinterpret_cast<const f4*>(mean + 4 * q), is = *reinterpret_cast<const f4*>(invstd + 4 * q);
  const f4 scale = g * is, shift = bt - mu * scale;
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const long long n_out = (long long)s.B * s.Ho * s.Wo * s.Q;
  const long long stride = (long long)gridDim.x * kBnThreads;
  for (long long o = (long long)blockIdx.x * kBnThreads + threadIdx.x; o < n_out; o += stride) {
    const long long orow = o / s.Q;
    const int wo = (int)(orow % s.Wo);
    const long long t = orow / s.Wo;
    const int ho = (int)(t % s.Ho);
    const long long b = t / s.Ho;
    int arg[4];
    f4 vraw;
    const f4 best = window_max(y, s, b, ho, wo, q, scale, shift, arg, &vraw);
    const f4 d = dz[o];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (best[e] > 0.f) {                            // ReLU passes the gradient at the arg-max
        const float xh = (vraw[e] - mu[e]) * is[e];   // xhat at the arg-max
        s1[e] += d[e];
        s2[e] = fmaf(d[e], xh, s2[e]);
      }
    }
  }
  float* p = partial + (size_t)blockIdx.x * 2 * s.C;
  quad_reduce_store(s1, s2, s.Q, lds, p, p + s.C);
}

__global__ __launch_bounds__(kFinCh * 16) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int nblk, int C, double n_rows, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ coef, float* __restrict__ dzero) {
  __shared__ double lds[2][kFinCh][16];
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, lane = threadIdx.x / kFinCh;
  double s1, s2;
  partial_sums(partial, nblk, C, c, lane, lds, &s1, &s2);
  if (c >= C || lane != 0) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  if (dzero) dzero[c] = 0.f;               // the exact zero gradient of a constant the norm cancels
  coef[c] = (float)(s1 / n_rows);          // mean of dy
  coef[C + c] = (float)(s2 / n_rows);      // mean of dy * xhat
}

// dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)), dy = dz at the window's arg-max
// where the activation is positive, 0 elsewhere.  One thread per (window, channel quad): every y
// of the window is read once and every dx written once.  Positions that no window covers (odd
// lengths: H % ph rows, W % pw columns) have dy = 0 and are written by the tail loop.
__global__ __launch_bounds__(kBnThreads) void bnrp_bwd_apply_kernel(
    const f4* __restrict__ y, const f4* __restrict__ dz, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ coef, f4* __restrict__ dx,
    BnShape s) {
  const int q = threadIdx.x % s.Q;
  const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * q), bt = *reinterpret_cast<const f4*>(beta + 4 * q);
  const f4 mu = *reinterpret_cast<const f4*>(mean + 4 * q), is = *reinterpret_cast<const f4*>(invstd + 4 * q);
  const f4 c1 = *reinterpret_cast<const f4*>(coef + 4 * q), c2 = *reinterpret_cast<const f4*>(coef + s.C + 4 * q);
  const f4 scale = g * is, shift = bt - mu * scale;
  const long long stride = (long long)gridDim.x * kBnThreads;
  const long long n_out = (long long)s.B * s.Ho * s.Wo * s.Q;
  for (long long o = (long long)blockIdx.x * kBnThreads + threadIdx.x; o < n_out; o += stride) {
    const long long orow = o / s.Q;
    const int wo = (int)(orow % s.Wo);
    const long long t = orow / s.Wo;
    const int ho = (int)(t % s.Ho);
    const long long b = t / s.Ho;
    // pass 1 over the window: arg-max of relu(a) (first maximum)
    f4 best = {-1.f, -1.f, -1.f, -1.f};
    int arg[4] = {0, 0, 0, 0};
    for (int i = 0; i < s.ph; ++i)
      for (int j = 0; j < s.pw; ++j) {
        const long long row = (b * s.H + (ho * s.ph + i)) * s.W + (wo * s.pw + j);
        const f4 a = f4_fma(y[row * s.Q + q], scale, shift);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = a[e] > 0.f ? a[e] : 0.f;
          if (r > best[e]) {
            best[e] = r;
            arg[e] = i * s.pw + j;
          }
        }
      }
    const f4 d = dz[o];
    // pass 2 (the window is in cache): dx for every position
    for (int i = 0; i < s.ph; ++i)
      for (int j = 0; j < s.pw; ++j) {
        const long long row = (b * s.H + (ho * s.ph + i)) * s.W + (wo * s.pw + j);
        const f4 v = y[row * s.Q + q];
        f4 dy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (best[e] > 0.f && arg[e] == i * s.pw + j) dy[e] = d[e];
        const f4 xh = (v - mu) * is;
        dx[row * s.Q + q] = scale * (dy - c1 - xh * c2);
      }
  }
  // uncovered positions: columns w >= Wo*pw of every row, then rows h >= Ho*ph of the covered columns
  const int wc = s.Wo * s.pw, hc = s.Ho * s.ph;
  const long long n_col = (long long)s.B * s.H * (s.W - wc) * s.Q;
  for (long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x; i < n_col; i += stride) {
    const long long r = i / s.Q;
    const int w = wc + (int)(r % (s.W - wc));
    const long long bh = r / (s.W - wc);                           // b * H + h
    const long long e = (bh * s.W + w) * s.Q + q;
    const f4 xh = (y[e] - mu) * is;
    dx[e] = scale * (-c1 - xh * c2);
  }
  const long long n_row = (long long)s.B * (s.H - hc) * wc * s.Q;
  for (long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x; i < n_row; i += stride) {
    const long long r = i / s.Q;
    const int w = (int)(r % wc);
    const long long t = r / wc;
    const int h = hc + (int)(t % (s.H - hc));
    const long long b = t / (s.H - hc);
    const long long e = ((b * s.H + h) * s.W + w) * s.Q + q;
    const f4 xh = (y[e] - mu) * is;
    dx[e] = scale * (-c1 - xh * c2);
  }
}

// ------------------------------------------------------------------------------------ fast paths
// The three streaming kernels above, specialised for the window shapes ResNet9 uses ((1,1), (1,2),
// (2,2)) and index ranges that fit 32 bits.  The generic versions walk the window in a loop with
// runtime bounds: one global load, one s_waitcnt vmcnt(0), the next load — and the backward
// reduction then waits once more for dz.  A forward pass hides that behind its store (fire and
// forget), a pure reader does not: bnrp_bwd_reduce_kernel ran at 2.4-2.8 TB/s where the forward
// apply reaches 6.0 (profiles/r2_resnet1d_step_kernels.csv).  Here every thread handles TWO pooled
// rows per iteration and issues all of their loads (2 x PH x PW of y, 2 of dz) before the first
// use; row -> (b, ho, wo) is 32-bit arithmetic (the generic 64-bit divisions are ~100 instructions
// each).
template <int PH, int PW>
__device__ __forceinline__ unsigned window_row(const BnShape& s, unsigned r) {
  const unsigned wo = r % (unsigned)s.Wo, t = r / (unsigned)s.Wo;
  const unsigned ho = t % (unsigned)s.Ho, b = t / (unsigned)s.Ho;
  return (b * (unsigned)s.H + ho * PH) * (unsigned)s.W + wo * PW;      // first input row of the window
}

template <int PH, int PW>
__device__ __forceinline__ void window_load(const f4* __restrict__ y, const BnShape& s, unsigned row0,
                                            int q, f4 (&v)[PH * PW]) {
#pragma unroll
  for (int i = 0; i < PH; ++i)
#pragma unroll
    for (int j = 0; j < PW; ++j)
      v[i * PW + j] = y[(size_t)(row0 + (unsigned)i * (unsigned)s.W + (unsigned)j) * s.Q + q];
}

// window_max on registers: maximum of relu(scale * y + shift), index of its FIRST occurrence, and
// y itself there.
template <int N>
__device__ __forceinline__ f4 window_best(const f4 (&v)[N], f4 scale, f4 shift, int (&arg)[4], f4* raw) {
  f4 best = {-1.f, -1.f, -1.f, -1.f}, vb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const f4 a = f4_fma(v[k], scale, shift);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float r = a[e] > 0.f ? a[e] : 0.f;
      if (r > best[e]) {
        best[e] = r;
        arg[e] = k;
        vb[e] = v[k][e];
      }
    }
  }
  *raw = vb;
  return best;
}

template <int PH, int PW>
__global__ __launch_bounds__(kBnThreads) void bnrp_apply_win_kernel(
    const f4* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ invstd, const f4* __restrict__ skip,
    f4* __restrict__ z, BnShape s) {
  const int q = threadIdx.x % s.Q;
  const unsigned R = kBnThreads / s.Q, rloc = threadIdx.x / s.Q;
  const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * q), bt = *reinterpret_cast<const f4*>(beta + 4 * q);
  const f4 mu = *reinterpret_cast<const f4*>(mean + 4 * q), is = *reinterpret_cast<const f4*>(invstd + 4 * q);
  const f4 scale = g * is, shift = bt - mu * scale;
  const unsigned n_rows = (unsigned)s.B * s.Ho * s.Wo, stride = gridDim.x * R;
  for (unsigned r = blockIdx.x * R + rloc; r < n_rows; r += 2 * stride) {
    const bool two = r + stride < n_rows;
    const unsigned r2 = two ? r + stride : r;
    f4 va[PH * PW], vb[PH * PW];
    window_load<PH, PW>(y, s, window_row<PH, PW>(s, r), q, va);
    window_load<PH, PW>(y, s, window_row<PH, PW>(s, r2), q, vb);
    f4 ka = {0.f, 0.f, 0.f, 0.f}, kb = ka;
    if (skip) {
      ka = skip[(size_t)r * s.Q + q];
      kb = skip[(size_t)r2 * s.Q + q];
    }
    int arg[4];
    f4 raw;
    z[(size_t)r * s.Q + q] = window_best<PH * PW>(va, scale, shift, arg, &raw) + ka;
    if (two) z[(size_t)r2 * s.Q + q] = window_best<PH * PW>(vb, scale, shift, arg, &raw) + kb;
  }
}

template <int PH, int PW>
__global__ __launch_bounds__(kBnThreads) void bnrp_bwd_reduce_win_kernel(
    const f4* __restrict__ y, const f4* __restrict__ dz, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ partial, BnShape s) {
  __shared__ f4 lds[2 * kBnThreads];
  const int q = threadIdx.x % s.Q;
  const unsigned R = kBnThreads / s.Q, rloc = threadIdx.x / s.Q;
  const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * q), bt = *reinterpret_cast<const f4*>(beta + 4 * q);
  const f4 mu = *reinterpret_cast<const f4*>(mean + 4 * q), is = *reinterpret_cast<const f4*>(invstd + 4 * q);
  const f4 scale = g * is, shift = bt - mu * scale;
  f4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const unsigned n_rows = (unsigned)s.B * s.Ho * s.Wo, stride = gridDim.x * R;
  for (unsigned r = blockIdx.x * R + rloc; r < n_rows; r += 2 * stride) {
    const bool two = r + stride < n_rows;
    const unsigned r2 = two ? r + stride : r;
    f4 va[PH * PW], vb[PH * PW];
    window_load<PH, PW>(y, s, window_row<PH, PW>(s, r), q, va);
    window_load<PH, PW>(y, s, window_row<PH, PW>(s, r2), q, vb);
    const f4 da = dz[(size_t)r * s.Q + q];
    f4 db = dz[(size_t)r2 * s.Q + q];
    if (!two) db = f4{0.f, 0.f, 0.f, 0.f};
    int arg[4];
    f4 raw;
    f4 best = window_best<PH * PW>(va, scale, shift, arg, &raw);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = best[e] > 0.f ? da[e] : 0.f;    // ReLU passes the gradient at the arg-max
      s1[e] += d;
      s2[e] = fmaf(d, (raw[e] - mu[e]) * is[e], s2[e]);
    }
    best = window_best<PH * PW>(vb, scale, shift, arg, &raw);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = best[e] > 0.f ? db[e] : 0.f;
      s1[e] += d;
      s2[e] = fmaf(d, (raw[e] - mu[e]) * is[e], s2[e]);
    }
  }
  float* p = partial + (size_t)blockIdx.x * 2 * s.C;
  quad_reduce_store(s1, s2, s.Q, lds, p, p + s.C);
}

template <int PH, int PW>
__global__ __launch_bounds__(kBnThreads) void bnrp_bwd_apply_win_kernel(
    const f4* __restrict__ y, const f4* __restrict__ dz, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ coef, f4* __restrict__ dx,
    BnShape s) {
  const int q = threadIdx.x % s.Q;
  const unsigned R = kBnThreads / s.Q, rloc = threadIdx.x / s.Q;
  const f4 g = *reinterpret_cast<const f4*>(gamma + 4 * q), bt = *reinterpret_cast<const f4*>(beta + 4 * q);
  const f4 mu = *reinterpret_cast<const f4*>(mean + 4 * q), is = *reinterpret_cast<const f4*>(invstd + 4 * q);
  const f4 c1 = *reinterpret_cast<const f4*>(coef + 4 * q), c2 = *reinterpret_cast<const f4*>(coef + s.C + 4 * q);
  const f4 scale = g * is, shift = bt - mu * scale;
  const unsigned n_rows = (unsigned)s.B * s.Ho * s.Wo, stride = gridDim.x * R;
  auto emit = [&](const f4 (&v)[PH * PW], f4 d, unsigned row0) {
    int arg[4] = {0, 0, 0, 0};
    f4 raw;
    const f4 best = window_best<PH * PW>(v, scale, shift, arg, &raw);
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        f4 dy = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (best[e] > 0.f && arg[e] == i * PW + j) dy[e] = d[e];
        const f4 xh = (v[i * PW + j] - mu) * is;
        dx[(size_t)(row0 + (unsigned)i * (unsigned)s.W + (unsigned)j) * s.Q + q] = scale * (dy - c1 - xh * c2);
      }
  };
  for (unsigned r = blockIdx.x * R + rloc; r < n_rows; r += 2 * stride) {
    const bool two = r + stride < n_rows;
    const unsigned r2 = two ? r + stride : r;
    const unsigned rowa = window_row<PH, PW>(s, r), rowb = window_row<PH, PW>(s, r2);
    f4 va[PH * PW], vb[PH * PW];
    window_load<PH, PW>(y, s, rowa, q, va);
    window_load<PH, PW>(y, s, rowb, q, vb);
    const f4 da = dz[(size_t)r * s.Q + q], db = dz[(size_t)r2 * s.Q + q];
    emit(va, da, rowa);
    if (two) emit(vb, db, rowb);
  }
  // uncovered positions (odd lengths), as in the generic kernel
  const long long stride4 = (long long)gridDim.x * kBnThreads;
  const int wc = s.Wo * s.pw, hc = s.Ho * s.ph;
  const long long n_col = (long long)s.B * s.H * (s.W - wc) * s.Q;
  for (long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x; i < n_col; i += stride4) {
    const long long r = i / s.Q;
    const int w = wc + (int)(r % (s.W - wc));
    const long long bh = r / (s.W - wc);
    const long long e = (bh * s.W + w) * s.Q + q;
    const f4 xh = (y[e] - mu) * is;
    dx[e] = scale * (-c1 - xh * c2);
  }
  const long long n_row = (long long)s.B * (s.H - hc) * wc * s.Q;
  for (long long i = (long long)blockIdx.x * kBnThreads + threadIdx.x; i < n_row; i += stride4) {
    const long long r = i / s.Q;
    const int w = (int)(r % wc);
    const long long t = r / wc;
    const int h = hc + (int)(t % (s.H - hc));
    const long long b = t / (s.H - hc);
    const long long e = ((b * s.H + h) * s.W + w) * s.Q + q;
    const f4 xh = (y[e] - mu) * is;
    dx[e] = scale * (-c1 - xh * c2);
  }
}

// 0: generic kernels; 1: (1,1), 2: (1,2), 3: (2,2) windows with 32-bit indexing
inline int bn_fast_kind(const BnShape& s) {
  if ((long long)s.B * s.H * s.W * s.Q >= (1ll << 31)) return 0;
  if (s.ph == 1 && s.pw == 1) return 1;
  if (s.ph == 1 && s.pw == 2) return 2;
  if (s.ph == 2 && s.pw == 2) return 3;
  return 0;
}

inline int bn_blocks(long long n4) {
  long long b = (n4 + (long long)kBnThreads * 8 - 1) / ((long long)kBnThreads * 8);
  if (b < 1) b = 1;
  return (int)(b > kBnMaxBlocks ? kBnMaxBlocks : b);
}

inline bool bn_shape(BnShape* s, int B, int H, int W, int C, int ph, int pw) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || ph <= 0 || pw <= 0 || (C & 3)) return false;
  const int Q = C / 4;
  if (Q > kBnThreads || kBnThreads % Q) return false;
  *s = BnShape{B, H, W, C, ph, pw, H / ph, W / pw, Q};
  return s->Ho > 0 && s->Wo > 0;
}

}  // namespace pcgmix

extern "C" long long pcgmix_bnrp_workspace_floats(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  return (long long)pcgmix::kBnMaxBlocks * 2 * C + 2 * C;
}

extern "C" int pcgmix_bnrp_fwd_f32(const float* y, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum,
                                   float eps, const float* mean_shift, long long* batches_tracked,
                                   const float* skip, float* z, float* mean, float* invstd,
                                   float* workspace, int B, int H, int W, int C, int ph, int pw,
                                   pcgmix_stream_t stream) {
  using namespace pcgmix;
  BnShape s;
  if (!y || !gamma || !beta || !z || !mean || !invstd || !workspace || !bn_shape(&s, B, H, W, C, ph, pw))
    return hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(skip) |
       reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
       reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(invstd)) & 15)
    return hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long rows = (long long)B * H * W, n4 = rows * s.Q;
  const int nblk = bn_blocks(n4);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(kBnThreads), 0, st,
                     reinterpret_cast<const f4*>(y), n4, s.Q, workspace, C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * 16), 0, st, workspace, nblk, C,
                     (double)rows, eps, momentum, mean, invstd, running_mean, running_var, mean_shift,
                     batches_tracked);
  const long long n_out = (long long)B * s.Ho * s.Wo * s.Q;
  const dim3 ag(bn_blocks(n_out * 2)), ab(kBnThreads);
  const f4* y4 = reinterpret_cast<const f4*>(y);
  const f4* k4 = reinterpret_cast<const f4*>(skip);
  f4* z4 = reinterpret_cast<f4*>(z);
  switch (bn_fast_kind(s)) {
    case 1: hipLaunchKernelGGL((bnrp_apply_win_kernel<1, 1>), ag, ab, 0, st, y4, gamma, beta, mean, invstd, k4, z4, s); break;
    case 2: hipLaunchKernelGGL((bnrp_apply_win_kernel<1, 2>), ag, ab, 0, st, y4, gamma, beta, mean, invstd, k4, z4, s); break;
    case 3: hipLaunchKernelGGL((bnrp_apply_win_kernel<2, 2>), ag, ab, 0, st, y4, gamma, beta, mean, invstd, k4, z4, s); break;
    default: hipLaunchKernelGGL(bnrp_apply_kernel, ag, ab, 0, st, y4, gamma, beta, mean, invstd, k4, z4, s);
  }
  return (int)hipGetLastError();
}

extern "C" int pcgmix_bnrp_bwd_f32(const float* y, const float* dz, const float* gamma,
                                   const float* beta, const float* mean, const float* invstd,
                                   float* dx, float* dgamma, float* dbeta, float* dzero,
                                   float* workspace, int B, int H, int W, int C, int ph, int pw,
                                   pcgmix_stream_t stream) {
  using namespace pcgmix;
  BnShape s;
  if (!y || !dz || !gamma || !beta || !mean || !invstd || !dx || !dgamma || !dbeta || !workspace ||
      !bn_shape(&s, B, H, W, C, ph, pw))
    return hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dz) |
       reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(workspace) |
       reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
       reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(invstd)) & 15)
    return hipErrorInvalidValue;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long long rows = (long long)B * H * W;
  const long long n_out = (long long)B * s.Ho * s.Wo * s.Q;
  const int nblk = bn_blocks(n_out * 2);
  float* coef = workspace + (size_t)kBnMaxBlocks * 2 * C;
  const dim3 rg(nblk), rb(kBnThreads);
  const f4* y4 = reinterpret_cast<const f4*>(y);
  const f4* dz4 = reinterpret_cast<const f4*>(dz);
  f4* dx4 = reinterpret_cast<f4*>(dx);
  const int kind = bn_fast_kind(s);
  switch (kind) {
    case 1: hipLaunchKernelGGL((bnrp_bwd_reduce_win_kernel<1, 1>), rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, workspace, s); break;
    case 2: hipLaunchKernelGGL((bnrp_bwd_reduce_win_kernel<1, 2>), rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, workspace, s); break;
    case 3: hipLaunchKernelGGL((bnrp_bwd_reduce_win_kernel<2, 2>), rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, workspace, s); break;
    default: hipLaunchKernelGGL(bnrp_bwd_reduce_kernel, rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, workspace, s);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * 16), 0, st, workspace, nblk,
                     C, (double)rows, dgamma, dbeta, coef, dzero);
  switch (kind) {
    case 1: hipLaunchKernelGGL((bnrp_bwd_apply_win_kernel<1, 1>), rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, coef, dx4, s); break;
    case 2: hipLaunchKernelGGL((bnrp_bwd_apply_win_kernel<1, 2>), rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, coef, dx4, s); break;
    case 3: hipLaunchKernelGGL((bnrp_bwd_apply_win_kernel<2, 2>), rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, coef, dx4, s); break;
    default: hipLaunchKernelGGL(bnrp_bwd_apply_kernel, rg, rb, 0, st, y4, dz4, gamma, beta, mean, invstd, coef, dx4, s);
  }
  return (int)hipGetLastError();
}
