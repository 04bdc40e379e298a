// pcgmix_potes.hip — the Potes 1D-CNN's convolutional branch as two fused kernels (gfx950).
//
// Reference: models.py:359-381 (conv_block_1d, CNN_potes.cnn1) — per band-pass channel
//     Conv1d(1->8, k5, pad1) + ReLU + MaxPool(2)  ->  Conv1d(8->4, k5, pad1) + ReLU + MaxPool(2)
// applied with the SAME weights to all four bands (models.py:444-455), i.e. to N = 4*B rows.
// Through MIOpen this stack costs ~1.8 ms per bs=256 step on MI355X (a naive direct-conv
// fallback, layout transposes, separate ReLU / pooling / pooling-backward kernels;
// profiles/r1_bench_kernel_stats.csv) for ~40 MB of unavoidable HBM traffic.  Here:
//
//   potes_fwd_kernel   reads each input row once, keeps the 8-channel intermediate in LDS,
//                      writes the pooled (N,4,P2) activations: 4*T + 16*P2 bytes per row.
//   potes_bwd_kernel   recomputes the forward tile from the input (cheaper than storing
//                      the 8-channel intermediate: 2.6x the input size), back-propagates through
//                      pool/ReLU/conv2/pool/ReLU and reduces the 212 weight/bias gradients per
//                      block in registers; a second tiny kernel sums the per-block partials
//                      (deterministic, no float atomics).  Reads 4*T + 16*P2 bytes per row.
//
// Small-channel direct convolutions are not GEMM-shaped (K = 5 or 40): this is VALU + LDS work
// bounded by HBM, not MFMA work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kC1 = 8, kC2 = 4, kK = 5;
constexpr int kNW1 = kC1 * kK;        // 40
constexpr int kNW2 = kC2 * kC1 * kK;  // 160
constexpr int kNGrad = kNW1 + kC1 + kNW2 + kC2;  // 212: [gw1 | gb1 | gw2 | gb2]
constexpr int kPotThreads = 256;

struct PotesDims {
  int T, L1, P1, L2, P2;
};
__host__ __device__ inline PotesDims potes_dims(int T) {
  PotesDims d;
  d.T = T;
  d.L1 = T - 2;       // conv k5 pad1
  d.P1 = d.L1 / 2;    // MaxPool1d(2), floor
  d.L2 = d.P1 - 2;
  d.P2 = d.L2 / 2;
  return d;
}

struct PotesWeights {  // LDS copy, broadcast-read
  float w1[kNW1], b1[kC1], w2[kNW2], b2[kC2];
};

__device__ __forceinline__ void load_weights(PotesWeights* W, const float* w1, const float* b1,
                                             const float* w2, const float* b2) {
  for (int i = threadIdx.x; i < kNW1; i += kPotThreads) W->w1[i] = w1[i];
  for (int i = threadIdx.x; i < kC1; i += kPotThreads) W->b1[i] = b1[i];
  for (int i = threadIdx.x; i < kNW2; i += kPotThreads) W->w2[i] = w2[i];
  for (int i = threadIdx.x; i < kC2; i += kPotThreads) W->b2[i] = b2[i];
}

// Stage x[xlo .. xlo+nx) of one row into LDS, zero outside [0, T).
__device__ __forceinline__ void stage_x(float* xs, const float* __restrict__ xrow, int xlo, int nx,
                                        int T) {
  for (int u = threadIdx.x; u < nx; u += kPotThreads) {
    const int g = xlo + u;
    xs[u] = (g >= 0 && g < T) ? xrow[g] : 0.f;
  }
}

// Layer 1 for pooled positions q = qlo + qq, qq in [0, nq): a1s[ci][qq] (0 outside [0,P1) —
// that IS conv2's zero padding) and, if sel != nullptr, which conv output won the pool and
// survived the ReLU: 0 none, 1 first (i = 2q), 2 second (i = 2q+1).
__device__ __forceinline__ void layer1(const PotesWeights& W, const float* xs, float* a1s,
                                       uint8_t* sel, int qlo, int nq, int P1) {
  for (int idx = threadIdx.x; idx < kC1 * nq; idx += kPotThreads) {
    const int ci = idx / nq, qq = idx - ci * nq;
    const int q = qlo + qq;
    float a = 0.f;
    uint8_t s = 0;
    if (q >= 0 && q < P1) {
      float za = W.b1[ci], zb = W.b1[ci];
#pragma unroll
      for (int k = 0; k < kK; ++k) {
        za = fmaf(W.w1[ci * kK + k], xs[2 * qq + k], za);
        zb = fmaf(W.w1[ci * kK + k], xs[2 * qq + 1 + k], zb);
      }
      const float ra = fmaxf(za, 0.f), rb = fmaxf(zb, 0.f);
      // torch's max-pool keeps the FIRST maximum (strict '>' scan)
      if (rb > ra) { a = rb; s = 2; } else { a = ra; s = ra > 0.f ? 1 : 0; }
    }
    a1s[idx] = a;
    if (sel) sel[idx] = s;
  }
}

// ---------------------------------------------------------------------------------- forward
constexpr int kFwdTP = 256;                     // pooled outputs per block
constexpr int kFwdNQ = 2 * kFwdTP + 4;
constexpr int kFwdNX = 4 * kFwdTP + 12;

__global__ __launch_bounds__(kPotThreads) void potes_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ h2, int N,
    int T) {
  __shared__ PotesWeights W;
  __shared__ float xs[kFwdNX];
  __shared__ float a1s[kC1 * kFwdNQ];
  const PotesDims d = potes_dims(T);
  const int n = blockIdx.y, p0 = blockIdx.x * kFwdTP;
  const int qlo = 2 * p0 - 1, xlo = 2 * qlo - 1;
  load_weights(&W, w1, b1, w2, b2);
  stage_x(xs, x + (size_t)n * T, xlo, kFwdNX, T);
  __syncthreads();
  layer1(W, xs, a1s, nullptr, qlo, kFwdNQ, d.P1);
  __syncthreads();
  const int p = p0 + threadIdx.x;
  if (p < d.P2) {
#pragma unroll
    for (int co = 0; co < kC2; ++co) {
      float za = W.b2[co], zb = W.b2[co];
#pragma unroll
      for (int ci = 0; ci < kC1; ++ci)
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          const float w = W.w2[(co * kC1 + ci) * kK + k];
          za = fmaf(w, a1s[ci * kFwdNQ + 2 * threadIdx.x + k], za);
          zb = fmaf(w, a1s[ci * kFwdNQ + 2 * threadIdx.x + 1 + k], zb);
        }
      h2[((size_t)n * kC2 + co) * d.P2 + p] = fmaxf(fmaxf(za, 0.f), fmaxf(zb, 0.f));
    }
  }
}

// ---------------------------------------------------------------------------------- backward
constexpr int kBwdTP = 128;                     // OWNED pooled outputs per work item
constexpr int kBwdNP = kBwdTP + 3;              // extended: p0-2 .. p0+TP
constexpr int kBwdNJ = 2 * kBwdTP + 6;          // conv2 outputs j: 2p0-4 .. 2p0+2TP+1
constexpr int kBwdNQ = 2 * kBwdTP + 10;         // a1 positions:   2p0-5 .. 2p0+2TP+4
constexpr int kBwdNX = 4 * kBwdTP + 24;         // x positions:    4p0-11 .. 4p0+4TP+12
constexpr int kBwdNI = 4 * kBwdTP;              // OWNED conv1 outputs i: 4p0 .. 4p0+4TP-1

__global__ __launch_bounds__(kPotThreads) void potes_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gh2, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ partial /* gridDim.x * 212 */, int N, int T) {
  __shared__ PotesWeights W;
  __shared__ float xs[kBwdNX];
  __shared__ float a1s[kC1 * kBwdNQ];
  __shared__ uint8_t sel1[kC1 * kBwdNQ];
  __shared__ float dz2s[kC2 * kBwdNJ];
  __shared__ float dz1s[kC1 * kBwdNI];
  __shared__ float red[4 * kNGrad];
  const PotesDims d = potes_dims(T);
  // +2: the a1 positions 2*P2 .. 2*P2+2 still receive gradient from the last pooled outputs
  const int tiles = (d.P2 + 2 + kBwdTP - 1) / kBwdTP;
  const long long work = (long long)N * tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_weights(&W, w1, b1, w2, b2);

  // gradient entry e in [0,212) is owned, for the whole kernel, by lane (e & 63) slot (e >> 6)
  // of EVERY wave; wave w sums the w-th quarter of each tile's positions.
  float acc[4] = {0.f, 0.f, 0.f, 0.f};

  for (long long item = blockIdx.x; item < work; item += gridDim.x) {
    const int n = (int)(item / tiles), p0 = (int)(item - (long long)n * tiles) * kBwdTP;
    const int qlo = 2 * p0 - 5, xlo = 2 * qlo - 1;
    __syncthreads();  // previous item's LDS fully consumed
    stage_x(xs, x + (size_t)n * T, xlo, kBwdNX, T);
    __syncthreads();
    layer1(W, xs, a1s, sel1, qlo, kBwdNQ, d.P1);
    __syncthreads();
    // conv2 + ReLU + pool on the extended range, straight to dz2 = dL/dz2
    for (int idx = threadIdx.x; idx < kC2 * kBwdNP; idx += kPotThreads) {
      const int co = idx / kBwdNP, pp = idx - co * kBwdNP;
      const int pe = p0 - 2 + pp;
      float da = 0.f, db = 0.f;
      if (pe >= 0 && pe < d.P2) {
        float za = W.b2[co], zb = W.b2[co];
#pragma unroll
        for (int ci = 0; ci < kC1; ++ci)
#pragma unroll
          for (int k = 0; k < kK; ++k) {
            const float w = W.w2[(co * kC1 + ci) * kK + k];
            za = fmaf(w, a1s[ci * kBwdNQ + 2 * pp + k], za);
            zb = fmaf(w, a1s[ci * kBwdNQ + 2 * pp + 1 + k], zb);
          }
        const float g = gh2[((size_t)n * kC2 + co) * d.P2 + pe];
        const float ra = fmaxf(za, 0.f), rb = fmaxf(zb, 0.f);
        if (rb > ra) db = g; else if (ra > 0.f) da = g;
      }
      dz2s[co * kBwdNJ + 2 * pp] = da;
      dz2s[co * kBwdNJ + 2 * pp + 1] = db;
    }
    __syncthreads();
    // back through conv2 to the owned a1 positions q = 2p0 + r, then pool1/ReLU1 -> dz1
    for (int idx = threadIdx.x; idx < kC1 * 2 * kBwdTP; idx += kPotThreads) {
      const int ci = idx / (2 * kBwdTP), r = idx - ci * (2 * kBwdTP);
      float da1 = 0.f;
#pragma unroll
      for (int co = 0; co < kC2; ++co)
#pragma unroll
        for (int k = 0; k < kK; ++k)
          da1 = fmaf(dz2s[co * kBwdNJ + r + 5 - k], W.w2[(co * kC1 + ci) * kK + k], da1);
      const uint8_t s = sel1[ci * kBwdNQ + r + 5];   // 0 for q outside [0,P1)
      dz1s[ci * kBwdNI + 2 * r] = s == 1 ? da1 : 0.f;
      dz1s[ci * kBwdNI + 2 * r + 1] = s == 2 ? da1 : 0.f;
    }
    __syncthreads();
    // the 212 reductions; each wave takes a quarter of the owned positions
    const int i_lo = wave * (kBwdNI / 4), i_hi = i_lo + kBwdNI / 4;          // conv1 outputs
    const int s_lo = wave * (2 * kBwdTP / 4), s_hi = s_lo + 2 * kBwdTP / 4;  // conv2 outputs
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
      const int e = slot * 64 + lane;
      if (e >= kNGrad) continue;
      float a = 0.f;
      if (e < kNW1) {                                  // gw1[ci][k] += dz1[ci][i] * x[i-1+k]
        const int ci = e / kK, k = e - ci * kK;
        for (int ii = i_lo; ii < i_hi; ++ii) a = fmaf(dz1s[ci * kBwdNI + ii], xs[ii + 10 + k], a);
      } else if (e < kNW1 + kC1) {                     // gb1[ci]
        const int ci = e - kNW1;
        for (int ii = i_lo; ii < i_hi; ++ii) a += dz1s[ci * kBwdNI + ii];
      } else if (e < kNW1 + kC1 + kNW2) {              // gw2[co][ci][k] += dz2[co][j] * a1[ci][j-1+k]
        const int f = e - kNW1 - kC1;
        const int co = f / (kC1 * kK), ci = (f / kK) % kC1, k = f % kK;
        for (int s = s_lo; s < s_hi; ++s)
          a = fmaf(dz2s[co * kBwdNJ + s + 4], a1s[ci * kBwdNQ + s + 4 + k], a);
      } else {                                         // gb2[co]
        const int co = e - kNW1 - kC1 - kNW2;
        for (int s = s_lo; s < s_hi; ++s) a += dz2s[co * kBwdNJ + s + 4];
      }
      acc[slot] += a;
    }
  }
  // combine the four waves' quarters, one partial vector per block
  __syncthreads();
#pragma unroll
  for (int slot = 0; slot < 4; ++slot) {
    const int e = slot * 64 + lane;
    if (e < kNGrad) red[wave * kNGrad + e] = acc[slot];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kNGrad; e += kPotThreads)
    partial[(size_t)blockIdx.x * kNGrad + e] =
        (red[e] + red[kNGrad + e]) + (red[2 * kNGrad + e] + red[3 * kNGrad + e]);
}

// Sum the per-block partial vectors in a fixed order: grads[e] = sum_g partial[g][e].
__global__ __launch_bounds__(kPotThreads) void potes_reduce_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ grads,
                                                                   int G) {
  __shared__ float red[kPotThreads];
  const int e = blockIdx.x;
  float a = 0.f;
  for (int g = threadIdx.x; g < G; g += kPotThreads) a += partial[(size_t)g * kNGrad + e];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = kPotThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) grads[e] = red[0];
}

}  // namespace pcgmix

extern "C" int pcgmix_potes_out_len(int T) {
  return T < 14 ? 0 : pcgmix::potes_dims(T).P2;
}

extern "C" int pcgmix_potes_bwd_blocks(int N, int T) {
  if (N <= 0 || T < 14) return 0;
  const pcgmix::PotesDims d = pcgmix::potes_dims(T);
  const long long work = (long long)N * ((d.P2 + 2 + pcgmix::kBwdTP - 1) / pcgmix::kBwdTP);
  return (int)(work < 1024 ? work : 1024);  // 4 persistent blocks per CU
}

extern "C" int pcgmix_potes_stack_fwd_f32(const float* x, const float* w1, const float* b1,
                                          const float* w2, const float* b2, float* h2, int N, int T,
                                          pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !w1 || !b1 || !w2 || !b2 || !h2 || N < 0 || T < 14 || N > 65535 * 1)
    return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  const PotesDims d = potes_dims(T);
  dim3 grid((unsigned)((d.P2 + kFwdTP - 1) / kFwdTP), (unsigned)N), block(kPotThreads);
  hipLaunchKernelGGL(potes_fwd_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), x, w1,
                     b1, w2, b2, h2, N, T);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_stack_bwd_f32(const float* x, const float* grad_h2, const float* w1,
                                          const float* b1, const float* w2, const float* b2,
                                          float* partial, float* grads, int N, int T,
                                          pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !grad_h2 || !w1 || !b1 || !w2 || !b2 || !partial || !grads || N <= 0 || T < 14)
    return hipErrorInvalidValue;
  const int G = pcgmix_potes_bwd_blocks(N, T);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(potes_bwd_kernel, dim3((unsigned)G), dim3(kPotThreads), 0, s, x, grad_h2, w1,
                     b1, w2, b2, partial, N, T);
  hipLaunchKernelGGL(potes_reduce_kernel, dim3(kNGrad), dim3(kPotThreads), 0, s, partial, grads, G);
  return (int)hipGetLastError();
}
