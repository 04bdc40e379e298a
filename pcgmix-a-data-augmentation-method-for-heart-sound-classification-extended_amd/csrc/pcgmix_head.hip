// pcgmix_head.hip — the Potes classifier head and the soft-target cross entropy as a handful of
// kernels (gfx950).
//
// Reference: models.py:376-381, 456-465 (CNN_potes: Dropout(.25) -> Flatten/concat -> dimreduc
// Linear(K->20) -> ReLU -> Dropout(.5) -> Linear(20->C)) and train_model.py:45-54 (CELoss:
// mean_b(-sum_c log_softmax(logits)[b,c] * target[b,c])).
//
// Through torch this part of the bs=256 train step is ~30 launches of 3-5 us each (ReLU, two
// dropouts, a 20->2 GEMM, log-softmax, mul, two reductions, two negations, their backward twins,
// three more GEMMs, bias reductions, masked scales) for a few KB of data, plus two hipBLASLt GEMMs
// and a masked scale over the 20 MB feature matrix: ~120 us of a ~300 us step.  Here:
//
//   potes_tail_fwd_kernel   split-K partials of dimreduc -> +bias -> ReLU -> dropout mask ->
//                           Linear(20->C): z (kept for backward) and logits
//   soft_ce_fwd_kernel      loss scalar, one block
//   soft_ce_bwd_kernel      dlogits
//   potes_tail_bwd_kernel   dlogits -> dW2, db2, dz (through dropout and ReLU), db1; one block
//   potes_head_bwd_kernel   dz -> dW1 = dz^T x and dx = mask1 * scale1 * (dz W1) in ONE pass over
//                           the feature matrix x (read once, written once: 9 bytes/element)
//
// Dropout: the caller hands over bytes (same shape as what they mask) and a threshold; an element
// is kept iff its byte >= thr.  thr = 1 reads a 0/1 mask; thr = 256*p reads uniformly random
// bytes from torch's generator (one `random_()` call for both masks: seeding behaves as with
// nn.Dropout; p = 0.25 and 0.5 are exact in 1/256ths).  The feature dropout is applied where the
// features are read (split-K forward, fused backward): no separate pass over the 20 MB matrix.
// All reductions run in a fixed order (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kHeadO = 20;        // dimreduc width (models.py:376)
constexpr int kHeadMaxC = 8;      // classes
constexpr int kTailRows = 4;      // batch rows per block in the forward tail (x 20 x 4 lanes)
constexpr int kTailBwdGroups = 48;
constexpr int kHbCols = 64;

typedef float f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------ tail, forward
// Four lanes (q = t & 3) share one z[row][o]: lane q sums the K-split partials q, q+4, ... with
// up to four loads in flight, then the four sub-sums are added in a fixed order by shuffles.
// The first kTailRows*C threads then form the logits.
__global__ __launch_bounds__(kTailRows* kHeadO * 4) void potes_tail_fwd_kernel(
    const float* __restrict__ partial, int KS, const float* __restrict__ b1,
    const uint8_t* __restrict__ mask2, float scale2, int thr2, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ z, float* __restrict__ logits, int B, int C) {
  __shared__ float h[kTailRows][kHeadO];
  const int t = threadIdx.x, q = t & 3, e = t >> 2;            // e = r * kHeadO + o
  const int r = e / kHeadO, o = e - r * kHeadO;
  const int row = blockIdx.x * kTailRows + r;
  const size_t i = (size_t)(row < B ? row : 0) * kHeadO + o;
  const size_t plane = (size_t)B * kHeadO;
  float v = 0.f;
  int ks = q;
  for (; ks + 12 < KS; ks += 16) {                  // four independent loads in flight
    const float t0 = partial[(size_t)ks * plane + i], t1 = partial[(size_t)(ks + 4) * plane + i],
                t2 = partial[(size_t)(ks + 8) * plane + i], t3 = partial[(size_t)(ks + 12) * plane + i];
    v += t0; v += t1; v += t2; v += t3;
  }
  for (; ks < KS; ks += 4) v += partial[(size_t)ks * plane + i];
  v += __shfl_xor(v, 1, 64);                        // (q0 + q1), (q2 + q3): same value in both lanes
  v += __shfl_xor(v, 2, 64);
  float hv = 0.f;
  if (row < B && q == 0) {
    v += b1 ? b1[o] : 0.f;
    z[i] = v;
    hv = v > 0.f ? v : 0.f;
    if (mask2) hv = (int)mask2[i] >= thr2 ? hv * scale2 : 0.f;
  }
  if (q == 0) h[r][o] = hv;
  __syncthreads();
  if (t < kTailRows * C) {
    const int rr = t / C, c = t - rr * C;
    const int row2 = blockIdx.x * kTailRows + rr;
    if (row2 < B) {
      float a = b2 ? b2[c] : 0.f;
#pragma unroll
      for (int k = 0; k < kHeadO; ++k) a = fmaf(h[rr][k], w2[c * kHeadO + k], a);
      logits[(size_t)row2 * C + c] = a;
    }
  }
}

// ------------------------------------------------------------------------------ tail, saliency
// Frozen model in eval mode, gradient of a class score w.r.t. the input (saliency.py:52-61): the
// logits themselves are not needed, only dz = (z > 0) * (seed W2) with seed = d score / d logits
// (one-hot of the label).  One kernel instead of potes_tail_fwd_kernel + potes_tail_bwd_kernel;
// same summation order of the partials as the forward tail, same fmaf chain over the classes as
// the backward tail.
__global__ __launch_bounds__(kTailRows* kHeadO * 4) void potes_tail_saliency_kernel(
    const float* __restrict__ partial, int KS, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ seed, float* __restrict__ dz, int B,
    int C) {
  const int t = threadIdx.x, q = t & 3, e = t >> 2;            // e = r * kHeadO + o
  const int r = e / kHeadO, o = e - r * kHeadO;
  const int row = blockIdx.x * kTailRows + r;
  const size_t i = (size_t)(row < B ? row : 0) * kHeadO + o;
  const size_t plane = (size_t)B * kHeadO;
  float v = 0.f;
  int ks = q;
  for (; ks + 12 < KS; ks += 16) {
    const float t0 = partial[(size_t)ks * plane + i], t1 = partial[(size_t)(ks + 4) * plane + i],
                t2 = partial[(size_t)(ks + 8) * plane + i], t3 = partial[(size_t)(ks + 12) * plane + i];
    v += t0; v += t1; v += t2; v += t3;
  }
  for (; ks < KS; ks += 4) v += partial[(size_t)ks * plane + i];
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  if (row < B && q == 0) {
    v += b1 ? b1[o] : 0.f;
    const float fac = v > 0.f ? 1.f : 0.f;
    float sacc = 0.f;
    for (int c = 0; c < C; ++c) sacc = fmaf(seed[(size_t)row * C + c], w2[c * kHeadO + o], sacc);
    dz[i] = sacc * fac;
  }
}

// ------------------------------------------------------------------------------ tail, backward
// Block 0 (the others only zero dW1's buffer for potes_head_bwd_kernel).  Thread (g, o): rows g, g+48, ... of column o.  dz = (z > 0) * m2 * (dlogits W2);
// dW2[c][o] = sum_b dlogits[b][c] h[b][o]; db1[o] = sum_b dz[b][o]; db2[c] = sum_b dlogits[b][c].
__global__ __launch_bounds__(kTailBwdGroups* kHeadO) void potes_tail_bwd_kernel(
    const float* __restrict__ dlogits, const float* __restrict__ z,
    const uint8_t* __restrict__ mask2, float scale2, int thr2, const float* __restrict__ w2,
    float* __restrict__ dz, float* __restrict__ dw2, float* __restrict__ db2,
    float* __restrict__ db1, int B, int C, float* __restrict__ zero, long long n_zero) {
  if (blockIdx.x > 0) {        // blocks 1.. clear the buffer the next kernel accumulates dW1 into
    const long long per = (n_zero + gridDim.x - 2) / (gridDim.x - 1);
    const long long lo = (long long)(blockIdx.x - 1) * per;
    const long long hi = lo + per < n_zero ? lo + per : n_zero;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) zero[i] = 0.f;
    return;
  }
  __shared__ float red[kTailBwdGroups][kHeadMaxC + 1][kHeadO];
  __shared__ float redc[kTailBwdGroups][kHeadMaxC];
  const int t = threadIdx.x, g = t / kHeadO, o = t - g * kHeadO;
  float w2c[kHeadMaxC], adw[kHeadMaxC], adb2[kHeadMaxC];
#pragma unroll
  for (int c = 0; c < kHeadMaxC; ++c) {
    w2c[c] = c < C ? w2[c * kHeadO + o] : 0.f;
    adw[c] = 0.f;
    adb2[c] = 0.f;
  }
  float adb1 = 0.f;
  for (int b = g; b < B; b += kTailBwdGroups) {
    const size_t i = (size_t)b * kHeadO + o;
    const float zz = z[i];
    float fac = zz > 0.f ? 1.f : 0.f;
    if (mask2) fac = (int)mask2[i] >= thr2 ? fac * scale2 : 0.f;
    const float hv = zz * fac;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < kHeadMaxC; ++c) {
      if (c < C) {
        const float dl = dlogits[(size_t)b * C + c];
        s = fmaf(dl, w2c[c], s);
        adw[c] = fmaf(dl, hv, adw[c]);
        adb2[c] += dl;
      }
    }
    const float d = s * fac;
    dz[i] = d;
    adb1 += d;
  }
#pragma unroll
  for (int c = 0; c < kHeadMaxC; ++c) red[g][c][o] = adw[c];
  red[g][kHeadMaxC][o] = adb1;
  if (o == 0) {
#pragma unroll
    for (int c = 0; c < kHeadMaxC; ++c) redc[g][c] = adb2[c];
  }
  __syncthreads();
  if (t < (C + 1) * kHeadO) {                      // c == C: db1
    const int c = t / kHeadO, oo = t - c * kHeadO;
    const int slot = c < C ? c : kHeadMaxC;
    float a = 0.f;
    for (int gg = 0; gg < kTailBwdGroups; ++gg) a += red[gg][slot][oo];
    if (c < C) dw2[c * kHeadO + oo] = a;
    else if (db1) db1[oo] = a;
  } else if (t >= 512 && t < 512 + C && db2) {
    const int c = t - 512;
    float a = 0.f;
    for (int gg = 0; gg < kTailBwdGroups; ++gg) a += redc[gg][c];
    db2[c] = a;
  }
}

// ------------------------------------------------------------------------------ head, backward
// Block = 64 feature columns x one HALF of the batch rows (gridDim.y = 2).  Lane = column, wave w
// takes rows w, w+4, ... of each 64-row chunk; W1's column (20 values) and the 20 dW1 accumulators
// live in registers, dz rows are LDS broadcasts.  x is the matrix the forward multiplied (after
// dropout); mask1/scale1 carry the Dropout(.25) backward to dx.
//
// Measured on MI355X (profiles/probes/head_bwd_variants.hip, B=256, K=19968): the kernel is bound
// by its 45 MB of traffic plus latency, not by the 40 FMAs per element (packing them or feeding dz
// from SGPRs changes nothing).  One 512-thread block per column tile holding all rows runs load,
// compute and store phases in lock-step over the whole GPU (21 us); several small co-resident
// blocks per CU overlap them (12.5 us).  The two row halves add their dW1 tiles with float
// atomicAdd onto a zeroed buffer — two addends per element, so the sum does not depend on arrival
// order (deterministic); more than two splits would.
constexpr int kHbWaves = 4, kHbRows = 64, kHbSplit = 2;

// Layout of the fused tail+loss kernel's per-row-block contributions (see potes_tail_loss_kernel):
constexpr int kTlStride = 192;   // floats per row block: dW2[c*20+o] at 0..159, db2 at 160..167,
                                 // db1 at 168..187, loss at 188
constexpr int kTlSeg = 4;        // the row blocks are summed in four segments, then combined in order

// Column t of the contributions of segment sg, summed in a FIXED order with eight loads in flight.
__device__ __forceinline__ float tl_segment_sum(const float* __restrict__ ws, int nrb, int t, int sg) {
  const int per = (nrb + kTlSeg - 1) / kTlSeg;
  const int g0 = sg * per, g1 = g0 + per < nrb ? g0 + per : nrb;
  float a = 0.f;
  int g = g0;
  for (; g + 8 <= g1; g += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ws[(size_t)(g + j) * kTlStride + t];
#pragma unroll
    for (int j = 0; j < 8; ++j) a += v[j];
  }
  for (; g < g1; ++g) a += ws[(size_t)g * kTlStride + t];
  return a;
}

// Where column t of the summed contributions goes: index into small = [dW2 (C x 20) | db2 (C) |
// db1 (20)], -1 for the loss column, -2 for nothing.
__device__ __forceinline__ int tl_small_index(int t, int C) {
  if (t < kHeadMaxC * kHeadO) {
    const int c = t / kHeadO, o = t - c * kHeadO;
    return c < C ? c * kHeadO + o : -2;
  }
  if (t < kHeadMaxC * kHeadO + kHeadMaxC) {
    const int c = t - kHeadMaxC * kHeadO;
    return c < C ? C * kHeadO + c : -2;
  }
  if (t < kHeadMaxC * kHeadO + kHeadMaxC + kHeadO) return C * kHeadO + C + (t - kHeadMaxC * kHeadO - kHeadMaxC);
  return t == kHeadMaxC * kHeadO + kHeadMaxC + kHeadO ? -1 : -2;
}

// NEED_DW = false: the weights are frozen (saliency model): only dx is produced — no read of x,
// no dW1 accumulators, no reduction, no atomics.
template <bool MASKED, bool NEED_DW>
__global__ __launch_bounds__(kHbWaves * 64) void potes_head_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const uint8_t* __restrict__ mask1,
    float scale1, int thr1, int bits1, const float* __restrict__ w1, float* __restrict__ dw1,
    float* __restrict__ dx, int B, int K, const float* __restrict__ gscale,
    const float* __restrict__ small_in, float* __restrict__ small_out, int n_small,
    const float* __restrict__ tl_ws, int tl_nrb, float* __restrict__ tl_loss, int tl_C) {
  __shared__ __align__(16) float dzl[kHbRows * kHeadO];
  // gscale (device scalar, may be NULL = 1): the gradient that arrived at the loss — dz was formed
  // for d loss = 1 by the fused tail+loss kernel; everything downstream is linear in it.  Block
  // (0,0) also scales the small gradients (dW2, db2, db1) that kernel produced.
  const float gs = gscale ? gscale[0] : 1.f;
  __shared__ float red[kHbWaves][kHeadO][kHbCols];
  if (small_out && blockIdx.x == 0 && blockIdx.y == 0) {
    if (tl_ws) {
      // The fused tail+loss forward left its finalize step to this block (captured training step:
      // one launch less per replay): column sums of the per-row-block contributions in
      // potes_tail_loss_finalize_kernel's order, the loss, and the small gradients times gs.
      float* seg = &red[0][0][0];                     // `red` is not in use before the end
      for (int it = threadIdx.x; it < kTlSeg * kTlStride; it += kHbWaves * 64)
        seg[it] = tl_segment_sum(tl_ws, tl_nrb, it % kTlStride, it / kTlStride);
      __syncthreads();
      for (int t = threadIdx.x; t < kTlStride; t += kHbWaves * 64) {
        const float a = ((seg[t] + seg[kTlStride + t]) + seg[2 * kTlStride + t]) + seg[3 * kTlStride + t];
        const int idx = tl_small_index(t, tl_C);
        if (idx >= 0) small_out[idx] = a * gs;
        else if (idx == -1) tl_loss[0] = a / (float)B;
      }
      __syncthreads();
    } else {
      for (int i = threadIdx.x; i < n_small; i += kHbWaves * 64) small_out[i] = small_in[i] * gs;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * kHbCols, k = k0 + lane;
  const bool valid = k < K;
  const int bh = (B + kHbSplit - 1) / kHbSplit;
  const int row_lo = blockIdx.y * bh, row_hi = min(B, row_lo + bh);
  float wcol[kHeadO], acc[kHeadO];
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) {
    wcol[o] = w1[(size_t)o * K + (valid ? k : 0)];     // clamped, not predicated
    acc[o] = 0.f;
  }
  for (int b0 = row_lo; b0 < row_hi; b0 += kHbRows) {
    const int nb = row_hi - b0 < kHbRows ? row_hi - b0 : kHbRows;
    __syncthreads();
    for (int i = threadIdx.x; i < nb * (kHeadO / 4); i += kHbWaves * 64)
      *reinterpret_cast<f4*>(dzl + 4 * i) =
          reinterpret_cast<const f4*>(dz + (size_t)b0 * kHeadO)[i] * gs;   // b0 * 80 B: 16-byte aligned
    __syncthreads();
    // all of this wave's rows of the chunk are requested before the first is used; the loads are
    // unconditional on a clamped address (a predicated load puts a branch and a vmcnt(0) wait
    // between consecutive requests)
    constexpr int kPer = kHbRows / kHbWaves;
    float xv[kPer];
    uint8_t mb[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + kHbWaves * j;
      const size_t e = (valid && r < nb) ? (size_t)(b0 + r) * K + k : 0;
      xv[j] = NEED_DW ? x[e] : 0.f;                 // both loads unconditional: a load that
      mb[j] = MASKED ? mask1[(e * (size_t)bits1) >> 3] : 1;   // depends on the mask would serialise them
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + kHbWaves * j;
      if (r < nb) {                                 // wave-uniform
        // x holds the features BEFORE the dropout: what the forward multiplied is mask*scale*x
        // element (row, k) owns bits1 random bits at bit offset (row*K + k)*bits1; K % 4 == 0
        const bool kept = !MASKED ||
            (int)((mb[j] >> ((k * bits1) & 7)) & ((1u << bits1) - 1u)) >= thr1;
        const float xm = MASKED ? (kept ? xv[j] * scale1 : 0.f) : xv[j];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < kHeadO / 4; ++q) {
          const f4 d = *reinterpret_cast<const f4*>(dzl + r * kHeadO + 4 * q);
          s = fmaf(d.x, wcol[4 * q], s);
          s = fmaf(d.y, wcol[4 * q + 1], s);
          s = fmaf(d.z, wcol[4 * q + 2], s);
          s = fmaf(d.w, wcol[4 * q + 3], s);
          if (NEED_DW) {
            acc[4 * q] = fmaf(d.x, xm, acc[4 * q]);
            acc[4 * q + 1] = fmaf(d.y, xm, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(d.z, xm, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(d.w, xm, acc[4 * q + 3]);
          }
        }
        if (valid && dx) dx[(size_t)(b0 + r) * K + k] = kept ? s * scale1 : 0.f;
      }
    }
  }
  if (!NEED_DW) return;
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) red[wave][o][lane] = acc[o];
  __syncthreads();
  for (int i = threadIdx.x; i < kHeadO * kHbCols; i += kHbWaves * 64) {
    const int o = i / kHbCols, l = i - o * kHbCols;
    if (k0 + l < K) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < kHbWaves; ++w) a += red[w][o][l];
      atomicAdd(dw1 + (size_t)o * K + k0 + l, a);   // onto zero; exactly kHbSplit == 2 addends
    }
  }
}

// ------------------------------------------------------------------------------ soft-target CE
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void soft_ce_fwd_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ target,
                                                          float* __restrict__ loss, int B, int C) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* x = logits + (size_t)b * C;
    const float* t = target + (size_t)b * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - m);
    const float lse = logf(se);
    float rl = 0.f;
    for (int c = 0; c < C; ++c) rl += (x[c] - m - lse) * t[c];
    acc -= rl;
  }
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) loss[0] = tot / (float)B;
}

// dlogits[b][c] = gout / B * (softmax[b][c] * sum_c' t[b][c'] - t[b][c])
__global__ __launch_bounds__(256) void soft_ce_bwd_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ gout,
                                                          float* __restrict__ dlogits, int B,
                                                          int C) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float g = gout[0] / (float)B;
  const float* x = logits + (size_t)b * C;
  const float* t = target + (size_t)b * C;
  float m = x[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float se = 0.f, ts = 0.f;
  for (int c = 0; c < C; ++c) {
    se += expf(x[c] - m);
    ts += t[c];
  }
  const float lse = logf(se);
  for (int c = 0; c < C; ++c)
    dlogits[(size_t)b * C + c] = g * (expf(x[c] - m - lse) * ts - t[c]);
}

// ------------------------------------------------------------------------------ tail + loss, fused
// potes_tail_fwd_kernel, soft_ce_fwd_kernel, soft_ce_bwd_kernel and potes_tail_bwd_kernel are four
// launches of 4.6-7.2 us for a few KB of data; everything they compute is local to a batch row
// except five batch reductions (the loss, dW2, db2, db1).  potes_tail_loss_kernel does the row-local
// part for 4 rows per block — z, logits, the row's soft-target cross entropy, dlogits, dz (for
// d loss = 1) — and writes its rows' contributions to the reductions; potes_tail_loss_finalize_kernel
// adds the per-block contributions in a fixed order.  Two launches instead of four.
__global__ __launch_bounds__(kTailRows* kHeadO * 4) void potes_tail_loss_kernel(
    const float* __restrict__ partial, int KS, const float* __restrict__ b1,
    const uint8_t* __restrict__ mask2, float scale2, int thr2, const float* __restrict__ w2,
    const float* __restrict__ b2, const float* __restrict__ target, float* __restrict__ z,
    float* __restrict__ logits, float* __restrict__ dz, float* __restrict__ ws,
    float* __restrict__ zero, long long n_zero, int B, int C, int nrb, int target_kind) {
  if ((int)blockIdx.x >= nrb) {    // the other blocks clear the buffer head_bwd accumulates dW1 into
    const int nz = (int)gridDim.x - nrb;
    const long long per = (n_zero + nz - 1) / nz;
    const long long lo = (long long)((int)blockIdx.x - nrb) * per;
    const long long hi = lo + per < n_zero ? lo + per : n_zero;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) zero[i] = 0.f;
    return;
  }
  __shared__ float h[kTailRows][kHeadO], fac[kTailRows][kHeadO], dzl[kTailRows][kHeadO];
  __shared__ float lg[kTailRows][kHeadMaxC], dl[kTailRows][kHeadMaxC], rl[kTailRows];
  const int t = threadIdx.x, q = t & 3, e = t >> 2;            // e = r * kHeadO + o
  const int r = e / kHeadO, o = e - r * kHeadO;
  const int row = blockIdx.x * kTailRows + r;
  const size_t i = (size_t)(row < B ? row : 0) * kHeadO + o;
  const size_t plane = (size_t)B * kHeadO;
  float v = 0.f;
  int ks = q;
  for (; ks + 12 < KS; ks += 16) {                  // four independent loads in flight
    const float t0 = partial[(size_t)ks * plane + i], t1 = partial[(size_t)(ks + 4) * plane + i],
                t2 = partial[(size_t)(ks + 8) * plane + i], t3 = partial[(size_t)(ks + 12) * plane + i];
    v += t0; v += t1; v += t2; v += t3;
  }
  for (; ks < KS; ks += 4) v += partial[(size_t)ks * plane + i];
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  if (q == 0) {
    float hv = 0.f, f = 0.f;
    if (row < B) {
      v += b1 ? b1[o] : 0.f;
      z[i] = v;
      f = v > 0.f ? 1.f : 0.f;
      if (mask2) f = (int)mask2[i] >= thr2 ? f * scale2 : 0.f;
      hv = v * f;
    }
    h[r][o] = hv;
    fac[r][o] = f;
  }
  __syncthreads();
  if (t < kTailRows * C) {
    const int rr = t / C, c = t - rr * C;
    const int row2 = blockIdx.x * kTailRows + rr;
    float a = 0.f;
    if (row2 < B) {
      a = b2 ? b2[c] : 0.f;
#pragma unroll
      for (int k = 0; k < kHeadO; ++k) a = fmaf(h[rr][k], w2[c * kHeadO + k], a);
      logits[(size_t)row2 * C + c] = a;
    }
    lg[rr][c] = a;
  }
  __syncthreads();
  if (t < kTailRows) {             // the row's soft-target CE (train_model.py:45-54) and dlogits
    const int row2 = blockIdx.x * kTailRows + t;
    float loss_r = 0.f;
    if (row2 < B) {
      // target_kind 1: `target` is a uint8 class label per row (hard targets: one byte instead of
      // C floats — small enough to travel in kernel arguments, see GraphedTrainStep)
      float tg[kHeadMaxC];
      if (target_kind) {
        const int lab = reinterpret_cast<const uint8_t*>(target)[row2];
        for (int c = 0; c < C; ++c) tg[c] = c == lab ? 1.f : 0.f;
      } else {
        for (int c = 0; c < C; ++c) tg[c] = target[(size_t)row2 * C + c];
      }
      float m = lg[t][0];
      for (int c = 1; c < C; ++c) m = fmaxf(m, lg[t][c]);
      float se = 0.f, ts = 0.f;
      for (int c = 0; c < C; ++c) {
        se += expf(lg[t][c] - m);
        ts += tg[c];
      }
      const float lse = logf(se), inv_b = 1.f / (float)B;
      for (int c = 0; c < C; ++c) {
        loss_r -= (lg[t][c] - m - lse) * tg[c];
        dl[t][c] = inv_b * (expf(lg[t][c] - m - lse) * ts - tg[c]);
      }
    } else {
      for (int c = 0; c < C; ++c) dl[t][c] = 0.f;
    }
    rl[t] = loss_r;
  }
  __syncthreads();
  if (q == 0) {                    // dz = (z > 0) * m2 * (dlogits W2), for d loss = 1
    float sacc = 0.f;
    for (int c = 0; c < C; ++c) sacc = fmaf(dl[r][c], w2[c * kHeadO + o], sacc);
    const float d = sacc * fac[r][o];
    dzl[r][o] = d;
    if (row < B) dz[i] = d;
  }
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * kTlStride;
  if (t < kHeadMaxC * kHeadO) {    // dW2[c][o] contribution of these rows
    const int c = t / kHeadO, oo = t - c * kHeadO;
    float a = 0.f;
    if (c < C)
      for (int rr = 0; rr < kTailRows; ++rr) a = fmaf(dl[rr][c], h[rr][oo], a);
    out[t] = a;
  } else if (t < kHeadMaxC * kHeadO + kHeadMaxC) {
    const int c = t - kHeadMaxC * kHeadO;
    float a = 0.f;
    if (c < C)
      for (int rr = 0; rr < kTailRows; ++rr) a += dl[rr][c];
    out[t] = a;
  } else if (t < kHeadMaxC * kHeadO + kHeadMaxC + kHeadO) {
    const int oo = t - kHeadMaxC * kHeadO - kHeadMaxC;
    float a = 0.f;
    for (int rr = 0; rr < kTailRows; ++rr) a += dzl[rr][oo];
    out[t] = a;
  } else if (t == kHeadMaxC * kHeadO + kHeadMaxC + kHeadO) {
    out[t] = (rl[0] + rl[1]) + (rl[2] + rl[3]);
  }
}

// Column t of the per-block contributions, summed in a FIXED order: four threads per column
// take a quarter of the row blocks each (their loads all in flight together), then thread 0 of
// the column adds the four partial sums in order.  (One thread walking all 64 row blocks took
// 16 us: 64 dependent-latency loads.)
__global__ __launch_bounds__(kTlStride* kTlSeg) void potes_tail_loss_finalize_kernel(
    const float* __restrict__ ws, int nrb, float* __restrict__ loss, float* __restrict__ small,
    int B, int C) {
  // small = [dW2 (C x 20) | db2 (C) | db1 (20)], for d loss = 1
  __shared__ float seg[kTlSeg][kTlStride];
  const int t = threadIdx.x % kTlStride, sg = threadIdx.x / kTlStride;
  seg[sg][t] = tl_segment_sum(ws, nrb, t, sg);
  __syncthreads();
  if (sg != 0) return;
  const float a = ((seg[0][t] + seg[1][t]) + seg[2][t]) + seg[3][t];
  const int idx = tl_small_index(t, C);
  if (idx >= 0) small[idx] = a;
  else if (idx == -1) loss[0] = a / (float)B;
}

}  // namespace pcgmix

extern "C" int pcgmix_soft_ce_fwd_f32(const float* logits, const float* target, float* loss, int B,
                                      int C, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!logits || !target || !loss || B <= 0 || C <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(soft_ce_fwd_kernel, dim3(1), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), logits, target, loss, B, C);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_soft_ce_bwd_f32(const float* logits, const float* target, const float* gout,
                                      float* dlogits, int B, int C, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!logits || !target || !gout || !dlogits || B <= 0 || C <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(soft_ce_bwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), logits, target, gout, dlogits, B, C);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_head_fwd_f32(const float* x, const uint8_t* mask1, float scale1,
                                         int thr1, int bits1, const float* w1, const float* b1,
                                         const uint8_t* mask2, float scale2, int thr2,
                                         const float* w2, const float* b2, float* partial, float* z,
                                         float* logits, int B, int K, int C,
                                         pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!w2 || !z || !logits || C <= 0 || C > kHeadMaxC) return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const hipError_t e =
      launch_skinny_partial(x, w1, partial, B, K, kHeadO, s, mask1, scale1, thr1, bits1);
  if (e != hipSuccess) return (int)e;
  const int KS = pcgmix_skinny_linear_splits(B, K);
  hipLaunchKernelGGL(potes_tail_fwd_kernel, dim3((unsigned)((B + kTailRows - 1) / kTailRows)),
                     dim3(kTailRows * kHeadO * 4), 0, s, partial, KS, b1, mask2, scale2, thr2, w2, b2,
                     z, logits, B, C);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_head_bwd_f32(const float* dlogits, const float* z, const uint8_t* mask2,
                                         float scale2, int thr2, const float* w2, const float* x,
                                         const uint8_t* mask1, float scale1, int thr1, int bits1,
                                         const float* w1, float* dz, float* dw2, float* db2,
                                         float* db1, float* dw1, float* dx, int B, int K, int C,
                                         pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!dlogits || !z || !w2 || !x || !w1 || !dz || !dw2 || (!dw1 && !dx) || B <= 0 || K <= 0 ||
      C <= 0 || C > kHeadMaxC || (reinterpret_cast<uintptr_t>(dz) & 15) ||
      (mask1 && bits1 != 1 && bits1 != 2 && bits1 != 4 && bits1 != 8) || (K & 3))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long n_dw1 = dw1 ? (long long)kHeadO * K : 0;
  const unsigned zero_blocks = (unsigned)((n_dw1 + 4095) / 4096);
  hipLaunchKernelGGL(potes_tail_bwd_kernel, dim3(1 + zero_blocks), dim3(kTailBwdGroups * kHeadO), 0,
                     s, dlogits, z, mask2, scale2, thr2, w2, dz, dw2, db2, db1, B, C, dw1, n_dw1);
  const dim3 grid((unsigned)((K + kHbCols - 1) / kHbCols), kHbSplit), block(kHbWaves * 64);
  if (!dw1) {                      // frozen weights: dx only
    if (mask1)
      hipLaunchKernelGGL((potes_head_bwd_kernel<true, false>), grid, block, 0, s, dz, x, mask1, scale1,
                         thr1, bits1, w1, dw1, dx, B, K, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0);
    else
      hipLaunchKernelGGL((potes_head_bwd_kernel<false, false>), grid, block, 0, s, dz, x, mask1, 1.0f,
                         0, 8, w1, dw1, dx, B, K, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0);
  } else if (mask1) {
    hipLaunchKernelGGL((potes_head_bwd_kernel<true, true>), grid, block, 0, s, dz, x, mask1, scale1,
                       thr1, bits1, w1, dw1, dx, B, K, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0);
  } else {
    hipLaunchKernelGGL((potes_head_bwd_kernel<false, true>), grid, block, 0, s, dz, x, mask1, 1.0f, 0,
                       8, w1, dw1, dx, B, K, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0);
  }
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_head_saliency_f32(const float* x, const float* w1, const float* b1,
                                              const float* w2, const float* seed, float* partial,
                                              float* dz, float* dx, int B, int K, int C,
                                              pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !w1 || !w2 || !seed || !partial || !dz || !dx || B <= 0 || K <= 0 || (K & 3) || C <= 0 ||
      C > kHeadMaxC || (reinterpret_cast<uintptr_t>(dz) & 15))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const hipError_t e = launch_skinny_partial(x, w1, partial, B, K, kHeadO, s, nullptr, 1.0f, 0, 8);
  if (e != hipSuccess) return (int)e;
  const int KS = pcgmix_skinny_linear_splits(B, K);
  hipLaunchKernelGGL(potes_tail_saliency_kernel, dim3((unsigned)((B + kTailRows - 1) / kTailRows)),
                     dim3(kTailRows * kHeadO * 4), 0, s, partial, KS, b1, w2, seed, dz, B, C);
  const dim3 grid((unsigned)((K + kHbCols - 1) / kHbCols), kHbSplit), block(kHbWaves * 64);
  hipLaunchKernelGGL((potes_head_bwd_kernel<false, false>), grid, block, 0, s, dz, x, nullptr, 1.0f, 0,
                     8, w1, nullptr, dx, B, K, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, 0);
  return (int)hipGetLastError();
}

extern "C" long long pcgmix_potes_head_loss_workspace_floats(int B) {
  return B <= 0 ? 0 : (long long)((B + pcgmix::kTailRows - 1) / pcgmix::kTailRows) * pcgmix::kTlStride;
}

extern "C" int pcgmix_potes_head_loss_fwd_f32(const float* x, const uint8_t* mask1, float scale1,
                                              int thr1, int bits1, const float* w1, const float* b1,
                                              const uint8_t* mask2, float scale2, int thr2,
                                              const float* w2, const float* b2, const float* target,
                                              float* partial, float* z, float* logits, float* dz,
                                              float* loss, float* small, float* ws, float* dw1_zero,
                                              int defer_finalize, int target_kind, int B, int K,
                                              int C, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!w2 || !target || !z || !logits || !dz || !loss || !small || !ws || C <= 0 || C > kHeadMaxC ||
      (reinterpret_cast<uintptr_t>(dz) & 15) || (target_kind != 0 && target_kind != 1))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const hipError_t e =
      launch_skinny_partial(x, w1, partial, B, K, kHeadO, s, mask1, scale1, thr1, bits1);
  if (e != hipSuccess) return (int)e;
  const int KS = pcgmix_skinny_linear_splits(B, K);
  const int nrb = (B + kTailRows - 1) / kTailRows;
  const long long n_zero = dw1_zero ? (long long)kHeadO * K : 0;
  const unsigned zero_blocks = (unsigned)((n_zero + 4095) / 4096);
  hipLaunchKernelGGL(potes_tail_loss_kernel, dim3((unsigned)nrb + zero_blocks),
                     dim3(kTailRows * kHeadO * 4), 0, s, partial, KS, b1, mask2, scale2, thr2, w2, b2,
                     target, z, logits, dz, ws, dw1_zero, n_zero, B, C, nrb, target_kind);
  if (!defer_finalize)
    hipLaunchKernelGGL(potes_tail_loss_finalize_kernel, dim3(1), dim3(kTlStride * kTlSeg), 0, s, ws,
                       nrb, loss, small, B, C);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_head_loss_bwd_f32(const float* dz, const float* gscale, const float* x,
                                              const uint8_t* mask1, float scale1, int thr1, int bits1,
                                              const float* w1, const float* small_in,
                                              float* small_out, float* dw1, float* dx,
                                              const float* deferred_ws, float* deferred_loss, int B,
                                              int K, int C, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!dz || !x || !w1 || (!dw1 && !dx) || B <= 0 || K <= 0 || (K & 3) || C <= 0 || C > kHeadMaxC ||
      (reinterpret_cast<uintptr_t>(dz) & 15) ||
      (mask1 && bits1 != 1 && bits1 != 2 && bits1 != 4 && bits1 != 8) ||
      (small_out && !small_in && !deferred_ws) || (deferred_ws && (!deferred_loss || !small_out)))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int n_small = C * kHeadO + C + kHeadO;
  const dim3 grid((unsigned)((K + kHbCols - 1) / kHbCols), kHbSplit), block(kHbWaves * 64);
#define PCGMIX_HB(M, D)                                                                          \
  hipLaunchKernelGGL((potes_head_bwd_kernel<M, D>), grid, block, 0, s, dz, x, mask1,              \
                     mask1 ? scale1 : 1.0f, mask1 ? thr1 : 0, mask1 ? bits1 : 8, w1, dw1, dx, B, K, \
                     gscale, small_in, small_out, n_small, deferred_ws,                          \
                     (B + kTailRows - 1) / kTailRows, deferred_loss, C)
  if (!dw1) { if (mask1) PCGMIX_HB(true, false); else PCGMIX_HB(false, false); }
  else { if (mask1) PCGMIX_HB(true, true); else PCGMIX_HB(false, true); }
#undef PCGMIX_HB
  return (int)hipGetLastError();
}
