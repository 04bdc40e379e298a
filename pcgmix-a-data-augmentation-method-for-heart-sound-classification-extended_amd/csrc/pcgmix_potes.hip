// pcgmix_potes.hip — the Potes 1D-CNN's convolutional branch as fused kernels (gfx950).
//
// Reference: models.py:359-381 (conv_block_1d, CNN_potes.cnn1) — per band-pass channel
//     Conv1d(1->8, k5, pad1) + ReLU + MaxPool(2)  ->  Conv1d(8->4, k5, pad1) + ReLU + MaxPool(2)
// applied with the SAME weights to all four bands (models.py:444-455), i.e. to N = 4*B rows.
// Through MIOpen this stack costs ~1.8 ms per bs=256 step on MI355X (a naive direct-conv
// fallback, layout transposes, separate ReLU / pooling / pooling-backward kernels;
// profiles/r1_bench_kernel_stats.csv) for ~40 MB of unavoidable HBM traffic.  Here:
//
//   potes_fwd_mfma_kernel         reads each input row once, keeps the 8-channel intermediate in
//                                 LDS, writes the pooled (N,4,P2) activations and, on request, the
//                                 ReLU/pool routing of both layers (m2, s1): 4*T + 16*P2 bytes per
//                                 row; block-form matrix instruction v_mfma_f32_4x4x1_16b_f32.
//   potes_bwd_pair_kernel         weight gradients from x, dL/dh2 and m2: recomputes layer 1 on the
//                                 lane that consumes it, packed multiply-adds over channel pairs,
//                                 212 gradients per block in registers; per-block partials are summed
//                                 by a second tiny kernel or inside the optimiser launch
//                                 (deterministic, no float atomics).
//   potes_input_grad_pair_kernel  dL/dx from dL/dh2, m2 and s1 (saliency maps).
//   potes_bwd_kernel<false>, potes_input_grad_kernel: the same two gradients for callers that kept
//                                 no routing (they recompute the whole forward per tile).
// The VALU forward and the earlier mask-based backward kernels (rounds 1-3: potes_fwd_kernel,
// potes_bwd_kernel<true>, potes_bwd_fused_kernel, potes_input_grad_mask_kernel) were removed from
// the product library in round 4; they are in the history at commit 2fa984b and their measurements
// in profiles/r2_*, r3_*.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <cmath>

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

// Bytes per (row, channel) of the packed first-layer selectors (see layer1_t, s1g).
__host__ __device__ inline int potes_s1_row_bytes(const PotesDims& d) { return (d.P1 >> 2) + 1; }

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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_load12(const float* p, float (&v)[12]) {  // p 16-byte aligned
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4),
           c = *reinterpret_cast<const f4*>(p + 8);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void lds_load8(const float* p, float (&v)[8]) {  // p 16-byte aligned
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Layer 1 (conv 1->8 k5 + ReLU + pool 2) for pooled positions q = qlo + qq, qq in [0, nq),
// nq % 4 == 0.  Work item = (channel, 4 consecutive qq): 12 staged inputs feed 8 conv outputs.
// a1s[ci*nq + qq] is 0 outside [0,P1) — that IS conv2's zero padding.  If sel != nullptr it gets
// which conv output won the pool and survived the ReLU: 0 none, 1 first (i = 2q), 2 second.
// SWZ (forward kernel): xs and a1s are stored as two planes of float4 — even-indexed float4 in
// plane E, odd-indexed in plane O (plane strides xplane / aplane floats) — so that the three
// 16-byte reads of a 12-float window starting at float4 index 2g become E[g], O[g], E[g+1]:
// consecutive lanes read consecutive 16 bytes.  With the plain layout the windows start 32 bytes
// apart and every ds_read_b128 is a 2-way bank conflict (64 banks x 4 B, 16 lanes per group).
// s1g (forward kernel only, may be nullptr): global plane (8, own_hi bytes) of this row that
// receives the same pool/ReLU selectors, four per byte (own_lo unused) — what the mask-based
// backward kernels read instead of recomputing this layer.
// c[m] = bias + sum_k w[k] * xw[m + k], m < 8, as float2 pairs (see potes_fwd_kernel): the ONE
// place where the first layer's arithmetic order is written down — forward and every backward
// recompute it through this function, so they agree on which ReLUs are alive and which element
// wins each max-pool.
__device__ __forceinline__ void conv1_window(const float (&xw)[12], const float (&w)[kK], float bias,
                                             float (&c)[8]) {
  f2 ce[4], co_[3];
  float o0 = 0.f, o7 = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) ce[m] = f2{bias, bias};
#pragma unroll
  for (int m = 0; m < 3; ++m) co_[m] = f2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kK; k += 2) {
    const f2 wk = {w[k], w[k]};
#pragma unroll
    for (int m = 0; m < 4; ++m)
      ce[m] = __builtin_elementwise_fma(wk, f2{xw[2 * m + k], xw[2 * m + k + 1]}, ce[m]);
  }
#pragma unroll
  for (int k = 1; k < kK; k += 2) {
    const f2 wk = {w[k], w[k]};
#pragma unroll
    for (int m = 0; m < 3; ++m)
      co_[m] = __builtin_elementwise_fma(wk, f2{xw[2 * m + 1 + k], xw[2 * m + 2 + k]}, co_[m]);
    o0 = fmaf(w[k], xw[k], o0);
    o7 = fmaf(w[k], xw[7 + k], o7);
  }
  c[0] = ce[0].x + o0;
  c[7] = ce[3].y + o7;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    c[2 * m + 1] = ce[m].y + co_[m].x;
    c[2 * m + 2] = ce[m + 1].x + co_[m].y;
  }
}

// ReLU + MaxPool(2) of one pair of conv outputs, branch-free: value, and which of the two won and
// survived (0 none, 1 first, 2 second — torch's max-pool keeps the FIRST maximum: strict '>').
// Written as selects: the nested-if form compiled to an exec-mask branch per position
// (s_and_saveexec / s_cbranch_execz / s_or exec plus hazard nops, ~10 scalar instructions around
// three vector ones).
__device__ __forceinline__ void relu_pool2(float za, float zb, bool valid, float& a, uint32_t& sc) {
  const float ra = fmaxf(za, 0.f), rb = fmaxf(zb, 0.f);
  const bool second = rb > ra;
  const float best = second ? rb : ra;
  const uint32_t code = second ? 2u : (ra > 0.f ? 1u : 0u);
  a = valid ? best : 0.f;
  sc = valid ? code : 0u;
}

template <bool SWZ>
__device__ __forceinline__ void layer1_t(const PotesWeights& W, const float* xs, float* a1s,
                                         uint8_t* sel, int qlo, int nq, int P1, int xplane,
                                         int aplane, uint8_t* __restrict__ s1g = nullptr,
                                         int own_lo = 0, int own_hi = 0, int g_first = 0) {
  // groups g_first .. nq/4-1 of every channel (g_first > 0: the caller never reads the first
  // 4*g_first positions and the remaining items fill whole rounds of the block)
  const int groups = nq / 4 - g_first;
  for (int item = threadIdx.x; item < kC1 * groups; item += kPotThreads) {
    const int ci = item / groups, g = g_first + item - ci * groups;
    float xw[12], w[kK];
    if (SWZ) {
      const f4 a = *reinterpret_cast<const f4*>(xs + 4 * g),
               b = *reinterpret_cast<const f4*>(xs + xplane + 4 * g),
               c = *reinterpret_cast<const f4*>(xs + 4 * g + 4);
      xw[0] = a.x; xw[1] = a.y; xw[2] = a.z; xw[3] = a.w; xw[4] = b.x; xw[5] = b.y; xw[6] = b.z;
      xw[7] = b.w; xw[8] = c.x; xw[9] = c.y; xw[10] = c.z; xw[11] = c.w;
    } else {
      lds_load12(xs + 8 * g, xw);
    }
#pragma unroll
    for (int k = 0; k < kK; ++k) w[k] = W.w1[ci * kK + k];
    const float bias = W.b1[ci];
    float c[8];
    conv1_window(xw, w, bias, c);
    f4 out;
    uint32_t sels = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = qlo + 4 * g + u;
      float a;
      uint32_t sc;
      relu_pool2(c[2 * u], c[2 * u + 1], q >= 0 && q < P1, a, sc);
      out[u] = a;
      sels |= sc << (8 * u);
    }
    if (SWZ) {
      *reinterpret_cast<f4*>(a1s + ci * 2 * aplane + (g & 1) * aplane + 4 * (g >> 1)) = out;
      if (s1g) {
        // Packed: selector of position q in bits 2*((q+1)&3) of byte (q+1)>>2 of the channel's
        // row (stride own_hi = P1/4 + 1 bytes).  qlo + 1 is a multiple of 4, so a work item's four
        // positions are exactly one byte; the one group that two neighbouring tiles both compute
        // (the halo) gets the same byte from both — same inputs, same arithmetic.
        const int bi = ((qlo + 1) >> 2) + g;
        if (bi >= 0 && bi < own_hi)
          s1g[(size_t)ci * own_hi + bi] = (uint8_t)((sels & 3u) | ((sels >> 6) & 0xcu) |
                                                    ((sels >> 12) & 0x30u) | ((sels >> 18) & 0xc0u));
      }
    } else {
      *reinterpret_cast<f4*>(a1s + ci * nq + 4 * g) = out;
      if (sel) *reinterpret_cast<uint32_t*>(sel + ci * nq + 4 * g) = sels;
    }
  }
}

__device__ __forceinline__ void layer1(const PotesWeights& W, const float* xs, float* a1s,
                                       uint8_t* sel, int qlo, int nq, int P1, int g_first = 0) {
  layer1_t<false>(W, xs, a1s, sel, qlo, nq, P1, 0, 0, nullptr, 0, 0, g_first);
}

// ---------------------------------------------------------------------------------- forward
constexpr int kFwdTP = 252;                     // pooled outputs per block (4 per lane, wave = co):
                                                // 2*252+4 = 508 layer-1 positions = 127 groups x 8
                                                // channels = 1016 work items = 4 rounds of 256 threads
                                                // (256 outputs would need a 5th round for 8 items)
constexpr int kFwdNQ = 2 * kFwdTP + 4;          // 508

// SAVE: also write what the mask-based backward kernels need so that they do not recompute the
// forward: m2 (N, 4, ceil(P2/4)) — per pooled output 2 bits (0 = ReLU-dead, 1 = first conv output
// of the pooled pair won, 2 = second), four outputs per byte — and, if s1 != nullptr, s1
// (N, 8, P1) — the same selector for the first layer, one byte per pooled position.
//
// rnd != nullptr (SAVE only): the launch also fills rnd[0 .. rnd_n16) (16-byte words) with the
// uniformly random bytes the classifier head's dropouts read (pcgmix_head.hip) — word i =
// counter_hash(key, 4i .. 4i+3), key = two 32-bit words in device memory (or, key == nullptr, the
// two kernel arguments).  In a captured training step this replaces an eager `random_()` launch
// before every replay: the key changes per replay (it rides with the step payload), the graph
// does not.
__device__ __forceinline__ uint32_t counter_hash(uint32_t i, uint32_t k0, uint32_t k1) {
  uint32_t h = i * 0x9E3779B1u + k0;      // murmur3's 32-bit finaliser, keyed before and inside
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h ^= k1;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// ---------------------------------------------------------------------------------- forward, MFMA
// The same stack on the matrix cores (round 3).  A k5 convolution with 8 or 4 output channels is a
// GEMM with a tiny N, which the block form v_mfma_f32_4x4x1_16b_f32 fits exactly: 16 independent
// 4x4 outer products per instruction, D_b[i][j] += A_b[i] * B_b[j].  Operand map (probed on the
// hardware, profiles/probes/mfma_f32_4x4_rate.hip): A lane 4b+i = row i of block b, B lane 4b+j =
// column j, D lane 4b+j register i = element (i, j).  Here
//     row    = a conv output POSITION  (lane L = 4b+i owns positions 8L .. 8L+7, one per accumulator r)
//     column = an output CHANNEL       (lane&3 = channel; B = that channel's weight, one VGPR per tap)
// so with the lane's 12-float input window xw[0..11] in registers (three conflict-free
// ds_read_b128 from the even/odd float4 planes) the update for tap k and position offset r is
//     D[r] = mfma(xw[r + k], w[k], D[r])
// — no operand assembly at all (the VALU kernel spends about one v_mov per packed multiply-add on
// it), 64 distinct LDS words feed 40 instructions, and the K = 1 form makes every accumulator an
// in-order fmaf chain from the bias (exact f32: what a scalar loop over (ci, k) would give).
// D[r] register i' of lane 4b+j is position 32b + 8i' + r of channel j: a lane ends up with 32
// CONSECUTIVE positions of one channel, so ReLU + MaxPool(2) and the routing codes are in-lane.
// Measured issue interval of this instruction: ~12 cycles (two passes + 4), 256 multiply-adds each,
// i.e. 2/3 of the f32 peak when nothing else issues; VALU work between two of them costs ~7 cycles
// extra on top of its own, so the matrix instructions are kept in runs of 40 or more.
//
// Block = 256 threads = one tile of kFwdTP = 252 pooled outputs of one row, as in the VALU kernel:
// 504 conv2 positions fed by 508 a1 positions = 1016 conv1 positions.
//   layer 1: wave w takes conv1 unit u = w >> 1 (512 positions, 8 per lane) and channel half
//            h = w & 1: one run of 40 instructions on 8 independent accumulators; pooled values to
//            the a1 rows in LDS (row pitch 516 floats = 16 B mod 256: the 16-byte stores of a
//            quarter wave, 64 B apart per block and one row apart per channel, tile the banks);
//   layer 2: wave w takes conv2 positions 128w .. 128w+127, two per lane (window = 6 floats, three
//            ds_read_b64, conflict-free), 8 x 5 x 2 = 80 instructions; even and odd input channels
//            accumulate separately (four independent chains) and are added at the end.  A lane ends
//            with 4 consecutive pooled outputs of channel j: stores and the m2 byte as in the VALU
//            kernel.
// 480 matrix instructions per tile, 120 per wave.
constexpr int kMfXE = 132 * 4;                  // x: even float4 plane (indices 0..128 used), floats
constexpr int kMfXFloats = (132 + 128) * 4;     // + odd plane (0..127)
constexpr int kMfXN = 1028;                     // generic path: staged samples (float4 index <= 256)
constexpr int kMfAPitch = 516;                  // a1 row: positions 0..515

__device__ __forceinline__ void mf_window(const float* planeE, const float* planeO, int g,
                                          float (&w)[12]) {
  const f4 a = *reinterpret_cast<const f4*>(planeE + 4 * g),
           b = *reinterpret_cast<const f4*>(planeO + 4 * g),
           c = *reinterpret_cast<const f4*>(planeE + 4 * g + 4);
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
}

// The matrix instruction as inline assembly: the compiler's scheduler otherwise re-orders a run
// (depth-first along one accumulator, results read out through v_accvgpr_read in between), which
// turns 8 independent chains into dependent pairs with VALU instructions among them.  Volatile
// statements keep their order; the accumulators stay in VGPRs.  The hazard recogniser does not
// see an MFMA in an asm statement, so a run is bracketed by explicit wait states: mf_run_begin
// after the VALU writes of the operands, mf_run_end before anything reads the accumulators
// (a 2-pass XDL result needs 5; 16 are cheap next to a run of 40).
__device__ __forceinline__ void mfma4(f4& acc, float a, float b) {
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first tap of a chain: the accumulator starts from c (a bias quad, or the inline constant 0)
__device__ __forceinline__ void mfma4_from(f4& acc, float a, float b, const f4& c) {
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void mfma4_from0(f4& acc, float a, float b) {
  asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mf_run_begin() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 4");
}
__device__ __forceinline__ void mf_run_end() {
  asm volatile("s_nop 7\n\ts_nop 7");
  __builtin_amdgcn_sched_barrier(0);
}

// Persistent blocks: tile = item of a strided list (item -> row item / tiles, tile item % tiles);
// the NEXT item's samples are requested from global memory into registers before the current
// item's matrix work starts and go to LDS after it, so the load latency (the kernel's skeleton —
// stage, barrier, compute, barrier, compute, store — used to run once per block: 13.5 us of the
// 27 us VALU kernel) hides behind 120 matrix instructions per wave.
struct MfStage {           // one thread's share of a tile's samples
  f4 v;                    // aligned path: x[4 p0 - 4 + 4t .. +3]; generic path: elements t + 256 c,
  float v4;                // c < 4, and c = 4
};

// Tile element e is x[4 p0 - 3 + e].  Elements 0 .. 1019 feed the 1016 conv1 positions whose
// pooled values reach a kept output; the aligned path stages -1 .. 1022 (element -1 lands in the
// even plane's padding), the rest of the planes is never initialised and only ever feeds matrix
// rows (positions) that nobody keeps.
__device__ __forceinline__ float mf_x_at(const float* __restrict__ xrow, int g, int T) {
  const float val = xrow[g < 0 ? 0 : (g >= T ? T - 1 : g)];
  return (g >= 0 && g < T) ? val : 0.f;
}
__device__ __forceinline__ void mf_stage_load(MfStage& st, const float* __restrict__ xrow, int p0,
                                              int T, bool fast) {
  const int t = threadIdx.x;
  if (fast) {
    // rows are 16-byte aligned and T % 4 == 0: a quad is wholly inside the row or wholly outside
    const int g0 = 4 * p0 - 4 + 4 * t;
    const int gc = g0 < 0 ? 0 : (g0 >= T ? T - 4 : g0);
    const f4 v = *reinterpret_cast<const f4*>(xrow + gc);
    const bool in = g0 >= 0 && g0 < T;
    st.v = f4{in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f};
  } else {
    const int g = 4 * p0 - 3 + t;
    st.v = f4{mf_x_at(xrow, g, T), mf_x_at(xrow, g + 256, T), mf_x_at(xrow, g + 512, T),
              mf_x_at(xrow, g + 768, T)};
    st.v4 = mf_x_at(xrow, g + 1024, T);
  }
}

__device__ __forceinline__ void mf_put(float* xs, int e, float v) {   // tile element e -> planes
  const int i = e >> 2;
  xs[(i & 1) * kMfXE + 4 * (i >> 1) + (e & 3)] = v;
}
__device__ __forceinline__ void mf_stage_store(const MfStage& st, float* xs, bool fast) {
  const int t = threadIdx.x;
  if (fast) {
    mf_put(xs, 4 * t - 1, st.v.x);
    mf_put(xs, 4 * t, st.v.y);
    mf_put(xs, 4 * t + 1, st.v.z);
    mf_put(xs, 4 * t + 2, st.v.w);
  } else {
    mf_put(xs, t, st.v.x);
    mf_put(xs, t + 256, st.v.y);
    mf_put(xs, t + 512, st.v.z);
    mf_put(xs, t + 768, st.v.w);
    if (t + 1024 < kMfXN) mf_put(xs, t + 1024, st.v4);
  }
}

// max(a, b, 0) in one instruction.  fmaxf(fmaxf(a, 0), fmaxf(b, 0)) compiles to three v_max_f32
// plus a canonicalising v_max(x, x) per operand under the IEEE mode; the operands here are finite.
__device__ __forceinline__ float relu_max2(float a, float b) {
  float r;
  asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <bool SAVE>
__global__ __launch_bounds__(kPotThreads, 4) void potes_fwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ h2,
    uint8_t* __restrict__ m2, uint8_t* __restrict__ s1, int N, int T, uint4* __restrict__ rnd,
    long long rnd_n16, const uint32_t* __restrict__ key, uint32_t key_lo, uint32_t key_hi) {
  if (SAVE && rnd) {
    const uint32_t k0 = key ? key[0] : key_lo, k1 = key ? key[1] : key_hi;
    const long long stride = (long long)gridDim.x * kPotThreads;
    for (long long i = (long long)blockIdx.x * kPotThreads + threadIdx.x; i < rnd_n16; i += stride) {
      const uint32_t c = (uint32_t)i * 4u;
      rnd[i] = make_uint4(counter_hash(c, k0, k1), counter_hash(c + 1, k0, k1),
                          counter_hash(c + 2, k0, k1), counter_hash(c + 3, k0, k1));
    }
  }
  __shared__ __align__(16) float xs[kMfXFloats];
  __shared__ __align__(16) float a1s[kC1 * kMfAPitch];
  const PotesDims d = potes_dims(T);
  const int tiles = (d.P2 + kFwdTP - 1) / kFwdTP, items = N * tiles;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int blk = lane >> 2, j = lane & 3;
  const int u = wave >> 1, h = wave & 1;
  const bool fast = !(T & 3) && !(reinterpret_cast<uintptr_t>(x) & 15);
  // this lane's B operands: channel j + 4h of layer 1, channel j of layer 2
  float wb1[kK], wb2[kC1][kK];
#pragma unroll
  for (int k = 0; k < kK; ++k) wb1[k] = w1[(j + 4 * h) * kK + k];
#pragma unroll
  for (int ci = 0; ci < kC1; ++ci)
#pragma unroll
    for (int k = 0; k < kK; ++k) wb2[ci][k] = w2[(j * kC1 + ci) * kK + k];
  const float bias1 = b1[j + 4 * h], bias2 = b2[j];
  const f4 bias1q = {bias1, bias1, bias1, bias1}, bias2q = {bias2, bias2, bias2, bias2};
  // positions 512 .. 515 of every a1 row are read (by conv2 positions nobody keeps), never written
  if (threadIdx.x < kC1) *reinterpret_cast<f4*>(a1s + threadIdx.x * kMfAPitch + 512) = f4{0.f, 0.f, 0.f, 0.f};
  const int s1row = potes_s1_row_bytes(d), m2row = (d.P2 + 3) / 4;

  int item = blockIdx.x;
  MfStage st;
  st.v = f4{0.f, 0.f, 0.f, 0.f};
  st.v4 = 0.f;
  if (item < items) mf_stage_load(st, x + (size_t)(item / tiles) * T, (item % tiles) * kFwdTP, T, fast);
  for (; item < items; item += gridDim.x) {
    const int n = item / tiles, p0 = (item - n * tiles) * kFwdTP;
    const int qlo = 2 * p0 - 1;
    mf_stage_store(st, xs, fast);
    __syncthreads();
    {
      const int nxt = item + gridDim.x;
      if (nxt < items) mf_stage_load(st, x + (size_t)(nxt / tiles) * T, (nxt % tiles) * kFwdTP, T, fast);
    }
    // ---- layer 1: conv1 positions 512u + 8 lane + r of channel j + 4h.  The lane's 16 pooled
    // values are a1 positions 256u + 16 blk + 4 i + t (t <-> accumulators 2t, 2t+1).
    {
      float xw[12];
      mf_window(xs, xs + kMfXE, 64 * u + lane, xw);
      f4 acc[8];
      mf_run_begin();
#pragma unroll
      for (int r = 0; r < 8; ++r) mfma4_from(acc[r], xw[r], wb1[0], bias1q);
#pragma unroll
      for (int k = 1; k < kK; ++k)
#pragma unroll
        for (int r = 0; r < 8; ++r) mfma4(acc[r], xw[r + k], wb1[k]);
      mf_run_end();
      const int qq0 = 256 * u + 16 * blk, qb = qlo + qq0;
      const bool edge = qlo + 256 * u < 0 || qlo + 256 * u + 256 > d.P1;   // wave-uniform
      float* arow = a1s + (j + 4 * h) * kMfAPitch + qq0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f4 out;
#pragma unroll
        for (int t = 0; t < 4; ++t) out[t] = relu_max2(acc[2 * t][i], acc[2 * t + 1][i]);
        if (edge) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int q = qb + 4 * i + t;
            out[t] = (q >= 0 && q < d.P1) ? out[t] : 0.f;
          }
        }
        *reinterpret_cast<f4*>(arow + 4 * i) = out;
      }
      if (SAVE && s1) {
        // packed selectors (layer1_t): position q in bits 2*((q+1)&3) of byte (q+1)>>2; q + 1 =
        // 2 p0 + 4 g + t with g = 64u + 4 blk + i, so a float4 group is exactly byte p0/2 + g and
        // the lane's four groups are four consecutive bytes: one (unaligned) 32-bit store.
        uint32_t sel4 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            // relu_pool2's rule: 2 if the second candidate's ReLU is strictly larger, else 1 if the
            // first survives its ReLU, else 0; 0 outside [0, P1)
            const float za = acc[2 * t][i], zb = acc[2 * t + 1][i];
            uint32_t sc = (zb > za && zb > 0.f) ? 2u : (za > 0.f ? 1u : 0u);
            if (edge) {
              const int q = qb + 4 * i + t;
              sc = (q >= 0 && q < d.P1) ? sc : 0u;
            }
            sel4 |= sc << (8 * i + 2 * t);
          }
        uint8_t* row = s1 + ((size_t)n * kC1 + j + 4 * h) * s1row;
        const int g0 = 64 * u + 4 * blk, bi0 = (p0 >> 1) + g0;
        if (g0 + 3 < kFwdNQ / 4 && bi0 + 3 < s1row) {
          typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
          *reinterpret_cast<u32_unaligned*>(row + bi0) = sel4;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (g0 + i < kFwdNQ / 4 && bi0 + i < s1row) row[bi0 + i] = (uint8_t)(sel4 >> (8 * i));
        }
      }
    }
    __syncthreads();
    // ---- layer 2: conv2 positions 128 wave + 2 lane + r of channel j
    f4 acc2[2][2];
    {
      // channel pairs: the next pair's windows are requested before the current run of 20
      float aw[2][2][6];
      const float* abase = a1s + 128 * wave + 2 * lane;
      auto window = [&](int ci, float (&w)[6]) {
        const f2 a = *reinterpret_cast<const f2*>(abase + ci * kMfAPitch),
                 b = *reinterpret_cast<const f2*>(abase + ci * kMfAPitch + 2),
                 c = *reinterpret_cast<const f2*>(abase + ci * kMfAPitch + 4);
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = c.x; w[5] = c.y;
      };
      window(0, aw[0][0]);
      window(1, aw[0][1]);
#pragma unroll
      for (int cp = 0; cp < kC1; cp += 2) {
        const int cur = (cp >> 1) & 1;
        if (cp + 2 < kC1) {
          window(cp + 2, aw[cur ^ 1][0]);
          window(cp + 3, aw[cur ^ 1][1]);
        }
        mf_run_begin();
#pragma unroll
        for (int k = 0; k < kK; ++k)
#pragma unroll
          for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              if (cp == 0 && k == 0) {             // the chains start: bias on the even channels
                if (par == 0) mfma4_from(acc2[0][r], aw[cur][0][r], wb2[0][0], bias2q);
                else mfma4_from0(acc2[1][r], aw[cur][1][r], wb2[1][0]);
              } else {
                mfma4(acc2[par][r], aw[cur][par][r + k], wb2[cp + par][k]);
              }
            }
        mf_run_end();
      }
    }
    // register i of acc2[.][r] is conv2 position 128 wave + 8 blk + 2 i + r: pooled output
    // pp = 64 wave + 4 blk + i  (r = 0, 1 are its two candidates)
    const int pp = 64 * wave + 4 * blk, p = p0 + pp;
    f4 o;
    uint32_t code = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // relu_pool2's rule: 2 if the second candidate's ReLU is strictly larger, else 1 if the
      // first survives its ReLU, else 0
      const float za = acc2[0][0][i] + acc2[1][0][i], zb = acc2[0][1][i] + acc2[1][1][i];
      o[i] = relu_max2(za, zb);
      if (SAVE) code |= ((zb > za && zb > 0.f) ? 2u : (za > 0.f ? 1u : 0u)) << (2 * i);
    }
    if (pp < kFwdTP) {                               // else: outputs owned by the next tile
      float* dst = h2 + ((size_t)n * kC2 + j) * d.P2 + p;
      uint8_t* mdst = m2 + ((size_t)n * kC2 + j) * m2row + (p >> 2);
      if (p0 + kFwdTP <= d.P2) {                     // tile wholly inside the row (block-uniform)
        if (SAVE) *mdst = (uint8_t)code;
        if ((d.P2 & 3) == 0) {
          *reinterpret_cast<f4*>(dst) = o;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[i] = o[i];
        }
      } else {
        if (SAVE && p < d.P2) {
          uint32_t keep = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (p + i < d.P2) keep |= 3u << (2 * i);
          *mdst = (uint8_t)(code & keep);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (p + i < d.P2) dst[i] = o[i];
      }
    }
  }
}

// Sum over the 64 lanes of a wave on the VALU alone (DPP: no LDS traffic), total in lane 63.
// quad swaps, half-row and row mirrors leave every lane of a row with the row's sum; row_bcast15 /
// row_bcast31 carry it across the four rows.  The kernel-end reduction of the backward's 53
// accumulators was 318 ds_bpermute_b32 per wave through __shfl_xor — 6.5 us of a 78 us kernel, all
// of it in the tail where every block of the launch does nothing else.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                 0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);    // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);    // row_mirror
  v = dpp_add<0x142, 0xa>(v);    // row_bcast15 into rows 1, 3
  v = dpp_add<0x143, 0xc>(v);    // row_bcast31 into rows 2, 3
  return v;
}

// ---------------------------------------------------------------------------------- backward
// Work item = (row, tile).  A tile OWNS kBwdTP pooled outputs p0 .. p0+TP-1, i.e. conv2 outputs
// j = 2p0+s (s < 2TP), layer-1 pooled positions q = 2p0+r (r < 2TP) and conv1 outputs
// i = 4p0+ii (ii < 4TP); every position is owned by exactly one tile.  To get the complete
// dL/da1 of its owned q it recomputes conv2 on the extended range p0-2 .. p0+TP (128 values).
constexpr int kBwdTP = 125;
constexpr int kBwdNP = 128;                     // extended pooled outputs pe = p0-2+pp
constexpr int kBwdNJ = 2 * kBwdNP;              // 256 conv2 outputs, j = 2p0-4+jj
constexpr int kBwdNQ = 260;                     // a1 positions q = 2p0-5+qq
constexpr int kBwdNX = 524;                     // x positions 4p0-11+u
constexpr int kBwdNS = 2 * kBwdTP;              // 250 owned j / owned q
// (4 * kBwdTP = 500 owned conv1 outputs i per tile)
constexpr int kBwdNIpad = 512;
constexpr int kNAcc = 53;                       // private accumulators per lane (see below)

__device__ __forceinline__ int potes_bwd_tiles(const PotesDims& d) {
  // +2: the a1 positions 2*P2 .. 2*P2+2 still receive gradient from the last pooled outputs
  return (d.P2 + 2 + kBwdTP - 1) / kBwdTP;
}

// MASK: the second layer's ReLU/pool routing comes from the forward's m2 bytes instead of being
// recomputed (no conv2 pass, one barrier phase less): dz2 = route(gh2, m2).
template <bool MASK>
__global__ __launch_bounds__(kPotThreads) void potes_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gh2, const uint8_t* __restrict__ m2,
    const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ partial /* gridDim.x * 212 */, int N, int T) {
  __shared__ PotesWeights W;
  __shared__ __align__(16) float xs[kBwdNX + 4];
  __shared__ __align__(16) float a1s[kC1 * kBwdNQ];
  __shared__ __align__(16) uint8_t sel1[kC1 * kBwdNQ];
  __shared__ __align__(16) float dz2s[kC2 * (kBwdNJ + 12)];   // +12: read window of the last lanes
  __shared__ float red[4 * kNAcc];
  constexpr int kDz2Row = kBwdNJ + 12;
  const PotesDims d = potes_dims(T);
  const int tiles = potes_bwd_tiles(d);
  const unsigned work = (unsigned)N * (unsigned)tiles;     // N <= 65535 rows, tiles < 2^15
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_weights(&W, w1, b1, w2, b2);
  for (int i = threadIdx.x; i < kC2 * kDz2Row; i += kPotThreads) dz2s[i] = 0.f;

  // Private accumulators, reduced across lanes once at the end of the kernel:
  //   wave w owns output channel co = w of conv2 and input channels ci = 2w, 2w+1 of conv1
  //   acc2[ci][k] = gw2[w][ci][k] (40), accb2 = gb2[w], acc1[c][k] = gw1[2w+c][k] (10),
  //   accb1[c] = gb1[2w+c]
  float acc2[kC1][kK], acc1[2][kK], accb1[2] = {0.f, 0.f}, accb2 = 0.f;
#pragma unroll
  for (int ci = 0; ci < kC1; ++ci)
#pragma unroll
    for (int k = 0; k < kK; ++k) acc2[ci][k] = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int k = 0; k < kK; ++k) acc1[c][k] = 0.f;

  // Software prefetch: the global loads of item k+1 (its x window and this lane's two output
  // gradients) are issued while item k is being processed, so that the ~2 us HBM/L2 latency is
  // not paid twice per item with only two resident blocks per CU.
  constexpr int kXPer = (kBwdNX + 4 + kPotThreads - 1) / kPotThreads;   // 3
  float xr[kXPer], gr[2];
  uint32_t mr[2] = {0u, 0u};
  const int m2s = (d.P2 + 3) / 4;
  auto prefetch = [&](unsigned it) {
    const int n = (int)(it / (unsigned)tiles), p0 = (int)(it - (unsigned)n * (unsigned)tiles) * kBwdTP;
    const int xlo = 2 * (2 * p0 - 5) - 1;
    const float* xrow = x + (size_t)n * T;
#pragma unroll
    for (int j = 0; j < kXPer; ++j) {
      const int u = threadIdx.x + j * kPotThreads, g = xlo + u;
      xr[j] = (u < kBwdNX + 4 && g >= 0 && g < T) ? xrow[g] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pe = p0 - 2 + 2 * lane + u;
      gr[u] = (pe >= 0 && pe < d.P2) ? gh2[((size_t)n * kC2 + wave) * d.P2 + pe] : 0.f;
      if (MASK)
        mr[u] = (pe >= 0 && pe < d.P2)
                    ? (m2[((size_t)n * kC2 + wave) * m2s + (pe >> 2)] >> (2 * (pe & 3))) & 3u
                    : 0u;
    }
  };
  if (blockIdx.x < work) prefetch(blockIdx.x);

  for (unsigned item = blockIdx.x; item < work; item += gridDim.x) {
    const int n = (int)(item / (unsigned)tiles), p0 = (int)(item - (unsigned)n * (unsigned)tiles) * kBwdTP;
    const int qlo = 2 * p0 - 5;
    (void)n;
    __syncthreads();  // previous item's LDS fully consumed
#pragma unroll
    for (int j = 0; j < kXPer; ++j) {
      const int u = threadIdx.x + j * kPotThreads;
      if (u < kBwdNX + 4) xs[u] = xr[j];
    }
    const float g_cur[2] = {gr[0], gr[1]};
    if (MASK) {   // dz2 on the extended range straight from the saved routing (wave = co)
      f4 dz;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        dz[2 * u] = mr[u] == 1u ? g_cur[u] : 0.f;
        dz[2 * u + 1] = mr[u] == 2u ? g_cur[u] : 0.f;
      }
      *reinterpret_cast<f4*>(dz2s + wave * kDz2Row + 4 * lane) = dz;
    }
    if (item + gridDim.x < work) prefetch(item + gridDim.x);
    __syncthreads();
    // Without the conv2 recompute nothing reads a1s[0..3] / sel1[0..4] (the weight gradient reads
    // a1 from index 4, the routing from index 5): 64 groups x 8 channels = exactly two rounds of
    // the block instead of two and a 8-thread third.
    layer1(W, xs, a1s, sel1, qlo, kBwdNQ, d.P1, MASK ? 1 : 0);
    __syncthreads();
    if (!MASK) {  // conv2 + ReLU + pool on the extended range -> dz2 = dL/dz2 (wave = co, 2 pooled per lane)
      const int co = __builtin_amdgcn_readfirstlane(wave);
      float aw[8], za[2], zb[2];
      za[0] = za[1] = zb[0] = zb[1] = b2[co];
#pragma unroll 2
      for (int ci = 0; ci < kC1; ++ci) {
        float w[kK];
        lds_load8(a1s + ci * kBwdNQ + 4 * lane, aw);
#pragma unroll
        for (int k = 0; k < kK; ++k) w[k] = w2[(co * kC1 + ci) * kK + k];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int k = 0; k < kK; ++k) {
            za[u] = fmaf(w[k], aw[2 * u + k], za[u]);
            zb[u] = fmaf(w[k], aw[2 * u + 1 + k], zb[u]);
          }
      }
      f4 dz;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pe = p0 - 2 + 2 * lane + u;
        float da = 0.f, db = 0.f;
        if (pe >= 0 && pe < d.P2) {
          const float g = g_cur[u];
          const float ra = fmaxf(za[u], 0.f), rb = fmaxf(zb[u], 0.f);
          if (rb > ra) db = g; else if (ra > 0.f) da = g;
        }
        dz[2 * u] = da;
        dz[2 * u + 1] = db;
      }
      *reinterpret_cast<f4*>(dz2s + co * kDz2Row + 4 * lane) = dz;
    }
    if (!MASK) __syncthreads();
    {  // back through conv2 to the owned a1 positions q = 2p0+r, then pool1/ReLU1 -> dz1
       // wave w -> channels 2w, 2w+1; lane -> r0 = 4*lane .. +3
      const int r0 = 4 * lane;
      const int ci0 = __builtin_amdgcn_readfirstlane(2 * wave);
      float da1[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
      for (int co = 0; co < kC2; ++co) {     // one dz2 window per co serves both channels
        float dw[12];
        lds_load12(dz2s + co * kDz2Row + r0, dw);   // dz2 index r+5-k, r = r0+u: r0+1 .. r0+8
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float w[kK];
#pragma unroll
          for (int k = 0; k < kK; ++k) w[k] = w2[(co * kC1 + ci0 + c) * kK + k];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < kK; ++k) da1[c][u] = fmaf(dw[u + 5 - k], w[k], da1[c][u]);
        }
      }
      // ... routed through pool1/ReLU1 to the conv1 outputs i = 4p0 + 8*lane + (0..7), and used
      // right here for gw1 / gb1: the lane that owns these a1 positions for channels 2w, 2w+1 is
      // the lane that owns the corresponding conv1 outputs — dz1 never goes through LDS (it did:
      // 16 KB per block, two 32-byte-stride accesses per lane and a barrier, for nothing).
      float xw[12];                                    // x index ii+10+k: 8*lane+10 .. 8*lane+21
      if (r0 < kBwdNS) {
#pragma unroll
        for (int u = 0; u < 12; u += 2) {
          const float2 v = *reinterpret_cast<const float2*>(xs + 2 * r0 + 10 + u);
          xw[u] = v.x;
          xw[u + 1] = v.y;
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ci = ci0 + c;
        float dd[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + u;
          const uint8_t sc = r < kBwdNS ? sel1[ci * kBwdNQ + r + 5] : 0;   // 0 outside [0,P1) too
          dd[2 * u] = sc == 1 ? da1[c][u] : 0.f;
          dd[2 * u + 1] = sc == 2 ? da1[c][u] : 0.f;
        }
        if (r0 < kBwdNS) {
          accb1[c] += ((dd[0] + dd[1]) + (dd[2] + dd[3])) + ((dd[4] + dd[5]) + (dd[6] + dd[7]));
#pragma unroll
          for (int k = 0; k < kK; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc1[c][k] = fmaf(dd[u], xw[u + k], acc1[c][k]);
        }
      }
    }
    {  // gw2 / gb2: wave = co, lane -> owned s0 = 4*lane .. +3 (s < 250)
      const int co = wave, s0 = 4 * lane;
      const f4 dv = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + s0 + 4);
      float dd[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u >= kBwdNS) dd[u] = 0.f;            // computed for the halo, owned by the next tile
      accb2 += (dd[0] + dd[1]) + (dd[2] + dd[3]);
      if (s0 < kBwdNS) {
#pragma unroll
        for (int ci = 0; ci < kC1; ++ci) {   // acc2 is indexed statically: keep fully unrolled
          float aw[8];
          lds_load8(a1s + ci * kBwdNQ + s0 + 4, aw);   // a1 index s+4+k
#pragma unroll
          for (int k = 0; k < kK; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc2[ci][k] = fmaf(dd[u], aw[u + k], acc2[ci][k]);
        }
      }
    }
  }

  // ---- one cross-lane reduction per kernel, then one partial vector per block ----------------
  float flat[kNAcc];
#pragma unroll
  for (int ci = 0; ci < kC1; ++ci)
#pragma unroll
    for (int k = 0; k < kK; ++k) flat[ci * kK + k] = acc2[ci][k];
  flat[40] = accb2;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
#pragma unroll
    for (int k = 0; k < kK; ++k) flat[41 + c * kK + k] = acc1[c][k];
    flat[51 + c] = accb1[c];
  }
#pragma unroll
  for (int e = 0; e < kNAcc; ++e) flat[e] = wave_sum_lane63(flat[e]);
  __syncthreads();
  if (lane == 63)
#pragma unroll
    for (int e = 0; e < kNAcc; ++e) red[wave * kNAcc + e] = flat[e];
  __syncthreads();
  for (int e = threadIdx.x; e < kNGrad; e += kPotThreads) {
    float v;
    if (e < kNW1) {                       // gw1[ci][k]
      const int ci = e / kK, k = e - ci * kK;
      v = red[(ci >> 1) * kNAcc + 41 + (ci & 1) * kK + k];
    } else if (e < kNW1 + kC1) {          // gb1[ci]
      const int ci = e - kNW1;
      v = red[(ci >> 1) * kNAcc + 51 + (ci & 1)];
    } else if (e < kNW1 + kC1 + kNW2) {   // gw2[co][ci][k]
      const int f = e - kNW1 - kC1;
      const int co = f / (kC1 * kK), rest = f - co * (kC1 * kK);
      v = red[co * kNAcc + rest];
    } else {                              // gb2[co]
      v = red[(e - kNW1 - kC1 - kNW2) * kNAcc + 40];
    }
    partial[(size_t)blockIdx.x * kNGrad + e] = v;
  }
}

// ---------------------------------------------------------------------------------- backward, channel pairs
// potes_bwd_fused_kernel is VALU-issue-bound (profiles/r3_potes_bwd_sq_counters.json: 31.1 M VALU
// instructions, 71 % of the launch) and of the ~455 multiply-add-related instructions a wave spends
// per item only 236 are v_pk_fma_f32 — the other 219 are v_mov_b32 that assemble even-aligned
// register pairs for operands that start at an odd offset (pairs run along POSITIONS there: a tap k
// shifts the window by k, and every odd k breaks the alignment).  Here the two halves of every
// packed multiply-add are the wave's two CHANNELS instead (wave w owns first-layer channels 2w,
// 2w+1): one operand is a channel pair that is a pair by construction — first-layer weights,
// second-layer weights (both wave-uniform: SGPR pairs), routed gradients, the a1 rows, which live
// pair-interleaved in LDS — and the other is ONE position's value, broadcast to both halves by
// op_sel from whichever half of its natural register pair it sits in.  No operand assembly at all.
// And the second-layer weight gradient uses what the routing masks say: of the two conv2 outputs
// under a pooled output only the winner carries gradient, so a lane takes ONE shifted 5-tap window
// of a1 per pooled output (the shift, 0 or 1, moves the LDS address) instead of two overlapping
// windows against a half-zero dz2 quad: half the multiply-adds of that phase.
// Per lane and item: layer 1 40, back through conv2 80, gw1 40, gw2 40 packed multiply-adds
// (fused kernel: 36 + 80 + 40 + 80 and 219 moves).  Same tiles, same ownership, same partial layout
// and epilogue as the fused kernel.  The first layer is now an in-order fmaf chain from the bias
// over k = 0..4 — bit for bit what the matrix-core forward computes (potes_fwd_mfma_kernel), so
// the recomputed ReLU/pool routing of layer 1 IS the forward's.
__device__ __forceinline__ void pkfma_lo(f2& acc, const f2& a, const f2& b) {   // acc += a * b.x
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pkfma_hi(f2& acc, const f2& a, const f2& b) {   // acc += a * b.y
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pkfma_s_lo(f2& acc, const f2& a_uniform, const f2& b) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(a_uniform), "v"(b));
}
__device__ __forceinline__ void pkfma_s_hi(f2& acc, const f2& a_uniform, const f2& b) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(a_uniform), "v"(b));
}
// acc += a * (element j of the float sequence stored as pairs p[0], p[1], ...)
template <int J, int N>
__device__ __forceinline__ void pkfma_at(f2& acc, const f2& a, const f2 (&p)[N]) {
  if (J & 1) pkfma_hi(acc, a, p[J >> 1]); else pkfma_lo(acc, a, p[J >> 1]);
}
template <int J, int N>
__device__ __forceinline__ void pkfma_s_at(f2& acc, const f2& a_uniform, const f2 (&p)[N]) {
  if (J & 1) pkfma_s_hi(acc, a_uniform, p[J >> 1]); else pkfma_s_lo(acc, a_uniform, p[J >> 1]);
}

template <int M>
struct PairConv1 {   // cp[m] += w[k] * x[m + k], k = 0..4 in order, for m = M .. 7
  template <int K>
  static __device__ __forceinline__ void taps(f2& c, const f2 (&w)[kK], const f2 (&xp)[6]) {
    if constexpr (K < kK) {
      pkfma_s_at<M + K>(c, w[K], xp);
      taps<K + 1>(c, w, xp);
    }
  }
  static __device__ __forceinline__ void run(f2 (&cp)[8], const f2 (&w)[kK], const f2 (&xp)[6]) {
    if constexpr (M < 8) {
      taps<0>(cp[M], w, xp);
      PairConv1<M + 1>::run(cp, w, xp);
    }
  }
};
template <int U>
struct PairDgrad {   // da[u] += w2pair[k] * dz2[u + 4 - k], u = U .. 3
  template <int K>
  static __device__ __forceinline__ void taps(f2& d, const f2 (&w)[kK], const f2 (&dwp)[4]) {
    if constexpr (K < kK) {
      pkfma_s_at<U + 4 - K>(d, w[K], dwp);
      taps<K + 1>(d, w, dwp);
    }
  }
  static __device__ __forceinline__ void run(f2 (&da)[4], const f2 (&w)[kK], const f2 (&dwp)[4]) {
    if constexpr (U < 4) {
      taps<0>(da[U], w, dwp);
      PairDgrad<U + 1>::run(da, w, dwp);
    }
  }
};
template <int U>
struct PairDgradV {  // the same with the weight pairs in vector registers
  template <int K>
  static __device__ __forceinline__ void taps(f2& d, const f2 (&w)[kK], const f2 (&dwp)[4]) {
    if constexpr (K < kK) {
      pkfma_at<U + 4 - K>(d, w[K], dwp);
      taps<K + 1>(d, w, dwp);
    }
  }
  static __device__ __forceinline__ void run(f2 (&da)[4], const f2 (&w)[kK], const f2 (&dwp)[4]) {
    if constexpr (U < 4) {
      taps<0>(da[U], w, dwp);
      PairDgradV<U + 1>::run(da, w, dwp);
    }
  }
};
template <int I>
struct PairGw1 {     // acc[k] += dd[i] * x[i + k], i = I .. 7
  template <int K>
  static __device__ __forceinline__ void taps(f2 (&acc)[kK], const f2& dd, const f2 (&xp)[6]) {
    if constexpr (K < kK) {
      pkfma_at<I + K>(acc[K], dd, xp);
      taps<K + 1>(acc, dd, xp);
    }
  }
  static __device__ __forceinline__ void run(f2 (&acc)[kK], const f2 (&dd)[8], const f2 (&xp)[6]) {
    if constexpr (I < 8) {
      taps<0>(acc, dd[I], xp);
      PairGw1<I + 1>::run(acc, dd, xp);
    }
  }
};

// One block barrier per item: the staging buffers (x window, routed dz2) are double-buffered, and a
// wave needs nothing from the other waves after it — it owns its channel pair through ALL phases:
// layer 1, back through conv2, gw1, and the second-layer weight gradient towards its two channels
// for all four output channels (its a1 rows cross lanes through a wave-private LDS region, which
// needs no barrier: a wave's LDS operations execute in order).  The first version of this kernel
// kept the fused kernel's three barriers per item and was exactly as slow as it (73.6 us against
// 73.8) with 44 % fewer VALU instructions: the fused kernel's "71 % VALU-issue-bound" was three
// blocks per CU each waiting at barriers most of the time, not arithmetic.
__global__ __launch_bounds__(kPotThreads, 4) void potes_bwd_pair_kernel(
    const float* __restrict__ x, const float* __restrict__ gh2, const uint8_t* __restrict__ m2,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float* __restrict__ partial /* gridDim.x * 212 */, int N, int T) {
  constexpr int kDz2Row = kBwdNJ + 12;
  constexpr int kXsLen = kBwdNX + 4;
  __shared__ __align__(16) float xs2[2][kXsLen];
  __shared__ __align__(16) float dz2s2[2][kC2 * kDz2Row];
  // a1 of the wave's channel pair, one (c0, c1) pair = 8 bytes per position, in FOUR planes by
  // position mod 4 (plane pitch kA1Plane pairs = 0 mod 32): lane l works on positions 4l + c, so for
  // any fixed c the 64 lanes read consecutive 8-byte words of one plane, and lanes whose window is
  // shifted by one (the winner of their pooled pair) read another plane at the same word offset —
  // no two lanes of a pass share a bank.  (Position-major rows, lane stride 32 bytes, cost four-way
  // conflicts on every one of the 40 window reads: 16.5 M conflict cycles, 62 % of the LDS pipe's
  // active time, profiles/r4_potes_bwd_pair_sq_counters.json "v3".)
  constexpr int kA1Plane = 96;                                  // >= kBwdNQ / 4 + 1, multiple of 32
  __shared__ __align__(16) float a1p[4 * 4 * kA1Plane * 2];     // [wave][plane][word][2]
  __shared__ float red[4 * kNAcc];
  // second-layer weights as (towards channel 2w, towards 2w+1) pairs, [wave][co][k]: 40 SGPRs per
  // wave next to the 12 of the first layer did not fit (59 scalar spills, 55 v_readlane per item);
  // from LDS they are five broadcast 8-byte reads per co
  __shared__ __align__(8) float w2ps[4 * kC2 * kK * 2];
  const PotesDims d = potes_dims(T);
  const int tiles = potes_bwd_tiles(d);
  const unsigned work = (unsigned)N * (unsigned)tiles;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 2 * kC2 * kDz2Row; i += kPotThreads) (&dz2s2[0][0])[i] = 0.f;

  // wave-uniform weight pairs (scalar loads): first-layer taps and bias of channels 2w, 2w+1, and
  // for every (co, k) the second-layer weights towards those two channels
  const int c0 = 2 * wave;
  f2 w1p[kK];
#pragma unroll
  for (int k = 0; k < kK; ++k) w1p[k] = f2{w1[c0 * kK + k], w1[(c0 + 1) * kK + k]};
  const f2 b1p = {b1[c0], b1[c0 + 1]};
  for (int i = threadIdx.x; i < 4 * kC2 * kK * 2; i += kPotThreads) {
    const int h = i & 1, k = (i >> 1) % kK, co = (i >> 1) / kK % kC2, wv = (i >> 1) / (kK * kC2);
    w2ps[i] = w2[(co * kC1 + 2 * wv + h) * kK + k];
  }

  f2 acc2p[kC2][kK], acc1p[kK], accb1p = {0.f, 0.f};   // acc2p[co][k] = (ci = 2w, ci = 2w+1)
  float accb2 = 0.f;                                    // gb2[co = wave]
#pragma unroll
  for (int co = 0; co < kC2; ++co)
#pragma unroll
    for (int k = 0; k < kK; ++k) acc2p[co][k] = f2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kK; ++k) acc1p[k] = f2{0.f, 0.f};

  // The next item's inputs are requested into registers here and consumed one item later.  Every
  // load goes to a clamped in-range address UNCONDITIONALLY and nothing is computed from a loaded
  // value before the next iteration: `cond ? p[i] : 0` compiles to a branch around the load, and
  // the routing byte's shift-and-mask right behind its load put an `s_waitcnt vmcnt` into that
  // branch — two exposed memory round trips per item, which is what bounded the fused kernel and
  // the first two versions of this one at ~5.5 us per item (three different instruction counts,
  // barrier counts and occupancies, one and the same 73 us).
  constexpr int kXPer = (kXsLen + kPotThreads - 1) / kPotThreads;   // 3
  float xr[kXPer], gr[2];
  uint32_t mb[2] = {0u, 0u};           // raw routing bytes; decoded when they are used
  int p0_pref = 0;                     // the tile the registers belong to
  const int m2s = (d.P2 + 3) / 4;
  auto prefetch = [&](unsigned it) {
    const int n = (int)(it / (unsigned)tiles), p0 = (int)(it - (unsigned)n * (unsigned)tiles) * kBwdTP;
    p0_pref = p0;
    const int xlo = 2 * (2 * p0 - 5) - 1;
    const float* xrow = x + (size_t)n * T;
#pragma unroll
    for (int j = 0; j < kXPer; ++j) {
      const int g = xlo + (int)threadIdx.x + j * kPotThreads;
      xr[j] = xrow[g < 0 ? 0 : (g >= T ? T - 1 : g)];
    }
    const float* grow = gh2 + ((size_t)n * kC2 + wave) * d.P2;
    const uint8_t* mrow = m2 + ((size_t)n * kC2 + wave) * m2s;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pe = p0 - 2 + 2 * lane + u, pc = pe < 0 ? 0 : (pe >= d.P2 ? d.P2 - 1 : pe);
      gr[u] = grow[pc];
      mb[u] = mrow[pc >> 2];
    }
  };
  if (blockIdx.x < work) prefetch(blockIdx.x);
  __syncthreads();                                   // dz2s zeroed before the first item writes it
  const int R0 = 4 * lane;                           // a1 positions r = R0 - 1 + u, u < 4
  float* const a1w = a1p + (size_t)wave * 4 * kA1Plane * 2;   // this wave's planes

  int buf = 0;
  for (unsigned item = blockIdx.x; item < work; item += gridDim.x, buf ^= 1) {
    const int p0 = (int)(item % (unsigned)tiles) * kBwdTP;
    float* const xs = xs2[buf];
    float* const dz2s = dz2s2[buf];
    // (no barrier in front: the buffers of this item were last READ two items ago, and every wave
    // has passed the previous item's barrier since)
    {
      const int xlo = 2 * (2 * p0_pref - 5) - 1;
#pragma unroll
      for (int j = 0; j < kXPer; ++j) {
        const int u = threadIdx.x + j * kPotThreads, g = xlo + u;
        if (u < kXsLen) xs[u] = (g >= 0 && g < T) ? xr[j] : 0.f;
      }
      f4 dz;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pe = p0_pref - 2 + 2 * lane + u;
        const bool in = pe >= 0 && pe < d.P2;
        const uint32_t code = in ? (mb[u] >> (2 * (pe & 3))) & 3u : 0u;
        dz[2 * u] = code == 1u ? gr[u] : 0.f;
        dz[2 * u + 1] = code == 2u ? gr[u] : 0.f;
      }
      *reinterpret_cast<f4*>(dz2s + wave * kDz2Row + 4 * lane) = dz;
    }
    if (item + gridDim.x < work) prefetch(item + gridDim.x);
    __syncthreads();
    f2 xp[6];
    {
      const f4 a = *reinterpret_cast<const f4*>(xs + 2 * R0 + 8),
               b = *reinterpret_cast<const f4*>(xs + 2 * R0 + 12),
               c = *reinterpret_cast<const f4*>(xs + 2 * R0 + 16);
      xp[0] = f2{a.x, a.y}; xp[1] = f2{a.z, a.w}; xp[2] = f2{b.x, b.y};
      xp[3] = f2{b.z, b.w}; xp[4] = f2{c.x, c.y}; xp[5] = f2{c.z, c.w};
    }
    uint32_t sel[2] = {0u, 0u};                        // 2 bits per position u, per channel
    {  // ---- layer 1 of channels 2w, 2w+1 at a1 positions r = R0 - 1 + u -> wave-private LDS rows
      f2 cp[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) cp[m] = b1p;
      PairConv1<0>::run(cp, w1p, xp);
      f2 a1v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = 2 * p0 + R0 - 1 + u, r = R0 - 1 + u;
        const bool valid = q >= 0 && q < d.P1, owned = r >= 0 && r < kBwdNS;
        // relu_pool2's rule with one v_max3 for the value (finite operands): code 2 if the second
        // candidate's ReLU is strictly larger, else 1 if the first survives its ReLU, else 0
        const float za0 = cp[2 * u].x, zb0 = cp[2 * u + 1].x, za1 = cp[2 * u].y, zb1 = cp[2 * u + 1].y;
        a1v[u].x = valid ? relu_max2(za0, zb0) : 0.f;
        a1v[u].y = valid ? relu_max2(za1, zb1) : 0.f;
        const uint32_t sc0 = (zb0 > za0 && zb0 > 0.f) ? 2u : (za0 > 0.f ? 1u : 0u);
        const uint32_t sc1 = (zb1 > za1 && zb1 > 0.f) ? 2u : (za1 > 0.f ? 1u : 0u);
        if (valid && owned) {
          sel[0] |= sc0 << (2 * u);
          sel[1] |= sc1 << (2 * u);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)                      // position R0 + 4 + u: plane u, word lane + 1
        *reinterpret_cast<f2*>(a1w + (size_t)(u * kA1Plane + lane + 1) * 2) = a1v[u];
    }
    {  // ---- back through conv2 to these positions, routed by the selectors into gw1 / gb1
      f2 da[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int co = 0; co < kC2; ++co) {
        const f4 lo = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + R0),
                 hi = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + R0 + 4);
        const f2 dwp[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
        f2 wv[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k)
          wv[k] = *reinterpret_cast<const f2*>(w2ps + ((wave * kC2 + co) * kK + k) * 2);
        PairDgradV<0>::run(da, wv, dwp);
      }
      f2 dd[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t s0 = (sel[0] >> (2 * u)) & 3u, s1 = (sel[1] >> (2 * u)) & 3u;
        dd[2 * u] = f2{s0 == 1u ? da[u].x : 0.f, s1 == 1u ? da[u].y : 0.f};
        dd[2 * u + 1] = f2{s0 == 2u ? da[u].x : 0.f, s1 == 2u ? da[u].y : 0.f};
      }
      accb1p += ((dd[0] + dd[1]) + (dd[2] + dd[3])) + ((dd[4] + dd[5]) + (dd[6] + dd[7]));
      PairGw1<0>::run(acc1p, dd, xp);
    }
    // the wave's a1 rows are complete in LDS for the wave itself (its LDS operations run in order);
    // only the compiler has to be told not to move the reads below across the stores above
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {  // ---- gw2 towards channels 2w, 2w+1: owned conv2 positions s = s0 .. s0+3, all four co.
       // a1 needed by position s0 + u, tap k: position 4 lane + 4 + u + k — eight pairs, read once
       // (plane (u + k) & 3, word lane + 1 + ((u + k) >> 2): fixed offsets, conflict-free) and used
       // by every co.  (Taking only the pooled pair's WINNER — one shifted 5-tap window per pooled
       // output — halves the multiply-adds, but the shift is per lane and per co: 40 window reads
       // with a select in every address instead of 8 fixed ones; measured 60.1 us against this.)
      const int s0 = 4 * lane;
      if (s0 < kBwdNS) {
        f2 aw[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
          aw[c] = *reinterpret_cast<const f2*>(a1w + (size_t)((c & 3) * kA1Plane + lane + 1 + (c >> 2)) * 2);
#pragma unroll
        for (int co = 0; co < kC2; ++co) {
          f4 dv = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + s0 + 4);
          if (s0 + 2 >= kBwdNS) { dv.z = 0.f; dv.w = 0.f; }     // (kBwdNS is even)
          if (co == wave) accb2 += (dv.x + dv.y) + (dv.z + dv.w);
          const f2 dp[2] = {{dv.x, dv.y}, {dv.z, dv.w}};
#pragma unroll
          for (int k = 0; k < kK; ++k) {
            pkfma_lo(acc2p[co][k], aw[k], dp[0]);
            pkfma_hi(acc2p[co][k], aw[k + 1], dp[0]);
            pkfma_lo(acc2p[co][k], aw[k + 2], dp[1]);
            pkfma_hi(acc2p[co][k], aw[k + 3], dp[1]);
          }
        }
      }
    }
  }

  float flat[kNAcc];   // [co*10 + c*5 + k] gw2 towards channel 2w+c | [40] gb2[w] | gw1 | gb1
#pragma unroll
  for (int co = 0; co < kC2; ++co)
#pragma unroll
    for (int k = 0; k < kK; ++k) {
      flat[co * 10 + k] = acc2p[co][k].x;
      flat[co * 10 + kK + k] = acc2p[co][k].y;
    }
  flat[40] = accb2;
#pragma unroll
  for (int k = 0; k < kK; ++k) {
    flat[41 + k] = acc1p[k].x;
    flat[41 + kK + k] = acc1p[k].y;
  }
  flat[51] = accb1p.x;
  flat[52] = accb1p.y;
#pragma unroll
  for (int e = 0; e < kNAcc; ++e) flat[e] = wave_sum_lane63(flat[e]);
  __syncthreads();
  if (lane == 63)
#pragma unroll
    for (int e = 0; e < kNAcc; ++e) red[wave * kNAcc + e] = flat[e];
  __syncthreads();
  for (int e = threadIdx.x; e < kNGrad; e += kPotThreads) {
    float v;
    if (e < kNW1) {
      const int ci = e / kK, k = e - ci * kK;
      v = red[(ci >> 1) * kNAcc + 41 + (ci & 1) * kK + k];
    } else if (e < kNW1 + kC1) {
      const int ci = e - kNW1;
      v = red[(ci >> 1) * kNAcc + 51 + (ci & 1)];
    } else if (e < kNW1 + kC1 + kNW2) {
      const int f = e - kNW1 - kC1;
      const int co = f / (kC1 * kK), rest = f - co * (kC1 * kK);
      const int ci = rest / kK, k = rest - ci * kK;
      v = red[(ci >> 1) * kNAcc + co * 10 + (ci & 1) * kK + k];
    } else {
      v = red[(e - kNW1 - kC1 - kNW2) * kNAcc + 40];
    }
    partial[(size_t)blockIdx.x * kNGrad + e] = v;
  }
}

// ---------------------------------------------------------------------------------- input gradient
// dL/dx for saliency maps (saliency.py:52-61 differentiates the class score w.r.t. the INPUT).
// Tile g owns the input positions u = 4*p0 + v, v < 4*TP (p0 = g*TP, TP = 124).  dL/dx[u] needs
// dz1 at i = u-3 .. u+1, hence dL/da1 at q = 2p0-2 .. 2p0+2TP, hence dz2 at j = 2p0-5 .. 2p0+2TP+1,
// i.e. the pooled outputs p0-3 .. p0+TP (128 values): the same extended sizes as the weight-gradient
// kernel with every origin shifted by one pooled position.
constexpr int kInTP = 124;
constexpr int kInNR = 2 * kInTP + 3;            // 251 a1 positions with a complete gradient
constexpr int kInNU = 4 * kInTP;                // 496 owned inputs

__global__ __launch_bounds__(kPotThreads) void potes_input_grad_kernel(
    const float* __restrict__ x, const float* __restrict__ gh2, const float* __restrict__ w1,
    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ gx, int N, int T) {
  __shared__ PotesWeights W;
  __shared__ __align__(16) float xs[kBwdNX + 4];
  __shared__ __align__(16) float a1s[kC1 * kBwdNQ];
  __shared__ __align__(16) uint8_t sel1[kC1 * kBwdNQ];
  constexpr int kDz2Row = kBwdNJ + 12;
  __shared__ __align__(16) float dz2s[kC2 * kDz2Row];
  __shared__ __align__(16) float dz1s[kC1 * kBwdNIpad];
  const PotesDims d = potes_dims(T);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.y, p0 = blockIdx.x * kInTP;
  const int qlo = 2 * p0 - 7, xlo = 2 * qlo - 1;      // a1 origin, x origin (4p0 - 15)
  load_weights(&W, w1, b1, w2, b2);
  for (int i = threadIdx.x; i < kC2 * kDz2Row; i += kPotThreads) dz2s[i] = 0.f;
  stage_x(xs, x + (size_t)n * T, xlo, kBwdNX + 4, T);
  __syncthreads();
  layer1(W, xs, a1s, sel1, qlo, kBwdNQ, d.P1);
  __syncthreads();
  {  // conv2 + ReLU + pool on pe = p0-3+pp, pp < 128 -> dz2 (wave = co, 2 pooled per lane)
    const int co = __builtin_amdgcn_readfirstlane(wave);
    float aw[8], za[2], zb[2];
    za[0] = za[1] = zb[0] = zb[1] = b2[co];
#pragma unroll
    for (int ci = 0; ci < kC1; ++ci) {
      float w[kK];
      lds_load8(a1s + ci * kBwdNQ + 4 * lane, aw);
#pragma unroll
      for (int k = 0; k < kK; ++k) w[k] = w2[(co * kC1 + ci) * kK + k];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          za[u] = fmaf(w[k], aw[2 * u + k], za[u]);
          zb[u] = fmaf(w[k], aw[2 * u + 1 + k], zb[u]);
        }
    }
    f4 dz;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pe = p0 - 3 + 2 * lane + u;
      float da = 0.f, db = 0.f;
      if (pe >= 0 && pe < d.P2) {
        const float g = gh2[((size_t)n * kC2 + co) * d.P2 + pe];
        const float ra = fmaxf(za[u], 0.f), rb = fmaxf(zb[u], 0.f);
        if (rb > ra) db = g; else if (ra > 0.f) da = g;
      }
      dz[2 * u] = da;
      dz[2 * u + 1] = db;
    }
    *reinterpret_cast<f4*>(dz2s + co * kDz2Row + 4 * lane) = dz;
  }
  __syncthreads();
  {  // dL/da1 at q = 2p0-2+r (r < 251), through pool1/ReLU1 -> dz1 at i = 4p0-4+2r(+1)
    const int r0 = 4 * lane;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ci = __builtin_amdgcn_readfirstlane(2 * wave + c);
      float da1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int co = 0; co < kC2; ++co) {
        float dw[12], w[kK];
        lds_load12(dz2s + co * kDz2Row + r0, dw);   // dz2 index r+5-k
#pragma unroll
        for (int k = 0; k < kK; ++k) w[k] = w2[(co * kC1 + ci) * kK + k];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < kK; ++k) da1[u] = fmaf(dw[u + 5 - k], w[k], da1[u]);
      }
      float out[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u;
        const uint8_t sc = r < kInNR ? sel1[ci * kBwdNQ + r + 5] : 0;
        out[2 * u] = sc == 1 ? da1[u] : 0.f;
        out[2 * u + 1] = sc == 2 ? da1[u] : 0.f;
      }
      f4* dst = reinterpret_cast<f4*>(dz1s + ci * kBwdNIpad + 8 * lane);
      dst[0] = f4{out[0], out[1], out[2], out[3]};
      dst[1] = f4{out[4], out[5], out[6], out[7]};
    }
  }
  __syncthreads();
  {  // transposed conv1: dx[u] = sum_ci sum_k dz1[ci][u+1-k] * w1[ci][k]; dz1 index v+5-k
    const int v0 = 2 * threadIdx.x;
    if (v0 < kInNU) {
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int ci = 0; ci < kC1; ++ci) {
        float dw[6], w[kK];
#pragma unroll
        for (int j = 0; j < 6; j += 2) {             // dz1 index v0+1 .. v0+6 (8-byte aligned + 1)
          dw[j] = dz1s[ci * kBwdNIpad + v0 + 1 + j];
          dw[j + 1] = dz1s[ci * kBwdNIpad + v0 + 2 + j];
        }
#pragma unroll
        for (int k = 0; k < kK; ++k) w[k] = W.w1[ci * kK + k];
#pragma unroll
        for (int k = 0; k < kK; ++k) {
          acc0 = fmaf(dw[4 - k], w[k], acc0);        // v0+5-k   -> offset 4-k from v0+1
          acc1 = fmaf(dw[5 - k], w[k], acc1);        // v0+1+5-k
        }
      }
      const int u = 4 * p0 + v0;
      float* dst = gx + (size_t)n * T + u;
      if (u < T) dst[0] = acc0;
      if (u + 1 < T) dst[1] = acc1;
    }
  }
}

// The same gradient from the forward's saved routing (m2, s1) instead of a recomputed forward:
// neither x nor the activations are needed — dz2 = route(gh2, m2), dL/da1 = conv2^T dz2,
// dz1 = route(dL/da1, s1), dL/dx = conv1^T dz1.  Same tile geometry.  The packed multiply-adds run
// over CHANNEL pairs (the weight-gradient kernel's recipe, potes_bwd_pair_kernel; the round-2/3
// kernel with scalar multiply-adds over positions, potes_input_grad_mask_kernel, 44.5 us against
// 30.6, is in the history at 2fa984b): wave w back-propagates through conv2
// towards channels 2w, 2w+1 with the two channels as the halves of every v_pk_fma_f32 (weights as
// (c0, c1) pairs — broadcast 8-byte LDS reads —, the dz2 value broadcast by op_sel), the routed
// first-layer gradient goes to LDS pair-interleaved, and the transposed first layer multiplies
// genuine pairs: (dz1[c0][i], dz1[c1][i]) x (w1[c0][k], w1[c1][k]), the two halves added once at
// the end.  120 packed multiply-adds per lane instead of 240 scalar ones.  dz1 lives in eight
// planes by position mod 8 (pitch 72 words = 8 mod 32): the stores (lane l, element i -> plane i,
// word l) and the reads (thread t, position 2t + 1 + j -> four planes x eight words per 32 lanes)
// are bank-conflict free, and a position is one 8-byte word wherever it is, so a thread can own the
// ALIGNED outputs 2t, 2t+1 (one 8-byte global store).  All global loads are unconditional, from
// clamped addresses, and decoded after the barrier that already waits for them.
// Persistent blocks (item = (row, tile), strided): the next item's bytes are requested into
// registers before the current item's first barrier and decoded after its last one.  Two barriers
// per item are inherent (dz2 crosses waves into the conv2 back-propagation, dz1 crosses waves into
// the transposed first layer), and with two of them every LDS buffer's writers and readers are
// already a barrier apart in both directions: no third one, no double buffering.
constexpr int kInPlane = 68;     // = 4 mod 16: planes p, p+2, p+4, p+6 start 16 banks apart
__global__ __launch_bounds__(kPotThreads) void potes_input_grad_pair_kernel(
    const float* __restrict__ gh2, const uint8_t* __restrict__ m2, const uint8_t* __restrict__ s1,
    const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ gx, int N,
    int T) {
  constexpr int kDz2Row = kBwdNJ + 12;
  __shared__ __align__(16) float dz2s[kC2 * kDz2Row];
  __shared__ __align__(16) float dz1p[4 * 8 * kInPlane * 2];   // [channel pair][plane][word][2]
  __shared__ __align__(8) float w2ps[4 * kC2 * kK * 2];        // [wave][co][k][(2w, 2w+1)]
  const PotesDims d = potes_dims(T);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles = (T + kInNU - 1) / kInNU;
  const unsigned work = (unsigned)N * (unsigned)tiles;
  const int m2s = (d.P2 + 3) / 4, s1row = potes_s1_row_bytes(d);
  for (int i = threadIdx.x; i < 4 * kC2 * kK * 2; i += kPotThreads) {
    const int h = i & 1, k = (i >> 1) % kK, co = (i >> 1) / kK % kC2, wv = (i >> 1) / (kK * kC2);
    w2ps[i] = w2[(co * kC1 + 2 * wv + h) * kK + k];
  }
  for (int i = threadIdx.x; i < kC2 * 12; i += kPotThreads)     // the rows' 12-float pads stay zero
    dz2s[(i / 12) * kDz2Row + kBwdNJ + i % 12] = 0.f;
  // first-layer weight pairs of all four channel pairs: block-uniform, scalar registers
  f2 w1p[4][kK];
#pragma unroll
  for (int cp = 0; cp < 4; ++cp)
#pragma unroll
    for (int k = 0; k < kK; ++k) w1p[cp][k] = f2{w1[(2 * cp) * kK + k], w1[(2 * cp + 1) * kK + k]};

  // ---- what an item reads from memory: unconditional loads from clamped addresses, decoded later
  uint32_t sb[2][2];                       // first-layer selector bytes b0 = b1 - 1, b1 per channel
  float g[2];
  uint32_t mbyte[2];
  int n_pref = 0, p0_pref = 0;
  auto prefetch = [&](unsigned it) {
    const int n = (int)(it / (unsigned)tiles), p0 = (int)(it - (unsigned)n * (unsigned)tiles) * kInTP;
    n_pref = n;
    p0_pref = p0;
    const int sb1 = (p0 >> 1) + lane, sb0 = sb1 - 1;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const uint8_t* row = s1 + ((size_t)n * kC1 + 2 * wave + c) * s1row;
      sb[c][0] = row[sb0 < 0 ? 0 : (sb0 >= s1row ? s1row - 1 : sb0)];
      sb[c][1] = row[sb1 < 0 ? 0 : (sb1 >= s1row ? s1row - 1 : sb1)];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pe = p0 - 3 + 2 * lane + u, pc = pe < 0 ? 0 : (pe >= d.P2 ? d.P2 - 1 : pe);
      g[u] = gh2[((size_t)n * kC2 + wave) * d.P2 + pc];
      mbyte[u] = m2[((size_t)n * kC2 + wave) * m2s + (pc >> 2)];
    }
  };
  if (blockIdx.x < work) prefetch(blockIdx.x);

  for (unsigned item = blockIdx.x; item < work; item += gridDim.x) {
    const int n = n_pref, p0 = p0_pref;
    {  // dz2 at pe = p0-3+2*lane+u (wave = co)
      f4 dz;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pe = p0 - 3 + 2 * lane + u;
        const uint32_t code = (pe >= 0 && pe < d.P2) ? (mbyte[u] >> (2 * (pe & 3))) & 3u : 0u;
        dz[2 * u] = code == 1u ? g[u] : 0.f;
        dz[2 * u + 1] = code == 2u ? g[u] : 0.f;
      }
      *reinterpret_cast<f4*>(dz2s + wave * kDz2Row + 4 * lane) = dz;
    }
    // first-layer selectors of this lane's 4 positions q = 2p0-2+4*lane+u for its two channels
    // (packed: position q sits in bits 2*((q+1)&3) of byte (q+1)>>2; q+1 = 2*p0 - 1 + 4*lane + u
    // with 2*p0 a multiple of 4, so u = 0 is the top pair of one byte and u = 1..3 the low pairs
    // of the next)
    uint32_t sc[2][4];
    {
      const int sb1 = (p0 >> 1) + lane, sb0 = sb1 - 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t v0 = (sb0 >= 0 && sb0 < s1row) ? sb[c][0] : 0u;
        const uint32_t v1 = (sb1 >= 0 && sb1 < s1row) ? sb[c][1] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = 4 * lane + u, q = 2 * p0 - 2 + r;
          const uint32_t code = u == 0 ? (v0 >> 6) & 3u : (v1 >> (2 * (u - 1))) & 3u;
          sc[c][u] = (r < kInNR && q >= 0 && q < d.P1) ? code : 0u;
        }
      }
    }
    if (item + gridDim.x < work) prefetch(item + gridDim.x);
    __syncthreads();
    {  // dL/da1 of channels 2w, 2w+1 at q = 2p0-2+r, r = 4 lane + u, routed -> dz1 at index 2r (+1)
      const int r0 = 4 * lane;
      f2 da[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int co = 0; co < kC2; ++co) {
        const f4 a = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + r0),
                 b = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + r0 + 4),
                 c = *reinterpret_cast<const f4*>(dz2s + co * kDz2Row + r0 + 8);
        const f2 dwp[6] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}, {c.x, c.y}, {c.z, c.w}};
        f2 wv[kK];
#pragma unroll
        for (int k = 0; k < kK; ++k)
          wv[k] = *reinterpret_cast<const f2*>(w2ps + ((wave * kC2 + co) * kK + k) * 2);
        // da[u] += wv[k] * dz2[u + 5 - k]
        pkfma_at<5>(da[0], wv[0], dwp); pkfma_at<4>(da[0], wv[1], dwp); pkfma_at<3>(da[0], wv[2], dwp);
        pkfma_at<2>(da[0], wv[3], dwp); pkfma_at<1>(da[0], wv[4], dwp);
        pkfma_at<6>(da[1], wv[0], dwp); pkfma_at<5>(da[1], wv[1], dwp); pkfma_at<4>(da[1], wv[2], dwp);
        pkfma_at<3>(da[1], wv[3], dwp); pkfma_at<2>(da[1], wv[4], dwp);
        pkfma_at<7>(da[2], wv[0], dwp); pkfma_at<6>(da[2], wv[1], dwp); pkfma_at<5>(da[2], wv[2], dwp);
        pkfma_at<4>(da[2], wv[3], dwp); pkfma_at<3>(da[2], wv[4], dwp);
        pkfma_at<8>(da[3], wv[0], dwp); pkfma_at<7>(da[3], wv[1], dwp); pkfma_at<6>(da[3], wv[2], dwp);
        pkfma_at<5>(da[3], wv[3], dwp); pkfma_at<4>(da[3], wv[4], dwp);
      }
      float* base = dz1p + ((size_t)wave * 8 * kInPlane + lane) * 2;   // element i -> plane i, word lane
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f2 first = {sc[0][u] == 1u ? da[u].x : 0.f, sc[1][u] == 1u ? da[u].y : 0.f};
        const f2 second = {sc[0][u] == 2u ? da[u].x : 0.f, sc[1][u] == 2u ? da[u].y : 0.f};
        *reinterpret_cast<f2*>(base + (size_t)(2 * u) * kInPlane * 2) = first;
        *reinterpret_cast<f2*>(base + (size_t)(2 * u + 1) * kInPlane * 2) = second;
      }
    }
    __syncthreads();
    {  // transposed conv1: dx[v] = sum_ci sum_k dz1[ci][v+5-k] * w1[ci][k]; thread t owns v = 2t, 2t+1
       // and reads dz1 positions 2t+1 .. 2t+6 of every channel pair
      const int t = threadIdx.x, v0 = 2 * t;
      if (v0 < kInNU) {
        int off[6];                                  // word offsets (in floats) inside a pair's planes
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int pos = v0 + 1 + j;
          off[j] = ((pos & 7) * kInPlane + (pos >> 3)) * 2;
        }
        f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
          f2 dp[6];
#pragma unroll
          for (int j = 0; j < 6; ++j)
            dp[j] = *reinterpret_cast<const f2*>(dz1p + (size_t)cp * 8 * kInPlane * 2 + off[j]);
#pragma unroll
          for (int k = 0; k < kK; ++k) {
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "s"(w1p[cp][k]), "v"(dp[4 - k]));
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "s"(w1p[cp][k]), "v"(dp[5 - k]));
          }
        }
        const int u = 4 * p0 + v0;
        float* dst = gx + (size_t)n * T + u;
        const float o0 = acc0.x + acc0.y, o1 = acc1.x + acc1.y;
        if (u + 1 < T && !(reinterpret_cast<uintptr_t>(dst) & 7)) {
          __builtin_nontemporal_store(f2{o0, o1}, reinterpret_cast<f2*>(dst));
        } else {
          if (u < T) dst[0] = o0;
          if (u + 1 < T) dst[1] = o1;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- dimreduc
// z[B][O] = h[B][K] . W[O][K]^T + bias for a SKINNY output (O <= 32; the Potes head is 19968 -> 20,
// models.py:376).  hipBLASLt runs this shape as 32 workgroups with no split-K: 53 us at bs=256
// for a 20 MB read.  The first version here (VALU dot products against an LDS copy of W, 16 rows x
// 1024 columns per block) ran 18 us, bound by ds_read_b128 of W and instruction issue.  This one
// uses the f32-input matrix instruction, v_mfma_f32_32x32x2_f32 (exact f32, same peak as the f32
// VALU but none of its issue slots and no LDS in the inner loop):
//   D[o][b] += W[o][k] * h[b][k]      A = W (rows o >= O are zero lanes), B = h^T, 32 x 32 x 2
// Block = 32 batch rows x one 1024-wide K chunk, 4 waves x 256 columns.  Per 64 columns a lane
// (r = lane & 31, half = lane >> 5) takes the 32 consecutive floats h[row r][k0 + 32 half ..]
// and the same span of W[r][..] (through LDS, see the kernel); MFMA step j multiplies element
// j of both (the k index may be permuted freely inside a reduction as long as A and B agree).
// The four waves' 32x32 tiles are added in a fixed order through LDS; per-chunk partials are
// summed in a fixed order by the consumer (deterministic).
constexpr int kSkinnyMaxO = 32;
constexpr int kSkinnyRows = 32;     // batch rows per block
constexpr int kSkinnyChunk = 512;   // K elements per block (one partial per chunk)
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kSkinnyWaves = 4;     // 128 columns per wave, in steps of 64
constexpr int kSkStep = 64;         // columns per MFMA round
constexpr int kSkStride = 68;       // LDS row stride in floats (16-byte aligned, +4 against banks)

// The matrix instruction wants lane = (row, k-half): read straight from memory that is 32-byte
// pieces of 32 rows per request (14.5 us).  So each wave loads its 32 x 64 tile of h and O x 64
// tile of W row-contiguously (4 rows x 256 B per request), parks them in LDS and reads them back
// in operand order; the next step's global loads are in flight while the MFMAs run.
// MASK: h is the feature matrix BEFORE Dropout(p1); the dropout is applied while the tile is
// parked in LDS: element e = b*K + k owns `bits` (1, 2, 4 or 8) consecutive random bits of `mask`
// (bit offset e*bits), kept iff their value >= thr, kept values times `scale`.  Spares the
// separate dropout pass (a 20 MB write and re-read at bs=256).
// BITS: 0 = the mask's bits per element is the run-time argument; 2 = compile-time (Dropout(.25),
// the reference's value: constant shifts and masks in the decode).
template <int O, bool MASK, int BITS = 0>
__global__ __launch_bounds__(kSkinnyWaves * 64) void skinny_linear_partial_kernel(
    const float* __restrict__ h, const float* __restrict__ W, float* __restrict__ partial, int B,
    int K, const uint8_t* __restrict__ mask, float scale, int thr, int bits_rt) {
  const int bits = BITS ? BITS : bits_rt;
  constexpr int kWRows = (O + 3) / 4 * 4;                                // W tile rows in LDS
  constexpr int kWaveFloats = (32 + kWRows) * kSkStride;
  __shared__ __align__(16) float smem[kSkinnyWaves * kWaveFloats];
  constexpr int kPerWave = kSkinnyChunk / kSkinnyWaves;                  // 256 columns
  constexpr int kSteps = kPerWave / kSkStep;                             // 4
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int row_base = blockIdx.x * kSkinnyRows, ks = blockIdx.y;
  const int k_w = ks * kSkinnyChunk + wave * kPerWave;
  float* xs = smem + wave * kWaveFloats;            // [32][kSkStride]
  float* ws = xs + 32 * kSkStride;                  // [kWRows][kSkStride]
  // loader mapping: request `it` covers rows 4 it + (lane >> 4), float4 column lane & 15
  const int lr = lane >> 4, lc = 4 * (lane & 15);
  f4 gx[8], gw[kWRows / 4];
  uint32_t gm[8];
  auto fetch = [&](int step) {
    const int k = k_w + kSkStep * step + lc;
    const bool ok = k < K;                          // K % 4 == 0: a float4 is inside or outside
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = row_base + 4 * it + lr;
      const size_t e = (size_t)(row < B ? row : B - 1) * K + (ok ? k : 0);
      const f4 v = *reinterpret_cast<const f4*>(h + e);
      gx[it] = ok ? v : z4;
      if (MASK) {     // the 4*bits random bits of this float4 (e % 4 == 0: byte- or word-aligned)
        const size_t bit = e * (size_t)bits;
        gm[it] = bits == 8   ? *reinterpret_cast<const uint32_t*>(mask + e)
                 : bits == 4 ? (uint32_t)*reinterpret_cast<const uint16_t*>(mask + (bit >> 3))
                             : (uint32_t)mask[bit >> 3] >> (bit & 7);
      }
    }
#pragma unroll
    for (int it = 0; it < kWRows / 4; ++it) {
      const int o = 4 * it + lr;
      const f4 v = *reinterpret_cast<const f4*>(W + (size_t)(o < O ? o : 0) * K + (ok ? k : 0));
      gw[it] = (ok && o < O) ? v : z4;
    }
  };
  fetch(0);
  f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 1
  for (int step = 0; step < kSteps; ++step) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      f4 v = gx[it];
      if (MASK) {
        const uint32_t m = gm[it], fm = (1u << bits) - 1u;
        v.x = (int)(m & fm) >= thr ? v.x * scale : 0.f;
        v.y = (int)((m >> bits) & fm) >= thr ? v.y * scale : 0.f;
        v.z = (int)((m >> (2 * bits)) & fm) >= thr ? v.z * scale : 0.f;
        v.w = (int)((m >> (3 * bits)) & fm) >= thr ? v.w * scale : 0.f;
      }
      *reinterpret_cast<f4*>(xs + (4 * it + lr) * kSkStride + lc) = v;
    }
#pragma unroll
    for (int it = 0; it < kWRows / 4; ++it)
      *reinterpret_cast<f4*>(ws + (4 * it + lr) * kSkStride + lc) = gw[it];
    if (step + 1 < kSteps) fetch(step + 1);
    __syncthreads();                                // tiles complete (all waves run in step)
    f4 xa[8], wa[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      xa[q] = *reinterpret_cast<const f4*>(xs + r * kSkStride + 32 * half + 4 * q);
      const f4 z4 = {0.f, 0.f, 0.f, 0.f};
      wa[q] = r < O ? *reinterpret_cast<const f4*>(ws + (r < O ? r : 0) * kSkStride + 32 * half + 4 * q)
                    : z4;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q].x, xa[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q].y, xa[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q].z, xa[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q].w, xa[q].w, acc, 0, 0, 0);
    }
    __syncthreads();                                // tiles consumed before they are overwritten
  }
  // fixed-order sum of the waves' tiles (reusing the staging memory), then one partial per block
  float* red = smem;                                // [kSkinnyWaves - 1][16][64]
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[((wave - 1) * 16 + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < kSkinnyWaves - 1; ++w)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += red[(w * 16 + i) * 64 + lane];
  // C/D layout: column = lane & 31 (batch row), row o = (i & 3) + 8 (i >> 2) + 4 half
  const int row = row_base + r;
  if (row < B) {
    float* dst = partial + ((size_t)ks * B + row) * O;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = (i & 3) + 8 * (i >> 2) + 4 * half;
      if (o < O) dst[o] = acc[i];
    }
  }
}

__global__ void skinny_linear_reduce_kernel(const float* __restrict__ partial,
                                            const float* __restrict__ bias, float* __restrict__ z,
                                            int B, int O, int KS) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * O) return;
  float v = bias ? bias[i % O] : 0.f;
  for (int ks = 0; ks < KS; ++ks) v += partial[(size_t)ks * B * O + i];
  z[i] = v;
}

// Sum the per-block partial vectors in a fixed order: grads[e] = sum_g partial[g][e].
// Column e of the G partial rows, summed by one 256-thread block in a fixed order (thread t takes
// rows t, t + 256, ...; then a tree over the threads): the value is in red[0] for thread 0.
__device__ __forceinline__ float potes_reduce_column(const float* __restrict__ partial, int G, int e,
                                                     float* red) {
  float a = 0.f;
  for (int g = threadIdx.x; g < G; g += kPotThreads) a += partial[(size_t)g * kNGrad + e];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = kPotThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  return red[0];
}

__global__ __launch_bounds__(kPotThreads) void potes_reduce_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ grads,
                                                                   int G) {
  __shared__ float red[kPotThreads];
  const float v = potes_reduce_column(partial, G, blockIdx.x, red);
  if (threadIdx.x == 0) grads[blockIdx.x] = v;
}

// ---------------------------------------------------------------------------------- optimiser
// clip_grad_value_ + Adam (L2 weight decay) for one parameter tensor in one pass
// (train_model.py:557-558, 404-407, 566).  torch's foreach/fused Adam launches 512-thread blocks
// per 65536-element chunk: the 400k-element `dimreduc.weight` gets 7 blocks (41 us), and value
// clipping is two more foreach launches.  Same update rule as torch.optim.Adam:
//   g = clamp(g, -clip, clip) + wd * p;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p,
                                                        const float* __restrict__ g,
                                                        float* __restrict__ m,
                                                        float* __restrict__ v, long long n,
                                                        float clip, float wd, float one_m_b1,
                                                        float b2, float one_m_b2, float step_size,
                                                        float inv_bc2_sqrt, float eps) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float pi = p[i];
    gi = fmaf(wd, pi, gi);
    float mi = m[i], vi = v[i];
    mi = fmaf(one_m_b1, gi - mi, mi);                 // exp_avg.lerp_(grad, 1 - beta1)
    vi = fmaf(one_m_b2 * gi, gi, b2 * vi);            // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// All parameter tensors of a model in ONE launch: eight ~3 us launches (six of them on tensors of
// <= 160 elements) become one.  The tensor table travels by value in the kernel arguments; a block
// owns kAdamEPB consecutive elements of one tensor and finds it by a uniform scan of blk_start.
constexpr int kAdamMaxTensors = 32;
constexpr int kAdamEPB = 1024;
struct AdamTable {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  long long n[kAdamMaxTensors];
  int blk_start[kAdamMaxTensors + 1];
  int count;
};

// hyper != nullptr: the eight scalars are read from device memory instead (clip, wd, 1-b1, b2,
// 1-b2, step_size, 1/sqrt(bc2), eps — pcgmix_adam_hyper's layout): a launch captured in a
// hipGraph then follows OneCycleLR's lr/beta1 and the bias corrections from replay to replay.
// partial != nullptr: the launch carries kNGrad EXTRA blocks behind the table's own — block
// red_first + e sums column e of the conv stack's per-block gradient partials exactly as
// potes_reduce_kernel does (same order, same bits), stores it at grads[e] (what the parameters'
// .grad tensors alias) and applies the update to the one element it belongs to: the tensor of the
// table whose gradient pointer lies inside grads[0 .. kNGrad) (such tensors own no blocks of the
// table).  One launch instead of two at the end of a captured training step.
__global__ __launch_bounds__(256) void adam_clip_multi_kernel(AdamTable tab, float clip, float wd,
                                                              float one_m_b1, float b2,
                                                              float one_m_b2, float step_size,
                                                              float inv_bc2_sqrt, float eps,
                                                              const float* __restrict__ hyper,
                                                              const float* __restrict__ partial,
                                                              float* __restrict__ grads, int G,
                                                              int red_first) {
  if (hyper) {
    clip = hyper[0]; wd = hyper[1]; one_m_b1 = hyper[2]; b2 = hyper[3];
    one_m_b2 = hyper[4]; step_size = hyper[5]; inv_bc2_sqrt = hyper[6]; eps = hyper[7];
  }
  if (partial && (int)blockIdx.x >= red_first) {
    __shared__ float red[kPotThreads];
    const int e = (int)blockIdx.x - red_first;
    float gi = potes_reduce_column(partial, G, e, red);
    if (threadIdx.x != 0) return;
    grads[e] = gi;
    const float* ge = grads + e;
    for (int t = 0; t < tab.count; ++t) {
      if (ge >= tab.g[t] && ge < tab.g[t] + tab.n[t]) {
        const long long i = ge - tab.g[t];
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        const float pi = tab.p[t][i];
        gi = fmaf(wd, pi, gi);
        float mi = tab.m[t][i], vi = tab.v[t][i];
        mi = fmaf(one_m_b1, gi - mi, mi);
        vi = fmaf(one_m_b2 * gi, gi, b2 * vi);
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        tab.p[t][i] = pi - step_size * (mi / denom);
        tab.m[t][i] = mi;
        tab.v[t][i] = vi;
        return;
      }
    }
    return;
  }
  int t = 0;
  while (t + 1 < tab.count && (int)blockIdx.x >= tab.blk_start[t + 1]) ++t;
  float* __restrict__ p = tab.p[t];
  const float* __restrict__ g = tab.g[t];
  float* __restrict__ m = tab.m[t];
  float* __restrict__ v = tab.v[t];
  const long long n = tab.n[t];
  const long long base = (long long)((int)blockIdx.x - tab.blk_start[t]) * kAdamEPB;
#pragma unroll
  for (int j = 0; j < kAdamEPB / 256; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i >= n) break;
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float pi = p[i];
    gi = fmaf(wd, pi, gi);
    float mi = m[i], vi = v[i];
    mi = fmaf(one_m_b1, gi - mi, mi);
    vi = fmaf(one_m_b2 * gi, gi, b2 * vi);
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

}  // namespace pcgmix

extern "C" int pcgmix_adam_hyper(float clip, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, long long step, float* out8) {
  if (!out8 || step < 1) return hipErrorInvalidValue;
  // bias corrections in float64 on the host, as torch computes them from Python floats
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  out8[0] = clip;
  out8[1] = weight_decay;
  out8[2] = 1.0f - beta1;
  out8[3] = beta2;
  out8[4] = 1.0f - beta2;
  out8[5] = (float)((double)lr / bc1);
  out8[6] = (float)(1.0 / std::sqrt(bc2));
  out8[7] = eps;
  return hipSuccess;
}

static int adam_multi_launch(int n_tensors, float* const* p, const float* const* g, float* const* m,
                             float* const* v, const long long* n, const float* h8,
                             const float* hyper_dev, hipStream_t stream,
                             const float* partial = nullptr, float* grads = nullptr, int G = 0);

extern "C" int pcgmix_adam_clip_multi_dev_f32(int n_tensors, float* const* p, const float* const* g,
                                              float* const* m, float* const* v, const long long* n,
                                              const float* hyper_dev, pcgmix_stream_t stream) {
  if (n_tensors < 0 || !hyper_dev || (n_tensors > 0 && (!p || !g || !m || !v || !n)))
    return hipErrorInvalidValue;
  const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return adam_multi_launch(n_tensors, p, g, m, v, n, zero8, hyper_dev,
                           reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pcgmix_adam_clip_multi_reduce_dev_f32(int n_tensors, float* const* p,
                                                     const float* const* g, float* const* m,
                                                     float* const* v, const long long* n,
                                                     const float* hyper_dev, const float* partial,
                                                     float* grads, int G, pcgmix_stream_t stream) {
  if (n_tensors <= 0 || !hyper_dev || !p || !g || !m || !v || !n || !partial || !grads || G <= 0)
    return hipErrorInvalidValue;
  const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return adam_multi_launch(n_tensors, p, g, m, v, n, zero8, hyper_dev,
                           reinterpret_cast<hipStream_t>(stream), partial, grads, G);
}

extern "C" int pcgmix_potes_reduce_f32(const float* partial, float* grads, int G, pcgmix_stream_t stream) {
  if (!partial || !grads || G <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pcgmix::potes_reduce_kernel, dim3(pcgmix::kNGrad), dim3(pcgmix::kPotThreads), 0,
                     reinterpret_cast<hipStream_t>(stream), partial, grads, G);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_adam_clip_multi_f32(int n_tensors, float* const* p, const float* const* g,
                                          float* const* m, float* const* v, const long long* n,
                                          float clip, float lr, float beta1, float beta2, float eps,
                                          float weight_decay, long long step,
                                          pcgmix_stream_t stream) {
  if (n_tensors < 0 || step < 1 || (n_tensors > 0 && (!p || !g || !m || !v || !n)))
    return hipErrorInvalidValue;
  float h8[8];
  pcgmix_adam_hyper(clip, lr, beta1, beta2, eps, weight_decay, step, h8);
  return adam_multi_launch(n_tensors, p, g, m, v, n, h8, nullptr,
                           reinterpret_cast<hipStream_t>(stream));
}

static int adam_multi_launch(int n_tensors, float* const* p, const float* const* g, float* const* m,
                             float* const* v, const long long* n, const float* h8,
                             const float* hyper_dev, hipStream_t stream, const float* partial,
                             float* grads, int G) {
  using namespace pcgmix;
  if (partial && n_tensors > kAdamMaxTensors) return hipErrorInvalidValue;   // one table, one launch
  for (int first = 0; first < n_tensors; first += kAdamMaxTensors) {
    AdamTable tab;
    tab.count = 0;
    int blocks = 0;
    const int last = first + kAdamMaxTensors < n_tensors ? first + kAdamMaxTensors : n_tensors;
    for (int i = first; i < last; ++i) {
      if (n[i] < 0 || (n[i] > 0 && (!p[i] || !g[i] || !m[i] || !v[i]))) return hipErrorInvalidValue;
      if (n[i] == 0) continue;
      // tensors whose gradient lives in grads[0 .. kNGrad) are updated by the reduction blocks
      const bool deferred = partial && g[i] >= grads && g[i] < grads + kNGrad;
      const long long nb = deferred ? 0 : (n[i] + kAdamEPB - 1) / kAdamEPB;
      if (nb > (1ll << 30) - blocks) return hipErrorInvalidValue;
      const int k = tab.count++;
      tab.p[k] = p[i]; tab.g[k] = g[i]; tab.m[k] = m[i]; tab.v[k] = v[i]; tab.n[k] = n[i];
      tab.blk_start[k] = blocks;
      blocks += (int)nb;
    }
    if (tab.count == 0) continue;
    tab.blk_start[tab.count] = blocks;
    const int red_first = blocks;
    if (partial) blocks += kNGrad;
    hipLaunchKernelGGL(adam_clip_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, tab,
                       h8[0], h8[1], h8[2], h8[3], h8[4], h8[5], h8[6], h8[7], hyper_dev, partial, grads, G,
                       red_first);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return (int)err;
  }
  return hipSuccess;
}

extern "C" int pcgmix_adam_clip_f32(float* p, const float* g, float* m, float* v, long long n,
                                    float clip, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, long long step, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!p || !g || !m || !v || n < 0 || step < 1) return hipErrorInvalidValue;
  if (n == 0) return hipSuccess;
  // bias corrections in float64 on the host, as torch computes them from Python floats
  const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p, g, m, v, n, clip, weight_decay,
                     1.0f - beta1, beta2, 1.0f - beta2, step_size, inv_bc2_sqrt, eps);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_out_len(int T) {
  return T < 14 ? 0 : pcgmix::potes_dims(T).P2;
}

extern "C" int pcgmix_potes_bwd_blocks(int N, int T) {
  if (N <= 0 || T < 14) return 0;
  const pcgmix::PotesDims d = pcgmix::potes_dims(T);
  const long long work = (long long)N * ((d.P2 + 2 + pcgmix::kBwdTP - 1) / pcgmix::kBwdTP);  // = tiles
  // persistent blocks: 4 per CU (128 VGPRs, 4 waves per SIMD: 51.9 us at 1024 blocks, 53.0 at 768,
  // 52.6 at 1536; profiles/r4_potes_bwd_sweep.txt)
  long long cap = 1024;
  if (const char* env = getenv("PCGMIX_POTES_BWD_BLOCKS")) {   // tuning runs
    const long long v = atoll(env);
    if (v >= 1 && v <= 65535) cap = v;
  }
  return (int)(work < cap ? work : cap);
}

// Persistent blocks of the matrix-core forward: four per CU (128 VGPRs), tiles handed out by stride.
static unsigned potes_fwd_mfma_blocks(int N, const pcgmix::PotesDims& d) {
  const long long items = (long long)N * ((d.P2 + pcgmix::kFwdTP - 1) / pcgmix::kFwdTP);
  long long cap = 4 * 256;
  if (const char* env = getenv("PCGMIX_POTES_FWD_BLOCKS")) {   // tuning runs
    const long long v = atoll(env);
    if (v >= 1 && v <= (1 << 20)) cap = v;
  }
  return (unsigned)(items < cap ? items : cap);
}

extern "C" int pcgmix_potes_stack_fwd_f32(const float* x, const float* w1, const float* b1,
                                          const float* w2, const float* b2, float* h2, int N, int T,
                                          pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !w1 || !b1 || !w2 || !b2 || !h2 || N < 0 || T < 14 || N > 65535 * 1)
    return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  const PotesDims d = potes_dims(T);
  hipLaunchKernelGGL(potes_fwd_mfma_kernel<false>, dim3(potes_fwd_mfma_blocks(N, d)),
                     dim3(kPotThreads), 0, reinterpret_cast<hipStream_t>(stream), x, w1, b1, w2, b2, h2,
                     nullptr, nullptr, N, T, nullptr, 0ll, nullptr, 0u, 0u);
  return (int)hipGetLastError();
}

extern "C" long long pcgmix_potes_mask_bytes(int N, int T, int layer) {
  if (N <= 0 || T < 14) return 0;
  const pcgmix::PotesDims d = pcgmix::potes_dims(T);
  return layer == 2 ? (long long)N * pcgmix::kC2 * ((d.P2 + 3) / 4)
                    : (layer == 1 ? (long long)N * pcgmix::kC1 * pcgmix::potes_s1_row_bytes(d) : 0);
}

extern "C" int pcgmix_potes_stack_fwd_save_f32(const float* x, const float* w1, const float* b1,
                                               const float* w2, const float* b2, float* h2,
                                               uint8_t* m2, uint8_t* s1, int N, int T,
                                               uint8_t* rnd_out, long long rnd_bytes,
                                               const uint32_t* key_dev, uint64_t key,
                                               pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !w1 || !b1 || !w2 || !b2 || !h2 || !m2 || N < 0 || T < 14 || N > 65535)
    return hipErrorInvalidValue;
  if (rnd_out && (rnd_bytes <= 0 || (rnd_bytes & 15) || rnd_bytes > (16ll << 30) ||
                  (reinterpret_cast<uintptr_t>(rnd_out) & 15) || N == 0))
    return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  const PotesDims d = potes_dims(T);
  hipLaunchKernelGGL(potes_fwd_mfma_kernel<true>, dim3(potes_fwd_mfma_blocks(N, d)),
                     dim3(kPotThreads), 0, reinterpret_cast<hipStream_t>(stream), x, w1, b1, w2, b2, h2, m2, s1, N, T,
                     reinterpret_cast<uint4*>(rnd_out), rnd_out ? rnd_bytes / 16 : 0ll, key_dev,
                     (uint32_t)key, (uint32_t)(key >> 32));
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_stack_input_grad_f32(const float* x, const float* grad_h2,
                                                 const float* w1, const float* b1, const float* w2,
                                                 const float* b2, float* grad_x, int N, int T,
                                                 pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !grad_h2 || !w1 || !b1 || !w2 || !b2 || !grad_x || N < 0 || N > 65535 || T < 14)
    return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  dim3 grid((unsigned)((T + kInNU - 1) / kInNU), (unsigned)N), block(kPotThreads);
  hipLaunchKernelGGL(potes_input_grad_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream),
                     x, grad_h2, w1, b1, w2, b2, grad_x, N, T);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_stack_bwd_f32(const float* x, const float* grad_h2, const float* w1,
                                          const float* b1, const float* w2, const float* b2,
                                          float* partial, float* grads, int N, int T,
                                          pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !grad_h2 || !w1 || !b1 || !w2 || !b2 || !partial || !grads || N <= 0 || N > 65535 || T < 14)
    return hipErrorInvalidValue;
  const int G = pcgmix_potes_bwd_blocks(N, T);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(potes_bwd_kernel<false>, dim3((unsigned)G), dim3(kPotThreads), 0, s, x, grad_h2,
                     nullptr, w1, b1, w2, b2, partial, N, T);
  hipLaunchKernelGGL(potes_reduce_kernel, dim3(kNGrad), dim3(kPotThreads), 0, s, partial, grads, G);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_stack_bwd_mask_f32(const float* x, const float* grad_h2,
                                               const uint8_t* m2, const float* w1, const float* b1,
                                               const float* w2, const float* b2, float* partial,
                                               float* grads, int N, int T, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !grad_h2 || !m2 || !w1 || !b1 || !w2 || !b2 || !partial || N <= 0 || N > 65535 || T < 14)
    return hipErrorInvalidValue;
  const int G = pcgmix_potes_bwd_blocks(N, T);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(potes_bwd_pair_kernel, dim3((unsigned)G), dim3(kPotThreads), 0, s, x, grad_h2,
                     m2, w1, b1, w2, b2, partial, N, T);
  if (grads)      // NULL: the caller reduces later (pcgmix_adam_clip_multi_reduce_dev_f32 / pcgmix_potes_reduce_f32)
    hipLaunchKernelGGL(potes_reduce_kernel, dim3(kNGrad), dim3(kPotThreads), 0, s, partial, grads, G);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_potes_stack_input_grad_mask_f32(const float* grad_h2, const uint8_t* m2,
                                                      const uint8_t* s1, const float* w1,
                                                      const float* w2, float* grad_x, int N, int T,
                                                      pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!grad_h2 || !m2 || !s1 || !w1 || !w2 || !grad_x || N < 0 || N > 65535 || T < 14)
    return hipErrorInvalidValue;
  if (N == 0) return hipSuccess;
  // persistent blocks: 6 resident per CU (81 VGPRs, 22 KB of LDS); twice that many, the second
  // half starting as the first ends, evens out the tail: 1024: 34.0, 1536: 33.2, 2048: 31.3,
  // 3072: 30.6 us at N = 1024 x 5000 (profiles/r4_potes_ingrad_pair.txt)
  long long cap = 3072;
  if (const char* env = getenv("PCGMIX_POTES_INGRAD_BLOCKS")) {   // tuning runs
    const long long v = atoll(env);
    if (v >= 1 && v <= 65535) cap = v;
  }
  const long long work = (long long)N * ((T + kInNU - 1) / kInNU);
  hipLaunchKernelGGL(potes_input_grad_pair_kernel, dim3((unsigned)(work < cap ? work : cap)),
                     dim3(kPotThreads), 0, reinterpret_cast<hipStream_t>(stream), grad_h2, m2, s1, w1,
                     w2, grad_x, N, T);
  return (int)hipGetLastError();
}

extern "C" int pcgmix_skinny_linear_splits(int B, int K) {
  if (B <= 0 || K <= 0) return 0;
  return (K + pcgmix::kSkinnyChunk - 1) / pcgmix::kSkinnyChunk;   // one partial per K chunk
}

namespace pcgmix {
// Launch the split-K partial products of z = h W^T (shared with the fused head, pcgmix_head.hip).
hipError_t launch_skinny_partial(const float* h, const float* W, float* partial, int B, int K,
                                 int O, hipStream_t s, const uint8_t* mask, float scale, int thr,
                                 int bits) {
  if (!h || !W || !partial || B <= 0 || K <= 0 || (K & 3) || O <= 0 || O > kSkinnyMaxO)
    return hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(W)) & 15)
    return hipErrorInvalidValue;
  if (mask && ((reinterpret_cast<uintptr_t>(mask) & 3) ||
               (bits != 1 && bits != 2 && bits != 4 && bits != 8)))
    return hipErrorInvalidValue;
  const int KS = pcgmix_skinny_linear_splits(B, K);
  dim3 grid((unsigned)((B + kSkinnyRows - 1) / kSkinnyRows), (unsigned)KS),
      block(kSkinnyWaves * 64);
  if (O == 20 && mask && bits == 2) {
    hipLaunchKernelGGL((skinny_linear_partial_kernel<20, true, 2>), grid, block, 0, s, h, W, partial, B,
                       K, mask, scale, thr, bits);
  } else if (O == 20 && mask) {
    hipLaunchKernelGGL((skinny_linear_partial_kernel<20, true>), grid, block, 0, s, h, W, partial, B, K,
                       mask, scale, thr, bits);
  } else if (mask) {
    return hipErrorInvalidValue;                      // the masked variant exists for the Potes head
  } else if (O == 20) {
    hipLaunchKernelGGL((skinny_linear_partial_kernel<20, false>), grid, block, 0, s, h, W, partial, B,
                       K, nullptr, 1.f, 0, 8);
  } else if (O == 8) {
    hipLaunchKernelGGL((skinny_linear_partial_kernel<8, false>), grid, block, 0, s, h, W, partial, B,
                       K, nullptr, 1.f, 0, 8);
  } else if (O == 16) {
    hipLaunchKernelGGL((skinny_linear_partial_kernel<16, false>), grid, block, 0, s, h, W, partial, B,
                       K, nullptr, 1.f, 0, 8);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
}  // namespace pcgmix

extern "C" int pcgmix_skinny_linear_fwd_f32(const float* h, const float* W, const float* bias,
                                            float* partial, float* z, int B, int K, int O,
                                            pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!z) return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const hipError_t e = launch_skinny_partial(h, W, partial, B, K, O, s, nullptr, 1.f, 0, 8);
  if (e != hipSuccess) return (int)e;
  const int KS = pcgmix_skinny_linear_splits(B, K);
  const int n = B * O;
  hipLaunchKernelGGL(skinny_linear_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     partial, bias, z, B, O, KS);
  return (int)hipGetLastError();
}
