// pcgmix_mix.hip — fused segment-aware splice (+ magnitude warp) for gfx950 (MI355X).
//
// One launch replaces the reference's per-sample Python loop of slice blends
// (augmentations.py:289-337, 909-917, 969-977; 2D: augmentations2d.py:206-221) and, when knots
// are given, magnitude_warp with its host round trip (augmentations.py:674-683, 924-928).
//
// Work decomposition: blockIdx.y = sample b, blockIdx.x = a chunk of EPB consecutive elements of
// that sample's flattened (C, T) plane.  Everything that depends only on b (own and partner
// boundaries, displacement, partner base) is wave-uniform and lives in SGPRs; each lane owns
// VEC consecutive samples of one row, classifies them into the four heart states with unsigned
// range compares (boundaries are cumulative, so no scan is needed), reads its own row with one
// 16-byte load and the partner row at the shifted position, blends with separate fp32
// mul/mul/add (bit-compatible with torch's unfused ops), optionally multiplies by the spline in
// fp64 and stores 16 bytes.  HBM-bound: 12 algorithmic bytes per element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcgmix_kernels.h"

namespace pcgmix {

constexpr int kThreads = 256;
constexpr int kBatchPerGridZ = 32768;  // gridDim.y is capped at 65535: b = z * 32768 + y

// 16-byte vector whose address is only known to be 4-byte aligned (partner rows are read at
// an arbitrary sample offset); gfx950 global loads handle the misalignment in hardware.
typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float4_a __attribute__((ext_vector_type(4), aligned(16)));

struct StateMap {
  int a[4];      // own-side start of the blended range of state k
  int n[4];      // its length (0 = nothing to blend)
  int delta[4];  // partner index = own index + delta
};

// Boundaries of sample b and its partner -> blended ranges.  All inputs are block-uniform.
__device__ __forceinline__ StateMap make_state_map(const int32_t* __restrict__ frames,
                                                   const int32_t* __restrict__ off, int b, int m,
                                                   int T, const float2* __restrict__ part = nullptr) {
  StateMap sm;
  int f1[5], f2[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    f1[k] = frames[b * 5 + k];
    f2[k] = frames[m * 5 + k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int len1 = f1[k + 1] - f1[k];
    int len2 = f2[k + 1] - f2[k];
    int gap = len2 - len1;
    int agap = gap < 0 ? -gap : gap;
    int o = off ? off[b * 4 + k] : 0;
    if (part) {  // salopt_finalize_kernel's rule: greatest value, smallest displacement on ties
      float bv = -INFINITY;
      int bd = 0x7fffffff;
#pragma unroll
      for (int z = 0; z < kDispSplit; ++z) {
        const float2 p = part[((size_t)b * 4 + k) * kDispSplit + z];
        const int dd = __float_as_int(p.y);
        if (p.x > bv || (p.x == bv && dd < bd)) {
          bv = p.x;
          bd = dd;
        }
      }
      o = bd == 0x7fffffff ? 0 : bd;
    }
    o = o < 0 ? 0 : (o > agap ? agap : o);
    int a = f1[k] + (gap < 0 ? o : 0);
    int s = f2[k] + (gap > 0 ? o : 0);
    int n = len1 < len2 ? len1 : len2;
    // never blend outside the row on either side (malformed frames are rejected on the host;
    // this keeps the kernel memory-safe regardless)
    if (a < 0 || s < 0) n = 0;
    if (n > T - a) n = T - a;
    if (n > T - s) n = T - s;
    if (n < 0) n = 0;
    sm.a[k] = a;
    sm.n[k] = n;
    sm.delta[k] = s - a;
  }
  return sm;
}

// Returns the partner-minus-own index shift for sample position t, or INT_MIN if t is not
// inside a blended range.
__device__ __forceinline__ int blend_shift(const StateMap& sm, int t, bool& hit) {
  int d = 0;
  hit = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    bool in = (unsigned)(t - sm.a[k]) < (unsigned)sm.n[k];
    d = in ? sm.delta[k] : d;
    hit = hit || in;
  }
  return d;
}

__device__ __forceinline__ float blend(float own, float other, float lam, float oml) {
  // x*lam + partner*(1-lam) as three separately rounded fp32 ops (augmentations.py:294)
  return __fadd_rn(__fmul_rn(own, lam), __fmul_rn(other, oml));
}

// ---- magnitude-warp spline -------------------------------------------------------------------
// LDS image built per block: for every channel the chunk touches, one 6-double record per
// cubic piece {c0, c1, c2, c3, brk[p], pad} (scipy PPoly layout: c0 multiplies s^3), followed by
// the integer thresholds thr[i] = ceil(brk[i]) so that "brk[i] <= t" is an integer compare.
constexpr int kRec = 6;

__device__ __forceinline__ int spline_piece(const int* __restrict__ thr, int n_knots, int t) {
  int p = 0;  // searchsorted(brk, t, 'right') - 1, clipped to the last piece
  for (int i = 1; i <= n_knots - 2; ++i) p += (t >= thr[i]) ? 1 : 0;
  return p;
}

// scipy's PPoly evaluation order: c3 + c2*s + c1*s^2 + c0*s^3 with a running power, every
// operation rounded separately in float64; then float32(float64(v) * w) as numpy stores it.
__device__ __forceinline__ float spline_scale(float v, int t, double c0, double c1, double c2,
                                              double c3, double brk) {
  const double s = __dsub_rn((double)t, brk);
  double r = __dadd_rn(c3, __dmul_rn(c2, s));
  double z = __dmul_rn(s, s);
  r = __dadd_rn(r, __dmul_rn(c1, z));
  z = __dmul_rn(z, s);
  r = __dadd_rn(r, __dmul_rn(c0, z));
  return __double2float_rn(__dmul_rn((double)v, r));
}

template <int VEC>
__device__ __forceinline__ void spline_apply(float (&out)[VEC], const double* __restrict__ tab,
                                             const int* __restrict__ thr, int n_knots, int t0) {
  const int p0 = spline_piece(thr, n_knots, t0);
  const int p1 = VEC > 1 ? spline_piece(thr, n_knots, t0 + VEC - 1) : p0;
  if (p0 == p1) {  // the usual case: pieces are hundreds of samples long
    const double* r = tab + p0 * kRec;
    const double c0 = r[0], c1 = r[1], c2 = r[2], c3 = r[3], brk = r[4];
#pragma unroll
    for (int e = 0; e < VEC; ++e) out[e] = spline_scale(out[e], t0 + e, c0, c1, c2, c3, brk);
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double* r = tab + spline_piece(thr, n_knots, t0 + e) * kRec;
      out[e] = spline_scale(out[e], t0 + e, r[0], r[1], r[2], r[3], r[4]);
    }
  }
}

// VEC = 4: T % 4 == 0, rows are 16-byte aligned, U quads per lane (epb = 256*4*U).
// VEC = 1: any T, one element per lane and iteration.
// The kernel proper: sample b (block-uniform), its partner m and their blended ranges are known.
template <int VEC, bool WARP, int U>
__device__ __forceinline__ void mix_body(
    const float* __restrict__ x, float* __restrict__ y, float lam, float oml,
    const double* __restrict__ knots, const double* __restrict__ spline_op, int n_knots,
    const int32_t* __restrict__ zero_rect, int C, int T, int epb, int b, int m, const StateMap& sm,
    double* lds, int chunk = (int)blockIdx.x) {
  // optional zeroed rectangle (rows = index along C, columns = index along T), block-uniform
  int zr0 = 0, zr1 = 0, zc0 = 0, zc1 = 0;
  if (zero_rect) {
    zr0 = zero_rect[b * 4 + 0];
    zr1 = zero_rect[b * 4 + 1];
    zc0 = zero_rect[b * 4 + 2];
    zc1 = zero_rect[b * 4 + 3];
  }

  const int plane = C * T;
  const int chunk0 = chunk * epb;
  const size_t own_base = (size_t)b * plane;
  const size_t par_base = (size_t)m * plane;

  // Spline records for the channels this chunk touches (coef = op * knots[b,:,c]) are built in
  // LDS AFTER the lane's data loads have been issued, so their dependent-load latency overlaps
  // with the HBM latency of the batch data instead of preceding it.
  const int rec_per_ch = (n_knots - 1) * kRec;
  const int c_lo = WARP ? chunk0 / T : 0;
  const int* thr = WARP ? reinterpret_cast<const int*>(lds + (size_t)(epb / T + 2) * rec_per_ch)
                        : nullptr;
  auto build_records = [&]() {
    int last = chunk0 + epb - 1;
    if (last > plane - 1) last = plane - 1;
    const int nch = last / T - c_lo + 1;
    int* thr_w = reinterpret_cast<int*>(lds + (size_t)(epb / T + 2) * rec_per_ch);
    for (int i = threadIdx.x; i < n_knots; i += kThreads) thr_w[i] = (int)ceil(spline_op[i]);
    const int total = nch * rec_per_ch;
    for (int i = threadIdx.x; i < total; i += kThreads) {
      const int c = c_lo + i / rec_per_ch;
      const int r = i % rec_per_ch;
      const int piece = r / kRec, j4 = r % kRec;
      double acc = 0.0;
      if (j4 < 4) {
        const double* mrow = spline_op + n_knots + (size_t)(piece * 4 + j4) * n_knots;
        for (int j = 0; j < n_knots; ++j)
          acc = __dadd_rn(acc, __dmul_rn(mrow[j], knots[((size_t)b * n_knots + j) * C + c]));
      } else if (j4 == 4) {
        acc = spline_op[piece];
      }
      lds[i] = acc;
    }
    __syncthreads();
  };

  // With U = 4 the extra live registers of "loads first" cost a wave of occupancy (129 VGPRs), so
  // the fattest variant builds its records up front.
  constexpr bool kRecordsFirst = U >= 4;
  if constexpr (VEC == 4) {
    if (WARP && kRecordsFirst) build_records();
    // Two phases so that all U own-row and partner-row loads of a lane are in flight together.
    // Phase 1 is branch-free: the partner quad is fetched with ONE unaligned 16-byte load at
    // the shift of the first blended element (clamped into the row); elements of the quad that
    // blend with a different shift (a state boundary inside the quad) or whose quad load had to
    // be clamped are patched by a scalar load in phase 2 (at most a handful of quads per row).
    float4_a own[U];
    float4_u par[U];
    int t0s[U], cs[U], masks[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int i = chunk0 + (q * kThreads + (int)threadIdx.x) * 4;
      const bool valid = i < plane;
      const int ii = valid ? i : 0;
      const int c = ii / T;
      const int t0 = ii - c * T;
      bool hit[4];
      int d[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = blend_shift(sm, t0 + e, hit[e]);
      const int dsel = hit[0] ? d[0] : hit[1] ? d[1] : hit[2] ? d[2] : hit[3] ? d[3] : 0;
      int src0 = t0 + dsel;
      src0 = src0 < 0 ? 0 : (src0 > T - 4 ? T - 4 : src0);
      const bool clamped = src0 != t0 + dsel;
      int mask = valid ? 0x100 : 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (hit[e]) mask |= 1 << e;
        if (hit[e] && (clamped || d[e] != dsel)) mask |= 16 << e;
      }
      own[q] = *reinterpret_cast<const float4_a*>(x + own_base + ii);
      // The zero-padded tail and the unmatched part of longer states never touch the partner row —
      // but the load is NOT predicated: `if (mask & 0xf) pz = *p` compiles to a branch around the load
      // with an `s_waitcnt vmcnt(0)` inside it, so a lane's second own/partner pair was issued only
      // after its first had come back (two memory round trips per lane in the kernel that bounds the
      // strict-signature step).  A quad without a blended element re-reads its own quad instead: the
      // line is in flight already, no byte more leaves memory, and all 2 U loads are in flight together.
      size_t poff = (mask & 0xf) ? par_base + (size_t)c * T + src0 : own_base + ii;
      asm volatile("" : "+v"(poff));  // opaque (the OFFSET, so that the access stays a global load): otherwise the
                                      // compiler reuses the own quad's RESULT for the fallback and predicates the
                                      // partner load again, behind a wait for the own one
      par[q] = *reinterpret_cast<const float4_u*>(x + poff);
      t0s[q] = t0;
      cs[q] = c;
      masks[q] = mask;
    }
    if (WARP && !kRecordsFirst) build_records();
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int mask = masks[q];
      if (!(mask & 0x100)) continue;
      const int t0 = t0s[q], c = cs[q];
      float o[4] = {own[q].x, own[q].y, own[q].z, own[q].w};
      float pv[4] = {par[q].x, par[q].y, par[q].z, par[q].w};
      float out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = pv[e];
        if (mask & (16 << e)) {  // rare: patch with the element's own shift
          bool h;
          const int de = blend_shift(sm, t0 + e, h);
          v = x[par_base + (size_t)c * T + t0 + e + de];
        }
        out[e] = (mask & (1 << e)) ? blend(o[e], v, lam, oml) : o[e];
      }
      if (WARP) spline_apply<4>(out, lds + (size_t)(c - c_lo) * rec_per_ch, thr, n_knots, t0);
      if (c >= zr0 && c < zr1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t0 + e >= zc0 && t0 + e < zc1) out[e] = 0.f;
      }
      float4_a v;
      v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
      const int i = chunk0 + (q * kThreads + (int)threadIdx.x) * 4;
      // written once, read by a later kernel on other XCDs: a non-temporal store keeps the output out of
      // this XCD's L2 (no allocation, nothing to write back when the kernel ends).  (256,4,5000) 9.05 ->
      // 8.59 us, 16384 x 4 x 5000 507 -> 466 us (profiles/r4_nt_stores.txt)
      __builtin_nontemporal_store(v, reinterpret_cast<float4_a*>(y + own_base + i));
    }
  } else {
    if (WARP) build_records();
    for (int i = chunk0 + threadIdx.x; i < chunk0 + epb && i < plane; i += kThreads) {
      const int c = i / T;
      const int t0 = i - c * T;
      bool hit;
      const int d = blend_shift(sm, t0, hit);
      float out[1] = {x[own_base + i]};
      if (hit) out[0] = blend(out[0], x[par_base + (size_t)c * T + t0 + d], lam, oml);
      if (WARP) spline_apply<1>(out, lds + (size_t)(c - c_lo) * rec_per_ch, thr, n_knots, t0);
      if (c >= zr0 && c < zr1 && t0 >= zc0 && t0 < zc1) out[0] = 0.f;
      y[own_base + i] = out[0];
    }
  }
}

template <int VEC, bool WARP, int U>
__global__ __launch_bounds__(kThreads) void mix_warp_kernel(
    const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ frames,
    const int32_t* __restrict__ mix_idx, const int32_t* __restrict__ off, float lam, float oml,
    const double* __restrict__ knots, const double* __restrict__ spline_op, int n_knots,
    const int32_t* __restrict__ zero_rect, int B, int C, int T, int epb,
    const uint4* __restrict__ pay_src, uint4* __restrict__ pay_dst, int pay_n16,
    const float2* __restrict__ disp_part, const PartnerPack pk) {
  extern __shared__ __align__(16) double lds[];  // spline records per channel, then thresholds

  // Step payload (pcgmix_ctx_set_payload): a few KB that travelled with the index block and
  // belong somewhere else on the device; block (0,0,0) forwards them.
  if (pay_n16 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0)
    for (int i = threadIdx.x; i < pay_n16; i += kThreads) pay_dst[i] = pay_src[i];

  const int b = blockIdx.z * kBatchPerGridZ + blockIdx.y;
  if (b >= B) return;  // block-uniform
  int m = pk.n ? partner_get(pk, b) : mix_idx[b];
  m = (m < 0 || m >= B) ? b : m;  // memory safety; validated on the host as well
  const StateMap sm = make_state_map(frames, off, b, m, T, disp_part);
  mix_body<VEC, WARP, U>(x, y, lam, oml, knots, spline_op, n_knots, zero_rect, C, T, epb, b, m, sm, lds);
}

// ---- splice + warp, one lane = one quad of sample positions for ALL channels -------------------
// The spline's break points are the same for every channel (linspace(0, T-1, n),
// augmentations.py:676), so everything that depends on the sample position only — the state
// classification and partner shift of the splice, the spline piece, s = t - brk, s^2, s^3 and the
// int -> float64 conversion — is computed once per position and reused by the C channels of the
// row group (mix_body computes it per (channel, position): 4 of its 13 float64 operations per
// element, plus all of the integer work).  Same arithmetic per element, in the same order, each
// operation rounded on its own: bit-identical output (tests/test_mix_gpu.py).
// Block = kThreads * 4 * UT consecutive positions of one sample, all channels; LDS = the C * (n-1)
// coefficient records of the sample, built while the first loads are in flight.
// knots_b: the n_knots * C knots of THIS sample (device memory, or device-readable host memory for the
// armed launch).  LATE (the armed launch): the sample's partner and blended ranges are not known when the
// block starts — everything that does not depend on them (the own rows' first loads, operator and
// knots into LDS, the coefficient records, every position's spline piece and powers) runs first, then
// `await(sm, m)` (block-uniform; false = give up) delivers them.
struct TqKnown {
  __device__ bool operator()(StateMap&, int&) const { return true; }
};
template <int CG, int UT, bool LATE = false, class Await = TqKnown>
__device__ __forceinline__ void tq_body(
    const float* __restrict__ x, float* __restrict__ y, float lam, float oml,
    const double* __restrict__ knots_b, const double* __restrict__ spline_op, int n_knots, int C, int T,
    int b, int m, StateMap sm, double* lds, Await await = Await()) {
  const size_t own_base = (size_t)b * C * T;
  size_t par_base = (size_t)m * C * T;
  const int rec_per_ch = (n_knots - 1) * kRec;
  int* thr = reinterpret_cast<int*>(lds + (size_t)C * rec_per_ch);
  const int tq0 = blockIdx.x * (kThreads * 4 * UT);

  // per position quad: where the partner quad comes from and which elements blend (as mix_body)
  int t0s[UT], masks[UT], src0s[UT];
#pragma unroll
  for (int q = 0; q < UT; ++q) {
    const int t0 = tq0 + (q * kThreads + (int)threadIdx.x) * 4;
    t0s[q] = t0 < T ? t0 : 0;
    masks[q] = t0 < T ? 0x100 : 0;
    src0s[q] = 0;
  }
  auto classify = [&]() {
#pragma unroll
    for (int q = 0; q < UT; ++q) {
      const int tt = t0s[q];
      bool hit[4];
      int d[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = blend_shift(sm, tt + e, hit[e]);
      const int dsel = hit[0] ? d[0] : hit[1] ? d[1] : hit[2] ? d[2] : hit[3] ? d[3] : 0;
      int src0 = tt + dsel;
      src0 = src0 < 0 ? 0 : (src0 > T - 4 ? T - 4 : src0);
      const bool clamped = src0 != tt + dsel;
      int mask = masks[q] & 0x100;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (hit[e]) mask |= 1 << e;
        if (hit[e] && (clamped || d[e] != dsel)) mask |= 16 << e;
      }
      masks[q] = mask;
      src0s[q] = src0;
    }
  };
  float4_a own[UT][CG];
  float4_u par[UT][CG];
  auto issue_loads = [&](int c0) {
#pragma unroll
    for (int q = 0; q < UT; ++q)
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        const size_t row = (size_t)(c0 + cc) * T;
        own[q][cc] = *reinterpret_cast<const float4_a*>(x + own_base + row + t0s[q]);
        // unpredicated, as in mix_body: a quad without a blended element re-reads its own quad
        size_t poff = (masks[q] & 0xf) ? par_base + row + src0s[q] : own_base + row + t0s[q];
        asm volatile("" : "+v"(poff));
        par[q][cc] = *reinterpret_cast<const float4_u*>(x + poff);
      }
  };
  if (!LATE) {
    classify();
    issue_loads(0);
  }
  // coefficient records of all channels of this sample: coef = op * knots[b, :, c].  The operator
  // and the sample's knots are staged in LDS first — ONE round trip to memory; the dot products
  // with their operands in global memory were n_knots dependent round trips on every block's
  // critical path (each iteration: two loads, s_waitcnt vmcnt(0), one multiply-add).
  const int n_op = n_knots + 4 * (n_knots - 1) * n_knots;
  double* opl = reinterpret_cast<double*>(thr + ((n_knots + 1) & ~1));
  double* knl = opl + n_op;
  for (int i = threadIdx.x; i < n_op + n_knots * C; i += kThreads) {
    if (LATE && i >= n_op) {       // written by another block of this launch: agent-scope load (no stale L2 line)
      const unsigned long long bits = __hip_atomic_load(
          reinterpret_cast<const unsigned long long*>(knots_b) + (i - n_op), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      opl[i] = __longlong_as_double((long long)bits);
    } else {
      opl[i] = i < n_op ? spline_op[i] : knots_b[i - n_op];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_knots; i += kThreads) thr[i] = (int)ceil(opl[i]);
  for (int i = threadIdx.x; i < C * rec_per_ch; i += kThreads) {
    const int c = i / rec_per_ch, r = i % rec_per_ch;
    const int piece = r / kRec, j4 = r % kRec;
    double acc = 0.0;
    if (j4 < 4) {
      const double* mrow = opl + n_knots + (size_t)(piece * 4 + j4) * n_knots;
      for (int j = 0; j < n_knots; ++j) acc = __dadd_rn(acc, __dmul_rn(mrow[j], knl[j * C + c]));
    } else if (j4 == 4) {
      acc = opl[piece];
    }
    lds[i] = acc;
  }
  __syncthreads();
  // per position: piece, s, s^2, s^3 (scipy's running power, each product rounded)
  int pc[UT][4];
  double s1[UT][4], s2[UT][4], s3[UT][4];
#pragma unroll
  for (int q = 0; q < UT; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = t0s[q] + e;
      const int p = spline_piece(thr, n_knots, t);
      const double brk = lds[p * kRec + 4];
      const double s = __dsub_rn((double)t, brk);
      const double z = __dmul_rn(s, s);
      pc[q][e] = p;
      s1[q][e] = s;
      s2[q][e] = z;
      s3[q][e] = __dmul_rn(z, s);
    }
  if (LATE) {
    // the own rows' first loads go out here, behind the records and the positions: at the kernel's entry
    // they (20 MB at once from every block) stood in front of block (0,0)'s label loads — labels 3.0 us
    // after the entry instead of 1.0 — and they have the whole wait to arrive anyway
#pragma unroll
    for (int q = 0; q < UT; ++q)
#pragma unroll
      for (int cc = 0; cc < CG; ++cc)
        own[q][cc] = *reinterpret_cast<const float4_a*>(x + own_base + (size_t)cc * T + t0s[q]);
    if (!await(sm, m)) return;                   // block-uniform
    par_base = (size_t)m * C * T;
    classify();
#pragma unroll
    for (int q = 0; q < UT; ++q)
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        const size_t row = (size_t)cc * T;
        size_t poff = (masks[q] & 0xf) ? par_base + row + src0s[q] : own_base + row + t0s[q];
        asm volatile("" : "+v"(poff));
        par[q][cc] = *reinterpret_cast<const float4_u*>(x + poff);
      }
  }
  for (int c0 = 0; c0 < C; c0 += CG) {
    if (c0) issue_loads(c0);
#pragma unroll
    for (int q = 0; q < UT; ++q) {
      const int mask = masks[q];
      if (!(mask & 0x100)) continue;
      const int t0 = t0s[q];
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        const int c = c0 + cc;
        const float o[4] = {own[q][cc].x, own[q][cc].y, own[q][cc].z, own[q][cc].w};
        const float pv[4] = {par[q][cc].x, par[q][cc].y, par[q][cc].z, par[q][cc].w};
        const double* rec = lds + (size_t)c * rec_per_ch;
        float bl[4], out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = pv[e];
          if (mask & (16 << e)) {  // rare: patch with the element's own shift
            bool h;
            const int de = blend_shift(sm, t0 + e, h);
            v = x[par_base + (size_t)c * T + t0 + e + de];
          }
          bl[e] = (mask & (1 << e)) ? blend(o[e], v, lam, oml) : o[e];
        }
        if (pc[q][0] == pc[q][3]) {  // the usual case: pieces are hundreds of samples long
          const double* r = rec + pc[q][0] * kRec;
          const double k0 = r[0], k1 = r[1], k2 = r[2], k3 = r[3];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            double w = __dadd_rn(k3, __dmul_rn(k2, s1[q][e]));
            w = __dadd_rn(w, __dmul_rn(k1, s2[q][e]));
            w = __dadd_rn(w, __dmul_rn(k0, s3[q][e]));
            out[e] = __double2float_rn(__dmul_rn((double)bl[e], w));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double* r = rec + pc[q][e] * kRec;
            double w = __dadd_rn(r[3], __dmul_rn(r[2], s1[q][e]));
            w = __dadd_rn(w, __dmul_rn(r[1], s2[q][e]));
            w = __dadd_rn(w, __dmul_rn(r[0], s3[q][e]));
            out[e] = __double2float_rn(__dmul_rn((double)bl[e], w));
          }
        }
        float4_a v4;
        v4.x = out[0]; v4.y = out[1]; v4.z = out[2]; v4.w = out[3];
        __builtin_nontemporal_store(v4, reinterpret_cast<float4_a*>(y + own_base + (size_t)c * T + t0));
      }
    }
  }
}

template <int CG, int UT>
__global__ __launch_bounds__(kThreads) void mix_warp_tq_kernel(
    const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ frames,
    const int32_t* __restrict__ mix_idx, const int32_t* __restrict__ off, float lam, float oml,
    const double* __restrict__ knots, const double* __restrict__ spline_op, int n_knots, int B, int C,
    int T, const uint4* __restrict__ pay_src, uint4* __restrict__ pay_dst, int pay_n16,
    const float2* __restrict__ disp_part, const PartnerPack pk) {
  extern __shared__ __align__(16) double lds[];  // C * (n_knots - 1) records, then thresholds
  if (pay_n16 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0)
    for (int i = threadIdx.x; i < pay_n16; i += kThreads) pay_dst[i] = pay_src[i];
  const int b = blockIdx.z * kBatchPerGridZ + blockIdx.y;
  if (b >= B) return;  // block-uniform
  int m = pk.n ? partner_get(pk, b) : mix_idx[b];
  m = (m < 0 || m >= B) ? b : m;
  const StateMap sm = make_state_map(frames, off, b, m, T, disp_part);
  tq_body<CG, UT>(x, y, lam, oml, knots + (size_t)b * n_knots * C, spline_op, n_knots, C, T, b, m, sm, lds);
}

// The same kernel with the whole index block — boundaries and partners of up to kPackB samples
// as int16 — in its ARGUMENTS (3 KB of the 4 KB kernarg segment): block-uniform data that the
// blocks read with scalar loads, and no host-to-device copy in front of the launch.  For the
// step context's plain step at B <= 256, T <= 32767 (BASELINE configs[1]): the 6 KB blit copy
// was 4.7 us of GPU time in the middle of the label -> host -> copy -> splice chain that bounds a
// strict-signature step.
struct IdxPack {            // int16 values two per dword: boundaries fr[b*5+k] at i = b*5+k, partners at
  int32_t w[kPackB * 3];    // i = kPackB*5 + b.  Dwords so that the block-uniform reads become s_load_dword
};                          // (there is no scalar 16-bit load; int16 members are fetched through VGPRs).
__device__ __forceinline__ int pack_get(const IdxPack& p, int i) {
  const int w = p.w[i >> 1];
  return (i & 1) ? (w >> 16) : ((int)((unsigned)w << 16) >> 16);
}

struct PayPack {           // a small step payload in the arguments (see launch_mix_karg)
  uint4 w[kPackPayBytes / 16];
  int n16;
};

template <bool WARP, int U>
__global__ __launch_bounds__(kThreads) void mix_warp_karg_kernel(
    const float* __restrict__ x, float* __restrict__ y, const IdxPack pack, float lam, float oml,
    const double* __restrict__ knots, const double* __restrict__ spline_op, int n_knots, int B,
    int C, int T, int epb, const PayPack pay, uint4* __restrict__ pay_dst) {
  extern __shared__ __align__(16) double lds[];
  if (pay.n16 && (blockIdx.x | blockIdx.y) == 0 && (int)threadIdx.x < pay.n16)
    pay_dst[threadIdx.x] = pay.w[threadIdx.x];
  const int b = blockIdx.y;
  if (b >= B) return;
  int m = pack_get(pack, kPackB * 5 + b);
  m = (m < 0 || m >= B) ? b : m;
  // five boundaries = five consecutive halfwords = three consecutive dwords from (5 b) >> 1: fetched as
  // three INDEPENDENT scalar loads per sample (own ones beside the partner index, the partner's behind
  // it) and taken apart with shifts — one pack_get per boundary was a chain of eleven dependent loads
  int f1[5], f2[5];
  {
    const int j1 = (b * 5) >> 1, j2 = (m * 5) >> 1;
    const int o1 = (b * 5) & 1, o2 = (m * 5) & 1;
    const int a0 = pack.w[j1], a1 = pack.w[j1 + 1], a2 = pack.w[j1 + 2];
    const int c0 = pack.w[j2], c1 = pack.w[j2 + 1], c2 = pack.w[j2 + 2];
    auto half = [](int w0, int w1, int w2, int h) {       // halfword h (0..5) of three dwords, signed
      const int w = h < 2 ? w0 : (h < 4 ? w1 : w2);
      return (h & 1) ? (w >> 16) : ((int)((unsigned)w << 16) >> 16);
    };
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      f1[k] = half(a0, a1, a2, k + o1);
      f2[k] = half(c0, c1, c2, k + o2);
    }
  }
  StateMap sm;
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // make_state_map without offsets
    const int a = f1[k], s = f2[k];
    const int len1 = f1[k + 1] - a, len2 = f2[k + 1] - s;
    int n = len1 < len2 ? len1 : len2;
    if (a < 0 || s < 0) n = 0;
    if (n > T - a) n = T - a;
    if (n > T - s) n = T - s;
    if (n < 0) n = 0;
    sm.a[k] = a;
    sm.n[k] = n;
    sm.delta[k] = s - a;
  }
  mix_body<4, WARP, U>(x, y, lam, oml, knots, spline_op, n_knots, nullptr, C, T, epb, b, m, sm, lds);
}

// The plain splice armed before its index block exists (pcgmix_kernels.h, ArmedArgs).
#ifdef PCGMIX_PHASE_CLOCK
// probe build: per launch parity (seq & 1) and sample: block (0,b) entry | relay saw the host's record |
// block (1,b) saw the relayed record | block (1,b) done; [..][kPackB] = block (0,0): entry | labels flagged
__device__ long long g_armed_clock[2][kPackB + 1][4];
#define PCGMIX_ACLOCK(b, i) g_armed_clock[a.seq & 1][b][i] = (long long)wall_clock64()
#else
#define PCGMIX_ACLOCK(b, i)
#endif
template <int U>
__global__ __launch_bounds__(kThreads) void mix_armed_kernel(
    const float* __restrict__ x, float* __restrict__ y, const ArmedArgs a, int B, int C, int T, int epb,
    int chunks, const PayPack pay, uint4* __restrict__ pay_dst) {
  extern __shared__ __align__(16) double lds[];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  if ((blockIdx.x | blockIdx.y) == 0) {
    if (threadIdx.x == 0) { PCGMIX_ACLOCK(kPackB, 0); }
    if (pay.n16 && (int)threadIdx.x < pay.n16) pay_dst[threadIdx.x] = pay.w[threadIdx.x];
    // labels: ONE wave, lane l = rows 4l .. 4l+3, one byte each under the step's token in ONE 8-byte
    // word — valid by itself like the records, so neither a fence nor a flag (and no wait for the
    // stores to be acknowledged across the link) stands between the arg-max and the host
    if (threadIdx.x < 64) {
      const unsigned long long w = ((unsigned long long)a.token << 32) | onehot_argmax4(a.ohe, a.K, 4 * lane, B);
      __hip_atomic_store(a.lab64 + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (threadIdx.x == 0) { PCGMIX_ACLOCK(kPackB, 1); }
    }
  }
  if (b >= B) return;
  const unsigned long long t0 = wall_clock64();
  const uint32_t go = a.seq, stop = a.seq | kArmedAbort;
  const int word = lane < 7 ? lane : 6;          // six index words and lambda (word 6: float bits)
  if (blockIdx.x == 0 && threadIdx.x == 0) { PCGMIX_ACLOCK(b, 0); }
  if (blockIdx.x == 0 && threadIdx.x < 64) {     // this sample's relay: host record -> device record
    const unsigned long long* src = a.rec_h + (size_t)b * kArmedRecWords + word;
    unsigned long long w;
    bool aborted = false;
    for (;;) {
      w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const uint32_t st = (uint32_t)(w >> 32);
      if (__all(st == go)) break;
      if (__any(st == stop) || wall_clock64() - t0 > a.timeout_ticks) { aborted = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (aborted) {
      w = (unsigned long long)stop << 32;
      if (lane == 0) __hip_atomic_store(a.abort_h, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane < 7)
      __hip_atomic_store(a.rec_d + (size_t)b * kArmedRecWords + lane, w, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) { PCGMIX_ACLOCK(b, 1); }
  }
  unsigned long long w;
  {                                              // every wave: the sample's record on the device
    const unsigned long long* src = a.rec_d + (size_t)b * kArmedRecWords + word;
    for (;;) {
      w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t st = (uint32_t)(w >> 32);
      if (__all(st == go)) break;
      if (__any(st == stop) || wall_clock64() - t0 > 2 * a.timeout_ticks) return;
      __builtin_amdgcn_s_sleep(12);
    }
  }
  if (blockIdx.x == 1 && threadIdx.x == 0) { PCGMIX_ACLOCK(b, 2); }
  const int lo = (int)(uint32_t)w;
  int f1[5], f2[5], m, lam_bits;
  {
    const int w0 = __builtin_amdgcn_readlane(lo, 0), w1 = __builtin_amdgcn_readlane(lo, 1),
              w2 = __builtin_amdgcn_readlane(lo, 2), w3 = __builtin_amdgcn_readlane(lo, 3),
              w4 = __builtin_amdgcn_readlane(lo, 4), w5 = __builtin_amdgcn_readlane(lo, 5);
    lam_bits = __builtin_amdgcn_readlane(lo, 6);
    auto lo16 = [](int v) { return (int)((unsigned)v << 16) >> 16; };
    f1[0] = lo16(w0); f1[1] = w0 >> 16; f1[2] = lo16(w1); f1[3] = w1 >> 16; f1[4] = lo16(w2);
    m = w2 >> 16;
    f2[0] = lo16(w3); f2[1] = w3 >> 16; f2[2] = lo16(w4); f2[3] = w4 >> 16; f2[4] = lo16(w5);
  }
  m = (m < 0 || m >= B) ? b : m;
  StateMap sm;
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // make_state_map without offsets, as mix_warp_karg_kernel
    const int a0 = f1[k], s = f2[k];
    const int len1 = f1[k + 1] - a0, len2 = f2[k + 1] - s;
    int n = len1 < len2 ? len1 : len2;
    if (a0 < 0 || s < 0) n = 0;
    if (n > T - a0) n = T - a0;
    if (n > T - s) n = T - s;
    if (n < 0) n = 0;
    sm.a[k] = a0;
    sm.n[k] = n;
    sm.delta[k] = s - a0;
  }
  // two chunks of the sample per block: (256,4,5000) is 2,560 chunks, more blocks than the chip holds at
  // once (2,048 of four waves) — the last fifth would start, and begin to wait, when the first ones leave
  // lambda travels with the record: a caller may launch before it has drawn it (pcgmix_augment_plain_begin)
  const float lam = __int_as_float(lam_bits), oml = 1.0f - lam;
  for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x)
    mix_body<4, false, U>(x, y, lam, oml, nullptr, nullptr, 0, nullptr, C, T, epb, b, m, sm, lds, chunk);
  if (blockIdx.x == 1 && threadIdx.x == 0) { PCGMIX_ACLOCK(b, 3); }
}

// The splice + warp kernel armed the same way (PCGmix+ through the strict signature): the knots are known
// at launch time — the library drew them ahead — and sit in device-readable pinned memory, 8 * n_knots * C
// bytes per sample: every block stages the operator and its sample's knots into LDS while it waits for
// the sample's record, so neither a label launch nor a fetch launch nor their starts are left in the
// chain.  One wave per block polls; the verdict reaches the others through LDS, block-uniform (the body
// has barriers).
template <int CG, int UT>
__global__ __launch_bounds__(kThreads) void mix_warp_tq_armed_kernel(
    const float* __restrict__ x, float* __restrict__ y, const ArmedArgs a, float lam, float oml,
    const double* __restrict__ knots_host, double* __restrict__ knots_dev,
    const double* __restrict__ spline_op, int n_knots, int B, int C, int T, const PayPack pay,
    uint4* __restrict__ pay_dst) {
  extern __shared__ __align__(16) double lds[];
  __shared__ unsigned long long rec_s[8];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  if ((blockIdx.x | blockIdx.y) == 0) {
    if (threadIdx.x == 0) { PCGMIX_ACLOCK(kPackB, 0); }
    if (pay.n16 && (int)threadIdx.x < pay.n16) pay_dst[threadIdx.x] = pay.w[threadIdx.x];
    if (threadIdx.x < 64) {                      // labels, as mix_armed_kernel
      const unsigned long long w = ((unsigned long long)a.token << 32) | onehot_argmax4(a.ohe, a.K, 4 * lane, B);
      __hip_atomic_store(a.lab64 + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (threadIdx.x == 0) { PCGMIX_ACLOCK(kPackB, 1); }
    }
  }
  if (b >= B) return;
  const unsigned long long t0 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { PCGMIX_ACLOCK(b, 0); }
  // The sample's knots cross the link ONCE: its relay copies them from the pinned slot to the slot's device
  // twin and stamps word 7 of the device record; the sample's blocks wait for that stamp — it comes
  // microseconds before the record can — and stage from device memory.  With every block reading its
  // 8 * n_knots * C bytes from the pinned slot itself, 2,560 small reads (245 KB at (256,4,5000)) were in
  // flight across the link when the label words and the relays' first polls had to cross it: labels
  // 2.2 us after the entry instead of 0.6, records seen 7.7 us after the labels instead of 4.4.
  const int nkc = n_knots * C;
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    // agent-scope stores (written through, no L2 write-back) and a wait for their acknowledgement instead of
    // a release fence: 256 relays issuing buffer_wbl2 at once held up everything behind them — the label
    // words reached the host 8 us late
    for (int i = lane; i < nkc; i += 64)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(knots_dev) + (size_t)b * nkc + i,
                         reinterpret_cast<const unsigned long long*>(knots_host)[(size_t)b * nkc + i],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
      __hip_atomic_store(a.rec_d + (size_t)b * kArmedRecWords + 7, (unsigned long long)a.seq << 32,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x < 64) {
    const unsigned long long* src = a.rec_d + (size_t)b * kArmedRecWords + 7;
    bool ready = false;
    for (;;) {
      const uint32_t st = (uint32_t)(__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
      if (st == a.seq) { ready = true; break; }
      if (wall_clock64() - t0 > 2 * a.timeout_ticks) break;
      __builtin_amdgcn_s_sleep(4);
    }
    if (lane == 0) {
      rec_s[6] = ready ? 1ull : 0ull;
      if (!ready)                                // cannot happen while the grid is resident; the host must know
        __hip_atomic_store(a.abort_h, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (!rec_s[6]) return;                         // block-uniform
  auto await = [&](StateMap& sm, int& m) -> bool {
    const uint32_t go = a.seq, stop = a.seq | kArmedAbort;
    const int word = lane < 6 ? lane : 5;
    if (threadIdx.x < 64) {
      unsigned long long w;
      if (blockIdx.x == 0) {                     // this sample's relay: host record -> device record
        const unsigned long long* src = a.rec_h + (size_t)b * kArmedRecWords + word;
        bool aborted = false;
        for (;;) {
          w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          const uint32_t st = (uint32_t)(w >> 32);
          if (__all(st == go)) break;
          if (__any(st == stop) || wall_clock64() - t0 > a.timeout_ticks) { aborted = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        if (aborted) {
          w = (unsigned long long)stop << 32;
          if (lane == 0) __hip_atomic_store(a.abort_h, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (lane < 6)
          __hip_atomic_store(a.rec_d + (size_t)b * kArmedRecWords + lane, w, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) { PCGMIX_ACLOCK(b, 1); }
      }
      const unsigned long long* src = a.rec_d + (size_t)b * kArmedRecWords + word;
      bool ok = false;
      for (;;) {
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st = (uint32_t)(w >> 32);
        if (__all(st == go)) { ok = true; break; }
        if (__any(st == stop) || wall_clock64() - t0 > 2 * a.timeout_ticks) break;
        __builtin_amdgcn_s_sleep(12);
      }
      if (lane < 6) rec_s[lane] = w;
      if (lane == 6) rec_s[7] = ok ? 1ull : 0ull;
    }
    __syncthreads();
    if (blockIdx.x == 1 && threadIdx.x == 0) { PCGMIX_ACLOCK(b, 2); }
    if (!rec_s[7]) return false;
    int f1[5], f2[5];
    auto lo16 = [](int v) { return (int)((unsigned)v << 16) >> 16; };
    const int w0 = (int)(uint32_t)rec_s[0], w1 = (int)(uint32_t)rec_s[1], w2 = (int)(uint32_t)rec_s[2],
              w3 = (int)(uint32_t)rec_s[3], w4 = (int)(uint32_t)rec_s[4], w5 = (int)(uint32_t)rec_s[5];
    f1[0] = lo16(w0); f1[1] = w0 >> 16; f1[2] = lo16(w1); f1[3] = w1 >> 16; f1[4] = lo16(w2);
    m = w2 >> 16;
    f2[0] = lo16(w3); f2[1] = w3 >> 16; f2[2] = lo16(w4); f2[3] = w4 >> 16; f2[4] = lo16(w5);
    m = (m < 0 || m >= B) ? b : m;
#pragma unroll
    for (int k = 0; k < 4; ++k) {        // make_state_map without offsets
      const int a0 = f1[k], s = f2[k];
      const int len1 = f1[k + 1] - a0, len2 = f2[k + 1] - s;
      int n = len1 < len2 ? len1 : len2;
      if (a0 < 0 || s < 0) n = 0;
      if (n > T - a0) n = T - a0;
      if (n > T - s) n = T - s;
      if (n < 0) n = 0;
      sm.a[k] = a0;
      sm.n[k] = n;
      sm.delta[k] = s - a0;
    }
    return true;
  };
  StateMap sm0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { sm0.a[k] = 0; sm0.n[k] = 0; sm0.delta[k] = 0; }
  tq_body<CG, UT, true>(x, y, lam, oml, knots_dev + (size_t)b * nkc, spline_op, n_knots, C, T, b, b, sm0, lds,
                        await);
  if (blockIdx.x == 1 && threadIdx.x == 0) { PCGMIX_ACLOCK(b, 3); }
}

// Elements of one sample's plane per block: 1024 * U.  Fatter blocks amortise the block
// prologue (dependent index loads, spline records) and keep more loads in flight per lane;
// thinner ones waste fewer lanes on the last chunk of a short plane.  Chosen from measurements
// on MI355X ("Block shape"); PCGMIX_MIX_UNROLL overrides it for tuning runs.
static int choose_unroll(long long B, long long plane, bool warp) {
  int U = plane >= 8192 ? 2 : 1;
  if (warp && plane >= 16384 && B * plane >= (64LL << 20)) U = 4;
  if (const char* env = getenv("PCGMIX_MIX_UNROLL")) {
    const int v = atoi(env);
    if (v == 1 || v == 2 || v == 4) U = v;
  }
  return U;
}

}  // namespace pcgmix

extern "C" int pcgmix_mix_variant(int B, int C, int T, int warp, int aligned16, int* vec,
                                  int* unroll) {
  if (B < 0 || C <= 0 || T <= 0 || !vec || !unroll) return hipErrorInvalidValue;
  const bool vec4 = (T % 4 == 0) && aligned16;
  *vec = vec4 ? 4 : 1;
  *unroll = vec4 ? pcgmix::choose_unroll(B, (long long)C * T, warp != 0) : 1;
  return hipSuccess;
}

// Which instantiation launch_mix_warp launches — the ONE place that decides it (the launcher and the
// name query below both ask here, so a benchmark cannot pair its timing with the wrong rocprofv3 row).
namespace pcgmix {
struct MixVariant {
  bool tq;         // mix_warp_tq_kernel<CG, UT> (one lane = one position quad for all channels)
  int CG, UT;      // tq: channels whose loads a lane keeps in flight; quads per lane
  int vec, U;      // otherwise mix_warp_kernel<vec, warp, U>
  size_t lds_tq;   // tq: records | thresholds (padded to 8 bytes) | staged operator | staged knots
};
static MixVariant choose_mix_variant(long long B, int C, int T, bool vec4, int n_knots,
                                     bool zero_rect) {
  MixVariant v{};
  const bool warp = n_knots > 0;
  const long long plane = (long long)C * T;
  v.vec = vec4 ? 4 : 1;
  v.U = vec4 ? choose_unroll(B, plane, warp) : 1;
  if (!(vec4 && warp && !zero_rect && C <= 64)) return v;
  v.lds_tq = sizeof(double) * ((size_t)C * (n_knots - 1) * kRec + (size_t)n_knots +
                               4 * (size_t)(n_knots - 1) * n_knots + (size_t)n_knots * C) +
             sizeof(int) * (size_t)((n_knots + 1) & ~1);
  if (v.lds_tq > 64 * 1024) return v;
  v.tq = true;
  // two quads per lane only where the registers allow it (CG = 4, UT = 2 needs 256 VGPRs)
  v.UT = (C % 4 != 0 && T >= 4096 && B * plane >= (64LL << 20)) ? 2 : 1;
  // channels whose loads a lane keeps in flight.  Measured at C = 4 (MI355X, back to back):
  // (256,4,5000) 14.2 / 11.9 / 12.1 us for 4 / 2 / 1, saturating 16384x4x5000 672 / 596 / 573 us
  // (fewer registers, more waves: 144 / 94 / 76 VGPRs)
  v.CG = (C % 2 == 0 && B * plane < (64LL << 20)) ? 2 : 1;
  return v;
}
}  // namespace pcgmix

// Name of the kernel instantiation pcgmix_mix_warp_f32 launches for this problem (what rocprofv3
// lists), so that a benchmark can match its own timing with the profiler's rows.  n_knots = 0: no
// warp; zero_rect: the call carries 2D mask rectangles.
extern "C" int pcgmix_mix_kernel_name(int B, int C, int T, int n_knots, int zero_rect,
                                      int aligned16, char* buf, int buf_len) {
  if (B < 0 || C <= 0 || T <= 0 || n_knots < 0 || n_knots == 1 || n_knots > 64 || !buf ||
      buf_len <= 0)
    return hipErrorInvalidValue;
  const bool vec4 = (T % 4 == 0) && aligned16;
  const pcgmix::MixVariant v = pcgmix::choose_mix_variant(B, C, T, vec4, n_knots, zero_rect != 0);
  if (v.tq)
    snprintf(buf, (size_t)buf_len, "pcgmix::mix_warp_tq_kernel<%d, %d>", v.CG, v.UT);
  else
    snprintf(buf, (size_t)buf_len, "pcgmix::mix_warp_kernel<%d, %s, %d>", v.vec,
             n_knots ? "true" : "false", v.U);
  return hipSuccess;
}

extern "C" int pcgmix_mix_warp_f32(const float* x, float* y, const int32_t* frames,
                                   const int32_t* mix_idx, const int32_t* off, float lam,
                                   const double* knots, const double* spline_op, int n_knots,
                                   const int32_t* zero_rect, int B, int C, int T,
                                   pcgmix_stream_t stream) {
  return pcgmix::launch_mix_warp(x, y, frames, mix_idx, off, lam, knots, spline_op, n_knots,
                                 zero_rect, B, C, T, reinterpret_cast<hipStream_t>(stream), nullptr,
                                 nullptr, 0);
}

extern "C" int pcgmix_mix_karg_f32(const float* x, float* y, const int16_t* frames16,
                                   const int16_t* mix16, float lam, int B, int C, int T,
                                   pcgmix_stream_t stream) {
  return pcgmix::launch_mix_karg(x, y, frames16, mix16, lam, B, C, T,
                                 reinterpret_cast<hipStream_t>(stream));
}

extern "C" int pcgmix_mix_karg_variant(int B, int C, int T, int* unroll) {
  if (B <= 0 || B > pcgmix::kPackB || C <= 0 || T <= 0 || T > 32767 || (T & 3) || !unroll) return 0;
  *unroll = pcgmix::choose_unroll(B, (long long)C * T, false) >= 2 ? 2 : 1;
  return 1;
}

int pcgmix::launch_mix_karg(const float* x, float* y, const int16_t* frames16, const int16_t* mix16,
                            float lam, int B, int C, int T, hipStream_t s, const void* pay_host,
                            int pay_bytes, void* pay_dst) {
  using namespace pcgmix;
  if (pay_bytes < 0 || pay_bytes > kPackPayBytes ||
      (pay_bytes > 0 && (!pay_host || !pay_dst || (reinterpret_cast<uintptr_t>(pay_dst) & 15))))
    return hipErrorInvalidValue;
  PayPack pay;
  pay.n16 = (pay_bytes + 15) / 16;
  if (pay_bytes) {
    memset(pay.w, 0, sizeof(pay.w));
    memcpy(pay.w, pay_host, (size_t)pay_bytes);
  }
  if (!x || !y || !frames16 || !mix16 || x == y || B <= 0 || B > kPackB || C <= 0 || T <= 0 ||
      T > 32767 || (T & 3) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15))
    return hipErrorInvalidValue;
  const long long plane = (long long)C * T;
  if (plane > 0x7fffffffLL) return hipErrorInvalidValue;
  IdxPack pack;
  int16_t* p16 = reinterpret_cast<int16_t*>(pack.w);
  memcpy(p16, frames16, sizeof(int16_t) * (size_t)B * 5);
  memcpy(p16 + kPackB * 5, mix16, sizeof(int16_t) * (size_t)B);
  const int U = choose_unroll(B, plane, false);
  const int epb = kThreads * 4 * (U == 4 ? 2 : U);
  const unsigned chunks = (unsigned)((plane + epb - 1) / epb);
  const float oml = 1.0f - lam;
  dim3 grid(chunks, (unsigned)B), block(kThreads);
  if (U >= 2)
    hipLaunchKernelGGL((mix_warp_karg_kernel<false, 2>), grid, block, 0, s, x, y, pack, lam, oml,
                       nullptr, nullptr, 0, B, C, T, epb, pay, static_cast<uint4*>(pay_dst));
  else
    hipLaunchKernelGGL((mix_warp_karg_kernel<false, 1>), grid, block, 0, s, x, y, pack, lam, oml,
                       nullptr, nullptr, 0, B, C, T, epb, pay, static_cast<uint4*>(pay_dst));
  return (int)hipGetLastError();
}

#ifdef PCGMIX_PHASE_CLOCK
extern "C" int pcgmix_armed_phase_clock(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcgmix::g_armed_clock), sizeof(long long) * 2 * (pcgmix::kPackB + 1) * 4);
}
#endif

int pcgmix::launch_mix_armed(const float* x, float* y, const ArmedArgs& a, int B, int C, int T,
                             hipStream_t s, const void* pay_host, int pay_bytes, void* pay_dst) {
  using namespace pcgmix;
  if (pay_bytes < 0 || pay_bytes > kPackPayBytes ||
      (pay_bytes > 0 && (!pay_host || !pay_dst || (reinterpret_cast<uintptr_t>(pay_dst) & 15))))
    return hipErrorInvalidValue;
  if (!x || !y || x == y || B <= 0 || B > kPackB || C <= 0 || T <= 0 || T > 32767 || (T & 3) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) || !a.ohe || a.K <= 0 ||
      !a.lab64 || a.K > 256 || !a.rec_h || !a.rec_d || !a.abort_h || a.seq == 0 || (a.seq & kArmedAbort) ||
      ((reinterpret_cast<uintptr_t>(a.rec_h) | reinterpret_cast<uintptr_t>(a.rec_d) |
        reinterpret_cast<uintptr_t>(a.lab64)) & 7))
    return hipErrorInvalidValue;
  const long long plane = (long long)C * T;
  if (plane > 0x7fffffffLL) return hipErrorInvalidValue;
  PayPack pay;
  pay.n16 = (pay_bytes + 15) / 16;
  if (pay_bytes) {
    memset(pay.w, 0, sizeof(pay.w));
    memcpy(pay.w, pay_host, (size_t)pay_bytes);
  }
  const int U = choose_unroll(B, plane, false);
  const int epb = kThreads * 4 * (U == 4 ? 2 : U);
  const unsigned chunks = (unsigned)((plane + epb - 1) / epb);
  // every block resident at once: at most 2,048 blocks of four waves, each taking its sample's chunks
  // blockIdx.x, blockIdx.x + gridDim.x, ...
  unsigned per_sample = chunks;
  while (per_sample > 1 && per_sample * (unsigned)B > 2048u) per_sample = (per_sample + 1) / 2;
  dim3 grid(per_sample, (unsigned)B), block(kThreads);
  if (U >= 2)
    hipLaunchKernelGGL((mix_armed_kernel<2>), grid, block, 0, s, x, y, a, B, C, T, epb,
                       (int)chunks, pay, static_cast<uint4*>(pay_dst));
  else
    hipLaunchKernelGGL((mix_armed_kernel<1>), grid, block, 0, s, x, y, a, B, C, T, epb,
                       (int)chunks, pay, static_cast<uint4*>(pay_dst));
  return (int)hipGetLastError();
}

// 1 if launch_mix_tq_armed takes this problem (the tq instantiation exists for it and its grid is resident)
int pcgmix::mix_tq_armed_ok(int B, int C, int T, int n_knots) {
  if (B <= 0 || B > kPackB || C <= 0 || T <= 0 || T > 32767 || (T & 3) || n_knots < 2 || n_knots > 64) return 0;
  const MixVariant mv = choose_mix_variant(B, C, T, true, n_knots, false);
  if (!mv.tq) return 0;
  const long long blocks = (long long)((T + kThreads * 4 * mv.UT - 1) / (kThreads * 4 * mv.UT)) * B;
  return blocks <= 2048 && mv.lds_tq <= 32 * 1024;
}

int pcgmix::launch_mix_tq_armed(const float* x, float* y, const ArmedArgs& a, float lam,
                                const double* knots_host, double* knots_dev, const double* spline_op, int n_knots, int B,
                                int C, int T, hipStream_t s, const void* pay_host, int pay_bytes,
                                void* pay_dst) {
  using namespace pcgmix;
  if (pay_bytes < 0 || pay_bytes > kPackPayBytes ||
      (pay_bytes > 0 && (!pay_host || !pay_dst || (reinterpret_cast<uintptr_t>(pay_dst) & 15))))
    return hipErrorInvalidValue;
  if (!x || !y || x == y || !knots_host || !knots_dev || !spline_op || !mix_tq_armed_ok(B, C, T, n_knots) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) || !a.ohe || a.K <= 0 ||
      !a.lab64 || a.K > 256 || !a.rec_h || !a.rec_d || !a.abort_h || a.seq == 0 || (a.seq & kArmedAbort) ||
      ((reinterpret_cast<uintptr_t>(a.rec_h) | reinterpret_cast<uintptr_t>(a.rec_d) |
        reinterpret_cast<uintptr_t>(a.lab64) | reinterpret_cast<uintptr_t>(knots_host)) & 7))
    return hipErrorInvalidValue;
  PayPack pay;
  pay.n16 = (pay_bytes + 15) / 16;
  if (pay_bytes) {
    memset(pay.w, 0, sizeof(pay.w));
    memcpy(pay.w, pay_host, (size_t)pay_bytes);
  }
  const MixVariant mv = choose_mix_variant(B, C, T, true, n_knots, false);
  const float oml = 1.0f - lam;
  dim3 grid((unsigned)((T + kThreads * 4 * mv.UT - 1) / (kThreads * 4 * mv.UT)), (unsigned)B), block(kThreads);
#define PCGMIX_LAUNCH_TQA(CGV, UTV)                                                                  \
  hipLaunchKernelGGL((mix_warp_tq_armed_kernel<CGV, UTV>), grid, block, mv.lds_tq, s, x, y, a, lam,  \
                     oml, knots_host, knots_dev, spline_op, n_knots, B, C, T, pay,                   \
                     static_cast<uint4*>(pay_dst))
  if (mv.CG == 4) { if (mv.UT == 2) PCGMIX_LAUNCH_TQA(4, 2); else PCGMIX_LAUNCH_TQA(4, 1); }
  else if (mv.CG == 2) { if (mv.UT == 2) PCGMIX_LAUNCH_TQA(2, 2); else PCGMIX_LAUNCH_TQA(2, 1); }
  else { if (mv.UT == 2) PCGMIX_LAUNCH_TQA(1, 2); else PCGMIX_LAUNCH_TQA(1, 1); }
#undef PCGMIX_LAUNCH_TQA
  return (int)hipGetLastError();
}

int pcgmix::launch_mix_warp(const float* x, float* y, const int32_t* frames, const int32_t* mix_idx,
                            const int32_t* off, float lam, const double* knots,
                            const double* spline_op, int n_knots, const int32_t* zero_rect, int B,
                            int C, int T, hipStream_t s, const void* pay_src_v, void* pay_dst_v,
                            int pay_n16, const float2* disp_part, const int16_t* partners16) {
  using namespace pcgmix;
  const uint4* pay_src = static_cast<const uint4*>(pay_src_v);
  uint4* pay_dst = static_cast<uint4*>(pay_dst_v);
  if (pay_n16 < 0 || (pay_n16 > 0 && (!pay_src || !pay_dst ||
                                      ((reinterpret_cast<uintptr_t>(pay_src) |
                                        reinterpret_cast<uintptr_t>(pay_dst)) & 15))))
    return hipErrorInvalidValue;
  const PartnerPack pk = make_partner_pack(partners16, B);
  if (!x || !y || !frames || (!mix_idx && !pk.n) || x == y) return hipErrorInvalidValue;
  if (B < 0 || C <= 0 || T <= 0) return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  const bool warp = knots != nullptr;
  if (warp && (!spline_op || n_knots < 2 || n_knots > 64)) return hipErrorInvalidValue;
  const long long plane = (long long)C * T;
  if (plane > 0x7fffffffLL || B > kBatchPerGridZ * 1024) return hipErrorInvalidValue;

  const bool vec4 = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const int U = choose_unroll(B, plane, warp);
  const int epb = kThreads * 4 * U;
  const unsigned chunks = (unsigned)((plane + epb - 1) / epb);
  size_t lds = 0;
  if (warp) {
    const int nch = epb / T + 2;
    lds = sizeof(double) * (size_t)nch * (n_knots - 1) * kRec + sizeof(int) * (size_t)n_knots;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
  }
  const float oml = 1.0f - lam;  // float32 subtraction, as torch's (1 - lam) on a float32 tensor

  const unsigned gy = (unsigned)(B < kBatchPerGridZ ? B : kBatchPerGridZ);
  const unsigned gz = (unsigned)((B + kBatchPerGridZ - 1) / kBatchPerGridZ);
  dim3 grid(chunks, gy, gz), block(kThreads);
#define PCGMIX_LAUNCH(V, W, UU)                                                              \
  hipLaunchKernelGGL((mix_warp_kernel<V, W, UU>), grid, block, lds, s, x, y, frames, mix_idx, \
                     off, lam, oml, knots, spline_op, n_knots, zero_rect, B, C, T, epb, pay_src,  \
                     pay_dst, pay_n16, disp_part, pk)
#define PCGMIX_LAUNCH_U(W)                                \
  do {                                                    \
    if (U == 4) PCGMIX_LAUNCH(4, W, 4);                   \
    else if (U == 2) PCGMIX_LAUNCH(4, W, 2);              \
    else PCGMIX_LAUNCH(4, W, 1);                          \
  } while (0)
  const MixVariant mv = choose_mix_variant(B, C, T, vec4, warp ? n_knots : 0, zero_rect != nullptr);
  if (mv.tq) {
    // one lane = one position quad for all channels (mix_warp_tq_kernel)
    const int UT = mv.UT, CG = mv.CG;
    const size_t lds_tq = mv.lds_tq;
    dim3 grid_tq((unsigned)((T + kThreads * 4 * UT - 1) / (kThreads * 4 * UT)), gy, gz);
#define PCGMIX_LAUNCH_TQ(CGV, UTV)                                                                  \
  hipLaunchKernelGGL((mix_warp_tq_kernel<CGV, UTV>), grid_tq, block, lds_tq, s, x, y, frames,        \
                     mix_idx, off, lam, oml, knots, spline_op, n_knots, B, C, T, pay_src, pay_dst,  \
                     pay_n16, disp_part, pk)
    if (CG == 4) { if (UT == 2) PCGMIX_LAUNCH_TQ(4, 2); else PCGMIX_LAUNCH_TQ(4, 1); }
    else if (CG == 2) { if (UT == 2) PCGMIX_LAUNCH_TQ(2, 2); else PCGMIX_LAUNCH_TQ(2, 1); }
    else { if (UT == 2) PCGMIX_LAUNCH_TQ(1, 2); else PCGMIX_LAUNCH_TQ(1, 1); }
#undef PCGMIX_LAUNCH_TQ
    return (int)hipGetLastError();
  }
  if (vec4) {
    if (warp) PCGMIX_LAUNCH_U(true); else PCGMIX_LAUNCH_U(false);
  } else {
    if (warp) PCGMIX_LAUNCH(1, true, 1); else PCGMIX_LAUNCH(1, false, 1);
  }
#undef PCGMIX_LAUNCH_U
#undef PCGMIX_LAUNCH
  return (int)hipGetLastError();
}
