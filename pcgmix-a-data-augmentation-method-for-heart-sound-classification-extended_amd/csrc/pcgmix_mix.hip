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
                                                   int T) {
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

// Spline value at integer position t for the piece table `cf` (4 doubles per piece, scipy
// layout c0..c3 with c0 the cubic term).  Order of operations as scipy's PPoly evaluation.
__device__ __forceinline__ double spline_at(const double* __restrict__ cf,
                                            const double* __restrict__ brk, int n_knots, int t) {
  double td = (double)t;
  int p = 0;
  for (int i = 1; i <= n_knots - 2; ++i) p += (brk[i] <= td) ? 1 : 0;
  double s = __dsub_rn(td, brk[p]);
  const double* c = cf + p * 4;
  double r = c[3];
  r = __dadd_rn(r, __dmul_rn(c[2], s));
  double z = __dmul_rn(s, s);
  r = __dadd_rn(r, __dmul_rn(c[1], z));
  z = __dmul_rn(z, s);
  r = __dadd_rn(r, __dmul_rn(c[0], z));
  return r;
}

// VEC = 4: T % 4 == 0, rows are 16-byte aligned.  VEC = 1: any T.
template <int VEC, bool WARP>
__global__ __launch_bounds__(kThreads) void mix_warp_kernel(
    const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ frames,
    const int32_t* __restrict__ mix_idx, const int32_t* __restrict__ off, float lam, float oml,
    const double* __restrict__ knots, const double* __restrict__ spline_op, int n_knots, int B,
    int C, int T, int epb) {
  extern __shared__ __align__(16) double lds[];  // [n_knots] break points, then coef tables

  const int b = blockIdx.z * kBatchPerGridZ + blockIdx.y;
  if (b >= B) return;  // block-uniform
  int m = mix_idx[b];
  m = (m < 0 || m >= B) ? b : m;  // memory safety; validated on the host as well
  const StateMap sm = make_state_map(frames, off, b, m, T);

  const int plane = C * T;
  const int chunk0 = blockIdx.x * epb;
  const size_t own_base = (size_t)b * plane;
  const size_t par_base = (size_t)m * plane;

  int c_lo = 0;
  if (WARP) {
    // Coefficient tables for the channels this chunk touches: coef = op * knots[b,:,c].
    c_lo = chunk0 / T;
    int last = chunk0 + epb - 1;
    if (last > plane - 1) last = plane - 1;
    const int c_hi = last / T;
    const int pieces4 = (n_knots - 1) * 4;
    double* brk = lds;
    double* cf = lds + n_knots;
    for (int i = threadIdx.x; i < n_knots; i += kThreads) brk[i] = spline_op[i];
    const int total = (c_hi - c_lo + 1) * pieces4;
    for (int i = threadIdx.x; i < total; i += kThreads) {
      const int c = c_lo + i / pieces4;
      const int row = i % pieces4;
      const double* mrow = spline_op + n_knots + (size_t)row * n_knots;
      double acc = 0.0;
      for (int j = 0; j < n_knots; ++j)
        acc = __dadd_rn(acc, __dmul_rn(mrow[j], knots[((size_t)b * n_knots + j) * C + c]));
      cf[i] = acc;
    }
    __syncthreads();
  }

  for (int i = chunk0 + threadIdx.x * VEC; i < chunk0 + epb && i < plane; i += kThreads * VEC) {
    const int c = i / T;
    const int t0 = i - c * T;
    float own[VEC], out[VEC];
    if constexpr (VEC == 4) {
      float4_a v = *reinterpret_cast<const float4_a*>(x + own_base + i);
      own[0] = v.x; own[1] = v.y; own[2] = v.z; own[3] = v.w;
    } else {
      own[0] = x[own_base + i];
    }
    const float* prow = x + par_base + (size_t)c * T;

    bool hit[VEC];
    int d[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) d[e] = blend_shift(sm, t0 + e, hit[e]);

    bool uniform = false;
    if constexpr (VEC == 4)
      uniform = hit[0] && hit[1] && hit[2] && hit[3] && d[0] == d[1] && d[0] == d[2] &&
                d[0] == d[3];
    if (uniform) {
      if constexpr (VEC == 4) {
        float4_u p = *reinterpret_cast<const float4_u*>(prow + t0 + d[0]);
        out[0] = blend(own[0], p.x, lam, oml);
        out[1] = blend(own[1], p.y, lam, oml);
        out[2] = blend(own[2], p.z, lam, oml);
        out[3] = blend(own[3], p.w, lam, oml);
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float o = own[e];
        if (hit[e]) o = blend(o, prow[t0 + e + d[e]], lam, oml);
        out[e] = o;
      }
    }

    if (WARP) {
      const double* brk = lds;
      const double* cf = lds + n_knots + (size_t)(c - c_lo) * (n_knots - 1) * 4;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double w = spline_at(cf, brk, n_knots, t0 + e);
        out[e] = __double2float_rn(__dmul_rn((double)out[e], w));
      }
    }

    if constexpr (VEC == 4) {
      float4_a v;
      v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
      *reinterpret_cast<float4_a*>(y + own_base + i) = v;
    } else {
      y[own_base + i] = out[0];
    }
  }
}

}  // namespace pcgmix

extern "C" int pcgmix_mix_warp_f32(const float* x, float* y, const int32_t* frames,
                                   const int32_t* mix_idx, const int32_t* off, float lam,
                                   const double* knots, const double* spline_op, int n_knots,
                                   int B, int C, int T, pcgmix_stream_t stream) {
  using namespace pcgmix;
  if (!x || !y || !frames || !mix_idx || x == y) return hipErrorInvalidValue;
  if (B < 0 || C <= 0 || T <= 0) return hipErrorInvalidValue;
  if (B == 0) return hipSuccess;
  const bool warp = knots != nullptr;
  if (warp && (!spline_op || n_knots < 2 || n_knots > 64)) return hipErrorInvalidValue;
  const long long plane = (long long)C * T;
  if (plane > 0x7fffffffLL || B > kBatchPerGridZ * 1024) return hipErrorInvalidValue;

  const bool vec4 = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const int epb = kThreads * 4;  // elements of one sample's plane per block
  const unsigned chunks = (unsigned)((plane + epb - 1) / epb);
  size_t lds = 0;
  if (warp) {
    const int nch = epb / T + 2;
    lds = sizeof(double) * ((size_t)n_knots + (size_t)nch * (n_knots - 1) * 4);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
  }
  const float oml = 1.0f - lam;  // float32 subtraction, as torch's (1 - lam) on a float32 tensor

  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned gy = (unsigned)(B < kBatchPerGridZ ? B : kBatchPerGridZ);
  const unsigned gz = (unsigned)((B + kBatchPerGridZ - 1) / kBatchPerGridZ);
  dim3 grid(chunks, gy, gz), block(kThreads);
#define PCGMIX_LAUNCH(V, W)                                                                   \
  hipLaunchKernelGGL((mix_warp_kernel<V, W>), grid, block, lds, s, x, y, frames, mix_idx, off, \
                     lam, oml, knots, spline_op, n_knots, B, C, T, epb)
  if (vec4) {
    if (warp) PCGMIX_LAUNCH(4, true); else PCGMIX_LAUNCH(4, false);
  } else {
    if (warp) PCGMIX_LAUNCH(1, true); else PCGMIX_LAUNCH(1, false);
  }
#undef PCGMIX_LAUNCH
  return (int)hipGetLastError();
}
