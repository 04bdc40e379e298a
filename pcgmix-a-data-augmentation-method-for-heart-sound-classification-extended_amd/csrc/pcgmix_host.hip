// pcgmix_host.hip — host-only entry points of libpcgmix_hip.so (no device code).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "pcgmix_kernels.h"

extern "C" int pcgmix_abi_version(void) { return PCGMIX_ABI_VERSION; }

extern "C" const char* pcgmix_error_string(int err) {
  return hipGetErrorString(static_cast<hipError_t>(err));
}

extern "C" int pcgmix_spline_operator_size(int n_knots) {
  return n_knots < 2 ? 0 : n_knots + 4 * (n_knots - 1) * n_knots;
}

namespace {

// Solve A s = rhs (n x n dense, partial pivoting).  n <= 64.
bool solve_dense(std::vector<double>& A, std::vector<double>& rhs, int n) {
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r)
      if (std::fabs(A[r * n + col]) > std::fabs(A[piv * n + col])) piv = r;
    if (A[piv * n + col] == 0.0) return false;
    if (piv != col) {
      for (int c = 0; c < n; ++c) std::swap(A[piv * n + c], A[col * n + c]);
      std::swap(rhs[piv], rhs[col]);
    }
    for (int r = col + 1; r < n; ++r) {
      const double f = A[r * n + col] / A[col * n + col];
      if (f == 0.0) continue;
      for (int c = col; c < n; ++c) A[r * n + c] -= f * A[col * n + c];
      rhs[r] -= f * rhs[col];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double acc = rhs[r];
    for (int c = r + 1; c < n; ++c) acc -= A[r * n + c] * rhs[c];
    rhs[r] = acc / A[r * n + r];
  }
  return true;
}

// Piecewise-cubic coefficients (scipy PPoly layout, c[j][p] multiplies (t-x[p])^(3-j)) of the
// not-a-knot spline through (x[i], yv[i]).  Same equations as scipy's CubicSpline: unknowns
// are the first derivatives at the knots; interior rows are the C2 conditions, the two end
// rows the not-a-knot conditions; n == 3 is the single parabola, n == 2 the straight line.
bool notaknot_coeffs(const std::vector<double>& x, const std::vector<double>& yv,
                     std::vector<double>& coef /* (n-1)*4 */) {
  const int n = (int)x.size();
  std::vector<double> dx(n - 1), slope(n - 1), s(n);
  for (int i = 0; i < n - 1; ++i) {
    dx[i] = x[i + 1] - x[i];
    slope[i] = (yv[i + 1] - yv[i]) / dx[i];
  }
  if (n == 2) {
    s[0] = s[1] = slope[0];
  } else {
    std::vector<double> A((size_t)n * n, 0.0), rhs(n, 0.0);
    if (n == 3) {
      A[0] = 1.0; A[1] = 1.0;
      A[3] = dx[1]; A[4] = 2.0 * (dx[0] + dx[1]); A[5] = dx[0];
      A[7] = 1.0; A[8] = 1.0;
      rhs[0] = 2.0 * slope[0];
      rhs[1] = 3.0 * (dx[0] * slope[1] + dx[1] * slope[0]);
      rhs[2] = 2.0 * slope[1];
    } else {
      for (int i = 1; i < n - 1; ++i) {
        A[i * n + i - 1] = dx[i];
        A[i * n + i] = 2.0 * (dx[i - 1] + dx[i]);
        A[i * n + i + 1] = dx[i - 1];
        rhs[i] = 3.0 * (dx[i] * slope[i - 1] + dx[i - 1] * slope[i]);
      }
      double d = x[2] - x[0];
      A[0] = dx[1];
      A[1] = d;
      rhs[0] = ((dx[0] + 2.0 * d) * dx[1] * slope[0] + dx[0] * dx[0] * slope[1]) / d;
      d = x[n - 1] - x[n - 3];
      A[(n - 1) * n + n - 1] = dx[n - 3];
      A[(n - 1) * n + n - 2] = d;
      rhs[n - 1] = (dx[n - 2] * dx[n - 2] * slope[n - 3] +
                    (2.0 * d + dx[n - 2]) * dx[n - 3] * slope[n - 2]) / d;
    }
    if (!solve_dense(A, rhs, n)) return false;
    s = rhs;
  }
  for (int p = 0; p < n - 1; ++p) {
    const double t = (s[p] + s[p + 1] - 2.0 * slope[p]) / dx[p];
    coef[p * 4 + 0] = t / dx[p];
    coef[p * 4 + 1] = (slope[p] - s[p]) / dx[p] - t;
    coef[p * 4 + 2] = s[p];
    coef[p * 4 + 3] = yv[p];
  }
  return true;
}

}  // namespace

extern "C" int pcgmix_spline_operator_f64(int T, int n_knots, double* op) {
  if (!op || n_knots < 2 || n_knots > 64 || T < 2) return hipErrorInvalidValue;
  const int n = n_knots;
  // numpy.linspace(0, T-1, n): step = (T-1)/(n-1); y[i] = i*step; last point forced to T-1.
  std::vector<double> x(n);
  const double step = (double)(T - 1) / (double)(n - 1);
  for (int i = 0; i < n; ++i) x[i] = (double)i * step;
  x[n - 1] = (double)(T - 1);
  for (int i = 0; i < n; ++i) op[i] = x[i];
  double* M = op + n;
  std::vector<double> e(n), coef((size_t)(n - 1) * 4);
  for (int i = 0; i < n; ++i) {
    std::fill(e.begin(), e.end(), 0.0);
    e[i] = 1.0;
    if (!notaknot_coeffs(x, e, coef)) return hipErrorInvalidValue;
    for (int r = 0; r < (n - 1) * 4; ++r) M[(size_t)r * n + i] = coef[r];
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// CPython's random.Random, restated: MT19937 (Matsumoto & Nishimura reference algorithm, as
// Modules/_randommodule.c uses it) plus Lib/random.py's sample()/_randbelow()/uniform().
namespace {

struct PyRandom {
  uint32_t mt[624];
  int idx;

  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < 624; ++i)
      mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }

  void init_by_array(const uint32_t* key, int len) {
    init_genrand(19650218u);
    int i = 1, j = 0;
    for (int k = (624 > len ? 624 : len); k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
      ++i; ++j;
      if (i >= 624) { mt[0] = mt[623]; i = 1; }
      if (j >= len) j = 0;
    }
    for (int k = 623; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
      ++i;
      if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
  }

  // random.seed(int): key = little-endian 32-bit words of abs(seed), at least one word
  explicit PyRandom(uint64_t seed) {
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    init_by_array(key, key[1] ? 2 : 1);
  }

  uint32_t next32() {
    if (idx >= 624) {
      int kk;
      for (kk = 0; kk < 624 - 397; ++kk) {
        uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      for (; kk < 623; ++kk) {
        uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
      mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }

  double random() {  // 53-bit resolution, genrand_res53
    const uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
  }

  // getrandbits(k), 1 <= k <= 64: little-endian 32-bit words, the last one shifted down
  uint64_t getrandbits(int k) {
    if (k <= 32) return next32() >> (32 - k);
    const uint64_t lo = next32();
    const uint64_t hi = next32() >> (64 - k);
    return lo | (hi << 32);
  }

  // Random._randbelow_with_getrandbits(n), n >= 1
  uint64_t randbelow(uint64_t n) {
    int k = 0;
    for (uint64_t v = n; v; v >>= 1) ++k;  // n.bit_length()
    uint64_t r = getrandbits(k);
    while (r >= n) r = getrandbits(k);
    return r;
  }
};

}  // namespace

extern "C" double pcgmix_py_uniform01(uint64_t seed) {
  PyRandom r(seed);
  return 0.0 + (1.0 - 0.0) * r.random();  // a + (b - a) * random()
}

extern "C" int64_t pcgmix_py_randint0(uint64_t seed, int64_t hi) {
  if (hi < 0) return -1;
  PyRandom r(seed);
  return (int64_t)r.randbelow((uint64_t)hi + 1u);  // randrange(0, hi + 1)
}

extern "C" int pcgmix_partner_permutation_i64(const int32_t* group_id, int B, int n_groups,
                                              uint64_t seed, int64_t* mix) {
  if (!group_id || !mix || B < 0 || n_groups < 0) return hipErrorInvalidValue;
  std::vector<std::vector<int64_t>> members((size_t)n_groups);
  for (int b = 0; b < B; ++b) {
    const int g = group_id[b];
    if (g < 0 || g >= n_groups) return hipErrorInvalidValue;
    members[(size_t)g].push_back(b);
  }
  std::vector<int64_t> pool;
  for (const auto& idx : members) {
    const size_t n = idx.size();
    if (!n) continue;
    PyRandom rng(seed);  // a fresh Random(seed) per group, as the reference
    pool = idx;
    // sample(population, k = n): pool branch of Lib/random.py
    for (size_t i = 0; i < n; ++i) {
      const uint64_t j = rng.randbelow((uint64_t)(n - i));
      mix[idx[i]] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// Validate the boundaries and pack the per-step index block the kernels read:
//   int32 frames[B][5] | int32 mix[B] | int32 rand_off[B][4] (optional) | int32 rect[B][4] (optional)
// Returns 0, or 1 frames not monotone / negative, 2 cycle end beyond T, 3 partner out of range.
extern "C" int pcgmix_pack_plan_i32(const int64_t* frames, const int64_t* mix,
                                    const int32_t* rand_off, const int32_t* rect, int B, int T,
                                    int32_t* out) {
  if (!frames || !mix || !out || B < 0) return hipErrorInvalidValue;
  int32_t* f = out;
  int32_t* m = out + (size_t)B * 5;
  for (int b = 0; b < B; ++b) {
    const int64_t* r = frames + (size_t)b * 5;
    if (r[0] < 0) return 1;
    for (int k = 0; k < 4; ++k)
      if (r[k + 1] < r[k]) return 1;
    if (r[4] > T) return 2;
    for (int k = 0; k < 5; ++k) f[b * 5 + k] = (int32_t)r[k];
    if (mix[b] < 0 || mix[b] >= B) return 3;
    m[b] = (int32_t)mix[b];
  }
  int32_t* o = m + B;
  if (rand_off) {
    for (int i = 0; i < B * 4; ++i) o[i] = rand_off[i];
    o += (size_t)B * 4;
  }
  if (rect)
    for (int i = 0; i < B * 4; ++i) o[i] = rect[i];
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The whole per-step host prologue of the plain PCGmix methods in one call: partner draw, boundary
// validation, packing into pinned staging, ONE async H2D copy, kernel launch.  In Python the same
// steps cost ~45 us per call (numpy temporaries, three ctypes crossings, torch's copy_ dispatch)
// for a 10 us kernel; here ~10 us.
extern "C" int pcgmix_splice_same_label_f32(const float* x, float* y, const int64_t* labels,
                                            const int64_t* frames, uint64_t step, float lam,
                                            const double* knots, const double* spline_op,
                                            int n_knots, void* staging, void* dev_idx,
                                            int64_t* mix_out, int B, int C, int T,
                                            pcgmix_stream_t stream) {
  if (!x || !y || !labels || !frames || !staging || !dev_idx || !mix_out || B <= 0 || C <= 0 ||
      T <= 0 || (knots && (!spline_op || n_knots < 2)))
    return hipErrorInvalidValue;
  // groups of equal label in order of first appearance (augmentations.py:500-510)
  std::vector<int64_t> keys;
  std::vector<std::vector<int64_t>> members;
  for (int b = 0; b < B; ++b) {
    size_t g = 0;
    while (g < keys.size() && keys[g] != labels[b]) ++g;
    if (g == keys.size()) {
      keys.push_back(labels[b]);
      members.emplace_back();
    }
    members[g].push_back(b);
  }
  // a fresh Random(step) per group: initialise the generator once, copy its state per group
  const PyRandom seeded(step);
  std::vector<int64_t> pool;
  for (const auto& idx : members) {
    const size_t n = idx.size();
    PyRandom rng = seeded;
    pool = idx;
    for (size_t i = 0; i < n; ++i) {                 // sample(population, k = n): pool branch
      const uint64_t j = rng.randbelow((uint64_t)(n - i));
      mix_out[idx[i]] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  }
  int32_t* st = static_cast<int32_t*>(staging);
  const int perr = pcgmix_pack_plan_i32(frames, mix_out, nullptr, nullptr, B, T, st);
  if (perr) return perr > 0 && perr <= 3 ? -perr : perr;   // -1..-3: malformed boundaries
  const size_t n_int = (size_t)B * 6, n_int_pad = (n_int + 1) & ~(size_t)1;
  size_t nbytes = n_int_pad * 4;
  const double* knots_dev = nullptr;
  if (knots) {
    const size_t nk = (size_t)B * n_knots * C;
    std::memcpy(reinterpret_cast<char*>(staging) + nbytes, knots, nk * sizeof(double));
    knots_dev = reinterpret_cast<const double*>(static_cast<char*>(dev_idx) + nbytes);
    nbytes += nk * sizeof(double);
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const hipError_t ce = hipMemcpyAsync(dev_idx, staging, nbytes, hipMemcpyHostToDevice, s);
  if (ce != hipSuccess) return (int)ce;
  const int32_t* d = static_cast<const int32_t*>(dev_idx);
  return pcgmix_mix_warp_f32(x, y, d, d + (size_t)B * 5, nullptr, lam, knots_dev, spline_op,
                             knots ? n_knots : 0, nullptr, B, C, T, stream);
}

// The label read-back the reference's signature forces (target_ohe lives on the device,
// augmentations.py:501) done here as well: D2H of the one-hot matrix into pinned memory, stream
// synchronisation, first-maximum argmax — then the call above.  Saves the torch dispatch of the
// copy and a second ctypes crossing per step.
extern "C" int pcgmix_splice_same_label_ohe_f32(const float* x, float* y,
                                                const int64_t* target_ohe_dev, int num_classes,
                                                int64_t* ohe_pinned, const int64_t* frames,
                                                uint64_t step, float lam, const double* knots,
                                                const double* spline_op, int n_knots,
                                                void* staging, void* dev_idx, int64_t* mix_out,
                                                int B, int C, int T, pcgmix_stream_t stream) {
  if (!target_ohe_dev || !ohe_pinned || num_classes <= 0 || B <= 0) return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * num_classes;
  hipError_t e = hipMemcpyAsync(ohe_pinned, target_ohe_dev, n * sizeof(int64_t),
                                hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize(s);
  if (e != hipSuccess) return (int)e;
  std::vector<int64_t> labels((size_t)B);
  for (int b = 0; b < B; ++b) {
    const int64_t* row = ohe_pinned + (size_t)b * num_classes;
    int best = 0;
    for (int c = 1; c < num_classes; ++c)
      if (row[c] > row[best]) best = c;               // first maximum, as torch.max / np.argmax
    labels[(size_t)b] = best;
  }
  return pcgmix_splice_same_label_f32(x, y, labels.data(), frames, step, lam, knots, spline_op,
                                      n_knots, staging, dev_idx, mix_out, B, C, T, stream);
}

extern "C" long long pcgmix_splice_staging_bytes(int B, int C, int n_knots) {
  if (B < 0 || C < 0 || n_knots < 0) return 0;
  const long long n_int_pad = ((long long)B * 6 + 1) & ~1ll;
  return n_int_pad * 4 + (long long)B * n_knots * C * 8;
}
