// pcgmix_host.hip — host-side entry points of libpcgmix_hip.so: the reference's host RNG streams
// restated in C, plan packing, and the per-device step context (pinned staging ring, device
// scratch ring, label read-back) behind pcgmix_augment_plain_f32.  The only device code here is
// the one-block label arg-max that feeds the read-back.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <pthread.h>
#include <time.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "pcgmix_kernels.h"

namespace {
// The step-context entry points stage through pinned slots they may have to (re)allocate, wait on
// events, spin on host-mapped memory and carry per-step index data in kernel ARGUMENTS (frozen at
// capture): none of that belongs in a stream capture.  They refuse a capturing stream up front,
// before any of their first-call work (hipHostMalloc / hipMalloc / hipMemcpy) can invalidate it.
bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
}  // namespace

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
    // init_genrand(19650218) is the same 624 words every time: computed once, copied after that
    // (a third of the 1871 dependent steps of a seeding)
    static const PyRandom* base = [] {
      PyRandom* b = static_cast<PyRandom*>(::operator new(sizeof(PyRandom)));
      b->init_genrand(19650218u);
      return b;
    }();
    std::memcpy(mt, base->mt, sizeof(mt));
    idx = 624;
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

  void regen() {
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
  // the regeneration the first draw from a freshly seeded state starts with: done once per seeding
  // instead of once per COPY of the seeded state (the partner draw copies it per label group)
  void twist() {
    if (idx >= 624) regen();
  }

  uint32_t next32() {
    if (idx >= 624) regen();
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
    const int k = 64 - __builtin_clzll(n);  // n.bit_length(), n >= 1
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
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;   // synchronises below
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


// ------------------------------------------------------------------------------------------------
// Per-device step context.  Everything the per-step prologue of a plain PCGmix method needs
// besides the batch itself lives here, allocated once: a ring of pinned staging buffers with
// their device twins and one event per slot (a slot is reused only after the kernel that read
// its device twin has finished), host-mapped memory for the label read-back, and the constant
// spline operators per (T, n_knots).  One context per device and host thread; not thread-safe.
namespace {

constexpr int kSlots = 8;
constexpr int kSlotGroup = 4;     // slots per completion event

struct Slot {
  char* pinned = nullptr;
  char* dev = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool busy = false;
};

// Class labels from a one-hot (B, K) int64 matrix: first maximum per row (torch.max / np.argmax
// semantics, augmentations.py:501), written to host-mapped memory, then a system-scope release
// of `token` into the flag word the host spins on.  One block: B is a few hundred.
// seed (device, may be nullptr): float (B, K), 1 at the row's label and 0 elsewhere — the
// gradient seed of the saliency model's class score (saliency.py:52-61), for free.
__global__ __launch_bounds__(256) void label_argmax_kernel(const int64_t* __restrict__ ohe, int K,
                                                           int B, int32_t* __restrict__ lab,
                                                           uint32_t* flag, uint32_t token,
                                                           float* __restrict__ seed) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int best = pcgmix::onehot_argmax(ohe, K, b);
    lab[b] = best;
    if (seed)
      for (int c = 0; c < K; ++c) seed[(size_t)b * K + c] = c == best ? 1.f : 0.f;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// label_argmax_kernel that also delivers the step's boundaries: frames (B,5) arrive as int16 in
// the kernel ARGUMENTS (2.5 KB; B <= kPackB, T <= 32767) and are written out as the int32 array
// the saliency post-processing, the search and the splice read — one launch instead of a launch
// and a 5 KB host-to-device copy.
struct FramePack {
  int32_t w[pcgmix::kPackB * 5 / 2];
};
__global__ __launch_bounds__(256) void label_frames_kernel(const int64_t* __restrict__ ohe, int K,
                                                           int B, int32_t* __restrict__ lab,
                                                           uint32_t* flag, uint32_t token,
                                                           float* __restrict__ seed,
                                                           const FramePack fp,
                                                           int32_t* __restrict__ frames_out) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int best = pcgmix::onehot_argmax(ohe, K, b);
    lab[b] = best;
    if (seed)
      for (int c = 0; c < K; ++c) seed[(size_t)b * K + c] = c == best ? 1.f : 0.f;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(flag, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // the boundaries after the flag: the host does not wait for them
  for (int i = threadIdx.x; i < B * 5; i += blockDim.x) {
    const int w = fp.w[i >> 1];
    frames_out[i] = (i & 1) ? (w >> 16) : ((int)((unsigned)w << 16) >> 16);
  }
}

// The begin of a saliency-guided step whose class labels are already on the HOST (a training loop
// that still holds the loader's CPU targets): no label read-back, so ONE launch delivers everything
// the step's first half needs, all of it in the kernel ARGUMENTS — the labels (uint8) become the
// float one-hot gradient seed of the saliency pass (saliency.py:52-61), the boundaries (int16)
// become the int32 array post-processing, search and splice read, and a pending step payload of
// up to kPackPayBytes (pcgmix_ctx_set_payload: targets, dropout key, optimiser scalars of a
// captured training step) goes to its static device address.
struct LabelPack {
  int32_t w[pcgmix::kPackB / 4];
};
struct StepPayPack {
  uint4 w[pcgmix::kPackPayBytes / 16];
};
__global__ __launch_bounds__(256) void seed_frames_kernel(const LabelPack lp, int K, int B,
                                                          float* __restrict__ seed,
                                                          const FramePack fp,
                                                          int32_t* __restrict__ frames_out,
                                                          const StepPayPack pay,
                                                          uint4* __restrict__ pay_dst, int pay_n16) {
  if ((int)threadIdx.x < pay_n16) pay_dst[threadIdx.x] = pay.w[threadIdx.x];
  if (seed)
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      const int lab = (lp.w[b >> 2] >> (8 * (b & 3))) & 0xff;
      for (int c = 0; c < K; ++c) seed[(size_t)b * K + c] = c == lab ? 1.f : 0.f;
    }
  for (int i = threadIdx.x; i < B * 5; i += blockDim.x) {
    const int w = fp.w[i >> 1];
    frames_out[i] = (i & 1) ? (w >> 16) : ((int)((unsigned)w << 16) >> 16);
  }
}

// The same from device memory, for batches beyond kPackB: [frames int32 B*5 | labels int32 B].
__global__ __launch_bounds__(256) void seed_from_labels_kernel(const int32_t* __restrict__ blk, int K,
                                                               int B, float* __restrict__ seed,
                                                               int32_t* __restrict__ frames_out) {
  const int i0 = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  if (seed)
    for (int b = i0; b < B; b += stride) {
      const int lab = blk[B * 5 + b];
      for (int c = 0; c < K; ++c) seed[(size_t)b * K + c] = c == lab ? 1.f : 0.f;
    }
  for (int i = i0; i < B * 5; i += stride) frames_out[i] = blk[i];
}

// Host-to-device copy as a kernel: n16 16-byte words from device-readable host memory (a pinned
// staging slot) to device memory.  hipMemcpyAsync does the same with a blit kernel up to 16 KB;
// above that it takes the SDMA path, which costs ~25 us of stream stall per copy on MI355X
// (profiles/r2_cfg3_step_timeline.txt, first version) — ten times the transfer itself at 50 KB.
__global__ __launch_bounds__(256) void fetch_kernel(const uint4* __restrict__ src,
                                                    uint4* __restrict__ dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

struct pcgmix_ctx {
  double phase_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // accumulated host time per phase (diagnostic)
  long long calls = 0;
  int device = 0;
  Slot slot[kSlots];
  int next = 0;
  int32_t* lab = nullptr;          // host-mapped, coherent
  size_t lab_cap = 0;
  uint32_t* flag = nullptr;        // host-mapped, coherent
  uint32_t token = 0;
  std::map<std::pair<int, int>, double*> ops;   // (T, n_knots) -> device operator
  uint64_t gate_step = ~0ull;      // generator seeded for this step (pcgmix_ctx_gate), reusable
  PyRandom* seeded = nullptr;
  std::vector<int64_t> keys, pool, idx;   // per-step scratch of the partner draw
  std::vector<int> gid;
  std::vector<char> payload;       // pcgmix_ctx_set_payload: rides with the next index block
  void* payload_dst = nullptr;
  void* ws = nullptr;              // displacement-search workspace of the saliency-guided step
  size_t ws_cap = 0;
  int sal_B = 0, sal_max_len = 0;  // what pcgmix_ctx_salopt_begin saw
  int32_t sal_frames_h[pcgmix::kPackB * 5];   // ... and the boundaries (B <= kPackB), for the search's plan
  bool sal_frames_known = false;
  // the armed plain step (pcgmix_kernels.h, ArmedArgs)
  unsigned long long* rec_h = nullptr;   // host-mapped, coherent: kPackB records
  unsigned long long* rec_d = nullptr;   // device relay
  uint32_t armed_seq = 0;
  struct ArmedOpen {                     // an armed kernel is waiting for its records
    bool open = false, warp = false;
    uint32_t seq = 0;
    std::chrono::steady_clock::time_point t_launch;
    const float* x = nullptr;
    float* y = nullptr;
    int B = 0, C = 0, T = 0, my_slot = 0;
    hipStream_t s = nullptr;
  } armed_open;
  hipStream_t armed_stream = nullptr;    // stream of the last armed launch
  bool armed_any = false;
  long long armed_calls = 0, armed_slow = 0, armed_aborted = 0;
  // Random(step + 1) seeded (1,247 dependent steps, 1.9 us) and regenerated on a helper thread while
  // this step runs: the next call copies 2.5 KB instead.  The helper spins for half a millisecond
  // after a job, then sleeps.
  struct SeedAhead {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> job{0};          // 0 idle | 1 posted | 2 done
    std::atomic<bool> quit{false};
    uint64_t step = 0;
    PyRandom* state = nullptr;
    int fork_generation = 0;       // of the process that started the helper
    long long hits = 0, misses = 0;
  } ahead;
  unsigned long long armed_timeout = 100000000ull;   // 1 s of the 100 MHz clock, from the kernel's start
  int armed_stall_ms = 0;                            // tests: host stall in front of the record write
};

// Host-to-device copy as a kernel launch (see fetch_kernel) for callers with their own pinned
// staging: src = pinned host memory (hipHostMalloc / torch pin_memory: device-readable at the same
// address), both 16-byte aligned; nbytes is rounded up to whole 16-byte words, which both buffers
// must hold.
extern "C" int pcgmix_fetch_h2d(const void* src_pinned, void* dst_dev, size_t nbytes,
                                pcgmix_stream_t stream) {
  if (!src_pinned || !dst_dev || nbytes > (1ull << 31) ||
      ((reinterpret_cast<uintptr_t>(src_pinned) | reinterpret_cast<uintptr_t>(dst_dev)) & 15))
    return hipErrorInvalidValue;
  if (nbytes == 0) return hipSuccess;
  const int n16 = (int)((nbytes + 15) / 16);
  const int blocks = (n16 + 255) / 256 < 32 ? (n16 + 255) / 256 : 32;
  hipLaunchKernelGGL(fetch_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), static_cast<const uint4*>(src_pinned),
                     static_cast<uint4*>(dst_dev), n16);
  return (int)hipGetLastError();
}

namespace {

// A forked child has the parent's context bytes but not its helper thread: everything that would wait for
// the helper first checks that no fork happened since it was started.
std::atomic<int> g_ctx_fork_generation{0};
void ctx_mark_forked() { g_ctx_fork_generation.fetch_add(1, std::memory_order_relaxed); }

void seed_ahead_run(pcgmix_ctx* c) {
  pcgmix_ctx::SeedAhead& a = c->ahead;
  for (;;) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (a.job.load(std::memory_order_acquire) != 1 && !a.quit.load(std::memory_order_relaxed)) {
      if ((++spins & 63) || std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(500)) {
        _mm_pause();
        continue;
      }
      std::unique_lock<std::mutex> lk(a.mu);
      a.cv.wait(lk, [&] { return a.job.load(std::memory_order_acquire) == 1 || a.quit.load(); });
    }
    if (a.quit.load()) return;
    new (a.state) PyRandom(a.step);
    a.state->twist();
    a.job.store(2, std::memory_order_release);
  }
}

// c->seeded = Random(step), regenerated: from the helper if it has this step, else here
void seed_for_step(pcgmix_ctx* c, uint64_t step) {
  if (c->gate_step == step) return;
  pcgmix_ctx::SeedAhead& a = c->ahead;
  bool have = false;
  if (a.state && a.fork_generation == g_ctx_fork_generation.load(std::memory_order_relaxed) &&
      a.job.load(std::memory_order_acquire) != 0) {
    while (a.job.load(std::memory_order_acquire) == 1) _mm_pause();      // <= one seeding
    have = a.step == step;
    if (have) std::memcpy(static_cast<void*>(c->seeded), a.state, sizeof(PyRandom));
    a.job.store(0, std::memory_order_relaxed);
  }
  if (have) {
    ++a.hits;
  } else {
    new (c->seeded) PyRandom(step);
    c->seeded->twist();
    ++a.misses;
  }
  c->gate_step = step;
}

// ask the helper for Random(step) (the step a training loop calls next); starts it on first use
void seed_ahead_post(pcgmix_ctx* c, uint64_t step) {
  pcgmix_ctx::SeedAhead& a = c->ahead;
  static const bool enabled = getenv("PCGMIX_NO_SEED_AHEAD") == nullptr;
  if (!enabled) return;
  if (!a.state) {
    static const int once = pthread_atfork(nullptr, nullptr, ctx_mark_forked);
    (void)once;
    a.state = static_cast<PyRandom*>(::operator new(sizeof(PyRandom)));
    a.fork_generation = g_ctx_fork_generation.load(std::memory_order_relaxed);
    a.th = std::thread([c] { seed_ahead_run(c); });
  }
  if (a.fork_generation != g_ctx_fork_generation.load(std::memory_order_relaxed)) return;   // forked child
  if (a.job.load(std::memory_order_acquire) == 1) return;      // still busy with an older request
  a.step = step;
  a.job.store(1, std::memory_order_release);
  { std::lock_guard<std::mutex> lk(a.mu); }          // the helper is before its check or asleep
  a.cv.notify_one();
}

}  // namespace

extern "C" int pcgmix_ctx_create(int device, pcgmix_ctx** out) {
  if (!out) return hipErrorInvalidValue;
  int prev = 0;
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) return (int)e;
  if ((e = hipSetDevice(device)) != hipSuccess) return (int)e;
  pcgmix_ctx* c = new pcgmix_ctx();
  c->device = device;
  c->seeded = static_cast<PyRandom*>(::operator new(sizeof(PyRandom)));
  e = hipHostMalloc(reinterpret_cast<void**>(&c->flag), 64, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) {
    std::memset(c->flag, 0, 64);       // word 0: label token, word 8: abort word of the armed step
    for (int i = 0; i < kSlots && e == hipSuccess; ++i)
      e = hipEventCreateWithFlags(&c->slot[i].ev, hipEventDisableTiming);
  }
  (void)hipSetDevice(prev);
  if (e != hipSuccess) { delete c; return (int)e; }
  *out = c;
  return hipSuccess;
}

extern "C" void pcgmix_ctx_destroy(pcgmix_ctx* c) {
  if (!c) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& s : c->slot) {
    if (s.pinned) (void)hipHostFree(s.pinned);
    if (s.dev) (void)hipFree(s.dev);
    if (s.ev) (void)hipEventDestroy(s.ev);
  }
  if (c->ahead.state) {
    if (c->ahead.fork_generation == g_ctx_fork_generation.load(std::memory_order_relaxed)) {
      {
        std::lock_guard<std::mutex> lk(c->ahead.mu);
        c->ahead.quit.store(true);
      }
      c->ahead.cv.notify_all();
      if (c->ahead.th.joinable()) c->ahead.th.join();
    } else if (c->ahead.th.joinable()) {
      c->ahead.th.detach();
    }
    ::operator delete(c->ahead.state);
  }
  if (c->lab) (void)hipHostFree(c->lab);
  if (c->flag) (void)hipHostFree(c->flag);
  if (c->ws) (void)hipFree(c->ws);
  if (c->rec_h) (void)hipHostFree(c->rec_h);
  if (c->rec_d) (void)hipFree(c->rec_d);
  for (auto& kv : c->ops) (void)hipFree(kv.second);
  ::operator delete(c->seeded);
  (void)hipSetDevice(prev);
  delete c;
}

// random.Random(step).uniform(0, 1) with the seeded generator kept in the context, so that the
// step call that follows a passed gate does not initialise MT19937 a second time.
extern "C" double pcgmix_ctx_gate(pcgmix_ctx* c, uint64_t step) {
  if (!c) return 2.0;
  seed_for_step(c, step);
  PyRandom r = *c->seeded;
  return 0.0 + (1.0 - 0.0) * r.random();
}

namespace {

hipError_t slot_reserve(pcgmix_ctx* c, int i, size_t nbytes) {
  Slot& s = c->slot[i];
  if (i % kSlotGroup == 0) {                  // entering a group: its previous round must be done
    Slot& g = c->slot[i + kSlotGroup - 1];
    if (g.busy) {
      hipError_t e = hipEventSynchronize(g.ev);
      if (e != hipSuccess) return e;
      g.busy = false;
    }
  }
  if (s.cap >= nbytes) return hipSuccess;
  // A step needs more staging than this slot has: grow EVERY slot now.  One slot at a time put a
  // hipHostMalloc + hipMalloc (~250 us) on each of the first eight calls of a larger shape — five
  // of them inside a 5-step warm-up and three at the head of the timed region behind it (round 4:
  // calls of 246 / 249 / 243 us, then 28 us, at the head of the durmixmagwarp (256,4,5000) leg).
  // The other slots' device twins may still be read by launches in flight (a group's event is
  // recorded at its last slot only), so the device is drained first; this happens once per shape.
  size_t cap = 8192;
  while (cap < nbytes) cap <<= 1;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  for (int j = 0; j < kSlots; ++j) {
    Slot& t = c->slot[j];
    t.busy = false;
    if (t.cap >= cap) continue;
    if (t.pinned) (void)hipHostFree(t.pinned);
    if (t.dev) (void)hipFree(t.dev);
    t.pinned = t.dev = nullptr;
    t.cap = 0;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&t.pinned), cap, hipHostMallocDefault)) != hipSuccess)
      return e;
    if ((e = hipMalloc(reinterpret_cast<void**>(&t.dev), cap)) != hipSuccess) return e;
    t.cap = cap;
  }
  return hipSuccess;
}

hipError_t spline_op_device(pcgmix_ctx* c, int T, int n_knots, const double** out) {
  auto it = c->ops.find({T, n_knots});
  if (it != c->ops.end()) { *out = it->second; return hipSuccess; }
  const int n = pcgmix_spline_operator_size(n_knots);
  std::vector<double> host((size_t)n);
  int err = pcgmix_spline_operator_f64(T, n_knots, host.data());
  if (err) return (hipError_t)err;
  double* d = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * n);
  if (e != hipSuccess) return e;
  if ((e = hipMemcpy(d, host.data(), sizeof(double) * n, hipMemcpyHostToDevice)) != hipSuccess) {
    (void)hipFree(d);
    return e;
  }
  c->ops[{T, n_knots}] = d;
  *out = d;
  return hipSuccess;
}

}  // namespace

namespace {

// One staging slot's first `nbytes` to its device twin on `s`.
hipError_t upload_slot(const Slot& sl, size_t nbytes, hipStream_t s) {
  static const size_t limit = [] {
    const char* env = getenv("PCGMIX_BLIT_LIMIT");     // tuning runs
    return env ? (size_t)atoll(env) : (size_t)16384;
  }();
  if (nbytes <= limit)
    return hipMemcpyAsync(sl.dev, sl.pinned, nbytes, hipMemcpyHostToDevice, s);
  const int n16 = (int)((nbytes + 15) / 16);          // slots are sized in powers of two >= 8 KB
  const int blocks = (n16 + 255) / 256 < 32 ? (n16 + 255) / 256 : 32;
  hipLaunchKernelGGL(fetch_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(sl.pinned), reinterpret_cast<uint4*>(sl.dev), n16);
  return hipGetLastError();
}

// Boundaries (int64 (B,5), host) validated against the signal length and packed as int32.
// Returns 0, -1 (negative / decreasing) or -2 (cycle end beyond T); *max_len = longest state.
int pack_frames(const int64_t* frames, int B, int T, int32_t* st, int* max_len) {
  int bad = 0;
  int64_t longest = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t* r = frames + (size_t)b * 5;
    if (r[0] < 0) bad = bad ? bad : -1;
    for (int k = 0; k < 4; ++k) {
      if (r[k + 1] < r[k]) bad = bad ? bad : -1;
      if (r[k + 1] - r[k] > longest) longest = r[k + 1] - r[k];
    }
    if (r[4] > T) bad = bad ? bad : -2;
    for (int k = 0; k < 5; ++k) st[b * 5 + k] = (int32_t)r[k];
  }
  if (max_len) *max_len = (int)(longest > T ? T : longest);
  return bad;
}

// The same as int16 for the launches that carry the boundaries in their arguments (B <= kPackB,
// T <= 32767).  Branch-free passes (the compiler vectorises them; the row-by-row version with its
// first-failure bookkeeping was 1.2 us of a 17 us step); the exact code of the FIRST offending row,
// as pack_frames reports it, is worked out only when something is wrong.
__attribute__((always_inline)) inline bool frames16_scan(const int64_t* frames, int n, int T,
                                                        int16_t* fr16) {
  int64_t lo = 0, hi = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t v = frames[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
    fr16[i] = (int16_t)v;
  }
  int dec = 0;
  for (int i = 0; i + 1 < n; ++i) dec |= (int)(frames[i + 1] < frames[i]) & (int)((i + 1) % 5 != 0);
  return lo < 0 || hi > T || dec;
}
__attribute__((target("avx2"))) bool frames16_scan_avx2(const int64_t* f, int n, int T, int16_t* o) {
  return frames16_scan(f, n, T, o);
}
bool frames16_scan_base(const int64_t* f, int n, int T, int16_t* o) { return frames16_scan(f, n, T, o); }

int pack_frames16(const int64_t* frames, int B, int T, int16_t* fr16) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (!(avx2 ? frames16_scan_avx2(frames, B * 5, T, fr16) : frames16_scan_base(frames, B * 5, T, fr16)))
    return 0;
  int bad = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t* r = frames + (size_t)b * 5;
    if (r[0] < 0) bad = bad ? bad : -1;
    for (int k = 0; k < 4; ++k)
      if (r[k + 1] < r[k]) bad = bad ? bad : -1;
    if (r[4] > T) bad = bad ? bad : -2;
  }
  return bad;
}

// Groups of equal label in order of first appearance (augmentations.py:500-510), each permuted by
// a fresh Random(step).sample: one initialisation (c->seeded), state copied per group.  Scratch
// lives in the context (no allocation per step): gid[b] = group of sample b, members of group g =
// the samples b with gid[b] == g in ascending order.
void draw_partners(pcgmix_ctx* c, const int64_t* labels, int B, int64_t* mix_out, int32_t* mixp,
                   int16_t* mixp16 = nullptr) {
  c->keys.clear();
  c->gid.resize((size_t)B);
  c->pool.resize((size_t)B);
  c->idx.resize((size_t)B);
  for (int b = 0; b < B; ++b) {
    size_t g = 0;
    while (g < c->keys.size() && c->keys[g] != labels[b]) ++g;
    if (g == c->keys.size()) c->keys.push_back(labels[b]);
    c->gid[(size_t)b] = (int)g;
  }
  for (size_t g = 0; g < c->keys.size(); ++g) {
    size_t n = 0;
    for (int b = 0; b < B; ++b)
      if (c->gid[(size_t)b] == (int)g) c->idx[n++] = b;
    PyRandom rng = *c->seeded;
    std::memcpy(c->pool.data(), c->idx.data(), n * sizeof(int64_t));
    for (size_t i = 0; i < n; ++i) {                 // sample(population, k = n): pool branch
      const uint64_t j = rng.randbelow((uint64_t)(n - i));
      mix_out[c->idx[i]] = c->pool[j];
      if (mixp16) mixp16[c->idx[i]] = (int16_t)c->pool[j];
      else mixp[c->idx[i]] = (int32_t)c->pool[j];
      c->pool[j] = c->pool[n - i - 1];
    }
  }
}

// An event per GROUP of kSlotGroup slots: recorded behind the group's last enqueued work, waited
// for before the group's first reuse (one stream-order point covers all four: fewer API calls).
hipError_t slot_commit(pcgmix_ctx* c, int my_slot, hipStream_t s) {
  if (my_slot % kSlotGroup == kSlotGroup - 1) {
    const hipError_t e = hipEventRecord(c->slot[my_slot].ev, s);
    if (e != hipSuccess) return e;
    c->slot[my_slot].busy = true;
  }
  c->next = (my_slot + 1) % kSlots;
  return hipSuccess;
}

struct DeviceGuard {      // make the context's device current for the call
  int cur = 0, dev = 0;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) : dev(device) {
    err = hipGetDevice(&cur);
    if (err == hipSuccess && cur != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (cur != dev) (void)hipSetDevice(cur);
  }
};

// Enqueue the label arg-max for a (B, K) one-hot matrix on `s`; the labels land in c->lab and the
// flag word takes the returned token.
// Capture check, label memory sized for B, next token: what precedes the launch of a label kernel.
hipError_t labels_prepare(pcgmix_ctx* c, int B, hipStream_t s) {
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;   // a host wait cannot be captured
  if (c->lab_cap < (size_t)B) {
    if (c->lab) (void)hipHostFree(c->lab);
    c->lab = nullptr;
    c->lab_cap = 0;
    size_t cap = 1024;
    while (cap < (size_t)B) cap <<= 1;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->lab), cap * sizeof(int32_t),
                                 hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return e;
    std::memset(c->lab, 0, cap * sizeof(int32_t));
    c->lab_cap = cap;
  }
  // tokens start at 0x100: the armed step reads this memory as 8-byte words with the token on top, and
  // a pair of int32 labels left by another label kernel (values < 256) must never look like one
  ++c->token;
  if (c->token < 0x100u) c->token = 0x100u;
  return hipSuccess;
}

hipError_t labels_begin(pcgmix_ctx* c, const int64_t* ohe_dev, int K, int B, hipStream_t s,
                        float* seed = nullptr) {
  const hipError_t e = labels_prepare(c, B, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(label_argmax_kernel, dim3(1), dim3(256), 0, s, ohe_dev, K, B, c->lab, c->flag,
                     c->token, seed);
  return hipGetLastError();
}

// Spin until the flag word shows the current token (2 ms, then a stream synchronisation).
hipError_t labels_wait(pcgmix_ctx* c, hipStream_t s) {
  const uint32_t want = c->token;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n(c->flag, __ATOMIC_ACQUIRE) != want) {
    _mm_pause();
    if ((++spins & 1023u) == 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
      hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess) return e;
      if (__atomic_load_n(c->flag, __ATOMIC_ACQUIRE) != want) return hipErrorUnknown;
    }
  }
  return hipSuccess;
}

// The same wait while an ARMED kernel sits behind the label write on `s`: a stream synchronisation
// would wait for that kernel, which waits for us.  Spin, then poll politely; give up after a minute.
hipError_t labels_wait_armed(pcgmix_ctx* c, int B, int64_t* labels) {
  const uint32_t want = c->token;
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(c->lab);
  const int n = (B + 3) / 4;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  bool slow = false;
  int have = 0;                               // words 0 .. have-1 carry this step's token
  for (;;) {
    while (have < n) {
      const unsigned long long v = __atomic_load_n(w + have, __ATOMIC_RELAXED);
      if ((uint32_t)(v >> 32) != want) break;
      for (int j = 0; j < 4 && 4 * have + j < B; ++j) labels[4 * have + j] = (int64_t)((v >> (8 * j)) & 0xff);
      ++have;
    }
    if (have == n) return hipSuccess;
    if (!slow) {
      _mm_pause();
      if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
        slow = true;
    } else {
      timespec ts{0, 20000};
      nanosleep(&ts, nullptr);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return hipErrorLaunchTimeOut;
    }
  }
}

hipError_t armed_prepare(pcgmix_ctx* c, hipStream_t s) {
  if (!c->rec_h) {
    const size_t bytes = sizeof(unsigned long long) * pcgmix::kPackB * pcgmix::kArmedRecWords;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->rec_h), bytes,
                                 hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { c->rec_h = nullptr; return e; }
    std::memset(c->rec_h, 0, bytes);
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->rec_d), bytes)) != hipSuccess ||
        (e = hipMemset(c->rec_d, 0, bytes)) != hipSuccess) {      // synchronous: once per context
      (void)hipHostFree(c->rec_h);
      if (c->rec_d) (void)hipFree(c->rec_d);
      c->rec_h = nullptr;
      c->rec_d = nullptr;
      return e;
    }
  }
  // the records are one set per context: an armed kernel still queued on ANOTHER stream must have read
  // its records before they are rewritten (on the same stream the label wait already implies it)
  // (the whole device, not the other stream: its handle may have been destroyed by its owner since)
  if (c->armed_any && c->armed_stream != s) {
    const hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
  }
  c->armed_seq = c->armed_seq >= 0x7ffffffeu ? 1u : c->armed_seq + 1u;
  return hipSuccess;
}

// stamp = seq: the step's records; stamp = seq | kArmedAbort: every relay gives up
void armed_write(pcgmix_ctx* c, uint32_t stamp, const int16_t* fr16, const int16_t* mix16, int B, float lam = 0.f) {
  const unsigned long long hi = (unsigned long long)stamp << 32;
  for (int b = 0; b < B; ++b) {
    unsigned long long* r = c->rec_h + (size_t)b * pcgmix::kArmedRecWords;
    uint16_t v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fr16) {
      const int m = (mix16[b] < 0 || mix16[b] >= B) ? b : mix16[b];
      for (int k = 0; k < 5; ++k) {
        v[k] = (uint16_t)fr16[b * 5 + k];
        v[6 + k] = (uint16_t)fr16[m * 5 + k];
      }
      v[5] = (uint16_t)mix16[b];
    }
    for (int i = 0; i < 6; ++i)
      __atomic_store_n(r + i, hi | ((unsigned long long)v[2 * i + 1] << 16) | v[2 * i], __ATOMIC_RELAXED);
    uint32_t lam_bits;
    std::memcpy(&lam_bits, &lam, 4);
    __atomic_store_n(r + 6, hi | lam_bits, __ATOMIC_RELAXED);
  }
}

}  // namespace

// The label read-back on its own, in two halves, for callers that have GPU work to enqueue in
// between (the saliency-guided step enqueues the frozen model's graph): begin = the arg-max kernel
// on `stream`; wait = the spin, then int64 class labels in `labels_out` (host, B).
extern "C" int pcgmix_ctx_labels_begin(pcgmix_ctx* c, const int64_t* target_ohe_dev, int num_classes,
                                       int B, float* seed_out, pcgmix_stream_t stream) {
  if (!c || !target_ohe_dev || num_classes <= 0 || B <= 0) return hipErrorInvalidValue;
  int cur = 0;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return (int)e;
  if (cur != c->device && (e = hipSetDevice(c->device)) != hipSuccess) return (int)e;
  e = labels_begin(c, target_ohe_dev, num_classes, B, reinterpret_cast<hipStream_t>(stream), seed_out);
  if (cur != c->device) (void)hipSetDevice(cur);
  return (int)e;
}

extern "C" int pcgmix_ctx_labels_wait(pcgmix_ctx* c, int64_t* labels_out, int B,
                                      pcgmix_stream_t stream) {
  if (!c || !labels_out || B <= 0 || (size_t)B > c->lab_cap) return hipErrorInvalidValue;
  const hipError_t e = labels_wait(c, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return (int)e;
  for (int b = 0; b < B; ++b) labels_out[b] = c->lab[b];
  return hipSuccess;
}

// Bytes the caller wants on the device together with the next step (e.g. the float targets the
// loss reads, optimiser hyper-parameters, a dropout key): they are appended to the step's index
// block — same pinned slot, same single H2D copy — and block (0,0,0) of the splice kernel writes
// them to dst_dev, so they are in place for whatever the caller enqueues behind the step.  One
// shot: consumed by the next successful pcgmix_augment_plain_f32 on this context.  bytes == 0
// withdraws a pending payload.
extern "C" int pcgmix_ctx_set_payload(pcgmix_ctx* c, const void* host, size_t bytes, void* dst_dev) {
  if (!c || (bytes && (!host || !dst_dev)) || bytes > (1u << 20) ||
      (reinterpret_cast<uintptr_t>(dst_dev) & 15))
    return hipErrorInvalidValue;
  c->payload.assign((bytes + 15) & ~(size_t)15, 0);
  if (bytes) std::memcpy(c->payload.data(), host, bytes);
  c->payload_dst = bytes ? dst_dev : nullptr;
  return hipSuccess;
}

// A pending payload on its own: the step it was meant to ride with did not run a plain splice
// (probability gate, another method).  Same staging ring, one H2D copy straight to dst_dev.
extern "C" int pcgmix_ctx_flush_payload(pcgmix_ctx* c, pcgmix_stream_t stream) {
  if (!c) return hipErrorInvalidValue;
  if (c->payload.empty()) return hipSuccess;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;
  int cur = 0;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return (int)e;
  if (cur != c->device && (e = hipSetDevice(c->device)) != hipSuccess) return (int)e;
  const int my_slot = c->next;
  Slot& sl = c->slot[my_slot];
  e = slot_reserve(c, my_slot, c->payload.size());
  if (e == hipSuccess) {
    std::memcpy(sl.pinned, c->payload.data(), c->payload.size());
    e = hipMemcpyAsync(c->payload_dst, sl.pinned, c->payload.size(), hipMemcpyHostToDevice, s);
  }
  if (e == hipSuccess && my_slot % kSlotGroup == kSlotGroup - 1) {
    if ((e = hipEventRecord(sl.ev, s)) == hipSuccess) sl.busy = true;
  }
  if (e == hipSuccess) {
    c->next = (my_slot + 1) % kSlots;
    c->payload.clear();
    c->payload_dst = nullptr;
  }
  if (cur != c->device) (void)hipSetDevice(cur);
  return (int)e;
}

namespace {

// One fired step of a plain method, as pcgmix_augment_plain_f32 received it, and the three ways it is
// carried out.  phase_ns accumulates the host time between laps (pcgmix_ctx_phase_times).
struct PlainStep {
  pcgmix_ctx* c;
  const float* x;
  float* y;
  const int64_t* target_ohe_dev;
  int num_classes;
  const int64_t* labels_host;
  const int64_t* frames;
  uint64_t step;
  float lam;
  const double* knots;
  int n_knots;
  int64_t* mix_out;
  int B, C, T;
  hipStream_t s;
  pcgmix_stream_t stream;
  std::chrono::steady_clock::time_point tp;
  void lap(int i) {
    const auto now = std::chrono::steady_clock::now();
    c->phase_ns[i] += std::chrono::duration<double, std::nano>(now - tp).count();
    tp = now;
  }
};

// Strict signature, small plain batch (BASELINE configs[1]): ONE launch at the start of the call —
//    the splice ARMED before its index block exists (pcgmix_kernels.h, ArmedArgs).  Its first block
//    does the label arg-max; the host picks the labels up, draws the partners and writes one stamped
//    record per sample into host-mapped memory, where the waiting blocks find them.  Against the
//    two-launch chain below this takes the splice's launch (3 us of host time, 3-4 us until the
//    command processor has started it) and the label kernel's own start out of the chain
//    label -> host -> splice that bounds the step.
//    The same for splice + warp (durmixmagwarp): the knots go into a pinned slot and cross the link
//    inside the kernel, once per sample, so the label launch AND the fetch launch of the staged path
//    leave the chain.
// First half: everything up to and including the launch.  lam is needed here only for the splice + warp
// kernel (its knots come with lam out of one draw); the plain kernel takes it from the record.
int armed_begin(PlainStep& p, bool arm_warp) {
  pcgmix_ctx* c = p.c;
  const int B = p.B, C = p.C, T = p.T;
  hipStream_t s = p.s;
  hipError_t e = hipSuccess;
  pcgmix_ctx::ArmedOpen& o = c->armed_open;
  if (o.open) {                              // a begin whose finish never came (the caller raised in between)
    armed_write(c, o.seq | pcgmix::kArmedAbort, nullptr, nullptr, o.B);
    if (o.warp) (void)slot_commit(c, o.my_slot, o.s);
    o.open = false;
  }
  const int my_slot = c->next;               // (splice + warp only: the slot that carries the knots)
  Slot& sl = c->slot[my_slot];
  const double* op_dev = nullptr;
  if ((e = labels_prepare(c, B, s)) != hipSuccess) return (int)e;
  if ((e = armed_prepare(c, s)) != hipSuccess) return (int)e;
  if (arm_warp) {
    const size_t nkb = (size_t)B * p.n_knots * C * sizeof(double);
    if ((e = slot_reserve(c, my_slot, nkb)) != hipSuccess) return (int)e;
    if ((e = spline_op_device(c, T, p.n_knots, &op_dev)) != hipSuccess) return (int)e;
    std::memcpy(sl.pinned, p.knots, nkb);
  }
  pcgmix::ArmedArgs a;
  a.ohe = p.target_ohe_dev;
  a.K = p.num_classes;
  a.lab64 = reinterpret_cast<unsigned long long*>(c->lab);
  a.token = c->token;
  a.rec_h = c->rec_h;
  a.rec_d = c->rec_d;
  a.abort_h = c->flag + 8;
  a.seq = c->armed_seq;
  a.timeout_ticks = c->armed_timeout;
  o.t_launch = std::chrono::steady_clock::now();
  const int err =
      arm_warp ? pcgmix::launch_mix_tq_armed(p.x, p.y, a, p.lam, reinterpret_cast<const double*>(sl.pinned),
                                             reinterpret_cast<double*>(sl.dev), op_dev, p.n_knots, B, C, T, s,
                                             c->payload.data(), (int)c->payload.size(), c->payload_dst)
               : pcgmix::launch_mix_armed(p.x, p.y, a, B, C, T, s, c->payload.data(), (int)c->payload.size(),
                                          c->payload_dst);
  if (err) return err;
  c->armed_stream = s;
  c->armed_any = true;
  c->payload.clear();                        // (it travelled in the launch's arguments)
  c->payload_dst = nullptr;
  o.open = true;
  o.warp = arm_warp;
  o.seq = a.seq;
  o.x = p.x;
  o.y = p.y;
  o.B = B; o.C = C; o.T = T;
  o.my_slot = my_slot;
  o.s = s;
  p.lap(0);
  return hipSuccess;
}

// Second half: the kernel is waiting — every way out of here writes its records.
int armed_finish(PlainStep& p) {
  pcgmix_ctx* c = p.c;
  pcgmix_ctx::ArmedOpen& o = c->armed_open;
  const int B = o.B, C = o.C, T = o.T;
  hipStream_t s = o.s;
  hipError_t e = hipSuccess;
  o.open = false;
  int16_t fr16[pcgmix::kPackB * 5], mix16[pcgmix::kPackB];
  const int bad16 = pack_frames16(p.frames, B, T, fr16);
  p.lap(1);
  seed_for_step(c, p.step);
  p.lap(2);
  int64_t lab64a[pcgmix::kPackB];
  if (bad16 || (e = labels_wait_armed(c, B, lab64a)) != hipSuccess) {
    armed_write(c, o.seq | pcgmix::kArmedAbort, nullptr, nullptr, B);
    if (o.warp) (void)slot_commit(c, o.my_slot, s);        // the kernel may still read the slot's knots
    return bad16 ? bad16 : (int)e;
  }
  p.lap(3);
  draw_partners(c, lab64a, B, p.mix_out, nullptr, mix16);
  p.lap(4);
  if (c->armed_stall_ms > 0) {
    timespec ts{c->armed_stall_ms / 1000, (long)(c->armed_stall_ms % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
  }
  armed_write(c, o.seq, fr16, mix16, B, p.lam);
  p.lap(5);
  ++c->armed_calls;
  if (o.warp && (e = slot_commit(c, o.my_slot, s)) != hipSuccess) return (int)e;
  p.lap(6);
  // The relays give up 1 s after the kernel STARTED, which is later than t_launch: records written
  // within 0.4 s of the launch were in time whatever happened in between.  Otherwise (a debugger, a
  // stopped process, a host that lost its CPU for that long): wait for the kernel and look; if the
  // blocks gave up, the step again through the unarmed path, with the labels this call holds.
  if (std::chrono::steady_clock::now() - o.t_launch > std::chrono::milliseconds(400)) {
    ++c->armed_slow;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return (int)e;
    if (__atomic_load_n(c->flag + 8, __ATOMIC_ACQUIRE) == o.seq) {
      ++c->armed_aborted;
      return pcgmix_augment_plain_f32(c, o.x, o.y, nullptr, 0, lab64a, p.frames, p.step, p.lam, p.knots,
                                      p.n_knots, p.mix_out, B, C, T, reinterpret_cast<pcgmix_stream_t>(s));
    }
  }
  p.lap(7);
  seed_ahead_post(c, p.step + 1);
  ++c->calls;
  return hipSuccess;
}

int step_armed(PlainStep& p, bool arm_warp) {
  const int err = armed_begin(p, arm_warp);
  return err ? err : armed_finish(p);
}

// Small plain batches (BASELINE configs[1]: B = 256): the index block travels in the splice
//     kernel's ARGUMENTS — no staging slot, no host-to-device copy in the chain label kernel ->
//     host -> splice that bounds a strict-signature step.
int step_karg(PlainStep& p) {
  pcgmix_ctx* c = p.c;
  const float* x = p.x;
  float* y = p.y;
  const int64_t* frames = p.frames;
  const uint64_t step = p.step;
  const float lam = p.lam;
  const int B = p.B, C = p.C, T = p.T;
  int64_t* mix_out = p.mix_out;
  hipStream_t s = p.s;
  hipError_t e = hipSuccess;
  const int64_t* labels_host = p.labels_host;
  const bool readback = labels_host == nullptr;
  int16_t fr16[pcgmix::kPackB * 5], mix16[pcgmix::kPackB];
  const int bad16 = pack_frames16(frames, B, T, fr16);
  p.lap(1);
  seed_for_step(c, step);
  p.lap(2);
  std::vector<int64_t> lab64k;
  const int64_t* labels_k = labels_host;
  if (readback) {
    if ((e = labels_wait(c, s)) != hipSuccess) return (int)e;
    lab64k.resize((size_t)B);
    for (int b = 0; b < B; ++b) lab64k[(size_t)b] = c->lab[b];
    labels_k = lab64k.data();
  }
  p.lap(3);
  if (bad16) return bad16;
  draw_partners(c, labels_k, B, mix_out, nullptr, mix16);
  p.lap(4);
  p.lap(5);
  const int err = pcgmix::launch_mix_karg(x, y, fr16, mix16, lam, B, C, T, s, c->payload.data(),
                                          (int)c->payload.size(), c->payload_dst);
  if (err) return err;
  c->payload.clear();
  c->payload_dst = nullptr;
  p.lap(6);
  p.lap(7);
  seed_ahead_post(c, step + 1);
  ++c->calls;
  return hipSuccess;
}

// Everything else: a staging slot carries boundaries, partners, knots and payload in one copy.
int step_staged(PlainStep& p) {
  pcgmix_ctx* c = p.c;
  const float* x = p.x;
  float* y = p.y;
  const int64_t* frames = p.frames;
  const uint64_t step = p.step;
  const float lam = p.lam;
  const double* knots = p.knots;
  const int n_knots = p.n_knots, B = p.B, C = p.C, T = p.T;
  int64_t* mix_out = p.mix_out;
  hipStream_t s = p.s;
  hipError_t e = hipSuccess;
  const int64_t* labels_host = p.labels_host;
  const bool readback = labels_host == nullptr;
  // 2. staging slot, boundaries validated and packed, knots copied, generator seeded
  const size_t n_int = (size_t)B * 6, n_int_pad = (n_int + 1) & ~(size_t)1;
  const size_t nk = knots ? (size_t)B * n_knots * C : 0;
  const size_t pay_off = (n_int_pad * 4 + nk * sizeof(double) + 15) & ~(size_t)15;
  const size_t nbytes = pay_off + c->payload.size();
  const int my_slot = c->next;           // advanced only when the step has been enqueued
  Slot& sl = c->slot[my_slot];
  if ((e = slot_reserve(c, my_slot, nbytes)) != hipSuccess) return (int)e;
  p.lap(1);
  int32_t* st = reinterpret_cast<int32_t*>(sl.pinned);
  const int bad = pack_frames(frames, B, T, st, nullptr);
  const double* op_dev = nullptr;
  const double* knots_dev = nullptr;
  if (knots) {
    if ((e = spline_op_device(c, T, n_knots, &op_dev)) != hipSuccess) return (int)e;
    std::memcpy(sl.pinned + n_int_pad * 4, knots, nk * sizeof(double));
    knots_dev = reinterpret_cast<const double*>(sl.dev + n_int_pad * 4);
  }
  if (!c->payload.empty()) std::memcpy(sl.pinned + pay_off, c->payload.data(), c->payload.size());
  seed_for_step(c, step);

  p.lap(2);
  // 3. wait for the labels (the one host wait the reference's signature forces,
  //    augmentations.py:501): spin on the flag word the kernel releases; if it does not show up
  //    within 2 ms fall back to a stream synchronisation
  std::vector<int64_t> lab64;
  const int64_t* labels = labels_host;
  if (readback) {
    if ((e = labels_wait(c, s)) != hipSuccess) return (int)e;
    lab64.resize((size_t)B);
    for (int b = 0; b < B; ++b) lab64[(size_t)b] = c->lab[b];
    labels = lab64.data();
  }
  p.lap(3);
  if (bad) return bad;                       // malformed boundaries: nothing else is enqueued

  // 4. partners: groups of equal label, each permuted by a fresh Random(step).sample
  draw_partners(c, labels, B, mix_out, st + (size_t)B * 5);
  p.lap(4);
  // 5. one H2D copy, the launch, the slot's event behind it.  (Round 4 tried the fetch on a side
  //    stream of the context, beside the previous step's kernels: PCGmix+ train step 129.0 -> 128.0 us,
  //    but the extra HIP stream can share a hardware queue with the stream a PIPELINED step augments on
  //    and serialise it — cfg3 train 229 -> 282 us inside the full bench; removed,
  //    profiles/r4_magwarp_fetch_ahead.txt.)
  if ((e = upload_slot(sl, nbytes, s)) != hipSuccess) return (int)e;
  p.lap(5);
  const int32_t* d = reinterpret_cast<const int32_t*>(sl.dev);
  const int err = pcgmix::launch_mix_warp(x, y, d, d + (size_t)B * 5, nullptr, lam, knots_dev, op_dev,
                                          knots ? n_knots : 0, nullptr, B, C, T, s,
                                          sl.dev + pay_off, c->payload_dst,
                                          (int)(c->payload.size() / 16));
  if (err) return err;
  c->payload.clear();
  c->payload_dst = nullptr;
  p.lap(6);
  if ((e = slot_commit(c, my_slot, s)) != hipSuccess) return (int)e;
  p.lap(7);
  seed_ahead_post(c, step + 1);
  ++c->calls;
  return hipSuccess;
}

}  // namespace

extern "C" int pcgmix_augment_plain_f32(pcgmix_ctx* c, const float* x, float* y,
                                        const int64_t* target_ohe_dev, int num_classes,
                                        const int64_t* labels_host, const int64_t* frames,
                                        uint64_t step, float lam, const double* knots,
                                        int n_knots, int64_t* mix_out, int B, int C, int T,
                                        pcgmix_stream_t stream) {
  if (!c || !x || !y || !frames || !mix_out || B <= 0 || C <= 0 || T <= 0 ||
      (!target_ohe_dev && !labels_host) || (target_ohe_dev && num_classes <= 0) ||
      (knots && n_knots < 2))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;
  PlainStep p{c, x, y, target_ohe_dev, num_classes, labels_host, frames, step, lam, knots, n_knots, mix_out,
              B, C, T, s, stream, std::chrono::steady_clock::now()};
  int cur = 0;
  hipError_t e = hipGetDevice(&cur);
  if (e != hipSuccess) return (int)e;
  if (cur != c->device && (e = hipSetDevice(c->device)) != hipSuccess) return (int)e;
  struct Restore {
    int cur, dev;
    ~Restore() { if (cur != dev) (void)hipSetDevice(cur); }
  } restore{cur, c->device};

  const bool readback = labels_host == nullptr;
  static const bool karg_ok = getenv("PCGMIX_NO_KARG") == nullptr;     // tuning / A-B runs
  const bool small = karg_ok && !knots && c->payload.size() <= (size_t)pcgmix::kPackPayBytes &&
                     B <= pcgmix::kPackB && T <= 32767 && !(T & 3) &&
                     !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15);

  static const bool armed_ok = getenv("PCGMIX_NO_ARMED") == nullptr;
  const bool arm_base = armed_ok && readback && num_classes <= 256 &&
                        c->payload.size() <= (size_t)pcgmix::kPackPayBytes &&
                        !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15);
  const bool arm_plain = arm_base && small;
  const bool arm_warp = arm_base && knots && pcgmix::mix_tq_armed_ok(B, C, T, n_knots);
  if (arm_plain || arm_warp) return step_armed(p, arm_warp);

  // start the label read-back first: what follows up to the partner draw does not need the labels
  // and runs while the GPU finishes what precedes this call on `stream`
  if (readback && (e = labels_begin(c, target_ohe_dev, num_classes, B, s)) != hipSuccess)
    return (int)e;
  p.lap(0);
  return small ? step_karg(p) : step_staged(p);
}

// The armed plain step in two calls, for a caller that has host work of its own between the launch and the
// moment lambda is known (the Python binding draws lambda from numpy's stream there): begin = validation,
// eligibility, the launch; finish = boundaries, labels, partners, records.
extern "C" int pcgmix_augment_plain_begin(pcgmix_ctx* c, const float* x, float* y,
                                          const int64_t* target_ohe_dev, int num_classes, int B, int C,
                                          int T, pcgmix_stream_t stream) {
  if (!c || !x || !y || !target_ohe_dev || num_classes <= 0 || B <= 0 || C <= 0 || T <= 0)
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;
  static const bool karg_ok = getenv("PCGMIX_NO_KARG") == nullptr;
  static const bool armed_ok = getenv("PCGMIX_NO_ARMED") == nullptr;
  if (!(armed_ok && karg_ok && num_classes <= 256 && c->payload.size() <= (size_t)pcgmix::kPackPayBytes &&
        B <= pcgmix::kPackB && T <= 32767 && !(T & 3) &&
        !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)))
    return PCGMIX_NOT_ARMED;
  DeviceGuard guard(c->device);
  if (guard.err != hipSuccess) return (int)guard.err;
  PlainStep p{c, x, y, target_ohe_dev, num_classes, nullptr, nullptr, 0, 0.f, nullptr, 0, nullptr,
              B, C, T, s, stream, std::chrono::steady_clock::now()};
  return armed_begin(p, false);
}

extern "C" int pcgmix_augment_plain_finish(pcgmix_ctx* c, const int64_t* frames, uint64_t step, float lam,
                                           int64_t* mix_out) {
  if (!c || !frames || !mix_out) return hipErrorInvalidValue;
  if (!c->armed_open.open) return hipErrorNotReady;
  DeviceGuard guard(c->device);
  if (guard.err != hipSuccess) return (int)guard.err;
  PlainStep p{c, c->armed_open.x, c->armed_open.y, nullptr, 0, nullptr, frames, step, lam, nullptr, 0, mix_out,
              c->armed_open.B, c->armed_open.C, c->armed_open.T, c->armed_open.s,
              reinterpret_cast<pcgmix_stream_t>(c->armed_open.s), std::chrono::steady_clock::now()};
  return armed_finish(p);
}

extern "C" int pcgmix_ctx_armed_stats(pcgmix_ctx* c, long long* out3) {
  if (!c || !out3) return hipErrorInvalidValue;
  out3[0] = c->armed_calls;
  out3[1] = c->armed_slow;
  out3[2] = c->armed_aborted;
  return hipSuccess;
}

extern "C" int pcgmix_ctx_armed_debug(pcgmix_ctx* c, unsigned long long timeout_ticks, int stall_ms) {
  if (!c || stall_ms < 0 || stall_ms > 5000) return hipErrorInvalidValue;
  c->armed_timeout = timeout_ticks ? timeout_ticks : 100000000ull;
  c->armed_stall_ms = stall_ms;
  return hipSuccess;
}

// ---- the saliency-guided step in two calls, with the caller's saliency pass in between ----------
// begin: the label arg-max kernel (labels -> host-mapped memory, float one-hot -> seed_out: the
// saliency pass's gradient seed), the boundaries validated, packed and sent to frames_dst_dev
// (what the saliency post-processing, the search and the splice read).
extern "C" int pcgmix_ctx_salopt_begin(pcgmix_ctx* c, const int64_t* target_ohe_dev, int num_classes,
                                       float* seed_out, const int64_t* frames,
                                       int32_t* frames_dst_dev, int B, int T,
                                       pcgmix_stream_t stream) {
  if (!c || !frames || !frames_dst_dev || B <= 0 || T <= 0 || (target_ohe_dev && num_classes <= 0))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;
  DeviceGuard guard(c->device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return (int)e;
  if (target_ohe_dev && B <= pcgmix::kPackB && T <= 32767) {
    // boundaries in the label kernel's arguments: one launch, no copy
    FramePack fp;
    int16_t* p16 = reinterpret_cast<int16_t*>(fp.w);
    int bad16 = 0;
    int64_t longest = 0;
    for (int b = 0; b < B; ++b) {
      const int64_t* r = frames + (size_t)b * 5;
      if (r[0] < 0) bad16 = bad16 ? bad16 : -1;
      for (int k = 0; k < 4; ++k) {
        if (r[k + 1] < r[k]) bad16 = bad16 ? bad16 : -1;
        if (r[k + 1] - r[k] > longest) longest = r[k + 1] - r[k];
      }
      if (r[4] > T) bad16 = bad16 ? bad16 : -2;
      for (int k = 0; k < 5; ++k) {
        p16[b * 5 + k] = (int16_t)r[k];
        c->sal_frames_h[b * 5 + k] = (int32_t)r[k];
      }
    }
    c->sal_frames_known = false;
    if (bad16) return bad16;
    c->sal_frames_known = true;
    if ((e = labels_prepare(c, B, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(label_frames_kernel, dim3(1), dim3(256), 0, s, target_ohe_dev, num_classes, B,
                       c->lab, c->flag, c->token, seed_out, fp, frames_dst_dev);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    c->sal_B = B;
    c->sal_max_len = (int)(longest > T ? T : longest);
    return hipSuccess;
  }
  if (target_ohe_dev && (e = labels_begin(c, target_ohe_dev, num_classes, B, s, seed_out)) != hipSuccess)
    return (int)e;
  const int my_slot = c->next;
  if ((e = slot_reserve(c, my_slot, (size_t)B * 20)) != hipSuccess) return (int)e;
  Slot& sl = c->slot[my_slot];
  int max_len = 0;
  c->sal_frames_known = false;
  const int bad = pack_frames(frames, B, T, reinterpret_cast<int32_t*>(sl.pinned), &max_len);
  if (bad) return bad;
  if ((e = hipMemcpyAsync(frames_dst_dev, sl.pinned, (size_t)B * 20, hipMemcpyHostToDevice, s)) !=
      hipSuccess)
    return (int)e;
  c->sal_B = B;
  c->sal_max_len = max_len;
  return (int)slot_commit(c, my_slot, s);
}

// begin with the class labels on the HOST: no read-back, no flag.  Up to kPackB samples everything
// rides in the arguments of ONE launch (seed_frames_kernel), a pending step payload of up to
// kPackPayBytes included (it is then consumed); larger batches stage [frames | labels] through a
// slot and leave the payload pending (pcgmix_ctx_flush_payload).
extern "C" int pcgmix_ctx_salopt_begin_labels(pcgmix_ctx* c, const int64_t* labels_host,
                                              int num_classes, float* seed_out,
                                              const int64_t* frames, int32_t* frames_dst_dev, int B,
                                              int T, pcgmix_stream_t stream) {
  if (!c || !labels_host || !frames || !frames_dst_dev || B <= 0 || T <= 0 || num_classes <= 0)
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;
  DeviceGuard guard(c->device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return (int)e;
  for (int b = 0; b < B; ++b)
    if (labels_host[b] < 0 || labels_host[b] >= num_classes) return hipErrorInvalidValue;
  if (B <= pcgmix::kPackB && T <= 32767 && num_classes <= 256) {
    FramePack fp;
    LabelPack lp;
    StepPayPack pay;
    int16_t* p16 = reinterpret_cast<int16_t*>(fp.w);
    uint8_t* p8 = reinterpret_cast<uint8_t*>(lp.w);
    int bad16 = 0;
    int64_t longest = 0;
    for (int b = 0; b < B; ++b) {
      const int64_t* r = frames + (size_t)b * 5;
      if (r[0] < 0) bad16 = bad16 ? bad16 : -1;
      for (int k = 0; k < 4; ++k) {
        if (r[k + 1] < r[k]) bad16 = bad16 ? bad16 : -1;
        if (r[k + 1] - r[k] > longest) longest = r[k + 1] - r[k];
      }
      if (r[4] > T) bad16 = bad16 ? bad16 : -2;
      for (int k = 0; k < 5; ++k) {
        p16[b * 5 + k] = (int16_t)r[k];
        c->sal_frames_h[b * 5 + k] = (int32_t)r[k];
      }
      p8[b] = (uint8_t)labels_host[b];
    }
    c->sal_frames_known = false;
    if (bad16) return bad16;
    c->sal_frames_known = true;
    int pay_n16 = 0;
    void* pay_dst = nullptr;
    if (!c->payload.empty() && c->payload.size() <= (size_t)pcgmix::kPackPayBytes) {
      std::memcpy(pay.w, c->payload.data(), c->payload.size());
      pay_n16 = (int)(c->payload.size() / 16);
      pay_dst = c->payload_dst;
    }
    hipLaunchKernelGGL(seed_frames_kernel, dim3(1), dim3(256), 0, s, lp, num_classes, B, seed_out, fp,
                       frames_dst_dev, pay, static_cast<uint4*>(pay_dst), pay_n16);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if (pay_n16) {
      c->payload.clear();
      c->payload_dst = nullptr;
    }
    c->sal_B = B;
    c->sal_max_len = (int)(longest > T ? T : longest);
    return hipSuccess;
  }
  const int my_slot = c->next;
  if ((e = slot_reserve(c, my_slot, (size_t)B * 24)) != hipSuccess) return (int)e;
  Slot& sl = c->slot[my_slot];
  int32_t* st = reinterpret_cast<int32_t*>(sl.pinned);
  int max_len = 0;
  c->sal_frames_known = false;
  const int bad = pack_frames(frames, B, T, st, &max_len);
  if (bad) return bad;
  for (int b = 0; b < B; ++b) st[(size_t)B * 5 + b] = (int32_t)labels_host[b];
  if ((e = upload_slot(sl, (size_t)B * 24, s)) != hipSuccess) return (int)e;
  const int blocks = (B * 5 + 255) / 256 < 64 ? (B * 5 + 255) / 256 : 64;
  hipLaunchKernelGGL(seed_from_labels_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                     reinterpret_cast<const int32_t*>(sl.dev), num_classes, B, seed_out,
                     frames_dst_dev);
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  c->sal_B = B;
  c->sal_max_len = max_len;
  return (int)slot_commit(c, my_slot, s);
}

// finish: labels picked up (or handed over), same-label partners drawn, [mix | knots] sent in one
// copy, displacement search and fused splice(+warp) enqueued (pcgmix_salopt_mix_warp_f32's pair
// of kernels; the search workspace is the context's).  frames_dev = begin's frames_dst_dev.
extern "C" int pcgmix_ctx_salopt_finish(pcgmix_ctx* c, const float* x, float* y, const float* sal,
                                        const int32_t* frames_dev, const int64_t* labels_host,
                                        uint64_t step, float lam, int mode, const double* knots,
                                        int n_knots, int64_t* mix_out, int B, int C, int T,
                                        pcgmix_stream_t stream) {
  if (!c || !x || !y || !sal || !frames_dev || !mix_out || B <= 0 || C <= 0 || T <= 0 ||
      (knots && n_knots < 2) || (mode != 0 && mode != 1) || (!labels_host && (size_t)B > c->lab_cap))
    return hipErrorInvalidValue;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stream_is_capturing(s)) return hipErrorStreamCaptureUnsupported;
  DeviceGuard guard(c->device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return (int)e;
  // slot: [partners int32 B | pad to 16 bytes | knots float64].  The partners go down in a small
  // hipMemcpyAsync (blit kernel); the knots (49 KB at bs 256: SDMA territory, ~25 us of stream
  // stall) are fetched from the pinned slot by an idle block of the search kernel instead.
  const size_t n_mix_pad = ((size_t)B + 3) & ~(size_t)3;
  const size_t nk = knots ? (((size_t)B * n_knots * C + 1) & ~(size_t)1) : 0;   // whole 16-byte words
  const size_t nbytes = n_mix_pad * 4 + nk * sizeof(double);
  const int my_slot = c->next;
  if ((e = slot_reserve(c, my_slot, nbytes)) != hipSuccess) return (int)e;
  Slot& sl = c->slot[my_slot];
  const double* op_dev = nullptr;
  const double* knots_dev = nullptr;
  if (knots) {
    if ((e = spline_op_device(c, T, n_knots, &op_dev)) != hipSuccess) return (int)e;
    std::memcpy(sl.pinned + n_mix_pad * 4, knots, (size_t)B * n_knots * C * sizeof(double));
    knots_dev = reinterpret_cast<const double*>(sl.dev + n_mix_pad * 4);
  }
  const size_t ws_bytes = (size_t)pcgmix_salopt_workspace_bytes(B);
  if (c->ws_cap < ws_bytes) {
    if (c->ws) (void)hipFree(c->ws);   // synchronises: nothing in flight still reads it afterwards
    c->ws = nullptr;
    c->ws_cap = 0;
    if ((e = hipMalloc(&c->ws, ws_bytes)) != hipSuccess) return (int)e;
    c->ws_cap = ws_bytes;
  }
  seed_for_step(c, step);
  std::vector<int64_t> lab64;
  const int64_t* labels = labels_host;
  if (!labels) {
    if ((e = labels_wait(c, s)) != hipSuccess) return (int)e;
    lab64.resize((size_t)B);
    for (int b = 0; b < B; ++b) lab64[(size_t)b] = c->lab[b];
    labels = lab64.data();
  }
  // Up to 256 samples the partners travel in the arguments of the two launches (PartnerPack);
  // beyond that, in a small copy to the slot's device twin.
  int16_t mix16[pcgmix::kPackB];
  const bool in_args = B <= pcgmix::kPackB;
  draw_partners(c, labels, B, mix_out, reinterpret_cast<int32_t*>(sl.pinned), in_args ? mix16 : nullptr);
  const int32_t* mix_dev = nullptr;
  if (!in_args) {
    if ((e = hipMemcpyAsync(sl.dev, sl.pinned, n_mix_pad * 4, hipMemcpyHostToDevice, s)) != hipSuccess)
      return (int)e;
    mix_dev = reinterpret_cast<const int32_t*>(sl.dev);
  }
  // begin kept the boundaries (B <= kPackB): the search is launched as the list of its blocks that
  // have candidates, longest chain first (plan_salopt_blocks) — host work beside the saliency pass
  pcgmix::DispPlan plan;
  plan.n = 0;
  if (in_args && c->sal_frames_known && c->sal_B == B)
    pcgmix::plan_salopt_blocks(c->sal_frames_h, nullptr, mix16, B, T, c->sal_max_len, &plan);
  int err = pcgmix::launch_salopt_search(sal, frames_dev, mix_dev, lam, mode, nullptr, c->ws,
                                         c->sal_B == B ? c->sal_max_len : 0, B, T, s,
                                         sl.pinned + n_mix_pad * 4, sl.dev + n_mix_pad * 4,
                                         (int)(nk * sizeof(double) / 16), in_args ? mix16 : nullptr,
                                         plan.n ? &plan : nullptr);
  if (err) return err;
  err = pcgmix::launch_mix_warp(x, y, frames_dev, mix_dev, nullptr, lam, knots_dev, op_dev,
                                knots ? n_knots : 0, nullptr, B, C, T, s, nullptr, nullptr, 0,
                                static_cast<const float2*>(c->ws), in_args ? mix16 : nullptr);
  if (err) return err;
  return (int)slot_commit(c, my_slot, s);
}

// The frozen CNN_potes saliency pass in ONE call: conv stack forward saving its routing -> head
// (split-K product, dz = (z > 0) * (seed W2), dx = dz W1) -> input gradient from the routing ->
// post-processing.  The same six launches as the four entry points it strings together
// (pcgmix_potes_stack_fwd_save_f32, pcgmix_potes_head_saliency_f32,
// pcgmix_potes_stack_input_grad_mask_f32, pcgmix_saliency_post_f32), without a trip through the
// binding between them: the GPU idled 4-5 us in front of the input gradient and of the forward
// while Python assembled the next call (profiles/r4_cfg3_step_timeline.txt).
extern "C" int pcgmix_potes_saliency_pass_f32(
    const float* x, const float* cw1, const float* cb1, const float* cw2, const float* cb2, float* h2,
    uint8_t* m2, uint8_t* s1, const float* hw1, const float* hb1, const float* hw2, const float* seed,
    float* partial, float* dz, float* gfeat, float* gx, const int32_t* frames, float* sal, int ksize,
    double sigma, int B, int T, int K, int n_classes, pcgmix_stream_t stream) {
  if (B <= 0 || T <= 0) return hipErrorInvalidValue;
  int err = pcgmix_potes_stack_fwd_save_f32(x, cw1, cb1, cw2, cb2, h2, m2, s1, B * 4, T, nullptr, 0, nullptr, 0,
                                            stream);
  if (err) return err;
  err = pcgmix_potes_head_saliency_f32(h2, hw1, hb1, hw2, seed, partial, dz, gfeat, B, K, n_classes, stream);
  if (err) return err;
  err = pcgmix_potes_stack_input_grad_mask_f32(gfeat, m2, s1, cw1, cw2, gx, B * 4, T, stream);
  if (err) return err;
  return pcgmix_saliency_post_f32(gx, frames, sal, ksize, sigma, B, 4, T, stream);
}

// Diagnostic: mean host nanoseconds per call spent in the 8 phases of pcgmix_augment_plain_f32
// since the last query (label kernel launch | slot reserve | pack + seed | label wait | grouping +
// permutation | H2D enqueue | kernel launch | event record); resets the accumulators.
extern "C" long long pcgmix_ctx_phase_times(pcgmix_ctx* c, double* out8) {
  if (!c || !out8) return 0;
  const long long n = c->calls;
  for (int i = 0; i < 8; ++i) {
    out8[i] = n ? c->phase_ns[i] / (double)n : 0.0;
    c->phase_ns[i] = 0.0;
  }
  c->calls = 0;
  return n;
}
