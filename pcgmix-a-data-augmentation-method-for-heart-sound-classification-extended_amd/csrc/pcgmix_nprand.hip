// pcgmix_nprand.hip — numpy's LEGACY global random stream restated in C, for the two draws every
// PCGmix step takes from it (host code only, no device code in this file):
//
//     np.random.seed(step); lam = np.random.beta(alpha, alpha)          augmentations.py:659-666
//     random_warps = np.random.normal(1.0, sigma, (B, knot + 2, C))     augmentations.py:677
//
// The reference leaves the GLOBAL stream positioned behind those draws; the knots must be the
// very doubles numpy draws (tests/golden `knots`).  numpy spends ~110 us on the 6144 normals of a
// (256, 6, 4) block — an order of magnitude more than the fused splice+warp kernel they feed — and
// a get_state/set_state round trip costs another ~90 us, so a private RandomState on a Python
// worker thread cannot hide it.  Here instead:
//
//   * MT19937 (numpy/random/src/mt19937/mt19937.c: init_genrand seeding for an integer seed,
//     block regeneration, tempering, 53-bit doubles from two words), the legacy polar Gaussian
//     (legacy-distributions.c legacy_gauss: pairs, the SECOND deviate of a pair is returned first,
//     the first one is cached) and Johnk's beta (legacy_beta, a <= 1 and b <= 1) are restated with
//     the same operations in the same order (this file is compiled with -ffp-contract=off; log,
//     pow, exp and sqrt are the process's libm, the functions numpy itself calls);
//   * every rejection-loop trial of the polar method consumes exactly four 32-bit words, so the
//     trials of one 624-word block are evaluated branch-free four at a time and compacted
//     afterwards; the logarithms run over the accepted pairs only;
//   * the generator works IN PLACE on numpy's own state memory
//     (np.random.get_bit_generator().ctypes.state_address -> struct { uint32 key[624]; int pos; }),
//     which is how the global stream ends up exactly where the reference leaves it without a
//     set_state; numpy's Gaussian cache flag is outside that struct, so the entry points insist on
//     an even count of normals (the cache is then empty before and after, as after np.random.seed);
//   * pcgmix_npdraw: the (lam, knots) of step s + 1 depend on (s + 1, alpha, sigma, n) only, so a
//     worker thread draws them on a private state while the caller is busy with step s; a call
//     whose key matches picks the block up and copies the final generator state into numpy's.
//
// Bit-identity with numpy (seed, beta, normal, final state) is asserted by tests/test_host_logic.py
// on every platform the tests run on.
#include <pthread.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <immintrin.h>

#include "pcgmix_hip.h"

namespace {

struct NpMt {           // numpy/random/src/mt19937/mt19937.h: mt19937_state
  uint32_t key[624];
  int pos;
};

constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;

typedef uint32_t v4u __attribute__((vector_size(16)));

inline v4u ld4(const uint32_t* p) { v4u v; std::memcpy(&v, p, 16); return v; }
inline void st4(uint32_t* p, v4u v) { std::memcpy(p, &v, 16); }

inline uint32_t twist(uint32_t a, uint32_t b, uint32_t far) {
  const uint32_t y = (a & kUpper) | (b & kLower);
  return far ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);
}
inline v4u twist4(v4u a, v4u b, v4u far) {
  const v4u y = (a & kUpper) | (b & kLower);
  return far ^ (y >> 1) ^ ((0u - (y & 1u)) & kMatrix);
}

// mt19937_gen: the next 624 raw words.  Word k depends on words k, k+1 (old) and k+397 (old for
// k < 227, new — written 227 steps earlier — after that): four words at a time are independent.
void mt_gen(NpMt* s) {
  uint32_t* mt = s->key;
  int k = 0;
  for (; k + 4 <= 624 - 397; k += 4) st4(mt + k, twist4(ld4(mt + k), ld4(mt + k + 1), ld4(mt + k + 397)));
  for (; k < 624 - 397; ++k) mt[k] = twist(mt[k], mt[k + 1], mt[k + 397]);
  for (; k + 4 <= 623; k += 4) st4(mt + k, twist4(ld4(mt + k), ld4(mt + k + 1), ld4(mt + k - 227)));
  for (; k < 623; ++k) mt[k] = twist(mt[k], mt[k + 1], mt[k - 227]);
  mt[623] = twist(mt[623], mt[0], mt[396]);
  s->pos = 0;
}

inline uint32_t temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline uint32_t mt_next(NpMt* s) {
  if (s->pos >= 624) mt_gen(s);
  return temper(s->key[s->pos++]);
}

inline double mt_double(NpMt* s) {      // mt19937_next_double
  const int32_t a = (int32_t)(mt_next(s) >> 5), b = (int32_t)(mt_next(s) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

void mt_seed(NpMt* s, uint32_t seed) {  // mt19937_seed (RandomState.seed(int))
  for (int pos = 0; pos < 624; ++pos) {
    s->key[pos] = seed;
    seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
  }
  s->pos = 624;
}

// legacy_beta, Johnk's branch (a <= 1 and b <= 1)
double beta_johnk(NpMt* s, double a, double b) {
  for (;;) {
    const double U = mt_double(s), V = mt_double(s);
    const double X = std::pow(U, 1.0 / a), Y = std::pow(V, 1.0 / b);
    const double XpY = X + Y;
    if (XpY <= 1.0 && U + V > 0.0) {
      if (XpY > 0) return X / XpY;
      double logX = std::log(U) / a, logY = std::log(V) / b;
      const double logM = logX > logY ? logX : logY;
      logX -= logM;
      logY -= logM;
      return std::exp(logX - std::log(std::exp(logX) + std::exp(logY)));
    }
  }
}

// The polar method's trials of one generator block, branch-free.  Trial t takes words 4t .. 4t+3
// = two doubles; as 64-bit lanes q = w_even | w_odd << 32 a lane IS one double's two words:
//   a = temper(w_even) >> 5 (27 bits), b = temper(w_odd) >> 6 (26 bits),
//   d = (a * 67108864.0 + b) / 9007199254740992.0   (every operation exact; int -> double through
//       the 2^52 trick, exact below 2^52), x = 2.0 * d - 1.0.
// NL = 64-bit lanes per vector (2: one trial, SSE2 — the x86-64 baseline; 4: two trials, AVX2).
// xs[2t], xs[2t+1] = x1, x2 of trial t; sq = their squares (summed pairwise by the caller).
template <int NL>
struct Lanes {
  typedef uint32_t vu32 __attribute__((vector_size(NL * 8)));
  typedef uint64_t vu64 __attribute__((vector_size(NL * 8)));
  typedef double vf64 __attribute__((vector_size(NL * 8)));
  static inline void trials(const uint32_t* w, int t0, int t1, double* xs, double* sq) {
    const int per = NL / 2;                   // trials per vector
    for (int t = t0; t < t1; t += per) {
      vu32 y;
      std::memcpy(&y, w + 4 * t, sizeof(y));
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680u;
      y ^= (y << 15) & 0xefc60000u;
      y ^= (y >> 18);
      vu64 q;
      std::memcpy(&q, &y, sizeof(q));
      const vu64 a = (q & 0xffffffffull) >> 5, b = q >> 38;
      const vu64 ma = a | 0x4330000000000000ull, mb = b | 0x4330000000000000ull;
      vf64 da, db;
      std::memcpy(&da, &ma, sizeof(da));
      std::memcpy(&db, &mb, sizeof(db));
      da -= 4503599627370496.0;
      db -= 4503599627370496.0;
      const vf64 d = (da * 67108864.0 + db) / 9007199254740992.0;
      const vf64 x = 2.0 * d - 1.0;
      const vf64 x2 = x * x;
      std::memcpy(xs + 2 * t, &x, sizeof(x));
      std::memcpy(sq + 2 * t, &x2, sizeof(x2));
    }
  }
};

void trials_sse2(const uint32_t* w, int t0, int t1, double* xs, double* sq) {
  Lanes<2>::trials(w, t0, t1, xs, sq);
}
__attribute__((target("avx2"))) void trials_avx2(const uint32_t* w, int t0, int t1, double* xs,
                                                 double* sq) {
  if (t0 & 1) { Lanes<2>::trials(w, t0, t0 + 1, xs, sq); ++t0; }      // 156 trials: t1 is even
  Lanes<4>::trials(w, t0, t1, xs, sq);
}
typedef void (*trials_fn)(const uint32_t*, int, int, double*, double*);
const trials_fn g_trials = __builtin_cpu_supports("avx2") ? trials_avx2 : trials_sse2;

// n normals (n even) = n/2 accepted pairs of the polar method, written as numpy's fill loop
// writes them: out[2i] = loc + scale * (f * x2), out[2i+1] = loc + scale * (f * x1).
// `cancel` (may be null): polled once per generator block by the worker's jobs.
bool normal_fill(NpMt* s, double loc, double scale, long long n, double* out,
                 const std::atomic<int>* cancel, std::vector<double>& scratch) {
  const long long pairs = n / 2;
  if ((long long)scratch.size() < 3 * pairs) scratch.resize((size_t)(3 * pairs));
  double* ax1 = scratch.data();
  double* ax2 = ax1 + pairs;
  double* ar2 = ax2 + pairs;
  long long got = 0;
  auto scalar_trial = [&] {
    const double x1 = 2.0 * mt_double(s) - 1.0, x2 = 2.0 * mt_double(s) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    if (!(r2 >= 1.0 || r2 == 0.0)) { ax1[got] = x1; ax2[got] = x2; ar2[got] = r2; ++got; }
  };
  alignas(32) double xs[320], sq[320];
  while (got < pairs) {
    if (cancel && cancel->load(std::memory_order_relaxed)) return false;
    if (s->pos & 3) {               // a caller's stream that is not at a multiple of four words:
      scalar_trial();               // word by word (a trial may straddle two blocks)
      continue;
    }
    if (s->pos >= 624) mt_gen(s);
    const int t0 = s->pos >> 2;     // trials of this block: words 4t .. 4t+3
    g_trials(s->key, t0, 156, xs, sq);
    int t = t0;
    for (; t < 156 && got < pairs; ++t) {
      const double r2 = sq[2 * t] + sq[2 * t + 1];
      ax1[got] = xs[2 * t];
      ax2[got] = xs[2 * t + 1];
      ar2[got] = r2;
      got += !(r2 >= 1.0 || r2 == 0.0);               // (a rejected trial is overwritten)
    }
    s->pos = 4 * t;
  }
  // f = sqrt(-2.0 * log(r2) / r2): the libm calls on their own, the rest vectorises
  for (long long i0 = 0; i0 < pairs; i0 += 256) {
    double lg[256], f[256];
    const int m = (int)(pairs - i0 < 256 ? pairs - i0 : 256);
    for (int i = 0; i < m; ++i) lg[i] = std::log(ar2[i0 + i]);
    for (int i = 0; i < m; ++i) f[i] = std::sqrt(-2.0 * lg[i] / ar2[i0 + i]);
    for (int i = 0; i < m; ++i) {
      out[2 * (i0 + i)] = loc + scale * (f[i] * ax2[i0 + i]);
      out[2 * (i0 + i) + 1] = loc + scale * (f[i] * ax1[i0 + i]);
    }
  }
  return true;
}

}  // namespace

// ---- primitives (tests, callers with their own loop) ----------------------------------------------

extern "C" int pcgmix_np_seed(void* mt_state, uint32_t seed) {
  if (!mt_state) return 1;
  mt_seed(static_cast<NpMt*>(mt_state), seed);
  return 0;
}

extern "C" int pcgmix_np_beta(void* mt_state, double a, double b, double* out) {
  if (!mt_state || !out || !(a > 0.0) || !(b > 0.0) || a > 1.0 || b > 1.0) return 1;
  NpMt* s = static_cast<NpMt*>(mt_state);
  if (s->pos < 0 || s->pos > 624) return 1;
  *out = beta_johnk(s, a, b);
  return 0;
}

extern "C" int pcgmix_np_normal_fill(void* mt_state, double loc, double scale, long long n,
                                     double* out) {
  if (!mt_state || n < 0 || (n & 1) || (n && !out)) return 1;
  NpMt* s = static_cast<NpMt*>(mt_state);
  if (s->pos < 0 || s->pos > 624) return 1;
  std::vector<double> scratch;
  normal_fill(s, loc, scale, n, out, nullptr, scratch);
  return 0;
}

// ---- one step's (lambda, knots), drawn ahead ------------------------------------------------------

namespace {
// A forked child has the object but not its threads: it draws inline from then on.
std::atomic<int> g_fork_generation{0};
void mark_forked() { g_fork_generation.fetch_add(1, std::memory_order_relaxed); }

struct DrawKey {
  uint32_t seed = 0;
  double alpha = 0, sigma = 0;
  long long n = -1;
  bool operator==(const DrawKey& o) const {
    return seed == o.seed && alpha == o.alpha && sigma == o.sigma && n == o.n;
  }
};

// One block = what one step consumes; entry e serves the seeds congruent to e and owns a thread.
struct DrawEntry {
  DrawKey key;
  NpMt end;                         // generator state behind the draws
  double lam = 0;
  std::vector<double> knots, scratch;
  std::thread th;
  std::atomic<int> job{0};          // 0 idle, 1 posted / running, 2 done, 3 cancelled
  std::atomic<int> cancel{0};

  bool draw(const DrawKey& k, const std::atomic<int>* stop) {
    mt_seed(&end, k.seed);
    lam = beta_johnk(&end, k.alpha, k.alpha);
    if ((long long)knots.size() < k.n) knots.resize((size_t)k.n);
    return normal_fill(&end, 1.0, k.sigma, k.n, knots.data(), stop, scratch);
  }
};
}  // namespace

struct pcgmix_npdraw {
  static constexpr int kMaxEntries = 4;
  static constexpr int kSpinUs = 2000;
  DrawEntry entry[kMaxEntries];
  int entries = 1;                  // lookahead + 1
  bool use_workers = false;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> quit{false};
  long long hits = 0, misses = 0;
  int fork_generation = 0;

  void run(DrawEntry* e) {
    for (;;) {
      // a hot loop posts a job every few ten microseconds: spin for a while before sleeping (a
      // sleeping worker costs its wake-up, tens of microseconds and more under a hypervisor, right
      // when a short timed region starts)
      const auto t0 = std::chrono::steady_clock::now();
      int spins = 0;
      while (e->job.load(std::memory_order_acquire) != 1 && !quit.load(std::memory_order_relaxed)) {
        if ((++spins & 63) ||
            std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(kSpinUs)) {
          _mm_pause();
          continue;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return e->job.load(std::memory_order_acquire) == 1 || quit.load(); });
      }
      if (quit.load()) return;
      const bool ok = e->draw(e->key, &e->cancel);       // key: posted by the caller
      e->job.store(ok ? 2 : 3, std::memory_order_release);
    }
  }

  // Wait until entry e is not running; returns its final job state and leaves it idle.
  int settle(DrawEntry* e, bool abandon) {
    int j = e->job.load(std::memory_order_acquire);
    if (j == 0) return 0;
    if (abandon) e->cancel.store(1, std::memory_order_relaxed);
    while ((j = e->job.load(std::memory_order_acquire)) == 1) _mm_pause();   // <= one block of work
    e->cancel.store(0, std::memory_order_relaxed);
    e->job.store(0, std::memory_order_relaxed);
    return j;
  }
};

extern "C" int pcgmix_npdraw_create(pcgmix_npdraw** out, int lookahead) {
  if (!out || lookahead < 0 || lookahead >= pcgmix_npdraw::kMaxEntries) return 1;
  static const int once = pthread_atfork(nullptr, nullptr, mark_forked);
  (void)once;
  pcgmix_npdraw* d = new pcgmix_npdraw();
  d->fork_generation = g_fork_generation.load();
  d->entries = lookahead + 1;
  d->use_workers = lookahead > 0;
  if (d->use_workers)
    for (int i = 0; i < d->entries; ++i) {
      DrawEntry* e = &d->entry[i];
      e->th = std::thread([d, e] { d->run(e); });
    }
  *out = d;
  return 0;
}

extern "C" void pcgmix_npdraw_destroy(pcgmix_npdraw* d) {
  if (!d) return;
  const bool forked = d->fork_generation != g_fork_generation.load();
  if (!forked) {
    for (auto& e : d->entry) e.cancel.store(1);
    {
      std::lock_guard<std::mutex> lk(d->mu);
      d->quit.store(true);
    }
    d->cv.notify_all();
  }
  for (auto& e : d->entry)
    if (e.th.joinable()) { if (forked) e.th.detach(); else e.th.join(); }
  delete d;
}

extern "C" int pcgmix_npdraw_step(pcgmix_npdraw* d, uint32_t seed, double alpha, double sigma,
                                  long long n, void* np_state, double* lam, const double** knots,
                                  int* hit) {
  if (!d || !lam || !knots || n < 0 || (n & 1) || !(alpha > 0.0) || alpha > 1.0) return 1;
  DrawKey k;
  k.seed = seed; k.alpha = alpha; k.sigma = sigma; k.n = n;
  if (d->use_workers && d->fork_generation != g_fork_generation.load(std::memory_order_relaxed)) {
    d->use_workers = false;         // (the thread objects are detached at destroy)
    for (auto& e : d->entry) e.job.store(0);
  }
  DrawEntry* e = &d->entry[seed % (uint32_t)d->entries];
  bool have = false;
  if (d->use_workers && e->job.load(std::memory_order_acquire) != 0) {
    const bool match = e->key == k;
    have = d->settle(e, !match) == 2 && match;
  }
  auto post_ahead = [&] {           // the next steps' blocks, while the caller works on this one
    bool posted = false;
    for (int a = 1; a < d->entries; ++a) {
      DrawKey nk = k;
      nk.seed = seed + (uint32_t)a;
      DrawEntry* ne = &d->entry[nk.seed % (uint32_t)d->entries];
      if (ne == e) continue;                             // (seed wrapped around 2^32)
      if (ne->job.load(std::memory_order_acquire) != 0) {
        if (ne->key == nk) continue;                     // already on its way (or there)
        d->settle(ne, true);
      }
      ne->key = nk;
      ne->job.store(1, std::memory_order_release);
      posted = true;
    }
    if (posted) {
      { std::lock_guard<std::mutex> lk(d->mu); }          // a worker is before its check or asleep
      d->cv.notify_all();
    }
  };
  if (d->use_workers) post_ahead();   // (before an inline draw: the workers start beside it)
  if (have) {
    ++d->hits;
  } else {
    e->key = k;
    e->draw(k, nullptr);
    ++d->misses;
  }
  if (np_state) std::memcpy(np_state, &e->end, sizeof(NpMt));
  *lam = e->lam;
  *knots = e->knots.data();
  if (hit) *hit = have ? 1 : 0;
  return 0;
}

extern "C" long long pcgmix_npdraw_stats(pcgmix_npdraw* d, long long* misses) {
  if (!d) return -1;
  if (misses) *misses = d->misses;
  return d->hits;
}
