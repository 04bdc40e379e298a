// Internal header shared by the kernel translation units of libpcgmix_hip.so.
#ifndef PCGMIX_KERNELS_H
#define PCGMIX_KERNELS_H
#include <hip/hip_runtime.h>

#include "pcgmix_hip.h"   // public C ABI (include/)

namespace pcgmix {

// Large dynamic-LDS opt-in is a per-device property of the kernel: raise it once per
// (kernel, device) — `done` is a bitmask indexed by the current device (<= 64 devices).
inline hipError_t allow_large_lds(const void* kernel, unsigned long long* done, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (*done & bit) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) *done |= bit;
  return e;
}

// pcgmix_potes.hip: split-K partial products of the skinny linear layer (see there).
// mask != nullptr: h holds the features before dropout; element e owns `bits` random bits of mask
// (bit offset e*bits), kept iff their value >= thr, times scale.
hipError_t launch_skinny_partial(const float* h, const float* W, float* partial, int B, int K,
                                 int O, hipStream_t s, const uint8_t* mask, float scale, int thr,
                                 int bits);

// pcgmix_mix.hip: pcgmix_mix_warp_f32 plus an optional payload of pay_n16 16-byte words that
// block (0,0,0) copies from pay_src to pay_dst (both 16-byte aligned device addresses).
// disp_part != nullptr: the per-state offsets are not read from `off` but reduced by every block
// from the displacement search's per-block results (kDispSplit float2 {value, displacement bits}
// per (sample, state); pcgmix_saliency.hip) — the search's own finalize launch is then not needed.
constexpr int kDispSplit = 4;
int launch_mix_warp(const float* x, float* y, const int32_t* frames, const int32_t* mix_idx,
                    const int32_t* off, float lam, const double* knots, const double* spline_op,
                    int n_knots, const int32_t* zero_rect, int B, int C, int T, hipStream_t s,
                    const void* pay_src, void* pay_dst, int pay_n16,
                    const float2* disp_part = nullptr, const int16_t* partners16 = nullptr);

// pcgmix_mix.hip: the plain splice (no offsets, no warp, no rectangle) with its index block in
// the kernel ARGUMENTS instead of device memory: frames (B,5) and partners (B) as int16 in host
// memory, B <= kPackB, T <= 32767, T % 4 == 0, x and y 16-byte aligned.  Returns
// hipErrorInvalidValue when the shape does not qualify (the caller then takes the copy path).
constexpr int kPackB = 256;
constexpr int kPackPayBytes = 320;   // step payload that still fits next to the index block
// Partner indices of up to kPackB samples as int16, two per dword, passed BY VALUE in the kernel
// arguments of the displacement search and of the splice (saliency-guided step: the partners are
// the only per-step index data that depends on the labels — carried by the launches themselves,
// they need no host-to-device copy).  n == 0: not in use, the kernels read mix_idx from memory.
struct PartnerPack {
  int32_t w[kPackB / 2];
  int n;
};
__device__ __forceinline__ int partner_get(const PartnerPack& p, int b) {
  const int w = p.w[b >> 1];
  return (b & 1) ? (w >> 16) : ((int)((unsigned)w << 16) >> 16);
}
inline PartnerPack make_partner_pack(const int16_t* partners16, int B) {
  PartnerPack pk;
  pk.n = 0;
  if (partners16 && B > 0 && B <= kPackB) {
    int16_t* p16 = reinterpret_cast<int16_t*>(pk.w);
    for (int b = 0; b < B; ++b) p16[b] = partners16[b];
    pk.n = B;
  }
  return pk;
}
// pay: up to kPackPayBytes bytes (host) that also travel in the arguments; block (0,0) writes them,
// rounded up to 16, to pay_dst (device, 16-byte aligned) — the step payload of pcgmix_ctx_set_payload.
int launch_mix_karg(const float* x, float* y, const int16_t* frames16, const int16_t* mix16, float lam,
                    int B, int C, int T, hipStream_t s, const void* pay = nullptr, int pay_bytes = 0,
                    void* pay_dst = nullptr);

// The plain splice launched BEFORE its index block exists ("armed"): a strict-signature step is the
// chain  previous splice -> label arg-max -> host (labels -> partners) -> splice,  and with two
// launches the host's reaction includes a launch (3 us) and the time the command processor takes to
// start it (3-4 us).  Armed, ONE kernel is enqueued at the start of the call: its block (0,0) does
// the label arg-max (labels under the step's token to host-mapped memory), then every
// sample's first block polls that sample's 64-byte RECORD in host-mapped memory until the host has
// written it, relays it to device memory, and the sample's other blocks pick it up there.
// A record is six 8-byte words, each = (stamp << 32) | two int16 values; the stamp is the step's
// sequence number, so a word is valid by itself (8-byte aligned loads and stores are single-copy
// atomic on both sides): no flag, no ordering between words, no torn record.  Values: own
// boundaries f[0..4], partner index, the partner's boundaries f[0..4], 0; word 6 = the step's lambda
// (float bits) for the plain launch, which may be made before lambda is drawn.
// stamp | kArmedAbort in a record = "give up": written by the host (bad input after the launch) or
// by a relay that saw nothing within timeout_ticks of the 100 MHz clock (then also *abort_h = seq);
// every wave leaves without touching y.
constexpr uint32_t kArmedAbort = 0x80000000u;
constexpr int kArmedRecWords = 8;            // 64 bytes per sample: six index words, lambda, (device) knots-ready
struct ArmedArgs {
  const int64_t* ohe;                        // (B, K) one-hot on the device
  int K;
  unsigned long long* lab64;                 // host-mapped: 64 words = (token << 32) | labels of rows 4l .. 4l+3,
  uint32_t token;                            //   one byte each (K <= 256)
  const unsigned long long* rec_h;           // host-mapped records (kPackB x kArmedRecWords)
  unsigned long long* rec_d;                 // device relay of the same shape
  uint32_t* abort_h;                         // host-mapped
  uint32_t seq;                              // 1 .. 0x7fffffff
  unsigned long long timeout_ticks;
};
int launch_mix_armed(const float* x, float* y, const ArmedArgs& a, int B, int C, int T, hipStream_t s,
                     const void* pay = nullptr, int pay_bytes = 0, void* pay_dst = nullptr);
// The same for the splice + warp kernel (durmixmagwarp): knots_host = this step's knots (B, n_knots, C)
// float64 in device-readable pinned memory, knots_dev = device scratch of the same size (each sample's
// relay copies its knots there once), spline_op the constant operator on the device.
int mix_tq_armed_ok(int B, int C, int T, int n_knots);
int launch_mix_tq_armed(const float* x, float* y, const ArmedArgs& a, float lam, const double* knots_host,
                        double* knots_dev, const double* spline_op, int n_knots, int B, int C, int T, hipStream_t s,
                        const void* pay = nullptr, int pay_bytes = 0, void* pay_dst = nullptr);
// first maximum of row b of a (B, K) int64 one-hot matrix (torch.max / np.argmax, augmentations.py:501)
__device__ __forceinline__ int onehot_argmax(const int64_t* __restrict__ ohe, int K, int b) {
  const int64_t* row = ohe + (size_t)b * K;
  if (K == 2 && !(reinterpret_cast<uintptr_t>(ohe) & 15)) {
    // two classes (the reference's data sets): the row is ONE 16-byte load; the loop below is a load,
    // a wait and a compare per class — two memory round trips in the kernel the host waits for
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    const ll2 v = *reinterpret_cast<const ll2*>(row);
    return v.y > v.x ? 1 : 0;
  }
  int best = 0;
  int64_t bv = row[0];
  for (int c = 1; c < K; ++c) {
    const int64_t v = row[c];
    if (v > bv) { bv = v; best = c; }
  }
  return best;
}

// Labels of rows r0 .. r0+3 (rows >= B read row B-1 and report 0) packed one byte each.  For two classes
// the four 16-byte rows are loaded FIRST and compared afterwards: four calls of onehot_argmax compile to
// load, wait, compare four times over — four memory round trips in front of the word the host waits for.
__device__ __forceinline__ uint32_t onehot_argmax4(const int64_t* __restrict__ ohe, int K, int r0, int B) {
  uint32_t packed = 0;
  if (K == 2 && !(reinterpret_cast<uintptr_t>(ohe) & 15)) {
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    ll2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = r0 + j < B ? r0 + j : B - 1;
      v[j] = *reinterpret_cast<const ll2*>(ohe + (size_t)r * 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) packed |= (uint32_t)((r0 + j < B && v[j].y > v[j].x) ? 1 : 0) << (8 * j);
    return packed;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + j;
    const int best = onehot_argmax(ohe, K, r < B ? r : B - 1);
    packed |= (uint32_t)(r < B ? best & 0xff : 0) << (8 * j);
  }
  return packed;
}

// pcgmix_saliency.hip: the displacement search of pcgmix_salopt_disp_f32; disp == nullptr leaves
// the per-block results in `workspace` for launch_mix_warp's disp_part.
// pay_*: pay_n16 16-byte words that one otherwise idle block copies from pay_src (device-readable
// host memory) to pay_dst while the search runs.
// plan != nullptr (plan->n > 0): the launch holds exactly the listed blocks, heaviest first
// (plan_salopt_blocks), instead of the full (sample, slice, state) grid.
constexpr int kDispPlanMax = 1408;   // with the PartnerPack still inside the 4 KB of kernel arguments
struct DispPlan {
  int n;                             // 0: no plan (natural grid)
  uint16_t e[kDispPlanMax];          // (sample << 4) | (state << 2) | slice
};
struct DispNoPlan { int n; };
// Host: the blocks of the search that have candidates — slice z of pair (b, k) holds the
// displacements z*256 .. z*256+255, +1024, ... — ordered by the length of their chain of sums
// (own state's length x passes), longest first, so that the dispatcher deals the long ones out
// first and evenly; a pair without a search gets its slice 0 (which marks all four slices empty).
// frames_h: (B,5) as the device copy; partners as int32 or int16 (one of them).  false: B too
// large or more blocks than the plan holds — launch without a plan.
bool plan_salopt_blocks(const int32_t* frames_h, const int32_t* mix_h, const int16_t* mix16, int B,
                        int T, int max_len, DispPlan* out);
int launch_salopt_search(const float* sal, const int32_t* frames, const int32_t* mix_idx, float lam,
                         int mode, int32_t* disp, void* workspace, int max_len, int B, int T,
                         hipStream_t s, const void* pay_src = nullptr, void* pay_dst = nullptr,
                         int pay_n16 = 0, const int16_t* partners16 = nullptr,
                         const DispPlan* plan = nullptr);

}  // namespace pcgmix
#endif
