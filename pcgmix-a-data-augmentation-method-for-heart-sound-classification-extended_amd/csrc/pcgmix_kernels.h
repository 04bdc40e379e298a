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
