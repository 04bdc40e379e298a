// Probe (round 4): issue rate of the float64 VALU instructions the log-mel kernel uses beside its
// f64 matrix instructions — v_fma_f64, v_add_f64, v_mul_f64, v_cvt_f64_f32 — and of
// v_mfma_f64_16x16x4_f64 alone and with float64 VALU work from a SIMD-mate wave.
// 8 independent chains per wave, 4 waves per SIMD (1024 blocks of 256 threads).
//   hipcc --offload-arch=gfx950 -O3 -o f64_valu_rate f64_valu_rate.hip && ./f64_valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int WHICH>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double a[8];
  float f[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    f[i] = 1.0f + 1e-3f * (threadIdx.x + i);
  }
  const double x = 1.0000001, y = 1e-9;
  d4 acc = {0, 0, 0, 0};
  const int wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (WHICH == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if (WHICH == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      if (WHICH == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (WHICH == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      if (WHICH == 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], x, acc, 0, 0, 0);
      if (WHICH == 5) {   // waves 0-2 of the block: matrix instructions; wave 3: v_fma_f64
        if (wave < 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], x, acc, 0, 0, 0);
        else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      }
      if (WHICH == 6) {   // the same with v_add_f64
        if (wave < 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], x, acc, 0, 0, 0);
        else asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      }
    }
  }
  double s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WHICH>
void run(const char* name, double* out, int iters) {
  const int blocks = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<WHICH><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<WHICH><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / (double(iters) * 8 * (blocks * 4.0 / 1024.0));
  printf("%-58s %8.3f ms  %6.2f ns per wave-instruction per SIMD = %5.1f cycles @2.4 GHz\n", name, ms, per, per * 2.4);
  fflush(stdout);
}

int main() {
  double* out;
  hipMalloc(&out, 1024 * 256 * 8);
  run<0>("v_fma_f64", out, 2000);
  run<1>("v_add_f64", out, 2000);
  run<2>("v_mul_f64", out, 2000);
  run<3>("v_cvt_f64_f32", out, 2000);
  run<4>("v_mfma_f64_16x16x4_f64 (4 waves per SIMD)", out, 400);
  run<5>("3 waves v_mfma_f64 + 1 wave v_fma_f64 per SIMD (per instr)", out, 400);
  run<6>("3 waves v_mfma_f64 + 1 wave v_add_f64 per SIMD (per instr)", out, 400);
  return 0;
}
