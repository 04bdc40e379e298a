// Probe (round 4): issue rate of v_pk_fma_f32 on gfx950 in the forms potes_bwd_pair_kernel uses —
// plain VGPR pairs, op_sel broadcast of one half, an SGPR pair as one source — next to v_fma_f32.
// 8 independent accumulator chains per wave, 4 waves per SIMD (1024 blocks of 256 threads): the
// issue port is what is timed.
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_rate pk_fma_rate.hip && ./pk_fma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WHICH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float sx, float sy) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{1.0f + 1e-3f * (threadIdx.x + i), 0.5f};
  const f2 x = {1.0000001f, 0.9999999f}, y = {1e-9f, 2e-9f};
  const f2 su = {sx, sy};                          // wave-uniform: an SGPR pair
  float b[8];
  for (int i = 0; i < 8; ++i) b[i] = a[i].x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (WHICH == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(x.x), "v"(y.x));
      if (WHICH == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      if (WHICH == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "v"(y));
      if (WHICH == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(x), "v"(y));
      if (WHICH == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(su), "v"(y));
      if (WHICH == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      if (WHICH == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + b[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WHICH>
void run(const char* name, float* out) {
  const int iters = 4000, blocks = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<WHICH><<<blocks, 256>>>(out, iters, 1.0000001f, 0.9999999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<WHICH><<<blocks, 256>>>(out, iters, 1.0000001f, 0.9999999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: blocks*4 waves / 1024 SIMDs, each iters*8 instructions
  const double per = ms * 1e6 / (double(iters) * 8 * (blocks * 4.0 / 1024.0));
  printf("%-44s %7.3f ms  %5.2f ns per wave-instruction per SIMD = %4.1f cycles @2.4 GHz\n", name, ms, per, per * 2.4);
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  run<0>("v_fma_f32", out);
  run<1>("v_pk_fma_f32 (VGPR pairs)", out);
  run<2>("v_pk_fma_f32 op_sel_hi:[1,0,1] (bcast lo)", out);
  run<3>("v_pk_fma_f32 op_sel:[0,1,0] (bcast hi)", out);
  run<4>("v_pk_fma_f32 SGPR-pair source + bcast", out);
  run<5>("v_pk_add_f32", out);
  run<6>("v_pk_mul_f32", out);
  return 0;
}
