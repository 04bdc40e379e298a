// Probe: issue rate of the float64 VALU operations the magnitude-warp spline uses on gfx950
// (is v_fma_f64 the fast one and v_mul_f64 / v_add_f64 quarter rate?).
// One kernel per instruction; each wave issues 8 independent chains of the instruction back to
// back (inline asm: the compiler can neither fold nor reorder them), 4 waves per SIMD so that the
// issue port, not the dependent latency, is what is timed.
//   hipcc --offload-arch=gfx950 -O3 -o f64_valu_rate f64_valu_rate.hip && ./f64_valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHAIN8(OP)                                                                              \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                    \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),          \
                 "+v"(a[6]), "+v"(a[7])                                                          \
               : "v"(x), "v"(y))

#define FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define ADD(i) "v_add_f64 %" #i ", %" #i ", %9\n"

template <int WHICH>
__global__ __launch_bounds__(256) void k_f64(double* out, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  const double x = 1.0000001, y = 1e-9;
  for (int it = 0; it < iters; ++it) {
    if (WHICH == 0) CHAIN8(FMA);
    if (WHICH == 1) CHAIN8(MUL);
    if (WHICH == 2) CHAIN8(ADD);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// conversions: int32 -> f64, f32 -> f64, f64 -> f32 (each feeds the next input so nothing folds)
template <int WHICH>
__global__ __launch_bounds__(256) void k_cvt(double* out, int iters) {
  double d[8];
  float f[8];
  int n[8];
  for (int i = 0; i < 8; ++i) {
    d[i] = 1.0 + threadIdx.x + i;
    f[i] = 1.0f + threadIdx.x + i;
    n[i] = threadIdx.x + i;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (WHICH == 0) asm volatile("v_cvt_f64_i32 %0, %1\n" : "=v"(d[i]) : "v"(n[i]));
      if (WHICH == 1) asm volatile("v_cvt_f64_f32 %0, %1\n" : "=v"(d[i]) : "v"(f[i]));
      if (WHICH == 2) asm volatile("v_cvt_f32_f64 %0, %1\n" : "=v"(f[i]) : "v"(d[i]));
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i] + f[i] + n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// f32 reference point: v_fma_f32 and v_pk_fma_f32 through the same harness
template <int WHICH>
__global__ __launch_bounds__(256) void k_f32(double* out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);
  const float x = 1.0000001f, y = 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a[i]) : "v"(x), "v"(y));
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, double* out) {
  const int iters = 4000, blocks = 256 * 4;      // 4 blocks of 4 waves per CU: 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x 8 instructions
  const double per_instr_ns = ms * 1e6 / (4.0 * iters * 8.0);
  printf("%-16s %.3f ms  %.2f ns per wave-instruction per SIMD = %.1f cycles @2.4 GHz\n", name, ms,
         per_instr_ns, per_instr_ns * 2.4);
}

int main() {
  double* out;
  hipMalloc(&out, 1024 * 256 * 8);
  run("v_fma_f64", k_f64<0>, out);
  run("v_mul_f64", k_f64<1>, out);
  run("v_add_f64", k_f64<2>, out);
  run("v_cvt_f64_i32", k_cvt<0>, out);
  run("v_cvt_f64_f32", k_cvt<1>, out);
  run("v_cvt_f32_f64", k_cvt<2>, out);
  run("v_fma_f32", k_f32<0>, out);
  return 0;
}
