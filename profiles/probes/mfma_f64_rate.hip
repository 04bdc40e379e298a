// Probe: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (one wave per SIMD, 5 independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[5];
  for (int i = 0; i < 5; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* out;
  hipMalloc(&out, 256 * 256 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1, 256}) {
    k<<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_mfma_ns = ms * 1e6 / (iters * 5.0);
    printf("blocks=%d: %.3f ms, %.1f ns per MFMA per wave (%.0f cycles @2.4GHz), %.2f TFLOP/s\n", blocks, ms,
           per_mfma_ns, per_mfma_ns * 2.4, blocks * 4.0 * iters * 5 * 2048 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
