// Probe: v_mfma_f32_4x4x1_16b_f32 on gfx950 — operand/result lane map and issue rate.
//   layout : D = A (x) B per block with random operands, checked against the hypothesis
//            A: lane 4b+i = row i of block b;  B: lane 4b+j = column j of block b;
//            D: lane 4b+j, register i = element (i, j) of block b.
//   rate   : 8 independent accumulators, one and two waves per SIMD, with and without four
//            v_max_f32 between the matrix instructions (does the VALU share the issue slot?).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4 profiles/probes/mfma_f32_4x4_rate.hip && /tmp/mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[threadIdx.x * 4 + r] = acc[r];
}

template <int FILL>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters) {
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.0f;
  float f[4] = {a, b, a + b, a - b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      if (FILL) {
#pragma unroll
        for (int j = 0; j < FILL; ++j) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[j & 3]) : "v"(b));
      }
    }
  }
  float s = f[0] + f[1] + f[2] + f[3];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FILL>
static void run_rate(float* out, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<FILL><<<256, threads>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate_kernel<FILL><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 256.0;
  const double per = ms * 1e6 / (iters * 8.0 * waves_per_simd);
  printf("fill=%d waves/SIMD=%.0f: %.3f ms, %.2f ns per MFMA per SIMD (%.1f cycles @2.4GHz), %.1f TFLOP/s\n", FILL,
         waves_per_simd, ms, per, per * 2.4, 256.0 * (threads / 64) * iters * 8 * 512 / (ms * 1e-3) / 1e12);
}

int main() {
  float ha[64], hb[64], hd[256], *a, *b, *d, *out;
  srand(1);
  for (int i = 0; i < 64; ++i) { ha[i] = rand() % 17 - 8; hb[i] = rand() % 13 - 6; }
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice);
  hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l >> 2, j = l & 3;
      const float want = ha[4 * blk + r] * hb[4 * blk + j];
      if (hd[l * 4 + r] != want) ++bad;
    }
  printf("layout hypothesis (D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j]): %s (%d of 256 differ)\n",
         bad ? "WRONG" : "confirmed", bad);
  if (bad)
    for (int l = 0; l < 8; ++l)
      printf("  lane %d: %g %g %g %g   (a=%g b=%g)\n", l, hd[4 * l], hd[4 * l + 1], hd[4 * l + 2], hd[4 * l + 3], ha[l], hb[l]);
  run_rate<0>(out, 256);
  run_rate<0>(out, 512);
  run_rate<1>(out, 256);
  run_rate<2>(out, 256);
  run_rate<4>(out, 256);
  run_rate<2>(out, 512);
  return 0;
}
