// Probe: read bandwidth of a (256 x 19968) f32 matrix (20 MB) as a function of how a block walks
// it: each 256-thread block reads ROWS rows x RUN contiguous floats per row with 16-byte loads.
// hipcc -O3 --offload-arch=gfx950 run_length_bw.hip -o run_length_bw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int ROWS, int RUN>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ x, float* __restrict__ out, int B, int K) {
  constexpr int F4_PER_ROW = RUN / 4;               // float4 per row run
  constexpr int TOTAL = ROWS * F4_PER_ROW;          // float4 per block
  constexpr int PER = TOTAL / 256;
  const int c0 = blockIdx.x * RUN, r0 = blockIdx.y * ROWS;
  f4 v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = threadIdx.x + 256 * i;
    const int r = idx / F4_PER_ROW, j = idx - r * F4_PER_ROW;
    int col = c0 + 4 * j; if (col >= K) col = 0;
    v[i] = *reinterpret_cast<const f4*>(x + (size_t)(r0 + r) * K + col);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  if (s == 12345.678f) out[0] = s;
}
template <int ROWS, int RUN> void run(const float* x, float* out, int B, int K) {
  dim3 g((K + RUN - 1) / RUN, B / ROWS);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rd<ROWS, RUN>), g, dim3(256), 0, 0, x, out, B, K);
  hipEventRecord(a);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((rd<ROWS, RUN>), g, dim3(256), 0, 0, x, out, B, K);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("rows/block %3d  run %5d floats  blocks %5d  %6.1f us  %5.2f TB/s\n", ROWS, RUN, g.x * g.y, ms * 10.f,
         (double)B * K * 4 / (ms * 1e-5) / 1e12);
}
int main() {
  const int B = 256, K = 19968;
  float *x, *out; hipMalloc(&x, (size_t)B * K * 4); hipMalloc(&out, 4); hipMemset(x, 0, (size_t)B * K * 4);
  run<64, 64>(x, out, B, K);
  run<32, 128>(x, out, B, K);
  run<16, 256>(x, out, B, K);
  run<32, 256>(x, out, B, K);
  run<8, 512>(x, out, B, K);
  run<16, 512>(x, out, B, K);
  run<32, 512>(x, out, B, K);
  run<4, 1024>(x, out, B, K);
  run<16, 1024>(x, out, B, K);
  run<2, 2048>(x, out, B, K);
  run<8, 2048>(x, out, B, K);
  run<1, 4096>(x, out, B, K);
  run<4, 4096>(x, out, B, K);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
