// Probe: where does the split-K skinny linear (256 x 19968 -> 20) spend its time?
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off skinny_variants.hip -o skinny_variants.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int O = 20;
template <int CHUNK, bool FILL, bool FMA, bool SHFL, int RPT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ h, const float* __restrict__ W,
                                         float* __restrict__ partial, int B, int K) {
  __shared__ __align__(16) float wl[O * CHUNK];
  constexpr int kCols = CHUNK / 64;
  const int r0 = blockIdx.x * 16 * RPT, ks = blockIdx.y, k_lo = ks * CHUNK;
  const int kn = (K - k_lo < CHUNK) ? K - k_lo : CHUNK;
  const int rg = threadIdx.x >> 4, jl = threadIdx.x & 15;
  f4 a[RPT][kCols];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int row = r0 + RPT * rg + r;
    const float* hr = h + (size_t)(row < B ? row : B - 1) * K + k_lo;
#pragma unroll
    for (int c = 0; c < kCols; ++c) { const int j = 4 * (jl + 16 * c); a[r][c] = *reinterpret_cast<const f4*>(hr + (j < kn ? j : 0)); }
  }
  if (FILL) {
    for (int i = threadIdx.x * 4; i < O * CHUNK; i += 1024) {
      const int o = i / CHUNK, j = i - o * CHUNK;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < kn) v = *reinterpret_cast<const f4*>(W + (size_t)o * K + k_lo + j);
      *reinterpret_cast<f4*>(wl + i) = v;
    }
  }
  __syncthreads();
  float acc[RPT][O];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int o = 0; o < O; ++o) acc[r][o] = 0.f;
#pragma unroll
  for (int c = 0; c < kCols; ++c) {
    const int j = 4 * (jl + 16 * c);
#pragma unroll
    for (int o = 0; o < O; ++o) {
      if (FMA) {
        const f4 w = *reinterpret_cast<const f4*>(wl + o * CHUNK + j);
#pragma unroll
        for (int r = 0; r < RPT; ++r)
          acc[r][o] = fmaf(a[r][c].w, w.w, fmaf(a[r][c].z, w.z, fmaf(a[r][c].y, w.y, fmaf(a[r][c].x, w.x, acc[r][o]))));
      } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r][o] += a[r][c].x;
      }
    }
  }
  if (SHFL) {
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int o = 0; o < O; ++o)
#pragma unroll
        for (int s = 8; s > 0; s >>= 1) acc[r][o] += __shfl_xor(acc[r][o], s, 64);
  }
  if (jl == 0) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) { const int row = r0 + RPT * rg + r;
#pragma unroll
      for (int o = 0; o < O; ++o) if (row < B) partial[((size_t)ks * B + row) * O + o] = acc[r][o]; }
  }
}
template <int CHUNK, bool FILL, bool FMA, bool SHFL, int RPT> void run(const char* name, const float* h, const float* W, float* p, int B, int K) {
  dim3 g((B + 16 * RPT - 1) / (16 * RPT), (K + CHUNK - 1) / CHUNK);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<CHUNK, FILL, FMA, SHFL, RPT>), g, dim3(256), 0, 0, h, W, p, B, K);
  hipEventRecord(a);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k<CHUNK, FILL, FMA, SHFL, RPT>), g, dim3(256), 0, 0, h, W, p, B, K);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-58s blocks %4d  %6.1f us\n", name, g.x * g.y, ms * 10.f);
}
int main() {
  const int B = 256, K = 19968;
  float *h, *W, *p; hipMalloc(&h, (size_t)B * K * 4); hipMalloc(&W, 20 * K * 4); hipMalloc(&p, 64 * B * 20 * 4 * 8);
  hipMemset(h, 0, (size_t)B * K * 4); hipMemset(W, 0, 20 * K * 4);
#define R(C, F, M, S, RP) run<C, F, M, S, RP>("chunk " #C " fill=" #F " fma=" #M " shfl=" #S " rows/thread=" #RP, h, W, p, B, K)
  R(512, true, true, true, 2);
  R(512, false, true, true, 2);
  R(512, true, false, true, 2);
  R(512, true, true, false, 2);
  R(512, false, false, false, 2);
  R(512, false, true, false, 2);
  R(256, true, true, true, 2);
  R(256, true, true, true, 4);
  R(512, true, true, true, 1);
  R(1024, true, true, true, 1);
  R(256, true, true, true, 1);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
}
