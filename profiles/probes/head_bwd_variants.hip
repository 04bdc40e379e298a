// Probe: where does potes_head_bwd_kernel's time go?  Variants of the kernel with pieces removed.
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off head_bwd_variants.hip -o head_bwd_variants
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <dlfcn.h>
#include <vector>
constexpr int kHeadO = 20, kHbCols = 64, kHbRows = 256;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WAVES, bool MASKED, bool FMA, bool STORE, bool LOADX>
__global__ __launch_bounds__(WAVES * 64) void k(const float* __restrict__ dz, const float* __restrict__ x,
    const uint8_t* __restrict__ mask1, float scale1, const float* __restrict__ w1, float* __restrict__ dw1,
    float* __restrict__ dx, int B, int K) {
  __shared__ __align__(16) float dzl[kHbRows * kHeadO];
  __shared__ float red[8][kHeadO][kHbCols];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * kHbCols, kk = k0 + lane;
  const bool valid = kk < K;
  float wcol[kHeadO], acc[kHeadO];
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) { wcol[o] = w1[(size_t)o * K + (valid ? kk : 0)]; acc[o] = 0.f; }
  for (int b0 = 0; b0 < B; b0 += kHbRows) {
    const int nb = B - b0 < kHbRows ? B - b0 : kHbRows;
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 5; i += WAVES * 64)
      *reinterpret_cast<f4*>(dzl + 4 * i) = reinterpret_cast<const f4*>(dz + (size_t)b0 * kHeadO)[i];
    __syncthreads();
    constexpr int kPer = kHbRows / WAVES;
    float xv[kPer]; uint8_t mb[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + WAVES * j;
      const size_t e = (valid && r < nb) ? (size_t)(b0 + r) * K + kk : 0;
      xv[j] = LOADX ? x[e] : (float)r;
      mb[j] = MASKED ? mask1[e] : 1;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + WAVES * j;
      if (r < nb) {
        float s = 0.f;
        if (FMA) {
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const f4 d = *reinterpret_cast<const f4*>(dzl + r * kHeadO + 4 * q);
            s = fmaf(d.x, wcol[4 * q], s); s = fmaf(d.y, wcol[4 * q + 1], s);
            s = fmaf(d.z, wcol[4 * q + 2], s); s = fmaf(d.w, wcol[4 * q + 3], s);
            acc[4 * q] = fmaf(d.x, xv[j], acc[4 * q]); acc[4 * q + 1] = fmaf(d.y, xv[j], acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(d.z, xv[j], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(d.w, xv[j], acc[4 * q + 3]);
          }
        } else { s = xv[j]; acc[j % 20] += xv[j]; }
        if (STORE) { if (valid) dx[(size_t)(b0 + r) * K + kk] = mb[j] ? s * scale1 : 0.f; }
        else acc[(j + 1) % 20] += mb[j] ? s : 0.f;
      }
    }
  }
  if (wave < 8) {
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) red[wave][o][lane] = acc[o];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kHeadO * kHbCols; i += WAVES * 64) {
    const int o = i / kHbCols, l = i - o * kHbCols;
    if (k0 + l < K) { float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[w][o][l];
      dw1[(size_t)o * K + k0 + l] = a; }
  }
}

// rows split over gridDim.y = 2 halves; dW1 by float atomicAdd onto zeroed memory (two addends per
// element: order-independent).  8 waves x 16 rows per 128-row chunk -> ~100 VGPRs, blocks co-reside.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int WAVES, int ROWS, bool MASKED, bool PK = false>
__global__ __launch_bounds__(WAVES * 64) void k2(const float* __restrict__ dz, const float* __restrict__ x,
    const uint8_t* __restrict__ mask1, float scale1, const float* __restrict__ w1, float* __restrict__ dw1,
    float* __restrict__ dx, int B, int K) {
  __shared__ __align__(16) float dzl[ROWS * kHeadO];
  __shared__ float red[WAVES][kHeadO][kHbCols];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * kHbCols, kk = k0 + lane;
  const bool valid = kk < K;
  const int bh = (B + gridDim.y - 1) / gridDim.y;
  const int row_lo = blockIdx.y * bh, row_hi = min(B, row_lo + bh);
  float wcol[kHeadO], acc[kHeadO];
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) { wcol[o] = w1[(size_t)o * K + (valid ? kk : 0)]; acc[o] = 0.f; }
  for (int b0 = row_lo; b0 < row_hi; b0 += ROWS) {
    const int nb = row_hi - b0 < ROWS ? row_hi - b0 : ROWS;
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 5; i += WAVES * 64)
      *reinterpret_cast<f4*>(dzl + 4 * i) = reinterpret_cast<const f4*>(dz + (size_t)b0 * kHeadO)[i];
    __syncthreads();
    constexpr int kPer = ROWS / WAVES;
    float xv[kPer]; uint8_t mb[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + WAVES * j;
      const size_t e = (valid && r < nb) ? (size_t)(b0 + r) * K + kk : 0;
      xv[j] = x[e];
      mb[j] = MASKED ? mask1[e] : 1;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + WAVES * j;
      if (r < nb) {
        float s = 0.f;
        if (PK) {
          f2 s2 = {0.f, 0.f};
          const f2 xx = {xv[j], xv[j]};
#pragma unroll
          for (int q = 0; q < 10; ++q) {
            const f2 d = *reinterpret_cast<const f2*>(dzl + r * kHeadO + 2 * q);
            const f2 w = {wcol[2 * q], wcol[2 * q + 1]};
            s2 = __builtin_elementwise_fma(d, w, s2);
            f2 a = {acc[2 * q], acc[2 * q + 1]};
            a = __builtin_elementwise_fma(d, xx, a);
            acc[2 * q] = a.x; acc[2 * q + 1] = a.y;
          }
          s = s2.x + s2.y;
        } else {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const f4 d = *reinterpret_cast<const f4*>(dzl + r * kHeadO + 4 * q);
          s = fmaf(d.x, wcol[4 * q], s); s = fmaf(d.y, wcol[4 * q + 1], s);
          s = fmaf(d.z, wcol[4 * q + 2], s); s = fmaf(d.w, wcol[4 * q + 3], s);
          acc[4 * q] = fmaf(d.x, xv[j], acc[4 * q]); acc[4 * q + 1] = fmaf(d.y, xv[j], acc[4 * q + 1]);
          acc[4 * q + 2] = fmaf(d.z, xv[j], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(d.w, xv[j], acc[4 * q + 3]);
        }
        }
        if (valid) dx[(size_t)(b0 + r) * K + kk] = mb[j] ? s * scale1 : 0.f;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) red[wave][o][lane] = acc[o];
  __syncthreads();
  for (int i = threadIdx.x; i < kHeadO * kHbCols; i += WAVES * 64) {
    const int o = i / kHbCols, l = i - o * kHbCols;
    if (k0 + l < K) { float a = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) a += red[w][o][l];
      if (gridDim.y == 1) dw1[(size_t)o * K + k0 + l] = a; else atomicAdd(dw1 + (size_t)o * K + k0 + l, a); }
  }
}

// dz rows are wave-uniform: fetch them with scalar loads (SGPR operands of the FMAs), no LDS.
template <int WAVES, int ROWS, bool MASKED>
__global__ __launch_bounds__(WAVES * 64) void k3(const float* __restrict__ dz, const float* __restrict__ x,
    const uint8_t* __restrict__ mask1, float scale1, const float* __restrict__ w1, float* __restrict__ dw1,
    float* __restrict__ dx, int B, int K) {
  __shared__ float red[WAVES][kHeadO][kHbCols];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k0 = blockIdx.x * kHbCols, kk = k0 + lane;
  const bool valid = kk < K;
  const int bh = (B + gridDim.y - 1) / gridDim.y;
  const int row_lo = blockIdx.y * bh, row_hi = min(B, row_lo + bh);
  float wcol[kHeadO], acc[kHeadO];
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) { wcol[o] = w1[(size_t)o * K + (valid ? kk : 0)]; acc[o] = 0.f; }
  for (int b0 = row_lo; b0 < row_hi; b0 += ROWS) {
    const int nb = row_hi - b0 < ROWS ? row_hi - b0 : ROWS;
    constexpr int kPer = ROWS / WAVES;
    float xv[kPer]; uint8_t mb[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + WAVES * j;
      const size_t e = (valid && r < nb) ? (size_t)(b0 + r) * K + kk : 0;
      xv[j] = x[e];
      mb[j] = MASKED ? mask1[e] : 1;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int r = wave + WAVES * j;
      if (r < nb) {
        const float* __restrict__ drow = dz + (size_t)(b0 + r) * kHeadO;   // uniform address
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < kHeadO; ++o) {
          const float d = drow[o];
          s = fmaf(d, wcol[o], s);
          acc[o] = fmaf(d, xv[j], acc[o]);
        }
        if (valid) dx[(size_t)(b0 + r) * K + kk] = mb[j] ? s * scale1 : 0.f;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < kHeadO; ++o) red[wave][o][lane] = acc[o];
  __syncthreads();
  for (int i = threadIdx.x; i < kHeadO * kHbCols; i += WAVES * 64) {
    const int o = i / kHbCols, l = i - o * kHbCols;
    if (k0 + l < K) { float a = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) a += red[w][o][l];
      if (gridDim.y == 1) dw1[(size_t)o * K + k0 + l] = a; else atomicAdd(dw1 + (size_t)o * K + k0 + l, a); }
  }
}

__global__ void fill(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((h & 0xffff) / 65536.f - 0.5f) * ((h >> 16 & 3) ? 1.f : 0.f); }
}
__global__ void fillb(uint8_t* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (h & 3) != 0; }
}
template <typename F> void timeit(const char* name, F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < 100; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-40s %7.1f us\n", name, ms * 10.f);
}

int main() {
  const int B = 256, K = 19968;
  float *dz, *x, *w1, *dw1, *dx; uint8_t* m;
  hipMalloc(&dz, B * 20 * 4); hipMalloc(&x, (size_t)B * K * 4); hipMalloc(&w1, 20 * K * 4);
  hipMalloc(&dw1, 20 * K * 4); hipMalloc(&dx, (size_t)B * K * 4); hipMalloc(&m, (size_t)B * K);
  hipMemset(dz, 0, B * 20 * 4); hipMemset(x, 0, (size_t)B * K * 4); hipMemset(w1, 0, 20 * K * 4); hipMemset(m, 1, (size_t)B * K);
  if (getenv("RANDOM_DATA")) {
    size_t n = (size_t)B * K;
    hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, 0, x, n, 1u);
    hipLaunchKernelGGL(fill, dim3((20 * K + 255) / 256), dim3(256), 0, 0, w1, (size_t)20 * K, 2u);
    hipLaunchKernelGGL(fill, dim3((B * 20 + 255) / 256), dim3(256), 0, 0, dz, (size_t)B * 20, 3u);
    hipLaunchKernelGGL(fillb, dim3((n + 255) / 256), dim3(256), 0, 0, m, n);
    hipDeviceSynchronize();
  }
  dim3 g((K + 63) / 64);
#define RUN(W, M, F, S, L) timeit(#W " waves mask=" #M " fma=" #F " store=" #S " loadx=" #L, [&] { hipLaunchKernelGGL((k<W, M, F, S, L>), g, dim3(W * 64), 0, 0, dz, x, m, 1.33f, w1, dw1, dx, B, K); })
  RUN(8, true, true, true, true);
  RUN(8, false, true, true, true);
  RUN(8, true, false, true, true);
  RUN(8, true, true, false, true);
  RUN(8, true, true, true, false);
  RUN(8, true, false, false, true);
  RUN(8, false, false, true, false);
  RUN(4, true, true, true, true);
  RUN(16, true, true, true, true);
#define RUN2(W, R, Y) timeit("split rows: " #W " waves, " #R "-row chunks, grid.y=" #Y, [&] { if (Y > 1) hipMemsetAsync(dw1, 0, 20 * K * 4, 0); hipLaunchKernelGGL((k2<W, R, true>), dim3((K + 63) / 64, Y), dim3(W * 64), 0, 0, dz, x, m, 1.33f, w1, dw1, dx, B, K); })
#define RUN3(W, R, Y) timeit("packed, split rows: " #W " waves, " #R "-row chunks, grid.y=" #Y, [&] { if (Y > 1) hipMemsetAsync(dw1, 0, 20 * K * 4, 0); hipLaunchKernelGGL((k2<W, R, true, true>), dim3((K + 63) / 64, Y), dim3(W * 64), 0, 0, dz, x, m, 1.33f, w1, dw1, dx, B, K); })
#define RUN4(W, R, Y) timeit("scalar dz: " #W " waves, " #R "-row chunks, grid.y=" #Y, [&] { if (Y > 1) hipMemsetAsync(dw1, 0, 20 * K * 4, 0); hipLaunchKernelGGL((k3<W, R, true>), dim3((K + 63) / 64, Y), dim3(W * 64), 0, 0, dz, x, m, 1.33f, w1, dw1, dx, B, K); })
  RUN4(8, 256, 1);
  RUN4(8, 128, 1);
  RUN4(8, 128, 2);
  RUN4(8, 64, 2);
  RUN4(4, 64, 2);
  RUN4(4, 32, 4);
  RUN4(16, 256, 1);
  RUN4(16, 128, 2);
  timeit("memset dw1 alone", [&] { hipMemsetAsync(dw1, 0, 20 * K * 4, 0); });
  RUN3(8, 256, 1);
  RUN3(8, 128, 1);
  RUN3(8, 128, 2);
  RUN3(8, 64, 2);
  RUN3(4, 64, 2);
  RUN3(4, 32, 4);
  RUN2(8, 128, 2);
  RUN2(8, 128, 1);
  RUN2(8, 64, 2);
  RUN2(8, 64, 4);
  RUN2(4, 64, 2);
  RUN2(4, 64, 4);
  RUN2(4, 32, 4);
  RUN2(4, 32, 8);
  {  // the product library's own entry point on the same buffers
    void* h = dlopen(getenv("PCGMIX_SO") ? getenv("PCGMIX_SO") : "libpcgmix_hip.so", RTLD_NOW);
    if (h) {
      typedef int (*fn_t)(const float*, const float*, const uint8_t*, float, const float*, const float*,
                          const uint8_t*, float, const float*, float*, float*, float*, float*, float*, float*,
                          int, int, int, void*);
      fn_t fn = (fn_t)dlsym(h, "pcgmix_potes_head_bwd_f32");
      float *dl, *z, *w2, *dw2, *db2, *db1; uint8_t* m2;
      hipMalloc(&dl, B * 2 * 4); hipMalloc(&z, B * 20 * 4); hipMalloc(&w2, 40 * 4); hipMalloc(&dw2, 40 * 4);
      hipMalloc(&db2, 8); hipMalloc(&db1, 80); hipMalloc(&m2, B * 20);
      hipMemset(dl, 0, B * 8); hipMemset(z, 0, B * 80); hipMemset(w2, 0, 160); hipMemset(m2, 1, B * 20);
      timeit("product .so tail_bwd + head_bwd", [&] { fn(dl, z, m2, 2.f, w2, x, m, 1.33f, w1, dz, dw2, db2, db1, dw1, dx, B, K, 2, nullptr); });
    } else printf("dlopen failed: %s\n", dlerror());
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
