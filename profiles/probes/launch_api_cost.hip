// Probe: host cost of one kernel launch with a 4 KB by-value argument block (the splice kernel's
// index pack) and with 64 bytes of arguments (the label kernel), through hipLaunchKernelGGL and
// through hipModuleLaunchKernel on a pre-resolved function with a packed argument buffer.
// Measured (MI355X, second repeat): 4 KB arguments 2.77 vs 2.55 us, 64 B 2.49 vs 2.36 us — 0.1-0.2 us
// per launch: not worth a second launch path in the library.
// hipcc --offload-arch=gfx950 -O3 -o build_probe/launch_api profiles/probes/launch_api_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Big { int v[1000]; };
__global__ void kbig(Big b, int* out) { if (b.v[threadIdx.x & 7] == 12345) out[0] = 1; }
__global__ void ksmall(const long long* a, int n, int c, int* l, unsigned* f, unsigned s, float* g) {
  if (n == 12345) l[0] = 1;
}
template <class F> static double per_call_us(F f, int n) {
  for (int i = 0; i < 200; ++i) f();
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  int* out; hipMalloc(&out, 64);
  hipStream_t s; hipStreamCreate(&s);
  Big b; memset(&b, 0, sizeof b);
  hipFunction_t fbig, fsmall;
  if (hipGetFuncBySymbol(&fbig, (const void*)kbig) != hipSuccess || hipGetFuncBySymbol(&fsmall, (const void*)ksmall) != hipSuccess) {
    printf("hipGetFuncBySymbol failed\n"); return 1;
  }
  struct { Big b; int* out; } abig; abig.b = b; abig.out = out;
  size_t sbig = sizeof abig;
  void* xbig[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &abig, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sbig, HIP_LAUNCH_PARAM_END};
  struct { const long long* a; int n; int c; int* l; unsigned* f; unsigned s; float* g; } asm_ = {nullptr, 1, 2, out, nullptr, 3u, nullptr};
  size_t ssm = sizeof asm_;
  void* xsm[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &asm_, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ssm, HIP_LAUNCH_PARAM_END};
  const int N = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    printf("4 KB arguments : hipLaunchKernelGGL %.2f us   hipModuleLaunchKernel %.2f us\n",
           per_call_us([&] { hipLaunchKernelGGL(kbig, dim3(1000), dim3(256), 0, s, b, out); }, N),
           per_call_us([&] { hipModuleLaunchKernel(fbig, 1000, 1, 1, 256, 1, 1, 0, s, nullptr, xbig); }, N));
    printf("64 B arguments : hipLaunchKernelGGL %.2f us   hipModuleLaunchKernel %.2f us\n",
           per_call_us([&] { hipLaunchKernelGGL(ksmall, dim3(1), dim3(256), 0, s, (const long long*)nullptr, 1, 2, out, (unsigned*)nullptr, 3u, (float*)nullptr); }, N),
           per_call_us([&] { hipModuleLaunchKernel(fsmall, 1, 1, 1, 256, 1, 1, 0, s, nullptr, xsm); }, N));
  }
  return 0;
}
