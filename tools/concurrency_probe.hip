// Does a kernel on one stream run beside a long-running grid on another?  tools/concurrency_probe.hip
//   hipcc --offload-arch=gfx950 -O2 tools/concurrency_probe.hip -o tools/concurrency_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__global__ void __launch_bounds__(64) spin(unsigned long long ticks, unsigned long long *sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); n++; }
  if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = n;
}
__global__ void __launch_bounds__(256) small(float *p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
int main() {
  float *buf; unsigned long long *sink;
  CK(hipMalloc(&buf, 4 << 20)); CK(hipMalloc(&sink, 64));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipLaunchKernelGGL(small, dim3(4096), dim3(256), 0, b, buf, 1 << 20); CK(hipStreamSynchronize(b));
  for (int grid : {1, 256, 2048, 4096, 6144, 8192}) {
    hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, a, 500000ull /* 5 ms */, sink);
    std::this_thread::sleep_for(std::chrono::microseconds(300));
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(small, dim3(4096), dim3(256), 0, b, buf, 1 << 20);
    CK(hipStreamSynchronize(b));
    const double us_b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(a));
    const double us_a = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    std::printf("spinning grid of %5d one-wave workgroups on stream a: a small kernel on stream b finished after %7.0f us (the grid after %7.0f us)\n", grid, us_b, us_a);
  }
  return 0;
}
