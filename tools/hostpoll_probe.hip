// hostpoll_probe.hip -- what a persistent step kernel's hand-shake with the host costs on this box.
//   hipcc --offload-arch=gfx950 -O2 tools/hostpoll_probe.hip -o tools/hostpoll_probe.bin && tools/hostpoll_probe.bin
// (1) one wave polls a word in host-pinned coherent memory (system-scope loads) and echoes what it sees into
//     another host word: host -> device -> host round trip, and the duration of one device read of host memory;
// (2) the same wave republishes the word into device memory (agent-scope store) while N other waves poll that
//     copy (agent-scope loads) and count the ticks they waited: host -> pump -> workers latency;
// every spin is bounded (the kernel gives up after ~0.5 s) so a mistake cannot hang the box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef unsigned long long u64;

__device__ __forceinline__ u64 now100() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

// block 0: the pump.  blocks 1..: workers polling the device copy.
__global__ void __launch_bounds__(64) probe(const u64 *host_word, u64 *host_echo, u64 *dev_word, u64 *worker_seen, u64 *read_ticks,
                                            u64 last, u64 give_up_ticks) {
  const u64 t0 = now100();
  if (blockIdx.x == 0) {
    u64 seen = 0, reads = 0, ticks = 0;
    while (seen < last) {
      const u64 a = now100();
      const u64 h = __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const u64 b = now100();
      ticks += b - a; reads++;
      if (h != seen) {
        seen = h;
        if (threadIdx.x == 0) {
          __hip_atomic_store(dev_word, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(host_echo, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if (b - t0 > give_up_ticks) break;
    }
    if (threadIdx.x == 0) { read_ticks[0] = ticks; read_ticks[1] = reads; }
  } else {
    u64 seen = 0;
    while (seen < last) {
      const u64 d = __hip_atomic_load(dev_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d != seen) seen = d;
      else __builtin_amdgcn_s_sleep(1);
      if (now100() - t0 > give_up_ticks) break;
    }
    if (threadIdx.x == 0) worker_seen[blockIdx.x] = seen;
  }
}

int main(int argc, char **argv) {
  const int workers = argc > 1 ? std::atoi(argv[1]) : 2048;
  const u64 rounds = 20000;
  u64 *host = nullptr;
  CK(hipHostMalloc((void **)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  volatile u64 *hword = host, *hecho = host + 64;
  *hword = 0; *hecho = 0;
  u64 *d_host = nullptr;
  CK(hipHostGetDevicePointer((void **)&d_host, host, 0));
  u64 *dev = nullptr;
  CK(hipMalloc((void **)&dev, (size_t)(workers + 16 + 64) * 8));
  CK(hipMemset(dev, 0, (size_t)(workers + 16 + 64) * 8));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipLaunchKernelGGL(probe, dim3(1 + workers), dim3(64), 0, st, d_host, d_host + 64, dev, dev + 16, dev + 8, rounds, (u64)50000000);
  CK(hipGetLastError());
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  // round trips: write k, wait for the echo
  std::vector<double> rt;
  bool lost = false;
  for (u64 k = 1; k <= rounds && !lost; k++) {
    const auto a = std::chrono::steady_clock::now();
    *hword = k;
    while (*hecho != k) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 1.0) { lost = true; break; }
    }
    rt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count());
  }
  if (lost) *hword = rounds;   // let the kernel finish
  CK(hipStreamSynchronize(st));
  std::vector<u64> out((size_t)workers + 16 + 64);
  CK(hipMemcpy(out.data(), dev, out.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0, mx = 0, mn = 1e30;
  for (double x : rt) { sum += x; mx = x > mx ? x : mx; mn = x < mn ? x : mn; }
  size_t behind = 0;
  for (int w = 1; w <= workers; w++) behind += out[16 + w] != rounds;
  std::printf("host -> device -> host round trip over %zu rounds: mean %.2f us, min %.2f, max %.2f%s\n", rt.size(), sum / rt.size(), mn, mx,
              lost ? "  (ECHO LOST: the device never saw a host write)" : "");
  std::printf("one device read of host memory: %.2f us (mean of %llu reads)\n", out[9] ? out[8] * 0.01 / out[9] : 0.0, out[9]);
  std::printf("%d worker waves polling the device copy: %zu did not reach the last value\n", workers, behind);

  // one-way host write -> all workers have seen it is what a step hand-off costs; measured through the echo above
  // plus the workers' own poll (agent-scope load, L2 miss): bounded by the round trip.
  CK(hipFree(dev));
  CK(hipHostFree(host));
  return (lost || behind) ? 2 : 0;
}
