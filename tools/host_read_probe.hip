// how long one wave waits for a word of pinned host memory: vector load (system scope) against scalar loads of 1, 2, 16 dwords
//   hipcc --offload-arch=gfx950 -O3 tools/host_read_probe.hip -o gpurun_out/host_read_probe && gpurun_out/host_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }
__global__ void probe(const u64 *host, u64 *out, int reps) {
  u64 sink = 0;
  u64 t0 = now();
  for (int i = 0; i < reps; i++) sink += __hip_atomic_load(host + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  u64 t1 = now();
  for (int i = 0; i < reps; i++) { unsigned v; asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(host) : "memory"); sink += v; }
  u64 t2 = now();
  for (int i = 0; i < reps; i++) { u64 v; asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(host) : "memory"); sink += v; }
  u64 t3 = now();
  for (int i = 0; i < reps; i++) { u32x16 v; asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(host) : "memory"); sink += v[0] + v[15]; }
  u64 t4 = now();
  for (int i = 0; i < reps; i++) { u64 v, w; asm volatile("s_load_dwordx2 %0, %2, 0x0 glc\n\ts_load_dwordx2 %1, %2, 0x100 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v), "=&s"(w) : "s"(host) : "memory"); sink += v + w; }
  u64 t5 = now();
  for (int i = 0; i < reps; i++) { u64 v; asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_inv" : "=&s"(v) : "s"(host) : "memory"); sink += v; }
  u64 t6 = now();
  for (int i = 0; i < reps; i++) sink += __hip_atomic_load(host + (threadIdx.x & 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  u64 t7 = now();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t6 - t5; out[6] = t7 - t6; out[7] = sink; }
}
int main() {
  u64 *host, *dhost, *out;
  hipHostMalloc(&host, 4096, hipHostMallocDefault);
  for (int i = 0; i < 512; i++) host[i] = i;
  hipHostGetDevicePointer((void **)&dhost, host, 0);
  hipMalloc(&out, 64);
  const int reps = 2000;
  for (int pass = 0; pass < 2; pass++) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dhost, out, reps);
    hipDeviceSynchronize();
  }
  u64 r[8];
  hipMemcpy(r, out, 64, hipMemcpyDeviceToHost);
  const char *names[7] = {"vector load, 64 lanes x 8 B", "s_load_dword glc", "s_load_dwordx2 glc", "s_load_dwordx16 glc", "two s_load_dwordx2 glc in flight",
                          "s_load_dwordx2 + s_dcache_inv", "vector load, one address in all lanes"};
  for (int i = 0; i < 7; i++) std::printf("%-40s %.2f us per read\n", names[i], r[i] / 100.0 / reps);
  return 0;
}
