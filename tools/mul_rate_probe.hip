// issue cost of the 32 x 32 -> 64 bit multiply on gfx950: v_mul_lo_u32 + v_mul_hi_u32 against one v_mad_u64_u32 (one wave per SIMD,
// eight independent chains)   hipcc --offload-arch=gfx950 -O3 tools/mul_rate_probe.hip -o variants/mul_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out, uint64_t *cyc, int reps) {
  uint32_t c[8];
  for (int i = 0; i < 8; i++) c[i] = threadIdx.x * 2654435761u + i;
  const uint32_t M = 0xD2511F53u;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint32_t h = __umulhi(M, c[i]), l = M * c[i]; c[i] = h ^ l ^ (uint32_t)r; }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t p; uint64_t carry;
      asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"(c[i]), "v"(M));
      c[i] = (uint32_t)(p >> 32) ^ (uint32_t)p ^ (uint32_t)r;
    }
  }
  uint64_t t2 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { c[i] = (c[i] ^ (uint32_t)r) + (c[i] >> 3); }
  }
  uint64_t t3 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= c[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
}
int main() {
  uint32_t *out; uint64_t *cyc;
  hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
  const int reps = 20000;
  for (int pass = 0; pass < 2; pass++) { hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, out, cyc, reps); hipDeviceSynchronize(); }
  uint64_t r[3];
  hipMemcpy(r, cyc, 24, hipMemcpyDeviceToHost);
  std::printf("per 32x32->64 product (one wave per SIMD, clock counter units): mul_lo + mul_hi + 2 xor: %.1f   mad_u64_u32 + 2 xor + shift: %.1f   (3 simple ops: %.1f)\n",
              (double)r[0] / reps / 8, (double)r[1] / reps / 8, (double)r[2] / reps / 8);
  return 0;
}
