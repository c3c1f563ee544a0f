// tools/valu_rate_probe.hip -- development micro-benchmark (not part of the product): issue cost of the vector
// instructions the step kernel's noise path is made of, one wave per SIMD, eight independent chains per instruction
// kind, cycles per wave-instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void __launch_bounds__(64) probe(unsigned long long *out, int iters, float seed) {
  float f[8];
  double d[8];
  unsigned u[8];
  unsigned long long w[8];
  for (int k = 0; k < 8; k++) { f[k] = seed + k; d[k] = seed + 2.0 * k; u[k] = (unsigned)(seed * 1000) + k * 977u + threadIdx.x; w[k] = u[k]; }
  unsigned long long mask = 0x5555555555555555ull + (unsigned long long)iters, masks[2] = {0, 0};
  asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#define ONE(k)                                                                                              \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[k]));                                   \
    if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[k]));                                   \
    if (KIND == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(w[k]) : "v"(u[k]) : "vcc");      \
    if (KIND == 3) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[k]));                                    \
    if (KIND == 4) asm volatile("v_mul_hi_u32 %0, %0, %0" : "+v"(u[k]));                                    \
    if (KIND == 5) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[k]) : "v"(u[k]));                           \
    if (KIND == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));                           \
    if (KIND == 7) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[k]));                                       \
    if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));                                           \
    if (KIND == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));                                          \
    if (KIND == 10) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[k]));                                      \
    if (KIND == 11) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[k]));                                      \
    if (KIND == 12) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[k]));                                      \
    if (KIND == 13) asm volatile("v_alignbit_b32 %0, %0, %0, 31" : "+v"(u[k]));                             \
    if (KIND == 14) asm volatile("v_min_u32 %0, %0, %0" : "+v"(u[k]));                                      \
    if (KIND == 15) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[k]));                             \
    if (KIND == 16) asm volatile("v_cmp_gt_f64 vcc, %0, %0" : : "v"(d[k]) : "vcc");                         \
    if (KIND == 17) asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(u[k]));                                  \
    if (KIND == 18) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[k]));                               \
    if (KIND == 19) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[k]));                                   \
    if (KIND == 20) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[k]));                                          \
    if (KIND == 21) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));                          \
    if (KIND == 22) asm volatile("v_cmp_gt_f32 vcc, %0, %0" : : "v"(f[k]) : "vcc");                         \
    if (KIND == 23) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(u[k]));                             \
    if (KIND == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %0, %1" : "+v"(u[k]) : "s"(mask));                  \
    if (KIND == 25) asm volatile("v_cmp_gt_f32_e64 %0, %1, %1" : "=s"(masks[k & 1]) : "v"(f[k]));               \
    if (KIND == 26) asm volatile("v_and_b32 %0, %0, %0" : "+v"(u[k]));                                          \
    if (KIND == 27) asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(u[k]));                                  \
    if (KIND == 28) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(f[k]));                                          \
    if (KIND == 29) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(u[k]));
    REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sink = 0;
  for (int k = 0; k < 8; k++) sink += f[k] + (float)d[k] + (float)u[k] + (float)w[k];
  sink += (float)(masks[0] + masks[1]);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)sink; }
}

template <int KIND>
void run(const char *name, unsigned long long *dev, int waves_per_simd) {
  const int iters = 2000, blocks = 1024 * waves_per_simd;
  probe<KIND><<<blocks, 64>>>(dev, iters, 1.5f);
  probe<KIND><<<blocks, 64>>>(dev, iters, 1.5f);
  hipDeviceSynchronize();
  static unsigned long long host[2 * 16384];
  hipMemcpy(host, dev, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int b = 0; b < blocks; b++) sum += (double)host[2 * b];
  // s_memtime ticks at 100 MHz on this part; report ticks per instruction and let the fp32 FMA line calibrate
  printf("%-20s %d wave(s)/SIMD: %7.3f ticks per instruction of one wave, %6.3f per instruction of the SIMD\n", name, waves_per_simd, sum / blocks / (iters * 32.0), sum / blocks / (iters * 32.0) / waves_per_simd);
}

int main() {
  unsigned long long *dev;
  hipMalloc(&dev, sizeof(unsigned long long) * 2 * 16384);
  for (int w = 1; w <= 8; w *= 2) {
    run<0>("v_fma_f32", dev, w); run<18>("v_pk_fma_f32", dev, w); run<1>("v_fma_f64", dev, w); run<10>("v_add_f64", dev, w); run<11>("v_mul_f64", dev, w);
    run<2>("v_mad_u64_u32", dev, w); run<3>("v_mul_lo_u32", dev, w); run<4>("v_mul_hi_u32", dev, w); run<17>("v_mul_u32_u24", dev, w);
    run<5>("v_cvt_f64_u32", dev, w); run<6>("v_cvt_f32_f64", dev, w); run<21>("v_cvt_f64_f32", dev, w); run<7>("v_cvt_f32_u32", dev, w);
    run<8>("v_rcp_f32", dev, w); run<20>("v_rsq_f32", dev, w); run<9>("v_sqrt_f32", dev, w); run<19>("v_frexp_mant_f64", dev, w); run<23>("v_ldexp_f64", dev, w);
    run<12>("v_add_u32", dev, w); run<13>("v_alignbit_b32", dev, w); run<14>("v_min_u32", dev, w); run<15>("v_cndmask_b32", dev, w);
    run<16>("v_cmp_gt_f64", dev, w); run<22>("v_cmp_gt_f32", dev, w); run<24>("v_cndmask_e64 sgpr", dev, w); run<25>("v_cmp_f32_e64 sgpr", dev, w);
    run<26>("v_and_b32", dev, w); run<27>("v_lshl_add_u32", dev, w); run<28>("v_mul_f32", dev, w); run<29>("v_cvt_u32_f32", dev, w);
  }
  return 0;
}
