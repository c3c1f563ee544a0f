// tools/stream_probe.hip -- development micro-benchmark (not part of the product):
// does the step kernel's access pattern (24 dword read streams + 17 write
// streams, planar SoA) stream slower than the same bytes laid out as 64-vehicle
// tiles (one contiguous block per wave), or than dwordx2 / dwordx4 per lane?
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NR 24
#define NW 17

// planar: comp k of vehicle i at base[k*S + i]
template <int VPL>
__global__ void __launch_bounds__(256) planar(const float *__restrict__ in, float *__restrict__ out, long S, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * VPL;
  if (i >= n) return;
  float acc[VPL];
  float v[NR][VPL];
#pragma unroll
  for (int k = 0; k < NR; k++) {
    if (VPL == 1) v[k][0] = in[k * S + i];
    if (VPL == 2) { float2 t = *(const float2 *)&in[k * S + i]; v[k][0] = t.x; v[k][1] = t.y; }
    if (VPL == 4) { float4 t = *(const float4 *)&in[k * S + i]; v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w; }
  }
#pragma unroll
  for (int j = 0; j < VPL; j++) { acc[j] = 0; for (int k = NW; k < NR; k++) acc[j] += v[k][j]; }
#pragma unroll
  for (int k = 0; k < NW; k++) {
    if (VPL == 1) out[k * S + i] = v[k][0] * 1.0001f + acc[0];
    if (VPL == 2) *(float2 *)&out[k * S + i] = make_float2(v[k][0] * 1.0001f + acc[0], v[k][1] * 1.0001f + acc[1]);
    if (VPL == 4) *(float4 *)&out[k * S + i] = make_float4(v[k][0] * 1.0001f + acc[0], v[k][1] * 1.0001f + acc[1], v[k][2] * 1.0001f + acc[2], v[k][3] * 1.0001f + acc[3]);
  }
}

// planar through ONE buffer resource: address = rsrc base + per-lane 32-bit offset + scalar component offset
// (buffer_load_dword v, voff, s[rsrc], soff offen): no per-lane address arithmetic at all
__global__ void __launch_bounds__(64) planar_buffer(float *base, long S, long n) {
  const unsigned i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(NR * S * 4), 0x00020000);
  const unsigned off = i * 4u;
  const int S4 = (int)S * 4;
  float v[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, k * S4, 0));
  float acc = 0;
  for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NW; k++) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k] * 1.0001f + acc), r, off, k * S4, 0);
}
// the same with WORK dependent-free FMAs per loaded value between the loads and the stores: what arithmetic of the
// step kernel's size (~230 vector instructions off tick, ~800 on tick) costs a launch that streams these bytes
template <int NRR, int NWW, int WORK>
__global__ void __launch_bounds__(64) planar_buffer_work(float *base, long S, long n) {
  const unsigned i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(NR * S * 4), 0x00020000);
  const unsigned off = i * 4u;
  const int S4 = (int)S * 4;
  float v[NRR];
#pragma unroll
  for (int k = 0; k < NRR; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, k * S4, 0));
  float acc = 0;                                 // every load is used, whatever WORK is
#pragma unroll
  for (int k = NWW; k < NRR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NWW; k++) v[k] += acc;
#pragma unroll
  for (int w = 0; w < WORK; w++) {
#pragma unroll
    for (int k = 0; k < NWW; k++) v[k] = __builtin_fmaf(v[k], 1.0000001f, v[(k + 1 + w) % NRR] * 1e-9f);   // 2 instructions each
  }
#pragma unroll
  for (int k = 0; k < NWW; k++) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k]), r, off, k * S4, 0);
}
__global__ void __launch_bounds__(64) planar64(const float *__restrict__ in, float *__restrict__ out, long S, long n) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float v[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = in[k * S + i];
  float acc = 0;
  for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NW; k++) out[k * S + i] = v[k] * 1.0001f + acc;
}

// tiled: 64-vehicle tile = NR rows of 64 floats; comp k of vehicle i at base[(i/64)*NR*64 + k*64 + i%64]
__global__ void __launch_bounds__(256) tiled(const float *__restrict__ in, float *__restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long base = (i >> 6) * (NR * 64) + (i & 63);
  float v[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = in[base + k * 64];
  float acc = 0;
  for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NW; k++) out[base + k * 64] = v[k] * 1.0001f + acc;
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : (1 << 20);   // 1<<22, 1<<24: beyond the 256 MiB Infinity Cache
  long S = n;
  float *a, *b;
  const long SMAX = n + (1 << 16);
  hipMalloc(&a, sizeof(float) * NR * SMAX);
  hipMalloc(&b, sizeof(float) * NR * SMAX);
  hipMemset(a, 0, sizeof(float) * NR * SMAX);
  hipMemset(b, 0, sizeof(float) * NR * SMAX);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)n * 4 * (NR + NW);
  // variants 11..13: dword / dwordx2 / dwordx4 per lane with the engine's odd-multiple-of-256 stride
  const long pads[] = {0, 0, 0, 0, 0, 256, 1024 + 256, 4096 + 256, 16384 + 1024 + 64, 64, 32768 + 2048 + 128, 256, 256, 256};
  for (int variant = 0; variant < 14; variant++) {
    S = n + pads[variant];
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      for (int it = 0; it < (n > (1 << 21) ? 20 : 100); it++) {
        // in-place like the engine: read and write the same buffer
        switch (variant) {
          case 0: planar<1><<<(n + 255) / 256, 256>>>(a, a, S, n); break;
          case 1: planar<2><<<(n / 2 + 255) / 256, 256>>>(a, a, S, n); break;
          case 2: planar<4><<<(n / 4 + 255) / 256, 256>>>(a, a, S, n); break;
          case 3: tiled<<<(n + 255) / 256, 256>>>(a, a, n); break;
          case 4: planar<1><<<(n + 255) / 256, 256>>>(a, b, S, n); break;  // out of place
          case 12: planar<2><<<(n / 2 + 255) / 256, 256>>>(a, a, S, n); break;
          case 13: planar<4><<<(n / 4 + 255) / 256, 256>>>(a, a, S, n); break;
          default: planar<1><<<(n + 255) / 256, 256>>>(a, a, S, n); break;  // padded stride
        }
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ms *= (n > (1 << 21) ? 5.0f : 1.0f);   // normalise to 100 launches
      if (ms < best) best = ms;
    }
    const char *names[] = {"planar dword in-place", "planar dwordx2 in-place", "planar dwordx4 in-place", "tiled-64 dword in-place", "planar dword out-of-place"};
    if (variant < 5) printf("%-28s %.2f us/launch  %.0f GB/s\n", names[variant], best * 10, bytes / (best * 1e-5) / 1e9);
    else if (variant >= 11) printf("planar dwordx%d pad 256        %.2f us/launch  %.0f GB/s\n", 1 << (variant - 11), best * 10, bytes / (best * 1e-5) / 1e9);
    else printf("planar dword pad %-10ld  %.2f us/launch  %.0f GB/s\n", pads[variant], best * 10, bytes / (best * 1e-5) / 1e9);
  }
  // one-wave workgroups like the engine: global addressing vs one buffer resource (stride n + 256)
  S = n + 256;
  for (int variant = 0; variant < 2; variant++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      for (int it = 0; it < (n > (1 << 21) ? 20 : 100); it++) {
        if (variant == 0) planar64<<<(n + 63) / 64, 64>>>(a, a, S, n);
        else planar_buffer<<<(n + 63) / 64, 64>>>(a, S, n);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ms *= (n > (1 << 21) ? 5.0f : 1.0f);
      if (ms < best) best = ms;
    }
    printf("%-28s %.2f us/launch  %.0f GB/s\n", variant ? "64-lane groups, buffer rsrc" : "64-lane groups, global", best * 10, bytes / (best * 1e-5) / 1e9);
  }
  // 20 read + 13 write streams (132 B: the off-tick launch) and 24 + 17 (164 B: the tick launch) with 0 / ~230 / ~800
  // vector instructions between loads and stores
  for (int variant = 0; variant < 6; variant++) {
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0);
      for (int it = 0; it < (n > (1 << 21) ? 20 : 100); it++) {
        switch (variant) {
          case 0: planar_buffer_work<20, 13, 0><<<(n + 63) / 64, 64>>>(a, S, n); break;
          case 1: planar_buffer_work<20, 13, 9><<<(n + 63) / 64, 64>>>(a, S, n); break;     // 9 * 13 * 2 = 234
          case 2: planar_buffer_work<20, 13, 31><<<(n + 63) / 64, 64>>>(a, S, n); break;    // 806
          case 3: planar_buffer_work<24, 17, 0><<<(n + 63) / 64, 64>>>(a, S, n); break;
          case 4: planar_buffer_work<24, 17, 7><<<(n + 63) / 64, 64>>>(a, S, n); break;     // 238
          case 5: planar_buffer_work<24, 17, 24><<<(n + 63) / 64, 64>>>(a, S, n); break;    // 816
        }
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ms *= (n > (1 << 21) ? 5.0f : 1.0f);
      if (ms < best) best = ms;
    }
    const char *names[] = {"132 B, no arithmetic", "132 B, ~230 instructions", "132 B, ~800 instructions", "164 B, no arithmetic", "164 B, ~230 instructions", "164 B, ~800 instructions"};
    printf("%-28s %.2f us/launch\n", names[variant], best * 10);
  }
  return 0;
}
