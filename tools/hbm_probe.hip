// tools/hbm_probe.hip -- development micro-benchmark (not part of the product): what the step kernel's traffic shape
// can stream BEYOND the 256 MiB Infinity Cache (2^22 .. 2^24 vehicles), and which knob moves it.
// Shape of the off-tick launch of the bench workload: 20 planar dword streams read, the first 13 of them written back
// in place (132 B per vehicle); trivial arithmetic.  Knobs: cache-policy bits on loads / stores (nt, sc1), which
// workgroup streams which range (dispatch order vs one contiguous range per XCD), waves that loop over many chunks
// (the resident grid's shape), tile layouts (all components of T vehicles contiguous), 16 B per lane inside tiles, out
// of place, two halves on two streams; plus the box's own ceilings: float4 copy / read / write of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/hbm_probe.bin && tools/hbm_probe.bin [log2 n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>

#define NR 20
#define NW 13
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(float *base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}

// chunk (64 vehicles) a workgroup takes: dispatch order, or XCD x (= blockIdx % 8: workgroups go round-robin over the
// XCDs) takes the x-th contiguous eighth of the chunks
template <bool XCD>
__device__ __forceinline__ unsigned chunk_of(unsigned b, unsigned nb) {
  if (!XCD) return b;
  const unsigned per = nb >> 3;          // nb is a multiple of 8 here
  return (b & 7u) * per + (b >> 3);
}

// planar, one-wave workgroups, one chunk per workgroup; LA / SA = aux bits of the loads / stores
template <int LA, int SA, bool XCD, bool OUTOF>
__global__ void __launch_bounds__(64) planar_k(float *in, float *out, long S, long n) {
  const unsigned c = chunk_of<XCD>(blockIdx.x, gridDim.x);
  const unsigned i = c * 64 + threadIdx.x;
  if (i >= n) return;
  __amdgpu_buffer_rsrc_t ri = rsrc(in, NR * S * 4), ro = rsrc(OUTOF ? out : in, NR * S * 4);
  const unsigned off = i * 4u;
  const int S4 = (int)S * 4;
  float v[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ri, off, k * S4, LA));
  float acc = 0;
#pragma unroll
  for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NW; k++) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k] * 1.0001f + acc), ro, off, k * S4, SA);
}

// the resident grid's shape: `gridDim.x` one-wave workgroups, each loops over chunks w, w + W, ... (INTERLEAVED) or over
// its own contiguous run of chunks
template <int LA, int SA, bool CONTIG>
__global__ void __launch_bounds__(64) looping_k(float *in, long S, long n) {
  const unsigned W = gridDim.x, w = blockIdx.x;
  const unsigned chunks = (unsigned)((n + 63) / 64);
  __amdgpu_buffer_rsrc_t r = rsrc(in, NR * S * 4);
  const int S4 = (int)S * 4;
  const unsigned per = (chunks + W - 1) / W;
  for (unsigned t = 0; t < per; t++) {
    const unsigned c = CONTIG ? w * per + t : w + t * W;
    if (c >= chunks) break;
    const unsigned off = (c * 64 + threadIdx.x) * 4u;
    float v[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, k * S4, LA));
    float acc = 0;
#pragma unroll
    for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
    for (int k = 0; k < NW; k++) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k] * 1.0001f + acc), r, off, k * S4, SA);
  }
}

// tiles: all NR components of T vehicles contiguous (component k of vehicle i at tile(i) * NR * T + k * T + i % T);
// one wave per 64 vehicles, dword per lane
template <int T, int LA, int SA, bool XCD>
__global__ void __launch_bounds__(64) tiled_k(float *in, long n) {
  const unsigned c = chunk_of<XCD>(blockIdx.x, gridDim.x);
  const unsigned i = c * 64 + threadIdx.x;
  if (i >= n) return;
  __amdgpu_buffer_rsrc_t r = rsrc(in, (long)NR * n * 4);
  const unsigned off = ((i / T) * (unsigned)(NR * T) + (i % T)) * 4u;
  float v[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, k * T * 4, LA));
  float acc = 0;
#pragma unroll
  for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NW; k++) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[k] * 1.0001f + acc), r, off, k * T * 4, SA);
}

// tiles of 256 vehicles, a wave takes a whole tile: 4 vehicles per lane, 16 B per lane and access (1 KiB per wave-instruction)
template <int LA, int SA>
__global__ void __launch_bounds__(64) tiled256x4_k(float *in, long n) {
  const unsigned tile = blockIdx.x;
  if ((long)tile * 256 >= n) return;
  __amdgpu_buffer_rsrc_t r = rsrc(in, (long)NR * n * 4);
  const unsigned off = (tile * (unsigned)(NR * 256) + threadIdx.x * 4u) * 4u;
  typedef float __attribute__((ext_vector_type(4))) f4;
  typedef unsigned __attribute__((ext_vector_type(4))) u4;
  f4 v[NR];
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, k * 256 * 4, LA));
  f4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int k = NW; k < NR; k++) acc += v[k];
#pragma unroll
  for (int k = 0; k < NW; k++) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v[k] * 1.0001f + acc), r, off, k * 256 * 4, SA);
}

// ceilings of the box for the same bytes: 16 B per lane, 256-thread workgroups, grid-stride
__global__ void __launch_bounds__(256) copy_k(const float4 *__restrict__ a, float4 *__restrict__ b, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) rmw_k(float4 *a, long n4) {      // read-modify-write in place, one stream
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) { float4 t = a[i]; t.x += 1.0f; a[i] = t; }
}
__global__ void __launch_bounds__(256) read_k(const float4 *__restrict__ a, float *sink, long n4) {
  float s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) { float4 t = a[i]; s += t.x + t.y + t.z + t.w; }
  if (s == 12345.678f) *sink = s;
}
__global__ void __launch_bounds__(256) write_k(float4 *a, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) a[i] = make_float4(1, 2, 3, 4);
}

int main(int argc, char **argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 22;
  const long n = 1L << lg;
  const long S = n + 256;                 // the engine's odd multiple of 256
  float *a, *b;
  CK(hipMalloc(&a, sizeof(float) * NR * S));
  CK(hipMalloc(&b, sizeof(float) * NR * S));
  CK(hipMemset(a, 0, sizeof(float) * NR * S));
  CK(hipMemset(b, 0, sizeof(float) * NR * S));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1, ej;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  const double bytes = (double)n * 4 * (NR + NW);
  const int launches = 20;
  const unsigned nb = (unsigned)(n / 64);
  printf("n = 2^%d vehicles, %d read + %d write planar dword streams = %.0f MB per launch\n", lg, NR, NW, bytes / 1e6);

  auto run = [&](const char *name, std::function<void()> f, double b_per_launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s0));
      for (int it = 0; it < launches; it++) f();
      CK(hipEventRecord(e1, s0));
      CK(hipEventSynchronize(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double us = best * 1e3 / launches;
    printf("%-58s %8.2f us/launch  %6.0f GB/s\n", name, us, b_per_launch / (us * 1e-6) / 1e9);
    fflush(stdout);
  };
#define L1(K, ...) [&]() { hipLaunchKernelGGL(K, dim3(nb), dim3(64), 0, s0, __VA_ARGS__); }
  const int NT = 2, SC1 = 16, SC0 = 1;
  run("planar, dispatch order (the engine's launch)", L1((planar_k<0, 0, false, false>), a, a, S, n), bytes);
  run("planar, loads nt", L1((planar_k<NT, 0, false, false>), a, a, S, n), bytes);
  run("planar, stores nt", L1((planar_k<0, NT, false, false>), a, a, S, n), bytes);
  run("planar, loads nt + stores nt", L1((planar_k<NT, NT, false, false>), a, a, S, n), bytes);
  run("planar, loads sc1", L1((planar_k<SC1, 0, false, false>), a, a, S, n), bytes);
  run("planar, stores sc1 (write-through, line dropped)", L1((planar_k<0, SC1, false, false>), a, a, S, n), bytes);
  run("planar, stores sc0 sc1", L1((planar_k<0, SC0 | SC1, false, false>), a, a, S, n), bytes);
  run("planar, loads nt + stores sc1", L1((planar_k<NT, SC1, false, false>), a, a, S, n), bytes);
  run("planar, loads sc1 nt + stores sc1 nt", L1((planar_k<SC1 | NT, SC1 | NT, false, false>), a, a, S, n), bytes);
  run("planar, one contiguous eighth per XCD", L1((planar_k<0, 0, true, false>), a, a, S, n), bytes);
  run("planar, eighth per XCD, loads nt + stores nt", L1((planar_k<NT, NT, true, false>), a, a, S, n), bytes);
  run("planar, out of place", L1((planar_k<0, 0, false, true>), a, b, S, n), bytes);
  run("planar, out of place, nt + nt", L1((planar_k<NT, NT, false, true>), a, b, S, n), bytes);
  run("planar, out of place, eighth per XCD", L1((planar_k<0, 0, true, true>), a, b, S, n), bytes);
  // two halves on two streams, like afe_set_split_stepping
  {
    const long half = n / 2;
    auto two = [&](int la_sa) {
      CK(hipEventRecord(ej, s0)); CK(hipStreamWaitEvent(s1, ej, 0));
      for (int it = 0; it < launches; it++) {
        if (la_sa == 0) {
          hipLaunchKernelGGL((planar_k<0, 0, false, false>), dim3(nb / 2), dim3(64), 0, s0, a, a, S, half);
          hipLaunchKernelGGL((planar_k<0, 0, false, false>), dim3(nb / 2), dim3(64), 0, s1, a + half, a + half, S, half);
        } else {
          hipLaunchKernelGGL((planar_k<2, 2, false, false>), dim3(nb / 2), dim3(64), 0, s0, a, a, S, half);
          hipLaunchKernelGGL((planar_k<2, 2, false, false>), dim3(nb / 2), dim3(64), 0, s1, a + half, a + half, S, half);
        }
      }
      CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0));
    };
    float best[2] = {1e30f, 1e30f};
    for (int v = 0; v < 2; v++)
      for (int rep = 0; rep < 4; rep++) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s0));
        two(v);
        CK(hipEventRecord(e1, s0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best[v]) best[v] = ms;
      }
    for (int v = 0; v < 2; v++) {
      const double us = best[v] * 1e3 / launches;
      printf("%-58s %8.2f us/step    %6.0f GB/s\n", v ? "planar, two halves on two streams, nt + nt" : "planar, two halves on two streams", us, bytes / (us * 1e-6) / 1e9);
    }
  }
  // the resident grid's shape
  for (int waves : {2048, 4096, 6143, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "looping waves (%d), chunks interleaved", waves);
    run(nm, [&]() { hipLaunchKernelGGL((looping_k<0, 0, false>), dim3(waves), dim3(64), 0, s0, a, S, n); }, bytes);
    snprintf(nm, sizeof nm, "looping waves (%d), chunks interleaved, nt + nt", waves);
    run(nm, [&]() { hipLaunchKernelGGL((looping_k<2, 2, false>), dim3(waves), dim3(64), 0, s0, a, S, n); }, bytes);
    snprintf(nm, sizeof nm, "looping waves (%d), contiguous run per wave", waves);
    run(nm, [&]() { hipLaunchKernelGGL((looping_k<0, 0, true>), dim3(waves), dim3(64), 0, s0, a, S, n); }, bytes);
  }
  // tiles
  run("tiles of 64 vehicles", L1((tiled_k<64, 0, 0, false>), a, n), bytes);
  run("tiles of 256 vehicles", L1((tiled_k<256, 0, 0, false>), a, n), bytes);
  run("tiles of 1024 vehicles", L1((tiled_k<1024, 0, 0, false>), a, n), bytes);
  run("tiles of 4096 vehicles", L1((tiled_k<4096, 0, 0, false>), a, n), bytes);
  run("tiles of 16384 vehicles", L1((tiled_k<16384, 0, 0, false>), a, n), bytes);
  run("tiles of 1024 vehicles, nt + nt", L1((tiled_k<1024, 2, 2, false>), a, n), bytes);
  run("tiles of 1024 vehicles, eighth per XCD", L1((tiled_k<1024, 0, 0, true>), a, n), bytes);
  run("tiles of 256 vehicles, 16 B per lane (wave = tile)", [&]() { hipLaunchKernelGGL((tiled256x4_k<0, 0>), dim3((unsigned)(n / 256)), dim3(64), 0, s0, a, n); }, bytes);
  run("tiles of 256 vehicles, 16 B per lane, nt + nt", [&]() { hipLaunchKernelGGL((tiled256x4_k<2, 2>), dim3((unsigned)(n / 256)), dim3(64), 0, s0, a, n); }, bytes);
  // the box's own ceilings over the same footprint (NR * n floats)
  const long n4 = (long)NR * n / 4;
  const double fb = (double)NR * n * 4;
  run("float4 copy a -> b (1 read + 1 write stream)", [&]() { hipLaunchKernelGGL(copy_k, dim3(256 * 16), dim3(256), 0, s0, (const float4 *)a, (float4 *)b, n4); }, 2 * fb);
  run("float4 read-modify-write in place", [&]() { hipLaunchKernelGGL(rmw_k, dim3(256 * 16), dim3(256), 0, s0, (float4 *)a, n4); }, 2 * fb);
  run("float4 read", [&]() { hipLaunchKernelGGL(read_k, dim3(256 * 16), dim3(256), 0, s0, (const float4 *)a, b, n4); }, fb);
  run("float4 write", [&]() { hipLaunchKernelGGL(write_k, dim3(256 * 16), dim3(256), 0, s0, (float4 *)b, n4); }, fb);
  return 0;
}
