// pk_probe.hip -- round-5 review item 7, as a bounded experiment: what two vehicles per lane with gfx950's packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) could buy the rigid-body part of the vehicle step at the shard sizes that
// are bound by vector issue (131 072 and 262 144 vehicles).
//
// The probe is the step's ARITHMETIC SHAPE, not the engine: the rigid-body update of afe_kernels.hip run_vehicle (four
// lag-free motors, torque sums, R(att), angular momentum and acceleration, body drag, acceleration, integration, series
// quaternion increment, renormalisation) written once on a type T and instantiated for T = float (one vehicle per lane:
// the engine's layout) and T = float2 (two vehicles per lane: every + - * fma is one packed instruction for both), with
// the engine's memory pattern per step (13 state words in and out, 4 commands, 3 force components, planar dword streams,
// one-wave workgroups, state through memory every step, K steps per launch like a resident grid's worker).  No noise, no
// clamps, no ground contact: what is left out is what cannot be packed (integer generator, selects) -- the probe gives
// the UPPER bound of the gain on the packable part.
//   hipcc -O3 --offload-arch=gfx950 tools/pk_probe.hip -o tools/pk_probe.bin && tools/pk_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ T fm(T a, T b, T c);
template <> __device__ __forceinline__ float fm<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ f2 fm<f2>(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T bc(float x);
template <> __device__ __forceinline__ float bc<float>(float x) { return x; }
template <> __device__ __forceinline__ f2 bc<f2>(float x) { return (f2){x, x}; }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ f2 rsq(f2 x) { return (f2){__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
__device__ __forceinline__ float absf(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ f2 absf(f2 x) { return __builtin_elementwise_abs(x); }

struct Params { float kf, ktau, inv_mass, dt, hdt, mpx[4], mpy[4], I[9], Iinv[9], drag[3]; };

template <typename T>
__device__ __forceinline__ void step(const Params &P, T &px, T &py, T &pz, T &vx, T &vy, T &vz, T &q0, T &q1, T &q2, T &q3, T &wx, T &wy, T &wz,
                                     const T cmd[4], T fex, T fey, T fez) {
#pragma clang fp contract(off)
  T Fz = bc<T>(0), Tx = bc<T>(0), Ty = bc<T>(0), Tz = bc<T>(0);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const T w = cmd[m];
    const T thrust = bc<T>(P.kf) * w * absf(w);
    const T aero = bc<T>(-P.ktau) * w * absf(w);
    Fz = Fz + thrust;
    Tx = Tx + bc<T>(P.mpy[m]) * thrust;
    Ty = Ty - bc<T>(P.mpx[m]) * thrust;
    Tz = Tz + aero * bc<T>((m & 1) ? -1.0f : 1.0f);
  }
  T R[9];
  {
    const T r0 = q0 * q0, r1 = q1 * q1, r2 = q2 * q2, r3 = q3 * q3;
    const T a = bc<T>(2) * q0, b = bc<T>(2) * q1, c = bc<T>(2) * q2;
    R[0] = r0 + r1 - r2 - r3; R[1] = fm(b, q2, -(a * q3)); R[2] = fm(b, q3, a * q2);
    R[3] = fm(b, q2, a * q3); R[4] = r0 - r1 + r2 - r3; R[5] = fm(c, q3, -(a * q1));
    R[6] = fm(b, q3, -(a * q2)); R[7] = fm(c, q3, a * q1); R[8] = r0 - r1 - r2 + r3;
  }
  const T Lx = fm(bc<T>(P.I[2]), wz, fm(bc<T>(P.I[1]), wy, bc<T>(P.I[0]) * wx));
  const T Ly = fm(bc<T>(P.I[5]), wz, fm(bc<T>(P.I[4]), wy, bc<T>(P.I[3]) * wx));
  const T Lz = fm(bc<T>(P.I[8]), wz, fm(bc<T>(P.I[7]), wy, bc<T>(P.I[6]) * wx));
  const T cx = fm(wy, Lz, -(wz * Ly)), cy = fm(wz, Lx, -(wx * Lz)), cz = fm(wx, Ly, -(wy * Lx));
  const T ux = Tx - cx, uy = Ty - cy, uz = Tz - cz;
  const T aax = fm(bc<T>(P.Iinv[2]), uz, fm(bc<T>(P.Iinv[1]), uy, bc<T>(P.Iinv[0]) * ux));
  const T aay = fm(bc<T>(P.Iinv[5]), uz, fm(bc<T>(P.Iinv[4]), uy, bc<T>(P.Iinv[3]) * ux));
  const T aaz = fm(bc<T>(P.Iinv[8]), uz, fm(bc<T>(P.Iinv[7]), uy, bc<T>(P.Iinv[6]) * ux));
  const T vbx = fm(R[6], vz, fm(R[3], vy, R[0] * vx)), vby = fm(R[7], vz, fm(R[4], vy, R[1] * vx)), vbz = fm(R[8], vz, fm(R[5], vy, R[2] * vx));
  const T Fbx = bc<T>(P.drag[0]) * (-vbx), Fby = bc<T>(P.drag[1]) * (-vby), Fbz = fm(bc<T>(P.drag[2]), -vbz, Fz);
  const T im = bc<T>(P.inv_mass);
  const T ax = (fm(R[2], Fbz, fm(R[1], Fby, R[0] * Fbx)) + fex) * im;
  const T ay = (fm(R[5], Fbz, fm(R[4], Fby, R[3] * Fbx)) + fey) * im;
  const T az = bc<T>(-9.81f) + (fm(R[8], Fbz, fm(R[7], Fby, R[6] * Fbx)) + fez) * im;
  const T dt = bc<T>(P.dt), hdt = bc<T>(P.hdt);
  const T npx = fm(dt, fm(hdt, ax, vx), px), npy = fm(dt, fm(hdt, ay, vy), py), npz = fm(dt, fm(hdt, az, vz), pz);
  const T nvx = fm(dt, ax, vx), nvy = fm(dt, ay, vy), nvz = fm(dt, az, vz);
  const T rx = dt * wx, ry = dt * wy, rz = dt * wz;
  const T t = fm(rz, rz, fm(ry, ry, rx * rx));
  const T h2 = bc<T>(0.25f) * t;
  const T cs = fm(h2, fm(h2, fm(h2, fm(h2, bc<T>(2.4801587e-5f), bc<T>(-1.3888889e-3f)), bc<T>(4.1666667e-2f)), bc<T>(-0.5f)), bc<T>(1.0f));
  const T sc = bc<T>(0.5f) * fm(h2, fm(h2, fm(h2, fm(h2, bc<T>(2.7557319e-6f), bc<T>(-1.9841270e-4f)), bc<T>(8.3333333e-3f)), bc<T>(-1.6666667e-1f)), bc<T>(1.0f));
  const T d0 = cs, d1 = sc * rx, d2 = sc * ry, d3 = sc * rz;
  T n0 = fm(-d3, q3, fm(-d2, q2, fm(-d1, q1, d0 * q0)));
  T n1 = fm(-d2, q3, fm(d3, q2, fm(d0, q1, d1 * q0)));
  T n2 = fm(d1, q3, fm(d0, q2, fm(-d3, q1, d2 * q0)));
  T n3 = fm(d0, q3, fm(-d1, q2, fm(d2, q1, d3 * q0)));
  const T inv = rsq(fm(n3, n3, fm(n2, n2, fm(n1, n1, n0 * n0))));
  n0 = n0 * inv; n1 = n1 * inv; n2 = n2 * inv; n3 = n3 * inv;
  px = npx; py = npy; pz = npz; vx = nvx; vy = nvy; vz = nvz; q0 = n0; q1 = n1; q2 = n2; q3 = n3;
  wx = fm(dt, aax, wx); wy = fm(dt, aay, wy); wz = fm(dt, aaz, wz);
}

// one vehicle per lane: wave w steps chunks w, w + waves, ... (64 vehicles each) through K steps, state through memory
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) probe_scalar(Params P, float *state, const float *in, int64_t stride, int chunks, int K) {
  for (int s = 0; s < K; s++)
    for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
      const int64_t i = (int64_t)c * 64 + threadIdx.x;
      float v[13], cmd[4], f[3];
#pragma unroll
      for (int k = 0; k < 13; k++) v[k] = state[k * stride + i];
#pragma unroll
      for (int k = 0; k < 4; k++) cmd[k] = in[k * stride + i];
#pragma unroll
      for (int k = 0; k < 3; k++) f[k] = in[(4 + k) * stride + i];
      step<float>(P, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], cmd, f[0], f[1], f[2]);
#pragma unroll
      for (int k = 0; k < 13; k++) state[k * stride + i] = v[k];
    }
}

// two vehicles per lane: chunk pair (2c, 2c + 1) in the two halves of every register pair
template <int WAVES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) probe_packed(Params P, float *state, const float *in, int64_t stride, int pairs, int K) {
  for (int s = 0; s < K; s++)
    for (int c = blockIdx.x; c < pairs; c += gridDim.x) {
      const int64_t i = (int64_t)c * 128 + threadIdx.x;
      f2 v[13], cmd[4], f[3];
#pragma unroll
      for (int k = 0; k < 13; k++) v[k] = (f2){state[k * stride + i], state[k * stride + i + 64]};
#pragma unroll
      for (int k = 0; k < 4; k++) cmd[k] = (f2){in[k * stride + i], in[k * stride + i + 64]};
#pragma unroll
      for (int k = 0; k < 3; k++) f[k] = (f2){in[(4 + k) * stride + i], in[(4 + k) * stride + i + 64]};
      step<f2>(P, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[12], cmd, f[0], f[1], f[2]);
#pragma unroll
      for (int k = 0; k < 13; k++) { state[k * stride + i] = v[k].x; state[k * stride + i + 64] = v[k].y; }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
  const int K = 400;
  Params P = {};
  P.kf = 1.07e-8f; P.ktau = 6.4e-11f; P.inv_mass = 1.0f / 0.032f; P.dt = 1e-3f; P.hdt = 5e-4f;
  const float arm = 0.033f;
  const float sx[4] = {1, -1, -1, 1}, sy[4] = {-1, -1, 1, 1};
  for (int m = 0; m < 4; m++) { P.mpx[m] = arm * sx[m]; P.mpy[m] = arm * sy[m]; }
  for (int k = 0; k < 9; k++) { P.I[k] = 0; P.Iinv[k] = 0; }
  P.I[0] = P.I[4] = 1.6e-5f; P.I[8] = 2.9e-5f; P.Iinv[0] = P.Iinv[4] = 1 / 1.6e-5f; P.Iinv[8] = 1 / 2.9e-5f;
  P.drag[0] = P.drag[1] = P.drag[2] = 0.01f;
  for (int64_t n : {65536, 131072, 262144, 524288}) {
    const int64_t stride = n + 256 * 3;
    std::vector<float> st((size_t)13 * stride, 0.0f), in((size_t)7 * stride, 0.0f);
    for (int64_t i = 0; i < n; i++) {
      st[2 * stride + i] = 3.5f; st[6 * stride + i] = 1.0f;
      for (int k = 0; k < 4; k++) in[k * stride + i] = 2708.0f + (float)(i % 7);
      in[4 * stride + i] = 0.01f * (float)(i % 5);
    }
    float *d_st, *d_in;
    CK(hipMalloc(&d_st, st.size() * 4)); CK(hipMalloc(&d_in, in.size() * 4));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int chunks = (int)(n / 64);
    auto run = [&](const char *name, int waves, auto launch) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipMemcpy(d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("%8ld vehicles  %-34s %5d waves: %7.3f us per step\n", (long)n, name, waves, best * 1e3f / K);
    };
    const int cap6 = 6 * 4 * 256, cap3 = 3 * 4 * 256, cap4 = 4 * 4 * 256;
    int w = chunks < cap6 ? chunks : cap6;
    run("one vehicle per lane (6 waves/SIMD)", w, [&] { hipLaunchKernelGGL(probe_scalar, dim3(w), dim3(64), 0, 0, P, d_st, d_in, stride, chunks, K); });
    int w3 = chunks / 2 < cap3 ? chunks / 2 : cap3;
    run("two per lane, packed (3 waves/SIMD)", w3, [&] { hipLaunchKernelGGL(probe_packed<3>, dim3(w3), dim3(64), 0, 0, P, d_st, d_in, stride, chunks / 2, K); });
    int w4 = chunks / 2 < cap4 ? chunks / 2 : cap4;
    run("two per lane, packed (4 waves/SIMD)", w4, [&] { hipLaunchKernelGGL(probe_packed<4>, dim3(w4), dim3(64), 0, 0, P, d_st, d_in, stride, chunks / 2, K); });
    const int cap5 = 5 * 4 * 256;
    int w5 = chunks / 2 < cap5 ? chunks / 2 : cap5;
    run("two per lane, packed (5 waves/SIMD)", w5, [&] { hipLaunchKernelGGL(probe_packed<5>, dim3(w5), dim3(64), 0, 0, P, d_st, d_in, stride, chunks / 2, K); });
    CK(hipFree(d_st)); CK(hipFree(d_in));
  }
  return 0;
}
