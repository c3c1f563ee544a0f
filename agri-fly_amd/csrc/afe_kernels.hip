// afe_kernels.hip -- gfx950 (MI355X) kernels of the batched quadrotor engine.
//
// One lane = one vehicle.  A wave64 load of one state component is one
// contiguous 256-B segment of a planar SoA slab, so every global access is
// fully coalesced; the ~24 component loads of a vehicle are independent and
// are all issued before the first use, which is what keeps enough bytes in
// flight to stream at HBM rate.  Per-type constants are staged into LDS once
// per workgroup.  There is no dense contraction on this path (largest matrix
// is 3x3), hence no MFMA; the bound is HBM bandwidth.
//
// The arithmetic follows the reference statement by statement, in the same
// order (citations: Components/Components/Simulation/Quadcopter_T.cpp,
// Motor.cpp, Common/Common/Math/Rotation.hpp, Vec3.hpp of agri-fly), so the
// fp64 instantiation tracks the CPU oracle to rounding error and the fp32
// instantiation differs from it only by fp32 rounding.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afe_device.h"

namespace afe {

// ---------------------------------------------------------------------------
// scalar math, float / double
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ void m_sincos(float x, float *s, float *c) { sincosf(x, s, c); }
__device__ __forceinline__ void m_sincos(double x, double *s, double *c) { sincos(x, s, c); }

// Rotation<Real>::GetRotationMatrix, Rotation.hpp:196-220 (literal)
template <typename R>
__device__ __forceinline__ void rot_matrix(R v0, R v1, R v2, R v3, R M[9]) {
  const R r0 = v0 * v0, r1 = v1 * v1, r2 = v2 * v2, r3 = v3 * v3;
  M[0] = r0 + r1 - r2 - r3;
  M[1] = 2 * v1 * v2 - 2 * v0 * v3;
  M[2] = 2 * v1 * v3 + 2 * v0 * v2;
  M[3] = 2 * v1 * v2 + 2 * v0 * v3;
  M[4] = r0 - r1 + r2 - r3;
  M[5] = 2 * v2 * v3 - 2 * v0 * v1;
  M[6] = 2 * v1 * v3 - 2 * v0 * v2;
  M[7] = 2 * v2 * v3 + 2 * v0 * v1;
  M[8] = r0 - r1 - r2 + r3;
}

// Matrix<Real,3,3> * Vec3<Real>, Vec3.hpp:201-210 (accumulates from 0)
template <typename R, typename M>
__device__ __forceinline__ void mat_vec(const M *A, R x, R y, R z, R &ox, R &oy, R &oz) {
  ox = ((R(0) + R(A[0]) * x) + R(A[1]) * y) + R(A[2]) * z;
  oy = ((R(0) + R(A[3]) * x) + R(A[4]) * y) + R(A[5]) * z;
  oz = ((R(0) + R(A[6]) * x) + R(A[7]) * y) + R(A[8]) * z;
}

// ---------------------------------------------------------------------------
// IMU noise: std::minstd_rand0 + libstdc++ std::normal_distribution<double>
// (reference Quadcopter_T.hpp:122-123; bits/random.tcc), always in double.
__device__ __forceinline__ uint32_t minstd_next(uint32_t &s) {
  // x <- 16807 x mod (2^31 - 1); Mersenne reduction: hi*2^31 + lo == hi + lo
  const uint64_t p = (uint64_t)s * 16807u;
  uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
  if (r >= 2147483647u) r -= 2147483647u;
  s = r;
  return r;
}

__device__ __forceinline__ double canonical53(uint32_t &s) {
#pragma clang fp contract(off)
  // generate_canonical<double,53>: two engine calls, R = 2147483646
  double sum = (double)(minstd_next(s) - 1u);
  sum = sum + (double)(minstd_next(s) - 1u) * 2147483646.0;
  double ret = sum / 4611686009837453312.0;  // (double)(R*R as long double)
  if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;  // nextafter(1, 0)
  return ret;
}

// one Marsaglia polar pair; `first` is what the first operator() call returns
__device__ __forceinline__ void normal_pair(uint32_t &s, double &first, double &second) {
#pragma clang fp contract(off)
  double x, y, r2;
  do {
    x = 2.0 * canonical53(s) - 1.0;
    y = 2.0 * canonical53(s) - 1.0;
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0.0);
  const double mult = sqrt(-2 * log(r2) / r2);
  first = y * mult;
  second = x * mult;
}

// ---------------------------------------------------------------------------
// The vehicle step.
template <typename R, bool FEXT, bool TEXT, bool NOISE, bool RENORM>
__global__ void __launch_bounds__(256)
afe_step_kernel(const StepView<R> v) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  {
    // stage the type table: n_types * sizeof(DevParams) bytes as dwords
    const uint32_t *src = reinterpret_cast<const uint32_t *>(v.table);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds_raw);
    const int nwords = v.n_types * (int)(sizeof(DevParams<R>) / 4);
    for (int k = threadIdx.x; k < nwords; k += 256) dst[k] = src[k];
  }
  __syncthreads();

  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= v.n) return;
  const int64_t S = v.stride;

  // ---- issue every load up front (independent, coalesced) ----
  R px = v.pos[i], py = v.pos[S + i], pz = v.pos[2 * S + i];
  R vx = v.vel[i], vy = v.vel[S + i], vz = v.vel[2 * S + i];
  R q0 = v.att[i], q1 = v.att[S + i], q2 = v.att[2 * S + i], q3 = v.att[3 * S + i];
  R wx = v.ang_vel[i], wy = v.ang_vel[S + i], wz = v.ang_vel[2 * S + i];
  R ms[4] = {v.motor[i], v.motor[S + i], v.motor[2 * S + i], v.motor[3 * S + i]};
  const float cmd_f[4] = {v.cmd[i], v.cmd[S + i], v.cmd[2 * S + i], v.cmd[3 * S + i]};
  R fex = 0, fey = 0, fez = 0, tex = 0, tey = 0, tez = 0;
  if (FEXT) { fex = v.ext_force[i]; fey = v.ext_force[S + i]; fez = v.ext_force[2 * S + i]; }
  if (TEXT) { tex = v.ext_torque[i]; tey = v.ext_torque[S + i]; tez = v.ext_torque[2 * S + i]; }
  uint32_t rng = 0;
  if (NOISE && v.tick_mask) rng = v.rng[i];
  const DevParams<R> &P = reinterpret_cast<const DevParams<R> *>(lds_raw)[v.type[i]];

  const R dt = v.dt;
  float gx = 0, gy = 0, gz = 0, ax_m = 0, ay_m = 0, az_m = 0;
  bool have_imu = false;

  // Motor.cpp:48-50: negative commands clamp to zero (the command is a float,
  // Quadcopter_T.hpp:100, widened at Quadcopter_T.cpp:98)
  R cmd[4];
#pragma unroll
  for (int m = 0; m < 4; m++) { cmd[m] = (R)cmd_f[m]; if (cmd[m] < 0) cmd[m] = 0; }

  for (int step = 0; step < v.n_steps; step++) {
    // ---- 4 motors: Motor::Run, Motor.cpp:39-84 ----
    R Fz = 0;                       // totalForce_b (thrust axes are all +z)
    R Tx = 0, Ty = 0, Tz = 0;       // totalTorque_b
    R Lm[4];                        // rotor angular momenta (about z)
    const R c = P.c_lag;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const R spin = (R)AFE_MOTOR_SPIN(m);
      const R old = ms[m];
      R w = c * old + (1 - c) * cmd[m];                      // :60
      if (w > P.wmax) w = P.wmax; else if (w < P.wmin) w = P.wmin;  // :62-66
      ms[m] = w;
      const R thrust = P.kf * w * m_abs(w);                  // :70 (along +z)
      const R aero = -P.ktau * w * m_abs(w);                 // :73 (along spin*z)
      const R ang_acc = (w - old) / dt;                      // :78
      // torque = aero*axis + p x (0,0,thrust) - ang_acc*J*axis   :71-79
      const R tz_m = (aero * spin) - (ang_acc * P.Jm) * spin;
      Fz = Fz + thrust;                                      // Quadcopter_T.cpp:102
      Tx = Tx + (P.mpy[m] * thrust);                         // Vec3.hpp:106-109
      Ty = Ty + (-(P.mpx[m] * thrust));                      // z*rx - x*rz, rx = 0
      Tz = Tz + tz_m;
      Lm[m] = (w * P.Jm) * spin;                             // Motor.cpp:68
    }

    R Rm[9];
    rot_matrix<R>(q0, q1, q2, q3, Rm);   // R(att); R(att.Inverse()) == Rm^T bitwise

    if (TEXT) {                          // Quadcopter_T.cpp:106
      Tx = Tx + (Rm[0] * tex + Rm[3] * tey + Rm[6] * tez);
      Ty = Ty + (Rm[1] * tex + Rm[4] * tey + Rm[7] * tez);
      Tz = Tz + (Rm[2] * tex + Rm[5] * tey + Rm[8] * tez);
    }

    // angular momentum and acceleration, Quadcopter_T.cpp:113-120
    R Lx, Ly, Lzz;
    mat_vec<R>(P.I, wx, wy, wz, Lx, Ly, Lzz);
    Lzz = (((Lzz + Lm[0]) + Lm[1]) + Lm[2]) + Lm[3];
    const R cx = wy * Lzz - wz * Ly;     // _angVel.Cross(angMomentum)
    const R cy = wz * Lx - wx * Lzz;
    const R cz = wx * Ly - wy * Lx;
    R aax, aay, aaz;
    mat_vec<R>(P.Iinv, Tx - cx, Ty - cy, Tz - cz, aax, aay, aaz);

    // body drag, Quadcopter_T.cpp:123-128
    const R vbx = Rm[0] * vx + Rm[3] * vy + Rm[6] * vz;
    const R vby = Rm[1] * vx + Rm[4] * vy + Rm[7] * vz;
    const R vbz = Rm[2] * vx + Rm[5] * vy + Rm[8] * vz;
    const R Fbx = P.drag[0] * (-vbx);
    const R Fby = P.drag[1] * (-vby);
    const R Fbz = Fz + P.drag[2] * (-vbz);

    // acceleration, Quadcopter_T.cpp:131-132
    R accx = R(0) + ((Rm[0] * Fbx + Rm[1] * Fby + Rm[2] * Fbz) + fex) / P.mass;
    R accy = R(0) + ((Rm[3] * Fbx + Rm[4] * Fby + Rm[5] * Fbz) + fey) / P.mass;
    R accz = R(-9.81) + ((Rm[6] * Fbx + Rm[7] * Fby + Rm[8] * Fbz) + fez) / P.mass;

    // integration, Quadcopter_T.cpp:140-143 (old vel / old angVel / old att)
    R npx = (px + dt * vx) + dt * (dt * (R(0.5) * accx));
    R npy = (py + dt * vy) + dt * (dt * (R(0.5) * accy));
    R npz = (pz + dt * vz) + dt * (dt * (R(0.5) * accz));
    R nvx = vx + dt * accx, nvy = vy + dt * accy, nvz = vz + dt * accz;
    // FromRotationVector(angVel*dt), Rotation.hpp:84-97
    const R rx = dt * wx, ry = dt * wy, rz = dt * wz;
    const R theta = m_sqrt(rx * rx + ry * ry + rz * rz);
    R d0 = 1, d1 = 0, d2 = 0, d3 = 0;
    if (theta >= R(4.84813681e-6)) {
      R sn, cs;
      m_sincos(theta * R(0.5), &sn, &cs);
      d0 = cs;
      d1 = sn * (rx / theta);
      d2 = sn * (ry / theta);
      d3 = sn * (rz / theta);
    }
    // att * dq, Rotation.hpp:124-131 (this = att, r1 = dq)
    R n0 = d0 * q0 - d1 * q1 - d2 * q2 - d3 * q3;
    R n1 = d1 * q0 + d0 * q1 + d3 * q2 - d2 * q3;
    R n2 = d2 * q0 - d3 * q1 + d0 * q2 + d1 * q3;
    R n3 = d3 * q0 + d2 * q1 - d1 * q2 + d0 * q3;
    R nwx = wx + dt * aax, nwy = wy + dt * aay, nwz = wz + dt * aaz;

    if (RENORM) {
      // fp32 storage only: the reference keeps |q| = 1 to 5e-14 over 1e4 steps
      // without ever normalising; fp32 needs this to stay inside tolerance.
      const R inv = R(1) / m_sqrt(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
      n0 *= inv; n1 *= inv; n2 *= inv; n3 *= inv;
    }

    // ground contact, Quadcopter_T.cpp:146-151
    if ((npz <= 0) && (nvz < 0)) {
      npz = 0; nvz = 0; accz = 0;
      nwx = 0; nwy = 0; nwz = 0;
    }
    px = npx; py = npy; pz = npz;
    vx = nvx; vy = nvy; vz = nvz;
    q0 = n0; q1 = n1; q2 = n2; q3 = n3;
    wx = nwx; wy = nwy; wz = nwz;

    // ---- onboard-logic gate fired on this sub-step: IMU synthesis ----
    if ((v.tick_mask >> step) & 1ull) {                      // Quadcopter_T.cpp:159
      float ng[3] = {0, 0, 0}, na[3] = {0, 0, 0};
      if (NOISE) {
        // g++ evaluates the ctor arguments right to left (Quadcopter_T.cpp:
        // 167-169,176-178): z <- draw 1, y <- 2, x <- 3
        double d[6];
        normal_pair(rng, d[0], d[1]);
        normal_pair(rng, d[2], d[3]);
        normal_pair(rng, d[4], d[5]);
        ng[0] = v.sigma_gyro * (float)d[2];
        ng[1] = v.sigma_gyro * (float)d[1];
        ng[2] = v.sigma_gyro * (float)d[0];
        na[0] = v.sigma_acc * (float)d[5];
        na[1] = v.sigma_acc * (float)d[4];
        na[2] = v.sigma_acc * (float)d[3];
      }
      float tx_, ty_, tz_;
      mat_vec<float>(P.Rimu, (float)wx, (float)wy, (float)wz, tx_, ty_, tz_);  // :165-166
      gx = tx_ + ng[0]; gy = ty_ + ng[1]; gz = tz_ + ng[2];                    // :167-170
      // _att.Inverse() * (acc + (0,0,9.81)) with the NEW attitude, :174
      R Rn[9];
      rot_matrix<R>(q0, q1, q2, q3, Rn);
      const R sx = accx + R(0), sy = accy + R(0), sz = accz + R(9.81);
      const R bx = Rn[0] * sx + Rn[3] * sy + Rn[6] * sz;
      const R by = Rn[1] * sx + Rn[4] * sy + Rn[7] * sz;
      const R bz = Rn[2] * sx + Rn[5] * sy + Rn[8] * sz;
      mat_vec<float>(P.Rimu, (float)bx, (float)by, (float)bz, tx_, ty_, tz_);  // :175
      ax_m = tx_ + na[0]; ay_m = ty_ + na[1]; az_m = tz_ + na[2];              // :176-179
      have_imu = true;
    }
  }

  // ---- write back (in place: same lines this lane just read) ----
  v.pos[i] = px; v.pos[S + i] = py; v.pos[2 * S + i] = pz;
  v.vel[i] = vx; v.vel[S + i] = vy; v.vel[2 * S + i] = vz;
  v.att[i] = q0; v.att[S + i] = q1; v.att[2 * S + i] = q2; v.att[3 * S + i] = q3;
  v.ang_vel[i] = wx; v.ang_vel[S + i] = wy; v.ang_vel[2 * S + i] = wz;
  v.motor[i] = ms[0]; v.motor[S + i] = ms[1]; v.motor[2 * S + i] = ms[2]; v.motor[3 * S + i] = ms[3];
  if (have_imu) {
    v.gyro[i] = gx; v.gyro[S + i] = gy; v.gyro[2 * S + i] = gz;
    v.acc[i] = ax_m; v.acc[S + i] = ay_m; v.acc[2 * S + i] = az_m;
    if (NOISE) v.rng[i] = rng;
  }
}

template <typename R>
static int launch_step(const StepView<R> &v, const LaunchFlags &f, hipStream_t st) {
  if (v.n <= 0) return 0;
  const unsigned grid = (unsigned)((v.n + 255) / 256);
  const size_t lds = (size_t)v.n_types * sizeof(DevParams<R>);
#define AFE_LAUNCH(FE, TE, NO, RE) \
  hipLaunchKernelGGL((afe_step_kernel<R, FE, TE, NO, RE>), dim3(grid), dim3(256), lds, st, v)
#define AFE_SEL_RE(FE, TE, NO) do { if (f.renorm) AFE_LAUNCH(FE, TE, NO, true); else AFE_LAUNCH(FE, TE, NO, false); } while (0)
#define AFE_SEL_NO(FE, TE) do { if (f.noise) AFE_SEL_RE(FE, TE, true); else AFE_SEL_RE(FE, TE, false); } while (0)
#define AFE_SEL_TE(FE) do { if (f.ext_torque) AFE_SEL_NO(FE, true); else AFE_SEL_NO(FE, false); } while (0)
  if (f.ext_force) AFE_SEL_TE(true); else AFE_SEL_TE(false);
#undef AFE_SEL_TE
#undef AFE_SEL_NO
#undef AFE_SEL_RE
#undef AFE_LAUNCH
  return (int)hipGetLastError();
}

int launch_step_f32(const StepView<float> &v, const LaunchFlags &f, void *stream) {
  return launch_step<float>(v, f, (hipStream_t)stream);
}
int launch_step_f64(const StepView<double> &v, const LaunchFlags &f, void *stream) {
  return launch_step<double>(v, f, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// RNG seeding (Quadcopter_T.cpp:27: default-constructed engine => seed 1)
__global__ void __launch_bounds__(256)
afe_seed_kernel(uint32_t *rng, int64_t n, int64_t first_global, int policy) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t s = 1u;
  if (policy == 1) {
    // linear_congruential_engine::seed(s): s mod m, and 0 -> 1
    s = (uint32_t)((uint64_t)(1 + first_global + i) % 2147483647ull);
    if (s == 0) s = 1u;
  }
  rng[i] = s;
}

int launch_seed_rng(uint32_t *rng, int64_t n, int64_t first_global, int policy, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(afe_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, rng, n, first_global, policy);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// shared-world query support
template <typename R>
__global__ void __launch_bounds__(256)
afe_pack_positions_kernel(const R *pos, int64_t stride, int64_t n, float *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = (float)pos[i];
  out[n + i] = (float)pos[stride + i];
  out[2 * n + i] = (float)pos[2 * stride + i];
}

int launch_pack_positions_f32(const float *pos, int64_t stride, int64_t n, float *out, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(afe_pack_positions_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pos, stride, n, out);
  return (int)hipGetLastError();
}
int launch_pack_positions_f64(const double *pos, int64_t stride, int64_t n, float *out, void *stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(afe_pack_positions_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, pos, stride, n, out);
  return (int)hipGetLastError();
}

// Brute-force nearest neighbour of each local vehicle among the gathered
// ensemble: candidates are streamed through LDS in 256-vehicle tiles so each
// global position is read once per workgroup, not once per lane.
__global__ void __launch_bounds__(256)
afe_nearest_kernel(const float *self_xyz, int64_t n_self, int64_t first_global,
                   const float *all_xyz, int64_t n_all, float *dist2, int32_t *index) {
  __shared__ float tx[256], ty[256], tz[256];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n_self;
  float x = 0, y = 0, z = 0;
  if (live) { x = self_xyz[i]; y = self_xyz[n_self + i]; z = self_xyz[2 * n_self + i]; }
  const int64_t me = first_global + i;
  float best = 3.4e38f;
  int32_t best_j = -1;
  for (int64_t base = 0; base < n_all; base += 256) {
    const int64_t j = base + threadIdx.x;
    if (j < n_all) { tx[threadIdx.x] = all_xyz[j]; ty[threadIdx.x] = all_xyz[n_all + j]; tz[threadIdx.x] = all_xyz[2 * n_all + j]; }
    __syncthreads();
    const int lim = (int)((n_all - base) < 256 ? (n_all - base) : 256);
    for (int k = 0; k < lim; k++) {
      const float dx = tx[k] - x, dy = ty[k] - y, dz = tz[k] - z;
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < best && (base + k) != me) { best = d; best_j = (int32_t)(base + k); }
    }
    __syncthreads();
  }
  if (live) { dist2[i] = best; index[i] = best_j; }
}

int launch_nearest_neighbour(const float *self_xyz, int64_t n_self, int64_t first_global,
                             const float *all_xyz, int64_t n_all, float *dist2,
                             int32_t *index, void *stream) {
  if (n_self <= 0) return 0;
  hipLaunchKernelGGL(afe_nearest_kernel, dim3((unsigned)((n_self + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, self_xyz, n_self, first_global, all_xyz, n_all, dist2, index);
  return (int)hipGetLastError();
}

}  // namespace afe
