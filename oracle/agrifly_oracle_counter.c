/* agrifly_oracle_counter.c -- see agrifly_oracle_counter.h.  TEST INFRASTRUCTURE ONLY. */
#include "agrifly_oracle_counter.h"

#include <math.h>
#include <stddef.h>

/* Philox4x32-10 (Salmon et al., SC'11; Random123 philox.h): ten rounds of
 *   (c0, c1, c2, c3) <- (hi(M1 c2) ^ c1 ^ k0, lo(M1 c2), hi(M0 c0) ^ c3 ^ k1, lo(M0 c0))
 * with the key bumped by the Weyl constants between rounds. */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void ora_counter_block(uint64_t seed, uint64_t index, unsigned stream, unsigned block, uint64_t ordinal, uint32_t out[4]) {
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t ctr[4] = {(uint32_t)index, (uint32_t)((index >> 32) & 0xffffu) | ((uint32_t)stream << 16) | ((uint32_t)block << 24),
                           (uint32_t)ordinal, (uint32_t)(ordinal >> 32)};
  ora_philox4x32_10(ctr, key, out);
}

void ora_counter_normals4(const uint32_t w[4], double z[4]) {
  const double two_pi = 6.283185307179586476925286766559;
  for (int k = 0; k < 2; k++) {
    const double u_r = ((double)(w[2 * k] >> 9) + 0.5) * (1.0 / 8388608.0);      /* 2^-23 */
    const double u_a = (double)(w[2 * k + 1] >> 8) * (1.0 / 16777216.0);          /* 2^-24 */
    const double r = sqrt(-2.0 * log(u_r));
    z[2 * k] = r * cos(two_pi * u_a);
    z[2 * k + 1] = r * sin(two_pi * u_a);
  }
}

void ora_imu_normals(uint64_t seed, uint64_t index, uint64_t tick, double z[6]) {
  uint32_t w[4];
  double a[4], b[4];
  ora_counter_block(seed, index, ORA_STREAM_IMU, 0, tick, w);
  ora_counter_normals4(w, a);
  ora_counter_block(seed, index, ORA_STREAM_IMU, 1, tick, w);
  ora_counter_normals4(w, b);
  z[0] = a[0]; z[1] = a[1]; z[2] = a[2]; z[3] = a[3]; z[4] = b[0]; z[5] = b[1];
}

void ora_gust_force(uint64_t seed, uint64_t index, uint64_t n_global, uint64_t epoch, double sigma_max, double force[3]) {
  uint32_t w[4];
  double z[4];
  ora_counter_block(seed, index, ORA_STREAM_GUST, 0, epoch, w);
  ora_counter_normals4(w, z);
  const double sigma = sigma_max * (double)index / (double)(n_global > 1 ? n_global - 1 : 1);
  for (int k = 0; k < 3; k++) force[k] = sigma * z[k];
}

/* Under the counter policy the six normals of a tick come from Philox and go into the step in the reference's DRAW
 * order (ora_quad_step_normals): gyro x y z = z0 z1 z2 are draws 3 2 1, accelerometer x y z = z3 z4 z5 draws 6 5 4.  The
 * libstdc++ stream is neither drawn from nor advanced (round-4 review: the earlier version ran the step with sigma = 0,
 * which still drew and discarded six libstdc++ normals per tick -- the most expensive item of a CPU step -- and made
 * the CPU baseline of this policy slower than the algorithm it stands for). */
void ora_step_batch_counter(int64_t n, int n_steps, const ora_params *table, const uint8_t *types, double *pos, double *vel,
                            double *att, double *ang_vel, double *motor_speed, uint32_t *rng, const float *motor_cmd,
                            double *ext_force, const double *ext_torque, uint64_t dt_us, const uint8_t *tick_per_step,
                            float *gyro, float *acc, int use_counter_noise, uint64_t seed, uint64_t first_global,
                            uint64_t tick_base, uint64_t gust_seed, uint64_t gust_period_us, uint64_t t0_us, uint64_t n_global, double sigma_max) {
  const double dt = (double)((double)dt_us * 1e-6);     /* Timer::GetSeconds<double>, Timer.hpp:36-38 */
  const int threads = ora_get_batch_threads();
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
  for (int64_t i = 0; i < n; i++) {
    const ora_params p = table[types ? types[i] : 0];
    ora_state s;
    for (int k = 0; k < 3; k++) { s.pos[k] = pos[k * n + i]; s.vel[k] = vel[k * n + i]; s.ang_vel[k] = ang_vel[k * n + i]; }
    for (int k = 0; k < 4; k++) { s.att[k] = att[k * n + i]; s.motor_speed[k] = motor_speed[k * n + i]; }
    s.rng = rng ? rng[i] : 1u;
    const float cmd[4] = {motor_cmd[0 * n + i], motor_cmd[1 * n + i], motor_cmd[2 * n + i], motor_cmd[3 * n + i]};
    double fe[3] = {0, 0, 0}, te[3] = {0, 0, 0};
    if (ext_force && !gust_period_us) for (int k = 0; k < 3; k++) fe[k] = ext_force[k * n + i];
    if (ext_torque) for (int k = 0; k < 3; k++) te[k] = ext_torque[k * n + i];
    uint64_t ticks = tick_base, gust_epoch = ~(uint64_t)0;
    for (int st = 0; st < n_steps; st++) {
      if (gust_period_us) {          /* piecewise constant: evaluated when a step starts in a new epoch, like the engine does */
        const uint64_t epoch = (t0_us + (uint64_t)st * dt_us) / gust_period_us;
        if (epoch != gust_epoch) { ora_gust_force(gust_seed, first_global + (uint64_t)i, n_global, epoch, sigma_max, fe); gust_epoch = epoch; }
      }
      const int tick = tick_per_step ? tick_per_step[st] : 0;
      float g[3], a[3];
      if (tick && use_counter_noise) {
        double z[6];
        ora_imu_normals(seed, first_global + (uint64_t)i, ticks, z);
        const double draws[6] = {z[2], z[1], z[0], z[5], z[4], z[3]};
        ora_quad_step_normals(&p, &s, cmd, fe, te, dt, tick, draws, g, a, 0);
      } else {
        ora_quad_step(&p, &s, cmd, fe, te, dt, tick, g, a, 0);
      }
      if (tick) {
        ticks++;
        for (int k = 0; k < 3; k++) { if (gyro) gyro[k * n + i] = g[k]; if (acc) acc[k * n + i] = a[k]; }
      }
    }
    for (int k = 0; k < 3; k++) { pos[k * n + i] = s.pos[k]; vel[k * n + i] = s.vel[k]; ang_vel[k * n + i] = s.ang_vel[k]; }
    for (int k = 0; k < 4; k++) { att[k * n + i] = s.att[k]; motor_speed[k * n + i] = s.motor_speed[k]; }
    if (rng) rng[i] = s.rng;
    if (ext_force && gust_period_us) for (int k = 0; k < 3; k++) ext_force[k * n + i] = fe[k];
  }
}
