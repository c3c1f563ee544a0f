/*
 * agrifly_oracle_counter.h -- CPU checker for the engine's counter-based noise: the Monte-Carlo seed policy
 * (AFE_SEED_COUNTER) and the wind-gust process of BASELINE config 4 (afe_set_gust_process).
 *
 * TEST INFRASTRUCTURE ONLY (see agrifly_oracle.h).
 *
 * There is NO reference code behind these two: the reference seeds every vehicle's std::default_random_engine with 1
 * (Quadcopter_T.cpp:27) and has no gust model at all -- SetExternalForce (Quadcopter_T.hpp:45, applied at
 * Quadcopter_T.cpp:132) is the port, what drives it is the caller's business (SURVEY 8d config 4 asks for "seeded,
 * N(0, sigma^2), sigma swept 0...0.5 N, piecewise-constant 100 ms").  What is pinned is the third-party algorithm:
 *   Philox4x32-10, J. K. Salmon, M. A. Moraes, R. O. Dror, D. E. Shaw, "Parallel random numbers: as easy as 1, 2, 3",
 *   SC'11; Random123 library v1.14 (philox.h), checked against that library's published known-answer vectors
 *   (tests/test_counter_oracle.py).
 * Everything after the generator is this repository's own definition, stated here once and implemented twice
 * (here in double with libm, on the device in the engine's precision):
 *   words x0..x3 of one block; radius uniform  u_r = ((x_even >> 9) + 0.5) * 2^-23   in (0, 1), exact in fp32
 *                              angle  uniform  u_a =  (x_odd  >> 8)        * 2^-24   in [0, 1), exact in fp32
 *   Box-Muller:  r = sqrt(-2 ln u_r);  z_a = r cos(2 pi u_a);  z_b = r sin(2 pi u_a)      (|z| <= 5.65)
 *   block address: key = (seed low, seed high); counter = (index low, index high (16 bits) | stream << 16 | block << 24,
 *                  ordinal low, ordinal high); stream 1 = IMU noise (ordinal = logic-tick number, blocks 0 and 1:
 *                  gyro x y z = z0 z1 z2, accelerometer x y z = z3 z4 z5), stream 2 = gusts (ordinal = epoch, block 0:
 *                  force x y z = sigma_i (z0 z1 z2), sigma_i = sigma_max * index / (n_global - 1)).
 */
#ifndef AGRIFLY_ORACLE_COUNTER_H
#define AGRIFLY_ORACLE_COUNTER_H
#include <stdint.h>

#include "agrifly_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_STREAM_IMU 1u
#define ORA_STREAM_GUST 2u

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* the block at (seed, index, stream, block, ordinal) */
void ora_counter_block(uint64_t seed, uint64_t index, unsigned stream, unsigned block, uint64_t ordinal, uint32_t out[4]);
/* the two Box-Muller pairs of a block: z[0..3] */
void ora_counter_normals4(const uint32_t words[4], double z[4]);
/* six N(0,1) of vehicle `index` at logic tick `tick`: gyro x y z, accelerometer x y z */
void ora_imu_normals(uint64_t seed, uint64_t index, uint64_t tick, double z[6]);
/* gust force of vehicle `index` during epoch `epoch` */
void ora_gust_force(uint64_t seed, uint64_t index, uint64_t n_global, uint64_t epoch, double sigma_max, double force[3]);

/* ora_step_batch (agrifly_oracle.h) with the counter policy: the six normals of a tick come from
 * ora_imu_normals(seed, first_global + i, tick_base + ticks so far) instead of the vehicle's libstdc++ stream, and --
 * when gust_period_us != 0 -- the external force of every step is ora_gust_force(gust_seed, ...) at epoch (t0_us + step * dt_us) /
 * gust_period_us (ext_force is then an OUTPUT: the force of the last step).  use_counter_noise = 0 keeps the
 * libstdc++ stream (rng) and only adds the gusts. */
void ora_step_batch_counter(int64_t n, int n_steps, const ora_params *table, const uint8_t *types, double *pos, double *vel,
                            double *att, double *ang_vel, double *motor_speed, uint32_t *rng, const float *motor_cmd,
                            double *ext_force, const double *ext_torque, uint64_t dt_us, const uint8_t *tick_per_step,
                            float *gyro, float *acc, int use_counter_noise, uint64_t seed, uint64_t first_global,
                            uint64_t tick_base, uint64_t gust_seed, uint64_t gust_period_us, uint64_t t0_us, uint64_t n_global, double sigma_max);

#ifdef __cplusplus
}
#endif
#endif
