/* agrifly_oracle_world.c -- see agrifly_oracle_world.h.  TEST INFRASTRUCTURE ONLY. */
#include "agrifly_oracle_world.h"

#include <math.h>

static int finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

void ora_nearest_neighbour(const float *all_xyz, int64_t n_all, int64_t first, int64_t count,
                           float *dist2, int32_t *index) {
  const float *X = all_xyz, *Y = all_xyz + n_all, *Z = all_xyz + 2 * n_all;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < count; k++) {
    const int64_t me = first + k;
    float best = 3.4e38f;
    int32_t best_j = -1;
    if (finite3(X[me], Y[me], Z[me])) {
      for (int64_t j = 0; j < n_all; j++) {
        if (j == me) continue;
        const float dx = X[j] - X[me], dy = Y[j] - Y[me], dz = Z[j] - Z[me];
        const float d = (float)((float)(dx * dx) + (float)(dy * dy)) + (float)(dz * dz);
        if (d < best) { best = d; best_j = (int32_t)j; }   /* NaN distances never win */
      }
    }
    dist2[k] = best;
    index[k] = best_j;
  }
}

/* ---- std::mt19937 (bits/random.tcc mersenne_twister_engine) ---- */
static void mt_seed(ora_uwb *u, uint32_t seed) {
  u->mt[0] = seed;
  for (int i = 1; i < 624; i++) u->mt[i] = 1812433253u * (u->mt[i - 1] ^ (u->mt[i - 1] >> 30)) + (uint32_t)i;
  u->idx = 624;
}

uint32_t ora_uwb_mt_next(ora_uwb *u) {
  if (u->idx >= 624) {
    for (int k = 0; k < 624; k++) {
      const uint32_t y = (u->mt[k] & 0x80000000u) | (u->mt[(k + 1) % 624] & 0x7fffffffu);
      u->mt[k] = u->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    u->idx = 0;
  }
  uint32_t z = u->mt[u->idx++];
  z ^= (z >> 11);
  z ^= (z << 7) & 0x9d2c5680u;
  z ^= (z << 15) & 0xefc60000u;
  z ^= (z >> 18);
  return z;
}

/* std::generate_canonical<double, 53>(mt19937): k = ceil(53 / 32) = 2 draws, R = 2^32 */
double ora_uwb_canonical(ora_uwb *u) {
  double sum = (double)ora_uwb_mt_next(u);
  sum += (double)ora_uwb_mt_next(u) * 4294967296.0;
  double ret = sum / 18446744073709551616.0;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}

/* std::normal_distribution<double>(0,1)::operator(): Marsaglia polar with the
 * second value of a pair kept for the next call */
double ora_uwb_normal(ora_uwb *u) {
  double ret;
  if (u->saved_available) {
    u->saved_available = 0;
    ret = u->saved;
  } else {
    double x, y, r2;
    do {
      x = 2.0 * ora_uwb_canonical(u) - 1.0;
      y = 2.0 * ora_uwb_canonical(u) - 1.0;
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = sqrt(-2 * log(r2) / r2);
    u->saved = x * mult;
    u->saved_available = 1;
    ret = y * mult;
  }
  return ret * 1.0 + 0.0; /* stddev 1, mean 0 */
}

void ora_uwb_init(ora_uwb *u, double noise_std, double outlier_prob, double outlier_std) {
  mt_seed(u, 0u); /* UWBNetwork.cpp:19 */
  u->saved_available = 0;
  u->saved = 0.0;
  u->noise_std = noise_std;
  u->outlier_prob = outlier_prob;
  u->outlier_std = outlier_std;
}

double ora_uwb_draw(ora_uwb *u, int *is_outlier) {
  /* distUniform(rng): generate_canonical * (b - a) + a with (a, b) = (0, 1) */
  const double uni = ora_uwb_canonical(u) * (1.0 - 0.0) + 0.0;
  if (uni < u->outlier_prob) { /* UWBNetwork.cpp:67 */
    if (is_outlier) *is_outlier = 1;
    return ora_uwb_normal(u) * u->outlier_std; /* :68 */
  }
  if (is_outlier) *is_outlier = 0;
  return ora_uwb_normal(u) * u->noise_std; /* :70 */
}

float ora_uwb_range(ora_uwb *u, const double p_req[3], const double p_res[3], int *is_outlier) {
  int out = 0;
  const double noise = ora_uwb_draw(u, &out);
  if (is_outlier) *is_outlier = out;
  if (out) return (float)noise; /* meas.range is a float, UWBRadio.hpp:21 */
  const double dx = p_req[0] - p_res[0], dy = p_req[1] - p_res[1], dz = p_req[2] - p_res[2];
  return (float)(sqrt(dx * dx + dy * dy + dz * dz) + noise); /* :71 */
}
