/*
 * agrifly_oracle_world.h -- CPU checker for the shared-world consumers of the
 * gathered position buffer.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as agrifly_oracle.h): only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * UWB ranging restates Simulation::UWBNetwork::Run,
 *   Components/Components/Simulation/UWBNetwork.cpp:4-6,19,66-71
 * (std::mt19937 seeded 0; per completed transaction one
 * std::uniform_real_distribution<double>(0,1) draw and one
 * std::normal_distribution<double>(0,1) draw, whose cached second value persists
 * across transactions; range narrowed to float, UWBRadio.hpp:21).
 * PARITY STATUS: the noise stream is PINNED against libstdc++ itself
 * (oracle/ref_uwb_probe.cpp -> tests/golden/uwb_kat.json: the same generator /
 * distribution classes in the call-site shape of UWBNetwork.cpp).  The geometric
 * part is |p_req - p_res| in double (Vec3.hpp:113-116, 165-172), unpinned like
 * the rest of Vec3 (UWBNetwork.cpp cannot be compiled here: Vec3.hpp includes
 * Matrix.hpp -> <Eigen/Dense>, absent from the image).
 *
 * Nearest neighbour has no reference counterpart (the reference has no
 * neighbour query; SURVEY 8e derives it from the north_star's "shared-world
 * neighbour/collision queries"): the checker is the O(n^2) definition itself.
 */
#ifndef AGRIFLY_ORACLE_WORLD_H
#define AGRIFLY_ORACLE_WORLD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* all_xyz: planar fp32 [3][n_all].  For vehicles first..first+count-1: squared
 * fp32 distance ((dx*dx + dy*dy) + dz*dz, every operation rounded to float) to
 * the nearest OTHER vehicle and its index; lowest index among equals; 3.4e38 / -1
 * when there is none or the vehicle's own position is not finite. */
void ora_nearest_neighbour(const float *all_xyz, int64_t n_all, int64_t first, int64_t count,
                           float *dist2, int32_t *index);

typedef struct ora_uwb {
  uint32_t mt[624];
  int idx;
  int saved_available; /* normal_distribution::_M_saved_available */
  double saved;
  double noise_std, outlier_prob, outlier_std;
} ora_uwb;

void ora_uwb_init(ora_uwb *u, double noise_std, double outlier_prob, double outlier_std); /* rng.seed(0) */
/* one completed transaction, UWBNetwork.cpp:66-71; positions in double */
float ora_uwb_range(ora_uwb *u, const double p_req[3], const double p_res[3], int *is_outlier);
/* the draws alone: noise term (already scaled) and the outlier decision */
double ora_uwb_draw(ora_uwb *u, int *is_outlier);
/* raw pieces, for the known-answer test */
uint32_t ora_uwb_mt_next(ora_uwb *u);
double ora_uwb_canonical(ora_uwb *u);
double ora_uwb_normal(ora_uwb *u);

#ifdef __cplusplus
}
#endif
#endif
