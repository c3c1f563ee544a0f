/*
 * agrifly_oracle_planner.h -- CPU restatement of the RAPPIDS depth-image planner
 * (SURVEY.md 8f row f3): the sources under Components/Components/DepthImagePlanner,
 * Components/Components/TrajectoryGenerator and Common/Common/Math/{RootFinder,
 * Trajectory}.hpp, with a deterministic candidate COUNT in place of the
 * reference's wall-clock budget (DepthImagePlanner.cpp:123-126).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as agrifly_oracle.h).
 * PARITY STATUS
 *   pinned   : cubic / quartic root finder and the single-axis min-jerk
 *              trajectory against the reference's own RootFinder.hpp and
 *              SingleAxisTrajectory.{hpp,cpp} (stand-alone sources compiled in
 *              place: oracle/_ref/traj_probe -> tests/golden/planner_math_kat.json);
 *              candidate sampling against libstdc++ std::mt19937 +
 *              uniform_real_distribution in the reference's call shape.
 *   unpinned : everything that needs Vec3 / cv::Mat (RapidTrajectoryGenerator
 *              feasibility tests, monotonic sections, pyramid inflation, the
 *              search loop): restated line by line, reference quirks included
 *              (e.g. DepthImagePlanner.cpp:648 assigns bottomShrinkTemp to
 *              rightEdgeShrunk).
 */
#ifndef AGRIFLY_ORACLE_PLANNER_H
#define AGRIFLY_ORACLE_PLANNER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

unsigned ora_solve_cubic(double a, double b, double c, double x[3]);             /* RootFinder.hpp:55-96 */
unsigned ora_solve_quartic(double a, double b, double c, double d, double r[4]); /* RootFinder.hpp:104-176 */

typedef struct ora_axis {  /* SingleAxisTrajectory */
  double p0, v0, a0, pf, vf, af;
  double a, b, g, cost;
  double peak_t[2];
  int peak_init;
} ora_axis;
void ora_axis_generate(ora_axis *ax, double Tf);                                  /* SingleAxisTrajectory.cpp:59-107, full goal */
void ora_axis_minmax_acc(ora_axis *ax, double *amin, double *amax, double t1, double t2); /* :118-155 */
double ora_axis_max_jerk_sq(const ora_axis *ax, double t1, double t2);           /* :164-176 */
double ora_axis_pos(const ora_axis *ax, double t);
double ora_axis_vel(const ora_axis *ax, double t);
double ora_axis_acc(const ora_axis *ax, double t);

/* TrajectoryTestResult bits, DepthImagePlanner.hpp:38-44 */
enum { ORA_LOW_COST = 1, ORA_DYN_FEASIBLE = 2, ORA_VEL_ADMISSIBLE = 4, ORA_COLLISION_FREE = 8 };

typedef struct ora_planner_config {
  int width, height;
  double depth_scale, focal_length, cx, cy;
  double true_vehicle_radius, planning_vehicle_radius, min_checking_dist;
  double min_thrust, max_thrust, max_ang_vel, max_velocity, min_section_time; /* ctor defaults 5,30,20,5,0.02 */
  int max_pyramids;        /* _maxNumPyramids */
  int pixel_buffer;        /* _pyramidSearchPixelBuffer = 2 */
  int cost_type;           /* 0: ExplorationCost (direction), 1: main.cpp goal cost */
  double cost_vec[3];      /* exploration direction, or goal in the camera frame */
} ora_planner_config;
void ora_planner_default_config(ora_planner_config *c, int width, int height, double depth_scale,
                                double focal_length, double true_radius, double planning_radius,
                                double min_checking_dist);

typedef struct ora_plan_result {
  int found;               /* FindLowestCostTrajectory's return value */
  int best_index;          /* index of the winning candidate, -1 if none */
  double best_cost;
  double coeffs[6][3];     /* CommonMath::Trajectory of the winner: t^5 .. t^0 */
  double tf;
  int n_generated, n_cost_checks, n_collision_checks, n_velocity_checks, n_collision_free, n_pyramids;
} ora_plan_result;

/* samples[k] = {pixelX, pixelY, depth, time} of candidate k, as drawn by
 * RandomTrajectoryGenerator::GetNextCandidateTrajectory (DepthImagePlanner.hpp:
 * 393-404).  flags_out (optional) gets the TrajectoryTestResult of each candidate. */
void ora_planner_run(const ora_planner_config *cfg, const uint16_t *depth, const double vel0[3],
                     const double acc0[3], const double grav[3], const double (*samples)[4],
                     int n_candidates, ora_plan_result *out, uint8_t *flags_out);

/* libstdc++ std::mt19937(seed) + four uniform_real_distribution<> in g++'s
 * argument-evaluation order of the reference call site: depth, pixelY, pixelX
 * are drawn in that order, then time. */
void ora_planner_samples(uint32_t seed, int width, int height, int n, double (*samples)[4]);

/* brute-force check used by the tests (the reference's own self-check,
 * IsCollisionFreeGroundTruth, DepthImagePlanner.cpp:1031-1098, is a sampled ray
 * test; this is a denser independent version, not a restatement) */
int ora_planner_sampled_collision(const ora_planner_config *cfg, const uint16_t *depth,
                                  const double coeffs[6][3], double tf, int n_samples);

/* test hook: move the k-th acos / cos / pow result of the calling thread's next run by `ulps` (k < 0: off) */
void ora_planner_nudge(long call_index, int ulps);
long ora_planner_nudge_calls(void);

#ifdef __cplusplus
}
#endif
#endif
