/*
 * agrifly_oracle_planner.c -- see agrifly_oracle_planner.h.  TEST INFRASTRUCTURE.
 * double arithmetic, reference operation order; citations are file:line under
 * /root/reference.  DIP = Components/Components/DepthImagePlanner/DepthImagePlanner.cpp,
 * RTG = Components/Components/TrajectoryGenerator/RapidTrajectoryGenerator.{hpp,cpp},
 * SAT = .../SingleAxisTrajectory.{hpp,cpp}.
 */
#include "agrifly_oracle_planner.h"

#include <float.h>
#include <math.h>

/* std::min / std::max semantics (return the first argument on ties / unordered) */
#define SMIN(a, b) (((b) < (a)) ? (b) : (a))
#define SMAX(a, b) (((a) < (b)) ? (b) : (a))
#include <stdlib.h>
#include <string.h>

/* ---- Common/Common/Math/RootFinder.hpp ---------------------------------- */
/* :40-44: the constants are FLOATS in the reference */
static const float RF_PI = 3.141592653589793238463;
static const float RF_EPS = 1e-12;
#define RF_2PI ((float)(2 * RF_PI))

/* Test hook (sensitivity analysis, tests/campaigns/planner_campaign.py): the k-th acos / cos / pow result of the
 * calling thread can be moved by a few ulps.  libm implementations differ by an ulp in these (glibc vs the
 * device's math library); a planner outcome that flips under such a nudge is one no two platforms agree on.
 * Off unless ora_planner_nudge() was called on this thread. */
static _Thread_local long ora_nudge_at = -1, ora_nudge_seen = 0;
static _Thread_local int ora_nudge_ulps = 0;
void ora_planner_nudge(long call_index, int ulps) { ora_nudge_at = call_index; ora_nudge_ulps = ulps; ora_nudge_seen = 0; }
long ora_planner_nudge_calls(void) { return ora_nudge_seen; }
static double ora_tr(double v) {
  if (ora_nudge_seen++ == ora_nudge_at) {
    int k;
    for (k = 0; k < abs(ora_nudge_ulps); k++) v = nextafter(v, ora_nudge_ulps > 0 ? INFINITY : -INFINITY);
  }
  return v;
}

unsigned ora_solve_cubic(double a, double b, double c, double *x) {
  /* :55-96 */
  double a2 = a * a;
  double q = (a2 - 3 * b) / 9;
  double r = (a * (2 * a2 - 9 * b) + 27 * c) / 54;
  double r2 = r * r;
  double q3 = q * q * q;
  double A, B;
  if (r2 < q3) {
    double t = r / sqrt(q3);
    if (t < -1) t = -1;
    if (t > 1) t = 1;
    t = ora_tr(acos(t));
    a /= 3;
    q = -2 * sqrt(q);
    x[0] = q * ora_tr(cos(t / 3)) - a;
    x[1] = q * ora_tr(cos((t + (double)RF_2PI) / (double)3)) - a;
    x[2] = q * ora_tr(cos((t - (double)RF_2PI) / (double)3)) - a;
    return 3;
  } else {
    A = -ora_tr(pow(fabs(r) + sqrt(r2 - q3), 1. / 3));
    if (r < 0) A = -A;
    B = (fabs(A) < (double)RF_EPS ? 0 : q / A);
    a /= 3;
    x[0] = (A + B) - a;
    x[1] = (double)(-0.5) * (A + B) - a;
    x[2] = (double)(0.5) * sqrt((double)(3.)) * (A - B);
    if (fabs(x[2]) < (double)RF_EPS) {
      x[2] = x[1];
      return 2;
    }
    return 1;
  }
}

unsigned ora_solve_quartic(double a, double b, double c, double d, double *root) {
  /* :104-176 */
  double a3 = -b;
  double b3 = a * c - 4. * d;
  double c3 = -a * a * d - c * c + 4. * b * d;
  int rCnt = 0;
  double x3[3];
  unsigned iZeroes = ora_solve_cubic(a3, b3, c3, x3);
  double q1, q2, p1, p2, D, sqD, y;
  y = x3[0];
  if (iZeroes != 1) {
    if (fabs(x3[1]) > fabs(y)) y = x3[1];
    if (fabs(x3[2]) > fabs(y)) y = x3[2];
  }
  D = y * y - 4 * d;
  if (fabs(D) < (double)RF_EPS) {
    q1 = q2 = y * 0.5;
    D = a * a - 4. * (b - y);
    if (fabs(D) < (double)RF_EPS) {
      p1 = p2 = a * 0.5;
    } else {
      sqD = sqrt(D);
      p1 = (a + sqD) * 0.5;
      p2 = (a - sqD) * 0.5;
    }
  } else {
    sqD = sqrt(D);
    q1 = (y + sqD) * 0.5;
    q2 = (y - sqD) * 0.5;
    p1 = (a * q1 - c) / (q1 - q2);
    p2 = (c - a * q2) / (q1 - q2);
  }
  D = p1 * p1 - 4 * q1;
  if (!(D < 0.0)) {
    sqD = sqrt(D);
    root[rCnt] = (-p1 + sqD) * 0.5; ++rCnt;
    root[rCnt] = (-p1 - sqD) * 0.5; ++rCnt;
  }
  D = p2 * p2 - 4 * q2;
  if (!(D < 0.0)) {
    sqD = sqrt(D);
    root[rCnt] = (-p2 + sqD) * 0.5; ++rCnt;
    root[rCnt] = (-p2 - sqD) * 0.5; ++rCnt;
  }
  return rCnt;
}

/* ---- SingleAxisTrajectory ------------------------------------------------ */
double ora_axis_acc(const ora_axis *x, double t) { /* SAT.hpp GetAcceleration */
  return x->a0 + x->g * t + (1 / 2.0) * x->b * t * t + (1 / 6.0) * x->a * t * t * t;
}
double ora_axis_vel(const ora_axis *x, double t) {
  return x->v0 + x->a0 * t + (1 / 2.0) * x->g * t * t + (1 / 6.0) * x->b * t * t * t + (1 / 24.0) * x->a * t * t * t * t;
}
double ora_axis_pos(const ora_axis *x, double t) {
  return x->p0 + x->v0 * t + (1 / 2.0) * x->a0 * t * t + (1 / 6.0) * x->g * t * t * t +
         (1 / 24.0) * x->b * t * t * t * t + (1 / 120.0) * x->a * t * t * t * t * t;
}
static double axis_jerk(const ora_axis *x, double t) { return x->g + x->b * t + (1 / 2.0) * x->a * t * t; }

void ora_axis_generate(ora_axis *x, double Tf) {
  /* SAT.cpp:59-107, branch _posGoalDefined && _velGoalDefined && _accGoalDefined */
  double delta_a = x->af - x->a0;
  double delta_v = x->vf - x->v0 - x->a0 * Tf;
  double delta_p = x->pf - x->p0 - x->v0 * Tf - 0.5 * x->a0 * Tf * Tf;
  const double T2 = Tf * Tf;
  const double T3 = T2 * Tf;
  const double T4 = T3 * Tf;
  const double T5 = T4 * Tf;
  x->a = (60 * T2 * delta_a - 360 * Tf * delta_v + 720 * 1 * delta_p) / T5;
  x->b = (-24 * T3 * delta_a + 168 * T2 * delta_v - 360 * Tf * delta_p) / T5;
  x->g = (3 * T4 * delta_a - 24 * T3 * delta_v + 60 * T2 * delta_p) / T5;
  x->cost = x->g * x->g + x->b * x->g * Tf + x->b * x->b * T2 / 3.0 + x->a * x->g * T2 / 3.0 +
            x->a * x->b * T3 / 4.0 + x->a * x->a * T4 / 20.0;
  x->peak_init = 0; /* Reset() precedes every Generate in the planner (DIP.hpp:397) */
}

void ora_axis_minmax_acc(ora_axis *x, double *aMinOut, double *aMaxOut, double t1, double t2) {
  /* SAT.cpp:118-155 */
  if (!x->peak_init) {
    if (x->a) {
      double det = x->b * x->b - 2 * x->g * x->a;
      if (det < 0) {
        x->peak_t[0] = 0;
        x->peak_t[1] = 0;
      } else {
        x->peak_t[0] = (-x->b + sqrt(det)) / x->a;
        x->peak_t[1] = (-x->b - sqrt(det)) / x->a;
      }
    } else {
      if (x->b) x->peak_t[0] = -x->g / x->b;
      else x->peak_t[0] = 0;
      x->peak_t[1] = 0;
    }
    x->peak_init = 1;
  }
  *aMinOut = SMIN(ora_axis_acc(x, t1), ora_axis_acc(x, t2));
  *aMaxOut = SMAX(ora_axis_acc(x, t1), ora_axis_acc(x, t2));
  for (int i = 0; i < 2; i++) {
    if (x->peak_t[i] <= t1) continue;
    if (x->peak_t[i] >= t2) continue;
    *aMinOut = SMIN(*aMinOut, ora_axis_acc(x, x->peak_t[i]));
    *aMaxOut = SMAX(*aMaxOut, ora_axis_acc(x, x->peak_t[i]));
  }
}

double ora_axis_max_jerk_sq(const ora_axis *x, double t1, double t2) {
  /* SAT.cpp:164-176 */
  double jMaxSqr = SMAX(pow(axis_jerk(x, t1), 2), pow(axis_jerk(x, t2), 2));
  if (x->a) {
    double tMax = -x->b / x->a;
    if (tMax > t1 && tMax < t2) jMaxSqr = SMAX(pow(axis_jerk(x, tMax), 2), jMaxSqr);
  }
  return jMaxSqr;
}

/* ---- RapidTrajectoryGenerator ------------------------------------------- */
typedef struct {
  ora_axis ax[3];
  double grav[3];
  double tf;
} rtg;

enum { IN_FEASIBLE = 0, IN_INDETERMINABLE = 1, IN_THRUST_HIGH = 2, IN_THRUST_LOW = 3 };

static double rtg_thrust(const rtg *g, double t) { /* RTG.hpp GetThrust: (acc - grav).GetNorm2() */
  const double x = ora_axis_acc(&g->ax[0], t) - g->grav[0];
  const double y = ora_axis_acc(&g->ax[1], t) - g->grav[1];
  const double z = ora_axis_acc(&g->ax[2], t) - g->grav[2];
  return sqrt(x * x + y * y + z * z);
}

static int rtg_input_section(rtg *g, double fminAllowed, double fmaxAllowed, double wmaxAllowed, double t1,
                             double t2, double minTimeSection) {
  /* RTG.cpp:75-150 */
  if (t2 - t1 < minTimeSection) return IN_INDETERMINABLE;
  if (SMAX(rtg_thrust(g, t1), rtg_thrust(g, t2)) > fmaxAllowed) return IN_THRUST_HIGH;
  if (SMIN(rtg_thrust(g, t1), rtg_thrust(g, t2)) < fminAllowed) return IN_THRUST_LOW;
  double fminSqr = 0, fmaxSqr = 0, jmaxSqr = 0;
  for (int i = 0; i < 3; i++) {
    double amin, amax;
    ora_axis_minmax_acc(&g->ax[i], &amin, &amax, t1, t2);
    double v1 = amin - g->grav[i];
    double v2 = amax - g->grav[i];
    if (SMAX(pow(v1, 2), pow(v2, 2)) > pow(fmaxAllowed, 2)) return IN_THRUST_HIGH;
    if (v1 * v2 < 0) fminSqr += 0;
    else fminSqr += pow(SMIN(fabs(v1), fabs(v2)), 2);
    fmaxSqr += pow(SMAX(fabs(v1), fabs(v2)), 2);
    jmaxSqr += ora_axis_max_jerk_sq(&g->ax[i], t1, t2);
  }
  double fmin_ = sqrt(fminSqr);
  double fmax_ = sqrt(fmaxSqr);
  double wBound;
  if (fminSqr > 1e-6) wBound = sqrt(jmaxSqr / fminSqr);
  else wBound = DBL_MAX;
  if (fmax_ < fminAllowed) return IN_THRUST_LOW;
  if (fmin_ > fmaxAllowed) return IN_THRUST_HIGH;
  if (fmin_ < fminAllowed || fmax_ > fmaxAllowed || wBound > wmaxAllowed) {
    double tHalf = (t1 + t2) / 2;
    int r1 = rtg_input_section(g, fminAllowed, fmaxAllowed, wmaxAllowed, t1, tHalf, minTimeSection);
    if (r1 == IN_FEASIBLE) return rtg_input_section(g, fminAllowed, fmaxAllowed, wmaxAllowed, tHalf, t2, minTimeSection);
    return r1;
  }
  return IN_FEASIBLE;
}

static int rtg_velocity_feasible(const rtg *g, double vmax) {
  /* RTG.cpp:163-208; returns 1 for StateFeasible */
  for (int dim = 0; dim < 3; dim++) {
    double c[4];
    c[0] = g->ax[dim].a / 6.0;
    c[1] = g->ax[dim].b / 2.0;
    c[2] = g->ax[dim].g / 1.0;
    c[3] = g->ax[dim].a0;
    double roots[3 + 2];
    unsigned rootCount = 0;
    if (fabs(c[0]) > 1e-6) rootCount = ora_solve_cubic(c[1] / c[0], c[2] / c[0], c[3] / c[0], roots);
    else return 0;
    roots[rootCount] = 0;
    roots[rootCount + 1] = g->tf;
    for (unsigned i = 0; i < (rootCount + 2); i++) {
      if (roots[i] < 0) continue;
      if (roots[i] > g->tf) continue;
      const double vx = ora_axis_vel(&g->ax[0], roots[i]);
      const double vy = ora_axis_vel(&g->ax[1], roots[i]);
      const double vz = ora_axis_vel(&g->ax[2], roots[i]);
      if (fabs(vx) >= vmax || fabs(vy) >= vmax || fabs(vz) >= vmax) return 0;
    }
  }
  return 1;
}

static void rtg_trajectory(const rtg *g, double c[6][3]) {
  /* RTG.hpp GetTrajectory */
  for (int i = 0; i < 3; i++) {
    c[0][i] = g->ax[i].a / 120;
    c[1][i] = g->ax[i].b / 24;
    c[2][i] = g->ax[i].g / 6;
    c[3][i] = ora_axis_acc(&g->ax[i], 0) / 2;
    c[4][i] = ora_axis_vel(&g->ax[i], 0);
    c[5][i] = ora_axis_pos(&g->ax[i], 0);
  }
}

/* ---- CommonMath::Trajectory / MonotonicTrajectory ------------------------ */
typedef struct {
  double t0, t1;
  int increasing;
} mono;

static double traj_axis(const double c[6][3], int i, double t) { /* Trajectory.hpp:90-96 */
  return c[0][i] * t * t * t * t * t + c[1][i] * t * t * t * t + c[2][i] * t * t * t + c[3][i] * t * t + c[4][i] * t +
         c[5][i];
}
static mono mono_make(const double c[6][3], double t0, double t1) { /* MonotonicTrajectory.hpp:34-41 */
  mono m = {t0, t1, 0};
  m.increasing = traj_axis(c, 2, t0) < traj_axis(c, 2, t1);
  return m;
}
static double mono_deepest(const double c[6][3], const mono *m) { /* :46-60 */
  return m->increasing ? traj_axis(c, 2, m->t1) : traj_axis(c, 2, m->t0);
}

static void sort_doubles(double *a, int n) { /* std::sort on a handful of doubles */
  for (int i = 1; i < n; i++) {
    double v = a[i];
    int j = i;
    while (j > 0 && v < a[j - 1]) { a[j] = a[j - 1]; j--; }
    a[j] = v;
  }
}

/* ---- Pyramid ------------------------------------------------------------- */
typedef struct {
  double depth;
  int right, top, left, bottom;
  double normal[4][3];
} pyramid;

typedef struct {
  const ora_planner_config *cfg;
  const uint16_t *depth;
  pyramid *pyr;
  int n_pyr, cap_pyr;
} planner;

static void unit_cross(const double a[3], const double b[3], double o[3]) {
  /* Pyramid.hpp:52-57: Cross().GetUnitVector(); Vec3.hpp:126-129 truncates the norm to float */
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  const float n = sqrt(x * x + y * y + z * z);
  o[0] = x / n; o[1] = y / n; o[2] = z / n;
}

static void deproject(const ora_planner_config *c, double x, double y, double depth, double o[3]) {
  /* DIP.hpp:274-279 */
  o[0] = depth * ((x - c->cx) / c->focal_length);
  o[1] = depth * ((y - c->cy) / c->focal_length);
  o[2] = depth * 1;
}

static int inflate_pyramid(planner *P, int x0, int y0, double minimumDepth, pyramid *out) {
  /* DIP.cpp:456-970 */
  const ora_planner_config *c = P->cfg;
  const uint16_t *img = P->depth;
  const int W = c->width, H = c->height, buf = c->pixel_buffer;
  int imageEdgeOffset = c->focal_length * c->true_vehicle_radius / c->min_checking_dist;
  if (x0 <= imageEdgeOffset + buf + 1 || x0 > W - imageEdgeOffset - buf - 1 || y0 <= imageEdgeOffset + buf + 1 ||
      y0 > H - imageEdgeOffset - buf - 1)
    return 0;
  uint16_t minimumPyramidDepth = (uint16_t)((minimumDepth + c->planning_vehicle_radius) / c->depth_scale);
  int initPixSearchRadius = c->focal_length * c->planning_vehicle_radius / (c->depth_scale * minimumPyramidDepth);
  if (2 * initPixSearchRadius >= (W < H ? W : H) - 2 * imageEdgeOffset) return 0;

  int leftEdge, topEdge, rightEdge, bottomEdge;
  if (y0 - initPixSearchRadius < imageEdgeOffset) {
    topEdge = imageEdgeOffset;
    bottomEdge = topEdge + 2 * initPixSearchRadius;
  } else {
    bottomEdge = (H - imageEdgeOffset - 1 < y0 + initPixSearchRadius) ? H - imageEdgeOffset - 1 : y0 + initPixSearchRadius;
    topEdge = bottomEdge - 2 * initPixSearchRadius;
  }
  if (x0 - initPixSearchRadius < imageEdgeOffset) {
    leftEdge = imageEdgeOffset;
    rightEdge = leftEdge + 2 * initPixSearchRadius;
  } else {
    rightEdge = (W - imageEdgeOffset - 1 < x0 + initPixSearchRadius) ? W - imageEdgeOffset - 1 : x0 + initPixSearchRadius;
    leftEdge = rightEdge - 2 * initPixSearchRadius;
  }
  uint16_t ignoreDist = (uint16_t)(c->true_vehicle_radius / c->depth_scale);
  uint16_t pixDist;
  for (int y = topEdge; y < bottomEdge; y++)
    for (int x = leftEdge; x < rightEdge; x++) {
      pixDist = img[y * W + x];
      if (pixDist <= minimumPyramidDepth && pixDist > ignoreDist) return 0;
    }

  uint16_t maxDepthExpandedPyramid = 65535;
  int rightFree = 1, topFree = 1, leftFree = 1, bottomFree = 1;
  while (rightFree || topFree || leftFree || bottomFree) {
    if (rightFree) {
      if (rightEdge < W - imageEdgeOffset - 1) {
        for (int y = topEdge; y <= bottomEdge; y++) {
          pixDist = img[y * W + rightEdge + 1];
          if (pixDist > ignoreDist) {
            if (pixDist < minimumPyramidDepth) { rightFree = 0; rightEdge--; break; }
            if (pixDist < maxDepthExpandedPyramid) maxDepthExpandedPyramid = pixDist;
          }
        }
        rightEdge++;
      } else rightFree = 0;
    }
    if (topFree) {
      if (topEdge > imageEdgeOffset) {
        for (int x = leftEdge; x <= rightEdge; x++) {
          pixDist = img[(topEdge - 1) * W + x];
          if (pixDist > ignoreDist) {
            if (pixDist < minimumPyramidDepth) { topFree = 0; topEdge++; break; }
            if (pixDist < maxDepthExpandedPyramid) maxDepthExpandedPyramid = pixDist;
          }
        }
        topEdge--;
      } else topFree = 0;
    }
    if (leftFree) {
      if (leftEdge > imageEdgeOffset) {
        for (int y = topEdge; y <= bottomEdge; y++) {
          pixDist = img[y * W + leftEdge - 1];
          if (pixDist > ignoreDist) {
            if (pixDist < minimumPyramidDepth) { leftFree = 0; leftEdge++; break; }
            if (pixDist < maxDepthExpandedPyramid) maxDepthExpandedPyramid = pixDist;
          }
        }
        leftEdge--;
      } else leftFree = 0;
    }
    if (bottomFree) {
      if (bottomEdge < H - imageEdgeOffset - 1) {
        for (int x = leftEdge; x <= rightEdge; x++) {
          pixDist = img[(bottomEdge + 1) * W + x];
          if (pixDist > ignoreDist) {
            if (pixDist < minimumPyramidDepth) { bottomFree = 0; bottomEdge--; break; }
            if (pixDist < maxDepthExpandedPyramid) maxDepthExpandedPyramid = pixDist;
          }
        }
        bottomEdge++;
      } else bottomFree = 0;
    }
  }

  int rightEdgeShrunk = W - 1 - imageEdgeOffset;
  int leftEdgeShrunk = imageEdgeOffset;
  int topEdgeShrunk = imageEdgeOffset;
  int bottomEdgeShrunk = H - 1 - imageEdgeOffset;
  int numerator = c->focal_length * c->planning_vehicle_radius / c->depth_scale;

  /* right side, :617-661 */
  for (int x = rightEdge; x < W; x++)
    for (int y = topEdge; y <= bottomEdge; y++) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if (numerator > (x - rightEdgeShrunk) * pixDist) {
          int rightShrinkTemp = x - (int)(numerator / pixDist);
          if (x0 > rightShrinkTemp - buf) {
            int topShrinkTemp = y + (int)(numerator / pixDist);
            int bottomShrinkTemp = y - (int)(numerator / pixDist);
            if (y0 < topShrinkTemp + buf && y0 > bottomShrinkTemp - buf) return 0;
            else if (y0 < topShrinkTemp + buf) bottomEdgeShrunk = bottomShrinkTemp;
            else if (y0 > bottomShrinkTemp - buf) topEdgeShrunk = topShrinkTemp;
            else {
              int uShrinkLostArea = (topShrinkTemp - topEdgeShrunk);
              int dShrinkLostArea = (bottomEdgeShrunk - bottomShrinkTemp);
              if (dShrinkLostArea > uShrinkLostArea) topEdgeShrunk = topShrinkTemp;
              else rightEdgeShrunk = bottomShrinkTemp; /* sic: DIP.cpp:648 */
            }
          } else rightEdgeShrunk = rightShrinkTemp;
        }
      }
    }
  /* left side, :663-698 */
  for (int x = leftEdge; x >= 0; x--)
    for (int y = topEdge; y <= bottomEdge; y++) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if ((leftEdgeShrunk - x) * pixDist < numerator) {
          int leftShrinkTemp = x + (int)(numerator / pixDist);
          if (x0 < leftShrinkTemp + buf) {
            int topShrinkTemp = y + (int)(numerator / pixDist);
            int bottomShrinkTemp = y - (int)(numerator / pixDist);
            if (y0 < topShrinkTemp + buf && y0 > bottomShrinkTemp - buf) return 0;
            else if (y0 < topShrinkTemp + buf) bottomEdgeShrunk = bottomShrinkTemp;
            else if (y0 > bottomShrinkTemp - buf) topEdgeShrunk = topShrinkTemp;
            else {
              int uShrinkLostArea = (topShrinkTemp - topEdgeShrunk);
              int dShrinkLostArea = (bottomEdgeShrunk - bottomShrinkTemp);
              if (dShrinkLostArea > uShrinkLostArea) topEdgeShrunk = topShrinkTemp;
              else bottomEdgeShrunk = bottomShrinkTemp;
            }
          } else leftEdgeShrunk = leftShrinkTemp;
        }
      }
    }
  if (leftEdgeShrunk + buf > rightEdgeShrunk - buf) return 0;

  /* top side, :705-744 */
  for (int y = topEdge; y >= 0; y--)
    for (int x = leftEdge; x <= rightEdge; x++) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if ((topEdgeShrunk - y) * pixDist < numerator) {
          int topShrinkTemp = y + (int)(numerator / pixDist);
          if (y0 < topShrinkTemp + buf) {
            int rightShrinkTemp = x - (int)(numerator / pixDist);
            int leftShrinkTemp = x + (int)(numerator / pixDist);
            if (x0 > rightShrinkTemp - buf && x0 < leftShrinkTemp + buf) return 0;
            else if (x0 > rightShrinkTemp - buf) leftEdgeShrunk = leftShrinkTemp;
            else if (x0 < leftShrinkTemp + buf) rightEdgeShrunk = rightShrinkTemp;
            else {
              int rShrinkLostArea = (rightEdgeShrunk - rightShrinkTemp);
              int lShrinkLostArea = (leftShrinkTemp - leftEdgeShrunk);
              if (rShrinkLostArea > lShrinkLostArea) leftEdgeShrunk = leftShrinkTemp;
              else rightEdgeShrunk = rightShrinkTemp;
            }
          } else topEdgeShrunk = topShrinkTemp;
        }
      }
    }
  /* bottom side, :746-785 */
  for (int y = bottomEdge; y < H; y++)
    for (int x = leftEdge; x <= rightEdge; x++) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if (numerator > (y - bottomEdgeShrunk) * pixDist) {
          int bottomShrinkTemp = y - (int)(numerator / pixDist);
          if (y0 > bottomShrinkTemp - buf) {
            int rightShrinkTemp = x - (int)(numerator / pixDist);
            int leftShrinkTemp = x + (int)(numerator / pixDist);
            if (x0 > rightShrinkTemp - buf && x0 < leftShrinkTemp + buf) return 0;
            else if (x0 > rightShrinkTemp - buf) leftEdgeShrunk = leftShrinkTemp;
            else if (x0 < leftShrinkTemp + buf) rightEdgeShrunk = rightShrinkTemp;
            else {
              int rShrinkLostArea = (rightEdgeShrunk - rightShrinkTemp);
              int lShrinkLostArea = (leftShrinkTemp - leftEdgeShrunk);
              if (rShrinkLostArea > lShrinkLostArea) leftEdgeShrunk = leftShrinkTemp;
              else rightEdgeShrunk = rightShrinkTemp;
            }
          } else bottomEdgeShrunk = bottomShrinkTemp;
        }
      }
    }
  if (topEdgeShrunk + buf > bottomEdgeShrunk - buf) return 0;

  /* top right corner, :794-829 */
  for (int y = topEdge; y >= 0; y--)
    for (int x = rightEdge; x < W; x++) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if (numerator > (x - rightEdgeShrunk) * pixDist && (topEdgeShrunk - y) * pixDist < numerator) {
          int rightShrinkTemp = x - (int)(numerator / pixDist);
          int topShrinkTemp = y + (int)(numerator / pixDist);
          if (x0 > rightShrinkTemp - buf && y0 < topShrinkTemp + buf) return 0;
          else if (x0 > rightShrinkTemp - buf) topEdgeShrunk = topShrinkTemp;
          else if (y0 < topShrinkTemp + buf) rightEdgeShrunk = rightShrinkTemp;
          else {
            int rShrinkLostArea = (rightEdgeShrunk - rightShrinkTemp) * (bottomEdgeShrunk - topEdgeShrunk);
            int uShrinkLostArea = (topShrinkTemp - topEdgeShrunk) * (rightEdgeShrunk - leftEdgeShrunk);
            if (rShrinkLostArea > uShrinkLostArea) topEdgeShrunk = topShrinkTemp;
            else rightEdgeShrunk = rightShrinkTemp;
          }
        }
      }
    }
  /* bottom right corner, :831-866 */
  for (int y = bottomEdge; y < H; y++)
    for (int x = rightEdge; x < W; x++) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if (numerator > (x - rightEdgeShrunk) * pixDist && numerator > (y - bottomEdgeShrunk) * pixDist) {
          int rightShrinkTemp = x - (int)(numerator / pixDist);
          int bottomShrinkTemp = y - (int)(numerator / pixDist);
          if (x0 > rightShrinkTemp - buf && y0 > bottomShrinkTemp - buf) return 0;
          else if (x0 > rightShrinkTemp - buf) bottomEdgeShrunk = bottomShrinkTemp;
          else if (y0 > bottomShrinkTemp - buf) rightEdgeShrunk = rightShrinkTemp;
          else {
            int rShrinkLostArea = (rightEdgeShrunk - rightShrinkTemp) * (bottomEdgeShrunk - topEdgeShrunk);
            int dShrinkLostArea = (bottomEdgeShrunk - bottomShrinkTemp) * (rightEdgeShrunk - leftEdgeShrunk);
            if (rShrinkLostArea > dShrinkLostArea) bottomEdgeShrunk = bottomShrinkTemp;
            else rightEdgeShrunk = rightShrinkTemp;
          }
        }
      }
    }
  /* top left corner, :868-903 */
  for (int y = topEdge; y >= 0; y--)
    for (int x = leftEdge; x >= 0; x--) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if ((leftEdgeShrunk - x) * pixDist < numerator && (topEdgeShrunk - y) * pixDist < numerator) {
          int leftShrinkTemp = x + (int)(numerator / pixDist);
          int topShrinkTemp = y + (int)(numerator / pixDist);
          if (x0 < leftShrinkTemp + buf && y0 < topShrinkTemp + buf) return 0;
          else if (x0 < leftShrinkTemp + buf) topEdgeShrunk = topShrinkTemp;
          else if (y0 < topShrinkTemp + buf) leftEdgeShrunk = leftShrinkTemp;
          else {
            int lShrinkLostArea = (leftShrinkTemp - leftEdgeShrunk) * (bottomEdgeShrunk - topEdgeShrunk);
            int uShrinkLostArea = (topShrinkTemp - topEdgeShrunk) * (rightEdgeShrunk - leftEdgeShrunk);
            if (lShrinkLostArea > uShrinkLostArea) topEdgeShrunk = topShrinkTemp;
            else leftEdgeShrunk = leftShrinkTemp;
          }
        }
      }
    }
  /* bottom left corner, :905-940 */
  for (int y = bottomEdge; y < H; y++)
    for (int x = leftEdge; x >= 0; x--) {
      pixDist = img[y * W + x];
      if (pixDist > ignoreDist && pixDist < maxDepthExpandedPyramid) {
        if ((leftEdgeShrunk - x) * pixDist < numerator && numerator > (y - bottomEdgeShrunk) * pixDist) {
          int leftShrinkTemp = x + (int)(numerator / pixDist);
          int bottomShrinkTemp = y - (int)(numerator / pixDist);
          if (x0 < leftShrinkTemp + buf && y0 > bottomShrinkTemp - buf) return 0;
          else if (x0 < leftShrinkTemp + buf) bottomEdgeShrunk = bottomShrinkTemp;
          else if (y0 > bottomShrinkTemp - buf) leftEdgeShrunk = leftShrinkTemp;
          else {
            int lShrinkLostArea = (leftShrinkTemp - leftEdgeShrunk) * (bottomEdgeShrunk - topEdgeShrunk);
            int dShrinkLostArea = (bottomEdgeShrunk - bottomShrinkTemp) * (rightEdgeShrunk - leftEdgeShrunk);
            if (lShrinkLostArea > dShrinkLostArea) bottomEdgeShrunk = bottomShrinkTemp;
            else leftEdgeShrunk = leftShrinkTemp;
          }
        }
      }
    }

  /* :942-966 */
  double depth = maxDepthExpandedPyramid * c->depth_scale - c->planning_vehicle_radius;
  double corners[4][3];
  deproject(c, (double)rightEdgeShrunk, (double)topEdgeShrunk, depth, corners[0]);
  deproject(c, (double)leftEdgeShrunk, (double)topEdgeShrunk, depth, corners[1]);
  deproject(c, (double)leftEdgeShrunk, (double)bottomEdgeShrunk, depth, corners[2]);
  deproject(c, (double)rightEdgeShrunk, (double)bottomEdgeShrunk, depth, corners[3]);
  out->depth = depth;
  out->right = rightEdgeShrunk; out->top = topEdgeShrunk; out->left = leftEdgeShrunk; out->bottom = bottomEdgeShrunk;
  unit_cross(corners[0], corners[1], out->normal[0]); /* Pyramid.hpp:52-57 */
  unit_cross(corners[1], corners[2], out->normal[1]);
  unit_cross(corners[2], corners[3], out->normal[2]);
  unit_cross(corners[3], corners[0], out->normal[3]);
  return 1;
}

static int find_containing_pyramid(const planner *P, double px, double py, double depth, pyramid *out) {
  /* DIP.cpp:356-380: std::lower_bound on depth, then linear scan */
  int first = 0;
  while (first < P->n_pyr && P->pyr[first].depth < depth) first++;
  const int buf = P->cfg->pixel_buffer;
  for (int k = first; k < P->n_pyr; k++) {
    const pyramid *p = &P->pyr[k];
    if (p->left + buf < px && px < p->right - buf && p->top + buf < py && py < p->bottom - buf) {
      *out = *p;
      return 1;
    }
  }
  return 0;
}

static int deepest_collision_time(const double coeffs[6][3], const mono *m, const pyramid *p, double *outT) {
  /* DIP.cpp:382-454 */
  int collides = 0;
  *outT = m->increasing ? m->t0 : m->t1;
  for (int f = 0; f < 4; f++) {
    double c[5] = {0, 0, 0, 0, 0};
    for (int dim = 0; dim < 3; dim++) {
      c[0] += p->normal[f][dim] * coeffs[0][dim];
      c[1] += p->normal[f][dim] * coeffs[1][dim];
      c[2] += p->normal[f][dim] * coeffs[2][dim];
      c[3] += p->normal[f][dim] * coeffs[3][dim];
      c[4] += p->normal[f][dim] * coeffs[4][dim];
    }
    double roots[4];
    unsigned rootCount;
    if (fabs(c[0]) > 1e-6) rootCount = ora_solve_quartic(c[1] / c[0], c[2] / c[0], c[3] / c[0], c[4] / c[0], roots);
    else rootCount = ora_solve_cubic(c[2] / c[1], c[3] / c[1], c[4] / c[1], roots);
    sort_doubles(roots, (int)rootCount);
    if (m->increasing) {
      for (int i = (int)rootCount - 1; i >= 0; i--) {
        if (roots[i] > m->t1) continue;
        else if (roots[i] > m->t0) {
          if (roots[i] > *outT) { *outT = roots[i]; collides = 1; break; }
        } else break;
      }
    } else {
      for (int i = 0; i < (int)rootCount; i++) {
        if (roots[i] < m->t0) continue;
        else if (roots[i] < m->t1) {
          if (roots[i] < *outT) { *outT = roots[i]; collides = 1; break; }
        } else break;
      }
    }
  }
  return collides;
}

static int is_collision_free(planner *P, const double coeffs[6][3], double tf) {
  /* GetMonotonicSections, DIP.cpp:303-354 */
  double c[5];
  for (int i = 0; i < 5; i++) c[i] = (5 - i) * coeffs[i][2]; /* Trajectory.hpp:122-129 */
  double roots[6];
  roots[0] = 0;
  roots[1] = tf;
  unsigned rootCount;
  if (fabs(c[0]) > 1e-6) rootCount = ora_solve_quartic(c[1] / c[0], c[2] / c[0], c[3] / c[0], c[4] / c[0], roots + 2);
  else rootCount = ora_solve_cubic(c[2] / c[1], c[3] / c[1], c[4] / c[1], roots + 2);
  sort_doubles(roots, (int)rootCount + 2);
  mono sec[16];
  int ns = 0;
  for (unsigned i = 0; i < rootCount + 1; i++) {
    if (roots[i] < 0) continue;
    else if (fabs(roots[i] - roots[i + 1]) < 1e-6) continue;
    else if (roots[i] >= tf) break;
    if (roots[i + 1] <= tf) sec[ns++] = mono_make(coeffs, roots[i], roots[i + 1]);
    else break;
  }
  /* std::sort by deepest depth (insertion sort for such short ranges) */
  for (int i = 1; i < ns; i++) {
    mono v = sec[i];
    int j = i;
    while (j > 0 && mono_deepest(coeffs, &v) < mono_deepest(coeffs, &sec[j - 1])) { sec[j] = sec[j - 1]; j--; }
    sec[j] = v;
  }
  /* IsCollisionFree, DIP.cpp:214-301 (no wall-clock exits) */
  const ora_planner_config *cfg = P->cfg;
  while (ns > 0) {
    mono m = sec[--ns];
    double startP[3], endP[3];
    const double ts = m.increasing ? m.t0 : m.t1, te = m.increasing ? m.t1 : m.t0;
    for (int i = 0; i < 3; i++) { startP[i] = traj_axis(coeffs, i, ts); endP[i] = traj_axis(coeffs, i, te); }
    if (startP[2] < cfg->min_checking_dist && endP[2] < cfg->min_checking_dist) continue;
    const double px = endP[0] * cfg->focal_length / endP[2] + cfg->cx; /* DIP.hpp:287-290 */
    const double py = endP[1] * cfg->focal_length / endP[2] + cfg->cy;
    pyramid pyr;
    if (!find_containing_pyramid(P, px, py, endP[2], &pyr)) {
      if (P->n_pyr >= cfg->max_pyramids) return 0;
      if (!inflate_pyramid(P, (int)px, (int)py, endP[2], &pyr)) return 0;
      int idx = 0; /* std::lower_bound(begin, end, pyr): first element not less than pyr */
      while (idx < P->n_pyr && P->pyr[idx].depth < pyr.depth) idx++;
      if (P->n_pyr == P->cap_pyr) {
        P->cap_pyr = P->cap_pyr ? 2 * P->cap_pyr : 16;
        P->pyr = (pyramid *)realloc(P->pyr, sizeof(pyramid) * (size_t)P->cap_pyr);
      }
      memmove(&P->pyr[idx + 1], &P->pyr[idx], sizeof(pyramid) * (size_t)(P->n_pyr - idx));
      P->pyr[idx] = pyr;
      P->n_pyr++;
    }
    double tcol;
    if (deepest_collision_time(coeffs, &m, &pyr, &tcol)) {
      if (ns < 16) sec[ns++] = m.increasing ? mono_make(coeffs, m.t0, tcol) : mono_make(coeffs, tcol, m.t1);
      else return 0;
    }
  }
  return 1;
}

void ora_planner_default_config(ora_planner_config *c, int width, int height, double depth_scale,
                                double focal_length, double true_radius, double planning_radius,
                                double min_checking_dist) {
  memset(c, 0, sizeof(*c));
  c->width = width; c->height = height;
  c->depth_scale = depth_scale; c->focal_length = focal_length;
  c->cx = width / 2.0; c->cy = height / 2.0; /* main.cpp:484-488 */
  c->true_vehicle_radius = true_radius; c->planning_vehicle_radius = planning_radius;
  c->min_checking_dist = min_checking_dist;
  c->min_thrust = 5; c->max_thrust = 30; c->max_ang_vel = 20; c->max_velocity = 5; c->min_section_time = 0.02; /* DIP.cpp:43-50 */
  c->max_pyramids = 2147483647;
  c->pixel_buffer = 2; /* DIP.cpp:59 */
  c->cost_type = 0;
  c->cost_vec[2] = 1.0;
}

void ora_planner_run(const ora_planner_config *cfg, const uint16_t *depth, const double vel0[3],
                     const double acc0[3], const double grav[3], const double (*samples)[4],
                     int n_candidates, ora_plan_result *out, uint8_t *flags_out) {
  /* FindLowestCostTrajectory, DIP.cpp:91-212, with a candidate count instead of a time budget */
  planner P = {cfg, depth, 0, 0, 0};
  memset(out, 0, sizeof(*out));
  out->best_index = -1;
  double bestCost = DBL_MAX;
  rtg cand;
  memset(&cand, 0, sizeof(cand));
  for (int i = 0; i < 3; i++) { cand.ax[i].p0 = 0; cand.ax[i].v0 = vel0[i]; cand.ax[i].a0 = acc0[i]; cand.grav[i] = grav[i]; }
  for (int k = 0; k < n_candidates; k++) {
    /* GetNextCandidateTrajectory, DIP.hpp:393-404 */
    double posf[3];
    deproject(cfg, samples[k][0], samples[k][1], samples[k][2], posf);
    for (int i = 0; i < 3; i++) { cand.ax[i].pf = posf[i]; cand.ax[i].vf = 0; cand.ax[i].af = 0; }
    cand.tf = samples[k][3];
    for (int i = 0; i < 3; i++) ora_axis_generate(&cand.ax[i], cand.tf);
    out->n_generated++;
    /* cost */
    const double dur = cand.tf;
    const double ex = ora_axis_pos(&cand.ax[0], dur), ey = ora_axis_pos(&cand.ax[1], dur), ez = ora_axis_pos(&cand.ax[2], dur);
    double cost;
    if (cfg->cost_type == 0) { /* ExplorationCost::GetCost, DIP.hpp:488-492 */
      cost = -(cfg->cost_vec[0] * ex + cfg->cost_vec[1] * ey + cfg->cost_vec[2] * ez) / dur;
    } else { /* Simulator/Rappids_Simulator/main.cpp:86-107 */
      const double gx = cfg->cost_vec[0], gy = cfg->cost_vec[1], gz = cfg->cost_vec[2];
      const double SG = sqrt((gx - 0) * (gx - 0) + (gy - 0) * (gy - 0) + (gz - 0) * (gz - 0));
      const double PiG = sqrt((gx - ex) * (gx - ex) + (gy - ey) * (gy - ey) + (gz - ez) * (gz - ez));
      cost = -(SG - PiG) / dur;
    }
    unsigned result = 0;
    if (cost < bestCost) {
      result |= ORA_LOW_COST;
      out->n_cost_checks++;
      int res = rtg_input_section(&cand, cfg->min_thrust, cfg->max_thrust, cfg->max_ang_vel, 0, cand.tf, cfg->min_section_time);
      if (res == IN_FEASIBLE) {
        result |= ORA_DYN_FEASIBLE;
        out->n_collision_checks++;
        if (rtg_velocity_feasible(&cand, cfg->max_velocity)) {
          result |= ORA_VEL_ADMISSIBLE;
          out->n_velocity_checks++;
          double coeffs[6][3];
          rtg_trajectory(&cand, coeffs);
          if (is_collision_free(&P, coeffs, cand.tf)) {
            result |= ORA_COLLISION_FREE;
            out->found = 1;
            bestCost = cost;
            out->n_collision_free++;
            out->best_index = k;
            out->best_cost = cost;
            memcpy(out->coeffs, coeffs, sizeof(coeffs));
            out->tf = cand.tf;
          }
        }
      }
    }
    if (flags_out) flags_out[k] = (uint8_t)result;
  }
  out->n_pyramids = P.n_pyr;
  free(P.pyr);
}

/* ---- candidate sampling: libstdc++ mt19937 + uniform_real_distribution<double> ---- */
typedef struct { uint32_t mt[624]; int idx; } mt19937;
static void mt_seed(mt19937 *m, uint32_t seed) {
  m->mt[0] = seed;
  for (int i = 1; i < 624; i++) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
  m->idx = 624;
}
static uint32_t mt_next(mt19937 *m) {
  if (m->idx >= 624) {
    for (int i = 0; i < 624; i++) {
      uint32_t y = (m->mt[i] & 0x80000000u) | (m->mt[(i + 1) % 624] & 0x7fffffffu);
      m->mt[i] = m->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    m->idx = 0;
  }
  uint32_t y = m->mt[m->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
static double mt_canonical(mt19937 *m) { /* generate_canonical<double,53>(mt19937): two 32-bit draws */
  double sum = (double)mt_next(m);
  sum += (double)mt_next(m) * 4294967296.0;
  double ret = sum / 18446744073709551616.0;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}
static double mt_uniform(mt19937 *m, double a, double b) { return (mt_canonical(m) * (b - a)) + a; }

void ora_planner_samples(uint32_t seed, int width, int height, int n, double (*samples)[4]) {
  /* RandomTrajectoryGenerator default ctor, DIP.hpp:349-366; g++ evaluates the three
   * arguments of DeprojectPixelToPoint right to left: depth, pixelY, pixelX */
  mt19937 m;
  mt_seed(&m, seed);
  for (int k = 0; k < n; k++) {
    samples[k][2] = mt_uniform(&m, 1.5, 3.0);
    samples[k][1] = mt_uniform(&m, 0.1 * height, 0.9 * height);
    samples[k][0] = mt_uniform(&m, 0.1 * width, 0.9 * width);
    samples[k][3] = mt_uniform(&m, 2.0, 3.0);
  }
}

int ora_planner_sampled_collision(const ora_planner_config *cfg, const uint16_t *depth, const double coeffs[6][3],
                                  double tf, int n_samples) {
  /* independent dense check: does the planning-radius sphere at sampled times poke in front of any
   * depth pixel it covers?  returns 1 if a collision is seen */
  for (int s = 0; s <= n_samples; s++) {
    const double t = tf * s / n_samples;
    const double x = traj_axis(coeffs, 0, t), y = traj_axis(coeffs, 1, t), z = traj_axis(coeffs, 2, t);
    if (z < cfg->min_checking_dist) continue;
    const double r = cfg->planning_vehicle_radius;
    const int u0 = (int)floor((x - r) * cfg->focal_length / z + cfg->cx), u1 = (int)ceil((x + r) * cfg->focal_length / z + cfg->cx);
    const int v0 = (int)floor((y - r) * cfg->focal_length / z + cfg->cy), v1 = (int)ceil((y + r) * cfg->focal_length / z + cfg->cy);
    for (int v = v0; v <= v1; v++)
      for (int u = u0; u <= u1; u++) {
        if (u < 0 || v < 0 || u >= cfg->width || v >= cfg->height) return 1;
        const double d = depth[v * cfg->width + u] * cfg->depth_scale;
        if (d > cfg->true_vehicle_radius && d < z - r) return 1;
      }
  }
  return 0;
}
