// afe_planner.hip -- batched RAPPIDS depth-image planner for gfx950 (SURVEY.md 8f
// row f3): one planner instance per vehicle, a fixed number of candidate
// motion primitives per plan instead of the reference's wall-clock budget.
//
// Reference (agri-fly tree): Components/Components/DepthImagePlanner/
// DepthImagePlanner.{hpp,cpp} ("DIP"), Pyramid.hpp, MonotonicTrajectory.hpp,
// Components/Components/TrajectoryGenerator/{RapidTrajectoryGenerator,
// SingleAxisTrajectory}.{hpp,cpp} ("RTG", "SAT"), Common/Common/Math/
// {RootFinder,Trajectory}.hpp.
//
// First correct form (round 1): ONE LANE = ONE PLANNER.  The search is
// sequential by construction -- a candidate is only examined if it beats the best
// cost so far, and every collision check reads and grows the plan's sorted pyramid
// list (DIP.cpp:143-190,214-301) -- so the parallel axis is the ensemble.  The
// pyramid list lives in a per-planner slab in HBM; the depth image is read through
// the caches (planners sharing an image share its lines).  All arithmetic is
// double, as in the reference, with FMA contraction off so the polynomial
// coefficients are bit-identical to a CPU evaluation; only acos/cos/pow differ
// from libm by an ulp.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "afe_planner.h"

namespace afe {
namespace {

#define PL_MIN(a, b) (((b) < (a)) ? (b) : (a))  // std::min / std::max semantics
#define PL_MAX(a, b) (((a) < (b)) ? (b) : (a))

// ---- RootFinder.hpp -------------------------------------------------------
// :40-44 the constants are floats in the reference
#define PL_2PI ((double)(float)(2 * (float)3.141592653589793238463))
#define PL_EPS ((double)(float)1e-12)

__device__ unsigned solve_cubic(double a, double b, double c, double *x) {
#pragma clang fp contract(off)
  const double a2 = a * a;
  double q = (a2 - 3 * b) / 9;
  const double r = (a * (2 * a2 - 9 * b) + 27 * c) / 54;
  const double r2 = r * r;
  const double q3 = q * q * q;
  if (r2 < q3) {
    double t = r / sqrt(q3);
    if (t < -1) t = -1;
    if (t > 1) t = 1;
    t = acos(t);
    a /= 3;
    q = -2 * sqrt(q);
    x[0] = q * cos(t / 3) - a;
    x[1] = q * cos((t + PL_2PI) / 3.0) - a;
    x[2] = q * cos((t - PL_2PI) / 3.0) - a;
    return 3;
  }
  double A = -pow(fabs(r) + sqrt(r2 - q3), 1. / 3);
  if (r < 0) A = -A;
  const double B = (fabs(A) < PL_EPS ? 0 : q / A);
  a /= 3;
  x[0] = (A + B) - a;
  x[1] = -0.5 * (A + B) - a;
  x[2] = 0.5 * sqrt(3.) * (A - B);
  if (fabs(x[2]) < PL_EPS) {
    x[2] = x[1];
    return 2;
  }
  return 1;
}

__device__ unsigned solve_quartic(double a, double b, double c, double d, double *root) {
#pragma clang fp contract(off)
  double x3[3];
  const unsigned iZeroes = solve_cubic(-b, a * c - 4. * d, -a * a * d - c * c + 4. * b * d, x3);
  double y = x3[0];
  if (iZeroes != 1) {
    if (fabs(x3[1]) > fabs(y)) y = x3[1];
    if (fabs(x3[2]) > fabs(y)) y = x3[2];
  }
  double q1, q2, p1, p2;
  double D = y * y - 4 * d;
  if (fabs(D) < PL_EPS) {
    q1 = q2 = y * 0.5;
    D = a * a - 4. * (b - y);
    if (fabs(D) < PL_EPS) {
      p1 = p2 = a * 0.5;
    } else {
      const double s = sqrt(D);
      p1 = (a + s) * 0.5;
      p2 = (a - s) * 0.5;
    }
  } else {
    const double s = sqrt(D);
    q1 = (y + s) * 0.5;
    q2 = (y - s) * 0.5;
    p1 = (a * q1 - c) / (q1 - q2);
    p2 = (c - a * q2) / (q1 - q2);
  }
  unsigned n = 0;
  D = p1 * p1 - 4 * q1;
  if (!(D < 0.0)) {
    const double s = sqrt(D);
    root[n++] = (-p1 + s) * 0.5;
    root[n++] = (-p1 - s) * 0.5;
  }
  D = p2 * p2 - 4 * q2;
  if (!(D < 0.0)) {
    const double s = sqrt(D);
    root[n++] = (-p2 + s) * 0.5;
    root[n++] = (-p2 - s) * 0.5;
  }
  return n;
}

__device__ void sort_small(double *a, int n) {
  for (int i = 1; i < n; i++) {
    const double v = a[i];
    int j = i;
    while (j > 0 && v < a[j - 1]) { a[j] = a[j - 1]; j--; }
    a[j] = v;
  }
}

// ---- one candidate motion primitive (RTG + 3 x SAT) ------------------------
struct Cand {
  double v0[3], a0[3], grav[3];     // initial state (p0 = 0: camera-fixed frame) and gravity
  double al[3], be[3], ga[3];       // SAT _a, _b, _g per axis
  double peak[3][2];                // SAT _accPeakTimes
  double tf;
};

__device__ double c_acc(const Cand &k, int i, double t) {  // SAT.hpp GetAcceleration
#pragma clang fp contract(off)
  return k.a0[i] + k.ga[i] * t + (1 / 2.0) * k.be[i] * t * t + (1 / 6.0) * k.al[i] * t * t * t;
}
__device__ double c_vel(const Cand &k, int i, double t) {
#pragma clang fp contract(off)
  return k.v0[i] + k.a0[i] * t + (1 / 2.0) * k.ga[i] * t * t + (1 / 6.0) * k.be[i] * t * t * t +
         (1 / 24.0) * k.al[i] * t * t * t * t;
}
__device__ double c_pos(const Cand &k, int i, double t) {
#pragma clang fp contract(off)
  return 0.0 + k.v0[i] * t + (1 / 2.0) * k.a0[i] * t * t + (1 / 6.0) * k.ga[i] * t * t * t +
         (1 / 24.0) * k.be[i] * t * t * t * t + (1 / 120.0) * k.al[i] * t * t * t * t * t;
}
__device__ double c_jerk(const Cand &k, int i, double t) {
#pragma clang fp contract(off)
  return k.ga[i] + k.be[i] * t + (1 / 2.0) * k.al[i] * t * t;
}

// SAT.cpp:59-107 (goal position, velocity and acceleration all defined; goal velocity
// and acceleration are zero for every RAPPIDS candidate, DIP.hpp:398-401) and the
// acceleration peak times of SAT.cpp:119-140
__device__ void c_generate(Cand &k, const double pf[3], double Tf) {
#pragma clang fp contract(off)
  const double T2 = Tf * Tf, T3 = T2 * Tf, T4 = T3 * Tf, T5 = T4 * Tf;
  for (int i = 0; i < 3; i++) {
    const double da = 0.0 - k.a0[i];
    const double dv = 0.0 - k.v0[i] - k.a0[i] * Tf;
    const double dp = pf[i] - 0.0 - k.v0[i] * Tf - 0.5 * k.a0[i] * Tf * Tf;
    k.al[i] = (60 * T2 * da - 360 * Tf * dv + 720 * 1 * dp) / T5;
    k.be[i] = (-24 * T3 * da + 168 * T2 * dv - 360 * Tf * dp) / T5;
    k.ga[i] = (3 * T4 * da - 24 * T3 * dv + 60 * T2 * dp) / T5;
    if (k.al[i]) {
      const double det = k.be[i] * k.be[i] - 2 * k.ga[i] * k.al[i];
      if (det < 0) {
        k.peak[i][0] = 0;
        k.peak[i][1] = 0;
      } else {
        k.peak[i][0] = (-k.be[i] + sqrt(det)) / k.al[i];
        k.peak[i][1] = (-k.be[i] - sqrt(det)) / k.al[i];
      }
    } else {
      k.peak[i][0] = k.be[i] ? -k.ga[i] / k.be[i] : 0;
      k.peak[i][1] = 0;
    }
  }
  k.tf = Tf;
}

__device__ double c_thrust(const Cand &k, double t) {  // RTG.hpp GetThrust
#pragma clang fp contract(off)
  const double x = c_acc(k, 0, t) - k.grav[0], y = c_acc(k, 1, t) - k.grav[1], z = c_acc(k, 2, t) - k.grav[2];
  return sqrt(x * x + y * y + z * z);
}

enum { SEC_FEASIBLE = 0, SEC_INDETERMINABLE = 1, SEC_HIGH = 2, SEC_LOW = 3, SEC_SPLIT = 4 };

// one level of RTG.cpp:75-150 without the recursion: SEC_SPLIT means "indeterminate,
// bisect" (:130-145)
__device__ int input_section(const Cand &k, const PlannerConfig &cfg, double t1, double t2) {
#pragma clang fp contract(off)
  if (t2 - t1 < cfg.min_section_time) return SEC_INDETERMINABLE;
  const double f1 = c_thrust(k, t1), f2 = c_thrust(k, t2);
  if (PL_MAX(f1, f2) > cfg.max_thrust) return SEC_HIGH;
  if (PL_MIN(f1, f2) < cfg.min_thrust) return SEC_LOW;
  double fminSqr = 0, fmaxSqr = 0, jmaxSqr = 0;
  for (int i = 0; i < 3; i++) {
    // SAT.cpp:142-154 GetMinMaxAcc
    const double e1 = c_acc(k, i, t1), e2 = c_acc(k, i, t2);
    double amin = PL_MIN(e1, e2), amax = PL_MAX(e1, e2);
    for (int p = 0; p < 2; p++) {
      const double tp = k.peak[i][p];
      if (tp <= t1) continue;
      if (tp >= t2) continue;
      const double ap = c_acc(k, i, tp);
      amin = PL_MIN(amin, ap);
      amax = PL_MAX(amax, ap);
    }
    const double v1 = amin - k.grav[i], v2 = amax - k.grav[i];
    if (PL_MAX(v1 * v1, v2 * v2) > cfg.max_thrust * cfg.max_thrust) return SEC_HIGH;
    if (v1 * v2 < 0) fminSqr += 0;
    else { const double m = PL_MIN(fabs(v1), fabs(v2)); fminSqr += m * m; }
    { const double m = PL_MAX(fabs(v1), fabs(v2)); fmaxSqr += m * m; }
    // SAT.cpp:164-176 GetMaxJerkSquared
    const double j1 = c_jerk(k, i, t1), j2 = c_jerk(k, i, t2);
    double jm = PL_MAX(j1 * j1, j2 * j2);
    if (k.al[i]) {
      const double tMax = -k.be[i] / k.al[i];
      if (tMax > t1 && tMax < t2) { const double jp = c_jerk(k, i, tMax); jm = PL_MAX(jp * jp, jm); }
    }
    jmaxSqr += jm;
  }
  const double fmin = sqrt(fminSqr), fmax = sqrt(fmaxSqr);
  const double wBound = (fminSqr > 1e-6) ? sqrt(jmaxSqr / fminSqr) : 1.7976931348623157e308;
  if (fmax < cfg.min_thrust) return SEC_LOW;
  if (fmin > cfg.max_thrust) return SEC_HIGH;
  if (fmin < cfg.min_thrust || fmax > cfg.max_thrust || wBound > cfg.max_ang_vel) return SEC_SPLIT;
  return SEC_FEASIBLE;
}

// RTG.cpp:152-161 CheckInputFeasibility: depth-first bisection, left half first, stop at
// the first section that is not feasible -- the recursion of :130-145 as a stack of
// pending right halves
__device__ bool input_feasible(const Cand &k, const PlannerConfig &cfg) {
  double stack_t1[24], stack_t2[24];
  int sp = 0;
  stack_t1[0] = 0;
  stack_t2[0] = k.tf;
  sp = 1;
  while (sp > 0) {
    --sp;
    double t1 = stack_t1[sp], t2 = stack_t2[sp];
    for (;;) {
      const int r = input_section(k, cfg, t1, t2);
      if (r == SEC_FEASIBLE) break;
      if (r != SEC_SPLIT) return false;
      const double tHalf = (t1 + t2) / 2;
      if (sp >= 24) return false;
      stack_t1[sp] = tHalf;   // second half waits
      stack_t2[sp] = t2;
      sp++;
      t2 = tHalf;             // descend into the first half
    }
  }
  return true;
}

// RTG.cpp:163-208
__device__ bool velocity_feasible(const Cand &k, double vmax) {
#pragma clang fp contract(off)
  for (int dim = 0; dim < 3; dim++) {
    const double c0 = k.al[dim] / 6.0, c1 = k.be[dim] / 2.0, c2 = k.ga[dim] / 1.0, c3 = k.a0[dim];
    double roots[5];
    unsigned n;
    if (fabs(c0) > 1e-6) n = solve_cubic(c1 / c0, c2 / c0, c3 / c0, roots);
    else return false;
    roots[n] = 0;
    roots[n + 1] = k.tf;
    for (unsigned i = 0; i < n + 2; i++) {
      if (roots[i] < 0) continue;
      if (roots[i] > k.tf) continue;
      if (fabs(c_vel(k, 0, roots[i])) >= vmax || fabs(c_vel(k, 1, roots[i])) >= vmax ||
          fabs(c_vel(k, 2, roots[i])) >= vmax)
        return false;
    }
  }
  return true;
}

// ---- CommonMath::Trajectory: c[0] t^5 + ... + c[5] --------------------------
struct Poly {
  double c[6][3];
};
__device__ double p_axis(const Poly &p, int i, double t) {  // Trajectory.hpp:90-96
#pragma clang fp contract(off)
  return p.c[0][i] * t * t * t * t * t + p.c[1][i] * t * t * t * t + p.c[2][i] * t * t * t + p.c[3][i] * t * t +
         p.c[4][i] * t + p.c[5][i];
}
struct Section {  // MonotonicTrajectory
  double t0, t1;
  bool increasing;
};
__device__ Section make_section(const Poly &p, double t0, double t1) {
  Section s = {t0, t1, false};
  s.increasing = p_axis(p, 2, t0) < p_axis(p, 2, t1);
  return s;
}
__device__ double deepest(const Poly &p, const Section &s) { return s.increasing ? p_axis(p, 2, s.t1) : p_axis(p, 2, s.t0); }

// ---- pyramids ---------------------------------------------------------------
__device__ void unit_normal(const double a[3], const double b[3], double o[3]) {
#pragma clang fp contract(off)
  // Pyramid.hpp:52-57; Vec3::GetUnitVector narrows the norm to float (Vec3.hpp:126-129)
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  const float n = (float)sqrt(x * x + y * y + z * z);
  o[0] = x / n; o[1] = y / n; o[2] = z / n;
}
__device__ void deproject(const PlannerConfig &c, double x, double y, double depth, double o[3]) {
#pragma clang fp contract(off)
  o[0] = depth * ((x - c.cx) / c.focal_length);   // DIP.hpp:274-279
  o[1] = depth * ((y - c.cy) / c.focal_length);
  o[2] = depth * 1;
}

// shrink bookkeeping shared by the eight scans of DIP.cpp:617-940
struct Shrink {
  int right, left, top, bottom;
};

// DIP.cpp:456-970
__device__ bool inflate_pyramid(const PlannerConfig &c, const uint16_t *__restrict__ img, int x0, int y0,
                                double minimumDepth, PlannerPyramid &out) {
#pragma clang fp contract(off)
  const int W = c.width, H = c.height, buf = c.pixel_buffer;
  const int edgeOff = (int)(c.focal_length * c.true_vehicle_radius / c.min_checking_dist);
  if (x0 <= edgeOff + buf + 1 || x0 > W - edgeOff - buf - 1 || y0 <= edgeOff + buf + 1 || y0 > H - edgeOff - buf - 1)
    return false;
  const uint16_t minDepthPix = (uint16_t)((minimumDepth + c.planning_vehicle_radius) / c.depth_scale);
  const int initR = (int)(c.focal_length * c.planning_vehicle_radius / (c.depth_scale * minDepthPix));
  if (2 * initR >= (W < H ? W : H) - 2 * edgeOff) return false;
  int L, T, R, B;
  if (y0 - initR < edgeOff) { T = edgeOff; B = T + 2 * initR; }
  else { B = PL_MIN(H - edgeOff - 1, y0 + initR); T = B - 2 * initR; }
  if (x0 - initR < edgeOff) { L = edgeOff; R = L + 2 * initR; }
  else { R = PL_MIN(W - edgeOff - 1, x0 + initR); L = R - 2 * initR; }
  const uint16_t ignore = (uint16_t)(c.true_vehicle_radius / c.depth_scale);
  for (int y = T; y < B; y++)
    for (int x = L; x < R; x++) {
      const uint16_t d = img[y * W + x];
      if (d <= minDepthPix && d > ignore) return false;
    }
  // spiral expansion, :520-600
  uint16_t maxDepth = 65535;
  bool rFree = true, tFree = true, lFree = true, bFree = true;
  while (rFree || tFree || lFree || bFree) {
    if (rFree) {
      if (R < W - edgeOff - 1) {
        for (int y = T; y <= B; y++) {
          const uint16_t d = img[y * W + R + 1];
          if (d > ignore) {
            if (d < minDepthPix) { rFree = false; R--; break; }
            maxDepth = PL_MIN(maxDepth, d);
          }
        }
        R++;
      } else rFree = false;
    }
    if (tFree) {
      if (T > edgeOff) {
        for (int x = L; x <= R; x++) {
          const uint16_t d = img[(T - 1) * W + x];
          if (d > ignore) {
            if (d < minDepthPix) { tFree = false; T++; break; }
            maxDepth = PL_MIN(maxDepth, d);
          }
        }
        T--;
      } else tFree = false;
    }
    if (lFree) {
      if (L > edgeOff) {
        for (int y = T; y <= B; y++) {
          const uint16_t d = img[y * W + L - 1];
          if (d > ignore) {
            if (d < minDepthPix) { lFree = false; L++; break; }
            maxDepth = PL_MIN(maxDepth, d);
          }
        }
        L--;
      } else lFree = false;
    }
    if (bFree) {
      if (B < H - edgeOff - 1) {
        for (int x = L; x <= R; x++) {
          const uint16_t d = img[(B + 1) * W + x];
          if (d > ignore) {
            if (d < minDepthPix) { bFree = false; B--; break; }
            maxDepth = PL_MIN(maxDepth, d);
          }
        }
        B++;
      } else bFree = false;
    }
  }
  // shrink by the vehicle radius, :602-940
  Shrink s = {W - 1 - edgeOff, edgeOff, edgeOff, H - 1 - edgeOff};
  const int num = (int)(c.focal_length * c.planning_vehicle_radius / c.depth_scale);
#define PL_PIX(x, y) const uint16_t d = img[(y) * W + (x)]; if (d > ignore && d < maxDepth)
  for (int x = R; x < W; x++)            // right side, :617-661
    for (int y = T; y <= B; y++) {
      PL_PIX(x, y) {
        if (num > (x - s.right) * d) {
          const int rT = x - (int)(num / d);
          if (x0 > rT - buf) {
            const int tT = y + (int)(num / d), bT = y - (int)(num / d);
            if (y0 < tT + buf && y0 > bT - buf) return false;
            else if (y0 < tT + buf) s.bottom = bT;
            else if (y0 > bT - buf) s.top = tT;
            else if ((s.bottom - bT) > (tT - s.top)) s.top = tT;
            else s.right = bT;            // sic, DIP.cpp:648
          } else s.right = rT;
        }
      }
    }
  for (int x = L; x >= 0; x--)           // left side, :663-698
    for (int y = T; y <= B; y++) {
      PL_PIX(x, y) {
        if ((s.left - x) * d < num) {
          const int lT = x + (int)(num / d);
          if (x0 < lT + buf) {
            const int tT = y + (int)(num / d), bT = y - (int)(num / d);
            if (y0 < tT + buf && y0 > bT - buf) return false;
            else if (y0 < tT + buf) s.bottom = bT;
            else if (y0 > bT - buf) s.top = tT;
            else if ((s.bottom - bT) > (tT - s.top)) s.top = tT;
            else s.bottom = bT;
          } else s.left = lT;
        }
      }
    }
  if (s.left + buf > s.right - buf) return false;
  for (int y = T; y >= 0; y--)           // top side, :705-744
    for (int x = L; x <= R; x++) {
      PL_PIX(x, y) {
        if ((s.top - y) * d < num) {
          const int tT = y + (int)(num / d);
          if (y0 < tT + buf) {
            const int rT = x - (int)(num / d), lT = x + (int)(num / d);
            if (x0 > rT - buf && x0 < lT + buf) return false;
            else if (x0 > rT - buf) s.left = lT;
            else if (x0 < lT + buf) s.right = rT;
            else if ((s.right - rT) > (lT - s.left)) s.left = lT;
            else s.right = rT;
          } else s.top = tT;
        }
      }
    }
  for (int y = B; y < H; y++)            // bottom side, :746-785
    for (int x = L; x <= R; x++) {
      PL_PIX(x, y) {
        if (num > (y - s.bottom) * d) {
          const int bT = y - (int)(num / d);
          if (y0 > bT - buf) {
            const int rT = x - (int)(num / d), lT = x + (int)(num / d);
            if (x0 > rT - buf && x0 < lT + buf) return false;
            else if (x0 > rT - buf) s.left = lT;
            else if (x0 < lT + buf) s.right = rT;
            else if ((s.right - rT) > (lT - s.left)) s.left = lT;
            else s.right = rT;
          } else s.bottom = bT;
        }
      }
    }
  if (s.top + buf > s.bottom - buf) return false;
  for (int y = T; y >= 0; y--)           // top right corner, :794-829
    for (int x = R; x < W; x++) {
      PL_PIX(x, y) {
        if (num > (x - s.right) * d && (s.top - y) * d < num) {
          const int rT = x - (int)(num / d), tT = y + (int)(num / d);
          if (x0 > rT - buf && y0 < tT + buf) return false;
          else if (x0 > rT - buf) s.top = tT;
          else if (y0 < tT + buf) s.right = rT;
          else if ((s.right - rT) * (s.bottom - s.top) > (tT - s.top) * (s.right - s.left)) s.top = tT;
          else s.right = rT;
        }
      }
    }
  for (int y = B; y < H; y++)            // bottom right corner, :831-866
    for (int x = R; x < W; x++) {
      PL_PIX(x, y) {
        if (num > (x - s.right) * d && num > (y - s.bottom) * d) {
          const int rT = x - (int)(num / d), bT = y - (int)(num / d);
          if (x0 > rT - buf && y0 > bT - buf) return false;
          else if (x0 > rT - buf) s.bottom = bT;
          else if (y0 > bT - buf) s.right = rT;
          else if ((s.right - rT) * (s.bottom - s.top) > (s.bottom - bT) * (s.right - s.left)) s.bottom = bT;
          else s.right = rT;
        }
      }
    }
  for (int y = T; y >= 0; y--)           // top left corner, :868-903
    for (int x = L; x >= 0; x--) {
      PL_PIX(x, y) {
        if ((s.left - x) * d < num && (s.top - y) * d < num) {
          const int lT = x + (int)(num / d), tT = y + (int)(num / d);
          if (x0 < lT + buf && y0 < tT + buf) return false;
          else if (x0 < lT + buf) s.top = tT;
          else if (y0 < tT + buf) s.left = lT;
          else if ((lT - s.left) * (s.bottom - s.top) > (tT - s.top) * (s.right - s.left)) s.top = tT;
          else s.left = lT;
        }
      }
    }
  for (int y = B; y < H; y++)            // bottom left corner, :905-940
    for (int x = L; x >= 0; x--) {
      PL_PIX(x, y) {
        if ((s.left - x) * d < num && num > (y - s.bottom) * d) {
          const int lT = x + (int)(num / d), bT = y - (int)(num / d);
          if (x0 < lT + buf && y0 > bT - buf) return false;
          else if (x0 < lT + buf) s.bottom = bT;
          else if (y0 > bT - buf) s.left = lT;
          else if ((lT - s.left) * (s.bottom - s.top) > (s.bottom - bT) * (s.right - s.left)) s.bottom = bT;
          else s.left = lT;
        }
      }
    }
#undef PL_PIX
  // :942-966
  const double depth = maxDepth * c.depth_scale - c.planning_vehicle_radius;
  double k0[3], k1[3], k2[3], k3[3];
  deproject(c, (double)s.right, (double)s.top, depth, k0);
  deproject(c, (double)s.left, (double)s.top, depth, k1);
  deproject(c, (double)s.left, (double)s.bottom, depth, k2);
  deproject(c, (double)s.right, (double)s.bottom, depth, k3);
  out.depth = depth;
  out.right = s.right; out.top = s.top; out.left = s.left; out.bottom = s.bottom;
  unit_normal(k0, k1, out.normal[0]);
  unit_normal(k1, k2, out.normal[1]);
  unit_normal(k2, k3, out.normal[2]);
  unit_normal(k3, k0, out.normal[3]);
  return true;
}

// DIP.cpp:382-454
__device__ bool deepest_collision_time(const Poly &p, const Section &m, const PlannerPyramid &pyr, double &tOut) {
#pragma clang fp contract(off)
  bool collides = false;
  tOut = m.increasing ? m.t0 : m.t1;
  for (int f = 0; f < 4; f++) {
    double c[5] = {0, 0, 0, 0, 0};
    for (int dim = 0; dim < 3; dim++)
      for (int q = 0; q < 5; q++) c[q] += pyr.normal[f][dim] * p.c[q][dim];
    double roots[4];
    unsigned n;
    if (fabs(c[0]) > 1e-6) n = solve_quartic(c[1] / c[0], c[2] / c[0], c[3] / c[0], c[4] / c[0], roots);
    else n = solve_cubic(c[2] / c[1], c[3] / c[1], c[4] / c[1], roots);
    sort_small(roots, (int)n);
    if (m.increasing) {
      for (int i = (int)n - 1; i >= 0; i--) {
        if (roots[i] > m.t1) continue;
        else if (roots[i] > m.t0) {
          if (roots[i] > tOut) { tOut = roots[i]; collides = true; break; }
        } else break;
      }
    } else {
      for (int i = 0; i < (int)n; i++) {
        if (roots[i] < m.t0) continue;
        else if (roots[i] < m.t1) {
          if (roots[i] < tOut) { tOut = roots[i]; collides = true; break; }
        } else break;
      }
    }
  }
  return collides;
}

// GetMonotonicSections (DIP.cpp:303-354) + IsCollisionFree (:214-301)
__device__ bool collision_free(const PlannerConfig &cfg, const uint16_t *__restrict__ img, const Poly &p, double tf,
                               PlannerPyramid *pyr, int &nPyr, int maxPyr) {
#pragma clang fp contract(off)
  double dc[5];
  for (int i = 0; i < 5; i++) dc[i] = (5 - i) * p.c[i][2];
  double roots[6];
  roots[0] = 0;
  roots[1] = tf;
  unsigned n;
  if (fabs(dc[0]) > 1e-6) n = solve_quartic(dc[1] / dc[0], dc[2] / dc[0], dc[3] / dc[0], dc[4] / dc[0], roots + 2);
  else n = solve_cubic(dc[2] / dc[1], dc[3] / dc[1], dc[4] / dc[1], roots + 2);
  sort_small(roots, (int)n + 2);
  Section sec[8];
  int ns = 0;
  for (unsigned i = 0; i < n + 1; i++) {
    if (roots[i] < 0) continue;
    else if (fabs(roots[i] - roots[i + 1]) < 1e-6) continue;
    else if (roots[i] >= tf) break;
    if (roots[i + 1] <= tf) sec[ns++] = make_section(p, roots[i], roots[i + 1]);
    else break;
  }
  for (int i = 1; i < ns; i++) {  // std::sort by deepest point, ascending
    const Section v = sec[i];
    int j = i;
    while (j > 0 && deepest(p, v) < deepest(p, sec[j - 1])) { sec[j] = sec[j - 1]; j--; }
    sec[j] = v;
  }
  while (ns > 0) {
    const Section m = sec[--ns];
    const double ts = m.increasing ? m.t0 : m.t1, te = m.increasing ? m.t1 : m.t0;
    const double sz = p_axis(p, 2, ts);
    const double ex = p_axis(p, 0, te), ey = p_axis(p, 1, te), ez = p_axis(p, 2, te);
    if (sz < cfg.min_checking_dist && ez < cfg.min_checking_dist) continue;
    const double px = ex * cfg.focal_length / ez + cfg.cx;   // DIP.hpp:287-290
    const double py = ey * cfg.focal_length / ez + cfg.cy;
    // FindContainingPyramid, DIP.cpp:356-380
    int at = -1;
    {
      int first = 0;
      while (first < nPyr && pyr[first].depth < ez) first++;
      for (int q = first; q < nPyr; q++)
        if (pyr[q].left + cfg.pixel_buffer < px && px < pyr[q].right - cfg.pixel_buffer &&
            pyr[q].top + cfg.pixel_buffer < py && py < pyr[q].bottom - cfg.pixel_buffer) { at = q; break; }
    }
    if (at < 0) {
      if (nPyr >= maxPyr) return false;                        // _maxNumPyramids, :255-260
      PlannerPyramid fresh;
      if (!inflate_pyramid(cfg, img, (int)px, (int)py, ez, fresh)) return false;
      int idx = 0;                                             // std::lower_bound + insert, :269-271
      while (idx < nPyr && pyr[idx].depth < fresh.depth) idx++;
      for (int q = nPyr; q > idx; q--) pyr[q] = pyr[q - 1];
      pyr[idx] = fresh;
      nPyr++;
      at = idx;
    }
    double tcol;
    if (deepest_collision_time(p, m, pyr[at], tcol)) {
      if (ns >= 8) return false;
      sec[ns++] = m.increasing ? make_section(p, m.t0, tcol) : make_section(p, tcol, m.t1);
    }
  }
  return true;
}

}  // namespace

// FindLowestCostTrajectory, DIP.cpp:91-212, candidate count instead of a time budget
__global__ void __launch_bounds__(64) afe_rappids_kernel(const PlannerConfig cfg, const PlannerBatch b) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= b.n) return;
  const uint16_t *img = b.images + (int64_t)(b.image_index ? b.image_index[i] : i) * cfg.width * cfg.height;
  const double *samples = b.samples + (int64_t)(b.sample_table ? b.sample_table[i] : 0) * b.n_candidates * 4;
  PlannerPyramid *pyr = b.pyramids + i * b.max_pyramids;
  int nPyr = 0;
  Cand k;
  double cost_vec[3];
  for (int a = 0; a < 3; a++) {
    k.v0[a] = b.vel0[a * b.n + i];
    k.a0[a] = b.acc0[a * b.n + i];
    k.grav[a] = b.grav[a * b.n + i];
    cost_vec[a] = b.cost_vec ? b.cost_vec[a * b.n + i] : cfg.cost_vec[a];
  }
  PlanOutput o;
  o.found = 0; o.best_index = -1; o.best_cost = 1.7976931348623157e308; o.tf = 0;
  o.n_generated = o.n_cost_checks = o.n_collision_checks = o.n_velocity_checks = o.n_collision_free = 0;
  for (int q = 0; q < 6; q++) for (int a = 0; a < 3; a++) o.coeffs[q][a] = 0;
  double bestCost = 1.7976931348623157e308;
  for (int c = 0; c < b.n_candidates; c++) {
    double pf[3];
    deproject(cfg, samples[4 * c + 0], samples[4 * c + 1], samples[4 * c + 2], pf);   // DIP.hpp:393-404
    c_generate(k, pf, samples[4 * c + 3]);
    o.n_generated++;
    const double dur = k.tf;
    const double ex = c_pos(k, 0, dur), ey = c_pos(k, 1, dur), ez = c_pos(k, 2, dur);
    double cost;
    if (cfg.cost_type == 0) {        // ExplorationCost::GetCost, DIP.hpp:488-492
      cost = -(cost_vec[0] * ex + cost_vec[1] * ey + cost_vec[2] * ez) / dur;
    } else {                         // Simulator/Rappids_Simulator/main.cpp:86-107
      const double gx = cost_vec[0], gy = cost_vec[1], gz = cost_vec[2];
      const double SG = sqrt((gx - 0) * (gx - 0) + (gy - 0) * (gy - 0) + (gz - 0) * (gz - 0));
      const double PiG = sqrt((gx - ex) * (gx - ex) + (gy - ey) * (gy - ey) + (gz - ez) * (gz - ez));
      cost = -(SG - PiG) / dur;
    }
    unsigned result = 0;
    if (cost < bestCost) {
      result |= 1;
      o.n_cost_checks++;
      if (input_feasible(k, cfg)) {
        result |= 2;
        o.n_collision_checks++;
        if (velocity_feasible(k, cfg.max_velocity)) {
          result |= 4;
          o.n_velocity_checks++;
          Poly p;                    // RTG.hpp GetTrajectory
          for (int a = 0; a < 3; a++) {
            p.c[0][a] = k.al[a] / 120;
            p.c[1][a] = k.be[a] / 24;
            p.c[2][a] = k.ga[a] / 6;
            p.c[3][a] = c_acc(k, a, 0) / 2;
            p.c[4][a] = c_vel(k, a, 0);
            p.c[5][a] = c_pos(k, a, 0);
          }
          if (collision_free(cfg, img, p, k.tf, pyr, nPyr, b.max_pyramids)) {
            result |= 8;
            bestCost = cost;
            o.found = 1;
            o.n_collision_free++;
            o.best_index = c;
            o.best_cost = cost;
            o.tf = k.tf;
            for (int q = 0; q < 6; q++) for (int a = 0; a < 3; a++) o.coeffs[q][a] = p.c[q][a];
          }
        }
      }
    }
    if (b.flags) b.flags[i * b.n_candidates + c] = (uint8_t)result;
  }
  o.n_pyramids = nPyr;
  b.out[i] = o;
}

int launch_rappids(const PlannerConfig &cfg, const PlannerBatch &b, void *stream) {
  if (b.n <= 0) return 0;
  hipLaunchKernelGGL(afe_rappids_kernel, dim3((unsigned)((b.n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, cfg, b);
  return (int)hipGetLastError();
}

}  // namespace afe
