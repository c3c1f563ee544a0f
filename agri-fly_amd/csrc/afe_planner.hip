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
// The search is sequential by construction -- a candidate is only examined if it beats
// the best cost so far, every collision check reads and grows the plan's sorted pyramid
// list (DIP.cpp:143-190,214-301), every pixel test of InflatePyramid reads edges that
// earlier pixels moved -- so the reference's decision order is kept and the parallelism
// sits underneath it: candidates are pre-evaluated one per lane, then ONE WAVE RUNS ONE
// PLANNER, with the per-pixel work spread over its 64 lanes (see "bit images in LDS" and
// "wave-cooperative pixel scans" below).  The pyramid list lives in a per-planner slab in
// HBM.  All arithmetic is double, as in the reference, with FMA contraction off so the
// polynomial coefficients are bit-identical to a CPU evaluation; only acos/cos/pow differ
// from libm by an ulp.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>

#include "afe_planner.h"
#include "afe_host.h"   // afe_dev_env

namespace afe {
namespace {

// -DAFE_PLANNER_PROFILE: per-phase cycle totals (s_memtime), printed by launch_rappids; development only
#ifdef AFE_PLANNER_PROFILE
__device__ unsigned long long g_prof[40];   // [8]: the longest planner (cycles); [9] scan chunks examined, [10] of them holding a marked pixel, [11] scan calls
#define PL_T0(var) const unsigned long long var = __builtin_readcyclecounter()
__device__ unsigned long long g_longest[40];   // the same slots, of the longest planner alone
__shared__ unsigned long long s_mine[24];
#define PL_T1(var, slot) do { if (threadIdx.x == 0) { const unsigned long long d_ = __builtin_readcyclecounter() - var; atomicAdd(&g_prof[slot], d_); s_mine[slot] += d_; } } while (0)
#define PL_COUNT(slot, n) do { if (threadIdx.x == 0) { atomicAdd(&g_prof[slot], (unsigned long long)(n)); s_mine[slot] += (unsigned long long)(n); } } while (0)
struct PlScope {      // a phase's cycles however it is left (early returns included)
  unsigned long long t0; int slot;
  __device__ PlScope(int s) : t0(__builtin_readcyclecounter()), slot(s) {}
  __device__ ~PlScope() { if (threadIdx.x == 0) { const unsigned long long d_ = __builtin_readcyclecounter() - t0; atomicAdd(&g_prof[slot], d_); } }
};
#define PL_SCOPE(name, slot) PlScope name(slot)
#ifndef AFE_PLANNER_PROFILE_SCANS   // the per-chunk counters sit in the innermost loops and slow the kernel several times
#define PL_COUNT_SCAN(slot, n)
#else
#define PL_COUNT_SCAN(slot, n) PL_COUNT(slot, n)
#endif
#else
#define PL_SCOPE(name, slot)
#define PL_T0(var)
#define PL_T1(var, slot)
#define PL_COUNT(slot, n)
#define PL_COUNT_SCAN(slot, n)
#endif

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
#ifndef AFE_SHRINK_QUAL
#define AFE_SHRINK_QUAL            /* tools/planner_inline_probe.sh builds one variant with `volatile` here */
#endif
struct Shrink {
  AFE_SHRINK_QUAL int right, left, top, bottom;
};

// NOTE on `noinline` below (build_mask, first_blocking_ring, side_scan, corner_scan, inflate_pyramid): with these
// helpers inlined into the one very large search kernel, hipcc (ROCm 7.2) produced code in which
// some of the guarded corner scans were skipped although their guard held -- same inputs, same
// bit image, isolated copies of the same functions correct (found with a 640-plan campaign against
// the oracle on rendered orchard images, now tests/test_gpu_planner.py::test_campaign_...).  Keeping
// the wave-cooperative pieces as separate functions gives the structurizer small, reducible bodies;
// it costs nothing measurable and the campaign is the regression test.
// Round 2 re-test: the bit image is now bracketed by workgroup barriers (build_mask), which rules out the
// one cross-lane ordering this file relied on implicitly -- and with the helpers inlined again the campaign
// still fails (plan 79 of 640: one candidate reported collision-free that is not).  So the barriers stay
// because they are right, and `noinline` stays because it is needed.
// Round 3 (tools/planner_inline_probe.sh, one helper force-inlined at a time): only side_scan's inlining breaks the
// campaign, and only at the kernel's 128-VGPR budget -- the same source is correct at 168 or 256 VGPRs
// (amdgpu_waves_per_eu 3 / 2).  The failing build is the one where the scans' wave-uniform state lives across vector
// spills inside inflate_pyramid's divergent regions (planner section).
// (each helper's attribute is a macro so that a build can inline them one at a time: tools/planner_inline_probe.sh)
#ifndef AFE_NI_MASK
#define AFE_NI_MASK __attribute__((noinline))
#endif
#ifndef AFE_NI_RING
#define AFE_NI_RING __attribute__((noinline))
#endif
#ifndef AFE_NI_SIDE
#define AFE_NI_SIDE __attribute__((noinline))
#endif
#ifndef AFE_NI_CORNER
#define AFE_NI_CORNER __attribute__((noinline))
#endif
#ifndef AFE_NI_INFLATE
#define AFE_NI_INFLATE __attribute__((noinline))
#endif
// ---- wave-cooperative pixel scans ------------------------------------------------
// One wave runs one planner: everything outside the scans below is computed redundantly
// (and therefore convergently) by all 64 lanes; inside a scan lane l looks at pixel
// base + l of the reference's scan order.  The reference's loops are sequential -- a
// pixel's test reads edges that earlier pixels may have moved -- so a chunk is resolved
// as: ballot the lanes whose test holds under the CURRENT edges, let the first of them
// (in scan order) apply its update, re-test the lanes after it, repeat.  Between two
// updates the edges are constant, so this visits exactly the pixels the sequential loop
// would act on, in the same order.  For the four side scans the ordinary update is a
// running min / max of one edge (see side_scan), which a wave reduction applies to a whole
// chunk at once.
// (the result is the same in every lane; readfirstlane tells the compiler so, which moves whatever is
// derived from it -- edges, loop bounds, addresses -- into scalar registers and the scalar ALU)
// A value that is the same in every lane of the wave by construction, said so to the compiler (it lands in a scalar
// register and every decision made from it is a scalar branch).  The out-of-line helpers below take such values as
// arguments -- which arrive in vector registers, "divergent" as far as the compiler can know; decisions made from them
// inside loops whose other state is scalar were what the round-2 / round-3 `noinline` mystery came down to.
#ifndef AFE_PLANNER_DIVERGENT_ARGS
#define PL_UNIFORM(x) ((x) = __builtin_amdgcn_readfirstlane(x))
#else
#define PL_UNIFORM(x) ((void)0)
#endif
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o < v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m); v = o > v ? o : v; }
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ uint64_t lanes_from(int l) { return l >= 64 ? 0ull : (~0ull << l); }

// ---- bit images in LDS ----------------------------------------------------------------------
// Pyramid inflation asks two yes/no questions of every pixel: "nearer than minDepthPix?" during the
// spiral expansion (DIP.cpp:520-600) and "nearer than the pyramid's far plane?" during the shrink
// scans (:602-940).  Each is answered once per pyramid for the whole image by a sweep of fully
// independent loads (many in flight, no decision in between) that leaves ONE BIT per pixel in LDS:
// word (y, w) holds pixels x = 64w .. 64w+63 of row y.  The decision loops then run on the bit
// image -- a ring of the expansion is a handful of LDS reads and ballots, a shrink scan touches
// HBM only for the few chunks that contain a marked pixel.
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
// i / inner for 0 <= i < 2^24, 1 <= inner <= 2^15 as a multiply-high
__device__ __forceinline__ unsigned div_magic(int inner) { return (unsigned)(0x100000000ull / (unsigned)inner) + 1u; }
__device__ __forceinline__ int div_small(int i, int inner, unsigned magic) {
  return inner == 1 ? i : (int)__umulhi((unsigned)i, magic);
}

constexpr int kSweepBatch = 8;

// mask(x, y) = lo < d(x, y) < hi for the whole image
// WANT_MIN: also return the smallest marked depth (65535 if none) -- the shrink scans stop where
// even that depth could not reach an edge any more.  (Images whose rows are not whole 64-pixel words, or too large for
// the summaries' list in LDS; everything else goes through build_mask_sum below.)
template <bool WANT_MIN>
__device__ AFE_NI_MASK int build_mask(const uint16_t *__restrict__ img, int W, int H, int lane, uint64_t *mask, int WW, uint16_t lo,
                          uint16_t hi) {
  PL_UNIFORM(W); PL_UNIFORM(H); PL_UNIFORM(WW);
  { int l = lo, h = hi; PL_UNIFORM(l); PL_UNIFORM(h); lo = (uint16_t)l; hi = (uint16_t)h; }
  int lane_min = 65535;
  // The bit image is written by one set of lanes and read by others.  The block is a single wave,
  // so these barriers cost nothing, but they are what orders the LDS traffic: readers of the
  // previous image are done before it is overwritten, and the new image is complete (and the
  // compiler may not move LDS reads across) before anyone looks at it.
  __syncthreads();
  const int n = WW * H;
  const unsigned magic = div_magic(WW);
  for (int c0 = 0; c0 < n; c0 += kSweepBatch) {
    uint16_t d[kSweepBatch];
#pragma unroll
    for (int u = 0; u < kSweepBatch; u++) {
      const int c = c0 + u;
      d[u] = 0;
      if (c < n) {
        const int y = div_small(c, WW, magic), x = 64 * (c - y * WW) + lane;
        if (x < W) d[u] = img[y * W + x];
      }
    }
#pragma unroll
    for (int u = 0; u < kSweepBatch; u++) {
      const int c = c0 + u;
      if (c < n) {
        const bool marked = d[u] > lo && d[u] < hi;
        const uint64_t bits = __ballot(marked);
        if (lane == 0) mask[c] = bits;
        if (WANT_MIN && marked) lane_min = PL_MIN(lane_min, (int)d[u]);
      }
    }
  }
  __syncthreads();
  return WANT_MIN ? wave_min_i32(lane_min) : 65535;
}

// ---- the bit image from per-word summaries (round 6) --------------------------------------------------
// PlannerBatch::sums holds, per 64-pixel word of the image, the smallest depth above `ignore` and the largest depth (or
// 0xffff when a pixel at or below `ignore` sits in the word).  Both questions build_mask answers are "ignore < d < hi":
// a word whose smallest such depth is >= hi is all zeros, a word whose largest depth is < hi (and that holds no ignored
// pixel) is all ones; only the words a depth edge at `hi` runs through -- a tenth of an orchard view -- need their
// pixels.  Lane l decides word base + l from one dword; the undecided words of the batch are then taken eight at a
// time, eight lanes per word (one 16-byte vector = one byte of the bit image each), the k-th undecided word found by
// a push (ds_permute: word of rank k -> lane 8k) and a pull (ds_bpermute from lane & ~7) through the LDS crossbar, no
// LDS memory.  The range test itself is packed 16-bit arithmetic: t = d - (lo + 1) wraps the depths at or below lo
// ABOVE every depth in range, u = sat(t - (span - 1)) is zero exactly for t < span, min(u, 1) is the NOT-in-range
// bit of both halves of a dword: 4 vector instructions per two pixels instead of 10.
// (the three instructions themselves: from the builtins the compiler rebuilds the comparison -- two v_cmp, two
// v_cndmask and a v_perm per dword)
__device__ __forceinline__ unsigned not_in_range2(unsigned two_px, unsigned lo1, unsigned span1) {
  unsigned t, u, n;
  asm("v_pk_sub_u16 %0, %1, %2" : "=v"(t) : "v"(two_px), "s"(lo1));
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(u) : "v"(t), "s"(span1));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(n) : "v"(u), "s"(0x00010001u));
  return n;
}
// the 8 bits of 8 pixels (bit j = pixel j is in (lo, hi)); lo1 = lo + 1, span1 = hi - lo - 2 (>= 0: the caller knows a depth in range exists)
__device__ __forceinline__ unsigned range_bits8_packed(const uint4 q, unsigned lo1, unsigned span1) {
  unsigned acc = not_in_range2(q.x, lo1, span1);
  acc |= not_in_range2(q.y, lo1, span1) << 2;
  acc |= not_in_range2(q.z, lo1, span1) << 4;
  acc |= not_in_range2(q.w, lo1, span1) << 6;
  return ~(acc | (acc >> 15)) & 0xffu;      // low halves sit at bits 0, 2, 4, 6, high halves at 16, 18, 20, 22
}

// Memory round trips, not instructions, are what a planner's wave spends its time on (four waves share a SIMD and each
// has a few dozen vector instructions between two waits): the summaries arrive kSumBatch batches of 64 words at a time
// (one wait), the undecided words go to a list in LDS behind the bit image (kSumList entries; emptied when full), and the
// list is worked off 8 * kSumGroups words per trip -- for an orchard view five waits per bit image instead of nineteen.
constexpr int kSumBatch = 10, kSumGroups = 8, kSumList = 256;

__device__ __forceinline__ void sum_list_flush(const uint4 *__restrict__ src, uint8_t *bytes, const uint16_t *list, int count, int lane,
                                               unsigned lo1, unsigned span1) {
  for (int g = 0; g < count; g += 8 * kSumGroups) {
    uint4 q[kSumGroups];
    int vec[kSumGroups];
#pragma unroll
    for (int u = 0; u < kSumGroups; u++) {
      const int idx = g + 8 * u + (lane >> 3);
      vec[u] = -1;
      q[u] = make_uint4(0, 0, 0, 0);
      if (idx < count) { vec[u] = (int)list[idx] * 8 + (lane & 7); q[u] = src[vec[u]]; }
    }
#pragma unroll
    for (int u = 0; u < kSumGroups; u++)
      if (vec[u] >= 0) bytes[vec[u]] = (uint8_t)range_bits8_packed(q[u], lo1, span1);
  }
}

template <bool WANT_MIN>
__device__ AFE_NI_MASK int build_mask_sum(const uint16_t *__restrict__ img, const uint32_t *__restrict__ sums, int nwords, int lane,
                                          uint64_t *mask, uint16_t lo, uint16_t hi) {
  PL_UNIFORM(nwords);
  { int l = lo, h = hi; PL_UNIFORM(l); PL_UNIFORM(h); lo = (uint16_t)l; hi = (uint16_t)h; }
  int lane_min = 65535;
  __syncthreads();                      // readers of the previous image are done (see build_mask)
  const uint4 *src = (const uint4 *)img;
  uint8_t *bytes = (uint8_t *)mask;
  uint16_t *list = (uint16_t *)(mask + nwords);
  const unsigned lo1s = (lo + 1u) & 0xffffu, sp1s = ((unsigned)hi - (unsigned)lo - 2u) & 0xffffu;
  const unsigned lo1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lo1s | (lo1s << 16))), span1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(sp1s | (sp1s << 16)));
  int count = 0;
  for (int base0 = 0; base0 < nwords; base0 += 64 * kSumBatch) {
    unsigned sm[kSumBatch];
#pragma unroll
    for (int u = 0; u < kSumBatch; u++) {
      const int w = base0 + 64 * u + lane;
      sm[u] = 0xffffffffu;
      if (w < nwords) sm[u] = sums[w];
    }
#pragma unroll
    for (int u = 0; u < kSumBatch; u++) {
      const int w = base0 + 64 * u + lane;
      const bool valid = w < nwords;
      const unsigned mn = sm[u] & 0xffffu, mxe = sm[u] >> 16;
      const bool zero = mn >= (unsigned)hi, ones = mxe < (unsigned)hi;
      const bool open = valid && !zero && !ones;      // (then lo < mn < hi: hi >= lo + 2)
      if (valid && !open) mask[w] = ones ? ~0ull : 0ull;
      if (WANT_MIN && !zero) lane_min = PL_MIN(lane_min, (int)mn);    // the word's smallest depth above lo is below hi: it is marked
      const uint64_t b = __ballot(open);
      if (!b) continue;
      const int cnt = __popcll(b);
      if (count + cnt > kSumList) { sum_list_flush(src, bytes, list, count, lane, lo1, span1); count = 0; }
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
      if (open) list[count + rank] = (uint16_t)w;
      count += cnt;
    }
  }
  sum_list_flush(src, bytes, list, count, lane, lo1, span1);
  __syncthreads();
  return WANT_MIN ? wave_min_i32(lane_min) : 65535;
}

// first marked pixel of column x, rows ya..yb / of row y, columns xa..xb
__device__ __forceinline__ bool mask_col_hit(const uint64_t *mask, int WW, int lane, int x, int ya, int yb, int &first) {
  const int w = x >> 6, sh = x & 63;
  for (int base = ya; base <= yb; base += 64) {
    const int y = base + lane;
    const uint64_t word = y <= yb ? mask[y * WW + w] : 0ull;
    const uint64_t b = __ballot((word >> sh) & 1ull);
    if (b) { first = base + (int)__ffsll((unsigned long long)b) - 1; return true; }
  }
  return false;
}
__device__ __forceinline__ uint64_t clip_word(uint64_t word, int w, int xa, int xb) {
  if (w == (xa >> 6)) word &= ~0ull << (xa & 63);
  if (w == (xb >> 6)) word &= ~0ull >> (63 - (xb & 63));
  return word;
}
__device__ __forceinline__ bool mask_row_hit(const uint64_t *mask, int WW, int lane, int y, int xa, int xb, int &first) {
  const int wa = xa >> 6, wb = xb >> 6;
  for (int wbase = wa; wbase <= wb; wbase += 64) {
    const int w = wbase + lane;
    const uint64_t word = w <= wb ? clip_word(mask[y * WW + w], w, xa, xb) : 0ull;
    const uint64_t b = __ballot(word != 0);
    if (b) {
      const int l = (int)__ffsll((unsigned long long)b) - 1;
      first = 64 * (wbase + l) + (int)__ffsll((unsigned long long)shfl_u64(word, l)) - 1;
      return true;
    }
  }
  return false;
}
// any marked pixel in [xa, xb] x [ya, yb]?
__device__ __forceinline__ bool mask_region_any(const uint64_t *mask, int WW, int lane, int xa, int xb, int ya, int yb) {
  if (xa > xb || ya > yb) return false;
  const int wa = xa >> 6, nw = (xb >> 6) - wa + 1, total = nw * (yb - ya + 1);
  const unsigned magic = div_magic(nw);
  for (int base = 0; base < total; base += 64) {
    const int p = base + lane;
    uint64_t word = 0;
    if (p < total) {
      const int o = div_small(p, nw, magic), w = wa + (p - o * nw);
      word = clip_word(mask[(ya + o) * WW + w], w, xa, xb);
    }
    if (__ballot(word != 0)) return true;
  }
  return false;
}
__device__ __forceinline__ bool mask_bit(const uint64_t *mask, int WW, int x, int y) {
  return (mask[y * WW + (x >> 6)] >> (x & 63)) & 1ull;
}

// The expansion examines, in ring j (counted from the current rectangle), exactly the pixels at
// "ring distance" j = max(dx, dy) from it, dx / dy being the distance beyond the rectangle's
// right / left and top / bottom edge (0 in between) -- as long as every side that is still free keeps
// advancing.  The first ring in which a free side meets a marked pixel is therefore the smallest
// ring distance of any marked pixel inside the window the free sides can still reach; all rings
// before it are clear and can be taken in one step.  Returns that ring index (>= 1) or INT_MAX.
__device__ AFE_NI_RING int first_blocking_ring(const uint64_t *mask, int WW, int lane, int xlo, int xhi, int ylo, int yhi, int L,
                                   int R, int T, int B) {
  PL_UNIFORM(WW); PL_UNIFORM(xlo); PL_UNIFORM(xhi); PL_UNIFORM(ylo); PL_UNIFORM(yhi); PL_UNIFORM(L); PL_UNIFORM(R); PL_UNIFORM(T); PL_UNIFORM(B);
  const int wa = xlo >> 6, nw = (xhi >> 6) - wa + 1, total = nw * (yhi - ylo + 1);
  const unsigned magic = div_magic(nw);
  int best = 0x7fffffff;
  for (int base = 0; base < total; base += 64) {
    const int p = base + lane;
    if (p < total) {
      const int o = div_small(p, nw, magic), w = wa + (p - o * nw), y = ylo + o;
      const uint64_t word = clip_word(mask[y * WW + w], w, xlo, xhi);
      if (word) {
        const int dy = y < T ? T - y : (y > B ? y - B : 0);
        const int x_first = 64 * w, x_last = x_first + 63;
        // bits right of R: the nearest is the lowest one
        if (x_last > R) {
          const uint64_t part = x_first > R ? word : word & (~0ull << ((R + 1) & 63));
          if (part) { const int dx = x_first + (int)__ffsll((unsigned long long)part) - 1 - R; best = PL_MIN(best, PL_MAX(dx, dy)); }
        }
        // bits left of L: the nearest is the highest one
        if (x_first < L) {
          const uint64_t part = x_last < L ? word : word & ~(~0ull << (L & 63));
          if (part) { const int dx = L - (x_first + 63 - __clzll((long long)part)); best = PL_MIN(best, PL_MAX(dx, dy)); }
        }
        // bits in [L, R]: above or below the rectangle (inside it nothing is examined any more)
        if (dy > 0 && x_last >= L && x_first <= R && clip_word(word, w, PL_MAX(L, x_first), PL_MIN(R, x_last)))
          best = PL_MIN(best, dy);
      }
    }
  }
  return wave_min_i32(best);
}

// min over d > ignore of n pixels src[0], src[stride], ... (a blocked line's prefix)
__device__ __forceinline__ void min_line(const uint16_t *__restrict__ src, int stride, int n, int lane, uint16_t ignore,
                                         int &laneMin) {
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    if (i < n) { const uint16_t d = src[i * stride]; if (d > ignore) laneMin = PL_MIN(laneMin, (int)d); }
  }
}


enum { SIDE_RIGHT = 0, SIDE_LEFT = 1, SIDE_TOP = 2, SIDE_BOTTOM = 3 };

// One of the four side scans, DIP.cpp:617-785.  `total` pixels, pixel i at
// (xa + (i / inner) * dxo + (i % inner) * dxi, ya + ...).  Returns false for the
// reference's "return false".
// Scans walk `total` pixels in chunks of 64; kScanBatch chunks are loaded together (the loads do
// not depend on the edges) and then resolved one after the other, so one memory latency is paid
// per batch instead of per chunk.  i / inner for i < 2^24, inner <= 2^15 as a multiply-high.
#ifndef AFE_SCAN_BATCH
#define AFE_SCAN_BATCH 4
#endif
constexpr int kScanBatch = AFE_SCAN_BATCH;

template <int SIDE>
__device__ AFE_NI_SIDE bool side_scan(const uint16_t *__restrict__ src, int sx, int sy, const uint64_t *mask, int WW, int lane,
                          int total, int inner, int xa, int ya, int dxo, int dyo, int dxi, int dyi, int num, int buf,
                          int x0, int y0, int dmin, Shrink &s) {
  PL_UNIFORM(sx); PL_UNIFORM(sy); PL_UNIFORM(WW); PL_UNIFORM(total); PL_UNIFORM(inner); PL_UNIFORM(xa); PL_UNIFORM(ya);
  PL_UNIFORM(dxo); PL_UNIFORM(dyo); PL_UNIFORM(dxi); PL_UNIFORM(dyi); PL_UNIFORM(num); PL_UNIFORM(buf); PL_UNIFORM(x0); PL_UNIFORM(y0); PL_UNIFORM(dmin);
  const unsigned magic = div_magic(inner);
  for (int base0 = 0; base0 < total; base0 += 64 * kScanBatch) {
    {  // lines run outward from the rectangle: once even the nearest marked depth cannot reach the
       // (current) edge from this line, no later pixel of the scan can act
      const int o0 = div_small(base0, inner, magic);
      int dist;
      if (SIDE == SIDE_RIGHT) dist = xa + o0 * dxo - s.right;
      else if (SIDE == SIDE_LEFT) dist = s.left - (xa + o0 * dxo);
      else if (SIDE == SIDE_TOP) dist = s.top - (ya + o0 * dyo);
      else dist = ya + o0 * dyo - s.bottom;
      if (dist > 0 && dist * dmin >= num) break;
    }
    int xs[kScanBatch], ys[kScanBatch];
    bool vs[kScanBatch];
    uint16_t ds[kScanBatch];
#pragma unroll
    for (int c = 0; c < kScanBatch; c++) {
      const int i = base0 + 64 * c + lane;
      xs[c] = 0; ys[c] = 0; vs[c] = false;
      if (i < total) {
        const int o = div_small(i, inner, magic), r = i - o * inner;
        xs[c] = xa + o * dxo + r * dxi;
        ys[c] = ya + o * dyo + r * dyi;
        vs[c] = mask_bit(mask, WW, xs[c], ys[c]);          // ignore < d < maxDepth, from the bit image
      }
    }
#ifdef AFE_PLANNER_PROFILE
    for (int c = 0; c < kScanBatch; c++) if (base0 + 64 * c < total) { PL_COUNT_SCAN(9, 1); if (__ballot(vs[c])) PL_COUNT_SCAN(10, 1); }
#endif
#pragma unroll
    for (int c = 0; c < kScanBatch; c++) {                  // depths only where a marked pixel needs one
      ds[c] = 1;
      if (vs[c]) ds[c] = src[ys[c] * sy + xs[c] * sx];     // (1, W) on the image, (H, 1) on its transpose
    }
#pragma unroll
    for (int c = 0; c < kScanBatch; c++) {
      const int x = xs[c], y = ys[c];
      const uint16_t d = ds[c];
      const bool valid = vs[c];
      {  // no marked pixel of the chunk acts under the edges as they stand: nothing moves, next chunk
         // (before the integer division below, which most chunks of a cluttered image then never pay)
        bool hit0;
        if (SIDE == SIDE_RIGHT) hit0 = valid && num > (x - s.right) * (int)d;
        else if (SIDE == SIDE_LEFT) hit0 = valid && (s.left - x) * (int)d < num;
        else if (SIDE == SIDE_TOP) hit0 = valid && (s.top - y) * (int)d < num;
        else hit0 = valid && num > (y - s.bottom) * (int)d;
        if (!__ballot(hit0)) continue;
      }
      int k = 0;
      if (valid) k = (int)(num / d);
      // the edge this pixel asks for, and whether granting it would cut the seed pixel off
      int want;
      bool exceptional;
      if (SIDE == SIDE_RIGHT) { want = x - k; exceptional = x0 > want - buf; }
      else if (SIDE == SIDE_LEFT) { want = x + k; exceptional = x0 < want + buf; }
      else if (SIDE == SIDE_TOP) { want = y + k; exceptional = y0 < want + buf; }
      else { want = y - k; exceptional = y0 > want - buf; }
      const uint64_t excMask = __ballot(valid && exceptional);
      uint64_t todo = ~0ull;
      for (;;) {
        bool hit;
        if (SIDE == SIDE_RIGHT) hit = valid && num > (x - s.right) * (int)d;
        else if (SIDE == SIDE_LEFT) hit = valid && (s.left - x) * (int)d < num;
        else if (SIDE == SIDE_TOP) hit = valid && (s.top - y) * (int)d < num;
        else hit = valid && num > (y - s.bottom) * (int)d;
        const uint64_t b = __ballot(hit) & todo;
        if (!b) break;
        const uint64_t be = b & excMask;
        const int l = be ? (int)__ffsll((unsigned long long)be) - 1 : 64;
        const uint64_t pre = b & ~lanes_from(l);
        if (pre) {
          // ordinary updates before lane l: `edge = want` under the test is a running min (right,
          // bottom) or max (left, top) of `want` -- the test holds iff `want` is beyond the edge
          const bool mine = hit && ((pre >> lane) & 1ull);
          if (SIDE == SIDE_RIGHT) { const int m = wave_min_i32(mine ? want : 0x7fffffff); s.right = PL_MIN(s.right, m); }
          else if (SIDE == SIDE_LEFT) { const int m = wave_max_i32(mine ? want : -0x7fffffff); s.left = PL_MAX(s.left, m); }
          else if (SIDE == SIDE_TOP) { const int m = wave_max_i32(mine ? want : -0x7fffffff); s.top = PL_MAX(s.top, m); }
          else { const int m = wave_min_i32(mine ? want : 0x7fffffff); s.bottom = PL_MIN(s.bottom, m); }
          if (l >= 64) break;
          todo = lanes_from(l);      // lane l is re-tested against the moved edge
          continue;
        }
        // lane l is next in scan order and exceptional: the reference's inner branch
        const int xl = __builtin_amdgcn_readlane(x, l), yl = __builtin_amdgcn_readlane(y, l), kl = __builtin_amdgcn_readlane(k, l);
        if (SIDE == SIDE_RIGHT || SIDE == SIDE_LEFT) {
          const int tT = yl + kl, bT = yl - kl;
          if (y0 < tT + buf && y0 > bT - buf) return false;
          else if (y0 < tT + buf) s.bottom = bT;
          else if (y0 > bT - buf) s.top = tT;
          else if ((s.bottom - bT) > (tT - s.top)) s.top = tT;
          else if (SIDE == SIDE_RIGHT) s.right = bT;            // sic, DIP.cpp:648
          else s.bottom = bT;
        } else {
          const int rT = xl - kl, lT = xl + kl;
          if (x0 > rT - buf && x0 < lT + buf) return false;
          else if (x0 > rT - buf) s.left = lT;
          else if (x0 < lT + buf) s.right = rT;
          else if ((s.right - rT) > (lT - s.left)) s.left = lT;
          else s.right = rT;
        }
        todo = lanes_from(l + 1);
      }
    }
  }
  return true;
}

enum { CORNER_TR = 0, CORNER_BR = 1, CORNER_TL = 2, CORNER_BL = 3 };

// One of the four corner scans, DIP.cpp:794-940: rows outward from the top / bottom edge,
// pixels outward from the right / left edge.
template <int CORNER>
__device__ AFE_NI_CORNER bool corner_scan(const uint16_t *__restrict__ img, int W, const uint64_t *mask, int WW, int lane, int rows,
                            int inner, int xa, int ya, int num, int buf, int x0, int y0, int dmin, Shrink &s) {
  constexpr bool RIGHT = (CORNER == CORNER_TR || CORNER == CORNER_BR);
  constexpr bool TOP = (CORNER == CORNER_TR || CORNER == CORNER_TL);
  PL_UNIFORM(W); PL_UNIFORM(WW); PL_UNIFORM(rows); PL_UNIFORM(inner); PL_UNIFORM(xa); PL_UNIFORM(ya); PL_UNIFORM(num); PL_UNIFORM(buf);
  PL_UNIFORM(x0); PL_UNIFORM(y0); PL_UNIFORM(dmin);
  {  // within a corner scan the right edge only moves left (the left edge only right), so pixels
     // farther out than ceil(num / dmin) from where it stands now can never act: narrow the rows
    const int reach = (num + dmin - 1) / dmin;
    inner = PL_MIN(inner, RIGHT ? s.right + reach - xa : xa - s.left + reach);
    if (inner <= 0) return true;
  }
  const int total = rows * inner;
  const unsigned magic = div_magic(inner);
  for (int base0 = 0; base0 < total; base0 += 64 * kScanBatch) {
    {  // rows run outward from the top / bottom edge: see side_scan
      const int o0 = div_small(base0, inner, magic);
      const int dist = TOP ? s.top - (ya - o0) : (ya + o0) - s.bottom;
      if (dist > 0 && dist * dmin >= num) break;
    }
    int xs[kScanBatch], ys[kScanBatch];
    bool vs[kScanBatch];
    uint16_t ds[kScanBatch];
#pragma unroll
    for (int c = 0; c < kScanBatch; c++) {
      const int i = base0 + 64 * c + lane;
      xs[c] = 0; ys[c] = 0; vs[c] = false;
      if (i < total) {
        const int o = div_small(i, inner, magic), r = i - o * inner;
        xs[c] = RIGHT ? xa + r : xa - r;
        ys[c] = TOP ? ya - o : ya + o;
        vs[c] = mask_bit(mask, WW, xs[c], ys[c]);
      }
    }
#ifdef AFE_PLANNER_PROFILE
    for (int c = 0; c < kScanBatch; c++) if (base0 + 64 * c < total) { PL_COUNT_SCAN(9, 1); if (__ballot(vs[c])) PL_COUNT_SCAN(10, 1); }
#endif
#pragma unroll
    for (int c = 0; c < kScanBatch; c++) {
      ds[c] = 1;
      if (vs[c]) ds[c] = img[ys[c] * W + xs[c]];
    }
#pragma unroll
    for (int c = 0; c < kScanBatch; c++) {
      const int x = xs[c], y = ys[c];
      const uint16_t d = ds[c];
      const bool valid = vs[c];
      if (!__ballot(valid)) continue;
      uint64_t todo = ~0ull;
      for (;;) {
        const bool hx = RIGHT ? num > (x - s.right) * (int)d : (s.left - x) * (int)d < num;
        const bool hy = TOP ? (s.top - y) * (int)d < num : num > (y - s.bottom) * (int)d;
        const uint64_t b = __ballot(valid && hx && hy) & todo;
        if (!b) break;
        const int l = (int)__ffsll((unsigned long long)b) - 1;
        const int xl = __builtin_amdgcn_readlane(x, l), yl = __builtin_amdgcn_readlane(y, l);
        const int kl = num / __builtin_amdgcn_readlane((int)d, l);   // only the acting pixel's quotient is needed
        const int xT = RIGHT ? xl - kl : xl + kl;          // rightTemp / leftTemp
        const int yT = TOP ? yl + kl : yl - kl;            // topTemp / bottomTemp
        const bool cutX = RIGHT ? x0 > xT - buf : x0 < xT + buf;
        const bool cutY = TOP ? y0 < yT + buf : y0 > yT - buf;
        if (cutX && cutY) return false;
        bool moveY;
        if (cutX) moveY = true;
        else if (cutY) moveY = false;
        else {
          const int lossX = RIGHT ? (s.right - xT) : (xT - s.left);
          const int lossY = TOP ? (yT - s.top) : (s.bottom - yT);
          moveY = lossX * (s.bottom - s.top) > lossY * (s.right - s.left);
        }
        if (moveY) { if (TOP) s.top = yT; else s.bottom = yT; }
        else { if (RIGHT) s.right = xT; else s.left = xT; }
        todo = lanes_from(l + 1);
      }
    }
  }
  return true;
}

// DIP.cpp:456-970, executed by one wave (lane = 0..63, everything but the scans is uniform)
__device__ AFE_NI_INFLATE bool inflate_pyramid(const PlannerConfig &c, const uint16_t *__restrict__ img,
                                const uint16_t *__restrict__ imgT, int HT, const uint32_t *__restrict__ sums, uint64_t *mask, int lane, int x0, int y0,
                                double minimumDepth, PlannerPyramid &out) {
#pragma clang fp contract(off)
  PL_COUNT(22, 1);
#ifndef AFE_PLANNER_DIVERGENT_ARGS
  // Every lane of the wave holds the same seed pixel and depth, but as arguments of an out-of-line function they arrive
  // in vector registers and the compiler has to assume they differ: every decision made from them (the scans' "would this
  // cut the seed pixel off" tests, their `return false`) is then a DIVERGENT exit from loops whose other state -- the
  // edges, the ballot masks -- it keeps in scalar registers.  Saying that they are uniform makes those exits scalar
  // branches.  (Round 4: this is what the failing build of round 2 / 3 depended on -- tools/planner_opt_bisect.sh,
  // DESIGN.md planner section.)
  x0 = __builtin_amdgcn_readfirstlane(x0);
  y0 = __builtin_amdgcn_readfirstlane(y0);
  HT = __builtin_amdgcn_readfirstlane(HT);
  {
    const unsigned long long u = (unsigned long long)__double_as_longlong(minimumDepth);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    minimumDepth = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
#endif
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
  PL_SCOPE(sc_all, 24);
  {  // :505-518, any pixel of [L,R) x [T,B) nearer than minDepthPix
    PL_SCOPE(sc_seed, 25);
    // (the answer does not depend on the order the pixels are looked at: eight chunks of 64 are loaded together,
    // one memory round trip per 512 pixels instead of eight -- a plan in a cluttered image asks this ~90 times)
    const int w = R - L, total = w * (B - T);
    const unsigned magic = div_magic(w);
    for (int base = 0; base < total; base += 64 * kSweepBatch) {
      uint16_t d[kSweepBatch];
#pragma unroll
      for (int u = 0; u < kSweepBatch; u++) {
        const int i = base + 64 * u + lane;
        d[u] = 0;                                        // 0 <= ignore: never "bad"
        if (i < total) {
          const int o = div_small(i, w, magic);
          d[u] = img[(T + o) * W + L + (i - o * w)];
        }
      }
      bool bad = false;
#pragma unroll
      for (int u = 0; u < kSweepBatch; u++) bad |= d[u] <= minDepthPix && d[u] > ignore;
      if (__ballot(bad)) { PL_COUNT(23, 1); return false; }
    }
  }
  PL_T0(t_exp);
  // spiral expansion, :520-600, on the bit image "nearer than minDepthPix"
  const int WW = (W + 63) >> 6;
  PL_T0(t_m1);
  if (sums) build_mask_sum<false>(img, sums, WW * H, lane, mask, ignore, minDepthPix);
  else build_mask<false>(img, W, H, lane, mask, WW, ignore, minDepthPix);
  PL_T1(t_m1, 1);
  PL_T0(t_ring);
  const int L0 = L, T0 = T, R0 = R, B0 = B;
  // a blocked side: the line it was blocked on and how much of it was examined before the hit
  int rbX = 0, rbFrom = 0, rbN = 0, tbY = 0, tbFrom = 0, tbN = 0, lbX = 0, lbFrom = 0, lbN = 0, bbY = 0, bbFrom = 0, bbN = 0;
  bool rFree = true, tFree = true, lFree = true, bFree = true;
  while (rFree || tFree || lFree || bFree) {
    int first;
    {  // take all the clear rings in one step (see first_blocking_ring)
      const int xlo = lFree ? edgeOff : L, xhi = rFree ? W - edgeOff - 1 : R;
      const int ylo = tFree ? edgeOff : T, yhi = bFree ? H - edgeOff - 1 : B;
      const int ring = first_blocking_ring(mask, WW, lane, xlo, xhi, ylo, yhi, L, R, T, B);
      const int clear = ring == 0x7fffffff ? 0x3fffffff : ring - 1;
      if (rFree) R = PL_MIN(R + clear, W - edgeOff - 1);
      if (tFree) T = PL_MAX(T - clear, edgeOff);
      if (lFree) L = PL_MAX(L - clear, edgeOff);
      if (bFree) B = PL_MIN(B + clear, H - edgeOff - 1);
    }
    if (rFree) {
      if (R < W - edgeOff - 1) {
        if (mask_col_hit(mask, WW, lane, R + 1, T, B, first)) { rFree = false; rbX = R + 1; rbFrom = T; rbN = first - T; }
        else R++;
      } else rFree = false;
    }
    if (tFree) {
      if (T > edgeOff) {
        if (mask_row_hit(mask, WW, lane, T - 1, L, R, first)) { tFree = false; tbY = T - 1; tbFrom = L; tbN = first - L; }
        else T--;
      } else tFree = false;
    }
    if (lFree) {
      if (L > edgeOff) {
        if (mask_col_hit(mask, WW, lane, L - 1, T, B, first)) { lFree = false; lbX = L - 1; lbFrom = T; lbN = first - T; }
        else L--;
      } else lFree = false;
    }
    if (bFree) {
      if (B < H - edgeOff - 1) {
        if (mask_row_hit(mask, WW, lane, B + 1, L, R, first)) { bFree = false; bbY = B + 1; bbFrom = L; bbN = first - L; }
        else B++;
      } else bFree = false;
    }
  }
  PL_T1(t_ring, 7);
  // maxDepth (:531,546,561,576): the nearest d > ignore over every pixel the expansion examined =
  // the grown rectangle minus the initial one (each growth step adds exactly the line it examined)
  // plus the examined prefixes of the blocked lines
  int laneMin = 65535;
  {
    PL_SCOPE(sc_maxd, 26);
    if ((W & 63) == 0) {
      // 8 pixels (one 16-byte vector, never straddling a row since W % 8 == 0) per lane per load,
      // only the vectors that overlap columns L..R.  A vector that lies wholly inside the counted
      // region takes the packed path: t = d - (ignore + 1) in 16-bit wrap-around arithmetic maps
      // the ignored depths (d <= ignore) ABOVE every counted one, so a packed unsigned minimum of
      // t needs no per-pixel test.
      typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
      const uint4 *src = (const uint4 *)img;
      const int rowv = W >> 3, vL = L >> 3, nvr = (R >> 3) - vL + 1, nvec = nvr * (B - T + 1);
      const unsigned magic = div_magic(nvr);
      const unsigned short bias1 = (unsigned short)(ignore + 1);
      const u16x2 bias = {bias1, bias1};
      u16x2 packed_min = {(unsigned short)0xffff, (unsigned short)0xffff};
      for (int v0 = 0; v0 < nvec; v0 += 64 * kSweepBatch) {
        uint4 q[kSweepBatch];
#pragma unroll
        for (int u = 0; u < kSweepBatch; u++) {
          const int v = v0 + 64 * u + lane;
          q[u] = make_uint4(0, 0, 0, 0);
          if (v < nvec) {
            const int o = div_small(v, nvr, magic);
            q[u] = src[(T + o) * rowv + vL + (v - o * nvr)];
          }
        }
#pragma unroll
        for (int u = 0; u < kSweepBatch; u++) {
          const int v = v0 + 64 * u + lane;
          if (v >= nvec) continue;
          const int o = div_small(v, nvr, magic), xb = (vL + (v - o * nvr)) << 3, y = T + o;
          const bool initRow = y >= T0 && y <= B0;
          if (xb >= L && xb + 7 <= R && !(initRow && xb + 7 >= L0 && xb <= R0)) {
            packed_min = __builtin_elementwise_min(packed_min, __builtin_bit_cast(u16x2, q[u].x) - bias);
            packed_min = __builtin_elementwise_min(packed_min, __builtin_bit_cast(u16x2, q[u].y) - bias);
            packed_min = __builtin_elementwise_min(packed_min, __builtin_bit_cast(u16x2, q[u].z) - bias);
            packed_min = __builtin_elementwise_min(packed_min, __builtin_bit_cast(u16x2, q[u].w) - bias);
          } else {
            const unsigned px[8] = {q[u].x & 0xffffu, q[u].x >> 16, q[u].y & 0xffffu, q[u].y >> 16,
                                    q[u].z & 0xffffu, q[u].z >> 16, q[u].w & 0xffffu, q[u].w >> 16};
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const int x = xb + j;
              const bool counted = x >= L && x <= R && !(initRow && x >= L0 && x <= R0) && px[j] > ignore;
              if (counted) laneMin = PL_MIN(laneMin, (int)px[j]);
            }
          }
        }
      }
      const int t = PL_MIN((int)packed_min.x, (int)packed_min.y);
      if (t <= 65534 - (int)ignore) laneMin = PL_MIN(laneMin, t + (int)ignore + 1);   // else: only ignored depths seen
    } else {
      const int nxf = R - L + 1, total = nxf * (B - T + 1);
      const unsigned magic = div_magic(nxf);
      for (int base0 = 0; base0 < total; base0 += 64 * kSweepBatch) {
        uint16_t d[kSweepBatch];
#pragma unroll
        for (int u = 0; u < kSweepBatch; u++) {
          const int i = base0 + 64 * u + lane;
          d[u] = 0;
          if (i < total) {
            const int o = div_small(i, nxf, magic), x = L + (i - o * nxf), y = T + o;
            if (!(x >= L0 && x <= R0 && y >= T0 && y <= B0)) d[u] = img[y * W + x];
          }
        }
#pragma unroll
        for (int u = 0; u < kSweepBatch; u++)
          if (d[u] > ignore) laneMin = PL_MIN(laneMin, (int)d[u]);
      }
    }
    min_line(imgT + rbX * HT + rbFrom, 1, rbN, lane, ignore, laneMin);      // (HT: the transposed image's padded column length)
    min_line(img + tbY * W + tbFrom, 1, tbN, lane, ignore, laneMin);
    min_line(imgT + lbX * HT + lbFrom, 1, lbN, lane, ignore, laneMin);
    min_line(img + bbY * W + bbFrom, 1, bbN, lane, ignore, laneMin);
  }
  const uint16_t maxDepth = (uint16_t)wave_min_i32(laneMin);
  PL_T1(t_exp, 2);
  PL_T0(t_side);
  // shrink by the vehicle radius, :602-940, on the bit image "ignore < d < maxDepth"
  // dmin: the nearest marked pixel anywhere; (distance from an edge) * dmin >= num means no pixel
  // at that distance or beyond can move the edge (num / d <= num / dmin), so a scan stops there
  PL_SCOPE(sc_shrink, 27);
  const int dmin = sums ? build_mask_sum<true>(img, sums, WW * H, lane, mask, ignore, maxDepth)
                        : build_mask<true>(img, W, H, lane, mask, WW, ignore, maxDepth);
  Shrink s = {W - 1 - edgeOff, edgeOff, edgeOff, H - 1 - edgeOff};
  const int num = (int)(c.focal_length * c.planning_vehicle_radius / c.depth_scale);
  const int ny = B - T + 1, nx = R - L + 1;
  // right side :617-661 (columns R.. outward, rows T..B); left side :663-698
  if (mask_region_any(mask, WW, lane, R, W - 1, T, B) &&
      !side_scan<SIDE_RIGHT>(imgT, HT, 1, mask, WW, lane, (W - R) * ny, ny, R, T, 1, 0, 0, 1, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (mask_region_any(mask, WW, lane, 0, L, T, B) &&
      !side_scan<SIDE_LEFT>(imgT, HT, 1, mask, WW, lane, (L + 1) * ny, ny, L, T, -1, 0, 0, 1, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (s.left + buf > s.right - buf) return false;
  // top side :705-744 (rows T.. outward, columns L..R); bottom side :746-785
  if (mask_region_any(mask, WW, lane, L, R, 0, T) &&
      !side_scan<SIDE_TOP>(img, 1, W, mask, WW, lane, (T + 1) * nx, nx, L, T, 0, -1, 1, 0, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (mask_region_any(mask, WW, lane, L, R, B, H - 1) &&
      !side_scan<SIDE_BOTTOM>(img, 1, W, mask, WW, lane, (H - B) * nx, nx, L, B, 0, 1, 1, 0, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (s.top + buf > s.bottom - buf) return false;
  PL_T1(t_side, 3);
  PL_T0(t_corner);
  // corners :794-940
  if (mask_region_any(mask, WW, lane, R, W - 1, 0, T) &&
      !corner_scan<CORNER_TR>(img, W, mask, WW, lane, T + 1, W - R, R, T, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (mask_region_any(mask, WW, lane, R, W - 1, B, H - 1) &&
      !corner_scan<CORNER_BR>(img, W, mask, WW, lane, H - B, W - R, R, B, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (mask_region_any(mask, WW, lane, 0, L, 0, T) &&
      !corner_scan<CORNER_TL>(img, W, mask, WW, lane, T + 1, L + 1, L, T, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  if (mask_region_any(mask, WW, lane, 0, L, B, H - 1) &&
      !corner_scan<CORNER_BL>(img, W, mask, WW, lane, H - B, L + 1, L, B, num, buf, x0, y0, dmin, s)) { PL_COUNT(16, 1); return false; }
  PL_T1(t_corner, 4);
  PL_COUNT(5, 1);
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
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The reference walks the four lateral faces one after the other, keeping the deepest crossing
// found so far (increasing section: the largest root in (t0, t1]; decreasing: the smallest in
// [t0, t1)).  That running extreme is order-independent, so the four quartics are solved side by
// side -- lane l takes face l & 3 -- and combined with two shuffles; every lane ends with the result.
__device__ bool deepest_collision_time(const Poly &p, const Section &m, const PlannerPyramid &pyr, int lane,
                                       double &tOut) {
#pragma clang fp contract(off)
  const int f = lane & 3;
  double c[5] = {0, 0, 0, 0, 0};
  for (int dim = 0; dim < 3; dim++) {
    const double nf = pyr.normal[f][dim];
    for (int q = 0; q < 5; q++) c[q] += nf * p.c[q][dim];
  }
  double roots[4];
  unsigned n;
  if (fabs(c[0]) > 1e-6) n = solve_quartic(c[1] / c[0], c[2] / c[0], c[3] / c[0], c[4] / c[0], roots);
  else n = solve_cubic(c[2] / c[1], c[3] / c[1], c[4] / c[1], roots);
  sort_small(roots, (int)n);
  double cand = m.increasing ? m.t0 : m.t1;
  if (m.increasing) {
    for (int i = (int)n - 1; i >= 0; i--) {
      if (roots[i] > m.t1) continue;
      if (roots[i] > m.t0) cand = roots[i];
      break;
    }
  } else {
    for (int i = 0; i < (int)n; i++) {
      if (roots[i] < m.t0) continue;
      if (roots[i] < m.t1) cand = roots[i];
      break;
    }
  }
#pragma unroll
  for (int x = 1; x <= 2; x <<= 1) {
    const double o = shfl_xor_f64(cand, x);
    cand = m.increasing ? (o > cand ? o : cand) : (o < cand ? o : cand);
  }
  tOut = cand;
  return m.increasing ? cand > m.t0 : cand < m.t1;
}

// GetMonotonicSections (DIP.cpp:303-354): a function of the candidate alone
__device__ void monotonic_sections(const Poly &p, double tf, CandSections &out) {
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
  Section sec[5];
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
  out.n = ns;
  out.increasing = 0;
  for (int i = 0; i < 5; i++) {
    out.t[i][0] = i < ns ? sec[i].t0 : 0.0;
    out.t[i][1] = i < ns ? sec[i].t1 : 0.0;
    if (i < ns && sec[i].increasing) out.increasing |= 1u << i;
  }
}

// The plan's pyramid list, sorted by depth (DIP.cpp:269-271), as the wave holds it when a plan may have at most 64
// pyramids (REGKEYS): lane q keeps what FindContainingPyramid reads of the q-th pyramid -- depth and the four edges --
// and the number of its record in HBM (records stay in the order the pyramids were made: nothing is moved on an
// insert; only the normals are ever read back).  "First pyramid at or beyond this depth that contains the pixel"
// is then one comparison per lane and a ballot instead of a walk over up to 64 records in memory, and an insert is
// a shift by one lane.  (Round 5 profile: the walk and the insert's record moves were 15 % of a plan's cycles.)
struct PyrKeys {
  double depth;
  int right, top, left, bottom, slot;
};
__device__ __forceinline__ double shfl_up1_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__shfl_up((int)(unsigned)u, 1), hi = (unsigned)__shfl_up((int)(unsigned)(u >> 32), 1);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// IsCollisionFree (DIP.cpp:214-301) on the candidate's sections
template <bool REGKEYS>
__device__ bool collision_free(const PlannerConfig &cfg, const uint16_t *__restrict__ img,
                               const uint16_t *__restrict__ imgT, int HT, const uint32_t *__restrict__ sums, uint64_t *mask_lds, int lane, const Poly &p,
                               const CandSections &first, PlannerPyramid *pyr, PyrKeys &keys, int &nPyr, int maxPyr) {
#pragma clang fp contract(off)
  Section sec[16];   // pending sections (a std::vector in the reference; 16 like the CPU checker)
  int ns = first.n;
  for (int i = 0; i < 5; i++) {
    sec[i].t0 = first.t[i][0];
    sec[i].t1 = first.t[i][1];
    sec[i].increasing = (first.increasing >> i) & 1u;
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
    PL_T0(t_find);
    if (REGKEYS) {
      // the pyramids nearer than ez are a prefix of the sorted list: "from the first at or beyond ez on" is "not nearer"
      const bool mine = lane < nPyr && !(keys.depth < ez) && keys.left + cfg.pixel_buffer < px && px < keys.right - cfg.pixel_buffer &&
                        keys.top + cfg.pixel_buffer < py && py < keys.bottom - cfg.pixel_buffer;
      const uint64_t holds = __ballot(mine);
      if (holds) at = (int)__ffsll((unsigned long long)holds) - 1;
    } else {
      int first = 0;
      while (first < nPyr && pyr[first].depth < ez) first++;
      for (int q = first; q < nPyr; q++)
        if (pyr[q].left + cfg.pixel_buffer < px && px < pyr[q].right - cfg.pixel_buffer &&
            pyr[q].top + cfg.pixel_buffer < py && py < pyr[q].bottom - cfg.pixel_buffer) { at = q; break; }
    }
    PL_T1(t_find, 17);
    if (at < 0) {
      if (nPyr >= maxPyr) return false;                        // _maxNumPyramids, :255-260
      PlannerPyramid fresh;
      if (!inflate_pyramid(cfg, img, imgT, HT, sums, mask_lds, lane, (int)px, (int)py, ez, fresh)) return false;
      PL_T0(t_ins);
      int idx = 0;                                             // std::lower_bound + insert, :269-271
      if (REGKEYS) {
        idx = __popcll(__ballot(lane < nPyr && keys.depth < fresh.depth));   // (a prefix again)
        const double d_up = shfl_up1_f64(keys.depth);
        const int r_up = __shfl_up(keys.right, 1), t_up = __shfl_up(keys.top, 1), l_up = __shfl_up(keys.left, 1),
                  b_up = __shfl_up(keys.bottom, 1), s_up = __shfl_up(keys.slot, 1);
        if (lane > idx) { keys.depth = d_up; keys.right = r_up; keys.top = t_up; keys.left = l_up; keys.bottom = b_up; keys.slot = s_up; }
        if (lane == idx) { keys.depth = fresh.depth; keys.right = fresh.right; keys.top = fresh.top; keys.left = fresh.left; keys.bottom = fresh.bottom; keys.slot = nPyr; }
        pyr[nPyr] = fresh;          // every lane writes the same record (and reads back only what it wrote itself)
      } else {
        while (idx < nPyr && pyr[idx].depth < fresh.depth) idx++;
        for (int q = nPyr; q > idx; q--) pyr[q] = pyr[q - 1];
        pyr[idx] = fresh;
      }
      nPyr++;
      at = idx;
      PL_T1(t_ins, 18);
    }
    double tcol;
    PL_T0(t_deep);
    const bool hits = deepest_collision_time(p, m, pyr[REGKEYS ? __builtin_amdgcn_readlane(keys.slot, at) : at], lane, tcol);
    PL_T1(t_deep, 19);
    PL_COUNT(20, 1);
    if (hits) {
      if (ns >= 16) return false;
      sec[ns++] = m.increasing ? make_section(p, m.t0, tcol) : make_section(p, tcol, m.t1);
    }
  }
  return true;
}

}  // namespace

// ---- FindLowestCostTrajectory, DIP.cpp:91-212, candidate count instead of a time budget ------
// The reference examines candidates one after the other: cost first, then (only if it beats the
// best so far) input feasibility, velocity, collision.  Cost, input feasibility and velocity
// admissibility depend on nothing but the candidate itself, so they are evaluated for ALL
// candidates of all planners by afe_rappids_candidates_kernel (one lane per candidate), and
// afe_rappids_search_kernel (one wave per planner) then replays the reference's sequential
// decisions on those precomputed answers -- the candidates whose cost does not beat the running
// best are skipped 64 at a time by a ballot -- and runs the collision checks, the only part
// that needs the image, wave-cooperatively.  Flags and counters come out exactly as the
// sequential loop would have produced them.

__device__ __forceinline__ void load_planner_state(const PlannerConfig &cfg, const PlannerBatch &b, int64_t i, Cand &k,
                                                   double cost_vec[3]) {
  for (int a = 0; a < 3; a++) {
    k.v0[a] = b.vel0[a * b.n + i];
    k.a0[a] = b.acc0[a * b.n + i];
    k.grav[a] = b.grav[a * b.n + i];
    cost_vec[a] = b.cost_vec ? b.cost_vec[a * b.n + i] : cfg.cost_vec[a];
  }
}

__device__ __forceinline__ double candidate_cost(const PlannerConfig &cfg, const Cand &k, const double cost_vec[3]) {
#pragma clang fp contract(off)
  const double dur = k.tf;
  const double ex = c_pos(k, 0, dur), ey = c_pos(k, 1, dur), ez = c_pos(k, 2, dur);
  if (cfg.cost_type == 0)            // ExplorationCost::GetCost, DIP.hpp:488-492
    return -(cost_vec[0] * ex + cost_vec[1] * ey + cost_vec[2] * ez) / dur;
  // Simulator/Rappids_Simulator/main.cpp:86-107
  const double gx = cost_vec[0], gy = cost_vec[1], gz = cost_vec[2];
  const double SG = sqrt((gx - 0) * (gx - 0) + (gy - 0) * (gy - 0) + (gz - 0) * (gz - 0));
  const double PiG = sqrt((gx - ex) * (gx - ex) + (gy - ey) * (gy - ey) + (gz - ez) * (gz - ez));
  return -(SG - PiG) / dur;
}

enum { CAND_INPUT_FEASIBLE = 1, CAND_VELOCITY_OK = 2 };

__device__ __forceinline__ void candidate_poly(const Cand &k, Poly &p) {   // RTG.hpp GetTrajectory
#pragma clang fp contract(off)
  for (int a = 0; a < 3; a++) {
    p.c[0][a] = k.al[a] / 120;
    p.c[1][a] = k.be[a] / 24;
    p.c[2][a] = k.ga[a] / 6;
    p.c[3][a] = c_acc(k, a, 0) / 2;
    p.c[4][a] = c_vel(k, a, 0);
    p.c[5][a] = c_pos(k, a, 0);
  }
}

__global__ void __launch_bounds__(256) afe_rappids_candidates_kernel(const PlannerConfig cfg, const PlannerBatch b) {
#pragma clang fp contract(off)
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= b.n * b.n_candidates) return;
  const int64_t i = t / b.n_candidates;
  const int c = (int)(t - i * b.n_candidates);
  const double *sample = b.samples + ((int64_t)(b.sample_table ? b.sample_table[i] : 0) * b.n_candidates + c) * 4;
  Cand k;
  double cost_vec[3], pf[3];
  load_planner_state(cfg, b, i, k, cost_vec);
  deproject(cfg, sample[0], sample[1], sample[2], pf);             // DIP.hpp:393-404
  c_generate(k, pf, sample[3]);
  b.cand_cost[t] = candidate_cost(cfg, k, cost_vec);
  uint8_t bits = 0;
  if (input_feasible(k, cfg)) {
    bits |= CAND_INPUT_FEASIBLE;
    if (velocity_feasible(k, cfg.max_velocity)) {
      bits |= CAND_VELOCITY_OK;
      Poly p;
      candidate_poly(k, p);
      monotonic_sections(p, k.tf, b.cand_sections[t]);
    }
  }
  b.cand_bits[t] = bits;
}

#ifndef AFE_PLANNER_WAVES
#define AFE_PLANNER_WAVES 4
#endif
// REGKEYS: at most 64 pyramids per plan -- the sorted list lives in the wave's lanes (PyrKeys)
template <bool REGKEYS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AFE_PLANNER_WAVES, AFE_PLANNER_WAVES)))
afe_rappids_search_kernel(const PlannerConfig cfg, const PlannerBatch b) {
#pragma clang fp contract(off)
  int64_t i = blockIdx.x;            // one wave per planner
  if (b.ordered) {                   // the finishing round: most remaining work first
    int64_t j = blockIdx.x;
    int k = 0;
    for (; k < PlannerBatch::kBins; k++) {
      const int64_t cnt = b.bin_count[k];
      if (j < cnt) break;
      j -= cnt;
    }
    if (k == PlannerBatch::kBins) return;
    i = b.bin_list[(int64_t)k * b.n + j];
  }
#ifdef AFE_PLANNER_PROFILE
  if (threadIdx.x < 24) s_mine[threadIdx.x] = 0;
  __syncthreads();
#endif
  PL_T0(t_all);
  const int lane = threadIdx.x;
  extern __shared__ uint64_t mask_lds[];   // one bit per pixel of this planner's image, see build_mask
  const int64_t img_off = (int64_t)(b.image_index ? b.image_index[i] : i) * cfg.width * cfg.height;
  const uint16_t *img = b.images + img_off;
  const uint16_t *imgT = b.images_t + (int64_t)(b.image_index ? b.image_index[i] : i) * cfg.width * b.height_t;
  const uint32_t *sums = b.sums ? b.sums + (img_off >> 6) : nullptr;      // one dword per 64 pixels
  const double *samples = b.samples + (int64_t)(b.sample_table ? b.sample_table[i] : 0) * b.n_candidates * 4;
  const double *cand_cost = b.cand_cost + i * b.n_candidates;
  const uint8_t *cand_bits = b.cand_bits + i * b.n_candidates;
  const CandSections *__restrict__ cand_sections = b.cand_sections + i * b.n_candidates;
  PlannerPyramid *pyr = b.pyramids + i * b.max_pyramids;
  PlanOutput *out = b.out + i;
  PyrKeys keys = {0.0, 0, 0, 0, 0, 0};
  int nPyr = 0;
  int n_cost = 0, n_feasible = 0, n_velocity = 0, n_free = 0, best_index = -1;
  double bestCost = 1.7976931348623157e308;
  int start_base = 0, start_lane = 0;
  PlannerBatch::Resume *rs = b.resume ? b.resume + i : nullptr;
  if (rs && b.round > 0) {            // pick up where an earlier round left this planner
    if (rs->done) return;
    start_base = rs->base; start_lane = rs->lane;
    best_index = rs->best_index; n_cost = rs->n_cost; n_feasible = rs->n_feasible; n_velocity = rs->n_velocity; n_free = rs->n_free;
    nPyr = rs->n_pyr; bestCost = rs->best_cost;
    if (REGKEYS && lane < nPyr) {      // the sorted list as the interrupted launch left it
      keys.slot = b.pyr_order[i * 64 + lane];
      const PlannerPyramid &P = pyr[keys.slot];
      keys.depth = P.depth; keys.right = P.right; keys.top = P.top; keys.left = P.left; keys.bottom = P.bottom;
    }
  } else if (lane == 0) {             // the "nothing found" answer; overwritten below
    out->tf = 0;
    for (int q = 0; q < 6; q++) for (int a = 0; a < 3; a++) out->coeffs[q][a] = 0;
  }
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  for (int base = start_base; base < b.n_candidates; base += 64) {
    const int c = base + lane;
    const bool has = c < b.n_candidates;
    const double my_cost = has ? cand_cost[c] : 0.0;
    const unsigned my_bits = has ? cand_bits[c] : 0u;
    unsigned my_result = 0;
    const int first_lane = base == start_base ? start_lane : 0;     // lanes below it were settled by an earlier round
    uint64_t todo = lanes_from(first_lane);
    for (;;) {
      // candidates of this chunk that beat the best cost so far; the first one is the next the
      // sequential loop would look into (the best cost only falls, so the others stay skipped)
      const uint64_t pass = __ballot(has && my_cost < bestCost) & todo;
      if (!pass) break;
      const int l = (int)__ffsll((unsigned long long)pass) - 1;
      const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)my_bits, l);
      unsigned result = 1;
      n_cost++;
      if (bits & CAND_INPUT_FEASIBLE) {
        result |= 2;
        n_feasible++;
        if (bits & CAND_VELOCITY_OK) {
          result |= 4;
          n_velocity++;
          PL_T0(t_regen);
          const double *sample = samples + 4 * (base + l);
          Cand k;
          double cost_vec[3], pf[3];
          load_planner_state(cfg, b, i, k, cost_vec);
          deproject(cfg, sample[0], sample[1], sample[2], pf);
          c_generate(k, pf, sample[3]);
          Poly p;
          candidate_poly(k, p);
          PL_T1(t_regen, 21);
          PL_T0(t_cf);
          const bool cfree = collision_free<REGKEYS>(cfg, img, imgT, b.height_t, sums, mask_lds, lane, p, cand_sections[base + l], pyr, keys, nPyr, b.max_pyramids);
          PL_T1(t_cf, 6);
          if (cfree) {
            result |= 8;
            bestCost = __shfl(my_cost, l);
            n_free++;
            best_index = base + l;
            if (lane == 0) {
              out->tf = k.tf;
              for (int q = 0; q < 6; q++) for (int a = 0; a < 3; a++) out->coeffs[q][a] = p.c[q][a];
            }
          }
        }
      }
      if (lane == l) my_result = result;
      todo = lanes_from(l + 1);
      // out of budget for this round: everything up to candidate base + l is settled; write it down and leave
      if (b.budget_ticks && __builtin_amdgcn_s_memrealtime() - t_begin > (unsigned long long)b.budget_ticks) {
        if (b.flags && has && lane >= first_lane && lane <= l) b.flags[i * b.n_candidates + c] = (uint8_t)my_result;
        if (b.bin_count) {
          // what may still lie ahead: candidates behind this one that are admissible and cheaper than the best so far
          // (each costs a collision check unless a better one is found first) -- the planner's place in the finishing round
          int est = 0;
          for (int b2 = base; b2 < b.n_candidates; b2 += 64) {
            const int c2 = b2 + lane;
            const bool ahead = c2 < b.n_candidates && c2 > base + l &&
                               (cand_bits[c2] & (CAND_INPUT_FEASIBLE | CAND_VELOCITY_OK)) == (CAND_INPUT_FEASIBLE | CAND_VELOCITY_OK) && cand_cost[c2] < bestCost;
            est += __popcll(__ballot(ahead));
          }
          if (lane == 0) {
            const int bin = est >= 160 ? 0 : est >= 96 ? 1 : est >= 48 ? 2 : est >= 24 ? 3 : est >= 12 ? 4 : est >= 6 ? 5 : est >= 1 ? 6 : 7;
            const int at = atomicAdd(b.bin_count + bin, 1);
            b.bin_list[(int64_t)bin * b.n + at] = (int32_t)i;
          }
        }
        if (REGKEYS && lane < nPyr) b.pyr_order[i * 64 + lane] = (uint8_t)keys.slot;
        if (lane == 0) {
          rs->done = 0;
          rs->base = l == 63 ? base + 64 : base;
          rs->lane = l == 63 ? 0 : l + 1;
          rs->best_index = best_index; rs->n_cost = n_cost; rs->n_feasible = n_feasible; rs->n_velocity = n_velocity; rs->n_free = n_free;
          rs->n_pyr = nPyr; rs->best_cost = bestCost;
        }
        return;
      }
    }
    if (b.flags && has && lane >= first_lane) b.flags[i * b.n_candidates + c] = (uint8_t)my_result;
  }
  if (lane == 0) {
    if (rs) rs->done = 1;
    out->found = best_index >= 0;
    out->best_index = best_index;
    out->best_cost = bestCost;
    out->n_generated = b.n_candidates;
    out->n_cost_checks = n_cost;
    out->n_collision_checks = n_feasible;
    out->n_velocity_checks = n_velocity;
    out->n_collision_free = n_free;
    out->n_pyramids = nPyr;
  }
  PL_T1(t_all, 0);
#ifdef AFE_PLANNER_PROFILE
  if (threadIdx.x == 0) {
    const unsigned long long mine = __builtin_readcyclecounter() - t_all;
    if (atomicMax(&g_prof[8], mine) < mine) {
      for (int q = 0; q < 24; q++) g_longest[q] = s_mine[q];
      g_longest[8] = mine; g_longest[12] = (unsigned long long)n_cost; g_longest[13] = (unsigned long long)n_velocity; g_longest[14] = (unsigned long long)nPyr; g_longest[15] = (unsigned long long)i;
    }
  }
#endif
}

// images [n][H][W] -> images_t [n][W][H] through 64 x 64 LDS tiles, and -- `sums` given (width % 64 == 0) -- the
// per-word summaries of PlannerBatch::sums on the way: a tile row IS one 64-pixel word of the bit images.  With one image
// per planner (config 3: 65 536 images, 10 GB in, 10 GB out) this pass is as long as the search itself, so it moves
// bytes the wide way: a wave reads a row segment as 32 dwords (two pixels each) and writes a transposed column segment of
// 64 pixels as 32 dwords -- 128-byte runs on both sides (round 5's 32 x 32 tiles wrote 64-byte runs of single pixels:
// 2.1 TB/s; round 6: see DESIGN.md).  Even width and height take the dword path, anything else pixel by pixel.
__global__ void __launch_bounds__(256) afe_prepare_images_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                                 uint32_t *__restrict__ sums, int W, int H, int HT, unsigned ignore) {
  __shared__ uint16_t tile[64][66];
  const int64_t off = (int64_t)blockIdx.z * W * H, off_t = (int64_t)blockIdx.z * W * HT;
  const int t = threadIdx.x;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const bool even = ((W | H) & 1) == 0;
  if (even) {
    const int c2 = (t & 31) * 2, x = x0 + c2;
    for (int r = t >> 5; r < 64; r += 8) {
      const int y = y0 + r;
      uint32_t two = 0;
      if (x < W && y < H) two = *reinterpret_cast<const uint32_t *>(src + off + (int64_t)y * W + x);
      *reinterpret_cast<uint32_t *>(&tile[r][c2]) = two;
    }
  } else {
    const int tx = t & 63, x = x0 + tx;
    for (int r = t >> 6; r < 64; r += 4) {
      const int y = y0 + r;
      tile[r][tx] = (x < W && y < H) ? src[off + (int64_t)y * W + x] : (uint16_t)0;
    }
  }
  __syncthreads();
  if (even) {
    const int r2 = (t & 31) * 2, yo = y0 + r2;
    for (int c = t >> 5; c < 64; c += 8) {
      const int xo = x0 + c;
      if (xo < W && yo < H)
        *reinterpret_cast<uint32_t *>(dst + off_t + (int64_t)xo * HT + yo) = (uint32_t)tile[r2][c] | ((uint32_t)tile[r2 + 1][c] << 16);
    }
  } else {
    const int ty = t & 63, yo = y0 + ty;
    for (int c = t >> 6; c < 64; c += 4) {
      const int xo = x0 + c;
      if (xo < W && yo < H) dst[off_t + (int64_t)xo * HT + yo] = tile[ty][c];
    }
  }
  if (sums) {
    // rows of the tile, eight lanes with eight pixels each, 32 rows per pass
    for (int r = t >> 3; r < 64; r += 32) {
      const int seg = t & 7, y = y0 + r;
      unsigned mn = 0xffffu, mx = 0u;
      bool ign = false;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const unsigned d = tile[r][8 * seg + j];
        if (d > ignore) mn = d < mn ? d : mn; else ign = true;
        mx = d > mx ? d : mx;
      }
      unsigned packed = mn | ((ign ? 0xffffu : mx) << 16);
#pragma unroll
      for (int m = 1; m <= 4; m <<= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)packed, m);
        const unsigned lo = (o & 0xffffu) < (packed & 0xffffu) ? (o & 0xffffu) : (packed & 0xffffu);
        const unsigned hi = (o >> 16) > (packed >> 16) ? (o >> 16) : (packed >> 16);
        packed = lo | (hi << 16);
      }
      if (seg == 0 && y < H) sums[((int64_t)blockIdx.z * H + y) * (W >> 6) + blockIdx.x] = packed;
    }
  }
}

int launch_rappids(const PlannerConfig &cfg, const PlannerBatch &b, void *stream) {
  if (b.n <= 0) return 0;
  // grid.z is limited to 65535: one launch per run of that many images (config 3 has one image per planner, 65536)
  const unsigned ignore = (unsigned)(uint16_t)(cfg.true_vehicle_radius / cfg.depth_scale);     // as inflate_pyramid forms it
  for (int64_t i0 = 0; i0 < b.n_images; i0 += 65535) {
    const int64_t cnt = (b.n_images - i0) < 65535 ? (b.n_images - i0) : 65535;
    const int64_t off = i0 * cfg.width * cfg.height;
    hipLaunchKernelGGL(afe_prepare_images_kernel, dim3((cfg.width + 63) / 64, (cfg.height + 63) / 64, (unsigned)cnt),
                       dim3(256), 0, (hipStream_t)stream, b.images + off, b.images_t + i0 * cfg.width * b.height_t, b.sums ? b.sums + (off >> 6) : nullptr,
                       cfg.width, cfg.height, b.height_t, ignore);
  }
  // the bit image, and behind it the list of undecided words of build_mask_sum
  const unsigned mask_bytes = (unsigned)(((cfg.width + 63) >> 6) * cfg.height) * (unsigned)sizeof(uint64_t) + (b.sums ? kSumList * (unsigned)sizeof(uint16_t) : 0u);
#ifdef AFE_PLANNER_PROFILE
  unsigned long long zero[40] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof(zero));
#endif
  const int64_t n_cand = b.n * b.n_candidates;
  hipLaunchKernelGGL(afe_rappids_candidates_kernel, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     cfg, b);
  // A launch lasts as long as its longest planner, and planners differ by two orders of magnitude.  Two uses of the
  // interruptible search (PlannerBatch::Resume; exact by construction,
  // tests/test_gpu_planner.py::test_search_in_budgeted_rounds_is_the_uninterrupted_search) against that tail were
  // built and measured on 65 536 planners (profiles/r03_planner_rounds.txt, r03_planner_lpt.txt):
  //   * rounds with doubling budgets: SLOWER than one launch (cluttered orchard views 245 against 210 ms) -- the long
  //     planners wait for every round's queue of short ones.  Off; AFE_PLANNER_ROUNDS_FROM=<planners> switches it on
  //     (a host that time-slices planning against other work).
  //   * longest first (below): 210 -> 152 ms on the cluttered views, 50.6 -> 29 ms on the config-3 shape.
  PlannerBatch bb = b;
  bb.ordered = 0;
  void (*search)(const PlannerConfig, const PlannerBatch) = (b.max_pyramids <= 64 && b.pyr_order) ? afe_rappids_search_kernel<true> : afe_rappids_search_kernel<false>;
  int64_t rounds_from = INT64_MAX;
  if (const char *env = afe_dev_env("AFE_PLANNER_ROUNDS_FROM")) rounds_from = std::strtoll(env, nullptr, 10);
  // Longest first: a short sizing round (every planner works for at most `sizing` microseconds; most finish), then ONE
  // finishing round that starts the interrupted planners in the order of the work they may still have -- the launch
  // then ends when the work runs out, not when a long planner that happened to start late does.
  int64_t lpt_from = 16384;           // beyond four times what the chip holds at once (4 096 waves); measured: profiles/r03_planner_lpt.txt
  unsigned sizing_us = 400;
  if (const char *env = afe_dev_env("AFE_PLANNER_LPT_FROM")) lpt_from = std::strtoll(env, nullptr, 10);
  if (const char *env = afe_dev_env("AFE_PLANNER_SIZING_US")) sizing_us = (unsigned)std::strtoul(env, nullptr, 10);
  if (b.resume && b.bin_count && b.n > lpt_from && b.n <= rounds_from) {
    (void)hipMemsetAsync(b.bin_count, 0, PlannerBatch::kBins * sizeof(int32_t), (hipStream_t)stream);
    bb.round = 0; bb.budget_ticks = sizing_us * 100u;
    hipLaunchKernelGGL(search, dim3((unsigned)b.n), dim3(64), mask_bytes, (hipStream_t)stream, cfg, bb);
    bb.round = 1; bb.budget_ticks = 0; bb.ordered = 1;
    hipLaunchKernelGGL(search, dim3((unsigned)b.n), dim3(64), mask_bytes, (hipStream_t)stream, cfg, bb);
  } else if (!b.resume || b.n <= rounds_from) {
    bb.resume = nullptr; bb.budget_ticks = 0; bb.round = 0; bb.bin_count = nullptr;
    hipLaunchKernelGGL(search, dim3((unsigned)b.n), dim3(64), mask_bytes, (hipStream_t)stream, cfg, bb);
  } else {
    bb.bin_count = nullptr;
    unsigned budgets_us[16] = {1000, 2000, 4000, 8000, 16000, 32000};
    int n_rounds = 6;
    if (const char *env = afe_dev_env("AFE_PLANNER_ROUNDS_US")) {     // measurement aid: "500,1000,..." (a last unlimited round is always added)
      n_rounds = 0;
      for (const char *q = env; *q && n_rounds < 16;) {
        budgets_us[n_rounds++] = (unsigned)std::strtoul(q, nullptr, 10);
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
      }
    }
    for (int r = 0; r <= n_rounds; r++) {
      bb.round = r;
      bb.budget_ticks = r < n_rounds ? budgets_us[r] * 100u : 0u;
      hipLaunchKernelGGL(search, dim3((unsigned)b.n), dim3(64), mask_bytes, (hipStream_t)stream, cfg, bb);
    }
  }
#ifdef AFE_PLANNER_PROFILE
  unsigned long long prof[40];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof), sizeof(prof));
  fprintf(stderr, "planner profile (cycles per planner): total %.0f | collision_free %.0f | inflate: mask1 %.0f expansion(all) %.0f "
          "sides+mask2 %.0f corners %.0f | completed pyramids/planner %.2f ringloop %.0f | longest planner %.0f | scan chunks/planner %.0f, holding a marked pixel %.0f\n", (double)prof[0] / b.n, (double)prof[6] / b.n,
          (double)prof[1] / b.n, (double)prof[2] / b.n, (double)prof[3] / b.n, (double)prof[4] / b.n, (double)prof[5] / b.n, (double)prof[7] / b.n, (double)prof[8], (double)prof[9] / b.n, (double)prof[10] / b.n);
  fprintf(stderr, "  per planner (scoped, early exits included): inflate %.0f | seed check %.0f | max-depth sweep %.0f | mask2 + scans %.0f\n",
          (double)prof[24] / b.n, (double)prof[25] / b.n, (double)prof[26] / b.n, (double)prof[27] / b.n);
  fprintf(stderr, "  per planner: inflate calls %.2f, refused at the seed rectangle %.2f, refused inside a scan %.2f\n", (double)prof[22] / b.n, (double)prof[23] / b.n, (double)prof[16] / b.n);
  fprintf(stderr, "  per planner: regenerate %.0f | sections(unused) %.0f | find pyramid %.0f | insert %.0f | section quartics %.0f (%.1f of them)\n",
          (double)prof[21] / b.n, (double)prof[16] / b.n, (double)prof[17] / b.n, (double)prof[18] / b.n, (double)prof[19] / b.n, (double)prof[20] / b.n);
  (void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_longest), sizeof(prof));
  fprintf(stderr, "  longest: regenerate %llu | sections %llu | find pyramid %llu | insert %llu | section quartics %llu (%llu of them)\n",
          prof[21], prof[16], prof[17], prof[18], prof[19], prof[20]);
  fprintf(stderr, "  longest planner #%llu: total %llu | collision_free %llu | mask1 %llu expansion(all) %llu sides+mask2 %llu corners %llu | pyramids inflated %llu "
          "(kept %llu) | cost checks %llu, collision checks %llu | scan chunks %llu\n", prof[15], prof[8], prof[6], prof[1], prof[2], prof[3], prof[4], prof[5], prof[14], prof[12], prof[13], prof[9]);
#endif
  return (int)hipGetLastError();
}

}  // namespace afe
