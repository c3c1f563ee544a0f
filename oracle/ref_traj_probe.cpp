// oracle/ref_traj_probe.cpp -- TEST INFRASTRUCTURE.
// Known answers from the REFERENCE's own planner math that compiles stand-alone:
//   Common/Common/Math/RootFinder.hpp (needs <math.h> only)
//   Components/Components/TrajectoryGenerator/SingleAxisTrajectory.{hpp,cpp}
// plus libstdc++'s std::mt19937 + std::uniform_real_distribution<> in the shape of
// the planner's candidate generator (DepthImagePlanner.hpp:393-404: three draws as
// function arguments, then one more), to record this compiler's evaluation order.
// usage: traj_probe <n> <seed> -> one JSON object
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdint.h>
#include "Common/Math/RootFinder.hpp"
#include "Components/TrajectoryGenerator/SingleAxisTrajectory.hpp"
#include "Components/TrajectoryGenerator/SingleAxisTrajectory.cpp"

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
static double uni(uint32_t &s, double lo, double hi) { return lo + (hi - lo) * (double)(lcg(s) >> 8) / 16777216.0; }

struct P3 { double x, y, d; };
static P3 deproject(double x, double y, double depth) { P3 p = {x, y, depth}; return p; }

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 32;
  uint32_t s = argc > 2 ? (uint32_t) atol(argv[2]) : 1u;
  printf("{\"cubic\": [\n");
  for (int k = 0; k < n; k++) {
    double a = uni(s, -5, 5), b = uni(s, -5, 5), c = uni(s, -5, 5);
    if (k % 7 == 0) { a = 3; b = 3; c = 1; }            // triple root
    if (k % 7 == 1) { a = -4; b = 5; c = -2; }          // double root
    double x[3] = {0, 0, 0};
    unsigned cnt = RootFinder::solve_cubic<double>(a, b, c, x);
    printf(" {\"in\": [%.17g, %.17g, %.17g], \"count\": %u, \"x\": [%.17g, %.17g, %.17g]}%s\n", a, b, c, cnt, x[0], x[1], x[2], k + 1 < n ? "," : "");
  }
  printf("], \"quartic\": [\n");
  for (int k = 0; k < n; k++) {
    double a = uni(s, -5, 5), b = uni(s, -8, 8), c = uni(s, -8, 8), d = uni(s, -5, 5);
    if (k % 5 == 0) { a = -10; b = 35; c = -50; d = 24; }   // roots 1 2 3 4
    if (k % 5 == 1) { a = 0; b = -5; c = 0; d = 4; }        // +-1 +-2
    double r[4] = {0, 0, 0, 0};
    unsigned cnt = RootFinder::solve_quartic<double>(a, b, c, d, r);
    printf(" {\"in\": [%.17g, %.17g, %.17g, %.17g], \"count\": %u, \"x\": [%.17g, %.17g, %.17g, %.17g]}%s\n", a, b, c, d, cnt, r[0], r[1], r[2], r[3], k + 1 < n ? "," : "");
  }
  printf("], \"axis\": [\n");
  using RapidQuadrocopterTrajectoryGenerator::SingleAxisTrajectory;
  for (int k = 0; k < n; k++) {
    SingleAxisTrajectory ax;
    const double p0 = 0, v0 = uni(s, -3, 3), a0 = uni(s, -4, 4), pf = uni(s, -3, 3), tf = uni(s, 2, 3);
    ax.SetInitialState(p0, v0, a0);
    ax.SetGoalPosition(pf); ax.SetGoalVelocity(0); ax.SetGoalAcceleration(0);
    ax.GenerateTrajectory(tf);
    double t1 = uni(s, 0, 1), t2 = t1 + uni(s, 0.02, 2), amin, amax;
    if (t2 > tf) t2 = tf;
    ax.GetMinMaxAcc(amin, amax, t1, t2);
    double jsq = ax.GetMaxJerkSquared(t1, t2);
    printf(" {\"v0\": %.17g, \"a0\": %.17g, \"pf\": %.17g, \"tf\": %.17g, \"alpha\": %.17g, \"beta\": %.17g, \"gamma\": %.17g, \"cost\": %.17g,"
           " \"t1\": %.17g, \"t2\": %.17g, \"amin\": %.17g, \"amax\": %.17g, \"jmaxsq\": %.17g, \"pos_tf\": %.17g, \"vel_half\": %.17g}%s\n",
           v0, a0, pf, tf, ax.GetParamAlpha(), ax.GetParamBeta(), ax.GetParamGamma(), ax.GetCost(), t1, t2, amin, amax, jsq,
           ax.GetPosition(tf), ax.GetVelocity(0.5 * tf), k + 1 < n ? "," : "");
  }
  printf("], \"mt19937\": {");
  for (int seed = 0; seed < 2; seed++) {
    std::mt19937 gen(seed == 0 ? 0u : 20261002u);
    std::uniform_real_distribution<> px(0.1 * 320, 0.9 * 320), py(0.1 * 240, 0.9 * 240), dep(1.5, 3.0), tim(2.0, 3.0);
    printf("\"seed%d\": [", seed);
    for (int k = 0; k < 8; k++) {
      P3 p = deproject(px(gen), py(gen), dep(gen));   // the reference's call shape
      double t = tim(gen);
      printf("%s[%.17g, %.17g, %.17g, %.17g]", k ? ", " : "", p.x, p.y, p.d, t);
    }
    printf("], ");
  }
  { std::mt19937 g(0u); printf("\"raw0\": [%lu, %lu, %lu, %lu]}}\n", (unsigned long) g(), (unsigned long) g(), (unsigned long) g(), (unsigned long) g()); }
  return 0;
}
