// planned_trajectory.hpp -- what the Rappids_Simulator loop asks of the trajectory the planner returned
// (Simulator/Rappids_Simulator/main.cpp:558-608): position, velocity and acceleration along it, the thrust that
// flies it and the body rates that turn the thrust direction along it.  The planner hands over the winner as the
// six coefficient vectors of CommonMath::Trajectory (t^5 .. t^0, Trajectory.hpp:31-36; afe_plan_output::coeffs)
// in the camera-fixed frame it planned in, together with the gravity vector of that frame.
//   thrust(t)  = |a(t) - g|                                   RapidTrajectoryGenerator.hpp:192-194
//   normal(t)  = (a(t) - g) / |a(t) - g|                      :187-189
//   omega(t,h) = acos(n(t) . n(t+h)) / h about n(t) x n(t+h)   RapidTrajectoryGenerator.cpp:264-286
#pragma once
#include <cerrno>
#include <cmath>

#include "agrifly/standalone_types.hpp"

namespace agrifly_cli {

struct PlannedTrajectory {
  double c[6][3];
  double duration;
  Vec3d gravity;

  Vec3d Position(double t) const { return Eval(t, 0); }
  Vec3d Velocity(double t) const { return Eval(t, 1); }
  Vec3d Acceleration(double t) const { return Eval(t, 2); }
  double Thrust(double t) const { return (Acceleration(t) - gravity).GetNorm2(); }
  Vec3d Normal(double t) const {
    const Vec3d f = Acceleration(t) - gravity;
    return f / f.GetNorm2();
  }
  Vec3d Omega(double t, double step) const {
    const Vec3d n0 = Normal(t), n1 = Normal(t + step);
    const Vec3d turn = n0.Cross(n1);
    const double len = turn.GetNorm2();
    if (len <= 1e-6) return Vec3d(0, 0, 0);
    errno = 0;
    const double rate = std::acos(n0.Dot(n1)) / step;
    if (errno) return Vec3d(0, 0, 0);
    return rate * (turn / len);
  }

 private:
  Vec3d Eval(double t, int derivative) const {
    double out[3];
    for (int a = 0; a < 3; a++) {
      const double k5 = c[0][a], k4 = c[1][a], k3 = c[2][a], k2 = c[3][a], k1 = c[4][a], k0 = c[5][a];
      if (derivative == 0) out[a] = ((((k5 * t + k4) * t + k3) * t + k2) * t + k1) * t + k0;
      else if (derivative == 1) out[a] = (((5 * k5 * t + 4 * k4) * t + 3 * k3) * t + 2 * k2) * t + k1;
      else out[a] = ((20 * k5 * t + 12 * k4) * t + 6 * k3) * t + 2 * k2;
    }
    return Vec3d(out[0], out[1], out[2]);
  }
};

}  // namespace agrifly_cli
