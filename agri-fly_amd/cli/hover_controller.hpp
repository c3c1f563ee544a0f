// hover_controller.hpp -- the caller-side position / attitude controller of the Rappids_Simulator loop, restated for
// the headless program (see mocap_estimator.hpp for why restated).
//
// What it restates, in float like the sources and in their operation order:
//   * a PD law on position gives the acceleration to ask for (Logic/QuadcopterPositionController.hpp:22-28);
//     gravity is added, the result limited to 20 m/s^2 and to a vertical part of at least g / 2, and split into a
//     thrust magnitude -- projected on where the vehicle's z axis points now -- and a thrust direction
//     (Offboard/QuadcopterController.cpp:19-45);
//   * the attitude that tilts e3 onto that direction by the shortest rotation, then the commanded yaw (:47-70);
//   * body rates that close the attitude error with time constant tc_z about the full error rotation plus
//     (1/tc_xy - 1/tc_z) on the reduced, tilt-only part (Logic/QuadcopterAttitudeController.hpp:35-68).
// tests/test_mocap_estimator.py holds every output float against the numpy restatement of tests/offboard_stub.py.
#pragma once
#include <cmath>

#include "agrifly/standalone_types.hpp"

namespace agrifly_cli {

struct HoverController {
  // CF_MINIQUAD tuning (Logic/QuadcopterConstants.hpp:214-226) and the limits of QuadcopterController.cpp:5-9
  // The attitude time constants are DERIVED in the reference, in float: attControl_timeConst_xy = 0.04f * 2 and
  // attControl_timeConst_z = (0.04f * 5) * 2 -- and 0.04f * 5 rounds to 0.199999988, so the yaw constant is
  // 0.399999976, one ulp below 0.4f.  With the literal 0.4f every rate command is an ulp off and, a few seconds
  // into a flight, one 16-bit radio code differs from the reference's.
  float wn = 2.0f, zeta = 0.7f, tcTilt = 0.04f * 2, tcYaw = (0.04f * 5) * 2;
  float leastVertical = 0.5f * 9.81f, mostThrust = 20, leastThrust = -1;

  // angle between two unit vectors whose cosine is `c`, with the ends of acosf's domain pinned
  static float AngleFromCosine(float c, float oneish) {
    if (c >= oneish) return 0;
    if (c <= -oneish) return float(M_PI);
    return acosf(c);
  }

  Vec3f RatesFor(const Rotationf &wanted, const Rotationf &have) const {
    const Rotationf err = wanted.Inverse() * have;
    const Vec3f whole = err.ToRotationVector();
    const Vec3f upInErr = err.Inverse() * Vec3f(0, 0, 1);
    Vec3f tiltAxis = upInErr.Cross(Vec3f(0, 0, 1));
    const float tilt = AngleFromCosine(upInErr.Dot(Vec3f(0, 0, 1)), 1.0f);
    const float len = tiltAxis.GetNorm2();
    tiltAxis = len < 1e-12f ? Vec3f(0, 0, 0) : tiltAxis / len;
    const float gYaw = 1.0f / tcYaw, gTilt = 1.0f / tcTilt;
    return -gYaw * whole - (gTilt - gYaw) * tilt * tiltAxis;
  }

  // same argument list as Offboard::QuadcopterController::Run
  void Run(Vec3d const curPos, Vec3d const curVel, Rotationd const curAtt, Vec3d const desPos, Vec3d const desVel,
           Vec3d const desAcc, double const desiredYawAngle, Vec3d &outCmdAngVel, double &outCmdThrust) const {
    const Vec3f up(0, 0, 1);
    const Rotationf attNow(curAtt);
    Vec3f pull = (Vec3f(desPos) - Vec3f(curPos)) * wn * wn + (Vec3f(desVel) - Vec3f(curVel)) * 2 * wn * zeta + Vec3f(desAcc);
    pull = pull + Vec3f(0, 0, 9.81f);
    if (pull.GetNorm2() > mostThrust) pull *= mostThrust / pull.GetNorm2();
    if (pull.z < leastVertical) pull.z = leastVertical;
    const float size = pull.GetNorm2();
    const Vec3f along = pull / size;
    outCmdThrust = size * (attNow * up).Dot(along);
    if (outCmdThrust < leastThrust) outCmdThrust = leastThrust;

    const float turn = AngleFromCosine(along.Dot(up), 1 - 1e-12f);
    const Vec3f axis = up.Cross(along);
    const float axisLen = axis.GetNorm2();
    const Rotationf tilted = axisLen < 1e-6f ? Rotationf::Identity() : Rotationf::FromRotationVector(axis * (turn / axisLen));
    const Rotationf wanted = tilted * Rotationf::FromRotationVector(Vec3f(0, 0, float(desiredYawAngle)));
    outCmdAngVel = Vec3d(RatesFor(wanted, attNow));
  }

  // same argument list as Offboard::QuadcopterController::RunTracking (QuadcopterController.cpp:76-132): the
  // reference trajectory's thrust and body rates are fed forward, the PD law only closes the position error
  // (its acceleration feed-forward is deliberately zero there), and the attitude that points the thrust along
  // refAcc + correction + g is tracked with the same rate law as in Run
  void RunTracking(Vec3d const curPos, Vec3d const curVel, Rotationd const curAtt, Vec3d const refPos, Vec3d const refVel,
                   Vec3d const refAcc, double const desiredYawAngle, double const refThrust, Vec3d const refAngVel,
                   Vec3d &outCmdAngVel, double &outCmdThrust, Rotationf &outCmdAtt) const {
    const Vec3f up(0, 0, 1);
    const Rotationf attNow(curAtt);
    const Vec3f correction = (Vec3f(refPos) - Vec3f(curPos)) * wn * wn + (Vec3f(refVel) - Vec3f(curVel)) * 2 * wn * zeta + Vec3f(0, 0, 0);
    outCmdThrust = refThrust + correction.Dot(attNow * up);
    const Vec3f proper = Vec3f(refAcc) + correction + Vec3f(0, 0, 9.81f);
    const Vec3f along = proper / proper.GetNorm2();
    const float turn = AngleFromCosine(along.Dot(up), 1 - 1e-12f);
    const Vec3f axis = up.Cross(along);
    const float axisLen = axis.GetNorm2();
    const Rotationf tilted = axisLen < 1e-6f ? Rotationf::Identity() : Rotationf::FromRotationVector(axis * (turn / axisLen));
    outCmdAtt = tilted * Rotationf::FromRotationVector(Vec3f(0, 0, float(desiredYawAngle)));
    outCmdAngVel = refAngVel + Vec3d(RatesFor(outCmdAtt, attNow));
  }
};

}  // namespace agrifly_cli
