// hover_controller.hpp -- the caller-side position / attitude controller of the Rappids_Simulator loop, restated for
// the headless program (see mocap_estimator.hpp for why restated); float, like the reference.
#pragma once
#include <cmath>

#include "agrifly/standalone_types.hpp"

namespace agrifly_cli {

// Offboard::QuadcopterController::Run, QuadcopterController.cpp:11-74, with the position controller
// of Logic/QuadcopterPositionController.hpp:22-28 and the attitude controller of
// Logic/QuadcopterAttitudeController.hpp:35-68 -- float, like the reference.
struct HoverController {
  float natFreq, damping, tc_xy, tc_z;
  float minVerticalProperAcceleration, maxProperAcc, minProperAcc;
  HoverController() : natFreq(2.0f), damping(0.7f), tc_xy(0.08f), tc_z(0.4f),   // QuadcopterConstants.hpp:214-226
                      minVerticalProperAcceleration(0.5f * 9.81f), maxProperAcc(20), minProperAcc(-1) {}

  Vec3f GetDesiredAngularVelocity(const Rotationf desAttitude, const Rotationf estAttitude) const {
    Rotationf errAtt = (desAttitude.Inverse() * estAttitude);
    const Vec3f desRotVec = errAtt.ToRotationVector();
    Vec3f desRedAttRotAx = Vec3f(errAtt.Inverse() * Vec3f(0, 0, 1)).Cross(Vec3f(0, 0, 1));
    float desRedAttRotAn_cos = Vec3f(errAtt.Inverse() * Vec3f(0, 0, 1)).Dot(Vec3f(0, 0, 1));
    float desRedAttRotAn;
    if (desRedAttRotAn_cos >= 1.0f) desRedAttRotAn = 0;
    else if (desRedAttRotAn_cos <= -1.0f) desRedAttRotAn = float(M_PI);
    else desRedAttRotAn = acosf(desRedAttRotAn_cos);
    float n = desRedAttRotAx.GetNorm2();
    if (n < 1e-12f) desRedAttRotAx = Vec3f(0, 0, 0);
    else desRedAttRotAx = desRedAttRotAx / n;
    float k3 = (1.0f / tc_z);
    float k12 = (1.0f / tc_xy);
    return -k3 * desRotVec - (k12 - k3) * desRedAttRotAn * desRedAttRotAx;
  }

  void Run(Vec3d const curPos, Vec3d const curVel, Rotationd const curAtt, Vec3d const desPos, Vec3d const desVel,
           Vec3d const desAcc, double const desiredYawAngle, Vec3d &outCmdAngVel, double &outCmdThrust) const {
    Vec3f const e3(0, 0, 1);
    Vec3f const cmdAcc = (Vec3f(desPos) - Vec3f(curPos)) * natFreq * natFreq +
                         (Vec3f(desVel) - Vec3f(curVel)) * 2 * natFreq * damping + Vec3f(desAcc);
    Vec3f cmdProperAcc = cmdAcc + Vec3f(0, 0, 9.81f);
    if (cmdProperAcc.GetNorm2() > maxProperAcc) cmdProperAcc *= maxProperAcc / cmdProperAcc.GetNorm2();
    if (cmdProperAcc.z < minVerticalProperAcceleration) cmdProperAcc.z = minVerticalProperAcceleration;
    float const normCmdProperAcc = cmdProperAcc.GetNorm2();
    Vec3f const cmdThrustDir = cmdProperAcc / normCmdProperAcc;
    outCmdThrust = normCmdProperAcc * (Rotationf(curAtt) * Vec3f(0, 0, 1)).Dot(cmdThrustDir);
    if (outCmdThrust < minProperAcc) outCmdThrust = minProperAcc;
    Rotationf cmdAtt;
    const float cosAngle = cmdThrustDir.Dot(e3);
    float angle;
    if (cosAngle >= (1 - 1e-12f)) angle = 0;
    else if (cosAngle <= -(1 - 1e-12f)) angle = float(M_PI);
    else angle = acosf(cosAngle);
    Vec3f rotAx = e3.Cross(cmdThrustDir);
    const float n = rotAx.GetNorm2();
    if (n < 1e-6f) cmdAtt = Rotationf::Identity();
    else cmdAtt = Rotationf::FromRotationVector(rotAx * (angle / n));
    Rotationf cmdAttYawed = cmdAtt * Rotationf::FromRotationVector(Vec3f(0, 0, float(desiredYawAngle)));
    outCmdAngVel = Vec3d(GetDesiredAngularVelocity(cmdAttYawed, Rotationf(curAtt)));
  }
};

}  // namespace agrifly_cli
