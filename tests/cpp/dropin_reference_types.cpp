// Compile-and-link check of the façade's AGRIFLY_USE_REFERENCE_TYPES branch against agri-fly's OWN headers
// (tests/test_dropin_reference_headers.py builds it; -I /root/reference/Common, -I /root/reference/Components,
// -I tests/shim for <Eigen/Dense>, -DAGRIFLY_USE_REFERENCE_TYPES).  This is INTEGRATION.md section 2's snippet made whole:
// the vehicle of Simulator/Rappids_Simulator/main.cpp:146-218 constructed with the SAME argument list, but as
// agrifly::Quadcopter_T<Onboard::QuadcopterLogic>, held the way the loops hold it (main.cpp:211;
// AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:83) and driven through every member of the upper seam the two mains
// call.  It is never RUN by the CPU suite (it needs an MI355X); it exists so that a signature drift between the façade
// and the reference's SimulationObject6DOF / Quadcopter_T / logicType concept fails a build.  Pins nothing.
#include <memory>

#include "Common/Time/ManualTimer.hpp"
#include "Components/Logic/QuadcopterLogic.hpp"
#include "Components/Simulation/CommunicationsDelay.hpp"
#include "Components/Simulation/SimulationObject6DOF.hpp"
#include <Eigen/Dense>

#include "agrifly/Quadcopter_T.hpp"

namespace Simulation {
// what the tree's own typedef (Quadcopter_T.hpp:134) becomes in a tree that takes the engine
typedef agrifly::Quadcopter_T<Onboard::QuadcopterLogic> Quadcopter;
}

int main() {
  ManualTimer simTimer;                                                     // main.cpp:143
  // main.cpp:146-177, verbatim in meaning
  uint8_t vehicleId = 1;
  Onboard::QuadcopterConstants::QuadcopterType quadcopterType = Onboard::QuadcopterConstants::GetVehicleTypeFromID(vehicleId);
  Onboard::QuadcopterConstants vehConsts(quadcopterType);
  double const mass = vehConsts.mass;
  double const inertia_xx = vehConsts.inertia_xx;
  double const inertia_yy = inertia_xx;
  double const inertia_zz = vehConsts.inertia_zz;
  double armLength = vehConsts.armLength;
  double propThrustFromSpeedSqr = vehConsts.propellerThrustFromSpeedSqr;
  double propTorqueFromSpeedSqr = vehConsts.propellerTorqueFromThrust * vehConsts.propellerThrustFromSpeedSqr;
  double motorTimeConst = vehConsts.motorTimeConst;
  double motorInertia = vehConsts.motorInertia;
  double motorMinSpeed = vehConsts.motorMinSpeed;
  double motorMaxSpeed = vehConsts.motorMaxSpeed;
  Vec3d centreOfMassError = Vec3d(0, 0, 0);
  double const periodOnboardLogic = 1.0 / 500.0;
  double const timeDelayOffboardControlLoopTrue = 0.03;
  // main.cpp:203-209
  Eigen::Matrix<double, 3, 3> inertiaMatrix;
  inertiaMatrix << inertia_xx, 0, 0, 0, inertia_yy, 0, 0, 0, inertia_zz;
  Vec3d linDragCoeffB = Vec3d(vehConsts.linDragCoeffBx, vehConsts.linDragCoeffBy, vehConsts.linDragCoeffBz);

  // main.cpp:211-218: the same fifteen arguments in the same order
  std::shared_ptr<Simulation::Quadcopter> quad;
  quad.reset(new Simulation::Quadcopter(&simTimer, mass, inertiaMatrix, armLength, centreOfMassError, motorMinSpeed, motorMaxSpeed,
                                        propThrustFromSpeedSqr, propTorqueFromSpeedSqr, motorTimeConst, motorInertia, linDragCoeffB,
                                        vehicleId, quadcopterType, periodOnboardLogic));
  // AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:83: what SimVehicle::vehicle is
  std::shared_ptr<Simulation::SimulationObject6DOF> vehicle = quad;

  // main.cpp:266-270: the command channel carries the reference's RawMessage by value
  Simulation::CommunicationsDelay<RadioTypes::RadioMessageDecoded::RawMessage> cmdRadioChannel(&simTimer, timeDelayOffboardControlLoopTrue);

  // main.cpp:279-280
  vehicle->SetPosition(Vec3d(0.1, -0.2, 0));
  vehicle->SetAttitude(Rotationd::FromEulerYPR(0.1, 0, 0));
  vehicle->SetVelocity(Vec3d(0, 0, 0));
  vehicle->SetAngularVelocity(Vec3d(0, 0, 0));

  double sum = 0;
  for (int step = 0; step < 20; step++) {
    vehicle->Run();                                                         // main.cpp:391
    simTimer.AdvanceMicroSeconds(1000);                                     // main.cpp:392
    Vec3d simTruthPos(vehicle->GetPosition());                              // main.cpp:394-395
    Rotationd simTruthAtt(vehicle->GetAttitude());
    Vec3d vel = vehicle->GetVelocity(), angVel = vehicle->GetAngularVelocity();
    sum += simTruthPos.z + simTruthAtt[0] + vel.x + angVel.y + simTruthAtt.ToEulerYPR().x;
    TelemetryPacket::data_packet_t dataPacketRaw1, dataPacketRaw2;          // main.cpp:462, 666
    vehicle->GetTelemetryDataPackets(dataPacketRaw1, dataPacketRaw2);
    sum += dataPacketRaw1.type + dataPacketRaw2.packetNumber;
    Vec3d accMeasIMU, gyroMeasIMU;                                          // Simulator/main.cpp:444-446
    vehicle->GetAccelerometer(accMeasIMU);
    vehicle->GetRateGyro(gyroMeasIMU);
    sum += accMeasIMU.z + gyroMeasIMU.x;
    if (step % 10 == 0) {                                                   // main.cpp:700-733: a rates command goes up
      RadioTypes::RadioMessageDecoded::RawMessage rawMsg;
      RadioTypes::RadioMessageDecoded::CreateRatesCommand(0, 9.81f, Vec3f(0, 0, 0), rawMsg.raw);
      cmdRadioChannel.AddMessage(rawMsg);
    }
    if (cmdRadioChannel.HaveNewMessage()) vehicle->SetCommandRadioMsg(cmdRadioChannel.GetMessage());   // main.cpp:737-739
  }
  // the members a host reaches through the concrete type (Quadcopter_T.hpp:39-59)
  quad->SetExternalForce(Vec3d(0, 0, 0.01));
  quad->SetExternalTorque(Vec3d(0, 0, 0));
  sum += quad->GetMotorForce(0);
  Vec3f estPos, estVel, estAngVel;
  Rotationf estAtt;
  quad->GetEstimate(estPos, estVel, estAtt, estAngVel);
  vehicle->AddUWBRadioTarget(2, Vec3f(1, 2, 3));
  std::shared_ptr<Simulation::UWBRadio> radio = vehicle->GetRadio();
  sum += radio ? radio->GetId() : 0;
  return sum != sum ? 1 : 0;
}
