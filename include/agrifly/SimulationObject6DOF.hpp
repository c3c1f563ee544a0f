// agrifly/SimulationObject6DOF.hpp -- the abstract Vehicle API the reference's loops hold.
//
// Inside the agri-fly tree (AGRIFLY_USE_REFERENCE_TYPES) this header is only a forwarder to the
// tree's own Components/Simulation/SimulationObject6DOF.hpp, so that agrifly::Quadcopter_T derives
// from THE class `std::shared_ptr<Simulation::SimulationObject6DOF> vehicle` refers to
// (AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:83, Simulator/Rappids_Simulator/main.cpp:211).
//
// Outside the tree it declares classes of the same names, namespaces, members and virtuals:
//   Simulation::SimulationObject      Components/Components/Simulation/SimulationObject.hpp:9-22
//   Simulation::UWBRadio              Components/Components/Simulation/UWBRadio.hpp:17-100
//   Simulation::SimulationObject6DOF  Components/Components/Simulation/SimulationObject6DOF.hpp:12-85
// and the two payload types that cross the seam, byte-compatible with the reference's:
//   RadioTypes::RadioMessageDecoded::RawMessage   Common/Common/DataTypes/RadioTypes.hpp:39-71
//   TelemetryPacket::data_packet_t                Common/Common/DataTypes/TelemetryPacket.hpp:32-36
// A host written against the reference's Vehicle API compiles against these unchanged.
#pragma once

#ifdef AGRIFLY_USE_REFERENCE_TYPES
// (the tree's header names std::shared_ptr without including <memory> -- its own includers happen to have done so,
// Quadcopter_T.hpp:4; found by tests/test_dropin_reference_headers.py, which compiles this branch against the tree)
#include <stdint.h>

#include <memory>

#include "Components/Simulation/SimulationObject6DOF.hpp"
#else
#include <stdint.h>

#include <memory>

#include "../agrifly_engine.h"
#include "standalone_types.hpp"

namespace RadioTypes {
struct RadioMessageDecoded {
  enum { RAW_PACKET_SIZE = AFE_RADIO_PACKET_SIZE };   // 3 + 2 * 10 = 23, RadioTypes.hpp:41-52
  struct RawMessage {
    uint8_t raw[RAW_PACKET_SIZE];
  };
};
}  // namespace RadioTypes

namespace TelemetryPacket {
struct data_packet_t {   // TelemetryPacket.hpp:32-36
  uint8_t type;
  uint8_t packetNumber;
  uint16_t data[14];
} __attribute__((packed));
static_assert(sizeof(data_packet_t) == AFE_TELEMETRY_PACKET_SIZE, "telemetry packet is 30 bytes on the wire");
}  // namespace TelemetryPacket

namespace Simulation {

class SimulationObject {
 public:
  SimulationObject(BaseTimer *const timer) : _integrationTimer(timer) {}
  virtual ~SimulationObject() {}
  virtual void Run() = 0;

 protected:
  Timer _integrationTimer;
};

class UWBRadio : public SimulationObject {
 public:
  struct RangingMeasurement {
    bool haveNew;
    float range;
    uint8_t responderId;
    bool failure;
  };
  UWBRadio(BaseTimer *const timer, uint8_t myId)
      : SimulationObject(timer), _myUWBId(myId), _nextUWBRangingTargetId(0), _uwbTruePosition() {
    _meas.haveNew = false;
  }
  virtual ~UWBRadio() {}
  virtual void Run() {}
  void SetPosition(Vec3d const in) { _uwbTruePosition = in; }
  void SetNextRangingTarget(uint8_t id) { _nextUWBRangingTargetId = id; }
  Vec3d GetPosition() const { return _uwbTruePosition; }
  void SetMeasurement(RangingMeasurement meas) { _meas = meas; _meas.haveNew = true; }
  bool GetHaveNewMeasurement(void) { return _meas.haveNew; }
  RangingMeasurement GetMeasurement(void) {
    RangingMeasurement outMeas = _meas;
    _meas.haveNew = false;
    return outMeas;
  }
  uint8_t GetId() const { return _myUWBId; }
  uint8_t GetNextRangingTargetId() const { return _nextUWBRangingTargetId; }

 private:
  uint8_t _myUWBId;
  uint8_t _nextUWBRangingTargetId;
  Vec3d _uwbTruePosition;
  RangingMeasurement _meas;
};

class SimulationObject6DOF : public SimulationObject {
 public:
  SimulationObject6DOF(BaseTimer *const timer)
      : SimulationObject(timer), _pos(0, 0, 0), _vel(0, 0, 0), _att(Rotationd::Identity()), _angVel(0, 0, 0) {}
  virtual ~SimulationObject6DOF() {}

  virtual void Run() = 0;

  // non-virtual, on the base's own members -- exactly as in the reference
  Vec3d GetPosition() const { return _pos; }
  Vec3d GetVelocity() const { return _vel; }
  Rotationd GetAttitude() const { return _att; }
  Vec3d GetAngularVelocity() const { return _angVel; }
  void SetPosition(Vec3d in) { _pos = in; }
  void SetVelocity(Vec3d in) { _vel = in; }
  void SetAttitude(Rotationd in) { _att = in; }
  void SetAngularVelocity(Vec3d in) { _angVel = in; }

  virtual std::shared_ptr<UWBRadio> GetRadio() { return _radio; }
  virtual void AddUWBRadioTarget(uint8_t id, Vec3f pos) = 0;
  virtual void SetCommandRadioMsg(RadioTypes::RadioMessageDecoded::RawMessage const raw) = 0;
  virtual void GetTelemetryDataPackets(TelemetryPacket::data_packet_t &dataPacket1,
                                       TelemetryPacket::data_packet_t &dataPacket2) = 0;
  virtual void GetAccelerometer(Vec3d &acc) = 0;
  virtual void GetRateGyro(Vec3d &rateGyro) = 0;

 protected:
  Vec3d _pos;
  Vec3d _vel;
  Rotationd _att;   // <vector in world frame> = _att * <vector in body frame>
  Vec3d _angVel;
  std::shared_ptr<UWBRadio> _radio;
};

}  // namespace Simulation
#endif
