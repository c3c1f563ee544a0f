// agrifly/Wire.hpp -- C++ conveniences over the C ABI's wire formats for hosts
// that do not have agri-fly's Common/DataTypes headers (inside the agri-fly tree
// keep using RadioTypes.hpp / TelemetryPacket.hpp: the bytes are identical).
//   RawRadioMessage   == RadioTypes::RadioMessageDecoded::RawMessage (23 bytes)
//   DelayLine<T>      == Simulation::CommunicationsDelay<T>: release a message
//                        once now >= enqueue time + uint64(delay * 1e6) us, oldest
//                        first, one per poll (Components/Components/Simulation/
//                        CommunicationsDelay.hpp:10-52; main.cpp:282,673,737-739)
#pragma once
#include <stdint.h>

#include <deque>
#include <utility>

#include "../agrifly_engine.h"
#ifndef AGRIFLY_USE_REFERENCE_TYPES
#include "standalone_types.hpp"
#endif

namespace agrifly {

struct RawRadioMessage {
  uint8_t raw[AFE_RADIO_PACKET_SIZE];
};

inline RawRadioMessage MakeRatesCommand(uint8_t flags, float desTotalThrust, Vec3f desAngVel) {
  RawRadioMessage m;
  const float w[3] = {desAngVel.x, desAngVel.y, desAngVel.z};
  afe_radio_create_rates_command(flags, desTotalThrust, w, m.raw);
  return m;
}

inline afe_radio_message Decode(const RawRadioMessage &m) {
  afe_radio_message out;
  afe_radio_decode(m.raw, &out);
  return out;
}

template <class Message>
class DelayLine {
 public:
  DelayLine(BaseTimer *const clock, double delaySeconds)
      : clock_(clock), delay_us_(uint64_t(delaySeconds * 1e6)) {}
  void AddMessage(const Message &msg) { pending_.push_back(std::make_pair(clock_->GetMicroSeconds() + delay_us_, msg)); }
  bool HaveNewMessage() const { return !pending_.empty() && clock_->GetMicroSeconds() >= pending_.front().first; }
  Message GetMessage() {
    Message out = pending_.front().second;
    pending_.pop_front();
    return out;
  }

 private:
  BaseTimer *const clock_;
  const uint64_t delay_us_;
  std::deque<std::pair<uint64_t, Message> > pending_;
};

}  // namespace agrifly
