// tests/cpp/test_wire.cpp -- host-only (no GPU): agrifly/Wire.hpp's DelayLine against
// the cadence the reference's CommunicationsDelay shows in the committed fixture
// (tests/golden/timer_cadence.json "radio_delivered": one message enqueued every
// 10 loop iterations, 30 ms delay), and the radio helpers.  Prints JSON.
#include <cstdio>
#include <cstdlib>

#include "agrifly/Wire.hpp"

int main(int argc, char **argv) {
  const uint64_t advance_us = argc > 1 ? (uint64_t)atoll(argv[1]) : 1000;
  const int runs = argc > 2 ? atoi(argv[2]) : 40;
  ManualTimer simTimer;
  agrifly::DelayLine<int> radio(&simTimer, 0.03);
  std::printf("{\"radio_delivered\": [");
  for (int s = 0; s < runs; s++) {
    simTimer.AdvanceMicroSeconds(advance_us);
    if (s % 10 == 0) radio.AddMessage(s);
    int got = -1;
    if (radio.HaveNewMessage()) got = radio.GetMessage();
    std::printf("%s%d", s ? ", " : "", got);
  }
  agrifly::RawRadioMessage m = agrifly::MakeRatesCommand(0, 9.81f, Vec3f(0.5f, -0.25f, 1.0f));
  afe_radio_message d = agrifly::Decode(m);
  std::printf("], \"type\": %d, \"thrust\": %.9g, \"wz\": %.9g, \"bytes\": [", d.type, d.floats[0], d.floats[3]);
  for (int i = 0; i < AFE_RADIO_PACKET_SIZE; i++) std::printf("%s%u", i ? ", " : "", m.raw[i]);
  std::printf("]}\n");
  return 0;
}
