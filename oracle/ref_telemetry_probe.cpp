// oracle/ref_telemetry_probe.cpp -- TEST INFRASTRUCTURE.
// Known answers from the REFERENCE's own telemetry codec, compiled from where it
// lies (Common/Common/DataTypes/TelemetryPacket.hpp needs only libc).
// usage: telemetry_probe <n> <seed>  -> JSON list of {packet fields, bytes pt1, bytes pt2, decoded}
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <stdint.h>
#include "Common/DataTypes/TelemetryPacket.hpp"

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }
static float uni(uint32_t &s, float lo, float hi) { return lo + (hi - lo) * (float)(lcg(s) >> 8) / 16777216.0f; }

static void dump_bytes(const TelemetryPacket::data_packet_t &p) {
  const uint8_t *b = (const uint8_t *) &p;
  printf("[");
  for (unsigned i = 0; i < sizeof(p); i++) printf("%s%u", i ? "," : "", b[i]);
  printf("]");
}
static void dump_floats(const char *name, const float *f, int n, bool last = false) {
  printf("\"%s\": [", name);
  for (int i = 0; i < n; i++) {
    if (std::isnan(f[i])) printf("%snull", i ? "," : ""); else printf("%s%.9g", i ? "," : "", f[i]);
  }
  printf("]%s", last ? "" : ", ");
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16;
  uint32_t s = argc > 2 ? (uint32_t) atol(argv[2]) : 1u;
  printf("{\"sizeof_data_packet\": %u, \"cases\": [\n", (unsigned) sizeof(TelemetryPacket::data_packet_t));
  for (int k = 0; k < n; k++) {
    TelemetryPacket::TelemetryPacket t;
    // ranges deliberately exceed the codec's limits now and then (-> 0 / NaN)
    for (int i = 0; i < 3; i++) {
      t.accel[i] = uni(s, -33, 33); t.gyro[i] = uni(s, -38, 38); t.position[i] = uni(s, -32, 32);
      t.velocity[i] = uni(s, -32, 32); t.attitude[i] = uni(s, -1.05f, 1.05f);
    }
    for (int i = 0; i < 4; i++) t.motorForces[i] = uni(s, -0.5f, 10.5f);
    for (int i = 0; i < 6; i++) t.debugVals[i] = uni(s, -105, 105);
    t.battVoltage = uni(s, -0.5f, 15.5f);
    t.packetNumber = (uint8_t) (lcg(s) >> 24);
    t.panicReason = (uint8_t) (lcg(s) >> 29);
    t.warnings = (uint8_t) (lcg(s) >> 24);
    if (k == 0) { for (int i = 0; i < 3; i++) { t.accel[i] = 0; t.gyro[i] = 0; } t.battVoltage = 7.5f; }
    TelemetryPacket::data_packet_t p1, p2;
    memset(&p1, 0, sizeof(p1)); memset(&p2, 0, sizeof(p2));
    t.type = TelemetryPacket::PACKET_TYPE_QUAD_TELEMETRY_PT1;
    TelemetryPacket::EncodeTelemetryPacket(t, p1);
    t.type = TelemetryPacket::PACKET_TYPE_QUAD_TELEMETRY_PT2;
    TelemetryPacket::EncodeTelemetryPacket(t, p2);
    TelemetryPacket::TelemetryPacket d;
    memset(&d, 0, sizeof(d));
    TelemetryPacket::DecodeTelemetryPacket(p1, d);
    TelemetryPacket::DecodeTelemetryPacket(p2, d);
    printf(" {");
    dump_floats("accel", t.accel, 3); dump_floats("gyro", t.gyro, 3); dump_floats("motorForces", t.motorForces, 4);
    dump_floats("position", t.position, 3); dump_floats("battVoltage", &t.battVoltage, 1);
    dump_floats("velocity", t.velocity, 3); dump_floats("attitude", t.attitude, 3); dump_floats("debugVals", t.debugVals, 6);
    printf("\"packetNumber\": %u, \"panicReason\": %u, \"warnings\": %u, ", t.packetNumber, t.panicReason, t.warnings);
    printf("\"pt1\": "); dump_bytes(p1); printf(", \"pt2\": "); dump_bytes(p2); printf(", \"decoded\": {");
    dump_floats("accel", d.accel, 3); dump_floats("gyro", d.gyro, 3); dump_floats("motorForces", d.motorForces, 4);
    dump_floats("position", d.position, 3); dump_floats("battVoltage", &d.battVoltage, 1);
    dump_floats("velocity", d.velocity, 3); dump_floats("attitude", d.attitude, 3); dump_floats("debugVals", d.debugVals, 6, true);
    printf(", \"panicReason\": %u, \"warnings\": %u}}%s\n", d.panicReason, d.warnings, k + 1 < n ? "," : "");
  }
  printf("]}\n");
  return 0;
}
