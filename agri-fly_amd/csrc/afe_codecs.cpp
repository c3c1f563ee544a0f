// afe_codecs.cpp -- host-only wire formats either side of the vehicle step
// (SURVEY.md 8f row f2): the 23-byte uplink radio message and the 30-byte
// telemetry packets.  Integer / byte work: bit-exact with the reference's
// Common/Common/DataTypes/RadioTypes.hpp and TelemetryPacket.hpp.  No HIP.
#include <cmath>
#include <cstring>
#include <limits>

#include "../../include/agrifly_engine.h"

namespace {

// RadioTypes.hpp:39-71
enum { IDX_TYPE = 0, IDX_RESERVED = 1, IDX_FLAGS = 2, IDX_FLOATS = 3, ENC_SIZE = 2, ENC_MAX = 1 << 16,
       ENC_HALF = ENC_MAX / 2, NUM_FLOATS = 10, RAW_SIZE = IDX_FLOATS + ENC_SIZE * NUM_FLOATS };
static_assert(RAW_SIZE == AFE_RADIO_PACKET_SIZE, "radio packet size");
enum { MAX_THRUST = 35, MAX_RATES = 35, MAX_POS = 20, MAX_VEL = 10, MAX_ACC = 30, MAX_DEFAULT = 1 };
enum { T_INVALID = 0, T_KILL = 2, T_POSITION = 3, T_ACCELERATION = 4, T_RATES = 5, T_IDLE = 6 };  // RadioTypes.hpp:17-25

// encodeToRadioByte, RadioTypes.hpp:73-101
void encode16(float valIn, float limit, unsigned indx, uint8_t *bytes) {
  int out;
  if ((valIn > -limit) && (valIn < limit)) out = int(valIn * ENC_HALF / limit + 0.5f) + ENC_HALF;
  else if (valIn > -limit) out = ENC_MAX - 1;
  else if (valIn < limit) out = 0;
  else out = 0;  // NaN
  for (int i = 0; i < ENC_SIZE; i++) {
    if (indx + i >= RAW_SIZE) break;
    bytes[indx + i] = (out >> ((ENC_SIZE - i - 1) * 8)) % 256;
  }
}
// decodeFromRadioBytes, RadioTypes.hpp:103-116
float decode16(const uint8_t *bytesIn, unsigned indx, float limit) {
  int out = 0;
  for (int i = 0; i < ENC_SIZE; i++) {
    if (indx + i >= RAW_SIZE) break;
    out += bytesIn[indx + i] << ((ENC_SIZE - 1 - i) * 8);
  }
  return limit * (out - ENC_HALF) / float(ENC_HALF);
}

// TelemetryPacket.hpp:38-73
float map_to_ones(float x, float a, float b) { return ((x - a) / (b - a)) * 2 - 1; }
float map_to_ab(float x, float a, float b) { return ((x + 1) / 2) * (b - a) + a; }
uint16_t encode_ones(float t) {
  uint16_t out;
  if (t < -1 || t > 1) out = 0;
  else out = 32768 + 32767 * t;
  return out;
}
float decode_ones(uint16_t t) {
  if (t == 0) return std::numeric_limits<float>::quiet_NaN();
  return (t - 32768) / float(32768);
}
// TelemetryPacket.hpp:82-100
enum { R_ACC = 30, R_GYRO = 35, R_FORCE_MAX = 10, R_FORCE_MIN = 0, R_BATT_MAX = 15, R_BATT_MIN = 0, R_POS = 30,
       R_VEL = 30, R_ATT = 1, R_GENERIC = 100 };

void put16(uint8_t *p, int slot, uint16_t v) { std::memcpy(p + 2 + 2 * slot, &v, 2); }  // data_packet_t is packed
uint16_t get16(const uint8_t *p, int slot) { uint16_t v; std::memcpy(&v, p + 2 + 2 * slot, 2); return v; }

}  // namespace

extern "C" int afe_radio_create_rates_command(uint8_t flags, float des_total_thrust, const float des_ang_vel[3],
                                              uint8_t raw_out[AFE_RADIO_PACKET_SIZE]) {
  if (!des_ang_vel || !raw_out) return AFE_ERR_INVALID_ARG;
  // CreateRatesCommand, RadioTypes.hpp:158-171 (bytes it does not write are left alone there; zeroed here)
  std::memset(raw_out, 0, RAW_SIZE);
  raw_out[IDX_TYPE] = T_RATES;
  raw_out[IDX_RESERVED] = 0;
  raw_out[IDX_FLAGS] = flags;
  encode16(des_total_thrust, MAX_THRUST, IDX_FLOATS, raw_out);
  for (int i = 0; i < 3; i++) encode16(des_ang_vel[i], MAX_RATES, IDX_FLOATS + (i + 1) * ENC_SIZE, raw_out);
  return AFE_OK;
}

extern "C" int afe_radio_create_position_command(uint8_t flags, const float pos[3], const float vel[3],
                                                 const float acc[3], uint8_t raw_out[AFE_RADIO_PACKET_SIZE]) {
  if (!pos || !vel || !acc || !raw_out) return AFE_ERR_INVALID_ARG;
  std::memset(raw_out, 0, RAW_SIZE);  // CreatePositionCommand, RadioTypes.hpp:137-156
  raw_out[IDX_TYPE] = T_POSITION;
  raw_out[IDX_FLAGS] = flags;
  for (int i = 0; i < 3; i++) {
    encode16(pos[i], MAX_POS, IDX_FLOATS + (0 + i) * ENC_SIZE, raw_out);
    encode16(vel[i], MAX_VEL, IDX_FLOATS + (3 + i) * ENC_SIZE, raw_out);
    encode16(acc[i], MAX_ACC, IDX_FLOATS + (6 + i) * ENC_SIZE, raw_out);
  }
  return AFE_OK;
}

extern "C" int afe_radio_create_acceleration_command(uint8_t flags, const float acc[3], float yaw_rate,
                                                     uint8_t raw_out[AFE_RADIO_PACKET_SIZE]) {
  if (!acc || !raw_out) return AFE_ERR_INVALID_ARG;
  std::memset(raw_out, 0, RAW_SIZE);  // CreateAccelerationCommand, RadioTypes.hpp:173-187
  raw_out[IDX_TYPE] = T_ACCELERATION;
  raw_out[IDX_FLAGS] = flags;
  for (int i = 0; i < 3; i++) encode16(acc[i], MAX_ACC, IDX_FLOATS + i * ENC_SIZE, raw_out);
  encode16(yaw_rate, MAX_RATES, IDX_FLOATS + 3 * ENC_SIZE, raw_out);
  return AFE_OK;
}

extern "C" int afe_radio_create_simple_command(int type, uint8_t flags, uint8_t raw_out[AFE_RADIO_PACKET_SIZE]) {
  if (!raw_out || (type != T_KILL && type != T_IDLE)) return AFE_ERR_INVALID_ARG;
  std::memset(raw_out, 0, RAW_SIZE);  // CreateKillCommand / CreateIdleCommand, RadioTypes.hpp:123-135
  raw_out[IDX_TYPE] = (uint8_t)type;
  raw_out[IDX_FLAGS] = flags;
  return AFE_OK;
}

extern "C" int afe_radio_decode(const uint8_t raw[AFE_RADIO_PACKET_SIZE], afe_radio_message *out) {
  if (!raw || !out) return AFE_ERR_INVALID_ARG;
  // RadioMessageDecoded(raw), RadioTypes.hpp:189-240
  out->type = raw[IDX_TYPE];
  out->flags = raw[IDX_FLAGS];
  for (int i = 0; i < NUM_FLOATS; i++) out->floats[i] = 0.0f;  // the reference leaves unused fields uninitialised
  switch (out->type) {
    case T_POSITION:
      for (int i = 0; i < 3; i++) out->floats[i] = decode16(raw, IDX_FLOATS + i * ENC_SIZE, MAX_POS);
      for (int i = 3; i < 6; i++) out->floats[i] = decode16(raw, IDX_FLOATS + i * ENC_SIZE, MAX_VEL);
      for (int i = 6; i < 9; i++) out->floats[i] = decode16(raw, IDX_FLOATS + i * ENC_SIZE, MAX_ACC);
      break;
    case T_RATES:
      out->floats[0] = decode16(raw, IDX_FLOATS, MAX_THRUST);
      for (int i = 1; i < NUM_FLOATS; i++) out->floats[i] = decode16(raw, IDX_FLOATS + i * ENC_SIZE, MAX_RATES);
      break;
    case T_ACCELERATION:
      for (int i = 0; i < 3; i++) out->floats[i] = decode16(raw, IDX_FLOATS + i * ENC_SIZE, MAX_ACC);
      out->floats[3] = decode16(raw, IDX_FLOATS + 3 * ENC_SIZE, MAX_RATES);
      break;
    default:
      for (int i = 0; i < NUM_FLOATS; i++) out->floats[i] = decode16(raw, IDX_FLOATS + i * ENC_SIZE, MAX_DEFAULT);
      break;
  }
  return AFE_OK;
}

extern "C" int afe_telemetry_encode(const afe_telemetry_packet *src, uint8_t out[AFE_TELEMETRY_PACKET_SIZE]) {
  if (!src || !out) return AFE_ERR_INVALID_ARG;
  // EncodeTelemetryPacket, TelemetryPacket.hpp:122-166; data_packet_t = {u8 type, u8 packetNumber, u16 data[14]} packed
  out[0] = src->type;
  out[1] = src->packet_number;
  if (src->type == 0) {
    for (int i = 0; i < 3; i++) {
      put16(out, i + 0, encode_ones(map_to_ones(src->accel[i], -R_ACC, R_ACC)));
      put16(out, i + 3, encode_ones(map_to_ones(src->gyro[i], -R_GYRO, R_GYRO)));
    }
    for (int i = 0; i < 4; i++) put16(out, i + 6, encode_ones(map_to_ones(src->motor_forces[i], R_FORCE_MIN, R_FORCE_MAX)));
    for (int i = 0; i < 3; i++) put16(out, i + 10, encode_ones(map_to_ones(src->position[i], -R_POS, R_POS)));
    put16(out, 13, encode_ones(map_to_ones(src->batt_voltage, R_BATT_MIN, R_BATT_MAX)));
  } else if (src->type == 1) {
    for (int i = 0; i < 3; i++) {
      put16(out, i + 0, encode_ones(map_to_ones(src->velocity[i], -R_VEL, R_VEL)));
      put16(out, i + 3, encode_ones(map_to_ones(src->attitude[i], -R_ATT, R_ATT)));
    }
    for (int i = 0; i < 6; i++) put16(out, i + 6, encode_ones(map_to_ones(src->debug_vals[i], -R_GENERIC, R_GENERIC)));
    std::memcpy(out + 2 + 2 * 12, &src->panic_reason, 1);  // memcpy(&out.data[12], &src.panicReason, 1)
    std::memcpy(out + 2 + 2 * 13, &src->warnings, 1);
  }
  return AFE_OK;
}

extern "C" int afe_telemetry_decode(const uint8_t in[AFE_TELEMETRY_PACKET_SIZE], afe_telemetry_packet *out) {
  if (!in || !out) return AFE_ERR_INVALID_ARG;
  // DecodeTelemetryPacket, TelemetryPacket.hpp:168-207 (only the fields of in.type are written)
  out->type = in[0];
  out->packet_number = in[1];
  if (in[0] == 0) {
    for (int i = 0; i < 3; i++) {
      out->accel[i] = map_to_ab(decode_ones(get16(in, i + 0)), -R_ACC, R_ACC);
      out->gyro[i] = map_to_ab(decode_ones(get16(in, i + 3)), -R_GYRO, R_GYRO);
    }
    for (int i = 0; i < 4; i++) out->motor_forces[i] = map_to_ab(decode_ones(get16(in, i + 6)), R_FORCE_MIN, R_FORCE_MAX);
    for (int i = 0; i < 3; i++) out->position[i] = map_to_ab(decode_ones(get16(in, i + 10)), -R_POS, R_POS);
    out->batt_voltage = map_to_ab(decode_ones(get16(in, 13)), R_BATT_MIN, R_BATT_MAX);
  } else if (in[0] == 1) {
    for (int i = 0; i < 3; i++) {
      out->velocity[i] = map_to_ab(decode_ones(get16(in, i + 0)), -R_VEL, R_VEL);
      out->attitude[i] = map_to_ab(decode_ones(get16(in, i + 3)), -R_ATT, R_ATT);
    }
    for (int i = 0; i < 6; i++) out->debug_vals[i] = map_to_ab(decode_ones(get16(in, i + 6)), -R_GENERIC, R_GENERIC);
    std::memcpy(&out->panic_reason, in + 2 + 2 * 12, 1);
    std::memcpy(&out->warnings, in + 2 + 2 * 13, 1);
  }
  return AFE_OK;
}
