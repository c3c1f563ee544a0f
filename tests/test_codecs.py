"""Wire formats either side of the step, through the C ABI.
Telemetry is pinned against the reference's own TelemetryPacket.hpp (compiled in
place by oracle/_ref/telemetry_probe -> tests/golden/telemetry_kat.json); the
radio codec (RadioTypes.hpp needs Eigen through Vec3.hpp: unbuildable here) is a
restatement checked against a second, independent restatement in numpy
(tests/offboard_stub.py) and hand-derived byte patterns.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests.offboard_stub import radio_quantise


def _tp(afa, case, typ):
    t = afa.TelemetryPacket()
    t.type = typ
    t.packet_number = case["packetNumber"]
    for name, field in (("accel", "accel"), ("gyro", "gyro"), ("motorForces", "motor_forces"),
                        ("position", "position"), ("velocity", "velocity"), ("attitude", "attitude"),
                        ("debugVals", "debug_vals")):
        for i, v in enumerate(case[name]):
            getattr(t, field)[i] = v
    t.batt_voltage = case["battVoltage"][0]
    t.panic_reason, t.warnings = case["panicReason"], case["warnings"]
    return t


def test_telemetry_bytes_match_reference_header(afa, golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "telemetry_kat.json")))
    assert kat["sizeof_data_packet"] == afa.TELEMETRY_PACKET_SIZE == 30
    L = afa.library()
    n_out_of_range = 0
    for case in kat["cases"]:
        for typ, key in ((0, "pt1"), (1, "pt2")):
            t = _tp(afa, case, typ)
            out = np.zeros(30, np.uint8)
            assert L.afe_telemetry_encode(C.byref(t), out.ctypes.data) == 0
            want = np.array(case[key], np.uint8)
            # the reference leaves the high bytes of data[12], data[13] of part 2 unwritten (memset 0 in the probe)
            np.testing.assert_array_equal(out, want, err_msg=key)
        d = afa.TelemetryPacket()
        for key in ("pt1", "pt2"):
            raw = np.array(case[key], np.uint8)
            assert L.afe_telemetry_decode(raw.ctypes.data, C.byref(d)) == 0
        dec = case["decoded"]
        for name, field in (("accel", "accel"), ("gyro", "gyro"), ("motorForces", "motor_forces"),
                            ("position", "position"), ("velocity", "velocity"), ("attitude", "attitude"),
                            ("debugVals", "debug_vals")):
            for i, v in enumerate(dec[name]):
                got = getattr(d, field)[i]
                if v is None:                      # out-of-range value -> code 0 -> NaN
                    assert np.isnan(got)
                    n_out_of_range += 1
                else:
                    assert np.float32(got) == np.float32(v), (name, i)
        assert d.panic_reason == dec["panicReason"] and d.warnings == dec["warnings"]
    assert n_out_of_range > 10     # the fixture exercises the saturation path


def test_radio_rates_roundtrip_and_layout(afa):
    raw = afa.radio_create_rates_command(0x02, 9.81, [0.5, -0.25, 1.0])
    assert raw[0] == 5 and raw[1] == 0 and raw[2] == 0x02 and len(raw) == 23
    # big-endian 16-bit: 9.81 -> int(9.81*32768/35 + .5) + 32768 = 41952 = 0xA3E0
    code = int(np.float32(9.81) * np.float32(32768) / np.float32(35) + np.float32(0.5)) + 32768
    assert (int(raw[3]) << 8 | int(raw[4])) == code
    m = afa.radio_decode(raw)
    assert m.type == 5 and m.flags == 2
    want = np.concatenate([radio_quantise(np.float32([9.81]), 35), radio_quantise(np.float32([0.5, -0.25, 1.0]), 35)])
    np.testing.assert_array_equal(np.float32(list(m.floats)[:4]), want)
    # fields 4..9 of a rates packet were never written: decode as code 0 -> -35
    assert all(f == -35.0 for f in list(m.floats)[4:])


def test_radio_codec_against_independent_restatement(afa):
    rng = np.random.default_rng(8)
    vals = np.concatenate([rng.uniform(-40, 40, 4000), [0, 35, -35, 34.9999, -34.9999, np.nan, np.inf, -np.inf]]).astype(np.float32)
    for k in range(0, len(vals) - 3, 4):
        v = vals[k:k + 4]
        m = afa.radio_decode(afa.radio_create_rates_command(0, v[0], v[1:4]))
        np.testing.assert_array_equal(np.float32(list(m.floats)[:4]), radio_quantise(v, 35), err_msg=str(v))


def test_radio_other_message_types(afa):
    L = afa.library()
    raw = np.zeros(23, np.uint8)
    p, v, a = np.float32([1, -2, 3]), np.float32([0.5, 0, -9.99]), np.float32([0, 29.9, -31])
    assert L.afe_radio_create_position_command(1, p.ctypes.data, v.ctypes.data, a.ctypes.data, raw.ctypes.data) == 0
    m = afa.radio_decode(raw)
    assert m.type == 3
    np.testing.assert_array_equal(np.float32(list(m.floats)[:9]),
                                  np.concatenate([radio_quantise(p, 20), radio_quantise(v, 10), radio_quantise(a, 30)]))
    assert L.afe_radio_create_acceleration_command(0, a.ctypes.data, C.c_float(2.5), raw.ctypes.data) == 0
    m = afa.radio_decode(raw)
    assert m.type == 4
    np.testing.assert_array_equal(np.float32(list(m.floats)[:4]),
                                  np.concatenate([radio_quantise(a, 30), radio_quantise(np.float32([2.5]), 35)]))
    for t in (2, 6):
        assert L.afe_radio_create_simple_command(t, 0x80, raw.ctypes.data) == 0
        m = afa.radio_decode(raw)
        assert m.type == t and m.flags == 0x80
    assert L.afe_radio_create_simple_command(5, 0, raw.ctypes.data) == 1


def test_cpp_delay_line_matches_reference_communications_delay(golden_dir):
    """include/agrifly/Wire.hpp DelayLine vs the reference CommunicationsDelay.hpp
    behaviour recorded in tests/golden/timer_cadence.json (reference header compiled in place)"""
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp, "test_wire"])
    for c in json.load(open(os.path.join(golden_dir, "timer_cadence.json")))["cases"]:
        out = json.loads(subprocess.check_output([os.path.join(cpp, "test_wire"), str(c["advance_us"]),
                                                  str(len(c["radio_delivered"]))]))
        assert out["radio_delivered"] == c["radio_delivered"], c["loop_dt_expr"]
    assert out["type"] == 5 and abs(out["thrust"] - 9.81) < 35 / 32768 and out["bytes"][0] == 5
