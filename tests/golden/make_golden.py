#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

Run in the build container (needs /root/reference for the timer probe):

    python tests/golden/make_golden.py

* timer_cadence.json  <- oracle/_ref/timer_probe: the REFERENCE's own
  Timer / ManualTimer / CommunicationsDelay headers compiled from where they
  lie under /root/reference (they need no Eigen) and driven through the call
  sequence of Quadcopter_T::Run inside the Rappids_Simulator loop.
* rng_kat.json        <- oracle/_ref/rng_probe: libstdc++'s
  std::default_random_engine + std::normal_distribution<double>, the library
  code the reference's IMU noise comes from, compiled with g++.
* lpf_kat.json        <- oracle/_ref/lpf_probe: the REFERENCE's own
  LowPassFilterSecondOrder.hpp (stand-alone header) with the onboard logic's
  gyro / accelerometer settings.
* telemetry_kat.json  <- oracle/_ref/telemetry_probe: the REFERENCE's own
  TelemetryPacket.hpp (stand-alone header): packets -> 30-byte wire form -> back.
* planner_math_kat.json <- oracle/_ref/traj_probe: the REFERENCE's own RootFinder.hpp
  and SingleAxisTrajectory.{hpp,cpp} (stand-alone sources) + libstdc++ mt19937.
* uwb_kat.json        <- oracle/_ref/uwb_probe: libstdc++'s std::mt19937 +
  uniform_real_distribution + normal_distribution in the statement sequence of
  the reference's UWBNetwork::Run completion branch (UWBNetwork.cpp:4-6,19,66-71).
* oracle_regression.npz <- the oracle itself (NOT the reference): seeded
  single-step / rollout vectors that freeze the restatement so later edits of
  oracle/agrifly_oracle.c cannot drift silently.  It pins nothing against the
  reference (rigid-body parity is unpinned).
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TIMER_CASES = [
    # (loop dt [s] as the loop writes it, onboard logic period [s], runs)
    ("1.0/500.0", "1.0/500.0", 40),   # the reference's own setting (main.cpp:140,177)
    ("1.0/1000.0", "1.0/500.0", 40),  # BASELINE.json dt = 1 ms
    ("1.0/1000.0", "1.0/1000.0", 40),
    ("1.0/2000.0", "1.0/500.0", 40),
    ("1.0/1000.0", "0.0029", 60),     # period*1e6 truncates to 2899 us
    ("1.0/250.0", "1.0/500.0", 20),   # dt > period: one tick per run, growing lag
    ("1.0/300.0", "1.0/500.0", 30),   # uint64_t(dt*1e6) truncates to 3333 us
    ("1.0/1000.0", "1.0/30.0", 120),  # the image-request cadence (main.cpp:198-200)
]


def main():
    from oracle import oracle_py
    oracle_py.build(force=True)
    ref = os.path.join(ROOT, "oracle", "_ref")
    timer = os.path.join(ref, "timer_probe")
    rng = os.path.join(ref, "rng_probe")

    cases = []
    for dt_s, per_s, n in TIMER_CASES:
        dt, per = eval(dt_s), eval(per_s)
        out = subprocess.check_output([timer, repr(dt), repr(per), str(n)])
        rec = json.loads(out)
        rec["loop_dt_expr"], rec["period_expr"] = dt_s, per_s
        cases.append(rec)
    with open(os.path.join(HERE, "timer_cadence.json"), "w") as f:
        json.dump({"generator": "oracle/_ref/timer_probe (reference Timer.hpp, ManualTimer.hpp, "
                                "CommunicationsDelay.hpp compiled in place)",
                   "cases": cases}, f, indent=0)

    kats = []
    for args in (["600"], ["60", "2"], ["60", "12345"], ["60", "2147483646"], ["60", "4097"]):
        rec = json.loads(subprocess.check_output([rng] + args))
        rec["seed"] = int(args[1]) if len(args) > 1 else 1
        kats.append(rec)
    with open(os.path.join(HERE, "rng_kat.json"), "w") as f:
        json.dump({"generator": "oracle/_ref/rng_probe (libstdc++ <random>, g++)",
                   "gxx": subprocess.check_output(["g++", "--version"]).decode().splitlines()[0],
                   "streams": kats}, f, indent=0)

    lpf = os.path.join(ref, "lpf_probe")
    lpfs = []
    for per, cut, n in ((1 / 500.0, 200.0, 200), (1 / 500.0, 100.0, 200), (1 / 1000.0, 200.0, 200)):
        lpfs.append(json.loads(subprocess.check_output([lpf, repr(per), repr(cut), str(n)])))
    with open(os.path.join(HERE, "lpf_kat.json"), "w") as f:
        json.dump({"generator": "oracle/_ref/lpf_probe (reference LowPassFilterSecondOrder.hpp compiled in place, "
                                "LowPassFilterSecondOrder<float,float>)", "cases": lpfs}, f, indent=0)

    tel = json.loads(subprocess.check_output([os.path.join(ref, "telemetry_probe"), "48", "20261002"]))
    tel["generator"] = ("oracle/_ref/telemetry_probe (reference TelemetryPacket.hpp compiled in place; "
                        "EncodeTelemetryPacket / DecodeTelemetryPacket)")
    with open(os.path.join(HERE, "telemetry_kat.json"), "w") as f:
        json.dump(tel, f, indent=0)

    pm = json.loads(subprocess.check_output([os.path.join(ref, "traj_probe"), "64", "20261002"]))
    pm["generator"] = ("oracle/_ref/traj_probe (reference RootFinder.hpp and SingleAxisTrajectory.{hpp,cpp} compiled in "
                       "place; libstdc++ mt19937 + uniform_real_distribution in the planner's call shape)")
    with open(os.path.join(HERE, "planner_math_kat.json"), "w") as f:
        json.dump(pm, f, indent=0)

    uwb = []
    for args in (["200", "0.05", "0.1", "3.0"], ["200", "0.0", "0.0", "0.0"], ["101", "0.25", "0.5", "10.0"]):
        uwb.append(json.loads(subprocess.check_output([os.path.join(ref, "uwb_probe")] + args)))
    with open(os.path.join(HERE, "uwb_kat.json"), "w") as f:
        json.dump({"generator": "oracle/_ref/uwb_probe (libstdc++ <random>, g++; call-site shape of UWBNetwork.cpp:66-71; "
                                "true range of transaction k = 1 + k/8 m)", "cases": uwb}, f, indent=0)

    # --- oracle regression vectors (oracle-generated; not a reference pin) ---
    from tests.scenarios import random_ensemble
    ens = random_ensemble(n=256, seed=20261002)
    b0 = ens.to_oracle_batch()
    b0.step(1e-3, 1, ticks=[1])
    b1 = ens.to_oracle_batch()
    ticks100 = np.zeros(100, np.uint8)
    ticks100[1::2] = 1
    b1.step(1e-3, 100, ticks=ticks100)
    np.savez_compressed(
        os.path.join(HERE, "oracle_regression.npz"),
        seed=20261002,
        s1_pos=b0.pos, s1_vel=b0.vel, s1_att=b0.att, s1_ang_vel=b0.ang_vel,
        s1_motor=b0.motor_speed, s1_gyro=b0.gyro, s1_acc=b0.acc, s1_rng=b0.rng,
        s100_pos=b1.pos, s100_vel=b1.vel, s100_att=b1.att, s100_ang_vel=b1.ang_vel,
        s100_motor=b1.motor_speed, s100_gyro=b1.gyro, s100_acc=b1.acc, s100_rng=b1.rng)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
