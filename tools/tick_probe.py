#!/usr/bin/env python3
"""Runs the step kernel at a fixed logic cadence for profiling with rocprofv3:
    rocprofv3 --pmc ... -- python3 tools/tick_probe.py on|off|mix [vehicles] [launches]
(on: the logic gate fires every step; off: never; mix: every 2nd step; logic: mix with the on-device rates logic
closing the loop on a hover command; split: mix with afe_set_split_stepping(2))."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
afa = importlib.import_module("agri-fly_amd")

mode = sys.argv[1] if len(sys.argv) > 1 else "mix"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 60
period = {"on": 0.0005, "off": 1000.0, "mix": 0.002, "logic": 0.002, "split": 0.002}[mode]
p = afa.params_from_type(5)
data = afa.scenarios.gust_ensemble(n, p, seed=4)
e = afa.Ensemble(n)
e.set_type_table([p])
e.set_logic_period(period)
e.set_split_stepping(1)     # one launch per step unless the mode asks otherwise
e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
e.set_motor_cmds(data.motor_cmd)
e.set_external_force(data.ext_force)
if mode == "logic":
    import numpy as np
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
if mode == "split":
    e.set_split_stepping(2)
for _ in range(launches):
    e.step(1000, 1)
e.sync()
e.close()
