import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
afa = importlib.import_module("agri-fly_amd")
n = 1 << 20
p = afa.params_from_type(5)
data = afa.scenarios.gust_ensemble(n, p, seed=4)
def run(types, nt, logic=False, period=0.002):
    e = afa.Ensemble(n)
    e.set_split_stepping(1)      # kernel timing: one launch per step
    tab = [afa.params_from_type([5,1,2,4][k % 4]) for k in range(nt)]
    e.set_type_table(tab)
    if types is not None: e.set_vehicle_types(types)
    e.set_logic_period(period)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
    e.set_motor_cmds(data.motor_cmd); e.set_external_force(data.ext_force)
    if logic:
        e.set_rates_logic([afa.rates_logic_params_from_type([5,1,2,4][k % 4]) for k in range(nt)])
        e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    for _ in range(50): e.step(1000, 1)
    a, b = e.event(), e.event(); e.sync(); e.record(a)
    for _ in range(300): e.step(1000, 1)
    e.record(b); us = e.elapsed_ms(a, b) * 1e3 / 300
    bts = 0.5 * (e.algorithmic_bytes_per_step(True) + e.algorithmic_bytes_per_step(False)) if period == 0.002 else e.algorithmic_bytes_per_step(period < 0.001)
    e.close()
    return us, bts
rng = np.random.default_rng(0)
def runs(nt):     # fleets laid out type by type: one type per aligned run of 64 vehicles
    return (np.arange(n) * nt // n).astype(np.uint8)
for name, types, nt in (("uniform (kernel-arg params)", None, 1), ("4 types, random (LDS table)", rng.integers(0, 4, n).astype(np.uint8), 4),
                        ("4 types, type by type", runs(4), 4),
                        ("64 types, random", rng.integers(0, 64, n).astype(np.uint8), 64), ("64 types, type by type", runs(64), 64),
                        ("256 types, random", rng.integers(0, 256, n).astype(np.uint8), 256), ("256 types, type by type", runs(256), 256)):
    for logic in (False, True):
        us, bts = run(types, nt, logic)
        print("%-28s logic=%d  %.2f us/step  %.0f B/veh  %.0f GB/s" % (name, logic, us, bts, n * bts / us / 1e3))
