"""One-off soak (development): 2^20 vehicles hover closed-loop on the device (rates logic, IMU noise, gusts) for 100 s of
simulated time = 1e5 steps; every vehicle must stay finite, upright and near its set-point-free hover attitude."""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
n = 1 << 20
p = afa.params_from_type(5)
data = afa.scenarios.gust_ensemble(n, p, seed=11)
e = afa.Ensemble(n)
e.set_type_table([p])
e.set_logic_period(1 / 500)
e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
e.set_external_force(data.ext_force * 0.2)
e.set_rates_logic([afa.rates_logic_params_from_type(5)])
e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
t0 = time.time()
for chunk in range(100):
    e.step(1000, 1000)
    if chunk % 20 == 19:
        st = e.get_state()
        q = st["att"]
        print("t=%3d s: finite %s, |q|-1 max %.2e, tilt max %.3f rad, |w| max %.3f rad/s, speed max %.2f m/s"
              % (chunk + 1, bool(all(np.isfinite(st[k]).all() for k in st)), np.abs(np.linalg.norm(q, axis=0) - 1).max(),
                 (2 * np.arccos(np.clip(np.abs(q[0]), 0, 1))).max(), np.linalg.norm(st["ang_vel"], axis=0).max(),
                 np.linalg.norm(st["vel"], axis=0).max()), flush=True)
e.sync()
print("1e5 steps of 2^20 vehicles in %.1f s wall (%.3g vehicle-steps/s incl. read-backs)" % (time.time() - t0, n * 1e5 / (time.time() - t0)))
