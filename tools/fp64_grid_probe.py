"""fp64 resident grids at sizes around the residency of their kernels: does a grid of n / 64 workers start without the
pump having to park it and the host to shrink it (stderr says so)?   python tools/fp64_grid_probe.py"""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
from tests.scenarios import random_ensemble
import itertools
for logic, (mode, name), noise, force, n in itertools.product((False, True), ((afa.AFE_STEP_PERSISTENT, "persistent"), (afa.AFE_STEP_RESIDENT, "resident state")),
                                                       (None, afa.AFE_SEED_DECORRELATED, afa.AFE_SEED_COUNTER), (False, True), (140000, 200000)):
    if True:
        if True:
            ens = random_ensemble(n, seed=1, with_wrench=True, type_ids=(5,))
            d = ens.data
            e = afa.Ensemble(n, precision=afa.AFE_F64)
            e.set_type_table([afa.params_from_type(5)])
            e.set_logic_period(1 / 500)
            e.set_imu_noise(noise is not None, 0.1, 0.2, afa.AFE_SEED_DECORRELATED if noise is None else noise)
            e.set_state(d.pos, d.vel, d.att, d.ang_vel, d.motor_speed)
            e.set_motor_cmds(d.motor_cmd)
            if force: e.set_external_force(d.ext_force)
            if logic:
                e.set_rates_logic([afa.rates_logic_params_from_type(5)])
                e.set_rates_commands(np.full(n, 9.5, np.float32), np.zeros((3, n), np.float32))
            e.set_step_mode(mode)
            t0 = time.perf_counter()
            for _ in range(200):
                e.step(1000, 1)
            e.sync()
            print("fp64, logic %s, %s, noise %s, force %s, %d vehicles (%d chunks): 200 steps in %.1f ms" % (logic, name, noise, force, n, (n + 63) // 64, (time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
            e.close()
