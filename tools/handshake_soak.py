"""Soak of the resident grid's hand-shake (development): a host doing random things -- bursts of steps, pauses around the
grid's patience, getters, setters, configuration and mode changes, checkpoints -- on a launched engine with a device arena
and, side by side, on an engine in a random stepping mode with a random kind of arena; whenever the state is read it must
be the same bits.  The GPU tests run three seeds of this (tests/test_gpu_persistent.py); this runs as many as asked for.
   python tools/handshake_soak.py [first_seed] [n_seeds]"""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
from tests.scenarios import random_ensemble


def make(n, host_visible, mode, logic, seed):
    ens = random_ensemble(n, seed=seed, with_wrench=True, type_ids=(5,))
    d = ens.data
    e = afa.Ensemble(n, precision=afa.AFE_F32 if seed % 3 else afa.AFE_F64, host_visible=host_visible)
    e.set_type_table([afa.params_from_type(d.type_ids[0])])
    e.set_logic_period(1 / 500)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_state(d.pos, d.vel, d.att, d.ang_vel, d.motor_speed)
    e.set_motor_cmds(d.motor_cmd)
    e.set_external_force(d.ext_force)
    if logic:
        e.set_rates_logic([afa.rates_logic_params_from_type(d.type_ids[0])])
        rng = np.random.default_rng(3)
        e.set_rates_commands(np.full(n, 9.5, np.float32), (0.2 * rng.standard_normal((3, n))).astype(np.float32))
    e.set_split_stepping(1)
    e.set_step_mode(mode)
    return e, d


def everything(e):
    st = e.get_state()
    gyro, acc = e.get_imu()
    return dict(st, gyro=gyro, acc=acc, rng=e.get_rng_state(), cmd=e.get_motor_cmds(), time=np.array([e.time_us, e.logic_ticks]))


def same(a, b, what):
    xa, xb = everything(a), everything(b)
    for k in xa:
        assert np.array_equal(xa[k], xb[k], equal_nan=True), (what, k)


def one(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 700, 3000, 20011, 140000]))
    logic = bool(rng.integers(0, 2))
    hv = bool(rng.integers(0, 2)) and n <= 20011
    if os.environ.get("SOAK_ONLY_N") and n != int(os.environ["SOAK_ONLY_N"]):
        return n, logic, hv
    modes = [afa.AFE_STEP_PERSISTENT, afa.AFE_STEP_RESIDENT, afa.AFE_STEP_AUTO, afa.AFE_STEP_LAUNCH]
    a, d = make(n, False, afa.AFE_STEP_LAUNCH, logic, seed)
    b, _ = make(n, hv, int(rng.choice(modes[:3])), logic, seed)
    issued, blob, blob_a = 0, None, None
    for op in range(200):
        what = rng.choice(["step", "step", "step", "burst", "pause", "get", "get1", "cmd", "noise", "gust", "mode", "completed", "save", "load", "sync", "sync", "devsync"])
        if os.environ.get("SOAK_TRACE"): print("  seed %d op %d: %s" % (seed, op, what), file=sys.stderr, flush=True)
        if what == "step":
            dt, k = int(rng.choice([1000, 1000, 500, 2000])), int(rng.integers(1, 40))
            a.step(dt, k); b.step(dt, k); issued += k
        elif what == "burst":
            k = int(rng.integers(1, 300))
            for _ in range(k):
                b.step(1000, 1)
            a.step(1000, k); issued += k
        elif what == "pause":
            time.sleep(float(rng.choice([0.00005, 0.0002, 0.0004, 0.002])))
        elif what == "get":
            same(a, b, (seed, op))
        elif what == "get1":
            i = int(rng.integers(0, n))
            sa, sb = a.get_state(first=i, count=1), b.get_state(first=i, count=1)
            assert all(np.array_equal(sa[k], sb[k], equal_nan=True) for k in sa), (seed, op, "get1")
        elif what == "cmd":
            for e in (a, b):
                if logic:
                    e.set_rates_commands(np.full(n, 9.0 + op * 0.01, np.float32), np.zeros((3, n), np.float32))
                else:
                    e.set_motor_cmds(np.clip(d.motor_cmd * (1 + 0.001 * op), 0, None))
        elif what == "noise":
            pol = int(rng.choice([afa.AFE_SEED_DECORRELATED, afa.AFE_SEED_COUNTER]))
            on = bool(rng.integers(0, 4))
            for e in (a, b):
                e.set_imu_noise(on, 0.1, 0.2, pol)
        elif what == "gust":
            on, per = bool(rng.integers(0, 2)), int(rng.choice([7000, 30000, 100000]))
            for e in (a, b):
                e.set_gust_process(on, seed=5, sigma_max=0.3, period_us=per)
        elif what == "mode":
            m = int(rng.choice(modes))
            if os.environ.get("SOAK_VERBOSE"): print("  seed %d op %d: mode %d" % (seed, op, m), file=sys.stderr, flush=True)
            b.set_step_mode(m)
        elif what == "sync":             # round 4: waits for the steps, a resident grid stays; the completion word is true afterwards
            b.sync(); a.sync()
            assert b.steps_completed == issued == a.steps_completed, (seed, op, "sync")
        elif what == "devsync":          # a device-wide synchronise issued outside the engine: neither waits for the grid nor ends it
            torch.cuda.synchronize()
        elif what == "completed":
            assert b.steps_completed <= issued == a.steps_completed
        elif what == "save":
            blob, blob_a = b.save_checkpoint(), a.save_checkpoint()
        elif what == "load" and blob is not None:
            b.load_checkpoint(blob); a.load_checkpoint(blob_a)
    same(a, b, (seed, "end"))
    a.close(); b.close()
    return n, logic, hv


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = time.time()
for seed in range(first, first + count):
    t1 = time.time()
    n, logic, hv = one(seed)
    print("seed %d: %d vehicles, %s, logic %s, host-visible %s: ok (%.1f s)" % (seed, n, "fp32" if seed % 3 else "fp64", logic, hv, time.time() - t1), file=sys.stderr, flush=True)
print("%d seeds in %.0f s" % (count, time.time() - t0))
