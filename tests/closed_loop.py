"""Closed-loop drivers shared by the config-0/1 tests: the Rappids_Simulator
loop order (Run(); advance clock; offboard gate; radio delivery -- main.cpp:
330,391-392,471-476,737-739) around either the oracle or the HIP engine."""
import numpy as np

from tests.offboard_stub import OffboardHover


def fly_oracle(ora, n, seconds, dt_us=1000, period=1 / 500, seeds=None, log_every=10):
    b = ora.Batch(n, [ora.params_from_type(5)])          # at rest on the ground (main.cpp:279-280)
    b.rng[:] = 1 if seeds is None else seeds
    cl = ora.ClosedLoopBatch(b, [ora.logic_params_from_type(5, period)], period)
    off = OffboardHover(n)
    n_runs = int(round(seconds * 1e6 / dt_us))
    dts, ticks = ora.clock_ticks(dt_us * 1e-6, period, n_runs)
    log = []
    now = 0
    for it in range(n_runs):
        if dts[it] > 0:
            cl.step(dts[it], [ticks[it]])
        now += dt_us
        msg = off.maybe_command(now, b.pos, b.vel, b.att)
        if msg is not None:
            cl.set_rates_cmd(msg[0], msg[1])
        if it % log_every == 0:
            log.append(np.concatenate([b.pos[:, :].copy(), b.vel.copy(), b.att.copy(), b.ang_vel.copy()]))
    return b, np.array(log)


def fly_engine(afa, n, seconds, precision, dt_us=1000, period=1 / 500, decorrelated=False, log_every=10):
    off = OffboardHover(n)
    n_runs = int(round(seconds * 1e6 / dt_us))
    e = afa.Ensemble(n, precision=precision)
    e.set_type_table([afa.params_from_type(5)])
    e.set_logic_period(period)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED if decorrelated else afa.AFE_SEED_REFERENCE)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    log = []
    now = 0
    pending = 0
    for it in range(n_runs):
        if it > 0:
            pending += 1                     # Run() #0 is the dt == 0 early return
        now += dt_us
        gate = (now - off.reset_us) * 1e-6 > off.period
        due = bool(off.queue) and now >= off.queue[0][0]
        want_log = it % log_every == 0
        if gate or due or want_log:
            if pending:
                e.step(dt_us, pending)       # fused launch up to this event
                pending = 0
            st = e.get_state() if (gate or want_log) else None
            msg = off.maybe_command(now, st["pos"], st["vel"], st["att"]) if gate else (
                off.maybe_command(now, None, None, None) if due else None)
            if msg is not None:
                e.set_rates_commands(msg[0], msg[1])
            if want_log:
                log.append(np.concatenate([st["pos"], st["vel"], st["att"], st["ang_vel"]]))
    if pending:
        e.step(dt_us, pending)
    return e, np.array(log)
