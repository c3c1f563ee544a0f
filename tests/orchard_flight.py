"""System-level driver (test infrastructure): the RAPPIDS flight loop of
Simulator/Rappids_Simulator/main.cpp for a whole ensemble, with the AirSim/Unity side
replaced by the engine's own depth camera:

    every 1 ms        engine step (physics + on-device rates logic)         main.cpp:391-392
    every 10 ms       offboard loop: [every 3rd tick: render -> plan]       main.cpp:471-556
                      trajectory tracking (RunTracking) -> radio -> 30 ms   main.cpp:558-671,737-739

Everything on the vehicle side of the radio runs on the GPU (step, IMU, rates logic, depth
camera, planner; the depth images never leave HBM); the offboard tracking controller is a numpy
restatement like tests/offboard_stub.py (the caller side is out of scope, SURVEY.md section 2 row
10) fed with the true state instead of the mocap estimate.  Used by
tests/test_gpu_orchard_flight.py to check that the pieces compose: vehicles make progress through
a procedural orchard without touching a tree, and the run is reproducible.
"""
import numpy as np

from tests.offboard_stub import (F, _from_rotvec, _quat_inv, _quat_mul, _rotate, _to_rotvec, radio_quantise)


def quat_mul64(a, b):
    """Rotation.hpp:124-131 (this = a, r1 = b), arrays [4, n]"""
    return np.stack([b[0] * a[0] - b[1] * a[1] - b[2] * a[2] - b[3] * a[3],
                     b[1] * a[0] + b[0] * a[1] + b[3] * a[2] - b[2] * a[3],
                     b[2] * a[0] - b[3] * a[1] + b[0] * a[2] + b[1] * a[3],
                     b[3] * a[0] + b[2] * a[1] - b[1] * a[2] + b[0] * a[3]])


def rot_matrix64(q):
    """Rotation.hpp:196-220 -> [3, 3, n]"""
    r0, r1, r2, r3 = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    return np.array([[r0 + r1 - r2 - r3, 2 * q[1] * q[2] - 2 * q[0] * q[3], 2 * q[1] * q[3] + 2 * q[0] * q[2]],
                     [2 * q[1] * q[2] + 2 * q[0] * q[3], r0 - r1 + r2 - r3, 2 * q[2] * q[3] - 2 * q[0] * q[1]],
                     [2 * q[1] * q[3] - 2 * q[0] * q[2], 2 * q[2] * q[3] + 2 * q[0] * q[1], r0 - r1 - r2 + r3]])


def rotate64(R, v, inverse=False):
    return np.einsum("jin,jn->in", R, v) if inverse else np.einsum("ijn,jn->in", R, v)


def poly_eval(coeffs, t):
    """CommonMath::Trajectory (t^5 .. t^0), coeffs [n, 6, 3], t [n] -> pos, vel, acc [3, n]"""
    c = coeffs
    t = t[:, None]
    pos = ((((c[:, 0] * t + c[:, 1]) * t + c[:, 2]) * t + c[:, 3]) * t + c[:, 4]) * t + c[:, 5]
    vel = (((5 * c[:, 0] * t + 4 * c[:, 1]) * t + 3 * c[:, 2]) * t + 2 * c[:, 3]) * t + c[:, 4]
    acc = ((20 * c[:, 0] * t + 12 * c[:, 1]) * t + 6 * c[:, 2]) * t + 2 * c[:, 3]
    return pos.T, vel.T, acc.T


def unit(v):
    n = np.sqrt((v * v).sum(0))
    return v / np.where(n == 0, 1.0, n)


def traj_omega(coeffs, grav, t, step=0.02):
    """RapidTrajectoryGenerator::GetOmega, RapidTrajectoryGenerator.cpp:264-286"""
    n0 = unit(poly_eval(coeffs, t)[2] - grav)
    n1 = unit(poly_eval(coeffs, t + step)[2] - grav)
    cr = np.cross(n0.T, n1.T).T
    nrm = np.sqrt((cr * cr).sum(0))
    ang = np.arccos(np.clip((n0 * n1).sum(0), -1, 1)) / step
    return np.where(nrm <= 1e-6, 0.0, ang * cr / np.where(nrm <= 1e-6, 1.0, nrm))


def run_tracking(pos, vel, att, ref_pos, ref_vel, ref_acc, ref_thrust, ref_ang_vel, nat_freq=2.0, damping=0.7,
                 tc_xy=0.08, tc_z=0.4):
    """QuadcopterController::RunTracking, QuadcopterController.cpp:76-132 (float), desired yaw 0;
    tuning of the MINIQUAD (QuadcopterConstants.hpp:214-224, main.cpp:225-229)"""
    n = pos.shape[1]
    p, v, q = np.asarray(pos, F), np.asarray(vel, F), np.asarray(att, F)
    w, z = F(nat_freq), F(damping)
    acc_err = ((np.asarray(ref_pos, F) - p) * w * w + (np.asarray(ref_vel, F) - v) * F(2) * w * z).astype(F)
    e3 = np.zeros((3, n), F)
    e3[2] = 1
    thrust = (np.asarray(ref_thrust, F) + (acc_err * _rotate(q, e3)).sum(0)).astype(F)
    proper = (np.asarray(ref_acc, F) + acc_err + np.array([[0], [0], [9.81]], F)).astype(F)
    tdir = (proper / np.sqrt((proper * proper).sum(0))).astype(F)
    cosang = tdir[2]
    angle = np.where(cosang >= F(1 - 1e-12), F(0),
                     np.where(cosang <= F(-(1 - 1e-12)), F(np.pi), np.arccos(np.clip(cosang, -1, 1)))).astype(F)
    rot_ax = np.stack([-tdir[1], tdir[0], np.zeros(n, F)]).astype(F)
    nrm = np.sqrt((rot_ax * rot_ax).sum(0)).astype(F)
    tiny = nrm < F(1e-6)
    ref_att = _from_rotvec((rot_ax * (angle / np.where(tiny, F(1), nrm))).astype(F))
    ref_att[:, tiny] = np.array([[1], [0], [0], [0]], F)
    err = _quat_mul(_quat_inv(ref_att), q)                     # GetDesiredAngularVelocity, as in offboard_stub
    des_rot = _to_rotvec(err)
    z_in_err = _rotate(_quat_inv(err), e3)
    red_ax = np.stack([z_in_err[1], -z_in_err[0], np.zeros(n, F)]).astype(F)
    cos_red = z_in_err[2]
    red_an = np.where(cos_red >= F(1), F(0), np.where(cos_red <= F(-1), F(np.pi),
                                                        np.arccos(np.clip(cos_red, -1, 1)))).astype(F)
    nn = np.sqrt((red_ax * red_ax).sum(0)).astype(F)
    red_ax = np.where(nn < F(1e-12), F(0), red_ax / np.where(nn < F(1e-12), F(1), nn)).astype(F)
    k3, k12 = F(1) / F(tc_z), F(1) / F(tc_xy)
    ang_vel_err = (-k3 * des_rot - (k12 - k3) * red_an * red_ax).astype(F)
    return thrust, (np.asarray(ref_ang_vel, F) + ang_vel_err).astype(F)


def clearance(layout, pos):
    """distance-like margins of points [3, n] to the analytic trees: (horizontal distance to the
    nearest trunk surface while below its top, smallest canopy ellipsoid level (>1 = outside))"""
    dx = pos[0][None, :] - layout[:, 0][:, None]
    dy = pos[1][None, :] - layout[:, 1][:, None]
    horiz = np.sqrt(dx * dx + dy * dy) - layout[:, 2][:, None]
    below_top = pos[2][None, :] <= layout[:, 3][:, None] + 0.05
    trunk = np.where(below_top, horiz, np.inf).min(0)
    e = (((pos[:, None, :] - layout[:, 4:7].T[:, :, None]) / layout[:, 7:10].T[:, :, None]) ** 2).sum(0)
    return trunk, np.sqrt(e.min(0))


def fly_orchard(afa, n=48, seconds=6.0, seed=0, n_candidates=192, rows=6, cols=10, altitude=1.2, plan_every=3,
                log_every=2, start_planning=0.5, on_plan=None, on_tick=None, want_flags=False):
    """on_plan(frame): called after every render -> plan with a dict of everything that went in and came
    out (engine, scene, camera, device image buffer, planner inputs, outputs, flags) -- the hook the
    full-size in-loop test checks its subsample through; on_tick(tick, t, engine, state) after every
    10 ms of physics."""
    sc = afa.scenarios
    tris, layout = sc.orchard_mesh(rows=rows, cols=cols, seed=seed, return_layout=True)
    scene = afa.Scene(tris)
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    params = afa.params_from_type(5)
    rng = np.random.default_rng(seed)
    # start west of the orchard, facing +x: half of the vehicles on aisle centres, half in line with a tree row
    lane = rng.integers(0, rows - 1, n)
    on_row = rng.random(n) < 0.5
    y0 = np.where(on_row, lane * 4.0 + rng.uniform(-0.3, 0.3, n), lane * 4.0 + 2.0 + rng.uniform(-0.8, 0.8, n))
    pos0 = np.stack([np.full(n, -4.0) + rng.uniform(-1, 0, n), y0, np.full(n, altitude)])
    goal = np.stack([np.full(n, (cols - 1) * 3.0 + 8.0), y0, np.full(n, altitude)])
    att0 = np.tile(np.array([[1.0], [0.0], [0.0], [0.0]]), (1, n))
    e = afa.Ensemble(n, precision=afa.AFE_F32)
    e.set_type_table([params])
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    w_h = sc.hover_speed(params)
    e.set_state(pos0, np.zeros((3, n)), att0, np.zeros((3, n)), np.full((4, n), w_h))
    e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    buf = afa.DeviceBuffer(n * 240 * 320 * 2)
    cfg = afa.planner_default_config(320, 240, cam.depth_scale, cam.focal_length, 2 * params.arm_length,
                                     3 * params.arm_length, 0.5)            # main.cpp:167-169
    cfg.cost_type = 1
    samples = afa.planner_samples(0, 320, 240, n_candidates)                 # the reference re-seeds with 0 per plan

    dt_us, period_off, delay_us = 1000, 0.01, 30000
    planned = np.zeros(n, bool)
    coeffs = np.zeros((n, 6, 3))
    tf = np.zeros(n)
    t_plan = np.zeros(n)
    traj_R = np.tile(np.eye(3)[:, :, None], (1, 1, n))
    traj_off = pos0.copy()
    traj_grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n))
    prev_thrust = np.full(n, 9.81)
    queue = []
    log = {"t": [], "pos": [], "vel": [], "trunk": [], "canopy": [], "found": [], "plan_ms": [], "render_ms": []}
    n_ticks = int(round(seconds / period_off))
    for tick in range(1, n_ticks + 1):
        e.step(dt_us, 10)                                                   # 10 ms of physics, ticks inside the launch
        now = tick * 10 * dt_us
        t = now * 1e-6
        st = e.get_state()
        pos, vel, att = st["pos"], st["vel"], st["att"]
        if on_tick is not None:
            on_tick(tick, t, e, st)
        R_att = rot_matrix64(att)
        if t >= start_planning and tick % plan_every == 0:
            cam_q = quat_mul64(att, np.tile(mount[:, None], (1, n)))
            R_cam = rot_matrix64(cam_q)
            e3 = np.zeros((3, n))
            e3[2] = 1
            vel_c = rotate64(R_cam, vel, inverse=True)
            acc_c = rotate64(R_cam, rotate64(R_att, e3) * prev_thrust - np.array([[0], [0], [9.81]]), inverse=True)
            grav_c = rotate64(R_cam, np.tile(np.array([[0.0], [0.0], [-9.81]]), (1, n)), inverse=True)
            goal_c = rotate64(R_cam, goal - pos, inverse=True)
            ms_r = scene.render_engine(e, cam, mount, out=buf)
            out, flags, ms_p = afa.rappids_plan(cfg, buf, vel_c, acc_c, grav_c, samples, cost_vec=goal_c,
                                                want_flags=want_flags)
            plans = afa.plans_as_array(out)
            found = plans["found"].astype(bool)
            coeffs[found] = plans["coeffs"][found]
            tf[found] = plans["tf"][found]
            if on_plan is not None:
                on_plan(dict(tick=tick, t=t, engine=e, scene=scene, tris=tris, cam=cam, mount=mount, buf=buf, cfg=cfg,
                             samples=samples, state=st, vel_c=vel_c, acc_c=acc_c, grav_c=grav_c, goal_c=goal_c,
                             plans=plans, flags=flags, render_ms=ms_r, plan_ms=ms_p))
            t_plan[found] = t
            traj_R[:, :, found] = R_cam[:, :, found]
            traj_off[:, found] = pos[:, found]
            traj_grav[:, found] = grav_c[:, found]
            planned |= found
            log["found"].append(found.mean())
            log["plan_ms"].append(ms_p)
            log["render_ms"].append(ms_r)
        # tracking reference (main.cpp:558-608)
        tt = t - t_plan
        running = planned & (tt < tf)
        te = np.where(running, tt + 0.04, tf)
        p_c, v_c, a_c = poly_eval(coeffs, te)
        v_c[:, ~running] = 0
        a_c[:, ~running] = 0
        behind = p_c[2] < 0
        p_c[2, behind] = 0
        v_c[2, behind & (v_c[2] < 0)] = 0
        a_c[2, behind & (a_c[2] < 0)] = 0
        ref_pos = np.where(planned, rotate64(traj_R, p_c) + traj_off, pos0)
        ref_vel = np.where(planned, rotate64(traj_R, v_c), 0.0)
        ref_acc = np.where(planned, rotate64(traj_R, a_c), 0.0)
        acc_for_thrust = poly_eval(coeffs, te)[2]
        ref_thrust = np.where(planned, np.sqrt(((acc_for_thrust - traj_grav) ** 2).sum(0)), 9.81)
        om_c = traj_omega(coeffs, traj_grav, te)
        ref_w = np.where(planned, rotate64(R_att, rotate64(traj_R, om_c), inverse=True), 0.0)
        thrust, ang_vel = run_tracking(pos, vel, att, ref_pos, ref_vel, ref_acc, ref_thrust, ref_w)
        prev_thrust = thrust.astype(np.float64)
        queue.append((now + delay_us, radio_quantise(thrust, 35), radio_quantise(ang_vel, 35)))
        while queue and now >= queue[0][0]:
            _, th_q, w_q = queue.pop(0)
            e.set_rates_commands(th_q, w_q)
        if tick % log_every == 0:
            trunk, canopy = clearance(layout, pos)
            log["t"].append(t)
            log["pos"].append(pos.copy())
            log["vel"].append(vel.copy())
            log["trunk"].append(trunk)
            log["canopy"].append(canopy)
    buf.close()
    e.close()
    for k in ("t", "pos", "vel", "trunk", "canopy", "found", "plan_ms", "render_ms"):
        log[k] = np.array(log[k])
    log["pos0"], log["goal"], log["layout"], log["planned"] = pos0, goal, layout, planned
    return log
