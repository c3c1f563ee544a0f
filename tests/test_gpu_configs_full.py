"""BASELINE.json configs 3 and 5 AT THEIR SIZE on one MI355X, through the C ABI.

config 5: 262 144 vehicles + depth raycast against the orchard triangle mesh -- one GPU's shard of
          the 8-GPU split (32 768 views, a 5 GB image buffer) and then the whole ensemble in chunks;
config 3: 65 536 vehicles with the RAPPIDS planner IN THE LOOP: physics + IMU + onboard rates logic
          on the device -> depth image per vehicle from engine state -> plan per vehicle on it ->
          tracking controller -> radio -> physics, several camera frames.

At these sizes the CPU checkers cannot run everything, so each test combines size-independent
properties over the whole ensemble with a subsample that must be IDENTICAL to the checker
(tests/golden-style bit parity for images and plans, 1e-5 for the physics).  Needs an MI355X."""
import numpy as np
import pytest

from tests.orchard_flight import fly_orchard
from tests.scenarios import FLOORS, MEASUREMENTS, afa, rel_err_vec

pytestmark = pytest.mark.gpu
scen = afa.scenarios


def _ocam(ora, cam):
    c = ora.render_camera(cam.width, cam.height, cam.focal_length, cam.depth_scale, cam.max_count)
    c.cx, c.cy = cam.cx, cam.cy
    return c


def _near(tris, cam, pos, slack=0.5):
    """Triangles that can influence the image taken from `pos`: a hit at camera depth z >= max_count *
    depth_scale saturates to max_count exactly like a miss, and it cannot hide anything nearer, so a
    triangle whose every point is farther than that depth times the longest pixel-ray factor changes no
    count.  The checker itself stays brute force over whatever it is given; this only spares it
    triangles that are provably irrelevant (a 32 x 32-tree orchard spans 100 m, the camera sees 10 m)."""
    reach = cam.max_count * cam.depth_scale
    corner = np.sqrt(1.0 + (max(cam.cx, cam.width - cam.cx) / cam.focal_length) ** 2 +
                     (max(cam.cy, cam.height - cam.cy) / cam.focal_length) ** 2)
    t = tris.reshape(-1, 3, 3).astype(np.float64)
    lo, hi = t.min(1), t.max(1)
    d = np.maximum(np.maximum(lo - pos, pos - hi), 0.0)          # distance from pos to the triangle's box
    return tris[np.sqrt((d * d).sum(1)) <= reach * corner + slack]


def _wrap_u16(ptr, shape):
    import torch

    class _W:
        def __init__(self):
            self.__cuda_array_interface__ = dict(shape=shape, typestr="<u2", data=(ptr, False), version=2, strides=None)
    return torch.as_tensor(_W(), device="cuda")


def test_config5_262144_vehicles_depth_raycast(ora):
    import torch
    tris = scen.orchard_mesh(rows=32, cols=32, seed=1)
    scene = afa.Scene(tris)
    info = scene.info()
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    n, shard = 262144, 32768
    rng = np.random.default_rng(5)
    lo, hi = info["bounds"][:3], info["bounds"][3:]
    pos = np.stack([rng.uniform(lo[0] + 2, hi[0] - 2, n), rng.uniform(lo[1] + 2, hi[1] - 2, n), rng.uniform(0.5, 3.0, n)])
    att = scen.random_attitudes(rng, n, max_tilt_deg=20.0)
    params = afa.params_from_type(5)
    e = afa.Ensemble(n, precision=afa.AFE_F32)
    e.set_type_table([params])
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    e.set_state(pos, rng.normal(0, 0.5, (3, n)), att, np.zeros((3, n)), np.full((4, n), scen.hover_speed(params)))
    e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    e.step(1000, 33)                                   # one camera period of flight: poses come from engine state
    st = e.get_state()
    assert np.isfinite(st["pos"]).all()
    buf = afa.DeviceBuffer(shard * 240 * 320 * 2)      # 5.03 GB: one GPU's share of the 8-GPU split
    px = 240 * 320
    sub = np.sort(rng.choice(n, 64, replace=False))    # the views checked against the CPU checker
    oc = _ocam(ora, cam)
    total_ms, checked, stats = 0.0, 0, []
    for c in range(n // shard):
        ms = scene.render_engine(e, cam, mount, first=c * shard, count=shard, out=buf)
        total_ms += ms
        img = _wrap_u16(buf.ptr.value, (shard, 240, 320))
        torch.cuda.synchronize()
        flat = img.view(shard, px).to(torch.int32)
        stats.append(torch.stack([flat.amax(1), flat.amin(1), (flat == 255).sum(1), flat.sum(1)]).cpu().numpy())
        for g in sub[(sub >= c * shard) & (sub < (c + 1) * shard)]:
            k = int(g - c * shard)
            got = buf.download(np.uint16, (240, 320), offset_bytes=k * px * 2)
            p, q = st["pos"][:, g], st["att"][:, g]
            want = ora.render_depth(oc, _near(tris, cam, p), p, q, mount)
            np.testing.assert_array_equal(got, want, err_msg="vehicle %d" % g)
            if checked < 4:                            # and the culling argument itself, on a few views
                part, _ = scene.render(cam, p[:, None], q[:, None], mount)
                np.testing.assert_array_equal(part[0], got)
            checked += 1
        if c == 0:
            first_shard_ms = ms
    assert checked == 64
    s = np.concatenate(stats, axis=1)
    assert s[0].max() <= 255                           # 8-bit DepthVis widened to uint16
    frac_sky = s[2].sum() / (n * px)
    assert 0.05 < frac_sky < 0.9
    assert (s[1] < 255).mean() > 0.97                  # nearly every vehicle sees something within 10 m
    # the image is a function of the pose alone: identical poses -> identical images, whatever the chunk
    e.set_state(pos=st["pos"][:, :shard], att=st["att"][:, :shard], first=n - shard, count=shard)
    ms = scene.render_engine(e, cam, mount, first=n - shard, count=shard, out=buf)
    img = _wrap_u16(buf.ptr.value, (shard, px)).to(torch.int32)
    torch.cuda.synchronize()
    again = torch.stack([img.amax(1), img.amin(1), (img == 255).sum(1), img.sum(1)]).cpu().numpy()
    np.testing.assert_array_equal(again, stats[0])
    rays = n * px
    MEASUREMENTS["config5_depth_raycast"] = {"vehicles": n, "triangles": int(info["n_tri"]), "bvh_nodes": int(info["n_nodes"]),
                                             "shard_views": shard, "shard_ms": first_shard_ms, "total_ms": total_ms,
                                             "rays_per_s": rays / (total_ms * 1e-3), "oracle_views_identical": checked}
    print("\nconfig 5: %d views x 320x240 over %d triangles in %.0f ms (%.3g rays/s), shard of 32768 in %.0f ms"
          % (n, info["n_tri"], total_ms, rays / (total_ms * 1e-3), first_shard_ms))
    buf.close()
    e.close()


def test_config3_65536_vehicles_planner_in_the_loop(ora):
    """65 536 vehicles fly into the orchard; from t = 0.5 s every third offboard tick renders every
    vehicle's depth image from the engine's state and plans on it (images stay in HBM).  Whole-ensemble
    properties every frame; on three frames a 12-vehicle subsample is checked against the oracle CHAIN:
    image == checker's render of the engine's pose, plan == checker's planner on that image with the same
    inputs; and the first 40 ms of closed-loop physics (rates logic on the device, per-vehicle noise
    streams seeded by global index) == the oracle's closed loop for a 64-vehicle subsample."""
    n = 65536
    rng = np.random.default_rng(8)
    sub_plan = np.sort(rng.choice(n, 12, replace=False))
    sub_phys = np.sort(rng.choice(n, 64, replace=False))
    frames, phys = [], {}

    def on_tick(tick, t, e, st):
        if tick == 4:                                  # 40 steps, 19 logic ticks; the first radio packet lands after this read-out
            phys.update({k: st[k][:, sub_phys].copy() for k in ("pos", "vel", "att", "ang_vel")})
            phys["rng"] = e.get_rng_state()[sub_phys]
            phys["cmd"] = e.get_motor_cmds()[:, sub_phys]

    def on_plan(f):
        plans, flags, st = f["plans"], f["flags"], f["state"]
        found = plans["found"] == 1
        rec = {"t": f["t"], "found": float(found.mean()), "render_ms": f["render_ms"], "plan_ms": f["plan_ms"]}
        assert np.all((plans["best_index"] >= 0) == found)
        assert np.all(np.isin(flags, (0, 1, 3, 7, 15)))
        assert np.all(flags[np.nonzero(found)[0], plans["best_index"][found]] == 15)
        assert np.isfinite(plans["coeffs"][found]).all() and (plans["tf"][found] >= 2.0).all() and (plans["tf"][found] <= 3.0).all()
        if len(frames) in (0, 3, 6):                   # the oracle chain on a subsample
            cam, cfg = f["cam"], f["cfg"]
            oc = _ocam(ora, cam)
            ocfg = ora.planner_config(cam.width, cam.height, cam.depth_scale, cam.focal_length, cfg.true_vehicle_radius,
                                      cfg.planning_vehicle_radius, cfg.min_checking_dist)
            ocfg.max_pyramids = cfg.max_pyramids
            ocfg.cost_type = 1
            for i in sub_plan:
                got = f["buf"].download(np.uint16, (240, 320), offset_bytes=int(i) * 240 * 320 * 2)
                p, q = st["pos"][:, i], st["att"][:, i]
                want = ora.render_depth(oc, _near(f["tris"], cam, p), p, q, f["mount"])
                np.testing.assert_array_equal(got, want, err_msg="frame %d vehicle %d" % (len(frames), i))
                for a in range(3):
                    ocfg.cost_vec[a] = f["goal_c"][a, i]
                res, rflags = ora.planner_run(ocfg, want, f["vel_c"][:, i], f["acc_c"][:, i], f["grav_c"][:, i], f["samples"])
                assert (plans["found"][i], plans["best_index"][i]) == (res.found, res.best_index), (len(frames), i)
                np.testing.assert_array_equal(flags[i], rflags)
                assert (plans["n_cost_checks"][i], plans["n_collision_checks"][i], plans["n_collision_free"][i]) == \
                    (res.n_cost_checks, res.n_collision_checks, res.n_collision_free)
            rec["oracle_chain_checked"] = len(sub_plan)
        frames.append(rec)

    log = fly_orchard(afa, n=n, seconds=0.77, seed=0, n_candidates=192, on_plan=on_plan, on_tick=on_tick, want_flags=True,
                      log_every=77)
    assert len(frames) == 9 and sum("oracle_chain_checked" in f for f in frames) == 3
    assert np.isfinite(log["pos"]).all()
    assert min(f["found"] for f in frames) > 0.9
    assert log["trunk"].min() > 0.116 and log["canopy"].min() > 1.0           # nobody touched a tree
    advance = log["pos"][-1, 0] - log["pos0"][0]
    assert advance.min() > 0.01                                               # every vehicle is under way east (0.27 s after the first plan)

    # closed-loop physics of the first 40 ms (until the first radio packet arrives) against the oracle's closed loop (fp32 engine, 1e-5)
    params = ora.params_from_type(5)
    b = ora.Batch(len(sub_phys), [params])
    pos0 = log["pos0"][:, sub_phys]
    b.pos[:] = pos0
    b.motor_speed[:] = scen.hover_speed(afa.params_from_type(5))
    b.rng[:] = 1 + sub_phys                           # AFE_SEED_DECORRELATED: seed = 1 + GLOBAL index
    cl = ora.ClosedLoopBatch(b, [ora.logic_params_from_type(5, 1.0 / 500.0)], 1.0 / 500.0)
    cl.set_rates_cmd(np.full(len(sub_phys), 9.81, np.float32), np.zeros((3, len(sub_phys)), np.float32))
    cl.step(1e-3, afa.plan_ticks(1.0 / 500.0, 0, 1000, 40)[0])
    np.testing.assert_array_equal(phys["rng"], b.rng)
    worst = {}
    for k, ref in dict(pos=b.pos, vel=b.vel, att=b.att, ang_vel=b.ang_vel).items():
        worst[k] = rel_err_vec(phys[k], ref, FLOORS[k] if k in ("pos", "att") else 0.1)   # hover: v, w ~ 0 by construction
        assert worst[k] <= 1e-5, (k, worst[k])
    assert rel_err_vec(phys["cmd"], b.motor_cmd, 1.0) <= 1e-5
    MEASUREMENTS["config3_planner_in_loop"] = {
        "vehicles": n, "candidates": 192, "frames": frames, "physics_rel_err_first_40_steps": worst,
        "mean_render_ms": float(np.mean([f["render_ms"] for f in frames])),
        "mean_plan_ms": float(np.mean([f["plan_ms"] for f in frames]))}
    print("\nconfig 3 in loop: %d vehicles, %d frames, render %.0f ms + plan %.0f ms per frame, found %.3f..%.3f"
          % (n, len(frames), np.mean([f["render_ms"] for f in frames]), np.mean([f["plan_ms"] for f in frames]),
             min(f["found"] for f in frames), max(f["found"] for f in frames)))


@pytest.mark.parametrize("n,addressing", [(30_000_000, "buffer resources, offsets up to 4.1e9"),
                                          (40_000_000, "global addresses: the arena exceeds 4 GiB")])
def test_tens_of_millions_of_vehicles_address_their_slabs_correctly(n, addressing):
    """Sized for 288 GB: 3e7 vehicles are a 4.1 GB arena -- buffer-resource offsets right up to the 32-bit limit --
    and 4e7 a 5.5 GB one, beyond it, which takes the global-address instantiations for real (not via
    afe_set_addressing).  Three windows of 4 096 vehicles -- the first, the middle, the very last -- get random
    states, commands and gusts, everybody else rests at the origin; after 6 noisy steps the windows must match
    the oracle (seeded per global index) and a resting vehicle next to each window must still rest."""
    from oracle import oracle_py as ora
    from tests.scenarios import FLOORS, rel_err_vec
    w = 4096
    windows = [0, (n // 2) // 64 * 64 + 17, n - w]
    p = afa.params_from_type(5)
    with afa.Ensemble(n, first_global_index=5) as e:
        e.set_type_table([p])
        e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
        e.set_logic_period(1 / 500)
        assert e.step_kernel_info() == ("kernel arguments", "buffer" if n == 30_000_000 else "global")
        data = []
        for k, first in enumerate(windows):
            d = afa.scenarios.random_ensemble(w, 700 + k, type_ids=(5,), ground_fraction=0.0)
            d.pos[2] += 30
            cmd = np.minimum(d.motor_cmd, 900.0)
            e.set_state(d.pos, d.vel, d.att, d.ang_vel, d.motor_speed, first=first, count=w)
            e.set_motor_cmds(cmd, first=first, count=w)
            e.set_external_force(d.ext_force, first=first, count=w)
            data.append((d, cmd))
        steps = 6
        e.step(1000, steps)
        ticks = afa.plan_ticks(1 / 500, 0, 1000, steps)[0]
        for (d, cmd), first in zip(data, windows):
            st = e.get_state(first=first, count=w)
            gyro, acc = e.get_imu(first=first, count=w)
            b = ora.Batch(w, [ora.params_from_type(5)])
            b.pos[:], b.vel[:], b.att[:], b.ang_vel[:], b.motor_speed[:] = d.pos, d.vel, d.att, d.ang_vel, d.motor_speed
            b.motor_cmd[:] = cmd
            b.ext_force[:] = d.ext_force
            b.rng[:] = 5 + first + 1 + np.arange(w)
            b.step(1e-3, steps, ticks=ticks)
            for f in ("pos", "vel", "att", "ang_vel"):
                assert rel_err_vec(st[f], getattr(b, f), FLOORS[f]) <= 1e-5, (first, f)
            assert rel_err_vec(gyro, b.gyro, FLOORS["gyro"]) <= 1e-5 and rel_err_vec(acc, b.acc, FLOORS["acc"]) <= 1e-5
            assert np.array_equal(e.get_rng_state(first=first, count=w), b.rng)
        # bystanders: at rest at the origin with zero commands they fall freely for 6 ms, all alike
        rest = [e.get_state(first=f, count=1) for f in (w, windows[1] - 1, n - w - 1)]
        for r in rest[1:]:
            for f in r:
                assert np.array_equal(r[f], rest[0][f]), f
        assert abs(rest[0]["vel"][2, 0] + 9.81 * 0.006) < 1e-3 or rest[0]["pos"][2, 0] == 0.0
    MEASUREMENTS["large_ensemble_%d" % n] = {"vehicles": n, "addressing": addressing}
