#!/usr/bin/env python3
"""bench.py -- vehicle-steps/sec of the HIP vehicle-step path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole ensemble: one launch of the
step kernel advancing every vehicle by dt = 1 ms, state read from and written
back to HBM (no temporal fusion in the headline number).  Workload: the
config-4 shape of BASELINE.json -- a hovering MINIQUAD ensemble with a
per-vehicle wind-gust force through the SetExternalForce port, IMU synthesis
with on-device libstdc++-compatible noise at the 500 Hz onboard-logic cadence --
1,048,576 vehicles PER GPU (weak scaling; inputs resident in HBM before the
timed region).  For N > 1 the driver launches one rank per GPU under
torch.distributed.run; ranks own contiguous shards and step them with no
collective (the path has none); only the timing uses a barrier and a MAX.

Prints ONE JSON line on rank 0 (see the repo task contract), including
  roofline     -- algorithmic HBM bytes / measured kernel time vs 8 TB/s
  cpu_baseline -- the CPU oracle (port of the reference's algorithm) timed on
                  one host core over a bounded sample (rank 0, N = 1 only)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
DT_US = 1000
LOGIC_PERIOD = 1.0 / 500.0


def build_shard(afa, n_local, first_global, n_global, device, fext=True):
    p = afa.params_from_type(5)  # QC_TYPE_CF_MINIQUAD: vehicle id 1 of every shipped main
    data = afa.scenarios.gust_ensemble(n_local, p, seed=4, first_global=first_global, n_global=n_global)
    e = afa.Ensemble(n_local, precision=afa.AFE_F32, device=device, first_global_index=first_global)
    e.set_type_table([p])
    e.set_logic_period(LOGIC_PERIOD)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
    e.set_motor_cmds(data.motor_cmd)
    if fext:
        e.set_external_force(data.ext_force)
    return e


def time_steps(e, steps, per_launch, sync, barrier):
    """wall time of `steps` physics steps issued as launches of `per_launch`"""
    barrier()
    sync()
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        k = min(per_launch, steps - done)
        e.step(DT_US, k)
        done += k
    sync()
    barrier()
    return time.perf_counter() - t0


def kernel_time_events(e, launches):
    """average duration of one step-kernel launch, HIP events on the engine's
    stream bracketing a back-to-back run of single-step launches"""
    ev0, ev1 = e.event(), e.event()
    e.sync()
    e.record(ev0)
    for _ in range(launches):
        e.step(DT_US, 1)
    e.record(ev1)
    ms = e.elapsed_ms(ev0, ev1)
    e.destroy_event(ev0)
    e.destroy_event(ev1)
    return ms * 1e-3 / launches


def mean_bytes_per_step(e, afa, steps):
    ticks, _ = afa.plan_ticks(LOGIC_PERIOD, 0, DT_US, max(2, min(steps, 1000)))
    frac = float(ticks.mean())
    return frac * e.algorithmic_bytes_per_step(True) + (1 - frac) * e.algorithmic_bytes_per_step(False), frac


def per_kernel_breakdown(afa, n_local, device):
    """HIP-event launch time of the two instantiations the timed region
    alternates between: gate never firing (NOISE=0) / firing every step (NOISE=1)"""
    res = {}
    for name, period in (("off_tick", 1000.0), ("on_tick", 0.0005)):
        e = build_shard(afa, n_local, 0, n_local, device)
        e.set_logic_period(period)
        for _ in range(50):
            e.step(DT_US, 1)
        t = kernel_time_events(e, 300)
        b = e.algorithmic_bytes_per_step(name == "on_tick")
        res[name] = {"kernel_us": t * 1e6, "bytes_per_vehicle_step": b, "achieved_GBs": n_local * b / t / 1e9}
        e.close()
    return res


def committed_traffic(n_local):
    """PMC-derived HBM bytes per launch of this exact workload, from the rocprofv3
    summary committed under profiles/ (counters cannot be read inside the run)"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        w = t["workload"]
        if w["vehicles_per_gpu"] == n_local and w["dt_us"] == DT_US and w["fext"] and w["noise"]:
            return t["traffic_bytes_per_launch"], t["source"]
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def perception_rows(afa, n_views=512, n_planners=16384, n_candidates=256):
    """SURVEY 8f rows f4 + f3 on the config-5 / config-3 shapes, bounded to a fraction of a second:
    depth camera over a procedural orchard, then the RAPPIDS planner on those images (kept in HBM)."""
    tris = afa.scenarios.orchard_mesh(rows=32, cols=32, seed=1)
    scene = afa.Scene(tris)
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    rng = np.random.default_rng(4)
    pos = np.stack([rng.uniform(-5, 90, n_views), rng.uniform(-5, 120, n_views), rng.uniform(0.8, 2.5, n_views)])
    yaw = rng.uniform(-np.pi, np.pi, n_views)
    att = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)])
    e = afa.Ensemble(n_views, precision=afa.AFE_F32)
    e.set_type_table([afa.params_from_type(5)])
    e.set_state(pos, np.zeros((3, n_views)), att, np.zeros((3, n_views)), np.zeros((4, n_views)))
    buf = afa.DeviceBuffer(n_views * 240 * 320 * 2)
    scene.render_engine(e, cam, mount, out=buf)
    ms_render = min(scene.render_engine(e, cam, mount, out=buf) for _ in range(3))
    cfg = afa.planner_default_config(320, 240, cam.depth_scale, cam.focal_length, 0.116, 0.174, 0.5)
    idx = (np.arange(n_planners) % n_views).astype(np.int32)
    vel0 = np.stack([rng.normal(0, 0.3, n_planners), rng.normal(0, 0.2, n_planners), rng.uniform(0, 2.0, n_planners)])
    acc0 = rng.normal(0, 0.3, (3, n_planners))
    grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n_planners))
    samples = afa.planner_samples(0, 320, 240, n_candidates)
    ms_plan, found = 1e30, 0.0
    for _ in range(2):
        out, _, ms = afa.rappids_plan(cfg, buf, vel0, acc0, grav, samples, image_index=idx)
        ms_plan = min(ms_plan, ms)
    found = float(np.mean([o.found for o in out]))
    buf.close()
    e.close()
    # one camera frame of the closed perception loop (config 3 / 5): 33 ms of physics with the rates
    # logic on the device, one depth image per vehicle, one plan per vehicle on it -- all from engine state
    n_loop = 4096
    lane = rng.integers(0, 31, n_loop)
    p0 = np.stack([rng.uniform(-6.0, -3.0, n_loop), lane * 4.0 + 2.0 + rng.uniform(-0.8, 0.8, n_loop), np.full(n_loop, 1.2)])
    q0 = np.tile(np.array([[1.0], [0.0], [0.0], [0.0]]), (1, n_loop))
    params = afa.params_from_type(5)
    el = afa.Ensemble(n_loop, precision=afa.AFE_F32)
    el.set_type_table([params])
    el.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    el.set_rates_logic([afa.rates_logic_params_from_type(5)])
    el.set_state(p0, np.zeros((3, n_loop)), q0, np.zeros((3, n_loop)), np.full((4, n_loop), afa.scenarios.hover_speed(params)))
    el.set_rates_commands(np.full(n_loop, 9.81, np.float32), np.zeros((3, n_loop), np.float32))
    bl = afa.DeviceBuffer(n_loop * 240 * 320 * 2)
    v_c = np.stack([np.zeros(n_loop), np.zeros(n_loop), np.full(n_loop, 1.0)])
    zeros = np.zeros((3, n_loop))
    g_c = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n_loop))
    samples_l = afa.planner_samples(0, 320, 240, 192)
    frame = {}
    for _ in range(2):
        t0 = time.perf_counter()
        el.step(1000, 33)
        el.sync()
        t1 = time.perf_counter()
        ms_r = scene.render_engine(el, cam, mount, out=bl)
        outl, _, ms_p = afa.rappids_plan(cfg, bl, v_c, zeros, g_c, samples_l)
        frame = {"vehicles": n_loop, "physics_33_steps_ms": (t1 - t0) * 1e3, "render_ms": ms_r, "plan_ms": ms_p,
                 "candidates": 192, "fraction_found": float(np.mean([o.found for o in outl]))}
    frame["frame_ms"] = frame["physics_33_steps_ms"] + frame["render_ms"] + frame["plan_ms"]
    frame["realtime_factor_at_30Hz"] = 33.0 / frame["frame_ms"]
    bl.close()
    el.close()
    info = scene.info()
    return {"closed_perception_loop_frame": frame,
            "depth_camera": {"views": n_views, "image": "320x240", "triangles": int(info["n_tri"]),
                             "kernel_ms": ms_render, "rays_per_s": n_views * 76800 / (ms_render * 1e-3)},
            "rappids_planner": {"planners": n_planners, "candidates": n_candidates, "distinct_images": n_views,
                                "kernel_ms": ms_plan, "plans_per_s": n_planners / (ms_plan * 1e-3),
                                "fraction_found": found},
            "note": "images rendered from engine state and planned on without leaving HBM; results are "
                    "bit-identical to the CPU checkers in tests/test_gpu_render.py / test_gpu_planner.py"}


def cpu_baseline(afa, budget_vehicle_steps=100_000_000):
    """the oracle (double, scalar C, 1 thread) on a bounded sample of the same
    workload; test infrastructure used here only as the reported baseline"""
    from oracle import oracle_py
    n, steps = 16384, max(10, budget_vehicle_steps // 16384)
    p = afa.params_from_type(5)
    data = afa.scenarios.gust_ensemble(n, p, seed=4, n_global=1 << 20)
    b = oracle_py.Batch(n, [oracle_py.params_from_type(5)])
    b.pos[:], b.vel[:], b.att[:], b.ang_vel[:] = data.pos, data.vel, data.att, data.ang_vel
    b.motor_speed[:], b.motor_cmd[:] = data.motor_speed, data.motor_cmd
    b.ext_force[:] = data.ext_force
    b.rng[:] = 1 + np.arange(n)
    ticks, _ = afa.plan_ticks(LOGIC_PERIOD, 0, DT_US, steps)
    b.step(DT_US * 1e-6, 10, ticks=ticks[:10])  # warm
    t0 = time.perf_counter()
    b.step(DT_US * 1e-6, steps, ticks=ticks)
    dt = time.perf_counter() - t0
    out = {"value": n * steps / dt, "unit": "vehicle-steps/s", "cores": 1, "kind": "port",
           "sample": "%d vehicles x %d steps of the same workload (gust force, IMU+noise every 2nd step), "
                     "oracle/agrifly_oracle.c double precision, gcc -O2, 1 thread, %.1f s" % (n, steps, dt),
           "host_cpus": os.cpu_count()}
    # the same port spread over every host core (OpenMP over vehicles), SURVEY 8d CPU-baseline (ii)
    threads = os.cpu_count() or 1
    if threads > 1:
        oracle_py.lib().ora_set_batch_threads(threads)
        b.step(DT_US * 1e-6, 10, ticks=ticks[:10])
        steps_mt = steps * min(threads, 32) // 4
        ticks_mt, _ = afa.plan_ticks(LOGIC_PERIOD, 0, DT_US, steps_mt)
        t0 = time.perf_counter()
        b.step(DT_US * 1e-6, steps_mt, ticks=ticks_mt)
        dt_mt = time.perf_counter() - t0
        oracle_py.lib().ora_set_batch_threads(1)
        out["all_cores"] = {"value": n * steps_mt / dt_mt, "unit": "vehicle-steps/s", "cores": threads,
                            "sample": "%d vehicles x %d steps, %d OpenMP threads, %.1f s" % (n, steps_mt, threads, dt_mt)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--vehicles", type=int, default=1 << 20, help="vehicles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: RCCL even for one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is reserved
        # for the one JSON line, so the communicator is created (first barrier) with fd 1 -> fd 2
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    afa = importlib.import_module("agri-fly_amd")
    n_local = args.vehicles
    n_global = n_local * world
    e = build_shard(afa, n_local, rank * n_local, n_global, local_rank)

    # ---- the headline measurement: W warmup steps, then exactly K timed ----
    time_steps(e, args.warmup, 1, sync, barrier)
    elapsed = time_steps(e, args.steps, 1, sync, barrier)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = n_global * args.steps / elapsed

    out = None
    if rank == 0:
        bytes_step, tick_frac = mean_bytes_per_step(e, afa, args.steps)
        t_kernel = kernel_time_events(e, min(args.steps, 1000))
        achieved = n_local * bytes_step / t_kernel / 1e9
        traffic, traffic_src = committed_traffic(n_local)
        out = {
            "metric": "vehicle-steps/sec @dt=1ms",
            "value": value,
            "unit": "vehicle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "config 4 shape: hovering CF_MINIQUAD ensemble, per-vehicle wind-gust external force, "
                            "IMU synthesis + on-device minstd_rand0/normal noise at the 500 Hz logic gate, "
                            "one kernel launch per 1 ms step (no temporal fusion)",
                "vehicles_per_gpu": n_local,
                "vehicles_total": n_global,
                "dt_us": DT_US,
                "logic_period_s": LOGIC_PERIOD,
                "steps_per_launch": 1,
                "parallelism": "ensemble sharded contiguously, %d rank(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": n_local * bytes_step,
                "kernel": "afe::afe_step_kernel<float, FEXT=1, TEXT=0, NOISE, LOGIC=0, SINGLE=1> -- the timed region "
                          "alternates NOISE=0 (no logic tick) and NOISE=1 (tick: IMU + six Gaussian draws) launches",
                "kernel_us": t_kernel * 1e6,
                "algorithmic_bytes_per_vehicle_step": bytes_step,
                "imu_tick_fraction": tick_frac,
                "per_kernel": per_kernel_breakdown(afa, n_local, local_rank),
                "note": "achieved = mean algorithmic bytes per launch / mean HIP-event launch time over the "
                        "timed cadence; in-place state (%.0f MB per launch) fits the 256 MiB Infinity Cache; "
                        "traffic = bytes per launch from the committed rocprofv3 PMC summary"
                        % (n_local * bytes_step / 1e6),
            },
        }
        if world == 1 and not args.no_sweep:
            sweep = []
            for n in (1024, 4096, 65536, 262144, 4 << 20):
                es = build_shard(afa, n, 0, n, local_rank)
                k = 400 if n >= 262144 else 2000
                time_steps(es, 50, 1, sync, barrier)
                t1 = time_steps(es, k, 1, sync, barrier)
                tf = time_steps(es, k, 2, sync, barrier)
                t50 = time_steps(es, k, 50, sync, barrier)
                es.set_max_fused_steps(1)       # still one launch per step, issued by the engine's C++ loop
                tc = time_steps(es, k, k, sync, barrier)
                es.set_max_fused_steps(64)
                sweep.append({"vehicles": n, "vsteps_per_s": n * k / t1, "vsteps_per_s_native_loop": n * k / tc,
                              "vsteps_per_s_fused2": n * k / tf, "vsteps_per_s_fused50": n * k / t50})
                es.close()
            # config 2 closed on the GPU: on-device onboard rates logic, hover command
            closed = []
            for n in (4096, 1 << 20):
                es = build_shard(afa, n, 0, n, local_rank)
                es.set_rates_logic([afa.rates_logic_params_from_type(5)])
                es.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
                k = 400
                time_steps(es, 50, 1, sync, barrier)
                t1 = time_steps(es, k, 1, sync, barrier)
                t10 = time_steps(es, k, 10, sync, barrier)
                closed.append({"vehicles": n, "vsteps_per_s": n * k / t1, "vsteps_per_s_fused10": n * k / t10})
                es.close()
            out["closed_loop_on_device"] = closed
            out["perception_rows"] = perception_rows(afa)
            out["sweep"] = sweep
            out["sweep_note"] = ("vsteps_per_s = one launch per step issued from Python; native_loop = the same "
                                 "launches issued by afe_step's C++ loop (afe_set_max_fused_steps(1)); fused2 = two 1 ms steps per launch (nothing is observable between 500 Hz logic "
                                 "ticks); fused50 = 50 steps per launch, state in registers (open-loop commands)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(afa)
    e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
