#!/usr/bin/env python3
"""bench.py -- vehicle-steps/sec of the HIP vehicle-step path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole ensemble: every vehicle
advanced by dt = 1 ms, state read from and written back to HBM (no temporal
fusion in the headline number), one afe_step(e, dt, 1) call per step.  The
engine runs in AFE_STEP_AUTO: up to 2^20 vehicles per GPU one resident grid
serves the steps (afe_set_step_mode: no kernel boundary between two steps),
beyond that one launch per step and half of the shard.  The timed region is
repeated -- blocks of exactly K steps, each bracketed by barrier + synchronise --
until 50 ms have been timed; the line carries the median block.  Workload: the
config-4 shape of BASELINE.json -- a hovering MINIQUAD ensemble with a
per-vehicle wind-gust force from the on-device gust process (the
SetExternalForce port), IMU synthesis at the 500 Hz onboard-logic cadence with
Gaussian noise from the reference's own machinery -- one std::minstd_rand0 +
std::normal_distribution stream per vehicle, words bit-exact (AFE_SEED_DECORRELATED:
a reference vehicle seeded 1 + index draws the same numbers); the same workload on
the engine's counter-based generator (AFE_SEED_COUNTER: no per-vehicle word, no
rejection loop) is measured with the same protocol and printed beside it as
`counter_noise_policy` -- 1,048,576 vehicles PER GPU (weak scaling; inputs
resident in HBM before the timed region).  For N > 1 there is one rank process per GPU: either the caller
starts them (python -m torch.distributed.run ... bench.py --gpus N: RANK /
WORLD_SIZE are in the environment) or `python bench.py --gpus N` starts them
itself (child processes, before this process has touched a GPU) and relays
rank 0's line.  Ranks own contiguous shards and step them with no collective
(the path has none); only the timing uses a barrier and a MAX.  The one exchange
the path has -- the shared-world query: RCCL all-gather of positions + the
neighbour / UWB-ranging consumers, at 100 Hz of simulated time -- is timed
separately in `shared_world`.

Prints ONE JSON line on rank 0 (see the repo task contract), kept below 6 kB
(compact_line; tests/test_bench_line.py holds the bound), including
  roofline     -- algorithmic bytes / measured kernel time vs 8 TB/s, WHERE the working set lives (the 2^20 headline: the
                  256 MiB Infinity Cache; `traffic` is fabric-side bytes there), and `hbm_streaming`: the same kernels at
                  2^23 vehicles, where nothing survives a step on-die -- the path's true HBM row
  cpu_baseline -- the CPU oracle (port of the reference's algorithm) timed on
                  one host core over a bounded sample (rank 0, N = 1 only)
Everything else that is measured (size sweep, perception rows, disturbance
bins, shared-world worlds, notes) goes to the side file bench_detail.json
(and gpurun_out/bench_detail.json where that directory exists).
"""
import argparse
import importlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def _load_provenance():
    import importlib.util
    spec = importlib.util.spec_from_file_location("afe_provenance", os.path.join(ROOT, "agri-fly_amd", "provenance.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


PROV = _load_provenance()      # which kernel sources a committed counter summary was taken on (round-5 review item 5)
STALE = []                     # committed summaries this run refused to borrow from: their kernels have been edited since

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
DT_US = 1000
LOGIC_PERIOD = 1.0 / 500.0


GUST_SEED, NOISE_SEED, GUST_SIGMA_MAX, GUST_PERIOD_US = 4, 5, 0.5, 100000
INFINITY_CACHE_BYTES = 256 << 20      # MI355X_MICROARCH.md: 256 MiB die-level L3
L2_BYTES = 32 << 20                   # 8 XCDs x 4 MiB
L2_PEAK_GBS = 34500.0                 # MI355X_MICROARCH.md: aggregate L2 bandwidth
# The headline's IMU noise: True = the reference's own per-vehicle libstdc++ streams (what a reference vehicle seeded
# 1 + index reproduces word for word), False = the counter-based generator (faster: the `counter_noise_policy` row).
HEADLINE_EXACT_STREAMS = True


def build_shard(afa, n_local, first_global, n_global, device, fext=True, precision=None, exact_stream=None):
    """config 4: hovering CF_MINIQUAD ensemble in one shared world (4 m lattice by GLOBAL index), per-vehicle wind gusts
    from the on-device gust process (sigma swept 0 .. 0.5 N over the global index, resampled every 100 ms), IMU synthesis
    with Gaussian noise at the 500 Hz logic gate -- from the counter-based generator (AFE_SEED_COUNTER), or with
    exact_stream=True from per-vehicle libstdc++ minstd_rand0 / normal_distribution streams (AFE_SEED_DECORRELATED);
    None: the headline's policy (HEADLINE_EXACT_STREAMS)"""
    if exact_stream is None:
        exact_stream = HEADLINE_EXACT_STREAMS
    p = afa.params_from_type(5)  # QC_TYPE_CF_MINIQUAD: vehicle id 1 of every shipped main
    data = afa.scenarios.hover_ensemble(n_local, p)
    idx = np.arange(first_global, first_global + n_local)
    data.pos[0] = (idx % 1024) * 4.0        # the dynamics are translation invariant
    data.pos[1] = (idx // 1024) * 4.0
    e = afa.Ensemble(n_local, precision=afa.AFE_F32 if precision is None else precision, device=device,
                     first_global_index=first_global)
    e.set_type_table([p])
    e.set_logic_period(LOGIC_PERIOD)
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED if exact_stream else afa.AFE_SEED_COUNTER)
    e.set_noise_seed(NOISE_SEED)
    e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
    e.set_motor_cmds(data.motor_cmd)
    if fext:
        e.set_gust_process(True, seed=GUST_SEED, sigma_max=GUST_SIGMA_MAX, period_us=GUST_PERIOD_US, n_global=n_global)
    e.set_step_mode(headline_mode(afa, n_local))      # one resident grid up to 2^20 vehicles, (split) launches beyond
    return e


def headline_mode(afa, n):
    """the stepping the headline and its rows are measured with: the resident grid in its per-step-observable form
    (AFE_STEP_PERSISTENT: every step reads the state from memory and writes it back -- SURVEY 8d's accounting) up to 2^20
    vehicles, launches beyond.  (AFE_STEP_AUTO itself would take the resident-state form, which does not read the state
    back between steps authorised ahead: the `resident_state` companion.)"""
    if os.environ.get("AFE_BENCH_ONE_DEVICE") == "1":
        # the one-GPU test hook (several ranks on device 0): launches.  Two PROCESSES that each hold a resident grid sized for
        # the whole device starve each other's late workgroups (one run in ten ended with "persistent step kernel gave up
        # waiting"); one rank per GPU is the contract, and no number of that hook is meant.
        return afa.AFE_STEP_LAUNCH
    return afa.AFE_STEP_PERSISTENT if n <= (1 << 20) else afa.AFE_STEP_AUTO


def uses_persistent(afa, mode, n):
    mode = headline_mode(afa, n) if mode is None else mode
    return mode in (afa.AFE_STEP_PERSISTENT, afa.AFE_STEP_RESIDENT) or (mode == afa.AFE_STEP_AUTO and n <= (1 << 20))


def time_steps(e, steps, per_launch, sync, barrier):
    """wall time of `steps` physics steps issued as afe_step calls of `per_launch` steps each, bracketed by
    barrier + synchronise on both sides (e.sync() first: it ends a resident grid after the last authorised step)"""
    e.sync()          # no resident grid while the ranks meet (the collective's kernel wants CUs too)
    barrier()
    sync()
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        k = min(per_launch, steps - done)
        e.step(DT_US, k)
        done += k
    e.sync()
    sync()
    t1 = time.perf_counter()      # this rank's own finish; the caller takes the MAX over ranks (the closing barrier's own
    barrier()                     # latency is not part of anybody's K steps)
    return t1 - t0


def timed_blocks(e, steps, per_launch, sync, barrier, reduce_max, min_total_s=0.05, min_blocks=5, max_blocks=2000, own=None, settle_s=0.03):
    """blocks of exactly `steps` steps, each bracketed like time_steps, until `min_total_s` seconds have been timed;
    reduce_max makes every rank see the slowest rank's time of a block, so all ranks run the same number of blocks
    (the loop's conditions are evaluated on reduced values only).  `own` (a list) receives this rank's own times.
    Before the first timed block, untimed blocks of the same shape run for `settle_s` seconds: a device that has just been
    given an engine (allocation, uploads: idle compute units) takes tens of milliseconds to reach its clocks, and a row of
    a small shard is over in less (measured: 131 072 vehicles, 3.6 us per step in the first 50 ms, 3.05 after)."""
    settled = 0.0
    while settled < settle_s:
        settled += max(reduce_max(time_steps(e, steps, per_launch, sync, barrier)), 1e-6)
    # ... and the timed blocks are not served by the FIRST resident grid of an engine: that one is 1-5 % slower for as long
    # as it lives (20 % for the very first of a process; every later grid of the same engine is not: tools/sync_cost_probe.py,
    # DESIGN.md section 6 -- cause open).  Any getter ends it; here afe_grid_time does.
    if settle_s > 0 and hasattr(e, "grid_time"):
        e.grid_time()
    blocks, total = [], 0.0
    while len(blocks) < min_blocks or (total < min_total_s and len(blocks) < max_blocks):
        t_own = time_steps(e, steps, per_launch, sync, barrier)
        t = reduce_max(t_own)
        if own is not None:
            own.append(t_own)
        blocks.append(t)
        total += t
    return blocks


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def residency(footprint_bytes):
    """where a step's working set lives between two steps: the XCDs' L2s, the Infinity Cache, or nowhere on-die (HBM)"""
    if footprint_bytes <= L2_BYTES:
        return "l2"
    if footprint_bytes <= INFINITY_CACHE_BYTES * 0.94:      # (the guide: a table stays resident while table + stream fit in about 256 MiB)
        return "infinity_cache"
    return "hbm"


SIMDS = 1024      # 256 CUs x 4 SIMDs


def bound_fields(n, bytes_per_vehicle_step, seconds_per_step, ns_profile=None):
    """The honest ceiling of a row (round-4 review).  Rows whose working set fits the XCDs' L2s are not bound by any memory
    roofline: they get no HBM fraction but the L2-side rate of the algorithmic bytes against the L2 peak, and a bound by how
    many worker waves share a SIMD: from ~1.5 up the step is bound by vector-instruction ISSUE (131 072 vehicles: two waves
    per SIMD whose vector pipes are busy 77 % of the time -- the committed SQ counters of that very grid, attached to that
    row), below by the LATENCY of one wave's dependent chain (4 096 vehicles: one wave on 64 of 1 024 SIMDs).  Rows beyond
    the L2s get the fraction of the 8 TB/s HBM peak and where the working set really lives (infinity_cache: the bytes are
    served on-die; hbm: they cross the memory interface).  Residency goes by the DISTINCT bytes a step touches (the state is
    read and written in place: counted once; an IMU sample and an engine word exist once though a tick comes every 2nd step)."""
    distinct = n * max(1.0, bytes_per_vehicle_step - 40.0)
    where = residency(distinct)
    rate = n * bytes_per_vehicle_step / seconds_per_step / 1e9
    if where == "l2":
        waves_per_simd = ((n + 63) // 64) / float(SIMDS)
        out = {"bound": "valu_issue" if waves_per_simd >= 1.5 else "latency", "resident_in": "l2", "l2_GBs": rate, "l2_frac": rate / L2_PEAK_GBS}
        if ns_profile and n == 131072 and ns_profile.get("counters_stale"):
            out.update({"counters_stale": True})
        elif ns_profile and n == 131072:
            out.update({"valu_busy_frac": ns_profile.get("simd_valu_busy_frac"), "valu_active_frac_per_wave": ns_profile.get("valu_active_frac"),
                        "valu_instructions_per_wave_step": ns_profile.get("valu_instructions_per_wave_step"), "counters_from": ns_profile.get("counters_from")})
        return out
    return {"bound": "hbm", "resident_in": where, "frac": rate / HBM_PEAK_GBS}


def committed_ns_profile(exact_stream=None):
    """SQ counters of the north-star shard's resident grid under the given noise policy, from the newest committed summary"""
    import glob
    want = "reference_streams" if (HEADLINE_EXACT_STREAMS if exact_stream is None else exact_stream) else "counter"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ns_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("valu_active_frac_of_wave_cycles") and d.get("noise_policy", "counter") == want:
                if not PROV.taken_on_this_tree(d, PROV.STEP_KERNEL):       # the newest summary is of another kernel: nothing older is better
                    _stale(f)
                    return {"counters_stale": True, "stale_summary": os.path.relpath(f, ROOT)}
                return {"valu_active_frac": d["valu_active_frac_of_wave_cycles"], "wait_any_frac": d.get("wait_any_frac_of_wave_cycles"),
                        "simd_valu_busy_frac": d.get("simd_valu_busy_frac"),
                        "valu_instructions_per_wave_step": d.get("valu_instructions_per_wave_and_step"), "counters_from": os.path.relpath(f, ROOT)}
        except (OSError, ValueError):
            pass
    return None


def _stale(path):
    rel = os.path.relpath(path, ROOT)
    if rel not in STALE:
        STALE.append(rel)


def scaling_check(exp, exp_src, world, n_local, value, n_strong, strong_value):
    """This run's per-GPU rates beside what ONE GPU measured on the same shards with the same arguments
    (profiles/expected_rates.json, tools/expected_rates.py): shards do not communicate while stepping, so a ratio near 1 is
    linear scaling and anything well below it is the collectives between blocks, barrier skew or a slow device."""
    if not exp or world < 2:
        return None
    sc = {"from": exp_src, "n_gpus": world}
    w = (exp.get("weak_per_gpu") or {}).get(str(n_local)) or (exp.get("strong_shard") or {}).get(str(n_local))     # (one GPU on a shard of that size, same protocol)
    if w and value:
        sc["weak"] = {"vehicles_per_gpu": n_local, "expected_per_gpu": w["vsteps_per_s"], "measured_per_gpu": value / world, "ratio": value / world / w["vsteps_per_s"]}
    st = (exp.get("strong_shard") or {}).get(str(n_strong))
    if st and strong_value:
        sc["strong"] = {"vehicles_per_gpu": n_strong, "expected_per_gpu": st["vsteps_per_s"], "measured_per_gpu": strong_value / world,
                        "ratio": strong_value / world / st["vsteps_per_s"]}
    return sc if ("weak" in sc or "strong" in sc) else None


def committed_json(pattern, pick, sources=None):
    """the newest committed summary matching `pattern` that `pick` accepts; with `sources` (agri-fly_amd/provenance.py) only
    if it was taken on the kernel sources of this tree -- otherwise (None, None) and the file is listed in STALE"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(f))
            r = pick(d)
            if r is not None:
                if sources is not None and not PROV.taken_on_this_tree(d, sources):
                    _stale(f)
                    return None, None
                return r, os.path.relpath(f, ROOT)
        except (OSError, ValueError, KeyError, TypeError):
            pass
    return None, None


def kernel_time_events(e, launches):
    """device time per step of a back-to-back run of `launches` single-step afe_step calls.  Launch mode: HIP events on the
    engine's stream around that many kernel launches.  Resident grid: it runs on the engine's own AQL queue (no HIP stream
    sees it), so its time is the begin / end device timestamps of its dispatch (afe_grid_time: what rocprofv3
    --kernel-trace reports as the kernel's duration) -- one grid from its first wave to its exit, serving all the steps."""
    ev0, ev1 = e.event(), e.event()
    e.sync()
    e.grid_time()             # ends a resident grid and clears the account
    e.record(ev0)
    for _ in range(launches):
        e.step(DT_US, 1)
    e.record(ev1)
    ms = e.elapsed_ms(ev0, ev1)
    e.destroy_event(ev0)
    e.destroy_event(ev1)
    grid_s, grid_steps = e.grid_time()
    if grid_steps == launches and 0.5 * ms * 1e-3 <= grid_s <= 1.5 * ms * 1e-3:     # (nonsense under a profiler that owns the timestamps)
        return grid_s / launches
    return ms * 1e-3 / launches


def event_blocks(e, launches, min_total_s=0.05, min_blocks=5, max_blocks=400):
    """kernel_time_events repeated until min_total_s of device time has been measured: (median, min, max, repeats)"""
    ts, total = [], 0.0
    while len(ts) < min_blocks or (total < min_total_s and len(ts) < max_blocks):
        t = kernel_time_events(e, launches)
        ts.append(t)
        total += t * launches
    return median(ts), min(ts), max(ts), len(ts)


def mean_bytes_per_step(e, afa, steps):
    # (the cadence in steady state -- the timed blocks come after warm-up and earlier blocks: every 2nd step ticks at
    # dt = 1 ms and 500 Hz -- not the first `steps` steps of a fresh engine, whose first tick is the 3rd step)
    ticks, _ = afa.plan_ticks(LOGIC_PERIOD, 0, DT_US, 4000)
    frac = float(ticks[2000:].mean())
    return frac * e.algorithmic_bytes_per_step(True) + (1 - frac) * e.algorithmic_bytes_per_step(False), frac


def per_kernel_breakdown(afa, n_local, device):
    """HIP-event launch time of the two instantiations the timed region
    alternates between: gate never firing (NOISE=0) / firing every step (NOISE=1)"""
    res = {}
    for name, period in (("off_tick", 1000.0), ("on_tick", 0.0005)):
        e = build_shard(afa, n_local, 0, n_local, device)
        e.set_step_mode(afa.AFE_STEP_LAUNCH)
        e.set_split_stepping(1)      # the kernels themselves: one launch per step for the whole shard
        e.set_logic_period(period)
        for _ in range(50):
            e.step(DT_US, 1)
        t = kernel_time_events(e, 300)
        b = e.algorithmic_bytes_per_step(name == "on_tick")
        res[name] = {"kernel_us": t * 1e6, "bytes_per_vehicle_step": b, "achieved_GBs": n_local * b / t / 1e9}
        e.close()
    return res


def companion_rows(afa, n_local, device, sync, barrier, split=False):
    """Two companions of the headline on the same workload and ensemble size:
    fused_logic_period -- one launch per onboard-logic period (2 steps of 1 ms at 500 Hz): nothing is
      observable between two logic ticks (commands and wrench are held, the IMU is sampled at the tick),
      so this is what afe_step(dt, 2) does for a host that drives its logic at the tick cadence;
    f64 -- the same kernel in the reference's own precision (AFE_F64), one launch per step."""
    rows = {}
    e = build_shard(afa, n_local, 0, n_local, device)
    e.set_step_mode(afa.AFE_STEP_LAUNCH)         # a fused launch is a launch
    e.set_split_stepping(2 if n_local >= (1 << 19) else 1)
    split = n_local >= (1 << 19)
    k = 1000
    time_steps(e, 100, 2, sync, barrier)
    t = time_steps(e, k, 2, sync, barrier)
    ev0, ev1 = e.event(), e.event()
    e.sync()
    e.record(ev0)
    for _ in range(300):
        e.step(DT_US, 2)
    e.record(ev1)
    t_launch = e.elapsed_ms(ev0, ev1) * 1e-3 / 300
    b = e.algorithmic_bytes_per_step(True)      # one pass over the state with one tick in it
    rows["fused_logic_period"] = {"value": n_local * k / t, "unit": "vehicle-steps/s", "steps_per_launch": 2, "launches_per_pass": 2 if split else 1,
                                  "kernel_us": t_launch * 1e6, "algorithmic_bytes_per_launch_per_vehicle": b,
                                  "achieved_GBs": n_local * b / t_launch / 1e9, "frac": n_local * b / t_launch / 1e9 / HBM_PEAK_GBS,
                                  "note": "bitwise the same trajectory as the headline (tests/test_gpu_parity.py::"
                                          "test_fused_steps_equal_single_steps_bitwise); launched kernels (AFE_STEP_LAUNCH), two streams from 2^19 vehicles up"}
    e.destroy_event(ev0)
    e.destroy_event(ev1)
    e.close()
    # the resident grid taking the steps that are already authorised together (AFE_STEP_RESIDENT): inputs once per batch,
    # state in registers from step to step, every step's state stored as it is made
    e = build_shard(afa, n_local, 0, n_local, device)
    e.set_step_mode(afa.AFE_STEP_RESIDENT)
    time_steps(e, 100, 1, sync, barrier)
    k = 2000
    t = median([time_steps(e, k, 1, sync, barrier) for _ in range(3)])
    rows["resident_state"] = {"value": n_local * k / t, "unit": "vehicle-steps/s", "us_per_step": t / k * 1e6,
                              "stored_bytes_per_vehicle_step": 52 + 12, "stored_GBs": n_local * 64 / (t / k) / 1e9,
                              "note": "afe_set_step_mode(AFE_STEP_RESIDENT): bitwise the headline's trajectory; each step's state still lands in memory (52 B + "
                                      "24 B of IMU sample every 2nd step), but is not read back between the steps a host has authorised ahead of the device"}
    e.close()
    e = build_shard(afa, n_local, 0, n_local, device, precision=afa.AFE_F64)      # stepping: automatic, like the headline
    time_steps(e, 50, 1, sync, barrier)
    k = 400
    t = time_steps(e, k, 1, sync, barrier)
    t_kernel = kernel_time_events(e, 200)
    bytes_step, _ = mean_bytes_per_step(e, afa, k)
    rows["f64"] = {"value": n_local * k / t, "unit": "vehicle-steps/s", "dtype": "f64", "stepping": "persistent" if uses_persistent(afa, None, n_local) else "launches", "kernel_us": t_kernel * 1e6,
                   "algorithmic_bytes_per_vehicle_step": bytes_step, "achieved_GBs": n_local * bytes_step / t_kernel / 1e9,
                   "frac": n_local * bytes_step / t_kernel / 1e9 / HBM_PEAK_GBS}
    e.close()
    return rows


def committed_traffic(n_local, exact_stream):
    """PMC-derived bytes per step through the L2s' fabric side (FETCH_SIZE / WRITE_SIZE: Infinity-Cache hits are counted, so
    for a working set that fits the Infinity Cache this is L2 <-> Infinity Cache traffic, not HBM traffic) of this exact
    workload, from the rocprofv3 summary committed under profiles/ (counters cannot be read inside the run)"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        w = t["workload"]
        policy = w.get("noise_policy", "counter")
        if w["vehicles_per_gpu"] == n_local and w["dt_us"] == DT_US and w["fext"] and w["noise"] and policy == ("reference_streams" if exact_stream else "counter"):
            if not PROV.taken_on_this_tree(t, PROV.STEP_KERNEL):
                _stale(path)
                return None, None, None
            return t.get("traffic_bytes_per_step", t["traffic_bytes_per_launch"]), t["source"], t.get("rocprof_kernel_us_per_step")
    except (OSError, KeyError, ValueError):
        pass
    return None, None, None


def config1_row():
    """BASELINE config 1 / configs[0] -- the reference's own CPU-runnable case: ONE vehicle flown by the offboard loop of
    Simulator/Rappids_Simulator/main.cpp (mocap 200 Hz, control 100 Hz, 30 ms radio delay, the log every step) for
    simulated seconds at dt = 1 ms, through the C ABI by agri-fly_amd/bin/rappids_headless (host-visible engine, resident
    grid).  The program times its own loop (steady_clock), so process start and HIP initialisation are outside."""
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "agri-fly_amd", "bin", "rappids_headless")
    if not os.path.exists(exe):
        return {"error": "agri-fly_amd/bin/rappids_headless is not built (__graft_entry__.build() makes it)"}
    seconds, us = 10, None
    with tempfile.TemporaryDirectory() as tmp:
        best = 1e30
        for _ in range(2):
            r = subprocess.run([exe, "--seconds", str(seconds), "--dt-us", "1000", "--out", os.path.join(tmp, "sim.csv")],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            if r.returncode != 0:
                return {"error": "rappids_headless exit %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
            m = re.search(r"Loop wall time ([0-9.eE+-]+) s for (\d+) steps", r.stdout.decode())
            if not m:
                return {"error": "rappids_headless printed no loop wall time"}
            best = min(best, float(m.group(1)) / int(m.group(2)))
        us = best * 1e6
    return {"vehicles": 1, "dt_ms": 1.0, "steps_timed": seconds * 1000, "us_per_step": us,
            "vsteps_per_s": 1e6 / us, "realtime_factor": 1000.0 / us,
            "note": "one vehicle with the host in the loop of every step (state read back and logged every step, mocap estimator and "
                    "controller on the host); the program's own steady_clock around its loop (process start and HIP initialisation outside); "
                    "the reference's CPU loop does this in ~0.5 us per step, a device-arena engine in ~74"}


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
    # the config-3 size on the same images: 65 536 planners (longest-first scheduling applies from 16 385 up)
    n_big = 65536
    idx_b = (np.arange(n_big) % n_views).astype(np.int32)
    vel_b = np.stack([rng.normal(0, 0.3, n_big), rng.normal(0, 0.2, n_big), rng.uniform(0, 2.0, n_big)])
    acc_b = rng.normal(0, 0.3, (3, n_big))
    grav_b = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n_big))
    ms_big = min(afa.rappids_plan(cfg, buf, vel_b, acc_b, grav_b, samples, image_index=idx_b)[2] for _ in range(2))
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
    # ---- what bounds the two perception kernels (counting build / output counters) ----
    st, _ = scene.render_stats(cam, pos, att, mount)
    rays, waves = st["rays"], st["waves"]
    FP64_PEAK_TFLOPS = 78.6            # MI355X fp64 vector peak (spec), MI355X_MICROARCH.md
    mt_flops = 45.0                    # one Moeller-Trumbore evaluation to the end: 27 mul + 18 add/sub (+ 1 division)
    ray_flops = 9 * 2 + 6 + 4          # direction = R (u, v, 1), final floor(z / scale)
    fp64_flops = st["tri_fp64_tests_per_ray"] * mt_flops + rays * ray_flops
    node_bytes = st["nodes_per_wave"] * 64.0 + st["tri_box_tests_per_wave"] * 96.0   # scalar loads, served by L2
    rpmc, rsrc = committed_json("r*_render_pmc.json", lambda d: d if d.get("valu_issue_fraction_of_busy_cycles") else None, PROV.RENDER_KERNEL)
    ppmc, psrc = committed_json("r*_planner_pmc.json", lambda d: d if d.get("valu_issue_fraction_of_busy_cycles") else None, PROV.PLANNER_KERNEL)
    rpmc, ppmc = rpmc or {}, ppmc or {}
    render_roofline = {
        "bound": "vector-instruction issue (a vector instruction issues in most of the SIMDs' busy cycles: the committed SQ counters below), not HBM and not fp64 throughput",
        "valu_issue_fraction_of_busy_cycles": rpmc.get("valu_issue_fraction_of_busy_cycles"),
        "valu_instructions_per_ray": (rpmc.get("per_ray") or {}).get("valu_instructions", (rpmc.get("per_wave") or {}).get("valu_instructions", 0) / 64.0 or None),
        "counters_from": rsrc,
        "floor_valu_per_ray": render_floor(st)["floor_valu_per_ray"], "floor": render_floor(st),
        "per_ray": {"triangle_box_tests": st["tri_box_tests_per_ray"] / rays, "fp64_triangle_tests": st["tri_fp64_tests_per_ray"] / rays},
        "per_wave_of_64_rays": {"nodes_visited": st["nodes_per_wave"] / waves, "triangle_box_tests": st["tri_box_tests_per_wave"] / waves,
                                "fp64_triangle_tests_executed": st["tri_fp64_tests_per_wave"] / waves},
        "fp64_TFLOPs_achieved": fp64_flops / (ms_render * 1e-3) / 1e12, "fp64_peak_TFLOPs": FP64_PEAK_TFLOPS,
        "fp64_frac": fp64_flops / (ms_render * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
        "tree_bytes_GBs": node_bytes / (ms_render * 1e-3) / 1e9,
        "hbm_bytes_GBs": rays * 2.0 / (ms_render * 1e-3) / 1e9, "hbm_frac": rays * 2.0 / (ms_render * 1e-3) / 1e9 / HBM_PEAK_GBS}
    plans = afa.plans_as_array(out)
    px_bytes = 240 * 320 * 2
    # per pyramid the search kernel sweeps the image twice (the two bit images) and once more over the grown
    # rectangle; plus the candidate kernel's and the transpose's passes over the inputs
    plan_bytes = float(plans["n_pyramids"].sum()) * 2.0 * px_bytes + n_views * 2.0 * px_bytes + n_planners * n_candidates * 9.0
    planner_roofline = {
        "bound": "latency of a sequential search (one wave per planner; longest-first scheduling keeps the shader engines busy to the end of the launch: "
                 "the committed SQ counters below), not HBM",
        "valu_issue_fraction_of_busy_cycles": ppmc.get("valu_issue_fraction_of_busy_cycles"),
        "valu_instructions_per_plan": (ppmc.get("per_plan") or {}).get("valu_instructions"),
        "counters_from": psrc,
        "pyramids_per_plan": float(plans["n_pyramids"].mean()), "collision_checks_per_plan": float(plans["n_collision_checks"].mean()),
        "algorithmic_bytes": plan_bytes, "achieved_GBs": plan_bytes / (ms_plan * 1e-3) / 1e9,
        "hbm_frac": plan_bytes / (ms_plan * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return {"closed_perception_loop_frame": frame,
            "depth_camera": {"views": n_views, "image": "320x240", "triangles": int(info["n_tri"]),
                             "kernel_ms": ms_render, "rays_per_s": n_views * 76800 / (ms_render * 1e-3),
                             "roofline": render_roofline},
            "rappids_planner": {"planners": n_planners, "candidates": n_candidates, "distinct_images": n_views,
                                "kernel_ms": ms_plan, "plans_per_s": n_planners / (ms_plan * 1e-3),
                                "fraction_found": found, "roofline": planner_roofline,
                                "config3_size": {"planners": n_big, "kernel_ms": ms_big, "plans_per_s": n_big / (ms_big * 1e-3),
                                                 "scheduling": "longest first: a 0.4 ms sizing round, then one finishing round in the order of the collision checks still ahead"}},
            "note": "images rendered from engine state and planned on without leaving HBM; results are "
                    "bit-identical to the CPU checkers in tests/test_gpu_render.py / test_gpu_planner.py"}


def _rot64(q):
    """Rotation.hpp:196-220 for planar quaternions [4, n] -> [3, 3, n]"""
    r0, r1, r2, r3 = q[0] * q[0], q[1] * q[1], q[2] * q[2], q[3] * q[3]
    return np.array([[r0 + r1 - r2 - r3, 2 * q[1] * q[2] - 2 * q[0] * q[3], 2 * q[1] * q[3] + 2 * q[0] * q[2]],
                     [2 * q[1] * q[2] + 2 * q[0] * q[3], r0 - r1 + r2 - r3, 2 * q[2] * q[3] - 2 * q[0] * q[1]],
                     [2 * q[1] * q[3] - 2 * q[0] * q[2], 2 * q[2] * q[3] + 2 * q[0] * q[1], r0 - r1 - r2 + r3]])


def _qmul64(a, b):
    """Rotation.hpp:124-131 (this = a, r1 = b)"""
    return np.stack([b[0] * a[0] - b[1] * a[1] - b[2] * a[2] - b[3] * a[3], b[1] * a[0] + b[0] * a[1] + b[3] * a[2] - b[2] * a[3],
                     b[2] * a[0] - b[3] * a[1] + b[0] * a[2] + b[1] * a[3], b[3] * a[0] + b[2] * a[1] - b[1] * a[2] + b[0] * a[3]])


def config3_row(afa, device=0, n=65536, frames=3, n_candidates=192, steps_per_frame=30):
    """BASELINE config 3 AT ITS SIZE in the metric's unit: 65 536 vehicles with the RAPPIDS planner in the loop, the device
    side of one camera frame as tests/test_gpu_configs_full.py::test_config3_65536_vehicles_planner_in_the_loop flies it
    (tests/orchard_flight.py: the 6 x 10-tree orchard, vehicles west of it facing +x, 192 candidates, main.cpp's cost
    function towards a goal east of the orchard, a plan every 3rd offboard tick = 30 steps of 1 ms):
        30 steps of physics + IMU + on-device rates logic  ->  one depth image per vehicle from engine state (10 GB, stays
        in HBM)  ->  one plan per vehicle on its own image.
    The offboard tracking controller between two frames is host code outside SURVEY 8 (caller side) and is not timed: the
    rate command stays the hover command, as in the first 40 ms of the test.  vsteps_per_s = n x 30 / (physics + render + plan)."""
    sc = afa.scenarios
    rows, cols, altitude = 6, 10, 1.2
    tris = sc.orchard_mesh(rows=rows, cols=cols, seed=0)
    scene = afa.Scene(tris, device=device)
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    params = afa.params_from_type(5)
    rng = np.random.default_rng(0)
    lane = rng.integers(0, rows - 1, n)
    on_row = rng.random(n) < 0.5
    y0 = np.where(on_row, lane * 4.0 + rng.uniform(-0.3, 0.3, n), lane * 4.0 + 2.0 + rng.uniform(-0.8, 0.8, n))
    pos0 = np.stack([np.full(n, -4.0) + rng.uniform(-1, 0, n), y0, np.full(n, altitude)])
    goal = np.stack([np.full(n, (cols - 1) * 3.0 + 8.0), y0, np.full(n, altitude)])
    att0 = np.tile(np.array([[1.0], [0.0], [0.0], [0.0]]), (1, n))
    e = afa.Ensemble(n, precision=afa.AFE_F32, device=device)
    e.set_type_table([params])
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    e.set_state(pos0, np.zeros((3, n)), att0, np.zeros((3, n)), np.full((4, n), sc.hover_speed(params)))
    e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    buf = afa.DeviceBuffer(n * 240 * 320 * 2, device=device)
    cfg = afa.planner_default_config(320, 240, cam.depth_scale, cam.focal_length, 2 * params.arm_length, 3 * params.arm_length, 0.5)   # main.cpp:167-169
    cfg.cost_type = 1
    samples = afa.planner_samples(0, 320, 240, n_candidates)
    rec = []
    for f in range(frames + 1):                 # the first frame is warm-up (scratch allocation of the planner)
        t0 = time.perf_counter()
        e.step(1000, steps_per_frame)
        e.sync()
        ms_phys = (time.perf_counter() - t0) * 1e3
        st = e.get_state()
        pos, vel, att = st["pos"], st["vel"], st["att"]
        R_att = _rot64(att)
        R_cam = _rot64(_qmul64(att, np.tile(mount[:, None], (1, n))))
        inv = lambda R, v: np.einsum("jin,jn->in", R, v)
        e3 = np.zeros((3, n))
        e3[2] = 1
        vel_c = inv(R_cam, vel)
        acc_c = inv(R_cam, np.einsum("ijn,jn->in", R_att, e3) * 9.81 - np.array([[0], [0], [9.81]]))
        grav_c = inv(R_cam, np.tile(np.array([[0.0], [0.0], [-9.81]]), (1, n)))
        goal_c = inv(R_cam, goal - pos)
        ms_r = scene.render_engine(e, cam, mount, out=buf)
        out, _, ms_p = afa.rappids_plan(cfg, buf, vel_c, acc_c, grav_c, samples, cost_vec=goal_c, device=device)
        if f:
            plans = afa.plans_as_array(out)
            rec.append({"physics_ms": ms_phys, "render_ms": ms_r, "plan_ms": ms_p, "found": float(plans["found"].mean()),
                        "pyramids_per_plan": float(plans["n_pyramids"].mean()), "collision_checks_per_plan": float(plans["n_collision_checks"].mean())})
    buf.close()
    e.close()
    scene.close()
    afa.planner_release_scratch()
    m = {k: median([r[k] for r in rec]) for k in rec[0]}
    frame_ms = m["physics_ms"] + m["render_ms"] + m["plan_ms"]
    return {"vehicles": n, "frames_timed": frames, "steps_per_frame": steps_per_frame, "candidates": n_candidates, "triangles": int(len(tris)),
            "frame_ms": frame_ms, "physics_ms": m["physics_ms"], "render_ms": m["render_ms"], "plan_ms": m["plan_ms"],
            "vsteps_per_s": n * steps_per_frame / (frame_ms * 1e-3), "rays_per_s": n * 76800 / (m["render_ms"] * 1e-3), "plans_per_s": n / (m["plan_ms"] * 1e-3),
            "fraction_found": m["found"], "pyramids_per_plan": m["pyramids_per_plan"], "collision_checks_per_plan": m["collision_checks_per_plan"],
            "bound": "perception: depth camera (vector issue) + planner search; the physics is %.1f %% of the frame" % (100 * m["physics_ms"] / frame_ms)}


def config5_row(afa, device=0, n=262144, frames=2, shard=32768, steps_per_frame=33):
    """BASELINE config 5 AT ITS SIZE in the metric's unit: 262 144 vehicles + depth raycast against the orchard triangle mesh
    (32 x 32 trees), the device side of one camera frame as tests/test_gpu_configs_full.py::test_config5_... runs it: 33 steps
    of physics + IMU + rates logic (one 30 Hz camera period), then every vehicle's 320 x 240 depth image from engine state,
    in eight chunks of 32 768 views (one GPU's share of the 8-GPU split: a 5 GB image buffer)."""
    sc = afa.scenarios
    tris = sc.orchard_mesh(rows=32, cols=32, seed=1)
    scene = afa.Scene(tris, device=device)
    info = scene.info()
    cam = afa.camera_default(320, 240)
    mount = afa.camera_default_mount()
    rng = np.random.default_rng(5)
    lo, hi = info["bounds"][:3], info["bounds"][3:]
    pos = np.stack([rng.uniform(lo[0] + 2, hi[0] - 2, n), rng.uniform(lo[1] + 2, hi[1] - 2, n), rng.uniform(0.5, 3.0, n)])
    att = sc.random_attitudes(rng, n, max_tilt_deg=20.0)
    params = afa.params_from_type(5)
    e = afa.Ensemble(n, precision=afa.AFE_F32, device=device)
    e.set_type_table([params])
    e.set_imu_noise(True, 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    e.set_state(pos, rng.normal(0, 0.5, (3, n)), att, np.zeros((3, n)), np.full((4, n), sc.hover_speed(params)))
    e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    buf = afa.DeviceBuffer(shard * 240 * 320 * 2, device=device)
    rec = []
    for f in range(frames + 1):
        t0 = time.perf_counter()
        e.step(1000, steps_per_frame)
        e.sync()
        ms_phys = (time.perf_counter() - t0) * 1e3
        ms_r = sum(scene.render_engine(e, cam, mount, first=c * shard, count=shard, out=buf) for c in range(n // shard))
        if f:
            rec.append({"physics_ms": ms_phys, "render_ms": ms_r})
    # traversal counters of a sample of this frame's views (counting build of the kernel)
    st = e.get_state()
    k = 256
    pick = np.sort(rng.choice(n, k, replace=False))
    stats, _ = scene.render_stats(cam, st["pos"][:, pick], st["att"][:, pick], mount)
    buf.close()
    e.close()
    scene.close()
    m = {q: median([r[q] for r in rec]) for q in rec[0]}
    frame_ms = m["physics_ms"] + m["render_ms"]
    return {"vehicles": n, "frames_timed": frames, "steps_per_frame": steps_per_frame, "triangles": int(info["n_tri"]), "bvh_nodes": int(info["n_nodes"]),
            "views_per_chunk": shard, "frame_ms": frame_ms, "physics_ms": m["physics_ms"], "render_ms": m["render_ms"],
            "vsteps_per_s": n * steps_per_frame / (frame_ms * 1e-3), "rays_per_s": n * 76800 / (m["render_ms"] * 1e-3),
            "per_ray": render_per_ray(stats),
            "bound": "depth camera (vector issue); the physics is %.1f %% of the frame" % (100 * m["physics_ms"] / frame_ms)}


# Vector instructions of the depth camera's wave by what they are spent on, counted in the gfx950 listing of
# afe_render_depth_kernel<false> (make -C agri-fly_amd/csrc asm-style listing, ordered walk; round 6): ray set-up + octant vote +
# quantise / store; one PairNode visit (two boxes: 6 v_pk_fma + min3 / max3 + compares); one triangle's own box; one
# double-precision Moeller-Trumbore test to the end with the best-hit update (62 fp64 instructions, an IEEE division inside).
RENDER_VALU = {"ray_setup_and_output": 132, "node_visit": 16, "triangle_box_test": 8, "fp64_triangle_test": 84}


def render_floor(st):
    """What 64 rays cost in vector instructions by the counting build's tallies and the static costs above, and the FLOOR of
    any traversal that keeps the contract (every pixel's count from the double-precision test of the triangle it shows): ray
    set-up, ONE fp64 test per triangle a tile really shows, the store -- no node, no box, no test of a hidden triangle."""
    waves = float(st["waves"])
    nodes, boxes, mts, vis = (st[k] / waves for k in ("nodes_per_wave", "tri_box_tests_per_wave", "tri_fp64_tests_per_wave", "visible_triangles_per_wave"))
    c = RENDER_VALU
    model = c["ray_setup_and_output"] + nodes * c["node_visit"] + boxes * c["triangle_box_test"] + mts * c["fp64_triangle_test"]
    floor = c["ray_setup_and_output"] + vis * c["fp64_triangle_test"]
    return {"visible_triangles_per_wave": vis, "valu_per_wave_by_static_costs": model, "valu_per_ray_by_static_costs": model / 64.0,
            "valu_per_node_visit": c["node_visit"], "floor_valu_per_wave": floor, "floor_valu_per_ray": floor / 64.0, "floor_over_model": floor / model,
            "traversal_share": (nodes * c["node_visit"] + boxes * c["triangle_box_test"]) / model,
            "hidden_or_missed_fp64_tests_share": max(0.0, mts - vis) * c["fp64_triangle_test"] / model}


def render_per_ray(st):
    """what one ray of the depth camera costs in tree work (the counting build's totals over a batch): nodes its wave visits
    (one node per wave step, shared by the 64 rays of the tile), triangle-box tests and double-precision triangle tests"""
    rays, waves = float(st["rays"]), float(st["waves"])
    return {"nodes_visited_per_wave": st["nodes_per_wave"] / waves, "nodes_visited_per_ray": st["nodes_per_wave"] / rays,
            "triangle_box_tests_per_ray": st["tri_box_tests_per_ray"] / rays, "fp64_triangle_tests_per_ray": st["tri_fp64_tests_per_ray"] / rays,
            "triangle_box_tests_per_wave": st["tri_box_tests_per_wave"] / waves, "fp64_triangle_tests_per_wave": st["tri_fp64_tests_per_wave"] / waves,
            "floor": render_floor(st)}


def host_cores():
    """(threads this process may run on, cgroup CPU quota in cores or None, CPU model): os.cpu_count() is the HOST's count --
    a container is often given a slice of it (round 4: 256 reported, 8 usable)"""
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),):
        try:
            quota = parse(open(path).read())
        except (OSError, ValueError, IndexError):
            pass
    if quota is None:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return affinity, quota, model


def cpu_baseline(afa, budget_vehicle_steps=20_000_000, all_cores=True):
    """the oracle (double, scalar C) on a bounded sample of the same workload; test infrastructure used here only as the
    reported baseline.  Three legs, ~10 s together: one thread under the headline's noise policy (cpu_baseline.value: the same
    workload as `value`), one thread under the other policy, and every core this process may use (OpenMP over vehicles, >= 64 vehicles per thread, vehicles outer /
    steps inner: no shared data)."""
    from oracle import oracle_py
    p = afa.params_from_type(5)

    def batch(n):
        data = afa.scenarios.hover_ensemble(n, p)
        b = oracle_py.Batch(n, [oracle_py.params_from_type(5)])
        b.pos[:], b.vel[:], b.att[:], b.ang_vel[:] = data.pos, data.vel, data.att, data.ang_vel
        b.motor_speed[:], b.motor_cmd[:] = data.motor_speed, data.motor_cmd
        b.rng[:] = 1 + np.arange(n, dtype=np.uint32)          # AFE_SEED_DECORRELATED: seed = 1 + global index
        return b

    def run(b, k, t0_us, tick_base, counter):
        ticks, _ = afa.plan_ticks(LOGIC_PERIOD, 0, DT_US, k)
        t0 = time.perf_counter()
        oracle_py.step_counter(b, DT_US, k, ticks, counter_noise=counter, seed=NOISE_SEED, first_global=0, tick_base=tick_base,
                               gust_seed=GUST_SEED, gust_period_us=GUST_PERIOD_US, t0_us=t0_us, n_global=1 << 20, sigma_max=GUST_SIGMA_MAX)
        return time.perf_counter() - t0

    n = 16384
    steps = max(10, budget_vehicle_steps // n)
    head_counter = not HEADLINE_EXACT_STREAMS            # the headline's noise policy first: cpu_baseline.value is the SAME workload
    names = {True: "counter-based noise (Philox + Box-Muller, no libstdc++ draw)", False: "the reference's libstdc++ streams (minstd_rand0 + polar method, seed 1 + index)"}
    b = batch(n)
    run(b, 10, 0, 0, head_counter)  # warm
    dt = run(b, steps, 10 * DT_US, 5, head_counter)
    affinity, quota, model = host_cores()
    out = {"value": n * steps / dt, "unit": "vehicle-steps/s", "cores": 1, "kind": "port",
           "noise_policy": "counter" if head_counter else "reference_streams",
           "sample": "%d vehicles x %d steps of the same workload (gust process, IMU + %s every 2nd step), "
                     "oracle/agrifly_oracle.c + agrifly_oracle_counter.c double precision, gcc -O2, 1 thread, %.1f s" % (n, steps, names[head_counter], dt),
           "cpu_model": model, "host_cpus": os.cpu_count(), "affinity_cpus": affinity, "cgroup_quota_cpus": quota}
    # the other noise policy, one thread
    bx = batch(n)
    steps_x = max(10, steps // 2)
    run(bx, 10, 0, 0, not head_counter)
    dtx = run(bx, steps_x, 10 * DT_US, 5, not head_counter)
    out["other_noise_policy"] = {"value": n * steps_x / dtx, "unit": "vehicle-steps/s", "cores": 1, "noise_policy": "reference_streams" if head_counter else "counter",
                                 "sample": "%d vehicles x %d steps, %s, 1 thread, %.1f s" % (n, steps_x, names[not head_counter], dtx)}
    # the same port on every core this process may use, SURVEY 8d CPU-baseline (ii)
    threads = max(1, min(affinity, int(np.ceil(quota)) if quota else affinity))
    if threads > 1 and all_cores:
        n_mt = max(n, 64 * threads)
        bm = batch(n_mt)
        oracle_py.lib().ora_set_batch_threads(threads)
        run(bm, 10, 0, 0, head_counter)
        steps_mt = int(min(20000, max(100, 3.0 * out["value"] * threads / n_mt)))     # ~3 s if it scales
        dt_mt = run(bm, steps_mt, 10 * DT_US, 5, head_counter)
        oracle_py.lib().ora_set_batch_threads(1)
        v = n_mt * steps_mt / dt_mt
        out["all_cores"] = {"value": v, "unit": "vehicle-steps/s", "cores": threads, "speedup_over_one_thread": v / out["value"],
                            "sample": "%d vehicles x %d steps, %d OpenMP threads (%d vehicles each), %.1f s" % (n_mt, steps_mt, threads, n_mt // threads, dt_mt)}
    return out


def disturbance_sweep(afa, device, n=1 << 20, seconds=10.0):
    """BASELINE config 4 as the Monte-Carlo sweep it names: n vehicles hover in closed loop (on-device rates logic, thrust
    command g, rate command 0) for `seconds` under the gust process, sigma swept 0 .. 0.5 N over the vehicle index; the
    result is the RMS horizontal deviation from the start position per sigma bin, next to the closed form for a level
    vehicle under piecewise-constant white acceleration (Var x(T) = (sigma/m)^2 tau^4 sum_j (j + 1/2)^2)."""
    p = afa.params_from_type(5)
    e = build_shard(afa, n, 0, n, device)
    e.set_step_mode(afa.AFE_STEP_AUTO)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
    p0 = e.get_state()["pos"]
    steps = int(round(seconds * 1e6 / DT_US))
    t0 = time.perf_counter()
    e.step(DT_US, steps)
    e.sync()
    wall = time.perf_counter() - t0
    st = e.get_state()
    e.close()
    dev2 = (st["pos"][0] - p0[0]) ** 2 + (st["pos"][1] - p0[1]) ** 2
    sigma = GUST_SIGMA_MAX * np.arange(n) / (n - 1)
    K, tau = steps * DT_US // GUST_PERIOD_US, GUST_PERIOD_US * 1e-6
    unit = tau ** 4 * sum((j + 0.5) ** 2 for j in range(K))          # per axis, per (m/s^2)^2
    bins = []
    edges = np.linspace(0, n, 9).astype(int)
    for a, b in zip(edges[:-1], edges[1:]):
        closed = np.sqrt(2 * unit * np.mean((sigma[a:b] / p.mass) ** 2))
        bins.append({"sigma_N": [float(sigma[a]), float(sigma[b - 1])], "rms_xy_m": float(np.sqrt(dev2[a:b].mean())),
                     "gust_only_closed_form_m": float(closed), "on_the_ground_fraction": float((st["pos"][2, a:b] <= 0).mean())})
    return {"vehicles": n, "simulated_seconds": seconds, "steps": steps, "wall_s": wall, "vsteps_per_s": n * steps / wall,
            "gust": {"sigma_max_N": GUST_SIGMA_MAX, "epoch_ms": GUST_PERIOD_US / 1e3, "seed": GUST_SEED},
            "bins": bins,
            "note": "closed-loop hover: IMU synthesis + onboard rates logic on the device every 2 ms; no position or attitude loop (those are "
                    "offboard / host-side in the reference), so a vehicle drifts with the integrated gust acceleration -- the closed form -- plus what "
                    "the gyro noise does to its attitude through the rates loop (the sigma = 0 end of the sweep)"}


class stdout_to_stderr:
    """stdout carries exactly one JSON line.  RCCL prints a version banner through C stdio when a
    communicator comes up (and it may sit in the C buffer until exit), so communicators are created
    with fd 1 pointing at fd 2, and the C buffers are flushed before fd 1 is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def shared_world(afa, n_local, rank, world, local_rank, dist, torch, sync, barrier):
    """The path's only exchange, at its cadence: every 10 ms of simulated time (100 Hz) the shards all-gather their
    positions (afe_gather_positions: pack + ncclAllGather on the engine's stream) and run the consumers on the gathered
    buffer (uniform-grid nearest neighbour for every local vehicle; 1024 UWB ranging transactions).  Two worlds of
    defined age -- the 4 m lattice as it starts, and the same ensemble after 3000 steps of gusts -- each with the
    query in the engine's stream (the steps wait for it) and on a stream of its own (afe_nearest_neighbour_async: the
    next ten steps run beside it)."""
    n_all = n_local * world
    uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        uid.copy_(torch.from_numpy(afa.Comm.unique_id()))
    if dist is not None:
        dist.broadcast(uid, 0)
    with stdout_to_stderr():
        comm = afa.Comm(uid.cpu().numpy(), rank, world, device=local_rank)
    xyz = torch.empty((3, n_all), dtype=torch.float32, device="cuda")
    d2 = torch.empty(n_local, dtype=torch.float32, device="cuda")
    idx = torch.empty(n_local, dtype=torch.int32, device="cuda")
    net = afa.UwbNetwork(0.05, 0.01, 3.0)
    rng = np.random.default_rng(7)
    req = rng.integers(0, n_all, 1024).astype(np.int32)
    res = rng.integers(0, n_all, 1024).astype(np.int32)

    one_device = os.environ.get("AFE_BENCH_ONE_DEVICE") == "1"      # (gloo instead of RCCL: the collective's tensor lives on the host)

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if one_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = {"rccl_ranks": None, "vehicles_gathered": n_all, "allgather_bytes_per_rank": 12 * n_local,
           "query_cadence": "every 10 steps of 1 ms (100 Hz simulated time)", "neighbour_grid_reshaped_every_n_queries": 16,
           "vehicles_sorted_into_cells_every_n_queries": 8, "worlds": {}}
    for world_name, age in (("lattice_as_started", 0), ("after_3000_steps_of_gusts", 3000)):
        e = build_shard(afa, n_local, rank * n_local, n_all, local_rank)
        e.set_step_mode(afa.AFE_STEP_AUTO)               # ten steps per call: the engine fuses them
        with stdout_to_stderr():
            e.gather_positions(comm, xyz.data_ptr())     # RCCL's first call on this communicator
            e.sync()
        if age:
            e.step(DT_US, age)
        e.set_neighbour_grid_refresh(16)     # re-shape the grid (one read-back) every 16th query; exact either way
        e.set_neighbour_sort_reuse(8)        # sort into cells every 8th query, keep the order and refresh the positions in between; exact either way

        def query(asynchronous=False, with_uwb=False):
            e.gather_positions(comm, xyz.data_ptr())
            if asynchronous:
                e.nearest_neighbour_async(xyz.data_ptr(), n_all, d2.data_ptr(), idx.data_ptr())
            else:
                e.nearest_neighbour(xyz.data_ptr(), n_all, d2.data_ptr(), idx.data_ptr())
            if with_uwb:
                net.range(e, xyz.data_ptr(), n_all, req, res)

        for _ in range(3):
            query(with_uwb=True)
        e.sync(); sync(); barrier()
        parts = {}
        for name, fn in (("allgather_ms", lambda: e.gather_positions(comm, xyz.data_ptr())),
                         ("nearest_neighbour_ms", lambda: e.nearest_neighbour(xyz.data_ptr(), n_all, d2.data_ptr(), idx.data_ptr())),
                         ("uwb_1024_ranges_ms", lambda: net.range(e, xyz.data_ptr(), n_all, req, res))):
            ev0, ev1 = e.event(), e.event()
            barrier()
            e.record(ev0)
            for _ in range(10):
                fn()
            e.record(ev1)
            parts[name] = e.elapsed_ms(ev0, ev1) / 10
            e.destroy_event(ev0)
            e.destroy_event(ev1)

        # physics with a query every 10 steps (100 Hz at dt = 1 ms) vs physics alone
        def run(k_steps, every, asynchronous=False, per_call=10):
            barrier()
            e.query_sync(); e.sync(); sync()
            t0 = time.perf_counter()
            for s in range(0, k_steps, 10):
                for _ in range(10 // per_call):
                    e.step(DT_US, per_call)   # ten steps per call: nobody looks in between, the engine (AFE_STEP_AUTO) fuses them
                if every:
                    query(asynchronous)     # (the UWB read-back would serialise the host; timed above on its own)
            e.query_sync(); e.sync(); sync()
            barrier()
            return time.perf_counter() - t0
        k = 400
        # (the runs with queries first: the world ages with every step flown, and the query's cost follows its age --
        # the physics-only runs do not care)
        run(50, 10)
        t_sync = reduce_max(median([run(k, 10) for _ in range(3)]))
        run(50, 10, True)
        t_async = reduce_max(median([run(k, 10, True) for _ in range(3)]))
        t_without = reduce_max(median([run(k, 0) for _ in range(3)]))
        t_single = reduce_max(median([run(k, 0, per_call=1) for _ in range(3)]))   # the headline's protocol: every step its own call
        info = e.neighbour_grid_info()
        e.query_sync(); e.sync()
        w = dict(parts)
        w.update({"query_ms": parts["allgather_ms"] + parts["nearest_neighbour_ms"],
                  "vsteps_per_s_physics_only": n_all * k / t_without,
                  "vsteps_per_s_physics_only_one_step_per_call": n_all * k / t_single,
                  "vsteps_per_s_query_in_stream": n_all * k / t_sync, "fraction_query_in_stream": t_without / t_sync,
                  "vsteps_per_s_query_on_own_stream": n_all * k / t_async, "fraction_query_on_own_stream": t_without / t_async,
                  "neighbour_grid": {"dims": list(info["dims"]), "cell_size_m": info["cell_size"], "queries_finished_by_brute_force": info["n_bruteforce"]},
                  "min_separation_m": float(torch.sqrt(d2.min()).item())})
        out["worlds"][world_name] = w
        out["rccl_ranks"] = comm.info()[1]
        e.close()
    out["note"] = ("a query = all-gather + exact nearest neighbour of every local vehicle among all gathered ones, every 10 steps; the ten steps between two "
                   "queries are ONE afe_step call (nothing is observable in between, so AFE_STEP_AUTO runs them as a fused launch: state in registers, "
                   "~9 us per step at 2^20 vehicles instead of the ~19 of one observable step per call -- vsteps_per_s_physics_only_one_step_per_call is the "
                   "headline's protocol on the same engine).  fraction_* = rate with queries / rate without, both with ten steps per call: the query's "
                   "~0.1-0.15 ms now stand against ~0.09 ms of physics.  Both live on the memory system: running the query beside the next ten steps "
                   "(own stream) hides its launch gaps and little else")
    net.close()
    with stdout_to_stderr():
        comm.close()
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes here (agri-fly_amd/launch.py).
    This process never imports torch and never touches a GPU; it relays rank 0's JSON line and the exit codes."""
    launch = importlib.import_module("agri-fly_amd.launch")
    print(launch.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))


def shard_row(afa, n, device, sync, barrier, reduce_max, block_steps, mode=None, parts=None, min_total_s=0.05):
    """one ensemble size, measured like the headline (repeated bracketed blocks, median): microseconds per step,
    vehicle-steps/s and the roofline fraction on algorithmic bytes; mode/parts override the engine's automatic choice"""
    e = build_shard(afa, n, 0, n, device)
    if mode is not None:
        e.set_step_mode(mode)
    if parts is not None:
        e.set_split_stepping(parts)
    time_steps(e, 100, 1, sync, barrier)
    blocks = timed_blocks(e, block_steps, 1, sync, barrier, reduce_max, min_total_s=min_total_s)
    t = median(blocks) / block_steps
    bytes_step, _ = mean_bytes_per_step(e, afa, block_steps)
    row = {"vehicles": n, "us_per_step": t * 1e6, "vsteps_per_s": n / t, "algorithmic_GBs": n * bytes_step / t / 1e9,
           "stepping": "persistent" if uses_persistent(afa, mode, n) else ("split launches" if n >= (1 << 19) and parts != 1 else "launches"),
           "block_steps": block_steps, "repeats": len(blocks)}
    row.update(bound_fields(n, bytes_step, t, committed_ns_profile()))
    return e, row


LINE_LIMIT = 6000            # bytes of the printed JSON line (the driver's parser lost a 20 kB line in round 3)
DETAIL_FILE = "bench_detail.json"


def _r(x, sig=6):
    """numbers rounded to `sig` significant digits (the line is for reading and parsing, the side file keeps full precision)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d and d[k] is not None}


def compact_line(full):
    """The ONE printed line: the contract keys, a one-sentence workload, a compact roofline, cpu_baseline, the strong row
    (config 4 as stated), the north-star shard, the reference-exact noise row and {name: value} companions.  Everything
    else stays in `full`, which goes to bench_detail.json.  Never longer than LINE_LIMIT: if a caller's strings push it
    over, the optional objects are dropped one by one (least important first) until it fits."""
    cfg = full.get("config", {})
    roof = full.get("roofline") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg.get("workload_short", cfg.get("workload", ""))[:400],
                      "vehicles_per_gpu": cfg.get("vehicles_per_gpu"), "vehicles_total": cfg.get("vehicles_total"),
                      "dt_us": cfg.get("dt_us"), "steps_per_call": cfg.get("steps_per_call"), "stepping": cfg.get("stepping"),
                      "noise": cfg.get("noise"), "parallelism": cfg.get("parallelism_short", cfg.get("parallelism"))}
    line.update(_pick(full, ("repeats", "ms_per_step_min", "ms_per_step_max", "headline_note")) or {})
    r = _pick(roof, ("bound", "resident_in", "working_set_bytes", "infinity_cache_bytes", "achieved", "peak", "unit", "frac", "traffic", "traffic_is",
                     "traffic_source", "kernel_us", "kernel_us_min", "kernel_us_max",
                     "kernel_us_rocprof", "algorithmic_bytes_per_vehicle_step")) or {}
    r.setdefault("traffic", None)          # (the contract's key: a number from the committed PMC summary, or null)
    kr = roof.get("kernel_us_rocprof")
    if isinstance(kr, dict):      # the committed trace holds one figure per block length: the line carries this run's
        nums = {k: v for k, v in kr.items() if isinstance(v, (int, float))}
        r["kernel_us_rocprof"] = nums.get("blocks_of_%s_steps" % full.get("steps"), next(iter(nums.values()), None))
    r["kernel"] = (roof.get("kernel_short") or roof.get("kernel") or "")[:120]
    if roof.get("traffic_source"):
        r["traffic_source"] = str(roof["traffic_source"])[:60]
    pm = roof.get("peak_measured")
    if pm:
        r["peak_measured"] = [pm.get("GBs_164B"), pm.get("GBs_132B")]
        r["ratio_to_stream_probe"] = roof.get("frac_of_measured")
    if roof.get("steady_state"):
        r["steady_state"] = _pick(roof["steady_state"], ("steps", "kernel_us", "achieved_GBs", "ratio_to_stream_probe"))
    if roof.get("launch_mode"):
        r["launch_mode"] = _pick(roof["launch_mode"], ("kernel_us", "achieved_GBs", "ratio_to_stream_probe"))
    if roof.get("beyond_cache"):
        r["beyond_cache"] = _pick(roof["beyond_cache"], ("vehicles", "us_per_step", "frac", "resident_in", "stepping"))
    if roof.get("hbm_streaming"):
        r["hbm_streaming"] = _pick(roof["hbm_streaming"], ("vehicles", "us_per_step", "achieved", "frac", "frac_of_6290", "probe_GBs", "ratio_to_stream_probe", "pmc_over_algorithmic",
                                                           "pmc_from", "stepping", "error"))
    line["roofline"] = r
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind"))
        c["sample"] = str(cb.get("sample", ""))[:110]
        if cb.get("all_cores"):
            c["all_cores"] = _pick(cb["all_cores"], ("value", "cores", "speedup_over_one_thread"))
        if cb.get("other_noise_policy"):
            c["other_noise_policy"] = _pick(cb["other_noise_policy"], ("value", "cores", "noise_policy"))
        if cb.get("noise_policy"):
            c["noise_policy"] = cb["noise_policy"]
        c["cpu_model"] = str(cb.get("cpu_model", ""))[:60]
        line["cpu_baseline"] = c
    st = full.get("config4_as_stated")
    if st:
        c4 = _pick(st, ("scaling", "vehicles_total", "vehicles_per_gpu", "n_gpus", "value", "unit", "ms_per_step", "stepping", "bound", "resident_in", "frac", "l2_GBs", "l2_frac", "valu_busy_frac", "counters_stale"))
        if st.get("steady_state"):
            c4["steady_state"] = _pick(st["steady_state"], ("steps", "ms_per_step", "value"))
        line["config4_as_stated"] = c4
    ns = full.get("north_star_shard")
    if ns:
        line["north_star_shard"] = _pick(ns, ("vehicles_per_gpu", "us_per_step", "us_per_step_k_blocks", "vsteps_per_s_per_gpu", "bound", "resident_in", "l2_GBs", "l2_frac",
                                              "valu_busy_frac", "valu_active_frac_per_wave", "valu_instructions_per_wave_step", "counters_from", "counters_stale", "launch_mode_us_per_step"))
    rn = full.get("counter_noise_policy")
    if rn:
        line["counter_noise_policy"] = _pick(rn, ("value", "unit", "ms_per_step", "algorithmic_bytes_per_vehicle_step", "kernel_us", "achieved_GBs", "ratio_to_stream_probe", "resident_in", "stepping", "seed_policy"))
    sc = full.get("scaling_check")
    if sc:
        line["scaling_check"] = sc
    comp = full.get("companions")
    if comp:
        line["companions"] = {k: v.get("value") for k, v in comp.items() if isinstance(v, dict) and "value" in v}
    cl = full.get("closed_loop_on_device")
    if cl:
        line["closed_loop_on_device"] = [_pick(c, ("vehicles", "us_per_step", "us_per_step_auto", "vsteps_per_s", "bound", "resident_in", "frac", "l2_GBs")) for c in cl]       # (no instruction count is committed for the logic kernels: bound says latency for the L2-resident rows)
    pr = full.get("perception_rows")
    if pr and "depth_camera" in pr:
        dc, rp = pr["depth_camera"], pr["rappids_planner"]
        dcr, rpr = dc["roofline"], rp["roofline"]
        cam = {"views": dc["views"], "ms": dc["kernel_ms"], "rays_per_s": dc["rays_per_s"], "bound": "valu issue",
               "valu_issue_frac": dcr.get("valu_issue_fraction_of_busy_cycles"), "valu_per_ray": dcr.get("valu_instructions_per_ray"),
               "floor_valu_per_ray": dcr.get("floor_valu_per_ray"), "nodes_per_wave": (dcr.get("per_wave_of_64_rays") or {}).get("nodes_visited"),
               "tri_box_per_wave": (dcr.get("per_wave_of_64_rays") or {}).get("triangle_box_tests"),
               "tri_fp64_per_wave": (dcr.get("per_wave_of_64_rays") or {}).get("fp64_triangle_tests_executed"),
               "counters_from": dcr.get("counters_from")}
        pln = {"planners": rp["config3_size"]["planners"], "ms": rp["config3_size"]["kernel_ms"], "plans_per_s": rp["config3_size"]["plans_per_s"],
               "bound": "latency", "valu_issue_frac": rpr.get("valu_issue_fraction_of_busy_cycles"),
               "valu_per_plan": rpr.get("valu_instructions_per_plan"), "counters_from": rpr.get("counters_from")}
        for row in (cam, pln):
            if row["counters_from"] is None:
                row["counters_stale"] = True
        line["perception"] = {"depth_camera": {k: v for k, v in cam.items() if v is not None}, "planner": {k: v for k, v in pln.items() if v is not None}}
    for key in ("config3", "config5"):           # BASELINE configs 3 and 5 at their size, in the metric's unit
        row = full.get(key)
        if row:
            # (the SQ-counter figures of the two kernels are in `perception` just above; here the frame, in the metric's unit)
            line[key] = _pick(row, ("vehicles", "frame_ms", "physics_ms", "render_ms", "plan_ms", "vsteps_per_s", "rays_per_s", "plans_per_s", "bound_short",
                                    "model_valu_per_ray", "floor_valu_per_ray", "counters_stale", "error"))
    if full.get("counters_stale"):
        line["counters_stale"] = [str(x)[:48] for x in full["counters_stale"]][:6]
    sw = full.get("shared_world")
    if sw:
        if "error" in sw:
            line["shared_world"] = {"error": str(sw["error"])[:200]}
        else:
            line["shared_world"] = {"rccl_ranks": sw.get("rccl_ranks"),
                                    "fraction_query_in_stream": {k: w.get("fraction_query_in_stream") for k, w in sw.get("worlds", {}).items()},
                                    "vsteps_per_s_query_in_stream": {k: w.get("vsteps_per_s_query_in_stream") for k, w in sw.get("worlds", {}).items()}}
    c1 = full.get("config1_host_in_loop")
    if c1:
        line["config1_host_in_loop"] = _pick(c1, ("us_per_step", "realtime_factor", "error"))
    line["detail"] = full.get("detail", DETAIL_FILE)
    line = _r(line)
    for victim in ("config1_host_in_loop", "shared_world", "closed_loop_on_device", "companions", "counter_noise_policy",
                   "north_star_shard", "config4_as_stated", "perception", "config5", "config3"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(victim, None)
    return line


def write_detail(full):
    """the whole measurement record, full precision, next to bench.py (and under gpurun_out/ where that exists, so that a
    gpurun call brings it home); returns the name the line carries"""
    text = json.dumps(full, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(text + "\n")
            except OSError as ex:
                sys.stderr.write("bench.py: could not write %s: %s\n" % (os.path.join(d, DETAIL_FILE), ex))
    return DETAIL_FILE


_REAL_STDOUT = os.dup(1)       # the line goes here whatever fd 1 points at when it is printed
_PRINTED = __import__("threading").Lock()


def print_line_once(out):
    """exactly one JSON line on the real stdout, whoever gets here first (main thread or the watchdog): the compact
    line; the full record goes to the side file"""
    if not _PRINTED.acquire(blocking=False):
        return False
    sys.stdout.flush()
    out["detail"] = write_detail(out)
    os.write(_REAL_STDOUT, (json.dumps(compact_line(out)) + "\n").encode())
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--vehicles", type=int, default=1 << 20, help="vehicles per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-shared-world", action="store_true")
    ap.add_argument("--no-perception", action="store_true", help="skip the config-3 / config-5 rows")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed cadence (for rocprofv3 passes: no comparisons, no per-kernel breakdown, no sweep)")
    ap.add_argument("--step-mode", choices=("auto", "launch", "persistent"), default="auto")
    ap.add_argument("--noise", choices=("reference", "counter"), default=None,
                    help="the headline's IMU noise: the reference's per-vehicle libstdc++ streams (default) or the counter-based generator")
    ap.add_argument("--perception-only", action="store_true",
                    help="development: only the config-3 / config-5 rows (and the perception component rows) as one JSON object")
    ap.add_argument("--watchdog", type=int, default=240, help="seconds the shared-world part may take before the line is printed without it")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)     # before torch / the engine are imported: this process stays off the GPUs
    global HEADLINE_EXACT_STREAMS
    if args.noise is not None:
        HEADLINE_EXACT_STREAMS = args.noise == "reference"

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # test hook (tools/bench_two_ranks_one_gpu.sh): every rank on device 0 and gloo instead of RCCL, so that the N > 1
    # control flow -- shard bookkeeping, the strong-scaling row, the MAX over ranks -- can be run on a one-GPU box.
    # RCCL refuses two ranks on one device, so the shared-world part needs --no-shared-world there; no number is meant.
    one_device = os.environ.get("AFE_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if torch.cuda.device_count() < world and not one_device:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: RCCL even for one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():     # RCCL's banner must not land on stdout
            if one_device:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        # Before anything is timed: the process group IS N ranks over RCCL, one per GPU (round-4 review: RCCL had only ever
        # run with one rank; a group that came up smaller, or on another backend, must not produce a line that looks like N GPUs)
        if not one_device:
            ones = torch.ones(1, device="cuda")
            dist.all_reduce(ones)
            if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl" or int(ones.item()) != args.gpus:
                raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s) on backend %s (all-reduce of ones: %g)"
                                 % (args.gpus, dist.get_world_size(), dist.get_backend(), ones.item()))

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    afa = importlib.import_module("agri-fly_amd")
    if args.perception_only:
        rows = {"config3": config3_row(afa, local_rank), "config5": config5_row(afa, local_rank)}
        print(json.dumps(_r(rows)))
        return
    n_local = args.vehicles
    mode = {"auto": headline_mode(afa, n_local), "launch": afa.AFE_STEP_LAUNCH, "persistent": afa.AFE_STEP_PERSISTENT}[args.step_mode]
    n_global = n_local * world
    e = build_shard(afa, n_local, rank * n_local, n_global, local_rank)
    e.set_step_mode(mode)

    # ---- the headline measurement: W warmup steps, then blocks of exactly K timed steps (median block) ----
    time_steps(e, args.warmup, 1, sync, barrier)
    own_blocks = []
    e.grid_time()         # (ends a resident grid and clears its account: what follows is the timed region's own)
    blocks = timed_blocks(e, args.steps, 1, sync, barrier, reduce_max, own=own_blocks)
    # device time of the resident grid over exactly that region (settling blocks included): the begin / end timestamps of
    # its dispatch(es) on the engine's queue and the steps served -- what a rocprofv3 kernel trace of this command shows
    # for the same dispatch (tools/profile_summary_r04.py).  (0, 0) when the steps were launched kernels.
    grid_s, grid_steps = e.grid_time()
    persistent = uses_persistent(afa, mode, n_local)
    split = (not persistent) and n_local >= (1 << 19)
    elapsed = median(blocks)
    value = n_global * args.steps / elapsed

    # ---- config 4 as BASELINE.json states it: 2^20 vehicles in total, sharded over the ranks (strong scaling) ----
    strong = None
    n_strong = (1 << 20) // world
    if not args.headline_only:
        es = build_shard(afa, n_strong, rank * n_strong, n_strong * world, local_rank)
        es.set_step_mode(mode)
        time_steps(es, args.warmup, 1, sync, barrier)
        sblocks = timed_blocks(es, args.steps, 1, sync, barrier, reduce_max)
        long_steps = max(args.steps, 2000)
        slong = timed_blocks(es, long_steps, 1, sync, barrier, reduce_max, min_blocks=3)
        sbytes, _ = mean_bytes_per_step(es, afa, args.steps)
        ts, tl = median(sblocks) / args.steps, median(slong) / long_steps
        strong = {"what": "BASELINE config 4 as stated: 1,048,576 vehicles in total, %d per GPU on %d GPU(s); same workload, same timing protocol" % (n_strong, world),
                  "scaling": "strong", "vehicles_total": n_strong * world, "vehicles_per_gpu": n_strong, "n_gpus": world,
                  "stepping": "persistent" if uses_persistent(afa, mode, n_strong) else "launches",
                  "value": n_strong * world / ts, "unit": "vehicle-steps/s", "ms_per_step": ts * 1e3, "steps": args.steps, "repeats": len(sblocks),
                  "steady_state": {"steps": long_steps, "ms_per_step": tl * 1e3, "value": n_strong * world / tl}}
        strong.update(bound_fields(n_strong, sbytes, ts, committed_ns_profile()))     # per GPU: what bounds one rank's shard
        es.close()

    # ---- the same workload under the OTHER noise policy, the headline's protocol: the counter-based generator when the
    # headline runs on the reference's streams (per-vehicle std::minstd_rand0 + std::normal_distribution, bit-exact words and
    # polar-method decisions: Quadcopter_T.cpp:165-180), and the other way round with --noise counter ----
    other = None
    if not args.headline_only:
        ex = build_shard(afa, n_local, rank * n_local, n_global, local_rank, exact_stream=not HEADLINE_EXACT_STREAMS)
        ex.set_step_mode(mode)
        time_steps(ex, max(args.warmup, 50), 1, sync, barrier)
        ex.grid_time()
        xblocks = timed_blocks(ex, args.steps, 1, sync, barrier, reduce_max)
        xg_s, xg_steps = ex.grid_time()
        xbytes, _ = mean_bytes_per_step(ex, afa, args.steps)
        tx = median(xblocks) / args.steps
        tx_grid = xg_s / xg_steps if xg_steps > 0 and xg_s > 0 else None
        if tx_grid is not None and not (0.5 * tx <= tx_grid <= 1.5 * tx):
            tx_grid = None
        tx_kernel = (tx_grid if tx_grid is not None else event_blocks(ex, args.steps)[0]) if rank == 0 else None
        other = {"value": n_global / tx, "unit": "vehicle-steps/s", "ms_per_step": tx * 1e3, "steps": args.steps, "repeats": len(xblocks),
                 "algorithmic_bytes_per_vehicle_step": xbytes, "kernel_us": None if tx_kernel is None else tx_kernel * 1e6,
                 "frac": None if tx_kernel is None else n_local * xbytes / tx_kernel / 1e9 / HBM_PEAK_GBS,
                 "frac_wall": n_local * xbytes / tx / 1e9 / HBM_PEAK_GBS, "resident_in": residency(n_local * xbytes),
                 "stepping": "persistent" if uses_persistent(afa, mode, n_local) else "launches",
                 "seed_policy": "AFE_SEED_COUNTER" if HEADLINE_EXACT_STREAMS else "AFE_SEED_DECORRELATED",
                 "note": ("IMU noise from the counter-based generator (Philox4x32-10 + Box-Muller per vehicle and tick: no per-vehicle word, no rejection loop; "
                          "a stream no reference run produces)" if HEADLINE_EXACT_STREAMS else
                          "IMU noise from per-vehicle libstdc++-exact streams (seed 1 + global index)") +
                         "; same timing protocol as the headline (median of bracketed K-step blocks), frac from device time like roofline.frac"}
        ex.close()

    out = None
    if rank != 0:             # (rank 0 still takes the roofline's measurements on its engine, then lets it go as well)
        e.close()
        e = None
    if rank == 0:
        bytes_step, tick_frac = mean_bytes_per_step(e, afa, args.steps)
        # device time per step over the timed cadence: HIP events on the engine's stream around K steps, repeated
        t_block, t_kmin, t_kmax, k_rep = event_blocks(e, args.steps)
        # launch mode: HIP events around the K launches of a block.  Resident grid: one grid serves all the blocks of the
        # timed region -- its device time / the steps it served; the one-block-per-dispatch figure (a grid started and
        # parked around every block, as round 3 had to) stays in the record as kernel_us_dispatch_per_block
        # (under rocprofv3 the runtime's dispatch timestamps are the profiler's, not ours: what comes back is nonsense --
        # a figure far from the wall time per step is dropped for the event-style one)
        t_grid = grid_s / grid_steps if grid_steps > 0 and grid_s > 0 else None
        if t_grid is not None and not (0.5 * elapsed / args.steps <= t_grid <= 1.5 * elapsed / args.steps):
            t_grid, grid_steps = None, 0
        t_kernel = t_grid if t_grid is not None else t_block
        achieved = n_local * bytes_step / t_kernel / 1e9
        # the same over a long run: a resident grid's launch and exit (and a stream's first launches) amortised away
        long_steps = max(args.steps, 2000)
        t_long = median([kernel_time_events(e, long_steps) for _ in range(3)])
        # the same engine stepping by launches (one per step; per half of the shard from 2^19 vehicles up), for comparison
        t_launch = None
        if persistent and not args.headline_only:
            e.set_step_mode(afa.AFE_STEP_LAUNCH)
            time_steps(e, 100, 1, sync, lambda: None)
            t_launch = median([kernel_time_events(e, long_steps) for _ in range(3)])
            e.set_step_mode(mode)
        traffic, traffic_src, rocprof_us = committed_traffic(n_local, HEADLINE_EXACT_STREAMS)
        e_footprint = 52.0 + 16.0 + 12.0 + 24.0 + (4.0 if HEADLINE_EXACT_STREAMS else 0.0)      # distinct bytes a step touches per fp32 vehicle of this workload
        # the headline's engine is done.  (It goes before the rows of smaller shards are measured: a rank of an 8-GPU run
        # holds its own shard and nothing else, and a second large engine that has had a resident grid costs a small
        # engine's synchronised blocks ~0.5 us per step while it exists -- measured, tools/sync_cost_probe.py; cause open.)
        e.close()
        e = None
        # what the box streams in this launch shape, in this run (measured figure next to the nominal peak)
        probe = None
        if not args.headline_only:
            us164 = afa.stream_probe(n_local, 24, 17, 200, local_rank)
            us132 = afa.stream_probe(n_local, 20, 13, 200, local_rank)
            probe = {"GBs_164B": n_local * 164 / us164 / 1e3, "GBs_132B": n_local * 132 / us132 / 1e3,
                     "us_164B": us164, "us_132B": us132,
                     "what": "afe_stream_probe: back-to-back launches of a pure streaming kernel in the step kernel's shape (one-wave workgroups, "
                             "24+17 / 20+13 planar dword streams in place), same ensemble size, this run"}
        out = {
            "metric": "vehicle-steps/sec @dt=1ms",
            "value": value,
            "unit": "vehicle-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": len(blocks),
            "ms_per_step_min": min(blocks) / args.steps * 1e3,
            "ms_per_step_max": max(blocks) / args.steps * 1e3,
            "ms_per_step_rank0_own": median(own_blocks) / args.steps * 1e3,     # `ms_per_step` is the MAX over ranks, block by block
            "higher_is_better": True,
            "headline_note": "since round 5 on the reference's noise streams; rounds 1-4 quoted counter_noise_policy" if HEADLINE_EXACT_STREAMS else None,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload_short": "BASELINE config 4 per GPU: %d hovering CF_MINIQUAD vehicles, per-vehicle wind gusts (sigma 0..0.5 N, 100 ms epochs), IMU "
                                  "synthesis + noise at the 500 Hz logic gate, one afe_step call per 1 ms step, state through memory every step" % n_local,
                "noise": ("AFE_SEED_DECORRELATED: the reference's per-vehicle libstdc++ streams, bit-exact"
                          if HEADLINE_EXACT_STREAMS else
                          "AFE_SEED_COUNTER (Philox4x32-10 + Box-Muller); the reference's libstdc++ streams: reference_noise_streams"),
                "parallelism_short": "contiguous shards, %d rank(s), no data-path collective" % world,
                "workload": "config 4: hovering CF_MINIQUAD ensemble, per-vehicle wind gusts from the on-device gust process (sigma swept 0..0.5 N over "
                            "the global index, piecewise constant, resampled every 100 ms; afe_set_gust_process), IMU synthesis with Gaussian noise from " +
                            ("the reference's own machinery (one std::minstd_rand0 + std::normal_distribution stream per vehicle, seed 1 + global index: AFE_SEED_DECORRELATED)"
                             if HEADLINE_EXACT_STREAMS else "the counter-based generator (AFE_SEED_COUNTER: Philox4x32-10 + Box-Muller per vehicle and tick)") +
                            " at the 500 Hz logic gate, "
                            "one afe_step call per 1 ms step, state through HBM every step (no temporal fusion); " +
                            ("stepping by one resident grid (afe_set_step_mode: every wave advances its vehicles through each authorised step, "
                             "no kernel boundary between steps)" if persistent else
                             "one kernel launch per step" + (" and per half of the shard: the two halves step on two streams (afe_set_split_stepping)" if split else "")),
                "vehicles_per_gpu": n_local,
                "vehicles_total": n_global,
                "dt_us": DT_US,
                "logic_period_s": LOGIC_PERIOD,
                "steps_per_call": 1,
                "stepping": "persistent" if persistent else ("split launches" if split else "launches"),
                "timing": "median of %d blocks of exactly %d steps, each bracketed by barrier + synchronise (min / max in ms_per_step_min / _max)" % (len(blocks), args.steps),
                "parallelism": "ensemble sharded contiguously, %d rank(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                # where the headline's working set lives between two steps: every distinct byte a step touches (state 52 B, commands
                # 16, force 12, IMU sample 24, engine word 4 per vehicle) against the 256 MiB Infinity Cache.  Inside it, `frac` is
                # algorithmic bytes / time against the HBM peak with the Infinity Cache serving the bytes ("fits the 256 MB
                # Infinity Cache ... the reported fraction uses algorithmic bytes / time") -- the row that streams from HBM is hbm_streaming
                "resident_in": residency(n_local * e_footprint),
                "working_set_bytes": n_local * e_footprint,
                "infinity_cache_bytes": INFINITY_CACHE_BYTES,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "peak_measured": probe,
                "frac_of_measured": None if probe is None else achieved / probe["GBs_164B"],
                "traffic": traffic,
                "traffic_is": "L2-fabric bytes per step (Infinity-Cache hits counted): not HBM bytes here",
                "traffic_source": traffic_src,
                "kernel_short": ("afe_step_persistent_kernel<float,FEXT,NOISE=%s>" % ("libstdc++ streams" if HEADLINE_EXACT_STREAMS else "counter")) if persistent else "afe_step_kernel<float,FEXT,NOISE 0/1,SINGLE>",
                "kernel": (("afe::afe_step_persistent_kernel<float, FEXT=1, NOISE=%s, LOGIC=0>: one launch serves every step between two " % ("1 (libstdc++ streams)" if HEADLINE_EXACT_STREAMS else "2 (counter)")) +
                           "synchronisations; its rocprofv3 duration / the steps it served = kernel_us_rocprof, the HIP events around a block = kernel_us") if persistent else (
                           "afe::afe_step_kernel<float, FEXT=1, TEXT=0, NOISE, LOGIC=0, SINGLE=1>, 64-lane workgroups -- the timed region "
                           "alternates NOISE=0 (no logic tick) and NOISE=1 (tick: IMU + six Gaussian draws) launches"),
                "kernel_us": t_kernel * 1e6,
                "kernel_us_source": ("device timestamps of the resident grid's dispatch over the timed region / %d steps served (afe_grid_time)" % grid_steps
                                     if grid_steps > 0 and grid_s > 0 else "HIP events on the engine's stream around the launches of a block"),
                "kernel_us_dispatch_per_block": t_block * 1e6,
                "kernel_us_min": t_kmin * 1e6, "kernel_us_max": t_kmax * 1e6, "kernel_repeats": k_rep,
                # the committed rocprofv3 kernel trace of this command: the resident grid's own duration per step.  The events
                # above bracket the whole block on the stream -- dispatch of the grid, its ramp-up and the park hand-shake
                # included (about 25 us per block, i.e. 1.3 us per step in blocks of 20) -- so they read higher than the kernel
                "kernel_us_rocprof": rocprof_us,
                "steps_per_event_block": args.steps,
                "algorithmic_bytes_per_vehicle_step": bytes_step,
                "algorithmic_bytes_per_step": n_local * bytes_step,
                "imu_tick_fraction": tick_frac,
                "steady_state": {"steps": long_steps, "kernel_us": t_long * 1e6, "achieved_GBs": n_local * bytes_step / t_long / 1e9,
                                 "frac": n_local * bytes_step / t_long / 1e9 / HBM_PEAK_GBS},
                "launch_mode": None if t_launch is None else {
                    "kernel_us": t_launch * 1e6, "achieved_GBs": n_local * bytes_step / t_launch / 1e9,
                    "frac": n_local * bytes_step / t_launch / 1e9 / HBM_PEAK_GBS,
                    "note": "the same engine with afe_set_step_mode(AFE_STEP_LAUNCH): one kernel launch per step" +
                            (" and half of the shard (two streams)" if n_local >= (1 << 19) else "")},
                "per_kernel": None if args.headline_only else per_kernel_breakdown(afa, n_local, local_rank),
                "note": "achieved = algorithmic bytes per step / device time per step over the timed cadence (kernel_us).  The distinct bytes of a step "
                        "(%.0f MB) fit the 256 MiB Infinity Cache: resident_in says so, and traffic is fabric-side bytes, not HBM bytes.  beyond_cache: 2^22 "
                        "vehicles (the state alone still fits: cache-policy hints keep it on-die); hbm_streaming: 2^23 vehicles, nothing survives a step on-die -- "
                        "the row whose fraction is a fraction of HBM bandwidth" % (n_local * e_footprint / 1e6),
            },
            "config4_as_stated": strong,
            "counter_noise_policy" if HEADLINE_EXACT_STREAMS else "reference_noise_streams": other,
        }
        if world > 1:
            exp, exp_src = committed_json("expected_rates.json", lambda d: d)
            sc = scaling_check(exp, exp_src, world, n_local, value, n_strong, strong["value"] if strong else None)
            if sc:
                out["scaling_check"] = sc
        # rows whose bytes are served by the Infinity Cache: their ceiling is what a pure streaming kernel of the step kernel's
        # shape reaches at the same size and residency in this run (afe_stream_probe), not the HBM peak -- frac_of_probe
        if probe is not None:
            pg = probe["GBs_164B"]
            for row in (out["roofline"]["steady_state"], out["roofline"]["launch_mode"], other):
                if row and row.get("achieved_GBs"):
                    row["ratio_to_stream_probe"] = row["achieved_GBs"] / pg
                elif row and row.get("frac") is not None:
                    row["achieved_GBs"] = row["frac"] * HBM_PEAK_GBS
                    row["ratio_to_stream_probe"] = row["achieved_GBs"] / pg
        if world == 1 and not args.no_sweep and not args.headline_only:
            # beyond the Infinity Cache: 2^22 vehicles (620 MB per step)
            nb = 4 << 20
            eb, rowb = shard_row(afa, nb, local_rank, sync, barrier, reduce_max, 200)
            eb.sync()
            usb = afa.stream_probe(nb, 24, 17, 40, local_rank)
            eb.close()
            rowb["achieved_GBs"] = rowb["algorithmic_GBs"]
            rowb["peak_measured_GBs_164B"] = nb * 164 / usb / 1e3
            rowb["frac_of_measured"] = rowb["achieved_GBs"] / rowb["peak_measured_GBs_164B"]
            rowb["frac_of_6290"] = rowb["achieved_GBs"] / 6290.0      # the guide's achievable-from-HBM figure
            rowb["resident_in"] = "state in infinity_cache (cache policy 1), inputs and outputs from hbm"
            out["roofline"]["beyond_cache"] = rowb
            # THE HBM row: 2^23 vehicles.  The state alone is 436 MB: nothing a step touches is still on-die when the next
            # step comes for it, every algorithmic byte crosses the HBM interface, and the fabric-side counters ARE HBM
            # bytes here (profiles/r05_bc23_summary.json: PMC / algorithmic).  frac = algorithmic bytes / time / 8 TB/s;
            # frac_of_probe = against what afe_stream_probe streams in the same launch shape at the same size in this run.
            try:
                nh = 8 << 20
                eh, rowh = shard_row(afa, nh, local_rank, sync, barrier, reduce_max, 100)
                eh.sync()
                ush = afa.stream_probe(nh, 24, 17, 20, local_rank)
                eh.close()
                pmc, pmc_src = committed_json("r*_bc23_summary.json", lambda d: d["per_step"]["pmc_over_algorithmic"], PROV.STEP_KERNEL)
                out["roofline"]["hbm_streaming"] = {
                    "vehicles": nh, "us_per_step": rowh["us_per_step"], "vsteps_per_s": rowh["vsteps_per_s"], "achieved": rowh["algorithmic_GBs"], "unit": "GB/s",
                    "frac": rowh["algorithmic_GBs"] / HBM_PEAK_GBS, "probe_GBs": nh * 164 / ush / 1e3, "ratio_to_stream_probe": rowh["algorithmic_GBs"] / (nh * 164 / ush / 1e3),
                    "frac_of_6290": rowh["algorithmic_GBs"] / 6290.0, "pmc_over_algorithmic": pmc, "pmc_from": pmc_src,
                    "working_set_bytes": nh * e_footprint, "resident_in": "hbm", "stepping": rowh["stepping"], "block_steps": rowh["block_steps"], "repeats": rowh["repeats"],
                    "note": "launched kernels, two halves on two streams, cache policy by size (afe_set_cache_policy automatic); bounded by the device's "
                            "write path: read-only streams reach 6.6-6.8 TB/s on these boxes, write-only 4.4-4.9 (profiles/r04_hbm_probe_2p23.txt).  probe_GBs is "
                            "afe_stream_probe at the same size in this run -- a PLAIN streaming kernel (default cache policy, one stream), not a ceiling: the step "
                            "kernels' nt hints and two streams can beat it (ratio_to_stream_probe > 1); frac_of_6290 is against the guide's achievable HBM figure"}
            except Exception as ex:            # (a box short of memory must not cost the line)
                out["roofline"]["hbm_streaming"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            # the north-star shard: one GPU's share of config 4 on eight (131,072 vehicles), and its neighbours
            sweep = []
            for n in (1024, 4096, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20):
                es, row = shard_row(afa, n, local_rank, sync, barrier, reduce_max, 400)
                bytes_n, _ = mean_bytes_per_step(es, afa, 400)
                k = 400
                tr = None
                if n <= (1 << 20):
                    es.set_step_mode(afa.AFE_STEP_AUTO)
                    tr = median(timed_blocks(es, k, 1, sync, barrier, reduce_max, min_total_s=0.02, settle_s=0.005))
                    row["auto_resident_state"] = {"us_per_step": tr / k * 1e6, "vsteps_per_s": n * k / tr}
                es.set_step_mode(afa.AFE_STEP_LAUNCH)
                es.set_split_stepping(1)
                time_steps(es, 50, 1, sync, barrier)
                t1 = median([time_steps(es, k, 1, sync, barrier) for _ in range(3)])
                es.set_split_stepping(2)
                time_steps(es, 50, 1, sync, barrier)
                t2 = median([time_steps(es, k, 1, sync, barrier) for _ in range(3)])
                es.set_split_stepping(1)
                tf = median([time_steps(es, k, 2, sync, barrier) for _ in range(3)])
                t50 = median([time_steps(es, k, 50, sync, barrier) for _ in range(3)])
                fr = lambda t: n * bytes_n / (t / k) / 1e9
                row.update({"launches": {"us_per_step": t1 / k * 1e6, "vsteps_per_s": n * k / t1, "algorithmic_GBs": fr(t1)},
                            "split_launches": {"us_per_step": t2 / k * 1e6, "vsteps_per_s": n * k / t2, "algorithmic_GBs": fr(t2)},
                            "fused2": {"us_per_step": tf / k * 1e6, "vsteps_per_s": n * k / tf},
                            "fused50": {"us_per_step": t50 / k * 1e6, "vsteps_per_s": n * k / t50}})
                sweep.append(row)
                es.close()
            out["sweep"] = sweep
            out["sweep_note"] = ("per size: the headline's stepping (top level: us_per_step, vsteps_per_s, algorithmic_GBs; bound / resident_in say what the rate "
                                 "is a fraction OF -- the L2s' 34.5 TB/s for working sets that fit them (l2_frac; those rows are latency-bound), 8 TB/s beyond (frac); "
                                 "median of bracketed 400-step blocks), then the same shard stepped by launches on one stream, by launches on two streams "
                                 "(afe_set_split_stepping 2), with two steps per launch (fused2: nothing is observable between 500 Hz logic ticks) and with "
                                 "50 steps per launch (fused50: state in registers, open-loop commands).  131,072 vehicles is one GPU's shard of BASELINE "
                                 "config 4 on 8 GPUs")
            ns = [r for r in sweep if r["vehicles"] == 131072][0]
            bytes_ns = ns["algorithmic_GBs"] * 1e9 * ns["us_per_step"] * 1e-6 / 131072
            ek, rowk = shard_row(afa, 131072, local_rank, sync, barrier, reduce_max, args.steps)
            ek.close()
            # what ONE GPU delivers on the shard a rank of an N-GPU run of config 4 holds, in the driver's own protocol (blocks of
            # K steps): the figures a first multi-GPU run is judged against (tools/expected_rates.py -> profiles/expected_rates.json)
            shard_rates = {"131072": {"us_per_step_k_blocks": rowk["us_per_step"], "vsteps_per_s": rowk["vsteps_per_s"]}}
            for n_sh in (262144, 524288):
                esh, rsh = shard_row(afa, n_sh, local_rank, sync, barrier, reduce_max, args.steps)
                esh.close()
                shard_rates[str(n_sh)] = {"us_per_step_k_blocks": rsh["us_per_step"], "vsteps_per_s": rsh["vsteps_per_s"]}
            out["strong_shard_rates_one_gpu"] = {"k": args.steps, "rows": shard_rates}
            ns_row = {"vehicles_per_gpu": 131072, "us_per_step": ns["us_per_step"], "vsteps_per_s_per_gpu": ns["vsteps_per_s"],
                      "us_per_step_k_blocks": rowk["us_per_step"], "k": args.steps, "launch_mode_us_per_step": ns["launches"]["us_per_step"],
                      "note": "one GPU's shard of the 1M-vehicle ensemble on 8 GPUs, measured on this one GPU; shards do not communicate while stepping "
                              "(no data-path collective), so 8 ranks deliver 8x this rate up to barrier skew -- a projection until the driver's --gpus 8 run.  "
                              "No HBM fraction: the shard's 19 MB live in the XCDs' L2s (1.6 B per vehicle-step reach the fabric: profiles/r05_ns_summary.json); "
                              "two worker waves share every SIMD and the step is bound by vector-instruction issue: valu_busy_frac = the share of a SIMD's cycles in "
                              "which a vector instruction of one of its two waves executes (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per wave x 2, committed counters of "
                              "this grid, idle polls between blocks inside); l2_frac = algorithmic bytes against the L2s' 34.5 TB/s"}
            ns_row.update(bound_fields(131072, bytes_ns, ns["us_per_step"] * 1e-6, committed_ns_profile()))
            out["north_star_shard"] = ns_row
            # config 2 closed on the GPU: on-device onboard rates logic, hover command
            closed = []
            for n in (4096, 131072, 1 << 20):
                es = build_shard(afa, n, 0, n, local_rank)
                es.set_rates_logic([afa.rates_logic_params_from_type(5)])
                es.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
                k = 400
                t1 = median(timed_blocks(es, k, 1, sync, barrier, reduce_max, min_total_s=0.02)) 
                es.set_step_mode(afa.AFE_STEP_AUTO)          # one step per call: the resident-state form (state not read back between steps authorised ahead)
                ta = median(timed_blocks(es, k, 1, sync, barrier, reduce_max, min_total_s=0.02, settle_s=0.005))
                es.set_step_mode(afa.AFE_STEP_LAUNCH)
                t10 = median([time_steps(es, k, 10, sync, barrier) for _ in range(3)])
                b_mean, _ = mean_bytes_per_step(es, afa, k)   # state, force, commands; on ticks IMU, filter state, rate commands
                crow = {"vehicles": n, "vsteps_per_s": n * k / t1, "us_per_step": t1 / k * 1e6,
                        "algorithmic_bytes_per_vehicle_step": b_mean, "achieved_GBs": n * b_mean / (t1 / k) / 1e9,
                        "vsteps_per_s_fused10": n * k / t10, "us_per_step_auto": ta / k * 1e6, "vsteps_per_s_auto": n * k / ta}
                crow.update(bound_fields(n, b_mean, t1 / k, None))
                closed.append(crow)
                es.close()
            out["closed_loop_on_device"] = closed
            # PCIe-inclusive (never `value`): a host that hands the state over and takes it back around EVERY step -- what the
            # boundary costs when the inputs are not resident (the engine's design is that they are)
            ep = build_shard(afa, n_local, 0, n_local, local_rank)
            ep.set_step_mode(afa.AFE_STEP_LAUNCH)
            stp = ep.get_state(dtype=np.float32)
            tp = []
            for _ in range(4):
                t0 = time.perf_counter()
                ep.set_state(stp["pos"], stp["vel"], stp["att"], stp["ang_vel"], None, dtype=np.float32)
                ep.step(DT_US, 1)
                stp = ep.get_state(dtype=np.float32)
                tp.append(time.perf_counter() - t0)
            ep.close()
            out["pcie_inclusive"] = {"vehicles": n_local, "ms_per_step": min(tp) * 1e3, "vsteps_per_s": n_local / min(tp),
                                     "bytes_over_the_bus_per_step": n_local * (13 * 4 + 17 * 4),
                                     "note": "afe_set_state_f32 (pos, vel, att, ang_vel) + one step + afe_get_state_f32 (with rotor speeds) per step, "
                                             "pageable numpy buffers through the ctypes host; not the headline: inputs are resident there"}
            out["disturbance_sweep"] = disturbance_sweep(afa, local_rank)
            out["companions"] = companion_rows(afa, n_local, local_rank, sync, barrier, split)
            out["perception_rows"] = perception_rows(afa)
            out["config1_host_in_loop"] = config1_row()
        if world == 1 and not args.no_perception and not args.headline_only:
            # BASELINE configs 3 and 5 at their size, in the metric's unit: for these two the perception kernels ARE the hot path
            # (round-5 review item 1).  The SQ-counter figures ride along only while the committed summaries are of this tree's kernels.
            pr = out.get("perception_rows") or {}
            rr = (pr.get("depth_camera") or {}).get("roofline") or {}
            pl = (pr.get("rappids_planner") or {}).get("roofline") or {}
            for key, fn in (("config3", config3_row), ("config5", config5_row)):
                try:
                    row = fn(afa, local_rank)
                    row["valu_issue_frac"] = rr.get("valu_issue_fraction_of_busy_cycles")      # the camera bounds both frames
                    row["valu_per_ray"] = rr.get("valu_instructions_per_ray")
                    if key == "config5":
                        row["floor_valu_per_ray"] = row["per_ray"]["floor"]["floor_valu_per_ray"]
                        row["model_valu_per_ray"] = row["per_ray"]["floor"]["valu_per_ray_by_static_costs"]
                    if key == "config3":
                        row["valu_per_plan"] = pl.get("valu_instructions_per_plan")
                    if rr.get("counters_from") is None or (key == "config3" and pl.get("counters_from") is None):
                        row["counters_stale"] = True
                    row["bound_short"] = "depth camera: valu issue" if key == "config5" else "depth camera: valu issue; planner: latency"
                except Exception as ex:            # (a box short of memory must not cost the line)
                    row = {"error": "%s: %s" % (type(ex).__name__, ex)}
                out[key] = row
        if not args.no_cpu_baseline and not args.headline_only:
            # (N > 1: rank 0 alone, after its timed region -- the other ranks go on to the shared-world part and meet rank 0
            # there; a shorter sample and no all-cores leg, the cores are busy with the other ranks' processes)
            out["cpu_baseline"] = cpu_baseline(afa) if world == 1 else cpu_baseline(afa, budget_vehicle_steps=6_000_000, all_cores=False)
        if STALE:
            out["counters_stale"] = list(STALE)
    # The headline is measured.  What follows (the shared-world exchange: a second communicator, collectives at
    # query cadence) must never cost the line already in hand: if it has not finished within the watchdog's
    # time -- a rank stuck in a collective, whatever the cause -- rank 0 prints the line with the failure
    # recorded and every rank leaves with a non-zero status.
    import threading

    def bail():
        if rank == 0:
            out["shared_world"] = {"error": "did not finish within %d s; headline and roofline above are unaffected" % args.watchdog}
            print_line_once(out)
        os._exit(3)

    dog = threading.Timer(args.watchdog, bail)
    dog.daemon = True
    dog.start()
    sw_failed = False
    if not args.no_shared_world and not args.headline_only:
        try:
            sw = shared_world(afa, n_local, rank, world, local_rank, dist, torch, sync, barrier)   # collective: every rank
        except Exception as ex:                      # the other ranks may be waiting for this one: the watchdog frees them
            sw = {"error": "%s: %s" % (type(ex).__name__, ex)}
            sw_failed = True
            sys.stderr.write("bench.py rank %d: shared_world failed: %s\n" % (rank, sw["error"]))
        if rank == 0:
            out["shared_world"] = sw
    if e is not None:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    dog.cancel()
    if rank == 0:
        print_line_once(out)
    if sw_failed:
        sys.exit(4)


if __name__ == "__main__":
    main()
