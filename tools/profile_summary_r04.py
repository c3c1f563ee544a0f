#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_r04.sh <tag> [vehicles] (under gpurun_out/) into the committed summaries:
    python tools/profile_summary_r04.py <tag> "<build note>" [vehicles] [bytes per vehicle-step]
  profiles/<tag>_k20_kernel_stats.csv    rocprofv3 --stats of `bench.py --headline-only --steps 20 --warmup 5` (the driver's arguments)
  profiles/<tag>_kernel_stats.csv        the same with blocks of 2000 steps
  profiles/<tag>_full_kernel_stats.csv   the whole default bench
  profiles/<tag>_summary.json            per pass: every dispatch of the resident grid in the kernel trace beside the steps the
                                         engine says that grid served (AFE_GRID_LOG, dispatch order) -> device us per step;
                                         FETCH_SIZE / WRITE_SIZE per step (separate passes; FETCH_SIZE x 2 per the gfx950 note
                                         of MI355X_MICROARCH.md, KiB units); SQ counters per wave and step
  profiles/traffic.json                  PMC bytes per step + rocprof us per step of the bench workload (read by bench.py) -- 2^20 only
A resident grid now survives afe_sync: one dispatch serves every block of steps until something parks it (an entry point
that needs the stream, or 200 us without a new step), so a trace row's duration is meaningful only per step served."""
import sys as _sys, os as _os
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _provenance import STEP_KERNEL, PLANNER_KERNEL, RENDER_KERNEL, kernel_source_hashes
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_factors import factors
FETCH_FACTOR, WRITE_FACTOR, FACTOR_SOURCE = factors()
tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
BYTES_MEAN = float(sys.argv[4]) if len(sys.argv) > 4 else 144.0      # algorithmic B per vehicle-step: 132 off tick, 156 on tick, every 2nd step ticks
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")
KERNEL = "afe_step_persistent_kernel"


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    g.sort(key=os.path.getsize, reverse=True)      # (a pass that starts child processes leaves one set of files per process: the bench's own is the largest)
    return g[0] if g else None


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def rows(dirname, suffix):
    f = one(dirname + "/**/*_" + suffix + ".csv")
    return [r for r in csv.DictReader(open(f))] if f else []


def dispatches(dirname):
    """(start, duration ns) of the resident grid's dispatches in a pass's kernel trace, in start order"""
    r = [x for x in rows(dirname, "kernel_trace") if KERNEL in x["Kernel_Name"]]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return [(int(x["Start_Timestamp"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) for x in r]


def gridlog(name):
    f = os.path.join(out, name)
    if not os.path.exists(f):
        return []
    return [tuple(int(v) for v in line.split(",")) for line in open(f) if line.strip()]      # vehicles, workers, steps, engine's own ns


def paired(dirname, logname):
    d, g = dispatches(dirname), gridlog(logname)
    if len(d) == len(g) + 1:
        d = d[1:]         # the process's first dispatch is the primer on its throwaway queue (one wave, no steps, not in the engine's log)
    if not d or len(d) != len(g):
        return None, len(d), len(g)
    return [(dur, steps) for (_, dur), (_, _, steps, _) in zip(d, g)], len(d), len(g)


def counter_per_dispatch(dirname):
    agg = collections.defaultdict(list)
    for r in rows(dirname, "counter_collection"):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in agg.items()}


summary = {"build": note, "tag": tag, "script": "tools/profile_r04.sh %s %d" % (tag, N), "vehicles": N,
           "noise_policy": sys.argv[5] if len(sys.argv) > 5 else "counter",
           "kernel": "afe::afe_step_persistent_kernel<float, FEXT=1, NOISE (by policy), LOGIC=0, RESIDENT=0> on the engine's own AQL queue; "
                     "a dispatch serves every step authorised until it parks (AFE_GRID_LOG gives the steps per dispatch)",
           "algorithmic_bytes_per_vehicle_step": BYTES_MEAN}
for src, dst in (("prof_%s" % tag, "%s_kernel_stats.csv" % tag), ("prof_k20_%s" % tag, "%s_k20_kernel_stats.csv" % tag),
                 ("prof_full_%s" % tag, "%s_full_kernel_stats.csv" % tag)):
    f = one(src + "/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(prof, dst))
    d = one(src + "/**/*_domain_stats.csv")
    if d and src == "prof_" + tag:
        shutil.copy(d, os.path.join(prof, dst.replace("kernel_stats", "domain_stats")))

for name, src, log, block in (("blocks_of_20_steps", "prof_k20_%s" % tag, "gridlog_k20_%s.csv" % tag, 20),
                              ("blocks_of_2000_steps", "prof_%s" % tag, "gridlog_%s.csv" % tag, 2000)):
    p, nd, ng = paired(src, log)
    if p is None:
        summary[name] = {"error": "%d dispatches in the trace, %d grids in the engine's log" % (nd, ng)}
        continue
    served = [(dur, s) for dur, s in p if s > 0]
    # the grids that lived through the timed blocks (where workers are not kept together by issue priority a grid is retired after 512 steps): in dispatch order they come before the
    # first grid that served exactly one block (bench.py's event-style measurement follows its timed region)
    first_one = next((i for i, (_, s) in enumerate(served) if s == block), len(served))
    long_ = [(dur, s) for dur, s in served[:first_one] if s >= 256 or s > 4 * block]
    one_block = [dur / s for dur, s in served if s == block]            # grids parked after exactly one block (the event-style measurement)
    tot_ns, tot_steps = sum(d for d, _ in served), sum(s for _, s in served)
    rec = {"dispatches": nd, "steps_served": tot_steps, "device_ms": tot_ns / 1e6, "us_per_step_all_dispatches": tot_ns / tot_steps / 1e3}
    if long_:
        dur, s = sum(d for d, _ in long_), sum(st for _, st in long_)
        rec["resident_through_the_timed_blocks"] = {"dispatches": len(long_), "steps": s, "device_ms": dur / 1e6, "us_per_step": dur / s / 1e3,
                                                    "algorithmic_GBs": N * BYTES_MEAN / (dur / s), "frac_of_8TBs": N * BYTES_MEAN / (dur / s) / 8000.0,
                                                    "note": "the grids that lived through several blocks (2^20 vehicles: a grid is retired after 512 steps), each from its first wave to its park; "
                                                            "the host's pauses between blocks (barrier + synchronise + the clock) are inside"}
    if one_block:
        t = median(one_block)
        rec["one_block_per_dispatch"] = {"dispatches": len(one_block), "us_per_step": t / 1e3, "algorithmic_GBs": N * BYTES_MEAN / t,
                                         "frac_of_8TBs": N * BYTES_MEAN / t / 8000.0,
                                         "note": "grids that served exactly one block: dispatch, ramp-up and park inside"}
    summary[name] = rec

steps_of = {}
for cname, src, log in (("FETCH_SIZE", "pmc_fetch_%s" % tag, "gridlog_fetch_%s.csv" % tag), ("WRITE_SIZE", "pmc_write_%s" % tag, "gridlog_write_%s.csv" % tag)):
    c, g = counter_per_dispatch(src).get(cname), gridlog(log)
    if c and len(c) == len(g) + 1:
        c = c[1:]         # the primer
    if c and len(c) == len(g):
        tot, st = sum(v for v, (_, _, s, _) in zip(c, g) if s > 0), sum(s for (_, _, s, _) in g if s > 0)
        steps_of[cname] = (tot, st)
if len(steps_of) == 2:
    fs, fst = steps_of["FETCH_SIZE"]
    ws, wst = steps_of["WRITE_SIZE"]
    per_step = FETCH_FACTOR * 1024 * fs / fst + WRITE_FACTOR * 1024 * ws / wst
    summary["traffic"] = {"FETCH_SIZE_KiB_per_step": fs / fst, "WRITE_SIZE_KiB_per_step": ws / wst, "steps_counted": [fst, wst],
                          "fabric_bytes_per_step": per_step, "fabric_bytes_per_vehicle_step": per_step / N, "counter_factors": {"FETCH_SIZE": FETCH_FACTOR, "WRITE_SIZE": WRITE_FACTOR, "source": FACTOR_SOURCE},
                          "what": "bytes through the L2s' fabric side (TCC_EA0 requests): Infinity-Cache hits are counted, so this is L2 <-> Infinity Cache / HBM traffic, not HBM traffic",
                          "algorithmic_bytes_per_step": N * BYTES_MEAN, "ratio": per_step / (N * BYTES_MEAN)}
    if N == 1 << 20 and False:      # (round 5: the 2^20 record comes from tools/profile_summary_r03.py -- the grid is launched per block there)
        k20 = summary.get("blocks_of_20_steps", {}).get("resident_through_the_timed_blocks", {}).get("us_per_step")
        k2000 = summary.get("blocks_of_2000_steps", {}).get("resident_through_the_timed_blocks", {}).get("us_per_step")
        json.dump({"workload": {"vehicles_per_gpu": N, "fext": True, "noise": True, "dt_us": 1000, "logic_period_s": 0.002},
                   "traffic_bytes_per_step": per_step, "traffic_bytes_per_launch": per_step,
                   "rocprof_kernel_us_per_step": {"blocks_of_20_steps": k20, "blocks_of_2000_steps": k2000,
                                                  "source": "profiles/%s_summary.json (rocprofv3 --kernel-trace: the resident grid's dispatch that lived through the timed "
                                                            "blocks, duration / steps served)" % tag},
                   "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summed over the resident grid's dispatches / steps served, separate passes, "
                             "FETCH_SIZE / WRITE_SIZE scaled by the factors measured on known dword streams: profiles/*_fetch_calibration.json)" % tag}, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
sq, g = counter_per_dispatch("pmc_sq_%s" % tag), gridlog("gridlog_sq_%s.csv" % tag)
if sq.get("SQ_WAVES") and len(sq["SQ_WAVES"]) == len(g) + 1:
    sq = {k: v[1:] for k, v in sq.items()}     # the primer
if sq.get("SQ_WAVES") and len(sq["SQ_WAVES"]) == len(g):
    st = sum(s for (_, _, s, _) in g if s > 0)
    rec = {c: sum(v) for c, v in sorted(sq.items())}
    waves = rec["SQ_WAVES"] / max(1, len([1 for (_, _, s, _) in g if s > 0]))
    summary["sq_totals"] = rec
    summary["valu_instructions_per_wave_and_step"] = rec.get("SQ_INSTS_VALU", 0) / waves / st
    summary["salu_instructions_per_wave_and_step"] = rec.get("SQ_INSTS_SALU", 0) / waves / st
    wc = max(1.0, rec.get("SQ_WAVE_CYCLES", 0))
    # what bounds a grid whose state lives in the XCDs' L2s (round-4 review): share of the waves' resident cycles in which
    # a vector instruction is executing, and in which the wave waits for anything / for an instruction's operands
    summary["valu_active_frac_of_wave_cycles"] = rec.get("SQ_ACTIVE_INST_VALU", 0) / wc
    summary["wait_any_frac_of_wave_cycles"] = rec.get("SQ_WAIT_ANY", 0) / wc
    summary["wait_inst_any_frac_of_wave_cycles"] = rec.get("SQ_WAIT_INST_ANY", 0) / wc
    # a worker wave shares its SIMD with (workers / 1024 - 1) others: the SIMD's vector pipe is busy that many times the per-wave share
    workers_per_simd = max(1.0, (waves - 1) / 1024.0)
    summary["worker_waves_per_simd"] = workers_per_simd
    summary["simd_valu_busy_frac"] = summary["valu_active_frac_of_wave_cycles"] * workers_per_simd
    summary["note_sq"] = "a wave of the resident grid steps ceil(chunks / waves) chunks of 64 vehicles per step; the pump wave is one of SQ_WAVES; idle polling between blocks is inside"
summary["kernel_sources"] = kernel_source_hashes(STEP_KERNEL)       # bench.py borrows from this file only while these match the tree
json.dump(summary, open(os.path.join(prof, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1)[:5000])
