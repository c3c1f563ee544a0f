#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_r03.sh <tag> [vehicles] (under gpurun_out/) into the committed summaries:
    python tools/profile_summary_r03.py <tag> "<build note>" [vehicles]
  profiles/<tag>_kernel_stats.csv / _domain_stats.csv   rocprofv3 --stats of the headline cadence, blocks of 2000 steps
  profiles/<tag>_k20_kernel_stats.csv                   the same with the driver's arguments (blocks of 20 steps)
  profiles/<tag>_full_kernel_stats.csv                  the whole default bench
  profiles/<tag>_summary.json   the resident grid per step: duration / steps, algorithmic bytes, FETCH_SIZE / WRITE_SIZE
                                (separate passes; FETCH_SIZE x 2 per the gfx950 note of MI355X_MICROARCH.md, KiB units),
                                SQ counters per wave and step
  profiles/traffic.json         PMC HBM bytes per step of the bench workload (read by bench.py) -- only for 2^20 vehicles"""
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
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")
BYTES_MEAN = float(sys.argv[4]) if len(sys.argv) > 4 else 144.0      # algorithmic B per vehicle-step of the bench workload: 132 off tick, 156 on tick (counter noise: no engine word), every 2nd step ticks


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    g.sort(key=os.path.getsize, reverse=True)      # (a pass that starts child processes leaves one set of files per process: the bench's own is the largest)
    return g[0] if g else None


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def persistent_rows(dirname, suffix):
    f = one(dirname + "/**/*_" + suffix + ".csv")
    return [r for r in csv.DictReader(open(f))] if f else []


def launches(dirname):
    """durations (ns) of the resident grid's launches in a pass's kernel trace, in start order"""
    rows = [r for r in persistent_rows(dirname, "kernel_trace") if "afe_step_persistent_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]


def counter_per_launch(dirname):
    agg = collections.defaultdict(list)
    for r in persistent_rows(dirname, "counter_collection"):
        if "afe_step_persistent_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


summary = {"build": note, "tag": tag, "script": "tools/profile_r03.sh %s %d" % (tag, N), "vehicles": N, "noise_policy": sys.argv[6] if len(sys.argv) > 6 else "counter",
           "algorithmic_bytes_per_vehicle_step": BYTES_MEAN,
           "kernel": "afe::afe_step_persistent_kernel<float, FEXT=1, NOISE=%s, LOGIC=0, RESIDENT=0> (one launch per synchronised block of steps)"
                     % ("1 (libstdc++ streams)" if (len(sys.argv) > 6 and sys.argv[6] == "reference_streams") else "2 (counter)")}
for src, dst in (("prof_%s" % tag, "%s_kernel_stats.csv" % tag), ("prof_k20_%s" % tag, "%s_k20_kernel_stats.csv" % tag),
                 ("prof_full_%s" % tag, "%s_full_kernel_stats.csv" % tag)):
    f = one(src + "/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(prof, dst))
    d = one(src + "/**/*_domain_stats.csv")
    if d and src == "prof_" + tag:
        shutil.copy(d, os.path.join(prof, dst.replace("kernel_stats", "domain_stats")))

# (round 4: a grid is retired after 512 steps, so a 2000-step block is launches of 512, 512, 512 and 464 steps: the median launch serves 512)
LONG_STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
POLICY = sys.argv[6] if len(sys.argv) > 6 else "counter"          # the bench's noise policy in these passes: "reference_streams" | "counter"
for name, src, steps in (("blocks_of_2000_steps", "prof_%s" % tag, LONG_STEPS), ("blocks_of_20_steps", "prof_k20_%s" % tag, 20)):
    d = launches(src)
    if not d:
        continue
    full = [x for x in d if x > 0.5 * median(d)]          # the warm-up launch serves fewer steps than the blocks
    t = median(full) / steps
    summary[name] = {"launches": len(d), "median_launch_us": median(full) / 1e3, "us_per_step": t / 1e3,
                     "algorithmic_GBs": N * BYTES_MEAN / t, "frac_of_8TBs": N * BYTES_MEAN / t / 8000.0}
steps = 200
fetch, write, sq = counter_per_launch("pmc_fetch_%s" % tag), counter_per_launch("pmc_write_%s" % tag), counter_per_launch("pmc_sq_%s" % tag)
if fetch.get("FETCH_SIZE") and write.get("WRITE_SIZE"):
    fs, ws = median(fetch["FETCH_SIZE"]), median(write["WRITE_SIZE"])      # the 200-step blocks outnumber the warm-up launch
    per_step = (FETCH_FACTOR * 1024 * fs + WRITE_FACTOR * 1024 * ws) / steps
    summary["traffic"] = {"FETCH_SIZE_KiB_per_launch": fs, "WRITE_SIZE_KiB_per_launch": ws, "steps_per_launch": steps,
                          "fabric_bytes_per_step": per_step, "fabric_bytes_per_vehicle_step": per_step / N, "counter_factors": {"FETCH_SIZE": FETCH_FACTOR, "WRITE_SIZE": WRITE_FACTOR, "source": FACTOR_SOURCE},
                          "what": "bytes through the L2s' fabric side (TCC_EA0 requests): Infinity-Cache hits are counted, so this is L2 <-> Infinity Cache / HBM traffic, not HBM traffic",
                          "algorithmic_bytes_per_step": N * BYTES_MEAN, "ratio": per_step / (N * BYTES_MEAN)}
    if N == 1 << 20 and (len(sys.argv) <= 7 or sys.argv[7] != "no-traffic-json"):
        json.dump({"workload": {"vehicles_per_gpu": N, "fext": True, "noise": True, "noise_policy": POLICY, "dt_us": 1000, "logic_period_s": 0.002},
                   "kernel_sources": kernel_source_hashes(STEP_KERNEL),
                   "traffic_bytes_per_step": per_step,
                   "traffic_bytes_per_launch": per_step,
                   "rocprof_kernel_us_per_step": {"blocks_of_20_steps": summary.get("blocks_of_20_steps", {}).get("us_per_step"),
                                                  "blocks_of_2000_steps": summary.get("blocks_of_2000_steps", {}).get("us_per_step"),
                                                  "source": "profiles/%s_k20_kernel_stats.csv, profiles/%s_kernel_stats.csv (rocprofv3 --kernel-trace --stats; "
                                                            "median duration of the resident grid's launches / the steps each served)" % (tag, tag)},
                   "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over resident-grid launches of %d steps, separate passes, "
                             "FETCH_SIZE / WRITE_SIZE scaled by the factors measured on known dword streams: profiles/*_fetch_calibration.json)" % (tag, steps)}, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
if sq.get("SQ_WAVES"):
    rec = {c: median(v) for c, v in sorted(sq.items())}
    waves = rec["SQ_WAVES"]
    summary["sq_per_launch_of_200_steps"] = rec
    summary["valu_instructions_per_wave_and_step"] = rec.get("SQ_INSTS_VALU", 0) / waves / steps
    summary["salu_instructions_per_wave_and_step"] = rec.get("SQ_INSTS_SALU", 0) / waves / steps
    summary["note_sq"] = "a wave of the resident grid steps ceil(chunks / waves) chunks of 64 vehicles per step; the pump wave is one of SQ_WAVES"
foot = N * (52.0 + 16.0 + 12.0 + 24.0 + (4.0 if POLICY == "reference_streams" else 0.0))
summary["working_set_bytes"] = foot
summary["resident_in"] = "l2" if foot <= (32 << 20) else ("infinity_cache" if foot <= 0.94 * (256 << 20) else "hbm")
summary["note_frac"] = ("frac_of_8TBs is algorithmic bytes / kernel time against the HBM peak; with the working set resident in the Infinity Cache the bytes are served "
                        "on-die and the figure can exceed 1 -- the HBM row is profiles/r05_bc23_summary.json (2^23 vehicles)")
summary["kernel_sources"] = kernel_source_hashes(STEP_KERNEL)       # bench.py borrows from this file only while these match the tree
json.dump(summary, open(os.path.join(prof, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1)[:4000])
