#!/usr/bin/env python3
"""rocprofv3 output of tools/profile_r04_bc.sh <tag> [vehicles] (gpurun_out/) -> profiles/<tag>_kernel_stats.csv and
profiles/<tag>_summary.json: the HBM-proper regime (2^22 vehicles and beyond: launched kernels, two halves on two streams,
cache-policy hints by size).  Per step kernel: launches, average duration, FETCH_SIZE x 2 and WRITE_SIZE (KiB -> bytes, the
gfx950 note of MI355X_MICROARCH.md) per launch and per vehicle; per step (an off-tick and an on-tick launch pair alternate,
both halves concurrently): wall microseconds from the trace, algorithmic TB/s, fraction of 8 TB/s and of the guide's 6.29.
    python tools/profile_summary_r04_bc.py <tag> "<note>" [vehicles]"""
import sys as _sys, os as _os
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _provenance import STEP_KERNEL, PLANNER_KERNEL, RENDER_KERNEL, kernel_source_hashes
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_factors import factors
FETCH_FACTOR, WRITE_FACTOR, FACTOR_SOURCE = factors()
tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
BYTES = float(sys.argv[4]) if len(sys.argv) > 4 else 144.0      # algorithmic bytes per vehicle-step of the passes' noise policy (148: libstdc++ streams)
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def rows(dirname, suffix):
    f = one(dirname + "/**/*_" + suffix + ".csv")
    return [r for r in csv.DictReader(open(f))] if f else []


f = one("prof_%s/**/*_kernel_stats.csv" % tag)
if f:
    shutil.copy(f, os.path.join(prof, "%s_kernel_stats.csv" % tag))
trace = [r for r in rows("prof_%s" % tag, "kernel_trace") if "afe_step_kernel" in r["Kernel_Name"]]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
per_kernel = collections.defaultdict(list)
for r in trace:
    per_kernel[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
summary = {"tag": tag, "build": note, "vehicles": N, "script": "tools/profile_r04_bc.sh %s %d" % (tag, N), "kernels": {}}
counters = {}
for cname, d in (("FETCH_SIZE", "pmc_fetch_%s" % tag), ("WRITE_SIZE", "pmc_write_%s" % tag)):
    agg = collections.defaultdict(list)
    for r in rows(d, "counter_collection"):
        if "afe_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == cname:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    counters[cname] = agg
halves = 2 if N >= (1 << 19) else 1
for k, durs in per_kernel.items():
    body = durs[len(durs) // 5:]                      # past the warm-up
    fs, ws = counters["FETCH_SIZE"].get(k), counters["WRITE_SIZE"].get(k)
    rec = {"launches": len(durs), "avg_us": sum(body) / len(body) / 1e3, "vehicles_per_launch": N // halves}
    if fs and ws:
        fb, wb = FETCH_FACTOR * 1024 * sum(fs[len(fs) // 5:]) / len(fs[len(fs) // 5:]), WRITE_FACTOR * 1024 * sum(ws[len(ws) // 5:]) / len(ws[len(ws) // 5:])
        rec.update({"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "fetch_bytes_per_vehicle": fb / (N // halves),
                    "write_bytes_per_vehicle": wb / (N // halves), "traffic_GBs_in_the_launch": (fb + wb) / (rec["avg_us"] * 1e-6) / 1e9})
    summary["kernels"][k] = rec
# wall time per step from the trace: the timed region's launches come in groups of `halves` per step
if trace:
    body = trace[len(trace) // 5:]
    t0, t1 = int(body[0]["Start_Timestamp"]), int(body[-1]["End_Timestamp"])
    steps = len(body) / halves
    us = (t1 - t0) / steps / 1e3
    traffic = sum(r.get("fetch_bytes_per_launch", 0) + r.get("write_bytes_per_launch", 0) for r in summary["kernels"].values()) / max(1, len(summary["kernels"])) * halves
    summary["per_step"] = {"us_trace_span_per_step": us, "note": "first start to last end of the launches past the warm-up / steps (host gaps between blocks inside)",
                           "pmc_bytes_per_step_mean_of_the_two_kernels": traffic, "pmc_bytes_per_vehicle_step": traffic / N,
                           "algorithmic_bytes_per_vehicle_step": BYTES, "algorithmic_GBs": N * BYTES / (us * 1e-6) / 1e9,
                           "frac_of_8000": N * BYTES / (us * 1e-6) / 1e9 / 8000.0, "frac_of_6290": N * BYTES / (us * 1e-6) / 1e9 / 6290.0}
summary["counter_factors"] = {"FETCH_SIZE": FETCH_FACTOR, "WRITE_SIZE": WRITE_FACTOR, "source": FACTOR_SOURCE}
if "per_step" in summary:
    ps = summary["per_step"]
    ps["pmc_over_algorithmic"] = ps["pmc_bytes_per_vehicle_step"] / ps["algorithmic_bytes_per_vehicle_step"]
    ps["what"] = ("beyond the 256 MiB Infinity Cache nothing survives from one step to the next, so the fabric-side counters ARE HBM traffic here "
                  "(state 52 B x %d vehicles = %.0f MB)" % (N, 52.0 * N / 1e6))
summary["kernel_sources"] = kernel_source_hashes(STEP_KERNEL)       # bench.py borrows from this file only while these match the tree
json.dump(summary, open(os.path.join(prof, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3500])
