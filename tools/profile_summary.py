#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_r02.sh <tag> (under gpurun_out/) into the committed
summaries under profiles/:  python tools/profile_summary.py <tag> "<build note>"
  profiles/<tag>_kernel_stats.csv        rocprofv3 --stats table of the bench's timed cadence
  profiles/<tag>_full_kernel_stats.csv   the same for the whole default bench (shared world, perception, sweep)
  profiles/<tag>_summary.json            per step-kernel instantiation: launch time, algorithmic bytes,
                                         FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE x 2 per the
                                         gfx950 note of MI355X_MICROARCH.md, KiB units), SQ counters
  profiles/traffic.json                  PMC HBM bytes per launch of the bench workload (read by bench.py)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
# vehicles per LAUNCH: 2^20 on one stream, 2^19 when the bench steps its shard as two halves (afe_set_split_stepping)
N_LAUNCH = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
N_SHARD = 1 << 20
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None


def short(name):
    if "afe_step_kernel<" in name:
        args = name.split("afe_step_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return "step<%s,FEXT=%s,TEXT=%s,NOISE=%s,LOGIC=%s,SINGLE=%s>" % tuple(
            [args[0]] + [{"true": "1", "false": "0"}[a] for a in args[1:6]])
    return name.split("(")[0].replace("afe::", "").replace("(anonymous namespace)::", "").replace("void ", "")


def counters(dirname):
    f = one(dirname + "/**/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


summary = {"build": note, "tag": tag, "script": "tools/profile_r02.sh %s" % tag, "kernels": {}}
for src, dst in (("prof_%s" % tag, "%s_kernel_stats.csv" % tag), ("prof_full_%s" % tag, "%s_full_kernel_stats.csv" % tag)):
    f = one(src + "/**/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(prof, dst))
    d = one(src + "/**/*_domain_stats.csv")
    if d and src.startswith("prof_" + tag):
        shutil.copy(d, os.path.join(prof, dst.replace("kernel_stats", "domain_stats")))
f = one("prof_%s/**/*_kernel_stats.csv" % tag)
stats = {short(r["Name"]): r for r in csv.DictReader(open(f))} if f else {}
fetch, write, sq = counters("pmc_fetch_%s" % tag), counters("pmc_write_%s" % tag), counters("pmc_sq_%s" % tag)
N = N_LAUNCH
BYTES = {"0": 132.0, "1": 164.0}     # algorithmic B per vehicle-step of the bench workload, off / on tick
tot_alg = tot_pmc = tot_n = 0
for k, r in stats.items():
    if not k.startswith("step<float"):
        continue
    noise = k.split("NOISE=")[1][0]
    rec = {"calls": int(r["Calls"]), "average_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
           "algorithmic_bytes_per_vehicle": BYTES[noise], "algorithmic_GBs": N * BYTES[noise] / float(r["AverageNs"]),
           "frac_of_8TBs": N * BYTES[noise] / float(r["AverageNs"]) / 8000.0}
    if k in fetch and k in write:
        fs = fetch[k]["FETCH_SIZE"]
        ws = write[k]["WRITE_SIZE"]
        rec["FETCH_SIZE_KiB_mean"] = sum(fs) / len(fs)
        rec["WRITE_SIZE_KiB_mean"] = sum(ws) / len(ws)
        rec["hbm_bytes_per_launch"] = 2 * 1024 * rec["FETCH_SIZE_KiB_mean"] + 1024 * rec["WRITE_SIZE_KiB_mean"]
        rec["hbm_bytes_per_vehicle"] = rec["hbm_bytes_per_launch"] / N
        tot_alg += N * BYTES[noise] * len(fs)
        tot_pmc += rec["hbm_bytes_per_launch"] * len(fs)
        tot_n += len(fs)
    if k in sq:
        rec["sq_per_launch"] = {c: sum(v) / len(v) for c, v in sorted(sq[k].items())}
        w = rec["sq_per_launch"].get("SQ_WAVES", 0)
        if w:
            rec["valu_instructions_per_wave"] = rec["sq_per_launch"].get("SQ_INSTS_VALU", 0) / w
            rec["salu_instructions_per_wave"] = rec["sq_per_launch"].get("SQ_INSTS_SALU", 0) / w
    summary["kernels"][k] = rec
if tot_n:
    summary["traffic"] = {"pmc_bytes_per_launch_mean": tot_pmc / tot_n, "algorithmic_bytes_per_launch_mean": tot_alg / tot_n,
                          "ratio": tot_pmc / tot_alg}
    json.dump({"workload": {"vehicles_per_gpu": N_SHARD, "fext": True, "noise": True, "dt_us": 1000, "logic_period_s": 0.002},
               "vehicles_per_launch": N, "launches_per_step": N_SHARD // N,
               "traffic_bytes_per_launch": tot_pmc / tot_n,
               "traffic_bytes_per_step": tot_pmc / tot_n * (N_SHARD // N),
               "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                         "FETCH_SIZE x2 gfx950 correction)" % tag}, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
# how many step kernels are in flight over the timed cadence (1 on one stream; ~2 with split stepping): sum of the
# kernels' durations / time from the first start to the last end, from the kernel trace of the --stats pass
tr = one("prof_%s/**/*_kernel_trace.csv" % tag)
if tr:
    ks = [r for r in csv.DictReader(open(tr)) if "afe_step_kernel" in r["Kernel_Name"]]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    ks = ks[len(ks) // 5:]                       # past the warm-up
    if len(ks) > 10:
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
        span = max(int(r["End_Timestamp"]) for r in ks) - int(ks[0]["Start_Timestamp"])
        per_step = N_SHARD // N
        summary["cadence"] = {"step_kernels_in_flight_mean": busy / span, "launches": len(ks), "launches_per_step": per_step,
                              "time_per_step_us": span / (len(ks) / per_step) / 1e3,
                              "algorithmic_GBs_over_the_cadence": (tot_alg / tot_n if tot_n else 0) * len(ks) / span}
json.dump(summary, open(os.path.join(prof, "%s_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
