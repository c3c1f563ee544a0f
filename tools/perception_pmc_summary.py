#!/usr/bin/env python3
"""SQ counters of the two perception kernels (tools/planner_pmc.sh <tag> 65536 orchard, tools/render_pmc.sh <tag>) ->
profiles/<tag>_planner_pmc.json, profiles/<tag>_render_pmc.json: what bounds each kernel (share of the shader engines'
busy cycles in which a vector instruction issues), instructions per plan / per ray, and the rates.
    python tools/perception_pmc_summary.py <tag>
bench.py reads the two files for the `perception` rows' bound and fraction (counters cannot be read inside the run)."""
import sys as _sys, os as _os
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _provenance import STEP_KERNEL, PLANNER_KERNEL, RENDER_KERNEL, kernel_source_hashes
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def counters(dirs, needle):
    acc, launches = collections.defaultdict(float), 0
    for d in dirs:
        per = collections.defaultdict(int)
        for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if needle in r["Kernel_Name"]:
                    acc[r["Counter_Name"] + "@" + d] += float(r["Counter_Value"])
                    per[r["Counter_Name"]] += 1
        launches = max([launches] + list(per.values()))
    merged = {}
    for k, v in acc.items():
        merged.setdefault(k.split("@")[0], v)       # a counter collected in both passes: the first pass's
    return merged, launches


def durations(d, needle):
    t = []
    for f in glob.glob(os.path.join(out, d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return t


def fractions(c):
    busy, wc = max(1.0, c.get("SQ_BUSY_CYCLES", 0)), max(1.0, c.get("SQ_WAVE_CYCLES", 0))
    # the formula of profiles/r02d / r03: counters are sums over the device -- VALU-active cycles over 256 compute units,
    # busy cycles over 32 shader engines -- so per compute unit and busy cycle: ACTIVE / 256 / (BUSY / 32) = ACTIVE / (8 BUSY)
    return {"valu_issue_fraction_of_busy_cycles": c.get("SQ_ACTIVE_INST_VALU", 0) / 8.0 / busy,
            "valu_active_frac_of_wave_cycles": c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / wc if "SQ_WAIT_ANY" in c else None,
            "wait_inst_any_frac_of_wave_cycles": c.get("SQ_WAIT_INST_ANY", 0) / wc if "SQ_WAIT_INST_ANY" in c else None}


# ---- planner: 65 536 planners x 256 candidates on 512 orchard views, 3 timed calls + warm-up in the probe
c, n_launch = counters(["pmc_planner_%s" % tag, "pmc_planner2_%s" % tag], "rappids_search_kernel")
if c:
    calls = 3                 # tools/planner_probe.py: three timed calls, no separate warm-up
    planners = 65536
    plans = planners * calls
    t = sorted(durations("pmc_planner_%s" % tag, "rappids_search_kernel"))
    rec = {"kernel": "afe::afe_rappids_search_kernel (longest first from 16 385 planners up: a sizing launch and a finishing launch per call)",
           "workload": "tools/planner_probe.py 65536 orchard: 65 536 planners x 256 candidates on 512 views rendered inside the 32x32 orchard, %d calls" % calls,
           "search_launches": n_launch, "counters_sum_over_launches": {k: v for k, v in sorted(c.items())},
           "per_plan": {"valu_instructions": c.get("SQ_INSTS_VALU", 0) / plans, "salu_instructions": c.get("SQ_INSTS_SALU", 0) / plans,
                        "lds_instructions": c.get("SQ_INSTS_LDS", 0) / plans, "vmem_reads": c.get("SQ_INSTS_VMEM_RD", 0) / plans,
                        "wave_cycles": c.get("SQ_WAVE_CYCLES", 0) / plans},
           "search_kernel_ms_under_the_counter_pass": {"sum_per_call": sum(t) / calls if t else None, "launches": len(t)}}
    rec.update(fractions(c))
    rec["shader_engines_busy_fraction_note"] = "SQ_BUSY_CYCLES counts cycles in which a shader engine holds any wave; with longest-first scheduling the tail is gone (round 3: 0.57 -> 0.97 of the launch)"
    rec["bound"] = ("latency of a sequential search: one wave per planner, %.0f vector instructions per plan; a vector instruction is executing in %.0f %% of a "
                    "compute unit's busy cycles and in %.0f %% of a wave's resident cycles (it waits in %.0f %% of them)"
                    % (rec["per_plan"]["valu_instructions"], 100 * rec["valu_issue_fraction_of_busy_cycles"], 100 * rec["valu_active_frac_of_wave_cycles"],
                       100 * (rec["wait_any_frac_of_wave_cycles"] or 0)))
    rec["kernel_sources"] = kernel_source_hashes(PLANNER_KERNEL)
    json.dump(rec, open(os.path.join(prof, "%s_planner_pmc.json" % tag), "w"), indent=1)
    print(json.dumps(rec, indent=1)[:2500])

# ---- depth camera: 1 024 views of 320 x 240, 3 launches + the counting build's in the probe
c, n_launch = counters(["pmc_render_%s" % tag, "pmc_render2_%s" % tag], "render_depth_kernel<false>")
if not c:
    c, n_launch = counters(["pmc_render_%s" % tag, "pmc_render2_%s" % tag], "render_depth_kernel")
if c:
    waves = max(1.0, c.get("SQ_WAVES", 0))
    rays = waves * 64.0
    t = sorted(durations("pmc_render_%s" % tag, "render_depth_kernel"))
    rec = {"kernel": "afe::afe_render_depth_kernel<false> (+ afe::afe_tile_entry_kernel)",
           "workload": "tools/render_stats_probe.py: 1024 views of 320x240 over the 32x32-tree orchard", "launches": n_launch,
           "counters_sum_over_launches": {k: v for k, v in sorted(c.items())},
           "per_wave_of_64_rays": {"valu_instructions": c.get("SQ_INSTS_VALU", 0) / waves, "salu_instructions": c.get("SQ_INSTS_SALU", 0) / waves,
                                   "smem_instructions": c.get("SQ_INSTS_SMEM", 0) / waves},
           "per_ray": {"valu_instructions": c.get("SQ_INSTS_VALU", 0) / rays},
           "scalar_issue_fraction_of_busy_cycles": c.get("SQ_ACTIVE_INST_SCA", 0) / 8.0 / max(1.0, c.get("SQ_BUSY_CYCLES", 0)),
           "kernel_ms_under_the_counter_pass_median": t[len(t) // 2] if t else None}
    rec.update(fractions(c))
    rec["bound"] = ("vector-instruction issue: a vector instruction is executing in %.0f %% of a compute unit's busy cycles (%.0f per wave of 64 rays, %.1f per ray)"
                    % (100 * rec["valu_issue_fraction_of_busy_cycles"], rec["per_wave_of_64_rays"]["valu_instructions"], rec["per_ray"]["valu_instructions"]))
    rec["kernel_sources"] = kernel_source_hashes(RENDER_KERNEL)
    json.dump(rec, open(os.path.join(prof, "%s_render_pmc.json" % tag), "w"), indent=1)
    print(json.dumps(rec, indent=1)[:2500])
