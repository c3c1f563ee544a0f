"""bench.py's printed line: the driver parses ONE JSON line and lost a 20 kB one in round 3 (BENCH_r03.json parsed:
null).  compact_line keeps it below LINE_LIMIT whatever the full record holds; the full record goes to the side file.
Round 5 adds what the line must NOT say: no figure above 1 may be labelled a fraction of the HBM peak except the
headline's own `frac` (which states where its working set lives), rows whose working set fits the L2s carry no HBM
fraction at all, and the row that really streams from HBM (2^23 vehicles) is in the line."""
import copy
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def full_record():
    """a full record as bench.py wrote it on an MI355X this round (committed: profiles/r06_bench_detail_k20.json, 30+ kB)"""
    return json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail_k20.json")))


def _walk(x, path=""):
    if isinstance(x, dict):
        for k, v in x.items():
            yield from _walk(v, path + "/" + k)
    elif isinstance(x, list):
        for i, v in enumerate(x):
            yield from _walk(v, "%s[%d]" % (path, i))
    else:
        yield path, x


def test_full_size_record_gives_a_short_line_that_round_trips():
    full = full_record()
    assert len(json.dumps(full)) > 15000          # the record really is the long one
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT <= 6000, len(text)
    assert "\n" not in text
    back = json.loads(text)
    assert back == line
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == float("%.6g" % full["value"])
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "resident_in", "traffic_is", "hbm_streaming"):
        assert k in back["roofline"], k
    assert back["roofline"]["bound"] == "hbm" and back["roofline"]["peak"] == 8000.0
    for k in ("value", "unit", "cores", "kind", "sample", "all_cores"):
        assert k in back["cpu_baseline"], k
    assert set(back["config"]) >= {"workload", "vehicles_per_gpu", "vehicles_total", "stepping"}
    assert "model" not in back["config"]
    assert back["config4_as_stated"]["scaling"] == "strong"
    assert back["north_star_shard"]["vehicles_per_gpu"] == 131072
    assert back["counter_noise_policy"]["value"] > 0
    assert all(isinstance(v, float) for v in back["companions"].values())
    assert back["detail"] == bench.DETAIL_FILE
    for k in ("sweep", "sweep_note", "disturbance_sweep", "perception_rows"):
        assert k not in back
    # BASELINE configs 3 and 5 at their size, in the metric's unit (round-5 review item 1), and what bounds them
    c3, c5 = back["config3"], back["config5"]
    assert c3["vehicles"] == 65536 and c5["vehicles"] == 262144
    for row in (c3, c5):
        assert row["vsteps_per_s"] > 1e7 and row["frame_ms"] == pytest.approx(row["physics_ms"] + row["render_ms"] + row.get("plan_ms", 0.0), rel=1e-4)
        assert row["render_ms"] > 10 * row["physics_ms"]          # the perception kernels ARE the hot path of these two configs
    assert c3["plan_ms"] > 0 and 0 < c5["floor_valu_per_ray"] < c5["model_valu_per_ray"]
    cam, pln = back["perception"]["depth_camera"], back["perception"]["planner"]
    assert 0.5 < cam["valu_issue_frac"] <= 1.0 and cam["floor_valu_per_ray"] < cam["valu_per_ray"] and cam["counters_from"].startswith("profiles/r06")
    assert pln["valu_per_plan"] < 640530 / 1.4 and pln["counters_from"].startswith("profiles/r06")      # round 5: 640 530 vector instructions per plan


def test_the_line_labels_nothing_above_one_as_an_hbm_fraction():
    """round-4 review, "Next round" 1"""
    line = bench.compact_line(full_record())
    roof = line["roofline"]
    # the headline says where its bytes are served from; its traffic figure says what it counts
    assert roof["resident_in"] == "infinity_cache" and roof["working_set_bytes"] < roof["infinity_cache_bytes"]
    assert "not HBM" in roof["traffic_is"]
    # the true HBM row: 2^23 vehicles, a fraction of 8 000 GB/s below 1, the PMC bytes equal to the algorithmic ones
    hs = roof["hbm_streaming"]
    assert hs["vehicles"] == 1 << 23 and 0.5 < hs["frac"] < 1.0 and abs(hs["achieved"] / 8000.0 - hs["frac"]) < 1e-3
    assert 0.95 < hs["pmc_over_algorithmic"] < 1.10
    # rows that live in the L2s carry no HBM fraction but a bound, an L2-side rate and the vector pipes' share
    ns = line["north_star_shard"]
    assert "frac" not in ns and "launch_mode_frac" not in ns
    assert ns["resident_in"] == "l2" and ns["bound"] in ("valu_issue", "latency") and 0 < ns["l2_frac"] < 1 and 0 < ns["valu_busy_frac"] <= 1.0
    for row in line["closed_loop_on_device"]:
        if row["resident_in"] == "l2":
            assert "frac" not in row and row["bound"] in ("latency", "valu_issue")
        else:
            assert row["frac"] <= 1.0
    # every key called frac / frac_* anywhere in the line is a number not above 1 -- ratios to the plain stream probe are named ratios
    for path, v in _walk(line):
        key = path.rsplit("/", 1)[-1]
        if key.startswith("frac") or key.endswith("_frac") or key == "frac":
            assert v is None or v <= 1.0, (path, v)


def test_hostile_strings_and_missing_parts_still_fit():
    full = full_record()
    long = copy.deepcopy(full)
    long["config"]["workload"] = long["config"]["workload_short"] = "y" * 5000
    long["roofline"]["kernel"] = long["roofline"]["kernel_short"] = "k" * 5000
    long["roofline"]["traffic_source"] = "t" * 5000
    long["cpu_baseline"]["sample"] = "s" * 5000
    long["shared_world"] = {"error": "e" * 5000}
    long["companions"] = {"row_%03d" % i: {"value": 1.23456789e10 + i} for i in range(400)}     # pushes over: dropped, the contract stays
    text = json.dumps(bench.compact_line(long))
    assert len(text) <= bench.LINE_LIMIT
    back = json.loads(text)
    for k in CONTRACT:
        assert k in back, k
    # --headline-only / N > 1 records: no sweep, no cpu baseline, no strong row
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config", "roofline")}
    bare["config4_as_stated"] = None
    bare["roofline"] = dict(bare["roofline"], peak_measured=None, launch_mode=None, per_kernel=None)
    bare["roofline"].pop("beyond_cache", None)
    bare["roofline"].pop("hbm_streaming", None)
    back = json.loads(json.dumps(bench.compact_line(bare)))
    assert back["value"] > 0 and "cpu_baseline" not in back and "config4_as_stated" not in back


def test_non_finite_numbers_do_not_break_the_parser():
    full = full_record()
    full["roofline"]["traffic"] = float("nan")
    full["ms_per_step_max"] = float("inf")
    text = json.dumps(bench.compact_line(full))
    def refuse(token):          # json's hook for the bare tokens NaN / Infinity / -Infinity (the words may appear inside strings: "Infinity Cache")
        raise AssertionError("non-finite token %s in the line" % token)
    json.loads(text, parse_constant=refuse)


def test_side_file_holds_the_full_record(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    full = full_record()
    name = bench.write_detail(full)
    for d in (tmp_path, tmp_path / "gpurun_out"):
        assert json.load(open(d / name)) == full


def test_bound_fields_say_what_a_rate_is_a_fraction_of():
    ns = {"valu_instructions_per_wave_step": 460.0, "valu_active_frac": 0.39, "simd_valu_busy_frac": 0.78, "counters_from": "profiles/x.json"}
    small = bench.bound_fields(4096, 144.0, 1.6e-6, ns)                      # one wave on 64 of 1 024 SIMDs
    shard = bench.bound_fields(131072, 144.0, 2.0e-6, ns)                    # two waves per SIMD
    head = bench.bound_fields(1 << 20, 144.0, 19.5e-6, ns)
    big = bench.bound_fields(1 << 23, 144.0, 212e-6, ns)
    assert small["resident_in"] == "l2" and small["bound"] == "latency" and "frac" not in small
    assert shard["resident_in"] == "l2" and shard["bound"] == "valu_issue" and shard["valu_busy_frac"] == 0.78 and "frac" not in shard
    quarter = bench.bound_fields(262144, 148.0, 3.8e-6, ns)                  # 28 MB of distinct bytes: still the L2s', four waves per SIMD
    assert quarter["resident_in"] == "l2" and quarter["bound"] == "valu_issue" and "frac" not in quarter and "valu_busy_frac" not in quarter
    assert head == {"bound": "hbm", "resident_in": "infinity_cache", "frac": head["frac"]} and 0.9 < head["frac"] < 1.0
    assert big["resident_in"] == "hbm" and 0.6 < big["frac"] < 0.8
    assert bench.residency(109e6) == "infinity_cache" and bench.residency(19e6) == "l2" and bench.residency(905e6) == "hbm"


def test_scaling_check_puts_a_multi_gpu_run_beside_the_one_gpu_shard_rates():
    """round-4 review, "Next round" 3d: a first --gpus N run is judged against what one GPU delivers on the same shard"""
    exp = json.load(open(os.path.join(ROOT, "profiles", "expected_rates.json")))
    assert set(exp["strong_shard"]) >= {"131072", "262144", "524288", "1048576"} and "1048576" in exp["weak_per_gpu"]
    one = exp["weak_per_gpu"]["1048576"]["vsteps_per_s"]
    shard = exp["strong_shard"]["131072"]["vsteps_per_s"]
    sc = bench.scaling_check(exp, "profiles/expected_rates.json", 8, 1 << 20, 8 * 0.95 * one, 131072, 8 * 0.5 * shard)
    assert sc["n_gpus"] == 8
    assert abs(sc["weak"]["ratio"] - 0.95) < 1e-9 and sc["weak"]["measured_per_gpu"] == 0.95 * one
    assert abs(sc["strong"]["ratio"] - 0.5) < 1e-9 and sc["strong"]["vehicles_per_gpu"] == 131072
    assert bench.scaling_check(exp, "x", 1, 1 << 20, one, 1 << 20, one) is None           # one GPU: nothing to compare
    assert bench.scaling_check(None, None, 8, 1 << 20, one, 131072, shard) is None       # no committed rates
    line = bench.compact_line(dict(full_record(), scaling_check=sc))
    assert line["scaling_check"]["strong"]["ratio"] == 0.5


def _fake_profiles(tmp_path, monkeypatch, sources):
    """a profiles/ directory holding one planner and one north-star summary that claim the given kernel-source hashes"""
    prof = tmp_path / "profiles"
    prof.mkdir()
    json.dump({"valu_issue_fraction_of_busy_cycles": 0.5, "per_plan": {"valu_instructions": 4.0e5}, "kernel_sources": sources},
              open(prof / "r99_planner_pmc.json", "w"))
    json.dump({"valu_active_frac_of_wave_cycles": 0.38, "simd_valu_busy_frac": 0.76, "valu_instructions_per_wave_and_step": 556.0,
               "noise_policy": "reference_streams", "kernel_sources": sources}, open(prof / "r99_ns_summary.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "STALE", [])


def test_borrowed_counters_ride_along_only_while_the_kernel_sources_match(tmp_path, monkeypatch):
    """round-5 review item 5: every committed summary records sha256 of the kernel sources it was taken on; bench.py borrows
    from it only while they are the tree's, and says counters_stale otherwise"""
    now = bench.PROV.kernel_source_hashes()
    assert all(now.values()) and set(now) >= set(bench.PROV.STEP_KERNEL + bench.PROV.PLANNER_KERNEL + bench.PROV.RENDER_KERNEL)
    _fake_profiles(tmp_path, monkeypatch, now)
    pick = lambda d: d if d.get("valu_issue_fraction_of_busy_cycles") else None
    rec, src = bench.committed_json("r*_planner_pmc.json", pick, bench.PROV.PLANNER_KERNEL)
    assert rec["per_plan"]["valu_instructions"] == 4.0e5 and src == os.path.join("profiles", "r99_planner_pmc.json")
    ns = bench.committed_ns_profile(True)
    assert ns["simd_valu_busy_frac"] == 0.76 and ns["counters_from"].endswith("r99_ns_summary.json") and not bench.STALE
    row = bench.bound_fields(131072, 148.0, 2.5e-6, ns)
    assert row["valu_busy_frac"] == 0.76 and "counters_stale" not in row


def test_an_edited_kernel_drops_the_borrowed_counters(tmp_path, monkeypatch):
    now = bench.PROV.kernel_source_hashes()
    edited = dict(now, **{"afe_planner.hip": "0" * 16, "afe_kernels.hip": "f" * 16})
    _fake_profiles(tmp_path, monkeypatch, edited)
    pick = lambda d: d if d.get("valu_issue_fraction_of_busy_cycles") else None
    assert bench.committed_json("r*_planner_pmc.json", pick, bench.PROV.PLANNER_KERNEL) == (None, None)
    ns = bench.committed_ns_profile(True)
    assert ns.get("counters_stale") and "simd_valu_busy_frac" not in ns
    assert sorted(bench.STALE) == [os.path.join("profiles", "r99_ns_summary.json"), os.path.join("profiles", "r99_planner_pmc.json")]
    row = bench.bound_fields(131072, 148.0, 2.5e-6, ns)
    assert row["counters_stale"] is True and "valu_busy_frac" not in row and "counters_from" not in row
    # a summary without hashes (rounds 1-5) is stale by definition; without the `sources` argument nothing is checked
    json.dump({"valu_issue_fraction_of_busy_cycles": 0.5}, open(tmp_path / "profiles" / "r98_render_pmc.json", "w"))
    assert bench.committed_json("r*_render_pmc.json", pick, bench.PROV.RENDER_KERNEL) == (None, None)
    assert bench.committed_json("r*_render_pmc.json", pick)[0] is not None
    # and the printed line says so instead of printing last round's fractions
    full = full_record()
    full["counters_stale"] = list(bench.STALE)
    full["north_star_shard"] = dict({k: v for k, v in full["north_star_shard"].items() if not k.startswith("valu_") and k != "counters_from"}, counters_stale=True)
    full["config3"] = {"vehicles": 65536, "frame_ms": 98.9, "physics_ms": 0.13, "render_ms": 77.8, "plan_ms": 21.0, "vsteps_per_s": 1.99e7,
                       "rays_per_s": 6.5e10, "plans_per_s": 3.1e6, "bound_short": "depth camera: valu issue; planner: latency", "counters_stale": True}
    full["config5"] = {"vehicles": 262144, "frame_ms": 402.0, "physics_ms": 0.2, "render_ms": 401.7, "vsteps_per_s": 2.15e7, "rays_per_s": 5.0e10,
                       "bound_short": "depth camera: valu issue", "counters_stale": True}
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert line["counters_stale"] and line["north_star_shard"]["counters_stale"] is True and "valu_busy_frac" not in line["north_star_shard"]
    for key, n in (("config3", 65536), ("config5", 262144)):
        assert line[key]["vehicles"] == n and line[key]["vsteps_per_s"] > 1e7 and line[key]["counters_stale"] is True
        assert "valu_issue_frac" not in line[key]
    assert line["config3"]["plan_ms"] == 21.0 and "plan_ms" not in line["config5"]
