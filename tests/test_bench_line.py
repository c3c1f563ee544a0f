"""bench.py's printed line: the driver parses ONE JSON line and lost a 20 kB one in round 3 (BENCH_r03.json parsed:
null).  compact_line keeps it below LINE_LIMIT whatever the full record holds; the full record goes to the side file."""
import copy
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def full_record():
    """round 3's full 20 kB record (committed), with the keys this round adds"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r03c_bench_line_k20.json")))
    full["reference_noise_streams"] = {"value": 4.9674346308e10, "unit": "vehicle-steps/s", "ms_per_step": 0.021109004504978657, "steps": 20,
                                       "repeats": 124, "algorithmic_bytes_per_vehicle_step": 147.96800000000002, "kernel_us": 21.109004504978657,
                                       "frac": 0.9187767093150095, "frac_wall": 0.9, "stepping": "persistent", "seed_policy": "AFE_SEED_DECORRELATED",
                                       "note": "x" * 300}
    full["config"]["workload_short"] = "w" * 390
    full["config"]["noise"] = "AFE_SEED_COUNTER (Philox4x32-10 + Box-Muller); the reference's libstdc++ streams: reference_noise_streams"
    full["config"]["parallelism_short"] = "contiguous shards, 1 rank(s), no data-path collective"
    full["roofline"]["kernel_short"] = "afe_step_persistent_kernel<float,FEXT,NOISE=counter>"
    full["north_star_shard"]["us_per_step_k_blocks"] = 3.4123456789
    if "beyond_cache" in full["roofline"]:
        full["roofline"]["beyond_cache"]["frac_of_6290"] = 0.8456789123
    return full


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
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    assert back["roofline"]["bound"] == "hbm" and back["roofline"]["peak"] == 8000.0
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert set(back["config"]) >= {"workload", "vehicles_per_gpu", "vehicles_total", "stepping"}
    assert "model" not in back["config"]
    # what the review asked to see in the short line
    assert back["config4_as_stated"]["scaling"] == "strong"
    assert back["north_star_shard"]["vehicles_per_gpu"] == 131072
    assert back["reference_noise_streams"]["frac"] > 0
    assert all(isinstance(v, float) for v in back["companions"].values())
    assert back["detail"] == bench.DETAIL_FILE
    # nothing bulky came along
    for k in ("sweep", "sweep_note", "disturbance_sweep", "perception_rows"):
        assert k not in back


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
    back = json.loads(json.dumps(bench.compact_line(bare)))
    assert back["value"] > 0 and "cpu_baseline" not in back and "config4_as_stated" not in back


def test_non_finite_numbers_do_not_break_the_parser():
    full = full_record()
    full["roofline"]["traffic"] = float("nan")
    full["ms_per_step_max"] = float("inf")
    text = json.dumps(bench.compact_line(full))
    assert "NaN" not in text and "Infinity" not in text
    json.loads(text)


def test_side_file_holds_the_full_record(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    full = full_record()
    name = bench.write_detail(full)
    for d in (tmp_path, tmp_path / "gpurun_out"):
        assert json.load(open(d / name)) == full
