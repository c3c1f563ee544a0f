#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration on a KNOWN byte count in the step kernel's own access pattern.

MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read;
"other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The step kernels read
and write one DWORD per lane per slab component (a wave = one 256-B request per component).  afe_stream_probe_kernel
<NRD, NWR> is that pattern with nothing else in it: per element it reads NRD planar dword streams and writes the first NWR
of them back in place -- NRD * 4 bytes read, NWR * 4 bytes written, exactly.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/fetch_calibration.py      (then WRITE_SIZE: separate pass)
    python tools/fetch_calibration.py --summarise gpurun_out/pmc_cal_fetch_<tag> gpurun_out/pmc_cal_write_<tag> <tag>

Sizes: 2^23 elements (805 / 570 MB read / written per launch: nothing of one launch is still in the 256 MiB Infinity Cache
for the next) and 2^20 (101 MB: every launch after the first is served by the Infinity Cache -- the counters sit on the
L2's fabric side and must not care)."""
import collections
import csv
import glob
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [(1 << 23, 24, 17), (1 << 23, 20, 13), (1 << 20, 24, 17), (1 << 20, 20, 13)]
LAUNCHES = 6          # per repetition; afe_stream_probe runs 4 repetitions


def run():
    afa = importlib.import_module("agri-fly_amd")
    for n, nrd, nwr in CASES:
        us = afa.stream_probe(n, nrd, nwr, LAUNCHES, 0)
        print("%8d elements, %d read / %d written dwords each: %.1f us per launch, %.0f GB/s" % (n, nrd, nwr, us, n * 4 * (nrd + nwr) / us / 1e3), flush=True)


def per_dispatch(d, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "afe_stream_probe_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return vals


def summarise(dfetch, dwrite, tag):
    rows, text = [], []
    fetch, write = per_dispatch(dfetch, "FETCH_SIZE"), per_dispatch(dwrite, "WRITE_SIZE")
    for n, nrd, nwr in CASES:
        key = [k for k in fetch if ("<%d, %d>" % (nrd, nwr) in k[0] or "ILi%dELi%dE" % (nrd, nwr) in k[0]) and k[1] == n]
        if not key:
            continue
        f, w = sorted(fetch[key[0]]), sorted(write.get(key[0], [0.0]))
        fk, wk = f[len(f) // 2], w[len(w) // 2]                          # KiB per dispatch (rocprofv3's unit), median
        rd, wr = n * nrd * 4.0, n * nwr * 4.0
        rows.append({"elements": n, "read_dwords": nrd, "written_dwords": nwr, "bytes_read": rd, "bytes_written": wr,
                     "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "dispatches": [len(f), len(w)],
                     "fetch_factor": rd / (fk * 1024.0) if fk else None, "write_factor": wr / (wk * 1024.0) if wk else None,
                     "FETCH_SIZE_min_max_KiB": [f[0], f[-1]], "WRITE_SIZE_min_max_KiB": [w[0], w[-1]]})
        text.append("%8d elements x (%2d dwords read, %2d written): known %7.1f MB read, %7.1f MB written | FETCH_SIZE %10.0f KiB -> x %.4f | WRITE_SIZE %10.0f KiB -> x %.4f"
                    % (n, nrd, nwr, rd / 1e6, wr / 1e6, fk, rd / (fk * 1024.0) if fk else float("nan"), wk, wr / (wk * 1024.0) if wk else float("nan")))
    big = [r for r in rows if r["elements"] == 1 << 23 and r["read_dwords"] == 24]
    out = {"tag": tag, "kernel": "afe::afe_stream_probe_kernel<NRD, NWR> (one-wave workgroups, one dword per lane per planar stream, in place)",
           "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; factor = known bytes / (counter x 1024)",
           "cases": rows,
           "fetch_factor_dword_streams": big[0]["fetch_factor"] if big else None,
           "write_factor_dword_streams": big[0]["write_factor"] if big else None,
           "guide_factor_16B_per_lane": {"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_fetch_calibration.json" % tag), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", "%s_fetch_calibration.txt" % tag), "w") as fh:
        fh.write("FETCH_SIZE / WRITE_SIZE against known bytes, dword-per-lane planar streams (tools/fetch_calibration.py, rocprofv3 --pmc, one counter per pass)\n")
        fh.write("\n".join(text) + "\n")
        fh.write("factor used by tools/profile_summary_*.py for the step kernels (2^23 elements, 24 + 17 streams): FETCH_SIZE x %s, WRITE_SIZE x %s\n"
                 % (out["fetch_factor_dword_streams"], out["write_factor_dword_streams"]))
    print("\n".join(text))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        run()
