"""The factors that turn rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB) into bytes for the step kernels' access pattern:
measured on a known byte count (tools/fetch_calibration.py -> profiles/<tag>_fetch_calibration.json), newest round first;
the guide's figure for 16-B-per-lane streams (FETCH_SIZE x 2, WRITE_SIZE x 1) where no calibration is committed."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def factors():
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fetch_calibration.json")), reverse=True):
        try:
            d = json.load(open(f))
            ff, wf = d.get("fetch_factor_dword_streams"), d.get("write_factor_dword_streams")
            if ff and wf:
                return float(ff), float(wf), os.path.relpath(f, ROOT)
        except (OSError, ValueError):
            pass
    return 2.0, 1.0, "MI355X_MICROARCH.md (16 B per lane; uncalibrated for dword streams)"
