"""Which kernel sources a committed counter summary was taken on.

bench.py borrows figures that cannot be read inside a run (rocprofv3 PMC counters) from the summaries committed under
profiles/.  Every summary records `kernel_sources` = sha256 of the sources of the kernels it profiled; bench.py attaches a
borrowed figure only while those hashes still match the tree, and prints `counters_stale` instead once a kernel has been
edited without re-profiling (round-5 review, item 5)."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

# the sources whose text decides what the profiled kernel executes
STEP_KERNEL = ("afe_kernels.hip", "afe_device.h")
PLANNER_KERNEL = ("afe_planner.hip", "afe_planner.h")
RENDER_KERNEL = ("afe_render.hip", "afe_render.h")
ALL = STEP_KERNEL + PLANNER_KERNEL + RENDER_KERNEL


def kernel_source_hashes(names=ALL, csrc=CSRC):
    out = {}
    for n in names:
        try:
            with open(os.path.join(csrc, n), "rb") as f:
                out[n] = hashlib.sha256(f.read()).hexdigest()[:16]
        except OSError:
            out[n] = None
    return out


def taken_on_this_tree(record, names, csrc=CSRC):
    """True when `record` (a committed summary) carries the hashes of `names` and they are the tree's"""
    have = (record or {}).get("kernel_sources") or {}
    now = kernel_source_hashes(names, csrc)
    return all(now[n] is not None and have.get(n) == now[n] for n in names)
