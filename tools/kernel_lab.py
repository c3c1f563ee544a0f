#!/usr/bin/env python3
"""A/B timing of step-kernel builds on one MI355X (development tool).

    python tools/kernel_lab.py [--vehicles N] lib1.so lib2.so ...

For each library: HIP-event time of back-to-back single-step launches with the
logic gate never firing (off-tick launches), always firing (on-tick launches)
and at the bench cadence; prints one line per build.  Variants are interleaved
round-robin and repeated so that clock drift hits all of them alike.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, json, sys
sys.path.insert(0, %(root)r)
afa = importlib.import_module("agri-fly_amd")
import numpy as np
n = %(n)d
p = afa.params_from_type(5)
data = afa.scenarios.gust_ensemble(n, p, seed=4)
res = {}
for name, period, fext in (("off", 1000.0, True), ("on", 0.0005, True), ("mix", 0.002, True), ("off_nofext", 1000.0, False),
                           ("on_nonoise", 0.0005, True)):
    e = afa.Ensemble(n)
    e.set_type_table([p]); e.set_logic_period(period); e.set_split_stepping(1)   # kernel timing: one launch per step
    e.set_imu_noise(name != "on_nonoise", 0.1, 0.2, afa.AFE_SEED_DECORRELATED)
    e.set_state(data.pos, data.vel, data.att, data.ang_vel, data.motor_speed)
    e.set_motor_cmds(data.motor_cmd)
    if fext: e.set_external_force(data.ext_force)
    for _ in range(30): e.step(1000, 1)
    best = 1e9
    for rep in range(%(reps)d):
        a, b = e.event(), e.event()
        e.sync(); e.record(a)
        for _ in range(%(launches)d): e.step(1000, 1)
        e.record(b)
        best = min(best, e.elapsed_ms(a, b) * 1e3 / %(launches)d)
    res[name] = best
    e.close()
print(json.dumps(res))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vehicles", type=int, default=1 << 20)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    code = CHILD % dict(root=ROOT, n=a.vehicles, launches=a.launches, reps=a.reps)
    best = {}
    for rnd in range(a.rounds):
        for lib in a.libs:
            env = dict(os.environ, AGRIFLY_ENGINE_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(lib, "FAILED", out.stderr[-500:])
                continue
            r = json.loads(out.stdout.strip().splitlines()[-1])
            b = best.setdefault(lib, r)
            for k in r:
                b[k] = min(b[k], r[k])
    for lib in a.libs:
        if lib in best:
            r = best[lib]
            print("%-40s off %.2f us  on %.2f us  mix %.2f us  off_nofext %.2f us  on_nonoise %.2f us" %
                  (os.path.basename(lib), r["off"], r["on"], r["mix"], r["off_nofext"], r["on_nonoise"]))


if __name__ == "__main__":
    main()
