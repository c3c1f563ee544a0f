"""persistent stepping: us per step by ensemble size, resident worker waves per CU and run length
(AFE_PERSIST_WAVES_PER_CU / AFE_PERSIST_BALANCED are read when a grid is sized, so every point runs in a child process).
    python tools/persist_waves_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys, time
import torch
sys.path.insert(0, %r)
afa = importlib.import_module("agri-fly_amd")
import bench
for n in [int(x) for x in sys.argv[1:]]:
    e = bench.build_shard(afa, n, 0, n, 0)
    e.set_step_mode(afa.AFE_STEP_PERSISTENT)
    e.step(1000, 300); e.sync()
    out = []
    for steps in (200, 3000):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); e.step(1000, steps); e.sync(); best = min(best, time.perf_counter() - t0)
        out.append("%%.2f" %% (best / steps * 1e6))
    print("/".join(out), end=" ", flush=True)
    e.close()
''' % ROOT
sizes = [131072, 262144, 393216, 524288, 786432, 1 << 20]
print("us per step in runs of 200 / 3000 steps")
print("waves/CU       " + " ".join("%13d" % n for n in sizes))
for w, unb in ((8, 1), (12, 1), (16, 1), (20, 1), (24, 1), (24, 0)):
    env = dict(os.environ, AFE_PERSIST_WAVES_PER_CU=str(w))
    if not unb:
        env["AFE_PERSIST_BALANCED"] = "1"
    out = subprocess.run([sys.executable, "-c", CHILD] + [str(n) for n in sizes], env=env, capture_output=True, text=True, timeout=900)
    print("%8d %s " % (w, "     " if unb else "equal") + " ".join("%13s" % x for x in out.stdout.split()) + ("   " + out.stderr[-300:] if out.returncode else ""), flush=True)
