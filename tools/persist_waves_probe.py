"""persistent stepping: us per step by ensemble size and resident worker waves per CU (AFE_PERSIST_WAVES_PER_CU is read
when an engine first sizes its grid, so every point runs in a child process).
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
    steps = 3000
    e.step(1000, 300); e.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); e.step(1000, steps); e.sync(); best = min(best, time.perf_counter() - t0)
    print("%%.2f" %% (best / steps * 1e6), end=" ", flush=True)
    e.close()
''' % ROOT
sizes = [65536, 131072, 196608, 262144, 393216, 524288, 786432, 1 << 20]
print("waves/CU " + " ".join("%8d" % n for n in sizes))
for w in (4, 8, 12, 16, 20, 24, 28, 30, 31):
    out = subprocess.run([sys.executable, "-c", CHILD] + [str(n) for n in sizes], env=dict(os.environ, AFE_PERSIST_WAVES_PER_CU=str(w)),
                         capture_output=True, text=True, timeout=600)
    print("%8d " % w + " ".join("%8s" % x for x in out.stdout.split()) + ("   " + out.stderr[-300:] if out.returncode else ""), flush=True)
