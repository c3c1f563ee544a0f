"""Is the headless flight branch reproducible run to run?  N runs of rappids_headless --scene (the flight of
tests/test_gpu_headless.py), each hashed; with `load` a second process keeps the GPU busy meanwhile.  Prints the first row at
which a run differs from the first run and the trajectory-log row count."""
import hashlib, importlib, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
afa = importlib.import_module("agri-fly_amd")
EXE = os.environ.get("FLIGHT_EXE") or os.path.join(ROOT, "agri-fly_amd", "bin", "rappids_headless")
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
load = len(sys.argv) > 2 and sys.argv[2] == "load"
tris = afa.scenarios.orchard_mesh(rows=4, cols=8, seed=3)
tris = (tris.reshape(-1, 3, 3) + np.array([5.0, -2.0, 0.0])).reshape(-1, 9).astype(np.float32)
tmp = tempfile.mkdtemp()
mesh = os.path.join(tmp, "orchard.f32")
tris.tofile(mesh)
goal = [5.0 + 7 * 3.0 + 8.0, 0.0, 1.2]
args = ["--scene", mesh, "--goal"] + [str(g) for g in goal] + ["--hover", "1.2", "--start-flight", "2.0", "--seconds", "9.0", "--dt-us", "1000",
                                                                   "--candidates", "192", "--digits", "17", "--estimator", "truth"]
bg = None
if load:
    bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
first = None
for k in range(n_runs):
    out, traj, imgs = os.path.join(tmp, "sim%d.csv" % k), os.path.join(tmp, "traj%d.csv" % k), os.path.join(tmp, "img%d.csv" % k)
    subprocess.check_output([EXE, "--out", out, "--traj-log", traj, "--image-log", imgs] + args)
    rows, plans = open(out).read().split("\n"), open(traj).read().split("\n")
    images = open(imgs).read().split("\n")
    h = hashlib.sha256(("\n".join(rows)).encode()).hexdigest()[:12]
    if first is None:
        first = (rows, plans, images)
        print("run 0: %s, %d rows, %d plans" % (h, len(rows), len(plans)))
    else:
        d = next((i for i, (a, b) in enumerate(zip(first[0], rows)) if a != b), None)
        dp = next((i for i, (a, b) in enumerate(zip(first[1], plans)) if a != b), None)
        print("run %d: %s first differing state row %s, first differing plan %s%s" % (k, h, d, dp, "" if d is None else "  t = " + rows[d].split(",")[0]))
        di = next((i for i, (a, b) in enumerate(zip(first[2], images)) if a != b), None)
        if di is not None:
            a, b = first[2][di].split(","), images[di].split(",")
            print("   first differing image record %d at t = %s: checksum %s, pose %s" % (di, b[0], "DIFFERS" if a[1] != b[1] else "same", "DIFFERS" if a[2:] != b[2:] else "same"))
        if dp is not None:
            a, b = first[1][dp].split(","), plans[dp].split(",")
            print("   plan columns that differ:", [i for i, (x, y) in enumerate(zip(a, b)) if x != y][:20])
    if bg is not None and bg.poll() is not None:
        bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if bg is not None:
    bg.wait()
