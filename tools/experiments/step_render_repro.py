"""The headless loop's camera cadence without the planner: one vehicle (host-visible engine, resident grid) steps 1 ms at a
time with varying commands, every 33rd step its depth image is rendered from engine state into a device buffer and
hashed together with the pose read back.  The whole sequence twice (or more) under GPU load: hashes must repeat."""
import hashlib, importlib, os, subprocess, sys, time
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
afa = importlib.import_module("agri-fly_amd")
tris = afa.scenarios.orchard_mesh(rows=4, cols=8, seed=3)
tris = (tris.reshape(-1, 3, 3) + np.array([5.0, -2.0, 0.0])).reshape(-1, 9).astype(np.float32)
scene = afa.Scene(tris)
cam, mount = afa.camera_default(320, 240), afa.camera_default_mount()
p = afa.params_from_type(5)
host_visible = os.environ.get("REPRO_HOST_VISIBLE", "1") == "1"
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
time.sleep(20)
buf = afa.DeviceBuffer(240 * 320 * 2)
ref = None
for r in range(runs):
    e = afa.Ensemble(1, precision=afa.AFE_F64 if os.environ.get("REPRO_F64", "1") == "1" else afa.AFE_F32, host_visible=host_visible)
    e.set_type_table([p])
    e.set_logic_period(1 / 500)
    e.set_rates_logic([afa.rates_logic_params_from_type(5)])
    e.set_state(pos=np.array([[0.0], [0.0], [1.2]]), vel=np.array([[1.5], [0.0], [0.0]]), att=np.array([[1.0], [0], [0], [0]]),
                ang_vel=np.zeros((3, 1)), motor_speed=np.full((4, 1), afa.scenarios.hover_speed(p)))
    e.set_step_mode(afa.AFE_STEP_AUTO)
    seq = []
    for f in range(frames):
        for s in range(33):
            if s % 10 == 0:
                e.set_rates_commands(np.full(1, 9.81 + 0.3 * np.sin(0.01 * (f * 33 + s)), np.float32), np.array([[0.05 * np.sin(0.02 * f)], [0.04], [0.1 * np.cos(0.03 * f)]], np.float32))
            e.step(1000, 1)
        scene.render_engine(e, cam, mount, out=buf)
        st = e.get_state()
        img = buf.download(np.uint16, (240, 320))
        seq.append((hashlib.sha256(img.tobytes()).hexdigest()[:10], hashlib.sha256(st["pos"].tobytes() + st["att"].tobytes()).hexdigest()[:10]))
    e.close()
    if ref is None:
        ref = seq
        print("run 0: %d frames" % len(seq))
    else:
        bad = [(i, a[0] != b[0], a[1] != b[1]) for i, (a, b) in enumerate(zip(ref, seq)) if a != b]
        print("run %d: %d frames differ%s" % (r, len(bad), (" first: frame %d image %s pose %s" % (bad[0][0], "DIFFERS" if bad[0][1] else "same", "DIFFERS" if bad[0][2] else "same")) if bad else ""))
    if bg.poll() is not None:
        bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "experiments", "gpu_load.py"), os.environ.get("REPRO_LOAD_S", "120")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
bg.wait()
