mkdir -p gpurun_out/r03d
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03d/gpu_tests.txt 2>&1; echo "rc $?" >> gpurun_out/r03d/gpu_tests.txt
tail -15 gpurun_out/r03d/gpu_tests.txt
python - <<'PY'
import importlib, numpy as np
from oracle import oracle_py as ora
afa = importlib.import_module("agri-fly_amd")
from tests.test_gpu_counter import hover
n, first, seed = 1 << 16, 0, 1
e = hover(n, afa.AFE_F32, first_global=first)
e.set_imu_noise(True, 1.0, 0.0, afa.AFE_SEED_COUNTER); e.set_noise_seed(seed)
worst = 0
for tick in range(3):
    e.step(1000, e.steps_until_tick(1000))
    g, _ = e.get_imu()
    want = np.array([ora.imu_normals(seed, i, tick)[:3] for i in range(n)]).T
    err = np.abs(g - want); worst = max(worst, err.max())
print("fp32 counter normals, %d samples: worst |z - checker| = %.3g, mean %.3g" % (3 * 3 * n, worst, err.mean()))
PY
