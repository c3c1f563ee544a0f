"""small ensembles (BASELINE config 2 and below): microseconds per step by stepping mode, open loop and with the on-device
rates logic, one afe_step call per step, 400-step bracketed blocks.   python tools/small_n_probe.py [vehicles ...]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench
sync = torch.cuda.synchronize
modes = (("launch", afa.AFE_STEP_LAUNCH), ("persistent", afa.AFE_STEP_PERSISTENT), ("resident", afa.AFE_STEP_RESIDENT), ("auto", afa.AFE_STEP_AUTO))
for n in ([int(a) for a in sys.argv[1:]] or (1024, 4096, 16384, 65536, 131072)):
    for closed in (False, True):
        row = []
        for name, mode in modes:
            e = bench.build_shard(afa, n, 0, n, 0)
            if closed:
                e.set_rates_logic([afa.rates_logic_params_from_type(5)])
                e.set_rates_commands(np.full(n, 9.81, np.float32), np.zeros((3, n), np.float32))
            e.set_step_mode(mode)
            blocks = bench.timed_blocks(e, 400, 1, sync, lambda: None, lambda x: x, min_total_s=0.03)
            b20 = bench.timed_blocks(e, 20, 1, sync, lambda: None, lambda x: x, min_total_s=0.02, settle_s=0.0)
            row.append("%s %.2f (K=20: %.2f)" % (name, bench.median(blocks) / 400 * 1e6, bench.median(b20) / 20 * 1e6))
            e.close()
        print("%7d vehicles, %s: " % (n, "closed loop" if closed else "open loop  ") + " | ".join(row), flush=True)
