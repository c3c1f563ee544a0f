"""Helper of tests/test_gpu_multirank.py (run as a child process): a one-rank RCCL process group beside a full-size
resident step grid.  The target's `--gpus 8` run has every rank call collectives (barrier, MAX of the block time,
the positions all-gather) between blocks of steps served by a resident grid that fills every wave slot of the device:
the collective's kernel must get onto the device (the grid parks when the host goes quiet or an engine entry point needs
the stream) and must not change a bit of the trajectory.  Prints one JSON line."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bench
    afa = importlib.import_module("agri-fly_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    port = int(sys.argv[2])
    torch.cuda.set_device(0)
    with bench.stdout_to_stderr():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)                       # communicator comes up here
        torch.cuda.synchronize()
        comm = afa.Comm(afa.Comm.unique_id(), 0, 1, device=0)
    xyz = torch.empty((3, n), dtype=torch.float32, device="cuda")

    def fly(interleave):
        e = bench.build_shard(afa, n, 0, n, 0)
        e.set_step_mode(afa.AFE_STEP_PERSISTENT)
        e.set_resident_queue(1)                  # the grid that survives afe_sync: the case this probe is about
        rec = {}
        for _ in range(25):
            e.step(1000, 1)
        if interleave:
            rec["resident_before_collective"] = bool(e.persistent_running)
            t0 = time.perf_counter()
            x = torch.full((8,), 3.0, device="cuda")
            dist.all_reduce(x, op=dist.ReduceOp.MAX)          # torch's stream, RCCL's kernel: beside / after the grid
            torch.cuda.synchronize()
            rec["all_reduce_ms"] = (time.perf_counter() - t0) * 1e3
            rec["resident_after_collective"] = bool(e.persistent_running)     # (with compute units reserved the grid never left)
            rec["all_reduce_ok"] = bool((x == 3.0).all().item())
            # the same with the grid kept fed, as between two blocks of a run whose host does not pause
            ts = []
            for _ in range(10):
                e.step(1000, 1)
                t0 = time.perf_counter()
                dist.all_reduce(x, op=dist.ReduceOp.MAX)
                x.cpu()                                       # waits for the collective alone (torch's stream), not for the device
                ts.append((time.perf_counter() - t0) * 1e3)
            rec["all_reduce_ms_grid_fed"] = float(np.median(ts))
            rec["resident_after_fed_collectives"] = bool(e.persistent_running)
            for _ in range(5):
                e.step(1000, 1)                               # a grid is resident again
            rec["resident_before_gather"] = bool(e.persistent_running)
            t0 = time.perf_counter()
            with bench.stdout_to_stderr():
                e.gather_positions(comm, xyz.data_ptr())      # the product's exchange on the engine's stream
            e.sync()
            rec["gather_ms"] = (time.perf_counter() - t0) * 1e3
            rec["gathered_equals_state"] = bool(np.array_equal(xyz.cpu().numpy(), e.get_state(dtype=np.float32)["pos"]))
            for _ in range(20):
                e.step(1000, 1)
        else:
            for _ in range(35):           # (25 + the ten steps of the fed-grid collectives)
                e.step(1000, 1)
        e.sync()
        st = e.get_state()
        g, a = e.get_imu()
        e.close()
        return rec, dict(st, gyro=g, acc=a)

    _, ref = fly(False)
    rec, got = fly(True)
    rec["bits_identical"] = all(np.array_equal(ref[k], got[k], equal_nan=True) for k in ref)
    rec["vehicles"] = n
    with bench.stdout_to_stderr():
        comm.close()
        dist.destroy_process_group()
    os.write(bench._REAL_STDOUT, (json.dumps(rec) + "\n").encode())


if __name__ == "__main__":
    main()
