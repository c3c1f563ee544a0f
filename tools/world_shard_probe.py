"""Times the neighbour query of ONE shard against a gathered ensemble of 8 shards' worth of positions (the 8-GPU
weak-scaling shape, emulated on one device): AGRIFLY_ENGINE_LIB=<variant.so> python tools/world_shard_probe.py [steps]"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
afa = importlib.import_module("agri-fly_amd")
import bench

shards, n = 8, 1 << 20
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3800
big = bench.build_shard(afa, shards * n, 0, shards * n, 0)          # everybody, stepped as one ensemble
done = 0
while done < steps:
    k = min(50, steps - done)
    big.step(1000, k)
    done += k
xyz = torch.empty((3, shards * n), dtype=torch.float32, device="cuda")
big.pack_positions(xyz.data_ptr())
big.sync()
big.close()
print("world after %d steps packed" % steps, flush=True)
for rank in (0, 3):
    e = afa.Ensemble(n, precision=afa.AFE_F32, first_global_index=rank * n)
    d2 = torch.empty(n, dtype=torch.float32, device="cuda"); idx = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for refresh in (1, 16):
        e.set_neighbour_grid_refresh(refresh)
        for _ in range(3): e.nearest_neighbour(xyz.data_ptr(), shards * n, d2.data_ptr(), idx.data_ptr())
        e.sync()
        a, b = e.event(), e.event(); e.record(a)
        for _ in range(16): e.nearest_neighbour(xyz.data_ptr(), shards * n, d2.data_ptr(), idx.data_ptr())
        e.record(b); ms = e.elapsed_ms(a, b) / 16
        print("shard %d of %d (2^20 queries among %d positions), grid re-shaped every %2d queries: %.3f ms per query  %s"
              % (rank, shards, shards * n, refresh, ms, e.neighbour_grid_info()), flush=True)
    # the answers against the brute force on a subsample
    q = torch.randperm(n, device="cuda")[:2048].to(torch.int32).contiguous()
    bd = torch.full((n,), -1.0, dtype=torch.float32, device="cuda"); bi = torch.full((n,), -2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()   # q, bd, bi were filled on torch's stream; the engine works on its own
    e.nearest_neighbour_bruteforce(xyz.data_ptr(), shards * n, q.data_ptr(), q.numel(), bd.data_ptr(), bi.data_ptr())
    e.sync()
    ql = q.long()
    print("  2048-query subsample vs brute force: indices equal %s, distances equal %s"
          % (bool((bi[ql] == idx[ql]).all().item()), bool((bd[ql] == d2[ql]).all().item())))
    e.close()
