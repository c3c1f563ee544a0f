"""A stand-in rank program for the launcher test: joins a gloo group the way bench.py joins its RCCL
group (RANK / WORLD_SIZE / MASTER_* from the environment), proves the ranks really talk (all-reduce),
and prints one JSON line on rank 0.  `--fail-rank R` makes rank R exit non-zero after the group is up."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
dist.barrier()
fail = int(sys.argv[sys.argv.index("--fail-rank") + 1]) if "--fail-rank" in sys.argv else -1
if rank == fail:
    print("rank %d failing on purpose" % rank, file=sys.stderr)
    sys.exit(7)
if rank == 0:
    print("some RCCL-like banner on stdout")
    print(json.dumps({"n_ranks": world, "sum": float(t.item()), "local_rank": os.environ["LOCAL_RANK"]}))
dist.destroy_process_group()
