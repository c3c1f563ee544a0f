"""Multi-GPU host logic on CPU: shard partitioning and the position all-gather
through torch.distributed (gloo, world_size 2 and 3)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_exactly(afa):
    sh = afa.sharding
    for n in (0, 1, 7, 8, 1000, 1048576, 1048577):
        for w in (1, 2, 3, 4, 8):
            rs = sh.all_shard_ranges(n, w)
            assert rs[0][0] == 0 and sum(c for _, c in rs) == n
            for (f0, c0), (f1, _) in zip(rs, rs[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in rs) - min(c for _, c in rs) <= 1
            for v in {0, n // 3, n - 1} if n else set():
                r = sh.owner_of(v, n, w)
                assert rs[r][0] <= v < rs[r][0] + rs[r][1]
    with pytest.raises(ValueError):
        sh.shard_range(10, 2, 2)


def test_gust_scenario_shards_reproduce_unsharded_rows(afa):
    sc = afa.scenarios
    p = afa.params_from_type(5)
    n = 1000
    full = sc.gust_ensemble(n, p, seed=4)
    assert np.abs(full.ext_force[:, 0]).max() == 0.0       # sigma sweep starts at 0
    assert 0.2 < full.ext_force[:, n // 2:].std() < 0.5
    for w in (2, 3, 8):
        for r in range(w):
            first, count = afa.sharding.shard_range(n, r, w)
            part = sc.gust_ensemble(count, p, seed=4, first_global=first, n_global=n)
            np.testing.assert_array_equal(part.ext_force, full.ext_force[:, first:first + count])
            np.testing.assert_array_equal(part.motor_cmd, full.motor_cmd[:, first:first + count])


def _worker(rank, world, port, n_global, q):
    import importlib
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    afa = importlib.import_module("agri-fly_amd")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        rng = np.random.Generator(np.random.PCG64(7))
        allpos = rng.normal(size=(3, n_global)).astype(np.float32)
        first, count = afa.sharding.shard_range(n_global, rank, world)
        local = torch.from_numpy(np.ascontiguousarray(allpos[:, first:first + count]))
        got = afa.sharding.gather_positions(local, n_global).numpy()
        q.put((rank, bool(np.array_equal(got, allpos)), got.shape))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n_global", [(2, 4096), (2, 1001), (3, 1000)])
def test_gather_positions_equals_host_concatenate(world, n_global):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for _, ok, shape in results:
        assert ok and shape == (3, n_global)


def _exchange_worker(rank, world, port, counts, q):
    """one rank of the PRODUCT's exchange routine (afe_gather_exchange, the code behind afe_gather_positions and so behind
    bench.py's shared-world part) with gloo collectives on host memory in place of RCCL on device memory"""
    import importlib
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    afa = importlib.import_module("agri-fly_amd")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        n_all = int(sum(counts))
        rng = np.random.Generator(np.random.PCG64(11))
        allpos = rng.normal(size=(3, n_all)).astype(np.float32)
        first = int(sum(counts[:rank]))
        local = np.ascontiguousarray(allpos[:, first:first + counts[rank]])
        out = np.full((3, n_all), np.nan, np.float32)
        calls = {"all_gather": 0, "broadcast": 0}

        def all_gather(send, recv):
            calls["all_gather"] += 1
            dist.all_gather_into_tensor(torch.from_numpy(recv), torch.from_numpy(send.copy()))

        def broadcast(send, recv, root):
            calls["broadcast"] += 1
            t = torch.from_numpy(recv)
            if rank == root:
                t.copy_(torch.from_numpy(send.copy()))
            dist.broadcast(t, root)

        equal = len(set(counts)) == 1
        afa.gather_exchange(all_gather, broadcast, rank, world, None if equal else counts, local, out)
        # the consumer's bookkeeping on top: this shard's vehicles are global indices first .. first + count - 1
        d, i = afa.sharding.nearest_neighbour_reference(out, first, min(counts[rank], 8))
        dr, ir = afa.sharding.nearest_neighbour_reference(allpos, first, min(counts[rank], 8))
        q.put((rank, bool(np.array_equal(out, allpos)), calls, bool(np.array_equal(i, ir) and np.array_equal(d, dr))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(2048, 2048), (334, 333, 333), (5, 1000, 64), (1, 1)])
def test_the_products_exchange_routine_over_gloo(counts):
    """afe_gather_exchange -- counts, offsets, all-gather for equal shards, one broadcast per rank and component for
    unequal ones -- is what afe_gather_positions runs over RCCL; here every rank runs the same C routine over gloo.
    World sizes 2 and 3, equal, nearly equal and wildly unequal shards."""
    import torch.multiprocessing as mp
    world = len(counts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, list(counts), q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    equal = len(set(counts)) == 1
    for _, ok, calls, nn_ok in results:
        assert ok and nn_ok
        assert calls == ({"all_gather": 3, "broadcast": 0} if equal else {"all_gather": 0, "broadcast": 3 * world})


def test_exchange_routine_refuses_inconsistent_counts(afa):
    out = np.zeros((3, 10), np.float32)
    local = np.zeros((3, 4), np.float32)
    noop = lambda *a: None
    with pytest.raises(afa.AfeError):
        afa.gather_exchange(noop, noop, 0, 2, [5, 5], local, out)          # counts[rank] != n_local
    with pytest.raises(afa.AfeError):
        afa.gather_exchange(noop, noop, 0, 2, [4, 0], local, out)          # an empty shard
    with pytest.raises(afa.AfeError):
        afa.gather_exchange(noop, noop, 2, 2, None, local, out)            # rank outside the communicator


def test_nearest_neighbour_reference_helper(afa):
    xyz = np.array([[0, 1, 5, 5.5], [0, 0, 0, 0], [0, 0, 0, 0]], np.float32)
    d, i = afa.sharding.nearest_neighbour_reference(xyz, 0, 4)
    assert i.tolist() == [1, 0, 3, 2]
    np.testing.assert_allclose(d, [1, 1, 0.25, 0.25])


def test_launcher_starts_ranks_relays_rank0_and_propagates_failure(afa):
    """bench.py --gpus N without torch.distributed.run: agri-fly_amd/launch.py starts the ranks itself.
    World size 2 over gloo here; the rank program is tests/rank_probe.py."""
    import importlib
    import json
    launch = importlib.import_module("agri-fly_amd.launch")
    probe = os.path.join(ROOT, "tests", "rank_probe.py")
    line = launch.launch_ranks(probe, [], 2)
    rec = json.loads(line)
    assert rec == {"n_ranks": 2, "sum": 3.0, "local_rank": "0"}
    with pytest.raises(SystemExit) as ei:
        launch.launch_ranks(probe, ["--fail-rank", "1"], 2)
    assert "rank 1 exited with status 7" in str(ei.value)


def test_bench_parent_stays_off_the_gpu_when_it_launches():
    """the self-launch branch of bench.py runs before torch or the engine library are imported"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args)") < main.index("import torch")
    assert main.index("launch_ranks(args)") < main.index('importlib.import_module("agri-fly_amd")')
    top = src[:src.index("def build_shard")]
    assert "import torch" not in top
    launch_src = open(os.path.join(ROOT, "agri-fly_amd", "launch.py")).read()
    assert "import torch" not in launch_src and "hip" not in launch_src.replace("HIP call", "")


class _StubEngine:
    """what bench.time_steps needs of an engine: step and sync; a rank-dependent cost per step"""

    def __init__(self, seconds_per_step):
        self.cost, self.steps, self.syncs = seconds_per_step, 0, 0

    def step(self, dt_us, k):
        import time
        time.sleep(self.cost * k)
        self.steps += k

    def sync(self):
        self.syncs += 1


def _bench_flow_worker(rank, world, port, q):
    """bench.py's timing protocol for N > 1 -- time_steps / timed_blocks with a barrier and a MAX over ranks -- on a stub
    engine over gloo: every rank must run the same number of blocks (or a barrier never completes) and see the same,
    slowest-rank, time for each"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        e = _StubEngine(0.0002 * (1 + 4 * rank))       # rank 1 is five times slower
        own = []
        blocks = bench.timed_blocks(e, 5, 1, lambda: None, dist.barrier, reduce_max, min_total_s=0.03, min_blocks=3, own=own, settle_s=0.0)
        # a second protocol on the same group right behind it: nobody is left behind in a collective of the first
        blocks2 = bench.timed_blocks(e, 2, 2, lambda: None, dist.barrier, reduce_max, min_total_s=0.0, min_blocks=2, settle_s=0.0)
        q.put((rank, blocks, own, blocks2, e.steps))
    finally:
        dist.destroy_process_group()


def test_bench_timing_protocol_with_two_ranks_over_gloo():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_flow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, own0, b20, steps0), (_, b1, own1, b21, steps1) = results
    assert b0 == b1 and b20 == b21                     # the reduced times are the same numbers on every rank ...
    assert len(own0) == len(own1) == len(b0) >= 3      # ... so both ran the same number of blocks
    assert steps0 == steps1 == 5 * len(b0) + 2 * len(b20)
    for t, o0, o1 in zip(b0, own0, own1):
        assert t == max(o0, o1)                        # MAX over ranks, block by block
    assert sum(own1) > 2 * sum(own0)                   # and it is the slow rank's time that counts
    assert sum(b0) >= 0.03 or len(b0) == 2000
