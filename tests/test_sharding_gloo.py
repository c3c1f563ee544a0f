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
