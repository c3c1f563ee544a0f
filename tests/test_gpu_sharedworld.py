"""Shared-world query path on the GPU: afe_pack_positions, the RCCL all-gather
(world size 1 here; the multi-rank logic is covered by the gloo tests) and
afe_nearest_neighbour, with torch only as the device-buffer / communicator
plumbing.  Needs an MI355X."""
import os
import socket

import numpy as np
import pytest

from tests.scenarios import afa, random_ensemble

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("precision", [afa.AFE_F32, afa.AFE_F64])
def test_pack_positions_and_nearest_neighbour(precision):
    import torch
    n = 3000
    ens = random_ensemble(n, seed=21)
    with ens.to_engine(precision) as e:
        xyz = torch.empty((3, n), dtype=torch.float32, device="cuda")
        e.pack_positions(xyz.data_ptr())
        e.sync()
        np.testing.assert_array_equal(xyz.cpu().numpy(), ens.data.pos.astype(np.float32))
        # pretend this shard is vehicles [first, first+n) of a bigger gathered world
        first = 0
        d2 = torch.empty(n, dtype=torch.float32, device="cuda")
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr())
        e.sync()
        ref_d, ref_i = afa.sharding.nearest_neighbour_reference(ens.data.pos.astype(np.float32), first, n)
        np.testing.assert_array_equal(idx.cpu().numpy(), ref_i)
        np.testing.assert_allclose(d2.cpu().numpy(), ref_d, rtol=1e-5)


def test_nearest_neighbour_in_a_sharded_world():
    """shard = the middle third of a gathered ensemble: self-exclusion must use
    the GLOBAL index (first_global_index + i)"""
    import torch
    n_all, first, n = 3000, 1000, 1000
    ens = random_ensemble(n_all, seed=22)
    world = torch.from_numpy(ens.data.pos.astype(np.float32)).cuda().contiguous()
    part = afa.scenarios.EnsembleData(n)
    part.pos = np.ascontiguousarray(ens.data.pos[:, first:first + n])
    with afa.Ensemble(n, first_global_index=first) as e:
        e.set_type_table([afa.params_from_type(5)])
        e.set_state(pos=part.pos)
        d2 = torch.empty(n, dtype=torch.float32, device="cuda")
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        e.nearest_neighbour(world.data_ptr(), n_all, d2.data_ptr(), idx.data_ptr())
        e.sync()
    ref_d, ref_i = afa.sharding.nearest_neighbour_reference(ens.data.pos.astype(np.float32), first, n)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_i)


def test_rccl_gather_positions_world_size_1():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 4096
        ens = random_ensemble(n, seed=23)
        with ens.to_engine(afa.AFE_F32) as e:
            local = torch.empty((3, n), dtype=torch.float32, device="cuda")
            e.pack_positions(local.data_ptr())
            e.sync()
            allpos = afa.sharding.gather_positions(local, n)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(allpos.cpu().numpy(), ens.data.pos.astype(np.float32))
    finally:
        dist.destroy_process_group()


def test_device_view_matches_host_getters():
    """zero-copy view: torch can wrap the engine's slabs without a copy"""
    import ctypes
    import torch
    n = 1000
    ens = random_ensemble(n, seed=24)
    with ens.to_engine(afa.AFE_F32) as e:
        v = e.device_view()
        assert v.n_vehicles == n and v.stride == 1280 and v.state_elem_size == 4   # 256 x odd

        class _Wrap:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = dict(shape=shape, typestr="<f4", data=(ptr, False), version=2,
                                                     strides=None)
        pos = torch.as_tensor(_Wrap(v.pos, (3, v.stride)), device="cuda")[:, :n]
        np.testing.assert_array_equal(pos.cpu().numpy(), e.get_state(dtype=np.float32)["pos"])
        e.step(1000, 3)
        e.sync()
        np.testing.assert_array_equal(pos.cpu().numpy(), e.get_state(dtype=np.float32)["pos"])
