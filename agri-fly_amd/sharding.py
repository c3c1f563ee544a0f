"""Ensemble sharding over the GPUs of one node (one process per GPU).

The vehicle step has no inter-vehicle term (reference Quadcopter_T.cpp:85-203
reads only its own object; the multi-vehicle loop is a plain ``for``,
AIFS_ROS/hiperlab_rostools/src/Simulator/main.cpp:323-325), so ranks own
contiguous blocks of vehicles and step them with NO collective.  The only
exchange is the shared-world query (the reference's analogue is
UWBNetwork::Run reading other vehicles' true positions, UWBNetwork.cpp:54-84):
an all-gather of fp32 positions, 12 B/vehicle, at query cadence.

torch.distributed is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.
"""
import numpy as np


def shard_range(n_global, rank, world_size):
    """Contiguous block [first, first+count) of rank; sizes differ by <= 1."""
    if world_size < 1 or not (0 <= rank < world_size) or n_global < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_global, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def all_shard_ranges(n_global, world_size):
    return [shard_range(n_global, r, world_size) for r in range(world_size)]


def owner_of(vehicle, n_global, world_size):
    base, rem = divmod(n_global, world_size)
    edge = rem * (base + 1)
    if vehicle < edge:
        return vehicle // (base + 1)
    return rem + (vehicle - edge) // base


def gather_positions(local_xyz, n_global, group=None):
    """All-gather planar fp32 positions.

    local_xyz: torch tensor [3, count_local] (cuda for RCCL, cpu for gloo).
    Returns [3, n_global] on the same device, vehicles in global order.  Uneven
    shards are padded to the largest shard for the collective (one
    all_gather_into_tensor = one RCCL all-gather; at 1 M vehicles on 8 GPUs
    each rank contributes 1.5 MB).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    ranges = all_shard_ranges(n_global, world)
    max_count = max(c for _, c in ranges)
    count = local_xyz.shape[1]
    send = local_xyz
    if count != max_count:
        send = torch.zeros((3, max_count), dtype=local_xyz.dtype, device=local_xyz.device)
        send[:, :count] = local_xyz
    send = send.contiguous().view(-1)
    recv = torch.empty(world * 3 * max_count, dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, 3, max_count)
    if all(c == max_count for _, c in ranges):
        return recv.permute(1, 0, 2).reshape(3, n_global).contiguous()
    out = torch.empty((3, n_global), dtype=send.dtype, device=send.device)
    for r, (first, c) in enumerate(ranges):
        out[:, first:first + c] = recv[r, :, :c]
    return out


def nearest_neighbour_reference(all_xyz, first, count):
    """numpy brute force used by the tests to check afe_nearest_neighbour."""
    a = np.asarray(all_xyz, dtype=np.float32)
    out_d = np.empty(count, np.float32)
    out_i = np.empty(count, np.int32)
    for k in range(count):
        d = a - a[:, first + k:first + k + 1]
        d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
        d2[first + k] = np.inf
        out_i[k] = int(np.argmin(d2))
        out_d[k] = d2[out_i[k]]
    return out_d, out_i
