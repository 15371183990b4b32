"""Data-parallel sharding of parallel sequences over ranks (no counterpart in the
reference, which is single device: main.cpp:526-541).

Sequences of a global fraction never interact inside forward/backward; they only meet in the
weight-gradient SUM over patterns (LstmLayer.cu:502-510, FeedForwardLayer.cu:94-100) and in the
scalar error / #correct.  So rank r packs its own fraction from sequences r, r+G, r+2G, ... of the
length-sorted global list, and one all-reduce(SUM) over the flat weightUpdates arena followed by
the identical UpdateWeightFn on every rank keeps the replicas bit-identical.
"""
import numpy as np


def shard_indices(num_sequences, world_size, rank):
    """Round-robin over the (length-sorted) sequence list keeps per-rank T nearly equal."""
    return list(range(rank, num_sequences, world_size))


def shard_sequences(inputs, targets, world_size, rank, sort_by_length=True):
    order = list(range(len(inputs)))
    if sort_by_length:
        order.sort(key=lambda i: inputs[i].shape[0])
    mine = [order[i] for i in shard_indices(len(order), world_size, rank)]
    return [inputs[i] for i in mine], [targets[i] for i in mine]


def allreduce_sum_(flat, dist=None):
    """In-place SUM all-reduce of a flat gradient tensor (torch tensor, CPU/gloo or GPU/RCCL)."""
    if dist is None:
        import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def flatten_updates(layers):
    """Concatenate per-layer weightUpdates in layer order, padded to 4 floats per layer like the
    device arena (cn_api.cpp finalize())."""
    parts = []
    for g in layers:
        g = np.asarray(g, np.float32).reshape(-1)
        pad = (-g.size) % 4
        parts.append(np.concatenate([g, np.zeros(pad, np.float32)]) if pad else g)
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)


class DeviceArray:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can alias it
    (torch.as_tensor(DeviceArray(...), device='cuda')) for torch.distributed collectives."""

    def __init__(self, ptr, count, typestr="<f4"):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2,
        }
