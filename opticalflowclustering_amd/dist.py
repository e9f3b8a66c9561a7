"""Multi-GPU plumbing: one process per GPU (launched by torch.distributed.run), RCCL over xGMI for
the one exchange step of the path (the per-iteration all-reduce of the Lloyd partial sums).

torch.distributed is used only as the rendezvous: it carries the 128-byte RCCL unique id from rank 0
to the others and provides barrier / max-over-ranks for timing.  The data path never touches torch:
libofc opens its own RCCL communicator (ofc_dist_init) and issues ncclAllReduce on its own stream."""
import os

import numpy as np

from . import _lib
from ._lib import check, load, ptr


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_rccl(device, rank, world, broadcast_bytes):
    """broadcast_bytes(np.uint8[128]) -> np.uint8[128]: rank 0's id delivered to every rank"""
    uid = np.zeros(_lib.UNIQUE_ID_BYTES, np.uint8)
    if rank == 0:
        check(load().ofc_dist_unique_id(ptr(uid)))
    uid = np.ascontiguousarray(broadcast_bytes(uid), np.uint8)
    check(load().ofc_dist_init(device, rank, world, ptr(uid)))


def init_from_torch_env(device):
    """rendezvous through torch.distributed (gloo): returns (rank, world, barrier, allreduce_max)"""
    rank, world, _ = env_rank_world()
    if world == 1 and os.environ.get("OFC_FORCE_DIST") != "1":      # OFC_FORCE_DIST: rehearse the N>1 path on one GPU
        return 0, 1, (lambda: None), (lambda v: v)
    import torch
    import torch.distributed as td
    if not td.is_initialized():
        td.init_process_group(backend="gloo", rank=rank, world_size=world)

    def bcast(uid):
        t = torch.from_numpy(uid.copy())
        td.broadcast(t, src=0)
        return t.numpy()

    init_rccl(device, rank, world, bcast)

    def allreduce_max(v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    return rank, world, td.barrier, allreduce_max


def finalize():
    check(load().ofc_dist_finalize())
