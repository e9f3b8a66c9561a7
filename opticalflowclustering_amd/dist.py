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


def _unique_id():
    uid = np.zeros(_lib.UNIQUE_ID_BYTES, np.uint8)
    check(load().ofc_dist_unique_id(ptr(uid)))
    return uid


def _rccl_init(device, rank, world, uid):
    check(load().ofc_dist_init(device, rank, world, ptr(np.ascontiguousarray(uid, np.uint8))))


def init_rccl(device, rank, world, broadcast_bytes, vote_min):
    """Set up the RCCL communicator on every rank, or on none.
    broadcast_bytes(np.uint8[128]) -> np.uint8[128]: rank 0's array delivered to every rank;
    vote_min(float) -> float: minimum of the argument over all ranks.
    Returns None when every rank holds a communicator, else the reason (str) -- the SAME outcome on every rank, and the
    same sequence of collectives (vote, broadcast, vote) on every rank whatever failed where: a rank that cannot load
    librccl or create an id still votes, nobody enters the broadcast or ncclCommInitRank unless all can, and a rank whose
    ncclCommInitRank failed is found by the second vote (the others then drop their communicator)."""
    uid, err = np.zeros(_lib.UNIQUE_ID_BYTES, np.uint8), None
    try:
        uid = _unique_id()              # every rank: proves librccl loads here; rank 0's id is the one that is used
    except Exception as e:              # noqa: BLE001 -- whatever it is, the other ranks must not be left in a collective
        err = e
    if vote_min(0.0 if err is not None else 1.0) < 1.0:
        return "librccl unusable on %s" % ("this rank: %s" % err if err is not None else "another rank")
    uid = np.ascontiguousarray(broadcast_bytes(uid), np.uint8)
    try:
        _rccl_init(device, rank, world, uid)
    except Exception as e:              # noqa: BLE001
        err = e
    if vote_min(0.0 if err is not None else 1.0) < 1.0:
        if err is None:
            finalize()                  # this rank did get a communicator: drop it
        return "ncclCommInitRank failed on %s" % ("this rank: %s" % err if err is not None else "another rank")
    return None


TRANSPORT = "none"     # what init_from_torch_env set up: "rccl", "gloo-host" (fallback / OFC_DIST_TRANSPORT=gloo) or "none"


def init_from_torch_env(device):
    """rendezvous through torch.distributed (gloo): returns (rank, world, barrier, allreduce_max).
    The Lloyd exchange goes over RCCL (xGMI).  If the RCCL communicator cannot be set up on some rank (or
    OFC_DIST_TRANSPORT=gloo asks for it) every rank falls back, together, to the host transport over the same gloo group
    (ofc_dist_init_host): slower per iteration, same results."""
    global TRANSPORT
    rank, world, _ = env_rank_world()
    if world == 1 and os.environ.get("OFC_FORCE_DIST") != "1":      # OFC_FORCE_DIST: rehearse the N>1 path on one GPU
        return 0, 1, (lambda: None), (lambda v: v)
    import sys
    import torch
    import torch.distributed as td
    if not td.is_initialized():
        td.init_process_group(backend="gloo", rank=rank, world_size=world)

    def bcast(uid):
        t = torch.from_numpy(uid.copy())
        td.broadcast(t, src=0)
        return t.numpy()

    ops = {"sum": td.ReduceOp.SUM, "max": td.ReduceOp.MAX, "min": td.ReduceOp.MIN}

    def gloo_allreduce(arr, op):
        t = torch.from_numpy(np.ascontiguousarray(arr, np.float64).copy())
        td.all_reduce(t, op=ops[op])
        return t.numpy()

    def vote_min(v):
        return float(gloo_allreduce(np.array([v]), "min")[0])

    if os.environ.get("OFC_DIST_TRANSPORT") == "gloo":
        why = "OFC_DIST_TRANSPORT=gloo"
    else:
        why = init_rccl(device, rank, world, bcast, vote_min)
    if why is not None:                                        # every rank took the same branch
        if rank == 0:
            print("dist: RCCL not used (%s); Lloyd exchange over the gloo host transport" % why, file=sys.stderr)
        init_host(device, rank, world, gloo_allreduce)
        TRANSPORT = "gloo-host"
    else:
        TRANSPORT = "rccl"

    def allreduce_max(v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    return rank, world, td.barrier, allreduce_max


_HOST_CB = None      # keeps the ctypes callback alive while the library holds it


def init_host(device, rank, world, allreduce):
    """transport provided by the caller: allreduce(np.float64 array, op) -> reduced array, op in {'sum','max','min'} (the
    collective signature of sharded.fit_sharded).  gloo/MPI across nodes, or a pipe between processes sharing a GPU."""
    import ctypes as C
    global _HOST_CB
    proto = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_void_p)
    names = ("sum", "max", "min")

    def cb(buf, count, op, _user):
        try:
            a = np.ctypeslib.as_array(buf, shape=(count,))
            a[:] = np.asarray(allreduce(a.copy(), names[op]), np.float64)
            return 0
        except Exception:          # never let an exception cross the C ABI
            import traceback
            traceback.print_exc()
            return 1

    _HOST_CB = proto(cb)
    check(load().ofc_dist_init_host(device, rank, world, C.cast(_HOST_CB, C.c_void_p), None))


def finalize():
    check(load().ofc_dist_finalize())
