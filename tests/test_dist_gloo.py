"""The N>1 path on CPU: two gloo ranks run the host-driven sharded Lloyd driver (the product's
opticalflowclustering_amd.sharded.fit_sharded) over an ORACLE-backed shard (test-only backend; the product's
backend is DeviceShard = the HIP kernels) and must reproduce the single-process oracle fit: identical
labels and n_iter, centres <= 1e-9 -- including an iteration with an empty-cluster relocation whose farthest
sample lives on the other rank."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O
from opticalflowclustering_amd.pipeline import shard_pairs
from opticalflowclustering_amd.sharded import fit_sharded


class OracleShard:
    """shard-local passes through the CPU oracle (checker only)"""

    def __init__(self, X):
        self.X = np.ascontiguousarray(X)
        self.N, self.d = self.X.shape
        self.labels = np.full(self.N, -1, np.int32)

    def colstats(self, mean, pass_):
        Xd = self.X.astype(np.float64)
        return Xd.sum(0) if pass_ == 0 else ((Xd - mean) ** 2).sum(0)

    def step(self, mean, centers_c, accumulate=True):
        return O.lloyd_partials(self.X, mean, centers_c, self.labels)

    def inertia(self, mean, centers_c):
        Xc = self.X.astype(np.float64) - mean
        return float(((Xc - centers_c[self.labels]) ** 2).sum())

    def farthest(self, mean, centers_c, excl):
        Xc = self.X.astype(np.float64) - mean
        d2 = ((Xc - centers_c[self.labels]) ** 2).sum(1)
        d2[list(excl)] = -2
        if self.N == 0 or d2.max() < 0:
            return -1.0, -1, np.zeros(self.d), -1
        i = int(np.argmax(d2))
        return float(d2[i]), i, Xc[i], int(self.labels[i])


def make_case(name):
    rng = np.random.default_rng(5)
    if name == "uv":
        vel = rng.uniform(-4, 4, (5, 2))
        X = (vel[rng.integers(0, 5, 30000)] + 0.4 * rng.standard_normal((30000, 2))).astype(np.float32)
        return X, X[rng.choice(len(X), 5, replace=False)].astype(np.float64)
    if name == "reloc":
        X = rng.standard_normal((6000, 2))
        X[5000] = (9.0, 9.0)                                    # the farthest sample sits in rank 1's shard
        return X, np.array([[0.0, 0.0], [400.0, 400.0], [0.3, 0.3]])
    X = rng.integers(0, 256, (9000, 4), dtype=np.uint8)
    X[rng.random(9000) < 0.5] = 0
    return X, X[rng.choice(len(X), 3, replace=False)].astype(np.float64) + np.arange(3)[:, None] * 1e-3


def _worker(rank, world, port, name, q):
    import torch
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    X, C0 = make_case(name)
    a, b = shard_pairs(len(X), world, rank)
    shard = OracleShard(X[a:b])
    ops = {"sum": td.ReduceOp.SUM, "max": td.ReduceOp.MAX, "min": td.ReduceOp.MIN}

    def allreduce(arr, op):
        t = torch.from_numpy(np.ascontiguousarray(arr, np.float64).copy())
        td.all_reduce(t, op=ops[op])
        return t.numpy()

    cen, inertia, n_iter = fit_sharded(shard, C0, allreduce=allreduce, rank=rank)
    q.put((rank, cen, inertia, n_iter, shard.labels.copy()))
    td.barrier()
    td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("name", ["uv", "rgba", "reloc"])
def test_two_rank_gloo_equals_single_process_oracle(name):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, world = _free_port(), 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    X, C0 = make_case(name)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    assert res[0][3] == res[1][3] == n_iter
    assert np.array_equal(np.concatenate([res[0][4], res[1][4]]), lab)
    for r in res:
        assert np.abs(r[1] - cen).max() <= 1e-9
        assert abs(r[2] - inertia) <= 1e-10 * inertia
    assert np.array_equal(res[0][1], res[1][1])                  # every rank ends with the same centres


def test_single_shard_driver_equals_oracle():
    for name in ("uv", "rgba", "reloc"):
        X, C0 = make_case(name)
        shard = OracleShard(X)
        cen, inertia, n_iter = fit_sharded(shard, C0)
        oc, ol, oi, on = O.kmeans_fit(X, C0)
        assert n_iter == on and np.array_equal(shard.labels, ol) and np.abs(cen - oc).max() <= 1e-9


def _negotiation_worker(rank, world, port, scenario, q):
    """one rank of init_from_torch_env with the library calls replaced by stand-ins that fail where `scenario` says"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.pop("OFC_DIST_TRANSPORT", None)
    import numpy as np
    from opticalflowclustering_amd import _lib, dist
    calls = []

    def unique_id():
        calls.append("unique_id")
        if scenario == "id_fails_on_%d" % rank:
            raise _lib.OfcError(_lib.OFC_ECOMM, "librccl.so: cannot open shared object file")
        return np.full(_lib.UNIQUE_ID_BYTES, 40 + rank, np.uint8)

    def rccl_init(device, r, w, uid):
        calls.append("rccl_init:%d" % int(uid[0]))            # whose id arrived
        if scenario == "init_fails_on_%d" % rank:
            raise _lib.OfcError(_lib.OFC_ECOMM, "ncclCommInitRank failed")

    dist._unique_id, dist._rccl_init = unique_id, rccl_init
    dist.init_host = lambda device, r, w, allreduce: calls.append("init_host")
    dist.finalize = lambda: calls.append("finalize")
    r, w, barrier, allreduce_max = dist.init_from_torch_env(0)
    barrier()
    q.put((rank, dist.TRANSPORT, calls, allreduce_max(float(rank))))
    import torch.distributed as td
    td.destroy_process_group()


@pytest.mark.parametrize("scenario", ["id_fails_on_0", "id_fails_on_1", "init_fails_on_0", "init_fails_on_1", "all_fine"])
def test_rccl_setup_failure_on_one_rank_ends_on_the_host_transport_everywhere(scenario):
    """VERDICT r02 #4 / ADVICE: whichever rank cannot load librccl, create the id or join the communicator, every rank keeps
    issuing the same gloo collectives and all of them end, within seconds, on the same transport"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, world = _free_port(), 2
    procs = [ctx.Process(target=_negotiation_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = "rccl" if scenario == "all_fine" else "gloo-host"
    for rank, transport, calls, mx in res:
        assert transport == want, (rank, transport, calls)
        assert mx == 1.0                                            # the gloo group is still in step afterwards
        assert ("init_host" in calls) == (want == "gloo-host")
        if scenario.startswith("id_fails"):
            assert not any(c.startswith("rccl_init") for c in calls)     # nobody entered ncclCommInitRank
        else:
            assert "rccl_init:40" in calls                          # rank 0's id was the one delivered
    if scenario.startswith("init_fails"):
        ok_rank = 1 - int(scenario[-1])
        assert "finalize" in res[ok_rank][2] and "finalize" not in res[1 - ok_rank][2]
