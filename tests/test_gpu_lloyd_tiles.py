"""GPU parity of the tile-pruned Lloyd sweeps (csrc/lloyd_tiles.hip): a fit whose label-less iterations skip the tiles that
lie inside one Voronoi cell must be the fit sklearn computes (_kmeans.py:624-752) -- labels bit-exact, n_iter equal,
centres <= 1e-9, inertia <= 1e-10 relative against the CPU oracle, the sklearn goldens and the unpruned kernels."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lloyd_goldens.npz"))
UV_CASES = [c for c in sorted({k.split("/")[0] for k in Z.files if "/" in k})
            if Z[c + "/X"].dtype == np.float32 and Z[c + "/X"].shape[1] == 2 and len(Z[c + "/C0"]) <= 8]


class prune:
    """OFC_LLOYD_PRUNE for the fits inside the block (read by the library at every fit)"""

    def __init__(self, v):
        self.v = str(v)

    def __enter__(self):
        self.old = os.environ.get("OFC_LLOYD_PRUNE")
        os.environ["OFC_LLOYD_PRUNE"] = self.v

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("OFC_LLOYD_PRUNE", None)
        else:
            os.environ["OFC_LLOYD_PRUNE"] = self.old


def coherent_uv(N, k=5, run=700, noise=0.05, seed=0):
    """(u,v) vectors as a flow field has them: long runs of consecutive samples belong to one motion population"""
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-4, 4, (k, 2))
    pop = np.repeat(rng.integers(0, k, N // run + 2), run)[:N]
    return (vel[pop] + noise * rng.standard_normal((N, 2))).astype(np.float32), vel


def fit(X, C0, **kw):
    from opticalflowclustering_amd.cluster import KMeans, prune_stats
    km = KMeans(n_clusters=len(C0), init=C0, **kw).fit(X)
    return km, prune_stats()


def same_fit(km, cen, lab, inertia, n_iter):
    assert km.n_iter_ == n_iter
    assert np.array_equal(km.labels_, lab)
    assert np.abs(km.cluster_centers_ - cen).max() <= 1e-9
    assert abs(km.inertia_ - inertia) <= 1e-10 * max(inertia, 1e-300)


@pytest.mark.parametrize("policy", [2, 3])
@pytest.mark.parametrize("name", UV_CASES)
def test_sklearn_goldens_with_tile_sweeps(name, policy):
    """the sklearn-generated (u,v) goldens (incl. the 3-empty-cluster relocation case), tile sweeps on for any N"""
    X, C0 = Z[name + "/X"], Z[name + "/C0"]
    with prune(policy):
        km, st = fit(X, C0, max_iter=int(Z[name + "/max_iter"]), tol=float(Z[name + "/tol"]))
    same_fit(km, Z[name + "/centers"], Z[name + "/labels"], float(Z[name + "/inertia"]), int(Z[name + "/n_iter"]))
    assert st["tile_sweeps"] >= 1 or "reloc" in name      # (an iteration that stalls on an empty cluster is not counted)


@pytest.mark.parametrize("N", [64 * 4096 + 37, 64 * 64 * 3, 200_003, 64, 63 + 64, 5])
def test_coherent_field_pruned_equals_oracle_and_unpruned(N):
    X, vel = coherent_uv(N, seed=N)
    C0 = vel + 0.4 * np.random.default_rng(1).standard_normal(vel.shape)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    with prune(0):
        plain, st0 = fit(X, C0)
    with prune(2):
        tiled, st2 = fit(X, C0)
    with prune(3):
        forced, st3 = fit(X, C0)
    assert st0["tile_sweeps"] == 0
    for km in (plain, tiled, forced):
        same_fit(km, cen, lab, inertia, n_iter)
    assert np.abs(tiled.cluster_centers_ - plain.cluster_centers_).max() <= 1e-12
    if N >= 64 * 64:
        assert st2["tile_sweeps"] == n_iter, st2          # every label-less iteration
        if n_iter > 1:
            assert st2["pruned_sweeps"] >= 1 and st2["skip_fraction"] > 0.5 and st2["final_pruned"], st2
            assert st3["pruned_sweeps"] == st3["tile_sweeps"] and st3["final_pruned"], st3    # iteration 0 included


def test_incoherent_field_switches_pruning_off_and_stays_exact():
    """white-noise order (no two neighbours in the same population): almost no tile passes the box test; the device-side
    policy must leave the sweeps full, and forcing them to run pruned must still give the oracle's fit"""
    rng = np.random.default_rng(5)
    N = 300_000
    vel = rng.uniform(-4, 4, (5, 2))
    X = (vel[rng.integers(0, 5, N)] + 0.3 * rng.standard_normal((N, 2))).astype(np.float32)
    C0 = X[rng.choice(N, 5, replace=False)].astype(np.float64)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    with prune(2):
        km, st = fit(X, C0)
    same_fit(km, cen, lab, inertia, n_iter)
    assert st["pruned_sweeps"] == 0 and st["tile_sweeps"] >= 1 and not st["final_pruned"], st
    with prune(3):
        km, st = fit(X, C0)
    same_fit(km, cen, lab, inertia, n_iter)
    assert st["pruned_sweeps"] >= 1 and st["skip_fraction"] < 0.05, st


def test_tiles_on_a_voronoi_edge_are_walked():
    """two populations whose boundary runs through every tile's box: centres at (+-1, 0), samples alternate around u = 0
    at distance 1e-7..1e-3 -- the box test must reject those tiles (margin), labels stay the per-sample argmin"""
    rng = np.random.default_rng(9)
    N = 64 * 2048
    u = (rng.choice([-1.0, 1.0], N) * 10.0 ** rng.uniform(-7, -3, N))
    X = np.stack([u, rng.uniform(-1, 1, N)], 1).astype(np.float32)
    X[: N // 2, 0] += 1.0      # first half: clearly population 1 -> pure tiles; second half straddles the edge
    X[: N // 4, 0] -= 2.0      # first quarter: clearly population 0
    C0 = np.array([[-1.0, 0.0], [1.0, 0.0]])
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    with prune(3):
        km, st = fit(X, C0)
    same_fit(km, cen, lab, inertia, n_iter)
    assert 0.2 < st["skip_fraction"] < 1.0, st


def test_relocation_with_tile_sweeps():
    """an empty cluster met by a pruned sweep: the fit hands over to the labelled sweeps and relocates as sklearn does"""
    X, vel = coherent_uv(64 * 1500 + 11, k=3, seed=3)
    C0 = np.concatenate([vel, [[300.0, 300.0], [-250.0, 400.0]]])
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    for policy in (2, 3):
        with prune(policy):
            km, st = fit(X, C0)
        same_fit(km, cen, lab, inertia, n_iter)
    assert np.bincount(lab, minlength=5).min() > 0


@pytest.mark.parametrize("world", [2, 8])
def test_loopback_world_with_tile_sweeps(world):
    """the all-reduced tile counts drive the same mode decision on every (emulated) rank"""
    from opticalflowclustering_amd._lib import check, load
    X, vel = coherent_uv(64 * 900 + 5, seed=12)
    C0 = vel + 0.3
    with prune(2):
        ref, _ = fit(np.concatenate([X] * world), C0)
        check(load().ofc_dist_loopback(world))
        try:
            km, st = fit(X, C0)
        finally:
            check(load().ofc_dist_loopback(1))
    assert km.n_iter_ == ref.n_iter_ and np.array_equal(km.labels_, ref.labels_[: len(X)])
    assert np.allclose(km.cluster_centers_, ref.cluster_centers_, rtol=1e-11, atol=1e-11)
    assert st["pruned_sweeps"] >= 1


def test_tight_clusters_inertia_of_skipped_tiles():
    """populations far apart, noise 1e-4: the inertia of a tile that is not read comes from its scatter about its own mean
    plus 64 |mean - c|^2 (no cancellation), and must still meet the 1e-10 relative bar"""
    X, vel = coherent_uv(64 * 2500 + 3, k=4, run=900, noise=1e-4, seed=21)
    C0 = vel + 0.2
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    with prune(3):
        km, st = fit(X, C0)
    same_fit(km, cen, lab, inertia, n_iter)
    assert st["final_pruned"] and st["skip_fraction"] > 0.8, st


def test_sweep_bench_hook_runs_every_form():
    from opticalflowclustering_amd import _lib, stages
    X, vel = coherent_uv(64 * 3000, seed=2)
    buf = _lib.DeviceBuffer(X.nbytes).upload(X)
    mean = X.astype(np.float64).mean(0)
    for what in range(5):
        assert stages.bench_lloyd_sweep(buf.ptr, len(X), vel, mean, what, iters=2) > 0
    buf.free()


def _tiles_two_rank_worker(rank, conn, split, q):
    """one of two processes sharing the GPU: the in-library driver with tile sweeps on, its own (uneven) shard of a coherent
    field, the per-iteration exchange over a pipe (ofc_dist_init_host)"""
    import numpy as np
    from opticalflowclustering_amd import _lib, dist
    from opticalflowclustering_amd.cluster import kmeans_fit_dev, prune_stats
    X, vel = coherent_uv(64 * 2200 + 29, seed=31)
    cut = int(len(X) * split)
    Xs = np.ascontiguousarray(X[:cut] if rank == 0 else X[cut:])
    fn = {"sum": np.add, "max": np.maximum, "min": np.minimum}

    def allreduce(arr, op):
        conn.send(arr)
        other = conn.recv()
        return fn[op](arr, other) if rank == 0 else fn[op](other, arr)

    dist.init_host(0, rank, 2, allreduce)
    buf = _lib.DeviceBuffer(Xs.nbytes, 0).upload(Xs)
    lab = _lib.DeviceBuffer(len(Xs), 0)
    cen, inertia, n_iter = kmeans_fit_dev(buf.ptr, _lib.F32, len(Xs), 2, vel + 0.35, labels_ptr=lab.ptr)
    labels = lab.download((len(Xs),), np.uint8)
    st = prune_stats()
    dist.finalize()
    q.put((rank, cen, inertia, n_iter, labels.astype(np.int32), st))


@pytest.mark.parametrize("split", [0.5, 0.23])
def test_two_real_ranks_with_tile_sweeps(split, monkeypatch):
    """two processes, different shards, every label-less iteration and the final E-step pruned on both: the all-reduced
    records (incl. the tile counters) must give the single-process oracle fit"""
    import multiprocessing as mp
    monkeypatch.setenv("OFC_LLOYD_PRUNE", "2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    c0, c1 = ctx.Pipe()
    procs = [ctx.Process(target=_tiles_two_rank_worker, args=(r, c, split, q)) for r, c in ((0, c0), (1, c1))]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    X, vel = coherent_uv(64 * 2200 + 29, seed=31)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, vel + 0.35)
    assert res[0][3] == res[1][3] == n_iter
    assert np.array_equal(np.concatenate([res[0][4], res[1][4]]), lab)
    for r in res:
        assert np.abs(r[1] - cen).max() <= 1e-9 and abs(r[2] - inertia) <= 1e-10 * inertia
        assert r[5]["pruned_sweeps"] >= 1 and r[5]["final_pruned"], r[5]
    assert np.array_equal(res[0][1], res[1][1])


def test_prune_stats_rejects_null():
    from opticalflowclustering_amd import _lib
    assert _lib.load().ofc_lloyd_prune_stats(0, None) == _lib.OFC_EINVAL
