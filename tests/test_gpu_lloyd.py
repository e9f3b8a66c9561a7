"""GPU parity of the Lloyd kernels through the C ABI: bit-exact labels against the CPU oracle AND against
the sklearn-generated goldens (fixed init), n_iter identical, centres <= 1e-9, inertia <= 1e-10 rel."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lloyd_goldens.npz"))
CASES = sorted({k.split("/")[0] for k in Z.files if "/" in k})


@pytest.fixture(scope="module")
def KMeans():
    from opticalflowclustering_amd.cluster import KMeans
    return KMeans


@pytest.mark.parametrize("name", CASES)
def test_fit_matches_sklearn_golden(KMeans, name):
    X, C0 = Z[name + "/X"], Z[name + "/C0"]
    km = KMeans(n_clusters=len(C0), init=C0, max_iter=int(Z[name + "/max_iter"]), tol=float(Z[name + "/tol"])).fit(X)
    assert km.n_iter_ == int(Z[name + "/n_iter"])
    assert np.array_equal(km.labels_, Z[name + "/labels"])
    assert np.abs(km.cluster_centers_ - Z[name + "/centers"]).max() <= 1e-9
    assert abs(km.inertia_ - float(Z[name + "/inertia"])) <= 1e-10 * float(Z[name + "/inertia"])
    assert np.array_equal(km.predict(X), Z[name + "/predict"])


@pytest.mark.parametrize("dtype,d,k,N", [(np.uint8, 4, 1, 2601), (np.uint8, 4, 3, 5852), (np.uint8, 4, 8, 23562),
                                         (np.float32, 2, 5, 100003), (np.float32, 2, 2, 7), (np.float64, 3, 4, 4099),
                                         (np.float32, 1, 3, 1000), (np.float64, 4, 12, 20000), (np.uint8, 3, 16, 9000)])
def test_fit_matches_oracle(KMeans, dtype, d, k, N):
    rng = np.random.default_rng(N + k)
    if dtype == np.uint8:
        X = rng.integers(0, 256, (N, d), dtype=np.uint8)
        X[rng.random(N) < 0.6] = 0
    else:
        cen = rng.uniform(-5, 5, (k, d))
        X = (cen[rng.integers(0, k, N)] + 0.7 * rng.standard_normal((N, d))).astype(dtype)
    C0 = X[rng.choice(N, k, replace=False)].astype(np.float64) + (np.arange(k)[:, None] * 1e-3)
    km = KMeans(n_clusters=k, init=C0).fit(X)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    assert km.n_iter_ == n_iter
    assert np.array_equal(km.labels_, lab)
    assert np.abs(km.cluster_centers_ - cen).max() <= 1e-9
    assert abs(km.inertia_ - inertia) <= 1e-10 * max(inertia, 1e-300)
    assert np.array_equal(km.predict(X), O.kmeans_predict(X, cen))


def test_two_empty_clusters_relocated(KMeans):
    rng = np.random.default_rng(7)
    X = rng.standard_normal((3000, 2))
    C0 = np.array([[0.0, 0.0], [300.0, 300.0], [-400.0, 250.0], [0.5, 0.5]])
    km = KMeans(n_clusters=4, init=C0).fit(X)
    cen, lab, _, n_iter = O.kmeans_fit(X, C0)
    assert km.n_iter_ == n_iter and np.array_equal(km.labels_, lab)
    assert np.abs(km.cluster_centers_ - cen).max() <= 1e-9
    assert np.bincount(lab, minlength=4).min() > 0


def test_duplicate_points_more_clusters_than_distinct(KMeans):
    X = np.zeros((50, 4), np.uint8)
    X[25:] = 7
    C0 = np.array([[0, 0, 0, 0], [7, 7, 7, 7], [3, 3, 3, 3.0]])
    km = KMeans(n_clusters=3, init=C0).fit(X)
    cen, lab, _, n_iter = O.kmeans_fit(X, C0)
    assert km.n_iter_ == n_iter and np.array_equal(km.labels_, lab)
    assert np.allclose(km.cluster_centers_, cen, atol=1e-12)


def test_errors(KMeans):
    with pytest.raises(ValueError):
        KMeans(n_clusters=3, init=np.zeros((3, 4))).fit(np.zeros((2, 4), np.uint8))
    with pytest.raises(ValueError):
        KMeans(n_clusters=2, init=np.zeros((3, 4))).fit(np.zeros((9, 4), np.uint8))
    from opticalflowclustering_amd._lib import OfcError
    with pytest.raises(OfcError):
        KMeans(n_clusters=17).fit(np.random.default_rng(0).random((100, 2)))     # k > 16: outside kernel range


def test_caller_data_untouched_and_seeded_init_reproducible(KMeans):
    rng = np.random.default_rng(3)
    X = rng.integers(0, 256, (4000, 4), dtype=np.uint8)
    X0 = X.copy()
    a = KMeans(n_clusters=3).fit(X)
    b = KMeans(n_clusters=3).fit(X)
    assert np.array_equal(X, X0)
    assert np.array_equal(a.labels_, b.labels_) and np.array_equal(a.cluster_centers_, b.cluster_centers_)


def test_full_size_properties_uv(KMeans):
    """one 1080p frame of (u,v) vectors (cfg2 unit): size-independent properties instead of the oracle:
    labels == argmin distance to the returned centres, centres == mean of their members, inertia consistent."""
    rng = np.random.default_rng(11)
    N = 1920 * 1080
    vel = rng.uniform(-4, 4, (5, 2))
    X = (vel[rng.integers(0, 5, N)] + 0.3 * rng.standard_normal((N, 2))).astype(np.float32)
    C0 = X[rng.choice(N, 5, replace=False)].astype(np.float64)
    km = KMeans(n_clusters=5, init=C0).fit(X)
    Xd = X.astype(np.float64)
    d2 = ((Xd[:, None, :] - km.cluster_centers_[None]) ** 2).sum(-1)
    near = d2.argmin(1)
    margin = np.sort(d2, 1)
    amb = (margin[:, 1] - margin[:, 0]) < 1e-9
    assert np.array_equal(near[~amb], km.labels_[~amb])
    for j in range(5):
        assert np.abs(Xd[km.labels_ == j].mean(0) - km.cluster_centers_[j]).max() < 1e-9 or km.n_iter_ == 300
    assert abs(d2[np.arange(N), km.labels_].sum() - km.inertia_) <= 1e-9 * km.inertia_


def test_dist_world1_allreduce():
    """RCCL plumbing with a 1-rank communicator on the one GPU of the box"""
    import ctypes as C
    from opticalflowclustering_amd import _lib
    lib = _lib.load()
    uid = np.zeros(128, np.uint8)
    _lib.check(lib.ofc_dist_unique_id(_lib.ptr(uid)))
    _lib.check(lib.ofc_dist_init(0, 0, 1, _lib.ptr(uid)))
    buf = _lib.DeviceBuffer(8 * 5).upload(np.arange(5, dtype=np.float64))
    _lib.check(lib.ofc_dist_allreduce_f64(0, C.c_void_p(buf.ptr), 5))
    assert np.array_equal(buf.download((5,), np.float64), np.arange(5.0))
    _lib.check(lib.ofc_dist_finalize())


KPP = np.load(os.path.join(os.path.dirname(__file__), "golden", "kpp_goldens.npz"))
KPP_CASES = sorted({k.split("/")[0] for k in KPP.files})


@pytest.mark.parametrize("name", KPP_CASES)
def test_kmeans_plusplus_matches_sklearn_for_a_seed(KMeans, name):
    """KMeans(n_clusters=k, init='k-means++', random_state=seed): same seed rows, labels, iteration count as sklearn"""
    from opticalflowclustering_amd.cluster import kmeans_plusplus
    from oracle import oracle as O
    X, k, seed = KPP[name + "/X"], int(KPP[name + "/k"]), int(KPP[name + "/seed"])
    _, idx = kmeans_plusplus(X, k, seed)
    assert np.array_equal(idx, KPP[name + "/indices"])
    # the device step against its CPU restatement
    import ctypes as C
    from opticalflowclustering_amd._lib import check, load, ptr
    from opticalflowclustering_amd.cluster import _DT
    mean = X.astype(np.float64).mean(axis=0)
    cand = np.ascontiguousarray(idx[:3], np.int64)
    closest = O.kpp_candidates(X, mean, idx[:1])[0][0]
    out, pots = np.empty((3, len(X))), np.empty(3)
    check(load().ofc_kpp_candidates(0, ptr(X), _DT[X.dtype], len(X), X.shape[1], ptr(mean), ptr(cand), 3, ptr(closest),
                                    ptr(out), ptr(pots)))
    want, wpots = O.kpp_candidates(X, mean, cand, closest)
    assert np.allclose(out, want, rtol=1e-12, atol=1e-9) and np.allclose(pots, wpots, rtol=1e-12)
    km = KMeans(n_clusters=k, init="k-means++", random_state=seed).fit(X)
    assert km.n_iter_ == int(KPP[name + "/n_iter"])
    assert np.array_equal(km.labels_, KPP[name + "/labels"])
    assert np.allclose(km.cluster_centers_, KPP[name + "/centers"], rtol=1e-9, atol=1e-9)


def test_fuzz_shapes_against_oracle(KMeans):
    """seeded sweep over dtype x d x k x N (incl. N < 256, N not a multiple of 4, k up to 16, clustered and uniform
    data): labels, iteration count and centres against the CPU oracle"""
    rng = np.random.default_rng(2024)
    n_checked = 0
    for trial in range(36):
        dtype = [np.uint8, np.float32, np.float64][trial % 3]
        d = int(rng.integers(1, 5))
        k = int(rng.choice([1, 2, 3, 5, 8, 9, 13, 16]))
        N = int(rng.choice([k, k + 1, 37, 255, 257, 1023, 4099, 20001]))
        N = max(N, k)
        if trial % 2:
            cen = rng.uniform(0, 200, (k, d))
            X = cen[rng.integers(0, k, N)] + rng.normal(0, 6, (N, d))
        else:
            X = rng.uniform(0, 255, (N, d))
        X = np.clip(X, 0, 255).astype(dtype)
        rows = rng.choice(N, k, replace=False)
        C0 = X[rows].astype(np.float64) + rng.normal(0, 0.5, (k, d))
        cen, lab, inertia, n_it = O.kmeans_fit(X, C0)
        km = KMeans(n_clusters=k, init=C0).fit(X)
        msg = f"trial {trial}: dtype {np.dtype(dtype).name} d {d} k {k} N {N}"
        assert km.n_iter_ == n_it, msg
        assert np.array_equal(km.labels_, lab), msg
        assert np.allclose(km.cluster_centers_, cen, rtol=1e-9, atol=1e-7), msg
        assert abs(km.inertia_ - inertia) <= 1e-9 * max(1.0, abs(inertia)), msg
        assert np.array_equal(km.predict(X), O.kmeans_predict(X, cen)), msg
        n_checked += 1
    assert n_checked == 36


def _two_rank_worker(rank, conn, name, q):
    """one of two processes sharing the GPU: its half of the rows on the device, the host-driven sharded driver, and a
    pipe as the collective"""
    import numpy as np
    from opticalflowclustering_amd import _lib
    from opticalflowclustering_amd.cluster import _DT
    from opticalflowclustering_amd.pipeline import shard_pairs
    from opticalflowclustering_amd.sharded import DeviceShard, fit_sharded
    from tests.test_dist_gloo import make_case
    X, C0 = make_case(name)
    if X.dtype not in _DT:
        X = X.astype(np.float64)
    a, b = shard_pairs(len(X), 2, rank)
    Xs = np.ascontiguousarray(X[a:b])
    buf = _lib.DeviceBuffer(Xs.nbytes, 0)
    buf.upload(Xs)
    shard = DeviceShard(buf.ptr, _DT[Xs.dtype], len(Xs), Xs.shape[1])
    fn = {"sum": np.add, "max": np.maximum, "min": np.minimum}

    def allreduce(arr, op):
        arr = np.ascontiguousarray(arr, np.float64)
        conn.send(arr)
        other = conn.recv()
        return fn[op](arr, other) if rank == 0 else fn[op](other, arr)     # same operand order on both ranks

    cen, inertia, n_iter = fit_sharded(shard, C0, allreduce=allreduce, rank=rank)
    labels = np.empty(len(Xs), np.uint8)
    _lib.check(_lib.load().ofc_memcpy_d2h(0, _lib.ptr(labels), shard.labels, len(Xs)))
    q.put((rank, cen, inertia, n_iter, labels.astype(np.int32)))


@pytest.mark.parametrize("name", ["uv", "reloc", "rgba"])
def test_two_processes_share_the_gpu_sharded_driver(name):
    """world size 2 on ONE GPU without RCCL: two processes, each with half of the rows resident on the device, run
    sharded.fit_sharded over DeviceShard (the HIP kernels) with a pipe as the all-reduce -- must equal the single-process
    oracle fit, including the relocation whose farthest sample lives on the other rank"""
    import multiprocessing as mp
    from tests.test_dist_gloo import make_case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    c0, c1 = ctx.Pipe()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, c, name, q)) for r, c in ((0, c0), (1, c1))]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    X, C0 = make_case(name)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    assert res[0][3] == res[1][3] == n_iter
    assert np.array_equal(np.concatenate([res[0][4], res[1][4]]), lab)
    for r in res:
        assert np.abs(r[1] - cen).max() <= 1e-9 and abs(r[2] - inertia) <= 1e-10 * inertia
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("name,world", [("uv", 2), ("reloc", 2), ("uv", 8), ("rgba", 3)])
def test_loopback_world_equals_fit_over_concatenated_copies(KMeans, name, world):
    """the N>1 control flow of the in-library driver (all-reduced column sums / N, per-iteration totals in their own
    buffer, speculative iterations behind the halt flag, the relocation exchange, all-reduced inertia) on one GPU:
    ofc_dist_loopback(W) emulates W ranks that all hold this shard, so the fit must equal a single-rank fit over W
    concatenated copies -- same labels, same iteration count, same centres, W-fold inertia"""
    from opticalflowclustering_amd._lib import check, load
    from tests.test_dist_gloo import make_case
    X, C0 = make_case(name)
    ref = KMeans(n_clusters=len(C0), init=C0).fit(np.concatenate([X] * world))
    check(load().ofc_dist_loopback(world))
    try:
        km = KMeans(n_clusters=len(C0), init=C0).fit(X)
    finally:
        check(load().ofc_dist_loopback(1))
    assert km.n_iter_ == ref.n_iter_
    assert np.array_equal(km.labels_, ref.labels_[: len(X)])
    assert np.allclose(km.cluster_centers_, ref.cluster_centers_, rtol=1e-11, atol=1e-11)
    assert abs(km.inertia_ - ref.inertia_) <= 1e-10 * ref.inertia_


def _two_rank_library_worker(rank, conn, name, split, q):
    """one of two processes sharing the GPU, IN-LIBRARY driver (ofc_kmeans_fit_dev: statistics all-reduced, iterations
    enqueued behind the halt flag, relocation exchange with owner election) over a caller-provided host transport"""
    import numpy as np
    from opticalflowclustering_amd import _lib, dist
    from opticalflowclustering_amd.cluster import _DT, kmeans_fit_dev
    from tests.test_dist_gloo import make_case
    X, C0 = make_case(name)
    if X.dtype not in _DT:
        X = X.astype(np.float64)
    cut = int(len(X) * split)
    Xs = np.ascontiguousarray(X[:cut] if rank == 0 else X[cut:])
    fn = {"sum": np.add, "max": np.maximum, "min": np.minimum}

    def allreduce(arr, op):
        conn.send(arr)
        other = conn.recv()
        return fn[op](arr, other) if rank == 0 else fn[op](other, arr)     # same operand order on both ranks

    dist.init_host(0, rank, 2, allreduce)
    buf = _lib.DeviceBuffer(max(Xs.nbytes, 8), 0)
    if len(Xs):
        buf.upload(Xs)
    lab = _lib.DeviceBuffer(max(len(Xs), 1), 0)
    cen, inertia, n_iter = kmeans_fit_dev(buf.ptr, _DT[Xs.dtype], len(Xs), Xs.shape[1], C0, labels_ptr=lab.ptr)
    labels = lab.download((len(Xs),), np.uint8) if len(Xs) else np.zeros(0, np.uint8)
    dist.finalize()
    q.put((rank, cen, inertia, n_iter, labels.astype(np.int32)))


@pytest.mark.parametrize("name,split", [("uv", 0.5), ("reloc", 0.6), ("reloc", 0.9), ("rgba", 0.37), ("uv", 1.0)])
def test_two_processes_in_library_driver_with_different_shards(name, split):
    """the in-library N > 1 loop with REAL ranks: two processes, uneven shards (one of them empty in the last case), the
    per-iteration exchange carried by ofc_dist_init_host's callback over a pipe.  'reloc' puts the farthest sample on
    rank 1 (split 0.6: owner election on a rank other than 0; split 0.9: on the small shard).  Must equal the
    single-process oracle fit: labels, iteration count, centres, inertia."""
    import multiprocessing as mp
    from tests.test_dist_gloo import make_case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    c0, c1 = ctx.Pipe()
    procs = [ctx.Process(target=_two_rank_library_worker, args=(r, c, name, split, q)) for r, c in ((0, c0), (1, c1))]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    X, C0 = make_case(name)
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0)
    assert res[0][3] == res[1][3] == n_iter
    assert np.array_equal(np.concatenate([res[0][4], res[1][4]]), lab)
    for r in res:
        assert np.abs(r[1] - cen).max() <= 1e-9 and abs(r[2] - inertia) <= 1e-10 * inertia
    assert np.array_equal(res[0][1], res[1][1])


def test_new_entry_points_reject_bad_arguments():
    """error behaviour of the round-2 exports: negative codes + message, nothing crosses the ABI as an exception"""
    import ctypes as C
    from opticalflowclustering_amd import _lib
    lib = _lib.load()
    assert lib.ofc_dist_init_host(0, 0, 2, None, None) == _lib.OFC_EINVAL
    assert lib.ofc_dist_init_host(0, 3, 2, C.c_void_p(1), None) == _lib.OFC_EINVAL
    z = np.zeros((8, 8, 5), np.float32)
    f = np.zeros((8, 8, 2), np.float32)
    out = np.zeros((8, 8, 2), np.float32)
    assert lib.ofc_flow_iterate(0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), 8, 8, 15, 0, 0, 0, _lib.ptr(out)) == _lib.OFC_EINVAL
    assert lib.ofc_flow_iterate(0, None, _lib.ptr(z), _lib.ptr(f), 8, 8, 15, 1, 0, 0, _lib.ptr(out)) == _lib.OFC_EINVAL
    # the two-iteration and three-wave kernels are built for the reference's winsize only
    assert lib.ofc_flow_iterate(0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), 8, 8, 9, 2, 1, 0, _lib.ptr(out)) == _lib.OFC_EUNSUPPORTED
    assert b"winsize" in lib.ofc_last_error()
    assert lib.ofc_flow_iterate(0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), 8, 8, 9, 1, 2, 0, _lib.ptr(out)) == _lib.OFC_EUNSUPPORTED
    ms = C.c_float()
    assert lib.ofc_bench_flow_iters(0, 64, 64, 0, 1, 0, C.byref(ms)) == _lib.OFC_EINVAL
