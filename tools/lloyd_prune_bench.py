"""Lloyd over the full bench clip's (u,v) field with and without the tile sweeps (csrc/lloyd_tiles.hip): time per fit,
iterations, centres, share of tiles skipped.  usage: python tools/lloyd_prune_bench.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import INIT, auto_batch
from opticalflowclustering_amd import _lib
from opticalflowclustering_amd.cluster import prune_stats
from opticalflowclustering_amd.pipeline import ClipPipeline

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pipe = ClipPipeline(1920, 1080, T, batch_pairs=auto_batch(T - 1), n_engines=2)
pipe.synth(0)
pipe.run_flow()
lib = _lib.load()
res = {}
for policy in ("0", "1", "3", "0", "1"):
    os.environ["OFC_LLOYD_PRUNE"] = policy
    pipe.run_kmeans(INIT)
    _lib.check(lib.ofc_device_sync(0))
    t0 = time.perf_counter()
    for _ in range(3):
        cen, inertia, n_iter = pipe.run_kmeans(INIT)
    _lib.check(lib.ofc_device_sync(0))
    ms = (time.perf_counter() - t0) / 3 * 1e3
    lab = pipe.labels_host()
    print("OFC_LLOYD_PRUNE=%s: %.2f ms per fit, n_iter %d, inertia %.10e, %s" % (policy, ms, n_iter, inertia, prune_stats()), flush=True)
    if policy in res:
        assert np.array_equal(res[policy][0], cen) and np.array_equal(res[policy][1], lab), "not reproducible"
    res[policy] = (cen, lab, n_iter, inertia)
for p in ("1", "3"):
    print("policy %s vs 0: max centre diff %.3e, labels equal %s, n_iter %d / %d, inertia rel diff %.2e" % (
        p, np.abs(res[p][0] - res["0"][0]).max(), np.array_equal(res[p][1], res["0"][1]), res[p][2], res["0"][2],
        abs(res[p][3] - res["0"][3]) / res["0"][3]))
os.environ["OFC_LLOYD_PRUNE"] = "1"
os.environ["OFC_LLOYD_TRACE"] = "1"
pipe.run_kmeans(INIT)
pipe.close()

# ---- the worst case for the tile test: the same five populations with NO spatial coherence (every sample drawn from a
# random population), 2^26 samples: nothing can be skipped; what do the metadata pass and the policy cost? ----
from opticalflowclustering_amd.cluster import kmeans_fit_dev
del os.environ["OFC_LLOYD_TRACE"]
rng = np.random.default_rng(0)
N = 1 << 26
vel = rng.uniform(-4, 4, (5, 2)).astype(np.float32)
X = vel[rng.integers(0, 5, N)] + (0.3 * rng.standard_normal((N, 2))).astype(np.float32)
buf = _lib.DeviceBuffer(X.nbytes).upload(X)
lab = _lib.DeviceBuffer(N)
C0 = vel.astype(np.float64) + 0.5
for policy in ("0", "1", "0", "1"):
    os.environ["OFC_LLOYD_PRUNE"] = policy
    kmeans_fit_dev(buf.ptr, _lib.F32, N, 2, C0, labels_ptr=lab.ptr)
    _lib.check(lib.ofc_device_sync(0))
    t0 = time.perf_counter()
    for _ in range(5):
        cen, inertia, n_iter = kmeans_fit_dev(buf.ptr, _lib.F32, N, 2, C0, labels_ptr=lab.ptr)
    _lib.check(lib.ofc_device_sync(0))
    print("incoherent field, OFC_LLOYD_PRUNE=%s: %.3f ms per fit, n_iter %d, %s" % (
        policy, (time.perf_counter() - t0) / 5 * 1e3, n_iter, prune_stats()), flush=True)
