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
