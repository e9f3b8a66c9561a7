"""PCIe-inclusive rates (never bench.py's `value`): (1) the configs[4] shape -- 4K gray frames pushed from host memory
through the pinned double-buffered ingest, reduced to 350 cell-averaged (u,v) per pair, k=8 Lloyd over them; (2) the
per-frame drop-in ComputeOpticalFLow.compute() at 1080p (BGR frame up, BGR visualisation down, synchronous)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowclustering_amd import synth
from opticalflowclustering_amd.cluster import KMeans
from opticalflowclustering_amd.computeOpticalFlowModule import ComputeOpticalFLow
from opticalflowclustering_amd.stream import FlowStream

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H = 3840, 2160
p = synth.texture_params(0)
base = [synth.frame(W, H, 0.9 * t, -0.5 * t, p).astype(np.uint8) for t in range(8)]
fs = FlowStream(W, H, batch_pairs=8)
for rep in range(2):                       # first pass warms up allocations
    t0 = time.perf_counter()
    for t in range(n):
        fs.push(base[t % 8])
    cells = fs.finish()
    t1 = time.perf_counter()
    X = cells.reshape(-1, 2)
    km = KMeans(n_clusters=8, init="seeded-rows", random_state=0).fit(X)
    t2 = time.perf_counter()
print("cfg4 shape: %d 4K frames -> %d pairs: ingest+flow+cells %.1f ms/frame (%.0f Mpixels/s, PCIe-inclusive), "
      "k=8 Lloyd over %d cell vectors %.1f ms (%d iterations)" %
      (n, len(cells), (t1 - t0) / n * 1e3, (n - 1) * W * H / (t1 - t0) / 1e6, len(X), (t2 - t1) * 1e3, km.n_iter_))

W, H = 1920, 1080
g = [synth.frame(W, H, 1.1 * t, -0.6 * t, p).astype(np.uint8) for t in range(6)]
frames = [np.ascontiguousarray(np.stack([a, a, a], -1)) for a in g]
cf = ComputeOpticalFLow(frames[0])
for t in range(1, 6):
    cf.compute(frames[t])
t0 = time.perf_counter()
reps = 40
for t in range(reps):
    cf.compute(frames[t % 6])
dt = (time.perf_counter() - t0) / reps
print("ComputeOpticalFLow.compute at 1080p: %.2f ms/frame (%.0f Mpixels/s, synchronous, host BGR in / BGR out)" %
      (dt * 1e3, W * H / dt / 1e6))
cf.close()
