"""margin distribution of 64-point tiles at the converged centres of the full bench clip (decides what tile-level
distance bounds could skip in late Lloyd iterations)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import INIT
from opticalflowclustering_amd.pipeline import ClipPipeline

pipe = ClipPipeline(1920, 1080, 300, batch_pairs=32, n_engines=2)
pipe.synth(0)
pipe.run_flow()
centers, inertia, n_iter = pipe.run_kmeans(INIT, max_iter=300, tol=1e-4)
print("n_iter", n_iter, "centres", np.round(centers, 4).tolist())
idx = np.arange(0, 299, 23)
P = 1920 * 1080
for TILE in (64, 256):
    tot = uni = 0
    hist = np.zeros(5)
    for t in idx:
        X = pipe.sample_uv(np.arange(t * P, (t + 1) * P, dtype=np.int64)).astype(np.float64) if False else None
    break
F = pipe.flows_host()
pipe.close()
for TILE in (64, 256):
    res = []
    for t in idx:
        X = F[t].reshape(-1, 2).astype(np.float64)
        D = np.sqrt(((X[:, None, :] - centers[None]) ** 2).sum(-1))
        lab = D.argmin(1)
        srt = np.sort(D, 1)
        marg = (srt[:, 1] - srt[:, 0]).reshape(-1, TILE).min(1)
        L = lab.reshape(-1, TILE)
        u = (L == L[:, :1]).all(1)
        res.append([u.mean()] + [(u & (marg > th)).mean() for th in (0.01, 0.02, 0.05, 0.1, 0.2)])
    r = np.mean(res, 0)
    print("tile %3d: uniform %.3f | margin > 0.01: %.3f  0.02: %.3f  0.05: %.3f  0.1: %.3f  0.2: %.3f" % (TILE, *r))
