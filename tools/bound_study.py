"""How much of a Lloyd iteration could tile-level distance bounds (Hamerly) skip on the bench clip?  Computes the
flow of a few 1080p frames on the GPU, then replays Lloyd in numpy (f64) recording, per iteration, the fraction of
256-point tiles whose labels are uniform and whose margin survives the accumulated centre drift."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import INIT
from opticalflowclustering_amd.pipeline import ClipPipeline

T = int(sys.argv[1]) if len(sys.argv) > 1 else 9
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
TILE = int(sys.argv[3]) if len(sys.argv) > 3 else 256
pipe = ClipPipeline(1920, 1080, T, batch_pairs=8, n_engines=1)
pipe.synth(t0)
pipe.run_flow()
X = pipe.flows_host().reshape(-1, 2).astype(np.float64)
pipe.close()
N = len(X) // TILE * TILE
X = X[:N]
X0mean = X.mean(0)
X -= X0mean
C = INIT - X0mean
k = len(C)
bound = np.full(N // TILE, -np.inf)
tlabel = np.full(N // TILE, -1)
cum = np.zeros(k)
for it in range(20):
    D = np.sqrt(((X[:, None, :] - C[None]) ** 2).sum(-1))
    lab = D.argmin(1)
    srt = np.sort(D, 1)
    marg = (srt[:, 1] - srt[:, 0]).reshape(-1, TILE).min(1)
    L = lab.reshape(-1, TILE)
    uni = (L == L[:, :1]).all(1)
    # which tiles WOULD have been skipped this iteration (decided before computing): cached bound minus drift
    safe = (tlabel >= 0) & (bound - cum[np.maximum(tlabel, 0)] > 1e-4)
    wrong = safe & (~uni | (L[:, 0] != tlabel))
    print("iter %2d uniform tiles %.4f  skippable %.4f  (violations %d)  min-margin>0.01: %.4f" %
          (it, uni.mean(), safe.mean(), wrong.sum(), (uni & (marg > 0.01)).mean()), flush=True)
    # refresh the tiles that were not skipped
    ref = ~safe
    tlabel[ref] = np.where(uni[ref], L[ref, 0], -1)
    bound[ref] = marg[ref] + cum[np.maximum(tlabel[ref], 0)]
    Cn = np.stack([X[lab == j].mean(0) if (lab == j).any() else C[j] for j in range(k)])
    delta = np.sqrt(((Cn - C) ** 2).sum(1))
    for a in range(k):
        cum[a] += delta[a] + np.delete(delta, a).max()
    shift = (delta ** 2).sum()
    C = Cn
    if shift <= 1e-4 * X.var(0).mean():
        print("converged after", it + 1, "iterations; centre drift per iteration last:", delta)
        break
