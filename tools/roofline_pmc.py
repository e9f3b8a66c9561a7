"""workload of the roofline PMC passes: the kernels bench.py's `roofline` object names, launched exactly as bench.py
launches them (64 x 1080p images for k_polyexp<1,false>; two level-0 iterations over 32 resident 1080p pairs for
k_flow_iter<7,0>; the Lloyd sweeps -- full, pruned, metadata-building, final E-step -- over the 6.2e8 (u,v) vectors of the
300-frame bench clip at its converged centres)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from bench import CLIP_FRAMES, H, INIT, W, auto_batch
from opticalflowclustering_amd import _lib, stages
from opticalflowclustering_amd.pipeline import ClipPipeline
n_pe, n_fi, n_ll = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 3, 2)   # bench.py: 20, 10, 10
print("polyexp ms", stages.bench_polyexp(W, H, 64, n_pe, 0))
print("flow_iter ms", stages.bench_flow_iters(W, H, 32, n_fi, 0) / 2)
pipe = ClipPipeline(W, H, CLIP_FRAMES, batch_pairs=auto_batch(CLIP_FRAMES - 1), n_engines=2)
pipe.synth(0)
pipe.run_flow()
centers, _, n_iter = pipe.run_kmeans(INIT)
N = pipe.n_pairs * W * H
colsum = np.zeros(2)
_lib.check(_lib.load().ofc_lloyd_colstats_dev(0, pipe.flows.ptr, _lib.F32, N, 2, None, 0, _lib.ptr(colsum)))
for name, what in (("full", 0), ("pruned", 1), ("meta", 2), ("final", 3), ("final_pruned", 4)):
    print("lloyd", name, "ms", stages.bench_lloyd_sweep(pipe.flows.ptr, N, centers, colsum / N, what, n_ll))
pipe.close()
