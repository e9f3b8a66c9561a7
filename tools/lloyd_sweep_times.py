"""ms per launch of the five Lloyd sweep forms over the bench clip's resident (u,v) field (ofc_bench_lloyd_sweep)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CLIP_FRAMES, H, INIT, W, auto_batch
from opticalflowclustering_amd import _lib, stages
from opticalflowclustering_amd.pipeline import ClipPipeline

pipe = ClipPipeline(W, H, CLIP_FRAMES, batch_pairs=auto_batch(CLIP_FRAMES - 1), n_engines=2)
pipe.synth(0)
pipe.run_flow()
centers, _, n_iter = pipe.run_kmeans(INIT)
N = pipe.n_pairs * W * H
colsum = np.zeros(2)
_lib.check(_lib.load().ofc_lloyd_colstats_dev(0, pipe.flows.ptr, _lib.F32, N, 2, None, 0, _lib.ptr(colsum)))
for rep in range(2):
    print(" | ".join("%s %.3f" % (name, stages.bench_lloyd_sweep(pipe.flows.ptr, N, centers, colsum / N, what, 10))
                     for name, what in (("full", 0), ("pruned", 1), ("meta", 2), ("final", 3), ("final_pruned", 4), ("tiles_full", 5))))
pipe.close()
