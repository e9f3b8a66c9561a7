"""k_flow_iter against the software-pipelined experiment k_flow_iter_p (OFC_FLOW_PIPE=1, csrc/flow_experiments.hip): outputs must
be bit-identical; time per level-0 iteration over 32 resident 1080p pairs.  usage: python tools/flow_pipe_ab.py"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from opticalflowclustering_amd import stages, synth
from oracle import oracle as O
W, H = 700, 420
a, b = synth.translated_pair(W, H, 2.3, -1.1)
R0, R1 = O.polyexp(O.level_image(a, 0)), O.polyexp(O.level_image(b, 0))
rng = np.random.default_rng(0)
flow = (rng.standard_normal((H, W, 2)) * 2).astype(np.float32)
res = {}
for pipe in ("0", "1"):
    os.environ["OFC_FLOW_PIPE"] = pipe
    res[pipe] = stages.flow_iterate(R0, R1, flow, 3, mode=0)
print("pipelined == plain (bit-exact):", np.array_equal(res["0"], res["1"]), "max diff", np.abs(res["0"] - res["1"]).max())
for pipe in ("0", "1", "0", "1"):
    os.environ["OFC_FLOW_PIPE"] = pipe
    print("OFC_FLOW_PIPE=%s: %.4f ms per level-0 iteration over 32 pairs" % (pipe, stages.bench_flow_iters(1920, 1080, 32, 10, 0) / 2))
