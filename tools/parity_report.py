"""prints GPU-vs-oracle flow parity numbers (relative L2 and max abs) for a few synthetic cases"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowclustering_amd import synth
from opticalflowclustering_amd.flow import FlowEngine
from oracle import oracle as O

cases = [("translate(1.5,-0.75) 1080p", 1920, 1080, lambda W, H: synth.translated_pair(W, H, 1.5, -0.75)),
         ("translate(4,2.5) 720p", 1280, 720, lambda W, H: synth.translated_pair(W, H, 4.0, 2.5)),
         ("non-rigid 1080p", 1920, 1080, lambda W, H: synth.nonrigid_pair(W, H)[:2]),
         ("white noise 512", 512, 512, lambda W, H: synth.noise_pair(W, H)),
         ("flat bright + faint texture", 960, 540, lambda W, H: tuple(np.clip(248.0 + (f.astype(np.float64) - 127.5) * 0.04, 0, 255).astype(np.uint8)
                                                                       for f in synth.translated_pair(W, H, 2.0, 1.0))),
         ("dark, low contrast", 960, 540, lambda W, H: tuple((f // 16).astype(np.uint8) for f in synth.translated_pair(W, H, -1.0, 0.5))),
         ("half black / half texture", 960, 540, lambda W, H: tuple(np.where(np.arange(W)[None, :] < W // 2, 0, f).astype(np.uint8)
                                                                     for f in synth.translated_pair(W, H, 3.0, -2.0)))]
def bench_clip_pair(W, H):
    from opticalflowclustering_amd.pipeline import ClipPipeline
    pipe = ClipPipeline(W, H, 2, batch_pairs=1, n_engines=1)
    pipe.synth(t0=7, seed=0)
    fr = pipe.frames.download((2, H, W), np.uint8)
    pipe.close()
    return fr[0], fr[1]


def unrelated_pair(W, H):
    p = synth.texture_params(4)
    dx, dy, _ = synth.population_motion(W, H, 3, seed=3)
    return synth.frame(W, H, 2.7, -1.5, p), synth.frame(W, H, dx, dy, p)


cases.append(("bench clip pair (edges) 1080p", 1920, 1080, bench_clip_pair))
cases.append(("unrelated content 700x420", 700, 420, unrelated_pair))
if os.environ.get("OFC_POLYEXP_F64") == "1":
    print("polyexp horizontal sums in f64 (OFC_POLYEXP_F64=1)")
for name, W, H, gen in cases:
    a, b = gen(W, H)
    eng = FlowEngine(W, H)
    got = eng.calc(a, b)
    eng.close()
    want = O.farneback(a, b)
    d = (got - want).astype(np.float64)
    print("%-30s rel L2 %.2e   max|d| %.2e px   px>1e-3: %.4f %%   max|flow| %.2f" %
          (name, np.linalg.norm(d) / np.linalg.norm(want.astype(np.float64)), np.abs(d).max(),
           100.0 * (np.abs(d).max(-1) > 1e-3).mean(), np.abs(want).max()))
