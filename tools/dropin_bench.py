"""per-frame cost of the reference-named entry points on a 1080p clip (host frames in, CSV rows out)"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowclustering_amd import KmeanGrids, synth
from opticalflowclustering_amd.computeOpticalFlowModule import ComputeOpticalFLow

W, H, T = 1920, 1080, 41
p = synth.texture_params(0)
frames = np.stack([np.stack([synth.frame(W, H, 1.1 * t, -0.6 * t, p).astype(np.uint8)] * 3, -1) for t in range(T)])
d = tempfile.mkdtemp()
src = os.path.join(d, "clip.npy")
np.save(src, frames)
os.chdir(d)
for rep in range(2):
    t0 = time.perf_counter()
    rows = KmeanGrids.process_video(src, 1, os.path.join(d, "OutCSV", "clip.csv"), quiet=True)
    dt = time.perf_counter() - t0
print("KmeanGrids.process_video (flow + vis + 350-cell k=1 k-means + CSV): %.2f ms/frame over %d frames" % (dt / len(rows) * 1e3, len(rows)))
cf = ComputeOpticalFLow(frames[0])
t0 = time.perf_counter()
for t in range(1, T):
    cf.compute(frames[t])
print("ComputeOpticalFLow.compute alone: %.2f ms/frame" % ((time.perf_counter() - t0) / (T - 1) * 1e3))
cf.close()
