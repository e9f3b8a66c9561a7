"""time the last two iterations of a level as two single launches (mode 0) and as one two-iteration launch (mode 1)"""
import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages

for (W, H, n) in [(1920, 1080, 32), (960, 540, 32), (480, 270, 32), (1920, 1080, 38), (3840, 2160, 8)]:
    t0 = stages.bench_flow_iters(W, H, n, 10, 0)
    t1 = stages.bench_flow_iters(W, H, n, 10, 1)
    print("%dx%d x %d pairs: two launches %.3f ms, fused %.3f ms (%.2fx)" % (W, H, n, t0, t1, t0 / t1), flush=True)
