"""time the last two iterations of a level: two k_flow_iter launches (mode 0), one k_flow_iter2 launch (mode 1), two
k_flow_iter_w3 launches (mode 2)"""
import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages

for (W, H, n) in [(1920, 1080, 32), (960, 540, 32), (480, 270, 32), (240, 135, 32), (1920, 1080, 19), (3840, 2160, 8)]:
    t = [stages.bench_flow_iters(W, H, n, 10, m) for m in (0, 1, 2)]
    print("%dx%d x %d pairs: k_flow_iter x2 %.3f ms | k_flow_iter2 %.3f ms (%.2fx) | k_flow_iter_w3 x2 %.3f ms (%.2fx)"
          % (W, H, n, t[0], t[1], t[0] / t[1], t[2], t[0] / t[2]), flush=True)
