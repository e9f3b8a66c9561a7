"""workload of the roofline PMC passes: the two kernels bench.py's `roofline` object names, launched exactly as bench.py
launches them (64 x 1080p images for k_polyexp<1,false>; two level-0 iterations over 32 resident 1080p pairs for
k_flow_iter<7,0>)"""
import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages
print("polyexp ms", stages.bench_polyexp(1920, 1080, 64, 3, 0))
print("flow_iter ms", stages.bench_flow_iters(1920, 1080, 32, 3, 0) / 2)
