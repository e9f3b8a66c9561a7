"""workload of the roofline PMC passes: the two kernels bench.py's `roofline` object names, launched exactly as bench.py
launches them (64 x 1080p images for k_polyexp<1,false>; two level-0 iterations over 32 resident 1080p pairs for
k_flow_iter<7,0>)"""
import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages
n_pe, n_fi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 3)     # bench.py uses 20 and 10
print("polyexp ms", stages.bench_polyexp(1920, 1080, 64, n_pe, 0))
print("flow_iter ms", stages.bench_flow_iters(1920, 1080, 32, n_fi, 0) / 2)
