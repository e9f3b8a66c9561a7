"""workload for the PMC passes: the last two iterations of a 32-pair 1080p level, as two launches and as one"""
import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
print(stages.bench_flow_iters(1920, 1080, n, 2, 0), stages.bench_flow_iters(1920, 1080, n, 2, 1))
