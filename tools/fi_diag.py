import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages
print(stages.bench_flow_iters(1920, 1080, 32, 2, 0), stages.bench_flow_iters(1920, 1080, 32, 2, 2))
