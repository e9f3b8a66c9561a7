"""where a step of k_flow_iter spends its cycles: the stamped diagnostic build (s_memtime around every phase, summed per wave)"""
import sys
sys.path.insert(0, ".")
from opticalflowclustering_amd import stages
stages.bench_flow_iters(1920, 1080, 32, 1, 3)
