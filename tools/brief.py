"""print the headline fields of bench.py's JSON line read from stdin"""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(" ".join(sys.argv[1:]), "batch", d["config"]["flow_batch_pairs"], "value %.0f Mpx/s" % d["value"],
              "%.2f ms/step" % d["ms_per_step"], "roofline %.3f" % d["roofline"]["frac"], "iters", d["config"]["lloyd_iters"])
