"""per-step timeline from a rocprofv3 kernel trace of bench.py: for the LAST step, every launch in start order with its
duration, stream and the gap to the previous launch's end; then per-kernel totals per step"""
import csv
import glob
import sys
from collections import defaultdict

tr = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r["Queue_Id"])) for r in tr)
# a step starts at a k_level_dec<8 (first launch of flow_run for the first batch) following a Lloyd kernel
def lloydish(n):
    return "lloyd" in n or "reduce" in n or "rocclr" in n or "colstats" in n


starts = [i for i, e in enumerate(ev) if "k_level_dec<8" in e[2] and i > 0 and lloydish(ev[i - 1][2])]
if len(starts) < 2:
    print("could not find step boundaries", len(starts))
    sys.exit(0)
a, b = starts[-2], starts[-1]
step = ev[a:b]
t0 = step[0][0]
print("last full step: %d launches, %.3f ms from first start to last end" % (len(step), (max(e[1] for e in step) - t0) / 1e6))
short = lambda n: n.replace("void ofc::", "").replace("ofc::", "").split("(")[0][:34]
prev_end = t0
if len(sys.argv) > 2:
    for s, e, n, q in step:
        print("%9.1f us  +%7.1f  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, short(n)))
        prev_end = max(prev_end, e)
tot = defaultdict(lambda: [0, 0.0])
for s, e, n, q in step:
    tot[short(n)][0] += 1
    tot[short(n)][1] += (e - s) / 1e3
busy = 0.0
cur_s, cur_e = None, None
for s, e, n, q in step:            # union of busy intervals
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("GPU busy (union of kernel intervals) %.3f ms" % (busy / 1e6))
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("  %-36s n=%-3d %8.1f us" % (k, c, t))
