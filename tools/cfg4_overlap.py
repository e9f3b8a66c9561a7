"""from a rocprofv3 --kernel-trace --memory-copy-trace run of `bench.py --workload cfg4`: how much of the host-to-device
upload time runs while kernels run (the copy stream hiding the 8.3 MB per frame under the previous batch's flow)"""
import csv
import glob
import sys

root = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0])))
mt = list(csv.DictReader(open(glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)[0])))
ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kt)
cop = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", ""))) for r in mt]
h2d = [(s, e) for s, e, d in cop if "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper()]
# union of kernel intervals
merged = []
for s, e in ker:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])


def overlap(s, e):
    tot = 0
    for a, b in merged:
        if b <= s:
            continue
        if a >= e:
            break
        tot += min(e, b) - max(s, a)
    return tot


big = [(s, e) for s, e in h2d if e - s > 50_000]          # the frame-batch uploads (tens of MB), not the small control copies
t_copy = sum(e - s for s, e in big)
t_ov = sum(overlap(s, e) for s, e in big)
span = max(e for _, e in ker) - min(s for s, _ in ker)
print("kernels: %d launches, busy %.1f ms of a %.1f ms span" % (len(ker), sum(b - a for a, b in merged) / 1e6, span / 1e6))
print("host-to-device copies: %d in all, %d batch uploads, %.1f ms of copy time" % (len(h2d), len(big), t_copy / 1e6))
print("copy time overlapped by running kernels: %.1f ms = %.0f %%" % (t_ov / 1e6, 100.0 * t_ov / max(t_copy, 1)))
