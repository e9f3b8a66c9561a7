"""profiles/<tag>_pmc.json from the PMC passes of tools/roofline_pmc.sh: per kernel FETCH_SIZE / WRITE_SIZE means, the gfx950
read correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B; checked here against
TCC_EA0_RDREQ with no 32-B requests) and the fingerprint of the kernel source the passes belong to"""
import csv
import glob
import json
import sys
from collections import defaultdict

sys.path.insert(0, ".")
import bench

out, tag = sys.argv[1], sys.argv[2]
# the workload also runs the whole clip (all pyramid levels): of every kernel only the launches with the LARGEST grid count;
# k_flow_iter's level-0 and level-1 launches happen to share a grid size, so for it the bench hook's own launches are taken
# by position: it runs first, 1 preparing launch + 2 warm-up + 2 per repetition (tools/roofline_pmc.py: 3 repetitions in the
# counter passes, 10 in the --stats run)
rows = []
for path in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
gmax = defaultdict(int)
for r in rows:
    gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size"]))
acc = defaultdict(lambda: defaultdict(list))
for r in sorted(rows, key=lambda r: int(r.get("Dispatch_Id", 0))):
    if int(r["Grid_Size"]) == gmax[r["Kernel_Name"]] or "k_flow_iter<7, 0, false, false>" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for path in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    tr = list(csv.DictReader(open(path)))
    g2 = defaultdict(int)
    for r in tr:
        g2[r["Kernel_Name"]] = max(g2[r["Kernel_Name"]], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    for r in sorted(tr, key=lambda r: int(r["Start_Timestamp"])):
        if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) == g2[r["Kernel_Name"]] or \
                "k_flow_iter<7, 0, false, false>" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
W, H = 1920, 1080
want = {"k_polyexp": ("k_polyexp<1, false", 24 * W * H * 64, "64 x 1920x1080 images per launch"),
        "k_flow_iter": ("k_flow_iter<7, 0,", 56 * W * H * 32, "32 x 1920x1080 pairs per launch, level-0 iteration 2/3"),
        "k_lloyd_assign": ("k_lloyd_assign<2, 5, float, 3>", 8 * W * H * 299, "full label-less sweep, 299 x 1920x1080 (u,v) vectors"),
        "k_lloyd_tiles_pruned": ("k_lloyd_tiles<5, 0>", None, "pruned tile sweep over the same vectors, converged centres"),
        "k_tile_meta": ("k_tile_meta(", 8 * W * H * 299 + 40 * W * H * 299 // 64, "tile-metadata pass before iteration 0"),
        "k_lloyd_tiles_final": ("k_lloyd_tiles<5, 1>", None, "pruned final E-step: labels written + inertia"),
        "k_lloyd_final": ("k_lloyd_assign<2, 5, float, 2>", 9 * W * H * 299, "full final E-step: labels written + inertia")}
rec = {"source_sha16": bench.source_sha16(), "tag": tag, "kernels": {},
       "correction": "read bytes = 2 x FETCH_SIZE x 1024 (all read requests are 128 B: TCC_EA0_RDREQ_32B = 0 and "
                     "FETCH_SIZE x 1024 = TCC_EA0_RDREQ x 64); WRITE_SIZE x 1024 exact",
       "collected": "tools/roofline_pmc.sh: separate --pmc passes with --kernel-trace only, MI355X; summaries in "
                    "profiles/%s_roofline_pmc_summary.txt, durations in profiles/%s_roofline_kernel_stats.csv" % (tag, tag)}
for key, (pat, alg, cfg) in want.items():
    name = [k for k in acc if pat in k]
    if key == "k_flow_iter":            # the plain form (not the one that also sums the field)
        name = [k for k in name if "false, false>" in k]
    assert len(name) == 1, (pat, name)
    # the bench hooks launch each kernel several times with identical arguments; the fits before them launch the Lloyd
    # kernels with other centres (and the speculative no-op launches behind the halt flag): keep the launches of the hook =
    # the LAST ones (the pipeline's own launches come first)
    tail = 4 if "lloyd" in key else 0
    if key == "k_flow_iter":
        c = {k: sum(v[1:9]) / len(v[1:9]) for k, v in acc[name[0]].items()}
    else:
        c = {k: (sum(v[-tail:]) / tail if tail else sum(v) / len(v)) for k, v in acc[name[0]].items()}
    assert c.get("TCC_EA0_RDREQ_32B_sum", 0) == 0
    assert abs(c["FETCH_SIZE"] * 1024 - c["TCC_EA0_RDREQ_sum"] * 64) <= 0.02 * c["FETCH_SIZE"] * 1024, c
    traffic = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
    d = dur[name[0]][-12:] if "lloyd" in key else (dur[name[0]][1:23] if key == "k_flow_iter" else dur[name[0]])
    rec["kernels"][key] = {"symbol": pat, "config": cfg, "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
                           "read_bytes": 2 * c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024,
                           "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg,
                           "traffic_over_algorithmic": traffic / alg if alg else None,
                           "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1),
                           "avg_launch_us_rocprof": sum(d) / max(len(d), 1)}
json.dump(rec, open("profiles/%s_pmc.json" % tag, "w"), indent=1)
