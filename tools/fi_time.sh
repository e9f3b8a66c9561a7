#!/bin/bash
# usage: tools/fi_time.sh <tag> : average duration of the fused-iteration launches of one 32-pair 1080p batch (rocprofv3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fi_$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fi_$1 -- python3 bench.py --frames 33 --batch 32 --steps 3 --warmup 1 --no-cpu --engines 1 > gpurun_out/fi_$1.log 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
tr = list(csv.DictReader(open(glob.glob(f'gpurun_out/fi_{tag}/**/*kernel_trace.csv', recursive=True)[0])))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr if 'k_flow_iter<7, 0>' in r['Kernel_Name']]
d.sort(reverse=True)
n = len(d) // 4          # the level-0 launches are the slowest quarter... (2 of 9 per batch: top 2/9)
top = d[: max(1, len(d) * 2 // 9)]
print(tag, 'level-0 k_flow_iter<7,0>: n=%d mean %.1f us min %.1f us' % (len(top), sum(top) / len(top), min(top)))
PY
