#!/bin/bash
# usage: tools/fi_batch.sh <pairs> : per-pair time of the level-0 fused-iteration launches for a batch of <pairs>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
b=$1; f=$((b+1))
rm -rf gpurun_out/fib_$b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fib_$b -- python3 bench.py --frames $f --batch $b --steps 6 --warmup 2 --no-cpu --engines 1 > gpurun_out/fib_$b.log 2>&1
python3 - "$b" <<'PY'
import csv, glob, sys
b = int(sys.argv[1])
tr = list(csv.DictReader(open(glob.glob(f'gpurun_out/fib_{b}/**/*kernel_trace.csv', recursive=True)[0])))
for pat in ('k_flow_iter<7, 0>', 'k_flow_iter<7, 2>', 'k_polyexp<0, true>'):
    d = sorted(((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr if pat in r['Kernel_Name']), reverse=True)
    frac = 2 / 9 if '7, 0>' in pat else (1 / 3 if 'flow' in pat else 1.0)
    top = d[: max(1, int(len(d) * frac))]
    units = b if 'flow' in pat else b + 1
    print('batch %2d %-24s level-0 launches n=%d mean %.1f us -> %.2f us per %s' % (b, pat, len(top), sum(top) / len(top), sum(top) / len(top) / units, 'pair' if 'flow' in pat else 'frame'))
PY
