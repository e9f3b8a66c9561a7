#!/bin/bash
# usage: tools/round_check.sh <tag>   (on the GPU box, through gpurun) -- the round's closing measurements from the CURRENT source:
# GPU tests, parity report, PMC passes (profiles/<tag>_pmc.json), bench.py line + rocprofv3 kernel statistics of the same command
# (two engines, and single engine for exclusive kernel times), configs[4], smoke.
tag=${1:-r03}
mkdir -p gpurun_out profiles
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; rc=$?
tail -4 gpurun_out/gputest.log
[ $rc -ne 0 ] && exit $rc
grep -h "unrelated content" gpurun_out/gputest.log
timeout -k 10 300 python tools/parity_report.py > gpurun_out/parity.log 2>&1; tail -3 gpurun_out/parity.log
bash tools/roofline_pmc.sh $tag > gpurun_out/roofline_pmc.log 2>&1 || { tail -20 gpurun_out/roofline_pmc.log; exit 1; }
tail -5 gpurun_out/roofline_pmc.log | cut -c1-300
timeout -k 10 600 python bench.py > profiles/${tag}_bench_line.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
cut -c1-400 profiles/${tag}_bench_line.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in 2 1; do
  rm -rf gpurun_out/prof_e$e
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e$e -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extras --engines $e > gpurun_out/prof_e$e.log 2>&1 || { tail -5 gpurun_out/prof_e$e.log; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_e$e.log
done
cp $(find gpurun_out/prof_e2 -name "*kernel_stats.csv" | head -1) profiles/${tag}_bench_kernel_stats.csv
cp $(find gpurun_out/prof_e1 -name "*kernel_stats.csv" | head -1) profiles/${tag}_bench_1engine_kernel_stats.csv
timeout -k 10 300 python tools/lloyd_prune_bench.py 2>&1 | grep -v "ofc lloyd" > profiles/${tag}_lloyd_prune_bench.txt; tail -6 profiles/${tag}_lloyd_prune_bench.txt | cut -c1-200
timeout -k 10 120 python tools/flow_pipe_ab.py > profiles/${tag}_flow_pipe_ab.txt 2>&1; cat profiles/${tag}_flow_pipe_ab.txt
timeout -k 10 300 python bench.py --workload cfg4 --steps 2 > profiles/${tag}_cfg4_bench_line.json 2> gpurun_out/bench_cfg4.err || tail -5 gpurun_out/bench_cfg4.err
cut -c1-300 profiles/${tag}_cfg4_bench_line.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cp profiles/${tag}_* gpurun_out/ 2>/dev/null
