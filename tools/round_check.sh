#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; rc=$?
tail -6 gpurun_out/gputest.log
[ $rc -ne 0 ] && exit $rc
grep -h "unrelated content" gpurun_out/gputest.log
timeout -k 10 300 python tools/parity_report.py > gpurun_out/parity.log 2>&1; tail -3 gpurun_out/parity.log
OFC_POLYEXP_F64=1 timeout -k 10 300 python tools/parity_report.py > gpurun_out/parity_f64.log 2>&1; tail -3 gpurun_out/parity_f64.log
bash tools/roofline_pmc.sh ${1:-r02} > gpurun_out/roofline_pmc.log 2>&1 || { tail -20 gpurun_out/roofline_pmc.log; exit 1; }
tail -40 gpurun_out/roofline_pmc.log
cp profiles/*${1:-r02}* profiles/r02_pmc.json gpurun_out/ 2>/dev/null
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
cat gpurun_out/bench_default.json
timeout -k 10 300 python bench.py --workload cfg4 --steps 2 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err || tail -5 gpurun_out/bench_cfg4.err
cat gpurun_out/bench_cfg4.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
