#!/bin/bash
# one GPU session: full gpu test suite, parity report, default bench, batch/engine sweeps, cfg4
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; rc=$?
tail -15 gpurun_out/gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/parity_report.py > gpurun_out/parity.log 2>&1; cat gpurun_out/parity.log
OFC_POLYEXP_F64=1 timeout -k 10 300 python tools/parity_report.py > gpurun_out/parity_f64.log 2>&1; cat gpurun_out/parity_f64.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
cat gpurun_out/bench_default.json
for cfg in "38 1" "19 2" "19 1" "13 3" "13 2" "10 2"; do
  set -- $cfg
  python bench.py --no-cpu --no-extras --frames 39 --steps 20 --warmup 3 --batch $1 --engines $2 | python tools/brief.py "shard39 engines $2"
done
for cfg in "30 2" "32 2" "30 3" "25 2" "30 1"; do
  set -- $cfg
  python bench.py --no-cpu --no-extras --steps 5 --batch $1 --engines $2 | python tools/brief.py "full engines $2"
done
timeout -k 10 300 python bench.py --workload cfg4 --steps 2 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err || tail -5 gpurun_out/bench_cfg4.err
cat gpurun_out/bench_cfg4.json
