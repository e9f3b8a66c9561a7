#!/bin/bash
# N ranks of bench.py on ONE GPU over the gloo host transport (orchestration rehearsal; not a scaling measurement)
for n in 2 4; do
OFC_DIST_TRANSPORT=gloo OFC_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 2 --warmup 1 --no-cpu > gpurun_out/rehearse_$n.json 2> gpurun_out/rehearse_$n.err || { tail -5 gpurun_out/rehearse_$n.err; exit 1; }
python - $n <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/rehearse_{n}.json") if l.startswith("{")][-1])
print("N=%s (one GPU shared): %.1f ms/step, iters %d, batch %d, centres[0] %s, transport: %s" % (n, d["ms_per_step"], d["config"]["lloyd_iters"], d["config"]["flow_batch_pairs"], d["config"]["centers"][0], d["config"]["parallelism"][-30:]))
PY
done
