#!/bin/bash
# usage: tools/sweep_batch.sh : whole-clip step time for several flow batch sizes / engine counts
for cfg in "16 2" "24 2" "32 1" "32 2" "32 3" "40 2" "50 2" "64 2" "75 2" "100 1" "150 1"; do
  set -- $cfg
  python bench.py --no-cpu --steps 3 --warmup 1 --batch $1 --engines $2 | python tools/brief.py "engines $2"
done
