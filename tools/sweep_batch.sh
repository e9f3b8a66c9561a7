#!/bin/bash
# usage: tools/sweep_batch.sh : whole-clip step time for several flow batch sizes / engine counts
for cfg in "4 2" "6 2" "8 2" "12 2" "16 2" "24 2" "32 2" "48 2" "32 3" "16 3" "8 4"; do
  set -- $cfg
  python bench.py --no-cpu --steps 3 --warmup 1 --batch $1 --engines $2 | python tools/brief.py "engines $2"
done
