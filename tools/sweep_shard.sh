#!/bin/bash
# usage: tools/sweep_shard.sh <frames> : bench the clip step for several (batch, engines) splits
f=$1
for cfg in "32 2" "37 1" "40 1" "19 2" "20 2" "13 3" "10 2" "10 4"; do
  set -- $cfg
  python bench.py --no-cpu --frames $f --steps 20 --batch $1 --engines $2 | python tools/brief.py "frames $f engines $2"
done
