#!/bin/bash
for i in 1 2 3; do
  cp gpurun_keep_nopref.so opticalflowclustering_amd/libofc.so; python bench.py --no-cpu | python tools/brief.py nopref
  cp gpurun_keep_pref.so opticalflowclustering_amd/libofc.so; python bench.py --no-cpu | python tools/brief.py pref
done
