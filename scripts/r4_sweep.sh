#!/bin/bash
# usage: r4_sweep.sh "ENV1=.. ENV2=.." "ENV..." ...   one pipe_bench run per argument (env settings), last block printed
set -o pipefail
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== $cfg"
  env CACHE=${CACHE:-128000000} BLOCKS=${BLOCKS:-4} $cfg timeout -k 10 200 python scripts/pipe_bench.py > gpurun_out/r4_sw_$i.log 2>&1 || { tail -20 gpurun_out/r4_sw_$i.log; exit 1; }
  tail -1 gpurun_out/r4_sw_$i.log
  if grep -q "pipe prof" gpurun_out/r4_sw_$i.log; then python scripts/r4_prof.py gpurun_out/r4_sw_$i.log; fi
done
