#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r4_t_pipe.log 2>&1 || { tail -30 gpurun_out/r4_t_pipe.log; exit 1; }
tail -2 gpurun_out/r4_t_pipe.log
for C in 200000 32000000 128000000; do
  echo "== cache $C"
  CACHE=$C Q=256 E=40 BLOCKS=10 timeout -k 10 200 python scripts/pipe_bench.py > gpurun_out/r4_bal_$C.log 2>&1 || { tail -20 gpurun_out/r4_bal_$C.log; exit 1; }
  grep -h "block [2579]" gpurun_out/r4_bal_$C.log | cut -c1-120
  tail -1 gpurun_out/r4_bal_$C.log | grep -o "tree_wgs[^,]*" | head -1
done
