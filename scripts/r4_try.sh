#!/bin/bash
# round-4 iteration script: pipeline parity tests, then the quick throughput probe at a few settings
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r4_t_pipe.log 2>&1 || { tail -30 gpurun_out/r4_t_pipe.log; exit 1; }
tail -3 gpurun_out/r4_t_pipe.log
for T in ${TLIST:-96 128 160}; do
  echo "== tree wgs $T" 
  AZMI_PIPE_TREE_WGS=$T CACHE=128000000 BLOCKS=4 timeout -k 10 200 python scripts/pipe_bench.py > gpurun_out/r4_pb_T$T.log 2>&1 || { tail -20 gpurun_out/r4_pb_T$T.log; exit 1; }
  tail -2 gpurun_out/r4_pb_T$T.log
done
