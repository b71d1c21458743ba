#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for T in ${TLIST:-96 160}; do
  echo "== tree wgs $T"
  AZMI_PIPE_PROF=1 AZMI_PIPE_TREE_WGS=$T CACHE=128000000 BLOCKS=4 timeout -k 10 200 python scripts/pipe_bench.py > gpurun_out/r4_pp_T$T.log 2>&1 || { tail -20 gpurun_out/r4_pp_T$T.log; exit 1; }
  tail -1 gpurun_out/r4_pp_T$T.log; python scripts/r4_prof.py gpurun_out/r4_pp_T$T.log
done
