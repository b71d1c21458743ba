#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r4_groups.txt; : > $out
timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q -k "two_model or random_seats or gumbel or rejects or generic" >> $out 2>&1 || exit 1
GUMBEL=1 DRIVER=rounds BLOCKS=4 timeout -k 10 200 python scripts/pipe_groups_bench.py >> $out 2>&1
GUMBEL=1 DRIVER=pipeline BLOCKS=4 timeout -k 10 200 python scripts/pipe_groups_bench.py >> $out 2>&1
echo done >> $out
