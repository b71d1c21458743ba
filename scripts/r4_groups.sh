#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r4_groups.txt; : > $out
timeout -k 10 200 python -m pytest tests/test_gpu_pipeline.py -x -q -k "two_model or random_seats or gumbel or rejects" >> $out 2>&1 || exit 1
DRIVER=rounds timeout -k 10 200 python scripts/pipe_groups_bench.py >> $out 2>&1
DRIVER=pipeline timeout -k 10 200 python scripts/pipe_groups_bench.py >> $out 2>&1
DRIVER=pipeline AZMI_PIPE_GENERIC=1 BLOCKS=3 timeout -k 10 200 python scripts/pipe_groups_bench.py >> $out 2>&1
echo done >> $out
