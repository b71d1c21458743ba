#!/bin/bash
# bounded reproduction of the generic tree kernel's rare stall-cap error: the cross-check test itself, N processes one after the other
cd "$(dirname "$0")/.."
out=gpurun_out/repro_generic.txt; : > $out
for i in $(seq 1 ${N:-30}); do
  timeout -k 10 120 python -m pytest tests/test_gpu_pipeline.py -x -q -k "generic_tree_kernel" -s 2>&1 | grep -E "recovered pipeline error|pipeline dbg|passed|failed" >> $out
done
grep -c passed $out; grep -E "recovered|dbg|failed" $out | head -20
