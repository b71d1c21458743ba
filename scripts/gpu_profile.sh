#!/bin/bash
# Run on the GPU box (via gpurun): default bench line + rocprofv3 kernel stats of the same command.
# Outputs land in gpurun_out/; copy the summaries into profiles/ afterwards.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
rm -rf /tmp/prof && mkdir -p /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python3 bench.py --steps 6000 --warmup 6000 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv
ls -la /tmp/prof/* | head -20 > gpurun_out/prof_ls.txt
