#!/bin/bash
# Run on the GPU box (via gpurun): the driver's bench command + rocprofv3 kernel stats of the same workload.
# usage: gpu_profile.sh <tag> [bench args]   -> gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv (copy into profiles/ afterwards)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r2}; shift || true
python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"
rm -rf /tmp/prof && mkdir -p /tmp/prof
# the same workload under the kernel trace: a steady-state window after the same pre-roll (trace files grow with the launch count)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python3 bench.py --steps 4 --warmup 1 --profile-window --no-cpu-baseline --no-secondary "$@" > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_bench_prof.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 scripts/benchline.py gpurun_out/${tag}_bench.json $tag
head -6 gpurun_out/${tag}_kernel_stats.csv | cut -c1-220
