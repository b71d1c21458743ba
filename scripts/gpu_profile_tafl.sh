#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/proft && mkdir -p /tmp/proft
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o r -- python3 bench.py --game tawlbwrdd --engines 4 --steps 2000 --warmup 6000 > gpurun_out/bench_tafl_prof.json 2> gpurun_out/bench_tafl_prof.err
f=$(find /tmp/proft -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_tafl.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/kernel_stats_tafl.csv')))
for r in rows[:8]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Percentage'])
PY
