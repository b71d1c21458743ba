#!/bin/bash
# rocprofv3 kernel stats of a short Tawlbwrdd window (lock-step rounds, 4 shards) -> gpurun_out/r4_kernel_stats_tawlbwrdd.csv
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/kst && mkdir -p /tmp/kst
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o r -- python3 bench.py --worker --game tawlbwrdd --steps 2 --warmup 1 --preroll-factor 0.2 --profile-window --no-cpu-baseline --no-secondary "$@" > gpurun_out/r4_kst.log 2>&1 || { echo failed; tail -5 gpurun_out/r4_kst.log; exit 1; }
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/r4_kernel_stats_tawlbwrdd.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows[:14]:
        w.writerow(r)
for r in rows[1:11]: print(r[0][:90], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
rm -rf /tmp/kst
