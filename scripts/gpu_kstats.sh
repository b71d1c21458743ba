#!/bin/bash
# rocprofv3 kernel stats of a short bench window -> gpurun_out/$1_kernel_stats.csv  (usage: gpu_kstats.sh name [bench args])
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
name=$1; shift
rm -rf /tmp/prof && mkdir -p /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python3 bench.py --steps 3 --warmup 1 --preroll-factor 1.0 --no-cpu-baseline --no-secondary "$@" > gpurun_out/${name}_prof.json 2> gpurun_out/${name}_prof.err
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${name}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-60s calls %7s avg %8.1f us  %5.1f%%  min %7.1f max %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"]), float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
