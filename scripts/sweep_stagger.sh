#!/bin/bash
# usage: sweep_stagger.sh "US US ..." [bench args]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
us_list=$1; shift
for u in $us_list; do
  if [ "$u" = 0 ]; then unset AZMI_STAGGER_US; else export AZMI_STAGGER_US=$u; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/st_$u.json 2> gpurun_out/st_$u.err || { echo "bench failed $u"; tail -3 gpurun_out/st_$u.err; exit 1; }
  python3 - gpurun_out/st_$u.json $u <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("stagger_us", sys.argv[2], "games/s=%.1f" % d["value"], "net_ms=%.3f" % d["roofline"]["per_launch_event_ms"], "tree_ms=%.3f" % d["roofline_tree"]["per_launch_event_ms"], flush=True)
PY
done
