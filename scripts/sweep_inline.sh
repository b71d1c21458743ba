#!/bin/bash
# usage: sweep_inline.sh "N N ..." [bench args]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
ns=$1; shift
for n in $ns; do
  timeout -k 10 300 python3 bench.py --inline $n --no-cpu-baseline --no-secondary "$@" > gpurun_out/il_$n.json 2> gpurun_out/il_$n.err || { echo "bench failed $n"; tail -3 gpurun_out/il_$n.err; exit 1; }
  python3 - gpurun_out/il_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("inline", sys.argv[2], "games/s=%.1f" % d["value"], "net_ms=%.3f" % d["roofline"]["per_launch_event_ms"], "tree_ms=%.3f" % d["roofline_tree"]["per_launch_event_ms"], flush=True)
PY
done
