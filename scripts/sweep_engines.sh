#!/bin/bash
# engine-shard sweep of a wide game: usage sweep_engines.sh GAME "E:HWQ E:HWQ ..." [bench args]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
game=$1; shift; combos=$1; shift
for c in $combos; do
  e=${c%%:*}; q=${c##*:}
  timeout -k 10 300 python3 bench.py --game $game --engines $e --hwq $q --no-cpu-baseline --no-secondary "$@" > gpurun_out/sw_${game}_${e}_${q}.json 2> gpurun_out/sw_${game}_${e}_${q}.err || { echo "bench failed $c"; tail -3 gpurun_out/sw_${game}_${e}_${q}.err; exit 1; }
  python3 - gpurun_out/sw_${game}_${e}_${q}.json $c <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "games/s=%.1f" % d["value"], "net_ms=%.3f" % d["roofline"]["per_launch_event_ms"], "tree_ms=%.3f" % d["roofline_tree"]["per_launch_event_ms"], flush=True)
PY
done
