#!/bin/bash
# A/B of PlayParams overrides (bench.py AZMI_BENCH_OVERRIDES) on the GPU box: each argument is one override list
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  echo "== $v"
  AZMI_BENCH_OVERRIDES="$v" python bench.py --steps 6000 --warmup 30000 --no-cpu-baseline --no-secondary 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(round(d['value'],1),'games/s', round(d['ms_per_step'],4),'ms', 'tree',round(c['tree_kernel_ms'],4),'net',round(c['net_ms'],4),'hit',round(c['cache_hit_rate'],3),'sims/s',round(c['sims_per_s']/1e6,2))
"
done
