#!/bin/bash
# quick A/B of bench knobs on the GPU box; prints value / ms_per_step per variant
cd "${GRAFT_REPO_ROOT:-.}"
run() { echo "== $*"; python bench.py --steps 6000 --warmup 12000 --no-cpu-baseline --no-secondary "$@" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(round(d['value'],1),'games/s', round(d['ms_per_step'],4),'ms', 'tree',round(c['tree_kernel_ms'],4),'net',round(c['net_ms'],4),'hit',round(c['cache_hit_rate'],3),'sims/s',round(c['sims_per_s']/1e6,2))
    else: print(l.rstrip()[:200])
"; }
for v in "$@"; do run $v; done
