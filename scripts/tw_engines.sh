#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/tw_engines.txt; : > $out
for e in 4 8 4 8 16; do
  timeout -k 10 300 python bench.py --game tawlbwrdd --engines $e --warmup 1 --no-secondary --preroll-factor 1.0 --no-cpu-baseline --steps 13 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('engines $e: %.1f games/s %.2f Msims/s' % (d['value'], d['config']['sims_per_s'] / 1e6))" >> $out 2>&1
done
cat $out
