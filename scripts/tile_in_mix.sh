#!/bin/bash
# average duration of a net tile INSIDE the pipeline (beside the tree kernel): prof[8] = ticks the net workgroups spent in tiles, prof[7] = ticks waiting
cd "$(dirname "$0")/.."
AZMI_PIPE_PROF=1 CACHE=128000000 BLOCKS=6 E=100 timeout -k 10 300 python scripts/pipe_bench.py > gpurun_out/r5_tile_in_mix.txt 2>&1
python - <<'PY' >> gpurun_out/r5_tile_in_mix.txt
import re
L = open('gpurun_out/r5_tile_in_mix.txt').read().splitlines()
prof = [list(map(int, l.split(':')[1].split())) for l in L if l.startswith('pipe prof:')]
print('calls with prof lines:', len(prof))
a, b = prof[-4], prof[-1]
tt = [int(re.search(r'tiles_total (\d+)', l).group(1)) for l in L if 'tiles_total' in l]
nw = [int(re.search(r' N (\d+) ', l).group(1)) for l in L if 'tiles_total' in l]
print('tiles %d in the last 3 blocks; average tile %.1f us; net workgroups %d' % (tt[-1]-tt[-4], (b[8]-a[8]) / max(1, tt[-1]-tt[-4]) / 100.0, nw[-1]))
print('net wait ticks %d tile ticks %d -> busy %.3f' % (b[7]-a[7], b[8]-a[8], (b[8]-a[8]) / max(1, b[7]-a[7]+b[8]-a[8])))
PY
