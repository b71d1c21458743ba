cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kss && mkdir -p /tmp/kss
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kss -o r -- python3 bench.py --worker --game stargambit --warmup 1 --no-secondary --preroll-factor 0.5 --no-cpu-baseline --steps 60 > gpurun_out/r6_ks_sg.log 2>&1
f=$(find /tmp/kss -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:12]: print(r[0][:90], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
for i in 1 2 4; do timeout -k 10 300 python bench.py --worker --game stargambit --warmup 1 --no-secondary --preroll-factor 0.5 --no-cpu-baseline --steps 60 --inline $i 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('inline $i', 'games/s %.2f' % d['value'], 'sims/s %.0f' % d['config']['sims_per_s'], 'tree_ms %.3f net_ms %.3f hit %.3f' % (d['config']['tree_kernel_ms'], d['config']['net_ms'], d['config']['cache_hit_rate']))"; done
