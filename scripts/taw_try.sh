cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/ks && mkdir -p /tmp/ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o r -- python3 bench.py --game tawlbwrdd --steps 2 --warmup 1 --rounds-per-step 1024 --preroll-factor 0.3 --profile-window --no-cpu-baseline --no-secondary > gpurun_out/taw_ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:10]: print(r[0][:90], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
