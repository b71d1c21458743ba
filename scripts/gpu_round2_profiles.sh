#!/bin/bash
# Round-2 evidence for profiles/: bench lines, rocprofv3 kernel stats of the same command (short window) and the TCC traffic
# counters.   usage: gpu_round2_profiles.sh connect4|tawlbwrdd|stargambit
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
game=${1:-connect4}
case $game in
  connect4)   tag=r2;            args="";                                   bargs="--steps 20 --warmup 5" ;;
  tawlbwrdd)  tag=r2_tawlbwrdd;  args="--game tawlbwrdd";                   bargs="--game tawlbwrdd --steps 20 --warmup 2 --no-cpu-baseline --no-secondary" ;;
  stargambit) tag=r2_stargambit; args="--game stargambit";                  bargs="--game stargambit --preroll-factor 1 --steps 100 --warmup 2 --no-cpu-baseline --no-secondary" ;;
esac
echo "== bench ($game)"
python3 bench.py $bargs > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
tail -c 600 gpurun_out/${tag}_bench.json; echo
echo "== kernel stats ($game)"
rm -rf /tmp/ks && mkdir -p /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o r -- python3 bench.py $args --steps 2 --warmup 1 --rounds-per-step 1024 --preroll-factor 0.3 --profile-window --no-cpu-baseline --no-secondary > gpurun_out/${tag}_ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows[:13]:
        w.writerow(r)
for r in rows[1:7]: print(r[0][:70], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
echo "== pmc traffic ($game)"
bash scripts/gpu_pmc.sh $tag $args > gpurun_out/${tag}_pmc.log 2>&1; tail -8 gpurun_out/${tag}_pmc.log
