#!/bin/bash
# Round-3 evidence for profiles/ (Connect4 pipeline): rocprofv3 kernel stats of a short window of the bench worker, the TCC
# traffic counters of the net kernel alone, and the probe that shows why the co-resident pair cannot be counted.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
commit=${1:-unknown}
echo "== kernel stats"
rm -rf /tmp/ks && mkdir -p /tmp/ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o r -- python3 bench.py --worker --steps 3 --warmup 1 --preroll-factor 0.3 --profile-window --no-cpu-baseline --no-secondary > gpurun_out/r3_ks.log 2>&1 || { echo "kernel-stats run failed"; tail -5 gpurun_out/r3_ks.log; exit 1; }
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/r3_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    for r in rows[:14]:
        w.writerow(r)
for r in rows[1:9]: print(r[0][:80], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
tail -c 400 gpurun_out/r3_ks.log; echo
rm -rf /tmp/ks
echo "== pmc: net kernel alone, two drain sizes (a launch's traffic = the weight image once per XCD L2 + a per-position part)"
for n in 1536 7680; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${ctr}_$n && mkdir -p /tmp/pmc_${ctr}_$n
  N=$n timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${ctr}_$n -o r -- python3 scripts/pipe_net_pmc.py > gpurun_out/r3_pmc_${ctr}_$n.log 2>&1 || { echo "pmc $ctr $n failed"; tail -5 gpurun_out/r3_pmc_${ctr}_$n.log; exit 1; }
done
done
python3 - "$commit" <<'PY'
import csv, glob, collections, sys, os
with open("gpurun_out/r3_pmc_traffic.csv", "w") as f:
    f.write("kernel,dispatches,positions_per_dispatch,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch_raw,commit\n")
    for N in (1536, 7680):
        out = collections.defaultdict(dict)
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for fn in glob.glob(f"/tmp/pmc_{ctr}_{N}/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(fn)):
                    if r.get("Counter_Name") != ctr: continue
                    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(",", " ")[-70:]
                    acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
            for k, (tot, n) in acc.items():
                out[k][ctr] = (tot / max(n, 1), n)
        for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
            fs = d.get("FETCH_SIZE", (0, 0)); ws = d.get("WRITE_SIZE", (0, 0))
            if "k_pipe_net" not in k: continue
            f.write(f"{k},{max(fs[1], ws[1])},{N},{fs[0]:.2f},{ws[0]:.2f},{sys.argv[1]}\n")
print(open("gpurun_out/r3_pmc_traffic.csv").read())
PY
rm -rf /tmp/pmc_FETCH_SIZE_* /tmp/pmc_WRITE_SIZE_*
echo "== pmc: a pipeline epoch (expected: the serialised tree kernel times out)"
rm -rf /tmp/pmc_probe && mkdir -p /tmp/pmc_probe
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_probe -o r -- python3 scripts/pipe_pmc_probe.py > gpurun_out/r3_pmc_pipeline_probe.txt 2>&1
grep -a "pipeline under\|epochs ran" gpurun_out/r3_pmc_pipeline_probe.txt || tail -3 gpurun_out/r3_pmc_pipeline_probe.txt
rm -rf /tmp/pmc_probe
true
