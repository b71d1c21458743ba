#!/bin/bash
# Round-6 evidence for profiles/ (Connect4 pipeline + Tawlbwrdd).  usage: gpu_round6_profiles.sh <commit>
#  1. rocprofv3 kernel stats of a FULL-LENGTH window of the bench worker (the default steps / pre-roll: its AverageNs and the
#     bench's own event times describe the same epochs - VERDICT r4 item 9)          -> gpurun_out/r6_kernel_stats.csv
#  1b. the same for Tawlbwrdd (configs[2])                                              -> gpurun_out/r6_kernel_stats_tawlbwrdd.csv
#  1c. TCC traffic of the Tawlbwrdd round's kernels                                      -> gpurun_out/r6_pmc_traffic_tawlbwrdd.csv
#  2. TCC traffic counters of the net kernel ALONE draining a pre-filled ring           -> gpurun_out/r6_pmc_traffic.csv
#  3. the tree kernel ALONE (every seat EvalType.RANDOM: scripts/pipe_tree_only.py): FETCH_SIZE / WRITE_SIZE in separate passes,
#     then the L2 hit / request counters, the L1 -> L2 request counters and latency, the SQ issue / wait split
#                                                                                       -> gpurun_out/r6_pmc_tree.csv
# The program itself follows `--` (python3 ...): no env / bash -c hop under rocprofv3.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
commit=${1:-unknown}
echo "== kernel stats"
rm -rf /tmp/ks && mkdir -p /tmp/ks
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o r -- python3 bench.py --worker --no-cpu-baseline --no-secondary > gpurun_out/r6_ks.log 2>&1 || { echo "kernel-stats run failed"; tail -5 gpurun_out/r6_ks.log; exit 1; }
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/r6_kernel_stats.csv "$commit" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0] + ["commit"])
    for r in rows[1:14]:
        w.writerow(r + [sys.argv[3] if len(sys.argv) > 3 else ""])
for r in rows[1:9]: print(r[0][:80], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
tail -c 300 gpurun_out/r6_ks.log; echo
rm -rf /tmp/ks
echo "== kernel stats, Tawlbwrdd"
mkdir -p /tmp/kst
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o r -- python3 bench.py --worker --game tawlbwrdd --warmup 1 --no-secondary --preroll-factor 1.0 --no-cpu-baseline --steps 13 > gpurun_out/r6_ks_tawl.log 2>&1 || { echo "tawlbwrdd kernel-stats run failed"; tail -5 gpurun_out/r6_ks_tawl.log; exit 1; }
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
python3 - "$f" gpurun_out/r6_kernel_stats_tawlbwrdd.csv "$commit" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0] + ["commit"])
    for r in rows[1:12]:
        w.writerow(r + [sys.argv[3]])
for r in rows[1:8]: print(r[0][:80], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
rm -rf /tmp/kst
echo "== pmc traffic, Tawlbwrdd round kernels"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$ctr && mkdir -p /tmp/pmct_$ctr
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmct_$ctr -o r -- python3 bench.py --worker --game tawlbwrdd --warmup 0 --no-secondary --preroll-factor 0.05 --no-cpu-baseline --steps 1 --rounds-per-step 64 --profile-window > gpurun_out/r6_pmct_$ctr.log 2>&1 || { echo "tawlbwrdd pmc $ctr failed"; tail -5 gpurun_out/r6_pmct_$ctr.log; }
done
python3 - "$commit" <<'PY'
import csv, glob, collections, sys
out = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob(f"/tmp/pmct_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != ctr: continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(",", " ")[-70:]
            acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    for k, (tot, n) in acc.items():
        out[k][ctr] = (tot / max(n, 1), n)
with open("gpurun_out/r6_pmc_traffic_tawlbwrdd.csv", "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch_raw,commit\n")
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("WRITE_SIZE", (0, 0))[0]):
        fs = d.get("FETCH_SIZE", (0, 0)); ws = d.get("WRITE_SIZE", (0, 0))
        if max(fs[1], ws[1]) < 8: continue
        f.write(f"{k},{max(fs[1], ws[1])},{fs[0]:.2f},{ws[0]:.2f},{sys.argv[1]}\n")
print(open("gpurun_out/r6_pmc_traffic_tawlbwrdd.csv").read())
PY
rm -rf /tmp/pmct_*
echo "== kernel stats, StarGambit (configs[4] per GPU)"
mkdir -p /tmp/kss
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kss -o r -- python3 bench.py --worker --game stargambit --warmup 1 --no-secondary --preroll-factor 0.5 --no-cpu-baseline --steps 100 > gpurun_out/r6_ks_sg.log 2>&1 || { echo "stargambit kernel-stats run failed"; tail -5 gpurun_out/r6_ks_sg.log; }
f=$(find /tmp/kss -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" gpurun_out/r6_kernel_stats_stargambit.csv "$commit" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0] + ["commit"])
    for r in rows[1:12]:
        w.writerow(r + [sys.argv[3]])
for r in rows[1:9]: print(r[0][:80], r[1], "avg_us=%.1f" % (float(r[3]) / 1e3), r[4])
PY
rm -rf /tmp/kss
echo "== pmc traffic, StarGambit round kernels"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$ctr && mkdir -p /tmp/pmcs_$ctr
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcs_$ctr -o r -- python3 bench.py --worker --game stargambit --warmup 0 --no-secondary --preroll-factor 0.002 --no-cpu-baseline --steps 1 --rounds-per-step 64 --profile-window > gpurun_out/r6_pmcs_$ctr.log 2>&1 || { echo "stargambit pmc $ctr failed"; tail -5 gpurun_out/r6_pmcs_$ctr.log; }
done
python3 - "$commit" <<'PY'
import csv, glob, collections, sys
out = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob(f"/tmp/pmcs_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") != ctr: continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(",", " ")[-70:]
            acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    for k, (tot, n) in acc.items():
        out[k][ctr] = (tot / max(n, 1), n)
with open("gpurun_out/r6_pmc_traffic_stargambit.csv", "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch_raw,commit\n")
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("WRITE_SIZE", (0, 0))[0]):
        fs = d.get("FETCH_SIZE", (0, 0)); ws = d.get("WRITE_SIZE", (0, 0))
        if max(fs[1], ws[1]) < 8: continue
        f.write(f"{k},{max(fs[1], ws[1])},{fs[0]:.2f},{ws[0]:.2f},{sys.argv[1]}\n")
print(open("gpurun_out/r6_pmc_traffic_stargambit.csv").read())
PY
rm -rf /tmp/pmcs_*
echo "== pmc: net kernel alone, two drain sizes"
for n in 1536 7680; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${ctr}_$n && mkdir -p /tmp/pmc_${ctr}_$n
  N=$n timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${ctr}_$n -o r -- python3 scripts/pipe_net_pmc.py > gpurun_out/r6_pmc_${ctr}_$n.log 2>&1 || { echo "pmc $ctr $n failed"; tail -5 gpurun_out/r6_pmc_${ctr}_$n.log; exit 1; }
done
done
python3 - "$commit" <<'PY'
import csv, glob, collections, sys
with open("gpurun_out/r6_pmc_traffic.csv", "w") as f:
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
print(open("gpurun_out/r6_pmc_traffic.csv").read())
PY
rm -rf /tmp/pmc_FETCH_SIZE_* /tmp/pmc_WRITE_SIZE_*
echo "== pmc: tree kernel alone (RANDOM seats)"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_tree_$i && mkdir -p /tmp/pmc_tree_$i
  E=20 BLOCKS=2 STATS_OUT=gpurun_out/r6_tree_only_run_$i.json timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_tree_$i -o r -- python3 scripts/pipe_tree_only.py > gpurun_out/r6_pmc_tree_$i.log 2>&1 || { echo "tree pmc pass $i ($set) failed"; tail -5 gpurun_out/r6_pmc_tree_$i.log; continue; }
done
python3 - "$commit" <<'PY'
import csv, glob, collections, sys, json
acc = collections.defaultdict(lambda: [0.0, 0])
for fn in glob.glob("/tmp/pmc_tree_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_pipe_tree" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
sims = []
for fn in glob.glob("gpurun_out/r6_tree_only_run_*.json"):
    sims.append(json.load(open(fn))["sims_per_epoch"])
spe = sum(sims) / max(1, len(sims))
with open("gpurun_out/r6_pmc_tree.csv", "w") as f:
    f.write("kernel,counter,dispatches,value_per_dispatch,value_per_simulation,simulations_per_dispatch,commit\n")
    for k, (tot, n) in sorted(acc.items()):
        per = tot / max(1, n)
        f.write(f"k_pipe_tree<Connect4> alone (EvalType.RANDOM seats),{k},{n},{per:.2f},{per / max(1.0, spe):.4f},{spe:.0f},{sys.argv[1]}\n")
print(open("gpurun_out/r6_pmc_tree.csv").read())
PY
rm -rf /tmp/pmc_tree_*
true
