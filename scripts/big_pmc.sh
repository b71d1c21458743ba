#!/bin/bash
# instruction mix of the wide-game round kernel (rocprofv3 --pmc, the program itself after --)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf /tmp/bigpmc && mkdir -p /tmp/bigpmc
ROUNDS=256 REPS=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/bigpmc -o r -- python3 scripts/big_ab.py > gpurun_out/big_pmc.log 2>&1 || { tail -5 gpurun_out/big_pmc.log; exit 1; }
python3 - <<'PY' > gpurun_out/big_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("/tmp/bigpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "k_round_big" not in k and "k_leafnet_sp" not in k: continue
    w = d["SQ_WAVES"][0] / max(1, d["SQ_WAVES"][1])
    print(k, "dispatches", d["SQ_WAVES"][1], "waves/dispatch %.0f" % w)
    for c, (t, n) in sorted(d.items()):
        print("   %-22s per wave %.1f" % (c, t / n / max(1.0, w)))
PY
cat gpurun_out/big_pmc.txt
