#!/bin/bash
# SQ counters of the wide-game tree kernel on StarGambit (RANDOM evaluator: only the tree kernels run): where a simulation's
# time goes - instruction issue, LDS, vector memory waits.   usage: gpu_sg_pmc.sh [slots] [rounds]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
S=${1:-256}; R=${2:-96}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT"; do
  rm -rf /tmp/sgp && mkdir -p /tmp/sgp
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sgp -o r -- python3 scripts/sg_speed.py $S 800 1 $R 0 random > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/sgp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_round_big" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
for k, (t, n) in sorted(acc.items()): print(f"{k},{t/n:.1f},{n}")
PY
done
