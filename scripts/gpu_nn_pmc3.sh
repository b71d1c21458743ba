#!/bin/bash
# Instruction-mix and busy counters of the Connect4 leaf-net tile alone (profiles/r3_leafnet_pmc_<rows>.csv; round 2's
# r2_leafnet_pmc_<rows>.csv is the same measurement before the round-3 cuts of the tile's non-MFMA instructions).  usage: gpu_nn_pmc3.sh [rows]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
B=${1:-2904}
out=gpurun_out/r3_leafnet_pmc_$B.csv
python3 scripts/nnbench.py $B 200 2>/dev/null | grep -v amdgpu > $out
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES"; do
  rm -rf /tmp/nnp && mkdir -p /tmp/nnp
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/nnp -o r -- python3 scripts/nnbench.py $B 30 > /dev/null 2>&1
  python3 - >> $out <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/nnp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_leafnet" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
for k, (t, n) in sorted(acc.items()): print(f"{k},{t/n:.1f}")
PY
done
rm -rf /tmp/nnp
cat $out
