#!/bin/bash
# SQ counters of k_leafnet alone (standalone forward, 4096 rows): MFMA busy, wait buckets, LDS conflicts
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
python3 scripts/nnbench.py 4096 200; python3 scripts/nnbench.py 1024 200; python3 scripts/nnbench.py 2048 200
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/nnp && mkdir -p /tmp/nnp
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/nnp -o r -- python3 scripts/nnbench.py 4096 30 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/nnp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_leafnet" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
for k, (t, n) in sorted(acc.items()): print(f"{k},{t/n:.1f}")
PY
done
