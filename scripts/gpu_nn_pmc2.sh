cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
B=${1:-3072}
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | head -40
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/nnp && mkdir -p /tmp/nnp
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/nnp -o r -- python3 scripts/nnbench.py $B 30 > /dev/null 2>&1
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
