# the Connect4 tile's timing variants + in-kernel clock (profiles/r3_c4_tile_timing.txt), then the LDS bank-conflict counters of
# three variants (profiles/r3_c4_tile_pmc.csv)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/c4_tile_timing.hip -o /tmp/c4t || exit 1
/tmp/c4t 714 2904 > gpurun_out/r3_c4_tile_timing.txt 2>&1 || exit 1
cat gpurun_out/r3_c4_tile_timing.txt
C4T_NO_CLOCK=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/prof_tile -o tile -- /tmp/c4t 714 > /dev/null 2>&1
python3 - <<'PY'
import csv, collections, glob
f = glob.glob("gpurun_out/prof_tile/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("gpurun_out/r3_c4_tile_pmc.csv", "w")
out.write("kernel,dispatches,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,conflict_share,SQ_LDS_ADDR_CONFLICT,SQ_INSTS_LDS\n")
for k, d in acc.items():
    n = len(d["SQ_LDS_IDX_ACTIVE"]); m = lambda c: sum(d[c]) / max(1, len(d[c]))
    out.write("%s,%d,%.0f,%.0f,%.3f,%.0f,%.0f\n" % (k.replace(",", ";"), n, m("SQ_LDS_BANK_CONFLICT"), m("SQ_LDS_IDX_ACTIVE"), m("SQ_LDS_BANK_CONFLICT") / max(1.0, m("SQ_LDS_IDX_ACTIVE")), m("SQ_LDS_ADDR_CONFLICT"), m("SQ_INSTS_LDS")))
out.close()
print(open("gpurun_out/r3_c4_tile_pmc.csv").read())
PY
rm -rf gpurun_out/prof_tile
