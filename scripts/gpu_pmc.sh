#!/bin/bash
# HBM-side traffic of the round kernels from the TCC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (they do not fit one pass), kernel-trace only.  Summarised on the box (the raw CSVs
# are large); result: gpurun_out/pmc_traffic.csv with mean KB per dispatch and kernel.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr && mkdir -p /tmp/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o r -- python3 bench.py --steps 300 --warmup 900 --no-cpu-baseline --no-secondary > gpurun_out/pmc_$ctr.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
out = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"/tmp/pmc_{ctr}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            kn = r["Kernel_Name"]
            name = "k_leafnet<6 7 4 16>" if "k_leafnet<" in kn else kn.split("(")[0][-60:].replace(",", " ")
            acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    for k, (tot, n) in acc.items():
        out[k][ctr] = (tot / max(n, 1), n)
with open("gpurun_out/pmc_traffic.csv", "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch_raw\n")
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
        fs = d.get("FETCH_SIZE", (0, 0)); ws = d.get("WRITE_SIZE", (0, 0))
        if max(fs[1], ws[1]) < 50: continue
        f.write(f"{k},{max(fs[1], ws[1])},{fs[0]:.2f},{ws[0]:.2f}\n")
print(open("gpurun_out/pmc_traffic.csv").read())
PY
