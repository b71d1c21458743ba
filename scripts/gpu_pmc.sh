#!/bin/bash
# HBM-side traffic of the round kernels from the TCC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (they do not fit one pass), kernel-trace only.  Summarised on the box (the raw CSVs
# are large; COMMIT=<sha> stamps every row); result: gpurun_out/<tag>_pmc_traffic.csv with mean KB per dispatch and kernel (raw counter values:
# FETCH_SIZE is doubled by the reader, see bench.py).   usage: gpu_pmc.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=${1:-r2}; shift || true
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr && mkdir -p /tmp/pmc_$ctr
  # a short steady-state window: counter collection serialises the dispatches, so the pre-roll is shortened too
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o r -- python3 bench.py --steps 1 --warmup 0 --rounds-per-step 300 --preroll-factor 0.05 --profile-window --no-cpu-baseline --no-secondary "$@" > gpurun_out/${tag}_pmc_$ctr.log 2>&1
done
python3 - "$tag" "${COMMIT:-unknown}" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]; commit = sys.argv[2]
out = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"/tmp/pmc_{ctr}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            kn = r["Kernel_Name"]
            name = kn.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(",", " ")[-70:]
            acc[name][0] += float(r["Counter_Value"]); acc[name][1] += 1
    for k, (tot, n) in acc.items():
        out[k][ctr] = (tot / max(n, 1), n)
with open(f"gpurun_out/{tag}_pmc_traffic.csv", "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_dispatch_raw,WRITE_SIZE_KB_per_dispatch_raw,commit\n")
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
        fs = d.get("FETCH_SIZE", (0, 0)); ws = d.get("WRITE_SIZE", (0, 0))
        if max(fs[1], ws[1]) < 50: continue
        f.write(f"{k},{max(fs[1], ws[1])},{fs[0]:.2f},{ws[0]:.2f},{commit}\n")
print(open(f"gpurun_out/{tag}_pmc_traffic.csv").read())
PY
