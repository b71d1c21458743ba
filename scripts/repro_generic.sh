# bounded reproduction loop for the generic tree kernel's rare stall-cap error: N engine-iterations of the failing
# test's configuration, the first error printed with the kernel's diagnostic words (stderr: "pipeline dbg: ...")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${N:-2500} timeout -k 10 ${T:-1100} python scripts/repro_generic.py > gpurun_out/r5_repro_generic.txt 2>&1
tail -5 gpurun_out/r5_repro_generic.txt
