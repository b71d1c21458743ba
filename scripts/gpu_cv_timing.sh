cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 env $CVENV python scripts/cv_timing.py > gpurun_out/r5_cv_timing.txt 2>&1; rc=$?
cat gpurun_out/r5_cv_timing.txt
exit $rc
