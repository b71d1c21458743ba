# the conveyor: parity with the tile kernel, then its timing alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest scripts/experiments/test_conveyor_experiment.py -x -q > gpurun_out/r5_cv_test.txt 2>&1; rc=$?
tail -15 gpurun_out/r5_cv_test.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 env $CVENV python scripts/cv_timing.py > gpurun_out/r5_cv_timing.txt 2>&1; rc=$?
cat gpurun_out/r5_cv_timing.txt
exit $rc
