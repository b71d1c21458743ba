# the conveyor against the tile kernel (tests/test_gpu_conveyor.py), then its timing alone (azmi_debug_pipe_net_bench mode 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_conveyor.py -x -q "$@" > gpurun_out/r5_cv_test.txt 2>&1; rc=$?
tail -40 gpurun_out/r5_cv_test.txt
exit $rc
