# same-box A/B of the 4096 x 800 pipeline rate: the build in $BASE (a libazmi.so of the same ABI, AZMI_LIB) against the current one,
# alternating, REPS times; the pipeline tests first.  usage: BASE=alphazero-pybind11_amd/libazmi_base2.so REPS=2 bash scripts/gpu_ab_lib.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_ab_lib.txt; : > $out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r5_tests_pipeline.txt 2>&1; rc=$?
  tail -3 gpurun_out/r5_tests_pipeline.txt
  [ $rc -ne 0 ] && exit $rc
fi
for rep in $(seq 1 ${REPS:-2}); do
  for which in base new; do
    if [ $which = base ]; then export AZMI_LIB=$GRAFT_REPO_ROOT/$BASE; else unset AZMI_LIB; fi
    echo "== $which (rep $rep)" >> $out
    CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|rror" | cut -c1-200 >> $out || exit 1
  done
done
cat $out
