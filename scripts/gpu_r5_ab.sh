# round 5: regression tests of the pipeline, then the CU-split A/B on this box (scripts/pipe_bench.py: Connect4 4096 x 800, bench flags)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_conveyor.py tests/test_gpu_selfplay_harness.py -x -q > gpurun_out/r5_tests_pipeline.txt 2>&1; rc=$?
tail -5 gpurun_out/r5_tests_pipeline.txt
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/r5_cu_split_ab.txt
for split in 0 64 72 80 96; do
  echo "== AZMI_PIPE_CU_SPLIT=$split" >> gpurun_out/r5_cu_split_ab.txt
  AZMI_PIPE_CU_SPLIT=$split CACHE=128000000 Q=256 E=80 BLOCKS=5 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|preroll|Error|error" | cut -c1-220 >> gpurun_out/r5_cu_split_ab.txt || exit 1
done
cat gpurun_out/r5_cu_split_ab.txt
