# round 5: a tree-kernel change against the build before it (alphazero-pybind11_amd/libazmi_base.so, AZMI_LIB) on ONE box:
# pipeline regression tests with the new build, then tree-only rate and the 4096 x 800 pipeline rate, base / new alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_tree_ab.txt; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q > gpurun_out/r5_tests_pipeline.txt 2>&1; rc=$?
tail -3 gpurun_out/r5_tests_pipeline.txt
[ $rc -ne 0 ] && exit $rc
base=$GRAFT_REPO_ROOT/alphazero-pybind11_amd/libazmi_base.so
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export AZMI_LIB=$base; else unset AZMI_LIB; fi
    echo "== $which (rep $rep): tree kernel alone" >> $out
    E=20 BLOCKS=3 timeout -k 10 120 python scripts/pipe_tree_only.py 2>&1 | grep -E "block|rror" | cut -c1-200 >> $out || exit 1
    echo "== $which (rep $rep): pipeline 4096 x 800" >> $out
    CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|rror" | cut -c1-220 >> $out || exit 1
  done
done
unset AZMI_LIB
echo "== new, the conveyor as the net side (AZMI_PIPE_NET=conveyor)" >> $out
AZMI_PIPE_NET=conveyor CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|rror" | cut -c1-220 >> $out
cat $out
