# round 6 same-box A/B: tests first ($TESTS), then the 4096 x 800 pipeline rate of $BASE against the current build, alternating, REPS times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r6_ab.txt}; : > $out
if [ -n "$TESTS" ]; then
  timeout -k 10 900 python -m pytest $TESTS -x -q > gpurun_out/r6_ab_tests.txt 2>&1; rc=$?
  tail -3 gpurun_out/r6_ab_tests.txt
  [ $rc -ne 0 ] && exit $rc
fi
for rep in $(seq 1 ${REPS:-2}); do
  for which in base new; do
    if [ $which = base ]; then export AZMI_LIB=$GRAFT_REPO_ROOT/$BASE; else unset AZMI_LIB; fi
    echo "== $which (rep $rep)" >> $out
    CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|rror" | cut -c1-170 >> $out || exit 1
  done
done
cat $out
