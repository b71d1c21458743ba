# one-knob sweeps of the pipeline on one box (ENVNAME, VALUES): base first and last
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${OUT:-r6_knob_sweep.txt}; : > $out
for v in base $VALUES base; do
  if [ $v = base ]; then unset $ENVNAME; else export $ENVNAME=$v; fi
  echo "== $ENVNAME $v" >> $out
  CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block [123]|rror" | cut -c1-130 >> $out
done
cat $out
