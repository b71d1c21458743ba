cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_tree_wgs_sweep.txt; : > $out
for t in auto 128 136 152 160 auto; do
  if [ $t = auto ]; then unset AZMI_PIPE_TREE_WGS; else export AZMI_PIPE_TREE_WGS=$t; fi
  echo "== tree workgroups $t" >> $out
  CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|rror" | cut -c1-150 >> $out
done
cat $out
