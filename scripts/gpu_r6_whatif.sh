# what-if: the pipeline's rate with a shorter net (AZMI_WHATIF_NET_DEPTH residual blocks instead of 6): how much of it is tile latency
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_whatif_net_depth.txt; : > $out
for d in 6 4 2 1 6; do
  export AZMI_WHATIF_NET_DEPTH=$d
  echo "== net depth $d" >> $out
  AZMI_PIPE_PROF= CACHE=128000000 Q=256 E=80 BLOCKS=4 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|rror" | cut -c1-150 >> $out
done
cat $out
