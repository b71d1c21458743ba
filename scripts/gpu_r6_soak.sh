cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sh in net_random gumbel_two_nets slots_16384; do
  SHAPE=$sh N=$( [ $sh = slots_16384 ] && echo ${NBIG:-3000} || echo ${N:-20000} ) timeout -k 10 500 python scripts/soak_pipeline.py > gpurun_out/r6_soak_$sh.txt 2>&1; echo "rc $?" >> gpurun_out/r6_soak_$sh.txt
  tail -3 gpurun_out/r6_soak_$sh.txt
done
