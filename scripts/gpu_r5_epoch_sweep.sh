# round 5: epoch length sweep on this box (scripts/pipe_bench.py: Connect4 4096 x 800, bench flags; Q = simulations per slot and epoch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/r5_epoch_sweep.txt
for q in 256 512 1024 128; do
  e=$((20480 / q))
  echo "== sims_per_epoch = $q x 4096 slots, $e epochs per block" >> gpurun_out/r5_epoch_sweep.txt
  CACHE=128000000 Q=$q E=$e BLOCKS=5 PRE=1.0 timeout -k 10 240 python scripts/pipe_bench.py 2>&1 | grep -E "block|Error|error" | cut -c1-200 >> gpurun_out/r5_epoch_sweep.txt || exit 1
done
cat gpurun_out/r5_epoch_sweep.txt
