# StarGambit (configs[4] per GPU): tests, then the bench worker with the round split off / on (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_stargambit.py -x -q 2>&1 | tail -3 || exit 1
for mode in nosplit split nosplit split; do
  if [ $mode = nosplit ]; then export AZMI_NO_BIG_SPLIT=1; else unset AZMI_NO_BIG_SPLIT; fi
  timeout -k 10 400 python bench.py --worker --game stargambit --warmup 1 --no-secondary --preroll-factor 0.5 --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$mode', 'games/s %.2f' % d['value'], 'sims/s %.0f' % d['config']['sims_per_s'], 'tree_ms %.3f net_ms %.3f' % (d['config']['tree_kernel_ms'], d['config']['net_ms']), 'mfma agg %.3f' % d['roofline']['aggregate_frac'])
" >> gpurun_out/r6_sg_split_ab.txt
done
cat gpurun_out/r6_sg_split_ab.txt
