# round 6: the spatial tile's heads (policy softmax on all waves with one exponential per logit, pooling reads issued together):
# leaf-net parity tests, then a same-box A/B of BASELINE configs[2] (Tawlbwrdd 2048 x 400, PUCT then Gumbel) - $BASE against the current build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rc=0

[ $rc -ne 0 ] && exit $rc
out=gpurun_out/r6_tbw_heads_ab.txt; : > $out
for v in base new base new; do
  if [ $v = base ]; then export AZMI_LIB=$GRAFT_REPO_ROOT/$BASE; else unset AZMI_LIB; fi
  for g in "--steps 13" "--steps 17 --gumbel"; do
  timeout -k 10 300 python bench.py --worker --game tawlbwrdd --warmup 1 --no-secondary --preroll-factor 1.0 --no-cpu-baseline $g 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v $g', 'games/s %.1f' % d['value'], 'tree_ms %.4f net_ms %.4f' % (d['config']['tree_kernel_ms'], d['config']['net_ms']), 'sims/s %.2f M' % (d['config']['sims_per_s'] / 1e6))" >> $out || exit 1
  done
done
cat $out
