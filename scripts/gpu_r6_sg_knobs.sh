# StarGambit (configs[4] per GPU): the round's knobs on one box - hipGraph replay of the rounds, inline simulations per round
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r6_sg_knobs.txt; : > $out
run() { timeout -k 10 300 python bench.py --worker --game stargambit --warmup 1 --no-secondary --preroll-factor 0.5 --no-cpu-baseline --steps 100 $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'games/s %.2f' % d['value'], 'sims/s %.0f' % d['config']['sims_per_s'], 'tree_ms %.3f net_ms %.3f hit %.3f ms/round %.4f' % (d['config']['tree_kernel_ms'], d['config']['net_ms'], d['config']['cache_hit_rate'], d['config']['ms_per_round']))" >> $out; }
run default ""
AZMI_GRAPH=1 run graph ""
run inline1 "--inline 1"
run inline2 "--inline 2"
run inline16 "--inline 16"
run engines8 "--engines 8"
run engines2 "--engines 2"
run default ""
cat $out
