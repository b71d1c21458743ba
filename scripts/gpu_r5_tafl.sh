# round 5: the wide-game mover split (k_round_big_sim + k_round_big_move): parity tests of the Tafl family, then a same-box A/B of
# BASELINE configs[2] (Tawlbwrdd 2048 x 400) with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity_tawlbwrdd.py tests/test_gpu_tafl_family.py tests/test_gpu_gumbel.py tests/test_gpu_full_size.py tests/test_gpu_t3_nn_in_the_loop.py tests/test_gpu_groups_perms.py -x -q > gpurun_out/r5_tests_tafl.txt 2>&1; rc=$?
tail -5 gpurun_out/r5_tests_tafl.txt
[ $rc -ne 0 ] && exit $rc
for v in split nosplit split nosplit; do
  if [ $v = nosplit ]; then export AZMI_NO_BIG_SPLIT=1; else unset AZMI_NO_BIG_SPLIT; fi
  timeout -k 10 300 python bench.py --worker --game tawlbwrdd --warmup 1 --no-secondary --preroll-factor 1.0 --no-cpu-baseline --steps 13 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', 'games/s %.1f' % d['value'], 'tree_ms %.4f net_ms %.4f' % (d['config']['tree_kernel_ms'], d['config']['net_ms']), 'sims/s %.2f M' % (d['config']['sims_per_s'] / 1e6))" >> gpurun_out/r5_tafl_split_ab.txt || exit 1
done
cat gpurun_out/r5_tafl_split_ab.txt
