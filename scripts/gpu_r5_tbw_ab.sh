# round 5: the 11 x 11 spatial tile with ONE board per workgroup (build -DAZMI_SP11_TBW=1 -> libazmi_tbw1.so) against the 2-board tile
# on one box: net parity tests with the variant, then the Tawlbwrdd bench worker, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5_tbw_ab.txt; : > $out
var=$GRAFT_REPO_ROOT/alphazero-pybind11_amd/${VARIANT:-libazmi_tbw1.so}
AZMI_LIB=$var timeout -k 10 600 python -m pytest tests/test_gpu_leafnet.py -x -q > gpurun_out/r5_tests_leafnet_tbw1.txt 2>&1; rc=$?
tail -3 gpurun_out/r5_tests_leafnet_tbw1.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for which in base variant; do
    if [ $which = variant ]; then export AZMI_LIB=$var; else unset AZMI_LIB; fi
    echo "== $which (rep $rep)" >> $out
    timeout -k 10 300 python bench.py --worker --game tawlbwrdd --warmup 1 --no-secondary --preroll-factor 1.0 --no-cpu-baseline --steps 13 2> gpurun_out/r5_tbw_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print('games/s %.1f  ms/round %.4f  tree %.4f net %.4f  evals/s %.0f' % (d['value'], c['ms_per_round'], c['tree_kernel_ms'], c['net_ms'], c['leaf_evals_per_s']))
" >> $out || { tail -5 gpurun_out/r5_tbw_err.txt; exit 1; }
  done
done
cat $out
