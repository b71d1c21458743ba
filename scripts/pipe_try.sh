timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
for t in 2 0; do
AZMI_PIPE_TILE=$t timeout -k 10 500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench_pipe_$t.json 2> gpurun_out/bench_pipe.err; tail -2 gpurun_out/bench_pipe.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_pipe_$t.json').read().strip().splitlines()[-1])
c=d['config']; r=d['roofline']
print("TILE $t", d['value'], d['ms_per_step'], c['sims_per_s'], c['leaf_evals_per_s'], c['cache_hit_rate'], c['tree_kernel_ms'], c['net_ms'], r['achieved'], r['frac'])
PY
done
