timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
BLOCKS=1 PRE=0.5 E=100 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -4
