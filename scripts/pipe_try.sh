run() { echo "== $*"; env "$@" BLOCKS=4 PRE=2 E=100 Q=256 AZMI_PIPE_PROF=1 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -4; }
timeout -k 10 280 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2 &&
run X=1
