run() { echo "== $*"; env "$@" BLOCKS=3 PRE=2 E=300 Q=96 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -3; }
run AZMI_PIPE_SMALL_THR=0 &&
run AZMI_PIPE_SMALL_THR=60 &&
run AZMI_PIPE_SMALL_THR=180 &&
run AZMI_PIPE_SMALL_THR=480
