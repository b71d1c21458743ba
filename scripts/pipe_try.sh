run() { echo "== $*"; env "$@" BLOCKS=3 PRE=2 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -2; }
run Q=512 E=150 &&
run Q=1024 E=75 &&
run Q=256 E=300 AZMI_PIPE_MOVERS=16 &&
run Q=256 E=300 AZMI_PIPE_MOVERS=4
