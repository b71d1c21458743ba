run() { echo "== $*"; env "$@" BLOCKS=4 PRE=2 E=300 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -4; }
run Q=64 &&
run Q=80 &&
run Q=96 &&
run Q=128
