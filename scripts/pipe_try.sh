run() { echo "== $*"; env "$@" AZMI_PIPE_PROF=1 BLOCKS=2 PRE=2 E=300 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | grep block; }
run AZMI_PIPE_TREE_WGS=64 AZMI_PIPE_TILE=0
run AZMI_PIPE_TREE_WGS=96 AZMI_PIPE_TILE=0
run AZMI_PIPE_TREE_WGS=128 AZMI_PIPE_TILE=0
run AZMI_PIPE_TREE_WGS=96 AZMI_PIPE_TILE=2
