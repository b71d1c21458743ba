run() { echo "== $*"; env "$@" AZMI_PIPE_PROF=1 BLOCKS=2 PRE=2 E=300 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -3; }
run AZMI_PIPE_TREE_WGS=96
run AZMI_PIPE_TREE_WGS=128
run AZMI_PIPE_TREE_WGS=128 MAXI=2
