timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" CACHE=128000000 BLOCKS=6 PRE=3 Q=256 E=100 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | grep "block\|rror" | awk '/rror/{print} /block/{g+=$3; s+=$5; e+=$7; h+=$10; n++} END {if(n) printf "avg %.0f games/s %.2f Msims/s %.2f Mevals/s hit %.3f\n", g/n, s/n, e/n, h/n}'; }
run X=0 &&
run AZMI_PIPE_TREE_BLOCK=512 AZMI_PIPE_TREE_WGS=48 &&
run AZMI_PIPE_TREE_BLOCK=512 AZMI_PIPE_TREE_WGS=64
