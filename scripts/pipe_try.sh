timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -x -q -k "pipeline" 2>&1 | tail -2
run() { echo "== $*"; env "$@" BLOCKS=6 PRE=3 Q=256 E=100 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | grep block | awk '{g+=$3; s+=$5; e+=$7; h+=$10; n++} END {printf "avg %.0f games/s %.2f Msims/s %.2f Mevals/s hit %.3f\n", g/n, s/n, e/n, h/n}'; }
run CACHE=128000000 &&
run CACHE=128000000 Q=384 E=70 &&
run CACHE=128000000 Q=160 E=160 &&
run CACHE=32000000
