run() { echo "== $*"; env "$@" BLOCKS=4 PRE=1.5 Q=256 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | grep block | awk '{g+=$3; s+=$5; e+=$7; h+=$10; n++} END {printf "avg %.0f games/s %.2f Msims/s %.2f Mevals/s hit %.3f\n", g/n, s/n, e/n, h/n}'; }
run S=1024 E=400 &&
run S=2048 E=200 &&
run S=4096 E=100 &&
run S=8192 E=50 &&
run S=16384 E=25
