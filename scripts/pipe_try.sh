run() { echo "== $*"; env "$@" BLOCKS=6 PRE=2 timeout -k 10 280 python scripts/pipe_bench.py 2>&1 | grep -v amdgpu.ids | grep block | awk '{g+=$3; s+=$5; n++} END {printf "avg %.0f games/s %.2f Msims/s over %d blocks\n", g/n, s/n, n}'; }
run Q=192 E=130 &&
run Q=256 E=100 &&
run Q=320 E=80 &&
run Q=256 E=100 AZMI_PIPE_MOVERS=4 &&
run Q=256 E=100 AZMI_PIPE_IDLE_FRAC=0.03
